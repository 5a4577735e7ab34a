set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2c2
rocprofv3 --list-avail > $R/gpurun_out/r2c2/avail.txt 2>&1
grep -c . $R/gpurun_out/r2c2/avail.txt
echo done
