set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2v
df -h /tmp /dev/shm . > gpurun_out/r2v/df.txt 2>&1
cat gpurun_out/r2v/df.txt
free -g | head -2
AVAIL=$(df --output=avail -BG /tmp | tail -1 | tr -dc 0-9)
if [ "$AVAIL" -gt 110 ]; then N=100000000; else N=30000000; fi
echo "reads=$N" | tee -a gpurun_out/r2v/df.txt
LEON_CLI_READS=$N timeout -k 10 1100 python profiles/scripts/cli_at_scale.py > gpurun_out/r2v/cli.json 2> gpurun_out/r2v/cli.err
tail -c 2500 gpurun_out/r2v/cli.json
tail -n 3 gpurun_out/r2v/cli.err
