set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2c3
mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_READ_WAVEFRONTS TA_TOTAL_WAVEFRONTS GRBM_GUI_ACTIVE GRBM_TA_BUSY" "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_TOTAL_READ TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES" "TCC_REQ TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_TAG_STALL TCC_BUSY TCC_CYCLE" "MeanOccupancyPerCU MemUnitStalled SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "k_lookup_cand|k_walk" --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/p$i.json 2> $O/p$i.err || echo "pass $i failed"
done
echo done
