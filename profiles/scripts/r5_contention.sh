cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_contention.txt
: > $O
for spec in "0 0" "0.1 0.3" "0.5 0.3" "0.5 0.9" "0.9 0.9"; do
  set -- $spec
  for order in random sorted; do
    DECODE=0 DUP_RATE=$1 SKEW=$2 timeout -k 10 200 python profiles/scripts/structured_case.py 10000000 $order > /tmp/o.json 2>/dev/null || exit 1
    python3 - "$1" "$2" "$order" <<'PY' >> $O
import json, sys
d = json.load(open("/tmp/o.json"))
print("dup_rate %s skew %s %-6s: resolve %7.2f ms (sequential pass %6.2f, %8d reads), walk %6.2f, device %7.2f ms, anchors %8d, rounds %d" % (
    sys.argv[1], sys.argv[2], sys.argv[3], d["stages_ms"]["ms_resolve"], d["stages_ms"]["ms_resolve_chain"], d["resolve"]["reads_left_to_the_sequential_pass"],
    d["stages_ms"]["ms_walk"], d["stages_ms"]["ms_total"], d["anchors"], d["resolve"]["parallel_rounds"]))
PY
  done
done
cat $O
