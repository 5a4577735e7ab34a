#!/bin/bash
# the dictionary chain on the GPU box's host CPU (no GPU involved): the whole worker (helpers + chain) and the chain alone.
# (profiles/r3_chain_ab.txt also holds the previous formulation, range-carried, built from round 2's host_rc.h: `git show 12ee041:leon_amd/csrc/host_rc.h`)
cd "$(dirname "$0")"
set -e
g++ -O2 -std=c++17 -o /tmp/ab_new ab_new.cpp -lpthread
g++ -O2 -std=c++17 -o /tmp/ab_consumer ab_consumer.cpp -lpthread
grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null || true
/tmp/ab_new 6000000 31
for h in 2 3 4 6 8; do echo "helpers $h"; LEON_CHAIN_HELPERS=$h /tmp/ab_new 6000000 31 | tail -1; done
/tmp/ab_new 3000000 63 | tail -1
/tmp/ab_consumer 4000000
