#!/bin/bash
# Whole-step A/B of da_set_option settings on ONE box: interleaved bench.py runs (images/s, ms/step per line).
#   tools/step_ab.sh "gemm_nt_ws=0" "gemm_nt_ws=1" [rounds=3]          (run inside one gpurun call)
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for v in "$A" "$B"; do
    echo -n "$v   "
    DA_SET_OPTIONS="$v" timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1
  done
done
