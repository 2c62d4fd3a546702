mkdir -p gpurun_out/r4
for v in 0 1 0 1 0 1; do
  echo -n "gemm_nt_ws=$v   "
  DA_SET_OPTIONS="gemm_nt_ws=$v" timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1
done
