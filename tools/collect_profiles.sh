#!/bin/bash
# Collect the judged measurement artifacts on the GPU box (run through gpurun from the repo root; ROUND=r03 names them):
#   gpurun_out/final/box_probe.txt             tools/box_probe.py: this box's streaming bandwidth and long-K conv rate
#   gpurun_out/final/bench_256.json            the bench.py line (with the 512-px secondary block and cpu_baseline)
#   gpurun_out/final/per_shape_in_situ.txt     BENCH_SHAPES=1: time / TFLOP/s per GEMM / attention shape inside the step,
#                                              DEFAULT options, the shipped library
#   gpurun_out/final/stats/                    rocprofv3 --kernel-trace --stats of the default bench command
#   gpurun_out/final/pmc_fetch/, pmc_write/    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes
#   gpurun_out/final/bench_mb16.json           the reference YAML's device_train_microbatch_size: 16 (SD-2-base-256.yaml:87)
#   gpurun_out/final/bench_cfg3.json           BASELINE cfg 3: online VAE + text encode, batch 64 (and _b256: batch 256 = 2048 / 8 GPUs)
#   gpurun_out/final/bench_768v.json           BASELINE cfg 5: SD-2.1-768-v, latents 4x96x96, batch 16
# Copy the summaries into profiles/ afterwards (tools/pmc_hbm_summary.py gpurun_out/final profiles/${ROUND}_pmc_hbm_traffic.json).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
ROUND=${ROUND:-r04}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# which kind of box is this?  (the pool's MI355X differ by several per cent; PROBE_STRICT=1 stops on a slow one)
PROBE_MIN_TFS=${PROBE_MIN_TFS:-1300} python3 $R/tools/box_probe.py > $O/box_probe.txt 2>&1 || { cat $O/box_probe.txt; [ -z "$PROBE_STRICT" ] || exit 3; }
cat $O/box_probe.txt
BENCH_SHAPES=1 python3 $R/bench.py > $O/bench_256.json 2> $O/per_shape_in_situ.txt || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $ROUND -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
python3 $R/tools/trace_step_table.py $(ls $O/stats/*/*kernel_trace.csv $O/stats/*kernel_trace.csv 2>/dev/null | head -1) > $O/bench_kernel_steps.txt || exit 1
echo "stats done"
# the 512-px half of the metric (latents 4x64x64, batch 64): the same trace + per-step table
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats512 -o $ROUND -- python3 $R/bench.py --latent 64 --no-cpu-baseline --no-secondary > $O/bench_512_under_rocprof.json 2> $O/stats512.err || exit 1
python3 $R/tools/trace_step_table.py $(ls $O/stats512/*/*kernel_trace.csv $O/stats512/*kernel_trace.csv 2>/dev/null | head -1) > $O/bench_kernel_steps_512.txt || exit 1
echo "stats 512 done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 1 --warmup 1 > /dev/null 2> $O/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 1 --warmup 1 > /dev/null 2> $O/pmc_write.err || exit 1
echo "pmc done"
python3 $R/bench.py --microbatch 16 --no-cpu-baseline --no-kernel-timing --no-secondary --steps 3 --warmup 1 > $O/bench_mb16.json 2> $O/bench_mb16.err || exit 1
python3 $R/bench.py --full-pipeline --batch 64 --microbatch 64 --no-cpu-baseline --no-kernel-timing --no-secondary --steps 5 --warmup 2 > $O/bench_cfg3.json 2> $O/bench_cfg3.err || exit 1
python3 $R/bench.py --full-pipeline --batch 256 --microbatch 256 --no-cpu-baseline --no-kernel-timing --no-secondary --steps 3 --warmup 1 > $O/bench_cfg3_b256.json 2> $O/bench_cfg3_b256.err || exit 1
python3 $R/bench.py --latent 96 --no-cpu-baseline --no-kernel-timing --no-secondary --steps 5 --warmup 2 > $O/bench_768v.json 2> $O/bench_768v.err || exit 1
echo "extras done"
ls -R $O | head -40
