#!/bin/bash
# Collect the judged measurement artifacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/final/bench_256.json                         the bench.py line (with the 512-px secondary block and cpu_baseline)
#   gpurun_out/final/stats/                                  rocprofv3 --kernel-trace --stats of the default bench command
#   gpurun_out/final/pmc_fetch/, pmc_write/                  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes
# Copy the summaries into profiles/ afterwards (tools/pmc_hbm_summary.py gpurun_out/final profiles/r02_pmc_hbm_traffic.json).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_256.json 2> $O/bench_256.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o ${ROUND:-r02} -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 1 --warmup 1 > /dev/null 2> $O/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --no-secondary --steps 1 --warmup 1 > /dev/null 2> $O/pmc_write.err || exit 1
ls -R $O | head -40
