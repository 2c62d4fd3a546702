#!/bin/bash
# Counter passes over ONE short program (run on the GPU box through gpurun, from the repo root):
#   bash tools/pmc.sh <tag> python3 tools/one_gemm.py tn 256 32 320 320 3
# -> gpurun_out/pmc/<tag>/<pass>/ (rocprofv3 csv) and gpurun_out/pmc/<tag>.json (tools/pmc_summary.py).
# Each pass is its own rocprofv3 run with --kernel-trace only (gpurun refuses --pmc combined with other trace domains);
# the program follows `--` directly (no env/bash hop).  FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
O=$R/gpurun_out/pmc/$TAG
mkdir -p $O
ARGS=()
for a in "$@"; do if [ -e "$R/$a" ]; then ARGS+=("$R/$a"); else ARGS+=("$a"); fi; done   # repo-relative paths -> absolute
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
FAILED=()
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pass$i -o p -- "${ARGS[@]}" > $O/pass$i.out 2> $O/pass$i.err || { echo "pass $i failed"; tail -5 $O/pass$i.err; FAILED+=($i); }
done
# the summary names the passes that did not complete; a partial counter set must not look complete (exit 1)
python3 $R/tools/pmc_summary.py $O $O.json "${FAILED[@]}"
if [ ${#FAILED[@]} -gt 0 ]; then echo "pmc.sh: failed passes: ${FAILED[*]}"; exit 1; fi
