#!/bin/bash
# Build an alternate libdiffusion_amd (for tools/lib_ab.py) from the tree with ONE source replaced:
#   tools/build_alt.sh <name> <file.hip to use instead of csrc/<same basename>> [extra hipcc flags]
# -> tools/_ab/<name>.so   (objects of the other sources are taken from diffusion_amd/csrc as built)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
B=$(basename "$SRC" .hip)
mkdir -p $R/tools/_ab/obj
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I$R/include -I$R/diffusion_amd/csrc -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" -c "$SRC" -o $R/tools/_ab/obj/$NAME.$B.o
OBJS=""
for f in gemm_nt gemm_nt_v2 gemm_nt_v3 gemm_nt_ws gemm_tn gemm_tn_v2 attention norms pointwise; do
  if [ "$f" == "$B" ]; then OBJS="$OBJS $R/tools/_ab/obj/$NAME.$B.o"; else OBJS="$OBJS $R/diffusion_amd/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/tools/_ab/$NAME.so
echo built $R/tools/_ab/$NAME.so
