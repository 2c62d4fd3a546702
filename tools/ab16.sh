for v in 12 5 4 11; do echo "== nt v1 vs v$v (B=16)"; timeout -k 10 200 python tools/opt_ab.py gemm_nt_variant 1 $v 16 conv,lin 2>&1 | grep -v amdgpu; done
