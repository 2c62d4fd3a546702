"""ctypes binding of libdiffusion_amd.so (C ABI declared in include/diffusion_amd.h).

There is NO fallback: if the shared library is missing or an entry point returns non-zero, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DA_LIB_ALT=<file name>: another build of the library next to the shipped one (whole-program A/B of compile-time variants)
LIB_PATH = os.path.join(_HERE, os.environ.get('DA_LIB_ALT') or 'libdiffusion_amd.so')

_vp, _l, _i, _f, _fp, _ll = C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_void_p, C.c_void_p

# name -> argtypes (must mirror include/diffusion_amd.h exactly)
SIGNATURES = {
    'da_gemm_nt': [_vp, _l, _vp, _vp, _l, _fp, _vp, _l, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _fp, _l,
                   _vp],
    'da_set_option': [C.c_char_p, _i],
    'da_gemm_nt_variant_for': [_i, _i, _i, _i, _l],
    'da_gemm_nt_geglu': [_vp, _l, _vp, _vp, _l, _vp, _l, _fp, _i, _i, _i, _vp],
    'da_gemm_nt_geglu_bwd': [_vp, _l, _vp, _vp, _l, _vp, _l, _i, _i, _i, _vp],
    'da_gemm_tn_wgrad': [_vp, _l, _vp, _l, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fp, _l, _vp],
    'da_gemm_tn_variant_for': [_i, _i, _i, _i, _i, _i, _i, _i, _i],
    'da_attn_fwd': [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _fp, _i, _i, _i, _i, _f, _vp],
    'da_attn_fwd_causal': [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _fp, _i, _i, _i, _f, _vp],
    'da_attn_bwd': [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _fp, _fp, _vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _i,
                    _f, _vp],
    'da_norm_scratch_floats': [_i, _i, _i],
    'da_groupnorm_fwd': [_vp, _l, _vp, _l, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _f, _i, _vp],
    'da_groupnorm_bwd': [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i,
                         _vp],
    'da_layernorm_fwd': [_vp, _l, _vp, _l, _fp, _fp, _fp, _i, _i, _f, _vp],
    'da_layernorm_bwd': [_vp, _l, _vp, _l, _vp, _l, _vp, _l, _fp, _fp, _fp, _fp, _fp, _i, _i, _vp],
    'da_colsum_accum': [_vp, _l, _fp, _fp, _i, _i, _vp],
    'da_image_colsum': [_vp, _l, _vp, _l, _fp, _fp, _i, _i, _i, _vp],
    'da_geglu_fwd': [_vp, _l, _vp, _l, _i, _i, _vp],
    'da_geglu_bwd': [_vp, _l, _vp, _l, _vp, _l, _i, _i, _vp],
    'da_silu_fwd': [_vp, _l, _vp, _l, _i, _i, _vp],
    'da_gelu_fwd': [_vp, _l, _vp, _l, _i, _i, _vp],
    'da_silu_bwd': [_vp, _l, _vp, _l, _vp, _l, _i, _i, _vp],
    'da_add': [_vp, _l, _vp, _l, _vp, _l, _i, _i, _vp],
    'da_copy2d': [_vp, _l, _vp, _l, _i, _i, _vp],
    'da_upsample2x_fwd': [_vp, _vp, _i, _i, _i, _i, _vp],
    'da_upsample2x_bwd': [_vp, _vp, _i, _i, _i, _i, _vp],
    'da_timestep_embed': [_ll, _vp, _i, _i, _vp],
    'da_add_noise': [_fp, _fp, _ll, _fp, _fp, _vp, _fp, _i, _i, _i, _vp],
    'da_mse_loss': [_fp, _fp, _vp, _fp, _fp, _l, _f, _f, _i, _vp],
    'da_adamw': [_fp, _fp, _fp, _fp, _vp, _fp, _f, _l, _f, _f, _f, _f, _f, _i, _f, _vp],
    'da_cast_f32_bf16': [_fp, _vp, _l, _vp],
    'da_transpose_weight': [_vp, _vp, _i, _i, _i, _vp],
    'da_transpose_weights_batched': [_vp, _vp, _vp, _i, _i, _vp],
}
_RESTYPES = {'da_norm_scratch_floats': C.c_long}

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises NativeLibraryMissing when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must be imported first: it bundles its own libamdhip64 (soname libamdhip64.so.7); loading ours before it
    # would bring in /opt/rocm's copy as a SECOND HIP runtime, and torch-owned streams / pointers would be foreign to
    # the runtime our kernels are launched through (every launch then fails with hipErrorInvalidResourceHandle).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f'{LIB_PATH} not found: build it with `make -C diffusion_amd/csrc` (or __graft_entry__.build()). '
            'diffusion_amd has no CPU / PyTorch fallback for its kernels.')
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    # DA_SET_OPTIONS="key=value,key=value": da_set_option calls applied once at load (A/B runs of whole programs: bench.py,
    # the tests); an unknown key or rejected value is an error, not a silent default
    for kv in filter(None, os.environ.get('DA_SET_OPTIONS', '').split(',')):
        k, _, v = kv.partition('=')
        if lib.da_set_option(k.strip().encode(), int(v)) != 0:
            raise ValueError(f'DA_SET_OPTIONS: da_set_option({k.strip()!r}, {v}) was rejected')
    return lib


_ERR = {1: 'DA_ERR_SHAPE (arguments rejected, nothing launched)', 2: 'DA_ERR_LAUNCH (HIP error)'}


def call(name: str, *args):
    fn = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed: {_ERR.get(rc, rc)}')
