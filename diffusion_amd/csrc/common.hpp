// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define DEVINL __device__ __forceinline__

// status codes returned by every C-ABI entry point
#define DA_OK 0
#define DA_ERR_SHAPE 1
#define DA_ERR_LAUNCH 2

// hipGetLastError() is per-thread sticky-until-read: drop whatever an unrelated earlier runtime call left behind
#define DA_CLEAR_ERR() (void)hipGetLastError()

#define DA_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return DA_ERR_LAUNCH;            \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: remember it per device ordinal, not per process, so a
// process that drives several GPUs raises the limit on each of them (one bit per device in *done_mask).
static inline int da_ensure_dyn_smem(const void* fn, int bytes, unsigned long long* done_mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return DA_ERR_LAUNCH;
  if (*done_mask & (1ull << dev)) return DA_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return DA_ERR_LAUNCH;
  *done_mask |= 1ull << dev;
  return DA_OK;
}

DEVINL float bf2f(bf16 x) { return (float)x; }
DEVINL bf16 f2bf(float x) { return (bf16)x; }

DEVINL bf16x8 ld8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
DEVINL void st8(bf16* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }
// 16-byte store with the non-temporal hint (global_store_dwordx4 ... nt): output rows a kernel writes once and never reads
DEVINL void st8_nt(bf16* p, bf16x8 v) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4s;
  __builtin_nontemporal_store(__builtin_bit_cast(u32x4s, v), reinterpret_cast<u32x4s*>(p));
}
DEVINL bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
  return z;
}

// Exchange with lane ^ 32 through v_permlane32_swap (gfx950): a VALU op instead of the ds_bpermute round trip behind
// __shfl_xor(v, 32).  Returns the partner's value in every lane.
DEVINL float xor32(float v) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const unsigned x = __builtin_bit_cast(unsigned, v);
  const u32x2 r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  // r[0]: lanes 0-31 keep their value, lanes 32-63 receive lanes 0-31; r[1]: lanes 0-31 receive lanes 32-63
  return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

// value of lane ^ O for O = 8 (DPP row_ror:8 inside a 16-lane row), 16 (v_permlane16_swap), 32 (v_permlane32_swap)
template <int O>
DEVINL float lane_xor(float v) {
  static_assert(O == 8 || O == 16 || O == 32, "lane_xor");
  if constexpr (O == 8) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  } else if constexpr (O == 16) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const unsigned x = __builtin_bit_cast(unsigned, v);
    const u32x2 r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 16) ? r[0] : r[1]);
  } else {
    return xor32(v);
  }
}

DEVINL float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
DEVINL float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// sigmoid through v_exp_f32 + v_rcp_f32 (1 ulp each): the IEEE division behind `1.0f / (...)` is ~10 VALU instructions, and
// the GroupNorm-backward statistics pass (chan_reduce<1>: one silu' per element read) was bound by VALU issue, not HBM
DEVINL float sigmoid_f(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
DEVINL float silu_f(float x) { return x * sigmoid_f(x); }
// d/dx silu(x) = s + x*s*(1-s), s = sigmoid(x)
DEVINL float dsilu_f(float x) {
  const float s = sigmoid_f(x);
  return s * fmaf(x, 1.0f - s, 1.0f);
}
// erf-GELU (diffusers GEGLU uses F.gelu, approximate='none').  erf through Abramowitz-Stegun 7.1.26, branch-free:
//   erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2),  t = 1 / (1 + p z),  z >= 0          |error| <= 1.5e-7
// ~15 VALU instructions instead of the ~50 (forward) / ~90 (backward) of the library erff + expf, which made the
// GEGLU passes VALU-bound (444 us of pure VALU issue for the 32x32-level forward pass vs 406 us measured); gelu and
// its derivative stay within 5e-7 absolute of the fp64 values, far inside bf16 resolution.
// Returns Phi(x) = 0.5 (1 + erf(x / sqrt 2)) and e = exp(-x^2 / 2), which the derivative's Gaussian term reuses.
DEVINL float gelu_phi(float x, float& e) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  e = __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);
  const float er = copysignf(fmaf(-poly, e, 1.0f), x);
  return fmaf(0.5f, er, 0.5f);
}
DEVINL float gelu_f(float x) {
  float e;
  return x * gelu_phi(x, e);
}
DEVINL float dgelu_f(float x) {
  float e;
  const float cdf = gelu_phi(x, e);
  return fmaf(x * 0.39894228040143268f, e, cdf);
}

// q = (n * magic) >> 40, magic = floor(2^40/d)+1 (host computes magic).  With e = magic*d - 2^40 in (0, d] the quotient is
// exact while n * e < 2^40, i.e. for every n with n * d < 2^40 (and n < 2^24 so that n * magic fits 64 bits at d = 1):
// the launchers check both (tests/test_abi_and_host.py::test_fastdiv_exact_at_the_admitted_bounds mirrors the arithmetic)
struct FastDiv {
  unsigned long long magic;
  unsigned int d;
};
DEVINL unsigned int fdiv(unsigned int n, FastDiv f) { return (unsigned int)(((unsigned long long)n * f.magic) >> 40); }
static inline FastDiv make_fastdiv(unsigned int d) {
  FastDiv f;
  f.d = d;
  f.magic = ((1ull << 40) / d) + 1ull;
  return f;
}

// transposed LDS read (gfx950): per 16-lane group a 4-row x 16-col block of 16-bit elements is
// delivered column-major: lane i of the group gets column i, rows 0..3 in elements 0..3.
// Lane 4q+p of the group supplies the address of row q, columns 4p..4p+3 (8 bytes, 8-byte aligned).
DEVINL short4v lds_tr16_b64(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds_addr));
}

// The same read issued through inline asm, for loops that also fill LDS by DMA (global_load_lds): in front of the
// INTRINSIC the compiler's waitcnt pass assumes the pending DMA may alias and inserts s_waitcnt vmcnt(0), i.e. every
// K-step waited for the next stage it had just requested.  The asm form is invisible to that pass; the caller orders
// it with lds_wait<N>() (LDS reads return in order: N = reads that may still be outstanding), passing the registers
// it is about to consume so that their users cannot be scheduled above the wait.
DEVINL unsigned lds_offset(const void* p) {
  return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p;
}
DEVINL short4v lds_tr16_b64_asm(unsigned lds_byte_offset) {
  short4v r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_byte_offset));
  return r;
}
template <int N>
DEVINL void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <typename T>
DEVINL void reg_tie(T& r) {
  asm volatile("" : "+v"(r));
}
template <int N, typename... T>
DEVINL void lds_wait_for(T&... regs) {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  (reg_tie(regs), ...);
}
