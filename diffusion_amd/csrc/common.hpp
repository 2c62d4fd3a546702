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

DEVINL float bf2f(bf16 x) { return (float)x; }
DEVINL bf16 f2bf(float x) { return (bf16)x; }

DEVINL bf16x8 ld8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
DEVINL void st8(bf16* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }
DEVINL bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
  return z;
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

DEVINL float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// d/dx silu(x) = s + x*s*(1-s), s = sigmoid(x)
DEVINL float dsilu_f(float x) {
  float s = 1.0f / (1.0f + __expf(-x));
  return s * (1.0f + x * (1.0f - s));
}
DEVINL float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
DEVINL float dgelu_f(float x) {
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// exact n / d for n < 2^24, d >= 1:  q = (n * magic) >> 40, magic = floor(2^40/d)+1  (host computes magic)
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
