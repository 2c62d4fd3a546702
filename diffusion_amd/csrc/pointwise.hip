// HBM-bound pointwise / small kernels of the U-Net training step:
// GEGLU, SiLU, strided add / copy (skip-connection concat), nearest-2x upsample fwd/bwd, sinusoidal
// timestep embedding, forward-diffusion noising (NCHW fp32 -> NHWC-8 bf16), fused MSE loss + gradient,
// fused AdamW (fp32 master + moments, bf16 shadow write), weight-shadow transpose for dgrad.
// All use 16-B vector accesses along the contiguous (channel) axis and grid-stride loops.
#include "common.hpp"
#include "diffusion_amd.h"

namespace {

constexpr int PW_BLOCK = 256;
inline int pw_blocks(long n) {
  long b = (n + PW_BLOCK - 1) / PW_BLOCK;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (int)b;
}
#define GRID_STRIDE(i, n) \
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// ---- GEGLU: out = a * gelu(g), in = [a | g]
__global__ void geglu_fwd_kernel(const bf16* in, long ldi, bf16* out, long ldo, int nvec, long total) {
  GRID_STRIDE(i, total) {
    long row = i / nvec;
    int v = (int)(i - row * nvec);
    bf16x8 a = ld8(in + row * ldi + 8 * v);
    bf16x8 g = ld8(in + row * ldi + 8 * (nvec + v));
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(a[e]) * gelu_f(bf2f(g[e])));
    st8(out + row * ldo + 8 * v, o);
  }
}
__global__ void geglu_bwd_kernel(const bf16* in, long ldi, const bf16* dout, long lddo, bf16* din, long lddi,
                                 int nvec, long total) {
  GRID_STRIDE(i, total) {
    long row = i / nvec;
    int v = (int)(i - row * nvec);
    bf16x8 a = ld8(in + row * ldi + 8 * v);
    bf16x8 g = ld8(in + row * ldi + 8 * (nvec + v));
    bf16x8 d = ld8(dout + row * lddo + 8 * v);
    bf16x8 da, dg;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float gf = bf2f(g[e]), df = bf2f(d[e]);
      da[e] = f2bf(df * gelu_f(gf));
      dg[e] = f2bf(df * bf2f(a[e]) * dgelu_f(gf));
    }
    st8(din + row * lddi + 8 * v, da);
    st8(din + row * lddi + 8 * (nvec + v), dg);
  }
}

// ---- SiLU on a 2-D strided tensor
__global__ void silu_fwd_kernel(const bf16* x, long ldx, bf16* y, long ldy, int nvec, long total) {
  GRID_STRIDE(i, total) {
    long row = i / nvec;
    int v = (int)(i - row * nvec);
    bf16x8 a = ld8(x + row * ldx + 8 * v), o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(silu_f(bf2f(a[e])));
    st8(y + row * ldy + 8 * v, o);
  }
}
// ---- erf-GELU on a 2-D strided tensor (the text encoder's MLP activation; forward only - the encoder is frozen)
__global__ void gelu_fwd_kernel(const bf16* x, long ldx, bf16* y, long ldy, int nvec, long total) {
  GRID_STRIDE(i, total) {
    long row = i / nvec;
    int v = (int)(i - row * nvec);
    bf16x8 a = ld8(x + row * ldx + 8 * v), o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(gelu_f(bf2f(a[e])));
    st8(y + row * ldy + 8 * v, o);
  }
}
__global__ void silu_bwd_kernel(const bf16* x, long ldx, const bf16* dy, long lddy, bf16* dx, long lddx, int nvec,
                                long total) {
  GRID_STRIDE(i, total) {
    long row = i / nvec;
    int v = (int)(i - row * nvec);
    bf16x8 a = ld8(x + row * ldx + 8 * v), d = ld8(dy + row * lddy + 8 * v), o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(d[e]) * dsilu_f(bf2f(a[e])));
    st8(dx + row * lddx + 8 * v, o);
  }
}

// ---- strided add / copy
__global__ void add_kernel(const bf16* a, long lda, const bf16* b, long ldb, bf16* o, long ldo, int nvec,
                           long total) {
  GRID_STRIDE(i, total) {
    long row = i / nvec;
    int v = (int)(i - row * nvec);
    bf16x8 x = ld8(a + row * lda + 8 * v), y = ld8(b + row * ldb + 8 * v), r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = f2bf(bf2f(x[e]) + bf2f(y[e]));
    st8(o + row * ldo + 8 * v, r);
  }
}
__global__ void copy2d_kernel(const bf16* a, long lda, bf16* o, long ldo, int nvec, long total) {
  GRID_STRIDE(i, total) {
    long row = i / nvec;
    int v = (int)(i - row * nvec);
    st8(o + row * ldo + 8 * v, ld8(a + row * lda + 8 * v));
  }
}

// ---- nearest 2x upsample (NHWC)
__global__ void upsample2x_fwd_kernel(const bf16* x, bf16* y, int H, int W, int nvec, long total_out) {
  GRID_STRIDE(i, total_out) {
    long pix = i / nvec;
    int v = (int)(i - pix * nvec);
    int W2 = 2 * W, H2 = 2 * H;
    int ow = (int)(pix % W2);
    long t = pix / W2;
    int oh = (int)(t % H2);
    long b = t / H2;
    long src = (b * H + (oh >> 1)) * W + (ow >> 1);
    st8(y + pix * (long)(nvec * 8) + 8 * v, ld8(x + src * (long)(nvec * 8) + 8 * v));
  }
}
__global__ void upsample2x_bwd_kernel(const bf16* dy, bf16* dx, int H, int W, int nvec, long total_in) {
  GRID_STRIDE(i, total_in) {
    long pix = i / nvec;
    int v = (int)(i - pix * nvec);
    int w = (int)(pix % W);
    long t = pix / W;
    int h = (int)(t % H);
    long b = t / H;
    const long C = (long)nvec * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        long src = (b * 2 * H + 2 * h + dh) * 2 * W + 2 * w + dw;
        bf16x8 d = ld8(dy + src * C + 8 * v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += bf2f(d[e]);
      }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e]);
    st8(dx + pix * C + 8 * v, o);
  }
}

// ---- sinusoidal timestep embedding: [cos(t f_i) | sin(t f_i)], f_i = 10000^(-i/half)
__global__ void timestep_embed_kernel(const long long* t, bf16* out, int B, int dim) {
  const int half = dim >> 1;
  GRID_STRIDE(i, (long)B * half) {
    int b = (int)(i / half), k = (int)(i - (long)b * half);
    float f = expf(-9.210340371976184f * (float)k / (float)half);
    float a = (float)t[b] * f;
    out[(long)b * dim + k] = f2bf(cosf(a));
    out[(long)b * dim + half + k] = f2bf(sinf(a));
  }
}

// ---- forward diffusion: x_t = sqrt(ac[t]) x0 + sqrt(1-ac[t]) eps ; target = eps or v
// inputs NCHW fp32 [B,4,HW]; outputs NHWC with the 4 channels padded to 8 (pad = 0)
__global__ void add_noise_kernel(const float* x0, const float* eps, const long long* t, const float* sqrt_ac,
                                 const float* sqrt_1mac, bf16* xt, float* target, int HW, long total_pix,
                                 int v_pred) {
  GRID_STRIDE(i, total_pix) {
    long b = i / HW;
    int pix = (int)(i - b * HW);
    const float a = sqrt_ac[t[b]], s = sqrt_1mac[t[b]];
    bf16x8 o = zero8();
    float tg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x = x0[(b * 4 + c) * HW + pix], n = eps[(b * 4 + c) * HW + pix];
      o[c] = f2bf(a * x + s * n);
      tg[c] = v_pred ? (a * n - s * x) : n;
    }
    st8(xt + i * 8, o);
    *reinterpret_cast<f32x4*>(target + i * 8) = f32x4{tg[0], tg[1], tg[2], tg[3]};
    *reinterpret_cast<f32x4*>(target + i * 8 + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// ---- MSE loss over the 4 valid channels of NHWC-8 tensors, and its gradient
__global__ void mse_partial_kernel(const float* pred, const float* target, bf16* dpred, float* partial,
                                   long total_pix, float grad_coef) {
  __shared__ float sh[PW_BLOCK / 64];
  float acc = 0.f;
  GRID_STRIDE(i, total_pix) {
    f32x4 p = *reinterpret_cast<const f32x4*>(pred + i * 8);
    f32x4 q = *reinterpret_cast<const f32x4*>(target + i * 8);
    bf16x8 g = zero8();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float d = p[c] - q[c];
      acc += d * d;
      g[c] = f2bf(d * grad_coef);
    }
    st8(dpred + i * 8, g);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < PW_BLOCK / 64; ++w) s += sh[w];
    partial[blockIdx.x] = s;
  }
}
__global__ void mse_finalize_kernel(const float* partial, int n, float* loss, float inv_count, float weight,
                                    int accumulate) {
  __shared__ float sh[PW_BLOCK / 64];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < PW_BLOCK / 64; ++w) s += sh[w];
    s = s * inv_count * weight;
    loss[0] = accumulate ? loss[0] + s : s;
  }
}

// ---- fused AdamW (torch.optim.AdamW semantics), flat buffers; writes the bf16 compute shadow
__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, bf16* shadow, float* ema, float ema_s,
                             long n, float lr, float b1, float b2, float eps, float wd, float inv_bc1,
                             float inv_sqrt_bc2, float gscale) {
  GRID_STRIDE(i4, (n + 3) / 4) {
    long i = i4 * 4;
    if (i + 3 < n) {
      f32x4 pp = *reinterpret_cast<f32x4*>(p + i), gg = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
      bf16x4 sh;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float gr = gg[e] * gscale;
        float w = pp[e] * (1.f - lr * wd);
        mm[e] = b1 * mm[e] + (1.f - b1) * gr;
        vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
        float denom = sqrtf(vv[e]) * inv_sqrt_bc2 + eps;
        w -= lr * inv_bc1 * mm[e] / denom;
        pp[e] = w;
        sh[e] = f2bf(w);
      }
      if (ema) {  // EMA of the weights fused into the same pass (ema = s*ema + (1-s)*w_new)
        f32x4 ee = *reinterpret_cast<f32x4*>(ema + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) ee[e] = ema_s * ee[e] + (1.f - ema_s) * pp[e];
        *reinterpret_cast<f32x4*>(ema + i) = ee;
      }
      *reinterpret_cast<f32x4*>(p + i) = pp;
      *reinterpret_cast<f32x4*>(m + i) = mm;
      *reinterpret_cast<f32x4*>(v + i) = vv;
      *reinterpret_cast<bf16x4*>(shadow + i) = sh;
    } else {
      for (long j = i; j < n; ++j) {
        float gr = g[j] * gscale;
        float w = p[j] * (1.f - lr * wd);
        float mj = b1 * m[j] + (1.f - b1) * gr;
        float vj = b2 * v[j] + (1.f - b2) * gr * gr;
        w -= lr * inv_bc1 * mj / (sqrtf(vj) * inv_sqrt_bc2 + eps);
        p[j] = w; m[j] = mj; v[j] = vj; shadow[j] = f2bf(w);
        if (ema) ema[j] = ema_s * ema[j] + (1.f - ema_s) * w;
      }
    }
  }
}

__global__ void cast_f32_bf16_kernel(const float* s, bf16* d, long n) {
  GRID_STRIDE(i, n) d[i] = f2bf(s[i]);
}

// ---- dgrad weight shadow: dst[c][T-1-t][n] = src[n][t][c]   (taps flipped, channel roles swapped)
__global__ void transpose_weight_kernel(const bf16* src, bf16* dst, int N, int T, int C) {
  __shared__ bf16 tile[32][33];
  const int t = blockIdx.z;
  const int n0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int n = n0 + j, c = c0 + tx;
    tile[j][tx] = (n < N && c < C) ? src[((long)n * T + t) * C + c] : (bf16)0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, n = n0 + tx;
    if (c < C && n < N) dst[((long)c * T + (T - 1 - t)) * N + n] = tile[tx][j];
  }
}

// ---- all dgrad weight shadows in ONE launch: descriptor table {src_off, dst_off, N, T, C, first_block} per tensor
struct TransposeDesc {
  long src_off, dst_off;
  int N, T, C, first_block;
};
// 64 x 64 tiles, 16-B accesses on both sides (the 32 x 32 scalar version moved 64-B row segments and spent most of a
// 2-KiB tile's time in the descriptor search: 2.2 ms per step for 1.7 GB each way)
__global__ __launch_bounds__(256) void transpose_weights_batched_kernel(const bf16* src_base, bf16* dst_base,
                                                                        const TransposeDesc* desc, int ntensors) {
  __shared__ __attribute__((aligned(16))) bf16 tile[64][72];
  // binary search: last tensor whose first_block <= blockIdx.x
  int lo = 0, hi = ntensors - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (desc[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const TransposeDesc d = desc[lo];
  int b = blockIdx.x - d.first_block;
  const int tc = (d.C + 63) / 64, tn = (d.N + 63) / 64;
  const int t = b / (tc * tn);
  b -= t * tc * tn;
  const int n0 = (b / tc) * 64, c0 = (b % tc) * 64;
  const bf16* src = src_base + d.src_off;
  bf16* dst = dst_base + d.dst_off;
  const bool vec = !(d.C & 7) && !(d.N & 7) && !(d.src_off & 7) && !(d.dst_off & 7);
  if (vec) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int row = idx >> 3, ch = (idx & 7) * 8;
      const int n = n0 + row, c = c0 + ch;
      bf16x8 v = zero8();
      if (n < d.N && c < d.C) v = ld8(src + ((long)n * d.T + t) * d.C + c);
      *reinterpret_cast<bf16x8*>(&tile[row][ch]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int crow = idx >> 3, nch = (idx & 7) * 8;
      const int c = c0 + crow, n = n0 + nch;
      if (c < d.C && n < d.N) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = tile[nch + e][crow];
        st8(dst + ((long)c * d.T + (d.T - 1 - t)) * d.N + n, o);
      }
    }
  } else {
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
      const int row = idx >> 6, col = idx & 63;
      const int n = n0 + row, c = c0 + col;
      tile[row][col] = (n < d.N && c < d.C) ? src[((long)n * d.T + t) * d.C + c] : (bf16)0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
      const int crow = idx >> 6, ncol = idx & 63;
      const int c = c0 + crow, n = n0 + ncol;
      if (c < d.C && n < d.N) dst[((long)c * d.T + (d.T - 1 - t)) * d.N + n] = tile[ncol][crow];
    }
  }
}

}  // namespace

#define CHK8(x) if ((x) & 7) return DA_ERR_SHAPE

extern "C" int da_geglu_fwd(const void* in, long ldi, void* out, long ldo, int M, int Cout, hipStream_t s) {
  DA_CLEAR_ERR();
  if (M <= 0 || Cout <= 0) return DA_ERR_SHAPE;
  CHK8(Cout); CHK8(ldi); CHK8(ldo);
  long total = (long)M * (Cout >> 3);
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(pw_blocks(total)), dim3(PW_BLOCK), 0, s, (const bf16*)in, ldi,
                     (bf16*)out, ldo, Cout >> 3, total);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_geglu_bwd(const void* in, long ldi, const void* dout, long lddo, void* din, long lddi, int M,
                            int Cout, hipStream_t s) {
  DA_CLEAR_ERR();
  if (M <= 0 || Cout <= 0) return DA_ERR_SHAPE;
  CHK8(Cout); CHK8(ldi); CHK8(lddo); CHK8(lddi);
  long total = (long)M * (Cout >> 3);
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(pw_blocks(total)), dim3(PW_BLOCK), 0, s, (const bf16*)in, ldi,
                     (const bf16*)dout, lddo, (bf16*)din, lddi, Cout >> 3, total);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_silu_fwd(const void* x, long ldx, void* y, long ldy, int M, int C, hipStream_t s) {
  DA_CLEAR_ERR();
  if (M <= 0 || C <= 0) return DA_ERR_SHAPE;
  CHK8(C); CHK8(ldx); CHK8(ldy);
  long total = (long)M * (C >> 3);
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(pw_blocks(total)), dim3(PW_BLOCK), 0, s, (const bf16*)x, ldx, (bf16*)y,
                     ldy, C >> 3, total);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_gelu_fwd(const void* x, long ldx, void* y, long ldy, int M, int C, hipStream_t s) {
  DA_CLEAR_ERR();
  if (M <= 0 || C <= 0) return DA_ERR_SHAPE;
  CHK8(C); CHK8(ldx); CHK8(ldy);
  long total = (long)M * (C >> 3);
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(pw_blocks(total)), dim3(PW_BLOCK), 0, s, (const bf16*)x, ldx, (bf16*)y,
                     ldy, C >> 3, total);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_silu_bwd(const void* x, long ldx, const void* dy, long lddy, void* dx, long lddx, int M, int C,
                           hipStream_t s) {
  DA_CLEAR_ERR();
  if (M <= 0 || C <= 0) return DA_ERR_SHAPE;
  CHK8(C); CHK8(ldx); CHK8(lddy); CHK8(lddx);
  long total = (long)M * (C >> 3);
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(pw_blocks(total)), dim3(PW_BLOCK), 0, s, (const bf16*)x, ldx,
                     (const bf16*)dy, lddy, (bf16*)dx, lddx, C >> 3, total);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_add(const void* a, long lda, const void* b, long ldb, void* o, long ldo, int M, int C,
                      hipStream_t s) {
  DA_CLEAR_ERR();
  if (M <= 0 || C <= 0) return DA_ERR_SHAPE;
  CHK8(C); CHK8(lda); CHK8(ldb); CHK8(ldo);
  long total = (long)M * (C >> 3);
  hipLaunchKernelGGL(add_kernel, dim3(pw_blocks(total)), dim3(PW_BLOCK), 0, s, (const bf16*)a, lda, (const bf16*)b,
                     ldb, (bf16*)o, ldo, C >> 3, total);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_copy2d(const void* a, long lda, void* o, long ldo, int M, int C, hipStream_t s) {
  DA_CLEAR_ERR();
  if (M <= 0 || C <= 0) return DA_ERR_SHAPE;
  CHK8(C); CHK8(lda); CHK8(ldo);
  long total = (long)M * (C >> 3);
  hipLaunchKernelGGL(copy2d_kernel, dim3(pw_blocks(total)), dim3(PW_BLOCK), 0, s, (const bf16*)a, lda, (bf16*)o, ldo,
                     C >> 3, total);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_upsample2x_fwd(const void* x, void* y, int B, int H, int W, int C, hipStream_t s) {
  DA_CLEAR_ERR();
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return DA_ERR_SHAPE;
  CHK8(C);
  long total = (long)B * 4 * H * W * (C >> 3);
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(pw_blocks(total)), dim3(PW_BLOCK), 0, s, (const bf16*)x, (bf16*)y,
                     H, W, C >> 3, total);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_upsample2x_bwd(const void* dy, void* dx, int B, int H, int W, int C, hipStream_t s) {
  DA_CLEAR_ERR();
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return DA_ERR_SHAPE;
  CHK8(C);
  long total = (long)B * H * W * (C >> 3);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(pw_blocks(total)), dim3(PW_BLOCK), 0, s, (const bf16*)dy, (bf16*)dx,
                     H, W, C >> 3, total);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_timestep_embed(const long long* t, void* out, int B, int dim, hipStream_t s) {
  DA_CLEAR_ERR();
  if (B <= 0 || dim <= 0 || (dim & 1)) return DA_ERR_SHAPE;
  hipLaunchKernelGGL(timestep_embed_kernel, dim3(pw_blocks((long)B * dim / 2)), dim3(PW_BLOCK), 0, s, t, (bf16*)out,
                     B, dim);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_add_noise(const float* x0, const float* eps, const long long* t, const float* sqrt_ac,
                            const float* sqrt_1mac, void* xt, float* target, int B, int HW, int v_pred,
                            hipStream_t s) {
  DA_CLEAR_ERR();
  if (B <= 0 || HW <= 0) return DA_ERR_SHAPE;
  long total = (long)B * HW;
  hipLaunchKernelGGL(add_noise_kernel, dim3(pw_blocks(total)), dim3(PW_BLOCK), 0, s, x0, eps, t, sqrt_ac, sqrt_1mac,
                     (bf16*)xt, target, HW, total, v_pred);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_mse_loss(const float* pred, const float* target, void* dpred, float* loss, float* scratch,
                           long total_pix, float grad_coef, float weight, int accumulate, hipStream_t s) {
  DA_CLEAR_ERR();
  if (total_pix <= 0) return DA_ERR_SHAPE;
  int blocks = pw_blocks(total_pix);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(mse_partial_kernel, dim3(blocks), dim3(PW_BLOCK), 0, s, pred, target, (bf16*)dpred, scratch,
                     total_pix, grad_coef);
  DA_CHECK_LAUNCH();
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(PW_BLOCK), 0, s, scratch, blocks, loss,
                     1.0f / (4.0f * (float)total_pix), weight, accumulate);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_adamw(float* p, const float* g, float* m, float* v, void* shadow, float* ema, float ema_smoothing,
                        long n, float lr, float beta1, float beta2, float eps, float wd, int step, float grad_scale,
                        hipStream_t s) {
  DA_CLEAR_ERR();
  if (n <= 0 || step <= 0) return DA_ERR_SHAPE;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) || ((uintptr_t)shadow & 7))
    return DA_ERR_SHAPE;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(pw_blocks((n + 3) / 4)), dim3(PW_BLOCK), 0, s, p, g, m, v, (bf16*)shadow, ema,
                     ema_smoothing, n,
                     lr, beta1, beta2, eps, wd, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_cast_f32_bf16(const float* src, void* dst, long n, hipStream_t s) {
  DA_CLEAR_ERR();
  if (n <= 0) return DA_ERR_SHAPE;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(pw_blocks(n)), dim3(PW_BLOCK), 0, s, src, (bf16*)dst, n);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
extern "C" int da_transpose_weight(const void* src, void* dst, int N, int T, int C, hipStream_t s) {
  DA_CLEAR_ERR();
  if (N <= 0 || T <= 0 || C <= 0) return DA_ERR_SHAPE;
  hipLaunchKernelGGL(transpose_weight_kernel, dim3((C + 31) / 32, (N + 31) / 32, T), dim3(256), 0, s,
                     (const bf16*)src, (bf16*)dst, N, T, C);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_transpose_weights_batched(const void* src_base, void* dst_base, const void* desc, int ntensors,
                                            int total_blocks, hipStream_t s) {
  DA_CLEAR_ERR();
  if (ntensors <= 0 || total_blocks <= 0) return DA_ERR_SHAPE;
  hipLaunchKernelGGL(transpose_weights_batched_kernel, dim3(total_blocks), dim3(256), 0, s, (const bf16*)src_base,
                     (bf16*)dst_base, (const TransposeDesc*)desc, ntensors);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
