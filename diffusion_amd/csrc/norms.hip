// GroupNorm / LayerNorm forward+backward and per-channel reductions on NHWC bf16 activations
// (SURVEY.md K5, K7 and the bias-gradient column sums).  All HBM-bound: 16-B vector accesses along the
// channel axis, fp32 statistics, wave64 shuffles / LDS for the reductions, and NO atomics - every
// reduction is a fixed-order two-stage sum (partials per pixel chunk -> one finalize block per channel slice), so
// results - the norm-affine and bias gradients included - are bitwise reproducible run to run.
//
// chan_reduce<MODE>: each thread owns 8 consecutive channels and walks pixels; per-(image, channel)
//   partial sums of two quantities are written per pixel chunk:
//     MODE 0  GroupNorm forward statistics      (x, x^2)
//     MODE 1  GroupNorm(+SiLU) backward sums    (dz, dz*xhat),  dz = dy * silu'(z) when SiLU was fused
//     MODE 2  column sum                         (x, -)           (bias gradients)
#include "common.hpp"
#include "diffusion_amd.h"

extern int g_grad_overwrite;  // gemm_tn.hip: da_set_option("grad_overwrite") - write the gradient outputs instead of adding
// da_set_option("gn_resident", n): register-resident single-pass GroupNorm where the slab fits and the launch has >= n
// workgroups (0 = never, 1 = whenever it fits); "gn_resident_form": 0 auto | 1 16-wave forms only | 2 12-wave form above 8 vectors
int g_gn_resident = 192;
int g_gn_resident_form = 0;
int g_gn_resident_min_slab = 64 * 1024;  // "gn_resident_min_slab": bytes of one tensor per workgroup below which the multi-pass form runs

namespace {

struct ChanReduceParams {
  const bf16* X;
  const bf16* DY;
  const float* mean_rstd;  // [B][G][2]
  const float* gamma;
  const float* beta;
  float* partial;  // [B][nchunks][C][2]
  long ldx, lddy;
  int HW, C, G, cpg, nchunks, ppc, pxt, silu;
};

template <int MODE>
__global__ void chan_reduce_kernel(ChanReduceParams p) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int tid = threadIdx.x;
  // channels are split over blockIdx.z in slabs of <= 512 vectors (4096 channels)
  const int vec0 = blockIdx.z * 512;
  const int nvec = min(512, (p.C >> 3) - vec0);
  const int Cb = nvec * 8;
  const int lvec = tid % nvec, pl = tid / nvec;
  const int vec = vec0 + lvec;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * p.ppc;
  const int p1 = min(p.HW, p0 + p.ppc);
  const bool active = pl < p.pxt;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  // MODE 1: per-channel folded coefficients - xhat = x*rs + nm, z = x*za + zb (two FMAs per element instead of four ops;
  // with the silu' this pass is bound by VALU issue, not by HBM)
  float rs[8], nm[8], za[8], zb[8];
  if (MODE == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int c = 8 * vec + e;
      int g = c / p.cpg;
      const float mu = p.mean_rstd[((long)b * p.G + g) * 2];
      rs[e] = p.mean_rstd[((long)b * p.G + g) * 2 + 1];
      nm[e] = -mu * rs[e];
      za[e] = rs[e] * p.gamma[c];
      zb[e] = fmaf(nm[e], p.gamma[c], p.beta[c]);
    }
  }
  if (active) {
    for (int pix = p0 + pl; pix < p1; pix += p.pxt) {
      const long row = (long)b * p.HW + pix;
      bf16x8 x = ld8(p.X + row * p.ldx + 8 * vec);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = bf2f(x[e]);
          s1[e] += v;
          s2[e] += v * v;
        }
      } else if (MODE == 1) {
        bf16x8 dy = ld8(p.DY + row * p.lddy + 8 * vec);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xf = bf2f(x[e]);
          const float xh = fmaf(xf, rs[e], nm[e]);
          float dz = bf2f(dy[e]);
          if (p.silu) dz *= dsilu_f(fmaf(xf, za[e], zb[e]));
          s1[e] += dz;
          s2[e] += dz * xh;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] += bf2f(x[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[((long)pl * Cb + 8 * lvec + e) * 2] = s1[e];
      red[((long)pl * Cb + 8 * lvec + e) * 2 + 1] = s2[e];
    }
  }
  __syncthreads();
  float* out = p.partial + (((long)b * p.nchunks + chunk) * p.C + 8 * vec0) * 2;
  for (int c = tid; c < Cb; c += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int k = 0; k < p.pxt; ++k) {
      a += red[((long)k * Cb + c) * 2];
      q += red[((long)k * Cb + c) * 2 + 1];
    }
    out[c * 2] = a;
    out[c * 2 + 1] = q;
  }
}

int launch_chan_reduce(int mode, ChanReduceParams& p, int B, hipStream_t stream) {
  const int nvec_total = p.C >> 3;
  if (nvec_total < 1) return DA_ERR_SHAPE;
  const int zsplit = (nvec_total + 511) / 512;
  // every z-slab uses the same pxt, sized for the widest slab
  const int nvec = nvec_total < 512 ? nvec_total : 512;
  int pxt = 512 / nvec;
  if (pxt > p.HW) pxt = p.HW;
  if (pxt < 1) pxt = 1;
  p.pxt = pxt;
  p.ppc = (p.HW + p.nchunks - 1) / p.nchunks;
  const size_t smem = (size_t)pxt * nvec * 8 * 2 * sizeof(float);
  // the last slab may be narrower: its threads beyond nvec_last*pxt idle via the `active` predicate
  dim3 grid(p.nchunks, B, zsplit), block(nvec * pxt);
  if (mode == 0) hipLaunchKernelGGL(chan_reduce_kernel<0>, grid, block, smem, stream, p);
  else if (mode == 1) hipLaunchKernelGGL(chan_reduce_kernel<1>, grid, block, smem, stream, p);
  else hipLaunchKernelGGL(chan_reduce_kernel<2>, grid, block, smem, stream, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

int pick_chunks(int B, int HW) {
  int n = 2048 / (B > 0 ? B : 1);
  if (n > 64) n = 64;
  int maxc = HW / 16;
  if (n > maxc) n = maxc;
  if (n < 1) n = 1;
  return n;
}

// ---- GroupNorm forward finalize: partial -> mean/rstd [B][G][2] and per-(b,c) scale/shift [B][C][2]
// grid (B, ceil(G / GN_GPB)): a block owns GN_GPB groups of one image (one block per image was 24 us per call at
// batch 64: 64 blocks, each thread walking up to 10 channels x 32 chunks serially)
constexpr int GN_GPB = 8;

DEVINL void chunk_sum2(const float* pp, long stride, int nchunks, float& a, float& q) {
  float a0 = 0.f, a1 = 0.f, q0 = 0.f, q1 = 0.f;
  int k = 0;
  for (; k + 1 < nchunks; k += 2) {  // two independent chains keep two loads in flight
    const float2 v0 = *reinterpret_cast<const float2*>(pp + (long)k * stride);
    const float2 v1 = *reinterpret_cast<const float2*>(pp + (long)(k + 1) * stride);
    a0 += v0.x; q0 += v0.y;
    a1 += v1.x; q1 += v1.y;
  }
  if (k < nchunks) {
    const float2 v0 = *reinterpret_cast<const float2*>(pp + (long)k * stride);
    a0 += v0.x; q0 += v0.y;
  }
  a = a0 + a1;
  q = q0 + q1;
}

__global__ void gn_fwd_finalize_kernel(const float* partial, const float* gamma, const float* beta, float* mean_rstd,
                                       float* scale_shift, int C, int G, int cpg, int nchunks, int HW, float eps) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [GN_GPB*cpg][2] then [GN_GPB][2]
  const int b = blockIdx.x;
  const int g0 = blockIdx.y * GN_GPB, ng = min(GN_GPB, G - g0);
  const int c0 = g0 * cpg, nc = ng * cpg;
  float* gs = sh + 2 * GN_GPB * cpg;
  for (int cl = threadIdx.x; cl < nc; cl += blockDim.x) {
    float a, q;
    chunk_sum2(partial + ((long)b * nchunks * C + c0 + cl) * 2, (long)C * 2, nchunks, a, q);
    sh[2 * cl] = a;
    sh[2 * cl + 1] = q;
  }
  __syncthreads();
  for (int gl = threadIdx.x; gl < ng; gl += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int k = 0; k < cpg; ++k) {
      a += sh[2 * (gl * cpg + k)];
      q += sh[2 * (gl * cpg + k) + 1];
    }
    const float n = (float)cpg * (float)HW;
    float mean = a / n;
    float var = fmaxf(q / n - mean * mean, 0.f);
    float rstd = rsqrtf(var + eps);
    gs[2 * gl] = mean;
    gs[2 * gl + 1] = rstd;
    mean_rstd[((long)b * G + g0 + gl) * 2] = mean;
    mean_rstd[((long)b * G + g0 + gl) * 2 + 1] = rstd;
  }
  __syncthreads();
  for (int cl = threadIdx.x; cl < nc; cl += blockDim.x) {
    const int gl = cl / cpg, c = c0 + cl;
    float sc = gs[2 * gl + 1] * gamma[c];
    scale_shift[((long)b * C + c) * 2] = sc;
    scale_shift[((long)b * C + c) * 2 + 1] = beta[c] - gs[2 * gl] * sc;
  }
}

// y = x*scale + shift (+SiLU)
__global__ void gn_apply_kernel(const bf16* X, long ldx, bf16* Y, long ldy, const float* scale_shift, int HW, int C,
                                long total_vec, int silu) {
  const int nvec = C >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec; i += (long)gridDim.x * blockDim.x) {
    long row = i / nvec;
    int vec = (int)(i - row * nvec);
    int b = (int)(row / HW);
    bf16x8 x = ld8(X + row * ldx + 8 * vec);
    const float* ss = scale_shift + ((long)b * C + 8 * vec) * 2;
    bf16x8 y;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float z = bf2f(x[e]) * ss[2 * e] + ss[2 * e + 1];
      y[e] = f2bf(silu ? silu_f(z) : z);
    }
    st8(Y + row * ldy + 8 * vec, y);
  }
}

// ---- GroupNorm backward finalize: per (b,g) coefficients (mean(dxhat), mean(dxhat*xhat)); grid as the forward one
__global__ void gn_bwd_finalize_kernel(const float* partial, const float* gamma, float* coef, int C, int G, int cpg,
                                       int nchunks, int HW) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int b = blockIdx.x;
  const int g0 = blockIdx.y * GN_GPB, ng = min(GN_GPB, G - g0);
  const int c0 = g0 * cpg, nc = ng * cpg;
  for (int cl = threadIdx.x; cl < nc; cl += blockDim.x) {
    float a, q;
    chunk_sum2(partial + ((long)b * nchunks * C + c0 + cl) * 2, (long)C * 2, nchunks, a, q);
    sh[2 * cl] = a * gamma[c0 + cl];
    sh[2 * cl + 1] = q * gamma[c0 + cl];
  }
  __syncthreads();
  for (int gl = threadIdx.x; gl < ng; gl += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int k = 0; k < cpg; ++k) {
      a += sh[2 * (gl * cpg + k)];
      q += sh[2 * (gl * cpg + k) + 1];
    }
    const float n = (float)cpg * (float)HW;
    coef[((long)b * G + g0 + gl) * 2] = a / n;
    coef[((long)b * G + g0 + gl) * 2 + 1] = q / n;
  }
}

// ---- GroupNorm apply, forward and backward, "channel-stationary": a thread keeps its 8 channels' coefficients
// in registers and walks pixels (same geometry as chan_reduce), instead of re-deriving per-element group indices
// and re-loading statistics for every element.
//   fwd:  y  = act(x*sc + sh)                      sc = rstd*gamma, sh = beta - mean*sc
//   bwd:  dx = dz*sc + x*Q + R (+ Radd)            dz = dy*silu'(x*sc+sh) ; Q = -rstd^2*c2 ; R = -rstd*c1 + mean*rstd^2*c2
struct GnApplyParams {
  const bf16* X; const bf16* DY; const bf16* Radd; bf16* OUT;
  const float* mean_rstd; const float* coef; const float* gamma; const float* beta;
  long ldx, lddy, ldr, ldo;
  int HW, C, G, cpg, nchunks, ppc, pxt, silu;
};

template <bool BWD>
__global__ void gn_apply2_kernel(GnApplyParams p) {
  const int tid = threadIdx.x;
  const int vec0 = blockIdx.z * 512;
  const int nvec = min(512, (p.C >> 3) - vec0);
  const int lvec = tid % nvec, pl = tid / nvec;
  if (pl >= p.pxt) return;
  const int vec = vec0 + lvec;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * p.ppc;
  const int p1 = min(p.HW, p0 + p.ppc);
  float sc[8], sh[8], Q[8], R[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * vec + e;
    const int g = c / p.cpg;
    const float mu = p.mean_rstd[((long)b * p.G + g) * 2], rs = p.mean_rstd[((long)b * p.G + g) * 2 + 1];
    sc[e] = rs * p.gamma[c];
    sh[e] = p.beta[c] - mu * sc[e];
    if (BWD) {
      const float c1 = p.coef[((long)b * p.G + g) * 2], c2 = p.coef[((long)b * p.G + g) * 2 + 1];
      Q[e] = -rs * rs * c2;
      R[e] = -rs * c1 + mu * rs * rs * c2;
    }
  }
  for (int pix = p0 + pl; pix < p1; pix += p.pxt) {
    const long row = (long)b * p.HW + pix;
    const bf16x8 x = ld8(p.X + row * p.ldx + 8 * vec);
    bf16x8 o;
    if (!BWD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float z = fmaf(bf2f(x[e]), sc[e], sh[e]);
        o[e] = f2bf(p.silu ? silu_f(z) : z);
      }
    } else {
      const bf16x8 dy = ld8(p.DY + row * p.lddy + 8 * vec);
      const bf16x8 ra = p.Radd ? ld8(p.Radd + row * p.ldr + 8 * vec) : zero8();
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xf = bf2f(x[e]);
        float dz = bf2f(dy[e]);
        if (p.silu) dz *= dsilu_f(fmaf(xf, sc[e], sh[e]));
        o[e] = f2bf(fmaf(dz, sc[e], fmaf(xf, Q[e], R[e])) + bf2f(ra[e]));
      }
    }
    st8(p.OUT + row * p.ldo + 8 * vec, o);
  }
}

int launch_gn_apply2(bool bwd, GnApplyParams& p, int B, hipStream_t stream) {
  const int nvec_total = p.C >> 3;
  const int zsplit = (nvec_total + 511) / 512;
  const int nvec = nvec_total < 512 ? nvec_total : 512;
  int pxt = 256 / nvec;
  if (pxt < 1) pxt = 1;
  if (pxt > p.HW) pxt = p.HW;
  p.pxt = pxt;
  // enough (image, pixel-chunk) workgroups to keep >= 8 waves per CU busy, >= 4 pixels per thread
  int nch = (4096 + B * zsplit - 1) / (B * zsplit);
  int maxc = p.HW / (4 * pxt);
  if (nch > maxc) nch = maxc;
  if (nch < 1) nch = 1;
  p.nchunks = nch;
  p.ppc = (p.HW + nch - 1) / nch;
  dim3 grid(nch, B, zsplit), block(nvec * pxt);
  if (bwd) hipLaunchKernelGGL(gn_apply2_kernel<true>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(gn_apply2_kernel<false>, grid, block, 0, stream, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

// dx = rstd * (dz*gamma - c1 - xhat*c2) (+ Radd)
__global__ void gn_bwd_apply_kernel(const bf16* X, long ldx, const bf16* DY, long lddy, const bf16* Radd, long ldr,
                                    bf16* DX, long lddx, const float* mean_rstd, const float* coef,
                                    const float* gamma, const float* beta, int HW, int C, int G, int cpg,
                                    long total_vec, int silu) {
  const int nvec = C >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_vec; i += (long)gridDim.x * blockDim.x) {
    long row = i / nvec;
    int vec = (int)(i - row * nvec);
    int b = (int)(row / HW);
    bf16x8 x = ld8(X + row * ldx + 8 * vec);
    bf16x8 dy = ld8(DY + row * lddy + 8 * vec);
    bf16x8 ra = Radd ? ld8(Radd + row * ldr + 8 * vec) : zero8();
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int c = 8 * vec + e;
      int g = c / cpg;
      const float mu = mean_rstd[((long)b * G + g) * 2], rs = mean_rstd[((long)b * G + g) * 2 + 1];
      const float c1 = coef[((long)b * G + g) * 2], c2 = coef[((long)b * G + g) * 2 + 1];
      float xh = (bf2f(x[e]) - mu) * rs;
      float dz = bf2f(dy[e]);
      if (silu) dz *= dsilu_f(xh * gamma[c] + beta[c]);
      float dx = rs * (dz * gamma[c] - c1 - xh * c2);
      o[e] = f2bf(dx + bf2f(ra[e]));
    }
    st8(DX + row * lddx + 8 * vec, o);
  }
}

// out1[c] += sum_rows partial[row][c][0] ; out2[c] += sum_rows partial[row][c][1]
// ONE stage, fixed order, no atomics (bitwise reproducible): a block owns 16 channels (128-B row segments, float2 loads)
// and spreads the rows over 64 row-lanes, each with four independent chains (eight loads in flight per thread); the
// row-lanes meet in LDS and thread (0, cx) adds them in lane order.  The earlier form sliced the rows over blockIdx.y and
// let the slices meet in fp32 atomics (order-dependent last bits in every norm / bias gradient).
constexpr int CSF_CH = 16, CSF_RL = 64;
__global__ __launch_bounds__(CSF_CH * CSF_RL) void chan_sum_finalize_kernel(const float* partial, int nrows, int C,
                                                                            float* out1, float* out2, int overwrite) {
  __shared__ float sh[CSF_RL][CSF_CH][2];
  const int cx = threadIdx.x % CSF_CH, ry = threadIdx.x / CSF_CH;
  const int c = blockIdx.x * CSF_CH + cx;
  float a = 0.f, q = 0.f;
  if (c < C) {
    const float2* pp = reinterpret_cast<const float2*>(partial) + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
    int k = ry;
    for (; k + 3 * CSF_RL < nrows; k += 4 * CSF_RL) {
      const float2 v0 = pp[(long)k * C], v1 = pp[(long)(k + CSF_RL) * C], v2 = pp[(long)(k + 2 * CSF_RL) * C],
                   v3 = pp[(long)(k + 3 * CSF_RL) * C];
      a0 += v0.x; q0 += v0.y;
      a1 += v1.x; q1 += v1.y;
      a2 += v2.x; q2 += v2.y;
      a3 += v3.x; q3 += v3.y;
    }
    for (; k < nrows; k += CSF_RL) {
      const float2 v = pp[(long)k * C];
      a0 += v.x;
      q0 += v.y;
    }
    a = (a0 + a1) + (a2 + a3);
    q = (q0 + q1) + (q2 + q3);
  }
  sh[ry][cx][0] = a;
  sh[ry][cx][1] = q;
  __syncthreads();
  // 64 row-lanes -> 8 -> 1, every level in index order
  float la = 0.f, lq = 0.f;
  if (ry < 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      la += sh[ry * 8 + j][cx][0];
      lq += sh[ry * 8 + j][cx][1];
    }
  }
  __syncthreads();
  if (ry < 8) {
    sh[ry][cx][0] = la;
    sh[ry][cx][1] = lq;
  }
  __syncthreads();
  if (ry == 0 && c < C) {
    float ta = 0.f, tq = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ta += sh[j][cx][0];
      tq += sh[j][cx][1];
    }
    if (out1) out1[c] = overwrite ? ta : out1[c] + ta;
    if (out2) out2[c] = overwrite ? tq : out2[c] + tq;
  }
}

static inline dim3 chan_sum_grid(int nrows, int C) {
  (void)nrows;
  return dim3((C + CSF_CH - 1) / CSF_CH);
}

// out[b][c] = sum_chunks partial[b][chunk][c][0] (bf16, strided) ; db[c] += sum_b out[b][c] (fp32)
// block = 16 channels x 64 image-lanes; the image-lanes meet in LDS in lane order (no atomics, reproducible)
__global__ __launch_bounds__(CSF_CH * CSF_RL) void image_colsum_finalize_kernel(const float* partial, bf16* out, long ldo,
                                                                                float* db, int B, int nchunks, int C,
                                                                                int overwrite) {
  __shared__ float sh[CSF_RL][CSF_CH];
  const int cx = threadIdx.x % CSF_CH, ry = threadIdx.x / CSF_CH;
  const int c = blockIdx.x * CSF_CH + cx;
  float tot = 0.f;
  if (c < C) {
    for (int b = ry; b < B; b += CSF_RL) {
      const float* pp = partial + ((long)b * nchunks * C + c) * 2;
      float a0 = 0.f, a1 = 0.f;
      int k = 0;
      for (; k + 1 < nchunks; k += 2) {
        a0 += pp[(long)k * C * 2];
        a1 += pp[(long)(k + 1) * C * 2];
      }
      if (k < nchunks) a0 += pp[(long)k * C * 2];
      const float a = a0 + a1;
      out[(long)b * ldo + c] = f2bf(a);
      tot += a;
    }
  }
  sh[ry][cx] = tot;
  __syncthreads();
  float lt = 0.f;
  if (ry < 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) lt += sh[ry * 8 + j][cx];
  }
  __syncthreads();
  if (ry < 8) sh[ry][cx] = lt;
  __syncthreads();
  if (ry == 0 && c < C && db) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += sh[j][cx];
    db[c] = overwrite ? t : db[c] + t;
  }
}

// ---- LayerNorm: one wave per row, row kept in registers (C <= 1536)
constexpr int LN_MAXV = 3;

__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16* X, long ldx, bf16* Y, long ldy, const float* gamma,
                                                     const float* beta, float* mean_rstd, int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int nvec = C >> 3;
  const int wpb = blockDim.x >> 6;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < M; row += gridDim.x * wpb) {
    float v[LN_MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
      int vec = lane + 64 * k;
      bf16x8 x = vec < nvec ? ld8(X + (long)row * ldx + 8 * vec) : zero8();
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[k][e] = bf2f(x[e]);
        s += v[k][e];
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
      int vec = lane + 64 * k;
      if (vec < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float d = v[k][e] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) {
      mean_rstd[2 * (long)row] = mean;
      mean_rstd[2 * (long)row + 1] = rstd;
    }
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
      int vec = lane + 64 * k;
      if (vec < nvec) {
        bf16x8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = f2bf((v[k][e] - mean) * rstd * gamma[8 * vec + e] + beta[8 * vec + e]);
        st8(Y + (long)row * ldy + 8 * vec, y);
      }
    }
  }
}

// dx = rstd*(dy*gamma - mean(dy*gamma) - xhat*mean(dy*gamma*xhat)) (+Radd); per-block partial dgamma/dbeta
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16* X, long ldx, const bf16* DY, long lddy,
                                                     const bf16* Radd, long ldr, bf16* DX, long lddx,
                                                     const float* gamma, const float* mean_rstd, float* partial,
                                                     int M, int C) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][C][2]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nvec = C >> 3;
  const int wpb = blockDim.x >> 6;
  float dg[LN_MAXV][8], db[LN_MAXV][8], ga[LN_MAXV][8];
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int vec = lane + 64 * k;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dg[k][e] = 0.f;
      db[k][e] = 0.f;
      ga[k][e] = vec < nvec ? gamma[8 * vec + e] : 0.f;
    }
  }
  for (int row = blockIdx.x * wpb + wave; row < M; row += gridDim.x * wpb) {
    const float mean = mean_rstd[2 * (long)row], rstd = mean_rstd[2 * (long)row + 1];
    float xh[LN_MAXV][8], dyv[LN_MAXV][8];
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
      int vec = lane + 64 * k;
      bool ok = vec < nvec;
      bf16x8 x = ok ? ld8(X + (long)row * ldx + 8 * vec) : zero8();
      bf16x8 dy = ok ? ld8(DY + (long)row * lddy + 8 * vec) : zero8();
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[k][e] = ok ? (bf2f(x[e]) - mean) * rstd : 0.f;
        dyv[k][e] = bf2f(dy[e]);
        float dxh = dyv[k][e] * ga[k][e];
        a += dxh;
        q += dxh * xh[k][e];
        dg[k][e] += dyv[k][e] * xh[k][e];
        db[k][e] += dyv[k][e];
      }
    }
    a = wave_sum(a) / (float)C;
    q = wave_sum(q) / (float)C;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
      int vec = lane + 64 * k;
      if (vec < nvec) {
        bf16x8 ra = Radd ? ld8(Radd + (long)row * ldr + 8 * vec) : zero8();
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = f2bf(rstd * (dyv[k][e] * ga[k][e] - a - xh[k][e] * q) + bf2f(ra[e]));
        st8(DX + (long)row * lddx + 8 * vec, o);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    int vec = lane + 64 * k;
    if (vec < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[((long)wave * C + 8 * vec + e) * 2] = dg[k][e];
        red[((long)wave * C + 8 * vec + e) * 2 + 1] = db[k][e];
      }
    }
  }
  __syncthreads();
  float* out = partial + (long)blockIdx.x * C * 2;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int w = 0; w < wpb; ++w) {
      a += red[((long)w * C + c) * 2];
      q += red[((long)w * C + c) * 2 + 1];
    }
    out[c * 2] = a;
    out[c * 2 + 1] = q;
  }
}

// ---- LayerNorm fast path for C = 40 * LPR (320 / 640 / 1280: every transformer width of the SD-2 U-Net).
// LPR lanes share a row and each lane owns 5 vectors (40 channels), so a wave covers 64 / LPR rows at once and keeps
// 5 (fwd) or 10-15 (bwd) 1-KiB loads in flight instead of one 640-B row; gamma / beta sit in LDS.  The one-row-per-
// wave kernels above measured 2.8 (fwd) and 2.0 TB/s (bwd) at C = 320 against 5.6 TB/s for a streaming add.
// sum over the LPR lanes that share a row, result in every lane.  Up to 16 lanes it is pure DPP (quad permutes, then
// row_half_mirror and row_mirror pair each lane with one from the other half) instead of ds_bpermute round trips
// through the LDS crossbar, which sat on the critical path of every row group; the 32-lane step keeps the shuffle.
template <int CTRL>
DEVINL float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int LPR>
DEVINL float group_sum(float v) {
  v += dpp_f<0xB1>(v);                     // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);                     // quad_perm [2,3,0,1]
  if (LPR >= 8) v += dpp_f<0x141>(v);      // row_half_mirror: lane i <-> 7-i of each 8
  if (LPR >= 16) v += dpp_f<0x140>(v);     // row_mirror: lane i <-> 15-i of each 16
  if (LPR >= 32) v += lane_xor<16>(v);
  return v;
}

template <int LPR>
__global__ __launch_bounds__(256) void ln_fwd5_kernel(const bf16* X, long ldx, bf16* Y, long ldy, const float* gamma,
                                                      const float* beta, float* mean_rstd, int M, float eps) {
  constexpr int C = 40 * LPR, RPW = 64 / LPR;
  __shared__ __attribute__((aligned(16))) float gb[2 * C];
  for (int c = threadIdx.x; c < C; c += 256) {
    gb[c] = gamma[c];
    gb[C + c] = beta[c];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, j = lane % LPR;
  const long ngroups = ((long)M + RPW - 1) / RPW;
  for (long rg = (long)blockIdx.x * 4 + wave; rg < ngroups; rg += (long)gridDim.x * 4) {
    const long row = rg * RPW + sub;
    const bool ok = row < M;
    const long rowc = ok ? row : M - 1;  // clamp instead of predicating: the loads stay branch-free
    bf16x8 xv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) xv[k] = ld8(X + rowc * ldx + 8 * (j + LPR * k));
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += bf2f(xv[k][e]);
    const float mean = group_sum<LPR>(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = bf2f(xv[k][e]) - mean;
        q += d * d;
      }
    const float rstd = rsqrtf(group_sum<LPR>(q) * (1.f / C) + eps);
    if (ok) {
      if (j == 0) *reinterpret_cast<float2*>(mean_rstd + 2 * row) = make_float2(mean, rstd);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int c0 = 8 * (j + LPR * k);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gb + c0), g1 = *reinterpret_cast<const f32x4*>(gb + c0 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(gb + C + c0), b1 = *reinterpret_cast<const f32x4*>(gb + C + c0 + 4);
        bf16x8 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = f2bf((bf2f(xv[k][e]) - mean) * rstd * g0[e] + b0[e]);
          y[e + 4] = f2bf((bf2f(xv[k][e + 4]) - mean) * rstd * g1[e] + b1[e]);
        }
        st8(Y + row * ldy + c0, y);
      }
    }
  }
}

template <int LPR, bool HAS_R>
__global__ __launch_bounds__(256) void ln_bwd5_kernel(const bf16* X, long ldx, const bf16* DY, long lddy,
                                                      const bf16* Radd, long ldr, bf16* DX, long lddx,
                                                      const float* gamma, const float* mean_rstd, float* partial,
                                                      int M) {
  constexpr int C = 40 * LPR, RPW = 64 / LPR;
  __shared__ __attribute__((aligned(16))) float gsm[C];
  __shared__ __attribute__((aligned(16))) float red[4 * C * 2];  // [wave][C][2] partials
  for (int c = threadIdx.x; c < C; c += 256) gsm[c] = gamma[c];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, j = lane % LPR;
  float dg[5][8], db[5][8];
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dg[k][e] = 0.f;
      db[k][e] = 0.f;
    }
  const long ngroups = ((long)M + RPW - 1) / RPW;
  for (long rg = (long)blockIdx.x * 4 + wave; rg < ngroups; rg += (long)gridDim.x * 4) {
    const long row = rg * RPW + sub;
    const bool ok = row < M;
    const long rowc = ok ? row : M - 1;  // clamped, branch-free loads; a padded slot contributes with dy = 0
    const float okf = ok ? 1.f : 0.f;
    bf16x8 xv[5], dyv[5], rv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int c0 = 8 * (j + LPR * k);
      xv[k] = ld8(X + rowc * ldx + c0);
      dyv[k] = ld8(DY + rowc * lddy + c0);
      if (HAS_R) rv[k] = ld8(Radd + rowc * ldr + c0);
    }
    const float2 ms = *reinterpret_cast<const float2*>(mean_rstd + 2 * rowc);
    const float mean = ms.x, rstd = ms.y;
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int c0 = 8 * (j + LPR * k);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gsm + c0), g1 = *reinterpret_cast<const f32x4*>(gsm + c0 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (bf2f(xv[k][e]) - mean) * rstd;
        const float d = bf2f(dyv[k][e]) * okf;
        const float dxh = d * (e < 4 ? g0[e & 3] : g1[e & 3]);
        a += dxh;
        q += dxh * xh;
        dg[k][e] += d * xh;
        db[k][e] += d;
      }
    }
    a = group_sum<LPR>(a) * (1.f / C);
    q = group_sum<LPR>(q) * (1.f / C);
    if (ok) {
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const int c0 = 8 * (j + LPR * k);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gsm + c0), g1 = *reinterpret_cast<const f32x4*>(gsm + c0 + 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (bf2f(xv[k][e]) - mean) * rstd;
          float v = rstd * (bf2f(dyv[k][e]) * (e < 4 ? g0[e & 3] : g1[e & 3]) - a - xh * q);
          if (HAS_R) v += bf2f(rv[k][e]);
          o[e] = f2bf(v);
        }
        st8(DX + row * lddx + c0, o);
      }
    }
  }
  // fold the RPW row slots of the wave (lanes with equal j), then the 4 waves through LDS
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if constexpr (LPR <= 8) {
        dg[k][e] += lane_xor<8>(dg[k][e]);
        db[k][e] += lane_xor<8>(db[k][e]);
      }
      if constexpr (LPR <= 16) {
        dg[k][e] += lane_xor<16>(dg[k][e]);
        db[k][e] += lane_xor<16>(db[k][e]);
      }
      dg[k][e] += lane_xor<32>(dg[k][e]);
      db[k][e] += lane_xor<32>(db[k][e]);
      if (sub == 0) {
        red[(wave * C + 8 * (j + LPR * k) + e) * 2] = dg[k][e];
        red[(wave * C + 8 * (j + LPR * k) + e) * 2 + 1] = db[k][e];
      }
    }
  __syncthreads();
  float* out = partial + (long)blockIdx.x * C * 2;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += red[(w * C + c) * 2];
      q += red[(w * C + c) * 2 + 1];
    }
    out[c * 2] = a;
    out[c * 2 + 1] = q;
  }
}


// ---- GroupNorm with the slab resident in registers ("single pass"): a 1024-thread workgroup owns HW pixels x CW channels
// of ONE image (CW = whole groups, a multiple of 8), loads that slab with every 16-B load in flight at once, reduces it,
// and rewrites it from registers - x (and dy) cross the fabric once instead of twice.  Thread t keeps chunk j = t % chunks
// (8 channels) of pixels t / chunks + k*P, k < NL.  The arithmetic runs channel-pair-outer / pixel-inner so that only one
// pair's coefficients and accumulators are live beside the NL (forward) or 2*NL (backward) data vectors: the backward
// form must fit 2 x 11 vectors + everything else in the 128 VGPRs a 16-wave workgroup gets (or 2 x 14 in the 168 of the
// 12-wave form, which gn_res_plan takes when its last sweep is fuller).
// The parts of one image sit 8 workgroup ids apart (same XCD): when CW*2 bytes is not a multiple of the 128-B line
// (320 channels = 32 groups of 10) neighbouring parts share lines, and the shared lines are then served by one L2.
// Measured (tools/slab_pass.hip, 256 x 1024 px x 320 ch): forward 2 x 160 channels 5.6 TB/s of useful bytes, backward
// 4 x 80 channels 3.9 TB/s (160-B segments; 3.1 without the XCD placement), line-aligned shapes 5.8-6.2 TB/s.
struct GnResParams {
  const bf16* X; const bf16* DY; const bf16* Radd; bf16* OUT;
  const float* gamma; const float* beta;
  float* mean_rstd;   // [B][G][2]: written by the forward, read by the backward
  float* partial;     // backward: [B][C][2] per-image channel sums (dz, dz*xhat) for dbeta / dgamma
  long ldx, lddy, ldr, ldo;
  int HW, C, G, cpg, CW, chunks, P, parts, peers8, silu, nseg;
  float eps, inv_n;
};

// data vectors are kept as four 32-bit words (two bf16 each) and updated a word at a time, so that the compiler never holds
// a half-built vector as eight separate values
DEVINL u32x4 ldnt8w(const bf16* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
DEVINL void st8w(bf16* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
DEVINL u32x4 ldnt8o(const bf16* base, unsigned byte_off) {   // uniform base + 32-bit lane offset (saddr form)
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) + byte_off));
}
DEVINL void st8o(bf16* base, unsigned byte_off, u32x4 v) {
  *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
DEVINL u32x4 sel8(bool ok, u32x4 v) {
  u32x4 r;
  r[0] = ok ? v[0] : 0u; r[1] = ok ? v[1] : 0u; r[2] = ok ? v[2] : 0u; r[3] = ok ? v[3] : 0u;
  return r;
}
DEVINL float bflo(unsigned w) { return __uint_as_float(w << 16); }
DEVINL float bfhi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
DEVINL unsigned bfpack(float lo, float hi) {
  bf16x2 v;
  v[0] = f2bf(lo);
  v[1] = f2bf(hi);
  return __builtin_bit_cast(unsigned, v);
}

DEVINL void gn_res_block(const GnResParams& p, int& img, int& part) {
  const int id = blockIdx.x;
  if (p.peers8) {
    const int span = 8 * p.parts;
    const int g = id / span, r = id - g * span;
    part = r >> 3;
    img = g * 8 + (r & 7);
  } else {
    img = id / p.parts;
    part = id - img * p.parts;
  }
}

// red[P][CW] float2 per-(pixel lane, channel) -> chs[CW] float2 per channel; fixed order.  Returns with chs valid.
DEVINL void gn_res_reduce(const GnResParams& p, const float2* red, float2* red2, float2* chs) {
  const int t = threadIdx.x, T = blockDim.x;
  __syncthreads();
  const int rows = p.P < p.HW ? p.P : p.HW;   // pixel lanes beyond HW only hold zeros
  for (int idx = t; idx < p.nseg * p.CW; idx += T) {
    const int seg = idx / p.CW, c = idx - seg * p.CW;
    float a0 = 0.f, q0 = 0.f, a1 = 0.f, q1 = 0.f;
    int k = seg;
    for (; k + p.nseg < rows; k += 2 * p.nseg) {
      const float2 v0 = red[(long)k * p.CW + c], v1 = red[(long)(k + p.nseg) * p.CW + c];
      a0 += v0.x; q0 += v0.y;
      a1 += v1.x; q1 += v1.y;
    }
    if (k < rows) {
      const float2 v0 = red[(long)k * p.CW + c];
      a0 += v0.x; q0 += v0.y;
    }
    red2[idx] = make_float2(a0 + a1, q0 + q1);
  }
  __syncthreads();
  for (int c = t; c < p.CW; c += T) {
    float a = 0.f, q = 0.f;
    for (int sgi = 0; sgi < p.nseg; ++sgi) {
      const float2 v = red2[sgi * p.CW + c];
      a += v.x; q += v.y;
    }
    chs[c] = make_float2(a, q);
  }
  __syncthreads();
}

// LDS floats: red 2*(P+1)*CW | red2 2*nseg*CW | chs 2*CW | cf 4*CW | gs 4*(CW/cpg)
static inline size_t gn_res_lds_floats(int P, int CW, int nseg, int cpg) {
  return (size_t)2 * (P + 1) * CW + (size_t)2 * nseg * CW + 2 * CW + 4 * CW + 4 * (CW / cpg);
}

template <int NL, bool SILU>
__global__ __launch_bounds__(1024) void gn_res_fwd_kernel(GnResParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* red = reinterpret_cast<float2*>(smem);
  float2* red2 = red + (long)(p.P + 1) * p.CW;   // row P: where idle pixel lanes dump their (zero) sums
  float2* chs = red2 + (long)p.nseg * p.CW;
  float2* cf = chs + p.CW;                       // per channel (scale, shift)
  float2* gs = cf + 2 * p.CW;                    // per group (mean, rstd)
  int img, part;
  gn_res_block(p, img, part);
  const int t = threadIdx.x, j = t % p.chunks, p0 = t / p.chunks;
  const bool live = p0 < p.P;
  const int c0 = part * p.CW;
  // every lane loads (rows beyond the slab re-read its last row and are zeroed): no branches around the loads, one 32-bit
  // offset per vector from a uniform base
  const bf16* xb = p.X + (long)img * p.HW * p.ldx + c0;
  const int pl = live ? p0 : p.P - 1;
  const int pl0 = live ? p0 : p.P;
  u32x4 a[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int pix = pl + k * p.P;
    const int pc = pix < p.HW ? pix : p.HW - 1;
    a[k] = ldnt8o(xb, (unsigned)(pc * (int)p.ldx + 8 * j) * 2u);
  }
#pragma unroll
  for (int k = 0; k < NL; ++k) a[k] = sel8(live && pl + k * p.P < p.HW, a[k]);
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    float s1 = 0.f, s2 = 0.f, u1 = 0.f, u2 = 0.f;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const float v0 = bflo(a[k][w]), v1 = bfhi(a[k][w]);
      s1 += v0;
      s2 = fmaf(v0, v0, s2);
      u1 += v1;
      u2 = fmaf(v1, v1, u2);
    }
    red[pl0 * p.CW + 8 * j + 2 * w] = make_float2(s1, s2);       // unconditional: a branch here lets the compiler sink and
    red[pl0 * p.CW + 8 * j + 2 * w + 1] = make_float2(u1, u2);   // interleave the whole sweep (registers)
  }
  gn_res_reduce(p, red, red2, chs);
#pragma unroll
  for (int k = 0; k < NL; ++k) reg_tie(a[k]);   // the second sweep re-reads the packed values: nothing converted is kept across
  const int ng = p.CW / p.cpg, g0 = c0 / p.cpg;
  for (int gl = t; gl < ng; gl += blockDim.x) {
    float s = 0.f, q = 0.f;
    for (int i = 0; i < p.cpg; ++i) {
      const float2 v = chs[gl * p.cpg + i];
      s += v.x; q += v.y;
    }
    const float mean = s * p.inv_n;
    const float var = fmaxf(q * p.inv_n - mean * mean, 0.f);
    const float rstd = rsqrtf(var + p.eps);
    gs[gl] = make_float2(mean, rstd);
    p.mean_rstd[((long)img * p.G + g0 + gl) * 2] = mean;
    p.mean_rstd[((long)img * p.G + g0 + gl) * 2 + 1] = rstd;
  }
  __syncthreads();
  for (int c = t; c < p.CW; c += blockDim.x) {
    const float2 g = gs[c / p.cpg];
    const float sc = g.y * p.gamma[c0 + c];
    cf[c] = make_float2(sc, p.beta[c0 + c] - g.x * sc);
  }
  __syncthreads();
  if (!live) return;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float2 f0 = cf[8 * j + 2 * w], f1 = cf[8 * j + 2 * w + 1];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      float z0 = fmaf(bflo(a[k][w]), f0.x, f0.y), z1 = fmaf(bfhi(a[k][w]), f1.x, f1.y);
      if (SILU) {
        z0 = silu_f(z0);
        z1 = silu_f(z1);
      }
      a[k][w] = bfpack(z0, z1);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  bf16* yb = p.OUT + (long)img * p.HW * p.ldo + c0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int pix = p0 + k * p.P;
    if (pix < p.HW) st8o(yb, (unsigned)(pix * (int)p.ldo + 8 * j) * 2u, a[k]);
  }
}

// backward: dx = dz*sc + x*Q + R (+ Radd), dz = dy*silu'(x*sc+sh);  per group c1 = mean(gamma*dz), c2 = mean(gamma*dz*xhat),
// Q = -rstd^2*c2, R = -rstd*c1 + mean*rstd^2*c2 (as gn_apply2_kernel<true>).  dz is recomputed in the second sweep rather
// than kept (no registers for it); Radd lands in the dy registers once they are dead and is added to the bf16-rounded dx.
template <int NL, bool SILU, int T>
__global__ __launch_bounds__(T) void gn_res_bwd_kernel(GnResParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* red = reinterpret_cast<float2*>(smem);
  float2* red2 = red + (long)(p.P + 1) * p.CW;   // row P: where idle pixel lanes dump their (zero) sums
  float2* chs = red2 + (long)p.nseg * p.CW;
  float4* cf = reinterpret_cast<float4*>(chs + p.CW);   // per channel (rstd, -mean*rstd, rstd*gamma, shift)
  float4* gs = cf + p.CW;                                // per group (mean, rstd, Q, R)
  int img, part;
  gn_res_block(p, img, part);
  const int t = threadIdx.x, j = t % p.chunks, p0 = t / p.chunks;
  const bool live = p0 < p.P;
  const int c0 = part * p.CW;
  const bf16* xb = p.X + (long)img * p.HW * p.ldx + c0;
  const bf16* dyb = p.DY + (long)img * p.HW * p.lddy + c0;
  const int pl = live ? p0 : p.P - 1;
  const int pl0 = live ? p0 : p.P;
  unsigned a[NL][4], b[NL][4];   // plain words, not vectors: each is rewritten on its own (dz over dy, dx over x)
  {
    u32x4 va[NL], vb[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int pix = pl + k * p.P;
      const int pc = pix < p.HW ? pix : p.HW - 1;
      va[k] = ldnt8o(xb, (unsigned)(pc * (int)p.ldx + 8 * j) * 2u);
      vb[k] = ldnt8o(dyb, (unsigned)(pc * (int)p.lddy + 8 * j) * 2u);
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const bool ok = live && pl + k * p.P < p.HW;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a[k][w] = ok ? va[k][w] : 0u;
        b[k][w] = ok ? vb[k][w] : 0u;
        reg_tie(b[k][w]);   // materialise now: a select sunk to its use keeps the whole loaded tuple alive beside the new words
      }
    }
  }
  for (int c = t; c < p.CW; c += blockDim.x) {
    const int g = (c0 + c) / p.cpg;
    const float mu = p.mean_rstd[((long)img * p.G + g) * 2], rs = p.mean_rstd[((long)img * p.G + g) * 2 + 1];
    const float nm = -mu * rs, ga = p.gamma[c0 + c];
    cf[c] = make_float4(rs, nm, rs * ga, fmaf(nm, ga, p.beta[c0 + c]));
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float4 f0 = cf[8 * j + 2 * w], f1 = cf[8 * j + 2 * w + 1];
    float s1 = 0.f, s2 = 0.f, u1 = 0.f, u2 = 0.f;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const float x0 = bflo(a[k][w]), x1 = bfhi(a[k][w]);
      float d0 = bflo(b[k][w]), d1 = bfhi(b[k][w]);
      if (SILU) {   // dz replaces dy in its registers, rounded to bf16 (what a separate SiLU backward would hand to GroupNorm):
        d0 *= dsilu_f(fmaf(x0, f0.z, f0.w));   // the second sweep then needs no second exp / rcp per element
        d1 *= dsilu_f(fmaf(x1, f1.z, f1.w));
        unsigned pk = bfpack(d0, d1);
        reg_tie(pk);   // pack HERE: left alone, the compiler keeps both floats of every element until the second sweep
        b[k][w] = pk;
      }
      s1 += d0;
      s2 = fmaf(d0, fmaf(x0, f0.x, f0.y), s2);
      u1 += d1;
      u2 = fmaf(d1, fmaf(x1, f1.x, f1.y), u2);
      if (SILU) __builtin_amdgcn_sched_barrier(0);   // bound the transcendental chains in flight (registers)
    }
    red[pl0 * p.CW + 8 * j + 2 * w] = make_float2(s1, s2);       // unconditional: a branch here lets the compiler sink and
    red[pl0 * p.CW + 8 * j + 2 * w + 1] = make_float2(u1, u2);   // interleave the whole sweep (registers)
    __builtin_amdgcn_sched_barrier(0);
  }
  gn_res_reduce(p, red, red2, chs);
#pragma unroll
  for (int k = 0; k < NL; ++k) {   // the second sweep recomputes dz from the packed values instead of keeping 8*NL floats
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      reg_tie(a[k][w]);
      reg_tie(b[k][w]);
    }
  }
  int j2 = j, q0 = p0;   // ... and the store / Radd offsets are derived after this point, not carried through the first sweep
  reg_tie(j2);
  reg_tie(q0);
  for (int c = t; c < p.CW; c += blockDim.x) {
    const float2 v = chs[c];
    p.partial[((long)img * p.C + c0 + c) * 2] = v.x;
    p.partial[((long)img * p.C + c0 + c) * 2 + 1] = v.y;
  }
  const int ng = p.CW / p.cpg, g0 = c0 / p.cpg;
  for (int gl = t; gl < ng; gl += blockDim.x) {
    float s = 0.f, q = 0.f;
    for (int i = 0; i < p.cpg; ++i) {
      const float2 v = chs[gl * p.cpg + i];
      const float ga = p.gamma[c0 + gl * p.cpg + i];
      s = fmaf(v.x, ga, s);
      q = fmaf(v.y, ga, q);
    }
    const float c1 = s * p.inv_n, c2 = q * p.inv_n;
    const float mu = p.mean_rstd[((long)img * p.G + g0 + gl) * 2], rs = p.mean_rstd[((long)img * p.G + g0 + gl) * 2 + 1];
    gs[gl] = make_float4(mu, rs, -rs * rs * c2, -rs * c1 + mu * rs * rs * c2);
  }
  __syncthreads();
  if (!live) return;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float4 f0 = cf[8 * j2 + 2 * w], f1 = cf[8 * j2 + 2 * w + 1];
    const float4 g0q = gs[(8 * j2 + 2 * w) / p.cpg], g1q = gs[(8 * j2 + 2 * w + 1) / p.cpg];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const float x0 = bflo(a[k][w]), x1 = bfhi(a[k][w]);
      const float d0 = bflo(b[k][w]), d1 = bfhi(b[k][w]);
      a[k][w] = bfpack(fmaf(d0, f0.z, fmaf(x0, g0q.z, g0q.w)), fmaf(d1, f1.z, fmaf(x1, g1q.z, g1q.w)));
      if (SILU) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  bf16* ob = p.OUT + (long)img * p.HW * p.ldo + c0;
  if (p.Radd) {
    const bf16* rb = p.Radd + (long)img * p.HW * p.ldr + c0;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int pix = q0 + k * p.P;
      const int pc = pix < p.HW ? pix : p.HW - 1;
      const u32x4 v = ldnt8o(rb, (unsigned)(pc * (int)p.ldr + 8 * j2) * 2u);
#pragma unroll
      for (int w = 0; w < 4; ++w) b[k][w] = v[w];
    }
#pragma unroll
    for (int k = 0; k < NL; ++k)
#pragma unroll
      for (int w = 0; w < 4; ++w)
        a[k][w] = bfpack(bflo(a[k][w]) + bflo(b[k][w]), bfhi(a[k][w]) + bfhi(b[k][w]));
  }
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int pix = q0 + k * p.P;
    if (pix < p.HW) st8o(ob, (unsigned)(pix * (int)p.ldo + 8 * j2) * 2u, u32x4{a[k][0], a[k][1], a[k][2], a[k][3]});
  }
}

// Slab geometry for (B, HW, C, G): CW = whole groups, whole 16-B vectors, NL data vectors per thread <= max_nl.
// Prefers line-aligned segments, then the widest part that still gives every CU a workgroup.  false -> no fit (the
// multi-pass kernels take the call).
bool gn_res_plan(int B, int HW, int C, int G, long ld_min, long ld_max, bool bwd, GnResParams& p, int& nl, int& threads) {
  const int cpg = C / G;
  int unit = cpg;
  while (unit & 7) unit += cpg;                 // lcm(cpg, 8)
  if (unit > C || (C % unit)) return false;
  if ((long)HW * ld_max * 2 >= (1L << 31)) return false;   // 32-bit byte offsets inside an image
  // forward: 1024 threads, <= 21 vectors each.  backward (two resident tensors): 1024 threads x <= 8 vectors, or 768
  // threads (168 VGPRs) x <= 14
  const int nT = (bwd && g_gn_resident_form != 1) ? 2 : 1;
  const int Ts[2] = {1024, 768};
  const int maxnl[2] = {bwd ? (g_gn_resident_form == 2 ? 8 : 11) : 21, 14};
  double best = 1e30;
  int best_cw = 0, best_t = 0;
  for (int ti = 0; ti < nT; ++ti) {
    const int T = Ts[ti];
    for (int cw = unit; cw <= C; cw += unit) {
      if (C % cw) continue;
      const int chunks = cw / 8;
      if (chunks > T) break;
      const int P = T / chunks;
      const int n = (HW + P - 1) / P;
      if (n > maxnl[ti]) continue;
      const long wgs = (long)B * (C / cw);
      if (wgs < g_gn_resident) continue;         // too few workgroups for 256 CUs: the chunked kernels spread an image wider
      const bool aligned = ((cw * 2) % 128 == 0) && ((ld_min * 2) % 128 == 0);
      double cost = aligned ? 1.0 : (double)(cw * 2 + 128) / (double)(cw * 2);   // expected line over-fetch
      if (cost > 1.85) continue;                  // 80-B segments and narrower fetch more than the extra passes they save
      // ties: the widest slab (a workgroup's load -> reduce -> store chain is serial; 40-KB slabs measured 0.4-0.9x of the
      // multi-pass kernels, 160-KB ones 1.2-1.7x), the 16-wave form
      if ((long)HW * cw * 2 < g_gn_resident_min_slab) continue;
      if (gn_res_lds_floats(P, cw, T / cw < 1 ? 1 : T / cw, cpg) * sizeof(float) > 160 * 1024) continue;
      // ... then the form whose last sweep is fuller: HW / (P * n) of the loads carry data (1,024 px x 10 chunks: 0.91 with
      // 1,024 threads x 11 vectors, 0.96 with 768 x 14 - measured +2...5 % for the latter, -18 % where it forces a narrower slab)
      cost += 0.8 * (1.0 - (double)HW / ((double)P * n)) + 0.01 * ti - 0.002 * n * T / 1024.0;
      if (wgs < 256) cost += 0.05;
      if (cost < best) { best = cost; best_cw = cw; best_t = T; }
    }
  }
  if (!best_cw) return false;
  threads = best_t;
  p.CW = best_cw;
  p.chunks = best_cw / 8;
  p.P = best_t / p.chunks;
  p.parts = C / best_cw;
  p.peers8 = (B % 8 == 0 && p.parts > 1) ? 1 : 0;
  p.nseg = best_t / best_cw < 1 ? 1 : best_t / best_cw;
  nl = (HW + p.P - 1) / p.P;
  return true;
}

template <int NL, bool SILU>
int launch_gn_res_fwd2(const GnResParams& p, int B, hipStream_t stream) {
  const size_t smem = gn_res_lds_floats(p.P, p.CW, p.nseg, p.cpg) * sizeof(float);
  if (smem > 160 * 1024) return DA_ERR_SHAPE;
  static unsigned long long attr_done = 0;  // one bit per device
  if (da_ensure_dyn_smem((const void*)gn_res_fwd_kernel<NL, SILU>, 160 * 1024, &attr_done) != DA_OK) return DA_ERR_LAUNCH;
  hipLaunchKernelGGL((gn_res_fwd_kernel<NL, SILU>), dim3(B * p.parts), dim3(1024), smem, stream, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

template <int NL>
int launch_gn_res_fwd(const GnResParams& p, int B, hipStream_t stream) {
  return p.silu ? launch_gn_res_fwd2<NL, true>(p, B, stream) : launch_gn_res_fwd2<NL, false>(p, B, stream);
}

template <int NL, bool SILU, int T>
int launch_gn_res_bwd2(const GnResParams& p, int B, hipStream_t stream) {
  const size_t smem = gn_res_lds_floats(p.P, p.CW, p.nseg, p.cpg) * sizeof(float);
  if (smem > 160 * 1024) return DA_ERR_SHAPE;
  static unsigned long long attr_done = 0;  // one bit per device
  if (da_ensure_dyn_smem((const void*)gn_res_bwd_kernel<NL, SILU, T>, 160 * 1024, &attr_done) != DA_OK) return DA_ERR_LAUNCH;
  hipLaunchKernelGGL((gn_res_bwd_kernel<NL, SILU, T>), dim3(B * p.parts), dim3(T), smem, stream, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

template <int NL, int T>
int launch_gn_res_bwd(const GnResParams& p, int B, hipStream_t stream) {
  return p.silu ? launch_gn_res_bwd2<NL, true, T>(p, B, stream) : launch_gn_res_bwd2<NL, false, T>(p, B, stream);
}

int dispatch_gn_res_fwd(int nl, const GnResParams& p, int B, hipStream_t s) {
  if (nl <= 1) return launch_gn_res_fwd<1>(p, B, s);
  if (nl <= 2) return launch_gn_res_fwd<2>(p, B, s);
  if (nl <= 3) return launch_gn_res_fwd<3>(p, B, s);
  if (nl <= 4) return launch_gn_res_fwd<4>(p, B, s);
  if (nl <= 6) return launch_gn_res_fwd<6>(p, B, s);
  if (nl <= 8) return launch_gn_res_fwd<8>(p, B, s);
  if (nl <= 11) return launch_gn_res_fwd<11>(p, B, s);
  if (nl <= 16) return launch_gn_res_fwd<16>(p, B, s);
  return launch_gn_res_fwd<21>(p, B, s);
}

int dispatch_gn_res_bwd(int nl, int threads, const GnResParams& p, int B, hipStream_t s) {
  if (threads == 768) {
    if (nl <= 11) return launch_gn_res_bwd<11, 768>(p, B, s);
    return launch_gn_res_bwd<14, 768>(p, B, s);
  }
  if (nl <= 1) return launch_gn_res_bwd<1, 1024>(p, B, s);
  if (nl <= 2) return launch_gn_res_bwd<2, 1024>(p, B, s);
  if (nl <= 3) return launch_gn_res_bwd<3, 1024>(p, B, s);
  if (nl <= 4) return launch_gn_res_bwd<4, 1024>(p, B, s);
  if (nl <= 6) return launch_gn_res_bwd<6, 1024>(p, B, s);
  if (nl <= 8) return launch_gn_res_bwd<8, 1024>(p, B, s);
  return launch_gn_res_bwd<11, 1024>(p, B, s);
}

}  // namespace

extern "C" long da_norm_scratch_floats(int B, int HW, int C) {
  // upper bound of partial-sum floats any norm / colsum entry point below needs
  long a = (long)B * pick_chunks(B, HW) * C * 2;
  long b = 1024L * C * 2;  // LayerNorm backward: <= 1024 partial rows; colsum: <= 256
  return a > b ? a : b;
}

extern "C" int da_groupnorm_fwd(const void* X, long ldx, void* Y, long ldy, const float* gamma, const float* beta,
                                float* mean_rstd, float* scale_shift, float* scratch, int B, int HW, int C, int G,
                                float eps, int silu, hipStream_t stream) {
  DA_CLEAR_ERR();
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || (C % G) || (C & 7) || (ldx & 7) || (ldy & 7)) return DA_ERR_SHAPE;
  if (g_gn_resident) {
    GnResParams r = {};
    int nl = 0, threads = 0;
    if (gn_res_plan(B, HW, C, G, ldx < ldy ? ldx : ldy, ldx > ldy ? ldx : ldy, false, r, nl, threads)) {
      r.X = (const bf16*)X; r.ldx = ldx; r.OUT = (bf16*)Y; r.ldo = ldy; r.gamma = gamma; r.beta = beta;
      r.mean_rstd = mean_rstd; r.HW = HW; r.C = C; r.G = G; r.cpg = C / G; r.silu = silu; r.eps = eps;
      r.inv_n = 1.0f / ((float)(C / G) * (float)HW);
      return dispatch_gn_res_fwd(nl, r, B, stream);
    }
  }
  ChanReduceParams p = {};
  p.X = (const bf16*)X; p.ldx = ldx; p.partial = scratch;
  p.HW = HW; p.C = C; p.G = G; p.cpg = C / G; p.nchunks = pick_chunks(B, HW);
  int rc = launch_chan_reduce(0, p, B, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(gn_fwd_finalize_kernel, dim3(B, (G + GN_GPB - 1) / GN_GPB), dim3(256),
                     (size_t)(2 * GN_GPB * (C / G) + 2 * GN_GPB) * sizeof(float), stream,
                     scratch, gamma, beta, mean_rstd, scale_shift, C, G, C / G, p.nchunks, HW, eps);
  DA_CHECK_LAUNCH();
  GnApplyParams ap = {};
  ap.X = (const bf16*)X; ap.ldx = ldx; ap.OUT = (bf16*)Y; ap.ldo = ldy;
  ap.mean_rstd = mean_rstd; ap.gamma = gamma; ap.beta = beta;
  ap.HW = HW; ap.C = C; ap.G = G; ap.cpg = C / G; ap.silu = silu;
  return launch_gn_apply2(false, ap, B, stream);
}

extern "C" int da_groupnorm_bwd(const void* X, long ldx, const void* dY, long lddy, const void* Radd, long ldr,
                                void* dX, long lddx, const float* gamma, const float* beta, const float* mean_rstd,
                                float* dgamma, float* dbeta, float* coef, float* scratch, int B, int HW, int C, int G,
                                int silu, hipStream_t stream) {
  DA_CLEAR_ERR();
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || (C % G) || (C & 7) || (ldx & 7) || (lddy & 7) || (lddx & 7))
    return DA_ERR_SHAPE;
  if (Radd && (ldr & 7)) return DA_ERR_SHAPE;
  if (g_gn_resident) {
    GnResParams r = {};
    int nl = 0, threads = 0;
    long ldm = ldx < lddy ? ldx : lddy, ldM = ldx > lddy ? ldx : lddy;
    if (lddx < ldm) ldm = lddx;
    if (lddx > ldM) ldM = lddx;
    if (Radd && ldr > ldM) ldM = ldr;
    if (gn_res_plan(B, HW, C, G, ldm, ldM, true, r, nl, threads)) {
      r.X = (const bf16*)X; r.ldx = ldx; r.DY = (const bf16*)dY; r.lddy = lddy; r.Radd = (const bf16*)Radd; r.ldr = ldr;
      r.OUT = (bf16*)dX; r.ldo = lddx; r.gamma = gamma; r.beta = beta; r.mean_rstd = const_cast<float*>(mean_rstd);
      r.partial = scratch; r.HW = HW; r.C = C; r.G = G; r.cpg = C / G; r.silu = silu;
      r.inv_n = 1.0f / ((float)(C / G) * (float)HW);
      int rc = dispatch_gn_res_bwd(nl, threads, r, B, stream);
      if (rc) return rc;
      // dbeta[c] (+)= sum_b s1, dgamma[c] (+)= sum_b s2: one partial row per image
      hipLaunchKernelGGL(chan_sum_finalize_kernel, chan_sum_grid(B, C), dim3(CSF_CH * CSF_RL), 0, stream, scratch, B, C, dbeta,
                         dgamma, g_grad_overwrite);
      DA_CHECK_LAUNCH();
      return DA_OK;
    }
  }
  ChanReduceParams p = {};
  p.X = (const bf16*)X; p.ldx = ldx; p.DY = (const bf16*)dY; p.lddy = lddy;
  p.mean_rstd = mean_rstd; p.gamma = gamma; p.beta = beta; p.partial = scratch;
  p.HW = HW; p.C = C; p.G = G; p.cpg = C / G; p.nchunks = pick_chunks(B, HW); p.silu = silu;
  int rc = launch_chan_reduce(1, p, B, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B, (G + GN_GPB - 1) / GN_GPB), dim3(256),
                     (size_t)(2 * GN_GPB * (C / G)) * sizeof(float), stream, scratch,
                     gamma, coef, C, G, C / G, p.nchunks, HW);
  DA_CHECK_LAUNCH();
  // dgamma[c] += sum_b s2, dbeta[c] += sum_b s1
  hipLaunchKernelGGL(chan_sum_finalize_kernel, chan_sum_grid(B * p.nchunks, C), dim3(CSF_CH * CSF_RL), 0, stream, scratch,
                     B * p.nchunks, C, dbeta, dgamma, g_grad_overwrite);
  DA_CHECK_LAUNCH();
  GnApplyParams ap = {};
  ap.X = (const bf16*)X; ap.ldx = ldx; ap.DY = (const bf16*)dY; ap.lddy = lddy; ap.Radd = (const bf16*)Radd; ap.ldr = ldr;
  ap.OUT = (bf16*)dX; ap.ldo = lddx;
  ap.mean_rstd = mean_rstd; ap.coef = coef; ap.gamma = gamma; ap.beta = beta;
  ap.HW = HW; ap.C = C; ap.G = G; ap.cpg = C / G; ap.silu = silu;
  return launch_gn_apply2(true, ap, B, stream);
}

extern "C" int da_colsum_accum(const void* X, long ldx, float* out, float* scratch, int M, int C,
                               hipStream_t stream) {
  DA_CLEAR_ERR();
  if (M <= 0 || C <= 0 || (C & 7) || (ldx & 7)) return DA_ERR_SHAPE;
  ChanReduceParams p = {};
  p.X = (const bf16*)X; p.ldx = ldx; p.partial = scratch;
  p.HW = M; p.C = C; p.G = 1; p.cpg = C;
  int n = M / 64;
  if (n > 256) n = 256;
  if (n < 1) n = 1;
  p.nchunks = n;
  int rc = launch_chan_reduce(2, p, 1, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(chan_sum_finalize_kernel, chan_sum_grid(n, C), dim3(CSF_CH * CSF_RL), 0, stream, scratch, n, C, out,
                     (float*)nullptr, g_grad_overwrite);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_image_colsum(const void* X, long ldx, void* out, long ldo, float* db, float* scratch, int B, int HW,
                               int C, hipStream_t stream) {
  DA_CLEAR_ERR();
  if (B <= 0 || HW <= 0 || C <= 0 || (C & 7) || (ldx & 7)) return DA_ERR_SHAPE;
  ChanReduceParams p = {};
  p.X = (const bf16*)X; p.ldx = ldx; p.partial = scratch;
  p.HW = HW; p.C = C; p.G = 1; p.cpg = C; p.nchunks = pick_chunks(B, HW);
  int rc = launch_chan_reduce(2, p, B, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(image_colsum_finalize_kernel, dim3((C + CSF_CH - 1) / CSF_CH), dim3(CSF_CH * CSF_RL), 0, stream,
                     scratch, (bf16*)out, ldo, db, B, p.nchunks, C, g_grad_overwrite);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_layernorm_fwd(const void* X, long ldx, void* Y, long ldy, const float* gamma, const float* beta,
                                float* mean_rstd, int M, int C, float eps, hipStream_t stream) {
  DA_CLEAR_ERR();
  if (M <= 0 || C <= 0 || (C & 7) || C > 8 * 64 * LN_MAXV || (ldx & 7) || (ldy & 7)) return DA_ERR_SHAPE;
  if (C == 320 || C == 640 || C == 1280) {
    const int rpw = 64 / (C / 40);
    long b5 = ((long)M + 4 * rpw - 1) / (4 * rpw);
    if (b5 > 2048) b5 = 2048;
#define LN_FWD5(L) hipLaunchKernelGGL(ln_fwd5_kernel<L>, dim3((int)b5), dim3(256), 0, stream, (const bf16*)X, ldx, \
                                      (bf16*)Y, ldy, gamma, beta, mean_rstd, M, eps)
    if (C == 320) LN_FWD5(8); else if (C == 640) LN_FWD5(16); else LN_FWD5(32);
#undef LN_FWD5
    DA_CHECK_LAUNCH();
    return DA_OK;
  }
  int blocks = (M + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16*)X, ldx, (bf16*)Y, ldy, gamma,
                     beta, mean_rstd, M, C, eps);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_layernorm_bwd(const void* X, long ldx, const void* dY, long lddy, const void* Radd, long ldr,
                                void* dX, long lddx, const float* gamma, const float* mean_rstd, float* dgamma,
                                float* dbeta, float* scratch, int M, int C, hipStream_t stream) {
  DA_CLEAR_ERR();
  if (M <= 0 || C <= 0 || (C & 7) || C > 8 * 64 * LN_MAXV || (ldx & 7) || (lddy & 7) || (lddx & 7))
    return DA_ERR_SHAPE;
  if (Radd && (ldr & 7)) return DA_ERR_SHAPE;
  if (C == 320 || C == 640 || C == 1280) {
    const int rpw = 64 / (C / 40);
    // >= 8 row groups per wave amortise the dgamma/dbeta fold; <= 1024 partial rows in scratch (da_norm_scratch_floats)
    long b5 = ((long)M + 32 * rpw - 1) / (32 * rpw);
    if (b5 < 256) b5 = ((long)M + 4 * rpw - 1) / (4 * rpw) < 256 ? ((long)M + 4 * rpw - 1) / (4 * rpw) : 256;
    if (b5 > 1024) b5 = 1024;
#define LN_BWD5(L)                                                                                                   \
  do {                                                                                                               \
    if (Radd)                                                                                                        \
      hipLaunchKernelGGL((ln_bwd5_kernel<L, true>), dim3((int)b5), dim3(256), 0, stream, (const bf16*)X, ldx,         \
                         (const bf16*)dY, lddy, (const bf16*)Radd, ldr, (bf16*)dX, lddx, gamma, mean_rstd, scratch, M); \
    else                                                                                                             \
      hipLaunchKernelGGL((ln_bwd5_kernel<L, false>), dim3((int)b5), dim3(256), 0, stream, (const bf16*)X, ldx,        \
                         (const bf16*)dY, lddy, (const bf16*)Radd, ldr, (bf16*)dX, lddx, gamma, mean_rstd, scratch, M); \
  } while (0)
    if (C == 320) LN_BWD5(8); else if (C == 640) LN_BWD5(16); else LN_BWD5(32);
#undef LN_BWD5
    DA_CHECK_LAUNCH();
    hipLaunchKernelGGL(chan_sum_finalize_kernel, chan_sum_grid((int)b5, C), dim3(CSF_CH * CSF_RL), 0, stream, scratch, (int)b5, C,
                       dgamma, dbeta, g_grad_overwrite);
    DA_CHECK_LAUNCH();
    return DA_OK;
  }
  int blocks = (M + 3) / 4;
  if (blocks > 1024) blocks = 1024;  // 4 waves each: >= 16 waves per CU in flight for this HBM-bound pass
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), (size_t)4 * C * 2 * sizeof(float), stream,
                     (const bf16*)X, ldx, (const bf16*)dY, lddy, (const bf16*)Radd, ldr, (bf16*)dX, lddx, gamma,
                     mean_rstd, scratch, M, C);
  DA_CHECK_LAUNCH();
  hipLaunchKernelGGL(chan_sum_finalize_kernel, chan_sum_grid(blocks, C), dim3(CSF_CH * CSF_RL), 0, stream, scratch, blocks, C,
                     dgamma, dbeta, g_grad_overwrite);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
