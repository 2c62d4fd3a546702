// Implicit-GEMM "NT" kernel for gfx950:  C[M,N] = alpha * gather(A)[M,K] . W[N,K]^T  (+bias +rowbias +R)
//
// One kernel serves every K-contiguous contraction of the U-Net step:
//   * 3x3 / 1x1 convolution forward on NHWC activations, im2col-free: the A row of output pixel m for
//     tap (r,s) is the channel vector of input pixel (oh*stride+r-1, ow*stride+s-1); out-of-image taps
//     read zeros.  K = taps*Cin, W is [Cout][r][s][Cin].
//   * convolution dgrad: the same gather over dY with the flipped/transposed weight shadow
//     Wt[Cin][2-r][2-s][Cout] (mode 2 handles the stride-2 downsampler's dgrad parity rule).
//   * linear forward / dgrad (ksize 1, one "pixel" per row).
// Tile: 128(M) x 128(N) x 64(K) per 256-thread workgroup; 4 waves as 2x2, each 64x64 built from
// 4x4 v_mfma_f32_16x16x32_bf16 accumulators (fp32).  Global -> registers -> LDS staging (the gather
// and zero padding need per-row predicates, so no LDS-DMA), two LDS buffers, one barrier per K-step.
// LDS image per operand: [128 rows][64 k] bf16, 128-B rows, 16-B chunk index XOR ((row>>1)&7) so the
// ds_read_b128 fragment reads of the 16x16x32 operand map are bank-conflict free.
// Epilogue: accumulators -> per-wave fp32 LDS tile -> full 16-B row segments with fused
// bias / per-image row bias (timestep FiLM) / residual add -> bf16 (or fp32) global stores.
#include <string.h>

#include "common.hpp"
#include "diffusion_amd.h"

namespace {

struct GemmNTParams {
  const bf16* A;
  const bf16* W;
  void* C;
  const float* bias;
  const bf16* rowbias;
  const bf16* R;
  long lda, ldc, ldrb, ldr;
  int M, N, K, Cin;
  int Hin, Win, Hout, Wout;
  int ksize, mode;
  int out_fp32;
  float alpha;
  int tiles_m, tiles_n;
};

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;          // 32 KiB
constexpr int EPI_LD = 68;                               // fp32 row stride of the epilogue tile
constexpr int EPI_BYTES = 4 * 64 * EPI_LD * 4;           // 69632
constexpr int SMEM_BYTES = (2 * STAGE_BYTES > EPI_BYTES) ? 2 * STAGE_BYTES : EPI_BYTES;

DEVINL int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNTParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware remap: blocks b, b+8, ... share an XCD; give each XCD a contiguous run of tiles.
  const int nblk = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int scol = tid & 7;
  const int srow = tid >> 3;
  const int HWo = p.Hout * p.Wout;
  const int pad = (p.ksize == 3) ? 1 : 0;

  int pixbase[4], oh[4], ow[4];
  bool mval[4];
  const bf16* wptr[4];
  bool nval[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + srow + 32 * i;
    mval[i] = m < p.M;
    int mm = mval[i] ? m : 0;
    int b = mm / HWo;
    int rem = mm - b * HWo;
    oh[i] = rem / p.Wout;
    ow[i] = rem - oh[i] * p.Wout;
    pixbase[i] = b * p.Hin * p.Win;
    int n = n0 + srow + 32 * i;
    nval[i] = n < p.N;
    wptr[i] = p.W + (long)(nval[i] ? n : 0) * p.K;
  }

  int kk = scol * 8;
  int tap = kk / p.Cin;
  int cc = kk - tap * p.Cin;

  bf16x8 ra[4], rb[4];
  auto load_tile = [&]() {
    const bool kval = kk < p.K;
    int r = 0, s = 0;
    if (p.ksize == 3) {
      r = tap / 3;
      s = tap - 3 * r;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ih, iw;
      bool ok = kval && mval[i];
      if (p.mode == 0) {
        ih = oh[i] + r - pad;
        iw = ow[i] + s - pad;
      } else if (p.mode == 1) {
        ih = 2 * oh[i] + r - pad;
        iw = 2 * ow[i] + s - pad;
      } else if (p.mode == 4) {  // stride 2 over an image zero-padded at the bottom / right only (VAE downsampler)
        ih = 2 * oh[i] + r;
        iw = 2 * ow[i] + s;
      } else if (p.mode == 2) {
        int th = oh[i] + r - 1, tw = ow[i] + s - 1;
        ok = ok && !((th | tw) & 1);
        ih = th >> 1;
        iw = tw >> 1;
      } else {  // mode 3: conv over the nearest-2x upsampled input
        int th = oh[i] + r - 1, tw = ow[i] + s - 1;
        ok = ok && th >= 0 && tw >= 0 && th < p.Hout && tw < p.Wout;
        ih = th >> 1;
        iw = tw >> 1;
      }
      ok = ok && ih >= 0 && iw >= 0 && ih < p.Hin && iw < p.Win;
      ra[i] = ok ? ld8(p.A + (long)(pixbase[i] + ih * p.Win + iw) * p.lda + cc) : zero8();
      rb[i] = (kval && nval[i]) ? ld8(wptr[i] + kk) : zero8();
    }
    kk += BK;
    cc += BK;
    while (cc >= p.Cin) {
      cc -= p.Cin;
      ++tap;
    }
  };
  auto store_tile = [&](int buf) {
    char* Ab = smem + buf * STAGE_BYTES;
    char* Bb = Ab + BM * BK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = srow + 32 * i;
      int off = swz(row, scol);
      *reinterpret_cast<bf16x8*>(Ab + off) = ra[i];
      *reinterpret_cast<bf16x8*>(Bb + off) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char* Ab = smem + buf * STAGE_BYTES;
    const char* Bb = Ab + BM * BK * 2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[4], b[4];
      const int chunk = s * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int row = wm * 64 + i * 16 + (lane & 15);
        a[i] = *reinterpret_cast<const bf16x8*>(Ab + swz(row, chunk));
        int rowb = wn * 64 + i * 16 + (lane & 15);
        b[i] = *reinterpret_cast<const bf16x8*>(Bb + swz(rowb, chunk));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nk = (p.K + BK - 1) / BK;
  load_tile();
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) load_tile();
    compute(t & 1);
    if (t + 1 < nk) store_tile((t + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: acc -> per-wave fp32 LDS tile -> coalesced row segments
  float* ew = reinterpret_cast<float*>(smem) + wave * (64 * EPI_LD);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int row = i * 16 + (lane >> 4) * 4 + e;
        int col = j * 16 + (lane & 15);
        ew[row * EPI_LD + col] = acc[i][j][e];
      }
  __syncthreads();
  const int col8 = (lane & 7) * 8;
  const int n = n0 + wn * 64 + col8;
  if (n < p.N) {
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = p.bias ? p.bias[n + e] : 0.f;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      int row = pass * 8 + (lane >> 3);
      int m = m0 + wm * 64 + row;
      if (m >= p.M) continue;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&ew[row * EPI_LD + col8]);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(&ew[row * EPI_LD + col8 + 4]);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bv[e];
      if (p.rowbias) {
        int b = m / HWo;
        bf16x8 rbv = ld8(p.rowbias + (long)b * p.ldrb + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bf2f(rbv[e]);
      }
      if (p.R) {
        bf16x8 rv = ld8(p.R + (long)m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bf2f(rv[e]);
      }
      if (p.out_fp32) {
        float* cp = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
        *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
        st8(reinterpret_cast<bf16*>(p.C) + (long)m * p.ldc + n, o);
      }
    }
  }
}

}  // namespace

// large-tile LDS-DMA variant (gemm_nt_v2.hip)
int da_gemm_nt_v2_dispatch(int variant, int splits, float* ws, const void* A, long lda, const void* W, void* C, long ldc, const float* bias,
                           const void* rowbias, long ldrb, const void* R, long ldr, int M, int N, int K, int Cin,
                           int Hin, int Win, int Hout, int Wout, int ksize, int mode, int out_fp32, float alpha,
                           hipStream_t stream);

// streaming form of the short-K linears (gemm_nt_v3.hip); -1 = not a shape for it
int da_gemm_nt_v3_try(const void* A, long lda, const void* W, void* C, long ldc, const float* bias, const void* R, long ldr,
                      int M, int N, int K, hipStream_t stream);
extern int g_nt_stream, g_nt_stream_lw;
// weight-stationary form of the K = 320 linears (gemm_nt_ws.hip); -1 = not a shape for it
int da_gemm_nt_ws_try(const void* A, long lda, const void* W, const float* bias, const void* R, long ldr, void* C, long ldc,
                      int M, int N, int K, hipStream_t stream);
extern int g_nt_ws;

extern int g_tn_variant;  // gemm_tn.hip
extern int g_nt_korder;  // gemm_nt_v2.hip
extern int g_nt_persist;
extern int g_reserve_cus;
extern int g_nt_persist_conv;
extern int g_nt_de;
extern int g_grad_overwrite;
extern int g_attn_fused_bwd;  // attention.hip
extern int g_tn_ring;  // gemm_tn_v2.hip
extern int g_gn_resident, g_gn_resident_form, g_gn_resident_min_slab;  // norms.hip
int da_usable_cus(int cus);
static int g_nt_variant = 0;
static int g_nt_mfma32 = 0;   // da_set_option("gemm_nt_mfma32", 0 never | -1 for K <= 320 | 1 always): the 256x320 form on
                              // v_mfma_f32_32x32x16_bf16 (variant 15) where the cost model picks variant 12
static int g_nt_splitk = 1;  // da_set_option("gemm_nt_splitk", 0/1)  // 0 auto, 1 force v1 (128x128), 4 / 5 force v2 with BN 128 / 160 (when eligible)

// 1 -> gemm_nt_kernel (128x128), 4 / 5 / 10 / 12 -> gemm_nt2_kernel with BN 128 / 160 / 320 / 320.  12 is the default
// 256x320 form: 16 waves (4 per SIMD, 64x80 each, <= 128 VGPRs) instead of 10's 8 waves of 128x80 - the extra resident waves
// cover LDS latency and each other's epilogues (+28-39 % on the K <= 640 linears, parity on the longest-K convs).
// (11: the 4-wave 128x320x32
// instantiation, forced only: two workgroups per CU stay phase-locked, so it gains <= 5 % on one K=320 shape class).  *splits > 1: split-K over
// that many workgroups per tile (needs a workspace of splits*M*N floats) - used when the tile grid alone would
// leave most of the 256 CUs idle (small M: low-resolution layers, small microbatches).
static int g_nt_dispatch = 1;  // da_set_option("gemm_nt_dispatch", 0 legacy thresholds | 1 cost model)

// legacy rule (round 1): largest tile whose grid still fills most of the 256 CUs, split-K only for the 256x320 form
static int pick_nt_variant_legacy(int M, int N, int K, int Cin, long ws_floats, int* splits) {
  *splits = 1;
  const long tm = (M + 255) / 256;
  if (N % 320 == 0 && tm * (N / 320) >= 160) return 12;
  if (N % 160 == 0 && tm * (N / 160) >= 200) return 5;
  if (N % 256 == 0 && tm * (N / 256) >= 160) return 14;
  if (N % 160 != 0 && tm * ((N + 127) / 128) >= 200) return 4;
  if (N % 320 == 0 && g_nt_splitk) {
    const long tiles = tm * (N / 320);
    const int nk = K / 64;
    int s = (int)((256 + tiles - 1) / tiles);
    if (s > 8) s = 8;
    while (s > 1 && (nk / s < 8 || (long)s * M * N > ws_floats)) --s;  // >= 8 K-steps per split, workspace fits
    if (s > 1 && tiles * s >= 96) {
      *splits = s;
      return 12;
    }
  }
  return 1;
}

// Cost model over (tile form, split-K factor).  One workgroup per CU, so a launch takes
//   rounds(tiles * splits / 256 CUs) x (K-steps per split x step time + fixed prologue/epilogue) [+ split-K finalize].
// Step / fixed times per form are measured (tools/opt_ab.py, microbatch 16 and 256): the 16-wave 256x320 form retires a
// 64-deep step in ~1.5 us, the 8-wave 256x160 / 256x128 forms in ~1.1 us (latency-, not MFMA-bound), the 16-wave
// 256x256 form in ~1.3 us.  At large M every form runs many rounds and the widest tile wins (as the legacy rule had
// it); at small M (low-resolution levels, small microbatches) the legacy rule fell through to the 128x128 kernel -
// 36 % of the step time at the reference YAML's microbatch 16 - where a half-empty grid of large tiles is 20-35 % faster.
static int pick_nt_variant(int M, int N, int K, int Cin, long ws_floats, int* splits) {
  *splits = 1;
  if (Cin % 64 != 0) return 1;
  if (g_nt_variant == 4 || g_nt_variant == 5 || g_nt_variant == 10 || g_nt_variant == 11 || g_nt_variant == 12 || g_nt_variant == 14 ||
      g_nt_variant == 15 || g_nt_variant == 16 || g_nt_variant == 18)
    return g_nt_variant;
  if (g_nt_variant != 0) return 1;
  if (!g_nt_dispatch) return pick_nt_variant_legacy(M, N, K, Cin, ws_floats, splits);
  struct Form { int variant, bm, bn; double step_us, fixed_us; };
  // (the 16-wave 384x128 form: 1.37 us / step measured on the VAE encoder's 128-channel convs, +13-22 % over 256x128 there)
  static const Form forms[] = {{12, 256, 320, 1.5, 8.0}, {14, 256, 256, 1.3, 8.0}, {5, 256, 160, 1.16, 6.0}, {18, 384, 128, 1.37, 8.0},
                               {4, 256, 128, 1.07, 5.0}};
  const int nk = K / 64;
  const int ncu = da_usable_cus(256);  // da_set_option("reserve_cus")
  double best = 1e30;
  int best_v = 4, best_s = 1;
  for (const Form& f : forms) {
    const long tiles = ((M + f.bm - 1) / f.bm) * ((N + f.bn - 1) / f.bn);
    for (int s = 1; s <= (g_nt_splitk ? 8 : 1); ++s) {
      if (s > 1 && (nk / s < 4 || (long)s * M * N > ws_floats)) break;
      const long rounds = (tiles * s + ncu - 1) / ncu;
      const int steps = (nk + s - 1) / s;
      double t = (double)rounds * (steps * f.step_us + f.fixed_us);
      if (s > 1) t += 20.0 + ((double)s * M * N * 4.0 + (double)M * N * 2.0) / 5.0e6;  // finalize launch + slab traffic at ~5 TB/s
      if (t < best * 0.97) {  // ties go to the earlier (wider / unsplit) candidate
        best = t;
        best_v = f.variant;
        best_s = s;
      }
    }
  }
  *splits = best_s;
  // same tile on 32x32x16 matrix instructions (waves as 8 x 2): 10-15 % slower on long K (a third more LDS fragment reads);
  // 7-12 % faster on the 5-step K = 320 linears in isolation (instruction issue paces the short loop), no difference inside
  // the training step (540 vs 548 TFLOP/s on the 100 such launches) - kept as an option, off by default
  if (best_v == 12 && (g_nt_mfma32 == 1 || (g_nt_mfma32 < 0 && K <= 320))) return 15;
  return best_v;
}

extern "C" int da_gemm_nt_variant_for(int M, int N, int K, int Cin, long ws_floats) {
  int s;
  return pick_nt_variant(M, N, K, Cin, ws_floats, &s);
}

extern "C" int da_set_option(const char* key, int value) {
  if (key && !strcmp(key, "gemm_nt_variant")) {
    g_nt_variant = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gemm_nt_splitk")) {
    g_nt_splitk = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gemm_nt_dispatch")) {
    g_nt_dispatch = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gemm_nt_persist")) {
    g_nt_persist = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gemm_nt_korder")) {
    g_nt_korder = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gemm_nt_mfma32")) {
    g_nt_mfma32 = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gemm_tn_variant")) {
    g_tn_variant = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "attn_fused_bwd")) {
    g_attn_fused_bwd = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gemm_nt_ws")) {
    g_nt_ws = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gemm_tn_ring")) {
    if (value != 0 && value != 4 && value != 5) return DA_ERR_SHAPE;
    g_tn_ring = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gemm_nt_stream_lw")) {
    if (value != 4 && value != 16) return DA_ERR_SHAPE;
    g_nt_stream_lw = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gemm_nt_stream")) {
    g_nt_stream = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gemm_nt_de")) {
    g_nt_de = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gemm_nt_persist_conv")) {
    g_nt_persist_conv = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gn_resident")) {
    if (value < 0) return DA_ERR_SHAPE;
    g_gn_resident = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gn_resident_min_slab")) {
    if (value < 0) return DA_ERR_SHAPE;
    g_gn_resident_min_slab = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "gn_resident_form")) {
    if (value < 0 || value > 2) return DA_ERR_SHAPE;
    g_gn_resident_form = value;
    return DA_OK;
  }
  if (key && !strcmp(key, "grad_overwrite")) {
    g_grad_overwrite = value ? 1 : 0;
    return DA_OK;
  }
  if (key && !strcmp(key, "reserve_cus")) {
    if (value < 0 || value > 128) return DA_ERR_SHAPE;
    g_reserve_cus = value;
    return DA_OK;
  }
  return DA_ERR_SHAPE;
}

extern "C" int da_gemm_nt(const void* A, long lda, const void* W, void* C, long ldc, const float* bias,
                          const void* rowbias, long ldrb, const void* R, long ldr, int M, int N, int K, int Cin,
                          int Hin, int Win, int Hout, int Wout, int ksize, int mode, int out_fp32, float alpha,
                          float* splitk_ws, long splitk_ws_floats, hipStream_t stream) {
  DA_CLEAR_ERR();
  if (M <= 0 || N <= 0 || K <= 0) return DA_ERR_SHAPE;
  if ((N & 7) || (Cin & 7) || (K % Cin) || (lda & 7) || (ldc & 7)) return DA_ERR_SHAPE;
  if (ksize != 1 && ksize != 3) return DA_ERR_SHAPE;
  if (K != ksize * ksize * Cin) return DA_ERR_SHAPE;
  if (mode < 0 || mode > 4) return DA_ERR_SHAPE;
  if (mode == 4 && ksize != 3) return DA_ERR_SHAPE;
  if (Hout <= 0 || Wout <= 0 || (M % (Hout * Wout))) return DA_ERR_SHAPE;
  if (R && (ldr & 7)) return DA_ERR_SHAPE;
  if (rowbias && (ldrb & 7)) return DA_ERR_SHAPE;
  if (g_nt_ws && g_nt_variant == 0 && ksize == 1 && mode == 0 && !out_fp32 && alpha == 1.0f && !rowbias) {
    const int rc = da_gemm_nt_ws_try(A, lda, W, bias, R, ldr, C, ldc, M, N, K, stream);
    if (rc >= 0) return rc;
  }
  // linears whose K loop is short against their output: the streaming form (stores of a tile drained during the next
  // tile's K loop).  gemm_nt_stream: 0 off, 1 where it measured faster (K <= 640, enough row tiles for every CU), 2 wherever eligible
  if (g_nt_stream && g_nt_variant == 0 && ksize == 1 && mode == 0 && !out_fp32 && alpha == 1.0f && !rowbias &&
      (g_nt_stream == 2 || (K <= 640 && (long)((M + 127) / 128) * ((N + 319) / 320) >= 512))) {
    const int rc = da_gemm_nt_v3_try(A, lda, W, C, ldc, bias, R, ldr, M, N, K, stream);
    if (rc >= 0) return rc;
  }
  {
    int splits = 1;
    const int variant = pick_nt_variant(M, N, K, Cin, splitk_ws ? splitk_ws_floats : 0, &splits);
    if (variant != 1)
      return da_gemm_nt_v2_dispatch(variant, splits, splitk_ws, A, lda, W, C, ldc, bias, rowbias, ldrb, R, ldr, M, N, K, Cin, Hin, Win,
                                    Hout, Wout, ksize, mode, out_fp32, alpha, stream);
  }
  GemmNTParams p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = C; p.bias = bias;
  p.rowbias = (const bf16*)rowbias; p.R = (const bf16*)R;
  p.lda = lda; p.ldc = ldc; p.ldrb = ldrb; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.Cin = Cin;
  p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout;
  p.ksize = ksize; p.mode = mode; p.out_fp32 = out_fp32; p.alpha = alpha;
  p.tiles_m = (M + BM - 1) / BM;
  p.tiles_n = (N + BN - 1) / BN;
  static unsigned long long attr_done = 0;  // one bit per device
  if (da_ensure_dyn_smem((const void*)gemm_nt_kernel, SMEM_BYTES, &attr_done) != DA_OK) return DA_ERR_LAUNCH;
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(p.tiles_m * p.tiles_n), dim3(256), SMEM_BYTES, stream, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
