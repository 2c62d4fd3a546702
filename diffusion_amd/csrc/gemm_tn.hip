// Weight-gradient ("TN") implicit GEMM for gfx950:
//     dW[n][tap*Cin + c]  +=  sum_m  dY[m][n] * X[pixel(m) + tap][c]
// The contraction runs over output pixels m (the ROW index of both operands), so both operands reach
// the matrix cores through the gfx950 transposed LDS read (ds_read_b64_tr_b16): tiles are staged
// exactly as they lie in HBM ([m][channel], coalesced 16-B loads) and delivered k-major to the MFMA.
// Tile: 128(n) x 128(tap*Cin+c) fp32 accumulators per 256-thread workgroup, 32 pixels per step,
// 4 waves as 2x2 with 4x4 v_mfma_f32_16x16x32_bf16 tiles each.  LDS rows are padded to 288 B so the
// 8 rows a 32-lane half touches per transposed read land on disjoint banks.
// The pixel range is split over `splits` workgroups (small Cout x K layers have too few tiles to fill
// 256 CUs); partial tiles are stored as fp32 slabs in the caller's workspace and summed in a fixed order by
// tn1_slab_reduce_kernel (global_atomic_add_f32 only when no workspace was passed).
#include "common.hpp"
#include "diffusion_amd.h"

namespace {

struct GemmTNParams {
  const bf16* dY;
  const bf16* X;
  float* dW;
  long lddy, ldx;
  int M, N, Kt, Cin;
  int Hin, Win, Hout, Wout, ksize, mode;
  FastDiv div_hw, div_w;
  int tiles_n, tiles_k, splits, m_per_split;
  float* slab;  // splits > 1 with a workspace: partial tiles [tile][split][128][128] fp32, summed by tn1_slab_reduce_kernel
  int overwrite;  // da_set_option("grad_overwrite"): dW = ... instead of dW += ...
};

constexpr int TN_BM = 32;                 // pixels per step
constexpr int TN_LD = 288;                // bytes per LDS row (128 bf16 + 32 B pad)
constexpr int TN_TILE = TN_BM * TN_LD;    // 9216
constexpr int TN_STAGE = 2 * TN_TILE;     // 18432
constexpr int TN_SMEM = 2 * TN_STAGE;     // 36864

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTNParams p) {
  __shared__ __attribute__((aligned(16))) char smem[TN_SMEM];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;

  int bid = blockIdx.x;
  const int split = bid % p.splits;
  bid /= p.splits;
  const int tk = bid % p.tiles_k;
  const int tn = bid / p.tiles_k;
  const int n0 = tn * 128, k0 = tk * 128;
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);

  const int scol = tid & 15;
  const int srow = tid >> 4;  // rows srow, srow+16
  const int HWo = p.Hout * p.Wout;
  const int pad = (p.ksize == 3) ? 1 : 0;

  // fixed (tap, channel) of this thread's 16-B chunk of the X tile
  const int kk = k0 + scol * 8;
  const bool kval = kk < p.Kt;
  int tap = kval ? kk / p.Cin : 0;
  const int cc = kval ? kk - tap * p.Cin : 0;
  int r = 0, s = 0;
  if (p.ksize == 3) {
    r = tap / 3;
    s = tap - 3 * r;
  }
  const int ny = n0 + scol * 8;
  const bool nval = ny < p.N;

  bf16x8 ra[2], rb[2];
  int mcur = m_begin;
  auto load_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int m = mcur + srow + 16 * i;
      bool mv = m < m_end;
      ra[i] = (mv && nval) ? ld8(p.dY + (long)m * p.lddy + ny) : zero8();
      bool ok = mv && kval;
      unsigned mm = mv ? (unsigned)m : 0u;
      unsigned b = fdiv(mm, p.div_hw);
      unsigned rem = mm - b * (unsigned)HWo;
      int oh = (int)fdiv(rem, p.div_w);
      int ow = (int)rem - oh * p.Wout;
      int ih, iw;
      if (p.mode == 0) {
        ih = oh + r - pad;
        iw = ow + s - pad;
      } else if (p.mode == 1) {
        ih = 2 * oh + r - pad;
        iw = 2 * ow + s - pad;
      } else {  // mode 3: conv over the nearest-2x upsampled input
        int th = oh + r - 1, tw = ow + s - 1;
        ok = ok && th >= 0 && tw >= 0 && th < p.Hout && tw < p.Wout;
        ih = th >> 1;
        iw = tw >> 1;
      }
      ok = ok && ih >= 0 && iw >= 0 && ih < p.Hin && iw < p.Win;
      rb[i] = ok ? ld8(p.X + ((long)b * p.Hin * p.Win + (long)ih * p.Win + iw) * p.ldx + cc) : zero8();
    }
    mcur += TN_BM;
  };
  auto store_tile = [&](int buf) {
    char* Ab = smem + buf * TN_STAGE;
    char* Bb = Ab + TN_TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int off = (srow + 16 * i) * TN_LD + scol * 16;
      *reinterpret_cast<bf16x8*>(Ab + off) = ra[i];
      *reinterpret_cast<bf16x8*>(Bb + off) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed-read lane geometry: lane = 16g + 4q + p
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int tr_off = (4 * g + q) * TN_LD + pp * 8;

  auto compute = [&](int buf) {
    const char* Ab = smem + buf * TN_STAGE + tr_off + (wn * 64) * 2;
    const char* Bb = smem + buf * TN_STAGE + TN_TILE + tr_off + (wk * 64) * 2;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      short4v a0 = lds_tr16_b64(Ab + i * 32);
      short4v a1 = lds_tr16_b64(Ab + i * 32 + 16 * TN_LD);
      short4v b0 = lds_tr16_b64(Bb + i * 32);
      short4v b1 = lds_tr16_b64(Bb + i * 32 + 16 * TN_LD);
      typedef __attribute__((ext_vector_type(8))) short short8v;
      short8v av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
      short8v bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
      a[i] = __builtin_bit_cast(bf16x8, av);
      b[i] = __builtin_bit_cast(bf16x8, bv);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  };

  const int nsteps = (m_end - m_begin + TN_BM - 1) / TN_BM;
  if (nsteps <= 0) return;
  load_tile();
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < nsteps; ++t) {
    if (t + 1 < nsteps) load_tile();
    compute(t & 1);
    if (t + 1 < nsteps) store_tile((t + 1) & 1);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int kc = k0 + wk * 64 + j * 16 + (lane & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4 + e;
        if (p.slab) {
          const int nl = wn * 64 + i * 16 + (lane >> 4) * 4 + e, kl = wk * 64 + j * 16 + (lane & 15);
          p.slab[(((long)(tn * p.tiles_k + tk) * p.splits + split) * 128 + nl) * 128 + kl] = acc[i][j][e];
        } else if (n < p.N && kc < p.Kt) {
          float* dst = p.dW + (long)n * p.Kt + kc;
          if (p.splits == 1) *dst = p.overwrite ? acc[i][j][e] : *dst + acc[i][j][e];   // sole owner of the tile
          else unsafeAtomicAdd(dst, acc[i][j][e]);   // no workspace: order-dependent last bits
        }
      }
    }
}

// dW[n][k'] += sum_split slab[tile(n, k')][split][n % 128][k' % 128], 4 consecutive k' per thread, eight chains, fixed order
__global__ __launch_bounds__(256) void tn1_slab_reduce_kernel(GemmTNParams p) {
  const int kq = p.Kt >> 2;
  const long total = (long)p.N * kq;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int n = (int)(i / kq);
    const int kc = (int)(i - (long)n * kq) * 4;
    const int tn = n >> 7, tk = kc >> 7;
    const float* src = p.slab + (((long)(tn * p.tiles_k + tk) * p.splits) * 128 + (n & 127)) * 128 + (kc & 127);
    f32x4 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    int sp = 0;
    for (; sp + 7 < p.splits; sp += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += *reinterpret_cast<const f32x4*>(src + (long)(sp + u) * 128 * 128);
    }
    for (int u = 0; sp < p.splits; ++sp, ++u) acc[u & 7] += *reinterpret_cast<const f32x4*>(src + (long)sp * 128 * 128);
    f32x4* dst = reinterpret_cast<f32x4*>(p.dW + (long)n * p.Kt + kc);
    const f32x4 sum = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    *dst = p.overwrite ? sum : *dst + sum;
  }
}

}  // namespace

// large-tile LDS-DMA variant (gemm_tn_v2.hip)
int da_gemm_tn_v2_dispatch(int variant, const void* dY, long lddy, const void* X, long ldx, float* dW, float* dbias,
                           int M, int N, int Cin, int Hin, int Win, int Hout, int Wout, int ksize, int mode, float* ws,
                           long ws_floats, hipStream_t stream);
int da_gemm_tn_v2_fast_period(int M, int N, int Hin, int Win, int Hout, int Wout, int mode);
int g_tn_variant = 0;  // 0 auto, 1 force v1 (128x128x32), 2 force v2 (320x192x64); da_set_option
// da_set_option("grad_overwrite", 1): every gradient-PRODUCING entry point (da_gemm_tn_wgrad incl. its bias gradient,
// da_colsum_accum, da_image_colsum's db, the dgamma / dbeta of da_groupnorm_bwd / da_layernorm_bwd) writes its outputs
// instead of adding to them.  The trainer sets it for the first microbatch of a step, which then needs neither the 3.46 GB
// zero fill of the flat gradient nor the read half of 866 M read-add-writes.  0 (default) = accumulate, as the header says.
int g_grad_overwrite = 0;

static bool tn_takes_v2(int M, int N, int Kt) {
  // measured (tools/tn_ab.py, microbatch 16 / 64): the 320x192x64 kernel wins from 1,024 pixels up (1.1-2.3x), and at 256
  // pixels once the gradient itself is large (N * K' >= 8 M elements: 1.2-1.5x; below that the 128x128x32 kernel is 1.4x
  // faster)
  const bool big = (N >= 160) && (Kt >= 256) && (M >= 1024 || (M >= 256 && (long)N * Kt >= (8L << 20)));
  return g_tn_variant == 2 || g_tn_variant == 3 || (g_tn_variant == 0 && big);
}

extern "C" int da_gemm_tn_variant_for(int M, int N, int Cin, int Hin, int Win, int Hout, int Wout, int ksize, int mode) {
  if (!tn_takes_v2(M, N, ksize * ksize * Cin)) return 1;
  return da_gemm_tn_v2_fast_period(M, N, Hin, Win, Hout, Wout, mode) ? 3 : 2;
}

extern "C" int da_gemm_tn_wgrad(const void* dY, long lddy, const void* X, long ldx, float* dW, float* dbias,
                                float* scratch, int M, int N, int Cin, int Hin, int Win, int Hout, int Wout, int ksize,
                                int mode, float* split_ws, long split_ws_floats, hipStream_t stream) {
  DA_CLEAR_ERR();
  if (M <= 0 || N <= 0 || Cin <= 0) return DA_ERR_SHAPE;
  if ((N & 7) || (Cin & 7) || (lddy & 7) || (ldx & 7)) return DA_ERR_SHAPE;
  if (ksize != 1 && ksize != 3) return DA_ERR_SHAPE;
  if (mode != 0 && mode != 1 && mode != 3) return DA_ERR_SHAPE;
  if (M >= (1 << 24)) return DA_ERR_SHAPE;
  if (Hout <= 0 || Wout <= 0 || (M % (Hout * Wout))) return DA_ERR_SHAPE;
  // FastDiv (common.hpp) is exact while n * d < 2^40: pixel index / (Hout*Wout), k' index / Cin
  if (ksize == 3 && ((long)M * Hout * Wout >= (1L << 40) || 9L * Cin * Cin >= (1L << 40))) return DA_ERR_SHAPE;
  {
    const int Kt = ksize * ksize * Cin;
    if (tn_takes_v2(M, N, Kt))
      return da_gemm_tn_v2_dispatch(g_tn_variant == 2 ? 2 : 3, dY, lddy, X, ldx, dW, dbias, M, N, Cin, Hin, Win, Hout, Wout, ksize, mode,
                                    split_ws, split_ws ? split_ws_floats : 0, stream);
  }
  if (dbias) {  // small-shape path: separate fixed-order column sum
    if (!scratch) return DA_ERR_SHAPE;
    int rc = da_colsum_accum(dY, lddy, dbias, scratch, M, N, stream);
    if (rc) return rc;
  }
  GemmTNParams p;
  p.dY = (const bf16*)dY; p.X = (const bf16*)X; p.dW = dW;
  p.lddy = lddy; p.ldx = ldx;
  p.M = M; p.N = N; p.Cin = Cin; p.Kt = ksize * ksize * Cin;
  p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout; p.ksize = ksize; p.mode = mode;
  p.div_hw = make_fastdiv((unsigned)(Hout * Wout));
  p.div_w = make_fastdiv((unsigned)Wout);
  p.tiles_n = (N + 127) / 128;
  p.tiles_k = (p.Kt + 127) / 128;
  const int tiles = p.tiles_n * p.tiles_k;
  // aim for >= 1024 workgroups (4 per CU) while keeping >= 256 pixels per split
  int splits = (1024 + tiles - 1) / tiles;
  int max_splits = (M + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int mps = (M + splits - 1) / splits;
  mps = ((mps + TN_BM - 1) / TN_BM) * TN_BM;
  splits = (M + mps - 1) / mps;
  p.splits = splits;
  p.m_per_split = mps;
  // split tiles meet in a slab + fixed-order reduce when the caller passed a workspace (reproducible), else in atomics
  p.slab = nullptr;
  p.overwrite = g_grad_overwrite;
  if (splits > 1 && split_ws && (long)tiles * splits * 128 * 128 <= split_ws_floats && (p.Kt & 3) == 0) p.slab = split_ws;
  if (p.overwrite && splits > 1 && !p.slab)  // the atomic path can only add: start from zero
    if (hipMemsetAsync(dW, 0, (size_t)N * p.Kt * sizeof(float), stream) != hipSuccess) return DA_ERR_LAUNCH;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * splits), dim3(256), 0, stream, p);
  DA_CHECK_LAUNCH();
  if (p.slab) {
    long blocks = ((long)p.N * (p.Kt >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(tn1_slab_reduce_kernel, dim3((int)blocks), dim3(256), 0, stream, p);
    DA_CHECK_LAUNCH();
  }
  return DA_OK;
}
