// gemm_tn v2: large-tile LDS-DMA weight-gradient kernel (same contract as gemm_tn.hip).
//     dW[n][tap*Cin + c]  +=  sum_m  dY[m][n] * X[pixel(m) + tap][c]
// Tile: 320 (n) x 192 (k' = tap*Cin+c) fp32 accumulators per 512-thread workgroup (8 waves as 4 x 2, each
// 80 x 96 = 5 x 6 v_mfma_f32_16x16x32_bf16 tiles), 64 pixels per stage, 2 LDS stages, one workgroup per CU.  n = 320
// tiles every channel count of the SD-2 U-Net exactly and 192 divides 9*Cin; the 320 x 256 form spills.
// Both operands are staged exactly as they lie in HBM ([pixel][channel] rows) by global_load_lds_dwordx4 and
// reach the MFMA k-major through ds_read_b64_tr_b16 - issued through inline asm with counted lgkmcnt waits, because in
// front of the intrinsic the compiler waits (vmcnt(0)) for the LDS-DMA the step has just requested.  Conflict-free
// transposed reads need the 8 rows a 32-lane half touches to sit in different bank octets; LDS-DMA cannot pad rows, so
// the 16-B chunk index is XOR-ed on the SOURCE side:  dY rows (640 B = 2.5 bank rows) and X rows (384 B):
// chunk ^= 2*((row>>1)&3) (the odd half bank-row offset of odd rows supplies the third bit).  Out-of-range rows /
// columns / padding taps read a zero page.  The pixel range is split over workgroups; split tiles store fp32 slabs
// that tn_slab_reduce_kernel sums into dW (atomics only when the caller passes no workspace).
// FAST path: see the kernel's template comment.
#include "common.hpp"
#include "diffusion_amd.h"

int da_usable_cus(int cus);  // gemm_nt_v2.hip: #CUs less da_set_option("reserve_cus")
extern int g_grad_overwrite;  // gemm_tn.hip
int g_tn_ring = 0;  // da_set_option("gemm_tn_ring", 0 | 4 | 5): ring depth of the linear-layer form (gemm_tn2_kernel, RING)

// CT (template flag of the kernel): the products are taken TRANSPOSED (X fragment as the first MFMA operand): a lane's four
// accumulator registers of a tile are then four consecutive k' of one n, and the epilogue touches the 320 x 192 fp32 tile
// in 30 16-byte accesses per lane instead of 120 4-byte ones.  Same products, same sums, bit-identical dW.  Measured
// (tools/lib_ab.py tn, profiles/r03_ab_tn_ct.txt): the UNSPLIT tiles - a read-add-write of dW, the 1280-channel 3x3 layers -
// +5.5...+23.7 %; split tiles (plain slab stores) +-1 %, and -2.6...-6.8 % on the K' <= 640 linears.  So: CT = unsplit.

namespace {

struct GemmTN2Params {
  const bf16* dY;
  const bf16* X;
  float* dW;
  float* dbias;  // optional: dbias[n] += sum_m dY[m][n], produced by the tk == 0 workgroups
  long lddy, ldx;
  int M, N, Kt, Cin;
  int Hin, Win, Hout, Wout, ksize, mode;
  FastDiv div_hw, div_w, div_cin;
  int tiles_n, tiles_k, splits, m_per_split;
  float* slab;  // split > 1 with a workspace: tile partials are STORED here, [tile][split][320][BK] fp32, and summed
               // into dW by tn_slab_reduce_kernel (no atomics; fixed summation order)
  float* bslab;  // the bias-gradient partials of the same splits, [tn][split][320] fp32 (behind the tile slabs)
  int overwrite;  // da_set_option("grad_overwrite"): dW / dbias are written, not added to
  int period;  // FAST path: the border pattern of a lane's X rows repeats every `period` 64-pixel steps
  int ups64;   // FAST path, fused upsample with Wout == 64: a step is ONE output row, the source row advances every second step
};

__device__ __attribute__((aligned(256))) unsigned char g_zero_page_tn[256];

constexpr int T2_MS = 64;                 // pixels per stage
constexpr int T2_BN = 320;                // output tile rows (n)
constexpr int T2_SA = T2_BN * 2;          // 640 B per dY row
constexpr int T2_A_BYTES = T2_MS * T2_SA;  // 40 KiB
constexpr int T2_AJ = T2_A_BYTES / 1024 / 8;  // 5 DMA instructions per wave per stage

// source-side XOR of the 16-B chunk index: rows that are a whole number of 256-B bank rows need 3 bits from the
// row index; rows of an odd number of 128-B halves (640 B, 384 B) get one bit for free from the row parity
DEVINL int swA(int row) { return 2 * ((row >> 1) & 3); }
template <int BK>
DEVINL int swB(int row) { return (BK * 2) % 256 == 0 ? 2 * (row & 7) : 2 * ((row >> 1) & 3); }

DEVINL void glds16_tn(const void* gsrc, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// T2_BK: output tile columns (k'): 192 (256 also fits the swizzle scheme but spills registers).
// FAST: stride-1 gather (mode 0) with M % 64 == 0 and a border pattern that repeats every p.period <= 64 steps
// (HW % 64 == 0, or 64 % HW == 0 with period 1) - every stride-1 3x3 conv and every linear layer of the U-Net.  There the
// source pixel of a lane slot is LINEAR in the step (pixel + tap displacement), so the K loop needs no coordinate
// arithmetic: dY loads are uniform base + a constant lane offset, X loads add one select on a precomputed validity
// mask.  The generic path (strided / upsampled gathers, ragged M) recomputes coordinates each step: ~105 VALU
// instructions per step that run on every wave with the matrix pipe idle.
//
// RING (round 4; 0 = the two-stage form above): the linear layers (ksize 1, no gather, no border) stream operands that no
// other workgroup of the XCD has just fetched, and with ONE 64-pixel stage in flight behind a vmcnt(0) barrier a step lasts
// as long as a loaded HBM round trip (2.7-2.9 us against 0.8 us of MFMA: 16384x1280x1280 ran 76 us for 24 us of matrix
// work).  This form keeps RING - 1 half-stages (32 pixels = one MFMA K-step, 32 KB) in flight in a ring of RING slots
// behind COUNTED vmcnt waits: waves 0-4 request the 20 dY pieces of a half, waves 5-7 the 12 X pieces (4 per wave, so one
// count fits all), one barrier per half.  Same products in the same order: bit-identical to RING = 0.
template <int T2_BK, bool FAST, bool CT, int RING = 0>
__global__ __launch_bounds__(512, 2) void gemm_tn2_kernel(GemmTN2Params p) {
  constexpr int T2_SB = T2_BK * 2;
  constexpr int T2_B_BYTES = T2_MS * T2_SB;
  constexpr int T2_STAGE = T2_A_BYTES + T2_B_BYTES;
  constexpr int T2_BJ = T2_B_BYTES / 1024 / 8;
  constexpr int C16B = T2_SB / 16;  // 16-B chunks per X row
  constexpr int JT = T2_BK / 32;    // 16-column MFMA tiles per wave along k'
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave >> 1, wb = wave & 1;  // 4 waves along n (80 each) x 2 along k' (128 each)

  // XCD-aware order: workgroups b, b+8, ... share an XCD (and its L2).  Each XCD gets a contiguous run of ids, decoded
  // k'-tile fastest, so the workgroups running together on an XCD stream the SAME dY rows (same n-tile, same pixel
  // range) and the same X rows (other taps / channel tiles of them) - the L2 serves most of both operands.
  int bid = blockIdx.x;
  {
    const int nblk = p.tiles_n * p.tiles_k * p.splits;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tk = bid % p.tiles_k;
  bid /= p.tiles_k;
  const int split = bid % p.splits;
  const int tn = bid / p.splits;
  const int n0 = tn * T2_BN, k0 = tk * T2_BK;
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int nsteps = (m_end - m_begin + T2_MS - 1) / T2_MS;
  if (nsteps <= 0) return;
  const int HWo = p.Hout * p.Wout;
  const int pad = (p.ksize == 3) ? 1 : 0;
  const int gmul = (p.mode == 1) ? 2 : 1, gshift = (p.mode == 3) ? 1 : 0;
  const int hlim = (p.mode == 3) ? p.Hout : p.Hin, wlim = (p.mode == 3) ? p.Wout : p.Win;
  const char* zero = reinterpret_cast<const char*>(g_zero_page_tn);

  int mcur_f = m_begin;
  // ---- FAST-path descriptors (see the template comment); the generic ones follow
  unsigned fa_off[T2_AJ], fx_off[T2_BJ], fx_off1[T2_BJ], fx_m0[T2_BJ], fx_m1[T2_BJ];
  // source pixels per 64-pixel step and the (uniform) source element offset of step 0's first image row block
  const int src_step = (int)((long)T2_MS * p.Hin * p.Win / HWo);
  // the uniform base sits pad*(Win+1) source pixels below the block's first one, so every lane offset is >= 0
  const long x_margin = (long)pad * (p.Win + 1);
  const long x_base0 = ((long)(m_begin / T2_MS) * src_step - x_margin) * p.ldx;
  if constexpr (FAST && RING == 0) {
#pragma unroll
    for (int j = 0; j < T2_AJ; ++j) {
      const int ci = (wave * T2_AJ + j) * 64 + lane;
      const int row = ci / 40, pc = ci - row * 40;
      const int lc = pc ^ swA(row);
      const int n = min(n0 + lc * 8, p.N - 8);  // columns past N: clamped (their outputs are never stored)
      fa_off[j] = (unsigned)(row * (int)p.lddy + n);
    }
#pragma unroll
    for (int j = 0; j < T2_BJ; ++j) {
      const int ci = (wave * T2_BJ + j) * 64 + lane;
      const int row = ci / C16B, pc = ci - row * C16B;
      const int lc = pc ^ swB<T2_BK>(row);
      const int kk = k0 + lc * 8;
      const bool kok = kk < p.Kt;
      const unsigned tap = kok ? fdiv((unsigned)kk, p.div_cin) : 0u;
      const int c = kok ? kk - (int)tap * p.Cin : 0;
      const int r = (int)tap / 3, s2 = (int)tap - 3 * r;  // ksize 1: tap == 0
      // Source element of this slot at step 0 by the general gather rule (stride 1 / 2, fused nearest-2x upsample),
      // relative to X; later steps add a uniform stride (a 64-pixel step covers whole output rows, so the source
      // advances by 64 * Hin*Win / (Hout*Wout) pixels).  Out-of-image taps are masked below.
      unsigned m0 = 0, m1 = 0;
      {
        const unsigned mm = (unsigned)(m_begin + row);
        const unsigned bb = fdiv(mm, p.div_hw);
        const unsigned rem = mm - bb * (unsigned)HWo;
        const int oh = (int)fdiv(rem, p.div_w);
        const int ow = (int)rem - oh * p.Wout;
        // signed, NOT clamped (the slot is valid again at later steps and must keep a linear offset); >> floors
        const int ih = (oh * gmul + r - pad) >> gshift, iw = (ow * gmul + s2 - pad) >> gshift;
        fx_off[j] = (unsigned)(((long)bb * p.Hin * p.Win + (long)ih * p.Win + iw) * p.ldx + c - x_base0);
        fx_off1[j] = fx_off[j];
        if (p.ups64) {
          // one output row per step: source row (oh + r - 1) >> 1 = (oh >> 1) + cpar with cpar = (b + r - 1) >> 1 for the row's
          // parity b - two lane offsets relative to the uniform base (oh >> 1) * Win - margin, selected by the step's parity
          fx_off[j] = (unsigned)((long)((((0 + r - 1) >> 1) + 1) * p.Win + iw + 1) * p.ldx + c);
          fx_off1[j] = (unsigned)((long)((((1 + r - 1) >> 1) + 1) * p.Win + iw + 1) * p.ldx + c);
        }
      }
      if (kok) {
        if (p.ksize == 1 && p.mode == 0) {
          m0 = m1 = 0xffffffffu;
        } else {
          for (int ph = 0; ph < p.period; ++ph) {
            const unsigned mm = (unsigned)(m_begin + row + T2_MS * ph);
            const unsigned rem = mm - fdiv(mm, p.div_hw) * (unsigned)HWo;
            const int oh = (int)fdiv(rem, p.div_w);
            const int ow = (int)rem - oh * p.Wout;
            const int th = oh * gmul + r - pad, tw = ow * gmul + s2 - pad;
            const bool ok = (unsigned)th < (unsigned)hlim && (unsigned)tw < (unsigned)wlim;
            if (ph < 32) m0 |= (ok ? 1u : 0u) << ph;
            else m1 |= (ok ? 1u : 0u) << (ph - 32);
          }
        }
      }
      fx_m0[j] = m0;
      fx_m1[j] = m1;
    }
  }
  int phase = 0;
  auto issue_fast = [&](int stage, bool live) {
    char* Ab = smem + stage * T2_STAGE;
    char* Bb = Ab + T2_A_BYTES;
    const int mc = live ? mcur_f : m_begin;  // the step after the last re-reads step 0 (X rows masked off)
    const bf16* ab = p.dY + (long)mc * p.lddy;
#pragma unroll
    for (int j = 0; j < T2_AJ; ++j) glds16_tn(ab + fa_off[j], Ab + (wave * T2_AJ + j) * 1024);
    const int grow = mc / T2_MS;  // ups64: the global output row of this step
    const bf16* xb = p.X + ((p.ups64 ? (long)(grow >> 1) * p.Win : (long)grow * src_step) - x_margin) * p.ldx;
    const bool odd_row = p.ups64 && (grow & 1);
    const unsigned pb0 = (live && phase < 32) ? (1u << phase) : 0u;
    const unsigned pb1 = (live && phase >= 32) ? (1u << (phase - 32)) : 0u;
#pragma unroll
    for (int j = 0; j < T2_BJ; ++j) {
      const bool ok = ((fx_m0[j] & pb0) | (fx_m1[j] & pb1)) != 0;
      const void* src = ok ? (const void*)(xb + (odd_row ? fx_off1[j] : fx_off[j])) : (const void*)zero;
      glds16_tn(src, Bb + (wave * T2_BJ + j) * 1024);
    }
    mcur_f += T2_MS;
    phase = (phase + 1 == p.period) ? 0 : phase + 1;
  };

  // ---- DMA source descriptors (fixed per lane and instruction slot for the whole kernel)
  // dY image: chunk ci = (wave*AJ + j)*64 + lane ; row = ci / 40 ; physical chunk pc = ci % 40 ; logical lc = pc ^ swA(row)
  int a_row[T2_AJ];
  int a_off[T2_AJ];
  bool a_ok[T2_AJ];
#pragma unroll
  for (int j = 0; j < T2_AJ; ++j) {
    const int ci = (wave * T2_AJ + j) * 64 + lane;
    const int row = ci / 40, pc = ci - row * 40;
    const int lc = pc ^ swA(row);
    const int n = n0 + lc * 8;
    a_row[j] = row;
    a_ok[j] = n < p.N;
    a_off[j] = row * (int)p.lddy + n;
  }
  // X image: row = ci >> 5 ; pc = ci & 31 ; lc = pc ^ swB(row) ; k' = k0 + lc*8 -> (tap, c)
  int b_row[T2_BJ], b_dr[T2_BJ], b_ds[T2_BJ], b_c[T2_BJ];
  bool b_ok[T2_BJ];
#pragma unroll
  for (int j = 0; j < T2_BJ; ++j) {
    const int ci = (wave * T2_BJ + j) * 64 + lane;
    const int row = ci / C16B, pc = ci - row * C16B;
    const int lc = pc ^ swB<T2_BK>(row);
    const int kk = k0 + lc * 8;
    b_row[j] = row;
    b_ok[j] = kk < p.Kt;
    const unsigned tap = b_ok[j] ? fdiv((unsigned)kk, p.div_cin) : 0u;
    b_c[j] = b_ok[j] ? kk - (int)tap * p.Cin : 0;
    int r = 0, s = 0;
    if (p.ksize == 3) {
      r = (int)tap / 3;
      s = (int)tap - 3 * r;
    }
    b_dr[j] = r;
    b_ds[j] = s;
  }

  // pixel coordinates of each X-chunk's row, advanced incrementally by 64 rows per stage (no per-step division)
  int x_b[T2_BJ], x_oh[T2_BJ], x_ow[T2_BJ];
#pragma unroll
  for (int j = 0; j < T2_BJ; ++j) {
    const unsigned mm = (unsigned)(m_begin + b_row[j]);
    const unsigned bb = fdiv(mm, p.div_hw);
    const unsigned rem = mm - bb * (unsigned)HWo;
    x_b[j] = (int)bb;
    x_oh[j] = (int)fdiv(rem, p.div_w);
    x_ow[j] = (int)rem - x_oh[j] * p.Wout;
  }
  const int step_ow = T2_MS % p.Wout, step_oh = (T2_MS / p.Wout) % p.Hout, step_b = T2_MS / HWo;

  int mcur = m_begin;
  auto issue = [&](int stage) {
    char* Ab = smem + stage * T2_STAGE;
    char* Bb = Ab + T2_A_BYTES;
#pragma unroll
    for (int j = 0; j < T2_AJ; ++j) {
      const int m = mcur + a_row[j];
      const void* src = (a_ok[j] && m < m_end) ? (const void*)(p.dY + (long)mcur * p.lddy + a_off[j]) : (const void*)zero;
      glds16_tn(src, Ab + (wave * T2_AJ + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < T2_BJ; ++j) {
      const int m = mcur + b_row[j];
      bool ok = b_ok[j] && m < m_end;
      const int b = x_b[j], oh = x_oh[j], ow = x_ow[j];
      x_ow[j] += step_ow;
      if (x_ow[j] >= p.Wout) { x_ow[j] -= p.Wout; x_oh[j] += 1; }
      x_oh[j] += step_oh;
      if (x_oh[j] >= p.Hout) { x_oh[j] -= p.Hout; x_b[j] += 1; }
      x_b[j] += step_b;
      // branch-free tap geometry: stride-2 (mode 1) multiplies, the fused upsample (mode 3) shifts
      const int th = oh * gmul + b_dr[j] - pad, tw = ow * gmul + b_ds[j] - pad;
      ok = ok && (unsigned)th < (unsigned)hlim && (unsigned)tw < (unsigned)wlim;
      const int ih = th >> gshift, iw = tw >> gshift;
      const void* src =
          ok ? (const void*)(p.X + ((long)b * p.Hin * p.Win + (long)ih * p.Win + iw) * p.ldx + b_c[j]) : (const void*)zero;
      glds16_tn(src, Bb + (wave * T2_BJ + j) * 1024);
    }
    mcur += T2_MS;
  };

  f32x4 acc[5][JT];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < JT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  typedef __attribute__((ext_vector_type(8))) short short8v;
  // bias gradient = dY^T . 1 : one extra MFMA per dY fragment against an all-ones B fragment (waves wb == 0 of
  // the tk == 0 workgroups only), instead of a separate column-sum pass over dY
  const bool do_bias = (p.dbias != nullptr) && (tk == 0) && (wb == 0);
  f32x4 accb[5];
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
#pragma unroll
  for (int i = 0; i < 5; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // All transposed reads of a half-step are issued up front through the asm form (common.hpp: the intrinsic would
  // make the compiler wait for the DMA just issued), then two counted waits: the dY fragments + the first half of the
  // X fragments, 15 (+5 bias) MFMAs, the second half, 15 MFMAs.
  const unsigned smem_off = lds_offset(smem);
  auto compute_half_at = [&](const unsigned Ab, const unsigned Bb, int ms) {
    const int r0 = ms * 32 + 4 * g + q, r1 = r0 + 16;
    short4v ta0[5], ta1[5], tb0[JT], tb1[JT];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int ch = wa * 10 + i * 2 + (pp >> 1);
      ta0[i] = lds_tr16_b64_asm(Ab + r0 * T2_SA + ((ch ^ swA(r0)) << 4) + 8 * (pp & 1));
      ta1[i] = lds_tr16_b64_asm(Ab + r1 * T2_SA + ((ch ^ swA(r1)) << 4) + 8 * (pp & 1));
    }
    auto read_b = [&](int jj) {
      const int ch = wb * (JT * 2) + jj * 2 + (pp >> 1);
      tb0[jj] = lds_tr16_b64_asm(Bb + r0 * T2_SB + ((ch ^ swB<T2_BK>(r0)) << 4) + 8 * (pp & 1));
      tb1[jj] = lds_tr16_b64_asm(Bb + r1 * T2_SB + ((ch ^ swB<T2_BK>(r1)) << 4) + 8 * (pp & 1));
    };
    static_assert(JT == 6, "counted waits below assume 6 X fragments per wave");
    // FAST has the registers to request all six X fragments at once; the generic path (more live gather state)
    // requests the second three only after the first MFMA block, which keeps it free of spills
#pragma unroll
    for (int jj = 0; jj < (FAST ? JT : JT / 2); ++jj) read_b(jj);
    if constexpr (FAST)
      lds_wait_for<6>(ta0[0], ta1[0], ta0[1], ta1[1], ta0[2], ta1[2], ta0[3], ta1[3], ta0[4], ta1[4], tb0[0], tb1[0],
                      tb0[1], tb1[1], tb0[2], tb1[2]);
    else
      lds_wait_for<0>(ta0[0], ta1[0], ta0[1], ta1[1], ta0[2], ta1[2], ta0[3], ta1[3], ta0[4], ta1[4], tb0[0], tb1[0],
                      tb0[1], tb1[1], tb0[2], tb1[2]);
    bf16x8 a[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      short8v v = __builtin_shufflevector(ta0[i], ta1[i], 0, 1, 2, 3, 4, 5, 6, 7);
      a[i] = __builtin_bit_cast(bf16x8, v);
    }
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 5; ++i)
        accb[i] = CT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a[i], accb[i], 0, 0, 0)
                     : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i], 0, 0, 0);
    }
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
      if (jh == 1) {
        if constexpr (!FAST) {
#pragma unroll
          for (int jj = JT / 2; jj < JT; ++jj) read_b(jj);
        }
        lds_wait_for<0>(tb0[3], tb1[3], tb0[4], tb1[4], tb0[5], tb1[5]);
      }
      bf16x8 bfr[JT / 2];
#pragma unroll
      for (int jj = 0; jj < JT / 2; ++jj) {
        short8v v = __builtin_shufflevector(tb0[jh * (JT / 2) + jj], tb1[jh * (JT / 2) + jj], 0, 1, 2, 3, 4, 5, 6, 7);
        bfr[jj] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int jj = 0; jj < JT / 2; ++jj)
          acc[i][jh * (JT / 2) + jj] =
              CT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[jj], a[i], acc[i][jh * (JT / 2) + jj], 0, 0, 0)
                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bfr[jj], acc[i][jh * (JT / 2) + jj], 0, 0, 0);
    }
  };

  auto compute_half = [&](int stage, int ms) {
    const unsigned Ab = smem_off + stage * T2_STAGE;
    compute_half_at(Ab, Ab + T2_A_BYTES, ms);
  };

  if constexpr (RING > 0) {
    static_assert(!RING || (FAST && T2_BK == 192), "ring form: linear layers on the FAST path, 4 pieces per wave and half");
    constexpr int HALF_A = 32 * T2_SA, HALF = HALF_A + 32 * T2_SB;  // 20 KB of dY + 12 KB of X = 32 pixels
    const int nh = (m_end - m_begin) >> 5;  // FAST: M % 64 == 0 and splits are whole stages
    const bool is_a = wave < 5;
    unsigned off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (is_a) {
        const int ci = (wave * 4 + j) * 64 + lane;
        const int row = ci / 40, pc = ci - row * 40;
        const int n = min(n0 + (pc ^ swA(row)) * 8, p.N - 8);  // columns past N / Kt: clamped (never stored / never reduced)
        off[j] = (unsigned)(row * (int)p.lddy + n);
      } else {
        const int ci = ((wave - 5) * 4 + j) * 64 + lane;
        const int row = ci / C16B, pc = ci - row * C16B;
        const int kk = min(k0 + (pc ^ swB<T2_BK>(row)) * 8, p.Kt - 8);
        off[j] = (unsigned)(row * (int)p.ldx + kk);
      }
    }
    const bf16* src = is_a ? p.dY + (long)m_begin * p.lddy : p.X + (long)m_begin * p.ldx;
    const long hstride = 32 * (is_a ? p.lddy : p.ldx);
    char* dst0 = smem + (is_a ? wave * 4096 : HALF_A + (wave - 5) * 4096);
    int h_issue = 0, slot_issue = 0;
    auto issue_half = [&]() {
      const bf16* base = src + (h_issue < nh ? h_issue : 0) * hstride;  // past the end: half 0 again, into a free slot
      char* dst = dst0 + slot_issue * HALF;
#pragma unroll
      for (int j = 0; j < 4; ++j) glds16_tn(base + off[j], dst + j * 1024);
      ++h_issue;
      slot_issue = slot_issue + 1 == RING ? 0 : slot_issue + 1;
    };
#pragma unroll
    for (int h = 0; h < RING - 1; ++h) issue_half();
    int slot = 0;
    for (int h = 0; h < nh; ++h) {
      // my pieces of half h have landed (RING - 2 younger halves of 4 pieces stay in flight); behind the barrier
      // everybody's have, and everybody has read half h - 1 out of the slot the next request overwrites
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (RING - 2)) : "memory");
      issue_half();
      const unsigned Ab = smem_off + slot * HALF;
      compute_half_at(Ab, Ab + HALF_A, 0);
      slot = slot + 1 == RING ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail requests write LDS: not past the end of the workgroup
  } else if constexpr (FAST) {
    issue_fast(0, true);
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {  // one basic block: no branch around the issue
      // Placement of the request, measured (tools/tn_ab.py, batch 256): here for every wave; between the halves -7..+2 % by
      // shape; the two waves of a SIMD (w, w + 4) taking turns - one here, one between its halves, so that one's request
      // block (~900-1300 cycles for its 8 pieces) would run under the other's MFMAs - -6..-14 %; turns by wave parity +-2 %.
      issue_fast((t + 1) & 1, t + 1 < nsteps);
      compute_half(t & 1, 0);
      compute_half(t & 1, 1);
      __syncthreads();
    }
  } else {
    issue(0);
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
      compute_half(t & 1, 0);
      if (t + 1 < nsteps) issue((t + 1) & 1);
      compute_half(t & 1, 1);
      __syncthreads();
    }
  }

  if constexpr (CT) {
    // accumulator register e of tile (i, j): n = wa*80 + i*16 + (lane & 15), k' = wb*16*JT + j*16 + 4*(lane >> 4) + e
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int nl = wa * 80 + i * 16 + (lane & 15), kl = wb * (16 * JT) + j * 16 + (lane >> 4) * 4;
        const int n = n0 + nl, kc = k0 + kl;
        if (p.slab && p.splits > 1) {
          *reinterpret_cast<f32x4*>(p.slab + (((long)(tn * p.tiles_k + tk) * p.splits + split) * T2_BN + nl) * T2_BK + kl) = acc[i][j];
        } else if (n < p.N && kc < p.Kt) {  // Kt % 8 == 0 and kc % 4 == 0: the four k' are inside together
          float* dst = p.dW + (long)n * p.Kt + kc;
          if (p.splits == 1) {  // sole owner of this tile: plain write / read-add-write
            if (p.overwrite) *reinterpret_cast<f32x4*>(dst) = acc[i][j];
            else *reinterpret_cast<f32x4*>(dst) += acc[i][j];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + e, acc[i][j][e]);
          }
        }
      }
    if (do_bias && lane < 16) {  // every register of accb[i] holds the column sum of n = wa*80 + i*16 + (lane & 15)
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int nl = wa * 80 + i * 16 + lane, n = n0 + nl;
        if (p.slab && p.splits > 1) p.bslab[((long)tn * p.splits + split) * T2_BN + nl] = accb[i][0];  // summed by the reduce kernel
        else if (n < p.N) {
          if (p.splits == 1) p.dbias[n] = p.overwrite ? accb[i][0] : p.dbias[n] + accb[i][0];  // the only workgroup with this (tn, tk == 0)
          else unsafeAtomicAdd(p.dbias + n, accb[i][0]);  // no workspace: order-dependent last bits (as dW above)
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int kc = k0 + wb * (16 * JT) + j * 16 + (lane & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + wa * 80 + i * 16 + (lane >> 4) * 4 + e;
        if (p.slab && p.splits > 1) {
          // fp32 atomics retire at ~1 dword / clk / L2 channel (~48 us for the ~16 M of a 256-workgroup split grid):
          // store the partial tile instead and let the reduce kernel add the splits
          const int nl = wa * 80 + i * 16 + (lane >> 4) * 4 + e, kl = wb * (16 * JT) + j * 16 + (lane & 15);
          p.slab[(((long)(tn * p.tiles_k + tk) * p.splits + split) * T2_BN + nl) * T2_BK + kl] = acc[i][j][e];
        } else if (n < p.N && kc < p.Kt) {
          float* dst = p.dW + (long)n * p.Kt + kc;
          if (p.splits == 1) *dst = p.overwrite ? acc[i][j][e] : *dst + acc[i][j][e];  // sole owner of this tile
          else unsafeAtomicAdd(dst, acc[i][j][e]);
        }
      }
    }
  if (do_bias && (lane & 15) == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int nl = wa * 80 + i * 16 + (lane >> 4) * 4 + e, n = n0 + nl;
        if (p.slab && p.splits > 1) p.bslab[((long)tn * p.splits + split) * T2_BN + nl] = accb[i][e];  // summed by the reduce kernel
        else if (n < p.N) {
          if (p.splits == 1) p.dbias[n] = p.overwrite ? accb[i][e] : p.dbias[n] + accb[i][e];  // the only workgroup with this (tn, tk == 0)
          else unsafeAtomicAdd(p.dbias + n, accb[i][e]);  // no workspace: order-dependent last bits (as dW above)
        }
      }
  }
}

// dW[n][k'] += sum_split slab[tile(n, k')][split][n % 320][k' % BK]   (4 consecutive k' per thread): blocks [0, main_blocks).
// Blocks behind them (one per 16 bias elements, resident beside the others) reduce the bias-gradient partials:
// dbias[n] += sum_split bslab[tn][split][n % 320]; 16 lanes share an n (splits l, l + 16, ... on two chains), then a fixed
// butterfly - the same additions in the same order every run.
template <int BK>
__global__ __launch_bounds__(256) void tn_slab_reduce_kernel(GemmTN2Params p, int main_blocks) {
  if ((int)blockIdx.x >= main_blocks) {
    const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int n = ((int)blockIdx.x - main_blocks) * 16 + grp;
    if (n < p.N) {
      const int tn = n / T2_BN;
      const float* src = p.bslab + ((long)tn * p.splits) * T2_BN + (n - tn * T2_BN);
      float a0 = 0.f, a1 = 0.f;
      int sp = l;
      for (; sp + 16 < p.splits; sp += 32) {
        a0 += src[(long)sp * T2_BN];
        a1 += src[(long)(sp + 16) * T2_BN];
      }
      if (sp < p.splits) a0 += src[(long)sp * T2_BN];
      float a = a0 + a1;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o, 16);
      if (l == 0) p.dbias[n] = p.overwrite ? a : p.dbias[n] + a;
    }
    return;
  }
  const int kq = p.Kt >> 2;
  const long total = (long)p.N * kq;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)main_blocks * 256) {
    const int n = (int)(i / kq);
    const int kc = (int)(i - (long)n * kq) * 4;
    const int tn = n / T2_BN, tk = kc / BK;
    const float* src = p.slab + (((long)(tn * p.tiles_k + tk) * p.splits) * T2_BN + (n - tn * T2_BN)) * BK + (kc - tk * BK);
    // eight independent partial sums: up to 128 splits per element, and with two loads in flight the walk was
    // latency-bound (23 us for a 320 x 320 gradient); the order of the final adds is fixed, so dW stays reproducible
    f32x4 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    int sp = 0;
    for (; sp + 7 < p.splits; sp += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += *reinterpret_cast<const f32x4*>(src + (long)(sp + u) * T2_BN * BK);
    }
    for (int u = 0; sp < p.splits; ++sp, ++u) acc[u & 7] += *reinterpret_cast<const f32x4*>(src + (long)sp * T2_BN * BK);
    f32x4* dst = reinterpret_cast<f32x4*>(p.dW + (long)n * p.Kt + kc);
    const f32x4 sum = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    *dst = p.overwrite ? sum : *dst + sum;
  }
}

template <int BK, bool FAST>
int launch_tn2(GemmTN2Params p, float* ws, long ws_floats, hipStream_t stream) {
  constexpr int SMEM = 2 * (T2_A_BYTES + T2_MS * BK * 2);
  p.tiles_n = (p.N + T2_BN - 1) / T2_BN;
  p.tiles_k = (p.Kt + BK - 1) / BK;
  const int tiles = p.tiles_n * p.tiles_k;
  // One 512-thread workgroup per CU; the pixel range is split into k parts so that tiles*k fills the CUs (256, less the
  // ones da_set_option("reserve_cus") leaves to an overlapping collective).
  //  * tiles >= 90 % of the CUs: k = 1 - the workgroup owns its tile, plain read-add-write, no slabs;
  //  * otherwise the smallest k whose grid is (nearly) a whole number of rounds (last round >= 90 % full),
  //    capped so that the k extra fp32 tile writes stay below ~20 % of the GEMM time
  //    (k <= M/6000) but never below one full round; fallback: the fullest grid within the cap.
  const int ncu = da_usable_cus(256);
  int best = 1;
  if (tiles < (ncu * 29) / 32) {
    int kcap = p.M / 6000;
    const int one_round = (ncu + tiles - 1) / tiles;
    if (kcap < one_round) kcap = one_round;
    const int max_splits = p.M / 512 > 0 ? p.M / 512 : 1;
    if (kcap > max_splits) kcap = max_splits;
    double best_eff = -1.0;
    bool found = false;
    for (int k = 1; k <= kcap; ++k) {
      const long blocks = (long)tiles * k;
      const double eff = (double)blocks / (double)(((blocks + ncu - 1) / ncu) * ncu);
      if (eff >= 0.9 && blocks >= ncu - 6) {
        best = k;
        found = true;
        break;
      }
      if (eff >= best_eff - 1e-9) {  // ties -> more, shorter workgroups
        best_eff = eff;
        best = k;
      }
    }
    (void)found;
  }
  int mps = (p.M + best - 1) / best;
  mps = ((mps + T2_MS - 1) / T2_MS) * T2_MS;
  p.splits = (p.M + mps - 1) / mps;
  p.m_per_split = mps;
  static unsigned long long attr_done = 0, attr_done_ct = 0;  // one bit per device
  if (da_ensure_dyn_smem((const void*)gemm_tn2_kernel<BK, FAST, false>, SMEM, &attr_done) != DA_OK) return DA_ERR_LAUNCH;
  if (da_ensure_dyn_smem((const void*)gemm_tn2_kernel<BK, FAST, true>, SMEM, &attr_done_ct) != DA_OK) return DA_ERR_LAUNCH;
  p.slab = p.bslab = nullptr;
  {
    const long tile_floats = (long)tiles * p.splits * T2_BN * BK;
    const long bias_floats = p.dbias ? (long)p.tiles_n * p.splits * T2_BN : 0;
    if (p.splits > 1 && ws && tile_floats + bias_floats <= ws_floats && (p.Kt & 3) == 0) {
      p.slab = ws;
      p.bslab = ws + tile_floats;
    }
  }
  p.overwrite = g_grad_overwrite;
  if (p.overwrite && p.splits > 1 && !p.slab) {  // the atomic path can only add: start from zero
    if (hipMemsetAsync(p.dW, 0, (size_t)p.N * p.Kt * sizeof(float), stream) != hipSuccess) return DA_ERR_LAUNCH;
    if (p.dbias && hipMemsetAsync(p.dbias, 0, (size_t)p.N * sizeof(float), stream) != hipSuccess) return DA_ERR_LAUNCH;
  }
  bool ring_done = false;
  if constexpr (FAST && BK == 192) {
    // the ring form (see the kernel's template comment): linear layers only
    if (g_tn_ring && p.ksize == 1 && p.mode == 0 && p.Hin == p.Hout && p.Win == p.Wout && p.N % 8 == 0 && p.Kt % 8 == 0 && p.N >= 8 &&
        p.Kt >= 8 && 64 * p.lddy < (1L << 31) && 64 * p.ldx < (1L << 31)) {
      static unsigned long long ring_attr[4] = {0, 0, 0, 0};
      const dim3 grid(p.splits == 1 ? tiles : tiles * p.splits);
#define TN2_RING(R, CTV, slot)                                                                                              \
  {                                                                                                                       \
    if (da_ensure_dyn_smem((const void*)gemm_tn2_kernel<BK, FAST, CTV, R>, R * 32768, &ring_attr[slot]) != DA_OK) return DA_ERR_LAUNCH; \
    hipLaunchKernelGGL((gemm_tn2_kernel<BK, FAST, CTV, R>), grid, dim3(512), R * 32768, stream, p);                          \
  }
      if (g_tn_ring >= 5) {
        if (p.splits == 1) TN2_RING(5, true, 0) else TN2_RING(5, false, 1)
      } else {
        if (p.splits == 1) TN2_RING(4, true, 2) else TN2_RING(4, false, 3)
      }
#undef TN2_RING
      ring_done = true;
    }
  }
  if (!ring_done) {
    if (p.splits == 1) hipLaunchKernelGGL((gemm_tn2_kernel<BK, FAST, true>), dim3(tiles), dim3(512), SMEM, stream, p);
    else hipLaunchKernelGGL((gemm_tn2_kernel<BK, FAST, false>), dim3(tiles * p.splits), dim3(512), SMEM, stream, p);
  }
  DA_CHECK_LAUNCH();
  if (p.slab) {
    const long total = (long)p.N * (p.Kt >> 2);
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const int bias_blocks = p.dbias ? (p.N + 15) / 16 : 0;
    hipLaunchKernelGGL(tn_slab_reduce_kernel<BK>, dim3((int)blocks + bias_blocks), dim3(256), 0, stream, p, (int)blocks);
    DA_CHECK_LAUNCH();
  }
  return DA_OK;
}

}  // namespace

// FAST path of the 320x192x64 wgrad kernel: M a multiple of the 64-pixel step, a border pattern that repeats within
// 64 steps, and steps that cover whole output rows of the gather (so the source pixel advances uniformly): stride 1
// always, stride 2 when 64 % Wout == 0, fused upsample when 64 % (2*Wout) == 0.  Returns the period of the border mask
// in steps (0: not eligible, the generic gather runs).
int da_gemm_tn_v2_fast_period(int M, int N, int Hin, int Win, int Hout, int Wout, int mode) {
  const int HWo = Hout * Wout;
  const bool rows_ok = mode == 0 || (mode == 1 && T2_MS % Wout == 0) || (mode == 3 && T2_MS % (2 * Wout) == 0) ||
                       (mode == 3 && Wout == T2_MS && Hout % 2 == 0 && 2 * Hin == Hout && 2 * Win == Wout);  // ups64
  if (rows_ok && M % T2_MS == 0 && N >= 8 && ((long)T2_MS * Hin * Win) % HWo == 0) {
    if (HWo % T2_MS == 0 && HWo / T2_MS <= 64) return HWo / T2_MS;
    if (T2_MS % HWo == 0) return 1;
  }
  return 0;
}

// Called by da_gemm_tn_wgrad (gemm_tn.hip) after argument validation.
int da_gemm_tn_v2_dispatch(int variant, const void* dY, long lddy, const void* X, long ldx, float* dW, float* dbias,
                           int M, int N, int Cin, int Hin, int Win, int Hout, int Wout, int ksize, int mode, float* ws,
                           long ws_floats, hipStream_t stream) {
  GemmTN2Params p;
  p.dY = (const bf16*)dY; p.X = (const bf16*)X; p.dW = dW; p.dbias = dbias;
  p.lddy = lddy; p.ldx = ldx;
  p.M = M; p.N = N; p.Cin = Cin; p.Kt = ksize * ksize * Cin;
  p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout; p.ksize = ksize; p.mode = mode;
  p.div_hw = make_fastdiv((unsigned)(Hout * Wout));
  p.div_w = make_fastdiv((unsigned)Wout);
  p.div_cin = make_fastdiv((unsigned)Cin);
  p.tiles_n = p.tiles_k = p.splits = p.m_per_split = 0;
  p.slab = p.bslab = nullptr;
  p.overwrite = 0;
  (void)variant;  // the 320x256 instantiation (160 accumulators) spills on gfx950 and lost to 320x192 everywhere
  p.period = da_gemm_tn_v2_fast_period(M, N, Hin, Win, Hout, Wout, mode);
  p.ups64 = (p.period && mode == 3 && Wout == T2_MS) ? 1 : 0;
  return p.period ? launch_tn2<192, true>(p, ws, ws_floats, stream) : launch_tn2<192, false>(p, ws, ws_floats, stream);
}
