// gemm_nt v2: large-tile LDS-DMA variant of the implicit-GEMM "NT" kernel (same contract as gemm_nt.hip).
//
// Why: the 128x128x64 tile of v1 moves 32 KiB from L2 per 2.1 MFLOP (64 FLOP/B); at the bf16 MFMA peak that is
// ~39 TB/s of L2->LDS traffic, more than the chip has, and v1 measures ~0.5 PFLOP/s on every large shape.
// v2 is one template over (MFMA tiles per wave, wave grid, K-step): the default large form is 256 x 320 x 64 with 16
// waves (4 x 4, each 64 x 80 = 4 x 5 v_mfma_f32_16x16x32_bf16, <= 128 VGPRs, 4 waves per SIMD, 142 FLOP/B); 8-wave
// forms 256 x (320|160|128) x 64 and a 4-wave 128 x 320 x 32 form exist for small / odd shapes and as references.
// BN = 320 / 160 make every channel count of the SD-2 U-Net (multiples of 320) tile exactly.  One workgroup per CU,
// LDS filled by global_load_lds_dwordx4 (no staging registers, no ds_write pass).
// LDS-DMA writes lane-linearly (1 KiB = 8 rows x 128 B per wave instruction), so the XOR swizzle that keeps the
// ds_read_b128 fragment reads conflict-free is applied on the per-lane SOURCE address; out-of-image taps and
// out-of-range rows read a 128-B zero page instead of being predicated.
// Two LDS stages; the loads of K-step t+1 are issued during step t (at its start for 1x1 shapes, between its two MFMA
// halves for 3x3) and retired by the vmcnt(0)+barrier that ends the step.  Requires Cin % 64 == 0 (one tap per K-step).
// Epilogue variants: plain (bias / per-image row bias / residual), GEGLU forward, GEGLU backward, split-K slabs.
#include "common.hpp"
#include "diffusion_amd.h"
// NT2_NT=1 (compile-time experiment, round 4; default 0): the output rows of the short-K forms leave with the NON-TEMPORAL
// hint (global_store_dwordx4 ... nt).  In isolation (tools/lib_ab.py, the same launch repeated; profiles/r04_ab_nt_ntstore.txt)
// 262144 x 960 x 320 ran +20.8 %, x 320 x 320 +4...8 %, the fused GEGLU forward +5...12 % (3 x 3 convolutions -1...-3 %, GEGLU
// backward -3 %) - and the whole training step did not move (169.4 / 168.6 vs 169.1 / 169.1 ms, two interleaved pairs on one
// box): inside the step the consumer of those rows is the next launch, which finds a good part of a plainly stored 168 MB
// tensor in the 256 MB Infinity Cache and none of a non-temporal one.  Per-launch A/B numbers of byte-bound kernels do not
// carry over to the step; only whole-step A/Bs (DA_LIB_ALT / DA_SET_OPTIONS) decide.
#ifndef NT2_NT
#define NT2_NT 0
#endif
#if NT2_NT
#define NTST st8_nt
#else
#define NTST st8
#endif


// NT2_CT (convolution forms, i.e. !EARLY): the 16x16x32 products are taken TRANSPOSED (W fragment as the first MFMA operand),
// so a lane's four accumulator registers of a tile are four consecutive COLUMNS of one output row instead of four rows of
// one column: the epilogue stages a strip with one ds_write_b128 per tile (5 per strip) instead of four ds_write_b32 (20
// per strip; the LDS store path's 64 B/clk was ~0.55 us of a ~2 us strip).  Same products, same sums, bit-identical output.
// Measured (tools/lib_ab.py nt, profiles/r03_ab_nt_ct.txt): 3x3 convs +1.3...+3.6 %; the persistent linear / GEGLU forms
// -0.4...-3.7 % (their epilogue runs beside the next tile's first DMA and is VALU-issue, not LDS-store, bound), so those
// keep the row-of-column form.  0 = the old form everywhere (A/B).
#ifndef NT2_CT
#define NT2_CT 1
#endif

int g_nt_persist = -1;  // da_set_option("gemm_nt_persist", n): resident workgroups of the persistent forms (-1 = #CUs, 0 = off)
// da_set_option("reserve_cus", R): CUs left to somebody else - the RCCL channels of the gradient all-reduce that overlaps
// backward in a multi-GPU job.  Every grid that is sized to ONE ROUND of the chip (the persistent tile walks here, the
// weight-gradient pixel splits in gemm_tn_v2.hip, the cost model's round count in gemm_nt.hip) is sized to #CUs - R
// instead, so that a CU taken by a collective does not push a whole-CU workgroup into a second round.  0 = whole chip.
int g_reserve_cus = 0;
int g_nt_persist_conv = 1;  // da_set_option("gemm_nt_persist_conv", 0 | 1): the persistent tile walk for 3x3 convolutions
int g_nt_de = 1;            // da_set_option("gemm_nt_de", 0 | 1): direct (register -> HBM) epilogue where it applies (nt2_tile, DE)
int da_usable_cus(int cus) {
  int n = cus - g_reserve_cus;
  return n < 32 ? 32 : n;
}

namespace {

struct GemmNT2Params {
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
  int splits, ksteps_per_split;  // split-K: workgroup (tile, s) multiplies K-steps [s*kps, (s+1)*kps) into slab s
  long slab_stride;              // elements between fp32 partial slabs (split-K only)
  bf16* G;                       // GEGLU variant: gated output [M][inner] (C then holds the pre-activation [M][2*inner])
  long ldg;
  int inner;                     // GEGLU variant: hidden width; W rows [0, inner) = value, [inner, 2*inner) = gate
  int total_blocks;              // persistent (EARLY) form: tiles vblock = blockIdx.x, += gridDim.x, < total_blocks
  FastDiv div_hw, div_w;         // exact m / (Hout*Wout) and rem / Wout for m < 2^24 (persistent convolution form: a runtime
                                 // divisor's reciprocal would live in a VGPR across the tile walk)
  FastDiv div_nblk, div_tn;      // the same for the tile walk's own divisions (block -> split, tile row)
  int korder;                    // 3x3 K-loop order: 0 = tap-major (k = tap*Cin + c, as W is laid out), 1 = channel-chunk-major
                                 // with the 9 taps innermost (see the K-loop comment)
};

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

// LDS image of a [rows x BK] bf16 tile: rows of BK*2 bytes, 16-B chunks XOR-swizzled so that the 16 lanes of a
// ds_read_b128 quarter-wave (16 consecutive rows, same chunk) hit 16 distinct 16-B slots of the 256-B bank span
template <int BK>
DEVINL int swz_key(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
template <int BK>
DEVINL int swz2(int row, int chunk) { return row * (BK * 2) + ((chunk ^ swz_key<BK>(row)) << 4); }

// Workgroup barrier for the LDS strip exchange of the epilogues: waits for this wave's LDS traffic only.
// __syncthreads() would also wait for vmcnt(0), i.e. complete at every strip the residual rows (and, in the fused GEGLU
// backward, the saved activations) that are deliberately requested strips ahead.
DEVINL void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Diagnostic build only (-DDA_STAMPS, tools/nt2_stamps.py): wave 0 of every workgroup writes the shader clock at the phase
// boundaries of a tile, 16 slots per tile.
#ifdef DA_STAMPS
__device__ unsigned long long* g_stamp_buf;
#define STAMP(i)                                                                                      \
  do {                                                                                                \
    if (threadIdx.x == 0 && g_stamp_buf) g_stamp_buf[(long)vblock * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" int da_debug_set_stamps(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#else
#define STAMP(i) do {} while (0)
#endif

DEVINL void glds16(const void* gsrc, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// MT x NT = MFMA tiles (MF x MF, see MF below) per wave; WM x WN = wave grid; BM = MF*MT*WM, BN = MF*NT*WN; BK = K-step
// (64 | 32).  The shipped workhorse is <4, 5, 4, 4>: 256 x 320 x 64, 16 waves (4 per SIMD, <= 128 VGPRs), one workgroup per
// CU; the 8-wave forms serve small / odd shapes, the 4-wave BK-32 form (two workgroups per CU) is a test-only instantiation.
// UPS: gather mode 3 (conv over a nearest-2x upsampled image) - a compile-time split so that the K loop of the other
// modes stays one basic block.
// GEGLU: the feed-forward input projection with its activation fused (diffusers GEGLU: out = value * gelu(gate)).  A
// column tile is the 160 value rows of 160 hidden units followed by their 160 gate rows of W (the per-lane row offsets
// make that remap free), so both halves of every product sit in the same LDS strip; the epilogue stores the bf16
// pre-activation (kept for backward) and the gated output, which saves the separate pass that re-read the former.
// GEGLU == 2 is the backward counterpart on the dgrad of the FF output projection: the tile of d(gated) never goes to
// HBM; the epilogue reads the saved pre-activation, applies the GEGLU derivative and stores d(pre-activation).
// EARLY (ksize 1, mode 0 only: linears, 1x1 convs, the fused GEGLU forms): request the next stage at the START of the K-step
// instead of between its two MFMA halves (measured on the 16-wave form: +3-4 % on these shapes - A streamed from HBM,
// longer lead -, -2-4 % on 3x3 convs), address A rows by 32-bit byte offsets from a uniform base, and run as a
// PERSISTENT grid: one resident workgroup per CU walks the tile list and requests the next tile's first K-step before the
// current tile's epilogue (LDS map below).
// MF: MFMA shape.  16 = v_mfma_f32_16x16x32_bf16 (MT x NT tiles of 16 x 16 per wave); 32 = v_mfma_f32_32x32x16_bf16 (tiles of
// 32 x 32): the same FLOPs in half as many matrix instructions, each of which holds the SIMD's vector issue port for 8
// cycles whatever its shape (MI355X_MICROARCH.md, cycle constants) - with four waves per SIMD also issuing 18-24
// ds_read_b128 and the LDS-DMA pieces of the next stage every K-step, the 16x16x32 form spends 1,280 of a step's 2,560
// matrix-pipe cycles on MFMA issue alone and the step measures ~3,500.
// PERSIST: the resident-workgroup tile walk with the next tile's first K-step requested ahead of the epilogue (LDS map below).
// Always with EARLY; also - round 3 - for the 16-wave convolution form when a launch has more tiles than CUs.
// DE (round 4): DIRECT epilogue - the tile leaves the accumulators for HBM without the LDS strip exchange.  With the
// transposed products (CT) a lane's four registers of a 16x16 tile are four consecutive columns of one output row; packed
// to bf16 that is 8 bytes, and one v_permlane16_swap per dword between the tiles (i, j) and (i, j+1) turns them into 16
// contiguous bytes per lane (lane rows q = lane >> 4: q even -> tile j, q odd -> tile j+1, columns 8*(q>>1)..+7), i.e. one
// global_store_dwordx4 per tile pair covering 16 rows x 64 B.  No LDS traffic, no barrier, ~10 VALU per store: by the
// round-2 clock stamps the strip epilogue of a K = 320 tile took 13.6 us (4 strips x (LDS write + barrier + ~45 VALU per
// task + barrier)) against 10.2 us for its whole K loop, with the matrix pipe idle and the read stream dry - the K <= 640
// linears ran at 3.1-3.4 TB/s of algorithmic bytes.  The residual is loaded in the same lane layout (16 B per lane, the
// same shuffle backwards) in two halves of the row block, each into registers the K loop's fragments (first half) and the
// first half's accumulators (second half) have just left; sums and roundings are those of the strip epilogue, bit for bit.
// HASR: a residual is added (compile-time, so that the form without one carries none of its registers or branches).
template <int MT, int NT, int WM, int WN, int BK, bool UPS, int GEGLU, bool EARLY, int MF = 16, bool PERSIST = EARLY, bool DE = false,
          bool HASR = false>
DEVINL void nt2_tile(const GemmNT2Params& p, const int vblock0, char* smem) {
  constexpr int NW = WM * WN;
  constexpr bool CT = MF == 16 && GEGLU == 0 && ((NT2_CT && !EARLY) || DE);  // transposed products, see NT2_CT
  static_assert(!DE || (MF == 16 && GEGLU == 0 && ((NT % 2) == 0 || (MT % 2) == 0)), "direct epilogue: 16x16 tiles in pairs");
  constexpr int V2_BM = MF * MT * WM, V2_BK = BK;
  static_assert((NW == 16 || NW == 8 || NW == 4) && (BK == 64 || BK == 32), "wave grid");
  static_assert(MF == 16 || (MF == 32 && GEGLU == 0 && BK == 64), "MFMA shape");
  constexpr int BN = MF * NT * WN;
  constexpr int SR = MF == 16 ? 16 : 8;  // rows per epilogue strip: the rows one accumulator register quad covers per wave
  constexpr int RG = 512 / BK;                // tile rows per 1-KiB DMA group (8 | 16)
  constexpr int LR = BK / 8;                  // lanes (16-B chunks) per row
  constexpr int AJ = V2_BM / RG / NW;         // A row groups per wave
  constexpr int BGROUPS = BN / RG;            // 1-KiB row groups of the B tile
  constexpr int BJ = (BGROUPS + NW - 1) / NW; // B row groups per wave
  constexpr int A_BYTES = V2_BM * V2_BK * 2;
  constexpr int B_BYTES = BN * V2_BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  static_assert(V2_BM % (RG * NW) == 0, "A groups");
  // LDS map.  One tile per workgroup (convolutions): [stage 0 | stage 1 | bias row]; the epilogue's fp32 strips reuse the
  // stage buffers.  Persistent short-K forms (EARLY): [stage 0 | stage 1 ... | bias row x 2] with the strips starting at
  // stage 1 and running past it, clear of stage 0 - the NEXT tile's descriptors are set up and its first K-step is
  // requested right after the K loop, so that load (2.2-2.8 us of a 27 us tile by the clock stamps, plus ~0.9 us of
  // descriptor arithmetic waiting on nothing) is in flight under the epilogue instead of in front of the next K loop.
  constexpr int EPI_LD = BN + 4;
  constexpr int STRIP = SR * EPI_LD;  // floats per strip
  constexpr int STRIP_OFF = PERSIST ? STAGE : 0;
  constexpr int BIAS_OFF = !PERSIST ? 2 * STAGE : (STAGE + WM * STRIP * 4 > 2 * STAGE ? STAGE + WM * STRIP * 4 : 2 * STAGE);
  int vblock = vblock0;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // Epilogue work split: the WN waves of a row block share its 16-row strip as 64*WN (row, 8-column) tasks per pass.  The
  // strip has 2.5 passes of tasks at BN 320 (1.25 for the GEGLU forward), and waves land on SIMD (wave % 4) = wn: with the
  // ragged last pass always on wn 0.. the VALU-heavy epilogue (the strips measure VALU-issue bound: 2.0 us / strip with
  // tight spread, 4 us with the gelu) ran 3 passes on SIMD 0/1 and 2 on SIMD 2/3.  The last pass's wave slot is rotated by
  // the row block, so every SIMD gets the same share.
  auto task_of = [&](int ln, int pss, int tasks) {
    const bool ragged_last = (tasks % (64 * WN)) != 0 && pss == (tasks + 64 * WN - 1) / (64 * WN) - 1;
    const int slot = ragged_last ? (wn + WN - (wm * WN / 4) % WN) % WN : wn;  // SIMD of a wave = (wm * WN + wn) % 4
    return slot * 64 + ln + 64 * WN * pss;
  };
  // Per-lane values derived from the lane id inside the tile loop of the persistent forms would be hoisted out of it and
  // kept in VGPRs across the K loop, which has none to spare (128 per wave): each phase starts from a laundered copy, so
  // its lane-derived constants are recomputed per tile (a few VALU ops) instead of living - or spilling - through the loop.
  auto fresh_lane = [&]() {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    return ln;
  };
  STAMP(0);

  const int nblk = p.tiles_m * p.tiles_n;
  int split, tn, m0, n0;  // the tile the DMA descriptors below belong to (wave-uniform)
  auto locate = [&](int vb) {
    // (FastDiv only in the persistent convolution form: in the persistent linear form it costs 7 spills - measured -6...-16 %)
    split = (PERSIST && !EARLY) ? (int)fdiv((unsigned)vb, p.div_nblk) : vb / nblk;  // splits of a tile are nblk workgroups apart
    int bid = vb - split * nblk;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = (PERSIST && !EARLY) ? (int)fdiv((unsigned)bid, p.div_tn) : bid / p.tiles_n;
    tn = bid - tm * p.tiles_n;
    m0 = tm * V2_BM;
    n0 = tn * BN;
  };
  // weight / bias row behind tile column c (GEGLU: 160 value rows, then the 160 matching gate rows)
  auto wrow = [&](int c) {
    if constexpr (GEGLU == 1) return c < BN / 2 ? tn * (BN / 2) + c : p.inner + tn * (BN / 2) + (c - BN / 2);
    else return n0 + c;
  };
  const int HWo = p.Hout * p.Wout;
  // mode 4: stride 2 with zero padding at the bottom / right only (the VAE encoder's downsampler) = mode 1 without the
  // one-pixel shift
  // PCONV: the persistent convolution form is dispatched for stride-1 3x3 gathers only (mode 0: every multi-round conv and
  // conv dgrad of the U-Net) - its gather constants are compile-time, which keeps the tile walk free of the uniform-value
  // VGPRs (shift amounts, limits) the compiler otherwise carried - and spilled - across the K loop
  constexpr bool PCONV = PERSIST && !EARLY;
  const int pad = PCONV ? 1 : ((p.ksize == 3 && p.mode != 4) ? 1 : 0);
  const int gmul = PCONV ? 1 : ((p.mode == 1 || p.mode == 4) ? 2 : 1), gshift = PCONV ? 0 : ((p.mode == 2 || p.mode == 3) ? 1 : 0),
            pmask = PCONV ? 0 : ((p.mode == 2) ? 1 : 0);
  const int hlim = gshift ? 2 * p.Hin : p.Hin, wlim = gshift ? 2 * p.Win : p.Win;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  // bias for this column block -> LDS (behind the stage buffers), read back in the epilogue; zeros when absent
  // (LDS-DMA, like the tiles: no register round trip, no wait here - it has landed by the next __syncthreads)
  auto load_bias = [&](int buf) {
    if (tid < BN / 4) {
      const int n = wrow(tid * 4);
      const void* src = (p.bias && n < p.N) ? (const void*)(p.bias + n) : (const void*)zero;
      glds16(src, smem + BIAS_OFF + buf * (BN * 4) + wave * 1024);
    }
  };

  // ---- per-lane DMA sources, reduced ONCE per tile to what a K-step needs: the address arithmetic of a step runs on
  // every wave with the matrix pipe idle (it measured ~0.45 us of a ~2.2 us step when done from scratch each time).
  // A row j of this lane (row groups wave*AJ .. wave*AJ+AJ-1):
  //   abase[j]  pointer to the row's tap-(0,0) source pixel (+ this lane's swizzled 16-B chunk)
  //   amask[j]  bit t: tap t reads inside the image (modes: 0 stride 1, 1 stride 2, 2 dgrad of stride 2, 3 fused
  //             nearest-2x upsample); bits 16/17: parity of the output row / column (mode 3 only)
  // so that a step adds one wave-uniform offset (tap displacement + channel offset) and selects the zero page.
  const int ntaps = PCONV ? 9 : p.ksize * p.ksize;
  const bf16* abase[AJ];
  unsigned amask[AJ];
  unsigned aoff[AJ];  // EARLY forms, and the persistent convolution form (packed descriptors)
  unsigned woff[BJ];  // bytes
  int nk, tap, c0;
  const int nk_total = p.K / V2_BK;
  const bool tap_inner = p.korder != 0 && ntaps > 1;  // K-loop order, see below
  auto describe = [&]() {
  const int dl = fresh_lane();
  const int lrow = dl / LR, lchunk = dl % LR;
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int row = (wave * AJ + j) * RG + lrow;
    const int m = m0 + row;
    const bool mval = m < p.M;
    if constexpr (EARLY) {
      // EARLY is instantiated for ksize 1, mode 0 only: input pixel == output pixel, one tap.  The source is the uniform
      // base A + c0 plus a 32-bit per-lane BYTE offset (the host checks M * lda * 2 < 4 GiB), the SGPR-base addressing form
      // of the load: one VGPR per row group instead of a 64-bit pointer and a mask.  Rows past M read row 0: their
      // products land in rows the epilogue never stores.
#ifdef NT2_EXP_AHOT     // timing-only build: every tile reads the first 256 rows of A (L2-resident)
      aoff[j] = (unsigned)(row) * (unsigned)(p.lda * 2) + (lchunk ^ swz_key<BK>(row)) * 16;
#else
      aoff[j] = (unsigned)(mval ? m : 0) * (unsigned)(p.lda * 2) + (lchunk ^ swz_key<BK>(row)) * 16;
#endif
      continue;
    }
    const int mm = mval ? m : 0;
    int b, rem, oh;
    if constexpr (PERSIST) {
      b = (int)fdiv((unsigned)mm, p.div_hw);
      rem = mm - b * HWo;
      oh = (int)fdiv((unsigned)rem, p.div_w);
    } else {
      b = mm / HWo;
      rem = mm - b * HWo;
      oh = rem / p.Wout;
    }
    const int ow = rem - oh * p.Wout;
    unsigned mask = ((oh & 1) << 16) | ((ow & 1) << 17);
    for (int t = 0; t < ntaps; ++t) {
      const int r = (PCONV || p.ksize == 3) ? t / 3 : 0, s2 = (PCONV || p.ksize == 3) ? t - 3 * r : 0;
      const int th = oh * gmul + r - pad, tw = ow * gmul + s2 - pad;
      const bool ok = mval && (unsigned)th < (unsigned)hlim && (unsigned)tw < (unsigned)wlim && !((th | tw) & pmask);
      mask |= (ok ? 1u : 0u) << t;
    }
    const int bh = (oh * gmul - pad) >> gshift, bw = (ow * gmul - pad) >> gshift;
    if constexpr (PERSIST && !UPS) {
      // persistent convolution form: the walk keeps the descriptors of a tile live next to the accumulators, so they are
      // packed - a 32-bit byte offset from A (mod 2^32: the tap-(0,0) pixel of a border row lies before the image, the
      // displacement of a VALID tap brings the sum back inside; the host checks the activation is < 4 GiB) and both rows'
      // 9-bit tap masks in one register
      aoff[j] = (unsigned)(((long)(b * p.Hin * p.Win) + (long)bh * p.Win + bw) * p.lda * 2 + (lchunk ^ swz_key<BK>(row)) * 16);
      if (j == 0) amask[0] = mask & 0x1ffu;
      else amask[0] |= (mask & 0x1ffu) << (9 * j);
      continue;
    }
    amask[j] = mask;
    abase[j] = p.A + ((long)(b * p.Hin * p.Win) + (long)bh * p.Win + bw) * p.lda + (lchunk ^ swz_key<BK>(row)) * 8;
  }
  // B: row groups wave, wave+NW, ... (< BGROUPS): uniform base W + k0 plus a constant per-lane element offset.
  // Rows past N are clamped (their products land in columns the epilogue never stores).
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int g = wave + NW * j;
    const int row = g * RG + lrow;
    const int n = min(wrow(row), p.N - 1);
    woff[j] = ((unsigned)n * (unsigned)p.K + (lchunk ^ swz_key<BK>(row)) * 8) * 2;
  }

  const int kstep_begin = split * p.ksteps_per_split;
  nk = min(p.ksteps_per_split, nk_total - kstep_begin);
  // K-loop order.  W is [n][tap][c], and walking k linearly visits tap 0 of all Cin channels, then tap 1, ...: the 9
  // shifted reads of one activation row are Cin/64 steps apart, and with every CU of an XCD streaming its own row block
  // (32 x Cin/64 x 32 KiB between two taps) they fall out of the 4 MiB L2: 1.68 GB of fabric reads per launch on the
  // 320-channel 32x32 layers where 0.34 GB is algorithmic (rocprofv3 FETCH_SIZE, L2 hit 58 %).  korder 1 walks one
  // 64-channel chunk through its 9 taps before the next chunk: the re-reads are consecutive steps over a ~42 KiB footprint.
  // Same products, summed in a different order.
  if (tap_inner) {
    const int ch = kstep_begin / ntaps;
    tap = kstep_begin - ch * ntaps;
    c0 = ch * V2_BK;
  } else {
    const int kb = kstep_begin * V2_BK;
    tap = kb / p.Cin;
    c0 = kb - tap * p.Cin;
  }
  };  // describe()
  // live == false (the step after the last): A reads the zero page, B re-reads K-step 0 - no branch in the K loop
  auto issue = [&](int stage, bool live) {
    char* Ab = smem + stage * STAGE;
    char* Bb = Ab + A_BYTES;
    const int r = tap / 3, s2 = tap - 3 * r;  // ksize 1: tap stays 0
    const unsigned tapbit = live ? (1u << tap) : 0u;
    if constexpr (EARLY) {
      const char* au = reinterpret_cast<const char*>(p.A + (live ? c0 : 0));  // the step after the last re-reads K-step 0
#pragma unroll
      for (int j = 0; j < AJ; ++j) glds16(au + aoff[j], Ab + (wave * AJ + j) * 1024);
    } else if constexpr (!UPS) {
      // tap displacement in source pixels: r, s (stride 1 / 2) or (r+pad)/2 (dgrad of stride 2, only even taps valid)
      const int dr = (r + (gshift ? pad : 0)) >> gshift, ds = (s2 + (gshift ? pad : 0)) >> gshift;
      const long soff = (long)(dr * p.Win + ds) * p.lda + c0;
      if constexpr (PERSIST) {
        static_assert(AJ * 9 <= 32, "packed tap masks");
        const char* au = reinterpret_cast<const char*>(p.A);
        const unsigned sb = (unsigned)(soff * 2);
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
          const void* src = ((amask[0] >> (9 * j)) & tapbit) ? (const void*)(au + (unsigned)(aoff[j] + sb)) : (const void*)zero;
          glds16(src, Ab + (wave * AJ + j) * 1024);
        }
      } else {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
          const void* src = (amask[j] & tapbit) ? (const void*)(abase[j] + soff) : (const void*)zero;
          glds16(src, Ab + (wave * AJ + j) * 1024);
        }
      }
    } else {
      // nearest-2x upsample: source row of tap r is (oh + r - pad) >> 1, which depends on the parity of oh
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int ph = (amask[j] >> 16) & 1, pw = (amask[j] >> 17) & 1;
        const int dr = (r + pad * (1 - ph)) >> 1, ds = (s2 + pad * (1 - pw)) >> 1;
        const long soff = (long)(dr * p.Win + ds) * p.lda + c0;
        const void* src = (amask[j] & tapbit) ? (const void*)(abase[j] + soff) : (const void*)zero;
        glds16(src, Ab + (wave * AJ + j) * 1024);
      }
    }
    const char* wb = reinterpret_cast<const char*>(p.W + (live ? tap * p.Cin + c0 : 0));
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int g = wave + NW * j;
      if constexpr (PERSIST && !EARLY) {
        // persistent convolution form (full column tiles only, N % BN == 0: no row clamping): row group g = wave + NW*j lies
        // NW*RG rows of W behind group `wave` - a uniform offset on the base, ONE lane offset register for all groups
        if (BGROUPS % NW == 0 || g < BGROUPS) glds16(wb + (size_t)j * (NW * RG) * p.K * 2 + woff[0], Bb + g * 1024);
      } else {
        if (BGROUPS % NW == 0 || g < BGROUPS) glds16(wb + woff[j], Bb + g * 1024);
      }
    }
    // advance (selects, no branch): tap-inner walks the taps of one channel chunk; tap-major walks the chunks of one tap
    const int tap_n = tap + 1, c_n = c0 + V2_BK;
    const bool wrap_t = tap_n >= ntaps, wrap_c = c_n >= p.Cin;
    if (tap_inner) {
      tap = wrap_t ? 0 : tap_n;
      c0 = wrap_t ? c_n : c0;
    } else {
      c0 = wrap_c ? 0 : c_n;
      tap = wrap_c ? tap_n : tap;
    }
  };

  typedef float accv_t __attribute__((ext_vector_type(MF == 16 ? 4 : 16)));
  accv_t acc[MT][NT];

  auto compute_half = [&](int stage, int s) {
    const char* Ab = smem + stage * STAGE;
    const char* Bb = Ab + A_BYTES;
    bf16x8 a[MT], b[NT];
    // The fragment addresses are a handful of VALU ops from the lane id; they are recomputed per half-step (the empty asm
    // stops the compiler from hoisting them out of the K loop).  Hoisted, they push the 128-VGPR body over the edge: a
    // lane descriptor gets spilled, and its reload inside the K loop comes with an s_waitcnt vmcnt(0) - a wait for the
    // LDS-DMA requests issued just before it - i.e. one exposed load latency per K-step (seen in the ISA of the 3x3 form
    // after the K-order change, and of every form once the body sits in the persistent loop).
    int ln = lane;
    asm volatile("" : "+v"(ln));
    if constexpr (MF == 16) {
      const int chunk = s * 4 + (ln >> 4);
#pragma unroll
      for (int i = 0; i < MT; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(Ab + swz2<BK>(wm * (16 * MT) + i * 16 + (ln & 15), chunk));
#pragma unroll
      for (int j = 0; j < NT; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(Bb + swz2<BK>(wn * (16 * NT) + j * 16 + (ln & 15), chunk));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = CT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    } else {
      // two 16-deep slices per half-step; lane = (row % 32, k-half): the 16 rows a ds_read_b128 lane group touches are 8
      // even + 8 odd ones with 8 distinct (row >> 1) & 7 keys, so the 64-B-step swizzle stays conflict-free
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int chunk = s * 4 + ks * 2 + (ln >> 5);
#pragma unroll
        for (int i = 0; i < MT; ++i)
          a[i] = *reinterpret_cast<const bf16x8*>(Ab + swz2<BK>(wm * (32 * MT) + i * 32 + (ln & 31), chunk));
#pragma unroll
        for (int j = 0; j < NT; ++j)
          b[j] = *reinterpret_cast<const bf16x8*>(Bb + swz2<BK>(wn * (32 * NT) + j * 32 + (ln & 31), chunk));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  // stage 0 of the described tile has been requested; returns with the tile's products in acc and both stages idle
  // The accumulators start from the bias (plain forms with alpha == 1 and no split-K: the sum is the same up to fp32
  // rounding order): the epilogue's passes then need neither the 2 x 16-B LDS reads of the bias row per task - as much
  // LDS traffic as reading the strip itself - nor the registers to hold them next to the residual ring.
  const bool fold_bias = p.alpha == 1.0f && p.splits == 1;  // (the GEGLU entry points always launch with alpha 1, one split)
  auto kloop = [&](const int bbuf) {
  STAMP(1);
  __syncthreads();
  STAMP(2);
  if constexpr (CT) {  // register e of tile j = column wn*16*NT + j*16 + 4*(lane >> 4) + e
    const float* brow = reinterpret_cast<const float*>(smem + BIAS_OFF + bbuf * (BN * 4)) + wn * (MF * NT) + (fresh_lane() >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const f32x4 b4 = fold_bias ? *reinterpret_cast<const f32x4*>(brow + j * MF) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] = b4;
    }
  } else {
    const float* brow = reinterpret_cast<const float*>(smem + BIAS_OFF + bbuf * (BN * 4)) + wn * (MF * NT) + (fresh_lane() & (MF - 1));
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float b = fold_bias ? brow[j * MF] : 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < (MF == 16 ? 4 : 16); ++e) acc[i][j][e] = b;
    }
  }
#ifdef DA_STAMPS
  // K-loop phase sums of wave 0 (no memory traffic inside the loop): slots 8..11 = cycles from the step's start to the end
  // of MFMA half 0 | of the request block | of MFMA half 1 (all "issued") | of the barrier
  unsigned long long ph[4] = {0, 0, 0, 0}, tp = __builtin_amdgcn_s_memtime();
#define KSTAMP(i)                                         \
  do {                                                    \
    const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); \
    ph[i] += tn_ - tp;                                    \
    tp = tn_;                                             \
  } while (0)
#else
#define KSTAMP(i) do {} while (0)
#endif
  for (int t = 0; t < nk; ++t) {
    // the address arithmetic + DMA issue of step t+1 sits BETWEEN the two MFMA halves of step t: every wave leaves
    // the barrier at the same time, so issuing first would idle the matrix pipe of all four SIMDs during it
    if constexpr (BK == 64 && EARLY) {
      issue((t + 1) & 1, t + 1 < nk);
      compute_half(t & 1, 0);
      compute_half(t & 1, 1);
    } else if constexpr (BK == 64) {
      compute_half(t & 1, 0);
      KSTAMP(0);
      issue((t + 1) & 1, t + 1 < nk);
      KSTAMP(1);
      compute_half(t & 1, 1);
      KSTAMP(2);
    } else {  // one 32-deep MFMA pass per step; the co-resident workgroup covers the issue slot
      issue((t + 1) & 1, t + 1 < nk);
      compute_half(t & 1, 0);
    }
    __syncthreads();  // vmcnt(0): step t+1 has landed; barrier: everyone is done reading stage t
    KSTAMP(3);
  }
#ifdef DA_STAMPS
  if (threadIdx.x == 0 && g_stamp_buf)
    for (int i = 0; i < 4; ++i) g_stamp_buf[(long)vblock * 16 + 8 + i] = ph[i];
#endif
  STAMP(3);
  };  // kloop()

  // the epilogue of tile (m0e, n0e, tne, splite) with its bias row in LDS buffer bbuf
  auto epilogue = [&](const int m0e, const int n0e, const int tne, const int splite, const int bbuf) {
  if constexpr (DE) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    constexpr int NPJ = NT / 2;                    // column pairs (i, 2*pj), (i, 2*pj+1) per 16-row strip
    constexpr bool ODD = (NT & 1) != 0;            // odd NT: the last column tiles pair up over rows, (i-1, NT-1), (i, NT-1), i odd
    constexpr int IH = ODD ? ((MT / 2 + 1) / 2) * 2 : (MT + 1) / 2;  // strips of the first half (whole row pairs when ODD)
    const int ln = fresh_lane();
    char* const cb = reinterpret_cast<char*>(p.C);
    const char* const rb = reinterpret_cast<const char*>(p.R);
    // (row, column) of this lane's 16 bytes inside the tile: column pair pj of strip i / the odd column's row pair ending at strip i
    const int lrow = wm * (16 * MT) + (ln & 15), lq = (ln >> 4) & 1, lcol = wn * (16 * NT) + 8 * (ln >> 5);
    auto row_of = [&](int i) { return m0e + lrow + i * 16; };
    auto col_of = [&](int pj) { return n0e + lcol + 32 * pj + 16 * lq; };
    auto row5_of = [&](int i) { return m0e + lrow + (i - 1 + lq) * 16; };
    const int col5 = n0e + lcol + 16 * (NT - 1);
    u32x4 rin[MT][NPJ > 0 ? NPJ : 1], rin5[MT];
    // residual rows of strips [i_lo, i_hi): rows / columns past the matrix are clamped, not predicated (what they feed is
    // never stored; a branch per load would make the compiler wait vmcnt(0) per use beside the LDS-DMA requests in flight)
    auto fetch = [&](const int i_lo, const int i_hi) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i < i_lo || i >= i_hi) continue;
        const unsigned mo = (unsigned)min(row_of(i), p.M - 1) * (unsigned)(p.ldr * 2);
#pragma unroll
        for (int pj = 0; pj < NPJ; ++pj)
          rin[i][pj] = *reinterpret_cast<const u32x4*>(rb + (mo + (unsigned)min(col_of(pj), p.N - 8) * 2u));
        if constexpr (ODD) {
          if (i & 1)
            rin5[i] = *reinterpret_cast<const u32x4*>(rb + ((unsigned)min(row5_of(i), p.M - 1) * (unsigned)(p.ldr * 2) +
                                                            (unsigned)min(col5, p.N - 8) * 2u));
        }
      }
    };
    // the 16 bytes of a tile pair (a, b): [+ row bias] [+ residual] -> bf16 -> lane shuffle
    auto finish = [&](accv_t a, accv_t b, const u32x4 r, const int ma, const int mb, const int na, const int nb) {
      if (p.rowbias) {  // per-image row bias (the time-embedding projection behind a ResnetBlock2D's first convolution)
        const int ia = (PERSIST && !EARLY) ? (int)fdiv((unsigned)min(ma, p.M - 1), p.div_hw) : min(ma, p.M - 1) / HWo;
        const int ib = (PERSIST && !EARLY) ? (int)fdiv((unsigned)min(mb, p.M - 1), p.div_hw) : min(mb, p.M - 1) / HWo;
        const bf16x4 ra = *reinterpret_cast<const bf16x4*>(p.rowbias + (long)ia * p.ldrb + min(na, p.N - 4));
        const bf16x4 rb4 = *reinterpret_cast<const bf16x4*>(p.rowbias + (long)ib * p.ldrb + min(nb, p.N - 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[e] += bf2f(ra[e]);
          b[e] += bf2f(rb4[e]);
        }
      }
      if constexpr (HASR) {
        // the store shuffle backwards: dwords (0, 2) and (1, 3) swap back to (tile a, tile b) x (columns 0-1, columns 2-3)
        const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(r[0], r[2], false, false);
        const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(r[1], r[3], false, false);
        a[0] += __builtin_bit_cast(float, s0[0] << 16);
        a[1] += __builtin_bit_cast(float, s0[0] & 0xffff0000u);
        a[2] += __builtin_bit_cast(float, s1[0] << 16);
        a[3] += __builtin_bit_cast(float, s1[0] & 0xffff0000u);
        b[0] += __builtin_bit_cast(float, s0[1] << 16);
        b[1] += __builtin_bit_cast(float, s0[1] & 0xffff0000u);
        b[2] += __builtin_bit_cast(float, s1[1] << 16);
        b[3] += __builtin_bit_cast(float, s1[1] & 0xffff0000u);
      }
      bf16x2 a0, a1, b0, b1;
      a0[0] = f2bf(a[0]); a0[1] = f2bf(a[1]); a1[0] = f2bf(a[2]); a1[1] = f2bf(a[3]);
      b0[0] = f2bf(b[0]); b0[1] = f2bf(b[1]); b1[0] = f2bf(b[2]); b1[1] = f2bf(b[3]);
      const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, b0), false, false);
      const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a1), __builtin_bit_cast(unsigned, b1), false, false);
      return u32x4{s0[0], s1[0], s0[1], s1[1]};
    };
    const int aq = (ln >> 4) * 4;  // first column of this lane's accumulator quad inside a 16-column tile
    if constexpr (HASR) fetch(0, IH);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if constexpr (HASR) {
        if (i == IH) fetch(IH, MT);
      }
      const int m = row_of(i);
      const int ma = m0e + wm * (16 * MT) + i * 16 + (ln & 15);  // the accumulators' own row (before the shuffle)
#pragma unroll
      for (int pj = 0; pj < NPJ; ++pj) {
        const int n = col_of(pj);
        const int na = n0e + wn * (16 * NT) + 32 * pj + aq;
        const u32x4 o = finish(acc[i][2 * pj], acc[i][2 * pj + 1], rin[i][pj], ma, ma, na, na + 16);
#ifdef NT2_EXP_NOSTORE  // timing-only build (tools/build_alt.sh): the tile is computed and packed, nothing is stored
        asm volatile("" ::"v"(o));
#else
        if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(cb + ((unsigned)m * (unsigned)(p.ldc * 2) + (unsigned)n * 2u)) = o;
#endif
      }
      if constexpr (ODD) {
        if (i & 1) {
          const int m5 = row5_of(i);
          const int na = n0e + wn * (16 * NT) + 16 * (NT - 1) + aq;
          const u32x4 o = finish(acc[i - 1][NT - 1], acc[i][NT - 1], rin5[i], ma - 16, ma, na, na);
#ifdef NT2_EXP_NOSTORE
          asm volatile("" ::"v"(o));
#else
          if (m5 < p.M && col5 < p.N) *reinterpret_cast<u32x4*>(cb + ((unsigned)m5 * (unsigned)(p.ldc * 2) + (unsigned)col5 * 2u)) = o;
#endif
        }
      }
    }
    STAMP(4);
    return;
  }
  const float* bias_lds = reinterpret_cast<const float*>(smem + BIAS_OFF + bbuf * (BN * 4));
  float* const strips = reinterpret_cast<float*>(smem + STRIP_OFF);
  const int el = fresh_lane();
  if constexpr (GEGLU == 1) {
    static_assert(BN == 320 && !UPS, "GEGLU tile = 160 value + 160 gate columns");
    constexpr int GLD = BN + 4, GSTRIP = 16 * GLD, HC = BN / 2;  // strip layout as below; HC hidden units per tile
    constexpr int GTASKS = 16 * (HC / 8), GPASSES = (GTASKS + 64 * WN - 1) / (64 * WN);
    const int mrow0g = m0e + wm * (16 * MT);
    const int h0 = tne * HC;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float* ew = strips + wm * GSTRIP;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (CT) {
          *reinterpret_cast<f32x4*>(&ew[(el & 15) * GLD + wn * (16 * NT) + j * 16 + (el >> 4) * 4]) = acc[i][j];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            ew[((el >> 4) * 4 + e) * GLD + wn * (16 * NT) + j * 16 + (el & 15)] = acc[i][j][e];
        }
      }
      lds_barrier();
#pragma unroll
      for (int pss = 0; pss < GPASSES; ++pss) {
        const int task = task_of(el, pss, GTASKS);
        const int row = task / (HC / 8), c8 = (task - row * (HC / 8)) * 8;
        const int m = mrow0g + i * 16 + row;
        if (task < GTASKS && m < p.M && h0 + c8 < p.inner) {
          bf16x8 fv, fg, og;
#pragma unroll
          for (int q4 = 0; q4 < 2; ++q4) {
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(&ew[row * GLD + c8 + 4 * q4]);
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(&ew[row * GLD + HC + c8 + 4 * q4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // (bias: in the accumulators from the start, as in the plain form) rounded to bf16 BEFORE gating, exactly as
              // the two-kernel path (da_gemm_nt then da_geglu_fwd) does
              fv[4 * q4 + e] = f2bf(v4[e]);
              fg[4 * q4 + e] = f2bf(g4[e]);
              og[4 * q4 + e] = f2bf(bf2f(fv[4 * q4 + e]) * gelu_f(bf2f(fg[4 * q4 + e])));
            }
          }
          bf16* fp = reinterpret_cast<bf16*>(p.C) + (long)m * p.ldc + h0 + c8;
          NTST(fp, fv);
          NTST(fp + p.inner, fg);
          NTST(p.G + (long)m * p.ldg + h0 + c8, og);
        }
      }
      lds_barrier();  // single strip buffer: everyone is done reading before it is rewritten
      STAMP(4 + i);
    }
    return;
  }

  if constexpr (GEGLU == 2) {
    // C = d(pre-activation) [M][2*inner], G = saved pre-activation F [M][2*inner] (read only), tile = 320 columns of
    // d(gated); arithmetic and rounding points as da_geglu_bwd on a bf16 d(gated)
    constexpr int GLD = BN + 4, GSTRIP = 16 * GLD, CHB = BN / 8;
    constexpr int GTASKS = 16 * CHB, GPASSES = (GTASKS + 64 * WN - 1) / (64 * WN);
    const int mrow0g = m0e + wm * (16 * MT);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      float* ew = strips + wm * GSTRIP;
      bf16x8 fa[GPASSES], fg[GPASSES];  // this strip's saved value / gate rows, in flight across the LDS exchange
#pragma unroll
      for (int pss = 0; pss < GPASSES; ++pss) {
        const int task = task_of(el, pss, GTASKS);
        const int row = task / CHB, c8 = (task - row * CHB) * 8;
        const int m = mrow0g + i * 16 + row;
        fa[pss] = zero8();
        fg[pss] = zero8();
        if (task < GTASKS && m < p.M && n0e + c8 < p.inner) {
          const bf16* fp = p.G + (long)m * p.ldg + n0e + c8;
          fa[pss] = ld8(fp);
          fg[pss] = ld8(fp + p.inner);
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (CT) {
          *reinterpret_cast<f32x4*>(&ew[(el & 15) * GLD + wn * (16 * NT) + j * 16 + (el >> 4) * 4]) = acc[i][j];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            ew[((el >> 4) * 4 + e) * GLD + wn * (16 * NT) + j * 16 + (el & 15)] = acc[i][j][e];
        }
      }
      lds_barrier();
#pragma unroll
      for (int pss = 0; pss < GPASSES; ++pss) {
        const int task = task_of(el, pss, GTASKS);
        const int row = task / CHB, c8 = (task - row * CHB) * 8;
        const int m = mrow0g + i * 16 + row;
        if (task < GTASKS && m < p.M && n0e + c8 < p.inner) {
          bf16x8 da, dg;
#pragma unroll
          for (int q4 = 0; q4 < 2; ++q4) {
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(&ew[row * GLD + c8 + 4 * q4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float df = bf2f(f2bf(d4[e] * p.alpha)), gf = bf2f(fg[pss][4 * q4 + e]);
              da[4 * q4 + e] = f2bf(df * gelu_f(gf));
              dg[4 * q4 + e] = f2bf(df * bf2f(fa[pss][4 * q4 + e]) * dgelu_f(gf));
            }
          }
          bf16* op = reinterpret_cast<bf16*>(p.C) + (long)m * p.ldc + n0e + c8;
          st8(op, da);
          st8(op + p.inner, dg);
        }
      }
      lds_barrier();
      STAMP(4 + i);
    }
    return;
  }

  // ---- epilogue: the WN waves that share a row block stage their i-th 16-row MFMA strip side by side in LDS, then
  // leave it as full BN-wide rows (640 B contiguous per row at BN 320) - per-wave strips would store and read the
  // residual in 16*NT*2-byte pieces (160 B, straddling 128-B lines), which measured ~2.4 TB/s on the K=320 linears
  constexpr bool EPI_DB = !PERSIST && 2 * WM * STRIP * 4 <= 2 * STAGE;  // double-buffered strips: one barrier per strip
  constexpr int CH = BN / 8;                              // 8-column chunks per row
  constexpr int TASKS = SR * CH;                          // (row, chunk) pairs per strip, shared by 64*WN lanes
  constexpr int NSI = MT * (MF == 16 ? 1 : 4);            // strips per wave: one per 16-row tile | per register quad of a 32-row tile
  constexpr int PASSES = (TASKS + 64 * WN - 1) / (64 * WN);
  // residual rows are fetched ahead of their strip (the MFMA operand registers are dead by now): with a load -> wait ->
  // store chain per strip the epilogue exposed one HBM latency per strip, ~37 us per 256x320 tile on the K=320 linears.
  // 8-wave forms: a ring RD strips deep, refilled after a strip's stores; the 16-wave forms (128 VGPRs) have room for ONE
  // strip of residual next to the accumulators, and ~1.6 us of each strip's load stays exposed.  Two things that did not
  // fix that (clock stamps, tools/nt2_stamps.py): a two-strip ring spills - the spill of a just-requested register waits
  // for its load - and refilling each register right after its use (same footprint, a full strip of lead) makes the
  // compiler's s_waitcnt conservative: with the validity branches around every load it cannot count the requests issued
  // since, waits vmcnt(0) before each use, and the strips got 20 % slower.
  constexpr int RBUD = NW == 16 ? 3 : 10;  // 16-byte registers for the ring
  constexpr int RD = (NSI < RBUD / PASSES) ? NSI : (RBUD / PASSES < 1 ? 1 : RBUD / PASSES);
  int trow[PASSES], tcol[PASSES];
  bool tval[PASSES];
#pragma unroll
  for (int pss = 0; pss < PASSES; ++pss) {
    const int task = task_of(el, pss, TASKS);
    trow[pss] = task / CH;
    tcol[pss] = (task - trow[pss] * CH) * 8;
    tval[pss] = task < TASKS && n0e + tcol[pss] < p.N;
  }
  const int mrow0 = m0e + wm * (MF * MT);
  const bool has_r = p.R != nullptr && p.splits == 1;
  bf16x8 rres[RD][PASSES];
  auto fetch_r = [&](int i, int slot) {
#pragma unroll
    for (int pss = 0; pss < PASSES; ++pss) {
      const int m = mrow0 + i * SR + trow[pss];
      if (tval[pss] && m < p.M) rres[slot][pss] = ld8(p.R + (long)m * p.ldr + n0e + tcol[pss]);
    }
  };
#pragma unroll
  for (int i = 0; i < RD; ++i) {
#pragma unroll
    for (int pss = 0; pss < PASSES; ++pss) rres[i][pss] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (has_r) fetch_r(i, i);
  }
#pragma unroll
  for (int i = 0; i < NSI; ++i) {  // strip i = rows [i * SR, i * SR + SR) of the wave's row block
    float* ew = strips + ((EPI_DB ? (i & 1) * WM : 0) + wm) * STRIP;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if constexpr (CT) {
        *reinterpret_cast<f32x4*>(&ew[(el & 15) * EPI_LD + wn * (16 * NT) + j * 16 + (el >> 4) * 4]) = acc[i][j];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (MF == 16) ew[((el >> 4) * 4 + e) * EPI_LD + wn * (16 * NT) + j * 16 + (el & 15)] = acc[i][j][e];
          else ew[((el >> 5) * 4 + e) * EPI_LD + wn * (32 * NT) + j * 32 + (el & 31)] = acc[i / 4][j][(i & 3) * 4 + e];
        }
      }
    }
    lds_barrier();
#pragma unroll
    for (int pss = 0; pss < PASSES; ++pss) {
      const int row = trow[pss], col8 = tcol[pss];
      const int m = mrow0 + i * SR + row;
      const int n = n0e + col8;
      if (tval[pss] && m < p.M) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(&ew[row * EPI_LD + col8]);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(&ew[row * EPI_LD + col8 + 4]);
        if (p.splits > 1) {  // split-K partial: raw fp32 sums into this split's slab; finalize kernel does the epilogue
          float* cp = reinterpret_cast<float*>(p.C) + splite * p.slab_stride + (long)m * p.N + n;
          *reinterpret_cast<f32x4*>(cp) = v0;
          *reinterpret_cast<f32x4*>(cp + 4) = v1;
          continue;
        }
        float v[8];
        if (fold_bias) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = v0[e];
            v[e + 4] = v1[e];
          }
        } else {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_lds + col8);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias_lds + col8 + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = v0[e] * p.alpha + b0[e];
            v[e + 4] = v1[e] * p.alpha + b1[e];
          }
        }
        if (p.rowbias) {
          const int b = (PERSIST && !EARLY) ? (int)fdiv((unsigned)m, p.div_hw) : m / HWo;
          const bf16x8 rbv = ld8(p.rowbias + (long)b * p.ldrb + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bf2f(rbv[e]);
        }
        if (has_r) {
          const bf16x8 rv = rres[i % RD][pss];
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
          if constexpr (EARLY) NTST(reinterpret_cast<bf16*>(p.C) + (long)m * p.ldc + n, o);
          else st8(reinterpret_cast<bf16*>(p.C) + (long)m * p.ldc + n, o);
        }
      }
    }
    if (has_r && i + RD < NSI) fetch_r(i + RD, i % RD);
    if (!EPI_DB) lds_barrier();  // single strip buffer: everyone is done reading before it is rewritten
    STAMP(4 + i);
  }
  };  // epilogue()

  locate(vblock);
  load_bias(0);
  describe();
  issue(0, true);
  if constexpr (!PERSIST) {
    kloop(0);
    epilogue(m0, n0, tn, split, 0);
  } else {
    // a fixed grid of resident workgroups walks the tile list with stride gridDim.x
    for (int bbuf = 0;; bbuf ^= 1) {
      kloop(bbuf);
      const int m0e = m0, n0e = n0, tne = tn, splite = split;
      const int vnext = vblock + (int)gridDim.x;
      const bool more = vnext < p.total_blocks;
      if (more) {  // wave-uniform
        locate(vnext);
        load_bias(bbuf ^ 1);
        describe();
        issue(0, true);
      }
      STAMP(8);
      epilogue(m0e, n0e, tne, splite, bbuf);
      STAMP(12);
      if (!more) break;
      vblock = vnext;  // the next kloop() opens with __syncthreads: every wave has left the strips, stage 0 has landed
    }
  }
}

// The kernel.  Convolutions (EARLY == false): one tile per workgroup.  Linears and the fused GEGLU forms (EARLY == true,
// short K: a 256 x 320 tile lasts ~25 us, half of it the epilogue): a fixed grid of resident workgroups walks the tile
// list with stride gridDim.x inside nt2_tile, requesting the next tile's first K-step before the epilogue of the current
// one, so neither that load nor the epilogue's stores are waited for between tiles, and no workgroup is torn down and
// re-dispatched per tile.
template <int MT, int NT, int WM, int WN, int BK, bool UPS, int GEGLU = 0, bool EARLY = false, int MF = 16, bool PERSIST = EARLY, bool DE = false,
          bool HASR = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 16 ? 4 : 2)) void gemm_nt2_kernel(GemmNT2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  nt2_tile<MT, NT, WM, WN, BK, UPS, GEGLU, EARLY, MF, PERSIST, DE, HASR>(p, blockIdx.x, smem);
}

// split-K finalize: out[m][n] = alpha * sum_s slab[s][m][n] + bias[n] + rowbias[image(m)][n] + R[m][n]
__global__ void splitk_finalize_kernel(GemmNT2Params p, const float* ws) {
  const int nvec = p.N >> 3;
  const long total = (long)p.M * nvec;
  const int HWo = p.Hout * p.Wout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / nvec;
    const int n = (int)(i - m * nvec) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = 0; s < p.splits; ++s) {  // unrolled: the slab loads of several splits are in flight together
      const float* src = ws + s * p.slab_stride + m * p.N + n;
      const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += a[e];
        v[e + 4] += b[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + (p.bias ? p.bias[n + e] : 0.f);
    if (p.rowbias) {
      const bf16x8 rbv = ld8(p.rowbias + (m / HWo) * p.ldrb + n);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bf2f(rbv[e]);
    }
    if (p.R) {
      const bf16x8 rv = ld8(p.R + m * p.ldr + n);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bf2f(rv[e]);
    }
    if (p.out_fp32) {
      float* cp = reinterpret_cast<float*>(p.C) + m * p.ldc + n;
      *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
      st8(reinterpret_cast<bf16*>(p.C) + m * p.ldc + n, o);
    }
  }
}

// grid of the persistent forms: one resident workgroup per CU (g_nt_persist overrides the CU count; 0 = one tile per
// workgroup as in the convolution form)
static int persistent_grid(int total_blocks) {
  if (g_nt_persist == 0) return total_blocks;
  int n = g_nt_persist;
  if (n < 0) {
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return total_blocks;
    if (!cus[dev] && hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus[dev] = 256;
    n = da_usable_cus(cus[dev]);
  }
  return total_blocks < n ? total_blocks : n;
}

template <int MT, int NT, int WM, int WN, int BK, bool UPS, bool EARLY, int MF = 16, bool PERSIST = EARLY>
int launch_v2_mode(const GemmNT2Params& p0, int splits, float* ws, hipStream_t stream) {
  GemmNT2Params p = p0;
  constexpr int V2_BM = MF * MT * WM, V2_BK = BK, NTHREADS = 64 * WM * WN;
  constexpr int BN = MF * NT * WN;
  constexpr int STAGE = V2_BM * V2_BK * 2 + BN * V2_BK * 2, STRIPS = WM * (MF == 16 ? 16 : 8) * (BN + 4) * 4;
  // one tile per workgroup: 2 stages + bias row, strips inside the stages; persistent: strips behind stage 0, 2 bias rows
  constexpr int SMEM = PERSIST ? (STAGE + STRIPS > 2 * STAGE ? STAGE + STRIPS : 2 * STAGE) + 2 * BN * 4 : 2 * STAGE + BN * 4;
  static_assert(2 * STAGE >= STRIPS && SMEM <= 160 * 1024, "LDS map of nt2_tile");
  p.tiles_m = (p.M + V2_BM - 1) / V2_BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  static unsigned long long attr_done = 0;  // one bit per device
  if (da_ensure_dyn_smem((const void*)gemm_nt2_kernel<MT, NT, WM, WN, BK, UPS, 0, EARLY, MF, PERSIST>, SMEM, &attr_done) != DA_OK) return DA_ERR_LAUNCH;
  const int nk_total = p.K / V2_BK;
  p.splits = splits > 1 ? splits : 1;
  p.ksteps_per_split = (nk_total + p.splits - 1) / p.splits;
  p.splits = (nk_total + p.ksteps_per_split - 1) / p.ksteps_per_split;  // no empty splits
  p.slab_stride = (long)p.M * p.N;
  p.total_blocks = p.tiles_m * p.tiles_n * p.splits;
  p.div_nblk = make_fastdiv((unsigned)(p.tiles_m * p.tiles_n));
  p.div_tn = make_fastdiv((unsigned)p.tiles_n);
  if (PERSIST && (p.total_blocks >= (1 << 20) || p.M >= (1 << 24))) return DA_ERR_SHAPE;  // FastDiv range: n * d < 2^40
  const int grid = PERSIST ? persistent_grid(p.total_blocks) : p.total_blocks;
  if (p.splits > 1) {
    GemmNT2Params pk = p;
    pk.C = ws;  // partial slabs
    hipLaunchKernelGGL((gemm_nt2_kernel<MT, NT, WM, WN, BK, UPS, 0, EARLY, MF, PERSIST>), dim3(grid), dim3(NTHREADS), SMEM,
                       stream, pk);
    DA_CHECK_LAUNCH();
    const long total = (long)p.M * (p.N >> 3);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_finalize_kernel, dim3((int)blocks), dim3(256), 0, stream, p, (const float*)ws);
    DA_CHECK_LAUNCH();
    return DA_OK;
  }
  if constexpr (MF == 16 && ((NT % 2) == 0 || (MT % 2) == 0)) {
    // direct epilogue (see nt2_tile, DE).  Default (gemm_nt_de = 1): the convolution forms only - on the short-K linears it
    // measured -2...+2 % without a residual and -7...-13 % with one in isolation (its 16-row x 64-byte residual reads against
    // the strip path's whole 640-byte rows; profiles/r04_ab_nt_de.txt).  gemm_nt_de = 2 adds the linears without a residual,
    // 3 those with one (whole-step A/B switches).
    const bool form_ok = !EARLY || g_nt_de >= 3 || (g_nt_de == 2 && !p.R);
    const bool de = g_nt_de && form_ok && !p.out_fp32 && p.alpha == 1.0f && (p.N % 8) == 0 && (p.ldc % 8) == 0 && !((uintptr_t)p.C & 15) &&
                    (long)p.M * p.ldc * 2 < (1L << 32) &&
                    (!p.R || ((p.ldr % 8) == 0 && !((uintptr_t)p.R & 15) && (long)p.M * p.ldr * 2 < (1L << 32))) &&
                    (!p.rowbias || ((p.ldrb % 4) == 0 && !((uintptr_t)p.rowbias & 7)));
    if (de) {
      static unsigned long long attr_done_de[2] = {0, 0};
      if (p.R) {
        if (da_ensure_dyn_smem((const void*)gemm_nt2_kernel<MT, NT, WM, WN, BK, UPS, 0, EARLY, MF, PERSIST, true, true>, SMEM, &attr_done_de[1]) != DA_OK)
          return DA_ERR_LAUNCH;
        hipLaunchKernelGGL((gemm_nt2_kernel<MT, NT, WM, WN, BK, UPS, 0, EARLY, MF, PERSIST, true, true>), dim3(grid), dim3(NTHREADS), SMEM, stream, p);
      } else {
        if (da_ensure_dyn_smem((const void*)gemm_nt2_kernel<MT, NT, WM, WN, BK, UPS, 0, EARLY, MF, PERSIST, true, false>, SMEM, &attr_done_de[0]) != DA_OK)
          return DA_ERR_LAUNCH;
        hipLaunchKernelGGL((gemm_nt2_kernel<MT, NT, WM, WN, BK, UPS, 0, EARLY, MF, PERSIST, true, false>), dim3(grid), dim3(NTHREADS), SMEM, stream, p);
      }
      DA_CHECK_LAUNCH();
      return DA_OK;
    }
  }
  hipLaunchKernelGGL((gemm_nt2_kernel<MT, NT, WM, WN, BK, UPS, 0, EARLY, MF, PERSIST>), dim3(grid), dim3(NTHREADS), SMEM, stream, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

template <int GM>
int launch_v2_geglu(GemmNT2Params p, hipStream_t stream) {
  constexpr int BN = 320, STAGE = 256 * 64 * 2 + BN * 64 * 2, STRIPS = 4 * 16 * (BN + 4) * 4;
  constexpr int SMEM = (STAGE + STRIPS > 2 * STAGE ? STAGE + STRIPS : 2 * STAGE) + 2 * BN * 4;  // persistent LDS map of nt2_tile
  static_assert(SMEM <= 160 * 1024, "LDS");
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = GM == 1 ? (p.inner + BN / 2 - 1) / (BN / 2) : (p.inner + BN - 1) / BN;
  p.splits = 1;
  p.ksteps_per_split = p.K / 64;
  static unsigned long long attr_done = 0;  // one bit per device
  if (da_ensure_dyn_smem((const void*)gemm_nt2_kernel<4, 5, 4, 4, 64, false, GM, true>, SMEM, &attr_done) != DA_OK) return DA_ERR_LAUNCH;
  p.total_blocks = p.tiles_m * p.tiles_n;
  p.div_nblk = make_fastdiv((unsigned)p.total_blocks);
  p.div_tn = make_fastdiv((unsigned)p.tiles_n);
  if (p.total_blocks >= (1 << 20) || p.M >= (1 << 24)) return DA_ERR_SHAPE;  // FastDiv range: n * d < 2^40
  hipLaunchKernelGGL((gemm_nt2_kernel<4, 5, 4, 4, 64, false, GM, true>), dim3(persistent_grid(p.total_blocks)), dim3(1024), SMEM, stream, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

template <int MT, int NT, int WM, int WN, int BK, int MF = 16>
int launch_v2(const GemmNT2Params& p, int splits, float* ws, hipStream_t stream) {
  if (p.mode == 3) return launch_v2_mode<MT, NT, WM, WN, BK, true, false, MF>(p, splits, ws, stream);
  if ((long)p.N * p.K * 2 >= (1L << 32)) return DA_ERR_SHAPE;  // 32-bit byte offsets into W
  if constexpr (WM * WN == 16) {  // early issue only where it was measured: the 16-wave form on 1x1 shapes
    // (its A rows are addressed by 32-bit byte offsets from the base: larger activations take the generic form)
    if (p.ksize == 1 && (long)p.M * p.lda * 2 < (1L << 32)) return launch_v2_mode<MT, NT, WM, WN, BK, false, true, MF>(p, splits, ws, stream);
  }
  if constexpr (WM * WN == 16 && MF == 16 && MT == 4 && NT == 5) {
    // 3x3 convolutions on the 16-wave 256x320 form with more tiles than CUs: resident workgroups walk the tile list and
    // request the next tile's first K-step (and build its gather descriptors) ahead of the current tile's epilogue
    // (da_set_option("gemm_nt_persist_conv", 0) = one tile per workgroup, as before round 3)
    const long tiles = (long)((p.M + 255) / 256) * ((p.N + 319) / 320) * (splits > 1 ? splits : 1);
    if (g_nt_persist_conv && g_nt_persist != 0 && p.ksize == 3 && p.mode == 0 && tiles > da_usable_cus(256) && p.N % 320 == 0 &&
        p.M < (1 << 24) && (long)p.M * p.Hout * p.Wout < (1L << 40) &&   // FastDiv (common.hpp) exact while n * d < 2^40
        (long)(p.M / (p.Hout * p.Wout)) * p.Hin * p.Win * p.lda * 2 < (1L << 32))
      return launch_v2_mode<MT, NT, WM, WN, BK, false, false, MF, true>(p, splits, ws, stream);
  }
  return launch_v2_mode<MT, NT, WM, WN, BK, false, false, MF>(p, splits, ws, stream);
}

}  // namespace

int g_nt_korder = 1;  // da_set_option("gemm_nt_korder", 0 | 1)

// Called by da_gemm_nt (gemm_nt.hip) after argument validation.  variant: 4 -> BN 128, 5 -> BN 160, 10 -> BN 320.
int da_gemm_nt_v2_dispatch(int variant, int splits, float* ws, const void* A, long lda, const void* W, void* C, long ldc, const float* bias,
                           const void* rowbias, long ldrb, const void* R, long ldr, int M, int N, int K, int Cin,
                           int Hin, int Win, int Hout, int Wout, int ksize, int mode, int out_fp32, float alpha,
                           hipStream_t stream) {
  GemmNT2Params p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = C; p.bias = bias;
  p.rowbias = (const bf16*)rowbias; p.R = (const bf16*)R;
  p.lda = lda; p.ldc = ldc; p.ldrb = ldrb; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.Cin = Cin;
  p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout;
  p.ksize = ksize; p.mode = mode; p.out_fp32 = out_fp32; p.alpha = alpha;
  p.tiles_m = p.tiles_n = 0;
  p.splits = 1; p.ksteps_per_split = 0; p.slab_stride = 0;
  p.G = nullptr; p.ldg = 0; p.inner = 0;
  p.korder = g_nt_korder;
  p.total_blocks = 0;
  p.div_hw = make_fastdiv((unsigned)(Hout * Wout));
  p.div_w = make_fastdiv((unsigned)Wout);
  p.div_nblk = p.div_tn = make_fastdiv(1u);  // set by the launcher once the tile grid is known
  if (variant == 10) return launch_v2<8, 5, 2, 4, 64>(p, splits, ws, stream);
  if (variant == 11) return launch_v2<4, 10, 2, 2, 32>(p, 1, ws, stream);  // 128 x 320 x 32, 4 waves, 2 workgroups / CU
  if (variant == 12) return launch_v2<4, 5, 4, 4, 64>(p, splits, ws, stream);  // 256 x 320 x 64, 16 waves (4 / SIMD)
  if (variant == 15) return launch_v2<1, 5, 8, 2, 64, 32>(p, splits, ws, stream);     // 256 x 320 x 64, 16 waves as 8 x 2, 32x32x16 MFMA
  if (variant == 16) return launch_v2<2, 5, 4, 2, 64, 32>(p, splits, ws, stream);     // 256 x 320 x 64, 8 waves as 4 x 2 (64 x 160 each), 32x32x16 MFMA
  if (variant == 18) return launch_v2<3, 4, 8, 2, 64>(p, splits, ws, stream);         // 384 x 128 x 64, 16 waves as 8 x 2: N = 128 (VAE encoder, first level)
  if (variant == 14) return launch_v2<4, 4, 4, 4, 64>(p, splits, ws, stream);      // 256 x 256 x 64, 16 waves: N = 256 / 512 (VAE encoder)
  return variant == 5 ? launch_v2<4, 5, 4, 2, 64>(p, splits, ws, stream) : launch_v2<4, 4, 4, 2, 64>(p, splits, ws, stream);
}

int da_gemm_nt_geglu_ws_try(const void* A, long lda, const void* W, const float* bias, void* F, long ldf, void* G, long ldg,
                            int M, int inner, int K, hipStream_t stream);  // gemm_nt_ws.hip: -1 = not a shape for it

/* F[M][2*inner] = A[M][K] . W[2*inner][K]^T + bias ; G[M][inner] = F[:, :inner] * gelu(F[:, inner:]) in one launch */
extern "C" int da_gemm_nt_geglu(const void* A, long lda, const void* W, void* F, long ldf, void* G, long ldg,
                                const float* bias, int M, int inner, int K, hipStream_t stream) {
  DA_CLEAR_ERR();
  if (M <= 0 || inner <= 0 || K <= 0) return DA_ERR_SHAPE;
  if ((inner % 160) || (K % 64) || (lda & 7) || (ldf & 7) || (ldg & 7)) return DA_ERR_SHAPE;
  if ((long)M * lda * 2 >= (1L << 32) || (long)inner * K * 4 >= (1L << 32)) return DA_ERR_SHAPE;  // 32-bit byte offsets of the DMA sources
  {
    const int rc = da_gemm_nt_geglu_ws_try(A, lda, W, bias, F, ldf, G, ldg, M, inner, K, stream);  // weight-stationary form (K = 320)
    if (rc >= 0) return rc;
  }
  GemmNT2Params p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = F; p.bias = bias;
  p.rowbias = nullptr; p.R = nullptr;
  p.lda = lda; p.ldc = ldf; p.ldrb = 0; p.ldr = 0;
  p.M = M; p.N = 2 * inner; p.K = K; p.Cin = K;
  p.Hin = 1; p.Win = 1; p.Hout = 1; p.Wout = 1;
  p.ksize = 1; p.mode = 0; p.out_fp32 = 0; p.alpha = 1.0f;
  p.tiles_m = p.tiles_n = 0;
  p.splits = 1; p.ksteps_per_split = 0; p.slab_stride = 0;
  p.G = (bf16*)G; p.ldg = ldg; p.inner = inner; p.korder = 0; p.total_blocks = 0;
  p.div_hw = p.div_w = p.div_nblk = p.div_tn = make_fastdiv(1u);
  return launch_v2_geglu<1>(p, stream);
}

/* dF[M][2*inner] = geglu_bwd(F, dG) with dG[M][inner] = dY[M][K] . Wt[inner][K]^T never written to HBM */
extern "C" int da_gemm_nt_geglu_bwd(const void* dY, long lddy, const void* Wt, const void* F, long ldf, void* dF, long lddf,
                                    int M, int inner, int K, hipStream_t stream) {
  DA_CLEAR_ERR();
  if (M <= 0 || inner <= 0 || K <= 0) return DA_ERR_SHAPE;
  if ((inner % 320) || (K % 64) || (lddy & 7) || (ldf & 7) || (lddf & 7)) return DA_ERR_SHAPE;
  if ((long)M * lddy * 2 >= (1L << 32) || (long)inner * K * 2 >= (1L << 32)) return DA_ERR_SHAPE;  // 32-bit byte offsets of the DMA sources
  GemmNT2Params p;
  p.A = (const bf16*)dY; p.W = (const bf16*)Wt; p.C = dF; p.bias = nullptr;
  p.rowbias = nullptr; p.R = nullptr;
  p.lda = lddy; p.ldc = lddf; p.ldrb = 0; p.ldr = 0;
  p.M = M; p.N = inner; p.K = K; p.Cin = K;
  p.Hin = 1; p.Win = 1; p.Hout = 1; p.Wout = 1;
  p.ksize = 1; p.mode = 0; p.out_fp32 = 0; p.alpha = 1.0f;
  p.tiles_m = p.tiles_n = 0;
  p.splits = 1; p.ksteps_per_split = 0; p.slab_stride = 0;
  p.G = (bf16*)const_cast<void*>(F); p.ldg = ldf; p.inner = inner; p.korder = 0; p.total_blocks = 0;
  p.div_hw = p.div_w = p.div_nblk = p.div_tn = make_fastdiv(1u);
  return launch_v2_geglu<2>(p, stream);
}
