// gemm_nt v3: the STREAMING form of the linear layers with short K (C[M,N] = A[M,K] . W[N,K]^T + bias [+ R], bf16 in / out).
//
// Why a third form.  A [M,320] x [320,320] layer does 160 FLOP per byte it must move: it is bound by HBM, not by the
// matrix pipe, and the v2 kernel (gemm_nt_v2.hip: 256 x 320 tile, all 16 waves run the K loop, then all 16 run the
// epilogue) ran it at 3.4-4.2 TB/s of algorithmic bytes.  Timing-only builds of v2 (tools/build_alt.sh, round 4) showed where
// the rest goes: at 262144 x 320 x 320 the launch takes 95 us, 56 us with the output stores removed and 55 us with the A
// reads served from L2 as well - the K loop hides its reads completely, and the stores cost their full 168 MB / 39 us =
// 4.3 TB/s ON TOP, un-overlapped.  Cause: s_waitcnt vmcnt retires a wave's vector-memory operations IN ORDER, loads and
// stores alike, so the first K-step of the next tile (which must wait for its LDS-DMA tile) also waits for every store
// the wave issued before it - and a round of tiles (256 CUs x 164 KB = 42 MB) is more than the L2s hold, so those stores
// complete at the rate HBM takes them.  No epilogue scheduling fixes that (a direct register -> HBM epilogue measured the
// same time); the stores have to be YOUNGER than the loads the K loop waits for.
//
// The design.  128 x 320 x 64 tile, 16 waves as 4 x 4, each 32 x 80 = 2 x 5 v_mfma_f32_16x16x32_bf16 (transposed products:
// a lane's four registers of a 16 x 16 tile are four consecutive columns of one output row).  Half the accumulators of v2
// (40 registers per lane) leave room to keep the FINISHED previous tile - packed to bf16 and shuffled into 16 contiguous bytes per lane
// (v_permlane16_swap between neighbouring column tiles), 20 registers - through the next tile's K loop, and to store it
// there ONE 16-byte piece per K-step; the residual rows of the tile being computed arrive the same way (one or two
// 16-byte loads per step, same lane layout).  Every K-step issues, in this order,
//     B(t+1) -> LDS   (W slice, L2-resident, 2 stages)          A(t+2) -> LDS   (activation rows, HBM, 3 stages)
//     one store of the previous tile                            one or two residual loads of this tile
// computes step t, and ends with  s_waitcnt vmcnt(N) ; s_barrier  where N = the number of operations issued after
// B(t+1): the tile data of the next step has landed, everything younger - the A rows two steps ahead, the store, the
// residual loads - stays in flight across the barrier and gets a whole further step to complete.  Reads (A, R) and writes
// (C) are in flight together all the time; no phase of the kernel is store-only or load-only.  Resident workgroups walk the
// tile list (XCD-aware order), the prefetch cursors run across tile boundaries, the bias vector sits in LDS for the whole launch.
// Sums and roundings are those of gemm_nt_v2's epilogue (bias is the accumulators' start value, the residual is added in
// fp32 before the one rounding to bf16): bit-identical output.
#include "common.hpp"
#include "diffusion_amd.h"

int da_usable_cus(int cus);  // gemm_nt_v2.hip (da_set_option("reserve_cus"))
extern int g_nt_persist;

namespace {

struct GemmNT3Params {
  const bf16* A;
  const bf16* W;
  bf16* C;
  const float* bias;
  const bf16* R;
  long lda, ldc, ldr;
  int M, N, K;
  int tiles_m, tiles_n, total_tiles;
};

constexpr int T3_BM = 128, T3_BN = 320, T3_BK = 64;
constexpr int T3_A_STAGE = T3_BM * T3_BK * 2;   // 16 KiB
constexpr int T3_B_STAGE = T3_BN * T3_BK * 2;   // 40 KiB
constexpr int T3_NA = 3, T3_NB = 2;             // stages: A two steps ahead, B one
constexpr int T3_B_OFF = T3_NA * T3_A_STAGE;
constexpr int T3_BIAS_OFF = T3_B_OFF + T3_NB * T3_B_STAGE;
constexpr int T3_BIAS_MAX = 7168;               // floats (columns) of bias kept in LDS
constexpr int T3_SMEM = T3_BIAS_OFF + T3_BIAS_MAX * 4;
static_assert(T3_SMEM <= 160 * 1024, "LDS");

DEVINL void glds16_3(const void* gsrc, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// 16-byte load the compiler does not see: behind a tracked load it waits vmcnt(0) before every re-use of the destination
// registers (it cannot know about the counted waits below) and drains the tile requests just issued.  Completion is
// guaranteed by the step-end waits: an operation issued in step t is older than B(t+2), which the end of step t+1 waits for.
DEVINL u32x4 asm_load16(const void* uniform_base, unsigned byte_off) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(uniform_base) : "memory");
  return v;
}

// the step-end wait: all but the n youngest vector-memory operations of this wave are done; LDS reads done; barrier
DEVINL void step_sync(int n) {
  switch (n) {
    case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); break;
    case 64: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;   // no tile requests of this wave's own to wait for
    default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// LW: waves that issue the tile requests (16 = every wave its share; 4 = waves 0..3 only, one per SIMD - the other twelve
// then never wait on vmcnt in the K loop, so their stores stay in flight for as long as the memory system needs)
template <bool HASR, int LW>
__global__ __launch_bounds__(1024, 4) void gemm_nt3_kernel(GemmNT3Params p) {
  constexpr int NAJ = 16 / LW, NBJ = (40 + LW - 1) / LW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nk = p.K / T3_BK;
  const int grid = (int)gridDim.x;

  auto fresh_lane = [&]() {  // lane-derived constants are recomputed where used instead of living through the K loop
    int ln = lane;
    asm volatile("" : "+v"(ln));
    return ln;
  };
  // tile vb -> (m0, n0), XCD-aware: workgroups with equal (id % 8) share an XCD / L2 and take a contiguous run of tiles,
  // column tiles of one row block next to each other
  auto locate = [&](int vb, int& m0, int& n0) {
    const int nblk = p.total_tiles;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = vb & 7, idx = vb >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = bid / p.tiles_n;
    m0 = tm * T3_BM;
    n0 = (bid - tm * p.tiles_n) * T3_BN;
  };

  // ---- bias -> LDS, once (zeros when absent)
  {
    float* bl = reinterpret_cast<float*>(smem + T3_BIAS_OFF);
    for (int i = tid; i < p.tiles_n * T3_BN; i += 1024) bl[i] = (p.bias && i < p.N) ? p.bias[i] : 0.f;
  }

  // ---- prefetch cursors (wave-uniform position + this lane's source offsets); past the last tile they keep re-reading
  // their last valid source, so that every step issues the same number of requests
  int a_vb = blockIdx.x, a_k = 0, b_vb = blockIdx.x, b_k = 0;
  unsigned aoff[NAJ], woff[NBJ];
  const bool loader = wave < LW;
  auto a_describe = [&]() {
    int m0, n0;
    locate(a_vb, m0, n0);
    const int ln = fresh_lane();
#pragma unroll
    for (int j = 0; j < NAJ; ++j) {
      const int row = (wave + LW * j) * 8 + (ln >> 3);          // tile row of this lane's 16-byte chunk
#ifdef NT3_EXP_AHOT      // timing-only build: every tile reads the first 128 rows of A (L2-resident)
      const int m = row;
#else
      const int m = min(m0 + row, p.M - 1);                     // rows past M re-read the last row (never stored)
#endif
      aoff[j] = (unsigned)m * (unsigned)(p.lda * 2) + (unsigned)(((ln & 7) ^ ((row >> 1) & 7)) << 4);
    }
  };
  auto b_describe = [&]() {
    int m0, n0;
    locate(b_vb, m0, n0);
    const int ln = fresh_lane();
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
      const int row = (wave + LW * j) * 8 + (ln >> 3);          // row groups wave, wave+LW, ... (< 40)
      const int n = min(n0 + row, p.N - 1);
      woff[j] = (unsigned)n * (unsigned)(p.K * 2) + (unsigned)(((ln & 7) ^ ((row >> 1) & 7)) << 4);
    }
  };
  int a_issued = 0, b_issued = 0;  // steps issued so far (stage = count % stages)
  auto issue_a = [&]() {
    const char* src = reinterpret_cast<const char*>(p.A) + (unsigned)(a_k * (T3_BK * 2));
    char* dst = smem + (a_issued % T3_NA) * T3_A_STAGE;
#pragma unroll
    for (int j = 0; j < NAJ; ++j) glds16_3(src + aoff[j], dst + (wave + LW * j) * 1024);
    ++a_issued;
    if (++a_k == nk) {
      if (a_vb + grid < p.total_tiles) {
        a_vb += grid;
        a_k = 0;
        a_describe();
      } else {
        a_k = nk - 1;  // past the end: keep re-reading the last slice (lands in a stage nobody reads)
      }
    }
  };
  auto issue_b = [&]() {
    const char* src = reinterpret_cast<const char*>(p.W) + (unsigned)(b_k * (T3_BK * 2));
    char* dst = smem + T3_B_OFF + (b_issued % T3_NB) * T3_B_STAGE;
#pragma unroll
    for (int j = 0; j < NBJ; ++j)
      if ((LW * NBJ == 40) || wave + LW * j < 40) glds16_3(src + woff[j], dst + (wave + LW * j) * 1024);
    ++b_issued;
    if (++b_k == nk) {
      if (b_vb + grid < p.total_tiles) {
        b_vb += grid;
        b_k = 0;
        b_describe();
      } else {
        b_k = nk - 1;
      }
    }
  };

  // ---- the finished previous tile (outq) and the residual of the current one (rq), 16 bytes per lane and piece:
  // pieces 0,1 = strip 0 column pairs (0,1), (2,3); 2,3 = strip 1; 4 = column tile 4 of both strips
  u32x4 outq[5], rq[5];
  int pm0 = 0, pn0 = 0;      // tile the outq registers belong to
  bool have_prev = false;
  bool prev_whole = false;   // ... and whether it lies inside the matrix entirely (every lane of every wave stores)
  auto piece_off = [&](int piece, int m0, int n0, long ld, bool& ok) {
    const int ln = fresh_lane();
    const int q1 = (ln >> 4) & 1, lr = ln & 15, ch = ln >> 5;
    int m, n;
    if (piece < 4) {
      m = m0 + wm * 32 + (piece >> 1) * 16 + lr;
      n = n0 + wn * 80 + 32 * (piece & 1) + 16 * q1 + 8 * ch;
    } else {
      m = m0 + wm * 32 + q1 * 16 + lr;
      n = n0 + wn * 80 + 64 + 8 * ch;
    }
    ok = m < p.M && n < p.N;
#ifdef NT3_EXP_CHOT      // timing-only build: every tile stores to (and reads its residual from) the first 128 rows
    m -= m0;
#endif
    return (unsigned)min(m, p.M - 1) * (unsigned)(ld * 2) + (unsigned)min(n, p.N - 8) * 2u;
  };
  auto store_piece = [&](int piece) {
    bool ok;
    const unsigned off = piece_off(piece, pm0, pn0, p.ldc, ok);
#ifdef NT3_EXP_NOSTORE   // timing-only build (tools/build_alt.sh): everything but the stores
    asm volatile("" ::"v"(outq[piece]), "v"(off));
#else
    if (ok) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(p.C) + off) = outq[piece];
#endif
  };
  auto load_piece = [&](int piece, int m0, int n0) {
    bool ok;
    const unsigned off = piece_off(piece, m0, n0, p.ldr, ok);  // clamped, not predicated
    rq[piece] = asm_load16(p.R, off);
  };

  typedef float accv_t __attribute__((ext_vector_type(4)));
  accv_t acc[2][5];
  auto finish = [&](accv_t a, accv_t b, const u32x4 r) {
    if constexpr (HASR) {
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

  auto compute = [&](int a_stage, int b_stage) {
    const char* Ab = smem + a_stage * T3_A_STAGE;
    const char* Bb = smem + T3_B_OFF + b_stage * T3_B_STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      bf16x8 a[2], b[5];
      const int chunk = s * 4 + (ln >> 4);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wm * 32 + i * 16 + (ln & 15);
        a[i] = *reinterpret_cast<const bf16x8*>(Ab + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int row = wn * 80 + j * 16 + (ln & 15);
        b[j] = *reinterpret_cast<const bf16x8*>(Bb + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
  };

  // ---- prologue: A(0), B(0), A(1) of the first tile
  if (loader) {
    a_describe();
    b_describe();
    issue_a();
    issue_b();
    issue_a();
  }
  __syncthreads();  // (bias row written; also drains the three requests - once per launch)

  int g = 0;  // steps computed so far
  for (int vb = blockIdx.x; vb < p.total_tiles; vb += grid) {
    int m0, n0;
    locate(vb, m0, n0);
    {
      const float* brow = reinterpret_cast<const float*>(smem + T3_BIAS_OFF) + n0 + wn * 80 + (fresh_lane() >> 4) * 4;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(brow + j * 16);
        acc[0][j] = b4;
        acc[1][j] = b4;
      }
    }
    // Steps 0..4 carry the stores of the previous tile and the residual loads of this one; they are unrolled so that each
    // residual register is DEFINED (by the asm load) in straight-line code: inside a runtime `if (k == ...)` chain the
    // compiler joined the variants with register copies placed right behind the asm statement - copies of registers the
    // load had not filled yet (seen in the ISA of the first build).
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      if (loader) {
        issue_b();   // B(g+1)
        issue_a();   // A(g+2)
      }
      int after = NAJ;  // operations of this step issued after B: the A request(s), then ...
      if (have_prev) {  // ... one 16-byte piece of the finished previous tile
        store_piece(k);
        // a store whose lanes are all masked off (ragged last row / column tile) may be branched over: count it only
        // where it certainly executes - counting too few operations waits longer, never too short
        if (prev_whole) ++after;
      }
      if constexpr (HASR) {  // ... and this tile's residual: pieces 0,1 | 2 | 3 | 4 at steps 0..3 (a full step before use)
        if (k == 0) {
          load_piece(0, m0, n0);
          load_piece(1, m0, n0);
          after += 2;
        } else if (k < 4) {
          load_piece(k + 1, m0, n0);
          ++after;
        }
      }
      compute(g % T3_NA, g % T3_NB);
      step_sync(loader ? after : 64);
      ++g;
    }
    for (int k = 5; k < nk; ++k) {
      if (loader) {
        issue_b();
        issue_a();
      }
      compute(g % T3_NA, g % T3_NB);
      step_sync(loader ? NAJ : 64);
      ++g;
    }
    if constexpr (HASR && LW < 16) {
      // waves that issue no tile requests have waited for nothing so far: their residual loads are older than the one
      // store of step 4 (when that store was certainly issued), so at most that one may still be pending
      if (!loader) {
        if (have_prev && prev_whole) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    // the tile is complete: pack it (adding the residual) into the registers the NEXT tile's K loop stores from
    outq[0] = finish(acc[0][0], acc[0][1], rq[0]);
    outq[1] = finish(acc[0][2], acc[0][3], rq[1]);
    outq[2] = finish(acc[1][0], acc[1][1], rq[2]);
    outq[3] = finish(acc[1][2], acc[1][3], rq[3]);
    outq[4] = finish(acc[0][4], acc[1][4], rq[4]);
    pm0 = m0;
    pn0 = n0;
    have_prev = true;
    prev_whole = m0 + T3_BM <= p.M && n0 + T3_BN <= p.N;
  }
  if (have_prev) {
#pragma unroll
    for (int s = 0; s < 5; ++s) store_piece(s);
  }
  // the cursors ran past the last tile: nothing may still be writing this workgroup's LDS when it is handed on
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// da_set_option("gemm_nt_stream", 0 | 1 | 2): 0 (default) off, 1 for K <= 640 linears with >= 512 tiles, 2 wherever eligible.
// OFF by default: measured against the v2 form it is bit-identical and SLOWER (tools/lib_ab.py nt, profiles/r04_ab_nt_stream.txt):
// 262144 x 320 x 320 102 -> 108 us without a residual, 139 -> 153 us with one (every wave requesting tiles: 128 / 170 us).
// The timing-only builds say why the premise was only half right: with the stores removed it runs 71 us and with A served
// from L2 as well 66 us - so the loads ARE hidden - but the stores still cost +30 us although twelve of the sixteen waves
// never wait on vmcnt, and +16 us even when every tile stores to the same L2-resident 128 rows: it is the store path of
// the CU itself that the tile requests queue behind, not the wave's in-order counter alone.  Kept as an option with its
// bit-equality test; the next attempt at these shapes has to cut store INSTRUCTIONS or bytes, not reorder them.
int g_nt_stream = 0;
int g_nt_stream_lw = 4;  // da_set_option("gemm_nt_stream_lw", 4 | 16): waves that issue the tile requests

// Returns -1 when the shape is not one for this form (the caller falls through to gemm_nt_v2).
int da_gemm_nt_v3_try(const void* A, long lda, const void* W, void* C, long ldc, const float* bias, const void* R, long ldr,
                      int M, int N, int K, hipStream_t stream) {
  if (!g_nt_stream) return -1;
  if (K % 64 || K < 5 * 64 || (N & 7) || (lda & 7) || (ldc & 7) || (R && (ldr & 7))) return -1;
  if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)R) & 15) return -1;
  if ((long)M * lda * 2 >= (1L << 32) || (long)M * ldc * 2 >= (1L << 32) || (R && (long)M * ldr * 2 >= (1L << 32)) ||
      (long)N * K * 2 >= (1L << 32))
    return -1;
  GemmNT3Params p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.C = (bf16*)C; p.bias = bias; p.R = (const bf16*)R;
  p.lda = lda; p.ldc = ldc; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K;
  p.tiles_m = (M + T3_BM - 1) / T3_BM;
  p.tiles_n = (N + T3_BN - 1) / T3_BN;
  if (p.tiles_n * T3_BN > T3_BIAS_MAX) return -1;
  const long tiles = (long)p.tiles_m * p.tiles_n;
  if (tiles >= (1 << 24)) return -1;
  p.total_tiles = (int)tiles;
  int ncu = 256, dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  int grid = g_nt_persist > 0 ? g_nt_persist : da_usable_cus(ncu);
  if (grid > p.total_tiles) grid = p.total_tiles;
  static unsigned long long attr_done[4] = {0, 0, 0, 0};
#define NT3_LAUNCH(HR, LWV, SLOT)                                                                                        \
  do {                                                                                                                   \
    if (da_ensure_dyn_smem((const void*)gemm_nt3_kernel<HR, LWV>, T3_SMEM, &attr_done[SLOT]) != DA_OK) return DA_ERR_LAUNCH; \
    hipLaunchKernelGGL((gemm_nt3_kernel<HR, LWV>), dim3(grid), dim3(1024), T3_SMEM, stream, p);                           \
  } while (0)
  if (g_nt_stream_lw == 4) {
    if (R) NT3_LAUNCH(true, 4, 0); else NT3_LAUNCH(false, 4, 1);
  } else {
    if (R) NT3_LAUNCH(true, 16, 2); else NT3_LAUNCH(false, 16, 3);
  }
#undef NT3_LAUNCH
  DA_CHECK_LAUNCH();
  return DA_OK;
}
