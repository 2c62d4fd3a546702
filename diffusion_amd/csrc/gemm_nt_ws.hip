// gemm_nt_ws: weight-stationary linear layers for K = 320 and K = 640 (the level-0 / level-1 transformer widths);
// da_set_option("gemm_nt_ws", 0 | 1), on.   C[m][n] = sum_k A[m][k] * W[n][k]  (+ bias[n]) (+ R[m][n]),  bf16 in / out, fp32 sums.
// Written for K = 320 (N = 320 ... 1280); the K = 640 instantiation follows at the end of this comment.
// The tiled form (gemm_nt_v2.hip) stages a 40 KB slice of W per K-step next to the activation rows and re-reads both as MFMA
// fragments; at K = N = 320 the whole weight is 200 KB = 200 registers per lane of FOUR waves (one per SIMD, which gives
// each the whole 512-register file): wave w keeps the fragments of output columns 80w .. 80w+79 over all of K for the life of
// the workgroup, the LDS holds nothing but activation rows (a ring of 32-row x 640-byte stages requested three tiles ahead
// by LDS-DMA), and a 32 x 320 tile is 10 K-steps of 2 ds_read_b128 + 10 v_mfma_f32_16x16x32_bf16 per wave.  Products are taken
// transposed (W fragment first), so a lane's four accumulator registers are four consecutive columns of one row and leave
// through the direct epilogue of gemm_nt_v2.hip (bf16 pack, permlane16_swap between neighbouring column tiles, 16 bytes per
// lane).  One wave per SIMD has nobody to fill the gaps between its MFMAs, so the kernel is a two-buffer software pipeline:
// the products of tile k run beside the residual fetch, conversion, shuffle and stores of tile k - 1.  Every vector-memory
// operation is counted by hand (per step and wave 5 residual loads through inline asm, 5 tile requests, 5 stores) so that the
// waits leave the younger ones in flight.  Same products, same order of the K sum, same roundings as the tiled form:
// bit-identical (tests/test_kernels_gpu.py).  Measured (profiles/r04_ab_nt_ws.txt): 262144 x 320 x 320 88 -> 72 us (4.7 TB/s of
// its own bytes), with a residual 122 -> 97 (5.2 TB/s), 262144 x 960 x 320 234 -> 189; the step 168.8 -> 167.5 ms.
// K = 640 (WsCfg<20, 2>, opt-in: bit 1 of the option): a workgroup holds W for 128 output columns (160 KB; wave w columns
// 32w .. 32w+31 over all of K), the stages are 32 rows x 1280 B (40 KB, ring of four = the whole LDS, chunk XOR row & 15),
// N = 640 is five column blocks whose workgroups share an XCD and read the activation rows through its L2.  Bit-identical
// too; 65536 x 640 x 640 54 us against 57 us tiled in isolation (each of its 43 tiles per workgroup costs 10 tile requests
// and a barrier for 80 MFMAs per wave), the step -0.2 %: not switched on.
#include <type_traits>
#include "common.hpp"
#include "diffusion_amd.h"

int da_usable_cus(int cus);  // gemm_nt_v2.hip
int g_nt_ws = 1;             // da_set_option("gemm_nt_ws", bits): 1 = the K = 320 linears (on), 2 = K = 640 (off), 4 = the fused GEGLU forward at K = 320, 8 = K = 640 as two unpipelined workgroups per CU

namespace {

struct GemmWsParams {
  const bf16* A;
  const bf16* W;
  const float* bias;
  const bf16* R;
  bf16* C;
  long lda, ldr, ldc;
  int M, N, tiles_m, nb;
  bf16* G;      // MODE 1 (fused GEGLU forward): C = F[M][2*inner] (pre-activation), G[M][inner] = F[:, :inner] * gelu(F[:, inner:])
  long ldg;
  int inner;
};

constexpr int WS_BM = 32;  // rows per tile: two 16-row MFMA strips
#ifndef WS_NS_BUILD
#define WS_NS_BUILD 4
#endif
constexpr int WS_NS = WS_NS_BUILD;   // LDS stages (ring); 4 / 6 / 8 measured equal at K = 320 (tools/build_alt.sh -DWS_NS_BUILD=n)
constexpr int WS_AHEAD = WS_NS - 1;  // tiles requested ahead of the one being computed

// KS = K / 32 MFMA K-steps, NT = 16-column MFMA tiles per wave
// MODE 1 = fused GEGLU forward: a wave's NT column tiles are NT/2 tiles of VALUE columns and the NT/2 tiles of the GATE columns of the
// same hidden units (so the gating is lane-local); BN then counts hidden units per workgroup
template <int KS, int NT, int MODE = 0>
struct WsCfg {
  static constexpr int K = 32 * KS, BN = MODE == 1 ? 32 * NT : 64 * NT, ROWB = 64 * KS, STAGE = WS_BM * ROWB;
  static constexpr int PD = STAGE / 4096;          // LDS-DMA instructions per wave and tile
  static constexpr int NPJ = NT / 2, ODD = NT & 1;  // column-tile pairs per strip; a last tile paired across the two strips
  // 16-byte output pieces per lane and tile (= residual loads = stores); MODE 1: per strip F value, F gate and G, NT/4 pairs each
  static constexpr int PO = MODE == 1 ? 6 * (NT / 4) : 2 * NPJ + ODD;
  static_assert(MODE == 0 || NT == 4, "GEGLU form: two value + two gate tiles per wave");
  // source-side XOR of the 16-byte chunk index that makes ds_read_b128 of 16 consecutive rows conflict-free: rows of an odd
  // number of 128-byte halves (640 B) need three bits, (row >> 1) & 7; rows of whole 256-byte bank rows (1280 B) four, row & 15
  static constexpr int SWB = (ROWB % 256 == 0) ? 4 : 3;
  static constexpr int PER = 1 << (SWB - 2);       // fragment base registers per lane (K-steps s, s + PER, ... share one)
  // SPREAD: the PD requests of a step go out one per K-step, unconditionally (past the end the last tile is requested again:
  // an L2 hit into a stage nobody reads - no branch in the product stream, constant counts), instead of one block in front
  // of the products.  Measured: K = 640 (10 requests per 80 MFMAs) 61.8 -> 54.3 us at 65536 x 640 x 640; K = 320 (5 per 100)
  // +-1 %; the same behind a uniform branch per piece -7...-13 %.
#ifdef WS_SPREAD
  static constexpr bool SPREAD = WS_SPREAD != 0;
#else
  static constexpr bool SPREAD = KS >= 20;
#endif
  static_assert(STAGE % 4096 == 0 && WS_NS * STAGE <= 160 * 1024 && (ROWB / 16) % (1 << SWB) % 8 == 0, "stage geometry");
};

DEVINL void glds16_ws(const void* gsrc, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// a load the compiler does not count (see gemm_nt_v3.hip): kept in flight behind the hand-counted waits below
DEVINL u32x4 ws_load16(const void* uniform_base, unsigned byte_off) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(uniform_base) : "memory");
  return v;
}
// ds_read_b128 through inline asm with an immediate offset: the compiler's scheduler sinks plain LDS loads down to their
// first use when registers are tight (every K-step then waited lgkmcnt(0) for the fragment it had just requested: 20 exposed
// LDS round trips per tile, 1.9 us per tile for 0.76 us of MFMA); the asm form stays where it is written, one K-step ahead,
// and is ordered with lds_wait_for<N>() (common.hpp).
template <int OFF>
DEVINL bf16x8 ws_lds16(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int B, int E, typename F>
DEVINL void ws_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    ws_static_for<B + 1, E>(f);
  }
}

DEVINL void ws_wait_vm(int n) {  // all but the n youngest vector-memory operations of this wave are complete
  switch (n) {
#define WS_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
#define WS_W10(D) WS_W(D##0) WS_W(D##1) WS_W(D##2) WS_W(D##3) WS_W(D##4) WS_W(D##5) WS_W(D##6) WS_W(D##7) WS_W(D##8) WS_W(D##9)
    WS_W(1) WS_W(2) WS_W(3) WS_W(4) WS_W(5) WS_W(6) WS_W(7) WS_W(8) WS_W(9)
    WS_W10(1) WS_W10(2) WS_W10(3) WS_W10(4) WS_W10(5) WS_W(60)
#undef WS_W10
#undef WS_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // never more than it is safe to leave
  }
}

// PIPE = false: no software pipeline and ONE accumulator set - the kernel then fits 256 registers and runs as TWO workgroups per
// CU (two waves per SIMD fill each other's gaps), with two LDS stages each (one tile ahead).
template <int KS, int NT, bool HASR, int MODE = 0, bool PIPE = true>
__global__ __launch_bounds__(256, PIPE ? 1 : 2) void gemm_nt_ws_kernel(GemmWsParams p) {
  typedef WsCfg<KS, NT, MODE> G;
  constexpr int NS = PIPE ? WS_NS : 2, AHEAD = NS - 1;  // LDS stages of this form; tiles requested ahead
  static_assert(MODE == 0 || !HASR, "no residual in the GEGLU form");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NS stages of 32 activation rows
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  typedef float accv_t __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g4 = lane >> 4, l15 = lane & 15;
  // Workgroup -> (slot, column block).  A slot is a walk over the row tiles slot, slot + grid, ...; with N = nb x BN the nb
  // workgroups of a slot hold different weight columns and read the SAME activation rows, so they are placed on one XCD
  // (workgroup ids equal mod 8 share an XCD and its L2): per XCD gridDim.x / 8 workgroups = spx slots x nb column blocks,
  // the remainder (2 of 32 at nb = 3 or 5) exits - the rows are then fetched from HBM once, not nb times.
  const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
  const int spx = ((int)gridDim.x >> 3) / p.nb;
  if (idx >= spx * p.nb) return;
  const int n0 = (idx % p.nb) * G::BN;
  const int slot = xcd * spx + idx / p.nb;
  const int grid = 8 * spx;
  if (slot >= p.tiles_m) return;
  const int n_my = (p.tiles_m - slot + grid - 1) / grid;  // tiles of this workgroup: slot + k*grid

  // ---- tile requests: piece j of this wave covers LDS bytes (wave*PD + j)*1024 + lane*16 of the stage image [row][ROWB],
  // physical 16-byte chunk pc of row r holding logical chunk pc ^ swz(r)
  unsigned dsrc[G::PD];
#pragma unroll
  for (int j = 0; j < G::PD; ++j) {
    const int byte = (wave * G::PD + j) * 1024 + lane * 16;
    const int row = byte / G::ROWB, pc = (byte - row * G::ROWB) >> 4;
    const int swz = G::SWB == 3 ? (row >> 1) & 7 : row & 15;
    dsrc[j] = (unsigned)row * (unsigned)(p.lda * 2) + (unsigned)((pc ^ swz) << 4);
  }
  auto request = [&](int k) {  // k-th tile of this workgroup -> stage k % NS
    const char* base = reinterpret_cast<const char*>(p.A) + (long)(slot + k * grid) * WS_BM * p.lda * 2;
    char* dst = smem + (k % NS) * G::STAGE + wave * (G::PD * 1024);
#pragma unroll
    for (int j = 0; j < G::PD; ++j) glds16_ws(base + dsrc[j], dst + j * 1024);
  };
#ifdef WS_EXP_AHOT      // timing-only build: every request re-reads the workgroup's first tile (L2-resident)
#define WS_REQ(k) request_hot(k)
  auto request_hot = [&](int k) {
    const char* base = reinterpret_cast<const char*>(p.A) + (long)slot * WS_BM * p.lda * 2;
    char* dst = smem + (k % NS) * G::STAGE + wave * (G::PD * 1024);
#pragma unroll
    for (int j = 0; j < G::PD; ++j) glds16_ws(base + dsrc[j], dst + j * 1024);
  };
#else
#define WS_REQ(k) request(k)
#endif
  // Vector-memory operations this wave issues AFTER the requests of its tile k (k >= AHEAD; issued in step k - AHEAD):
  // what the wait in front of tile k may leave in flight.  A step j issues, in this order: the PO residual loads of tile j - 1
  // (j >= 1, HASR), the PD requests of tile j + AHEAD (while there is one), the PO stores of tile j - 1 (j >= 1).
  // (A pure function of k on purpose: running counters captured by the step lambda ended up in scratch memory, and every
  // scratch load comes with s_waitcnt vmcnt(0).)
  auto after_requests_of = [&](int k) {
    if constexpr (!PIPE) return (int)G::PO;  // the requests of tile k go out in the compute part of step k - 1: only its stores follow
    const int j0 = k - AHEAD;
    int n = j0 >= 1 ? G::PO : 0;
#pragma unroll
    for (int d = 1; d < AHEAD; ++d)
      n += (HASR ? G::PO : 0) + ((G::SPREAD || j0 + d + AHEAD < n_my) ? G::PD : 0) + G::PO;
    return n < 60 ? n : 60;  // the counter has 6 bits; leaving fewer in flight than allowed is always safe
  };

  // the first tiles are requested BEFORE the 160-200 KB of weight fragments (every workgroup fetches them at once - a few
  // microseconds during which HBM would otherwise idle), then everything is waited for together
#pragma unroll
  for (int k = 0; k < AHEAD; ++k)
    if (k < n_my) WS_REQ(k);
  // ---- the resident weight fragments: W rows n0 + 16*NT*wave + 16*jt + (lane & 15), k = 32*s + 8*(lane >> 4) .. + 7
  // (MODE 1: n0 is the workgroup's first hidden unit; tiles jt < NT/2 are its value rows of W, the others the gate rows)
  auto wcol = [&](int jt) {  // first output column (= row of W) of this wave's tile jt
    if constexpr (MODE == 1) return (jt < NT / 2 ? 0 : p.inner) + n0 + 8 * NT * wave + 16 * (jt % (NT / 2));
    else return n0 + 16 * NT * wave + 16 * jt;
  };
  bf16x8 wf[NT][KS];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const bf16* wp = p.W + (long)(wcol(jt) + l15) * G::K + 8 * g4;
#pragma unroll
    for (int s = 0; s < KS; ++s) wf[jt][s] = ld8(wp + 32 * s);
  }
  // bias of this lane's accumulator columns n0 + 16*NT*wave + 16*jt + 4*(lane >> 4) + e: the start value of every K sum
  accv_t bv[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[jt][e] = p.bias ? p.bias[wcol(jt) + 4 * g4 + e] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the counted waits below start from zero
  // ... and the compiler's own bookkeeping too: redefined here, the fragments are no pending loads to it (it would otherwise
  // wait vmcnt(0) for them in front of the first product - behind the tile requests in flight)
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wf[jt][s]));
    asm volatile("" : "+v"(bv[jt]));
  }

  // ---- this lane's 16 bytes of the output tile (the direct epilogue of gemm_nt_v2.hip with wm = 0, MT = 2, wn = wave):
  // pieces q = (strip i, column pair pj): q < NPJ -> (0, q), q < 2*NPJ -> (1, q - NPJ), q = 2*NPJ (NT odd) -> the last column
  // tile of both strips
  const int lq = g4 & 1, lcol = (MODE == 1 ? 8 : 16) * NT * wave + 8 * (lane >> 5);
  auto out_off = [&](int q, long ld) {
    if constexpr (MODE == 1) {  // q = 3*i + kind (F value, F gate, G): the value-tile pair's position; kind 1 adds inner columns
      const int row = l15 + 16 * (q / 3), col = lcol + 16 * lq + ((q % 3) == 1 ? p.inner : 0);
      return (unsigned)row * (unsigned)(ld * 2) + (unsigned)(n0 + col) * 2u;
    }
    const bool odd = q >= 2 * G::NPJ;
    const int i = q < G::NPJ ? 0 : 1, pj = q < G::NPJ ? q : q - G::NPJ;
    const int row = odd ? l15 + 16 * lq : l15 + 16 * i;
    const int col = odd ? lcol + 16 * (NT - 1) : lcol + 32 * pj + 16 * lq;
    return (unsigned)row * (unsigned)(ld * 2) + (unsigned)(n0 + col) * 2u;
  };
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

  // fragment (strip i, K-step s) = 16 bytes at row 16*i + (lane & 15), chunk (4*s + (lane >> 4)) ^ swz of the stage, swz the
  // lane's row swizzle.  With lo = (lane >> 4) ^ (swz & 3) and h = swz >> 2 that is byte 16*lo + 64*((s % PER) ^ h) +
  // 64*PER*(s / PER) of the row: PER base registers per lane and an IMMEDIATE - no address arithmetic in the K loop
  const unsigned smem_off = lds_offset(smem);
  const int fsw = G::SWB == 3 ? (lane >> 1) & 7 : l15;
  const unsigned frag_off = (unsigned)(l15 * G::ROWB) + (unsigned)((g4 ^ (fsw & 3)) << 4);
  typedef accv_t acc_t[2][NT];
  typedef std::integral_constant<bool, true> yes_t;
  typedef std::integral_constant<bool, false> no_t;

  // One step of the software pipeline: the products of tile k (-> cur) run beside the epilogue of tile k - 1 (<- prev): a
  // single wave per SIMD has nobody else to fill the gaps between its MFMAs, so the conversion, shuffle and store
  // instructions of the finished tile are placed between the products of the next one (the last PO K-steps; the residual of
  // the finished tile is requested at the start of the step and waited for in front of the first piece).
  u32x4 rin[G::PO];        // residual of the tile whose epilogue comes next
  bool requested = false;  // ... and whether tile requests were issued after its loads (they are the younger operations)
  auto step = [&](acc_t& cur, acc_t& prev, auto do_compute, auto do_epi, const int k) {
    constexpr bool C = decltype(do_compute)::value, E = decltype(do_epi)::value;
    const int st = k % NS;
    if constexpr (C) {
      if (k >= AHEAD) ws_wait_vm(after_requests_of(k));  // (the first AHEAD tiles landed behind the weight fragments)
      __builtin_amdgcn_s_barrier();  // everybody's pieces of tile k are in LDS; everybody is done with the stage requested next
      asm volatile("" ::: "memory");
    }
    // PIPE: the residual of tile k - 1 (its epilogue runs in this step); otherwise of tile k, ahead of its own products
    if constexpr (HASR && (PIPE ? E : C)) {
      const char* rb = reinterpret_cast<const char*>(p.R) + (long)(slot + (PIPE ? k - 1 : k) * grid) * WS_BM * p.ldr * 2;
#pragma unroll
      for (int q = 0; q < G::PO; ++q) rin[q] = ws_load16(rb, out_off(q, p.ldr));
    }
    // the PD requests of tile k + AHEAD: one block here, or (G::SPREAD) one per K-step below
    if constexpr (C || PIPE) requested = C && G::SPREAD;  // (an epilogue-only step of the unpipelined form keeps the compute part's)
    const int kreq = min(k + AHEAD, n_my - 1);
    const char* req_base = reinterpret_cast<const char*>(p.A) + (long)(slot + kreq * grid) * WS_BM * p.lda * 2;
    char* req_dst = smem + ((k + AHEAD) % NS) * G::STAGE + wave * (G::PD * 1024);
    static_assert(!G::SPREAD || G::PD <= KS - G::PO, "spread requests go out before the first output piece");
    if constexpr (C && !G::SPREAD) {
      if (k + AHEAD < n_my) {
        WS_REQ(k + AHEAD);
        requested = true;
      }
    }
    unsigned fbase[G::PER];
#pragma unroll
    for (int m = 0; m < G::PER; ++m) fbase[m] = smem_off + (unsigned)(st * G::STAGE) + frag_off + (unsigned)((m ^ (fsw >> 2)) << 6);
    char* cb = reinterpret_cast<char*>(p.C) + (long)(slot + (k - 1) * grid) * WS_BM * p.ldc * 2;
    auto piece = [&](int q) {  // the finished tile's 16-byte pieces in strip order
      u32x4 o;
      char* dst = cb + out_off(q, p.ldc);
      if constexpr (MODE == 1) {
        // value and gate rounded to bf16 BEFORE gating, exactly as the tiled fused form (gemm_nt_v2.hip, GEGLU == 1)
        const int i = q / 3, kind = q % 3;
        accv_t x[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float fv = bf2f(f2bf(prev[i][t][e])), fg = bf2f(f2bf(prev[i][2 + t][e]));
            x[t][e] = kind == 0 ? fv : kind == 1 ? fg : fv * gelu_f(fg);
          }
        o = finish(x[0], x[1], rin[q]);
        if (kind == 2) dst = reinterpret_cast<char*>(p.G) + (long)(slot + (k - 1) * grid) * WS_BM * p.ldg * 2 + out_off(q, p.ldg);
      } else {
        const bool odd = q >= 2 * G::NPJ;
        const int i = q < G::NPJ ? 0 : 1, pj = q < G::NPJ ? q : q - G::NPJ;
        o = odd ? finish(prev[0][NT - 1], prev[1][NT - 1], rin[q]) : finish(prev[i][2 * pj], prev[i][2 * pj + 1], rin[q]);
      }
#ifdef WS_EXP_NOSTORE   // timing-only build (tools/build_alt.sh): everything but the stores
      asm volatile("" ::"v"(o), "v"(dst));
#else
      *reinterpret_cast<u32x4*>(dst) = o;
#endif
    };
    bf16x8 a_cur[2], a_nxt[2];
    if constexpr (C) {
      a_cur[0] = ws_lds16<0>(fbase[0]);
      a_cur[1] = ws_lds16<16 * G::ROWB>(fbase[0]);
    }
    ws_static_for<0, KS>([&](auto S) {
      constexpr int s = decltype(S)::value;
      if constexpr (s == KS - G::PO && E && HASR) {
        // the residual has landed (only this step's requests are younger).  The registers are operands of the wait: without
        // that tie nothing stops the compiler from scheduling their first use in front of it.  ONE statement for both counts
        // (a run-time branch between two tied statements makes the compiler join their register operands with copies -
        // placed in front of the wait, they would read registers the loads have not filled)
        static_assert(G::PO == 5 || G::PO == 2, "operands of the residual wait");
        if constexpr (G::PO == 5)
          asm volatile("s_waitcnt vmcnt(%6)\n\t"
                       "s_cmp_lg_u32 %5, 0\n\t"
                       "s_cbranch_scc1 .Lws_r%=\n\t"
                       "s_waitcnt vmcnt(0)\n"
                       ".Lws_r%=:"
                       : "+v"(rin[0]), "+v"(rin[1]), "+v"(rin[2]), "+v"(rin[3]), "+v"(rin[G::PO - 1])
                       : "s"(__builtin_amdgcn_readfirstlane((int)requested)), "n"(G::PD)
                       : "scc", "memory");
        else
          asm volatile("s_waitcnt vmcnt(%3)\n\t"
                       "s_cmp_lg_u32 %2, 0\n\t"
                       "s_cbranch_scc1 .Lws_r%=\n\t"
                       "s_waitcnt vmcnt(0)\n"
                       ".Lws_r%=:"
                       : "+v"(rin[0]), "+v"(rin[G::PO - 1])
                       : "s"(__builtin_amdgcn_readfirstlane((int)requested)), "n"(G::PD)
                       : "scc", "memory");
      }
      if constexpr (C && G::SPREAD && s < G::PD) glds16_ws(req_base + dsrc[s], req_dst + s * 1024);
      if constexpr (C) {
        if constexpr (s + 1 < KS) {
          constexpr int OFF = 64 * G::PER * ((s + 1) / G::PER);
          a_nxt[0] = ws_lds16<OFF>(fbase[(s + 1) % G::PER]);
          a_nxt[1] = ws_lds16<OFF + 16 * G::ROWB>(fbase[(s + 1) % G::PER]);
          lds_wait_for<2>(a_cur[0], a_cur[1]);   // this step's fragments are here; the two just requested stay in flight
        } else {
          lds_wait_for<0>(a_cur[0], a_cur[1]);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (C) {
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
            cur[i][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jt][s], a_cur[i], s == 0 ? bv[jt] : cur[i][jt], 0, 0, 0);
        }
        if constexpr (E) {
          if (s >= KS - G::PO && i == 1) piece(s - (KS - G::PO));
        }
      }
      if constexpr (C) {
        a_cur[0] = a_nxt[0];
        a_cur[1] = a_nxt[1];
      }
    });
  };

  acc_t accA;
  if constexpr (!PIPE) {
    for (int k = 0; k < n_my; ++k) {
      step(accA, accA, yes_t{}, no_t{}, k);
      step(accA, accA, no_t{}, yes_t{}, k + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  acc_t accB;
  step(accA, accB, yes_t{}, no_t{}, 0);
  for (int k = 1;;) {
    if (k >= n_my) { step(accB, accA, no_t{}, yes_t{}, k); break; }
    step(accB, accA, yes_t{}, yes_t{}, k);
    ++k;
    if (k >= n_my) { step(accA, accB, no_t{}, yes_t{}, k); break; }
    step(accA, accB, yes_t{}, yes_t{}, k);
    ++k;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#undef WS_REQ

template <int KS, int NT, bool PIPE = true>
int launch_ws(GemmWsParams p, int N, int cus, hipStream_t stream) {
  typedef WsCfg<KS, NT> G;
  const int nb = N / G::BN, wgs = PIPE ? cus : 2 * cus;  // workgroups: one per CU, or two (the unpipelined 256-register form)
  if (nb < 1 || nb > 8 || N % G::BN) return -1;
  if ((long)p.tiles_m * nb < 8L * wgs) return -1;  // each workgroup loads 160-200 KB of W before its first 20-40 KB tile
  p.nb = nb;
  const int gx = (wgs / 8) * 8;  // whole rounds of the 8 XCDs (see the kernel's slot mapping)
  if (gx < 8 * nb) return -1;
  constexpr int SMEM = (PIPE ? WS_NS : 2) * G::STAGE;
  static unsigned long long attr_done[2] = {0, 0};
  if (p.R) {
    if (da_ensure_dyn_smem((const void*)gemm_nt_ws_kernel<KS, NT, true, 0, PIPE>, SMEM, &attr_done[1]) != DA_OK) return DA_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_nt_ws_kernel<KS, NT, true, 0, PIPE>), dim3(gx), dim3(256), SMEM, stream, p);
  } else {
    if (da_ensure_dyn_smem((const void*)gemm_nt_ws_kernel<KS, NT, false, 0, PIPE>, SMEM, &attr_done[0]) != DA_OK) return DA_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_nt_ws_kernel<KS, NT, false, 0, PIPE>), dim3(gx), dim3(256), SMEM, stream, p);
  }
  DA_CHECK_LAUNCH();
  return DA_OK;
}

}  // namespace

// Called by da_gemm_nt (gemm_nt.hip) for plain bf16 linears; -1 = not eligible (the tiled form takes the call).
int da_gemm_nt_ws_try(const void* A, long lda, const void* W, const float* bias, const void* R, long ldr, void* C, long ldc,
                      int M, int N, int K, hipStream_t stream) {
  if (!g_nt_ws || (K != 320 && K != 640) || M % WS_BM) return -1;  // whole 32-row tiles only
  if ((g_nt_ws & (K == 320 ? 1 : 2)) == 0) return -1;             // bit 0: the K = 320 form, bit 1: K = 640
  const int cus = da_usable_cus(256);
  if ((lda & 7) || (ldc & 7) || (R && (ldr & 7)) || ((size_t)A & 15) || ((size_t)C & 15) || ((size_t)W & 15) || (R && ((size_t)R & 15)))
    return -1;
  if (64 * lda * 2 >= (1L << 31) || 64 * ldc * 2 >= (1L << 31) || (R && 64 * ldr * 2 >= (1L << 31))) return -1;
  GemmWsParams p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.bias = bias; p.R = (const bf16*)R; p.C = (bf16*)C;
  p.lda = lda; p.ldr = ldr; p.ldc = ldc; p.M = M; p.N = N; p.tiles_m = M / WS_BM; p.nb = 1;
  p.G = nullptr; p.ldg = 0; p.inner = 0;
  if (K == 320) return N <= 1280 ? launch_ws<10, 5>(p, N, cus, stream) : -1;
  return (g_nt_ws & 8) ? launch_ws<20, 2, false>(p, N, cus, stream) : launch_ws<20, 2>(p, N, cus, stream);  // bit 3: two workgroups per CU
}

// Called by da_gemm_nt_geglu (gemm_nt_v2.hip); -1 = not eligible.  K = 320 only: 128 hidden units per workgroup (each wave
// 32 value + the 32 matching gate columns over all of K = 160 registers), inner / 128 column blocks per row walk on one XCD.
int da_gemm_nt_geglu_ws_try(const void* A, long lda, const void* W, const float* bias, void* F, long ldf, void* G, long ldg,
                            int M, int inner, int K, hipStream_t stream) {
  typedef WsCfg<10, 4, 1> Gc;
  if (!(g_nt_ws & 4) || K != Gc::K || M % WS_BM || inner % Gc::BN) return -1;
  const int nb = inner / Gc::BN, cus = da_usable_cus(256), gx = (cus / 8) * 8;
  if (nb < 1 || gx / 8 < nb || (long)(M / WS_BM) * nb < 8L * cus) return -1;
  if ((lda & 7) || (ldf & 7) || (ldg & 7) || ((size_t)A & 15) || ((size_t)F & 15) || ((size_t)G & 15) || ((size_t)W & 15)) return -1;
  if (64 * lda * 2 >= (1L << 31) || 64 * ldf * 2 >= (1L << 31) || 64 * ldg * 2 >= (1L << 31)) return -1;
  GemmWsParams p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.bias = bias; p.R = nullptr; p.C = (bf16*)F;
  p.lda = lda; p.ldr = 0; p.ldc = ldf; p.M = M; p.N = 2 * inner; p.tiles_m = M / WS_BM; p.nb = nb;
  p.G = (bf16*)G; p.ldg = ldg; p.inner = inner;
  constexpr int SMEM = WS_NS * Gc::STAGE;
  static unsigned long long attr_done = 0;
  if (da_ensure_dyn_smem((const void*)gemm_nt_ws_kernel<10, 4, false, 1>, SMEM, &attr_done) != DA_OK) return DA_ERR_LAUNCH;
  hipLaunchKernelGGL((gemm_nt_ws_kernel<10, 4, false, 1>), dim3(gx), dim3(256), SMEM, stream, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
