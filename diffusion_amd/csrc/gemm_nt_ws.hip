// gemm_nt_ws: weight-stationary linear layer for K = 320 (the level-0 transformer width); da_set_option("gemm_nt_ws", 0 | 1), on.
//     C[m][n] = sum_k A[m][k] * W[n][k]  (+ bias[n]) (+ R[m][n]),  bf16 in / out, fp32 accumulation,  N = 320, 640, 960, 1280
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
#include <type_traits>
#include "common.hpp"
#include "diffusion_amd.h"

int da_usable_cus(int cus);  // gemm_nt_v2.hip
int g_nt_ws = 1;             // da_set_option("gemm_nt_ws", 0 | 1)

namespace {

struct GemmWsParams {
  const bf16* A;
  const bf16* W;
  const float* bias;
  const bf16* R;
  bf16* C;
  long lda, ldr, ldc;
  int M, N, tiles_m, nb;
};

constexpr int WS_K = 320, WS_MT = 2, WS_BM = 16 * WS_MT, WS_BN = 320, WS_ROWB = WS_K * 2, WS_STAGE = WS_BM * WS_ROWB;
#ifndef WS_NS_BUILD
#define WS_NS_BUILD 4
#endif
constexpr int WS_NS = WS_NS_BUILD;  // LDS stages of 20 KB (ring); 4 / 6 / 8 measured equal (tools/build_alt.sh -DWS_NS_BUILD=n)
constexpr int WS_PIECES = WS_STAGE / 1024 / 4;  // LDS-DMA instructions per wave and tile (5); also the residual loads and the stores
constexpr int WS_AHEAD = WS_NS - 1;             // tiles requested ahead of the one being computed

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
    WS_W(5) WS_W(10) WS_W(15) WS_W(20) WS_W(25) WS_W(30) WS_W(35) WS_W(40) WS_W(45) WS_W(50) WS_W(55) WS_W(60)
#undef WS_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // never more than it is safe to leave
  }
}

template <bool HASR>
__global__ __launch_bounds__(256, 1) void gemm_nt_ws_kernel(GemmWsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // WS_NS stages of 32 activation rows
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  typedef float accv_t __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g4 = lane >> 4, l15 = lane & 15;
  // Workgroup -> (slot, column block).  A slot is a walk over the row tiles slot, slot + grid, ...; with N = nb x 320 the nb
  // workgroups of a slot hold different weight columns and read the SAME activation rows, so they are placed on one XCD
  // (workgroup ids equal mod 8 share an XCD and its L2): per XCD gridDim.x / 8 workgroups = spx slots x nb column blocks,
  // the remainder (2 of 32 at nb = 3) exits - the rows are then fetched from HBM once, not nb times.
  const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
  const int spx = ((int)gridDim.x >> 3) / p.nb;
  if (idx >= spx * p.nb) return;
  const int n0 = (idx % p.nb) * WS_BN;
  const int slot = xcd * spx + idx / p.nb;
  const int grid = 8 * spx;
  if (slot >= p.tiles_m) return;
  const int n_my = (p.tiles_m - slot + grid - 1) / grid;  // tiles of this workgroup: slot + k*grid

  // ---- tile requests: piece j of this wave covers LDS bytes (wave*10 + j)*1024 + lane*16 of the stage image
  // [row][640 B], physical 16-byte chunk pc of row r holding logical chunk pc ^ ((r >> 1) & 7) (conflict-free ds_read_b128)
  unsigned dsrc[WS_PIECES];
#pragma unroll
  for (int j = 0; j < WS_PIECES; ++j) {
    const int byte = (wave * WS_PIECES + j) * 1024 + lane * 16;
    const int row = byte / WS_ROWB, pc = (byte - row * WS_ROWB) >> 4;
    dsrc[j] = (unsigned)row * (unsigned)(p.lda * 2) + (unsigned)((pc ^ ((row >> 1) & 7)) << 4);
  }
  auto request = [&](int k) {  // k-th tile of this workgroup -> stage k % WS_NS
    const char* base = reinterpret_cast<const char*>(p.A) + (long)(slot + k * grid) * WS_BM * p.lda * 2;
    char* dst = smem + (k % WS_NS) * WS_STAGE + wave * (WS_PIECES * 1024);
#pragma unroll
    for (int j = 0; j < WS_PIECES; ++j) glds16_ws(base + dsrc[j], dst + j * 1024);
  };
#ifdef WS_EXP_AHOT      // timing-only build: every request re-reads the workgroup's first tile (L2-resident)
#define WS_REQ(k) request_hot(k)
  auto request_hot = [&](int k) {
    const char* base = reinterpret_cast<const char*>(p.A) + (long)slot * WS_BM * p.lda * 2;
    char* dst = smem + (k % WS_NS) * WS_STAGE + wave * (WS_PIECES * 1024);
#pragma unroll
    for (int j = 0; j < WS_PIECES; ++j) glds16_ws(base + dsrc[j], dst + j * 1024);
  };
#else
#define WS_REQ(k) request(k)
#endif
  // Vector-memory operations this wave issues AFTER the requests of its tile k (k >= WS_AHEAD; issued in step k - WS_AHEAD):
  // what the wait in front of tile k may leave in flight.  A step j issues, in this order: the residual loads of tile j - 1
  // (j >= 1, HASR), the requests of tile j + WS_AHEAD (while there is one), the stores of tile j - 1 (j >= 1); 5 each.
  // (A pure function of k on purpose: running counters captured by the step lambda ended up in scratch memory, and every
  // scratch load comes with s_waitcnt vmcnt(0).)
  auto after_requests_of = [&](int k) {
    const int j0 = k - WS_AHEAD;
    int n = j0 >= 1 ? WS_PIECES : 0;
#pragma unroll
    for (int d = 1; d < WS_AHEAD; ++d)
      n += (HASR ? WS_PIECES : 0) + (j0 + d + WS_AHEAD < n_my ? WS_PIECES : 0) + WS_PIECES;
    return n < 60 ? n : 60;  // the counter has 6 bits; leaving fewer in flight than allowed is always safe
  };

  // the first two tiles are requested BEFORE the 200 KB of weight fragments (every workgroup fetches them at once - a few
  // microseconds during which HBM would otherwise idle), then everything is waited for together
#pragma unroll
  for (int k = 0; k < WS_AHEAD; ++k)
    if (k < n_my) WS_REQ(k);
  // ---- the resident weight fragments: W rows n0 + 80*wave + 16*jt + (lane & 15), k = 32*s + 8*(lane >> 4) .. + 7
  bf16x8 wf[5][10];
  {
    const bf16* wp = p.W + (long)(n0 + 80 * wave + l15) * WS_K + 8 * g4;
#pragma unroll
    for (int jt = 0; jt < 5; ++jt)
#pragma unroll
      for (int s = 0; s < 10; ++s) wf[jt][s] = ld8(wp + (long)(16 * jt) * WS_K + 32 * s);
  }
  // bias of this lane's accumulator columns n0 + 80*wave + 16*jt + 4*(lane >> 4) + e: the start value of every K sum
  accv_t bv[5];
#pragma unroll
  for (int jt = 0; jt < 5; ++jt)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[jt][e] = p.bias ? p.bias[n0 + 80 * wave + 16 * jt + 4 * g4 + e] : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the counted waits below start from zero
  // ... and the compiler's own bookkeeping too: redefined here, the fragments are no pending loads to it (it would otherwise
  // wait vmcnt(0) for them in front of the first product - behind the tile requests in flight)
#pragma unroll
  for (int jt = 0; jt < 5; ++jt) {
#pragma unroll
    for (int s = 0; s < 10; ++s) asm volatile("" : "+v"(wf[jt][s]));
    asm volatile("" : "+v"(bv[jt]));
  }

  // ---- this lane's 16 bytes of the output tile (the direct epilogue of gemm_nt_v2.hip with wm = 0, MT = 4, wn = wave, NT = 5):
  // pieces (strip i, column pair pj = 0, 1); pj = 2: the fifth column tile of the strip pair (i - 1, i)
  const int lq = g4 & 1, lcol = 80 * wave + 8 * (lane >> 5);
  auto out_off = [&](int i, int pj, long ld) {
    const int row = (pj < 2) ? l15 + 16 * i : l15 + 16 * (i - 1 + lq);
    const int col = (pj < 2) ? lcol + 32 * pj + 16 * lq : lcol + 64;
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

  const unsigned smem_off = lds_offset(smem);
  const unsigned frag_off = (unsigned)(l15 * WS_ROWB);
  const int fsw = (lane >> 1) & 7;
  typedef accv_t acc_t[WS_MT][5];
  typedef std::integral_constant<bool, true> yes_t;
  typedef std::integral_constant<bool, false> no_t;

  // One step of the software pipeline: the products of tile k (-> cur) run beside the epilogue of tile k - 1 (<- prev): a
  // single wave per SIMD has nobody else to fill the gaps between its MFMAs, so the conversion, shuffle and store
  // instructions of the finished tile are placed between the products of the next one (second half of its K loop; the
  // residual of the finished tile is requested at the start of the step and waited for at half time).
  auto step = [&](acc_t& cur, acc_t& prev, auto do_compute, auto do_epi, const int k) {
    constexpr bool C = decltype(do_compute)::value, E = decltype(do_epi)::value;
    const int st = k % WS_NS;
    if constexpr (C) {
      if (k >= WS_AHEAD) ws_wait_vm(after_requests_of(k));  // (the first WS_AHEAD tiles landed behind the weight fragments)
      __builtin_amdgcn_s_barrier();  // everybody's pieces of tile k are in LDS; everybody is done with the stage requested next
      asm volatile("" ::: "memory");
    }
    u32x4 rin[WS_MT][2], rin5[WS_MT / 2];
    if constexpr (E && HASR) {
      const char* rb = reinterpret_cast<const char*>(p.R) + (long)(slot + (k - 1) * grid) * WS_BM * p.ldr * 2;
#pragma unroll
      for (int i = 0; i < WS_MT; ++i) {
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) rin[i][pj] = ws_load16(rb, out_off(i, pj, p.ldr));
        if (i & 1) rin5[i >> 1] = ws_load16(rb, out_off(i, 2, p.ldr));
      }
    }
    bool requested = false;
    if constexpr (C) {
      if (k + WS_AHEAD < n_my) {
        WS_REQ(k + WS_AHEAD);
        requested = true;
      }
    }
    // fragment (strip i, K-step s) = 16 bytes at row 16*i + (lane & 15), chunk (4*s + (lane >> 4)) ^ ((lane >> 1) & 7) of the
    // stage: with lo = (lane >> 4) ^ (fsw & 3) and b = fsw >> 2 that is byte 16*lo + 64*(s ^ b) of the row, i.e. an even /
    // odd base register per lane and an IMMEDIATE 64*(s & ~1) + 10240*i - no address arithmetic in the K loop
    const unsigned fb = smem_off + (unsigned)(st * WS_STAGE) + frag_off + (unsigned)(((g4 ^ (fsw & 3)) << 4));
    const unsigned fb_even = fb + (unsigned)((fsw >> 2) << 6), fb_odd = fb + (unsigned)((1 - (fsw >> 2)) << 6);
    char* cb = reinterpret_cast<char*>(p.C) + (long)(slot + (k - 1) * grid) * WS_BM * p.ldc * 2;
    auto piece = [&](int q) {  // q = 0..4: the finished tile's 16-byte pieces in strip order
      static_assert(WS_MT == 2, "piece table");
      constexpr int PI[5] = {0, 0, 1, 1, 1}, PP[5] = {0, 1, 0, 1, 2};
      const int i = PI[q], pj = PP[q];
      const u32x4 o = pj < 2 ? finish(prev[i][2 * pj], prev[i][2 * pj + 1], rin[i][pj]) : finish(prev[i - 1][4], prev[i][4], rin5[i >> 1]);
#ifdef WS_EXP_NOSTORE   // timing-only build (tools/build_alt.sh): everything but the stores
      asm volatile("" ::"v"(o));
#else
      *reinterpret_cast<u32x4*>(cb + out_off(i, pj, p.ldc)) = o;
#endif
    };
    bf16x8 a_cur[WS_MT], a_nxt[WS_MT];
    if constexpr (C) {
      a_cur[0] = ws_lds16<0>(fb_even);
      a_cur[1] = ws_lds16<16 * WS_ROWB>(fb_even);
    }
    ws_static_for<0, 10>([&](auto S) {
      constexpr int s = decltype(S)::value;
      if constexpr (s == 5 && E && HASR) {
        // the residual has landed (only this step's requests are younger).  The registers are operands of the wait: without
        // that tie nothing stops the compiler from scheduling their first use in front of it.  ONE statement for both counts
        // (a run-time branch between two tied statements makes the compiler join their register operands with copies -
        // placed in front of the wait, they would read registers the loads have not filled)
        asm volatile("s_waitcnt vmcnt(5)\n\t"
                     "s_cmp_lg_u32 %5, 0\n\t"
                     "s_cbranch_scc1 .Lws_r%=\n\t"
                     "s_waitcnt vmcnt(0)\n"
                     ".Lws_r%=:"
                     : "+v"(rin[0][0]), "+v"(rin[0][1]), "+v"(rin[1][0]), "+v"(rin[1][1]), "+v"(rin5[0])
                     : "s"((int)requested)
                     : "scc", "memory");
        static_assert(WS_PIECES == 5 && WS_MT == 2, "operands of the residual wait");
      }
      if constexpr (C) {
        if constexpr (s + 1 < 10) {
          constexpr int OFF = 64 * ((s + 1) & ~1);
          a_nxt[0] = ws_lds16<OFF>(((s + 1) & 1) ? fb_odd : fb_even);
          a_nxt[1] = ws_lds16<OFF + 16 * WS_ROWB>(((s + 1) & 1) ? fb_odd : fb_even);
          lds_wait_for<2>(a_cur[0], a_cur[1]);   // this step's fragments are here; the two just requested stay in flight
        } else {
          lds_wait_for<0>(a_cur[0], a_cur[1]);
        }
      }
#pragma unroll
      for (int i = 0; i < WS_MT; ++i) {
        if constexpr (C) {
#pragma unroll
          for (int jt = 0; jt < 5; ++jt)
            cur[i][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jt][s], a_cur[i], s == 0 ? bv[jt] : cur[i][jt], 0, 0, 0);
        }
        if constexpr (E) {
          if (s >= 5 && i == WS_MT - 1) piece(s - 5);
        }
      }
      if constexpr (C) {
#pragma unroll
        for (int i = 0; i < WS_MT; ++i) a_cur[i] = a_nxt[i];
      }
    });
  };

  acc_t accA, accB;
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

}  // namespace

// Called by da_gemm_nt (gemm_nt.hip) for plain bf16 linears; -1 = not eligible (the tiled form takes the call).
int da_gemm_nt_ws_try(const void* A, long lda, const void* W, const float* bias, const void* R, long ldr, void* C, long ldc,
                      int M, int N, int K, hipStream_t stream) {
  if (!g_nt_ws || K != WS_K || N < WS_BN || N % WS_BN || M % WS_BM) return -1;  // whole 32-row tiles only
  const int tiles_m = M / WS_BM;
  const int cus = da_usable_cus(256);
  if ((long)tiles_m * (N / WS_BN) < 8L * cus) return -1;  // each workgroup loads 200 KB of W before its first 20 KB tile
  if ((lda & 7) || (ldc & 7) || (R && (ldr & 7)) || ((size_t)A & 15) || ((size_t)C & 15) || ((size_t)W & 15) || (R && ((size_t)R & 15)))
    return -1;
  if (64 * lda * 2 >= (1L << 31) || 64 * ldc * 2 >= (1L << 31) || (R && 64 * ldr * 2 >= (1L << 31))) return -1;
  GemmWsParams p;
  p.A = (const bf16*)A; p.W = (const bf16*)W; p.bias = bias; p.R = (const bf16*)R; p.C = (bf16*)C;
  p.lda = lda; p.ldr = ldr; p.ldc = ldc; p.M = M; p.N = N; p.tiles_m = tiles_m;
  const int nb = N / WS_BN;
  if (nb > 4) return -1;
  p.nb = nb;
  const int gx = (cus / 8) * 8;  // whole rounds of the 8 XCDs (see the kernel's slot mapping)
  if (gx < 8 * nb) return -1;
  constexpr int SMEM = WS_NS * WS_STAGE;
  static unsigned long long attr_done[2] = {0, 0};
  if (R) {
    if (da_ensure_dyn_smem((const void*)gemm_nt_ws_kernel<true>, SMEM, &attr_done[1]) != DA_OK) return DA_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_nt_ws_kernel<true>), dim3(gx), dim3(256), SMEM, stream, p);
  } else {
    if (da_ensure_dyn_smem((const void*)gemm_nt_ws_kernel<false>, SMEM, &attr_done[0]) != DA_OK) return DA_ERR_LAUNCH;
    hipLaunchKernelGGL((gemm_nt_ws_kernel<false>), dim3(gx), dim3(256), SMEM, stream, p);
  }
  DA_CHECK_LAUNCH();
  return DA_OK;
}
