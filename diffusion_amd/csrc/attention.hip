// Flash-style attention for head_dim 64 on gfx950: forward, dQ and dK/dV kernels.
// No N x N score matrix is ever written to HBM.
//
// Orientation (v_mfma_f32_32x32x16_bf16, accumulator map col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)):
//   forward / dQ kernels are QUERY-major: S^T = K.Q^T puts the query on the lane, so the softmax row
//   statistics (running max, sum, LSE, delta) are per-lane scalars shared only with lane^32, and the
//   P^T / dS^T accumulators are directly the B operands of the following products (O^T = V^T.P^T,
//   dQ^T = K^T.dS^T) with no lane movement.  V^T / K^T reach the A operand through the transposed LDS
//   read (ds_read_b64_tr_b16) of a [key][d] image whose XOR-swizzled 128-B rows keep the reads conflict-free.
//   dK/dV kernel is KEY-major: S = Q.K^T and dP = dO.V^T put the key on the lane; each wave keeps
//   dK^T, dV^T of its 32 keys in accumulators while the workgroup sweeps the queries, so dK/dV need no
//   cross-workgroup sum; P and dS accumulators are again the B operands of dV^T += dO^T.P and
//   dK^T += Q^T.dS.  dQ is produced by the separate query-major kernel (recomputing S and dP) instead
//   of fp32 atomics, which keeps all three gradients bitwise reproducible.
// Every operand tile is ONE 128-B-row LDS image whose chunk swizzle serves row reads and transposed reads alike (swz_key).
// Softmax is computed in the exp2 domain: p = exp2(s*scale*log2e - L2), L2 = m + log2(sum) saved per row.
#include <type_traits>

#include "common.hpp"
#include "diffusion_amd.h"

namespace {

struct AttnParams {
  const bf16 *Q, *K, *V, *O, *dO;
  bf16 *Out, *dQ, *dK, *dV;
  float *L2, *Delta;
  long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int B, H, Nq, Nk;
  float sc;     // softmax scale * log2(e)
  float scale;  // softmax scale
};

// ONE LDS image per operand tile, [rows][64 bf16] with unpadded 128-B rows, serves both kinds of read: the 16-B chunk index of
// row r is XOR-ed with key(r) = (bit 1 of r) << 2 | (bits 2-3 of r).
//  * ds_read_b128 row fragments (16 rows x one chunk per lane group {0-3,12-15,20-27} / {4-11,16-19,28-31}): the row parity
//    picks the 128-B half of the 256-B bank span and key(r) takes all 8 values over the rows of a group, so the 16 lanes hit
//    16 distinct 16-B slots;
//  * ds_read_b64_tr_b16 (a 32-lane half touches one 64-B half of 4 consecutive rows r0..r0+3, r0 % 4 == 0): bits 2-3 of r are
//    constant over them (a permutation of the chunks inside the 64 bytes), bit 1 flips the 64-B half and bit 0 the 128-B
//    half - four distinct 64-B quarters of the bank span.
// Until round 3 the row image (key (r>>1)&7) and the transposed-read image (key bit 1 only) were separate copies of the same
// tile: twice the LDS-DMA requests (the per-step request block is wave time with the matrix pipe idle) and twice the LDS.
// Diagnostic build only (-DDA_STAMPS, tools/attn_stamps.py): waves 0 and 3 of the first workgroups of the dK/dV kernel keep
// the shader clock of five points of a query step in scalar registers and store them after the step's barrier
// (8 slots per (workgroup, wave, step)); never compiled into the shipped library.
#ifdef DA_STAMPS
__device__ unsigned long long* g_attn_stamp_buf;
__device__ int g_attn_stamp_wgs;
#define ASTAMP(v) v = __builtin_amdgcn_s_memtime()
#else
#define ASTAMP(v) do {} while (0)
#endif

constexpr int TR_LD = 128;
DEVINL int swz_key(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
DEVINL int tr_off(int row, int bytecol) { return row * TR_LD + (bytecol ^ (swz_key(row) << 4)); }

DEVINL int swz128(int row, int chunk) { return row * 128 + ((chunk ^ swz_key(row)) << 4); }

// A operand of the 32x32x16 MFMA for (rows = columns col0..col0+31 of a transposed-read image, k = 16 image rows
// starting at row0) in the accumulator-as-operand k order: element j <-> image row row0 + 8*(j>>2) + 4*(lane>>5) + (j&3).
// Issued in two steps because every loop here also fills LDS by DMA: the two transposed reads go through the asm form
// (common.hpp: against the intrinsic the compiler waits for every pending DMA), the caller waits with lds_wait_for<>
// and then joins the halves.
DEVINL void tr_frag_issue(unsigned img_off, int row0, int col0, int lane, short4v& t0, short4v& t1) {
  const int gg = lane >> 4, dgrp = gg & 1, hh = gg >> 1, qq = (lane >> 2) & 3, pp = lane & 3;
  const int row = row0 + 4 * hh + qq, bc = (col0 + 16 * dgrp + 4 * pp) * 2;
  const unsigned a = img_off + tr_off(row, bc);
  t0 = lds_tr16_b64_asm(a);
  // row + 8 flips bit 3 of the row = bit 1 of swz_key = byte bit 5 of the column (image offsets are multiples of 128 B)
  t1 = lds_tr16_b64_asm((a ^ 32u) + 8 * TR_LD);
}
DEVINL bf16x8 tr_frag_join(short4v t0, short4v t1) {
  typedef __attribute__((ext_vector_type(8))) short short8v;
  short8v v = __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

DEVINL bf16x8 pack8(const f32x16& x, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = f2bf(x[8 * s + j]);
  return r;
}

// The softmax arithmetic runs on register PAIRS: these kernels are bound by VALU issue slots (PMC: 8-17 VALU
// instructions per 32-cycle MFMA, matrix pipe ~25 % busy), and v_pk_fma/add/mul_f32 handle two fp32 per slot.
typedef float f32x2 __attribute__((ext_vector_type(2)));
DEVINL f32x2 pair(const f32x16& v, int i) { return f32x2{v[2 * i], v[2 * i + 1]}; }
DEVINL void set_pair(f32x16& v, int i, f32x2 x) { v[2 * i] = x.x; v[2 * i + 1] = x.y; }
// Scalar forms of the pair arithmetic for the FORWARD kernel: beside MFMAs a v_pk_*_f32 costs more than the two plain
// instructions it replaces (MI355X_MICROARCH.md, cycle constants), and the forward - one exp and ~4 other VALU ops per score
// against 8 MFMAs per 32x32 tile - measured +4...6 % at 256 / 1,024 tokens and +-1 % elsewhere with them
// (profiles/r03_ab_attn4.txt); the backward kernels measured +-2 % either way and keep the packed forms.  The empty asm keeps the
// compiler's SLP pass from re-pairing them.
DEVINL f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) {
  float x = fmaf(a.x, b.x, c.x), y = fmaf(a.y, b.y, c.y);
  asm("" : "+v"(x));
  asm("" : "+v"(y));
  return f32x2{x, y};
}
DEVINL f32x2 mul2(f32x2 a, f32x2 b) {
  float x = a.x * b.x, y = a.y * b.y;
  asm("" : "+v"(x));
  asm("" : "+v"(y));
  return f32x2{x, y};
}
DEVINL f32x2 add2(f32x2 a, f32x2 b) {
  float x = a.x + b.x, y = a.y + b.y;
  asm("" : "+v"(x));
  asm("" : "+v"(y));
  return f32x2{x, y};
}
DEVINL f32x2 exp2_2(f32x2 x) { return f32x2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }

DEVINL float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// XCD-aware workgroup order.  Workgroups are dispatched round-robin over the 8 XCDs (each with its own 4 MiB L2), so with
// the plain (block, head, image) grid the blocks of ONE head land on eight different XCDs and every XCD streams the
// K / V (or Q / dO) of ~24 heads at once - far more than its L2 - from the Infinity Cache / HBM (measured 6.1 TB/s
// for the 10.7 GB a 4096-token forward launch reads: the kernels were bound by that, not by MFMA, VALU or occupancy).
// Each XCD instead takes a contiguous run of (image, head, block) ids, block fastest: its ~96 resident workgroups
// then share the operands of ~3 heads (3 MiB), which its L2 serves at ~2.5x the rate.
DEVINL void xcd_block_id(int& blk, int& hd, int& b) {
  const int nb = gridDim.x, nh = gridDim.y;
  const int total = nb * nh * gridDim.z;
  const int w = blockIdx.x + nb * (blockIdx.y + nh * blockIdx.z);  // dispatch order
  const int q = total >> 3, r = total & 7;
  const int xcd = w & 7, idx = w >> 3;
  const int l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  blk = l % nb;
  const int t = l / nb;
  hd = t % nh;
  b = t / nh;
}

// LDS-DMA (global_load_lds): 64 lanes x 16 B (or 4 B) land lane-linearly at a wave-uniform LDS address; no staging
// registers, completion tracked by vmcnt.  Swizzles are therefore applied on the per-lane SOURCE address.
__device__ __attribute__((aligned(256))) unsigned char g_attn_zero[256];
DEVINL void dma16(const void* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
DEVINL void dma4(const void* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

DEVINL int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ------------------------------------------------------------------------------------------------
// forward.  Workgroup = 128 queries (4 waves x 32), 64-key tiles, LDS double-buffered (one barrier per tile),
// next tile requested by LDS-DMA while the current one is consumed.  Softmax per 64-key step:
// p = exp2(fma(s, sc, -m*sc)) on RAW scores (sc > 0 keeps the max order), rescale of O only when some lane's
// running max moved (wave-uniform branch), key masking only on the ragged tail tile.
// ------------------------------------------------------------------------------------------------
constexpr int FW_STAGE = 64 * 128 + 64 * TR_LD;

// CAUSAL (the text encoder's self-attention, Nq == Nk): key j contributes to query q only for j <= q - every tile is
// masked per lane like the ragged tail, and tiles wholly above a workgroup's last query are skipped.
template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 * FW_STAGE, dynamic: see attn_bwd_dkv_kernel
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  int blk, hd, b;
  xcd_block_id(blk, hd, b);
  const int q = blk * 128 + wave * 32 + r;
  const bool qv = q < p.Nq;

  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    qf[ks] = qv ? ld8(p.Q + ((long)b * p.Nq + q) * p.ldq + hd * 64 + ks * 16 + h * 8) : zero8();

  f32x16 o0, o1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
  float m = -INFINITY, l = 0.f;

  // causal: keys past the workgroup's last query are never needed
  const int nk_eff = CAUSAL ? min(p.Nk, blk * 128 + 128) : p.Nk;
  const int nt = (nk_eff + 63) / 64;
  // K / V tiles (64 keys) reach LDS by DMA, one tile ahead into the stage the previous step released: K as the row image,
  // V as the transposed-read image.  Wave w fills rows 8w..8w+7 and 32+8w..32+8w+7 of both.
  const int drow = wave * 8 + (lane >> 3), pc = lane & 7;
  const int lc = (pc ^ swz_key(drow)) * 8;  // source element offset behind physical chunk pc (rows drow, drow + 32: same key)
  const bf16* kp = p.K + ((long)b * p.Nk + drow) * p.ldk + hd * 64;
  const bf16* vp = p.V + ((long)b * p.Nk + drow) * p.ldv + hd * 64;
  const char* zero = reinterpret_cast<const char*>(g_attn_zero);
  auto dma = [&](int t, int st) {  // tiles are requested in order
    char* S = smem + st * FW_STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = t * 64 + drow + 32 * i < p.Nk;
      dma16(ok ? (const void*)(kp + 32 * i * p.ldk + lc) : (const void*)zero, S + i * 4096);
      dma16(ok ? (const void*)(vp + 32 * i * p.ldv + lc) : (const void*)zero, S + 8192 + i * 4096);
    }
    kp += 64 * p.ldk;
    vp += 64 * p.ldv;
  };
  auto sync_tile = [&]() {  // the requested tile has landed; everyone is done reading the current one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  dma(0, 0);
  sync_tile();
  // One 64-key step.  TAIL (the ragged last tile) is a compile-time split: written as a run-time `if`, the key masking
  // was if-converted into every step (32 v_cmp + 32 v_cndmask + index adds, a third of the loop's VALU work).
  auto step = [&](int t, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    if (t + 1 < nt) dma(t + 1, (t + 1) & 1);
    const char* Ks = smem + (t & 1) * FW_STAGE;
    const char* Vs = Ks + 64 * 128;
    f32x16 s0, s1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 k0 = *reinterpret_cast<const bf16x8*>(Ks + swz128(r, 2 * ks + h));
      bf16x8 k1 = *reinterpret_cast<const bf16x8*>(Ks + swz128(32 + r, 2 * ks + h));
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ks], s1, 0, 0, 0);
    }
    // V^T fragments for the P.V product: requested now (asm form), consumed after the softmax arithmetic
    short4v tv[2][4][2];
    const unsigned vto = lds_offset(Vs);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      tr_frag_issue(vto, 16 * s2, 0, lane, tv[s2][0][0], tv[s2][0][1]);
      tr_frag_issue(vto, 16 * s2, 32, lane, tv[s2][1][0], tv[s2][1][1]);
      tr_frag_issue(vto, 32 + 16 * s2, 0, lane, tv[s2][2][0], tv[s2][2][1]);
      tr_frag_issue(vto, 32 + 16 * s2, 32, lane, tv[s2][3][0], tv[s2][3][1]);
    }
    if constexpr (TAIL) {
      const int klim = CAUSAL ? min(p.Nk, q + 1) : p.Nk;  // first key this lane's query does not see
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int key = t * 64 + acc_row(i, lane);
        if (key >= klim) s0[i] = -INFINITY;
        if (key + 32 >= klim) s1[i] = -INFINITY;
      }
    }
    float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = max3(mx, s0[i], s1[i]);
    mx = fmaxf(mx, xor32(mx));
    if (__any(mx > m)) {
      const float mn = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f((m - mn) * p.sc);
      l *= alpha;
      const f32x2 al2 = {alpha, alpha};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        set_pair(o0, i, mul2(pair(o0, i), al2));
        set_pair(o1, i, mul2(pair(o1, i), al2));
      }
      m = mn;
    }
    const float msc = -m * p.sc;
    const f32x2 sc2 = {p.sc, p.sc}, msc2 = {msc, msc};
    f32x2 ls2 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x2 e0 = exp2_2(fma2(pair(s0, i), sc2, msc2));
      const f32x2 e1 = exp2_2(fma2(pair(s1, i), sc2, msc2));
      set_pair(s0, i, e0);
      set_pair(s1, i, e1);
      ls2 = add2(ls2, add2(e0, e1));
    }
    l += ls2.x + ls2.y;
    lds_wait_for<0>(tv[0][0][0], tv[0][0][1], tv[0][1][0], tv[0][1][1], tv[0][2][0], tv[0][2][1], tv[0][3][0], tv[0][3][1],
                    tv[1][0][0], tv[1][0][1], tv[1][1][0], tv[1][1][1], tv[1][2][0], tv[1][2][1], tv[1][3][0], tv[1][3][1]);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 pa = pack8(s0, s2), pb = pack8(s1, s2);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(tv[s2][0][0], tv[s2][0][1]), pa, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(tv[s2][1][0], tv[s2][1][1]), pa, o1, 0, 0, 0);
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(tv[s2][2][0], tv[s2][2][1]), pb, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(tv[s2][3][0], tv[s2][3][1]), pb, o1, 0, 0, 0);
    }
    sync_tile();
  };
  if constexpr (CAUSAL) {  // every tile takes the masking step
    for (int t = 0; t < nt; ++t) step(t, std::true_type{});
  } else {
    const int nfull = p.Nk / 64;
    for (int t = 0; t < nfull; ++t) step(t, std::false_type{});
    if (nfull < nt) step(nfull, std::true_type{});
  }
  const float lt = l + xor32(l);
  const float inv = 1.0f / lt;
  if (qv) {
    bf16* op = p.Out + ((long)b * p.Nq + q) * p.ldo + hd * 64;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      bf16x4 a, c;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = f2bf(o0[rg * 4 + e] * inv);
        c[e] = f2bf(o1[rg * 4 + e] * inv);
      }
      *reinterpret_cast<bf16x4*>(op + 8 * rg + 4 * h) = a;
      *reinterpret_cast<bf16x4*>(op + 32 + 8 * rg + 4 * h) = c;
    }
    if (h == 0) p.L2[((long)b * p.H + hd) * p.Nq + q] = m * p.sc + log2f(lt);
  }
}

// ------------------------------------------------------------------------------------------------
// backward, query-major: dQ (and delta = rowsum(dO*O), stored for the dK/dV kernel).  Same pipeline as forward.
// ------------------------------------------------------------------------------------------------
constexpr int DQ_STAGE = 2 * 64 * 128;  // K and V images (K is read by rows for S and transposed for dQ)

__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 * DQ_STAGE, dynamic: see attn_bwd_dkv_kernel
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  int blk, hd, b;
  xcd_block_id(blk, hd, b);
  const int q = blk * 128 + wave * 32 + r;
  const bool qv = q < p.Nq;

  bf16x8 qf[4], dof[4];
  float delta = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int c = hd * 64 + ks * 16 + h * 8;
    qf[ks] = qv ? ld8(p.Q + ((long)b * p.Nq + q) * p.ldq + c) : zero8();
    dof[ks] = qv ? ld8(p.dO + ((long)b * p.Nq + q) * p.lddo + c) : zero8();
    bf16x8 of = qv ? ld8(p.O + ((long)b * p.Nq + q) * p.ldo + c) : zero8();
#pragma unroll
    for (int j = 0; j < 8; ++j) delta += bf2f(dof[ks][j]) * bf2f(of[j]);
  }
  delta += xor32(delta);
  const long statidx = ((long)b * p.H + hd) * p.Nq + q;
  const float nL2q = qv ? -p.L2[statidx] : 0.f;
  if (qv && h == 0) p.Delta[statidx] = delta;

  f32x16 d0, d1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { d0[i] = 0.f; d1[i] = 0.f; }

  const int nt = (p.Nk + 63) / 64;
  // K / V tiles (64 keys) reach LDS by DMA, one tile ahead into the stage the previous step released: no staging
  // registers, no ds_write pass.  Wave w fills rows 8w..8w+7 and 32+8w..32+8w+7 of the two images.
  const int drow = wave * 8 + (lane >> 3), pc = lane & 7;
  const int lc = (pc ^ swz_key(drow)) * 8;
  const bf16* kp = p.K + ((long)b * p.Nk + drow) * p.ldk + hd * 64;
  const bf16* vp = p.V + ((long)b * p.Nk + drow) * p.ldv + hd * 64;
  const char* zero = reinterpret_cast<const char*>(g_attn_zero);
  auto dma = [&](int t, int st) {  // tiles are requested in order
    char* S = smem + st * DQ_STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = t * 64 + drow + 32 * i < p.Nk;
      dma16(ok ? (const void*)(kp + 32 * i * p.ldk + lc) : (const void*)zero, S + i * 4096);
      dma16(ok ? (const void*)(vp + 32 * i * p.ldv + lc) : (const void*)zero, S + 8192 + i * 4096);
    }
    kp += 64 * p.ldk;
    vp += 64 * p.ldv;
  };
  auto sync_tile = [&]() {  // the requested tile has landed; everyone is done reading the current one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  dma(0, 0);
  sync_tile();
  auto step = [&](int t, auto tail_tag) {  // TAIL: compile-time split, see attn_fwd_kernel
    constexpr bool TAIL = decltype(tail_tag)::value;
    if (t + 1 < nt) dma(t + 1, (t + 1) & 1);
    const char* Ks = smem + (t & 1) * DQ_STAGE;
    const char* Vs = Ks + 64 * 128;
    const char* Kt = Ks;  // the same image, read transposed
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int kb = half * 32;
      if (TAIL && t * 64 + kb >= p.Nk) break;
      // the row constant -delta is the START value of the dP accumulators: dP - delta leaves the MFMA chain ready
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = -delta; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + swz128(kb + r, 2 * ks + h));
        bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + swz128(kb + r, 2 * ks + h));
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp, 0, 0, 0);
      }
      // K^T fragments for the dQ product: requested now (asm form), consumed after the arithmetic below
      short4v tk0[2][2], tk1[2][2];
      const unsigned kto = lds_offset(Kt);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        tr_frag_issue(kto, kb + 16 * s2, 0, lane, tk0[s2][0], tk0[s2][1]);
        tr_frag_issue(kto, kb + 16 * s2, 32, lane, tk1[s2][0], tk1[s2][1]);
      }
      {
        const f32x2 sc2 = {p.sc, p.sc}, nl2 = {nL2q, nL2q};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x2 pv = exp2_2(__builtin_elementwise_fma(pair(s, i), sc2, nl2));
          set_pair(s, i, pv * pair(dp, i));
        }
      }
      if constexpr (TAIL) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (t * 64 + kb + acc_row(i, lane) >= p.Nk) s[i] = 0.f;
      }
      bf16x8 dsf[2] = {pack8(s, 0), pack8(s, 1)};
      lds_wait_for<0>(tk0[0][0], tk0[0][1], tk1[0][0], tk1[0][1], tk0[1][0], tk0[1][1], tk1[1][0], tk1[1][1]);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(tk0[s2][0], tk0[s2][1]), dsf[s2], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(tk1[s2][0], tk1[s2][1]), dsf[s2], d1, 0, 0, 0);
      }
    }
    sync_tile();
  };
  const int nfull = p.Nk / 64;
  for (int t = 0; t < nfull; ++t) step(t, std::false_type{});
  if (nfull < nt) step(nfull, std::true_type{});
  if (qv) {
    bf16* op = p.dQ + ((long)b * p.Nq + q) * p.lddq + hd * 64;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      bf16x4 a, c;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = f2bf(d0[rg * 4 + e] * p.scale);
        c[e] = f2bf(d1[rg * 4 + e] * p.scale);
      }
      *reinterpret_cast<bf16x4*>(op + 8 * rg + 4 * h) = a;
      *reinterpret_cast<bf16x4*>(op + 32 + 8 * rg + 4 * h) = c;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, key-major: dK, dV.  Workgroup = 128 keys (4 waves x 32), sweeps all queries in 32-row steps, QT rows per tile
// (three LDS stages filled by DMA two tiles ahead, one barrier per tile).
// ------------------------------------------------------------------------------------------------
// QT = query rows per tile: one LDS-DMA request block and one barrier per tile, QT / 32 sub-steps of 32 queries inside it
// (64 from 128 queries up: +3-4.5 % on the 256- to 9,216-token shapes with identical bits - half the barriers, and the
// dV / dK products of one sub-step run beside the loads and S / dP products of the next; 32 below: the 64-token level
// lost 3 %).  Stage = Q, dO images (read by rows and transposed) + L2, delta.
constexpr int kv_stage(int QT) { return 2 * QT * 128 + 2 * QT * 4; }

// (An 8-wave form - 256 keys per workgroup, one workgroup per CU, half the tile requests per key - measured 0...-10 % against
// two 4-wave workgroups per CU, profiles/r03_ab_attn2.txt: the two unsynchronised workgroups overlap better than one wide one.)
template <int KV_QT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnParams p) {
  constexpr int NW = 4, KV_SUB = KV_QT / 32, KV_STAGE = kv_stage(KV_QT);
  static_assert(KV_QT == 32 || KV_QT == 64, "query rows per tile");
  // dynamic LDS on purpose: against a static __shared__ array the compiler treats every LDS-DMA as a possibly
  // aliasing pending LDS write and puts s_waitcnt vmcnt(0) in front of the next ds_read, exposing the whole DMA latency
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 3 * KV_STAGE
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  int blk, hd, b;
  xcd_block_id(blk, hd, b);
  const int key = blk * (32 * NW) + wave * 32 + r;
  const bool kv = key < p.Nk;

  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int c = hd * 64 + ks * 16 + h * 8;
    kf[ks] = kv ? ld8(p.K + ((long)b * p.Nk + key) * p.ldk + c) : zero8();
    vf[ks] = kv ? ld8(p.V + ((long)b * p.Nk + key) * p.ldv + c) : zero8();
  }
  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { dk0[i] = 0.f; dk1[i] = 0.f; dv0[i] = 0.f; dv1[i] = 0.f; }

  // Q / dO tiles (QT queries) reach LDS by DMA, three stages, issued TWO tiles ahead: a 32-query step lasts about one
  // global-load latency, and with the register-staged prefetch consumed at the end of the same step the kernel spent
  // 55 % of its wave-cycles waiting (skipping the loads made it 31 % faster).  Wave w fills rows 8w..8w+7 of each 32-row
  // block of the two images (Q, dO); wave 0 also fetches the block's 32 L2 and 32 delta values.
  const int nt = (p.Nq + KV_QT - 1) / KV_QT;
  const int drow = (wave & 3) * 8 + (lane >> 3), pc = lane & 7;
  const int lc = (pc ^ swz_key(drow)) * 8;                 // source element offset behind physical chunk pc
  const bf16* qp = p.Q + ((long)b * p.Nq + drow) * p.ldq + hd * 64;
  const bf16* dop = p.dO + ((long)b * p.Nq + drow) * p.lddo + hd * 64;
  const float* statp = (lane < 32 ? p.L2 : p.Delta) + ((long)b * p.H + hd) * p.Nq + (lane & 31);
  const char* zero = reinterpret_cast<const char*>(g_attn_zero);
  auto dma = [&](int t, int st) {  // tiles are requested in order: the pointers advance by one tile per call
    char* S = smem + st * KV_STAGE + (wave & 3) * 1024;
#pragma unroll
    for (int sub = 0; sub < KV_SUB; ++sub) {
      const bool ok = t * KV_QT + sub * 32 + drow < p.Nq;
      dma16(ok ? (const void*)(qp + sub * 32 * p.ldq + lc) : (const void*)zero, S + sub * 4096);
      dma16(ok ? (const void*)(dop + sub * 32 * p.lddo + lc) : (const void*)zero, S + KV_QT * 128 + sub * 4096);
      if (wave == 0) {
        const bool ok2 = t * KV_QT + sub * 32 + (lane & 31) < p.Nq;
        dma4(ok2 ? (const void*)(statp + sub * 32) : (const void*)zero, smem + st * KV_STAGE + 2 * KV_QT * 128 + sub * 256);
      }
    }
    qp += KV_QT * p.ldq;
    dop += KV_QT * p.lddo;
    statp += KV_QT;
  };
  // wait until at most the DMAs of the newest requested tile are outstanding (2 per sub-step and wave, one more on
  // wave 0), then barrier
  auto sync_tiles = [&](bool newest_in_flight) {
    if (!newest_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * KV_SUB) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * KV_SUB) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // K / V fragments above: keep the counted waits below exact
  dma(0, 0);
  if (nt > 1) dma(1, 1);
  sync_tiles(nt > 1);
  for (int t = 0; t < nt; ++t) {
#ifdef DA_STAMPS
    unsigned long long st0, st1, st2, st3, st4;
#endif
    ASTAMP(st0);
    if (t + 2 < nt) dma(t + 2, (t + 2) % 3);  // its stage was last read in step t-1, released by that step's barrier
#pragma unroll
    for (int sub = 0; sub < KV_SUB; ++sub) {
    const char* Qs = smem + (t % 3) * KV_STAGE + sub * 4096;
    const char* Os = Qs + KV_QT * 128;
    const char* Qt = Qs;  // the same images, read transposed
    const char* Ot = Os;
    const float* Ls = reinterpret_cast<const float*>(smem + (t % 3) * KV_STAGE + 2 * KV_QT * 128 + sub * 256);
    const float* Ds = Ls + 32;
    // the row constants -delta[q] are the START values of the dP accumulators (row q of register i: acc_row)
    f32x16 s, dp;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ds + 8 * rg + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[4 * rg + e] = 0.f; dp[4 * rg + e] = -d4[e]; }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 qf = *reinterpret_cast<const bf16x8*>(Qs + swz128(r, 2 * ks + h));
      bf16x8 of = *reinterpret_cast<const bf16x8*>(Os + swz128(r, 2 * ks + h));
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[ks], s, 0, 0, 0);     // S[q][key]
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of, vf[ks], dp, 0, 0, 0);   // dP[q][key]
    }
    __builtin_amdgcn_sched_barrier(0);
    ASTAMP(st1);  // S / dP products issued
    // the transposed dO / Q fragments do not depend on the softmax arithmetic below: request them now (asm form, so the
    // compiler does not guard them with a wait for the tile DMA issued at the top of the step) and wait after it
    short4v to0[2][2], to1[2][2], tq0[2][2], tq1[2][2];
    const unsigned oto = lds_offset(Ot), qto = lds_offset(Qt);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      tr_frag_issue(oto, 16 * s2, 0, lane, to0[s2][0], to0[s2][1]);
      tr_frag_issue(oto, 16 * s2, 32, lane, to1[s2][0], to1[s2][1]);
      tr_frag_issue(qto, 16 * s2, 0, lane, tq0[s2][0], tq0[s2][1]);
      tr_frag_issue(qto, 16 * s2, 32, lane, tq1[s2][0], tq1[s2][1]);
    }
    f32x16 pr;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + 8 * rg + 4 * h);
      const f32x2 sc2 = {p.sc, p.sc};
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const int i = rg * 2 + e2;  // pair index: accumulator registers 2i, 2i+1 = rows 2*e2, 2*e2+1 of this group
        const f32x2 pv = exp2_2(__builtin_elementwise_fma(pair(s, i), sc2, -f32x2{l4[2 * e2], l4[2 * e2 + 1]}));
        set_pair(pr, i, pv);
        set_pair(s, i, pv * pair(dp, i));
      }
    }
    // rows beyond Nq in the last tile: Q = dO = 0, L2 = delta = 0 -> p = 1, dP - delta = 0, dS = 0, and dO^T.P adds 0.
    bf16x8 pf[2] = {pack8(pr, 0), pack8(pr, 1)};
    bf16x8 dsf[2] = {pack8(s, 0), pack8(s, 1)};
    ASTAMP(st2);  // softmax arithmetic issued
    lds_wait_for<0>(to0[0][0], to0[0][1], to1[0][0], to1[0][1], tq0[0][0], tq0[0][1], tq1[0][0], tq1[0][1], to0[1][0],
                    to0[1][1], to1[1][0], to1[1][1], tq0[1][0], tq0[1][1], tq1[1][0], tq1[1][1]);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(to0[s2][0], to0[s2][1]), pf[s2], dv0, 0, 0, 0);
      dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(to1[s2][0], to1[s2][1]), pf[s2], dv1, 0, 0, 0);
      dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(tq0[s2][0], tq0[s2][1]), dsf[s2], dk0, 0, 0, 0);
      dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(tq1[s2][0], tq1[s2][1]), dsf[s2], dk1, 0, 0, 0);
    }
    }
    ASTAMP(st3);  // dV / dK products issued
    sync_tiles(t + 2 < nt);  // tile t+1 has landed; everyone is done reading stage t % 3
    ASTAMP(st4);
#ifdef DA_STAMPS
    {
      const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      if (g_attn_stamp_buf && wg < g_attn_stamp_wgs && (wave == 0 || wave == 3) && lane == 0 && t < 32) {
        unsigned long long* o = g_attn_stamp_buf + (((long)wg * 2 + (wave == 3)) * 32 + t) * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = st4;
      }
    }
#endif
  }
  if (kv) {
    bf16* kp = p.dK + ((long)b * p.Nk + key) * p.lddk + hd * 64;
    bf16* vp = p.dV + ((long)b * p.Nk + key) * p.lddv + hd * 64;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      bf16x4 a, c, e0, e1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = f2bf(dk0[rg * 4 + e] * p.scale);
        c[e] = f2bf(dk1[rg * 4 + e] * p.scale);
        e0[e] = f2bf(dv0[rg * 4 + e]);
        e1[e] = f2bf(dv1[rg * 4 + e]);
      }
      *reinterpret_cast<bf16x4*>(kp + 8 * rg + 4 * h) = a;
      *reinterpret_cast<bf16x4*>(kp + 32 + 8 * rg + 4 * h) = c;
      *reinterpret_cast<bf16x4*>(vp + 8 * rg + 4 * h) = e0;
      *reinterpret_cast<bf16x4*>(vp + 32 + 8 * rg + 4 * h) = e1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward for SHORT key sequences (cross-attention: Nk = 77 text tokens <= 128), ONE kernel for dQ, dK and dV.
// The two-kernel backward reads Q and dO twice (once per kernel) and O once for 77 keys' worth of arithmetic: 0.84 GB of
// reads + 0.17 GB of dQ per level-0 layer at batch 256, 164 TFLOP/s.  Here the key-major sweep of attn_bwd_dkv_kernel
// (all keys of an (image, head) live in ONE workgroup: wave w holds keys 32w..32w+31 and their dK^T / dV^T accumulators)
// also produces dQ: every wave leaves its dS tile - bf16, [key][query] - in LDS before the tile's barrier, and after it
// each wave multiplies ALL keys' dS by a 16-column slice of K (K^T fragments kept in registers for the whole sweep):
// dQ^T[d slice][32 queries] = K^T . dS^T, three v_mfma_f32_16x16x32_bf16 per 16 queries, complete after one tile (no
// accumulation over key blocks, no atomics: bitwise reproducible), stored as 8 bytes per lane.  delta = rowsum(dO * O)
// is computed from the O tile (a third LDS image per stage) by the lanes that hold the dO row fragments anyway and
// reaches the accumulator rows through a wave-private 128-byte LDS scratch.  Q, dO and O are read once.
// 32-query tiles, three stages, requests two tiles ahead, one barrier per tile.
// ------------------------------------------------------------------------------------------------
constexpr int FU_STAGE = 3 * 32 * 128 + 256;     // Q, dO, O images + the 32 L2 values (one 4-byte LDS-DMA of wave 0: 64 lanes x 4 B)
constexpr int fu_ds(int NW) { return 32 * NW * 64; }   // dS^T image: [key 32*NW][query 32] bf16, 64-byte rows
constexpr int fu_smem(int NW) { return 3 * FU_STAGE + 2 * fu_ds(NW) + NW * 128; }

// NW = 4: up to 128 keys, two workgroups per CU (cross-attention, the 64-token level); NW = 8: up to 256 keys, one
// 8-wave workgroup per CU (the 256-token level): waves 0-3 request the tiles, every wave holds 32 keys, the dQ product is
// dealt as (16-column slice = wave & 3) x (16-query half = wave >> 2).
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_fused_kernel(AttnParams p) {
  constexpr int FU_DS = fu_ds(NW);
  constexpr int QTW = 8 / NW;  // 16-query tiles of the dQ product per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const dsbuf = smem + 3 * FU_STAGE;         // two dS^T images (tile parity); first holds the K image during set-up
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const dscr = reinterpret_cast<float*>(smem + 3 * FU_STAGE + 2 * FU_DS + wave * 128);  // this wave's delta row
  const int h = lane >> 5, r = lane & 31;
  int blk, hd, b;
  xcd_block_id(blk, hd, b);  // grid (1, H, B): blk == 0
  const int key = wave * 32 + r;
  const bool kv = key < p.Nk;
  const int nkb = (p.Nk + 31) / 32;  // 32-key blocks that hold real keys (<= NW)

  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int c = hd * 64 + ks * 16 + h * 8;
    kf[ks] = kv ? ld8(p.K + ((long)b * p.Nk + key) * p.ldk + c) : zero8();
    vf[ks] = kv ? ld8(p.V + ((long)b * p.Nk + key) * p.ldv + c) : zero8();
  }
  // ---- K^T fragments of this wave's 16-column slice (d = 16*wave .. +15) for the dQ product: the K rows go through a plain
  // [key][64] image once (rows of keys past Nk are zero, so whatever their dS holds adds nothing), read transposed
  {
    const int krow = tid >> 1, half = tid & 1;  // 32*NW rows x two 64-byte halves
    const bf16* src = p.K + ((long)b * p.Nk + krow) * p.ldk + hd * 64 + half * 32;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bf16x8 v = krow < p.Nk ? ld8(src + c * 8) : zero8();
      *reinterpret_cast<bf16x8*>(dsbuf + krow * 128 + half * 64 + c * 16) = v;
    }
  }
  __syncthreads();
  // A operand of v_mfma_f32_16x16x32_bf16: lane l holds A[row l&15][k = 8*(l>>4) + j] = K[key kb+8*(l>>4)+j][d0 + (l&15)]:
  // per 16-lane group two transposed 4-row x 16-column blocks (rows kb + 8g .. +3 and +4 .. +7, columns d0 .. d0+15)
  bf16x8 ktf[NW];
  const int dsl = wave & 3;  // this wave's 16-column slice of dQ
  {
    const int g = lane >> 4, i = lane & 15;
    const unsigned base = lds_offset(dsbuf) + (unsigned)((8 * g + (i >> 2)) * 128 + (16 * dsl + 4 * (i & 3)) * 2);
#pragma unroll
    for (int kb = 0; kb < NW; ++kb) {
      short4v t0 = lds_tr16_b64_asm(base + kb * 32 * 128);
      short4v t1 = lds_tr16_b64_asm(base + kb * 32 * 128 + 4 * 128);
      lds_wait_for<0>(t0, t1);
      ktf[kb] = tr_frag_join(t0, t1);
    }
  }
  __syncthreads();  // the K image is dead: its memory is the dS^T buffers from here on

  f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { dk0[i] = 0.f; dk1[i] = 0.f; dv0[i] = 0.f; dv1[i] = 0.f; }

  const int nt = (p.Nq + 31) / 32;
  const bool loader = wave < 4;  // waves 0-3 fill rows 8w .. 8w+7 of the three images
  const int drow = (wave & 3) * 8 + (lane >> 3), pc = lane & 7;
  const int lc = (pc ^ swz_key(drow)) * 8;
  const bf16* qp = p.Q + ((long)b * p.Nq + drow) * p.ldq + hd * 64;
  const bf16* dop = p.dO + ((long)b * p.Nq + drow) * p.lddo + hd * 64;
  const bf16* op = p.O + ((long)b * p.Nq + drow) * p.ldo + hd * 64;
  const long stat0 = ((long)b * p.H + hd) * p.Nq;
  const float* statp = p.L2 + stat0 + (lane & 31);
  const char* zero = reinterpret_cast<const char*>(g_attn_zero);
  auto dma = [&](int t, int st) {  // tiles are requested in order
    if (!loader) return;
    char* S = smem + st * FU_STAGE + wave * 1024;
    const bool ok = t * 32 + drow < p.Nq;
    dma16(ok ? (const void*)(qp + lc) : (const void*)zero, S);
    dma16(ok ? (const void*)(dop + lc) : (const void*)zero, S + 4096);
    dma16(ok ? (const void*)(op + lc) : (const void*)zero, S + 8192);
    if (wave == 0) {
      const bool ok2 = t * 32 + (lane & 31) < p.Nq;
      dma4(ok2 ? (const void*)statp : (const void*)zero, smem + st * FU_STAGE + 3 * 4096);
    }
    qp += 32 * p.ldq;
    dop += 32 * p.lddo;
    op += 32 * p.ldo;
    statp += 32;
  };
  auto sync_tiles = [&](bool newest_in_flight) {
    if (loader) {  // (the other waves have no tile requests of their own to wait for)
      if (!newest_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (wave == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // dQ rows of tile t from the dS^T image the tile left behind (all waves' keys), this wave's 16 columns
  auto dq_tile = [&](int t) {
    const int g = lane >> 4, i = lane & 15;
    const unsigned base = lds_offset(dsbuf + (t & 1) * FU_DS) + (unsigned)((8 * g + (i >> 2)) * 64 + (4 * (i & 3)) * 2);
    typedef float f32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int qi = 0; qi < QTW; ++qi) {
      const int qt = (wave >> 2) * QTW + qi;  // 16-query tile of the 32-query step
      f32x4v acc = {0.f, 0.f, 0.f, 0.f};
      short4v t0[NW], t1[NW];
#pragma unroll
      for (int kb = 0; kb < NW; ++kb) {
        t0[kb] = lds_tr16_b64_asm(base + kb * 32 * 64 + qt * 32);
        t1[kb] = lds_tr16_b64_asm(base + kb * 32 * 64 + qt * 32 + 4 * 64);
      }
#pragma unroll
      for (int kb = 0; kb < NW; ++kb) {
        lds_wait_for<0>(t0[kb], t1[kb]);
        if (kb < nkb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[kb], tr_frag_join(t0[kb], t1[kb]), acc, 0, 0, 0);
      }
      // D[d = 4*(l>>4) + e][q = l&15]: four consecutive columns of one dQ row per lane
      const int q = t * 32 + qt * 16 + i;
      if (q < p.Nq) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[e] * p.scale);
        *reinterpret_cast<bf16x4*>(p.dQ + ((long)b * p.Nq + q) * p.lddq + hd * 64 + 16 * dsl + 4 * g) = o;
      }
    }
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // K / V fragments above: keep the counted waits below exact
  dma(0, 0);
  if (nt > 1) dma(1, 1);
  sync_tiles(nt > 1);
  float delta_prev = 0.f;
  for (int t = 0; t < nt; ++t) {
    // the stores of tile t-1 (dQ rows, delta) go out BEFORE the requests of tile t+2: the step-end wait leaves exactly the
    // newest tile's requests in flight (vmcnt counts stores too, in order), so anything issued after them would have to be
    // counted as well - and a store whose lanes are all masked off might not be issued at all
    if (t > 0) {
      dq_tile(t - 1);  // (its dS^T image was completed by the barrier that ended step t-1)
      if (wave == 0 && h == 0 && (t - 1) * 32 + r < p.Nq) p.Delta[stat0 + (t - 1) * 32 + r] = delta_prev;
    }
    if (t + 2 < nt) dma(t + 2, (t + 2) % 3);
    const char* Qs = smem + (t % 3) * FU_STAGE;
    const char* Os = Qs + 4096;
    const char* Oo = Qs + 8192;
    const float* Ls = reinterpret_cast<const float*>(Qs + 3 * 4096);
    // dO and O row fragments of query r: this lane's 32 of the row's 64 columns -> half of delta[r]; the other half is in lane ^ 32
    bf16x8 of[4];
    float dpart = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      of[ks] = *reinterpret_cast<const bf16x8*>(Os + swz128(r, 2 * ks + h));
      const bf16x8 oo = *reinterpret_cast<const bf16x8*>(Oo + swz128(r, 2 * ks + h));
#pragma unroll
      for (int j = 0; j < 8; ++j) dpart = fmaf(bf2f(of[ks][j]), bf2f(oo[j]), dpart);
    }
    const float delta = dpart + xor32(dpart);
    if (h == 0) dscr[r] = delta;
    delta_prev = delta;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (same wave: the scratch row is written before it is read back)
    f32x16 s, dp;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(dscr + 8 * rg + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[4 * rg + e] = 0.f; dp[4 * rg + e] = -d4[e]; }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 qf = *reinterpret_cast<const bf16x8*>(Qs + swz128(r, 2 * ks + h));
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[ks], s, 0, 0, 0);       // S[q][key]
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of[ks], vf[ks], dp, 0, 0, 0);  // dP[q][key] - delta[q]
    }
    __builtin_amdgcn_sched_barrier(0);
    short4v to0[2][2], to1[2][2], tq0[2][2], tq1[2][2];
    const unsigned oto = lds_offset(Os), qto = lds_offset(Qs);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      tr_frag_issue(oto, 16 * s2, 0, lane, to0[s2][0], to0[s2][1]);
      tr_frag_issue(oto, 16 * s2, 32, lane, to1[s2][0], to1[s2][1]);
      tr_frag_issue(qto, 16 * s2, 0, lane, tq0[s2][0], tq0[s2][1]);
      tr_frag_issue(qto, 16 * s2, 32, lane, tq1[s2][0], tq1[s2][1]);
    }
    f32x16 pr;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + 8 * rg + 4 * h);
      const f32x2 sc2 = {p.sc, p.sc};
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const int i = rg * 2 + e2;
        const f32x2 pv = exp2_2(__builtin_elementwise_fma(pair(s, i), sc2, -f32x2{l4[2 * e2], l4[2 * e2 + 1]}));
        set_pair(pr, i, pv);
        set_pair(s, i, pv * pair(dp, i));
      }
    }
    bf16x8 pf[2] = {pack8(pr, 0), pack8(pr, 1)};
    bf16x8 dsf[2] = {pack8(s, 0), pack8(s, 1)};
    // dS^T -> LDS for the dQ product after the barrier: registers 4*rg .. 4*rg+3 are queries 8*rg + 4*h .. +3 of this lane's key
    {
      char* drow_p = dsbuf + (t & 1) * FU_DS + key * 64 + 8 * h;
      typedef __attribute__((ext_vector_type(4))) short short4w;
      typedef __attribute__((ext_vector_type(8))) short short8w;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const short8w v = __builtin_bit_cast(short8w, dsf[s2]);
        *reinterpret_cast<short4w*>(drow_p + 32 * s2) = __builtin_shufflevector(v, v, 0, 1, 2, 3);
        *reinterpret_cast<short4w*>(drow_p + 32 * s2 + 16) = __builtin_shufflevector(v, v, 4, 5, 6, 7);
      }
    }
    lds_wait_for<0>(to0[0][0], to0[0][1], to1[0][0], to1[0][1], tq0[0][0], tq0[0][1], tq1[0][0], tq1[0][1], to0[1][0],
                    to0[1][1], to1[1][0], to1[1][1], tq0[1][0], tq0[1][1], tq1[1][0], tq1[1][1]);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(to0[s2][0], to0[s2][1]), pf[s2], dv0, 0, 0, 0);
      dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(to1[s2][0], to1[s2][1]), pf[s2], dv1, 0, 0, 0);
      dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(tq0[s2][0], tq0[s2][1]), dsf[s2], dk0, 0, 0, 0);
      dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_join(tq1[s2][0], tq1[s2][1]), dsf[s2], dk1, 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the dS^T rows are in LDS before the barrier
    sync_tiles(t + 2 < nt);
  }
  dq_tile(nt - 1);
  if (wave == 0 && h == 0 && (nt - 1) * 32 + r < p.Nq) p.Delta[stat0 + (nt - 1) * 32 + r] = delta_prev;
  if (kv) {
    bf16* kp = p.dK + ((long)b * p.Nk + key) * p.lddk + hd * 64;
    bf16* vp = p.dV + ((long)b * p.Nk + key) * p.lddv + hd * 64;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      bf16x4 a, c, e0, e1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[e] = f2bf(dk0[rg * 4 + e] * p.scale);
        c[e] = f2bf(dk1[rg * 4 + e] * p.scale);
        e0[e] = f2bf(dv0[rg * 4 + e]);
        e1[e] = f2bf(dv1[rg * 4 + e]);
      }
      *reinterpret_cast<bf16x4*>(kp + 8 * rg + 4 * h) = a;
      *reinterpret_cast<bf16x4*>(kp + 32 + 8 * rg + 4 * h) = c;
      *reinterpret_cast<bf16x4*>(vp + 8 * rg + 4 * h) = e0;
      *reinterpret_cast<bf16x4*>(vp + 32 + 8 * rg + 4 * h) = e1;
    }
  }
}

int check(long ld) { return (ld & 7) ? 1 : 0; }

}  // namespace

int g_attn_fused_bwd = 1;  // da_set_option("attn_fused_bwd", 0 | 1): the one-kernel backward for Nk <= 128 (cross-attention)

#ifdef DA_STAMPS
extern "C" int da_debug_set_attn_stamps(void* buf, int wgs) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamp_buf), &buf, sizeof(buf)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamp_wgs), &wgs, sizeof(wgs)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int da_attn_fwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, void* O,
                           long ldo, float* L2, int B, int H, int Nq, int Nk, float scale, hipStream_t stream) {
  DA_CLEAR_ERR();
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return DA_ERR_SHAPE;
  if (check(ldq) || check(ldk) || check(ldv) || check(ldo)) return DA_ERR_SHAPE;
  AttnParams p = {};
  p.Q = (const bf16*)Q; p.K = (const bf16*)K; p.V = (const bf16*)V; p.Out = (bf16*)O; p.L2 = L2;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
  p.scale = scale; p.sc = scale * 1.4426950408889634f;
  hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3((Nq + 127) / 128, H, B), dim3(256), 2 * FW_STAGE, stream, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_attn_fwd_causal(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, void* O,
                                  long ldo, float* L2, int B, int H, int N, float scale, hipStream_t stream) {
  DA_CLEAR_ERR();
  if (B <= 0 || H <= 0 || N <= 0) return DA_ERR_SHAPE;
  if (check(ldq) || check(ldk) || check(ldv) || check(ldo)) return DA_ERR_SHAPE;
  AttnParams p = {};
  p.Q = (const bf16*)Q; p.K = (const bf16*)K; p.V = (const bf16*)V; p.Out = (bf16*)O; p.L2 = L2;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
  p.B = B; p.H = H; p.Nq = N; p.Nk = N;
  p.scale = scale; p.sc = scale * 1.4426950408889634f;
  hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3((N + 127) / 128, H, B), dim3(256), 2 * FW_STAGE, stream, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}

extern "C" int da_attn_bwd(const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, const void* O,
                           long ldo, const void* dO, long lddo, const float* L2, float* Delta, void* dQ, long lddq,
                           void* dK, long lddk, void* dV, long lddv, int B, int H, int Nq, int Nk, float scale,
                           hipStream_t stream) {
  DA_CLEAR_ERR();
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return DA_ERR_SHAPE;
  if (check(ldq) || check(ldk) || check(ldv) || check(ldo) || check(lddo) || check(lddq) || check(lddk) ||
      check(lddv))
    return DA_ERR_SHAPE;
  AttnParams p = {};
  p.Q = (const bf16*)Q; p.K = (const bf16*)K; p.V = (const bf16*)V; p.O = (const bf16*)O; p.dO = (const bf16*)dO;
  p.dQ = (bf16*)dQ; p.dK = (bf16*)dK; p.dV = (bf16*)dV;
  p.L2 = const_cast<float*>(L2); p.Delta = Delta;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
  p.scale = scale; p.sc = scale * 1.4426950408889634f;
  if (g_attn_fused_bwd && Nk <= 128 && Nq >= 64) {  // all keys of an (image, head) fit one workgroup: one kernel for dQ, dK, dV
    static unsigned long long attr_done = 0;
    if (da_ensure_dyn_smem((const void*)attn_bwd_fused_kernel<4>, fu_smem(4), &attr_done) != DA_OK) return DA_ERR_LAUNCH;
    hipLaunchKernelGGL(attn_bwd_fused_kernel<4>, dim3(1, H, B), dim3(256), fu_smem(4), stream, p);
    DA_CHECK_LAUNCH();
    return DA_OK;
  }
  if (g_attn_fused_bwd >= 2 && Nk <= 256 && Nq >= 64) {  // ... as one 8-wave workgroup (da_set_option("attn_fused_bwd", 2))
    static unsigned long long attr_done8 = 0;
    if (da_ensure_dyn_smem((const void*)attn_bwd_fused_kernel<8>, fu_smem(8), &attr_done8) != DA_OK) return DA_ERR_LAUNCH;
    hipLaunchKernelGGL(attn_bwd_fused_kernel<8>, dim3(1, H, B), dim3(512), fu_smem(8), stream, p);
    DA_CHECK_LAUNCH();
    return DA_OK;
  }
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((Nq + 127) / 128, H, B), dim3(256), 2 * DQ_STAGE, stream, p);
  DA_CHECK_LAUNCH();
  if (Nq >= 128)
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, dim3((Nk + 127) / 128, H, B), dim3(256), 3 * kv_stage(64), stream, p);
  else
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<32>, dim3((Nk + 127) / 128, H, B), dim3(256), 3 * kv_stage(32), stream, p);
  DA_CHECK_LAUNCH();
  return DA_OK;
}
