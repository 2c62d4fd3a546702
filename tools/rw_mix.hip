// Micro-benchmark behind DESIGN 4.3 (round 4): can ONE workgroup per CU keep HBM reads (LDS-DMA, as the GEMM K loops issue
// them) and HBM writes (16-byte register stores, as their epilogues issue them) in flight together, or do they serialise?
//   mode 0  "tile loop": every wave loads its share of a 160 KB tile, waits (vmcnt(0) + barrier), stores its share of a
//           160 KB output tile, next tile - the structure of gemm_nt2's persistent forms (the next loads wait for the stores:
//           vmcnt retires in order)
//   mode 1  split roles: waves 0-7 only load (and wait for their own loads), waves 8-15 only store and never wait;
//           one barrier per tile keeps them in step
//   mode 2  loads only          mode 3  stores only
//   mode 4  every wave loads AND stores each tile, but waits with vmcnt(N) that leaves this tile's stores in flight
//   mode 5  stores only, in the lane layout of the direct GEMM epilogue: a 256 x 640-byte tile, wave (wm, wn) owns rows
//           64*wm.. and bytes 160*wn..; one instruction = 16 rows x 64 contiguous bytes (4 lanes x 16 B), row stride 640 B
//   mode 6  the same with 16 rows x 128 contiguous bytes per instruction pair (8 lanes x 16 B: what two more lane swaps would give)
//   mode 7  MFMA only: every wave issues 10 x MFMA_PER_PIECE register-only v_mfma_f32_16x16x32_bf16 per tile (a 256 x 320 x 320
//           tile costs 200 per wave; -DMFMA_PER_PIECE=80 makes the matrix work about as long as the store stream)
//   mode 8  the same MFMAs, then the wave's share of the tile's stores (no waits): does the store stream run beside the matrix pipe?
//   mode 9  MFMAs interleaved with the stores (one store per 20 MFMAs)
//   mode 10 the GEMM structure: per tile 5 K-steps of {2 LDS-DMA pieces per wave, 40 MFMAs, vmcnt(0) + barrier}, then the
//           wave's 10 stores (the next tile's first wait also waits for them: in-order vmcnt)
//   mode 12 mode 10 plus the W stream of a K = N = 320 layer: 40 KB per K-step from a 200 KB L2-resident buffer (as gemm_nt2 streams W)
//   mode 11 the streaming structure (gemm_nt_v3): per K-step {2 LDS-DMA pieces, 2 stores of the PREVIOUS tile, 40 MFMAs,
//           vmcnt(2) + barrier}: the stores stay in flight across the step-end wait
// Build / run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/rw_mix.hip -o tools/_bin/rw_mix && tools/_bin/rw_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int TILE = 160 * 1024;   // bytes read and bytes written per tile
constexpr int PIECES = TILE / 1024;  // 1-KiB wave-instructions per tile (160)
#ifndef MFMA_PER_STEP
#define MFMA_PER_STEP 40         // MFMAs per wave and K-step in modes 10-12 (40 = a 256 x 320 x 64 step at full rate)
#endif
#ifndef MFMA_PER_PIECE
#define MFMA_PER_PIECE 80        // MFMAs per wave between two of its ten store instructions (modes 7-9)
#endif

__device__ __forceinline__ void glds16(const void* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(1024) void rw_kernel(const char* __restrict__ in, char* __restrict__ out, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 64 KiB landing zone, never read
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
  for (int t = 0; t < tiles_per_wg; ++t) {
    const long tile = (long)blockIdx.x * tiles_per_wg + t;
    const char* src = in + tile * TILE;
    char* dst = out + tile * TILE;
    if (MODE == 0 || MODE == 2 || MODE == 4) {
#pragma unroll
      for (int j = 0; j < PIECES / 16; ++j) {
        const int pc = wave + 16 * j;
        glds16(src + pc * 1024 + lane * 16, smem + (pc & 63) * 1024);
      }
    }
    if (MODE == 1 && wave < 8) {
#pragma unroll
      for (int j = 0; j < PIECES / 8; ++j) {
        const int pc = wave + 8 * j;
        glds16(src + pc * 1024 + lane * 16, smem + (pc & 63) * 1024);
      }
    }
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (MODE == 0 || MODE == 3 || MODE == 4) {
#pragma unroll
      for (int j = 0; j < PIECES / 16; ++j) {
        const int pc = wave + 16 * j;
        *reinterpret_cast<u32x4*>(dst + pc * 1024 + lane * 16) = v;
      }
    }
    if (MODE == 5 || MODE == 6) {
      const int wm = wave >> 2, wn = wave & 3;
      const int r = lane & 15, q = lane >> 4;
#pragma unroll
      for (int j = 0; j < PIECES / 16; ++j) {
        int row, col;
        if (MODE == 5) {  // j = 2*i + pj for j < 8: strip i, 64-byte column pair pj; j = 8, 9: the 32-byte fifth column of two strips each
          if (j < 8) { row = wm * 64 + (j >> 1) * 16 + r; col = wn * 160 + (j & 1) * 64 + q * 16; }
          else { row = wm * 64 + ((j - 8) * 2 + (q & 1)) * 16 + r; col = wn * 160 + 128 + (q >> 1) * 16; }
        } else {          // 8 lanes per row: 128 contiguous bytes, 8 rows per instruction
          const int r8 = lane & 7, q8 = lane >> 3;
          if (j < 8) { row = wm * 64 + j * 8 + r8; col = wn * 160 + q8 * 16; }
          else { row = wm * 64 + (j - 8) * 32 + (q8 >> 1) * 8 + r8; col = wn * 160 + 128 + (q8 & 1) * 16; }
        }
        *reinterpret_cast<u32x4*>(dst + row * 640 + col) = v;
      }
    }
    if (MODE == 7 || MODE == 8 || MODE == 9) {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      bf16x8 a, b;
#pragma unroll
      for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(t + e); }
#pragma unroll
      for (int j = 0; j < PIECES / 16; ++j) {
#pragma unroll
        for (int m = 0; m < MFMA_PER_PIECE / 4; ++m)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
        if (MODE == 9) {
          const int pc = wave + 16 * j;
          *reinterpret_cast<u32x4*>(dst + pc * 1024 + lane * 16) = v;
        }
      }
      if (MODE == 8) {
#pragma unroll
        for (int j = 0; j < PIECES / 16; ++j) {
          const int pc = wave + 16 * j;
          *reinterpret_cast<u32x4*>(dst + pc * 1024 + lane * 16) = v;
        }
      }
      asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
    }
    if (MODE == 10 || MODE == 11 || MODE == 12) {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      bf16x8 a, b;
#pragma unroll
      for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(t + e); }
#pragma unroll
      for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int pc = wave + 16 * (2 * k + j);
          glds16(src + pc * 1024 + lane * 16, smem + (pc & 63) * 1024);
        }
        if (MODE == 12) {   // W slice of this K-step: 40 pieces over 16 waves, always the same 200 KB (L2 / MALL resident)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int pc = wave + 16 * j;
            if (pc < 40) glds16(in + (k * 40 + pc) * 1024 + lane * 16, smem + ((pc + 7) & 63) * 1024);
          }
        }
        if (MODE == 11 && t > 0) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int pc = wave + 16 * (2 * k + j);
            *reinterpret_cast<u32x4*>(dst - TILE + pc * 1024 + lane * 16) = v;   // previous tile's rows
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // (register-only MFMAs are not ordered by the asm waits: pin the step's shape)
#pragma unroll
        for (int m = 0; m < MFMA_PER_STEP; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 11 && t > 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("" ::"v"(acc));
      if (MODE == 10 || MODE == 12 || t == tiles_per_wg - 1) {
#pragma unroll
        for (int j = 0; j < PIECES / 16; ++j) {
          const int pc = wave + 16 * j;
          *reinterpret_cast<u32x4*>(dst + pc * 1024 + lane * 16) = v;
        }
      }
    }
    if (MODE == 1 && wave >= 8) {
#pragma unroll
      for (int j = 0; j < PIECES / 8; ++j) {
        const int pc = (wave - 8) + 8 * j;
        *reinterpret_cast<u32x4*>(dst + pc * 1024 + lane * 16) = v;
      }
    }
    if (MODE == 1) {
      if (wave < 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (MODE == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (MODE == 4) {  // the loads of this tile are older than its stores: wait for them, leave the 10 stores in flight
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int MODE>
float run(const char* in, char* out, int tiles, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute((const void*)rw_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  rw_kernel<MODE><<<256, 1024, 64 * 1024>>>(in, out, tiles);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) rw_kernel<MODE><<<256, 1024, 64 * 1024>>>(in, out, tiles);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 16;
  const long bytes = 256L * tiles * TILE;
  char *in, *out;
  hipMalloc(&in, bytes); hipMalloc(&out, bytes);
  hipMemset(in, 1, bytes); hipMemset(out, 0, bytes);
  const char* names[13] = {"0 tile loop (load, wait, store)", "1 loader waves / storer waves", "2 loads only", "3 stores only",
                          "4 load+store per wave, counted wait", "5 stores only, 16 rows x 64 B per instr", "6 stores only, 8 rows x 128 B per instr",
                          "7 MFMA only", "8 MFMAs, then the stores", "9 MFMAs interleaved with the stores",
                          "10 GEMM structure (vmcnt(0) per step)", "11 streaming structure (stores behind counted waits)",
                          "12 GEMM structure + 40 KB of W per step from L2"};
  float us[13];
  for (int r = 0; r < 2; ++r) {
    us[0] = run<0>(in, out, tiles, 10); us[1] = run<1>(in, out, tiles, 10); us[2] = run<2>(in, out, tiles, 10);
    us[3] = run<3>(in, out, tiles, 10); us[4] = run<4>(in, out, tiles, 10); us[5] = run<5>(in, out, tiles, 10); us[6] = run<6>(in, out, tiles, 10);
    us[7] = run<7>(in, out, tiles, 10); us[8] = run<8>(in, out, tiles, 10); us[9] = run<9>(in, out, tiles, 10);
    us[10] = run<10>(in, out, tiles, 10); us[11] = run<11>(in, out, tiles, 10); us[12] = run<12>(in, out, tiles, 10);
  }
  printf("256 workgroups x %d tiles x %d KB read + %d KB written = %.0f MB each way\n", tiles, TILE / 1024, TILE / 1024, bytes / 1e6);
  for (int m = 0; m < 13; ++m) {
    const double moved = m == 7 ? 0.0 : ((m == 2 || m == 3 || (m >= 5 && m < 10)) ? bytes : 2.0 * bytes);
    printf("mode %-40s %8.1f us  %5.2f TB/s  %6.0f TFLOP/s\n", names[m], us[m], moved / us[m] / 1e6,
           m >= 10 ? 256.0 * tiles * 16 * (5.0 * MFMA_PER_STEP) * 16384.0 / us[m] / 1e6 : (m >= 7 ? 256.0 * tiles * 16 * (2.5 * MFMA_PER_PIECE) * 16384.0 / us[m] / 1e6 : 0.0));
  }
  if (argc > 2) {  // order check: the GEMM-structure modes again, interleaved
    for (int r = 0; r < 3; ++r) {
      const float a10 = run<10>(in, out, tiles, 10), a12 = run<12>(in, out, tiles, 10), a11 = run<11>(in, out, tiles, 10);
      printf("again: mode 10 %.1f us   mode 12 %.1f us   mode 11 %.1f us\n", a10, a12, a11);
    }
  }
  return 0;
}
