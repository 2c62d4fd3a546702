// Dev tool: can a GroupNorm pass hold its (image, channel-range) slab in registers and touch HBM once per tensor?
// Each workgroup loads HW pixels x CW channels of a [B*HW, C] bf16 tensor (16-B loads, all in flight at once), reduces,
// rescales and writes the slab back.  CW*2 bytes is not a multiple of the 128-B line for the real group widths (C/32 = 10
// channels), so neighbouring workgroups share lines: `peers8` places the parts of one image 8 workgroup ids apart (same XCD,
// same L2).  Prints effective GB/s of useful bytes.
// build: hipcc --offload-arch=gfx950 -O3 tools/slab_pass.hip -o gpurun_out/slab_pass ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

template <int NL, bool DUAL>
__global__ __launch_bounds__(1024) void slab_kernel(const u16x8* __restrict__ x, const u16x8* __restrict__ dy, u16x8* __restrict__ y, int HW,
                                                    int C8, int chunks, int P, int parts, int peers8) {
  int img, part;
  const int id = blockIdx.x;
  if (peers8) {
    const int g = id / (8 * parts), r = id - g * 8 * parts;
    part = r / 8;
    img = g * 8 + (r & 7);
  } else {
    img = id / parts;
    part = id - img * parts;
  }
  const int t = threadIdx.x;
  const int j = t % chunks, p0 = t / chunks;
  const bool live = p0 < P;
  const long base = (long)img * HW * C8 + part * chunks + j;
  u16x8 a[NL], b[DUAL ? NL : 1];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int p = p0 + k * P;
    const bool ok = live && p < HW;
    a[k] = ok ? __builtin_nontemporal_load(&x[base + (long)p * C8]) : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (DUAL) b[k] = ok ? __builtin_nontemporal_load(&dy[base + (long)p * C8]) : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  unsigned s = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k)
    for (int e = 0; e < 8; ++e) s += a[k][e] + (DUAL ? b[k][e] : 0);
  __shared__ unsigned red[1024];
  red[t] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  const unsigned short m = (unsigned short)red[0];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int p = p0 + k * P;
    if (live && p < HW) {
      u16x8 o = a[k];
      for (int e = 0; e < 8; ++e) o[e] = (unsigned short)(o[e] + m + (DUAL ? b[k][e] : 0));
      __builtin_nontemporal_store(o, &y[base + (long)p * C8]);
    }
  }
}

template <int NL, bool DUAL>
static void run(int B, int HW, int C, int CW, int peers8) {
  const int chunks = CW / 8, P = 1024 / chunks, parts = C / CW;
  if ((HW + P - 1) / P > NL) { printf("NL too small\n"); return; }
  const size_t n = (size_t)B * HW * C;
  u16x8 *x, *dy, *y;
  hipMalloc(&x, n * 2); hipMalloc(&dy, n * 2); hipMalloc(&y, n * 2);
  hipMemset(x, 1, n * 2); hipMemset(dy, 1, n * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((slab_kernel<NL, DUAL>), dim3(B * parts), dim3(1024), 0, 0, x, dy, y, HW, C / 8, chunks, P, parts, peers8);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = (double)n * 2 * (DUAL ? 3 : 2);
  printf("B=%d HW=%d C=%d CW=%d (%d B segments) NL=%d %s peers8=%d: %.1f us  %.2f TB/s useful\n", B, HW, C, CW, CW * 2, NL,
         DUAL ? "x+dy->dx" : "x->y", peers8, best * 1e3, bytes / best / 1e9);
  hipFree(x); hipFree(dy); hipFree(y);
}

int main() {
  for (int peers8 = 0; peers8 < 2; ++peers8) {
    run<21, false>(256, 1024, 320, 160, peers8);   // forward, 2 parts per image
    run<11, false>(256, 1024, 320, 80, peers8);    // forward, 4 parts
    run<11, true>(256, 1024, 320, 80, peers8);     // backward, 4 parts (x and dy resident)
    run<6, true>(256, 1024, 320, 40, peers8);      // backward, 8 parts
    run<21, false>(256, 1024, 640, 160, peers8);   // concatenated input
    run<11, true>(256, 256, 640, 320, peers8);     // level 1: 256 pixels, 2 parts of 320 channels (640-B segments, line aligned)
    run<11, false>(256, 256, 640, 320, peers8);
    run<11, true>(256, 64, 1280, 1280, peers8);    // level 2: whole image rows
  }
  return 0;
}
