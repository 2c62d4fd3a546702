// Sustained bf16 MFMA ceiling of the chip: register-resident v_mfma_f32_16x16x32_bf16 loop, no memory traffic (dev tool).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(const char* name, int blocks, int threads) {
  float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 1000);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * (threads / 64) * iters * NACC * 16.0 * 16 * 32 * 2;
    printf("%s rep %d: %.2f ms  %.1f TFLOP/s\n", name, rep, ms, fl / ms / 1e9);
  }
  hipFree(out);
}
int main() {
  run<8>("8 waves/CU (2/SIMD), 8 accumulators", 256, 512);
  run<16>("8 waves/CU (2/SIMD), 16 accumulators", 256, 512);
  run<8>("4 waves/CU (1/SIMD), 8 accumulators", 256, 256);
  return 0;
}
