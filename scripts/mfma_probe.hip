// Development probe: issue rate of v_mfma_f32_32x32x2_f32 as a function of (independent accumulator chains per
// wave) x (waves per SIMD).  hipcc --offload-arch=gfx950 -O3 scripts/mfma_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
  f32x16 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; c++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int c = 0; c < CH; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < CH; c++)
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
void run(int waves_per_simd) {
  float *out;
  hipMalloc(&out, 256 * 4 * 256 * 8 * sizeof(float));
  const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD of a CU
  const int iters = 20000 / CH;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f, 2.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 16 * CH * 4096.0;
  printf("chains/wave %d  waves/SIMD %d : %.1f TFLOP/s (%.2f ms)\n", CH, waves_per_simd, flops / ms / 1e9, ms);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 3, 4}) { run<1>(w); run<2>(w); run<4>(w); }
  return 0;
}
