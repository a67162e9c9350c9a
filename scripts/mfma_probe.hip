// Development probe: issue rate of v_mfma_f32_32x32x2_f32 as a function of (independent accumulator chains per
// wave) x (waves per SIMD) x (independent VALU instructions issued per MFMA: do they co-execute?).
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH, int NV>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
  f32x16 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; c++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; j++) v[j] = a0 * j + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
#pragma unroll
      for (int c = 0; c < CH; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; j++) v[j % 16] = v[j % 16] * 1.0001f + b0;   // independent VALU work (v_fma / v_mul+v_add)
    }
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < CH; c++)
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[c][i];
#pragma unroll
  for (int j = 0; j < 16; j++) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH, int NV>
void run(int waves_per_simd) {
  float *out;
  (void)hipMalloc(&out, 256 * 4 * 256 * 8 * sizeof(float));
  const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD of a CU
  const int iters = 20000 / CH;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CH, NV>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.f, 2.f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<CH, NV>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 16 * CH * 4096.0;
  printf("chains/wave %d  VALU/MFMA %2d  waves/SIMD %d : %.1f TFLOP/s (%.2f ms)\n", CH, NV / CH, waves_per_simd, flops / ms / 1e9, ms);
  (void)hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) { run<1, 0>(w); run<1, 2>(w); run<1, 4>(w); run<1, 8>(w); run<1, 16>(w); }
  run<2, 0>(1); run<4, 0>(1);
  return 0;
}
