// Development probe: the step structure of k_conv without any global memory traffic -- per step NQ x (one ds_read_b128 of the
// A tile + 4 v_mfma_f32_32x32x2_f32), a commit of NW ds_write_b128 + NV VALU instructions, NB workgroup barriers -- to
// see which part of the structure keeps the matrix pipe idle.  WPB waves share a row tile (as the 2 / 4 waves of a block).
// hipcc --offload-arch=gfx950 -O3 scripts/step_probe.hip -o step_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WPB, int NQ, int NW, int NV, int NB>
__global__ __launch_bounds__(WPB * 64) void k(float *out, int steps, float a0, float b0) {
  constexpr int LDA = NQ * 8 + 4;
  __shared__ __attribute__((aligned(16))) float As[32 * LDA];
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  f32x4 b = {b0, b0 + 1, b0 + 2, b0 + 3};
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; j++) v[j] = a0 * j + threadIdx.x;
  for (int i = threadIdx.x; i < 32 * LDA; i += WPB * 64) As[i] = a0 + i;
  __syncthreads();
  for (int st = 0; st < steps; st++) {
    // commit: NW ds_write_b128 + NV VALU
#pragma unroll
    for (int j = 0; j < NV; j++) v[j % 16] = v[j % 16] * 1.0001f + b0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const int row = (threadIdx.x / (NQ * 2) + w * (WPB * 64 / (NQ * 2))) & 31, c4 = threadIdx.x % (NQ * 2);
      *(f32x4 *)(As + row * LDA + c4 * 4) = f32x4{v[w], v[w + 1], v[w + 2], v[w + 3]};
    }
    if (NB >= 1) __syncthreads();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const f32x4 a = *(const f32x4 *)(As + r * LDA + q * 8 + h * 4);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (NB >= 2) __syncthreads();
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += acc[i];
#pragma unroll
  for (int j = 0; j < 16; j++) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WPB, int NQ, int NW, int NV, int NB>
void run(int waves_per_simd) {
  float *out;
  (void)hipMalloc(&out, (size_t)256 * 16 * 64 * 8 * sizeof(float));
  const int blocks = 256 * 4 * waves_per_simd / WPB;
  const int steps = 4000 / NQ;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<WPB, NQ, NW, NV, NB>), dim3(blocks), dim3(WPB * 64), 0, 0, out, 10, 1.f, 2.f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<WPB, NQ, NW, NV, NB>), dim3(blocks), dim3(WPB * 64), 0, 0, out, steps, 1.f, 2.f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * WPB * steps * NQ * 4 * 4096.0;
  printf("waves/block %d  q/step %2d  ds_write %d  VALU %3d  barriers %d  waves/SIMD %d : %6.1f TFLOP/s = %.2f of 157.3\n", WPB, NQ, NW, NV,
         NB, waves_per_simd, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
  (void)hipFree(out);
}
int main() {
  for (int w : {4, 2}) {
    run<2, 8, 0, 0, 0>(w);
    run<2, 8, 0, 0, 2>(w);
    run<2, 8, 4, 0, 2>(w);
    run<2, 8, 4, 32, 2>(w);
    run<2, 8, 4, 64, 2>(w);
    run<2, 8, 4, 64, 1>(w);
    run<2, 8, 4, 100, 2>(w);
    run<4, 16, 4, 64, 2>(w);
    run<4, 16, 4, 32, 2>(w);
    run<4, 16, 0, 0, 0>(w);
    run<1, 8, 4, 64, 0>(w);
  }
  return 0;
}
