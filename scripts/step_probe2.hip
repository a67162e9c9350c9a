// Development probe 2: step_probe + the WEIGHT traffic of k_conv -- every wave fetches, per q-iteration, one 1 KB fragment
// (64 lanes x 16 B) of the step's weight tile straight from global memory (an L2-resident array of 27 tiles), through a
// ring of QA iterations, blocks walking the 27 offsets from different starting points -- and optionally the GATHER
// traffic (NIT 16-byte loads per thread and step from random rows of a large array, committed to LDS a step later).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WPB, int NQ, int QA, bool WLOAD, bool GATHER, int NV, bool PRO = false, bool IDXL = false>
__global__ __launch_bounds__(WPB * 64) void k(float *out, int steps, const float *__restrict__ wbuf, const float *__restrict__ rows,
                                             const int *__restrict__ ridx, int n_rows) {
  static_assert(NQ * 8 <= 128 && WPB * 32 <= 128, "the probe's arrays hold 128-channel rows and 27 x 128 x 128 weights");
  constexpr int CIN = NQ * 8, COUT = WPB * 32, LDA = CIN + 4, LPR = CIN / 4, RPP = WPB * 64 / LPR, NIT = 32 / RPP;
  __shared__ __attribute__((aligned(16))) float As[32 * LDA];
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int grow = threadIdx.x / LPR, gc4 = threadIdx.x % LPR;
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; j++) v[j] = 1.0f * j + threadIdx.x;
  for (int i = threadIdx.x; i < 32 * LDA; i += WPB * 64) As[i] = 1.f + i;
  __syncthreads();
  const size_t tile = (size_t)CIN * COUT;             // floats of one offset's weights
  int kk = (blockIdx.x * 7) % 27;
  const float *wl = wbuf + ((size_t)h * COUT + wib * 32 + r) * 4;
  f32x4 ring[QA];
  f32x4 bconst = {2.f, 3.f, 4.f, 5.f};
  if (WLOAD) {
#pragma unroll
    for (int q = 0; q < QA; q++) ring[q] = *(const f32x4 *)(wl + kk * tile + (size_t)(2 * q) * COUT * 4);
  }
  f32x4 stage[NIT > 0 ? NIT : 1];
  int bi = blockIdx.x * 32;
  int idx[NIT > 0 ? NIT : 1];
#pragma unroll
  for (int it = 0; it < NIT; it++) idx[it] = (bi + it * RPP + grow) % n_rows;
  if (PRO) {
    // the chain a block of k_conv starts with: mask word -> (dependent) indices -> (dependent) rows -> commit
    const int m = __builtin_amdgcn_readfirstlane(ridx[blockIdx.x % n_rows]);
    kk = (kk + (m & 1)) % 27;
#pragma unroll
    for (int it = 0; it < NIT; it++) idx[it] = ridx[(m + bi + it * RPP + grow) % n_rows];
#pragma unroll
    for (int it = 0; it < NIT; it++) stage[it] = *(const f32x4 *)(rows + (size_t)idx[it] * CIN + gc4 * 4);
#pragma unroll
    for (int it = 0; it < NIT; it++) *(f32x4 *)(As + (it * RPP + grow) * LDA + gc4 * 4) = stage[it];
    __syncthreads();
  }
  for (int st = 0; st < steps; st++) {
    const int nk = (kk + 1) % 27;
    if (GATHER && st > 0) {
#pragma unroll
      for (int it = 0; it < NIT; it++) *(f32x4 *)(As + (it * RPP + grow) * LDA + gc4 * 4) = stage[it];
    }
#pragma unroll
    for (int j = 0; j < NV; j++) v[j % 16] = v[j % 16] * 1.0001f + 2.f;
    __syncthreads();
    if (GATHER) {
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int row = IDXL ? idx[it] : ridx[(bi + it * RPP + grow) % n_rows];
        stage[it] = *(const f32x4 *)(rows + (size_t)row * CIN + gc4 * 4);
      }
      bi += 32 * 977;
      if (IDXL) {     // the indices of the step after next (a load the NEXT step's gather addresses depend on)
#pragma unroll
        for (int it = 0; it < NIT; it++) idx[it] = ridx[(bi + it * RPP + grow) % n_rows];
      }
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      f32x4 b = WLOAD ? ring[q % QA] : bconst;
      if (WLOAD) {
        const int qq = (q + QA) % NQ;
        const size_t base = (q + QA < NQ ? kk : nk) * tile;
        ring[q % QA] = *(const f32x4 *)(wl + base + (size_t)(2 * qq) * COUT * 4);
      }
      const f32x4 a = *(const f32x4 *)(As + r * LDA + q * 8 + h * 4);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    kk = nk;
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += acc[i];
#pragma unroll
  for (int j = 0; j < 16; j++) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static float *g_w, *g_rows, *g_out;
static int *g_idx;
static const int kRows = 400000;
template <int WPB, int NQ, int QA, bool WLOAD, bool GATHER, int NV, bool PRO = false, bool IDXL = false>
void run(int waves_per_simd, const char *tag, int steps_arg = 16) {
  const int blocks = 256 * 4 * 4 / WPB * 6;     // six rounds at 4 waves per SIMD
  const size_t dyn = waves_per_simd >= 4 ? 0 : (waves_per_simd == 3 ? 40 : waves_per_simd == 2 ? 70 : 150) * 1024 / (16 / WPB > 0 ? 1 : 1);   // dynamic LDS that limits the resident workgroups
  // resident waves per SIMD = 160 KB / (static + dynamic LDS per block) * WPB / 4
  const size_t want_blocks_per_cu = (size_t)waves_per_simd * 4 / WPB;
  const size_t stat = (size_t)32 * (NQ * 8 + 4) * 4;
  size_t lds_per_block = waves_per_simd >= 4 ? 0 : (size_t)160 * 1024 / (want_blocks_per_cu + 0) - stat - 512;
  if (lds_per_block > 60 * 1024) lds_per_block = 60 * 1024;
  (void)dyn;
  const int steps = steps_arg;                                      // as a block of k_conv: ~16 active offsets
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<WPB, NQ, QA, WLOAD, GATHER, NV, PRO, IDXL>), dim3(blocks), dim3(WPB * 64), lds_per_block, 0, g_out, steps, g_w, g_rows, g_idx, kRows);
  (void)hipEventRecord(e0);
  for (int rep = 0; rep < 5; rep++)
    hipLaunchKernelGGL((k<WPB, NQ, QA, WLOAD, GATHER, NV, PRO, IDXL>), dim3(blocks), dim3(WPB * 64), lds_per_block, 0, g_out, steps, g_w, g_rows, g_idx, kRows);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = 5.0 * blocks * WPB * steps * NQ * 4 * 4096.0;
  printf("%-44s waves/block %d q/step %2d ring %d : %6.1f TFLOP/s = %.2f of 157.3\n", tag, WPB, NQ, QA, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
}
int main() {
  (void)hipMalloc(&g_w, (size_t)27 * 128 * 128 * 4);
  (void)hipMalloc(&g_rows, (size_t)kRows * 128 * 4);
  (void)hipMalloc(&g_idx, (size_t)kRows * 4);
  (void)hipMalloc(&g_out, (size_t)256 * 16 * 64 * 64 * sizeof(float));
  (void)hipMemset(g_w, 0, (size_t)27 * 128 * 128 * 4);
  (void)hipMemset(g_rows, 0, (size_t)kRows * 128 * 4);
  int *hidx = (int *)malloc(kRows * 4);
  srand(1);
  for (int i = 0; i < kRows; i++) hidx[i] = rand() % kRows;
  (void)hipMemcpy(g_idx, hidx, kRows * 4, hipMemcpyHostToDevice);
  run<2, 8, 8, false, false, 64>(4, "64->64 structure only, 8 q per barrier pair");
  run<2, 16, 8, false, false, 128>(4, "64->64 structure only, 16 q per barrier pair", 8);
  run<2, 8, 8, true, true, 64, true, true>(4, "64->64 all traffic, 8 q per barrier pair");
  run<2, 16, 8, true, true, 128, true, true>(4, "64->64 all traffic, 16 q per barrier pair", 8);
  return 0;
}
