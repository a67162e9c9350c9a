// a12. RPN head at inference (modeling/rpn/rpn_sparse3d.py:80-131, SingleConvRPNHead_Sparse3D): the three 1x1
// convolutions over the active sites of the selected maps -- t = relu(x W1^T + b1), objectness = t Wc^T + bc,
// regression = t Wr^T + br -- as ONE launch over the maps' rows where they lie (no concatenation, t never leaves the
// CU): three library GEMMs + a concatenation + an activation cost the launch thread 0.3 ms per building, more than the
// 0.6 GFLOP take on the matrix cores.
//
// A workgroup owns 32 site rows (the row space is the maps' rows laid end to end, which is the flattening order of
// cat_scales_obj_reg, rpn_sparse3d.py:19-77: scale, site, anchor) and C/32 waves.  Stage 1: wave w holds the 32x32
// accumulator of columns 32w.. of t (v_mfma_f32_32x32x2_f32 over Cin, A from the LDS tile of the rows, B from the
// k-interleaved packed weights as in k_conv); bias + ReLU; t replaces the rows in LDS.  Stage 2: the a objectness and
// 7a regression columns are one packed [C, 8a] operand (32 columns for a = 4), a 32-column tile per wave.
#include <algorithm>

#include "d3d_internal.h"

namespace d3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int kRpnMaxSlices = 8;   // channel slices per output tile in stage 2

struct RpnMaps {
  const float *p[D3D_RPN_MAX_MAPS];
  int start[D3D_RPN_MAX_MAPS + 1];  // first row of map m in the row space; start[n_maps] = all rows
  int n_maps;
};

// wp = packed[g][co][j] = W[co][4g + j] (W in nn.Linear layout [cout, cin]); one 32-column tile at column `colbase`,
// q-iterations [q0, q1) of the C / 8 (8 input channels each).  The weight fragments of QB iterations are requested together
// (they come straight from L2: one at a time, a 512-channel product waited 64 round trips).
template <int QB>
__device__ __forceinline__ f32x16 tile_product(const float *__restrict__ As, int lda, const float *__restrict__ wp,
                                               int cout, int colbase, int r, int h, int q0, int q1) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  const float *wl = wp + ((size_t)h * cout + colbase + r) * 4;
  for (int qb = q0; qb < q1; qb += QB) {
    f32x4 b[QB];
#pragma unroll
    for (int j = 0; j < QB; j++) {
      const int q = qb + j < q1 ? qb + j : q1 - 1;      // (a short last batch repeats its last fragment, unused)
      b[j] = *(const f32x4 *)(wl + (size_t)(2 * q) * cout * 4);
    }
#pragma unroll
    for (int j = 0; j < QB; j++) {
      if (qb + j < q1) {
        const f32x4 a = *(const f32x4 *)(As + r * lda + (qb + j) * 8 + h * 4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[j][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[j][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[j][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[j][3], acc, 0, 0, 0);
      }
    }
  }
  return acc;
}

// RELU_IN: the rows are rectified as they are loaded (the box head's fc6 output, roi_box_feature_extractors.py:110-115).
// stages: bit 0 = stage 1 (t = relu(rows W1^T + b1), kept in LDS; written to t_out when that is not null), bit 1 = stage 2
// (the two outputs from t); with stage 2 alone the rows ARE t.  Stage 2 reads the same fp32 values from LDS that stage 1
// would have stored to t_out, so a fused launch and the two stages launched apart give the same bits.
template <int C, bool RELU_IN>
__global__ __launch_bounds__(C / 32 * 64) void k_rpn_head(RpnMaps maps, const float *__restrict__ w1p,
                                                          const float *__restrict__ b1, const float *__restrict__ w2p,
                                                          const float *__restrict__ b2, int a, int out_tiles,
                                                          float *__restrict__ obj, float *__restrict__ reg, int stages,
                                                          float *__restrict__ t_out) {
  constexpr int W = C / 32, LDA = C + 4, LPR = C / 4;
  __shared__ __attribute__((aligned(16))) float As[32 * LDA];
  __shared__ float Ps[(W < kRpnMaxSlices * 2 ? W : kRpnMaxSlices * 2) * 1024];   // partial 32x32 tiles of stage 2
  const int n = maps.start[maps.n_maps];
  const int row0 = blockIdx.x * 32;
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6, r = lane & 31, h = lane >> 5;

  for (int i = tid; i < 32 * LPR; i += W * 64) {
    const int row = i / LPR, c4 = i % LPR, g = row0 + row;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (g < n) {
      int m = 0;
      while (m + 1 < maps.n_maps && g >= maps.start[m + 1]) m++;
      v = *(const f32x4 *)(maps.p[m] + (size_t)(g - maps.start[m]) * C + c4 * 4);
      if constexpr (RELU_IN) {
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = v[j] < 0.f ? 0.f : v[j];   // relu; a NaN stays a NaN
      }
    }
    *(f32x4 *)(As + row * LDA + c4 * 4) = v;
  }
  __syncthreads();
  // C/D layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  if (stages & 1) {
    const f32x16 acc = tile_product<16>(As, LDA, w1p, C, wib * 32, r, h, 0, C / 8);
    const int col = wib * 32 + r;
    const float bias = b1[col];
    __syncthreads();  // every wave has read the rows
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      float v = acc[i] + bias;
      v = v < 0.f ? 0.f : v;   // relu; a NaN stays a NaN
      As[row * LDA + col] = v;
      if (t_out && row0 + row < n) t_out[(size_t)(row0 + row) * C + col] = v;
    }
    __syncthreads();
  }
  if (!(stages & 2)) return;
  // Stage 2: the W waves split the product over (output tile, channel slice): S = W / out_tiles slices per tile (a power
  // of two), each a partial 32x32 tile in LDS; the partial tiles of an output tile are added in slice order (fixed:
  // the same bits from a fused launch and from this stage alone).
  const int nout = out_tiles * 32;
  int S = 1;
  while (2 * S * out_tiles <= W && 2 * S <= kRpnMaxSlices) S *= 2;
  const int nq = (C / 8) / S;
  for (int item = wib; item < out_tiles * S; item += W) {
    const int t = item / S, sl = item - t * S;
    const f32x16 acc = tile_product<8>(As, LDA, w2p, nout, t * 32, r, h, sl * nq, (sl + 1) * nq);
    float *pt = Ps + (size_t)item * 1024;
#pragma unroll
    for (int i = 0; i < 16; i++) pt[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
  }
  __syncthreads();
  for (int e = tid; e < out_tiles * 1024; e += W * 64) {
    const int t = e >> 10, row = (e >> 5) & 31, c = e & 31;
    const int col = t * 32 + c;
    const size_t g = (size_t)row0 + row;
    if (col >= 8 * a || g >= (size_t)n) continue;
    float v = 0.f;
    for (int sl = 0; sl < S; sl++) v += Ps[(size_t)(t * S + sl) * 1024 + row * 32 + c];
    v += b2[col];
    if (col < a)
      obj[g * a + col] = v;
    else
      reg[g * (7 * a) + (col - a)] = v;
  }
}

}  // namespace d3d

using namespace d3d;

extern "C" {

int d3d_rpn_head(const float *const *maps_host, const int *rows_host, int n_maps, int channels, const float *w1_packed,
                 const float *b1, const float *w2_packed, const float *b2, int a, float *objectness, float *regression,
                 void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(maps_host && rows_host && n_maps >= 1 && n_maps <= D3D_RPN_MAX_MAPS, "rpn_head: 1..%d maps",
              D3D_RPN_MAX_MAPS);
  D3D_REQUIRE(channels == 128 || channels == 256, "rpn_head: %d channels (built for 128 and 256)", channels);
  D3D_REQUIRE(a >= 1 && w1_packed && b1 && w2_packed && b2 && objectness && regression, "rpn_head: bad arguments");
  RpnMaps maps;
  long n = 0;
  for (int m = 0; m < D3D_RPN_MAX_MAPS; m++) {
    maps.p[m] = nullptr;
    maps.start[m] = (int)n;
    if (m < n_maps) {
      D3D_REQUIRE(rows_host[m] >= 0 && (rows_host[m] == 0 || maps_host[m]), "rpn_head: map %d", m);
      maps.p[m] = maps_host[m];
      n += rows_host[m];
    }
  }
  D3D_REQUIRE(n * 7L * a < (1L << 31), "rpn_head: %ld rows", n);
  maps.start[D3D_RPN_MAX_MAPS] = (int)n;
  for (int m = n_maps; m <= D3D_RPN_MAX_MAPS; m++) maps.start[m] = (int)n;
  maps.n_maps = n_maps;
  if (n == 0) return D3D_OK;
  const int out_tiles = (8 * a + 31) / 32;
  D3D_REQUIRE(out_tiles <= channels / 32, "rpn_head: %d output columns for %d channels (at most one 32-column tile per wave)", 8 * a, channels);
  const dim3 grid((unsigned)((n + 31) / 32));
  if (channels == 128)
    hipLaunchKernelGGL((k_rpn_head<128, false>), grid, dim3(256), 0, s, maps, w1_packed, b1, w2_packed, b2, a, out_tiles,
                       objectness, regression, 3, (float *)nullptr);
  else
    hipLaunchKernelGGL((k_rpn_head<256, false>), grid, dim3(512), 0, s, maps, w1_packed, b1, w2_packed, b2, a, out_tiles,
                       objectness, regression, 3, (float *)nullptr);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_mlp_heads(const float *x, int rows, int channels, int relu_in, const float *w1_packed, const float *b1,
                  float *t_out, const float *w2_packed, const float *b2, int a, float *out_a, float *out_7a, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(rows >= 0 && (channels == 128 || channels == 256 || channels == 512),
              "mlp_heads: %d channels (built for 128, 256 and 512)", channels);
  const int stages = (w1_packed ? 1 : 0) | (w2_packed ? 2 : 0);
  D3D_REQUIRE(stages != 0 && (!w1_packed || b1) && (!w2_packed || (b2 && a >= 1 && out_a && out_7a)) &&
                  (w2_packed || t_out), "mlp_heads: bad arguments");
  D3D_REQUIRE(!(relu_in && !w1_packed), "mlp_heads: the input is only rectified in front of the first stage");
  D3D_REQUIRE((long)rows * 7L * std::max(a, 1) < (1L << 31), "mlp_heads: %d rows", rows);
  if (rows == 0) return D3D_OK;
  D3D_REQUIRE(x, "mlp_heads: null input");
  RpnMaps maps;
  for (int m = 0; m < D3D_RPN_MAX_MAPS; m++) {
    maps.p[m] = m == 0 ? x : nullptr;
    maps.start[m] = m == 0 ? 0 : rows;
  }
  maps.start[D3D_RPN_MAX_MAPS] = rows;
  maps.n_maps = 1;
  const int out_tiles = w2_packed ? (8 * a + 31) / 32 : 0;
  D3D_REQUIRE(out_tiles <= channels / 32, "mlp_heads: %d output columns for %d channels", 8 * a, channels);
  const dim3 grid((unsigned)((rows + 31) / 32));
#define D3D_MLP_LAUNCH(CC, RI)                                                                                         \
  hipLaunchKernelGGL((k_rpn_head<CC, RI>), grid, dim3(CC / 32 * 64), 0, s, maps, w1_packed, b1, w2_packed, b2, a, out_tiles, \
                     out_a, out_7a, stages, t_out)
  if (channels == 128) {
    if (relu_in) D3D_MLP_LAUNCH(128, true); else D3D_MLP_LAUNCH(128, false);
  } else if (channels == 256) {
    if (relu_in) D3D_MLP_LAUNCH(256, true); else D3D_MLP_LAUNCH(256, false);
  } else {
    if (relu_in) D3D_MLP_LAUNCH(512, true); else D3D_MLP_LAUNCH(512, false);
  }
#undef D3D_MLP_LAUNCH
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

}  // extern "C"
