// a8/a9. BatchNorm (+ leaky ReLU) and feature-plane add.  HBM-bound: one fp64 column-sum pass
// (deterministic two-level reduction, no atomics) and one fused affine + activation pass.
#include "d3d_internal.h"

namespace d3d {

static constexpr int kStatBlocks = 512;  // partial-sum rows (>= 2 per CU)

// partial[b][0..C) = sum, partial[b][C..2C) = sum of squares over the rows of slice b
__global__ __launch_bounds__(256) void k_bn_partial(const float *__restrict__ x, int rows, int C,
                                                    double *__restrict__ partial) {
  extern __shared__ double red[];  // [256][2]
  const int tid = threadIdx.x;
  const int lanes_per_row = C < 256 ? C : 256;   // threads covering one row
  const int row_lanes = 256 / lanes_per_row;     // rows handled concurrently (C | 256 assumed if C<256)
  const int rl = tid / lanes_per_row, cl = tid % lanes_per_row;
  const int per = (rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  for (int c = cl; c < C; c += lanes_per_row) {
    double s = 0, ss = 0;
    if (rl < row_lanes)
      for (int r = r0 + rl; r < r1; r += row_lanes) {
        double v = (double)x[(size_t)r * C + c];
        s += v;
        ss += v * v;
      }
    red[tid * 2] = s;
    red[tid * 2 + 1] = ss;
    __syncthreads();
    if (rl == 0) {
      for (int j = 1; j < row_lanes; j++) {
        s += red[(j * lanes_per_row + cl) * 2];
        ss += red[(j * lanes_per_row + cl) * 2 + 1];
      }
      partial[(size_t)blockIdx.x * 2 * C + c] = s;
      partial[(size_t)blockIdx.x * 2 * C + C + c] = ss;
    }
    __syncthreads();
  }
}

// mode 0: batch_stats -> mean, unbiased var (torch .mean(0)/.var(0), batchNormalization.py:54-55)
// mode 1: train       -> save_mean, save_invstd, running update (BatchNormalization.cpp:20-38)
__global__ __launch_bounds__(256) void k_bn_finish(const double *__restrict__ partial, int nblk,
                                                   int rows, int C, int mode, float *o0, float *o1,
                                                   float *running_mean, float *running_var,
                                                   float eps, float momentum) {
  // block = 32 channels x 8 slices of the partial rows; fixed summation order (deterministic)
  __shared__ double red[256][2];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0, ss = 0;
  if (c < C)
    for (int b = sl; b < nblk; b += 8) {
      s += partial[(size_t)b * 2 * C + c];
      ss += partial[(size_t)b * 2 * C + C + c];
    }
  red[threadIdx.x][0] = s;
  red[threadIdx.x][1] = ss;
  __syncthreads();
  if (sl != 0 || c >= C) return;
  for (int j = 1; j < 8; j++) {
    s += red[j * 32 + cl][0];
    ss += red[j * 32 + cl][1];
  }
  double mean = s / rows;
  double m2 = ss - mean * mean * rows;  // sum of squared deviations
  if (m2 < 0) m2 = 0;
  if (mode == 0) {
    o0[c] = (float)mean;
    o1[c] = (float)(m2 / (rows - 1));
  } else if (mode == 2) {  // eval with batch statistics: invstd from the unbiased variance (float, then + eps)
    o0[c] = (float)mean;
    o1[c] = powf((float)(m2 / (rows - 1)) + eps, -0.5f);
  } else {
    o0[c] = (float)mean;
    running_mean[c] = momentum * running_mean[c] + (1 - momentum) * (float)mean;
    running_var[c] = momentum * running_var[c] + (1 - momentum) * (float)(m2 / (rows - 1));
    o1[c] = powf((float)(m2 / rows) + eps, -0.5f);
  }
}

__global__ void k_bn_eval_stats(const float *running_mean, const float *running_var, int C, float eps,
                                float *save_mean, float *save_invstd) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  save_mean[c] = running_mean[c];
  save_invstd[c] = powf(running_var[c] + eps, -0.5f);  // BatchNormalization.cpp:40-44
}

// y = leaky(x * w + b), w = invstd*gamma, b = -mean*w + beta (BatchNormalization.cpp:46-59)
__global__ __launch_bounds__(256) void k_bn_apply(const float *__restrict__ x, float *__restrict__ y,
                                                  size_t total, int C,
                                                  const float *__restrict__ save_mean,
                                                  const float *__restrict__ save_invstd,
                                                  const float *__restrict__ weight,
                                                  const float *__restrict__ bias, float leakiness) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= total) return;
  if ((C & 3) == 0) {
    const int c = (int)(i % C);
    float4 v = *(const float4 *)(x + i);
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float w = save_invstd[c + j] * (weight ? weight[c + j] : 1.f);
      float b = -save_mean[c + j] * w + (bias ? bias[c + j] : 0.f);
      float t = o[j] * w + b;
      o[j] = t * ((t > 0) ? 1.f : leakiness);
    }
    *(float4 *)(y + i) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    for (size_t k = i; k < i + 4 && k < total; k++) {
      const int c = (int)(k % C);
      float w = save_invstd[c] * (weight ? weight[c] : 1.f);
      float b = -save_mean[c] * w + (bias ? bias[c] : 0.f);
      float t = x[k] * w + b;
      y[k] = t * ((t > 0) ? 1.f : leakiness);
    }
  }
}

__global__ __launch_bounds__(256) void k_add(const float *__restrict__ a, const float *__restrict__ b,
                                             float *__restrict__ o, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 x = *(const float4 *)(a + i), y = *(const float4 *)(b + i);
    *(float4 *)(o + i) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  } else {
    for (; i < n; i++) o[i] = a[i] + b[i];
  }
}

static int run_partial(const float *in, int rows, int C, void *scratch, size_t scratch_bytes,
                       hipStream_t s, int *nblk_out) {
  D3D_REQUIRE(C > 0 && C <= 4096, "batch norm: planes=%d out of range", C);
  D3D_REQUIRE(C >= 256 ? (C % 256 == 0) : (256 % C == 0), "batch norm: planes=%d must divide 256 or be a multiple of 256", C);
  D3D_REQUIRE(scratch && scratch_bytes >= d3d_bn_scratch_bytes(C), "batch norm: scratch too small");
  int nblk = kStatBlocks;
  if (rows < nblk * 64) nblk = (rows + 63) / 64;
  if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(k_bn_partial, dim3(nblk), dim3(256), 256 * 2 * sizeof(double), s, in, rows, C, (double *)scratch);
  D3D_LAUNCH_CHECK();
  *nblk_out = nblk;
  return D3D_OK;
}

}  // namespace d3d

using namespace d3d;

extern "C" {

size_t d3d_bn_scratch_bytes(int planes) { return (size_t)kStatBlocks * 2 * planes * sizeof(double); }

int d3d_bn_batch_stats(const float *in, int rows, int planes, float *mean, float *var_unbiased,
                       void *scratch, size_t scratch_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(in && mean && var_unbiased && rows > 0, "bn_batch_stats: bad arguments");
  int nblk;
  int rc = run_partial(in, rows, planes, scratch, scratch_bytes, s, &nblk);
  if (rc) return rc;
  hipLaunchKernelGGL(k_bn_finish, dim3((planes + 31) / 32), dim3(256), 0, s, (const double *)scratch, nblk, rows, planes, 0, mean, var_unbiased, nullptr, nullptr, 0.f, 0.f);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_bn_batch_invstd(const float *in, int rows, int planes, float eps, float *mean, float *invstd,
                        void *scratch, size_t scratch_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(in && mean && invstd && rows > 0, "bn_batch_invstd: bad arguments");
  int nblk;
  int rc = run_partial(in, rows, planes, scratch, scratch_bytes, s, &nblk);
  if (rc) return rc;
  hipLaunchKernelGGL(k_bn_finish, dim3((planes + 31) / 32), dim3(256), 0, s, (const double *)scratch, nblk, rows, planes, 2, mean, invstd, nullptr, nullptr, eps, 0.f);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_bn_forward(const float *in, float *out, int rows, int planes, float *save_mean,
                   float *save_invstd, float *running_mean, float *running_var, const float *weight,
                   const float *bias, float eps, float momentum, int train, float leakiness,
                   void *scratch, size_t scratch_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(save_mean && save_invstd && running_mean && running_var && planes > 0 && rows >= 0, "bn_forward: bad arguments");
  if (rows == 0) return D3D_OK;
  D3D_REQUIRE(in && out, "bn_forward: null features");
  if (train) {
    int nblk;
    int rc = run_partial(in, rows, planes, scratch, scratch_bytes, s, &nblk);
    if (rc) return rc;
    hipLaunchKernelGGL(k_bn_finish, dim3((planes + 31) / 32), dim3(256), 0, s, (const double *)scratch, nblk, rows, planes, 1, save_mean, save_invstd, running_mean, running_var, eps, momentum);
  } else {
    hipLaunchKernelGGL(k_bn_eval_stats, dim3((planes + 63) / 64), dim3(64), 0, s, running_mean, running_var, planes, eps, save_mean, save_invstd);
  }
  size_t total = (size_t)rows * planes;
  hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)((total / 4 + 256) / 256)), dim3(256), 0, s, in, out, total, planes, save_mean, save_invstd, weight, bias, leakiness);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_bn_apply(const float *in, float *out, int rows, int planes, const float *mean, const float *invstd,
                 const float *weight, const float *bias, float leakiness, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (rows == 0) return D3D_OK;
  D3D_REQUIRE(in && out && mean && invstd && planes > 0 && rows > 0, "bn_apply: bad arguments");
  size_t total = (size_t)rows * planes;
  hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)((total / 4 + 256) / 256)), dim3(256), 0, s, in, out, total, planes, mean, invstd, weight, bias, leakiness);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_add(const float *a, const float *b, float *out, size_t n, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return D3D_OK;
  D3D_REQUIRE(a && b && out, "add: null pointer");
  hipLaunchKernelGGL(k_add, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, a, b, out, n);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

}  // extern "C"
