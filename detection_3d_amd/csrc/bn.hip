// a8/a9. BatchNorm (+ leaky ReLU) and feature-plane add.  HBM-bound: one fp64 column-sum pass
// (deterministic two-level reduction, no atomics) and one fused affine + activation pass.
#include "d3d_internal.h"

#include <type_traits>

namespace d3d {

static constexpr int kStatThreads = 1024;
static constexpr int kStatBlocks = 128;   // row slices: every workgroup pays an agent-scope release / acquire (an L2
                                          // write-back on this multi-XCD part) and its partials a reduction pass --
                                          // measured per building: 512 slices 0.60 ms, 256 0.53, 128 0.49, 64 0.53
                                          // (1024 slices for the 180 MB maps of a 4 x 1 M-point bf16 batch: no change --
                                          // those passes run beside the rulebooks' hash probes, which hold the HBM)
static constexpr int kStatGroup = 16;     // slices per first-level group
static constexpr int kStatMaxGroups = kStatBlocks / kStatGroup;
static constexpr size_t kTicketBytes = 256;  // [0] = groups done, [1 + g] = slices of group g done

// sums n rows of src[n][V] column-wise in a fixed order: thread (slice sl, value vi) adds rows sl, sl+SL, ...
// and slice 0 adds the slices in order; dst[v] receives the result.  All kStatThreads threads must call.
__device__ __forceinline__ void stat_reduce_rows(const double *__restrict__ src, int n, int V, double *__restrict__ dst,
                                                 double *red) {
  const int tid = threadIdx.x;
  const int VP = V < kStatThreads ? V : kStatThreads, SL = kStatThreads / VP;
  const int sl = tid / VP, vi = tid - sl * VP;
  for (int vb = 0; vb < V; vb += VP) {
    const int v = vb + vi;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int b = sl;
    for (; b + 3 * SL < n; b += 4 * SL) {
      a0 += src[(size_t)b * V + v];
      a1 += src[(size_t)(b + SL) * V + v];
      a2 += src[(size_t)(b + 2 * SL) * V + v];
      a3 += src[(size_t)(b + 3 * SL) * V + v];
    }
    for (; b < n; b += SL) a0 += src[(size_t)b * V + v];
    double acc = (a0 + a1) + (a2 + a3);
    red[tid] = acc;
    __syncthreads();
    if (sl == 0) {
      for (int q = 1; q < SL; q++) acc += red[q * VP + vi];
      dst[v] = acc;
    }
    __syncthreads();
  }
}

// 4 consecutive channels of a row as fp32: fp32 storage (16-byte load) or bf16 storage (8-byte load)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ f32x4_t load4(const T *p);
template <>
__device__ __forceinline__ f32x4_t load4<float>(const float *p) {
  return *(const f32x4_t *)p;
}
template <>
__device__ __forceinline__ f32x4_t load4<unsigned short>(const unsigned short *p) {
  const uint2 v = *(const uint2 *)p;
  return f32x4_t{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                 __uint_as_float(v.y & 0xffff0000u)};
}
__device__ __forceinline__ void store4(float *p, f32x4_t v) { *(f32x4_t *)p = v; }
__device__ __forceinline__ void store4(unsigned short *p, f32x4_t v) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 b;
#pragma unroll
  for (int j = 0; j < 4; j++) b[j] = (__bf16)v[j];   // round to nearest even
  *(uint2 *)p = __builtin_bit_cast(uint2, b);
}

// Column statistics in ONE launch.  Workgroup b sums slice b of the rows in fp64 (threads = float4 channel
// groups x concurrent rows, 4 rows in flight per thread) and parks sum / sum-of-squares in partial[b]; the
// last workgroup of each group of kStatGroup slices to arrive (ticket counter) adds that group's partials,
// and the last group to finish adds the group sums and writes the statistics.  Every sum has a fixed order,
// so the result does not depend on which workgroups happen to be last (deterministic, no float atomics).
//   mode 0: mean, unbiased var   (torch .mean(0)/.var(0), batchNormalization.py:54-55)
//   mode 1: train -> save_mean, save_invstd (biased), running update (BatchNormalization.cpp:20-38)
//   mode 2: mean, powf(unbiased var + eps, -0.5)   (eval with batch statistics)
// HBM-bound: rows * C * 4 bytes read once.  The tickets are zero on entry and are left zero.
// T = double: `x` holds src_rows rows of [2 C] column sums / sums of squares that the producing convolution left per
// row block (d3d_bn_prologue.out_stats); slice b adds its share of them instead of reading the tensor, the rest is the
// same two-level finish.  `rows` stays the number of tensor rows the statistics are over.
template <typename T>
__global__ __launch_bounds__(kStatThreads) void k_bn_stats(const T *__restrict__ x, int rows, int C,
                                                           unsigned int *tickets, double *partial, double *gpartial,
                                                           double *total, int mode, float *o0, float *o1,
                                                           float *running_mean, float *running_var, float eps,
                                                           float momentum, int src_rows) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ double red[4][kStatThreads];
  __shared__ int flag;
  const int tid = threadIdx.x;
  if constexpr (std::is_same<T, double>::value) {
    const int nb = gridDim.x, V2 = 2 * C;
    const int per_p = (src_rows + nb - 1) / nb;
    const int p0 = min(src_rows, (int)blockIdx.x * per_p), p1 = min(src_rows, p0 + per_p);
    stat_reduce_rows((const double *)x + (size_t)p0 * V2, p1 - p0, V2, partial + (size_t)blockIdx.x * V2, &red[0][0]);
  } else {
  const int C4 = C >> 2;                                    // float4 groups per row (C % 4 == 0)
  const int LPR = C4 < kStatThreads ? C4 : kStatThreads;    // threads covering one row
  const int RL = kStatThreads / LPR;                        // rows handled concurrently (LPR | 1024)
  const int rl = tid / LPR, cl = tid - rl * LPR;
  const int nblk = gridDim.x, V = 2 * C;
  const int per = (rows + nblk - 1) / nblk;
  const int r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  for (int g4 = cl; g4 < C4; g4 += LPR) {
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    int r = r0 + rl;
    for (; r + 3 * RL < r1; r += 4 * RL) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = load4<T>(x + (size_t)(r + u * RL) * C + g4 * 4);
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const double d = (double)v[u][j];
          s[j] += d;
          ss[j] += d * d;
        }
    }
    if (r < r1) {   // at most 3 passes left: their loads go out together too (absent rows add exact zeros)
      f32x4 v[3];
#pragma unroll
      for (int u = 0; u < 3; u++) {
        const int rr = r + u * RL;
        v[u] = rr < r1 ? load4<T>(x + (size_t)rr * C + g4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 3; u++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const double d = (double)v[u][j];
          s[j] += d;
          ss[j] += d * d;
        }
    }
    // rows of one channel group live in lanes cl, cl+LPR, ... of every wave: xor-butterfly inside the wave
    // (both partners compute the same sum), then one LDS entry per wave (or per row lane when LPR >= 64)
    if (LPR < 64)
      for (int d = LPR; d < 64; d <<= 1)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          s[j] += __shfl_xor(s[j], d, 64);
          ss[j] += __shfl_xor(ss[j], d, 64);
        }
    const int NE = LPR < 64 ? kStatThreads / 64 : RL;
    const int e = LPR < 64 ? (tid >> 6) : rl;
    const bool holder = LPR < 64 ? (tid & 63) < LPR : true;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      double *vals = half ? ss : s;
      if (holder)
#pragma unroll
        for (int j = 0; j < 4; j++) red[j][e * LPR + cl] = vals[j];
      __syncthreads();
      if (e == 0 && holder)
        for (int q = 1; q < NE; q++)
#pragma unroll
          for (int j = 0; j < 4; j++) vals[j] += red[j][q * LPR + cl];
      __syncthreads();
    }
    if (e == 0 && holder) {
      double *pp = partial + (size_t)blockIdx.x * V;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        pp[g4 * 4 + j] = s[j];
        pp[C + g4 * 4 + j] = ss[j];
      }
    }
  }
  }
  // ---- level 1: the last slice of a group to arrive adds the group's partials ----
  const int nblk = gridDim.x, V = 2 * C;
  const int ngroups = (nblk + kStatGroup - 1) / kStatGroup;
  const int grp = blockIdx.x / kStatGroup;
  const int gsize = min(kStatGroup, nblk - grp * kStatGroup);
  // one agent-scope release / acquire per workgroup (an L2 write-back / invalidate each on a multi-XCD part):
  // the barrier orders the other threads' stores before thread 0's fence and their loads after it
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    flag = atomicAdd(&tickets[1 + grp], 1u) == (unsigned)gsize - 1;
    if (flag) __threadfence();
  }
  __syncthreads();
  if (!flag) return;
  double *lvl1 = ngroups == 1 ? total : gpartial + (size_t)grp * V;
  stat_reduce_rows(partial + (size_t)grp * kStatGroup * V, gsize, V, lvl1, &red[0][0]);
  if (ngroups > 1) {
    // ---- level 2: the last group adds the group sums ----
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      flag = atomicAdd(&tickets[0], 1u) == (unsigned)ngroups - 1;
      if (flag) __threadfence();
    }
    __syncthreads();
    if (!flag) return;
    stat_reduce_rows(gpartial, ngroups, V, total, &red[0][0]);
  }
  __syncthreads();  // total[] was written by this workgroup
  for (int c = tid; c < C; c += kStatThreads) {
    const double sum = total[c], sq = total[C + c];
    const double mean = sum / rows;
    double m2 = sq - mean * mean * rows;  // sum of squared deviations
    if (m2 < 0) m2 = 0;
    if (mode == 0) {
      o0[c] = (float)mean;
      o1[c] = (float)(m2 / (rows - 1));
    } else if (mode == 2) {
      o0[c] = (float)mean;
      o1[c] = powf((float)(m2 / (rows - 1)) + eps, -0.5f);
    } else {
      o0[c] = (float)mean;
      running_mean[c] = momentum * running_mean[c] + (1 - momentum) * (float)mean;
      running_var[c] = momentum * running_var[c] + (1 - momentum) * (float)(m2 / (rows - 1));
      o1[c] = powf((float)(m2 / rows) + eps, -0.5f);
    }
  }
  if (tid <= ngroups) tickets[tid] = 0;
}

__global__ void k_bn_eval_stats(const float *running_mean, const float *running_var, int C, float eps,
                                float *save_mean, float *save_invstd) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  save_mean[c] = running_mean[c];
  save_invstd[c] = powf(running_var[c] + eps, -0.5f);  // BatchNormalization.cpp:40-44
}

// y = leaky(fma(x, w, b)), w = invstd*gamma, b = -mean*w + beta (BatchNormalization.cpp:46-59); leakiness 0 is
// max(t, 0): the sign of a zero and NaN -> 0 differ from t * 0, nothing else.
// A thread keeps ONE group of 4 channels and walks rows (grid-stride), so w and b are formed once per thread: with one
// float4 per thread the four parameter vectors are 4x the L1 traffic of the payload (measured ~1 TB/s).
// C4 = C / 4 divides the 256 threads of a workgroup (C in 4 .. 1024, powers of two).
template <typename T>
__global__ __launch_bounds__(256) void k_bn_apply_rows(const T *__restrict__ x, T *__restrict__ y, int rows,
                                                       int C, const float *__restrict__ save_mean,
                                                       const float *__restrict__ save_invstd,
                                                       const float *__restrict__ weight,
                                                       const float *__restrict__ bias, float leakiness) {
  const int C4 = C >> 2, RPI = 256 / C4;              // rows a workgroup covers per iteration
  const int cg = threadIdx.x % C4, rl = threadIdx.x / C4;
  const int c = cg * 4;
  d3d_f32x4 w, b;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    w[j] = save_invstd[c + j] * (weight ? weight[c + j] : 1.f);
    b[j] = -save_mean[c + j] * w[j] + (bias ? bias[c + j] : 0.f);
  }
  const size_t step = (size_t)gridDim.x * RPI;
  size_t r = (size_t)blockIdx.x * RPI + rl;
  for (; r + 3 * step < (size_t)rows; r += 4 * step) {   // 4 independent rows in flight
    d3d_f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = load4<T>(x + (r + u * step) * C + c);
#pragma unroll
    for (int u = 0; u < 4; u++) store4(y + (r + u * step) * C + c, bn_act(v[u], w, b, leakiness));
  }
  for (; r < (size_t)rows; r += step) store4(y + r * C + c, bn_act(load4<T>(x + r * C + c), w, b, leakiness));
}

// any other channel count
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
    const d3d_f32x4 v = *(const d3d_f32x4 *)(x + i);
    d3d_f32x4 w, b;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      w[j] = save_invstd[c + j] * (weight ? weight[c + j] : 1.f);
      b[j] = -save_mean[c + j] * w[j] + (bias ? bias[c + j] : 0.f);
    }
    *(d3d_f32x4 *)(y + i) = bn_act(v, w, b, leakiness);   // the same expression as the convolutions' fused prologue
  } else {
    for (size_t k = i; k < i + 4 && k < total; k++) {
      const int c = (int)(k % C);
      float w = save_invstd[c] * (weight ? weight[c] : 1.f);
      float b = -save_mean[c] * w + (bias ? bias[c] : 0.f);
      const d3d_f32x4 t = bn_act(d3d_f32x4{x[k], 0.f, 0.f, 0.f}, d3d_f32x4{w, 0.f, 0.f, 0.f}, d3d_f32x4{b, 0.f, 0.f, 0.f},
                                 leakiness);
      y[k] = t[0];
    }
  }
}

static void launch_bn_apply(const float *in, float *out, int rows, int planes, const float *mean, const float *invstd,
                            const float *weight, const float *bias, float leakiness, hipStream_t s) {
  const int C4 = planes >> 2;
  if ((planes & 3) == 0 && C4 >= 1 && C4 <= 256 && 256 % C4 == 0) {
    const int rpi = 256 / C4;
    const long need = ((long)rows + rpi - 1) / rpi;
    const unsigned blocks = (unsigned)std::max<long>(1, std::min<long>(need, 256 * 8));
    hipLaunchKernelGGL(k_bn_apply_rows<float>, dim3(blocks), dim3(256), 0, s, in, out, rows, planes, mean, invstd, weight,
                       bias, leakiness);
  } else {
    const size_t total = (size_t)rows * planes;
    hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)((total / 4 + 256) / 256)), dim3(256), 0, s, in, out, total, planes,
                       mean, invstd, weight, bias, leakiness);
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

template <typename T>
static int run_stats(const T *in, int rows, int C, void *scratch, size_t scratch_bytes, hipStream_t s, int mode,
                     float *o0, float *o1, float *running_mean, float *running_var, float eps, float momentum) {
  D3D_REQUIRE(C > 0 && C <= 4096 && C % 4 == 0, "batch norm: planes=%d must be a multiple of 4, <= 4096", C);
  const int C4 = C / 4;
  D3D_REQUIRE(kStatThreads % C4 == 0, "batch norm: planes/4=%d must divide %d", C4, kStatThreads);
  D3D_REQUIRE(scratch && scratch_bytes >= d3d_bn_scratch_bytes(C), "batch norm: scratch too small");
  D3D_REQUIRE(((uintptr_t)in & 15) == 0, "batch norm: features must be 16-byte aligned");
  const int RL = kStatThreads / C4;
  int nblk = (rows + 8 * RL - 1) / (8 * RL);  // >= 8 passes of the row lanes per workgroup
  if (nblk > kStatBlocks) nblk = kStatBlocks;
  if (nblk < 1) nblk = 1;
  double *partial = (double *)((char *)scratch + kTicketBytes);
  double *gpartial = partial + (size_t)kStatBlocks * 2 * C;
  double *total = gpartial + (size_t)kStatMaxGroups * 2 * C;
  hipLaunchKernelGGL(k_bn_stats<T>, dim3(nblk), dim3(kStatThreads), 0, s, in, rows, C, (unsigned int *)scratch, partial,
                     gpartial, total, mode, o0, o1, running_mean, running_var, eps, momentum, rows);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

// statistics from the column sums a convolution left per row block (src[src_rows][2 C], fp64)
static int run_stats_partials(const double *src, int src_rows, int rows, int C, void *scratch, size_t scratch_bytes,
                              hipStream_t s, int mode, float *mean, float *invstd, float eps) {
  D3D_REQUIRE(C > 0 && C <= 4096 && C % 4 == 0, "batch norm: planes=%d must be a multiple of 4, <= 4096", C);
  D3D_REQUIRE(scratch && scratch_bytes >= d3d_bn_scratch_bytes(C), "batch norm: scratch too small");
  const int V = 2 * C, VP = V < kStatThreads ? V : kStatThreads, SL = kStatThreads / VP;
  // >= 8 passes of the row lanes per workgroup and at most 64 slices (four first-level groups): the vectors are
  // 0.1-12 MB per launch.  Measured on the 500 k-point building (30 finishes): 16 slices 4.78 ms per building, 64
  // slices 4.75, 128 slices 4.76 (D3D_BN_SLICES).
  int nblk = (src_rows + 8 * SL - 1) / (8 * SL);
  static const int max_slices = [] {
    const char *e = getenv("D3D_BN_SLICES");
    return e ? atoi(e) : 64;
  }();
  nblk = std::max(1, std::min(nblk, std::min(max_slices, kStatBlocks)));
  double *partial = (double *)((char *)scratch + kTicketBytes);
  double *gpartial = partial + (size_t)kStatBlocks * 2 * C;
  double *total = gpartial + (size_t)kStatMaxGroups * 2 * C;
  hipLaunchKernelGGL(k_bn_stats<double>, dim3(nblk), dim3(kStatThreads), 0, s, src, rows, C, (unsigned int *)scratch,
                     partial, gpartial, total, mode, mean, invstd, nullptr, nullptr, eps, 0.f, src_rows);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

// fp32 rows [rows, cin] -> bf16 rows [rows, width] (width >= cin a multiple of 8; the channels past cin are zero): the
// input layer's per-voxel means as the bf16 backbone stores them.  One thread per 8 output channels (one 16-byte store).
__global__ __launch_bounds__(256) void k_rows_to_bf16(const float *__restrict__ in, long rows, int cin, int width,
                                                      unsigned short *__restrict__ out) {
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  const int g8 = width / 8;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * g8) return;
  const long row = t / g8;
  const int c0 = (int)(t - row * g8) * 8;
  const float *p = in + row * cin;
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; j++) b[j] = (__bf16)(c0 + j < cin ? p[c0 + j] : 0.f);   // round to nearest even, as torch's .to()
  *(uint4 *)(out + row * width + c0) = __builtin_bit_cast(uint4, b);
}

}  // namespace d3d

using namespace d3d;

extern "C" {

size_t d3d_bn_scratch_bytes(int planes) {
  return kTicketBytes + (size_t)(kStatBlocks + kStatMaxGroups + 1) * 2 * planes * sizeof(double);
}

int d3d_bn_batch_stats(const float *in, int rows, int planes, float *mean, float *var_unbiased,
                       void *scratch, size_t scratch_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(in && mean && var_unbiased && rows > 0, "bn_batch_stats: bad arguments");
  return run_stats(in, rows, planes, scratch, scratch_bytes, s, 0, mean, var_unbiased, nullptr, nullptr, 0.f, 0.f);
}

int d3d_bn_batch_invstd(const float *in, int rows, int planes, float eps, float *mean, float *invstd,
                        void *scratch, size_t scratch_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(in && mean && invstd && rows > 0, "bn_batch_invstd: bad arguments");
  return run_stats(in, rows, planes, scratch, scratch_bytes, s, 2, mean, invstd, nullptr, nullptr, eps, 0.f);
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
    int rc = run_stats(in, rows, planes, scratch, scratch_bytes, s, 1, save_mean, save_invstd, running_mean, running_var, eps, momentum);
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(k_bn_eval_stats, dim3((planes + 63) / 64), dim3(64), 0, s, running_mean, running_var, planes, eps, save_mean, save_invstd);
  }
  launch_bn_apply(in, out, rows, planes, save_mean, save_invstd, weight, bias, leakiness, s);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_bn_apply(const float *in, float *out, int rows, int planes, const float *mean, const float *invstd,
                 const float *weight, const float *bias, float leakiness, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (rows == 0) return D3D_OK;
  D3D_REQUIRE(in && out && mean && invstd && planes > 0 && rows > 0, "bn_apply: bad arguments");
  launch_bn_apply(in, out, rows, planes, mean, invstd, weight, bias, leakiness, s);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_bn_stats_from_partials(const double *partials, int partial_rows, int rows, int planes, float eps,
                               int want_invstd, float *mean, float *var_or_invstd, void *scratch, size_t scratch_bytes,
                               void *stream) {
  D3D_REQUIRE(partials && mean && var_or_invstd && partial_rows > 0 && rows > 0, "bn_stats_from_partials: bad arguments");
  return run_stats_partials(partials, partial_rows, rows, planes, scratch, scratch_bytes, (hipStream_t)stream,
                            want_invstd ? 2 : 0, mean, var_or_invstd, eps);
}

/* storage-type aware forms (d3d_dtype) of the two inference-side BatchNorm entry points */
int d3d_bn_batch_invstd_dt(const void *in, int rows, int planes, float eps, float *mean, float *invstd, void *scratch,
                           size_t scratch_bytes, int dtype, void *stream) {
  if (dtype == D3D_F32)
    return d3d_bn_batch_invstd((const float *)in, rows, planes, eps, mean, invstd, scratch, scratch_bytes, stream);
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(dtype == D3D_BF16 && in && mean && invstd && rows > 0, "bn_batch_invstd_dt: bad arguments");
  return run_stats((const unsigned short *)in, rows, planes, scratch, scratch_bytes, s, 2, mean, invstd, nullptr, nullptr,
                   eps, 0.f);
}

int d3d_bn_apply_dt(const void *in, void *out, int rows, int planes, const float *mean, const float *invstd,
                    const float *weight, const float *bias, float leakiness, int dtype, void *stream) {
  if (dtype == D3D_F32)
    return d3d_bn_apply((const float *)in, (float *)out, rows, planes, mean, invstd, weight, bias, leakiness, stream);
  hipStream_t s = (hipStream_t)stream;
  if (rows == 0) return D3D_OK;
  const int C4 = planes >> 2;
  D3D_REQUIRE(dtype == D3D_BF16 && in && out && mean && invstd && (planes & 3) == 0 && C4 >= 1 && C4 <= 256 &&
              256 % C4 == 0, "bn_apply_dt: bad arguments (planes=%d)", planes);
  const int rpi = 256 / C4;
  const long need = ((long)rows + rpi - 1) / rpi;
  const unsigned blocks = (unsigned)std::max<long>(1, std::min<long>(need, 256 * 8));
  hipLaunchKernelGGL(k_bn_apply_rows<unsigned short>, dim3(blocks), dim3(256), 0, s, (const unsigned short *)in,
                     (unsigned short *)out, rows, planes, mean, invstd, weight, bias, leakiness);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_rows_to_bf16(const float *in, long rows, int cin, int width, void *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(rows >= 0 && cin > 0 && width >= cin && width % 8 == 0, "rows_to_bf16: bad arguments (cin=%d, width=%d)", cin, width);
  if (rows == 0) return D3D_OK;
  D3D_REQUIRE(in && out && ((uintptr_t)out & 15) == 0, "rows_to_bf16: null or unaligned pointer");
  const long total = rows * (width / 8);
  hipLaunchKernelGGL(k_rows_to_bf16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, rows, cin, width,
                     (unsigned short *)out);
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
