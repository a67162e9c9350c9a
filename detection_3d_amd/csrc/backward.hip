// a7. Backward of the sparse ops (training rows): convolution dInput / dWeight, BatchNorm,
// input layer, sparse RoIAlign.  Reference: SCN/CUDA/Convolution.cu:249-442 (fused dI + dW with
// atomicAdd on dW), SCN/CPU/BatchNormalization.cpp:62-107, SCN/CPU/IOLayers.cpp:30-47,
// maskrcnn_benchmark/csrc/cuda/ROIAlignRotated3D_cuda.cu:182-354.
//
// dInput needs no new kernel: it is the forward gather-GEMM on the transposed rulebook with W^T
//   submanifold: nbr[i][k] = j  <=>  nbr[j][K-1-k] = i  -> same plan, offsets flipped;
//   strided conv: the deconvolution plan;  deconvolution: the convolution plan.
// dWeight[k] = sum over rules of offset k of in[r_in]^T (x) dOut[r_out]: MFMA with the rule index
// as the contraction dimension, partial tiles added with fp32 atomics (as the reference does).
#include <algorithm>

#include "d3d_internal.h"

namespace d3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// packed_t[k'][g][ci][j] = w[k][ci][4g+j]  (a conv weight with Cin' = cout, Cout' = cin), k' = flip ? K-1-k : k
__global__ void k_pack_weight_t(const float *__restrict__ w, int fv, int cin, int cout, int cp, int flip,
                                float *__restrict__ packed) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)fv * cp * cin;
  if (t >= total) return;
  int j = (int)(t & 3);
  long u = t >> 2;
  int ci = (int)(u % cin);
  u /= cin;
  int g = (int)(u % (cp / 4));
  int kp = (int)(u / (cp / 4));
  int k = flip ? fv - 1 - kp : kp;
  int co = 4 * g + j;
  packed[t] = co < cout ? w[((size_t)k * cin + ci) * cout + co] : 0.f;
}

static constexpr int kDwBlocksPerWg = 64;
static constexpr int kDwTargetWgs = 1024;
static constexpr int kDwMaxChunk = 8;

// One workgroup: offset k = blockIdx.y, a run of kDwBlocksPerWg row blocks, a group of <= 16 output tiles (32x32).
// The blocks of the run that have offset k are found with one ballot; their operands (32 gathered input rows and
// the 32 dOut rows) go through registers one block ahead and their row indices two blocks ahead, all loads
// branch-free (an absent row reads row 0 and is zeroed on the way to LDS), so the gathers of the next block fly
// under the MFMAs of the current one -- the same pipeline as the forward kernel.
// DET (d3d_conv_dw_deterministic): no atomics.  Workgroup blockIdx.x walks the runs blockIdx.x, + gridDim.x, ... in order
// (all of a run's active blocks, no chunks), keeps the sum in its accumulators and STORES it as partial `blockIdx.x` of
// dW (zeros when it met no rule); k_dw_reduce adds the partials to dW in a fixed order.
template <int CP, int COUT, bool DET>
__global__ __launch_bounds__(256) void k_conv_dw(const float *__restrict__ in, int cin,
                                                 const float *__restrict__ d_out,
                                                 const int32_t *__restrict__ nbrT, int npos,
                                                 const int32_t *__restrict__ rows,
                                                 const uint32_t *__restrict__ blkmask, int n_blk, int run,
                                                 int chunk, int nz_tiles, float *__restrict__ dW) {
  constexpr int NTI = CP / 32, NTJ = COUT / 32, T = NTI * NTJ;
  constexpr int TPG = T < 16 ? T : 16;             // tiles per group (grid.z)
  constexpr int TPW = (TPG + 3) / 4;               // tiles per wave
  constexpr int A4 = CP / 4, B4 = COUT / 4;        // float4 per row
  constexpr int NA = (32 * A4 + 255) / 256, NB = (32 * B4 + 255) / 256;  // float4 per thread and block
  static_assert(kDwBlocksPerWg == 64, "one ballot covers the longest run");
  __shared__ __attribute__((aligned(16))) float As[32 * CP];
  __shared__ __attribute__((aligned(16))) float Bs[32 * COUT];
  const int k = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 31, kk = lane >> 5;
  int b0 = blockIdx.x * run;
  const int n_runs = (n_blk + run - 1) / run;
  int rx = blockIdx.x;            // the run being walked (DET: then rx + gridDim.x, ...)
  // blocks of a run (<= 64) that have a rule at offset k
  auto run_mask = [&](int first) -> unsigned long long {
    const uint32_t mm = (lane < run && first + lane < n_blk) ? blkmask[first + lane] : 0u;
    return __ballot((mm >> k) & 1u);
  };
  unsigned long long active = run_mask(b0);
  // The run's active blocks are dealt out in chunks of `chunk` to the workgroups blockIdx.z / nz_tiles = 0, 1, ...:
  // a dense offset (the centre one is present in every block) no longer makes one workgroup walk the whole run
  // while the workgroups of the sparse offsets have long finished.
  if constexpr (!DET) {
    const int c = blockIdx.z / nz_tiles;
    for (int d = 0; d < c * chunk && active; d++) active &= active - 1;
    unsigned long long keep = 0, rest = active;
    for (int d = 0; d < chunk && rest; d++) {
      keep |= rest & (~rest + 1);
      rest &= rest - 1;
    }
    active = keep;
  }
  if constexpr (!DET) {
    if (!active) return;
  }
  const int zt = blockIdx.z % nz_tiles;   // tile group
  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
  const bool vec = (cin == CP);

  int ia[NA], ib[NB];           // row indices of the block whose operands are requested next
  f32x4 sa[NA], sb[NB];         // operands in flight
  uint32_t real_a = 0, real_b = 0;
  auto load_idx = [&](int b) {
#pragma unroll
    for (int j = 0; j < NA; j++) {
      const int e = threadIdx.x + j * 256;
      ia[j] = nbrT[(size_t)k * npos + b * 32 + (e < 32 * A4 ? e / A4 : 0)];
    }
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const int e = threadIdx.x + j * 256;
      ib[j] = rows[b * 32 + (e < 32 * B4 ? e / B4 : 0)];
    }
  };
  auto issue_data = [&]() {
    real_a = real_b = 0;
#pragma unroll
    for (int j = 0; j < NA; j++) {
      const int e = threadIdx.x + j * 256;
      const int c4 = e % A4;
      const int s = e < 32 * A4 ? ia[j] : -1;
      if (s >= 0) real_a |= 1u << j;
      const float *p = in + (size_t)(s < 0 ? 0 : s) * cin + c4 * 4;
      if (vec) {
        sa[j] = *(const f32x4 *)p;
      } else {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c4 * 4 + 0 < cin) v[0] = p[0];
        if (c4 * 4 + 1 < cin) v[1] = p[1];
        if (c4 * 4 + 2 < cin) v[2] = p[2];
        if (c4 * 4 + 3 < cin) v[3] = p[3];
        sa[j] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const int e = threadIdx.x + j * 256;
      const int c4 = e % B4;
      const int o = e < 32 * B4 ? ib[j] : -1;
      if (o >= 0) real_b |= 1u << j;
      sb[j] = *(const f32x4 *)(d_out + (size_t)(o < 0 ? 0 : o) * COUT + c4 * 4);
    }
  };
  auto commit = [&]() {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NA; j++) {
      const int e = threadIdx.x + j * 256;
      if (e < 32 * A4) *(f32x4 *)(As + (e / A4) * CP + (e % A4) * 4) = ((real_a >> j) & 1u) ? sa[j] : zero;
    }
#pragma unroll
    for (int j = 0; j < NB; j++) {
      const int e = threadIdx.x + j * 256;
      if (e < 32 * B4) *(f32x4 *)(Bs + (e / B4) * COUT + (e % B4) * 4) = ((real_b >> j) & 1u) ? sb[j] : zero;
    }
  };
  auto pop = [&]() -> int {  // next active block of the run (DET: of this workgroup's runs), or -1
    if constexpr (DET) {
      while (!active) {        // (uniform over the workgroup: every wave walks the same runs)
        rx += gridDim.x;
        if (rx >= n_runs) return -1;
        b0 = rx * run;
        active = run_mask(b0);
      }
    } else {
      if (!active) return -1;
    }
    const int j = __builtin_ctzll(active);
    active &= active - 1;
    return b0 + j;
  };

  int bc = pop();               // block being computed
  if (bc >= 0) {
    load_idx(bc);
    issue_data();
  }
  int bn = pop();               // block whose operands are requested during bc's MFMAs
  if (bn >= 0) load_idx(bn);
  while (bc >= 0) {
    commit();
    __syncthreads();
    const int bnn = bn >= 0 ? pop() : -1;
    if (bn >= 0) {
      issue_data();               // operands of bn (its indices arrived during the previous block)
      if (bnn >= 0) load_idx(bnn);
    }
#pragma unroll
    for (int t = 0; t < TPW; t++) {
      const int tile = zt * TPG + wave * TPW + t;
      if (wave * TPW + t >= TPG || tile >= T) continue;
      const int ti = tile / NTJ, tj = tile % NTJ;
#pragma unroll
      for (int s = 0; s < 16; s++) {
        const float a = As[(2 * s + kk) * CP + ti * 32 + i];
        const float bb = Bs[(2 * s + kk) * COUT + tj * 32 + i];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
    bc = bn;
    bn = bnn;
  }
#pragma unroll
  for (int t = 0; t < TPW; t++) {
    const int tile = zt * TPG + wave * TPW + t;
    if (wave * TPW + t >= TPG || tile >= T) continue;
    const int ti = tile / NTJ, tj = tile % NTJ;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int ci = ti * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
      const int co = tj * 32 + i;
      if constexpr (DET) {
        if (ci < cin) dW[(((size_t)blockIdx.x * gridDim.y + k) * cin + ci) * COUT + co] = acc[t][reg];
      } else {
        if (ci < cin) atomicAdd(dW + ((size_t)k * cin + ci) * COUT + co, acc[t][reg]);
      }
    }
  }
}

// dW[e] += partial[0][e] + partial[1][e] + ...  (in that order)
__global__ __launch_bounds__(256) void k_dw_reduce(const float *__restrict__ part, int n_part, size_t n, float *__restrict__ dW) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float v = 0.f;
  for (int g = 0; g < n_part; g++) v += part[(size_t)g * n + e];
  dW[e] += v;
}

static bool g_dw_deterministic = [] {
  const char *e = getenv("D3D_DW_DETERMINISTIC");
  return e && e[0] != '0';
}();
static constexpr int kDwDetPartials = 32;   // workgroups (and partial sums) per offset and tile group in deterministic mode

template <int CP, int COUT>
static int launch_dw_t(d3d_meta *m, const Plan &p, const float *in, int cin, const float *d_out, float *dW, hipStream_t s) {
  constexpr int T = (CP / 32) * (COUT / 32);
  constexpr int TPG = T < 16 ? T : 16;
  const int nz = (T + TPG - 1) / TPG;
  if (g_dw_deterministic) {
    // fixed summation order: run r belongs to workgroup r % G, which adds its runs' blocks in order; the G partial
    // sums are added in order by k_dw_reduce.  Scratch: G x (K, cin, COUT) floats of the feature lane.
    D3D_REQUIRE(m, "deterministic conv backward needs the metadata (its scratch lane)");
    const int run = kDwBlocksPerWg;
    const int n_runs = (p.n_blk + run - 1) / run;
    const int G = std::min(n_runs, kDwDetPartials);
    const size_t n = (size_t)p.K * cin * COUT;
    const size_t mark = m->feat_arena.used;
    float *part = m->feat_arena.get<float>((size_t)G * n);
    if (!part) {
      set_error("deterministic conv backward: %zu bytes of scratch do not fit the feature lane", (size_t)G * n * 4);
      return D3D_ERR_NOMEM;
    }
    hipLaunchKernelGGL((k_conv_dw<CP, COUT, true>), dim3(G, p.K, nz), dim3(256), 0, s, in, cin, d_out, p.nbrT, p.n_blk * 32,
                       p.rows, p.blkmask, p.n_blk, run, run, nz, part);
    hipLaunchKernelGGL(k_dw_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, G, n, dW);
    m->feat_arena.used = mark;   // stream-ordered scratch
    D3D_LAUNCH_CHECK();
    return D3D_OK;
  }
  // run length: long runs amortise the final atomics of a workgroup (T * 1024 of them), short runs keep a small
  // plan from being walked serially by a handful of workgroups -- aim at >= kDwTargetWgs workgroups.  Plans so large
  // that the run hits the 64 blocks one ballot covers deal the run's active blocks out in chunks of kDwMaxChunk
  // instead: the dense offsets (the centre one is in every block) would otherwise set the kernel's duration
  // (measured: Cin = Cout = 64 at 370 k rows 236 -> 186 us, 32 at 460 k rows 124 -> 90 us; smaller plans lose).
  long run = ((long)p.n_blk * p.K * nz + kDwTargetWgs - 1) / kDwTargetWgs;
  long chunk;
  if (run >= kDwBlocksPerWg) {
    run = kDwBlocksPerWg;
    chunk = kDwMaxChunk;
  } else {
    run = std::max<long>(2, run);
    chunk = run;
  }
  const int n_chunks = (int)((run + chunk - 1) / chunk);
  dim3 grid((unsigned)((p.n_blk + run - 1) / run), p.K, nz * n_chunks);
  hipLaunchKernelGGL((k_conv_dw<CP, COUT, false>), grid, dim3(256), 0, s, in, cin, d_out, p.nbrT, p.n_blk * 32, p.rows,
                     p.blkmask, p.n_blk, (int)run, (int)chunk, nz, dW);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}
template <int CP>
static int launch_dw_c(d3d_meta *m, const Plan &p, const float *in, int cin, const float *d_out, int cout, float *dW,
                       hipStream_t s) {
  switch (cout) {
    case 32: return launch_dw_t<CP, 32>(m, p, in, cin, d_out, dW, s);
    case 64: return launch_dw_t<CP, 64>(m, p, in, cin, d_out, dW, s);
    case 128: return launch_dw_t<CP, 128>(m, p, in, cin, d_out, dW, s);
    case 256: return launch_dw_t<CP, 256>(m, p, in, cin, d_out, dW, s);
  }
  set_error("conv backward: Cout=%d not supported", cout);
  return D3D_ERR_UNSUPPORTED;
}
// dW[K, cin, cout] += ...   (dW pre-zeroed by the caller, like the reference's d_weight)
static int launch_dw(d3d_meta *m, const Plan &p, const float *in, int cin, const float *d_out, int cout, float *dW,
                     hipStream_t s) {
  if (p.n_rows == 0) return D3D_OK;
  D3D_REQUIRE(in && d_out && dW, "conv backward: null pointer");
  if (cin <= 32) return launch_dw_c<32>(m, p, in, cin, d_out, cout, dW, s);
  if (cin <= 64) return launch_dw_c<64>(m, p, in, cin, d_out, cout, dW, s);
  if (cin <= 128) return launch_dw_c<128>(m, p, in, cin, d_out, cout, dW, s);
  if (cin <= 256) return launch_dw_c<256>(m, p, in, cin, d_out, cout, dW, s);
  set_error("conv backward: Cin=%d not supported", cin);
  return D3D_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------
// BatchNorm backward (CPU/BatchNormalization.cpp:62-107): partial sums of d' = dOut * relu' and
// (x - mean) d', fixed-order reduction, then the elementwise pass.
static constexpr int kBnBwdBlocks = 128;
__global__ __launch_bounds__(256) void k_bn_bwd_partial(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ dy, int rows, int C,
                                                        const float *__restrict__ mean, float leak,
                                                        double *__restrict__ partial) {
  extern __shared__ double red[];
  const int tid = threadIdx.x;
  const int lpr = C < 256 ? C : 256, row_lanes = 256 / lpr;
  const int rl = tid / lpr, cl = tid % lpr;
  const int per = (rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  for (int c = cl; c < C; c += lpr) {
    double s = 0, dp = 0;
    const float mu = mean[c];
    for (int r = r0 + rl; r < r1; r += row_lanes) {
      const size_t i = (size_t)r * C + c;
      const float d = dy[i] * ((y[i] > 0) ? 1.f : leak);
      s += (double)d;
      dp += (double)((x[i] - mu) * d);
    }
    red[tid * 2] = s;
    red[tid * 2 + 1] = dp;
    __syncthreads();
    if (rl == 0) {
      for (int j = 1; j < row_lanes; j++) {
        s += red[(j * lpr + cl) * 2];
        dp += red[(j * lpr + cl) * 2 + 1];
      }
      partial[(size_t)blockIdx.x * 2 * C + c] = s;
      partial[(size_t)blockIdx.x * 2 * C + C + c] = dp;
    }
    __syncthreads();
  }
}
// Same sums with 16-byte loads: threads = float4 channel groups x concurrent rows (1024 per workgroup), butterfly
// inside the wave, one LDS entry per wave -- the layout of k_bn_stats (bn.hip).  planes % 4 == 0, planes/4 | 1024.
__global__ __launch_bounds__(1024) void k_bn_bwd_partial4(const float *__restrict__ x, const float *__restrict__ y,
                                                          const float *__restrict__ dy, int rows, int C,
                                                          const float *__restrict__ mean, float leak,
                                                          double *__restrict__ partial) {
  __shared__ double red[4][1024];
  const int tid = threadIdx.x;
  const int C4 = C >> 2;
  const int LPR = C4 < 1024 ? C4 : 1024;
  const int RL = 1024 / LPR;
  const int rl = tid / LPR, cl = tid - rl * LPR;
  const int per = (rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  for (int g4 = cl; g4 < C4; g4 += LPR) {
    const f32x4 mu = *(const f32x4 *)(mean + g4 * 4);
    double s[4] = {0, 0, 0, 0}, dp[4] = {0, 0, 0, 0};
    for (int r = r0 + rl; r < r1; r += RL) {
      const size_t i = (size_t)r * C + g4 * 4;
      const f32x4 vx = *(const f32x4 *)(x + i), vy = *(const f32x4 *)(y + i), vd = *(const f32x4 *)(dy + i);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float d = vd[j] * ((vy[j] > 0) ? 1.f : leak);
        s[j] += (double)d;
        dp[j] += (double)((vx[j] - mu[j]) * d);
      }
    }
    if (LPR < 64)
      for (int d = LPR; d < 64; d <<= 1)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          s[j] += __shfl_xor(s[j], d, 64);
          dp[j] += __shfl_xor(dp[j], d, 64);
        }
    const int NE = LPR < 64 ? 16 : RL;
    const int e = LPR < 64 ? (tid >> 6) : rl;
    const bool holder = LPR < 64 ? (tid & 63) < LPR : true;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      double *vals = half ? dp : s;
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
      double *pp = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        pp[g4 * 4 + j] = s[j];
        pp[C + g4 * 4 + j] = dp[j];
      }
    }
  }
}
// dx for 4 channels per thread
__global__ __launch_bounds__(256) void k_bn_bwd_apply4(const float *__restrict__ x, const float *__restrict__ y,
                                                       const float *__restrict__ dy, float *__restrict__ dx,
                                                       size_t total4, int C, const float *__restrict__ mean,
                                                       const float *__restrict__ invstd,
                                                       const float *__restrict__ weight,
                                                       const float *__restrict__ grad_mean,
                                                       const float *__restrict__ kcoef, float leak) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total4) return;
  const size_t i = t * 4;
  const int c = (int)(i % C);
  const f32x4 vx = *(const f32x4 *)(x + i), vy = *(const f32x4 *)(y + i), vd = *(const f32x4 *)(dy + i);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float d = vd[j] * ((vy[j] > 0) ? 1.f : leak);
    o[j] = (d - grad_mean[c + j] - (vx[j] - mean[c + j]) * kcoef[c + j]) * invstd[c + j] * (weight ? weight[c + j] : 1.f);
  }
  *(f32x4 *)(dx + i) = o;
}
// the same arithmetic with one 4-channel group per thread walking rows (parameters read once per thread, see
// k_bn_apply_rows in bn.hip); C4 = C / 4 divides 256
__global__ __launch_bounds__(256) void k_bn_bwd_apply_rows(const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ dy, float *__restrict__ dx,
                                                           int rows, int C, const float *__restrict__ mean,
                                                           const float *__restrict__ invstd,
                                                           const float *__restrict__ weight,
                                                           const float *__restrict__ grad_mean,
                                                           const float *__restrict__ kcoef, float leak) {
  const int C4 = C >> 2, RPI = 256 / C4;
  const int cg = threadIdx.x % C4, rl = threadIdx.x / C4;
  const int c = cg * 4;
  f32x4 gm, mu, kc, is, wt;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    gm[j] = grad_mean[c + j];
    mu[j] = mean[c + j];
    kc[j] = kcoef[c + j];
    is[j] = invstd[c + j];
    wt[j] = weight ? weight[c + j] : 1.f;
  }
  const size_t step = (size_t)gridDim.x * RPI;
  auto one = [&](f32x4 vx, f32x4 vy, f32x4 vd) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float d = vd[j] * ((vy[j] > 0) ? 1.f : leak);
      o[j] = (d - gm[j] - (vx[j] - mu[j]) * kc[j]) * is[j] * wt[j];
    }
    return o;
  };
  size_t r = (size_t)blockIdx.x * RPI + rl;
  for (; r + step < (size_t)rows; r += 2 * step) {   // 2 rows = 6 loads in flight
    const size_t i0 = r * C + c, i1 = (r + step) * C + c;
    const f32x4 x0 = *(const f32x4 *)(x + i0), y0 = *(const f32x4 *)(y + i0), d0 = *(const f32x4 *)(dy + i0);
    const f32x4 x1 = *(const f32x4 *)(x + i1), y1 = *(const f32x4 *)(y + i1), d1 = *(const f32x4 *)(dy + i1);
    *(f32x4 *)(dx + i0) = one(x0, y0, d0);
    *(f32x4 *)(dx + i1) = one(x1, y1, d1);
  }
  for (; r < (size_t)rows; r += step) {
    const size_t i0 = r * C + c;
    *(f32x4 *)(dx + i0) = one(*(const f32x4 *)(x + i0), *(const f32x4 *)(y + i0), *(const f32x4 *)(dy + i0));
  }
}
__global__ __launch_bounds__(256) void k_bn_bwd_finish(const double *__restrict__ partial, int nblk, int rows,
                                                       int C, const float *__restrict__ invstd,
                                                       float *grad_mean, float *kcoef, float *d_weight,
                                                       float *d_bias) {
  __shared__ double red[256][2];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0, dp = 0;
  if (c < C)
    for (int b = sl; b < nblk; b += 8) {
      s += partial[(size_t)b * 2 * C + c];
      dp += partial[(size_t)b * 2 * C + C + c];
    }
  red[threadIdx.x][0] = s;
  red[threadIdx.x][1] = dp;
  __syncthreads();
  if (sl != 0 || c >= C) return;
  for (int j = 1; j < 8; j++) {
    s += red[j * 32 + cl][0];
    dp += red[j * 32 + cl][1];
  }
  const float is = invstd[c];
  if (d_bias) d_bias[c] = (float)s;
  if (d_weight) d_weight[c] = (float)dp * is;
  grad_mean[c] = (float)(s / rows);
  kcoef[c] = (float)dp * is * is / rows;
}
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float *__restrict__ x, const float *__restrict__ y,
                                                      const float *__restrict__ dy, float *__restrict__ dx,
                                                      size_t total, int C, const float *__restrict__ mean,
                                                      const float *__restrict__ invstd,
                                                      const float *__restrict__ weight,
                                                      const float *__restrict__ grad_mean,
                                                      const float *__restrict__ kcoef, float leak) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const float d = dy[i] * ((y[i] > 0) ? 1.f : leak);
  dx[i] = (d - grad_mean[c] - (x[i] - mean[c]) * kcoef[c]) * invstd[c] * (weight ? weight[c] : 1.f);
}

// Input layer backward (CPU/IOLayers.cpp:30-47): every point receives multiplier * d_out[site].
__global__ void k_input_backward(const float *__restrict__ d_out, int planes, const int32_t *__restrict__ off,
                                 const int32_t *__restrict__ idx, int n_active, int average,
                                 float *__restrict__ d_in) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n_active * planes) return;
  const int row = (int)(t / planes), c = (int)(t % planes);
  const int b = off[row], e = off[row + 1];
  const float mult = (average && e > b) ? (float)1 / (e - b) : (float)1;
  const float g = mult * d_out[t];
  for (int j = b; j < e; j++) d_in[(size_t)idx[j] * planes + c] = g;
}

}  // namespace d3d

using namespace d3d;

extern "C" {

int d3d_pack_conv_weight_transposed(const float *w, int fv, int cin, int cout, int flip, float *packed,
                                    void *stream) {
  hipStream_t s = (hipStream_t)stream;
  size_t n = d3d_packed_weight_floats(fv, cout, cin);
  D3D_REQUIRE(w && packed && n > 0, "pack_conv_weight_transposed: bad arguments (Cin=%d Cout=%d)", cin, cout);
  int cp = (int)(n / ((size_t)fv * cin));
  long total = (long)n;
  hipLaunchKernelGGL(k_pack_weight_t, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, fv, cin, cout, cp,
                     flip, packed);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

// d_in [rows_in, cin] (overwritten), d_weight [fv, cin, cout] (accumulated into; pre-zero it)
int d3d_subm_conv_backward(d3d_meta *m, const int *size, const int *filt, const float *in, int cin,
                           const float *packed_wt_flipped, int cout, const float *d_out, float *d_in,
                           float *d_weight, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && size && filt, "null argument");
  for (int d = 0; d < 3; d++) D3D_REQUIRE(filt[d] % 2 == 1, "submanifold backward needs odd filter sizes");
  int rc = d3d_subm_prepare(m, size, filt, stream, nullptr);
  if (rc) return rc;
  const Plan *p = find_plan(m, 0, size, filt, nullptr);
  if (d_in) {
    rc = launch_conv(m, *p, d_out, cout, packed_wt_flipped, cin, nullptr, d_in, s);
    if (rc) return rc;
  }
  if (d_weight) return launch_dw(m, *p, in, cin, d_out, cout, d_weight, s);
  return D3D_OK;
}

int d3d_conv_backward(d3d_meta *m, const int *in_size, const int *out_size, const int *filt,
                      const int *stride, const float *in, int cin, const float *packed_wt, int cout,
                      const float *d_out, float *d_in, float *d_weight, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && in_size && out_size && filt && stride, "null argument");
  const Plan *fwd = find_plan(m, 1, in_size, filt, stride);
  if (!fwd) {
    set_error("conv backward: forward rulebook not built");
    return D3D_ERR_STATE;
  }
  if (d_in) {
    const Plan *dec = nullptr;
    int rc = get_deconv_plan(m, in_size, filt, stride, s, &dec);
    if (rc) return rc;
    rc = launch_conv(m, *dec, d_out, cout, packed_wt, cin, nullptr, d_in, s);
    if (rc) return rc;
  }
  if (d_weight) return launch_dw(m, *fwd, in, cin, d_out, cout, d_weight, s);
  return D3D_OK;
}

int d3d_deconv_backward(d3d_meta *m, const int *in_size, const int *out_size, const int *filt,
                        const int *stride, const float *in, int cin, const float *packed_wt, int cout,
                        const float *d_out, float *d_in, float *d_weight, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && in_size && out_size && filt && stride, "null argument");
  const Plan *conv = find_plan(m, 1, out_size, filt, stride);  // fine -> coarse rulebook
  if (!conv) {
    set_error("deconv backward: strided rulebook not built");
    return D3D_ERR_STATE;
  }
  if (d_in) {
    int rc = launch_conv(m, *conv, d_out, cout, packed_wt, cin, nullptr, d_in, s);
    if (rc) return rc;
  }
  if (d_weight) {
    const Plan *dec = nullptr;
    int rc = get_deconv_plan(m, out_size, filt, stride, s, &dec);
    if (rc) return rc;
    return launch_dw(m, *dec, in, cin, d_out, cout, d_weight, s);
  }
  return D3D_OK;
}

int d3d_conv_dw_deterministic(int on) {
  const int was = g_dw_deterministic ? 1 : 0;
  if (on >= 0) g_dw_deterministic = on != 0;
  return was;
}

size_t d3d_bn_backward_scratch_bytes(int planes) {
  return (size_t)kBnBwdBlocks * 2 * planes * sizeof(double) + 2 * (size_t)planes * sizeof(float) + 512;
}

int d3d_bn_backward(const float *in, const float *out, const float *d_out, float *d_in, int rows, int planes,
                    const float *save_mean, const float *save_invstd, const float *weight, float *d_weight,
                    float *d_bias, float leakiness, void *scratch, size_t scratch_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(planes > 0 && rows >= 0 && save_mean && save_invstd, "bn_backward: bad arguments");
  if (rows == 0) return D3D_OK;
  D3D_REQUIRE(in && out && d_out && d_in, "bn_backward: null features");
  D3D_REQUIRE(planes >= 256 ? (planes % 256 == 0) : (256 % planes == 0), "bn_backward: planes=%d unsupported", planes);
  D3D_REQUIRE(scratch && scratch_bytes >= d3d_bn_backward_scratch_bytes(planes), "bn_backward: scratch too small");
  int nblk = kBnBwdBlocks;
  if (rows < nblk * 64) nblk = std::max(1, (rows + 63) / 64);
  double *partial = (double *)scratch;
  float *grad_mean = (float *)((char *)scratch + (size_t)kBnBwdBlocks * 2 * planes * sizeof(double));
  float *kcoef = grad_mean + planes;
  const bool vec4 = planes % 4 == 0 && 1024 % (planes / 4) == 0 && (((uintptr_t)in | (uintptr_t)out | (uintptr_t)d_out | (uintptr_t)d_in) & 15) == 0;
  if (vec4) {
    const int RL = 1024 / (planes / 4);
    nblk = std::max(1, std::min(kBnBwdBlocks, (rows + 8 * RL - 1) / (8 * RL)));
    hipLaunchKernelGGL(k_bn_bwd_partial4, dim3(nblk), dim3(1024), 0, s, in, out, d_out, rows, planes, save_mean, leakiness,
                       partial);
  } else {
    hipLaunchKernelGGL(k_bn_bwd_partial, dim3(nblk), dim3(256), 256 * 2 * sizeof(double), s, in, out, d_out, rows, planes,
                       save_mean, leakiness, partial);
  }
  hipLaunchKernelGGL(k_bn_bwd_finish, dim3((planes + 31) / 32), dim3(256), 0, s, partial, nblk, rows, planes,
                     save_invstd, grad_mean, kcoef, d_weight, d_bias);
  size_t total = (size_t)rows * planes;
  if (vec4 && planes / 4 <= 256 && 256 % (planes / 4) == 0) {
    const int rpi = 256 / (planes / 4);
    const long need = ((long)rows + rpi - 1) / rpi;
    const unsigned blocks = (unsigned)std::max<long>(1, std::min<long>(need, 256 * 8));
    hipLaunchKernelGGL(k_bn_bwd_apply_rows, dim3(blocks), dim3(256), 0, s, in, out, d_out, d_in, rows, planes, save_mean,
                       save_invstd, weight, grad_mean, kcoef, leakiness);
  } else if (vec4)
    hipLaunchKernelGGL(k_bn_bwd_apply4, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, in, out, d_out, d_in,
                       total / 4, planes, save_mean, save_invstd, weight, grad_mean, kcoef, leakiness);
  else
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, d_out, d_in, total,
                       planes, save_mean, save_invstd, weight, grad_mean, kcoef, leakiness);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_input_layer_backward(d3d_meta *m, const float *d_out, int planes, float *d_in, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && planes > 0, "bad arguments");
  if (!m->in_off) {
    set_error("input layer backward before build");
    return D3D_ERR_STATE;
  }
  if (m->in_active == 0) return D3D_OK;
  D3D_REQUIRE(d_out && d_in, "null gradient pointer");
  if (int rc = ensure_point_lists(m, s)) return rc;
  long total = (long)m->in_active * planes;
  hipLaunchKernelGGL(k_input_backward, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_out, planes,
                     m->in_off, m->in_idx, m->in_active, m->in_mode == 4 ? 1 : 0, d_in);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

}  // extern "C"
