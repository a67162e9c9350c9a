// a6 in bf16 storage (BASELINE.json configs[4]: "bf16 sparse conv ... HBM-bound gather/scatter, MFMA inner GEMM"; dtype
// dispatch of the reference: SCN/CUDA/Convolution.cu:444-521).  Same output-stationary plan as conv.hip -- a block of
// 32 output rows owned by COUT/32 waves, all filter offsets of the block in one launch, every output row written once --
// with feature rows, packed weights and outputs in bf16 and fp32 accumulation on v_mfma_f32_32x32x16_bf16.
//
// At bf16 the matrix work of a step (Cin/16 MFMAs of 32 cycles) is far shorter than the time to fetch its operands, so
// the kernel is built for bytes, not for matrix-pipe occupancy: half-size rows gathered whole (16 B per thread),
// weight fragments (one 16-B load per lane per MFMA) and row pieces of the NEXT step requested before the current
// step's MFMAs, the result tile staged through LDS in fp32 so that the residual is added before the single rounding and
// rows leave as 16-byte pieces.
#include <algorithm>
#include <cstdlib>

#include "d3d_internal.h"

namespace d3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;   // storage type at the ABI (raw bits)

__device__ __forceinline__ void wave_lds_sync_b() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

static inline int padded_cin_bf16(int cin) {
  if (cin <= 16) return 16;
  if (cin <= 32) return 32;
  if (cin <= 64) return 64;
  if (cin <= 128) return 128;
  if (cin <= 256) return 256;
  return -1;
}

// packed[k][g][co][j] = bf16(w[k][8g+j][co])  (zero for 8g+j >= cin); transposed: W^T of offset k (flip: K-1-k)
__global__ void k_pack_weight_bf16(const float *__restrict__ w, int fv, int cin, int cout, int cp,
                                   __bf16 *__restrict__ packed) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)fv * cp * cout;
  if (t >= total) return;
  int j = (int)(t & 7);
  long u = t >> 3;
  int co = (int)(u % cout);
  u /= cout;
  int g = (int)(u % (cp / 8));
  int k = (int)(u / (cp / 8));
  int ci = 8 * g + j;
  packed[t] = (__bf16)(ci < cin ? w[((size_t)k * cin + ci) * cout + co] : 0.f);
}

// 8 bf16 (one 16-byte piece of a row) <-> 8 floats
__device__ __forceinline__ void unpack8(u32x4 v, float *f) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    f[2 * j] = __uint_as_float(v[j] << 16);
    f[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float *f) {
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; j++) b[j] = (__bf16)f[j];   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return __builtin_bit_cast(u32x4, b);
}

// CT = Cin tile staged per step (16..128), NCT tiles cover the padded Cin; BPW independent row blocks per workgroup
// (COUT == 32 only); RB = consecutive 32-row blocks that walk the UNION of their offset masks together and share every
// weight fragment: a wave fetches the fragments of a step once and applies them to RB accumulator tiles.  At bf16 the
// kernel is bound by the bytes a CU can pull from L2, and with one row block per weight fetch two thirds of those bytes
// are weights; RB = 2 / 4 cuts them to a half / a quarter (blocks are mask-sorted neighbours, so the union adds little).
template <int CT, int NCT, int COUT, int BPW, int RB>
__global__ __launch_bounds__(BPW *(COUT / 32) * 64) void k_conv_bf16(
    const bf16_t *__restrict__ in, const bf16_t *__restrict__ wp, const int32_t *__restrict__ nbrT, int npos,
    const int32_t *__restrict__ rows, const uint32_t *__restrict__ blkmask, int n_blk,
    const bf16_t *__restrict__ residual, bf16_t *__restrict__ out, int n_split, float *__restrict__ partial, BnPre pre) {
  constexpr int WPBLK = COUT / 32;
  static_assert(WPBLK == 1 || BPW == 1, "row blocks sharing a workgroup must be single-wave");
  static_assert(RB == 1 || BPW == 1, "row blocks that share weight fragments form one workgroup");
  constexpr int TPB = WPBLK * 64;
  constexpr int ROWS = 32 * RB;
  constexpr int CP = CT * NCT;
  constexpr int LDA = CT + 8;            // bf16 elements per LDS row: +16 B keeps the 16-byte pieces aligned
  constexpr int LPR = CT / 8;            // threads per gathered row (16 B each)
  constexpr int RPP = TPB / LPR;         // rows per gather pass
  constexpr int NIT = (ROWS / RPP) > 0 ? (ROWS / RPP) : 1;
  constexpr int NQ = CT / 16;            // MFMAs (K = 16) per accumulator tile and step
  constexpr int LDO = COUT + 4;          // fp32 elements per row of the result tile
  constexpr int SM_A = ROWS * LDA * 2, SM_O = 32 * LDO * 4;
  constexpr int SM = SM_A > SM_O ? SM_A : SM_O;
  __shared__ __attribute__((aligned(16))) char smem[BPW * SM];

  const int slot = threadIdx.x / TPB, tib = threadIdx.x % TPB;
  const int blk = (blockIdx.x * BPW + slot) * RB;      // first row block of this group
  if (blk >= n_blk) return;  // BPW > 1 only when waves are independent (no barrier below)
  const int nsub = min(RB, n_blk - blk);               // row blocks of the group that exist
  bf16_t *As = (bf16_t *)(smem + slot * SM);
  float *Os = (float *)(smem + slot * SM);
  const int lane = tib & 63, wib = tib >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int grow = tib / LPR, gc8 = tib % LPR;
  const int colbase = wib * 32;

  auto block_sync = [&]() {
    if constexpr (WPBLK == 1)
      wave_lds_sync_b();
    else
      __syncthreads();
  };

  uint32_t mask = blkmask[blk];
#pragma unroll
  for (int j = 1; j < RB; j++) mask |= blkmask[min(blk + j, n_blk - 1)];
  mask = __builtin_amdgcn_readfirstlane(mask);
  if (n_split > 1) {
    uint32_t keep = 0;     // by offset index (not rank): independent of the block's other rows, see conv.hip
    for (uint32_t mm = mask; mm; mm &= mm - 1) {
      const int kk = __builtin_ctz(mm);
      if (kk % n_split == (int)blockIdx.y) keep |= 1u << kk;
    }
    mask = keep;
  }
  f32x16 acc[RB];
#pragma unroll
  for (int j = 0; j < RB; j++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[j][i] = 0.f;

  const int32_t *nb = nbrT + (size_t)blk * 32;
  const int rows_here = nsub * 32;
  int idx[NIT];
  u32x4 stage[NIT];
  bool absent[NIT];   // the staged piece belongs to a missing neighbour (it read row 0 and is replaced by zeros)
  int stage_ct = 0;
  // fused BatchNorm (+ leaky ReLU) of the producer: this thread always gathers the same 8 channels of a Cin tile
  float bnw[NCT][8], bnb[NCT][8];
  if (pre.mean) {
#pragma unroll
    for (int t = 0; t < NCT; t++)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int c = t * CT + gc8 * 8 + j;
        const float w = pre.invstd[c] * (pre.weight ? pre.weight[c] : 1.f);
        bnw[t][j] = w;
        bnb[t][j] = -pre.mean[c] * w + (pre.bias ? pre.bias[c] : 0.f);
      }
  }
  const uint32_t lane_piece = (uint32_t)gc8 * 16u, lane_idx = (uint32_t)grow * 4u;
  auto load_idx = [&](int k) {
    const char *kb = (const char *)(nb + (size_t)k * npos);
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int row = it * RPP + grow;
      idx[it] = row < rows_here ? *(const int32_t *)(kb + (lane_idx + (uint32_t)(it * RPP * 4))) : -1;
    }
  };
  auto issue_data = [&](int ct) {
    stage_ct = ct;
    const char *base = (const char *)(in + ct * CT);   // rows are CP * 2 bytes
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int s = idx[it];
      absent[it] = s < 0;
      stage[it] = *(const u32x4 *)(base + ((uint32_t)(s < 0 ? 0 : s) * (uint32_t)(CP * 2) + lane_piece));
    }
  };
  auto commit_gather = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int row = it * RPP + grow;
      u32x4 v = stage[it];
      if (pre.mean) {
        float f[8];
        unpack8(v, f);
        f32x4 wlo, whi, blo, bhi;     // element-wise selects: a pointer into the per-tile arrays would put them in scratch
#pragma unroll
        for (int j = 0; j < 4; j++) {
          wlo[j] = stage_ct == 0 ? bnw[0][j] : bnw[NCT - 1][j];
          whi[j] = stage_ct == 0 ? bnw[0][4 + j] : bnw[NCT - 1][4 + j];
          blo[j] = stage_ct == 0 ? bnb[0][j] : bnb[NCT - 1][j];
          bhi[j] = stage_ct == 0 ? bnb[0][4 + j] : bnb[NCT - 1][4 + j];
        }
        const f32x4 lo = bn_act(f32x4{f[0], f[1], f[2], f[3]}, wlo, blo, pre.leak);
        const f32x4 hi = bn_act(f32x4{f[4], f[5], f[6], f[7]}, whi, bhi, pre.leak);
        const float g[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        v = pack8(g);
      }
      if (absent[it]) v = u32x4{0u, 0u, 0u, 0u};    // exact zeros by a select: nothing of row 0 (NaN, Inf) leaks
      if (row < ROWS) *(u32x4 *)(As + row * LDA + gc8 * 8) = v;
    }
  };
  auto next_k = [&](int k) -> int {
    const uint32_t mm = k >= 31 ? 0u : (mask & ~((2u << k) - 1u));
    return mm ? __builtin_ctz(mm) : -1;
  };
  // weight fragments of one step: lane (r, h) of MFMA q reads the 8 k-values 16 q + 8 h .. + 7 of column colbase + r
  const uint32_t lane_b = (uint32_t)(h * COUT + r) * 16u;
  auto load_b = [&](int k, int ct, u32x4 *b) {
    const char *wk = (const char *)(wp + ((size_t)(k * (CP / 8) + ct * (CT / 8)) * COUT + colbase) * 8);
#pragma unroll
    for (int q = 0; q < NQ; q++) b[q] = *(const u32x4 *)(wk + (lane_b + (uint32_t)(2 * q * COUT * 16)));
  };

  int k = mask ? __builtin_ctz(mask) : -1;
  int ct = 0;
  u32x4 bcur[NQ], bnxt[NQ];
  // two independent waves of loads, both ahead of their use (as conv.hip): the row indices of the step after next and
  // the row pieces + weight fragments of the next step
  if (k >= 0) {
    load_idx(k);
    issue_data(0);
    const int k_after = NCT > 1 ? k : next_k(k);
    if (k_after >= 0 && k_after != k) load_idx(k_after);
    load_b(k, 0, bcur);
  }
  while (k >= 0) {
    commit_gather();
    block_sync();
    int nk = k, nct = ct + 1;
    if (nct == NCT) {
      nct = 0;
      nk = next_k(k);
    }
    if (nk >= 0) {
      issue_data(nct);   // idx[] holds offset nk
      const int k2 = (nct + 1 < NCT) ? nk : next_k(nk);
      if (k2 >= 0 && k2 != nk) load_idx(k2);
      load_b(nk, nct, bnxt);
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const bf16x8 b = __builtin_bit_cast(bf16x8, bcur[q]);
#pragma unroll
      for (int j = 0; j < RB; j++) {
        const bf16x8 a = *(const bf16x8 *)(As + (j * 32 + r) * LDA + q * 16 + h * 8);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      }
    }
    block_sync();
#pragma unroll
    for (int q = 0; q < NQ; q++) bcur[q] = bnxt[q];
    k = nk;
    ct = nct;
  }
  // ---- epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) ----
  if (n_split > 1) {
#pragma unroll
    for (int j = 0; j < RB; j++) {
      if (j >= nsub) break;
      float *pt = partial + ((size_t)blockIdx.y * npos + (size_t)(blk + j) * 32) * COUT;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const int row_in = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        pt[(size_t)row_in * COUT + colbase + r] = acc[j][reg];
      }
    }
    return;
  }
  // result tiles through LDS in fp32, one row block at a time (the A tile is dead: both syncs of the last step are
  // behind us), so that the residual is added before the single rounding and rows leave as 16-byte pieces
  constexpr int OPR = COUT / 8;            // threads per output row (8 channels = 16 B of bf16 each)
  constexpr int ORP = TPB / OPR;           // rows per pass (16)
  const int orow_l = tib / OPR, oc8 = tib % OPR;
#pragma unroll
  for (int j = 0; j < RB; j++) {
    if (j >= nsub) break;                  // uniform over the workgroup
    if (j > 0) block_sync();               // the previous tile has been read
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int row_in = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      Os[row_in * LDO + colbase + r] = acc[j][reg];
    }
    block_sync();
#pragma unroll
    for (int p = 0; p < 32 / ORP; p++) {
      const int row_in = p * ORP + orow_l;
      const int orow = rows[(blk + j) * 32 + row_in];
      if (orow < 0) continue;
      const f32x4 lo = *(const f32x4 *)(Os + row_in * LDO + oc8 * 8), hi = *(const f32x4 *)(Os + row_in * LDO + oc8 * 8 + 4);
      float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      const size_t o = (size_t)orow * COUT + oc8 * 8;
      if (residual) {
        float g[8];
        unpack8(*(const u32x4 *)(residual + o), g);
#pragma unroll
        for (int jj = 0; jj < 8; jj++) f[jj] += g[jj];
      }
      *(u32x4 *)(out + o) = pack8(f);
    }
  }
}

// out[rows[pos]] = bf16(sum_y partial[y][pos] (+ residual)), y in increasing order
__global__ __launch_bounds__(256) void k_conv_reduce_bf16(const float *__restrict__ partial, int n_split, int npos,
                                                          int cout8, const int32_t *__restrict__ rows,
                                                          const bf16_t *__restrict__ residual,
                                                          bf16_t *__restrict__ out) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)npos * cout8) return;
  const int pos = (int)(t / cout8), c8 = (int)(t % cout8);
  const int orow = rows[pos];
  if (orow < 0) return;
  float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int y = 0; y < n_split; y++) {
    const float *p = partial + (((size_t)y * npos + pos) * cout8 + c8) * 8;
    const f32x4 lo = *(const f32x4 *)p, hi = *(const f32x4 *)(p + 4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      f[j] += lo[j];
      f[4 + j] += hi[j];
    }
  }
  const size_t o = ((size_t)orow * cout8 + c8) * 8;
  if (residual) {
    float g[8];
    unpack8(*(const u32x4 *)(residual + o), g);
#pragma unroll
    for (int j = 0; j < 8; j++) f[j] += g[j];
  }
  *(u32x4 *)(out + o) = pack8(f);
}

static constexpr int kSplitTargetWavesB = 4096;

template <int CT, int NCT, int COUT, int BPW, int RB = 1>
static int launch_tb(d3d_meta *m, const Plan &p, const bf16_t *in, const bf16_t *wp, const bf16_t *residual, bf16_t *out,
                     hipStream_t s, BnPre pre) {
  constexpr int WPBLK = COUT / 32;
  constexpr int threads = BPW * WPBLK * 64;
  const int npos = p.n_blk * 32;
  const long waves = (long)((p.n_blk + RB - 1) / RB) * WPBLK;
  int n_split = 1;
  if (BPW == 1 && p.K > 1 && m && waves < kSplitTargetWavesB)
    n_split = (int)std::min<long>(p.K, (kSplitTargetWavesB + waves - 1) / waves);
  float *partial = nullptr;
  size_t mark = 0;
  if (n_split > 1) {
    mark = m->feat_arena.used;
    partial = m->feat_arena.get<float>((size_t)n_split * npos * COUT);
    if (!partial) n_split = 1;
  }
  const dim3 grid((p.n_blk + BPW * RB - 1) / (BPW * RB), n_split);
  hipEvent_t ev_start, ev_stop;
  conv_timing_take(&ev_start, &ev_stop);
  if (ev_start) (void)hipEventRecord(ev_start, s);
  hipLaunchKernelGGL((k_conv_bf16<CT, NCT, COUT, BPW, RB>), grid, dim3(threads), 0, s, in, wp, p.nbrT, npos, p.rows, p.blkmask,
                     p.n_blk, residual, out, n_split, partial, pre);
  if (ev_stop) (void)hipEventRecord(ev_stop, s);
  if (n_split > 1) {
    const long total = (long)npos * (COUT / 8);
    hipLaunchKernelGGL(k_conv_reduce_bf16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, partial, n_split, npos,
                       COUT / 8, p.rows, residual, out);
    m->feat_arena.used = mark;
  }
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

// row blocks sharing a weight fetch (d3d_conv_bf16_tuning): only for launches that still fill the chip afterwards
static int g_bf16_rb = 2;
static long g_bf16_rb_min_waves = 2L * kSplitTargetWavesB;   // waves the launch must still have

template <int CT, int NCT>
static int launch_cb(d3d_meta *m, const Plan &p, const bf16_t *in, const bf16_t *wp, int cout, const bf16_t *residual,
                     bf16_t *out, hipStream_t s, BnPre pre) {
  const int rb = g_bf16_rb;
  if (rb >= 2 && cout >= 64 && (long)p.n_blk * (cout / 32) >= g_bf16_rb_min_waves * rb) {
    if (rb >= 4) {
      switch (cout) {
        case 64: return launch_tb<CT, NCT, 64, 1, 4>(m, p, in, wp, residual, out, s, pre);
        case 128: return launch_tb<CT, NCT, 128, 1, 4>(m, p, in, wp, residual, out, s, pre);
      }
    }
    switch (cout) {
      case 64: return launch_tb<CT, NCT, 64, 1, 2>(m, p, in, wp, residual, out, s, pre);
      case 128: return launch_tb<CT, NCT, 128, 1, 2>(m, p, in, wp, residual, out, s, pre);
      case 256: return launch_tb<CT, NCT, 256, 1, 2>(m, p, in, wp, residual, out, s, pre);
    }
  }
  switch (cout) {
    case 32: return launch_tb<CT, NCT, 32, 4>(m, p, in, wp, residual, out, s, pre);
    case 64: return launch_tb<CT, NCT, 64, 1>(m, p, in, wp, residual, out, s, pre);
    case 128: return launch_tb<CT, NCT, 128, 1>(m, p, in, wp, residual, out, s, pre);
    case 256: return launch_tb<CT, NCT, 256, 1>(m, p, in, wp, residual, out, s, pre);
  }
  set_error("bf16 convolution: Cout=%d not supported (32, 64, 128, 256)", cout);
  return D3D_ERR_UNSUPPORTED;
}

int launch_conv_bf16(d3d_meta *m, const Plan &p, const void *in_, int cin, const void *packed_w, int cout,
                     const void *residual_, void *out_, hipStream_t s, const d3d_bn_prologue *bn) {
  if (bn && bn->out_stats_rows) *bn->out_stats_rows = 0;   // column statistics are an fp32-storage feature
  const bf16_t *in = (const bf16_t *)in_, *wp = (const bf16_t *)packed_w, *residual = (const bf16_t *)residual_;
  bf16_t *out = (bf16_t *)out_;
  if (p.n_rows == 0) {
    hipEvent_t a, b;
    conv_timing_take(&a, &b);
    if (a) (void)hipEventRecord(a, s);
    if (b) (void)hipEventRecord(b, s);
    return D3D_OK;
  }
  D3D_REQUIRE(in && wp && out, "bf16 convolution: null pointer");
  D3D_REQUIRE(padded_cin_bf16(cin) == cin, "bf16 convolution: feature rows must be stored with 16, 32, 64, 128 or 256 "
              "channels (got %d); pad narrower inputs with zero channels", cin);
  D3D_REQUIRE((size_t)p.n_in * (size_t)cin * 2 < ((size_t)1 << 32),
              "bf16 convolution: gathered tensor of %d rows x %d channels exceeds the 4 GiB of the 32-bit gather offsets", p.n_in, cin);
  D3D_REQUIRE((((uintptr_t)in | (uintptr_t)wp | (uintptr_t)out | (uintptr_t)residual) & 15) == 0,
              "bf16 convolution: tensors must be 16-byte aligned");
  BnPre pre = {nullptr, nullptr, nullptr, nullptr, 0.f};
  if (bn && bn->mean) {
    D3D_REQUIRE(bn->invstd, "fused BatchNorm prologue: null invstd");
    pre = {bn->mean, bn->invstd, bn->weight, bn->bias, bn->leakiness};
  }
  switch (cin) {
    case 16: return launch_cb<16, 1>(m, p, in, wp, cout, residual, out, s, pre);
    case 32: return launch_cb<32, 1>(m, p, in, wp, cout, residual, out, s, pre);
    case 64: return launch_cb<64, 1>(m, p, in, wp, cout, residual, out, s, pre);
    case 128: return launch_cb<128, 1>(m, p, in, wp, cout, residual, out, s, pre);
    case 256: return launch_cb<128, 2>(m, p, in, wp, cout, residual, out, s, pre);
  }
  return D3D_ERR_UNSUPPORTED;
}

}  // namespace d3d

using namespace d3d;

extern "C" {

int d3d_conv_bf16_tuning(int row_blocks, long min_waves) {
  D3D_REQUIRE(row_blocks == 1 || row_blocks == 2 || row_blocks == 4, "conv_bf16_tuning: row_blocks must be 1, 2 or 4");
  g_bf16_rb = row_blocks;
  g_bf16_rb_min_waves = min_waves >= 0 ? min_waves : 2L * kSplitTargetWavesB;
  return D3D_OK;
}

size_t d3d_packed_weight_bytes(int fv, int cin, int cout, int dtype) {
  if (dtype == D3D_F32) return d3d_packed_weight_floats(fv, cin, cout) * sizeof(float);
  const int cp = padded_cin_bf16(cin);
  if (dtype != D3D_BF16 || cp < 0 || fv <= 0 || cout <= 0) return 0;
  return (size_t)fv * cp * cout * 2;
}

int d3d_pack_conv_weight_dt(const float *w, int fv, int cin, int cout, void *packed, int dtype, void *stream) {
  if (dtype == D3D_F32) return d3d_pack_conv_weight(w, fv, cin, cout, (float *)packed, stream);
  hipStream_t s = (hipStream_t)stream;
  const int cp = padded_cin_bf16(cin);
  D3D_REQUIRE(dtype == D3D_BF16 && w && packed && fv > 0 && cout > 0 && cp > 0, "pack_conv_weight_dt: bad arguments (Cin=%d)", cin);
  const long total = (long)fv * cp * cout;
  hipLaunchKernelGGL(k_pack_weight_bf16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, fv, cin, cout, cp,
                     (__bf16 *)packed);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

}  // extern "C"
