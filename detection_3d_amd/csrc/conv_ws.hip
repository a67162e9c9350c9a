// a6. Sparse convolution forward, weight-sharing form for the large launches of the wide layers (Cin = Cout = 64 / 128).
//
// k_conv (conv.hip) gives a block of 32 output rows to COUT/32 waves that each fetch THEIR weight fragments of every
// (offset, Cin tile) step straight from L2: 16 KB per wave and step at 128 -> 128, requested ~1000 cycles before use
// against ~8000 cycles of L2 latency under load -- the matrix pipe is busy 0.39-0.49 of the time.
// Here a workgroup is RBW row blocks, ONE wave each (the wave holds all COUT/32 accumulator tiles of its block), and the
// weight tile of a step is read from L2 once per workgroup: every thread fetches 1/(64 RBW) of it a whole step ahead
// (registers), the tile is double-buffered in LDS in the packed (k-interleaved) layout it has in memory, and all waves
// take their B fragments from there.  L2 -> L1 bytes per MFMA drop 2-2.5x, the fetch has a step's matrix work to arrive
// under, and a step costs one workgroup barrier.  The waves walk the UNION of their blocks' offset masks (rows are sorted
// by mask: neighbouring blocks share most offsets); a wave whose block lacks the step's offset skips its MFMAs.
// Same products in the same order as k_conv: bit-identical results (tests run both, D3D_CONV_WS=0 switches it off).
#include <algorithm>

#include "d3d_internal.h"

namespace d3d {

typedef float wf32x16 __attribute__((ext_vector_type(16)));
typedef float wf32x4 __attribute__((ext_vector_type(4)));

// CP = Cin (= padded Cin: 16-byte row pieces), CT = Cin tile of a step, RBW = row blocks (waves) per workgroup
template <int CP, int COUT, int CT, int RBW>
__global__ __launch_bounds__(RBW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_conv_ws(const float *__restrict__ in, const float *__restrict__ wp,
                                                      const int32_t *__restrict__ nbrT, int npos,
                                                      const int32_t *__restrict__ rows,
                                                      const uint32_t *__restrict__ blkmask, int n_blk,
                                                      const float *__restrict__ residual, float *__restrict__ out,
                                                      BnPre pre, uint32_t in_bytes, double *__restrict__ stat) {
  constexpr int NCT = CP / CT, NT = COUT / 32;
  constexpr int LDA = CT + 4;              // +4 dwords: conflict-free ds_read_b128 of 32 rows
  constexpr int LPR = CT / 4;              // lanes per gathered row (16 B each)
  constexpr int RPP = 64 / LPR;            // rows per gather pass of the wave
  constexpr int NIT = 32 / RPP;
  constexpr int NQ = CT / 8;               // q-iterations (4 MFMAs per accumulator tile each) of a step
  constexpr int WT = CT * COUT;            // floats of a weight tile
  constexpr int WPT = WT / 4 / (RBW * 64); // 16-byte pieces of it per thread
  static_assert(WPT >= 1 && WT % (4 * RBW * 64) == 0, "weight tile must split evenly over the workgroup");
  __shared__ __attribute__((aligned(16))) float As_all[RBW * 32 * LDA];
  __shared__ __attribute__((aligned(16))) float Ws[2 * WT];
  __shared__ uint32_t umask_s;

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int blk = blockIdx.x * RBW + wave;
  const bool have = blk < n_blk;
  float *As = As_all + wave * 32 * LDA;
  const int r = lane & 31, h = lane >> 5;
  const int grow = lane / LPR, gc4 = lane % LPR;

  const uint32_t mymask = have ? __builtin_amdgcn_readfirstlane(blkmask[blk]) : 0u;
  if (threadIdx.x == 0) umask_s = 0;
  __syncthreads();
  if (lane == 0 && mymask) atomicOr(&umask_s, mymask);
  __syncthreads();
  const uint32_t umask = umask_s;
  const int rowid = have ? rows[blk * 32 + r] : -1;
  wf32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[nt][i] = 0.f;

  const int32_t *nb = nbrT + (size_t)(have ? blk : 0) * 32;
  int idx[NIT];
  wf32x4 stage[NIT];
  float mreal[NIT];
  int stage_ct = 0;
  wf32x4 bnw[NCT], bnb[NCT];
#pragma unroll
  for (int t = 0; t < NCT; t++) {
    bnw[t] = {1.f, 1.f, 1.f, 1.f};
    bnb[t] = {0.f, 0.f, 0.f, 0.f};
    if (pre.mean) {
      const int c = t * CT + gc4 * 4;
      const wf32x4 is = *(const wf32x4 *)(pre.invstd + c), mu = *(const wf32x4 *)(pre.mean + c);
      const wf32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
      const wf32x4 ga = pre.weight ? *(const wf32x4 *)(pre.weight + c) : one;
      const wf32x4 be = pre.bias ? *(const wf32x4 *)(pre.bias + c) : zero;
      bnw[t] = is * ga;
      bnb[t] = -mu * bnw[t] + be;
    }
  }
  const uint32_t lane_piece = (uint32_t)gc4 * 16u, lane_idx = (uint32_t)grow * 4u;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)in_bytes, 0x00020000);
  auto load_idx = [&](int k) {          // (a block without offset k reads -1 padding or real indices it never uses)
    const char *kb = (const char *)(nb + (size_t)k * npos);
#pragma unroll
    for (int it = 0; it < NIT; it++) idx[it] = *(const int32_t *)(kb + (lane_idx + (uint32_t)(it * RPP * 4)));
  };
  auto issue_data = [&](int ct, bool mine) {
    stage_ct = ct;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int s = mine ? idx[it] : -1;
      mreal[it] = s >= 0 ? 1.f : 0.f;
      const uint32_t off = s < 0 ? 0xfffffff0u : (uint32_t)s * (uint32_t)(CP * 4) + (uint32_t)(ct * CT * 4) + lane_piece;
      stage[it] = __builtin_bit_cast(wf32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)off, 0, 0));
    }
  };
  auto commit_gather = [&]() {
    wf32x4 bw = bnw[0], bb = bnb[0];
#pragma unroll
    for (int t = 1; t < NCT; t++)
      if (stage_ct == t) {
        bw = bnw[t];
        bb = bnb[t];
      }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int row = it * RPP + grow;
      wf32x4 v = stage[it];
      if (pre.mean) v = bn_act(v, bw, bb, pre.leak) * mreal[it];
      *(wf32x4 *)(As + row * LDA + gc4 * 4) = v;
    }
  };
  // the workgroup's share of a weight tile: piece p of thread t is 16 bytes at tile + (p * 64 RBW + t) * 16
  wf32x4 wst[WPT];
  auto issue_w = [&](int k, int ct) {
    const char *src = (const char *)(wp + ((size_t)(k * (CP / 4) + ct * (CT / 4)) * COUT) * 4);
#pragma unroll
    for (int p = 0; p < WPT; p++) wst[p] = *(const wf32x4 *)(src + (size_t)(p * RBW * 64 + (int)threadIdx.x) * 16);
  };
  auto commit_w = [&](int buf) {
    float *dst = Ws + buf * WT;
#pragma unroll
    for (int p = 0; p < WPT; p++) *(wf32x4 *)(dst + (size_t)(p * RBW * 64 + (int)threadIdx.x) * 4) = wst[p];
  };
  auto next_k = [&](int k) -> int {
    const uint32_t m = k >= 31 ? 0u : (umask & ~((2u << k) - 1u));
    return m ? __builtin_ctz(m) : -1;
  };

  int k = umask ? __builtin_ctz(umask) : -1;
  int ct = 0, buf = 0;
  if (k >= 0) {
    load_idx(k);
    issue_w(k, 0);
    issue_data(0, (mymask >> k) & 1u);
    commit_w(0);
    if (NCT == 1) {
      const int k1 = next_k(k);
      if (k1 >= 0) load_idx(k1);
    }
  }
  while (k >= 0) {
    const bool mine = (mymask >> k) & 1u;
    commit_gather();
    __syncthreads();                       // this step's weight tile (all waves) and row tile (own wave) are in LDS
    int nk = k, nct = ct + 1;
    if (nct == NCT) {
      nct = 0;
      nk = next_k(k);
    }
    if (nk >= 0) {
      issue_w(nk, nct);
      issue_data(nct, (mymask >> nk) & 1u);        // idx holds offset nk's rows
      const int k2 = (nct + 1 < NCT) ? nk : next_k(nk);
      if (k2 >= 0 && k2 != nk) load_idx(k2);
    }
    if (mine) {
      __builtin_amdgcn_s_setprio(1);
      const float *Wb = Ws + buf * WT;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const wf32x4 a = *(const wf32x4 *)(As + r * LDA + q * 8 + h * 4);
        wf32x4 b[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) b[nt] = *(const wf32x4 *)(Wb + ((size_t)(2 * q + h) * COUT + nt * 32 + r) * 4);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[nt][0], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[nt][1], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[nt][2], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[nt][3], acc[nt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
    if (nk >= 0) commit_w(buf ^ 1);        // the other buffer: nobody reads it before the next barrier
    buf ^= 1;
    k = nk;
    ct = nct;
  }
  if (!have) return;
  // ---- epilogue (as k_conv): C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
  double cs[NT], css[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++) cs[nt] = css[nt] = 0.0;
#pragma unroll
  for (int g4 = 0; g4 < 4; g4++) {
    int orow[4];
    float res[4][NT];
#pragma unroll
    for (int j = 0; j < 4; j++) orow[j] = __shfl(rowid, j + 8 * g4 + 4 * h, 64);
    if (residual) {
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) res[j][nt] = residual[(size_t)(orow[j] < 0 ? 0 : orow[j]) * COUT + nt * 32 + r];
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[nt][g4 * 4 + j] += res[j][nt];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (orow[j] < 0) continue;
#pragma unroll
      for (int nt = 0; nt < NT; nt++) out[(size_t)orow[j] * COUT + nt * 32 + r] = acc[nt][g4 * 4 + j];
      if (stat) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
          const double d = (double)acc[nt][g4 * 4 + j];
          cs[nt] += d;
          css[nt] += d * d;
        }
      }
    }
  }
  if (stat) {
    double *sp = stat + (size_t)blk * (2 * COUT);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const double a = cs[nt] + __shfl_xor(cs[nt], 32, 64), b = css[nt] + __shfl_xor(css[nt], 32, 64);
      if (h == 0) {
        sp[nt * 32 + r] = a;
        sp[COUT + nt * 32 + r] = b;
      }
    }
  }
}

// D3D_CONV_WS: 0 off (default), 1 on for the large launches, 2 on for every launch size (tests).  Measured on the
// 500 k-point building (scripts/conv_ws_probe.py, ms per building): 64 -> 64 0.83 with k_conv against 0.97 here, 128 -> 128
// 1.36-1.43 against 1.37-1.38 -- the weight fetch is not what holds k_conv back, and the lockstep of four row blocks over
// the union of their offset masks costs the narrow layer more than the shared fetch saves.  Kept as a measured variant.
static int g_ws_mode = [] {
  const char *e = getenv("D3D_CONV_WS");
  return e ? atoi(e) : 0;
}();
static constexpr int kWsMinBlocks = 2048;  // launches below this stay with k_conv (offset-split / latency-bound there)

// -> true when the launch was taken (k_conv_ws enqueued)
bool launch_conv_ws(const Plan &p, const float *in, int cin, const float *wp, int cout, const float *residual, float *out,
                    hipStream_t s, BnPre pre, double *stat, uint32_t in_bytes) {
  if (g_ws_mode == 0 || p.K <= 1) return false;
  if (g_ws_mode == 1 && p.n_blk < kWsMinBlocks) return false;
  const int npos = p.n_blk * 32;
  if (cin == 64 && cout == 64) {
    constexpr int RBW = 4;
    hipLaunchKernelGGL((k_conv_ws<64, 64, 64, RBW>), dim3((p.n_blk + RBW - 1) / RBW), dim3(RBW * 64), 0, s, in, wp, p.nbrT,
                       npos, p.rows, p.blkmask, p.n_blk, residual, out, pre, in_bytes, stat);
    return true;
  }
  if (cin == 128 && cout == 128) {
    constexpr int RBW = 4;
    hipLaunchKernelGGL((k_conv_ws<128, 128, 32, RBW>), dim3((p.n_blk + RBW - 1) / RBW), dim3(RBW * 64), 0, s, in, wp, p.nbrT,
                       npos, p.rows, p.blkmask, p.n_blk, residual, out, pre, in_bytes, stat);
    return true;
  }
  return false;
}

}  // namespace d3d

extern "C" int d3d_conv_ws_mode(int mode) {
  const int was = d3d::g_ws_mode;
  if (mode >= 0) d3d::g_ws_mode = mode;
  return was;
}
