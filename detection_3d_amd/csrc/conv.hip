// a6. Sparse convolution forward as ONE output-stationary launch per layer (the reference issues
// one kernel + one blocking rule memcpy per filter offset, SCN/CUDA/RuleBookIterator.h:15-32).
//
// Work decomposition: a block of 32 output rows (rows sorted by neighbour mask, see grid.hip
// finalize_plan) is owned by COUT/32/NT waves, each holding NT 32x32 accumulator tiles.  For
// every filter offset k present in the block's mask the waves gather the 32 input rows into an
// LDS tile (full rows, 16 B per thread, coalesced; register-staged one step ahead so that the
// loads overlap the matrix work), then run v_mfma_f32_32x32x2_f32 over Cin with the B operand
// (k-interleaved packed weights, L2-resident) read straight from global memory.  Accumulators
// stay in registers across all offsets; every output row is written exactly once (no atomics,
// deterministic, independent of how rows are grouped).
#include <algorithm>

#include "d3d_internal.h"

namespace d3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wave execute in issue order; this only stops the compiler from
  // moving LDS accesses of different lanes across the hand-off point.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

static inline int padded_cin(int cin) {
  if (cin <= 16) return 16;
  if (cin <= 32) return 32;
  if (cin <= 64) return 64;
  if (cin <= 128) return 128;
  if (cin <= 256) return 256;
  return -1;
}

// packed[k][g][co][j] = w[k][4g+j][co]  (zero for 4g+j >= cin)
__global__ void k_pack_weight(const float *__restrict__ w, int fv, int cin, int cout, int cp,
                              float *__restrict__ packed) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)fv * cp * cout;
  if (t >= total) return;
  int j = (int)(t & 3);
  long u = t >> 2;
  int co = (int)(u % cout);
  u /= cout;
  int g = (int)(u % (cp / 4));
  int k = (int)(u / (cp / 4));
  int ci = 4 * g + j;
  packed[t] = ci < cin ? w[((size_t)k * cin + ci) * cout + co] : 0.f;
}

// CT   = Cin tile staged in LDS per step (multiple of 8, <= 128); NCT tiles cover Cin
// NT   = 32-column accumulator tiles per wave; a row block is shared by WPBLK = COUT/32/NT waves
// BPW  = row blocks per workgroup (only with WPBLK == 1, where waves never synchronise)
// VEC  = Cin equals the padded CT * NCT (16-byte row pieces): branch-free, VALU-lean gather
//
// The fp32 MFMA shares the SIMD's vector ALU with ordinary VALU instructions (scripts/mfma_probe.hip: every VALU
// instruction issued by ANY wave of the SIMD takes ~3.5 cycles away from the matrix pipe, nothing co-executes), so
// the gather costs as few VALU instructions as possible: wave-uniform (scalar) base pointers + 32-bit per-lane byte
// offsets, one multiplier per row (1 = real, 0 = absent neighbour) instead of per-element selects, fused multiply-add
// + max for the BatchNorm + ReLU prologue.
// LATE = the next step's gather (and index) loads are issued after the step's FIRST q-iteration instead of ahead of its matrix
//        work, branch-free (a step without successor requests absent rows).  The compiler's s_waitcnt placement is
//        conservative at the loop header: the wait in front of the step's first MFMA -- for weight fragments requested a
//        step ago -- also covered every load issued since, i.e. the gathers just requested: each step began with the
//        wave stalled for a full gather round trip.  Issued behind the first MFMAs, only loads of the previous step are
//        outstanding at that wait, and the counted waits inside the straight-line q-loop leave the gathers in flight.
template <int CT, int NCT, int COUT, int NT, int BPW, bool VEC, bool LATE = false>
__global__ __launch_bounds__(BPW *(COUT / 32 / NT) * 64) void k_conv(
    const float *__restrict__ in, int cin, const float *__restrict__ wp,
    const int32_t *__restrict__ nbrT, int npos, const int32_t *__restrict__ rows,
    const uint32_t *__restrict__ blkmask, int n_blk, const float *__restrict__ residual,
    float *__restrict__ out, int n_split, float *__restrict__ partial, BnPre pre, uint32_t in_bytes,
    double *__restrict__ stat) {
  constexpr int WPBLK = COUT / 32 / NT;
  static_assert(WPBLK == 1 || BPW == 1, "row blocks sharing a workgroup must be single-wave");
  constexpr int TPB = WPBLK * 64;  // threads working on one row block
  constexpr int LDA = CT + 4;      // +4 dwords: conflict-free ds_read_b128 of 32 rows
  constexpr int CP = CT * NCT;
  constexpr int LPR = CT / 4;      // threads per gathered row (16 B each)
  constexpr int RPP = TPB / LPR;   // rows per gather pass
  constexpr int NIT = (32 / RPP) > 0 ? (32 / RPP) : 1;
  constexpr int NQ = CT / 8;       // q-iterations (4 MFMAs per accumulator tile each) of a step
  // weight fragments in flight (ring): 8 q-iterations = 2048 matrix cycles of lead (4: the 128 -> 128 family 0.91 ms per
  // building against 0.89, 2: 0.95; D3D_QA at compile time)
#ifndef D3D_QA
#define D3D_QA 8
#endif
  constexpr int QA = NQ < D3D_QA ? NQ : D3D_QA;
  static_assert(NQ % QA == 0, "ring depth must divide the q-iterations of a step");
  __shared__ __attribute__((aligned(16))) float smem[BPW * 32 * LDA];

  const int slot = threadIdx.x / TPB, tib = threadIdx.x % TPB;
  const int blk = blockIdx.x * BPW + slot;
  if (blk >= n_blk) return;  // BPW > 1 only when waves are independent (no barrier below)
  float *As = smem + slot * 32 * LDA;
  const int lane = tib & 63, wib = tib >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int grow = tib / LPR, gc4 = tib % LPR;
  const int colbase = wib * NT * 32;

  auto block_sync = [&]() {
    if constexpr (WPBLK == 1)
      wave_lds_sync();
    else
      __syncthreads();
  };

  // active offsets of the block; wave-uniform: keep it (and with it k, the weight / index base pointers and the loop
  // control) in SGPRs
  uint32_t mask = __builtin_amdgcn_readfirstlane(blkmask[blk]);
  if (n_split > 1) {
    // offset-split launch (few rows): this workgroup keeps every n_split-th active offset and
    // writes a partial tile; k_conv_reduce sums the partials in a fixed order.
    // by the offset's INDEX, not by its rank among the block's active offsets: which partial sum an offset of a row
    // lands in then does not depend on the other rows of the block, so the result is independent of how rows are
    // grouped into blocks (e.g. the same rows reached through plans of different builds)
    uint32_t keep = 0;
    for (uint32_t m = mask; m; m &= m - 1) {
      const int kk = __builtin_ctz(m);
      if (kk % n_split == (int)blockIdx.y) keep |= 1u << kk;
    }
    mask = keep;
  }
  const int rowid = rows[blk * 32 + r];
  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; nt++)
#pragma unroll
    for (int i = 0; i < 16; i++) acc[nt][i] = 0.f;

  const int32_t *nb = nbrT + (size_t)blk * 32;
  // Gather of (offset k, Cin tile ct) in two independent waves of loads, both issued ahead of their use:
  //   load_idx(k)   : the NIT input-row indices this thread needs for offset k   (one step before issue_data)
  //   issue_data(ct): the 16-byte row pieces, branch-free -- an absent neighbour reads row 0 and is zeroed
  //                   at commit time, so that no load waits for another one
  //   commit_gather : registers -> LDS (+ the fused BatchNorm), after the previous step's MFMAs
  int idx[NIT];
  f32x4 stage[NIT];
  float mreal[NIT];  // 1.f for a real row, 0.f for an absent one
  int stage_ct = 0;
  // optional fused BatchNorm + leaky ReLU of the producer layer (y = leaky(fma(x, w, b)), applied to real rows
  // only: a missing neighbour contributes zeros, as a zero row of the normalised tensor would not);
  // this thread always gathers the same 4 channels of a Cin tile, so w and b are fetched once
  f32x4 bnw[NCT], bnb[NCT];
#pragma unroll
  for (int t = 0; t < NCT; t++) {
    bnw[t] = {1.f, 1.f, 1.f, 1.f};
    bnb[t] = {0.f, 0.f, 0.f, 0.f};
    if (pre.mean) {
      const int c = t * CT + gc4 * 4;
      const f32x4 is = *(const f32x4 *)(pre.invstd + c), mu = *(const f32x4 *)(pre.mean + c);
      const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
      const f32x4 ga = pre.weight ? *(const f32x4 *)(pre.weight + c) : one;
      const f32x4 be = pre.bias ? *(const f32x4 *)(pre.bias + c) : zero;
      bnw[t] = is * ga;
      bnb[t] = -mu * bnw[t] + be;
    }
  }
  const uint32_t lane_piece = (uint32_t)gc4 * 16u, lane_idx = (uint32_t)grow * 4u;
  // The gathered tensor as a raw buffer of in_bytes: a row piece of an ABSENT neighbour is requested at an offset past
  // the end, which the hardware range check answers with zeros -- no branch, no select, and nothing of a real row
  // (row 0 used to stand in, and its NaN or Inf would have spread through the 0 * x of the commit) reaches the tile.
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)in_bytes, 0x00020000);
  auto load_idx = [&](int k) {
    const char *kb = (const char *)(nb + (size_t)k * npos);  // wave-uniform
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      if constexpr (RPP <= 32) {
        idx[it] = *(const int32_t *)(kb + (lane_idx + (uint32_t)(it * RPP * 4)));
      } else {  // a pass wider than the block (tiny Cin tile, many waves): threads past row 31 idle
        idx[it] = grow < 32 ? *(const int32_t *)(kb + lane_idx) : -1;
      }
    }
  };
  auto issue_data = [&](int ct) {
    stage_ct = ct;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int s = idx[it];
      mreal[it] = s >= 0 ? 1.f : 0.f;
      if constexpr (VEC) {
        // rows are CP * 4 bytes (< 4 GiB tensor); 0xfffffff0 + 16 exceeds any buffer size
        const uint32_t off = s < 0 ? 0xfffffff0u : (uint32_t)s * (uint32_t)(CP * 4) + (uint32_t)(ct * CT * 4) + lane_piece;
        stage[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)off, 0, 0));
      } else {
        const float *p = in + (size_t)(s < 0 ? 0 : s) * cin + ct * CT + gc4 * 4;
        const int c = ct * CT + gc4 * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c + 0 < cin) v[0] = p[0];
        if (c + 1 < cin) v[1] = p[1];
        if (c + 2 < cin) v[2] = p[2];
        if (c + 3 < cin) v[3] = p[3];
        stage[it] = v;
      }
    }
  };
  auto commit_gather = [&]() {
    const f32x4 bw = stage_ct == 0 ? bnw[0] : bnw[NCT - 1], bb = stage_ct == 0 ? bnb[0] : bnb[NCT - 1];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int row = it * RPP + grow;
      f32x4 v = stage[it];
      if constexpr (VEC) {
        // an absent row arrived as zeros; only the fused BatchNorm turns them into leaky(beta'), which the row's
        // multiplier (0 or 1, one VALU instruction per 2 elements) takes out again -- finite times 0
        if (pre.mean) v = bn_act(v, bw, bb, pre.leak) * mreal[it];
      } else {
        if (pre.mean) v = bn_act(v, bw, bb, pre.leak);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        v = mreal[it] != 0.f ? v : zero;     // a select: row 0 stood in for the absent neighbour
      }
      if (row < 32) *(f32x4 *)(As + row * LDA + gc4 * 4) = v;
    }
  };

  // step tokens: (offset k, Cin tile ct) in increasing (k, ct) order over the active offsets
  auto next_k = [&](int k) -> int {
    const uint32_t m = k >= 31 ? 0u : (mask & ~((2u << k) - 1u));
    return m ? __builtin_ctz(m) : -1;
  };
  int k = mask ? __builtin_ctz(mask) : -1;
  int ct = 0;
  // weight fragments (packed weights, L2-resident, shared by every block) come through a ring of QA q-iterations
  // in flight that runs across step boundaries: the last QA refills of a step fetch the first fragments of the
  // next one (the compiler alone keeps only ~1 load ahead)
  f32x4 ring[QA][NT];
  const uint32_t lane_b = (uint32_t)(h * COUT + r) * 16u;   // this lane's byte offset inside a weight fragment row
  if (k >= 0) {
    load_idx(k);
    issue_data(0);
    const int k_after = NCT > 1 ? k : next_k(k);   // indices of the step after this one
    if (k_after >= 0 && k_after != k) load_idx(k_after);
    const char *wk0 = (const char *)(wp + ((size_t)(k * (CP / 4)) * COUT + colbase) * 4);
#pragma unroll
    for (int q = 0; q < QA; q++)
#pragma unroll
      for (int nt = 0; nt < NT; nt++)
        ring[q][nt] = *(const f32x4 *)(wk0 + (lane_b + (uint32_t)((2 * q * COUT + nt * 32) * 16)));
  }
  while (k >= 0) {
    commit_gather();
    block_sync();
    // next (offset, tile) step
    int nk = k, nct = ct + 1;
    if (nct == NCT) {
      nct = 0;
      nk = next_k(k);
    }
    const int k2 = nk < 0 ? -1 : ((nct + 1 < NCT) ? nk : next_k(nk));   // the step after that: its indices are requested now
    if constexpr (!LATE) {
      if (nk >= 0) {
        issue_data(nct);  // loads fly while the matrix cores work; idx holds offset nk
        if (k2 >= 0 && k2 != nk) load_idx(k2);
      }
    }
    __builtin_amdgcn_s_setprio(1);
    // ---- 32 x (NT*32) += A[32 x CT] * W[k][CT x cols] ----
    const char *wk = (const char *)(wp + ((size_t)(k * (CP / 4) + ct * (CT / 4)) * COUT + colbase) * 4);
    const char *wk_next = nk >= 0 ? (const char *)(wp + ((size_t)(nk * (CP / 4) + nct * (CT / 4)) * COUT + colbase) * 4) : wk;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      f32x4 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; nt++) b[nt] = ring[q % QA][nt];
      {
        const char *src = (q + QA < NQ) ? wk : wk_next;   // wave-uniform base + per-lane byte offset
        const int qq = (q + QA) % NQ;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
          ring[q % QA][nt] = *(const f32x4 *)(src + (lane_b + (uint32_t)((2 * qq * COUT + nt * 32) * 16)));
      }
      const f32x4 a = *(const f32x4 *)(As + r * LDA + q * 8 + h * 4);
#pragma unroll
      for (int nt = 0; nt < NT; nt++) {
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[nt][0], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[nt][1], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[nt][2], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[nt][3], acc[nt], 0, 0, 0);
      }
      if constexpr (LATE) {
        if (q == 0) {
          // branch-free: without a next step the rows asked for are absent ones (zeros from the range check, never
          // committed), and the index loads repeat offset k's
          if (nk < 0) {
#pragma unroll
            for (int it = 0; it < NIT; it++) idx[it] = -1;
          }
          issue_data(nk >= 0 ? nct : 0);
          load_idx(k2 >= 0 ? k2 : k);
        }
      }
    }
    // order of the step's instruction stream: the gather / index loads up front, then per q-iteration one MFMA,
    // one LDS read (A of the next iteration), one weight load (ring refill QA iterations ahead), the other MFMAs
    if constexpr (!LATE) {
      __builtin_amdgcn_sched_group_barrier(0x020, 2 * NIT, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT - 1, 0);
      }
    } else {
      // LATE: A of q = 0, its first MFMA, A of q = 1, the ring refill, the other MFMAs of q = 0, THEN the gather and
      // index loads, then the remaining q-iterations as above
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT - 1, 0);
        if (q == 0) __builtin_amdgcn_sched_group_barrier(0x020, 2 * NIT, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    block_sync();
    k = nk;
    ct = nct;
  }
  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
  if (n_split > 1) {
    float *pt = partial + ((size_t)blockIdx.y * npos + (size_t)blk * 32) * COUT;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int row_in = (reg & 3) + 8 * (reg >> 2) + 4 * h;
#pragma unroll
      for (int nt = 0; nt < NT; nt++) pt[(size_t)row_in * COUT + colbase + nt * 32 + r] = acc[nt][reg];
    }
    return;
  }
  // four rows at a time: residual reads first (a padded row reads row 0 and is dropped), then adds and stores --
  // no load waits for another, and the epilogue does not set the kernel's register budget
  // `stat`: column sums and sums of squares (fp64) of the block's real rows, as they are stored -- the statistics of the
  // BatchNorm that follows are then a fixed-order sum of n_blk small vectors instead of a second pass over the tensor
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
        for (int nt = 0; nt < NT; nt++)
          res[j][nt] = residual[(size_t)(orow[j] < 0 ? 0 : orow[j]) * COUT + colbase + nt * 32 + r];
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[nt][g4 * 4 + j] += res[j][nt];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (orow[j] < 0) continue;
#pragma unroll
      for (int nt = 0; nt < NT; nt++) out[(size_t)orow[j] * COUT + colbase + nt * 32 + r] = acc[nt][g4 * 4 + j];
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
  if (stat) {   // rows 0-3, 8-11, .. live in lanes 0-31, the others in lanes 32-63: both halves form the same sum
    double *sp = stat + (size_t)blk * (2 * COUT);
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
      const double a = cs[nt] + __shfl_xor(cs[nt], 32, 64), b = css[nt] + __shfl_xor(css[nt], 32, 64);
      if (h == 0) {
        sp[colbase + nt * 32 + r] = a;
        sp[COUT + colbase + nt * 32 + r] = b;
      }
    }
  }
}

// out[rows[pos]] = sum_y partial[y][pos] (+ residual), y in increasing order.  `stat` (may be null): per-workgroup column
// sums / sums of squares of the rows it wrote, [gridDim.x][2 * cout] (see k_conv); a workgroup covers 1024 / cout rows.
__global__ __launch_bounds__(256) void k_conv_reduce(const float *__restrict__ partial, int n_split,
                                                     int npos, int cout4, const int32_t *__restrict__ rows,
                                                     const float *__restrict__ residual,
                                                     float *__restrict__ out, double *__restrict__ stat) {
  __shared__ double red[2][256][4];
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = t < (long)npos * cout4;
  const int pos = in_range ? (int)(t / cout4) : 0, c4 = (int)(t % cout4);
  const int orow = in_range ? rows[pos] : -1;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (orow >= 0) {
    for (int y = 0; y < n_split; y++) acc += *(const f32x4 *)(partial + (((size_t)y * npos + pos) * cout4 + c4) * 4);
    const size_t o = ((size_t)orow * cout4 + c4) * 4;
    if (residual) acc += *(const f32x4 *)(residual + o);
    *(f32x4 *)(out + o) = acc;
  }
  if (!stat) return;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const double d = orow >= 0 ? (double)acc[j] : 0.0;
    red[0][threadIdx.x][j] = d;
    red[1][threadIdx.x][j] = d * d;
  }
  __syncthreads();
  if ((int)threadIdx.x < cout4) {   // (256 is a multiple of cout4: thread c4 of the first row owns channel group c4)
    double *sp = stat + (size_t)blockIdx.x * (8 * cout4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      double a = 0.0, b = 0.0;
      for (int q = threadIdx.x; q < 256; q += cout4) {
        a += red[0][q][j];
        b += red[1][q][j];
      }
      sp[c4 * 4 + j] = a;
      sp[4 * cout4 + c4 * 4 + j] = b;
    }
  }
}

static constexpr int kSplitTargetWaves = 4096;  // below this many waves the launch is offset-split
static bool g_conv_late = [] {            // D3D_CONV_LATE=0: gathers issued ahead of the step's matrix work (A/B runs)
  const char *e = getenv("D3D_CONV_LATE");
  return !(e && e[0] == '0');
}();

// d3d_conv_time_next: HIP events the next k_conv launch of this thread is bracketed with (measurement only)
static thread_local hipEvent_t t_time_start = nullptr, t_time_stop = nullptr;
void conv_timing_take(hipEvent_t *start, hipEvent_t *stop) {
  *start = t_time_start;
  *stop = t_time_stop;
  t_time_start = t_time_stop = nullptr;
}

struct StatOut {   // d3d_bn_prologue.out_stats*: where the launch leaves the column statistics of its output
  double *buf;
  int cap;
  int *rows_host;
};

template <int CT, int NCT, int COUT, int NT, int BPW>
static int launch_t(d3d_meta *m, const Plan &p, const float *in, int cin, const float *wp,
                    const float *residual, float *out, hipStream_t s, BnPre pre, const StatOut &so) {
  constexpr int WPBLK = COUT / 32 / NT;
  constexpr int threads = BPW * WPBLK * 64;
  const int npos = p.n_blk * 32;
  const long waves = (long)p.n_blk * WPBLK;
  int n_split = 1;
  if (BPW == 1 && p.K > 1 && m) {
    if (waves < kSplitTargetWaves) n_split = (int)std::min<long>(p.K, (kSplitTargetWaves + waves - 1) / waves);
  }
  float *partial = nullptr;
  size_t mark = 0;
  if (n_split > 1) {
    mark = m->feat_arena.used;
    partial = m->feat_arena.get<float>((size_t)n_split * npos * COUT);
    if (!partial) n_split = 1;  // no room: fall back to the unsplit launch
  }
  const dim3 grid((p.n_blk + BPW - 1) / BPW, n_split);
  // column statistics of the output for the BatchNorm that follows: one vector pair per row block, or per workgroup of
  // the reduction when the launch is offset-split
  const long reduce_blocks = ((long)npos * (COUT / 4) + 255) / 256;
  const long stat_rows = n_split > 1 ? reduce_blocks : p.n_blk;
  double *stat = (so.buf && stat_rows <= so.cap) ? so.buf : nullptr;
  if (so.rows_host) *so.rows_host = stat ? (int)stat_rows : 0;
  const uint32_t in_bytes = (uint32_t)((size_t)p.n_in * (size_t)cin * 4);   // < 4 GiB (launch_conv checks)
  const hipEvent_t ev_start = t_time_start, ev_stop = t_time_stop;
  t_time_start = t_time_stop = nullptr;
  if (ev_start) (void)hipEventRecord(ev_start, s);
  if (cin == CT * NCT && n_split == 1 && launch_conv_ws(p, in, cin, wp, COUT, residual, out, s, pre, stat, in_bytes)) {
    // (taken by the weight-sharing kernel: same products in the same order)
  } else if (cin == CT * NCT && g_conv_late && CT >= 32)
    hipLaunchKernelGGL((k_conv<CT, NCT, COUT, NT, BPW, true, true>), grid, dim3(threads), 0, s, in, cin, wp, p.nbrT, npos,
                       p.rows, p.blkmask, p.n_blk, residual, out, n_split, partial, pre, in_bytes,
                       n_split > 1 ? nullptr : stat);
  else if (cin == CT * NCT)
    hipLaunchKernelGGL((k_conv<CT, NCT, COUT, NT, BPW, true>), grid, dim3(threads), 0, s, in, cin, wp, p.nbrT, npos,
                       p.rows, p.blkmask, p.n_blk, residual, out, n_split, partial, pre, in_bytes,
                       n_split > 1 ? nullptr : stat);
  else
    hipLaunchKernelGGL((k_conv<CT, NCT, COUT, NT, BPW, false>), grid, dim3(threads), 0, s, in, cin, wp, p.nbrT, npos,
                       p.rows, p.blkmask, p.n_blk, residual, out, n_split, partial, pre, in_bytes,
                       n_split > 1 ? nullptr : stat);
  if (ev_stop) (void)hipEventRecord(ev_stop, s);   // k_conv alone: the reduction of an offset-split launch follows
  if (n_split > 1) {
    const long total = (long)npos * (COUT / 4);
    hipLaunchKernelGGL(k_conv_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, partial, n_split,
                       npos, COUT / 4, p.rows, residual, out, stat);
    m->feat_arena.used = mark;  // stream-ordered scratch
  }
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

template <int CT, int NCT>
static int launch_c(d3d_meta *m, const Plan &p, const float *in, int cin, const float *wp, int cout,
                    const float *residual, float *out, hipStream_t s, BnPre pre, const StatOut &so) {
  switch (cout) {
    case 32: return launch_t<CT, NCT, 32, 1, 4>(m, p, in, cin, wp, residual, out, s, pre, so);    // 4 independent waves
    // (NT = 2 -- half as many waves per row block, no or fewer barriers -- measured slower; DESIGN.md lists the other
    //  variants that lost to this shape: 64-row groups, streaming workgroups, LDS-free A, shared weight tile)
    case 64: return launch_t<CT, NCT, 64, 1, 1>(m, p, in, cin, wp, residual, out, s, pre, so);    // 2 waves / block
    case 128: return launch_t<CT, NCT, 128, 1, 1>(m, p, in, cin, wp, residual, out, s, pre, so);  // 4 waves / block
    case 256: return launch_t<CT, NCT, 256, 1, 1>(m, p, in, cin, wp, residual, out, s, pre, so);  // 8 waves / block
  }
  set_error("convolution: Cout=%d not supported (32, 64, 128, 256)", cout);
  return D3D_ERR_UNSUPPORTED;
}

int launch_conv(d3d_meta *m, const Plan &p, const float *in, int cin, const float *packed_w, int cout,
                const float *residual, float *out, hipStream_t s, const d3d_bn_prologue *bn) {
  if (bn && bn->out_stats_rows) *bn->out_stats_rows = 0;
  if (p.n_rows == 0) {
    if (t_time_start) (void)hipEventRecord(t_time_start, s);
    if (t_time_stop) (void)hipEventRecord(t_time_stop, s);
    t_time_start = t_time_stop = nullptr;
    return D3D_OK;
  }
  D3D_REQUIRE(in && packed_w && out, "convolution: null pointer");
  D3D_REQUIRE((size_t)p.n_in * (size_t)cin * 4 < ((size_t)1 << 32),
              "convolution: gathered tensor of %d rows x %d channels exceeds the 4 GiB of the 32-bit gather offsets", p.n_in, cin);
  BnPre pre = {nullptr, nullptr, nullptr, nullptr, 0.f};
  StatOut so = {nullptr, 0, nullptr};
  if (bn) {
    so = {bn->out_stats, bn->out_stats_cap, bn->out_stats_rows};
    if (so.rows_host) *so.rows_host = 0;
  }
  if (bn && bn->mean) {
    D3D_REQUIRE(bn->invstd && cin % 8 == 0 && padded_cin(cin) == cin, "fused BatchNorm prologue needs Cin in {32,64,128,256}");
    pre = {bn->mean, bn->invstd, bn->weight, bn->bias, bn->leakiness};
  }
  switch (padded_cin(cin)) {
    case 16: return launch_c<16, 1>(m, p, in, cin, packed_w, cout, residual, out, s, pre, so);
    case 32: return launch_c<32, 1>(m, p, in, cin, packed_w, cout, residual, out, s, pre, so);
    case 64: return launch_c<64, 1>(m, p, in, cin, packed_w, cout, residual, out, s, pre, so);
    case 128: return launch_c<128, 1>(m, p, in, cin, packed_w, cout, residual, out, s, pre, so);
    case 256: return launch_c<128, 2>(m, p, in, cin, packed_w, cout, residual, out, s, pre, so);
  }
  set_error("convolution: Cin=%d not supported (<= 256)", cin);
  return D3D_ERR_UNSUPPORTED;
}

}  // namespace d3d

using namespace d3d;

extern "C" {

int d3d_conv_late_mode(int on) {
  const int was = g_conv_late ? 1 : 0;
  if (on >= 0) g_conv_late = on != 0;
  return was;
}

int d3d_conv_time_next(void *start_event, void *stop_event) {
  t_time_start = (hipEvent_t)start_event;
  t_time_stop = (hipEvent_t)stop_event;
  return D3D_OK;
}

size_t d3d_packed_weight_floats(int fv, int cin, int cout) {
  int cp = padded_cin(cin);
  if (cp < 0 || fv <= 0 || cout <= 0) return 0;
  return (size_t)fv * cp * cout;
}

int d3d_pack_conv_weight(const float *w, int fv, int cin, int cout, float *packed, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  int cp = padded_cin(cin);
  D3D_REQUIRE(w && packed && fv > 0 && cout > 0 && cp > 0, "pack_conv_weight: bad arguments (Cin=%d)", cin);
  long total = (long)fv * cp * cout;
  hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, fv, cin, cout, cp, packed);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_subm_conv_forward(d3d_meta *m, const int *size, const int *filt, const float *in, int cin,
                          const float *packed_w, int cout, const float *residual, float *out,
                          void *stream, double *macs_host, const d3d_bn_prologue *bn) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && size && filt, "null argument");
  int rc = d3d_subm_prepare(m, size, filt, stream, nullptr);
  if (rc) return rc;
  Plan *p = const_cast<Plan *>(find_plan(m, 0, size, filt, nullptr));
  if (macs_host) {
    long nr;
    rc = plan_rules(m, *p, s, &nr);
    if (rc) return rc;
    *macs_host = (double)nr * cin * cout;
  }
  return launch_conv(m, *p, in, cin, packed_w, cout, residual, out, s, bn);
}

int d3d_conv_forward(d3d_meta *m, const int *in_size, const int *out_size, const int *filt,
                     const int *stride, const float *in, int cin, const float *packed_w, int cout,
                     float *out, void *stream, double *macs_host, const d3d_bn_prologue *bn) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && in_size && out_size && filt && stride, "null argument");
  int rc = d3d_conv_prepare(m, in_size, out_size, filt, stride, stream, nullptr, nullptr);
  if (rc) return rc;
  Plan *p = const_cast<Plan *>(find_plan(m, 1, in_size, filt, stride));
  if (macs_host) {
    long nr;
    rc = plan_rules(m, *p, s, &nr);
    if (rc) return rc;
    *macs_host = (double)nr * cin * cout;
  }
  return launch_conv(m, *p, in, cin, packed_w, cout, nullptr, out, s, bn);
}

// Deconvolution: in = coarse features, out = fine features; reuses the strided rulebook of the
// matching convolution with the roles swapped (SCN/CPU/Deconvolution.cpp:17,33-37).
int d3d_deconv_forward(d3d_meta *m, const int *in_size, const int *out_size, const int *filt,
                       const int *stride, const float *in, int cin, const float *packed_w, int cout,
                       const float *residual, float *out, void *stream, double *macs_host,
                       const d3d_bn_prologue *bn) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && in_size && out_size && filt && stride, "null argument");
  const Plan *p = nullptr;
  int rc = get_deconv_plan(m, out_size, filt, stride, s, &p);
  if (rc) return rc;
  if (macs_host) {
    long nr;
    rc = plan_rules(m, *const_cast<Plan *>(p), s, &nr);
    if (rc) return rc;
    *macs_host = (double)nr * cin * cout;
  }
  return launch_conv(m, *p, in, cin, packed_w, cout, residual, out, s, bn);
}

// ---- storage-type aware forms (d3d_dtype): D3D_F32 forwards to the functions above, D3D_BF16 runs conv_bf16.hip.
// For bf16, `cin` is the stored row width (16, 32, 64, 128 or 256 channels; narrower inputs are zero padded).
int d3d_subm_conv_forward_dt(d3d_meta *m, const int *size, const int *filt, const void *in, int cin,
                             const void *packed_w, int cout, const void *residual, void *out, int dtype, void *stream,
                             double *macs_host, const d3d_bn_prologue *bn) {
  if (dtype == D3D_F32)
    return d3d_subm_conv_forward(m, size, filt, (const float *)in, cin, (const float *)packed_w, cout,
                                 (const float *)residual, (float *)out, stream, macs_host, bn);
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(dtype == D3D_BF16 && m && size && filt, "subm_conv_forward_dt: bad arguments");
  int rc = d3d_subm_prepare(m, size, filt, stream, nullptr);
  if (rc) return rc;
  Plan *p = const_cast<Plan *>(find_plan(m, 0, size, filt, nullptr));
  if (macs_host) {
    long nr;
    rc = plan_rules(m, *p, s, &nr);
    if (rc) return rc;
    *macs_host = (double)nr * cin * cout;
  }
  return launch_conv_bf16(m, *p, in, cin, packed_w, cout, residual, out, s, bn);
}

int d3d_conv_forward_dt(d3d_meta *m, const int *in_size, const int *out_size, const int *filt, const int *stride,
                        const void *in, int cin, const void *packed_w, int cout, void *out, int dtype, void *stream,
                        double *macs_host, const d3d_bn_prologue *bn) {
  if (dtype == D3D_F32)
    return d3d_conv_forward(m, in_size, out_size, filt, stride, (const float *)in, cin, (const float *)packed_w, cout,
                            (float *)out, stream, macs_host, bn);
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(dtype == D3D_BF16 && m && in_size && out_size && filt && stride, "conv_forward_dt: bad arguments");
  int rc = d3d_conv_prepare(m, in_size, out_size, filt, stride, stream, nullptr, nullptr);
  if (rc) return rc;
  Plan *p = const_cast<Plan *>(find_plan(m, 1, in_size, filt, stride));
  if (macs_host) {
    long nr;
    rc = plan_rules(m, *p, s, &nr);
    if (rc) return rc;
    *macs_host = (double)nr * cin * cout;
  }
  return launch_conv_bf16(m, *p, in, cin, packed_w, cout, nullptr, out, s, bn);
}

int d3d_deconv_forward_dt(d3d_meta *m, const int *in_size, const int *out_size, const int *filt, const int *stride,
                          const void *in, int cin, const void *packed_w, int cout, const void *residual, void *out,
                          int dtype, void *stream, double *macs_host, const d3d_bn_prologue *bn) {
  if (dtype == D3D_F32)
    return d3d_deconv_forward(m, in_size, out_size, filt, stride, (const float *)in, cin, (const float *)packed_w, cout,
                              (const float *)residual, (float *)out, stream, macs_host, bn);
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(dtype == D3D_BF16 && m && in_size && out_size && filt && stride, "deconv_forward_dt: bad arguments");
  const Plan *p = nullptr;
  int rc = get_deconv_plan(m, out_size, filt, stride, s, &p);
  if (rc) return rc;
  if (macs_host) {
    long nr;
    rc = plan_rules(m, *const_cast<Plan *>(p), s, &nr);
    if (rc) return rc;
    *macs_host = (double)nr * cin * cout;
  }
  return launch_conv_bf16(m, *p, in, cin, packed_w, cout, residual, out, s, bn);
}

}  // extern "C"
