// Sparse-grid metadata on the device: voxel hash-scatter (input layer), strided-conv output
// grids, submanifold / strided rulebooks ("plans"), spatial locations, sparse->dense.
// Replaces the single-threaded CPU rule builders of SCN/Metadata/* (see include/d3d_hip.h).
#include <algorithm>
#include <array>
#include <climits>
#include <condition_variable>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "d3d_internal.h"

// Every kernel of this file raises its waves' issue priority: they are short chains of latency-bound steps that share
// the CUs with the convolutions of the caller's stream, whose waves hold priority 1 through their matrix loops
// (conv.hip); at equal or lower priority a geometry wave waited behind them for every instruction it issued.  Measured
// on the 500 k-point building: k_plan_finish 0.48 -> 0.33 ms, k_subm_nbr 0.32 -> 0.23 ms per building, the pass 4.84 -> 4.78 ms.
#ifndef D3D_SIDE_PRIO_LEVEL
#define D3D_SIDE_PRIO_LEVEL 3
#endif
#define D3D_SIDE_PRIO() __builtin_amdgcn_s_setprio(D3D_SIDE_PRIO_LEVEL)
namespace d3d {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ------------------------------------------------------------------------------------------
// Exclusive scan of int32: wave64 shuffle scan -> block scan -> block-sum scan -> add.
static constexpr int kScanThreads = 256;
static constexpr int kScanItems = 8;
static constexpr int kScanTile = kScanThreads * kScanItems;

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_tiles(const int32_t *__restrict__ in,
                                                             int32_t *__restrict__ out, int n,
                                                             int32_t *__restrict__ tile_sums) {
  D3D_SIDE_PRIO();
  __shared__ int wave_tot[kScanThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long base = (long)blockIdx.x * kScanTile + (long)tid * kScanItems;
  int v[kScanItems];
  int sum = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; i++) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    sum += v[i];
  }
  int incl = wave_inclusive_scan(sum, lane);
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < wave; w++) wave_off += wave_tot[w];
  int run = wave_off + incl - sum;
#pragma unroll
  for (int i = 0; i < kScanItems; i++) {
    if (base + i < n) out[base + i] = run;
    run += v[i];
  }
  if (tid == kScanThreads - 1) tile_sums[blockIdx.x] = run;
}

// (one workgroup of 4 waves: it finds a free slot beside the convolutions of another stream at once -- as 16 waves it
//  waited ~0.1 ms per call for a CU to drain in the 4 x 1 M-point step)
static constexpr int kSumThreads = 256;
__global__ __launch_bounds__(kSumThreads) void k_scan_sums(int32_t *__restrict__ sums, int nb,
                                                           int32_t *__restrict__ total) {
  D3D_SIDE_PRIO();
  __shared__ int wave_tot[kSumThreads / 64];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += kSumThreads) {
    int i = base + tid;
    int v = i < nb ? sums[i] : 0;
    int incl = wave_inclusive_scan(v, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int off = carry_s;
    for (int w = 0; w < wave; w++) off += wave_tot[w];
    if (i < nb) sums[i] = off + incl - v;
    __syncthreads();
    if (tid == kSumThreads - 1) carry_s = off + incl;
    __syncthreads();
  }
  if (tid == 0 && total) *total = carry_s;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_add(int32_t *__restrict__ out, int n,
                                                           const int32_t *__restrict__ sums) {
  D3D_SIDE_PRIO();
  const long base = (long)blockIdx.x * kScanTile + (long)threadIdx.x * kScanItems;
  const int add = sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanItems; i++)
    if (base + i < n) out[base + i] += add;
}

int scan_exclusive_i32(const int32_t *in, int32_t *out, int n, int32_t *total_dev, Arena &scratch,
                       hipStream_t s) {
  if (n <= 0) {
    if (total_dev) D3D_HIP_CHECK(hipMemsetAsync(total_dev, 0, sizeof(int32_t), s));
    return D3D_OK;
  }
  int nb = (n + kScanTile - 1) / kScanTile;
  D3D_ALLOC(sums, int32_t, scratch, nb);
  hipLaunchKernelGGL(k_scan_tiles, dim3(nb), dim3(kScanThreads), 0, s, in, out, n, sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kSumThreads), 0, s, sums, nb, total_dev);
  hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(kScanThreads), 0, s, out, n, sums);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

// ------------------------------------------------------------------------------------------
// Stable LSD radix sort of (uint32 key, int32 value) pairs by the low `bits` bits of the key: three launches per pass
// and no cross-workgroup waiting -- per-tile digit histograms -> per-digit scan over the tiles -> ranked scatter -- with
// 8-, 9- or 10-bit digits, whichever covers `bits` in the fewest passes (27 bits: 3 x 9; 19 bits: 2 x 10; 8 bits: one).
// rocPRIM's device sort takes ~20 dependent launches of 5-7 us (block sort + merge passes) for the 10^4 .. 5*10^5 rows of
// a rulebook -- 110-190 us, almost all launch latency -- and its Onesweep variant 25-33 us per pass at these sizes (the
// look-back chain over a few dozen tiles is serial); this one is bound by its 3 launches per pass (~17 us).
// Descending order sorts the complemented digits, so it is stable too.
static constexpr int kRsThreads = 256;
static constexpr int kRsItems = 8;
static constexpr int kRsTile = kRsThreads * kRsItems;

template <int DB>
__device__ __forceinline__ uint32_t rs_digit(uint32_t k, int shift, bool desc) {
  return ((desc ? ~k : k) >> shift) & ((1u << DB) - 1u);
}

// hist[digit][tile] of one pass
// n_dev (may be null): the element count on the device, for a sort enqueued with an upper bound `n` (and its n_tiles)
template <int DB>
__global__ __launch_bounds__(kRsThreads) void k_rs_count(const uint32_t *__restrict__ keys, int n, int n_tiles, int shift,
                                                         int desc, uint32_t *__restrict__ hist,
                                                         const int32_t *__restrict__ n_dev) {
  D3D_SIDE_PRIO();
  constexpr int BINS = 1 << DB;
  __shared__ uint32_t h[BINS];
  if (n_dev) n = *n_dev;
  for (int b = threadIdx.x; b < BINS; b += kRsThreads) h[b] = 0;
  __syncthreads();
  const int base = blockIdx.x * kRsTile;
#pragma unroll
  for (int j = 0; j < kRsItems; j++) {
    const int i = base + j * kRsThreads + threadIdx.x;
    if (i < n) atomicAdd(&h[rs_digit<DB>(keys[i], shift, desc != 0)], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < BINS; b += kRsThreads) hist[(size_t)b * n_tiles + blockIdx.x] = h[b];
}

// one workgroup per digit: exclusive scan of its counts over the tiles (in place) and the digit's total
__global__ __launch_bounds__(kRsThreads) void k_rs_scan(uint32_t *__restrict__ hist, int n_tiles,
                                                        uint32_t *__restrict__ totals) {
  D3D_SIDE_PRIO();
  __shared__ uint32_t wave_tot[kRsThreads / 64];
  __shared__ uint32_t carry_s;
  uint32_t *row = hist + (size_t)blockIdx.x * n_tiles;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n_tiles; base += kRsThreads) {
    const int i = base + tid;
    const uint32_t v = i < n_tiles ? row[i] : 0u;
    const uint32_t incl = (uint32_t)wave_inclusive_scan((int)v, lane);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t off = carry_s;
    for (int w = 0; w < wave; w++) off += wave_tot[w];
    if (i < n_tiles) row[i] = off + incl - v;
    __syncthreads();
    if (tid == kRsThreads - 1) carry_s = off + incl;
    __syncthreads();
  }
  if (tid == 0) totals[blockIdx.x] = carry_s;
}

// Ranked scatter of one tile.  Wave w owns the tile's elements [512 w, 512 w + 512) and walks them in 8 rounds of 64
// consecutive ones; the lanes of a round that hold the same digit find each other with DB ballots, take consecutive
// ranks behind the wave's running count of that digit, and the waves' counts are chained in wave order: the rank of an
// element is the number of elements of its digit before it in the tile, i.e. the pass is stable.
template <int DB>
__global__ __launch_bounds__(kRsThreads) void k_rs_scatter(const uint32_t *__restrict__ keys_in,
                                                           const int32_t *__restrict__ vals_in, int n, int n_tiles,
                                                           int shift, int desc, const uint32_t *__restrict__ hist,
                                                           const uint32_t *__restrict__ totals,
                                                           uint32_t *__restrict__ keys_out,
                                                           int32_t *__restrict__ vals_out,
                                                           const int32_t *__restrict__ n_dev) {
  D3D_SIDE_PRIO();
  constexpr int NW = kRsThreads / 64, BINS = 1 << DB, PER = BINS / kRsThreads;
  if (n_dev) n = *n_dev;
  __shared__ uint32_t wcnt[NW][BINS];
  __shared__ uint32_t wave_tot[NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int b = tid; b < NW * BINS; b += kRsThreads) (&wcnt[0][0])[b] = 0;
  __syncthreads();
  const int base = blockIdx.x * kRsTile + wave * (64 * kRsItems);
  uint32_t key[kRsItems], dig[kRsItems], rank[kRsItems];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < kRsItems; r++) {
    const int i = base + r * 64 + lane;
    const bool ok = i < n;
    key[r] = ok ? keys_in[i] : 0u;
    const uint32_t d = rs_digit<DB>(key[r], shift, desc != 0);
    dig[r] = d;
    unsigned long long m = __ballot(ok);
#pragma unroll
    for (int b = 0; b < DB; b++) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    // every lane of a group reads the count before the group's first lane adds to it (LDS serves a wave in order)
    const uint32_t before = ok ? wcnt[wave][d] : 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    rank[r] = before + (uint32_t)__popcll(m & lt);
    if (ok && (m & lt) == 0) wcnt[wave][d] = before + (uint32_t)__popcll(m);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();
  // thread t, digits PER t .. PER t + PER - 1: where a digit of this tile starts = the digits before it (all tiles) +
  // the same digit in the tiles before this one, then the waves in order
  uint32_t pre[PER], sum = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    pre[j] = sum;
    sum += totals[tid * PER + j];
  }
  const uint32_t incl = (uint32_t)wave_inclusive_scan((int)sum, lane);
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t first = incl - sum;
  for (int w = 0; w < wave; w++) first += wave_tot[w];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int d = tid * PER + j;
    uint32_t off = first + pre[j] + hist[(size_t)d * n_tiles + blockIdx.x];
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const uint32_t c = wcnt[w][d];
      wcnt[w][d] = off;
      off += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRsItems; r++) {
    const int i = base + r * 64 + lane;
    if (i < n) {
      const uint32_t pos = wcnt[wave][dig[r]] + rank[r];
      if (keys_out) keys_out[pos] = key[r];
      vals_out[pos] = vals_in[i];
    }
  }
}

template <int DB>
static void rs_pass(const uint32_t *ksrc, const int32_t *vsrc, int n, int n_tiles, int shift, bool desc, uint32_t *hist,
                    uint32_t *totals, uint32_t *kdst, int32_t *vdst, hipStream_t s, const int32_t *n_dev) {
  hipLaunchKernelGGL(k_rs_count<DB>, dim3(n_tiles), dim3(kRsThreads), 0, s, ksrc, n, n_tiles, shift, desc ? 1 : 0, hist,
                     n_dev);
  hipLaunchKernelGGL(k_rs_scan, dim3(1 << DB), dim3(kRsThreads), 0, s, hist, n_tiles, totals);
  hipLaunchKernelGGL(k_rs_scatter<DB>, dim3(n_tiles), dim3(kRsThreads), 0, s, ksrc, vsrc, n, n_tiles, shift, desc ? 1 : 0,
                     hist, totals, kdst, vdst, n_dev);
}
static inline void rs_layout(int bits, int &passes, int &db) {
  passes = std::max(1, (bits + 9) / 10);
  db = std::min(10, std::max(8, (bits + passes - 1) / passes));
}
size_t sort_scratch_bytes(int n, int bits) {
  int passes, db;
  rs_layout(bits, passes, db);
  const size_t tiles = ((size_t)std::max(n, 1) + kRsTile - 1) / kRsTile;
  return (((size_t)1 << db) * (tiles + 1)) * 4 + 4 * ((size_t)std::max(n, 1) * 4 + 256) + 4096;
}

// keys_out may be null: the sorted keys are not wanted (saves the last pass's key stores).  `scratch` provides the
// intermediate buffers and the histograms (released by the caller's mark).
int sort_pairs_u32(const uint32_t *keys_in, uint32_t *keys_out, const int32_t *vals_in,
                   int32_t *vals_out, int n, int bits, Arena &scratch, hipStream_t s, bool descending,
                   const int32_t *n_dev) {
  if (n <= 0) return D3D_OK;
  int passes, db;
  rs_layout(bits, passes, db);
  const int n_tiles = (n + kRsTile - 1) / kRsTile;
  D3D_ALLOC(hist, uint32_t, scratch, ((size_t)n_tiles + 1) << db);
  uint32_t *totals = hist + ((size_t)n_tiles << db);
  uint32_t *ktmp[2] = {nullptr, nullptr};
  int32_t *vtmp[2] = {nullptr, nullptr};
  for (int j = 0; j < std::min(2, passes - 1); j++) {
    D3D_ALLOC(kt, uint32_t, scratch, n);
    D3D_ALLOC(vt, int32_t, scratch, n);
    ktmp[j] = kt;
    vtmp[j] = vt;
  }
  const uint32_t *ksrc = keys_in;
  const int32_t *vsrc = vals_in;
  for (int p = 0; p < passes; p++) {
    const bool last = p == passes - 1;
    uint32_t *kdst = last ? keys_out : ktmp[p & 1];
    int32_t *vdst = last ? vals_out : vtmp[p & 1];
    if (db == 8) rs_pass<8>(ksrc, vsrc, n, n_tiles, db * p, descending, hist, totals, kdst, vdst, s, n_dev);
    else if (db == 9) rs_pass<9>(ksrc, vsrc, n, n_tiles, db * p, descending, hist, totals, kdst, vdst, s, n_dev);
    else rs_pass<10>(ksrc, vsrc, n, n_tiles, db * p, descending, hist, totals, kdst, vdst, s, n_dev);
    ksrc = kdst;
    vsrc = vdst;
  }
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

__global__ void k_iota(int32_t *p, int n) {
  D3D_SIDE_PRIO();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = i;
}

struct CapGuard {  // temporarily shrinks the arena so that a raw table can sit at its far end
  Arena &a;
  size_t save;
  CapGuard(Arena &a_, size_t new_cap) : a(a_), save(a_.cap) { a.cap = new_cap; }
  ~CapGuard() { a.cap = save; }
};
static inline int next_pow2(long v) {
  long c = 1024;
  while (c < v) c <<= 1;
  return (int)c;
}
static inline dim3 grid1d(long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

// 0xFF fill of hash tables and rulebook arrays.  hipMemsetAsync's fill kernel reaches ~0.3 TB/s on the 12-16 MB arrays
// of the fine levels (53 us for a 16 MB table); 16-byte stores from enough workgroups run at HBM speed.
__global__ __launch_bounds__(256) void k_fill_ones(uint4 *__restrict__ p, size_t n16) {
  D3D_SIDE_PRIO();
  const uint4 v = make_uint4(~0u, ~0u, ~0u, ~0u);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
static hipError_t fill_ones(void *ptr, size_t bytes, hipStream_t s) {
  if (bytes < ((size_t)1 << 18) || ((uintptr_t)ptr & 15)) return hipMemsetAsync(ptr, 0xFF, bytes, s);
  const size_t n16 = bytes / 16;
  const unsigned blocks = (unsigned)std::min<size_t>((n16 + 1023) / 1024, 256 * 16);
  hipLaunchKernelGGL(k_fill_ones, dim3(blocks), dim3(256), 0, s, (uint4 *)ptr, n16);
  if (bytes & 15) return hipMemsetAsync((char *)ptr + n16 * 16, 0xFF, bytes & 15, s);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// a2. Input layer: hash insert + first-occurrence numbering (IOLayersRules.h:72-95).
// ext (may be null): device int32[4] = 1 + the largest x, y, z and example index over all points (zero-initialised by the
// caller): one atomicMax per dimension and 256-thread block.  The host reads it back with the site count and bounds the
// site counts of the coarser grids with it (grid chain below), so that no further count has to be read back.
__global__ __launch_bounds__(256) void k_insert_points(const int64_t *__restrict__ coords, int n, int ncols,
                                                       HashEntry *tab, int cap, int32_t *pslot, int32_t *ext) {
  D3D_SIDE_PRIO();
  __shared__ int red[4][4];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int v[4] = {0, 0, 0, 0};
  if (i < n) {
    const int64_t *c = coords + (size_t)i * ncols;
    int b = ncols == 4 ? (int)c[3] : 0;
    int slot = hash_insert(tab, cap, pack_key(b, (int)c[0], (int)c[1], (int)c[2]));
    atomicMin(&tab[slot].first, (uint32_t)i);
    pslot[i] = slot;
    v[0] = (int)c[0] + 1;
    v[1] = (int)c[1] + 1;
    v[2] = (int)c[2] + 1;
    v[3] = b + 1;
  }
  if (!ext) return;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    int m = v[d];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][d] = m;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int d = threadIdx.x;
    const int m = max(max(red[0][d], red[1][d]), max(red[2][d], red[3][d]));
    if (m > 0) atomicMax(&ext[d], m);
  }
}
__global__ void k_flag_first(const int32_t *__restrict__ pslot, const HashEntry *__restrict__ tab,
                             int n, int32_t *flag) {
  D3D_SIDE_PRIO();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (pslot[i] >= 0 && tab[pslot[i]].first == (uint32_t)i) ? 1 : 0;
}
__global__ void k_assign_input_sites(const int64_t *__restrict__ coords, int n, int ncols,
                                     const int32_t *__restrict__ pslot,
                                     const int32_t *__restrict__ flag,
                                     const int32_t *__restrict__ rank, HashEntry *tab, int32_t *loc) {
  D3D_SIDE_PRIO();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  int id = rank[i];
  tab[pslot[i]].val = id;
  const int64_t *c = coords + (size_t)i * ncols;
  loc[id * 4 + 0] = (int)c[0];
  loc[id * 4 + 1] = (int)c[1];
  loc[id * 4 + 2] = (int)c[2];
  loc[id * 4 + 3] = ncols == 4 ? (int)c[3] : 0;
}
__global__ void k_point_site(const int32_t *__restrict__ pslot, const HashEntry *__restrict__ tab,
                             int n, uint32_t *psite, int32_t *cnt) {
  D3D_SIDE_PRIO();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int s = tab[pslot[i]].val;
  psite[i] = (uint32_t)s;
  atomicAdd(&cnt[s], 1);
}
// a3. CPU/IOLayers.cpp:11-29: out[row] += mult * in[idx] in input order.
__global__ void k_input_forward(const float *__restrict__ in, int planes,
                                const int32_t *__restrict__ off, const int32_t *__restrict__ idx,
                                int n_active, int average, float *__restrict__ out) {
  D3D_SIDE_PRIO();
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n_active * planes) return;
  int row = (int)(t / planes), c = (int)(t % planes);
  int b = off[row], e = off[row + 1];
  int cnt = e - b;
  float mult = (average && cnt > 0) ? (float)1 / cnt : (float)1;
  float acc = 0.f;
  for (int j = b; j < e; j++) acc += mult * in[(size_t)idx[j] * planes + c];
  out[t] = acc;
}

// ------------------------------------------------------------------------------------------
// a5. Strided convolution: output grid + raw neighbour tables (ConvolutionRules.h:12-34).
struct ConvGeom {
  int filt[3], stride[3], out_size[3];
  int max_out;
};
__device__ __forceinline__ bool conv_entry(const ConvGeom &g, const int32_t *p, int j, int *o,
                                           int *off) {
  int lb[3], cnt[3];
#pragma unroll
  for (int d = 0; d < 3; d++) {
    int t = p[d] - g.filt[d] + g.stride[d];
    int l = t < 0 ? 0 : t / g.stride[d];          // max(0, (in - size + stride) / stride)
    int u = min(g.out_size[d] - 1, p[d] / g.stride[d]);
    lb[d] = l;
    cnt[d] = u - l + 1;
    if (cnt[d] <= 0) return false;
  }
  if (j >= cnt[0] * cnt[1] * cnt[2]) return false;
  int jz = j % cnt[2], jy = (j / cnt[2]) % cnt[1], jx = j / (cnt[2] * cnt[1]);
  o[0] = lb[0] + jx;
  o[1] = lb[1] + jy;
  o[2] = lb[2] + jz;
  *off = ((p[0] - o[0] * g.stride[0]) * g.filt[1] + (p[1] - o[1] * g.stride[1])) * g.filt[2] +
         (p[2] - o[2] * g.stride[2]);
  return true;
}
// n_in_dev (may be null): the input site count on the device, for a launch sized by an upper bound of it
__global__ void k_conv_insert(const int32_t *__restrict__ loc, long n_entries, ConvGeom g,
                              HashEntry *tab, int cap, int32_t *eslot, const int32_t *__restrict__ n_in_dev) {
  D3D_SIDE_PRIO();
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_in_dev) n_entries = (long)*n_in_dev * g.max_out;
  if (e >= n_entries) return;
  int i = (int)(e / g.max_out), j = (int)(e % g.max_out);
  const int32_t *p = loc + (size_t)i * 4;
  int o[3], off;
  if (!conv_entry(g, p, j, o, &off)) {
    eslot[e] = -1;
    return;
  }
  int slot = hash_insert(tab, cap, pack_key(p[3], o[0], o[1], o[2]));
  atomicMin(&tab[slot].first, (uint32_t)e);
  eslot[e] = slot;
}
__global__ void k_conv_assign(const int32_t *__restrict__ loc, long n_entries, ConvGeom g,
                              const int32_t *__restrict__ eslot, const int32_t *__restrict__ flag,
                              const int32_t *__restrict__ rank, HashEntry *tab, int32_t *loc_out) {
  D3D_SIDE_PRIO();
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_entries || !flag[e]) return;
  int i = (int)(e / g.max_out), j = (int)(e % g.max_out);
  const int32_t *p = loc + (size_t)i * 4;
  int o[3], off;
  conv_entry(g, p, j, o, &off);
  int id = rank[e];
  tab[eslot[e]].val = id;
  loc_out[id * 4 + 0] = o[0];
  loc_out[id * 4 + 1] = o[1];
  loc_out[id * 4 + 2] = o[2];
  loc_out[id * 4 + 3] = p[3];
}
// nbr_dec may be null (no deconvolution / backward view of this rulebook will be asked for)
__global__ void k_conv_fill(const int32_t *__restrict__ loc, long n_entries, ConvGeom g, int K,
                            const int32_t *__restrict__ eslot, const HashEntry *__restrict__ tab,
                            int32_t *nbr_fwd, int32_t *nbr_dec, const int32_t *__restrict__ n_in_dev) {
  D3D_SIDE_PRIO();
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_in_dev) n_entries = (long)*n_in_dev * g.max_out;
  if (e >= n_entries || eslot[e] < 0) return;
  int i = (int)(e / g.max_out), j = (int)(e % g.max_out);
  const int32_t *p = loc + (size_t)i * 4;
  int o[3], off;
  conv_entry(g, p, j, o, &off);
  int oid = tab[eslot[e]].val;
  nbr_fwd[(size_t)oid * K + off] = i;
  if (nbr_dec) nbr_dec[(size_t)i * K + off] = oid;
}

// First-touch numbering of a strided grid's output sites without a scan over the entries: a tile of kChainTile entries
// leaves its first-touch flags as 32 ballot words and their number (k_chain_flag); k_chain_assign then sums the counts of
// the tiles before its own (a few hundred integers, no cross-workgroup waiting), ranks its flags with popcounts and writes
// site ids, coordinates and -- the last tile -- the site count, all on the device.  Two launches for what k_flag_first,
// the three scan kernels and k_conv_assign did in five, and nothing the host has to read before the next level starts.
static constexpr int kChainTile = 2048;   // = 256 threads x 8 rounds; word w of a tile = its entries 64 w .. 64 w + 63
__global__ __launch_bounds__(256) void k_chain_flag(const int32_t *__restrict__ eslot, const HashEntry *__restrict__ tab,
                                                    long n_entries, int max_out, const int32_t *__restrict__ n_in_dev,
                                                    unsigned long long *__restrict__ flagbits,
                                                    int32_t *__restrict__ tile_cnt) {
  D3D_SIDE_PRIO();
  __shared__ int wcnt[4];
  if (n_in_dev) n_entries = (long)*n_in_dev * max_out;
  const long base = (long)blockIdx.x * kChainTile;
  if (base >= n_entries) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int c = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int w = j * 4 + wave;
    const long e = base + w * 64 + lane;
    bool f = false;
    if (e < n_entries) {
      const int sl = eslot[e];
      f = sl >= 0 && tab[sl].first == (uint32_t)e;
    }
    const unsigned long long bal = __ballot(f);
    if (lane == 0) flagbits[(size_t)blockIdx.x * 32 + w] = bal;
    c += __popcll(bal);
  }
  if (lane == 0) wcnt[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}
__global__ __launch_bounds__(256) void k_chain_assign(const int32_t *__restrict__ loc, long n_entries, ConvGeom g,
                                                      const int32_t *__restrict__ n_in_dev,
                                                      const int32_t *__restrict__ eslot,
                                                      const unsigned long long *__restrict__ flagbits,
                                                      const int32_t *__restrict__ tile_cnt, HashEntry *tab,
                                                      int32_t *__restrict__ loc_out, int32_t *__restrict__ n_out_dev) {
  D3D_SIDE_PRIO();
  __shared__ int red[4];
  __shared__ int wpre[33];
  __shared__ unsigned long long words[32];
  if (n_in_dev) n_entries = (long)*n_in_dev * g.max_out;
  if (n_entries <= 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out_dev = 0;
    return;
  }
  const int n_tiles = (int)((n_entries + kChainTile - 1) / kChainTile);
  if ((int)blockIdx.x >= n_tiles) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int part = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += 256) part += tile_cnt[t];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) red[wave] = part;
  if (threadIdx.x < 32) {
    const unsigned long long w = flagbits[(size_t)blockIdx.x * 32 + threadIdx.x];
    words[threadIdx.x] = w;
    const int pc = __popcll(w);
    int inc = pc;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    wpre[threadIdx.x] = inc - pc;
    if (threadIdx.x == 31) wpre[32] = inc;
  }
  __syncthreads();
  const int off = red[0] + red[1] + red[2] + red[3];
  const long base = (long)blockIdx.x * kChainTile;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int w = j * 4 + wave;
    const unsigned long long word = words[w];
    if (!((word >> lane) & 1ull)) continue;
    const long e = base + w * 64 + lane;
    const int id = off + wpre[w] + __popcll(word & ((1ull << lane) - 1ull));
    const int i = (int)(e / g.max_out), jj = (int)(e % g.max_out);
    const int32_t *p = loc + (size_t)i * 4;
    int o[3], offk;
    conv_entry(g, p, jj, o, &offk);
    tab[eslot[e]].val = id;
    loc_out[(size_t)id * 4 + 0] = o[0];
    loc_out[(size_t)id * 4 + 1] = o[1];
    loc_out[(size_t)id * 4 + 2] = o[2];
    loc_out[(size_t)id * 4 + 3] = p[3];
  }
  if ((int)blockIdx.x == n_tiles - 1 && threadIdx.x == 0) *n_out_dev = off + wpre[32];
}
// 0xFF fill of up to kFillSegs memory ranges in one launch (blockIdx.y = range)
static constexpr int kFillSegs = 48;
struct FillSegs {
  uint4 *ptr[kFillSegs];
  unsigned long long n16[kFillSegs];
};
__global__ __launch_bounds__(256) void k_fill_ones_multi(FillSegs f) {
  D3D_SIDE_PRIO();
  const uint4 v = make_uint4(~0u, ~0u, ~0u, ~0u);
  uint4 *__restrict__ p = f.ptr[blockIdx.y];
  const size_t n = f.n16[blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// the site counts of the chain's levels -> a pinned host array (one system-scope release store each)
static constexpr int kChainMax = 24;
struct CountPtrs {
  const int32_t *p[kChainMax];
  int n;
};
__global__ void k_store_counts(CountPtrs c, int32_t *__restrict__ host) {
  D3D_SIDE_PRIO();
  const int i = threadIdx.x;
  if (i < c.n) __hip_atomic_store(&host[i], *c.p[i], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The whole strided-grid build of a SMALL level in one single-workgroup launch: table + rulebook initialisation,
// insertion, first-touch flags, their exclusive scan in entry order, site numbering, and the forward / decoded
// rulebook fill -- what the large path spreads over 11 launches (3 fills, k_conv_insert, k_flag_first, 3 scan kernels,
// k_conv_assign, k_conv_fill).  The coarse pyramid levels are pure latency (each launch a dependent step of the chain
// the host waits on for the site count), so fewer steps is what matters; the results are the large path's, bit for bit.
// Threads keep their entries' slots in registers; table fields other threads wrote are read with agent-scope atomic
// loads (L2), never through a possibly stale L1 line.
static constexpr int kSmallGrid = 4096;                        // entries one workgroup takes (16 k: slower than the 11 launches)
static constexpr int kSmallGridThreads = 256;                  // 4 waves: finds a slot on a busy CU (see k_plan_small)
static constexpr int kSmallGridEPT = kSmallGrid / kSmallGridThreads;
// n_in_dev (may be null): the input site count on the device (n_entries / n_in are then upper bounds; the grid chain);
// prefilled: table and rulebook arrays already hold 0xFF (the chain's one fill launch); nbr_dec may be null.
__global__ __launch_bounds__(kSmallGridThreads) void k_conv_grid_small(const int32_t *__restrict__ loc, int n_entries, ConvGeom g,
                                                          int K, int n_in, HashEntry *tab, int cap,
                                                          int32_t *__restrict__ loc_out, int32_t *__restrict__ nbr_fwd,
                                                          int32_t *__restrict__ nbr_dec, int32_t *__restrict__ total,
                                                          const int32_t *__restrict__ n_in_dev, int prefilled) {
  D3D_SIDE_PRIO();
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  __shared__ int wsum[kSmallGridThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (n_in_dev) {
    n_in = *n_in_dev;
    n_entries = n_in * g.max_out;
  }
  if (!prefilled) {
    const u32x4 ones = {~0u, ~0u, ~0u, ~0u};
    for (int i = tid; i < cap; i += kSmallGridThreads) *(u32x4 *)&tab[i] = ones;
    if (nbr_dec)
      for (int i = tid; i < n_in * K + 1; i += kSmallGridThreads) nbr_dec[i] = -1;
    for (int i = tid; i < n_entries * K + 1; i += kSmallGridThreads) nbr_fwd[i] = -1;      // n_out <= n_entries rows are read later
    __syncthreads();   // every wave's stores have reached L2 ...
    if (tid == 0) __threadfence();   // ... one agent-scope fence for the workgroup (as in k_bn_stats)
    __syncthreads();
  }
  int slot_r[kSmallGridEPT];
#pragma unroll
  for (int it = 0; it < kSmallGridEPT; it++) {
    const int e = it * kSmallGridThreads + tid;
    int slot = -1;
    if (e < n_entries) {
      const int32_t *p = loc + (size_t)(e / g.max_out) * 4;
      int o[3], off;
      if (conv_entry(g, p, e % g.max_out, o, &off)) {
        slot = hash_insert(tab, cap, pack_key(p[3], o[0], o[1], o[2]));
        atomicMin(&tab[slot].first, (uint32_t)e);
      }
    }
    slot_r[it] = slot;
  }
  __syncthreads();   // every wave's stores have reached L2 ...
  if (tid == 0) __threadfence();   // ... one agent-scope fence for the workgroup (as in k_bn_stats)
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int it = 0; it < kSmallGridEPT; it++) {
    if (it * kSmallGridThreads >= n_entries) break;                                       // block-uniform
    const int e = it * kSmallGridThreads + tid, slot = slot_r[it];
    const bool first = slot >= 0 &&
                       __hip_atomic_load(&tab[slot].first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)e;
    const unsigned long long bal = __ballot(first);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kSmallGridThreads / 64; w++) {
      const int c = wsum[w];
      before += w < wave ? c : 0;
      all += c;
    }
    if (first) {
      const int id = base + before + __popcll(bal & ((1ull << lane) - 1ull));
      const int32_t *p = loc + (size_t)(e / g.max_out) * 4;
      int o[3], off;
      conv_entry(g, p, e % g.max_out, o, &off);
      __hip_atomic_store(&tab[slot].val, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      loc_out[id * 4 + 0] = o[0];
      loc_out[id * 4 + 1] = o[1];
      loc_out[id * 4 + 2] = o[2];
      loc_out[id * 4 + 3] = p[3];
    }
    base += all;
    __syncthreads();
  }
  if (tid == 0) *total = base;
  __syncthreads();   // every wave's stores have reached L2 ...
  if (tid == 0) __threadfence();   // ... one agent-scope fence for the workgroup (as in k_bn_stats)
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kSmallGridEPT; it++) {
    const int e = it * kSmallGridThreads + tid, slot = slot_r[it];
    if (slot < 0) continue;
    const int i = e / g.max_out;
    const int32_t *p = loc + (size_t)i * 4;
    int o[3], off;
    conv_entry(g, p, e % g.max_out, o, &off);
    const int oid = __hip_atomic_load(&tab[slot].val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    nbr_fwd[(size_t)oid * K + off] = i;
    if (nbr_dec) nbr_dec[(size_t)i * K + off] = oid;
  }
}

// a4. Submanifold neighbour probes (SubmanifoldConvolutionRules.h:13-45).  A 256-thread block owns 64
// output sites; its 64*K probes are spread over the threads (independent loads in flight), the [site][k]
// table is written coalesced, and the per-site offset masks are assembled in LDS.
// sort key of a plan row: (number of offsets << K) | offset mask -- rows with equal masks stay together and blocks come
// out ordered by weight (sorted descending: the heaviest blocks are dispatched first); the mask is its low K bits
__device__ __forceinline__ uint32_t plan_key(uint32_t m, int K) { return K <= 27 ? (((uint32_t)__popc(m) << K) | m) : m; }
__device__ __forceinline__ uint32_t plan_key_mask(uint32_t key, int K) { return K <= 27 ? (key & ((1u << K) - 1u)) : key; }

static constexpr int kNbrSites = 64;
// n_dev (may be null): the site count on the device, for a launch sized by an upper bound before the count is known
__global__ __launch_bounds__(256) void k_subm_nbr(const int32_t *__restrict__ loc, int n, int fx, int fy,
                                                  int fz, const HashEntry *__restrict__ tab, int cap,
                                                  int32_t *__restrict__ nbr, uint32_t *__restrict__ mask,
                                                  const int32_t *__restrict__ n_dev) {
  D3D_SIDE_PRIO();
  __shared__ uint32_t smask[kNbrSites];
  __shared__ int32_t sloc[kNbrSites * 4];
  const int K = fx * fy * fz;
  const int s0 = blockIdx.x * kNbrSites;
  if (n_dev) n = *n_dev;
  if (s0 >= n) return;
  const int ns = min(kNbrSites, n - s0);
  if (threadIdx.x < kNbrSites) smask[threadIdx.x] = 0;
  for (int e = threadIdx.x; e < ns * 4; e += 256) sloc[e] = loc[(size_t)s0 * 4 + e];
  __syncthreads();
  for (int e = threadIdx.x; e < ns * K; e += 256) {
    const int ls = e / K, k = e % K;
    const int dz = k % fz, dy = (k / fz) % fy, dx = k / (fz * fy);
    const int x = sloc[ls * 4] - fx / 2 + dx, y = sloc[ls * 4 + 1] - fy / 2 + dy, z = sloc[ls * 4 + 2] - fz / 2 + dz;
    int v = -1;
    if (x >= 0 && y >= 0 && z >= 0) v = hash_find(tab, cap, pack_key(sloc[ls * 4 + 3], x, y, z));
    nbr[(size_t)s0 * K + e] = v;
    if (v >= 0) atomicOr(&smask[ls], 1u << k);
  }
  __syncthreads();
  // the row's SORT KEY (popcount above the mask, plan_key): finalize_plan sorts it as it is
  if (threadIdx.x < ns) mask[s0 + threadIdx.x] = plan_key(smask[threadIdx.x], K);
}

// The same table for an odd-sized filter with HALF the probes: site j is the neighbour of site i at offset k exactly when
// i is the neighbour of j at the mirrored offset K-1-k, so only the offsets before the centre are probed and every hit is
// written twice (nbr[i][k] = j, nbr[j][K-1-k] = i).  `nbr` is pre-filled with -1 and `mask` with 0 by the caller (the
// offsets after the centre of a site are written only by their partners); the masks are OR-ed together with atomics
// (order independent).  On the fine levels 9 of 10 probes find nothing, so the second writes are few: 12.5 M probes
// become 6 M at level 0 for 0.5 M scattered stores.
__global__ __launch_bounds__(256) void k_subm_nbr_sym(const int32_t *__restrict__ loc, int n, int fx, int fy, int fz,
                                                      const HashEntry *__restrict__ tab, int cap,
                                                      int32_t *__restrict__ nbr, uint32_t *__restrict__ mask,
                                                      const int32_t *__restrict__ n_dev) {
  D3D_SIDE_PRIO();
  __shared__ uint32_t smask[kNbrSites];
  __shared__ int32_t sloc[kNbrSites * 4];
  const int K = fx * fy * fz, H = K / 2;       // offsets 0 .. H-1 are probed, H is the site itself
  const int s0 = blockIdx.x * kNbrSites;
  if (n_dev) n = *n_dev;
  if (s0 >= n) return;
  const int ns = min(kNbrSites, n - s0);
  if (threadIdx.x < kNbrSites) smask[threadIdx.x] = 0;
  for (int e = threadIdx.x; e < ns * 4; e += 256) sloc[e] = loc[(size_t)s0 * 4 + e];
  __syncthreads();
  for (int e = threadIdx.x; e < ns * (H + 1); e += 256) {
    const int ls = e / (H + 1), k = e % (H + 1);
    const int self = s0 + ls;
    if (k == H) {                                // the centre offset
      nbr[(size_t)self * K + H] = self;
      atomicOr(&smask[ls], 1u << H);
      continue;
    }
    const int dz = k % fz, dy = (k / fz) % fy, dx = k / (fz * fy);
    const int x = sloc[ls * 4] - fx / 2 + dx, y = sloc[ls * 4 + 1] - fy / 2 + dy, z = sloc[ls * 4 + 2] - fz / 2 + dz;
    int v = -1;
    if (x >= 0 && y >= 0 && z >= 0) v = hash_find(tab, cap, pack_key(sloc[ls * 4 + 3], x, y, z));
    if (v >= 0) {
      nbr[(size_t)self * K + k] = v;
      nbr[(size_t)v * K + (K - 1 - k)] = self;
      atomicOr(&smask[ls], 1u << k);
      atomicOr(&mask[v], 1u << (K - 1 - k));
    }
  }
  __syncthreads();
  if (threadIdx.x < ns) atomicOr(&mask[s0 + threadIdx.x], smask[threadIdx.x]);
}
// The neighbour table of a submanifold rulebook: nbr[n][K] and the rows' offset masks (the low K bits of the sort key).
// The half-probe form pays two fills: measured rulebook builds 0.285 -> 0.230 ms at 462 k sites, 0.245 -> 0.208 at 371 k,
// 0.113 -> 0.130 at 193 k, 0.084 -> 0.104 at 57 k.
static constexpr int kSymMinSites = 262144;
static bool subm_nbr_is_sym(int n_bound, const int *filt) {
  const int K = filt[0] * filt[1] * filt[2];
  return (filt[0] & 1) && (filt[1] & 1) && (filt[2] & 1) && K > 1 && n_bound >= kSymMinSites;
}
// the two fills the half-probe form needs (table -1, masks 0); a caller may run them ahead of time on another stream
static int subm_nbr_prefill(int n_bound, const int *filt, int32_t *nbr, uint32_t *mask, hipStream_t s) {
  const int K = filt[0] * filt[1] * filt[2];
  D3D_HIP_CHECK(fill_ones(nbr, sizeof(int32_t) * ((size_t)n_bound * K + 1), s));
  D3D_HIP_CHECK(hipMemsetAsync(mask, 0, sizeof(uint32_t) * (size_t)n_bound, s));
  return D3D_OK;
}
static int launch_subm_nbr(const int32_t *loc, int n_bound, const int *filt, const HashEntry *tab, int cap, int32_t *nbr,
                           uint32_t *mask, const int32_t *n_dev, hipStream_t s, bool prefilled = false) {
  if (n_bound <= 0) return D3D_OK;
  const int K = filt[0] * filt[1] * filt[2];
  if (subm_nbr_is_sym(n_bound, filt)) {
    if (!prefilled)
      if (int rc = subm_nbr_prefill(n_bound, filt, nbr, mask, s)) return rc;
    hipLaunchKernelGGL(k_subm_nbr_sym, grid1d(n_bound, kNbrSites), dim3(256), 0, s, loc, n_bound, filt[0], filt[1], filt[2],
                       tab, cap, nbr, mask, n_dev);
  } else {
    hipLaunchKernelGGL(k_subm_nbr, grid1d(n_bound, kNbrSites), dim3(256), 0, s, loc, n_bound, filt[0], filt[1], filt[2], tab,
                       cap, nbr, mask, n_dev);
  }
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

// ------------------------------------------------------------------------------------------
// Plan finalisation: per-row offset masks, sort rows by mask, transpose, block masks.
__global__ void k_row_mask(const int32_t *__restrict__ nbr, int n, int K, uint32_t *mask) {
  D3D_SIDE_PRIO();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t m = 0;
  for (int k = 0; k < K; k++) m |= (nbr[(size_t)i * K + k] >= 0 ? 1u : 0u) << k;
  mask[i] = plan_key(m, K);
}
// number of rules of a plan = valid entries of nbrT; only run when somebody asks for the MAC count
__global__ __launch_bounds__(256) void k_count_rules(const int32_t *__restrict__ nbrT, long total,
                                                     unsigned long long *n_rules) {
  D3D_SIDE_PRIO();
  __shared__ int wsum[4];
  int c = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    c += nbrT[i] >= 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(n_rules, (unsigned long long)(wsum[0] + wsum[1] + wsum[2] + wsum[3]));
}
// nbr[row][K] (row-major, site order) -> nbrT[K][npos] in plan order, through an LDS tile: a workgroup
// owns kTP consecutive plan positions, reads their K-entry rows as contiguous 4K-byte runs and writes
// kTP-entry runs of every offset's column.  The same pass pads rows[] to npos with -1 and derives the
// per-block offset masks (ballot over the 32 positions of a block), so that the plan needs one launch
// after the sort.  HBM-bound: 2 * npos * K * 4 bytes.
static constexpr int kTP = 128;
// n_dev (may be null): the row count on the device; the launch is then sized by an upper bound and the layout (row
// stride npos of nbrT) is derived from the true count, exactly as a launch that knew it would have laid it out
__global__ __launch_bounds__(256) void k_plan_finish(const int32_t *__restrict__ nbr, int32_t *__restrict__ rows,
                                                     int n_rows, int npos, int K, int32_t *__restrict__ nbrT,
                                                     uint32_t *__restrict__ blkmask, const int32_t *__restrict__ n_dev) {
  D3D_SIDE_PRIO();
  extern __shared__ int32_t tile[];  // [kTP][S], S odd
  __shared__ int32_t rloc[kTP];
  __shared__ uint32_t bm[kTP / 32];
  const int S = K | 1;
  const int p0 = blockIdx.x * kTP;
  if (n_dev) {
    n_rows = *n_dev;
    npos = ((n_rows + 31) / 32) * 32;
    if (p0 >= npos) return;
  }
  const int np = min(kTP, npos - p0);  // multiple of 32
  if (threadIdx.x < kTP) {
    const int p = p0 + threadIdx.x;
    int r = -1;
    if (p < n_rows)
      r = rows[p];
    else if (p < npos)
      rows[p] = -1;
    rloc[threadIdx.x] = r;
  }
  if (threadIdx.x < kTP / 32) bm[threadIdx.x] = 0;
  __syncthreads();
  for (int idx = threadIdx.x; idx < np * K; idx += 256) {
    const int p = idx / K, k = idx - p * K;
    const int r = rloc[p];
    tile[p * S + k] = r >= 0 ? nbr[(size_t)r * K + k] : -1;
  }
  __syncthreads();
  // np is a multiple of 32 and 256 a multiple of 64: a wave covers two whole blocks of one offset
  for (int idx = threadIdx.x; idx < kTP * K; idx += 256) {
    const int k = idx / kTP, p = idx - k * kTP;
    int v = -1;
    if (p < np) {
      v = tile[p * S + k];
      nbrT[(size_t)k * npos + p0 + p] = v;
    }
    const unsigned long long bal = __ballot(v >= 0);
    if ((threadIdx.x & 63) == 0) {
      if (bal & 0xffffffffull) atomicOr(&bm[p >> 5], 1u << k);
      if (bal >> 32) atomicOr(&bm[(p >> 5) + 1], 1u << k);
    }
  }
  __syncthreads();
  if (threadIdx.x < np / 32) blkmask[p0 / 32 + threadIdx.x] = bm[threadIdx.x];
}

// Row masks and the grouping sort of a SMALL plan (<= kSmallMax rows) in one single-workgroup launch (k_plan_finish
// follows).  The ten launches of 3-6 us each it replaces (mask, key, 7 radix-sort kernels) cost more in launch
// latency than the work; small layers are half of a network's plans.
// Sort: stable LSD radix, 4 passes of 4 bits on a 16-bit key in LDS; thread t owns a contiguous chunk of rows and
// the counter column cnt[digit][t], so ranks need no atomics and the order is deterministic.
// key16 = (K - popcount) << 11 | (mask if K <= 11 else an 11-bit hash of it): heaviest rows first, equal masks
// adjacent (hash collisions only cost a little padding).
static constexpr int kSmallMax = 8192;  // (16384 rows / 512 threads measured slower than the rocPRIM path)
static constexpr int kSmallThreads = 1024;
__global__ __launch_bounds__(kSmallThreads) void k_plan_small(const int32_t *__restrict__ nbr,
                                                              const uint32_t *__restrict__ mask_in, int n_rows,
                                                              int K, int32_t *__restrict__ rows) {
  D3D_SIDE_PRIO();
  __shared__ uint32_t buf[2][kSmallMax];          // (key16 << 16) | row
  __shared__ uint16_t cnt[16 * kSmallThreads];    // [digit][thread]
  __shared__ uint32_t wsum[kSmallThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n_rows + kSmallThreads - 1) / kSmallThreads;
  const int i0 = min(n_rows, tid * per), i1 = min(n_rows, i0 + per);
  for (int i = i0; i < i1; i++) {
    uint32_t m;
    if (mask_in)
      m = plan_key_mask(mask_in[i], K);
    else {
      m = 0;
      for (int k = 0; k < K; k++) m |= (nbr[(size_t)i * K + k] >= 0 ? 1u : 0u) << k;
    }
    const uint32_t lo = K <= 11 ? m : (m * 0x9E3779B1u) >> 21;
    buf[0][i] = ((((uint32_t)(K - __popc(m)) << 11) | lo) << 16) | (uint32_t)i;
  }
  __syncthreads();
  for (int pass = 0; pass < 4; pass++) {
    const uint32_t *src = buf[pass & 1];
    uint32_t *dst = buf[(pass & 1) ^ 1];
    const int shift = 16 + 4 * pass;
#pragma unroll
    for (int d = 0; d < 16; d++) cnt[d * kSmallThreads + tid] = 0;
    for (int i = i0; i < i1; i++) cnt[((src[i] >> shift) & 15u) * kSmallThreads + tid]++;
    __syncthreads();
    // exclusive scan of the flattened [digit][thread] counters: thread t owns entries [16 t, 16 t + 16)
    uint32_t loc[16], sum = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      loc[j] = sum;
      sum += cnt[tid * 16 + j];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = inc - sum;
    for (int w = 0; w < wave; w++) base += wsum[w];
#pragma unroll
    for (int j = 0; j < 16; j++) cnt[tid * 16 + j] = (uint16_t)(base + loc[j]);
    __syncthreads();
    for (int i = i0; i < i1; i++) {
      const uint32_t v = src[i];
      dst[cnt[((v >> shift) & 15u) * kSmallThreads + tid]++] = v;
    }
    __syncthreads();
  }
  const uint32_t *ord = buf[0];  // 4 passes: back in buffer 0
  for (int p = tid; p < n_rows; p += kSmallThreads) rows[p] = (int32_t)(ord[p] & 0xffffu);
}


// `mask_in` (may be null): per-row offset masks already computed by the caller together with the
// rule count in plan.n_rules_dev.  The rule count stays on the device until somebody asks for it.
int finalize_plan(d3d_meta *m, const int32_t *nbr, int n_rows, int K, Plan &plan, hipStream_t s,
                  uint32_t *mask_in) {
  D3D_REQUIRE(K >= 1 && K <= 32, "filter volume %d not supported (1..32)", K);
  Arena &A = lane_arena(m, s);
  plan.K = K;
  plan.n_rows = n_rows;
  plan.n_blk = (n_rows + 31) / 32;
  const int npos = plan.n_blk * 32;
  D3D_ALLOC(rows, int32_t, A, (size_t)npos + 1);
  D3D_ALLOC(nbrT, int32_t, A, (size_t)npos * K + 1);
  D3D_ALLOC(blkmask, uint32_t, A, (size_t)plan.n_blk + 1);
  plan.rows = rows;
  plan.nbrT = nbrT;
  plan.blkmask = blkmask;
  plan.n_rules = n_rows == 0 ? 0 : -1;
  if (n_rows == 0) return D3D_OK;
  if (n_rows <= kSmallMax) {
    hipLaunchKernelGGL(k_plan_small, dim3(1), dim3(kSmallThreads), 0, s, nbr, mask_in, n_rows, K, rows);
    hipLaunchKernelGGL(k_plan_finish, dim3((npos + kTP - 1) / kTP), dim3(256), (size_t)kTP * (K | 1) * sizeof(int32_t),
                       s, nbr, rows, n_rows, npos, K, nbrT, blkmask, (const int32_t *)nullptr);
    D3D_LAUNCH_CHECK();
    return D3D_OK;
  }
  size_t mark = A.used;
  uint32_t *mask = mask_in;
  if (!mask) {
    mask = A.get<uint32_t>(n_rows);
    if (!mask) {
      set_error("metadata arena exhausted while finalising a rulebook");
      return D3D_ERR_NOMEM;
    }
    hipLaunchKernelGGL(k_row_mask, grid1d(n_rows), dim3(256), 0, s, nbr, n_rows, K, mask);
  }
  // Exact, STABLE sort by (popcount, mask): rows of a mask class stay in site-id order, i.e. consecutive positions
  // of a block read (centre offset) and write nearly consecutive feature rows.  Measured alternative: grouping the rows
  // by hashing their masks into a class table (4 launches instead of ~19, same executed / useful steps within 10 %)
  // hands out positions by atomics, loses that order, and made the 64 -> 64 convolutions 29 % slower.  Leaving the
  // k = s = 2 plans unsorted (absent gathers cost no memory traffic) saves four sorts per building and costs the
  // strided convolutions 0.25 ms of zero tiles: 6.45 against 6.37 ms per building.
  D3D_REQUIRE(m->iota && n_rows <= m->iota_n, "finalize_plan: %d rows exceed the input layer's %d points", n_rows, m->iota_n);
  int rc = sort_pairs_u32(mask, nullptr, m->iota, rows, n_rows, std::min(K, 32), A, s, true, nullptr);   // low K bits: the mask
  if (rc) return rc;
  hipLaunchKernelGGL(k_plan_finish, dim3((npos + kTP - 1) / kTP), dim3(256), (size_t)kTP * (K | 1) * sizeof(int32_t),
                     s, nbr, rows, n_rows, npos, K, nbrT, blkmask, (const int32_t *)nullptr);
  D3D_LAUNCH_CHECK();
  A.used = mark;  // scratch released (stream-ordered reuse)
  return D3D_OK;
}

// While a geometry stream is set (d3d_meta_set_geometry_stream) the slab has two single-stream lanes: the geometry
// stream builds grids + strided rulebooks in `arena`, every other stream (the feature pass: submanifold / deconvolution
// rulebooks built by the first convolution that needs them, partial tiles, rule counts) works in `feat_arena`, so
// neither carves temporaries out of memory the other stream may still be using.
// A plan stream (d3d_meta_set_plan_stream) adds a third lane for the rulebooks that are views of an existing grid.
Arena &lane_arena(d3d_meta *m, hipStream_t s) {
  if (!m->geo_locked || s == m->geo_stream) return m->arena;
  return (m->plan_stream && s == m->plan_stream) ? m->plan_arena : m->feat_arena;
}

// New grids come from the geometry stream alone.
int check_build_stream(d3d_meta *m, hipStream_t s, const char *what) {
  if (m->geo_locked && s != m->geo_stream) {
    set_error("%s requested on a stream other than the metadata's geometry stream (d3d_meta_set_geometry_stream): "
              "prepare it there first", what);
    return D3D_ERR_STATE;
  }
  return D3D_OK;
}

// reads the rule count back (one stream sync) the first time it is asked for
int plan_rules(d3d_meta *m, Plan &p, hipStream_t s, long *out) {
  if (p.n_rules < 0) {
    Arena &A = m->feat_arena;   // may run on the feature stream while geometry is built on another one
    size_t mark = A.used;
    D3D_ALLOC(cnt, unsigned long long, A, 1);
    D3D_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(unsigned long long), s));
    const long total = (long)p.n_blk * 32 * p.K;
    hipLaunchKernelGGL(k_count_rules, dim3((unsigned)std::min<long>(1024, (total + 255) / 256)), dim3(256), 0, s,
                       p.nbrT, total, cnt);
    D3D_LAUNCH_CHECK();
    A.used = mark;
    // (a pinned word of its own: the geometry thread may be reading a site count back at the same time)
    D3D_HIP_CHECK(hipMemcpyAsync(&m->host_words[8], cnt, sizeof(long), hipMemcpyDeviceToHost, s));
    D3D_HIP_CHECK(hipStreamSynchronize(s));
    p.n_rules = m->host_words[8];
  }
  *out = p.n_rules;
  return D3D_OK;
}

__global__ void k_identity_plan(int32_t *rows, int32_t *nbrT, uint32_t *blkmask, int n, int npos, int n_blk) {
  D3D_SIDE_PRIO();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npos) {
    int v = i < n ? i : -1;
    rows[i] = v;
    nbrT[i] = v;
  }
  if (i < n_blk) blkmask[i] = 1u;
}

static PlanKey make_key(int kind, const int *in_size, const int *filt, const int *stride) {
  PlanKey k;
  k[0] = kind;
  for (int d = 0; d < 3; d++) {
    k[1 + d] = in_size[d];
    k[4 + d] = filt[d];
    k[7 + d] = stride ? stride[d] : 0;
  }
  return k;
}
const Plan *find_plan(d3d_meta *m, int kind, const int *in_size, const int *filt, const int *stride) {
  D3D_LOCK(m);
  auto it = m->plans.find(make_key(kind, in_size, filt, stride));
  return it == m->plans.end() ? nullptr : &it->second;
}

int get_deconv_plan(d3d_meta *m, const int *fine_size, const int *filt, const int *stride,
                    hipStream_t s, const Plan **out) {
  PlanKey key = make_key(2, fine_size, filt, stride);
  std::map<PlanKey, Plan>::iterator it;
  std::map<PlanKey, StridedRaw>::iterator raw;
  bool have, have_raw;
  {
    D3D_LOCK(m);
    it = m->plans.find(key);
    have = it != m->plans.end();
    raw = m->strided_raw.find(make_key(1, fine_size, filt, stride));
    have_raw = raw != m->strided_raw.end();
  }
  if (!have) {
    if (!have_raw) {
      set_error("deconvolution: no strided rulebook for this (size, filter, stride); run the "
                "matching convolution (d3d_conv_prepare) first");
      return D3D_ERR_STATE;
    }
    Plan p;
    int K = filt[0] * filt[1] * filt[2];
    int rc = finalize_plan(m, raw->second.nbr_dec, raw->second.n_in, K, p, s, nullptr);
    if (rc) return rc;
    D3D_LOCK(m);
    auto go = m->grids.find(raw->second.out_size);
    p.n_in = go == m->grids.end() ? 0 : go->second.n;   // gathers from the coarse tensor
    it = m->plans.emplace(key, p).first;
  }
  *out = &it->second;
  return D3D_OK;
}

// occupied extent of a grid (1 + max coordinate per axis), what sparse_3d_to_dense_2d crops the dense map to; and, when
// the grid's bounding box is small (host-side bounds `hext` from the input layer / the grid chain), a DENSE INDEX of that
// box: dense[((b * e0 + x) * e1 + y) * e2 + z] = 1 + site id, 0 = empty.  The rotated RoIAlign samples through it with
// one load per trilinear corner instead of a probe sequence through the hash table (a pyramid level the pooler reads is
// ~40 k cells: the index stays in L2 / L1).
static constexpr long kDenseMaxCells = 8L << 20;
__global__ void k_grid_extent(const int32_t *__restrict__ loc, int n, int32_t *__restrict__ extent,
                              int32_t *__restrict__ dense, int e0, int e1, int e2) {
  D3D_SIDE_PRIO();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int v[3] = {0, 0, 0};
  if (i < n) {
    const int x = loc[(size_t)i * 4], y = loc[(size_t)i * 4 + 1], z = loc[(size_t)i * 4 + 2], b = loc[(size_t)i * 4 + 3];
    v[0] = x + 1;
    v[1] = y + 1;
    v[2] = z + 1;
    if (dense) dense[(((size_t)b * e0 + x) * e1 + y) * e2 + z] = i + 1;
  }
#pragma unroll
  for (int d = 0; d < 3; d++) {
    int m = v[d];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(&extent[d], m);
  }
}
int grid_extent(d3d_meta *m, Grid &g, hipStream_t s) {
  if (g.extent) return D3D_OK;
  Arena &A = lane_arena(m, s);
  long cells = 0;
  if (g.hext[0] > 0 && g.hext[3] > 0) {
    cells = (long)g.hext[3] * g.hext[0] * g.hext[1] * g.hext[2];
    if (cells > kDenseMaxCells) cells = 0;
  }
  D3D_ALLOC(e, int32_t, A, 64 + (size_t)cells);      // [0..3] extent, [64..] dense index: one zero fill for both
  D3D_HIP_CHECK(hipMemsetAsync(e, 0, (64 + (size_t)cells) * sizeof(int32_t), s));
  int32_t *dense = cells ? e + 64 : nullptr;
  if (g.n > 0)
    hipLaunchKernelGGL(k_grid_extent, grid1d(g.n), dim3(256), 0, s, g.loc, g.n, e, dense, g.hext[0], g.hext[1], g.hext[2]);
  D3D_LAUNCH_CHECK();
  g.extent = e;
  g.dense = dense;
  return D3D_OK;
}

__global__ void k_locations(const int32_t *__restrict__ loc, int n, int64_t *__restrict__ out) {
  D3D_SIDE_PRIO();
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n * 4) out[t] = loc[t];
}
// AnchorGenerator for one map: site i, cell anchor a -> (loc_i / voxel_scale * stride + base_a.xyz, base_a.size, yaw)
struct AnchorBase {
  float v[16 * 7];
};
__global__ void k_anchors(const int32_t *__restrict__ loc, int n, int A, AnchorBase base, float vs, float s0, float s1,
                          float s2, float *__restrict__ out) {
  D3D_SIDE_PRIO();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * A) return;
  const int i = t / A, a = t - i * A;
  const int32_t *p = loc + (size_t)i * 4;
  const float *b = base.v + a * 7;
  float *o = out + (size_t)t * 7;
  // true division first, then the stride (anchor_generator_sparse3d.py:99), then + base (x + 0 keeps x)
  o[0] = (float)p[0] / vs * s0 + b[0];
  o[1] = (float)p[1] / vs * s1 + b[1];
  o[2] = (float)p[2] / vs * s2 + b[2];
  o[3] = 0.f + b[3];
  o[4] = 0.f + b[4];
  o[5] = 0.f + b[5];
  o[6] = 0.f + b[6];
}
// ... for up to kAnchorMaps maps in one launch: map m's sites are rows start[m] .. start[m+1]-1 of the row space
static constexpr int kAnchorMaps = 6, kAnchorA = 4;
struct AnchorMaps {
  const int32_t *loc[kAnchorMaps];
  int start[kAnchorMaps + 1];
  float stride[kAnchorMaps][3];
  float base[kAnchorMaps][kAnchorA * 7];
  int n_maps;
};
__global__ void k_anchors_maps(AnchorMaps am, int A, float vs, float *__restrict__ out) {
  D3D_SIDE_PRIO();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= am.start[am.n_maps] * A) return;
  const int row = t / A, a = t - row * A;
  int m = 0;
  while (m + 1 < am.n_maps && row >= am.start[m + 1]) m++;
  const int32_t *p = am.loc[m] + (size_t)(row - am.start[m]) * 4;
  const float *b = am.base[m] + a * 7;
  float *o = out + (size_t)t * 7;
  // the same operations, in the same order, as k_anchors
  o[0] = (float)p[0] / vs * am.stride[m][0] + b[0];
  o[1] = (float)p[1] / vs * am.stride[m][1] + b[1];
  o[2] = (float)p[2] / vs * am.stride[m][2] + b[2];
  o[3] = 0.f + b[3];
  o[4] = 0.f + b[4];
  o[5] = 0.f + b[5];
  o[6] = 0.f + b[6];
}
__global__ void k_sparse_to_dense(const float *__restrict__ in, int planes,
                                  const int32_t *__restrict__ loc, int n, int sx, int sy, int sz,
                                  float *__restrict__ out) {
  D3D_SIDE_PRIO();
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n * planes) return;
  // consecutive threads -> consecutive sites (coalesced-ish writes along z), plane-major loop
  int i = (int)(t % n), c = (int)(t / n);
  const int32_t *p = loc + (size_t)i * 4;
  size_t vol = (size_t)sx * sy * sz;
  size_t off = ((size_t)p[0] * sy + p[1]) * sz + p[2];
  out[((size_t)p[3] * planes + c) * vol + off] = in[(size_t)i * planes + c];
}
// SparseToDense backward (CPU/SparseToDense.cpp:22-33): d_in[i][c] = d_out[dense cell of site i][c]
__global__ void k_sparse_to_dense_bwd(const float *__restrict__ d_out, int planes, const int32_t *__restrict__ loc,
                                      int n, int sx, int sy, int sz, float *__restrict__ d_in) {
  D3D_SIDE_PRIO();
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)n * planes) return;
  int i = (int)(t % n), c = (int)(t / n);
  const int32_t *p = loc + (size_t)i * 4;
  size_t vol = (size_t)sx * sy * sz;
  size_t off = ((size_t)p[0] * sy + p[1]) * sz + p[2];
  d_in[(size_t)i * planes + c] = d_out[((size_t)p[3] * planes + c) * vol + off];
}
__global__ void k_export_plan(const int32_t *__restrict__ nbrT, const int32_t *__restrict__ rows,
                              int npos, int K, int swap, int32_t *triples, long capacity,
                              unsigned long long *count) {
  D3D_SIDE_PRIO();
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)npos * K) return;
  int k = (int)(t / npos), p = (int)(t % npos);
  int src = nbrT[t];
  if (src < 0) return;
  unsigned long long w = atomicAdd(count, 1ull);
  if ((long)w < capacity) {
    triples[w * 3 + 0] = swap ? rows[p] : src;
    triples[w * 3 + 1] = swap ? src : rows[p];
    triples[w * 3 + 2] = k;
  }
}
__global__ void k_export_input(const int32_t *a, int32_t *b, int n) {
  D3D_SIDE_PRIO();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i];
}

// a1 -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vox_min(const float *__restrict__ pcl, int n, int nfeat,
                                                 double scale,
                                                 unsigned long long *mins /*3, ordered-uint*/) {
  D3D_SIDE_PRIO();
  // per-axis min of (double)x*scale: grid-stride, wave shuffle, LDS, then ONE atomic per block
  // (doubles mapped to order-preserving uint64)
  __shared__ double red[4][3];
  double v[3] = {1e300, 1e300, 1e300};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    for (int d = 0; d < 3; d++) v[d] = fmin(v[d], (double)pcl[(size_t)i * nfeat + d] * scale);
  for (int d = 0; d < 3; d++) {
    double x = v[d];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) x = fmin(x, __shfl_xor(x, s, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][d] = x;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double x = fmin(fmin(red[0][threadIdx.x], red[1][threadIdx.x]), fmin(red[2][threadIdx.x], red[3][threadIdx.x]));
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    atomicMin(&mins[threadIdx.x], u);
  }
}
__device__ __forceinline__ double decode_ordered(unsigned long long u) {
  u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
  return __longlong_as_double((long long)u);
}
__global__ void k_vox_flag(const float *__restrict__ pcl, int n, int nfeat, double scale,
                           const unsigned long long *mins, int fx, int fy, int fz, int32_t *flag) {
  D3D_SIDE_PRIO();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int full[3] = {fx, fy, fz};
  bool ok = true;
  for (int d = 0; d < 3; d++) {
    double a = (double)pcl[(size_t)i * nfeat + d] * scale + (-decode_ordered(mins[d]));
    ok = ok && (a >= 0) && (a < (double)full[d]);
  }
  flag[i] = ok ? 1 : 0;
}
__global__ void k_vox_write(const float *__restrict__ pcl, int n, int nfeat, double scale,
                            const unsigned long long *mins, const int32_t *__restrict__ flag,
                            const int32_t *__restrict__ rank, int64_t *coords, float *feats) {
  D3D_SIDE_PRIO();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  int o = rank[i];
  for (int d = 0; d < 3; d++) {
    double a = (double)pcl[(size_t)i * nfeat + d] * scale + (-decode_ordered(mins[d]));
    coords[(size_t)o * 3 + d] = (int64_t)a;                  // trunc, suncg_dataset.py:173
    feats[(size_t)o * nfeat + d] = (float)(a / scale);        // :149
  }
  for (int c = 3; c < nfeat; c++) feats[(size_t)o * nfeat + c] = pcl[(size_t)i * nfeat + c];
}

}  // namespace d3d

using namespace d3d;

namespace d3d {
// The chain of strided grids of one scene, run by a thread of its own (d3d_geometry_async_start): every new grid costs
// one blocking read-back of its site count, and a caller that builds the chain itself cannot enqueue feature kernels
// while it waits.  The thread builds the listed rulebooks in order on the geometry stream and publishes, per entry,
// the output site count and an event; the caller picks an entry up when it needs it (d3d_geometry_async_wait).
// A new strided grid is complete (hash table, coordinates, site count) as soon as its count has been read back, before
// the strided rulebook that d3d_conv_prepare builds with it: it is entered into the metadata then, and a geometry thread
// may mark the moment on its stream (t_on_grid) so that the grid's views do not wait for the rulebook.
static thread_local void (*t_on_grid)(void *, hipStream_t) = nullptr;
static thread_local void *t_on_grid_arg = nullptr;
static void publish_grid(d3d_meta *m, Grid &go, int n_out, const int *out_size, hipStream_t s) {
  go.n = n_out;
  {
    D3D_LOCK(m);
    m->grids[Size3{out_size[0], out_size[1], out_size[2]}] = go;
  }
  if (t_on_grid) t_on_grid(t_on_grid_arg, s);
}

struct GeoAsync {
  std::thread th, th_views;                 // the grids (blocking read-backs) / the views behind them; started with the
  bool threads_up = false, stop = false;    // first chain of this metadata and kept (a handle serves scene after scene)
  int job = 0, job_done[2] = {0, 0};        // chains started / finished by each worker
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::array<int, 13>> specs;   // kind, in_size, out_size, filter, stride
  std::vector<int> n_out;
  std::vector<hipEvent_t> ev, gev;          // pools, reused from scene to scene: entry done / its grid complete
  std::vector<char> ready, grid_ready;      // entry built (grids) / enqueued (views); grid of a kind-1 entry complete
  int rc = 0, device = 0;
  std::string err;
  hipStream_t stream = nullptr, view_stream = nullptr;
};
// publishes the outcome of entry i; returns false when the chain has failed (here or in the other worker)
static bool geo_publish(GeoAsync *g, int i, int rc, int n_out) {
  std::lock_guard<std::mutex> lk(g->mu);
  if (rc == D3D_OK) {
    g->n_out[i] = n_out;
    g->ready[i] = 1;
  } else if (g->rc == D3D_OK) {
    g->rc = rc;
    g->err = g_err;
  }
  g->cv.notify_all();
  return g->rc == D3D_OK;
}
static int geo_begin(GeoAsync *g) {
  if (hipSetDevice(g->device) == hipSuccess) return D3D_OK;
  set_error("geometry thread: hipSetDevice(%d) failed", g->device);
  return D3D_ERR_HIP;
}
static void geo_run_grids(d3d_meta *m, GeoAsync *g);   // (call d3d_conv_prepare & co, defined with the C entry points)
static void geo_run_views(d3d_meta *m, GeoAsync *g);
static void geo_worker(d3d_meta *m, GeoAsync *g, int which) {
  int seen = 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv.wait(lk, [&] { return g->stop || g->job != seen; });
      if (g->stop) return;
      seen = g->job;
    }
    if (which == 0) geo_run_grids(m, g);
    else geo_run_views(m, g);
    std::lock_guard<std::mutex> lk(g->mu);
    g->job_done[which] = seen;
    g->cv.notify_all();
  }
}
static void geo_async_join(d3d_meta *m) {      // waits until both workers have finished the chain that was started
  GeoAsync *g = (GeoAsync *)m->geo_async;
  if (!g || !g->threads_up) return;
  std::unique_lock<std::mutex> lk(g->mu);
  g->cv.wait(lk, [&] { return g->job_done[0] == g->job && g->job_done[1] == g->job; });
}
static void geo_async_free(d3d_meta *m) {
  GeoAsync *g = (GeoAsync *)m->geo_async;
  if (!g) return;
  geo_async_join(m);
  if (g->threads_up) {
    {
      std::lock_guard<std::mutex> lk(g->mu);
      g->stop = true;
      g->cv.notify_all();
    }
    g->th.join();
    g->th_views.join();
  }
  for (hipEvent_t e : g->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : g->gev) (void)hipEventDestroy(e);
  delete g;
  m->geo_async = nullptr;
}
}  // namespace d3d

extern "C" {

const char *d3d_last_error(void) { return d3d::g_err; }
int d3d_abi_version(void) { return 1; }

int d3d_meta_create(d3d_meta **out, size_t arena_bytes) {
  D3D_REQUIRE(out && arena_bytes >= (1u << 20), "d3d_meta_create: bad arguments");
  d3d_meta *m = new d3d_meta();
  hipError_t e = hipMalloc((void **)&m->arena.base, arena_bytes);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu) failed: %s", arena_bytes, hipGetErrorString(e));
    delete m;
    return D3D_ERR_HIP;
  }
  const size_t feat_bytes = (arena_bytes / 3) & ~size_t(255);   // second lane (see lane_arena)
  m->arena.cap = m->arena_cap_full = arena_bytes - feat_bytes;
  m->feat_arena.base = m->arena.base + m->arena.cap;
  m->feat_arena.cap = m->feat_cap_full = feat_bytes;
  e = hipHostMalloc((void **)&m->host_words, 16 * sizeof(long), hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void **)&m->host_counts, 32 * sizeof(int32_t), hipHostMallocDefault);
  if (e != hipSuccess) {
    set_error("hipHostMalloc failed: %s", hipGetErrorString(e));
    (void)hipFree(m->arena.base);
    delete m;
    return D3D_ERR_HIP;
  }
  *out = m;
  return D3D_OK;
}
int d3d_meta_destroy(d3d_meta *m) {
  if (!m) return D3D_OK;
  geo_async_free(m);
  (void)hipFree(m->arena.base);
  (void)hipHostFree(m->host_words);
  if (m->host_counts) (void)hipHostFree(m->host_counts);
  if (m->count_ev) (void)hipEventDestroy(m->count_ev);
  if (m->chain_ev) (void)hipEventDestroy(m->chain_ev);
  if (m->grid_ev) (void)hipEventDestroy(m->grid_ev);
  if (m->lists_ev) (void)hipEventDestroy(m->lists_ev);
  if (m->fill_ev) (void)hipEventDestroy(m->fill_ev);
  if (m->aux_stream) (void)hipStreamDestroy(m->aux_stream);
  delete m;
  return D3D_OK;
}
int d3d_meta_clear(d3d_meta *m) {
  D3D_REQUIRE(m, "null metadata");
  geo_async_join(m);
  D3D_LOCK(m);
  m->arena.used = 0;
  m->arena.cap = m->arena_cap_full;
  m->pre_nbr = nullptr;
  m->pre_mask = nullptr;
  m->pre_plan_built = false;
  m->pre_tab = nullptr;
  m->lists_on_aux = false;
  m->feat_arena.used = 0;
  m->feat_arena.cap = m->feat_cap_full;
  m->plan_arena = Arena();
  m->plan_stream = nullptr;
  m->geo_locked = false;
  m->geo_stream = nullptr;
  m->grids.clear();
  m->plans.clear();
  m->strided_raw.clear();
  m->in_n = m->in_mode = m->in_active = 0;
  for (int d = 0; d < 4; d++) m->in_ext[d] = 0;
  m->in_off = m->in_idx = m->in_pslot = nullptr;
  m->pl_scratch = nullptr;
  m->pl_scratch_bytes = 0;
  m->iota = nullptr;
  m->iota_n = 0;
  m->in_lists = false;
  return D3D_OK;
}
int d3d_meta_set_geometry_stream(d3d_meta *m, void *stream, int enable) {
  D3D_REQUIRE(m, "null metadata");
  m->geo_locked = enable != 0;
  m->geo_stream = enable ? (hipStream_t)stream : nullptr;
  return D3D_OK;
}
int d3d_meta_set_plan_stream(d3d_meta *m, void *stream, int enable) {
  D3D_REQUIRE(m, "null metadata");
  if (!enable) {
    m->plan_stream = nullptr;   // the lane's rulebooks stay where they are until d3d_meta_clear
    return D3D_OK;
  }
  D3D_REQUIRE(m->geo_locked && stream && (hipStream_t)stream != m->geo_stream,
              "d3d_meta_set_plan_stream: set a geometry stream first, and a different one");
  if (!m->plan_arena.base) {    // first use for this scene: the upper 3/4 of the feature lane become the plan lane
    const size_t keep = (m->feat_arena.cap / 4) & ~size_t(255);
    D3D_REQUIRE(m->feat_arena.used <= keep, "d3d_meta_set_plan_stream: the feature lane is already %zu bytes deep",
                m->feat_arena.used);
    m->plan_arena.base = m->feat_arena.base + keep;
    m->plan_arena.cap = m->feat_arena.cap - keep;
    m->plan_arena.used = 0;
    m->feat_arena.cap = keep;
  }
  m->plan_stream = (hipStream_t)stream;
  return D3D_OK;
}
int d3d_meta_arena_used(d3d_meta *m, size_t *bytes_host) {
  D3D_REQUIRE(m && bytes_host, "null argument");
  *bytes_host = m->arena.used;
  return D3D_OK;
}

size_t d3d_voxelize_scratch_bytes(int n) {
  size_t nb = ((size_t)n + kScanTile - 1) / kScanTile;
  return 256 * 4 + (size_t)n * 8 + nb * 4 + 4096;
}
namespace d3d {
__global__ void k_store_word(const int32_t *__restrict__ src, int32_t *__restrict__ dst) {
  D3D_SIDE_PRIO();
  __hip_atomic_store(dst, *src, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// per host thread and device: a pinned word a kernel stores a count to and the event recorded behind that store
struct VoxWord {
  int32_t *word = nullptr;
  hipEvent_t ev = nullptr;
};
static VoxWord *vox_word() {
  static thread_local VoxWord words[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
    set_error("d3d_voxelize: no current device");
    return nullptr;
  }
  VoxWord &w = words[dev];
  if (!w.word) {
    hipError_t e = hipHostMalloc((void **)&w.word, 64, hipHostMallocPortable);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&w.ev, hipEventDisableTiming);
    if (e != hipSuccess) {
      set_error("d3d_voxelize: pinned word / event: %s", hipGetErrorString(e));
      w.word = nullptr;
      return nullptr;
    }
  }
  return &w;
}
}  // namespace d3d

int d3d_voxelize(const float *pcl, int n, int nfeat, double scale, const int *full, int64_t *coords,
                 float *feats, int *n_kept_host, void *scratch, size_t scratch_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(pcl && coords && feats && n_kept_host && full && nfeat >= 3 && n >= 0, "d3d_voxelize: bad arguments");
  D3D_REQUIRE(scratch_bytes >= d3d_voxelize_scratch_bytes(n), "d3d_voxelize: scratch too small");
  *n_kept_host = 0;
  if (n == 0) return D3D_OK;
  Arena A;
  A.base = (char *)scratch;
  A.cap = scratch_bytes;
  D3D_ALLOC(mins, unsigned long long, A, 4);
  D3D_ALLOC(flag, int32_t, A, n);
  D3D_ALLOC(rank, int32_t, A, n);
  D3D_HIP_CHECK(hipMemsetAsync(mins, 0xFF, 4 * sizeof(unsigned long long), s));
  hipLaunchKernelGGL(k_vox_min, dim3(std::min(1024u, grid1d(n).x)), dim3(256), 0, s, pcl, n, nfeat, scale, mins);
  hipLaunchKernelGGL(k_vox_flag, grid1d(n), dim3(256), 0, s, pcl, n, nfeat, scale, mins, full[0], full[1], full[2], flag);
  int rc = scan_exclusive_i32(flag, rank, n, (int32_t *)(mins + 3), A, s);
  if (rc) return rc;
  // the count goes to a pinned word by a store of its own launch, with an event behind it: the host waits for that
  // event while k_vox_write runs (a pageable hipMemcpy + stream synchronise took the write's time and a staged copy more)
  VoxWord *w = vox_word();
  if (!w) return D3D_ERR_HIP;
  hipLaunchKernelGGL(k_store_word, dim3(1), dim3(1), 0, s, (const int32_t *)(mins + 3), w->word);
  D3D_HIP_CHECK(hipEventRecord(w->ev, s));
  hipLaunchKernelGGL(k_vox_write, grid1d(n), dim3(256), 0, s, pcl, n, nfeat, scale, mins, flag, rank, coords, feats);
  D3D_LAUNCH_CHECK();
  D3D_HIP_CHECK(hipEventSynchronize(w->ev));
  *n_kept_host = *(volatile int32_t *)w->word;
  return D3D_OK;
}

namespace d3d {
static int build_point_lists(d3d_meta *m, hipStream_t s, bool by_bound);
}
static size_t point_list_scratch_bytes(int n) {   // what ensure_point_lists carves: site of every point, counts, scan sums, sort
  const size_t nn = (size_t)std::max(n, 1);
  return nn * 4 + 256 + (nn + 2) * 4 + 256 + ((nn + 1 + kScanTile - 1) / kScanTile) * 4 + 256 + sort_scratch_bytes(n, 32) + 4096;
}

int d3d_input_layer_build(d3d_meta *m, const int64_t *coords, int n, int ncols, const int *size,
                          int batch_size, int mode, void *stream, int *n_active_host) {
  return d3d_input_layer_build_prefetch(m, coords, n, ncols, size, batch_size, mode, nullptr, stream, n_active_host);
}

int d3d_input_layer_build_prefetch(d3d_meta *m, const int64_t *coords, int n, int ncols, const int *size,
                                   int batch_size, int mode, const int *prefetch_filter, void *stream,
                                   int *n_active_host) {
  hipStream_t s = (hipStream_t)stream;
  (void)batch_size;
  D3D_REQUIRE(m && size && n_active_host, "null argument");
  D3D_REQUIRE(ncols == 3 || ncols == 4, "coords must be [n,3] or [n,4], got %d columns", ncols);
  D3D_REQUIRE(mode == 3 || mode == 4, "input layer mode %d not supported (3=sum, 4=mean)", mode);
  D3D_REQUIRE(n >= 0 && (n == 0 || coords), "bad coords");
  {
    D3D_LOCK(m);
    if (!m->grids.empty()) {
      set_error("input layer: metadata already holds grids; call d3d_meta_clear first");
      return D3D_ERR_STATE;
    }
  }
  for (int d = 0; d < 3; d++) D3D_REQUIRE(size[d] > 0 && size[d] <= 32768, "spatial size out of range");
  Arena &A = m->arena;
  Grid g;
  for (int d = 0; d < 3; d++) g.size[d] = size[d];
  g.cap = next_pow2(2L * n);
  D3D_ALLOC(tab, HashEntry, A, g.cap);
  D3D_ALLOC(loc, int32_t, A, (size_t)n * 4 + 4);
  D3D_ALLOC(in_off, int32_t, A, (size_t)n + 2);
  D3D_ALLOC(in_idx, int32_t, A, (size_t)n + 1);
  D3D_ALLOC(pslot, int32_t, A, (size_t)n + 1);
  D3D_ALLOC(iota, int32_t, A, (size_t)n + 1);   // 0, 1, 2, ...: the values every plan sort permutes (no grid has more rows)
  {
    const size_t pl = point_list_scratch_bytes(n);
    D3D_ALLOC(pls, char, A, pl);
    m->pl_scratch = pls;
    m->pl_scratch_bytes = pl;
  }
  m->iota = iota;
  m->iota_n = n;
  if (n > 0) hipLaunchKernelGGL(k_iota, grid1d(n), dim3(256), 0, s, iota, n);
  m->in_pslot = pslot;
  m->in_lists = false;
  m->in_size = Size3{size[0], size[1], size[2]};
  g.tab = tab;
  g.loc = loc;
  m->in_n = n;
  m->in_mode = mode;
  m->in_off = in_off;
  m->in_idx = in_idx;
  int n_active = 0;
  if (n > 0) {
    // [0] site count, [1..4] extents of the points (k_insert_points).  NOT part of the scratch released below: the
    // prefetched neighbour probes read the count from it while other streams may already be allocating from this lane.
    D3D_ALLOC(total, int32_t, A, 8);
    // (the arrays of the rulebook that is enqueued below before the count is back: sized by the point count)
    const int pre_K = prefetch_filter ? prefetch_filter[0] * prefetch_filter[1] * prefetch_filter[2] : 0;
    const int nblk_b = (n + 31) / 32, npos_b = nblk_b * 32;
    int32_t *prows = nullptr, *pnbrT = nullptr;
    uint32_t *pblk = nullptr;
    if (pre_K > 1 && pre_K <= 32) {
      prows = A.get<int32_t>((size_t)npos_b + 1);
      pnbrT = A.get<int32_t>((size_t)npos_b * pre_K + 1);
      pblk = A.get<uint32_t>((size_t)nblk_b + 1);
    }
    size_t mark = A.used;
    D3D_ALLOC(flag, int32_t, A, n);
    D3D_ALLOC(rank, int32_t, A, n);
    // Everything of the pass's start that does not need the site count ON THE HOST is enqueued before the count is read
    // back, sized by the point count and reading the site count on the device, so that the GPU keeps working while the
    // count travels: on this stream the submanifold rulebook the caller will ask for first -- hash probes of every site,
    // the sort of the rows by offset mask, the transposed table (raw table, masks and sort scratch live at the top of
    // the geometry lane, which stays that much shorter for the scene) -- and, on a stream of the library's own, the
    // fills of that table (beside the point insertion) and the input layer's point lists (beside the probes; counters,
    // scan and sort cover n sites' worth of entries, sites past the true count hold no points).  Launch ORDER follows
    // the critical path: the host needs ~4 us per launch, and the probes must not queue behind the lists' 12 launches.
    bool pre = false, sym = false;
    size_t raw = 0, msk = 0, srt = 0;
    if (prefetch_filter && m->pl_scratch) {
      raw = (((size_t)n * pre_K + 1) * sizeof(int32_t) + 255) & ~size_t(255);
      msk = ((size_t)n * sizeof(uint32_t) + 511) & ~size_t(255);
      srt = (sort_scratch_bytes(n, std::min(pre_K, 32)) + 255) & ~size_t(255);
      pre = pre_K > 1 && pre_K <= 32 && prows && pnbrT && pblk && A.used + raw + msk + srt + (64u << 20) < A.cap;
    }
    Arena sc;
    if (pre) {
      if (!m->aux_stream) {
        D3D_HIP_CHECK(hipStreamCreateWithFlags(&m->aux_stream, hipStreamNonBlocking));
        D3D_HIP_CHECK(hipEventCreateWithFlags(&m->grid_ev, hipEventDisableTiming));
        D3D_HIP_CHECK(hipEventCreateWithFlags(&m->lists_ev, hipEventDisableTiming));
        D3D_HIP_CHECK(hipEventCreateWithFlags(&m->fill_ev, hipEventDisableTiming));
      }
      A.cap = (A.cap - raw - msk - srt) & ~size_t(255);
      sc.base = A.base + A.cap;
      sc.cap = srt;
      m->pre_mask = (uint32_t *)(A.base + A.cap + srt);
      m->pre_nbr = (int32_t *)(A.base + A.cap + srt + msk);
      for (int d = 0; d < 3; d++) m->pre_filt[d] = prefetch_filter[d];
      m->pre_stream = s;
      sym = subm_nbr_is_sym(n, prefetch_filter);
      // the library's stream joins the caller's here (arena reuse from scene to scene is ordered by the caller's stream)
      D3D_HIP_CHECK(hipEventRecord(m->grid_ev, s));
      D3D_HIP_CHECK(hipStreamWaitEvent(m->aux_stream, m->grid_ev, 0));
      if (sym) {
        if (int rc2 = subm_nbr_prefill(n, prefetch_filter, m->pre_nbr, m->pre_mask, m->aux_stream)) return rc2;
        D3D_HIP_CHECK(hipEventRecord(m->fill_ev, m->aux_stream));
      }
    }
    D3D_HIP_CHECK(fill_ones(tab, sizeof(HashEntry) * g.cap, s));
    D3D_HIP_CHECK(hipMemsetAsync(total, 0, 8 * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_insert_points, grid1d(n), dim3(256), 0, s, coords, n, ncols, tab, g.cap, pslot, total + 1);
    hipLaunchKernelGGL(k_flag_first, grid1d(n), dim3(256), 0, s, pslot, tab, n, flag);
    int rc = scan_exclusive_i32(flag, rank, n, total, A, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_assign_input_sites, grid1d(n), dim3(256), 0, s, coords, n, ncols, pslot, flag, rank, tab, loc);
    D3D_LAUNCH_CHECK();
    D3D_HIP_CHECK(hipMemcpyAsync(&m->host_words[0], total, 5 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    bool prefetched = false;
    if (pre) {
      if (!m->count_ev) D3D_HIP_CHECK(hipEventCreateWithFlags(&m->count_ev, hipEventDisableTiming));
      D3D_HIP_CHECK(hipEventRecord(m->count_ev, s));          // the host waits for the count, not for what follows
      // 1. the probes (the longest kernel of the start) ...
      if (sym) D3D_HIP_CHECK(hipStreamWaitEvent(s, m->fill_ev, 0));
      if (int rc2 = launch_subm_nbr(loc, n, prefetch_filter, tab, g.cap, m->pre_nbr, m->pre_mask, total, s, sym)) return rc2;
      prefetched = true;
      // 2. ... beside them the point lists, behind the grid (count_ev marks it) ...
      D3D_HIP_CHECK(hipStreamWaitEvent(m->aux_stream, m->count_ev, 0));
      m->pre_tab = tab;
      if (int rc2 = build_point_lists(m, m->aux_stream, true)) return rc2;
      D3D_HIP_CHECK(hipEventRecord(m->lists_ev, m->aux_stream));
      m->in_lists = true;
      m->lists_on_aux = true;
      // 3. ... then the rows sorted by offset mask (low K bits of the keys the probes left) and the transposed table
      Plan &p = m->pre_plan;
      p = Plan();
      p.K = pre_K;
      p.rows = prows;
      p.nbrT = pnbrT;
      p.blkmask = pblk;
      if (int rc2 = sort_pairs_u32(m->pre_mask, nullptr, m->iota, p.rows, n, std::min(pre_K, 32), sc, s, true, total)) return rc2;
      hipLaunchKernelGGL(k_plan_finish, dim3((npos_b + kTP - 1) / kTP), dim3(256), (size_t)kTP * (pre_K | 1) * sizeof(int32_t),
                         s, m->pre_nbr, p.rows, n, npos_b, pre_K, p.nbrT, p.blkmask, (const int32_t *)total);
      D3D_LAUNCH_CHECK();
      m->pre_plan_built = true;
    }
    if (prefetched) D3D_HIP_CHECK(hipEventSynchronize(m->count_ev));
    else D3D_HIP_CHECK(hipStreamSynchronize(s));
    n_active = (int)*(int32_t *)&m->host_words[0];
    for (int d = 0; d < 4; d++) m->in_ext[d] = ((const int32_t *)&m->host_words[0])[1 + d];
    A.used = mark;
  } else {
    D3D_HIP_CHECK(fill_ones(tab, sizeof(HashEntry) * g.cap, s));   // an empty grid still answers probes
  }
  g.n = n_active;
  for (int d = 0; d < 4; d++) g.hext[d] = m->in_ext[d];
  m->in_active = n_active;
  {
    D3D_LOCK(m);
    m->grids[Size3{size[0], size[1], size[2]}] = g;
    if (m->pre_plan_built) {        // the rulebook enqueued above, now that its row count is known on the host
      Plan &p = m->pre_plan;
      p.n_rows = p.n_in = n_active;
      p.n_blk = (n_active + 31) / 32;
      p.n_rules = n_active == 0 ? 0 : -1;
      m->plans.emplace(make_key(0, size, m->pre_filt, nullptr), p);
    }
  }
  *n_active_host = n_active;
  return D3D_OK;
}

}  // extern "C"

namespace d3d {
// Per-site point lists in input order (stable sort of point ids by site id), built by the first consumer on ITS
// stream with temporaries from that stream's lane of the arena: the grid exists as soon as d3d_input_layer_build
// returns, so a caller may start the level-0 rulebook on another stream while the lists are sorted here.
// by_bound: nothing in the build depends on the site count (counters, scan and sort cover all n points' worth of
// sites: the sites past the true count hold zero points), so the lists can be enqueued before the count is read back.
static int build_point_lists(d3d_meta *m, hipStream_t s, bool by_bound) {
  const int n = m->in_n, n_sites = by_bound ? m->in_n : m->in_active;
  std::map<Size3, Grid>::iterator it;
  HashEntry *tab = nullptr;
  if (by_bound) {
    tab = (HashEntry *)m->pre_tab;
  } else {
    D3D_LOCK(m);
    it = m->grids.find(m->in_size);
    D3D_REQUIRE(it != m->grids.end(), "input layer: grid not found");
    tab = it->second.tab;
  }
  Arena own;                      // the region the build set aside: independent of the stream's lane
  own.base = m->pl_scratch;
  own.cap = m->pl_scratch_bytes;
  Arena &A = m->pl_scratch ? own : lane_arena(m, s);
  size_t mark = A.used;
  D3D_ALLOC(psite, uint32_t, A, n);
  D3D_ALLOC(cnt, int32_t, A, (size_t)n + 1);
  D3D_REQUIRE(m->iota && n <= m->iota_n, "input layer: point index table missing");
  D3D_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(int32_t) * ((size_t)n_sites + 1), s));
  hipLaunchKernelGGL(k_point_site, grid1d(n), dim3(256), 0, s, m->in_pslot, tab, n, psite, cnt);
  int rc = scan_exclusive_i32(cnt, m->in_off, n_sites + 1, nullptr, A, s);
  if (rc) return rc;
  int bits = 1;
  while ((1L << bits) < n_sites) bits++;
  rc = sort_pairs_u32(psite, nullptr, m->iota, m->in_idx, n, bits, A, s, false);
  if (rc) return rc;
  D3D_LAUNCH_CHECK();
  A.used = mark;
  return D3D_OK;
}
int ensure_point_lists(d3d_meta *m, hipStream_t s) {
  if (m->in_n == 0) return D3D_OK;
  if (m->in_lists) {
    // built on the library's own stream by the input-layer build: this stream waits for them
    if (m->lists_on_aux) D3D_HIP_CHECK(hipStreamWaitEvent(s, m->lists_ev, 0));
    return D3D_OK;
  }
  int rc = build_point_lists(m, s, false);
  if (rc) return rc;
  m->in_lists = true;
  return D3D_OK;
}
}  // namespace d3d

extern "C" {


size_t d3d_sort_scratch_bytes(int n, int bits) { return sort_scratch_bytes(n, bits); }
int d3d_sort_pairs(const uint32_t *keys, const int32_t *vals, int n, int bits, int descending, uint32_t *keys_out,
                   int32_t *vals_out, void *scratch, size_t scratch_bytes, void *stream) {
  D3D_REQUIRE(n >= 0 && bits >= 1 && bits <= 32, "d3d_sort_pairs: bad arguments");
  if (n == 0) return D3D_OK;
  D3D_REQUIRE(keys && vals && vals_out && scratch, "d3d_sort_pairs: null pointer");
  D3D_REQUIRE(scratch_bytes >= sort_scratch_bytes(n, bits), "d3d_sort_pairs: scratch too small");
  Arena A;
  A.base = (char *)scratch;
  A.cap = scratch_bytes;
  return sort_pairs_u32(keys, keys_out, vals, vals_out, n, bits, A, (hipStream_t)stream, descending != 0);
}

int d3d_input_layer_prepare(d3d_meta *m, void *stream) {
  D3D_REQUIRE(m, "null metadata");
  if (!m->in_off) {
    set_error("input layer prepare before build");
    return D3D_ERR_STATE;
  }
  return ensure_point_lists(m, (hipStream_t)stream);
}

int d3d_input_layer_forward(d3d_meta *m, const float *feats, int planes, float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && planes > 0, "bad arguments");
  if (!m->in_off) {
    set_error("input layer forward before build");
    return D3D_ERR_STATE;
  }
  if (m->in_active == 0) return D3D_OK;
  D3D_REQUIRE(feats && out, "null feature pointer");
  if (int rc = ensure_point_lists(m, s)) return rc;
  hipLaunchKernelGGL(k_input_forward, grid1d((long)m->in_active * planes), dim3(256), 0, s, feats,
                     planes, m->in_off, m->in_idx, m->in_active, m->in_mode == 4 ? 1 : 0, out);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_input_layer_export(d3d_meta *m, int32_t *offsets, int32_t *idx, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && offsets && idx, "null argument");
  if (!m->in_off) {
    set_error("input layer export before build");
    return D3D_ERR_STATE;
  }
  if (int rc = ensure_point_lists(m, s)) return rc;
  hipLaunchKernelGGL(k_export_input, grid1d(m->in_active + 1), dim3(256), 0, s, m->in_off, offsets, m->in_active + 1);
  if (m->in_n) hipLaunchKernelGGL(k_export_input, grid1d(m->in_n), dim3(256), 0, s, m->in_idx, idx, m->in_n);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

static Grid *find_grid(d3d_meta *m, const int *size) {
  D3D_LOCK(m);
  auto it = m->grids.find(Size3{size[0], size[1], size[2]});
  return it == m->grids.end() ? nullptr : &it->second;
}

int d3d_get_n_active(d3d_meta *m, const int *size, int *n_host) {
  D3D_REQUIRE(m && size && n_host, "null argument");
  Grid *g = find_grid(m, size);
  if (!g) {
    set_error("no grid of spatial size [%d,%d,%d]", size[0], size[1], size[2]);
    return D3D_ERR_STATE;
  }
  *n_host = g->n;
  return D3D_OK;
}

int d3d_get_spatial_locations(d3d_meta *m, const int *size, int64_t *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && size, "null argument");
  Grid *g = find_grid(m, size);
  if (!g) {
    set_error("no grid of spatial size [%d,%d,%d]", size[0], size[1], size[2]);
    return D3D_ERR_STATE;
  }
  if (g->n == 0) return D3D_OK;
  D3D_REQUIRE(out, "null output");
  hipLaunchKernelGGL(k_locations, grid1d((long)g->n * 4), dim3(256), 0, s, g->loc, g->n, out);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_anchors(d3d_meta *m, const int *size, const float *base_host, int A, const float *stride_host,
                float voxel_scale, float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && size && base_host && stride_host && A >= 1 && A <= 16 && voxel_scale > 0, "anchors: bad arguments");
  Grid *g = find_grid(m, size);
  if (!g) {
    set_error("anchors: no grid of spatial size [%d,%d,%d]", size[0], size[1], size[2]);
    return D3D_ERR_STATE;
  }
  if (g->n == 0) return D3D_OK;
  D3D_REQUIRE(out, "null output");
  AnchorBase base;
  for (int i = 0; i < A * 7; i++) base.v[i] = base_host[i];
  hipLaunchKernelGGL(k_anchors, grid1d((long)g->n * A), dim3(256), 0, s, g->loc, g->n, A, base, voxel_scale,
                     stride_host[0], stride_host[1], stride_host[2], out);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_anchors_maps(d3d_meta *m, int n_maps, const int *sizes_host, const float *bases_host, int A,
                     const float *strides_host, float voxel_scale, float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && sizes_host && bases_host && strides_host && n_maps >= 1 && n_maps <= kAnchorMaps && A >= 1 &&
                  A <= kAnchorA && voxel_scale > 0,
              "anchors_maps: 1..%d maps, 1..%d anchors per site", kAnchorMaps, kAnchorA);
  AnchorMaps am = {};
  long n = 0;
  for (int i = 0; i < n_maps; i++) {
    Grid *g = find_grid(m, sizes_host + 3 * i);
    if (!g) {
      set_error("anchors: no grid of spatial size [%d,%d,%d]", sizes_host[3 * i], sizes_host[3 * i + 1], sizes_host[3 * i + 2]);
      return D3D_ERR_STATE;
    }
    am.loc[i] = g->loc;
    am.start[i] = (int)n;
    n += g->n;
    for (int d = 0; d < 3; d++) am.stride[i][d] = strides_host[3 * i + d];
    for (int j = 0; j < A * 7; j++) am.base[i][j] = bases_host[(size_t)i * A * 7 + j];
  }
  D3D_REQUIRE(n * A * 7 < (1L << 31), "anchors_maps: %ld sites", n);
  for (int i = n_maps; i <= kAnchorMaps; i++) am.start[i] = (int)n;
  am.n_maps = n_maps;
  if (n == 0) return D3D_OK;
  D3D_REQUIRE(out, "null output");
  hipLaunchKernelGGL(k_anchors_maps, grid1d(n * A), dim3(256), 0, s, am, A, voxel_scale, out);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_subm_prepare(d3d_meta *m, const int *size, const int *filt, void *stream, long *n_rules_host) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && size && filt, "null argument");
  PlanKey key = make_key(0, size, filt, nullptr);
  std::map<PlanKey, Plan>::iterator it;
  bool have;
  {
    D3D_LOCK(m);
    it = m->plans.find(key);
    have = it != m->plans.end();
  }
  if (!have) {
    Grid *g = find_grid(m, size);
    if (!g) {
      set_error("submanifold rulebook: no grid of spatial size [%d,%d,%d]", size[0], size[1], size[2]);
      return D3D_ERR_STATE;
    }
    int K = filt[0] * filt[1] * filt[2];
    D3D_REQUIRE(filt[0] > 0 && filt[1] > 0 && filt[2] > 0 && K <= 32, "filter volume %d not supported", K);
    Plan p;
    Arena &A = lane_arena(m, s);
    // raw table lives above the plan's persistent arrays: allocate persistent part first
    // (finalize_plan), so stage the raw table at the far end of the arena instead.
    size_t raw_bytes = ((size_t)g->n * K + 1) * sizeof(int32_t);
    if (A.used + 2 * raw_bytes + (size_t)g->n * 16 + (1 << 20) > A.cap) {
      set_error("metadata arena exhausted while building a submanifold rulebook");
      return D3D_ERR_NOMEM;
    }
    if (K == 1) {
      // 1x1x1 submanifold convolution: every site is its own (only) neighbour -> identity rulebook
      p.K = 1;
      p.n_rows = g->n;
      p.n_blk = (g->n + 31) / 32;
      p.n_rules = g->n;
      const int npos = p.n_blk * 32;
      D3D_ALLOC(rows, int32_t, A, (size_t)npos + 1);
      D3D_ALLOC(nbrT, int32_t, A, (size_t)npos + 1);
      D3D_ALLOC(blkmask, uint32_t, A, (size_t)p.n_blk + 1);
      p.rows = rows;
      p.nbrT = nbrT;
      p.blkmask = blkmask;
      if (npos) hipLaunchKernelGGL(k_identity_plan, grid1d(npos), dim3(256), 0, s, rows, nbrT, blkmask, g->n, npos, p.n_blk);
      D3D_LAUNCH_CHECK();
    } else if (m->pre_nbr && s == m->pre_stream && g->size[0] == m->in_size[0] && g->size[1] == m->in_size[1] &&
               g->size[2] == m->in_size[2] && filt[0] == m->pre_filt[0] && filt[1] == m->pre_filt[1] &&
               filt[2] == m->pre_filt[2]) {
      // the table d3d_input_layer_build_prefetch started on this stream
      int rc = finalize_plan(m, m->pre_nbr, g->n, K, p, s, m->pre_mask);
      if (rc) return rc;
    } else {
      int32_t *nbr = (int32_t *)(A.base + ((A.cap - raw_bytes) & ~size_t(255)));
      uint32_t *mask = (uint32_t *)((char *)nbr - (((size_t)g->n * 4 + 511) & ~size_t(255)));
      if (int rc2 = launch_subm_nbr(g->loc, g->n, filt, g->tab, g->cap, nbr, mask, nullptr, s)) return rc2;
      int rc;
      {
        CapGuard guard(A, (size_t)((char *)mask - A.base) & ~size_t(255));
        rc = finalize_plan(m, nbr, g->n, K, p, s, mask);
      }
      if (rc) return rc;
    }
    p.n_in = g->n;
    D3D_LOCK(m);
    it = m->plans.emplace(key, p).first;
  }
  if (n_rules_host) return plan_rules(m, it->second, s, n_rules_host);
  return D3D_OK;
}

int d3d_conv_prepare(d3d_meta *m, const int *in_size, const int *out_size, const int *filt,
                     const int *stride, void *stream, int *n_out_host, long *n_rules_host) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && in_size && out_size && filt && stride, "null argument");
  PlanKey key = make_key(1, in_size, filt, stride);
  std::map<PlanKey, Plan>::iterator it;
  bool have;
  {
    D3D_LOCK(m);
    it = m->plans.find(key);
    have = it != m->plans.end();
  }
  if (!have) {
    if (int rc = check_build_stream(m, s, "strided rulebook")) return rc;
    Grid *gi = find_grid(m, in_size);
    if (!gi) {
      set_error("strided rulebook: no grid of spatial size [%d,%d,%d]", in_size[0], in_size[1], in_size[2]);
      return D3D_ERR_STATE;
    }
    ConvGeom geo;
    int K = 1, max_out = 1;
    for (int d = 0; d < 3; d++) {
      D3D_REQUIRE(filt[d] > 0 && stride[d] > 0 && out_size[d] > 0, "bad filter/stride/size");
      D3D_REQUIRE((out_size[d] - 1) * stride[d] + filt[d] == in_size[d],
                  "convolution sizes inconsistent: (out-1)*stride+filter != in (convolution.py:37-38)");
      geo.filt[d] = filt[d];
      geo.stride[d] = stride[d];
      geo.out_size[d] = out_size[d];
      K *= filt[d];
      max_out *= std::min((filt[d] + stride[d] - 1) / stride[d], out_size[d]);
    }
    geo.max_out = max_out;
    D3D_REQUIRE(K <= 32, "filter volume %d not supported (<= 32)", K);
    D3D_REQUIRE(max_out <= 8, "each input site may feed at most 8 outputs");
    if (find_grid(m, out_size)) {
      set_error("strided rulebook: output grid [%d,%d,%d] already exists", out_size[0], out_size[1], out_size[2]);
      return D3D_ERR_STATE;
    }
    Arena &A = m->arena;
    const int n_in = gi->n;
    const long n_entries = (long)n_in * max_out;
    Grid go;
    for (int d = 0; d < 3; d++) go.size[d] = out_size[d];
    go.cap = next_pow2(2L * n_entries);
    D3D_ALLOC(tab, HashEntry, A, go.cap);
    D3D_ALLOC(loc_out, int32_t, A, (size_t)n_entries * 4 + 4);
    D3D_ALLOC(nbr_dec, int32_t, A, (size_t)n_in * K + 1);
    go.tab = tab;
    go.loc = loc_out;
    const bool small = n_entries > 0 && n_entries <= kSmallGrid;   // one launch does it all (k_conv_grid_small)
    if (!small) {
      D3D_HIP_CHECK(fill_ones(tab, sizeof(HashEntry) * go.cap, s));
      D3D_HIP_CHECK(fill_ones(nbr_dec, sizeof(int32_t) * ((size_t)n_in * K + 1), s));
    }
    Plan p;
    int n_out = 0;
    if (small) {
      size_t raw_bytes = ((size_t)n_entries * K + 1) * sizeof(int32_t);
      if (A.used + raw_bytes + (size_t)(1 << 20) > A.cap) {
        set_error("metadata arena exhausted while building a strided rulebook");
        return D3D_ERR_NOMEM;
      }
      int32_t *nbr_fwd = (int32_t *)(A.base + ((A.cap - raw_bytes) & ~size_t(255)));
      CapGuard guard(A, (A.cap - raw_bytes) & ~size_t(255));
      size_t mark = A.used;
      D3D_ALLOC(total, int32_t, A, 1);
      hipLaunchKernelGGL(k_conv_grid_small, dim3(1), dim3(kSmallGridThreads), 0, s, gi->loc, (int)n_entries, geo, K, n_in, tab, go.cap,
                         loc_out, nbr_fwd, nbr_dec, total, (const int32_t *)nullptr, 0);
      D3D_LAUNCH_CHECK();
      D3D_HIP_CHECK(hipMemcpyAsync(&m->host_words[1], total, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      D3D_HIP_CHECK(hipStreamSynchronize(s));
      n_out = (int)*(int32_t *)&m->host_words[1];
      A.used = mark;
      publish_grid(m, go, n_out, out_size, s);
      int rc = finalize_plan(m, nbr_fwd, n_out, K, p, s, nullptr);
      if (rc) return rc;
    } else if (n_entries > 0) {
      // raw forward table staged at the far end of the arena (size known only after the scan:
      // bound it by n_entries rows)
      size_t raw_bytes = ((size_t)n_entries * K + 1) * sizeof(int32_t);
      if (A.used + raw_bytes + 12 * (size_t)n_entries * sizeof(int32_t) + (size_t)(1 << 20) > A.cap) {
        set_error("metadata arena exhausted while building a strided rulebook");
        return D3D_ERR_NOMEM;
      }
      int32_t *nbr_fwd = (int32_t *)(A.base + ((A.cap - raw_bytes) & ~size_t(255)));
      CapGuard guard(A, (A.cap - raw_bytes) & ~size_t(255));
      size_t mark = A.used;
      D3D_ALLOC(eslot, int32_t, A, n_entries);
      D3D_ALLOC(flag, int32_t, A, n_entries);
      D3D_ALLOC(rank, int32_t, A, n_entries);
      D3D_ALLOC(total, int32_t, A, 1);
      hipLaunchKernelGGL(k_conv_insert, grid1d(n_entries), dim3(256), 0, s, gi->loc, n_entries, geo, tab, go.cap, eslot,
                         (const int32_t *)nullptr);
      hipLaunchKernelGGL(k_flag_first, grid1d(n_entries), dim3(256), 0, s, eslot, tab, (int)n_entries, flag);
      int rc = scan_exclusive_i32(flag, rank, (int)n_entries, total, A, s);
      if (rc) return rc;
      hipLaunchKernelGGL(k_conv_assign, grid1d(n_entries), dim3(256), 0, s, gi->loc, n_entries, geo, eslot, flag, rank, tab, loc_out);
      D3D_HIP_CHECK(hipMemcpyAsync(&m->host_words[1], total, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      D3D_HIP_CHECK(hipStreamSynchronize(s));
      n_out = (int)*(int32_t *)&m->host_words[1];
      publish_grid(m, go, n_out, out_size, s);
      D3D_HIP_CHECK(fill_ones(nbr_fwd, sizeof(int32_t) * ((size_t)n_out * K + 1), s));
      hipLaunchKernelGGL(k_conv_fill, grid1d(n_entries), dim3(256), 0, s, gi->loc, n_entries, geo, K, eslot, tab, nbr_fwd, nbr_dec,
                         (const int32_t *)nullptr);
      D3D_LAUNCH_CHECK();
      A.used = mark;
      rc = finalize_plan(m, nbr_fwd, n_out, K, p, s, nullptr);
      if (rc) return rc;
    } else {
      publish_grid(m, go, 0, out_size, s);
      int rc = finalize_plan(m, nullptr, 0, K, p, s, nullptr);
      if (rc) return rc;
    }
    D3D_LOCK(m);   // the raw table and the rulebook become visible together (the grid already is: publish_grid)
    StridedRaw raw;
    raw.nbr_dec = nbr_dec;
    raw.n_in = n_in;
    raw.out_size = Size3{out_size[0], out_size[1], out_size[2]};
    m->strided_raw[key] = raw;
    p.n_in = n_in;
    it = m->plans.emplace(key, p).first;
  }
  if (n_out_host) *n_out_host = it->second.n_rows;
  if (n_rules_host) return plan_rules(m, it->second, s, n_rules_host);
  return D3D_OK;
}

}  // extern "C" (reopened below)

namespace d3d {
// ------------------------------------------------------------------------------------------------------------------
// Grid chain: the strided grids (+ raw rule tables) of a whole pyramid enqueued WITHOUT a host read-back between the
// levels.  d3d_conv_prepare reads every new grid's site count back before it can size the next level's launches and
// allocations: 11 stream synchronisations per building, each with the GPU idle on that stream for the round trip.  Here
// every level is sized by an upper bound known ahead of time -- sites(out) <= min(entries(in), cells of the occupied box
// at that level), from the input grid's site count and coordinate extents, which arrive in ONE read-back -- its kernels
// read the true count from the device word the level before left, all tables get their 0xFF fill in one launch up
// front, and the counts of all levels come back together behind the last kernel (k_store_counts -> pinned words, one
// event).  Only then are the grids published and the rulebooks finalised (exact sizes), in order.
// Results are those of d3d_conv_prepare bit for bit: same insertion, same first-touch numbering, same tables.
struct ChainSpec {              // one strided rulebook: sizes as d3d_conv_prepare takes them
  int in_size[3], out_size[3], filt[3], stride[3];
  int need_dec;                 // keep the decoded table (deconvolution / backward view will be asked for)
};
typedef void (*ChainHook)(void *arg, int level, int n_out, hipStream_t s);   // level published / level's rulebook enqueued

static int run_grid_chain(d3d_meta *m, const std::vector<ChainSpec> &specs, hipStream_t s, std::vector<int> &n_out,
                          ChainHook on_grid, ChainHook on_done, void *hook_arg) {
  const int L = (int)specs.size();
  n_out.assign(L, 0);
  if (L == 0) return D3D_OK;
  D3D_REQUIRE(L <= kChainMax, "grid chain: %d levels (at most %d)", L, kChainMax);
  if (int rc = check_build_stream(m, s, "grid chain")) return rc;
  struct Level {
    ConvGeom geo;
    int K = 0;
    const int32_t *loc_in = nullptr;
    const int32_t *n_in_dev = nullptr;   // null: n_in_host is exact
    int n_in_host = 0, src = -1;         // src: chain level that produces the input grid, or -1
    long bound_in = 0, bound_entries = 0, bound_out = 0;
    int ext_out[4] = {0, 0, 0, 0};
    Grid go;
    int32_t *n_out_dev = nullptr, *eslot = nullptr, *tile_cnt = nullptr, *nbr_fwd = nullptr, *nbr_dec = nullptr;
    unsigned long long *flagbits = nullptr;
    bool small = false;
  };
  std::vector<Level> lv(L);
  std::map<Size3, int> made;             // output size -> chain level
  Arena &A = m->arena;
  const size_t cap_save = A.cap;
  struct Restore {                        // the raw tables sit at the far end of the lane until the rulebooks are enqueued
    Arena &a;
    size_t cap;
    ~Restore() { a.cap = cap; }
  } restore{A, cap_save};
  auto top = [&](size_t bytes) -> void * {
    bytes = (bytes + 255) & ~size_t(255);
    if (A.used + bytes + (size_t)(1 << 20) > A.cap) return nullptr;
    A.cap = (A.cap - bytes) & ~size_t(255);
    return A.base + A.cap;
  };
  FillSegs fs = {};
  int n_seg = 0;
  D3D_REQUIRE(3 * L <= kFillSegs, "grid chain: too many levels for one fill launch");
  auto add_fill = [&](void *p, size_t bytes) {   // bytes rounded up to 16 (allocations are 256-byte aligned and padded)
    fs.ptr[n_seg] = (uint4 *)p;
    fs.n16[n_seg] = (bytes + 15) / 16;
    n_seg++;
  };
  // A level that cannot be planned (bad sizes, missing input grid, arena exhausted) ends the chain in front of it: the
  // levels before it are built and published as usual, then its error is returned (what a caller that prepares the
  // levels one by one would see).
  auto plan_level = [&](int i) -> int {
    const ChainSpec &sp = specs[i];
    Level &v = lv[i];
    int K = 1, max_out = 1;
    for (int d = 0; d < 3; d++) {
      D3D_REQUIRE(sp.filt[d] > 0 && sp.stride[d] > 0 && sp.out_size[d] > 0, "bad filter/stride/size");
      D3D_REQUIRE((sp.out_size[d] - 1) * sp.stride[d] + sp.filt[d] == sp.in_size[d],
                  "convolution sizes inconsistent: (out-1)*stride+filter != in (convolution.py:37-38)");
      v.geo.filt[d] = sp.filt[d];
      v.geo.stride[d] = sp.stride[d];
      v.geo.out_size[d] = sp.out_size[d];
      K *= sp.filt[d];
      max_out *= std::min((sp.filt[d] + sp.stride[d] - 1) / sp.stride[d], sp.out_size[d]);
    }
    v.geo.max_out = max_out;
    v.K = K;
    D3D_REQUIRE(K <= 32, "filter volume %d not supported (<= 32)", K);
    D3D_REQUIRE(max_out <= 8, "each input site may feed at most 8 outputs");
    const Size3 in_sz{sp.in_size[0], sp.in_size[1], sp.in_size[2]}, out_sz{sp.out_size[0], sp.out_size[1], sp.out_size[2]};
    if (find_grid(m, sp.out_size) || made.count(out_sz)) {
      set_error("grid chain: output grid [%d,%d,%d] already exists", sp.out_size[0], sp.out_size[1], sp.out_size[2]);
      return D3D_ERR_STATE;
    }
    int ext_in[4];
    auto src = made.find(in_sz);
    if (src != made.end()) {
      const Level &u = lv[src->second];
      v.src = src->second;
      v.loc_in = u.go.loc;
      v.n_in_dev = u.n_out_dev;
      v.bound_in = u.bound_out;
      for (int d = 0; d < 4; d++) ext_in[d] = u.ext_out[d];
    } else {
      Grid *gi = find_grid(m, sp.in_size);
      if (!gi) {
        set_error("grid chain: no grid of spatial size [%d,%d,%d]", sp.in_size[0], sp.in_size[1], sp.in_size[2]);
        return D3D_ERR_STATE;
      }
      v.loc_in = gi->loc;
      v.n_in_host = gi->n;
      v.bound_in = gi->n;
      const bool known = gi->hext[0] > 0 && gi->hext[3] > 0;
      for (int d = 0; d < 3; d++) ext_in[d] = known ? gi->hext[d] : sp.in_size[d];
      ext_in[3] = known ? gi->hext[3] : 0;            // 0: number of examples unknown -> no cell bound
    }
    v.bound_entries = v.bound_in * max_out;
    long cells = ext_in[3] > 0 ? ext_in[3] : -1;
    for (int d = 0; d < 3; d++) {
      v.ext_out[d] = std::max(1, std::min(sp.out_size[d], (ext_in[d] - 1) / sp.stride[d] + 1));
      if (cells >= 0) cells = std::min<long>(cells * v.ext_out[d], (long)1 << 40);
    }
    v.ext_out[3] = ext_in[3];
    v.bound_out = cells >= 0 ? std::min(v.bound_entries, cells) : v.bound_entries;
    D3D_REQUIRE(v.bound_entries < (1L << 30), "grid chain: %ld candidate entries", v.bound_entries);
    v.small = v.bound_entries > 0 && v.bound_entries <= kSmallGrid;
    // persistent: table, coordinates, decoded rule table; temporaries (far end): entry slots, flags, raw forward table
    for (int d = 0; d < 3; d++) v.go.size[d] = sp.out_size[d];
    v.go.cap = next_pow2(2L * std::max<long>(v.bound_out, 1));
    D3D_ALLOC(tab, HashEntry, A, v.go.cap);
    D3D_ALLOC(loc_out, int32_t, A, (size_t)v.bound_out * 4 + 4);
    D3D_ALLOC(cnt, int32_t, A, 4);
    v.go.tab = tab;
    v.go.loc = loc_out;
    v.n_out_dev = cnt;
    if (v.bound_entries > 0) add_fill(tab, sizeof(HashEntry) * (size_t)v.go.cap);
    if (sp.need_dec) {
      D3D_ALLOC(dec, int32_t, A, (size_t)v.bound_in * K + 4);
      v.nbr_dec = dec;
      if (v.bound_entries > 0) add_fill(dec, sizeof(int32_t) * ((size_t)v.bound_in * K + 1));
    }
    if (v.bound_entries > 0) {
      const size_t fwd_bytes = sizeof(int32_t) * ((size_t)v.bound_out * K + 4);
      v.nbr_fwd = (int32_t *)top(fwd_bytes);
      if (!v.small) {
        const size_t tiles = ((size_t)v.bound_entries + kChainTile - 1) / kChainTile;
        v.eslot = (int32_t *)top(sizeof(int32_t) * (size_t)v.bound_entries);
        v.flagbits = (unsigned long long *)top(sizeof(unsigned long long) * tiles * 32);
        v.tile_cnt = (int32_t *)top(sizeof(int32_t) * tiles);
      }
      if (!v.nbr_fwd || (!v.small && (!v.eslot || !v.flagbits || !v.tile_cnt))) {
        set_error("metadata arena exhausted while building a grid chain");
        return D3D_ERR_NOMEM;
      }
      add_fill(v.nbr_fwd, fwd_bytes);
    }
    made[out_sz] = i;
    return D3D_OK;
  };
  int L_ok = L, fail_rc = D3D_OK;
  char fail_msg[512] = "";
  for (int i = 0; i < L; i++) {
    const size_t used0 = A.used, cap0 = A.cap;
    const int seg0 = n_seg;
    if (int rc = plan_level(i)) {
      A.used = used0;
      A.cap = cap0;
      n_seg = seg0;
      L_ok = i;
      fail_rc = rc;
      snprintf(fail_msg, sizeof(fail_msg), "%s", g_err);
      break;
    }
  }
  // one fill for every table of every level, then the levels back to back
  if (n_seg > 0) {
    size_t most = 0;
    for (int j = 0; j < n_seg; j++) most = std::max<size_t>(most, fs.n16[j]);
    const unsigned bx = (unsigned)std::max<size_t>(1, std::min<size_t>((most + 1023) / 1024, 512));
    hipLaunchKernelGGL(k_fill_ones_multi, dim3(bx, n_seg), dim3(256), 0, s, fs);
  }
  CountPtrs cp = {};
  cp.n = L_ok;
  for (int i = 0; i < L_ok; i++) {
    Level &v = lv[i];
    cp.p[i] = v.n_out_dev;
    if (v.bound_entries == 0) {
      D3D_HIP_CHECK(hipMemsetAsync(v.n_out_dev, 0, sizeof(int32_t), s));
      continue;
    }
    if (v.small) {
      hipLaunchKernelGGL(k_conv_grid_small, dim3(1), dim3(kSmallGridThreads), 0, s, v.loc_in, (int)v.bound_entries, v.geo, v.K,
                         (int)v.bound_in, v.go.tab, v.go.cap, v.go.loc, v.nbr_fwd, v.nbr_dec, v.n_out_dev, v.n_in_dev, 1);
      continue;
    }
    const long ne = v.bound_entries;
    const dim3 tiles((unsigned)((ne + kChainTile - 1) / kChainTile));
    hipLaunchKernelGGL(k_conv_insert, grid1d(ne), dim3(256), 0, s, v.loc_in, ne, v.geo, v.go.tab, v.go.cap, v.eslot, v.n_in_dev);
    hipLaunchKernelGGL(k_chain_flag, tiles, dim3(256), 0, s, v.eslot, v.go.tab, ne, v.geo.max_out, v.n_in_dev, v.flagbits,
                       v.tile_cnt);
    hipLaunchKernelGGL(k_chain_assign, tiles, dim3(256), 0, s, v.loc_in, ne, v.geo, v.n_in_dev, v.eslot, v.flagbits,
                       v.tile_cnt, v.go.tab, v.go.loc, v.n_out_dev);
    hipLaunchKernelGGL(k_conv_fill, grid1d(ne), dim3(256), 0, s, v.loc_in, ne, v.geo, v.K, v.eslot, v.go.tab, v.nbr_fwd,
                       v.nbr_dec, v.n_in_dev);
  }
  hipLaunchKernelGGL(k_store_counts, dim3(1), dim3(64), 0, s, cp, m->host_counts);
  D3D_LAUNCH_CHECK();
  if (!m->chain_ev) D3D_HIP_CHECK(hipEventCreateWithFlags(&m->chain_ev, hipEventDisableTiming));
  D3D_HIP_CHECK(hipEventRecord(m->chain_ev, s));
  D3D_HIP_CHECK(hipEventSynchronize(m->chain_ev));          // the one read-back of the chain
  for (int i = 0; i < L_ok; i++) {
    n_out[i] = ((volatile int32_t *)m->host_counts)[i];
    D3D_REQUIRE(n_out[i] >= 0 && n_out[i] <= lv[i].bound_out, "grid chain: level %d has %d sites, bound %ld", i, n_out[i],
                lv[i].bound_out);
  }
  // the grids exist: publish them all, then the rulebooks (exact sizes) in order
  for (int i = 0; i < L_ok; i++) {
    lv[i].go.n = n_out[i];
    for (int d = 0; d < 4; d++) lv[i].go.hext[d] = lv[i].ext_out[d];
    {
      D3D_LOCK(m);
      m->grids[Size3{specs[i].out_size[0], specs[i].out_size[1], specs[i].out_size[2]}] = lv[i].go;
    }
    if (on_grid) on_grid(hook_arg, i, n_out[i], s);
  }
  for (int i = 0; i < L_ok; i++) {
    Level &v = lv[i];
    const ChainSpec &sp = specs[i];
    const int n_in = v.src >= 0 ? n_out[v.src] : v.n_in_host;
    Plan p;
    int rc = finalize_plan(m, v.nbr_fwd, n_out[i], v.K, p, s, nullptr);
    if (rc) return rc;
    {
      D3D_LOCK(m);
      const PlanKey key = make_key(1, sp.in_size, sp.filt, sp.stride);
      if (v.nbr_dec) {
        StridedRaw raw;
        raw.nbr_dec = v.nbr_dec;
        raw.n_in = n_in;
        raw.out_size = Size3{sp.out_size[0], sp.out_size[1], sp.out_size[2]};
        m->strided_raw[key] = raw;
      }
      p.n_in = n_in;
      m->plans.emplace(key, p);
    }
    if (on_done) on_done(hook_arg, i, n_out[i], s);
  }
  if (fail_rc != D3D_OK) {
    set_error("%s", fail_msg);
    return fail_rc;
  }
  return D3D_OK;
}

// D3D_GRID_CHAIN=0 (or d3d_grid_chain_enable(0)): one d3d_conv_prepare (and read-back) per level, for A/B runs
static bool g_chain_enabled = [] {
  const char *e = getenv("D3D_GRID_CHAIN");
  return !(e && e[0] == '0');
}();

static int g_chain_head = [] {
  const char *e = getenv("D3D_GRID_CHAIN_HEAD");
  return e ? atoi(e) : 1;
}();

static void geo_run_grids(d3d_meta *m, GeoAsync *g) {      // the grids of the pyramid: two chains, two read-backs
  int rc = geo_begin(g);
  const int n = (int)g->specs.size();
  std::vector<int> rows;
  for (int i = 0; i < n; i++)
    if (g->specs[i][0] == 1 || g->specs[i][0] == 3) rows.push_back(i);
  if (rows.empty()) return;
  struct Hook {
    GeoAsync *g;
    const std::vector<int> *rows;
    bool failed;
    int first;                              // row of `rows` the running chain's level 0 is
  } hk = {g, &rows, false, 0};
  if (g_chain_enabled && rc == D3D_OK) {
    std::vector<ChainSpec> specs(rows.size());
    for (size_t j = 0; j < rows.size(); j++) {
      const int *sp = g->specs[rows[j]].data() + 1;
      for (int d = 0; d < 3; d++) {
        specs[j].in_size[d] = sp[d];
        specs[j].out_size[d] = sp[3 + d];
        specs[j].filt[d] = sp[6 + d];
        specs[j].stride[d] = sp[9 + d];
      }
      specs[j].need_dec = g->specs[rows[j]][0] == 1;      // kind 3: a strided grid whose decoded table nobody will ask for
    }
    // Two chains: the first level alone -- the feature pass wants it (and its rulebooks) a few hundred microseconds after
    // the input grid, before a chain over all levels has finished -- then every other level in one go.
    const ChainHook on_grid = [](void *a, int level, int, hipStream_t on) {       // the grid exists: its views may start
      Hook *h = (Hook *)a;
      const int i = (*h->rows)[h->first + level];
      if (hipEventRecord(h->g->gev[i], on) != hipSuccess) return;
      std::lock_guard<std::mutex> lk(h->g->mu);
      h->g->grid_ready[i] = 1;
      h->g->cv.notify_all();
    };
    const ChainHook on_done = [](void *a, int level, int n_sites, hipStream_t on) {   // its strided rulebook is enqueued
      Hook *h = (Hook *)a;
      const int i = (*h->rows)[h->first + level];
      int rc2 = D3D_OK;
      if (hipEventRecord(h->g->ev[i], on) != hipSuccess) {
        set_error("geometry thread: hipEventRecord failed");
        rc2 = D3D_ERR_HIP;
      }
      if (!geo_publish(h->g, i, rc2, n_sites)) h->failed = true;
    };
    const size_t cut = std::min<size_t>(g_chain_head, specs.size());
    for (int part = 0; part < 2 && rc == D3D_OK && !hk.failed; part++) {
      const size_t lo = part == 0 ? 0 : cut, hi = part == 0 ? cut : specs.size();
      if (lo >= hi) continue;
      std::vector<ChainSpec> seg(specs.begin() + lo, specs.begin() + hi);
      std::vector<int> n_out;
      hk.first = (int)lo;
      rc = run_grid_chain(m, seg, g->stream, n_out, on_grid, on_done, &hk);
    }
    if (rc != D3D_OK) geo_publish(g, rows[0], rc, 0);
    return;
  }
  for (int i : rows) {
    const int *sp = g->specs[i].data() + 1;
    int n_out = 0;
    struct Mark {
      GeoAsync *g;
      int i;
    } mk = {g, i};
    t_on_grid_arg = &mk;
    t_on_grid = [](void *a, hipStream_t on) {       // the grid exists: its submanifold views may start
      Mark *k = (Mark *)a;
      if (hipEventRecord(k->g->gev[k->i], on) != hipSuccess) return;   // (then the views wait for the whole entry)
      std::lock_guard<std::mutex> lk(k->g->mu);
      k->g->grid_ready[k->i] = 1;
      k->g->cv.notify_all();
    };
    if (rc == D3D_OK) rc = d3d_conv_prepare(m, sp, sp + 3, sp + 6, sp + 9, g->stream, &n_out, nullptr);
    t_on_grid = nullptr;
    if (rc == D3D_OK && hipEventRecord(g->ev[i], g->stream) != hipSuccess) {
      set_error("geometry thread: hipEventRecord failed");
      rc = D3D_ERR_HIP;
    }
    if (!geo_publish(g, i, rc, n_out)) return;
  }
}
static void geo_run_views(d3d_meta *m, GeoAsync *g) {      // the views: each behind the newest grid listed before it
  int rc = geo_begin(g);
  const int n = (int)g->specs.size();
  int dep = -1;
  for (int i = 0; i < n; i++) {
    const int kind = g->specs[i][0];
    if (kind == 1 || kind == 3) {
      dep = i;
      continue;
    }
    // a submanifold view needs the grid only; a deconvolution view the strided rulebook built with it
    bool early = false;
    if (dep >= 0) {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv.wait(lk, [&] { return g->ready[dep] || (kind == 0 && g->grid_ready[dep]) || g->rc != D3D_OK; });
      if (!g->ready[dep] && !(kind == 0 && g->grid_ready[dep])) return;
      early = !g->ready[dep];
    }
    const int *sp = g->specs[i].data() + 1;
    hipStream_t on = g->view_stream;
    if (rc == D3D_OK && dep >= 0 && hipStreamWaitEvent(on, early ? g->gev[dep] : g->ev[dep], 0) != hipSuccess) {
      set_error("geometry thread: hipStreamWaitEvent failed");
      rc = D3D_ERR_HIP;
    }
    if (rc == D3D_OK)
      rc = kind == 0 ? d3d_subm_prepare(m, sp, sp + 6, on, nullptr)
                     : d3d_deconv_prepare(m, sp, sp + 3, sp + 6, sp + 9, on, nullptr);
    if (rc == D3D_OK && hipEventRecord(g->ev[i], on) != hipSuccess) {
      set_error("geometry thread: hipEventRecord failed");
      rc = D3D_ERR_HIP;
    }
    if (!geo_publish(g, i, rc, 0)) return;
  }
}
}  // namespace d3d

extern "C" {

// (levels of the first chain: 1 = the first strided grid gets a read-back of its own, 0 = one chain for all levels)
int d3d_grid_chain_head(int levels) {
  const int was = g_chain_head;
  g_chain_head = levels < 0 ? 0 : levels;
  return was;
}
int d3d_grid_chain_enable(int on) {
  const int was = g_chain_enabled ? 1 : 0;
  g_chain_enabled = on != 0;
  return was;
}

int d3d_geometry_async_start(d3d_meta *m, const int *specs, int n, void *stream, void *view_stream) {
  D3D_REQUIRE(m && (n == 0 || specs) && n >= 0 && n <= 128, "geometry_async_start: bad arguments");
  D3D_REQUIRE(m->geo_locked && (hipStream_t)stream == m->geo_stream,
              "geometry_async_start: `stream` must be the metadata's geometry stream (d3d_meta_set_geometry_stream)");
  for (int i = 0; i < n; i++) {
    const int kind = specs[i * 13];
    D3D_REQUIRE(kind >= 0 && kind <= 3, "geometry_async_start: entry %d has kind %d (0 submanifold view, 1 strided grid, "
                "2 deconvolution view, 3 strided grid without a deconvolution view)", i, kind);
    D3D_REQUIRE(kind == 1 || kind == 3 || (view_stream && (hipStream_t)view_stream == m->plan_stream),
                "geometry_async_start: views need the metadata's plan stream (d3d_meta_set_plan_stream)");
  }
  geo_async_join(m);
  GeoAsync *g = (GeoAsync *)m->geo_async;
  if (!g) m->geo_async = g = new GeoAsync();
  g->specs.resize(n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 13; j++) g->specs[i][j] = specs[i * 13 + j];
  g->n_out.assign(n, 0);
  while ((int)g->ev.size() < n) {
    hipEvent_t e, e2;
    D3D_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    g->ev.push_back(e);
    D3D_HIP_CHECK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    g->gev.push_back(e2);
  }
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->ready.assign(n, 0);
    g->grid_ready.assign(n, 0);
    g->rc = D3D_OK;
    g->err.clear();
    g->stream = (hipStream_t)stream;
    g->view_stream = (hipStream_t)view_stream;
  }
  D3D_HIP_CHECK(hipGetDevice(&g->device));
  if (!g->threads_up) {
    g->th = std::thread(geo_worker, m, g, 0);
    g->th_views = std::thread(geo_worker, m, g, 1);
    g->threads_up = true;
  }
  std::lock_guard<std::mutex> lk(g->mu);
  g->job++;
  g->cv.notify_all();
  return D3D_OK;
}

int d3d_geometry_async_wait(d3d_meta *m, int index, int *n_out_host, void *wait_stream) {
  D3D_REQUIRE(m && m->geo_async, "geometry_async_wait: no chain was started");
  GeoAsync *g = (GeoAsync *)m->geo_async;
  D3D_REQUIRE(index >= 0 && index < (int)g->specs.size(), "geometry_async_wait: entry %d of %d", index, (int)g->specs.size());
  {
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [&] { return g->ready[index] || g->rc != D3D_OK; });
    if (!g->ready[index]) {
      set_error("geometry thread: %s", g->err.c_str());
      return g->rc;
    }
    if (n_out_host) *n_out_host = g->n_out[index];
  }
  D3D_HIP_CHECK(hipStreamWaitEvent((hipStream_t)wait_stream, g->ev[index], 0));
  return D3D_OK;
}

int d3d_geometry_async_finish(d3d_meta *m) {
  D3D_REQUIRE(m, "null metadata");
  geo_async_join(m);
  GeoAsync *g = (GeoAsync *)m->geo_async;
  if (g && g->rc != D3D_OK) {
    set_error("geometry thread: %s", g->err.c_str());
    return g->rc;
  }
  return D3D_OK;
}

int d3d_deconv_prepare(d3d_meta *m, const int *in_size, const int *out_size, const int *filt,
                       const int *stride, void *stream, long *n_rules_host) {
  (void)in_size;
  D3D_REQUIRE(m && out_size && filt && stride, "null argument");
  const Plan *p = nullptr;
  int rc = get_deconv_plan(m, out_size, filt, stride, (hipStream_t)stream, &p);
  if (rc) return rc;
  if (n_rules_host) return plan_rules(m, *const_cast<Plan *>(p), (hipStream_t)stream, n_rules_host);
  return D3D_OK;
}

int d3d_export_rules(d3d_meta *m, int kind, const int *in_size, const int *filt, const int *stride,
                     int32_t *triples, long capacity, long *n_host, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && in_size && filt && n_host && (kind == 0 || kind == 1 || kind == 2), "bad arguments");
  const Plan *p = find_plan(m, kind, in_size, filt, kind == 0 ? nullptr : stride);
  if (!p) {
    set_error("export: rulebook not built");
    return D3D_ERR_STATE;
  }
  *n_host = 0;
  if (p->n_rows == 0) return D3D_OK;
  Arena &A = lane_arena(m, s);
  size_t mark = A.used;
  D3D_ALLOC(cnt, unsigned long long, A, 1);
  D3D_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(unsigned long long), s));
  const int npos = p->n_blk * 32;
  // kind 2 (deconvolution plan): rows are fine sites = the rule's "in" side of the strided conv
  hipLaunchKernelGGL(k_export_plan, grid1d((long)npos * p->K), dim3(256), 0, s, p->nbrT, p->rows, npos, p->K, kind == 2 ? 1 : 0, triples, capacity, cnt);
  D3D_LAUNCH_CHECK();
  D3D_HIP_CHECK(hipMemcpyAsync(&m->host_words[9], cnt, sizeof(long), hipMemcpyDeviceToHost, s));
  D3D_HIP_CHECK(hipStreamSynchronize(s));
  *n_host = m->host_words[9];
  A.used = mark;
  return D3D_OK;
}

// debug / measurement: how well a plan's blocks are grouped.  executed = sum over blocks of 32 * popcount(block mask)
// (row-offset steps the convolution runs), useful = rules.  executed / useful = 1 is a perfect grouping.
int d3d_plan_stats(d3d_meta *m, int kind, const int *in_size, const int *filt, const int *stride, long *n_blocks_host,
                   long *executed_host, long *rules_host, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && in_size && filt && n_blocks_host && executed_host && rules_host, "bad arguments");
  Plan *p = const_cast<Plan *>(find_plan(m, kind, in_size, filt, kind == 0 ? nullptr : stride));
  if (!p) {
    set_error("plan_stats: rulebook not built");
    return D3D_ERR_STATE;
  }
  *n_blocks_host = p->n_blk;
  *executed_host = 0;
  int rc = plan_rules(m, *p, s, rules_host);
  if (rc || p->n_blk == 0) return rc;
  std::vector<uint32_t> bm(p->n_blk);
  D3D_HIP_CHECK(hipMemcpyAsync(bm.data(), p->blkmask, sizeof(uint32_t) * p->n_blk, hipMemcpyDeviceToHost, s));
  D3D_HIP_CHECK(hipStreamSynchronize(s));
  long ex = 0;
  for (uint32_t v : bm) ex += 32L * __builtin_popcount(v);
  *executed_host = ex;
  return D3D_OK;
}

int d3d_sparse_to_dense_forward(d3d_meta *m, const int *size, const float *in, int planes, int batch,
                                float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && size && out && planes > 0 && batch > 0, "bad arguments");
  Grid *g = find_grid(m, size);
  if (!g) {
    set_error("sparse_to_dense: no grid of spatial size [%d,%d,%d]", size[0], size[1], size[2]);
    return D3D_ERR_STATE;
  }
  size_t vol = (size_t)size[0] * size[1] * size[2];
  D3D_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(float) * vol * planes * batch, s));
  if (g->n == 0) return D3D_OK;
  D3D_REQUIRE(in, "null input");
  hipLaunchKernelGGL(k_sparse_to_dense, grid1d((long)g->n * planes), dim3(256), 0, s, in, planes, g->loc, g->n, size[0], size[1], size[2], out);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_sparse_to_dense_backward(d3d_meta *m, const int *size, const float *d_out, int planes, float *d_in,
                                 void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && size && planes > 0, "bad arguments");
  Grid *g = find_grid(m, size);
  if (!g) {
    set_error("sparse_to_dense backward: no grid of spatial size [%d,%d,%d]", size[0], size[1], size[2]);
    return D3D_ERR_STATE;
  }
  if (g->n == 0) return D3D_OK;
  D3D_REQUIRE(d_out && d_in, "null pointer");
  hipLaunchKernelGGL(k_sparse_to_dense_bwd, grid1d((long)g->n * planes), dim3(256), 0, s, d_out, planes, g->loc, g->n,
                     size[0], size[1], size[2], d_in);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

}  // extern "C"
