// Internal declarations shared by the translation units of libd3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/d3d_hip.h"

namespace d3d {

void set_error(const char *fmt, ...);

#define D3D_HIP_CHECK(expr)                                                          \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      d3d::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return D3D_ERR_HIP;                                                            \
    }                                                                                \
  } while (0)

#define D3D_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      d3d::set_error(__VA_ARGS__);    \
      return D3D_ERR_ARG;             \
    }                                 \
  } while (0)

#define D3D_LAUNCH_CHECK() D3D_HIP_CHECK(hipGetLastError())

// One HBM slab, bump-allocated; everything a scene needs (hash grids, rulebooks, scratch)
// lives here and is dropped by d3d_meta_clear.  Stream-ordered use on ONE stream.
struct Arena {
  char *base = nullptr;
  size_t cap = 0, used = 0;
  void *alloc(size_t bytes) {
    size_t off = (used + 255) & ~size_t(255);
    if (off + bytes > cap) return nullptr;
    used = off + bytes;
    return base + off;
  }
  template <typename T>
  T *get(size_t count) {
    return static_cast<T *>(alloc(count * sizeof(T)));
  }
};

#define D3D_ALLOC(ptr, T, arena, count)                                                \
  T *ptr = (arena).get<T>(count);                                                      \
  if (!ptr) {                                                                          \
    d3d::set_error("metadata arena exhausted (%zu of %zu bytes used, need %zu more)", \
                   (arena).used, (arena).cap, (size_t)(count) * sizeof(T));            \
    return D3D_ERR_NOMEM;                                                              \
  }

static constexpr uint64_t kEmptyKey = ~uint64_t(0);

// One 16-byte hash slot: a probe is a single aligned 16-B load.  Initialised to all ones
// (key = empty, val = -1, first = UINT_MAX).  `first` = smallest index that touched the slot
// while the grid was being built (first-occurrence numbering).
struct __attribute__((aligned(16))) HashEntry {
  uint64_t key;
  int32_t val;
  uint32_t first;
};

// Sparse grid of one spatial size: open-addressing hash (key -> site id) + site coordinates.
struct Grid {
  int size[3] = {0, 0, 0};
  int n = 0;        // active sites
  int cap = 0;      // hash capacity (power of two)
  HashEntry *tab = nullptr;
  int32_t *loc = nullptr;  // [n,4] x,y,z,b
  int32_t *extent = nullptr;  // device int[3]: 1 + max coordinate per axis (filled on first use, grid_extent)
  int32_t *dense = nullptr;    // 1 + site id per cell of the box [hext[3]][hext[0]][hext[1]][hext[2]], 0 = empty (grid_dense_index)
  int hext[4] = {0, 0, 0, 0};  // host-side UPPER BOUNDS of 1 + max x, y, z, example index (0: unknown); bound the site counts
                               // of the grids built from this one (grid chain)
};

// BatchNorm affine + leaky ReLU on 4 channels: y = leaky(fma(x, w, b)).  ONE definition for k_bn_apply and the fused
// prologue of the convolutions, so that fused and unfused inference give the same bits.  Explicit fma (the build
// uses -ffp-contract=off) and, for leakiness 0, max(t, 0): 1.5 VALU instructions per element instead of 3.5.
typedef float d3d_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d3d_f32x4 bn_act(d3d_f32x4 x, d3d_f32x4 w, d3d_f32x4 b, float leak) {
  d3d_f32x4 t;
#pragma unroll
  for (int j = 0; j < 4; j++) t[j] = __builtin_fmaf(x[j], w[j], b[j]);
  if (leak == 0.f) {
#pragma unroll
    for (int j = 0; j < 4; j++) asm("v_max_f32 %0, 0, %1" : "=v"(t[j]) : "v"(t[j]));  // fmaxf would add a canonicalising max
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) t[j] = t[j] > 0.f ? t[j] : t[j] * leak;
  }
  return t;
}

// Output-stationary rulebook ("plan") of one convolution:
//   rows   [n_blk*32]      output row of every position, sorted by neighbour mask, -1 padded
//   nbrT   [K][n_blk*32]   input row feeding position p through filter offset k, or -1
//   blkmask[n_blk]         OR over the block's 32 positions of their offset bit masks
struct Plan {
  int K = 0;
  int n_rows = 0;
  int n_in = 0;                               // rows of the tensor the plan gathers from
  int n_blk = 0;
  long n_rules = 0;                           // -1: not counted yet (plan_rules counts on demand)
  int32_t *rows = nullptr;
  int32_t *nbrT = nullptr;
  uint32_t *blkmask = nullptr;
};

typedef std::array<int, 3> Size3;
typedef std::array<int, 10> PlanKey;  // kind, in_size[3], filter[3], stride[3]

struct StridedRaw {  // kept so that the deconvolution plan can be finalised lazily
  int32_t *nbr_dec = nullptr;  // [n_in][K]
  int n_in = 0;
  Size3 out_size;
};

// order-preserving integer image of a float (larger float <=> larger integer; NaNs above +inf)
__device__ __forceinline__ uint32_t f32_ordered(float v) {
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// BoxCoder3D.decode of one row (box_coder_3d.py:38-65 / box_torch_ops.py:51-88, smooth_dim): e[7] encoding, a[7] anchor
// (xa, ya, za, wa, la, ha, ra), w[7] the coder's weights, size deltas clamped to `clip`; yaw wrapped to [-pi/2, pi/2).
__device__ __forceinline__ void box_decode_one(const float *e, const float *a, const float *w, float clip, float *o) {
  float t[7];
#pragma unroll
  for (int k = 0; k < 7; k++) t[k] = e[k] / w[k];
#pragma unroll
  for (int k = 3; k < 6; k++) t[k] = fminf(t[k], clip);
  const float xa = a[0], ya = a[1], za = a[2], wa = a[3], la = a[4], ha = a[5], ra = a[6];
  const float diagonal = sqrtf(la * la + wa * wa);
  o[0] = t[0] * diagonal + xa;
  o[1] = t[1] * diagonal + ya;
  o[2] = t[2] * ha + za;
  o[3] = (t[3] + 1) * wa;
  o[4] = (t[4] + 1) * la;
  o[5] = (t[5] + 1) * ha;
  const float pi = 3.14159265358979323846f;
  const float rg = t[6] + ra;
  o[6] = rg - floorf(rg / pi + 0.5f) * pi;
}

}  // namespace d3d

struct d3d_meta {
  d3d::Arena arena;       // first 2/3 of the slab: grids, rulebooks and the temporaries of building them
  d3d::Arena feat_arena;  // last 1/3: temporaries of the feature pass (offset-split partial tiles, rule counts) and,
                          // while a geometry stream is set, the rulebooks built on any other stream (grid.hip
                          // lane_arena) -- each part is only ever used by one stream at a time
  bool geo_locked = false;        // d3d_meta_set_geometry_stream: builds are only accepted on geo_stream
  hipStream_t geo_stream = nullptr;
  d3d::Arena plan_arena;          // third lane, carved out of feat_arena by d3d_meta_set_plan_stream: the submanifold /
  hipStream_t plan_stream = nullptr;  // deconvolution rulebooks built on plan_stream while both other streams work
  size_t feat_cap_full = 0;       // feat_arena.cap before the carve (restored by d3d_meta_clear)
  std::map<d3d::Size3, d3d::Grid> grids;
  std::map<d3d::PlanKey, d3d::Plan> plans;
  std::map<d3d::PlanKey, d3d::StridedRaw> strided_raw;
  // input layer
  int in_n = 0, in_mode = 0, in_active = 0;
  int32_t *in_off = nullptr, *in_idx = nullptr;
  int32_t *in_pslot = nullptr;   // hash slot of every input point (input for the point lists, built on first use)
  bool in_lists = false;         // in_off / in_idx filled (ensure_point_lists)
  char *pl_scratch = nullptr;    // temporaries of the point lists, set aside by the input-layer build: the lists can then
  size_t pl_scratch_bytes = 0;   // be sorted on ANY stream (no lane of the arena is touched)
  d3d::Size3 in_size = {0, 0, 0};  // spatial size of the input grid
  int in_ext[4] = {0, 0, 0, 0};    // 1 + largest x, y, z, example index of the input points (read back with the site count)
  int32_t *iota = nullptr;         // 0 .. in_n - 1 (values of the plan sorts), written once per scene
  int iota_n = 0;
  // pinned host words for size read-backs
  long *host_words = nullptr;
  int32_t *host_counts = nullptr;  // pinned int32[32]: the site counts of a grid chain, stored by k_store_counts
  hipEvent_t chain_ev = nullptr;   // recorded behind that store
  // neighbour table of the input grid started by d3d_input_layer_build_prefetch while the site count was read back
  // (top of the geometry lane, which is shortened by it until d3d_meta_clear); consumed by d3d_subm_prepare
  int32_t *pre_nbr = nullptr;
  uint32_t *pre_mask = nullptr;
  int pre_filt[3] = {0, 0, 0};
  hipStream_t pre_stream = nullptr;
  // ... and, sized by the point count like the table, the rest of that rulebook (sort by mask, transpose): enqueued
  // behind the probes before the host has the site count; registered as a plan once the count is back
  bool pre_plan_built = false;
  d3d::Plan pre_plan;              // (rows / blocks / gathered rows are filled in when the count is back)
  void *pre_tab = nullptr;         // the input grid's hash table (known before the grid is registered)
  // the input layer's point lists, built on a stream of the library's own right behind the grid (no count needed:
  // everything is sized by the point count); consumers wait for lists_ev
  hipStream_t aux_stream = nullptr;
  hipEvent_t grid_ev = nullptr, lists_ev = nullptr, fill_ev = nullptr;
  bool lists_on_aux = false;
  hipEvent_t count_ev = nullptr;   // marks the count copy, so that the host does not wait for the prefetch behind it
  size_t arena_cap_full = 0;
  // the three maps above may be read by the caller's thread while the geometry thread (d3d_geometry_async_start) adds
  // to them: every lookup / insertion holds `mu` (map nodes do not move, so the pointers handed out stay valid)
  std::recursive_mutex mu;
  void *geo_async = nullptr;       // grid.hip GeoAsync: the running / last geometry chain of this metadata
};
#define D3D_LOCK(m) std::lock_guard<std::recursive_mutex> d3d_lock_(m->mu)

namespace d3d {

// utilities implemented in grid.hip
int scan_exclusive_i32(const int32_t *in, int32_t *out, int n, int32_t *total_dev, Arena &scratch,
                       hipStream_t s);
int sort_pairs_u32(const uint32_t *keys_in, uint32_t *keys_out, const int32_t *vals_in,
                   int32_t *vals_out, int n, int bits, Arena &scratch, hipStream_t s, bool descending,
                   const int32_t *n_dev = nullptr);
int finalize_plan(d3d_meta *m, const int32_t *nbr, int n_rows, int K, Plan &plan, hipStream_t s,
                  uint32_t *mask_in);
int plan_rules(d3d_meta *m, Plan &p, hipStream_t s, long *out);
const Plan *find_plan(d3d_meta *m, int kind, const int *in_size, const int *filt, const int *stride);
int get_deconv_plan(d3d_meta *m, const int *fine_size, const int *filt, const int *stride,
                    hipStream_t s, const Plan **out);

int grid_extent(d3d_meta *m, Grid &g, hipStream_t s);  // grid.hip: ensures g.extent (and, for small boxes, g.dense)
int check_build_stream(d3d_meta *m, hipStream_t s, const char *what);
int ensure_point_lists(d3d_meta *m, hipStream_t s);
Arena &lane_arena(d3d_meta *m, hipStream_t s);

// conv.hip
struct BnPre {  // fused BatchNorm(+leaky ReLU) prologue of a convolution; mean == nullptr: none
  const float *mean, *invstd, *weight, *bias;
  float leak;
};
void conv_timing_take(hipEvent_t *start, hipEvent_t *stop);   // events armed by d3d_conv_time_next (or nullptr), disarms
// conv_bf16.hip
int launch_conv_bf16(d3d_meta *m, const Plan &p, const void *in, int cin, const void *packed_w, int cout,
                     const void *residual, void *out, hipStream_t s, const d3d_bn_prologue *bn);
int launch_conv(d3d_meta *m, const Plan &p, const float *in, int cin, const float *packed_w, int cout,
                const float *residual, float *out, hipStream_t s, const d3d_bn_prologue *bn = nullptr);
// conv_ws.hip: the weight-sharing kernel for the large launches of the wide layers; false = not taken
bool launch_conv_ws(const Plan &p, const float *in, int cin, const float *wp, int cout, const float *residual, float *out,
                    hipStream_t s, BnPre pre, double *stat, uint32_t in_bytes);

__device__ __forceinline__ uint64_t pack_key(int b, int x, int y, int z) {
  return ((uint64_t)(uint16_t)b << 48) | ((uint64_t)(uint16_t)x << 32) |
         ((uint64_t)(uint16_t)y << 16) | (uint64_t)(uint16_t)z;
}
// Locality-preserving slot choice: the 2x2x2 cell block a voxel belongs to is hashed to a group of 8
// consecutive 16-B slots (one 128-B line) and the voxel's position inside the block picks the slot, so
// the 27 neighbour probes of a site touch <= 8 lines (and the 8 trilinear corners of a sample <= 8,
// usually 1-2) instead of one random line each.  Collisions step by whole groups.
static constexpr uint64_t kLowBits = 0x0000000100010001ULL;  // bit 0 of x, y, z in the packed key
__device__ __forceinline__ uint32_t hash_key(uint64_t k) {
  const uint32_t intra = (uint32_t)((k >> 32) & 1) << 2 | (uint32_t)((k >> 16) & 1) << 1 | (uint32_t)(k & 1);
  k &= ~kLowBits;
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return ((uint32_t)k << 3) | intra;
}
static constexpr uint32_t kProbeStep = 8;
// Probe sequence: whole groups (same position inside the 2x2x2 block) first; after one full round over the cap/8
// groups the position moves on by one slot, so that the sequence reaches every slot of the table and an insertion
// terminates for ANY key set up to the table's capacity (a scene whose voxels all share one coordinate parity would
// otherwise fill its eighth of the table and probe for ever).
__device__ __forceinline__ uint32_t probe_next(uint32_t slot, uint32_t &round, int cap) {
  slot = (slot + kProbeStep) & (uint32_t)(cap - 1);
  if (++round == (uint32_t)cap / kProbeStep) {
    round = 0;
    slot = (slot + 1) & (uint32_t)(cap - 1);
  }
  return slot;
}
// returns the slot holding `key` (inserting it if absent)
__device__ __forceinline__ int hash_insert(HashEntry *tab, int cap, uint64_t key) {
  uint32_t slot = hash_key(key) & (uint32_t)(cap - 1), round = 0;
  while (true) {
    unsigned long long prev = atomicCAS((unsigned long long *)&tab[slot].key,
                                        (unsigned long long)kEmptyKey, (unsigned long long)key);
    if (prev == kEmptyKey || prev == key) return (int)slot;
    slot = probe_next(slot, round, cap);
  }
}
// returns the site id stored for `key`, or -1 (one 16-B load per probed slot)
__device__ __forceinline__ int hash_find(const HashEntry *__restrict__ tab, int cap, uint64_t key) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  uint32_t slot = hash_key(key) & (uint32_t)(cap - 1), round = 0;
  while (true) {
    const u32x4 e = *(const u32x4 *)&tab[slot];
    const uint64_t k = ((uint64_t)e[1] << 32) | e[0];
    if (k == key) return (int)e[2];
    if (k == kEmptyKey) return -1;
    slot = probe_next(slot, round, cap);
  }
}

// hash_find for N independent keys of a lane in lockstep: every probe round issues the N slot loads together, so the
// round trips of the keys overlap instead of following one another (each hash_find is a loop of its own, which the
// compiler does not interleave with the next one).  out[i] = site id or -1; keys with want[i] == false give -1.
template <int N>
__device__ __forceinline__ void hash_find_n(const HashEntry *__restrict__ tab, int cap, const uint64_t (&key)[N],
                                            const bool (&want)[N], int (&out)[N]) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  uint32_t slot[N], round[N];
  bool live[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    slot[i] = hash_key(key[i]) & (uint32_t)(cap - 1);
    round[i] = 0;
    live[i] = want[i];
    out[i] = -1;
  }
  while (true) {
    u32x4 e[N];
#pragma unroll
    for (int i = 0; i < N; i++) e[i] = *(const u32x4 *)&tab[live[i] ? slot[i] : 0u];   // unconditional: no branch per key
    bool more = false;
#pragma unroll
    for (int i = 0; i < N; i++) {
      if (!live[i]) continue;
      const uint64_t k = ((uint64_t)e[i][1] << 32) | e[i][0];
      if (k == key[i]) {
        out[i] = (int)e[i][2];
        live[i] = false;
      } else if (k == kEmptyKey) {
        live[i] = false;
      } else {
        slot[i] = probe_next(slot[i], round[i], cap);
        more = true;
      }
    }
    if (!more) break;
  }
}

}  // namespace d3d
