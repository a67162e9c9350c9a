// Exact top-k selection with a DEFINED order, and the reference-shaped rotated NMS entry built on it.
//
//   RPNPostProcessor.forward_for_single_feature_map   maskrcnn_benchmark/modeling/rpn/inference_3d.py:105-123
//       objectness.sigmoid() -> topk(pre_nms_top_n, sorted=True) -> gather regression / anchors -> decode
//   rotate_nms_3d                                     second/pytorch/core/box_torch_ops.py:489-514
//       topk(pre_max_size) -> rotate_nms_3d_cc (nms_cpu.py:32-44: argsort of -scores, greedy sweep) -> [:post_max_size]
//
// torch.topk / numpy's argsort leave the order among EQUAL scores to the implementation (and sigmoid saturates to exactly
// 1.0 for large logits, so ties are not exotic).  Here it is defined -- descending score, lower index first -- and
// oracle/detector_port.py follows the same rule.  One 1024-thread workgroup per segment:
//   1. radix select (11 + 11 + 10 bits, histograms in LDS) of the k-th largest order-preserving key T;
//   2. every element above T, and of those equal to T the ones with the lowest indices, collected into LDS as
//      (key << 32 | ~index) words;
//   3. bitonic sort of that list (<= 4096 words) in LDS, descending: key descending, index ascending.  Compare-exchange
//      steps whose partners lie within 128 elements stay inside one wave and need no workgroup barrier.
//   4. optional: the selected rows' box decode (BoxCoder3D.decode) written next to indices and scores.
// What it replaces in the RPN stage: sigmoid, a rocPRIM block sort + six merge passes + copies (~15 launches) and a
// gather + decode launch.
#include <algorithm>

#include <hip/hip_runtime.h>

#include "d3d_internal.h"

namespace d3d {

static constexpr int kTopkThreads = 1024, kTopkMax = 4096;

struct TopkArgs {
  const float *vals;        // element i of group g: vals[g * group_stride + i * elem_stride]
  int n, elem_stride, group_stride, n_groups;
  const int32_t *example;   // null, or the example index of every element (segment (b, g) takes example[i] == b)
  int k, apply_sigmoid;
  const float *reg;         // null, or regression rows: reg[i * reg_stride + 7 g .. + 7]
  int reg_stride;
  const float *anchors;     // [n, 7]
  float clip;
  int use_min;              // counts[s] = selected elements with value > min_value (a prefix of the sorted list)
  float min_value;
  int idx_mul, idx_add, idx_add_group;   // idx32 output = i * idx_mul + idx_add + g * idx_add_group
  int32_t *idx32;           // [S, k] outputs (each may be null); rows past min(k, elements) are not written
  int64_t *idx64;
  float *scores;
  float *props;             // [S, k, 7]
  int32_t *counts;          // [S] = min(k, elements of the segment)
  uint32_t *keys;           // null, or scratch [S, n4] (n4 = n rounded up to 4): the ordered keys, written by the first pass
                            // and re-read by the later ones with 16-byte loads (elements of other examples: key 0)
  int n4;
};

__device__ __forceinline__ float topk_value(const TopkArgs &a, int i, int g) {
  float x = a.vals[(size_t)g * a.group_stride + (size_t)i * a.elem_stride];
  if (a.apply_sigmoid) x = 1.0f / (1.0f + expf(-x));     // torch's sigmoid formula, IEEE division
  return x;
}

// block-wide: the bin (counted from the TOP) at which the running count reaches `want`; hist[NB] in LDS.
// -> sel[0] = bin, sel[1] = want - (count of the bins above it).  All threads call it; result valid after the barrier.
template <int NB>
__device__ __forceinline__ void find_bin_from_top(const uint32_t *hist, uint32_t want, uint32_t *wtot, uint32_t *sel) {
  constexpr int PER = NB / kTopkThreads;     // bins per thread (2 or 1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t loc[PER], sum = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    loc[j] = hist[NB - 1 - (tid * PER + j)];  // reversed: thread 0 holds the top bins
    sum += loc[j];
  }
  uint32_t inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  uint32_t before = inc - sum;
  for (int w = 0; w < wave; w++) before += wtot[w];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    if (before < want && want <= before + loc[j]) {
      sel[0] = (uint32_t)(NB - 1 - (tid * PER + j));
      sel[1] = want - before;
    }
    before += loc[j];
  }
  __syncthreads();
}

__global__ __launch_bounds__(kTopkThreads) void k_topk_select(TopkArgs a) {
  __shared__ uint32_t hist[2048];
  __shared__ unsigned long long sel[kTopkMax];
  __shared__ uint32_t wtot[kTopkThreads / 64];
  __shared__ uint32_t pick[2];
  __shared__ uint32_t n_sel, n_valid_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seg = blockIdx.x, g = seg % a.n_groups, b = seg / a.n_groups;
  const int n = a.n;
  auto valid = [&](int i) { return a.example == nullptr || a.example[i] == b; };

  // Key of element i: the order-preserving image of its value, at least 1; 0 marks an element of another example.
  // With key scratch the values are read (and the sigmoid evaluated) ONCE: the first pass stores the keys, the later
  // ones re-read them four per 16-byte load with 16 loads in flight per thread -- one workgroup walking ~10^5 elements
  // one dependent load at a time is bound by that load's latency (measured: 100 us for 150 k anchors, all four passes).
  uint32_t *const kbuf = a.keys ? a.keys + (size_t)seg * a.n4 : nullptr;
  auto key_of = [&](int i) -> uint32_t {
    if (!valid(i)) return 0u;
    return max(f32_ordered(topk_value(a, i, g)), 1u);
  };
  // f(key, index) over all elements, in batches of independent loads
  auto for_each_key = [&](auto f) {
    if (kbuf) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      constexpr int U = 4;
      const int nq = a.n4 / 4;                                  // 16-byte groups
      for (int q0 = 0; q0 < nq; q0 += kTopkThreads * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int q = q0 + u * kTopkThreads + tid;
          v[u] = q < nq ? *(const u32x4 *)(kbuf + (size_t)q * 4) : (u32x4){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int q = q0 + u * kTopkThreads + tid;
#pragma unroll
          for (int j = 0; j < 4; j++) f(v[u][j], q * 4 + j);
        }
      }
    } else {
      for (int i0 = 0; i0 < n; i0 += kTopkThreads) {
        const int i = i0 + tid;
        f(i < n ? key_of(i) : 0u, i);
      }
    }
  };
  // wave-aggregated histogram add: the lanes that share the first active lane's bin add once, together (scores of one
  // map crowd into few bins -- a random-init head puts every sigmoid next to 0.5 -- and 64 lanes adding to one LDS word
  // serialise)
  // A wave counts the run of elements that fall into ONE bin in scalar registers (wbin, wcnt) and adds to the LDS
  // histogram only when the bin changes (hist_flush at the end of a pass): with crowded scores a pass costs a wave a
  // handful of LDS atomics instead of one per 64 elements, all of them on the same word.
  uint32_t wbin = 0xffffffffu, wcnt = 0;
  auto hist_flush = [&]() {
    if (wcnt && lane == 0) atomicAdd(&hist[wbin], wcnt);
    wcnt = 0;
    wbin = 0xffffffffu;
  };
  auto hist_add = [&](bool in, uint32_t bin) {
    const unsigned long long todo = __ballot(in);
    if (todo) {
      const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, __builtin_ctzll(todo));
      const unsigned long long same = __ballot(in && bin == b0);
      if (b0 != wbin) {
        if (wcnt && lane == 0) atomicAdd(&hist[wbin], wcnt);
        wbin = b0;
        wcnt = 0;
      }
      wcnt += (uint32_t)__popcll(same);
      if (in && bin != b0) atomicAdd(&hist[bin], 1u);
    }
  };

  if (tid == 0) {
    n_valid_s = 0;
    n_sel = 0;
  }
  for (int i = tid; i < 2048; i += kTopkThreads) hist[i] = 0;
  __syncthreads();
  // pass 0: keys (stored when there is scratch), the segment's element count and the first histogram (bits 31..21)
  {
    uint32_t c = 0;
    if (kbuf) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      constexpr int U = 4;                                      // 16 independent value loads in flight per thread
      const int nq = a.n4 / 4;
      for (int q0 = 0; q0 < nq; q0 += kTopkThreads * U) {
        float x[U][4];
        bool ok[U][4];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int i = (q0 + u * kTopkThreads + tid) * 4 + j;
            ok[u][j] = i < n && valid(i);
            x[u][j] = ok[u][j] ? a.vals[(size_t)g * a.group_stride + (size_t)i * a.elem_stride] : 0.f;
          }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int q = q0 + u * kTopkThreads + tid;
          u32x4 kv;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            float v = x[u][j];
            if (a.apply_sigmoid) v = 1.0f / (1.0f + expf(-v));
            kv[j] = ok[u][j] ? max(f32_ordered(v), 1u) : 0u;
          }
          if (q < nq) *(u32x4 *)(kbuf + (size_t)q * 4) = kv;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            c += kv[j] != 0u;
            hist_add(kv[j] != 0u, kv[j] >> 21);
          }
        }
      }
    } else {
      for (int i0 = 0; i0 < n; i0 += kTopkThreads) {
        const int i = i0 + tid;
        const uint32_t key = i < n ? key_of(i) : 0u;
        c += key != 0u;
        hist_add(key != 0u, key >> 21);
      }
    }
    hist_flush();
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0 && c) atomicAdd(&n_valid_s, c);
  }
  __syncthreads();
  const int k_eff = min(a.k, (int)n_valid_s);
  if (tid == 0 && a.counts && (!a.use_min || k_eff == 0)) a.counts[seg] = k_eff;
  if (k_eff == 0) return;

  // 1. the k_eff-th largest key: three radix levels (11 + 11 + 10 bits); the first histogram is pass 0's
  uint32_t prefix = 0, want = (uint32_t)k_eff;
  find_bin_from_top<2048>(hist, want, wtot, pick);
  prefix = pick[0] << 21;
  want = pick[1];
  __syncthreads();
#pragma unroll
  for (int level = 1; level < 3; level++) {
    const int shift = level == 1 ? 10 : 0, bits = level == 1 ? 11 : 10;
    const uint32_t fixed_mask = 0xffffffffu << (shift + bits);
    for (int i = tid; i < 2048; i += kTopkThreads) hist[i] = 0;
    __syncthreads();
    for_each_key([&](uint32_t key, int) {
      hist_add(key != 0u && (key & fixed_mask) == prefix, (key >> shift) & ((1u << bits) - 1u));
    });
    hist_flush();
    __syncthreads();
    if (level == 1) find_bin_from_top<2048>(hist, want, wtot, pick);
    else find_bin_from_top<1024>(hist, want, wtot, pick);
    prefix |= pick[0] << shift;
    want = pick[1];
    __syncthreads();
  }
  const uint32_t T = prefix;                 // the k_eff-th largest key
  const uint32_t need_eq = want;             // how many elements equal to T belong to the top k_eff (>= 1)
  const uint32_t n_eq = hist[T & 1023u];     // level 3 fixed all 32 bits: this bin counts the keys equal to T
  __syncthreads();

  // 2. collect: keys above T (any order -- the sort follows) and the need_eq lowest-index keys equal to T
  const bool all_ties = n_eq == need_eq;
  for_each_key([&](uint32_t key, int i) {
    if (key > T || (all_ties && key == T && key != 0u)) {
      const uint32_t p = atomicAdd(&n_sel, 1u);
      sel[p] = ((unsigned long long)key << 32) | (uint32_t)(~(uint32_t)i);
    }
  });
  if (!all_ties) {
    // ties at the threshold: thread t owns the contiguous index range [t * per, t * per + per); ranks by a workgroup scan
    const int per = (n + kTopkThreads - 1) / kTopkThreads;
    const int i0 = min(n, tid * per), i1 = min(n, i0 + per);
    auto key_at = [&](int i) -> uint32_t { return kbuf ? kbuf[i] : key_of(i); };
    uint32_t mine = 0;
    for (int i = i0; i < i1; i++) mine += key_at(i) == T ? 1u : 0u;
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    __syncthreads();
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t rank = inc - mine;
    for (int w = 0; w < wave; w++) rank += wtot[w];
    for (int i = i0; i < i1 && rank < need_eq; i++) {
      if (key_at(i) == T) {
        const uint32_t p = atomicAdd(&n_sel, 1u);
        sel[p] = ((unsigned long long)T << 32) | (uint32_t)(~(uint32_t)i);
        rank++;
      }
    }
  }
  __syncthreads();
  // (n_sel == k_eff by construction)
  int P = 128;
  while (P < k_eff) P <<= 1;
  for (int i = k_eff + tid; i < P; i += kTopkThreads) sel[i] = 0ull;   // padding sorts last (a real word has ~index != 0)
  __syncthreads();

  // 3. bitonic sort, descending.  Pair t of a step with stride j: elements i = 2 j (t / j) + t % j and i + j.  For
  // j <= 64 the 64 pairs of a wave touch one 128-element window that no other wave touches: no workgroup barrier.
  int j_prev = 1 << 30;
  for (int k2 = 2; k2 <= P; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      if (j >= 128 || j_prev >= 128) __syncthreads();
      else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      for (int t = tid; t < P / 2; t += kTopkThreads) {
        const int i = 2 * j * (t / j) + (t % j), l = i + j;
        const bool desc = (i & k2) == 0;
        const unsigned long long x = sel[i], y = sel[l];
        if ((x < y) == desc) {
          sel[i] = y;
          sel[l] = x;
        }
      }
      j_prev = j;
    }
  }
  __syncthreads();

  // 4. outputs
  const float unit[7] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  for (int r = tid; r < k_eff; r += kTopkThreads) {
    const unsigned long long w = sel[r];
    const int i = (int)(~(uint32_t)w);
    const size_t o = (size_t)seg * a.k + r;
    if (a.use_min) {          // the list is sorted: the elements above min_value are a prefix
      const float v = topk_value(a, i, g);
      const bool in = v > a.min_value;
      bool next_in = false;
      if (r + 1 < k_eff) next_in = topk_value(a, (int)(~(uint32_t)sel[r + 1]), g) > a.min_value;
      if (in && !next_in) a.counts[seg] = r + 1;
      if (r == 0 && !in) a.counts[seg] = 0;
    }
    if (a.idx32) a.idx32[o] = i * a.idx_mul + a.idx_add + g * a.idx_add_group;
    if (a.idx64) a.idx64[o] = (int64_t)i;
    if (a.scores) a.scores[o] = topk_value(a, i, g);
    if (a.props) box_decode_one(a.reg + (size_t)i * a.reg_stride + 7 * g, a.anchors + (size_t)i * 7, unit, a.clip, a.props + o * 7);
  }
}

// keep int32 -> int64 (the reference returns a LongTensor), entries past the count untouched
__global__ void k_keep_to_i64(const int32_t *__restrict__ keep, const int32_t *__restrict__ n_keep, int cap,
                              int64_t *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap && i < *n_keep) out[i] = keep[i];
}

}  // namespace d3d

using namespace d3d;

extern "C" {

int d3d_topk_max(void) { return kTopkMax; }
size_t d3d_topk_scratch_bytes(int n, int segments) {
  return (size_t)std::max(segments, 0) * (size_t)((std::max(n, 0) + 3) & ~3) * sizeof(uint32_t);
}

int d3d_topk_segments(const float *vals, int n, int elem_stride, int group_stride, int n_groups, const int32_t *example,
                      int n_examples, int k, int apply_sigmoid, const float *min_value_host, const int *idx_map_host,
                      const float *reg, int reg_stride, const float *anchors, float clip, int32_t *idx32_out,
                      int64_t *idx64_out, float *scores_out, float *props_out, int32_t *counts_out, void *scratch,
                      size_t scratch_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(n >= 0 && n_groups >= 1 && n_examples >= 1 && elem_stride >= 1 && group_stride >= 0, "topk: bad shape");
  D3D_REQUIRE(k >= 0 && k <= kTopkMax, "topk: k = %d (at most %d)", k, kTopkMax);
  D3D_REQUIRE(counts_out, "topk: null counts");
  D3D_REQUIRE(n_examples == 1 || example, "topk: several examples need the example index of every element");
  const int S = n_groups * n_examples;
  if (n == 0 || k == 0) {
    D3D_HIP_CHECK(hipMemsetAsync(counts_out, 0, sizeof(int32_t) * S, s));
    return D3D_OK;
  }
  D3D_REQUIRE(vals, "topk: null values");
  D3D_REQUIRE(!props_out || (reg && anchors && reg_stride >= 7 * n_groups), "topk: decode needs regression rows and anchors");
  TopkArgs a;
  a.vals = vals; a.n = n; a.elem_stride = elem_stride; a.group_stride = group_stride; a.n_groups = n_groups;
  a.use_min = min_value_host != nullptr;
  a.min_value = min_value_host ? *min_value_host : 0.f;
  a.idx_mul = idx_map_host ? idx_map_host[0] : 1;
  a.idx_add = idx_map_host ? idx_map_host[1] : 0;
  a.idx_add_group = idx_map_host ? idx_map_host[2] : 0;
  a.example = n_examples > 1 ? example : nullptr;
  a.k = k; a.apply_sigmoid = apply_sigmoid;
  a.reg = reg; a.reg_stride = reg_stride; a.anchors = anchors; a.clip = clip;
  a.idx32 = idx32_out; a.idx64 = idx64_out; a.scores = scores_out; a.props = props_out; a.counts = counts_out;
  a.n4 = (n + 3) & ~3;
  a.keys = (scratch && scratch_bytes >= d3d_topk_scratch_bytes(n, S) && ((uintptr_t)scratch & 15) == 0) ? (uint32_t *)scratch
                                                                                                         : nullptr;
  hipLaunchKernelGGL(k_topk_select, dim3(S), dim3(kTopkThreads), 0, s, a);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

// (n: the number of input boxes -- the selection keeps their keys in the scratch; 0: without, slower for large n)
size_t d3d_rotate_nms_3d_scratch_bytes(int pre_max_size, int n) {
  const int k = std::max(0, std::min(pre_max_size, kTopkMax));
  return d3d_nms_batched_scratch_bytes(1, k) + (size_t)k * 8 + 1024 + 512 + d3d_topk_scratch_bytes(n, 1);
}

// rotate_nms_3d as the reference defines it (box_torch_ops.py:489-514 behind boxlist_nms_3d's size clamp,
// boxlist_ops_3d.py:14-62): boxes [n, 7] yx_zb, scores [n]; candidates = the pre_max_size best scores (ties: lower index
// first); sizes clamped for the IoU only (dy, dx >= aug_yx, dz >= aug_z); greedy rotated NMS at `thresh`; at most
// post_max_size survivors (<= 0: all).  keep_out int64 [min(n, pre_max_size)]: indices into the input, selection order;
// n_keep_dev int32 [1] on the device; n_keep_host (may be null): the count, read back (one stream synchronisation).
int d3d_rotate_nms_3d(const float *boxes, const float *scores, int n, int pre_max_size, int post_max_size, float thresh,
                      float aug_yx, float aug_z, int64_t *keep_out, int32_t *n_keep_dev, int *n_keep_host, void *scratch,
                      size_t scratch_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(n >= 0 && n_keep_dev, "rotate_nms_3d: bad arguments");
  const int k = std::min(n, pre_max_size > 0 ? pre_max_size : n);
  D3D_REQUIRE(k <= kTopkMax, "rotate_nms_3d: %d candidates (pass pre_max_size <= %d; the reference uses 2000)", k, kTopkMax);
  if (n_keep_host) *n_keep_host = 0;
  if (k == 0) {
    D3D_HIP_CHECK(hipMemsetAsync(n_keep_dev, 0, sizeof(int32_t), s));
    return D3D_OK;
  }
  D3D_REQUIRE(boxes && scores && keep_out && scratch && scratch_bytes >= d3d_rotate_nms_3d_scratch_bytes(k, 0),
              "rotate_nms_3d: bad buffers");
  char *base = (char *)scratch;
  int32_t *order = (int32_t *)base;                       // [k] candidates, descending score
  int32_t *keep32 = order + k;                            // [k]
  int32_t *cnt = keep32 + k;                              // [1] candidates (= k)
  const size_t off = (((size_t)k * 8 + 64) + 255) & ~size_t(255);
  // (key scratch for the selection: what lies behind the NMS's own scratch, if the caller provided that much)
  const size_t nms_bytes = d3d_nms_batched_scratch_bytes(1, k);
  const size_t key_off = (off + nms_bytes + 255) & ~size_t(255);
  void *key_scratch = scratch_bytes >= key_off + d3d_topk_scratch_bytes(n, 1) ? base + key_off : nullptr;
  int rc = d3d_topk_segments(scores, n, 1, 0, 1, nullptr, 1, k, 0, nullptr, nullptr, nullptr, 0, nullptr, 0.f, order, nullptr,
                             nullptr, nullptr, cnt, key_scratch, key_scratch ? d3d_topk_scratch_bytes(n, 1) : 0, stream);
  if (rc) return rc;
  rc = d3d_rotate_nms_3d_batched(boxes, order, k, cnt, 1, k, thresh, aug_yx, aug_z, post_max_size > 0 ? post_max_size : 0,
                                 keep32, n_keep_dev, base + off, scratch_bytes - off, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(k_keep_to_i64, dim3((k + 255) / 256), dim3(256), 0, s, keep32, n_keep_dev, k, keep_out);
  D3D_LAUNCH_CHECK();
  if (n_keep_host) {
    int32_t h = 0;
    D3D_HIP_CHECK(hipMemcpyAsync(&h, n_keep_dev, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    D3D_HIP_CHECK(hipStreamSynchronize(s));
    *n_keep_host = h;
  }
  return D3D_OK;
}

}  // extern "C"
