// a14-a18. Rotated BEV IoU x z-interval IoU, greedy rotated NMS, box decode -- all on device.
// The reference bounces GPU -> numpy -> numba-CUDA -> numpy -> spconv C++ -> GPU
// (utils3d/rotate_nms_3d_torch.py:64-83, second/core/non_max_suppression/nms_cpu.py:35-43).
//
// Arithmetic contract (shared with the oracle, compiled with -ffp-contract=off): fp32 for the
// geometry of second/core/non_max_suppression/nms_gpu.py, fp64 for the triangle-fan sum and the
// final ratio (numba promotes `x / 2.0`), cos/sin evaluated in fp64 and rounded to fp32.
#include "d3d_internal.h"

namespace d3d {

struct Quad {
  float p[8];  // 4 corners (x,y), order of rbbox_to_corners (nms_gpu.py:355-378)
};

__device__ __forceinline__ Quad make_quad(float xc, float yc, float xd, float yd, float angle) {
  const float a_cos = (float)cos((double)angle);
  const float a_sin = (float)sin((double)angle);
  const float cx[4] = {-xd / 2, -xd / 2, xd / 2, xd / 2};
  const float cy[4] = {-yd / 2, yd / 2, yd / 2, -yd / 2};
  Quad q;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    q.p[2 * i] = a_cos * cx[i] + a_sin * cy[i] + xc;
    q.p[2 * i + 1] = -a_sin * cx[i] + a_cos * cy[i] + yc;
  }
  return q;
}

// nms_gpu.py:310-328
__device__ __forceinline__ bool point_in_quad(float px, float py, const float *c) {
  const float ab0 = c[2] - c[0], ab1 = c[3] - c[1];
  const float ad0 = c[6] - c[0], ad1 = c[7] - c[1];
  const float ap0 = px - c[0], ap1 = py - c[1];
  const float abab = ab0 * ab0 + ab1 * ab1;
  const float abap = ab0 * ap0 + ab1 * ap1;
  const float adad = ad0 * ad0 + ad1 * ad1;
  const float adap = ad0 * ap0 + ad1 * ap1;
  return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

// nms_gpu.py:222-265
__device__ __forceinline__ bool seg_intersect(const float *p1, const float *p2, int i, int j,
                                              float *out) {
  const float A0 = p1[2 * i], A1 = p1[2 * i + 1];
  const float B0 = p1[2 * ((i + 1) & 3)], B1 = p1[2 * ((i + 1) & 3) + 1];
  const float C0 = p2[2 * j], C1 = p2[2 * j + 1];
  const float D0 = p2[2 * ((j + 1) & 3)], D1 = p2[2 * ((j + 1) & 3) + 1];
  const float BA0 = B0 - A0, BA1 = B1 - A1;
  const float DA0 = D0 - A0, CA0 = C0 - A0;
  const float DA1 = D1 - A1, CA1 = C1 - A1;
  const bool acd = DA1 * CA0 > CA1 * DA0;
  const bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
  if (acd == bcd) return false;
  const bool abc = CA1 * BA0 > BA1 * CA0;
  const bool abd = DA1 * BA0 > BA1 * DA0;
  if (abc == abd) return false;
  const float DC0 = D0 - C0, DC1 = D1 - C1;
  const float ABBA = A0 * B1 - B0 * A1;
  const float CDDC = C0 * D1 - D0 * C1;
  const float DH = BA1 * DC0 - BA0 * DC1;
  const float Dx = ABBA * DC0 - BA0 * CDDC;
  const float Dy = ABBA * DC1 - BA1 * CDDC;
  out[0] = Dx / DH;
  out[1] = Dy / DH;
  return true;
}

// The point list (up to 24 points) and the sort keys of inter() are indexed at run time: as private arrays they live in
// scratch memory.  PrivatePts is that form; LdsPts keeps them in a column of LDS the caller provides (element i of
// thread t at base[i * stride + t]) -- same operations in the same order.
struct PrivatePts {
  float p[48], v[24];
  __device__ __forceinline__ float &pt(int i) { return p[i]; }
  __device__ __forceinline__ float &key(int i) { return v[i]; }
};
struct LdsPts {
  float *base;
  int stride;
  __device__ __forceinline__ float &pt(int i) { return base[i * stride]; }
  __device__ __forceinline__ float &key(int i) { return base[(48 + i) * stride]; }
};
static constexpr int kLdsPtsFloats = 72;

// inter() of nms_gpu.py:381-395 on precomputed corners: q1 = rbbox1, q2 = rbbox2.
template <class Pts>
__device__ double quad_inter_f32(const Quad &q1, const Quad &q2, Pts B) {
  int num = 0;
  // nms_gpu.py:331-352
  for (int i = 0; i < 4; i++) {
    if (point_in_quad(q1.p[2 * i], q1.p[2 * i + 1], q2.p)) {
      B.pt(num * 2) = q1.p[2 * i];
      B.pt(num * 2 + 1) = q1.p[2 * i + 1];
      num++;
    }
    if (point_in_quad(q2.p[2 * i], q2.p[2 * i + 1], q1.p)) {
      B.pt(num * 2) = q2.p[2 * i];
      B.pt(num * 2 + 1) = q2.p[2 * i + 1];
      num++;
    }
  }
  float tmp[2];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      if (seg_intersect(q1.p, q2.p, i, j, tmp)) {
        B.pt(num * 2) = tmp[0];
        B.pt(num * 2 + 1) = tmp[1];
        num++;
      }
  if (num < 3) return 0.0;
  // nms_gpu.py:182-219
  float cx = 0.f, cy = 0.f;
  for (int i = 0; i < num; i++) {
    cx += B.pt(2 * i);
    cy += B.pt(2 * i + 1);
  }
  cx = (float)((double)cx / num);
  cy = (float)((double)cy / num);
  for (int i = 0; i < num; i++) {
    float v0 = B.pt(2 * i) - cx, v1 = B.pt(2 * i + 1) - cy;
    const float d = sqrtf(v0 * v0 + v1 * v1);
    v0 = v0 / d;
    v1 = v1 / d;
    if (v1 < 0) v0 = -2 - v0;
    B.key(i) = v0;
  }
  for (int i = 1; i < num; i++) {
    if (B.key(i - 1) > B.key(i)) {
      const float temp = B.key(i), tx = B.pt(2 * i), ty = B.pt(2 * i + 1);
      int j = i;
      while (j > 0 && B.key(j - 1) > temp) {
        B.key(j) = B.key(j - 1);
        B.pt(j * 2) = B.pt(j * 2 - 2);
        B.pt(j * 2 + 1) = B.pt(j * 2 - 1);
        j--;
      }
      B.key(j) = temp;
      B.pt(j * 2) = tx;
      B.pt(j * 2 + 1) = ty;
    }
  }
  // nms_gpu.py:166-179
  double area = 0.0;
  for (int i = 0; i < num - 2; i++) {
    const float a0 = B.pt(0), a1 = B.pt(1), b0 = B.pt(2 * i + 2), b1 = B.pt(2 * i + 3), c0 = B.pt(2 * i + 4),
                c1 = B.pt(2 * i + 5);
    const float v = (a0 - c0) * (b1 - c1) - (a1 - c1) * (b0 - c0);
    area += fabs((double)v / 2.0);
  }
  return area;
}

// devRotateIoUEval (nms_gpu.py:552-570): rbox1 = (q1, dims d1a x d1b), rbox2 likewise.
// `disjoint`: the caller knows that the quads' bounding boxes are strictly apart -- then inter() finds no corner inside
// the other quad and no edge crossing, i.e. returns exactly 0.0, and only the criterion's formula is left to evaluate
template <class Pts = PrivatePts>
__device__ __forceinline__ float iou_eval(const Quad &q1, float d1a, float d1b, const Quad &q2,
                                          float d2a, float d2b, int criterion, Pts B = Pts(), bool disjoint = false) {
  const float area1 = d1a * d1b, area2 = d2a * d2b;
  const double ai = disjoint ? 0.0 : quad_inter_f32(q1, q2, B);
  if (criterion == -1) return (float)(ai / ((double)(area1 + area2) - ai));
  if (criterion == 0) return (float)(ai / area1);
  if (criterion == 1) return (float)(ai / area2);
  if (criterion == 2) {
    const bool small = fminf(d2a, d2b) / fmaxf(d2a, d2b) < 0.25;
    if (small) return (float)(ai / ((double)area2 + fmax(0.0, (double)area1 * 0.5 - ai)));
    return (float)(ai / ((double)(area1 + area2) - ai));
  }
  return (float)ai;
}

// fp64 Sutherland-Hodgman clip of quad P by quad Q (both fp32 corners); returns the area.
// The two vertex lists (a quad clipped by four half-planes has at most 8 vertices) are indexed at run time; as private
// arrays they end up in scratch memory (528 B per lane, every access a round trip), so the caller hands in a column of
// LDS instead: element i of list l of thread t is buf[(l * 16 + i) * stride + t].  Same operations in the same order.
__device__ __forceinline__ double quad_inter_f64(const float *P, const float *Q, double *buf, int stride) {
  double *a = buf, *b = buf + 16 * stride;
  int na = 4;
#pragma unroll
  for (int i = 0; i < 8; i++) a[i * stride] = P[i];
  double sq = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int j = (i + 1) & 3;
    sq += (double)Q[2 * i] * Q[2 * j + 1] - (double)Q[2 * j] * Q[2 * i + 1];
  }
  const double sgn = sq >= 0 ? 1.0 : -1.0;
#pragma unroll
  for (int e = 0; e < 4; e++) {      // (unrolled: Q is then indexed statically and stays in registers)
    if (na <= 0) break;
    const double x1 = Q[2 * e], y1 = Q[2 * e + 1];
    const double x2 = Q[2 * ((e + 1) & 3)], y2 = Q[2 * ((e + 1) & 3) + 1];
    int nb = 0;
    for (int i = 0; i < na; i++) {
      const int j = (i + 1) % na;
      const double cx = a[(2 * i) * stride], cy = a[(2 * i + 1) * stride], nx = a[(2 * j) * stride],
                   ny = a[(2 * j + 1) * stride];
      const double dc = sgn * ((x2 - x1) * (cy - y1) - (y2 - y1) * (cx - x1));
      const double dn = sgn * ((x2 - x1) * (ny - y1) - (y2 - y1) * (nx - x1));
      if (dc >= 0) {
        b[(2 * nb) * stride] = cx;
        b[(2 * nb + 1) * stride] = cy;
        nb++;
      }
      if ((dc >= 0) != (dn >= 0)) {
        const double t = dc / (dc - dn);
        b[(2 * nb) * stride] = cx + t * (nx - cx);
        b[(2 * nb + 1) * stride] = cy + t * (ny - cy);
        nb++;
      }
    }
    na = nb;
    double *tmp = a;   // the clipped list is the next edge's input
    a = b;
    b = tmp;
  }
  if (na < 3) return 0.0;
  double s = 0;
  for (int i = 0; i < na; i++) {
    const int j = (i + 1) % na;
    s += a[(2 * i) * stride] * a[(2 * j + 1) * stride] - a[(2 * j) * stride] * a[(2 * i + 1) * stride];
  }
  return fabs(s) * 0.5;
}
__device__ __forceinline__ double quad_area_f64(const float *P) {
  double s = 0;
  for (int i = 0; i < 4; i++) {
    const int j = (i + 1) & 3;
    s += (double)P[2 * i] * P[2 * j + 1] - (double)P[2 * j] * P[2 * i + 1];
  }
  return fabs(s) * 0.5;
}

// ------------------------------------------------------------------------------------------
// IoU matrix: 64 x 64 tile per 256-thread block; lane = column (coalesced stores).
// mode 0: rotate_iou_gpu_eval on [*,5] boxes.  mode 1: boxes_iou_3d on [*,7] yx_zb boxes.
struct IouArgs {
  const float *rows;  // "boxes"/"targets"  [N, stride]
  const float *cols;  // "query"/"anchors"  [K, stride]
  int N, K, mode, criterion, only_xy;
  float aug[4];
  float *out;
};
struct BoxRec {
  Quad q;
  float d0, d1, z0, z1;
  float raw[5];
  float lo[2], hi[2];   // bounding box of the corners
};
__device__ __forceinline__ void load_box(const IouArgs &a, const float *base, int idx, bool is_row,
                                         BoxRec &r) {
  float xc, yc, d0, d1, ang;
  if (a.mode == 0) {
    const float *b = base + (size_t)idx * 5;
    xc = b[0]; yc = b[1]; d0 = b[2]; d1 = b[3]; ang = b[4];
    r.z0 = 0.f; r.z1 = 1.f;
  } else {
    const float *b = base + (size_t)idx * 7;
    const float cy = is_row ? a.aug[0] : a.aug[2], cz = is_row ? a.aug[1] : a.aug[3];
    xc = b[0]; yc = b[1]; d0 = fmaxf(b[3], cy); d1 = b[4]; ang = b[6];
    const float dz = fmaxf(b[5], cz);
    r.z0 = b[2];
    r.z1 = b[2] + dz;  // rotate_nms_3d_torch.py:15-16
  }
  r.d0 = d0; r.d1 = d1;
  r.raw[0] = xc; r.raw[1] = yc; r.raw[2] = d0; r.raw[3] = d1; r.raw[4] = ang;
  r.q = make_quad(xc, yc, d0, d1, ang);
#pragma unroll
  for (int d = 0; d < 2; d++) {
    r.lo[d] = fminf(fminf(r.q.p[d], r.q.p[2 + d]), fminf(r.q.p[4 + d], r.q.p[6 + d]));
    r.hi[d] = fmaxf(fmaxf(r.q.p[d], r.q.p[2 + d]), fmaxf(r.q.p[4 + d], r.q.p[6 + d]));
  }
}

__global__ __launch_bounds__(256) void k_iou_matrix(IouArgs a) {
  __shared__ BoxRec srow[64], scol[64];
  const int tid = threadIdx.x;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  if (tid < 64) {
    if (r0 + tid < a.N) load_box(a, a.rows, r0 + tid, true, srow[tid]);
  } else if (tid < 128) {
    const int t = tid - 64;
    if (c0 + t < a.K) load_box(a, a.cols, c0 + t, false, scol[t]);
  }
  __syncthreads();
  const int c = tid & 63, rg = tid >> 6;
  if (c0 + c >= a.K) return;
  const BoxRec &cb = scol[c];
  for (int i = 0; i < 16; i++) {
    const int rr = rg * 16 + i;
    if (r0 + rr >= a.N) break;
    const BoxRec &rb = srow[rr];
    // kernel order (nms_gpu.py:605-611): rbox1 = query (col), rbox2 = box (row).  Most pairs of an anchors x targets
    // matrix are far apart: their bounding boxes do not touch and the intersection is exactly 0 (NaN corners compare
    // false and take the full path)
    const bool apart = cb.hi[0] < rb.lo[0] || rb.hi[0] < cb.lo[0] || cb.hi[1] < rb.lo[1] || rb.hi[1] < cb.lo[1];
    float v = iou_eval(cb.q, cb.d0, cb.d1, rb.q, rb.d0, rb.d1, a.criterion, PrivatePts(), apart);
    bool same = true;  // check_same_boxes, nms_gpu.py:653-664
#pragma unroll
    for (int d = 0; d < 5; d++) same = same && (fabsf(rb.raw[d] - cb.raw[d]) < (float)1e-6);
    if (same) v = 1.f;
    if (a.mode == 1 && !a.only_xy) {
      const float overlap = fminf(cb.z1, rb.z1) - fmaxf(cb.z0, rb.z0);
      const float common = fmaxf(cb.z1, rb.z1) - fminf(cb.z0, rb.z0);
      v = v * (overlap / common);
    }
    a.out[(size_t)(r0 + rr) * a.K + c0 + c] = v;
  }
}

// ------------------------------------------------------------------------------------------
// NMS suppression masks: one wave per 64 x 64 tile; lanes = candidate boxes j; __ballot packs the
// 64 decisions "i suppresses j" of row i into one 64-bit word (wave64).
struct NmsBox {
  Quad q;
  float d0, d1, z0, z1, raw[5];
  float radius;  // BEV circumscribed circle, for the exact far-apart early-out
  double area;
};

// Conservative separating-axis test of two quads (rectangles): true only if they are separated by
// more than `margin` along one of the 4 edge directions -> they cannot touch, intersection area 0.
__device__ __forceinline__ bool quads_separated(const float *a, const float *b, float margin) {
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const float *p = q ? b : a;
#pragma unroll
    for (int e = 0; e < 2; e++) {
      float nx = p[2 * e + 3] - p[2 * e + 1], ny = -(p[2 * e + 2] - p[2 * e]);  // normal of edge e
      const float len = sqrtf(nx * nx + ny * ny);
      if (!(len > 0.f)) continue;
      nx /= len;
      ny /= len;
      float amin = 1e30f, amax = -1e30f, bmin = 1e30f, bmax = -1e30f;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float pa = a[2 * i] * nx + a[2 * i + 1] * ny, pb = b[2 * i] * nx + b[2 * i + 1] * ny;
        amin = fminf(amin, pa); amax = fmaxf(amax, pa);
        bmin = fminf(bmin, pb); bmax = fmaxf(bmax, pb);
      }
      if (amin > bmax + margin || bmin > amax + margin) return true;
    }
  }
  return false;
}
// A batch of independent candidate lists ("segments", e.g. the classes of the box head): candidate i of
// segment b is box order[b*stride + i] (b*stride + i without `order`), i < counts[b] (n_max without `counts`),
// already in descending score order.
struct NmsSegs {
  const int32_t *order;
  const int32_t *counts;
  int stride, n_max;
};
__device__ __forceinline__ int seg_count(const NmsSegs &g, int b) { return g.counts ? min(g.counts[b], g.n_max) : g.n_max; }
__device__ __forceinline__ int seg_box(const NmsSegs &g, int b, int i) {
  return g.order ? g.order[(size_t)b * g.stride + i] : b * g.stride + i;
}
// min_yx / min_z: the NMS_AUG_THICKNESS clamp of boxlist_nms_3d (dy, dx >= min_yx, dz >= min_z, IoU only)
__global__ void k_nms_prep(const float *__restrict__ boxes, NmsSegs g, float min_yx, float min_z,
                           NmsBox *__restrict__ rec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, sb = blockIdx.y;
  if (i >= seg_count(g, sb)) return;
  const float *b = boxes + (size_t)seg_box(g, sb, i) * 7;
  const float d0 = fmaxf(b[3], min_yx), d1 = fmaxf(b[4], min_yx), dz = fmaxf(b[5], min_z);
  NmsBox r;
  r.d0 = d0; r.d1 = d1;
  r.z0 = b[2]; r.z1 = b[2] + dz;
  r.raw[0] = b[0]; r.raw[1] = b[1]; r.raw[2] = d0; r.raw[3] = d1; r.raw[4] = b[6];
  r.q = make_quad(b[0], b[1], d0, d1, b[6]);
  r.radius = 0.5f * sqrtf(d0 * d0 + d1 * d1);
  r.area = quad_area_f64(r.q.p);
  rec[(size_t)sb * g.n_max + i] = r;
}
// Suppression masks in two passes so that the expensive geometry runs with full lanes:
//  k_nms_pairs  -- 256 threads per 64 x 64 tile (wave w: rows 16w..16w+15, lanes = candidate boxes j):
//                  cheap exact early-outs, survivors appended to a pair list (wave64 ballot + one atomic
//                  per wave).  The early-outs cannot change the decision:
//                   * z intervals do not overlap -> iou_z <= 0 (or NaN) -> gate `iou3d > 0` is false;
//                   * BEV circumscribed circles disjoint, or rectangles separated along an edge normal (margin
//                     ~1e-3 of the box size, far above fp32 rounding of metre-sized boxes) -> no corner inside
//                     the other box, no edges cross -> area 0 -> gate false.
//  k_nms_eval   -- one thread per listed pair: gate (fp32 IoU of nms_gpu.py x z IoU) and fp64 polygon IoU
//                  >= thresh; sets bit j of word [i][j/64] (atomicOr, order independent).
static constexpr int kNmsIdxBits = 12;  // candidates per segment <= 4096
__global__ __launch_bounds__(256) void k_nms_pairs(const NmsBox *__restrict__ rec, NmsSegs g, int2 *__restrict__ pairs,
                                                   unsigned int *__restrict__ n_pairs) {
  const int rt = blockIdx.y, ct = blockIdx.x, sb = blockIdx.z;
  const int n = seg_count(g, sb);
  if (ct < rt || ct * 64 >= n) return;
  rec += (size_t)sb * g.n_max;
  __shared__ NmsBox srow[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 64 && rt * 64 + threadIdx.x < n) srow[threadIdx.x] = rec[rt * 64 + threadIdx.x];
  const int j = ct * 64 + lane;
  NmsBox cb;
  if (j < n) cb = rec[j];
  __syncthreads();
  const int nrow = min(64, n - rt * 64);
  // the wave's 16 rows first (ballots stay in registers), then ONE counter atomic per wave: the pair counter is
  // a single address, and one atomic per (wave, row) serialises the whole grid on it
  unsigned long long bal[16];
  unsigned int total = 0;
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const int ii = wave * 16 + u;
    const int i = rt * 64 + ii;
    bool cand = false;
    if (ii < nrow && j < n && j > i) {
      const NmsBox &rb = srow[ii];
      const float overlap = fminf(cb.z1, rb.z1) - fmaxf(cb.z0, rb.z0);
      const float dx = cb.raw[0] - rb.raw[0], dy = cb.raw[1] - rb.raw[1];
      const float rr = (cb.radius + rb.radius) * 1.001f + 1e-4f;
      cand = overlap > 0.f && dx * dx + dy * dy <= rr * rr &&
             !quads_separated(rb.q.p, cb.q.p, 1e-3f * (1.f + cb.radius + rb.radius));
    }
    bal[u] = __ballot(cand);
    total += (unsigned int)__popcll(bal[u]);
  }
  if (total == 0) return;
  unsigned int base = 0;
  if (lane == 0) base = atomicAdd(n_pairs, total);
  base = __shfl(base, 0, 64);
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int u = 0; u < 16; u++) {
    if ((bal[u] >> lane) & 1ull)
      pairs[base + __popcll(bal[u] & lt)] = make_int2((sb << kNmsIdxBits) | (rt * 64 + wave * 16 + u), j);
    base += (unsigned int)__popcll(bal[u]);
  }
}
static constexpr int kEvalThreads = 128;   // x 32 doubles of LDS per thread (the clip's two vertex lists) = 32 KB
__global__ __launch_bounds__(kEvalThreads) void k_nms_eval(const NmsBox *__restrict__ rec, const int2 *__restrict__ pairs,
                                                           const unsigned int *__restrict__ n_pairs, int n_max, int ncb,
                                                           float thresh, unsigned long long *__restrict__ mask) {
  // one column per lane, in a region per WAVE: the fp32 gate's point list (72 floats), then the fp64 clip's two vertex
  // lists (32 doubles) reuse it.  The two layouts overlap in memory, which is safe only inside a wave (its lanes run
  // the gate of a pair together, then some of them the clip); another wave may be in the other phase.
  constexpr int kColBytes = kLdsPtsFloats * 4 > 256 ? kLdsPtsFloats * 4 : 256;
  __shared__ __attribute__((aligned(16))) char sm[kEvalThreads * kColBytes];
  char *region = sm + (threadIdx.x >> 6) * (64 * kColBytes);
  const int lane = threadIdx.x & 63;
  double *clip = reinterpret_cast<double *>(region) + lane;
  const LdsPts pts = {reinterpret_cast<float *>(region) + lane, 64};
  const unsigned int total = *n_pairs;
  for (unsigned int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
    int2 ij = pairs[p];
    const size_t seg0 = (size_t)(ij.x >> kNmsIdxBits) * n_max;  // first record / mask row of the segment
    ij.x &= (1 << kNmsIdxBits) - 1;
    const NmsBox rb = rec[seg0 + ij.x], cb = rec[seg0 + ij.y];
    // gate = boxes_iou_3d(dets, dets)[i, j] > 0 (nms_cpu.py:35, spconv nms.h)
    float v = iou_eval(cb.q, cb.d0, cb.d1, rb.q, rb.d0, rb.d1, -1, pts);
    bool same = true;
#pragma unroll
    for (int d = 0; d < 5; d++) same = same && (fabsf(rb.raw[d] - cb.raw[d]) < (float)1e-6);
    if (same) v = 1.f;
    const float overlap = fminf(cb.z1, rb.z1) - fmaxf(cb.z0, rb.z0);
    const float common = fmaxf(cb.z1, rb.z1) - fminf(cb.z0, rb.z0);
    v = v * (overlap / common);
    if (!(v > 0.0f)) continue;
    const double ia = quad_inter_f64(rb.q.p, cb.q.p, clip, 64);
    if (!(ia > 0)) continue;
    const double ua = rb.area + cb.area - ia;
    if (ua > 0 && ia / ua >= (double)thresh) {
      atomicOr(&mask[(seg0 + ij.x) * ncb + (ij.y >> 6)], 1ull << (ij.y & 63));
      // inside a 64-box chunk also the mirrored bit: a row's diagonal word then lists the EARLIER boxes of the chunk that
      // suppress it too, which lets the sweep resolve a chunk by fixed-point iteration over all lanes (k_nms_sweep_lds)
      if ((ij.x >> 6) == (ij.y >> 6)) atomicOr(&mask[(seg0 + ij.y) * ncb + (ij.x >> 6)], 1ull << (ij.x & 63));
    }
  }
}
// Greedy sweep by ONE wave: lane w owns word w of the "removed" bit vector.  Per 64-box chunk the
// intra-chunk chain is resolved on the diagonal words with v_readlane; the chunk's 64 mask rows are
// loaded unconditionally in one batch (independent loads) and OR-ed in for the kept boxes; the next
// chunk's diagonal word is fetched one chunk ahead.
__global__ __launch_bounds__(64) void k_nms_sweep(const unsigned long long *__restrict__ mask, NmsSegs g, int ncb,
                                                  int max_keep, int32_t *__restrict__ keep,
                                                  int32_t *__restrict__ n_keep) {
  const int lane = threadIdx.x, sb = blockIdx.x;
  const int n = __builtin_amdgcn_readfirstlane(seg_count(g, sb));  // wave-uniform: keep it (and all row pointers) in SGPRs
  mask += (size_t)sb * g.n_max * ncb;  // rows of ncb words; this segment uses the first ceil(n/64)
  keep += (size_t)sb * g.n_max;
  n_keep += sb;
  const int ncw = (n + 63) / 64;
  const unsigned long long lt = (1ull << lane) - 1ull;
  unsigned long long removed = 0;  // word `lane`
  int cnt = 0;
  // rows of chunk c, word `lane`.  Unconditional loads from a wave-uniform row pointer + the lane's word offset
  // (lanes past the row end re-read its last word, rows past the chunk end re-read its last row): neither is ever
  // used -- words left of the diagonal / beyond ncw are not read again, kept has no bit for a missing row -- and
  // the loads stay free of per-lane predication and 64-bit vector address arithmetic (the sweep is one wave:
  // every instruction is on the critical path)
  const unsigned int lane_off = (unsigned int)min(lane, max(ncb - 1, 0)) * 8u;  // byte offset of this lane's word
  auto load_rows = [&](int c, unsigned long long *w) {
    if (c >= ncw) return;
    const int base = c * 64;
    const int nrow = min(64, n - base);
    const unsigned long long *rowp = mask + (size_t)base * ncb;
#pragma unroll
    for (int b = 0; b < 64; b++) {
      const char *rp = (const char *)(rowp + (size_t)min(b, nrow - 1) * ncb);  // wave-uniform (SGPR pair)
      w[b] = *(const unsigned long long *)(rp + lane_off);
    }
  };
  auto load_diag = [&](int c) -> unsigned long long {
    const int base = c * 64;
    return (c < ncw && lane < min(64, n - base)) ? mask[(size_t)(base + lane) * ncb + c] : 0ull;
  };
  // one chunk: `w` holds its rows, `diag` its diagonal word; the rows of the chunk AFTER NEXT and the next diagonal
  // are requested before the dependent chain, so that a batch of 64 row loads has two chains to arrive under (one
  // chain is shorter than the batch's latency; the wave is alone on its SIMD and may use all 512 VGPRs)
  auto step = [&](int c, const unsigned long long *w, unsigned long long *w_fill, unsigned long long &diag) {
    const int base = c * 64;
    const int nrow = min(64, n - base);
    load_rows(c + 2, w_fill);
    const unsigned long long diag_next = load_diag(c + 1);
    // (requested before the chain: read through `order`, it is a global load the survivors' store would wait for)
    const int my_box = lane < nrow ? seg_box(g, sb, base + lane) : 0;
    // word c of the removed set, read into SGPRs: alive / kept / cnt and with them the whole chain stay scalar
    unsigned long long alive =
        ~(((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(removed >> 32), c) << 32) |
          (unsigned int)__builtin_amdgcn_readlane((int)removed, c));
    if (nrow < 64) alive &= (1ull << nrow) - 1ull;
    unsigned long long kept = 0;
    const unsigned int dlo = (unsigned int)diag, dhi = (unsigned int)(diag >> 32);
    // The dependent chain over the chunk's 64 boxes, a quarter of a chunk at a time: first the 16 diagonal words are moved
    // into scalar registers (independent v_readlane pairs, they pipeline), then the chain itself is pure scalar ALU --
    // test bit b of `alive`, clear the boxes that box b suppresses -- instead of two v_readlane round trips inside
    // every dependent step.  Rows past the segment's end have a zero word and no `alive` bit.
#pragma unroll
    for (int part = 0; part < 4; part++) {
      unsigned long long d[16];
#pragma unroll
      for (int b = 0; b < 16; b++)
        d[b] = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dhi, part * 16 + b) << 32) |
               (unsigned int)__builtin_amdgcn_readlane((int)dlo, part * 16 + b);
#pragma unroll
      for (int b = 0; b < 16; b++) {
        const unsigned long long bit = 1ull << (part * 16 + b);
        const bool on = (alive & bit) != 0;
        kept |= on ? bit : 0ull;
        alive &= on ? ~d[b] : ~0ull;
      }
    }
#pragma unroll
    for (int b = 0; b < 64; b++)
      if ((kept >> b) & 1ull) removed |= w[b];
    // survivors of the chunk, all lanes at once
    if ((kept >> lane) & 1ull) {
      const int pos = cnt + __popcll(kept & lt);
      if (pos < g.n_max) keep[pos] = my_box;
    }
    cnt += __popcll(kept);
    diag = diag_next;
  };
  unsigned long long wa[64], wb[64], wc[64];
  unsigned long long diag = load_diag(0);
  load_rows(0, wa);
  load_rows(1, wb);
  for (int c = 0; c < ncw && cnt < max_keep; c += 3) {   // the caller keeps at most max_keep survivors
    step(c, wa, wc, diag);
    if (c + 1 < ncw && cnt < max_keep) step(c + 1, wb, wa, diag);
    if (c + 2 < ncw && cnt < max_keep) step(c + 2, wc, wb, diag);
  }
  if (lane == 0) *n_keep = min(cnt, max_keep);  // the last chunk may overshoot the cap
}

// The same sweep with the mask rows staged through LDS by loader waves.  k_nms_sweep keeps three chunks of 64 rows in
// registers (384 VGPRs; the compiler moves them through the accumulation registers and scratch) and spends ~3.4 us per
// 64-box chunk, almost all of it waiting: the dependent chain itself is ~0.15 us.  Here the workgroup is one sweeper wave
// and kSwLoaders loader waves: a loader thread holds its 16-byte pieces of the next kSwStages chunks in registers (loads
// issued that many iterations ahead, so their latency is covered) and drops the oldest into one of two LDS slots while the
// sweeper works on the other; one workgroup barrier per chunk hands the slots over.  The sweeper reads its 64 diagonal
// words and, for the update of the removed set, the rows of the chunk from LDS (static offsets, conflict-free row
// stride of ncb + 1 words).  Same decisions in the same order: identical survivor lists.
static constexpr int kSwLoaders = 4, kSwStages = 4;
static constexpr int kSwThreads = 64 * (1 + kSwLoaders);
template <int NCBMAX>   // words per mask row <= NCBMAX (16: n <= 1024, 32: n <= 2048, 64: n <= 4096)
__global__ __launch_bounds__(kSwThreads) void k_nms_sweep_lds(const unsigned long long *__restrict__ mask, NmsSegs g,
                                                              int ncb, int max_keep, int32_t *__restrict__ keep,
                                                              int32_t *__restrict__ n_keep) {
  constexpr int NP = NCBMAX * 32 / (64 * kSwLoaders);      // 16-byte pieces of a chunk per loader thread
  static_assert(NP >= 1, "a chunk must give every loader thread a piece");
  extern __shared__ unsigned long long slots[];             // 2 x 64 rows x (ncb + 1) words
  __shared__ int stop_s;
  const int sb = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = seg_count(g, sb);
  mask += (size_t)sb * g.n_max * ncb;
  keep += (size_t)sb * g.n_max;
  n_keep += sb;
  const int ncw = (n + 63) / 64;
  const int rs = ncb + 1;                                    // LDS row stride in words
  const int slot_words = 64 * rs;
  if (threadIdx.x == 0) stop_s = 0;
  if (ncw == 0) {
    if (threadIdx.x == 0) *n_keep = 0;
    return;
  }
  const int half = ncb / 2 + (ncb & 1);                      // 16-byte pieces per row (the last may be half used)
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  // ---- loader side: piece p of thread tl covers words [2 q % half .., +2) of row q / half, q = tl + 64 kSwLoaders p
  const int tl = (int)threadIdx.x - 64;
  u64x2 st[kSwStages][NP];
  auto issue = [&](int c, u64x2 *dst) {                      // global loads of chunk c -> registers
    if (c >= ncw) return;
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const int q = tl + 64 * kSwLoaders * p;
      const int row = q / half, w2 = (q - row * half) * 2;
      u64x2 v = {0ull, 0ull};
      if (row < 64) {
        const unsigned long long *src = mask + (size_t)min(c * 64 + row, n - 1) * ncb + w2;
        v[0] = src[0];
        if (w2 + 1 < ncb) v[1] = src[1];
      }
      dst[p] = v;
    }
  };
  auto drop = [&](int c, const u64x2 *src) {                 // registers -> LDS slot c & 1
    if (c >= ncw) return;
    unsigned long long *slot = slots + (size_t)(c & 1) * slot_words;
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const int q = tl + 64 * kSwLoaders * p;
      const int row = q / half, w2 = (q - row * half) * 2;
      if (row < 64) {
        slot[row * rs + w2] = src[p][0];
        if (w2 + 1 < ncb) slot[row * rs + w2 + 1] = src[p][1];
      }
    }
  };
  // ---- sweeper state
  const unsigned long long lt = (1ull << lane) - 1ull;
  unsigned long long removed = 0;                            // word `lane` of the removed set
  int cnt = 0;
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < kSwStages; j++) issue(j, st[j]);
    drop(0, st[0]);
    issue(kSwStages, st[0]);
  }
  __syncthreads();
  for (int c0 = 0; c0 < ncw; c0 += kSwStages) {
    bool stop = false;
#pragma unroll
    for (int u = 0; u < kSwStages; u++) {
      const int c = c0 + u;
      if (c >= ncw) break;                                   // uniform
      if (wave > 0) {
        // chunk c + 1 (loaded kSwStages - 1 iterations ago) goes to the other slot; its stage is refilled
        drop(c + 1, st[(u + 1) % kSwStages]);
        issue(c + 1 + kSwStages, st[(u + 1) % kSwStages]);
      } else {
        const unsigned long long *slot = slots + (size_t)(c & 1) * slot_words;
        const int base = c * 64;
        const int nrow = min(64, n - base);
        const int my_box = lane < nrow ? seg_box(g, sb, base + lane) : 0;
        const unsigned long long diag = lane < nrow ? slot[lane * rs + c] : 0ull;
        unsigned long long alive =
            ~(((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(removed >> 32), c) << 32) |
              (unsigned int)__builtin_amdgcn_readlane((int)removed, c));
        if (nrow < 64) alive &= (1ull << nrow) - 1ull;
        // The chunk's greedy decision by fixed-point iteration over all lanes: box b is kept iff it is alive and no KEPT
        // earlier box of the chunk suppresses it.  K <- { b alive : (sup[b] & K) == 0 } starting from K = alive settles
        // box 0 after one step, box b once every earlier box is settled -- at most 64 steps, in practice the depth of
        // the longest suppression chain (a few) -- and its only fixed point is the greedy set.  One ballot per step
        // instead of a 64-step scalar chain fed by 128 lane reads.
        const unsigned long long sup = diag & lt;                 // earlier boxes of the chunk that suppress box `lane`
        const bool my_alive = (alive >> lane) & 1ull;
        unsigned long long kept = alive;
        for (int it = 0; it < 65; it++) {
          const unsigned long long knew = __ballot(my_alive && (sup & kept) == 0ull);
          if (knew == kept) break;
          kept = knew;
        }
        // removed |= rows of the kept boxes (word `lane`): 16 LDS reads in flight at a time
        const unsigned long long *col = slot + min(lane, ncb - 1);
#pragma unroll
        for (int part = 0; part < 4; part++) {
          unsigned long long w[16];
#pragma unroll
          for (int b = 0; b < 16; b++) w[b] = col[(part * 16 + b) * rs];
#pragma unroll
          for (int b = 0; b < 16; b++)
            if ((kept >> (part * 16 + b)) & 1ull) removed |= w[b];
        }
        if ((kept >> lane) & 1ull) {
          const int pos = cnt + __popcll(kept & lt);
          if (pos < g.n_max) keep[pos] = my_box;
        }
        cnt += __popcll(kept);
        if (cnt >= max_keep && lane == 0) stop_s = 1;
      }
      __syncthreads();
      if (stop_s) {
        stop = true;
        break;
      }
    }
    if (stop) break;
  }
  if (threadIdx.x == 0) *n_keep = min(cnt, max_keep);
}

// a14. BoxCoder3D.decode
__global__ void k_box_decode(const float *__restrict__ enc, const float *__restrict__ anchors, int n,
                             float w0, float w1, float w2, float w3, float w4, float w5, float w6,
                             float clip, float *__restrict__ out, const int64_t *__restrict__ rows, int nc) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t src = rows ? (size_t)rows[i] : (size_t)i;   // rows: the top-k selection, gathered here
  // nc > 1: enc holds nc class-wise encodings per anchor row (box_coder_3d.py decode of [n, 7 nc])
  const float *e = enc + src * 7, *a = anchors + (src / (size_t)nc) * 7;
  const float w[7] = {w0, w1, w2, w3, w4, w5, w6};
  box_decode_one(e, a, w, clip, out + (size_t)i * 7);
}

// survivors of the RPN's NMS into a list padded to P rows: out row i < *n_keep is box / score keep[i] with its sizes
// clamped from below (BoxList3D.clamp_size), the other rows repeat row 0 of the candidates (never pooled)
__global__ void k_gather_kept(const float *__restrict__ boxes, const float *__restrict__ scores,
                              const int32_t *__restrict__ keep, const int32_t *__restrict__ n_keep, int P,
                              float min_size, float *__restrict__ out_boxes, float *__restrict__ out_scores,
                              int32_t *__restrict__ count_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // the count, for the host: a store to pinned host memory that an event recorded behind this launch makes visible --
  // no copy engine and no second stream between the NMS and the host
  if (i == 0 && count_out) __hip_atomic_store(count_out, *n_keep, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (i >= P) return;
  const int j = i < *n_keep ? keep[i] : 0;
  const float *b = boxes + (size_t)j * 7;
  float *o = out_boxes + (size_t)i * 7;
#pragma unroll
  for (int k = 0; k < 7; k++) {
    float v = b[k];
    if (k >= 3 && k < 6) v = v < min_size ? min_size : v;   // torch.clamp(min=): a NaN stays
    o[k] = v;
  }
  out_scores[i] = scores[j];
}

}  // namespace d3d

using namespace d3d;

extern "C" {

int d3d_rotate_iou_eval(const float *boxes, int N, const float *query, int K, int criterion,
                        float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(N >= 0 && K >= 0, "rotate_iou_eval: negative size");
  if (N == 0 || K == 0) return D3D_OK;
  D3D_REQUIRE(boxes && query && out, "rotate_iou_eval: null pointer");
  IouArgs a;
  a.rows = boxes; a.cols = query; a.N = N; a.K = K; a.mode = 0; a.criterion = criterion; a.only_xy = 1;
  a.aug[0] = a.aug[1] = a.aug[2] = a.aug[3] = 0.f;
  a.out = out;
  hipLaunchKernelGGL(k_iou_matrix, dim3((K + 63) / 64, (N + 63) / 64), dim3(256), 0, s, a);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_boxes_iou_3d(const float *targets, int M, const float *anchors, int N, const float *aug_host,
                     int criterion, int only_xy, float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(M >= 0 && N >= 0, "boxes_iou_3d: negative size");
  if (M == 0 || N == 0) return D3D_OK;
  D3D_REQUIRE(targets && anchors && out, "boxes_iou_3d: null pointer");
  IouArgs a;
  a.rows = targets; a.cols = anchors; a.N = M; a.K = N; a.mode = 1; a.criterion = criterion; a.only_xy = only_xy;
  for (int i = 0; i < 4; i++) a.aug[i] = aug_host ? aug_host[i] : 0.f;
  a.out = out;
  hipLaunchKernelGGL(k_iou_matrix, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, s, a);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

size_t d3d_nms_batched_scratch_bytes(int segments, int n_max) {
  const size_t B = segments > 0 ? segments : 0, n = n_max > 0 ? n_max : 0, ncb = (n + 63) / 64;
  return B * n * ncb * 8 + B * n * sizeof(NmsBox) + (B * n * n / 2 + 64) * sizeof(int2) + 2048;
}
size_t d3d_nms_scratch_bytes(int n) { return d3d_nms_batched_scratch_bytes(1, n); }

int d3d_rotate_nms_3d_batched(const float *boxes, const int32_t *order, int stride, const int32_t *counts,
                              int segments, int n_max, float thresh, float min_yx, float min_z, int max_keep,
                              int32_t *keep, int32_t *n_keep, void *scratch, size_t scratch_bytes, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const int B = segments, n = n_max;
  D3D_REQUIRE(n >= 0 && n <= (1 << kNmsIdxBits), "rotate_nms_3d: n_max=%d out of range (<= 4096)", n);
  D3D_REQUIRE(B >= 0 && B <= 4096 && n_keep, "rotate_nms_3d: bad segment count %d or null n_keep", B);
  if (B == 0) return D3D_OK;
  if (n == 0) {
    D3D_HIP_CHECK(hipMemsetAsync(n_keep, 0, sizeof(int32_t) * B, s));
    return D3D_OK;
  }
  D3D_REQUIRE(boxes && keep && scratch && scratch_bytes >= d3d_nms_batched_scratch_bytes(B, n), "rotate_nms_3d: bad buffers");
  D3D_REQUIRE(B == 1 || order || stride >= n, "rotate_nms_3d: overlapping segments (stride %d < n_max %d)", stride, n);
  const int ncb = (n + 63) / 64;
  char *base = (char *)scratch;
  unsigned long long *mask = (unsigned long long *)base;
  size_t off = ((size_t)B * n * ncb * 8 + 255) & ~size_t(255);
  unsigned int *n_pairs = (unsigned int *)(base + off);
  off += 256;
  NmsBox *rec = (NmsBox *)(base + off);
  off = (off + (size_t)B * n * sizeof(NmsBox) + 255) & ~size_t(255);
  int2 *pairs = (int2 *)(base + off);
  const NmsSegs g = {order, counts, stride, n};
  D3D_HIP_CHECK(hipMemsetAsync(mask, 0, (((size_t)B * n * ncb * 8 + 255) & ~size_t(255)) + 256, s));  // masks + pair counter
  hipLaunchKernelGGL(k_nms_prep, dim3((n + 127) / 128, B), dim3(128), 0, s, boxes, g, min_yx, min_z, rec);
  hipLaunchKernelGGL(k_nms_pairs, dim3(ncb, ncb, B), dim3(256), 0, s, rec, g, pairs, n_pairs);
  hipLaunchKernelGGL(k_nms_eval, dim3(1024), dim3(kEvalThreads), 0, s, rec, pairs, n_pairs, n, ncb, thresh, mask);
  const int mk = max_keep > 0 ? max_keep : n;
  static const bool lds_sweep = [] {      // D3D_NMS_SWEEP=regs: the single-wave form (A/B runs)
    const char *e = getenv("D3D_NMS_SWEEP");
    return !(e && e[0] == 'r');
  }();
  if (lds_sweep) {
    const size_t lds = (size_t)2 * 64 * (ncb + 1) * sizeof(unsigned long long);
    if (ncb <= 16)
      hipLaunchKernelGGL(k_nms_sweep_lds<16>, dim3(B), dim3(kSwThreads), lds, s, mask, g, ncb, mk, keep, n_keep);
    else if (ncb <= 32)
      hipLaunchKernelGGL(k_nms_sweep_lds<32>, dim3(B), dim3(kSwThreads), lds, s, mask, g, ncb, mk, keep, n_keep);
    else
      hipLaunchKernelGGL(k_nms_sweep_lds<64>, dim3(B), dim3(kSwThreads), lds, s, mask, g, ncb, mk, keep, n_keep);
  } else {
    hipLaunchKernelGGL(k_nms_sweep, dim3(B), dim3(64), 0, s, mask, g, ncb, mk, keep, n_keep);
  }
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_rotate_nms_3d_sorted(const float *boxes, int n, float thresh, int max_keep, int32_t *keep,
                             int32_t *n_keep, void *scratch, size_t scratch_bytes, void *stream) {
  return d3d_rotate_nms_3d_batched(boxes, nullptr, 0, nullptr, 1, n, thresh, 0.f, 0.f, max_keep, keep, n_keep, scratch,
                                   scratch_bytes, stream);
}

// ---- box-head post-processing glue (inference.py:113-148), three launches instead of ~20 tensor ops ----
// masked class scores, class-major: sc[j][i] = prob[i][j+1] if > thresh else -1; counts[j] = candidates of class j+1
__global__ __launch_bounds__(256) void k_post_scores(const float *__restrict__ prob, int K, int nc, float thresh,
                                                     float *__restrict__ sc, int32_t *__restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  bool c = false;
  if (i < K) {
    const float v = prob[(size_t)i * nc + j + 1];
    c = v > thresh;
    sc[(size_t)j * K + i] = c ? v : -1.f;
  }
  const unsigned long long bal = __ballot(c);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(counts + j, __popcll(bal));
}
// order[j][i] = box index (RoI idx[j][i], class j+1) in the [K, nc] layout of the decoded boxes
__global__ __launch_bounds__(256) void k_post_order(const int64_t *__restrict__ idx, int K, int nc, int nseg,
                                                    int32_t *__restrict__ order) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)nseg * K) return;
  order[t] = (int32_t)(idx[t] * nc + (int64_t)(t / K) + 1);
}
// survivors of the batched NMS, class-major in selection order: flat box index + score, padding = (0, -1)
__global__ __launch_bounds__(256) void k_post_gather(const int32_t *__restrict__ keep, const int32_t *__restrict__ nk,
                                                     int nseg, int n_max, const float *__restrict__ prob_flat,
                                                     float *__restrict__ s_all, int64_t *__restrict__ flat_all) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nseg * n_max) return;
  const bool valid = (t % n_max) < nk[t / n_max];
  const int f = valid ? keep[t] : 0;
  flat_all[t] = f;
  s_all[t] = valid ? prob_flat[f] : -1.f;
}

// The cut to detections_per_img and the final gathers of inference.py:140-148 in ONE single-workgroup launch (what
// followed d3d_post_gather as ~12 tensor-library launches: top-k, compare, nonzero, three index ops, remainder).
// Candidates t = 0 .. segments * n_max - 1 (class-major, selection order): score s[t] = prob_flat[keep[t]] for
// t % n_max < n_keep[t / n_max], else -1.  thresh = the D-th largest s (D > 0 and D < #candidates; a padding -1 when
// fewer than D survive) clamped to >= 0; the selected set is { t : s[t] >= thresh } IN t ORDER -- like the reference's
// `keep = cls_scores >= image_thresh`, ties at the threshold all stay.  For every selected t: the box, the score and
// the label (box index % nc).  The D-th largest is found by a 4 x 8-bit radix select on the order-preserving integer
// image of the floats (exact), the compaction by a workgroup scan.
static constexpr int kSelMax = 8192, kSelThreads = 1024;
__global__ __launch_bounds__(kSelThreads) void k_post_select(const int32_t *__restrict__ keep,
                                                             const int32_t *__restrict__ nk, int nseg, int n_max,
                                                             const float *__restrict__ prob_flat,
                                                             const float *__restrict__ boxes, int nc, int D,
                                                             float *__restrict__ out_boxes, float *__restrict__ out_scores,
                                                             int64_t *__restrict__ out_labels, int32_t *__restrict__ out_n) {
  __shared__ uint32_t key[kSelMax];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t wsum[kSelThreads / 64];
  __shared__ uint32_t sel_prefix, sel_rank;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = nseg * n_max;
  for (int t = tid; t < N; t += kSelThreads) {
    const bool valid = (t % n_max) < nk[t / n_max];
    key[t] = f32_ordered(valid ? prob_flat[keep[t]] : -1.f);
  }
  uint32_t thresh_key = f32_ordered(0.f);     // s >= 0: every survivor
  if (D > 0 && D < N) {
    // radix select of the D-th largest key: fix 8 bits per pass, most significant first
    uint32_t prefix = 0, want = (uint32_t)D;   // `want`-th largest among the keys that match `prefix` on the fixed bits
    for (int pass = 0; pass < 4; pass++) {
      const int shift = 24 - 8 * pass;
      const uint32_t fixed_mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (int t = tid; t < N; t += kSelThreads) {
        const uint32_t k = key[t];
        if ((k & fixed_mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t acc = 0;
        int b = 255;
        for (; b > 0; b--) {
          if (acc + hist[b] >= want) break;
          acc += hist[b];
        }
        sel_prefix = prefix | ((uint32_t)b << shift);
        sel_rank = want - acc;
      }
      __syncthreads();
      prefix = sel_prefix;
      want = sel_rank;
      __syncthreads();
    }
    thresh_key = max(prefix, thresh_key);      // thresh.clamp_min(0)
  }
  __syncthreads();
  // ordered compaction: thread t owns the contiguous candidates [per * t, per * t + per)
  const int per = (N + kSelThreads - 1) / kSelThreads;
  const int t0 = min(N, tid * per), t1 = min(N, t0 + per);
  uint32_t mine = 0;
  for (int t = t0; t < t1; t++) mine += key[t] >= thresh_key;
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t pos = incl - mine;
  for (int w = 0; w < wave; w++) pos += wsum[w];
  if (tid == kSelThreads - 1)   // out_n may be a pinned host word: system-scope release, like k_gather_kept's count
    __hip_atomic_store(out_n, (int32_t)(pos + mine), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  for (int t = t0; t < t1; t++) {
    if (key[t] < thresh_key) continue;
    const int f = keep[t];
#pragma unroll
    for (int j = 0; j < 7; j++) out_boxes[(size_t)pos * 7 + j] = boxes[(size_t)f * 7 + j];
    out_scores[pos] = prob_flat[f];
    out_labels[pos] = (int64_t)(f % nc);
    pos++;
  }
}

int d3d_post_scores(const float *prob, int K, int nc, float thresh, float *sc, int32_t *counts, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(K >= 0 && nc >= 2, "post_scores: bad shape");
  D3D_REQUIRE(counts, "post_scores: null pointer");
  D3D_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int32_t) * (nc - 1), s));
  if (K == 0) return D3D_OK;
  D3D_REQUIRE(prob && sc, "post_scores: null pointer");
  hipLaunchKernelGGL(k_post_scores, dim3((K + 255) / 256, nc - 1), dim3(256), 0, s, prob, K, nc, thresh, sc, counts);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}
int d3d_post_order(const int64_t *idx, int K, int nc, int32_t *order, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(K >= 0 && nc >= 2, "post_order: bad shape");
  if (K == 0) return D3D_OK;
  D3D_REQUIRE(idx && order, "post_order: null pointer");
  const long total = (long)(nc - 1) * K;
  hipLaunchKernelGGL(k_post_order, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, idx, K, nc, nc - 1, order);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}
int d3d_post_gather(const int32_t *keep, const int32_t *n_keep, int segments, int n_max, const float *prob_flat,
                    float *scores, int64_t *flat, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(segments >= 0 && n_max >= 0, "post_gather: bad shape");
  if (segments == 0 || n_max == 0) return D3D_OK;
  D3D_REQUIRE(keep && n_keep && prob_flat && scores && flat, "post_gather: null pointer");
  hipLaunchKernelGGL(k_post_gather, dim3((segments * n_max + 255) / 256), dim3(256), 0, s, keep, n_keep, segments, n_max,
                     prob_flat, scores, flat);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_post_select_max(void) { return kSelMax; }
int d3d_post_select(const int32_t *keep, const int32_t *n_keep, int segments, int n_max, const float *prob_flat,
                    const float *boxes, int nc, int detections, float *out_boxes, float *out_scores, int64_t *out_labels,
                    int32_t *out_n, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(segments >= 0 && n_max >= 0 && nc >= 1 && out_n, "post_select: bad arguments");
  const long N = (long)segments * n_max;
  D3D_REQUIRE(N <= kSelMax, "post_select: %ld candidates exceed the %d of the single-workgroup selection", N, kSelMax);
  if (N == 0) {
    D3D_HIP_CHECK(hipMemsetAsync(out_n, 0, sizeof(int32_t), s));
    return D3D_OK;
  }
  D3D_REQUIRE(keep && n_keep && prob_flat && boxes && out_boxes && out_scores && out_labels, "post_select: null pointer");
  hipLaunchKernelGGL(k_post_select, dim3(1), dim3(kSelThreads), 0, s, keep, n_keep, segments, n_max, prob_flat, boxes, nc,
                     detections, out_boxes, out_scores, out_labels, out_n);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_box_decode(const float *enc, const float *anchors, int n, const float *weights_host,
                   float clip, float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return D3D_OK;
  D3D_REQUIRE(enc && anchors && out && weights_host && n > 0, "box_decode: bad arguments");
  const float *w = weights_host;
  hipLaunchKernelGGL(k_box_decode, dim3((n + 255) / 256), dim3(256), 0, s, enc, anchors, n, w[0], w[1], w[2], w[3], w[4], w[5], w[6], clip, out,
                     (const int64_t *)nullptr, 1);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_box_decode_rows(const float *enc, const float *anchors, const int64_t *rows, int n, const float *weights_host,
                        float clip, float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return D3D_OK;
  D3D_REQUIRE(enc && anchors && rows && out && weights_host && n > 0, "box_decode_rows: bad arguments");
  const float *w = weights_host;
  hipLaunchKernelGGL(k_box_decode, dim3((n + 255) / 256), dim3(256), 0, s, enc, anchors, n, w[0], w[1], w[2], w[3], w[4], w[5], w[6], clip, out,
                     rows, 1);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_box_decode_classes(const float *enc, const float *anchors, int n, int nc, const float *weights_host, float clip,
                           float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return D3D_OK;
  D3D_REQUIRE(enc && anchors && out && weights_host && n > 0 && nc >= 1, "box_decode_classes: bad arguments");
  const float *w = weights_host;
  const long total = (long)n * nc;
  D3D_REQUIRE(total < (1L << 31), "box_decode_classes: %ld boxes", total);
  hipLaunchKernelGGL(k_box_decode, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, enc, anchors, (int)total, w[0], w[1], w[2], w[3],
                     w[4], w[5], w[6], clip, out, (const int64_t *)nullptr, nc);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_gather_kept(const float *boxes, const float *scores, const int32_t *keep, const int32_t *n_keep_dev, int P,
                    float min_size, float *out_boxes, float *out_scores, int32_t *count_out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (P == 0) return D3D_OK;
  D3D_REQUIRE(boxes && scores && keep && n_keep_dev && out_boxes && out_scores && P > 0, "gather_kept: bad arguments");
  hipLaunchKernelGGL(k_gather_kept, dim3((P + 255) / 256), dim3(256), 0, s, boxes, scores, keep, n_keep_dev, P, min_size,
                     out_boxes, out_scores, count_out);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

}  // extern "C"
