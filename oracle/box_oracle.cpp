// TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's rotated-IoU / NMS / box-codec /
// RoIAlignRotated3D path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may load this library; the product path never does.
//
// PARITY PINNING
//  * rotated BEV IoU (second/core/non_max_suppression/nms_gpu.py): numba-CUDA, not runnable here
//    (numba absent, NVIDIA-only).  Restated operation by operation including numba's typing
//    rules (float32 arithmetic; `x / 2.0`, the triangle-area sum and the final ratio in
//    float64).  Pinned by the reference's recorded values in
//    second/core/non_max_suppression/test_nms_gpu.py:14-15 and analytic cases
//    (tests/test_oracle_boxes.py).  libdevice cosf/sinf are replaced by fp64 cos/sin rounded to
//    fp32 (<= 1 ulp from any faithful cosf); NVVM's possible FMA contraction is not modelled.
//  * greedy NMS: spconv.utils.rotate_non_max_suppression_cpu (traveller59/spconv v1.0/1.1,
//    shipped only as cp37 wheels under so_backups/) is un-vendored -> "parity unpinned".
//    Restated from its published algorithm: descending-score greedy, skip pair when the gate
//    matrix <= 0, suppress when BEV polygon IoU >= thresh.  boost::geometry's overlay is
//    replaced by Sutherland-Hodgman clipping in fp64.
//  * RoIAlignRotated3D: CUDA-only in the reference, restated from the .cu; no fixtures exist.
//
// File:line citations are into /root/reference/.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

// ---- second/core/non_max_suppression/nms_gpu.py:166-179 ---------------------------------
inline double trangle_area(const float *a, const float *b, const float *c) {
  float v = (a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0]);
  return (double)v / 2.0;
}
inline double poly_area(const float *int_pts, int num) {
  double area_val = 0.0;
  for (int i = 0; i < num - 2; i++)
    area_val += std::fabs(trangle_area(int_pts, int_pts + 2 * i + 2, int_pts + 2 * i + 4));
  return area_val;
}

// nms_gpu.py:182-219
void sort_vertex_in_convex_polygon(float *int_pts, int num) {
  if (num <= 0) return;
  float center[2] = {0.f, 0.f};
  for (int i = 0; i < num; i++) {
    center[0] += int_pts[2 * i];
    center[1] += int_pts[2 * i + 1];
  }
  center[0] = (float)((double)center[0] / num);
  center[1] = (float)((double)center[1] / num);
  float v[2];
  float vs[24];
  for (int i = 0; i < num; i++) {
    v[0] = int_pts[2 * i] - center[0];
    v[1] = int_pts[2 * i + 1] - center[1];
    float d = std::sqrt(v[0] * v[0] + v[1] * v[1]);
    v[0] = v[0] / d;
    v[1] = v[1] / d;
    if (v[1] < 0) v[0] = -2 - v[0];
    vs[i] = v[0];
  }
  for (int i = 1; i < num; i++) {
    if (vs[i - 1] > vs[i]) {
      float temp = vs[i];
      float tx = int_pts[2 * i];
      float ty = int_pts[2 * i + 1];
      int j = i;
      while (j > 0 && vs[j - 1] > temp) {
        vs[j] = vs[j - 1];
        int_pts[j * 2] = int_pts[j * 2 - 2];
        int_pts[j * 2 + 1] = int_pts[j * 2 - 1];
        j--;
      }
      vs[j] = temp;
      int_pts[j * 2] = tx;
      int_pts[j * 2 + 1] = ty;
    }
  }
}

// nms_gpu.py:222-265
bool line_segment_intersection(const float *pts1, const float *pts2, int i, int j, float *temp_pts) {
  float A[2], B[2], C[2], D[2];
  A[0] = pts1[2 * i];
  A[1] = pts1[2 * i + 1];
  B[0] = pts1[2 * ((i + 1) % 4)];
  B[1] = pts1[2 * ((i + 1) % 4) + 1];
  C[0] = pts2[2 * j];
  C[1] = pts2[2 * j + 1];
  D[0] = pts2[2 * ((j + 1) % 4)];
  D[1] = pts2[2 * ((j + 1) % 4) + 1];
  float BA0 = B[0] - A[0], BA1 = B[1] - A[1];
  float DA0 = D[0] - A[0], CA0 = C[0] - A[0];
  float DA1 = D[1] - A[1], CA1 = C[1] - A[1];
  bool acd = DA1 * CA0 > CA1 * DA0;
  bool bcd = (D[1] - B[1]) * (C[0] - B[0]) > (C[1] - B[1]) * (D[0] - B[0]);
  if (acd != bcd) {
    bool abc = CA1 * BA0 > BA1 * CA0;
    bool abd = DA1 * BA0 > BA1 * DA0;
    if (abc != abd) {
      float DC0 = D[0] - C[0], DC1 = D[1] - C[1];
      float ABBA = A[0] * B[1] - B[0] * A[1];
      float CDDC = C[0] * D[1] - D[0] * C[1];
      float DH = BA1 * DC0 - BA0 * DC1;
      float Dx = ABBA * DC0 - BA0 * CDDC;
      float Dy = ABBA * DC1 - BA1 * CDDC;
      temp_pts[0] = Dx / DH;
      temp_pts[1] = Dy / DH;
      return true;
    }
  }
  return false;
}

// nms_gpu.py:310-328.  g_eps_variant selects the commented-out tolerance form of :326-327
// (`eps = -1e-6`), the code state in which the values recorded in test_nms_gpu.py:14-15 were
// printed; it is used by orc_rotate_iou_raw only.
bool g_eps_variant = false;
bool point_in_quadrilateral(float pt_x, float pt_y, const float *corners) {
  float ab0 = corners[2] - corners[0], ab1 = corners[3] - corners[1];
  float ad0 = corners[6] - corners[0], ad1 = corners[7] - corners[1];
  float ap0 = pt_x - corners[0], ap1 = pt_y - corners[1];
  float abab = ab0 * ab0 + ab1 * ab1;
  float abap = ab0 * ap0 + ab1 * ap1;
  float adad = ad0 * ad0 + ad1 * ad1;
  float adap = ad0 * ap0 + ad1 * ap1;
  if (g_eps_variant) {
    const float eps = -1e-6f;
    return abab - abap >= eps && abap >= eps && adad - adap >= eps && adap >= eps;
  }
  return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

// nms_gpu.py:331-352.  The reference writes up to 24 candidate points into a 16-float local
// array (undefined behaviour past 8 points); here the buffer simply holds all 24.
int quadrilateral_intersection(const float *pts1, const float *pts2, float *int_pts) {
  int num = 0;
  for (int i = 0; i < 4; i++) {
    if (point_in_quadrilateral(pts1[2 * i], pts1[2 * i + 1], pts2)) {
      int_pts[num * 2] = pts1[2 * i];
      int_pts[num * 2 + 1] = pts1[2 * i + 1];
      num++;
    }
    if (point_in_quadrilateral(pts2[2 * i], pts2[2 * i + 1], pts1)) {
      int_pts[num * 2] = pts2[2 * i];
      int_pts[num * 2 + 1] = pts2[2 * i + 1];
      num++;
    }
  }
  float temp_pts[2];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      if (line_segment_intersection(pts1, pts2, i, j, temp_pts)) {
        int_pts[num * 2] = temp_pts[0];
        int_pts[num * 2 + 1] = temp_pts[1];
        num++;
      }
  return num;
}

// nms_gpu.py:355-378 (cos/sin: fp64 rounded to fp32, see header)
void rbbox_to_corners(float *corners, const float *rbbox) {
  float angle = rbbox[4];
  float a_cos = (float)std::cos((double)angle);
  float a_sin = (float)std::sin((double)angle);
  float center_x = rbbox[0], center_y = rbbox[1];
  float x_d = rbbox[2], y_d = rbbox[3];
  float cx[4], cy[4];
  cx[0] = -x_d / 2;
  cx[1] = -x_d / 2;
  cx[2] = x_d / 2;
  cx[3] = x_d / 2;
  cy[0] = -y_d / 2;
  cy[1] = y_d / 2;
  cy[2] = y_d / 2;
  cy[3] = -y_d / 2;
  for (int i = 0; i < 4; i++) {
    corners[2 * i] = a_cos * cx[i] + a_sin * cy[i] + center_x;
    corners[2 * i + 1] = -a_sin * cx[i] + a_cos * cy[i] + center_y;
  }
}

// nms_gpu.py:381-395
double inter(const float *rbbox1, const float *rbbox2) {
  float corners1[8], corners2[8], ic[48];
  rbbox_to_corners(corners1, rbbox1);
  rbbox_to_corners(corners2, rbbox2);
  int num = quadrilateral_intersection(corners1, corners2, ic);
  sort_vertex_in_convex_polygon(ic, num);
  return poly_area(ic, num);
}

// nms_gpu.py:552-570
float devRotateIoUEval(const float *rbox1, const float *rbox2, int criterion) {
  float area1 = rbox1[2] * rbox1[3];
  float area2 = rbox2[2] * rbox2[3];
  double area_inter = inter(rbox1, rbox2);
  if (criterion == -1) return (float)(area_inter / ((double)(area1 + area2) - area_inter));
  if (criterion == 0) return (float)(area_inter / area1);
  if (criterion == 1) return (float)(area_inter / area2);
  if (criterion == 2) {
    bool small = std::min(rbox2[2], rbox2[3]) / std::max(rbox2[2], rbox2[3]) < 0.25;
    if (small) return (float)(area_inter / ((double)area2 + std::max(0.0, (double)area1 * 0.5 - area_inter)));
    return (float)(area_inter / ((double)(area1 + area2) - area_inter));
  }
  return (float)area_inter;
}

// fp64 convex clipping (Sutherland-Hodgman) of quad P by quad Q; returns area of the result.
double shoelace(const double *p, int n) {
  double s = 0;
  for (int i = 0; i < n; i++) {
    int j = (i + 1) % n;
    s += p[2 * i] * p[2 * j + 1] - p[2 * j] * p[2 * i + 1];
  }
  return std::fabs(s) * 0.5;
}
double quad_inter_area(const float *P, const float *Q) {
  double a[32], b[32];
  int na = 4;
  for (int i = 0; i < 8; i++) a[i] = P[i];
  // orientation of Q
  double sq = 0;
  for (int i = 0; i < 4; i++) {
    int j = (i + 1) % 4;
    sq += (double)Q[2 * i] * Q[2 * j + 1] - (double)Q[2 * j] * Q[2 * i + 1];
  }
  double sgn = sq >= 0 ? 1.0 : -1.0;
  for (int e = 0; e < 4 && na > 0; e++) {
    double x1 = Q[2 * e], y1 = Q[2 * e + 1];
    double x2 = Q[2 * ((e + 1) % 4)], y2 = Q[2 * ((e + 1) % 4) + 1];
    int nb = 0;
    for (int i = 0; i < na; i++) {
      int j = (i + 1) % na;
      double cx = a[2 * i], cy = a[2 * i + 1], nx = a[2 * j], ny = a[2 * j + 1];
      double dc = sgn * ((x2 - x1) * (cy - y1) - (y2 - y1) * (cx - x1));
      double dn = sgn * ((x2 - x1) * (ny - y1) - (y2 - y1) * (nx - x1));
      if (dc >= 0) {
        b[2 * nb] = cx;
        b[2 * nb + 1] = cy;
        nb++;
      }
      if ((dc >= 0) != (dn >= 0)) {
        double t = dc / (dc - dn);
        b[2 * nb] = cx + t * (nx - cx);
        b[2 * nb + 1] = cy + t * (ny - cy);
        nb++;
      }
    }
    na = nb;
    std::memcpy(a, b, sizeof(double) * 2 * nb);
  }
  if (na < 3) return 0.0;
  return shoelace(a, na);
}

}  // namespace

extern "C" {

// rotate_iou_gpu_eval (nms_gpu.py:614-650) incl. check_same_boxes (:653-664).
// boxes [N,5], query [K,5] = (xc, yc, d0, d1, angle); out[N,K]; out[n,k] =
// devRotateIoUEval(query[k], boxes[n]) as the kernel does (:605-611).
void orc_rotate_iou_eval(const float *boxes, int N, const float *query, int K, int criterion,
                         float *out) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int n = 0; n < N; n++)
    for (int k = 0; k < K; k++) {
      float v = devRotateIoUEval(query + 5 * k, boxes + 5 * n, criterion);
      bool same = true;
      for (int d = 0; d < 5; d++)
        same = same && (std::fabs(boxes[5 * n + d] - query[5 * k + d]) < (float)1e-6);
      out[(size_t)n * K + k] = same ? 1.f : v;
    }
}

// Same without the check_same_boxes override: the state of the code when the values recorded in
// second/core/non_max_suppression/test_nms_gpu.py:14-15 were printed.
void orc_rotate_iou_raw(const float *boxes, int N, const float *query, int K, int criterion,
                        int eps_variant, float *out) {
  g_eps_variant = eps_variant != 0;
  for (int n = 0; n < N; n++)
    for (int k = 0; k < K; k++)
      out[(size_t)n * K + k] = devRotateIoUEval(query + 5 * k, boxes + 5 * n, criterion);
  g_eps_variant = false;
}

// boxes_iou_3d (utils3d/rotate_nms_3d_torch.py:23-88) + iou_one_dim (:7-21).
// targets [M,7], anchors [N,7] in yx_zb = (xc, yc, z_bot, d3, d4, dz, yaw).
// aug = {target_Y, target_Z, anchor_Y, anchor_Z} thickness clamps.  out [M,N].
void orc_boxes_iou_3d(const float *targets, int M, const float *anchors, int N, const float *aug,
                      int criterion, int only_xy, float *out) {
  std::vector<float> t2(5 * (size_t)M), a2(5 * (size_t)N), tz(2 * (size_t)M), az(2 * (size_t)N);
  for (int i = 0; i < M; i++) {
    const float *b = targets + 7 * i;
    float d3 = std::max(b[3], aug[0]), d5 = std::max(b[5], aug[1]);
    t2[5 * i + 0] = b[0]; t2[5 * i + 1] = b[1]; t2[5 * i + 2] = d3; t2[5 * i + 3] = b[4]; t2[5 * i + 4] = b[6];
    tz[2 * i] = b[2];
    tz[2 * i + 1] = b[2] + d5;
  }
  for (int i = 0; i < N; i++) {
    const float *b = anchors + 7 * i;
    float d3 = std::max(b[3], aug[2]), d5 = std::max(b[5], aug[3]);
    a2[5 * i + 0] = b[0]; a2[5 * i + 1] = b[1]; a2[5 * i + 2] = d3; a2[5 * i + 3] = b[4]; a2[5 * i + 4] = b[6];
    az[2 * i] = b[2];
    az[2 * i + 1] = b[2] + d5;
  }
  orc_rotate_iou_eval(t2.data(), M, a2.data(), N, criterion, out);
  if (only_xy) return;
  for (int i = 0; i < M; i++)
    for (int j = 0; j < N; j++) {
      float overlap = std::min(az[2 * j + 1], tz[2 * i + 1]) - std::max(az[2 * j], tz[2 * i]);
      float common = std::max(az[2 * j + 1], tz[2 * i + 1]) - std::min(az[2 * j], tz[2 * i]);
      out[(size_t)i * N + j] = out[(size_t)i * N + j] * (overlap / common);
    }
}

// center_to_corner_box2d (second/core/box_np_ops.py:374-394, corners_nd :176-207,
// rotation_2d :313-326) in float32: corners [n,4,2].
void orc_bev_corners(const float *xy, const float *dims, const float *angle, int n, float *corners) {
  const float nx[4] = {-0.5f, -0.5f, 0.5f, 0.5f}, ny[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
  for (int i = 0; i < n; i++) {
    float c = (float)std::cos((double)angle[i]), s = (float)std::sin((double)angle[i]);
    for (int k = 0; k < 4; k++) {
      float px = dims[2 * i] * nx[k], py = dims[2 * i + 1] * ny[k];
      corners[(i * 4 + k) * 2 + 0] = px * c + py * s + xy[2 * i];
      corners[(i * 4 + k) * 2 + 1] = px * (-s) + py * c + xy[2 * i + 1];
    }
  }
}

// spconv rotate_non_max_suppression_cpu (un-vendored; header).  corners [n,4,2], order[n],
// gate [n,n]; returns number kept, keep[] holds indices in selection order.
int orc_rotate_nms_cpu(const float *corners, const int32_t *order, const float *gate, int n,
                       float thresh, int32_t *keep) {
  std::vector<char> sup(n, 0);
  std::vector<double> area(n);
  for (int i = 0; i < n; i++) {
    double p[8];
    for (int k = 0; k < 8; k++) p[k] = corners[i * 8 + k];
    area[i] = shoelace(p, 4);
  }
  int nk = 0;
  for (int _i = 0; _i < n; _i++) {
    int i = order[_i];
    if (sup[i]) continue;
    keep[nk++] = i;
    for (int _j = _i + 1; _j < n; _j++) {
      int j = order[_j];
      if (sup[j]) continue;
      if (gate[(size_t)i * n + j] <= 0.0f) continue;
      double ia = quad_inter_area(corners + i * 8, corners + j * 8);
      if (ia <= 0) continue;
      double ua = area[i] + area[j] - ia;
      if (ua > 0 && ia / ua >= (double)thresh) sup[j] = 1;
    }
  }
  return nk;
}

// rotate_nms_3d_cc (second/core/non_max_suppression/nms_cpu.py:32-44) for boxes that were
// already clamped / top-k'd by the callers (boxlist_ops_3d.py:41-60, box_torch_ops.py:489-514).
// boxes [n,7] yx_zb, scores[n].  Order: descending score, ties -> lower index first.
int orc_rotate_nms_3d(const float *boxes, const float *scores, int n, float thresh, int32_t *keep) {
  if (n == 0) return 0;
  std::vector<float> iou((size_t)n * n);
  float aug[4] = {0, 0, 0, 0};
  orc_boxes_iou_3d(boxes, n, boxes, n, aug, -1, 0, iou.data());
  std::vector<int32_t> order(n);
  for (int i = 0; i < n; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scores[a] > scores[b]; });
  std::vector<float> xy(2 * (size_t)n), dims(2 * (size_t)n), ang(n), corners(8 * (size_t)n);
  for (int i = 0; i < n; i++) {
    xy[2 * i] = boxes[7 * i];
    xy[2 * i + 1] = boxes[7 * i + 1];
    dims[2 * i] = boxes[7 * i + 3];
    dims[2 * i + 1] = boxes[7 * i + 4];
    ang[i] = boxes[7 * i + 6];
  }
  orc_bev_corners(xy.data(), dims.data(), ang.data(), n, corners.data());
  return orc_rotate_nms_cpu(corners.data(), order.data(), iou.data(), n, thresh, keep);
}

// BoxCoder3D.decode (maskrcnn_benchmark/modeling/box_coder_3d.py:38-65) =
// weights division, clamp of size deltas, second_box_decode(smooth_dim=True)
// (second/pytorch/core/box_torch_ops.py:51-88), limit_period(yaw, 0.5, pi)
// (utils3d/geometric_torch.py:4-10).  enc [n,7], anchors [n,7] -> out [n,7], all fp32.
void orc_box_decode(const float *enc, const float *anchors, int n, const float *weights,
                    float clip, float *out) {
  const float pi = (float)M_PI;  // torch: python float pi applied to an fp32 tensor
  for (int i = 0; i < n; i++) {
    const float *e = enc + 7 * i, *a = anchors + 7 * i;
    float t[7];
    for (int k = 0; k < 7; k++) t[k] = e[k] / weights[k];
    for (int k = 3; k < 6; k++) t[k] = std::min(t[k], clip);
    float xa = a[0], ya = a[1], za = a[2], wa = a[3], la = a[4], ha = a[5], ra = a[6];
    float diagonal = std::sqrt(la * la + wa * wa);
    float *o = out + 7 * i;
    o[0] = t[0] * diagonal + xa;
    o[1] = t[1] * diagonal + ya;
    o[2] = t[2] * ha + za;
    o[3] = (t[3] + 1) * wa;
    o[4] = (t[4] + 1) * la;
    o[5] = (t[5] + 1) * ha;
    float rg = t[6] + ra;
    o[6] = rg - std::floor(rg / pi + 0.5f) * pi;
  }
}

// RoIAlignRotated3DForward (maskrcnn_benchmark/csrc/cuda/ROIAlignRotated3D_cuda.cu:15-85,
// :89-177), T=float.  input [B,C,H,W,Z], rois [K,8] = (b, cw, ch, cz, w, h, z, theta_deg),
// out [K,C,ph,pw,pz].  Keeps the `zsize > zsize` quirk of :27 (z above the map is clamped).
static float roi_interp(const float *d, int H, int W, int Z, float y, float x, float z) {
  if (y < -1.0 || y > H || x < -1.0 || x > W || z < -1.0) return 0;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  if (z <= 0) z = 0;
  int y_low = (int)y, x_low = (int)x, z_low = (int)z, y_high, x_high, z_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
  if (z_low >= Z - 1) { z_high = z_low = Z - 1; z = (float)z_low; } else z_high = z_low + 1;
  float ly = y - y_low, lx = x - x_low, lz = z - z_low;
  float hy = 1. - ly, hx = 1. - lx, hz = 1. - lz;
  float v1 = d[(y_low * W + x_low) * Z + z_low], v2 = d[(y_low * W + x_high) * Z + z_low];
  float v3 = d[(y_high * W + x_low) * Z + z_low], v4 = d[(y_high * W + x_high) * Z + z_low];
  float v5 = d[(y_low * W + x_low) * Z + z_high], v6 = d[(y_low * W + x_high) * Z + z_high];
  float v7 = d[(y_high * W + x_low) * Z + z_high], v8 = d[(y_high * W + x_high) * Z + z_high];
  float w1 = hy * hx * hz, w2 = hy * lx * hz, w3 = ly * hx * hz, w4 = ly * lx * hz;
  float w5 = hy * hx * lz, w6 = hy * lx * lz, w7 = ly * hx * lz, w8 = ly * lx * lz;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4 + w5 * v5 + w6 * v6 + w7 * v7 + w8 * v8);
}

void orc_roi_align_rotated_3d(const float *input, int B, int C, int H, int W, int Z,
                              const float *rois, int K, float spatial_scale, int ph_, int pw_,
                              int pz_, int sampling_ratio, float *out) {
  (void)B;
#pragma omp parallel for schedule(static)
  for (int n = 0; n < K; n++) {
    const float *r = rois + 8 * n;
    int b = (int)r[0];
    float cw = r[1] * spatial_scale, ch = r[2] * spatial_scale, cz = r[3] * spatial_scale;
    float rw = r[4] * spatial_scale, rh = r[5] * spatial_scale, rz = r[6] * spatial_scale;
    float theta = (float)((double)r[7] * M_PI / 180.0);
    rw = std::max(rw, 1.f);
    rh = std::max(rh, 1.f);
    rz = std::max(rz, 1.f);
    float bh = rh / (float)ph_, bw = rw / (float)pw_, bz = rz / (float)pz_;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)std::ceil(rh / ph_);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)std::ceil(rw / pw_);
    int gz = sampling_ratio > 0 ? sampling_ratio : (int)std::ceil(rz / pz_);
    float sh = (float)(-rh / 2.0), sw = (float)(-rw / 2.0), sz = (float)(-rz / 2.0);
    float cosT = (float)std::cos((double)theta), sinT = (float)std::sin((double)theta);
    const float count = (float)(gh * gw * gz);
    for (int c = 0; c < C; c++) {
      const float *d = input + ((size_t)b * C + c) * H * W * Z;
      for (int ph = 0; ph < ph_; ph++)
        for (int pw = 0; pw < pw_; pw++)
          for (int pz = 0; pz < pz_; pz++) {
            float acc = 0.f;
            for (int iy = 0; iy < gh; iy++) {
              const float yy = sh + ph * bh + (float)(iy + .5f) * bh / (float)gh;
              for (int ix = 0; ix < gw; ix++) {
                const float xx = sw + pw * bw + (float)(ix + .5f) * bw / (float)gw;
                for (int iz = 0; iz < gz; iz++) {
                  const float zz = sz + pz * bz + (float)(iz + .5f) * bz / (float)gz;
                  float x = xx * cosT + yy * sinT + cw;
                  float y = yy * cosT - xx * sinT + ch;
                  float z = zz + cz;
                  acc += roi_interp(d, H, W, Z, y, x, z);
                }
              }
            }
            acc /= count;
            out[((((size_t)n * C + c) * ph_ + ph) * pw_ + pw) * pz_ + pz] = acc;
          }
    }
  }
}

// RoIAlignRotated3DBackwardFeature (ROIAlignRotated3D_cuda.cu:182-354), T=float, into a zeroed dense
// gradient [B,C,H,W,Z].  Unlike the forward (:27), the backward bound test is `z > zsize` (:190):
// samples above the map get no gradient.  Serial (the reference uses atomicAdd).
void orc_roi_align_rotated_3d_backward(const float *top_diff, int B, int C, int H, int W, int Z,
                                       const float *rois, int K, float spatial_scale, int ph_, int pw_,
                                       int pz_, int sampling_ratio, float *bottom_diff) {
  std::memset(bottom_diff, 0, sizeof(float) * (size_t)B * C * H * W * Z);
  for (int n = 0; n < K; n++) {
    const float *r = rois + 8 * n;
    int b = (int)r[0];
    float cw = r[1] * spatial_scale, ch = r[2] * spatial_scale, cz = r[3] * spatial_scale;
    float rw = r[4] * spatial_scale, rh = r[5] * spatial_scale, rz = r[6] * spatial_scale;
    float theta = (float)((double)r[7] * M_PI / 180.0);
    rw = std::max(rw, 1.f);
    rh = std::max(rh, 1.f);
    rz = std::max(rz, 1.f);
    float bh = rh / (float)ph_, bw = rw / (float)pw_, bz = rz / (float)pz_;
    int gh = sampling_ratio > 0 ? sampling_ratio : (int)std::ceil(rh / ph_);
    int gw = sampling_ratio > 0 ? sampling_ratio : (int)std::ceil(rw / pw_);
    int gz = sampling_ratio > 0 ? sampling_ratio : (int)std::ceil(rz / pz_);
    float sh = (float)(-rh / 2.0), sw = (float)(-rw / 2.0), sz = (float)(-rz / 2.0);
    float cosT = (float)std::cos((double)theta), sinT = (float)std::sin((double)theta);
    const float count = (float)(gh * gw * gz);
    for (int c = 0; c < C; c++) {
      float *d = bottom_diff + ((size_t)b * C + c) * H * W * Z;
      for (int ph = 0; ph < ph_; ph++)
        for (int pw = 0; pw < pw_; pw++)
          for (int pz = 0; pz < pz_; pz++) {
            const float g = top_diff[((((size_t)n * C + c) * ph_ + ph) * pw_ + pw) * pz_ + pz];
            for (int iy = 0; iy < gh; iy++) {
              const float yy = sh + ph * bh + (float)(iy + .5f) * bh / (float)gh;
              for (int ix = 0; ix < gw; ix++) {
                const float xx = sw + pw * bw + (float)(ix + .5f) * bw / (float)gw;
                for (int iz = 0; iz < gz; iz++) {
                  const float zz = sz + pz * bz + (float)(iz + .5f) * bz / (float)gz;
                  float x = xx * cosT + yy * sinT + cw;
                  float y = yy * cosT - xx * sinT + ch;
                  float z = zz + cz;
                  if (y < -1.0 || y > H || x < -1.0 || x > W || z < -1.0 || z > Z) continue;
                  if (y <= 0) y = 0;
                  if (x <= 0) x = 0;
                  if (z <= 0) z = 0;
                  int yl = (int)y, xl = (int)x, zl = (int)z, yh, xh, zh;
                  if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
                  if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
                  if (zl >= Z - 1) { zh = zl = Z - 1; z = (float)zl; } else zh = zl + 1;
                  float ly = y - yl, lx = x - xl, lz = z - zl;
                  float hy = 1. - ly, hx = 1. - lx, hz = 1. - lz;
                  float w[8] = {hy * hx * hz, hy * lx * hz, ly * hx * hz, ly * lx * hz,
                                hy * hx * lz, hy * lx * lz, ly * hx * lz, ly * lx * lz};
                  int yy_[8] = {yl, yl, yh, yh, yl, yl, yh, yh}, xx_[8] = {xl, xh, xl, xh, xl, xh, xl, xh};
                  int zz_[8] = {zl, zl, zl, zl, zh, zh, zh, zh};
                  for (int q = 0; q < 8; q++) d[(yy_[q] * W + xx_[q]) * Z + zz_[q]] += g * w[q] / count;
                }
              }
            }
          }
    }
  }
}

}  // extern "C"
