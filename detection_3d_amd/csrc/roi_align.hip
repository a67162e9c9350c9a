// a21. RoIAlignRotated3D forward (maskrcnn_benchmark/csrc/cuda/ROIAlignRotated3D_cuda.cu:15-177).
//  * dense variant: same contract as _C.roi_align_rotated_3d_forward (input [B,C,H,W,Z]);
//  * sparse variant: samples the SparseConvNetTensor through its hash grid, so the 1.07 GB dense
//    map of sparse_3d_to_dense_2d (sparseconvnet/tools_3d_2d.py:7-48) is never materialised.
// Both keep the reference's `zsize > zsize` bound quirk (:27): z above the map is clamped.
#include "d3d_internal.h"

namespace d3d {

struct RoiGeom {
  int b;
  float cw, ch, cz, bh, bw, bz, sh, sw, sz, cosT, sinT;
  int gh, gw, gz;
};
__device__ __forceinline__ RoiGeom roi_geom(const float *r, float spatial_scale, int PH, int PW, int PZ,
                                            int sampling_ratio) {
  RoiGeom g;
  g.b = (int)r[0];
  g.cw = r[1] * spatial_scale;
  g.ch = r[2] * spatial_scale;
  g.cz = r[3] * spatial_scale;
  float rw = r[4] * spatial_scale, rh = r[5] * spatial_scale, rz = r[6] * spatial_scale;
  const float theta = (float)((double)r[7] * 3.14159265358979323846 / 180.0);
  rw = fmaxf(rw, 1.f);
  rh = fmaxf(rh, 1.f);
  rz = fmaxf(rz, 1.f);
  g.bh = rh / (float)PH;
  g.bw = rw / (float)PW;
  g.bz = rz / (float)PZ;
  g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / PH);
  g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / PW);
  g.gz = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rz / PZ);
  g.sh = (float)(-rh / 2.0);
  g.sw = (float)(-rw / 2.0);
  g.sz = (float)(-rz / 2.0);
  g.cosT = (float)cos((double)theta);
  g.sinT = (float)sin((double)theta);
  return g;
}
// sample position of (bin, sub-sample) in map coordinates (:150-163)
__device__ __forceinline__ void sample_pos(const RoiGeom &g, int ph, int pw, int pz, int iy, int ix,
                                           int iz, float &y, float &x, float &z) {
  const float yy = g.sh + ph * g.bh + (float)(iy + .5f) * g.bh / (float)g.gh;
  const float xx = g.sw + pw * g.bw + (float)(ix + .5f) * g.bw / (float)g.gw;
  const float zz = g.sz + pz * g.bz + (float)(iz + .5f) * g.bz / (float)g.gz;
  x = xx * g.cosT + yy * g.sinT + g.cw;
  y = yy * g.cosT - xx * g.sinT + g.ch;
  z = zz + g.cz;
}
// interpolation set-up of bilinear_interpolate (:15-62): returns false for an empty sample
struct Tri {
  int yl, yh, xl, xh, zl, zh;
  float ly, lx, lz, hy, hx, hz;
};
__device__ __forceinline__ bool tri_setup(float y, float x, float z, int H, int W, int Z, Tri &t) {
  if (y < -1.0 || y > H || x < -1.0 || x > W || z < -1.0) return false;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  if (z <= 0) z = 0;
  t.yl = (int)y;
  t.xl = (int)x;
  t.zl = (int)z;
  if (t.yl >= H - 1) { t.yh = t.yl = H - 1; y = (float)t.yl; } else t.yh = t.yl + 1;
  if (t.xl >= W - 1) { t.xh = t.xl = W - 1; x = (float)t.xl; } else t.xh = t.xl + 1;
  if (t.zl >= Z - 1) { t.zh = t.zl = Z - 1; z = (float)t.zl; } else t.zh = t.zl + 1;
  t.ly = y - t.yl;
  t.lx = x - t.xl;
  t.lz = z - t.zl;
  t.hy = 1. - t.ly;
  t.hx = 1. - t.lx;
  t.hz = 1. - t.lz;
  return true;
}

__global__ __launch_bounds__(256) void k_roi_dense(const float *__restrict__ input, int C, int H,
                                                   int W, int Z, const float *__restrict__ rois,
                                                   long nthreads, float spatial_scale, int PH, int PW,
                                                   int PZ, int sampling_ratio, float *__restrict__ out) {
  long index = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= nthreads) return;
  const int pz = index % PZ;
  const int pw = (index / PZ) % PW;
  const int ph = (index / PZ / PW) % PH;
  const int c = (index / PZ / PW / PH) % C;
  const int n = index / PZ / PW / PH / C;
  const RoiGeom g = roi_geom(rois + (size_t)n * 8, spatial_scale, PH, PW, PZ, sampling_ratio);
  const float *d = input + ((size_t)g.b * C + c) * H * W * Z;
  const float count = (float)(g.gh * g.gw * g.gz);
  float acc = 0.f;
  for (int iy = 0; iy < g.gh; iy++)
    for (int ix = 0; ix < g.gw; ix++)
      for (int iz = 0; iz < g.gz; iz++) {
        float y, x, z;
        sample_pos(g, ph, pw, pz, iy, ix, iz, y, x, z);
        Tri t;
        if (!tri_setup(y, x, z, H, W, Z, t)) continue;
        const float v1 = d[(t.yl * W + t.xl) * Z + t.zl], v2 = d[(t.yl * W + t.xh) * Z + t.zl];
        const float v3 = d[(t.yh * W + t.xl) * Z + t.zl], v4 = d[(t.yh * W + t.xh) * Z + t.zl];
        const float v5 = d[(t.yl * W + t.xl) * Z + t.zh], v6 = d[(t.yl * W + t.xh) * Z + t.zh];
        const float v7 = d[(t.yh * W + t.xl) * Z + t.zh], v8 = d[(t.yh * W + t.xh) * Z + t.zh];
        const float w1 = t.hy * t.hx * t.hz, w2 = t.hy * t.lx * t.hz, w3 = t.ly * t.hx * t.hz, w4 = t.ly * t.lx * t.hz;
        const float w5 = t.hy * t.hx * t.lz, w6 = t.hy * t.lx * t.lz, w7 = t.ly * t.hx * t.lz, w8 = t.ly * t.lx * t.lz;
        acc += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4 + w5 * v5 + w6 * v6 + w7 * v7 + w8 * v8);
      }
  out[index] = acc / count;
}

// Sparse variant.  Block = one RoI x one chunk of 128 channels (lane = 2 adjacent channels); a wave owns
// groups of kRoiG consecutive bins.  Per group and per step of 8 sub-samples the 64 lanes first resolve
// (sub-sample, corner) -> (row, weight) through the hash grid for all kRoiG bins at once (kRoiG independent
// probes in flight per lane), compact the taps that exist into a per-bin LDS list, and then every lane
// accumulates its two channels over the lists with 8 independent feature-row loads in flight (each a
// coalesced 512-B read of the wave).  Everything is latency-bound L2 traffic: the feature map of a pyramid
// level is a few MB; what matters is the number of loads in flight, not bytes.
// layout 0: out[n][c][ph][pw][pz] (the reference's); layout 1: out[n][ph][pw][c][pz] (rows of the box head's
// [1,1,pz] convolution seen as a GEMM).  roi_levels (optional): only RoIs with roi_levels[i] == level are pooled.
static constexpr int kRoiG = 4;   // 4 bins per group and 4 waves per SIMD (128 VGPRs, 4 spilled): 286 us for the bench's 1000 RoIs;
                                 // 8 bins at 3 waves (164 VGPRs) 368 us, 4 bins at 3 waves 357 us, 4 bins at 5 waves: 32 spills
static constexpr int kRoiCch = 128;
__device__ __forceinline__ void roi_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
static constexpr int kRoiWaves = 4;  // waves per RoI (8 measured the same; what helps is waves per SIMD, see kRoiG)
// One launch serves every pyramid level: a RoI's workgroup picks the map of roi_levels[n] (a launch per level leaves
// the workgroups of the other levels' RoIs to exit at once -- with two levels each launch fills half the wave slots).
struct RoiLevel {
  const HashEntry *tab;   // null: this level is not pooled by the launch
  const float *feats;
  const int32_t *extent;  // occupied extent of the grid on the device, or null: H, W, Z below
  const int32_t *dense;   // null, or the dense index of the grid's bounding box [b][e0][e1][e2] (1 + site id, 0 = empty)
  int cap, H, W, Z;
  int e0, e1, e2;
  float scale;
};
static constexpr int kRoiMaxLevels = 4;
struct RoiLevels {
  RoiLevel v[kRoiMaxLevels];
};
__global__ __launch_bounds__(kRoiWaves * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_roi_sparse(
    RoiLevels lv, int C, const float *__restrict__ rois, const int32_t *__restrict__ roi_levels,
    int PH, int PW, int PZ, int sampling_ratio, int layout, float *__restrict__ out) {
  const int n = blockIdx.x, cc = blockIdx.y;
  const int l = roi_levels ? __builtin_amdgcn_readfirstlane(roi_levels[n]) : 0;
  if (l < 0 || l >= kRoiMaxLevels) return;   // -1: a padding row (d3d_roi_prepare_counted)
  RoiLevel L = lv.v[0];                       // selects, not an indexed copy of the argument block
  if (l == 1) L = lv.v[1];
  if (l == 2) L = lv.v[2];
  if (l == 3) L = lv.v[3];
  if (!L.tab) return;                         // pooled from another pyramid level (by another launch)
  const HashEntry *__restrict__ tab = L.tab;
  const int32_t *__restrict__ dense = L.dense;
  const float *__restrict__ feats = L.feats;
  const int cap = L.cap;
  const float spatial_scale = L.scale;
  int H = L.H, W = L.W, Z = L.Z;
  if (L.extent) {  // crop = occupied extent of the grid, read on the device (no host round trip)
    H = L.extent[0];
    W = L.extent[1];
    Z = L.extent[2];
  }
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  __shared__ int2 list[kRoiWaves][kRoiG][64];  // (row, weight bits) of the taps that exist
  const int NB = PH * PW * PZ;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const RoiGeom g = roi_geom(rois + (size_t)n * 8, spatial_scale, PH, PW, PZ, sampling_ratio);
  const int NS = g.gh * g.gw * g.gz;
  const float count = (float)NS;
  const int c0 = cc * kRoiCch + 2 * lane;
  const bool ok0 = c0 < C, ok1 = c0 + 1 < C;
  const bool pair = ok1 && (C % 2 == 0);  // 8-byte aligned pair load
  const size_t n_out = (size_t)n;
  auto load2 = [&](int row) -> f32x2 {
    const float *p = feats + (size_t)row * C + c0;
    f32x2 v = {0.f, 0.f};
    if (pair)
      v = *(const f32x2 *)p;
    else {
      if (ok0) v[0] = p[0];
      if (ok1) v[1] = p[1];
    }
    return v;
  };
  const int ngroups = (NB + kRoiG - 1) / kRoiG;
  for (int grp = wave; grp < ngroups; grp += kRoiWaves) {
    const int b0 = grp * kRoiG;
    f32x2 acc[kRoiG];
#pragma unroll
    for (int gi = 0; gi < kRoiG; gi++) acc[gi] = {0.f, 0.f};
    for (int s0 = 0; s0 < NS; s0 += 8) {
      // ---- lanes = (sub-sample s0 + lane/8, corner lane%8), kRoiG bins at once ----
      const int s = s0 + (lane >> 3), corner = lane & 7;
      const int iz = s % g.gz, ix = (s / g.gz) % g.gw, iy = s / (g.gz * g.gw);
      const int zb = corner >> 2, yb = (corner >> 1) & 1, xb = corner & 1;
      int row[kRoiG];
      float wgt[kRoiG];
      // the probes of a lane four bins at a time in lockstep (hash_find_n: their round trips overlap; all kRoiG at
      // once costs a wave of occupancy in registers)
      static_assert(kRoiG % 4 == 0, "probe groups of four");
#pragma unroll
      for (int g0 = 0; g0 < kRoiG; g0 += 4) {
        uint64_t key[4];
        bool want[4];
        int r4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int bin = b0 + g0 + j;
          wgt[g0 + j] = 0.f;
          key[j] = 0;
          want[j] = false;
          if (s < NS && bin < NB) {
            const int pz = bin % PZ, pw = (bin / PZ) % PW, ph = bin / (PZ * PW);
            float y, x, z;
            sample_pos(g, ph, pw, pz, iy, ix, iz, y, x, z);
            Tri t;
            if (tri_setup(y, x, z, H, W, Z, t)) {
              wgt[g0 + j] = (yb ? t.ly : t.hy) * (xb ? t.lx : t.hx) * (zb ? t.lz : t.hz);
              // dense index [y][x][z]: y runs over the tensor's 1st spatial axis, x over the 2nd
              key[j] = pack_key(g.b, yb ? t.yh : t.yl, xb ? t.xh : t.xl, zb ? t.zh : t.zl);
              want[j] = true;
            }
          }
        }
        if (dense) {       // one load per corner: the cell of the grid's dense index (coordinates are inside the box:
                           // tri_setup clamps them to the occupied extent, which the box contains)
          int v4[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint64_t kk = key[j];
            const size_t cell = ((((size_t)(kk >> 48) * L.e0 + (size_t)((kk >> 32) & 0xffff)) * L.e1 +
                                  (size_t)((kk >> 16) & 0xffff)) * L.e2 + (size_t)(kk & 0xffff));
            v4[j] = want[j] ? dense[cell] : 0;
          }
#pragma unroll
          for (int j = 0; j < 4; j++) r4[j] = v4[j] - 1;
        } else {
          hash_find_n<4>(tab, cap, key, want, r4);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) row[g0 + j] = r4[j];
      }
      // the 64 taps of a bin fall into a handful of cells (its sub-samples are a fraction of a cell apart): merge
      // the taps of one cell (weights summed by a wave butterfly) so that each feature row is fetched once per bin
      // and step -- the kernel is bound by the L2 -> L1 bytes of those 512-B rows, not by instructions
      int cnt[kRoiG];
#pragma unroll
      for (int gi = 0; gi < kRoiG; gi++) {
        unsigned long long m = __ballot(row[gi] >= 0);
        int c = 0;
        while (m) {
          const int rl = __shfl(row[gi], __builtin_ctzll(m), 64);   // wave-uniform cell
          const bool mine = row[gi] == rl;
          float w = mine ? wgt[gi] : 0.f;
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) w += __shfl_xor(w, d, 64);
          if (lane == 0) list[wave][gi][c] = make_int2(rl, __float_as_int(w));
          c++;
          m &= ~__ballot(mine);
        }
        cnt[gi] = c;
      }
      roi_wave_sync();
      // ---- lanes = channel pairs: one entry per cell, batches of independent row loads ----
#pragma unroll
      for (int gi = 0; gi < kRoiG; gi++) {
        const int2 *L = list[wave][gi];
        const int nc = cnt[gi];
        int i = 0;
        for (; i + 8 <= nc; i += 8) {
          int2 e[8];
          f32x2 v[8];
#pragma unroll
          for (int j = 0; j < 8; j++) e[j] = L[i + j];
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] = load2(e[j].x);
#pragma unroll
          for (int j = 0; j < 8; j++) acc[gi] += __int_as_float(e[j].y) * v[j];
        }
        if (i + 4 <= nc) {
          int2 e[4];
          f32x2 v[4];
#pragma unroll
          for (int j = 0; j < 4; j++) e[j] = L[i + j];
#pragma unroll
          for (int j = 0; j < 4; j++) v[j] = load2(e[j].x);
#pragma unroll
          for (int j = 0; j < 4; j++) acc[gi] += __int_as_float(e[j].y) * v[j];
          i += 4;
        }
        if (i + 2 <= nc) {
          const int2 e0 = L[i], e1 = L[i + 1];
          const f32x2 v0 = load2(e0.x), v1 = load2(e1.x);
          acc[gi] += __int_as_float(e0.y) * v0;
          acc[gi] += __int_as_float(e1.y) * v1;
          i += 2;
        }
        if (i < nc) {
          const int2 e0 = L[i];
          acc[gi] += __int_as_float(e0.y) * load2(e0.x);
        }
      }
      roi_wave_sync();  // the lists are rewritten by the next step
    }
#pragma unroll
    for (int gi = 0; gi < kRoiG; gi++) {
      const int bin = b0 + gi;
      if (bin >= NB) break;
      const f32x2 r = acc[gi] / count;
      if (layout == 0) {
        float *o = out + (n_out * C + c0) * NB + bin;
        if (ok0) o[0] = r[0];
        if (ok1) o[NB] = r[1];
      } else {
        const int pz = bin % PZ, cell = bin / PZ;
        float *o = out + ((n_out * (size_t)(PH * PW) + cell) * C + c0) * PZ + pz;
        if (ok0) o[0] = r[0];
        if (ok1) o[PZ] = r[1];
      }
    }
  }
}

// Dense backward, _C.roi_align_rotated_3d_backward (ROIAlignRotated3D_cuda.cu:238-354): one thread per
// pooled element, 8 fp32 atomics per sub-sample into the zeroed dense gradient.
__global__ __launch_bounds__(256) void k_roi_dense_bwd(const float *__restrict__ top_diff, int C, int H, int W,
                                                       int Z, const float *__restrict__ rois, long nthreads,
                                                       float spatial_scale, int PH, int PW, int PZ,
                                                       int sampling_ratio, float *__restrict__ bottom_diff) {
  long index = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= nthreads) return;
  const int pz = index % PZ;
  const int pw = (index / PZ) % PW;
  const int ph = (index / PZ / PW) % PH;
  const int c = (index / PZ / PW / PH) % C;
  const int n = index / PZ / PW / PH / C;
  const RoiGeom g = roi_geom(rois + (size_t)n * 8, spatial_scale, PH, PW, PZ, sampling_ratio);
  float *d = bottom_diff + ((size_t)g.b * C + c) * H * W * Z;
  const float count = (float)(g.gh * g.gw * g.gz);
  const float top = top_diff[index];
  for (int iy = 0; iy < g.gh; iy++)
    for (int ix = 0; ix < g.gw; ix++)
      for (int iz = 0; iz < g.gz; iz++) {
        float y, x, z;
        sample_pos(g, ph, pw, pz, iy, ix, iz, y, x, z);
        Tri t;
        if (z > Z || !tri_setup(y, x, z, H, W, Z, t)) continue;   // backward bound test (:190)
        const float w[8] = {t.hy * t.hx * t.hz, t.hy * t.lx * t.hz, t.ly * t.hx * t.hz, t.ly * t.lx * t.hz,
                            t.hy * t.hx * t.lz, t.hy * t.lx * t.lz, t.ly * t.hx * t.lz, t.ly * t.lx * t.lz};
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int yy = (q >> 1) & 1 ? t.yh : t.yl, xx = q & 1 ? t.xh : t.xl, zz = q >> 2 ? t.zh : t.zl;
          atomicAdd(d + ((size_t)yy * W + xx) * Z + zz, top * w[q] / count);
        }
      }
}

// Backward of the sparse variant: the dense gradient of RoIAlignRotated3DBackwardFeature (:238-354)
// restricted to the active sites (what SparseToDense_updateGradInput would gather back).  Keeps the
// backward's own bound test `z > zsize` (:190).  fp32 atomics, like the reference.
static constexpr int kRoiBwdCch = 64;
__global__ __launch_bounds__(256) void k_roi_sparse_bwd(
    const HashEntry *__restrict__ tab, int cap, int C, int H, int W, int Z, const float *__restrict__ rois,
    float spatial_scale, int PH, int PW, int PZ, int sampling_ratio, const float *__restrict__ top_diff,
    float *__restrict__ d_feats) {
  extern __shared__ float tile[];  // [kRoiBwdCch][NB + 1]
  const int n = blockIdx.x, cc = blockIdx.y;
  const int NB = PH * PW * PZ, LD = NB + 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nch = min(kRoiBwdCch, C - cc * kRoiBwdCch);
  const float *g0 = top_diff + ((size_t)n * C + (size_t)cc * kRoiBwdCch) * NB;
  for (int idx = threadIdx.x; idx < nch * NB; idx += 256) tile[(idx / NB) * LD + idx % NB] = g0[idx];
  __syncthreads();
  const RoiGeom g = roi_geom(rois + (size_t)n * 8, spatial_scale, PH, PW, PZ, sampling_ratio);
  const int NS = g.gh * g.gw * g.gz;
  const float count = (float)NS;
  const int c = cc * kRoiBwdCch + lane;
  const bool cok = c < C;
  for (int bin = wave; bin < NB; bin += 4) {
    const int pz = bin % PZ, pw = (bin / PZ) % PW, ph = bin / (PZ * PW);
    const float top = cok ? tile[lane * LD + bin] : 0.f;
    for (int s0 = 0; s0 < NS; s0 += 8) {
      const int s = s0 + (lane >> 3), corner = lane & 7;
      int row = -1;
      float wgt = 0.f;
      if (s < NS) {
        const int iz = s % g.gz, ix = (s / g.gz) % g.gw, iy = s / (g.gz * g.gw);
        float y, x, z;
        sample_pos(g, ph, pw, pz, iy, ix, iz, y, x, z);
        Tri t;
        if (!(z > Z) && tri_setup(y, x, z, H, W, Z, t)) {
          const int zb = corner >> 2, yb = (corner >> 1) & 1, xb = corner & 1;
          wgt = (yb ? t.ly : t.hy) * (xb ? t.lx : t.hx) * (zb ? t.lz : t.hz);
          row = hash_find(tab, cap, pack_key(g.b, yb ? t.yh : t.yl, xb ? t.xh : t.xl, zb ? t.zh : t.zl));
        }
      }
      // the 64 taps of a step fall into a handful of cells: one atomic per (cell, channel) with the taps' weights summed
      // by a wave butterfly (as the forward does) instead of one per tap -- 5x fewer atomics on the hot rows
      unsigned long long m = __ballot(row >= 0);
      while (m) {
        const int rr = __shfl(row, __builtin_ctzll(m), 64);   // wave-uniform cell
        const bool mine = row == rr;
        float ww = mine ? wgt : 0.f;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) ww += __shfl_xor(ww, d, 64);
        if (cok) atomicAdd(d_feats + (size_t)rr * C + c, top * ww / count);
        m &= ~__ballot(mine);
      }
    }
  }
}

}  // namespace d3d

using namespace d3d;

// Pooler pre-processing in one launch (what the reference spreads over ~25 tensor ops): metric yx_zb proposals ->
// the op's RoI rows (batch id, centre x, centre y, centre z, size x, size y, size z, yaw in degrees, in pixels of the
// full-resolution grid) and the FPN level of every RoI.  The arithmetic is the reference's, operation by operation in
// fp32 (a division by a constant is the product with its fp32 reciprocal, as the tensor library evaluates it), so
// that the result equals the host-side chain bit for bit (tests/test_boxes_gpu.py).
struct RoiPrepScales {
  float v[8];
};
__global__ __launch_bounds__(256) void k_roi_prepare(const float *__restrict__ boxes, int n, float voxel_scale,
                                                     RoiPrepScales scales, int n_levels, float inv_canonical,
                                                     float *__restrict__ rois, int32_t *__restrict__ levels,
                                                     const int32_t *__restrict__ batch_ids,
                                                     const int32_t *__restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (count && i >= *count) {   // padding row of a list whose length is still on the device: no level pools it
    float *r = rois + (size_t)i * 8;
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = 0.f;
    if (levels) levels[i] = -1;
    return;
  }
  float b[7];
#pragma unroll
  for (int j = 0; j < 7; j++) b[j] = boxes[(size_t)i * 7 + j];
#pragma unroll
  for (int j = 0; j < 6; j++) b[j] *= voxel_scale;                  // convert_metric_to_pixel
  const float kHalfPi = (float)(3.14159265358979323846 * 0.5), kPi = (float)3.14159265358979323846;
  const float kInvPi = 1.f / kPi, kDeg = (float)(180.0 / 3.14159265358979323846);
  float yaw = b[6] + kHalfPi;                                        // yx_zb -> standard (bounding_box_3d.py:221-242)
  yaw = yaw - floorf(yaw * kInvPi + 0.f) * kPi;                      // limit_period(yaw, 0, pi)
  float *r = rois + (size_t)i * 8;
  r[0] = batch_ids ? (float)batch_ids[i] : 0.f;        // example index of the RoI (poolers_3d.py:112-118)
  r[1] = b[1];
  r[2] = b[0];
  r[3] = b[2] + b[5] * 0.5f;
  r[4] = b[3];
  r[5] = b[4];
  r[6] = b[5];
  r[7] = yaw * kDeg;
  if (levels) {                                                      // poolers_3d.py LevelMapper
    const float rate = sqrtf(fmaxf(b[3], b[4])) * inv_canonical;
    int best = 0;
    float bd = fabsf(scales.v[0] - rate);
    for (int l = 1; l < n_levels; l++) {
      const float d = fabsf(scales.v[l] - rate);
      if (d < bd) {
        bd = d;
        best = l;
      }
    }
    levels[i] = best;
  }
}

extern "C" {

int d3d_roi_prepare(const float *boxes_metric, int n, float voxel_scale, const float *scales_host, int n_levels,
                    float canonical_size, const int32_t *batch_ids, float *rois, int32_t *levels, void *stream) {
  return d3d_roi_prepare_counted(boxes_metric, n, nullptr, voxel_scale, scales_host, n_levels, canonical_size, batch_ids,
                                 rois, levels, stream);
}

int d3d_roi_prepare_counted(const float *boxes_metric, int n, const int32_t *count_dev, float voxel_scale,
                            const float *scales_host, int n_levels, float canonical_size, const int32_t *batch_ids,
                            float *rois, int32_t *levels, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(n >= 0 && n_levels >= 0 && n_levels <= 8, "roi_prepare: bad arguments (at most 8 levels)");
  if (n == 0) return D3D_OK;
  D3D_REQUIRE(boxes_metric && rois && (n_levels == 0 || scales_host), "roi_prepare: null pointer");
  RoiPrepScales sc = {};
  for (int l = 0; l < n_levels; l++) sc.v[l] = scales_host[l];
  hipLaunchKernelGGL(k_roi_prepare, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, boxes_metric, n, voxel_scale, sc,
                     n_levels, 1.f / canonical_size, rois, n_levels > 1 ? levels : nullptr, batch_ids, count_dev);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_roi_align_rotated_3d_forward(const float *input, int B, int C, int H, int W, int Z,
                                     const float *rois, int K, float spatial_scale, int ph, int pw,
                                     int pz, int sampling_ratio, float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Z > 0 && K >= 0 && ph > 0 && pw > 0 && pz > 0, "roi_align: bad shape");
  if (K == 0) return D3D_OK;
  D3D_REQUIRE(input && rois && out, "roi_align: null pointer");
  long nthreads = (long)K * C * ph * pw * pz;
  hipLaunchKernelGGL(k_roi_dense, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, input, C, H, W, Z, rois, nthreads, spatial_scale, ph, pw, pz, sampling_ratio, out);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

// fills lv (one pyramid level of a pooling launch) from the grid of spatial size `size`
static int roi_level_of(d3d_meta *m, const int *size, const float *feats, const int *crop, float spatial_scale,
                        hipStream_t s, RoiLevel *lv) {
  std::map<Size3, Grid>::iterator it;
  bool have_grid;
  {
    D3D_LOCK(m);
    it = m->grids.find(Size3{size[0], size[1], size[2]});
    have_grid = it != m->grids.end();
  }
  if (!have_grid) {
    set_error("roi_align_sparse: no grid of spatial size [%d,%d,%d]", size[0], size[1], size[2]);
    return D3D_ERR_STATE;
  }
  Grid &g = it->second;
  const int32_t *extent = nullptr;
  if (!crop) {  // NULL crop: the grid's own occupied extent, computed once on the device
    int rc = grid_extent(m, g, s);
    if (rc) return rc;
    extent = g.extent;
  }
  *lv = {g.tab, feats, extent, crop ? nullptr : g.dense, g.cap, crop ? crop[0] : 0, crop ? crop[1] : 0, crop ? crop[2] : 0,
         g.hext[0], g.hext[1], g.hext[2], spatial_scale};
  return D3D_OK;
}

int d3d_roi_align_rotated_3d_sparse_forward(d3d_meta *m, const int *size, const float *feats, int C,
                                            const int *crop, const float *rois, int K,
                                            float spatial_scale, int ph, int pw, int pz,
                                            int sampling_ratio, const int *roi_levels, int level, int layout,
                                            float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && size && C > 0 && K >= 0 && ph > 0 && pw > 0 && pz > 0, "roi_align_sparse: bad arguments");
  D3D_REQUIRE(!roi_levels || (level >= 0 && level < kRoiMaxLevels), "roi_align_sparse: level %d (< %d)", level, kRoiMaxLevels);
  RoiLevels lv = {};
  if (int rc = roi_level_of(m, size, feats, crop, spatial_scale, s, &lv.v[roi_levels ? level : 0])) return rc;
  if (K == 0) return D3D_OK;
  D3D_REQUIRE(feats && rois && out, "roi_align_sparse: null pointer");
  D3D_REQUIRE(layout == 0 || layout == 1, "roi_align_sparse: layout must be 0 ([K,C,ph,pw,pz]) or 1 ([K,ph,pw,C,pz])");
  hipLaunchKernelGGL(k_roi_sparse, dim3(K, (C + kRoiCch - 1) / kRoiCch), dim3(kRoiWaves * 64), 0, s, lv, C, rois,
                     roi_levels, ph, pw, pz, sampling_ratio, layout, out);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_roi_align_rotated_3d_sparse_forward_levels(d3d_meta *m, int n_levels, const int *sizes_host,
                                                   const float *const *feats_host, int C, const float *scales_host,
                                                   const float *rois, int K, int ph, int pw, int pz, int sampling_ratio,
                                                   const int *roi_levels, int layout, float *out, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && sizes_host && feats_host && scales_host && n_levels >= 1 && n_levels <= kRoiMaxLevels,
              "roi_align_sparse_levels: 1..%d levels", kRoiMaxLevels);
  D3D_REQUIRE(C > 0 && K >= 0 && ph > 0 && pw > 0 && pz > 0 && (roi_levels || n_levels == 1),
              "roi_align_sparse_levels: bad arguments");
  D3D_REQUIRE(layout == 0 || layout == 1, "roi_align_sparse: layout must be 0 ([K,C,ph,pw,pz]) or 1 ([K,ph,pw,C,pz])");
  RoiLevels lv = {};
  for (int l = 0; l < n_levels; l++) {
    D3D_REQUIRE(feats_host[l], "roi_align_sparse_levels: null feature pointer of level %d", l);
    if (int rc = roi_level_of(m, sizes_host + 3 * l, feats_host[l], nullptr, scales_host[l], s, &lv.v[l])) return rc;
  }
  if (K == 0) return D3D_OK;
  D3D_REQUIRE(rois && out, "roi_align_sparse_levels: null pointer");
  hipLaunchKernelGGL(k_roi_sparse, dim3(K, (C + kRoiCch - 1) / kRoiCch), dim3(kRoiWaves * 64), 0, s, lv, C, rois,
                     roi_levels, ph, pw, pz, sampling_ratio, layout, out);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

int d3d_roi_align_rotated_3d_backward(const float *top_diff, int B, int C, int H, int W, int Z,
                                      const float *rois, int K, float spatial_scale, int ph, int pw, int pz,
                                      int sampling_ratio, float *bottom_diff, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Z > 0 && K >= 0 && ph > 0 && pw > 0 && pz > 0 && bottom_diff,
              "roi_align_backward: bad arguments");
  D3D_HIP_CHECK(hipMemsetAsync(bottom_diff, 0, sizeof(float) * (size_t)B * C * H * W * Z, s));
  if (K == 0) return D3D_OK;
  D3D_REQUIRE(top_diff && rois, "roi_align_backward: null pointer");
  long nthreads = (long)K * C * ph * pw * pz;
  hipLaunchKernelGGL(k_roi_dense_bwd, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, top_diff, C, H, W, Z,
                     rois, nthreads, spatial_scale, ph, pw, pz, sampling_ratio, bottom_diff);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

// d_feats [n_active, C] is accumulated into (zero it first)
int d3d_roi_align_rotated_3d_sparse_backward(d3d_meta *m, const int *size, const float *top_diff, int C,
                                             const int *crop, const float *rois, int K, float spatial_scale,
                                             int ph, int pw, int pz, int sampling_ratio, float *d_feats,
                                             void *stream) {
  hipStream_t s = (hipStream_t)stream;
  D3D_REQUIRE(m && size && crop && C > 0 && K >= 0 && ph > 0 && pw > 0 && pz > 0, "roi_align_sparse_backward: bad arguments");
  std::map<Size3, Grid>::iterator it;
  bool have_grid;
  {
    D3D_LOCK(m);
    it = m->grids.find(Size3{size[0], size[1], size[2]});
    have_grid = it != m->grids.end();
  }
  if (!have_grid) {
    set_error("roi_align_sparse_backward: no grid of spatial size [%d,%d,%d]", size[0], size[1], size[2]);
    return D3D_ERR_STATE;
  }
  if (K == 0) return D3D_OK;
  D3D_REQUIRE(top_diff && rois && d_feats, "roi_align_sparse_backward: null pointer");
  const Grid &g = it->second;
  const int NB = ph * pw * pz;
  size_t lds = (size_t)kRoiBwdCch * (NB + 1) * sizeof(float);
  D3D_REQUIRE(lds <= 64 * 1024, "roi_align_sparse_backward: pooled volume %d too large", NB);
  hipLaunchKernelGGL(k_roi_sparse_bwd, dim3(K, (C + kRoiBwdCch - 1) / kRoiBwdCch), dim3(256), lds, s, g.tab, g.cap, C,
                     crop[0], crop[1], crop[2], rois, spatial_scale, ph, pw, pz, sampling_ratio, top_diff, d_feats);
  D3D_LAUNCH_CHECK();
  return D3D_OK;
}

}  // extern "C"
