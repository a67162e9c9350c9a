/*
 * d3d_hip.h -- C ABI of libd3d_hip.so, the MI355X (gfx950) implementation of the hot path of
 * zhupan007/Detection_3D:  points -> voxel hash-scatter -> sparse-3D-conv rulebooks +
 * gather-GEMM-scatter -> rotated IoU / NMS -> rotated 3-D RoIAlign.
 *
 * Conventions
 *  - every pointer argument is a DEVICE pointer unless its name ends in _host;
 *  - every function takes the hipStream_t to enqueue on (passed as void* so that this header
 *    needs no HIP include) and returns 0 on success or a negative d3d_status; it never throws.
 *    d3d_last_error() returns a thread-local message for the last failure;
 *  - functions that must tell the host a size (number of active sites ...) synchronise the
 *    stream once; all others are asynchronous;
 *  - feature matrices are row-major fp32 [rows, planes]; coordinates int64 [n, 3|4] (x,y,z[,b]).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * repository root; SCN = SparseConvNet/sparseconvnet/SCN).
 */
#ifndef D3D_HIP_H
#define D3D_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  D3D_OK = 0,
  D3D_ERR_ARG = -1,      /* bad argument (shape, null pointer, unsupported mode) */
  D3D_ERR_HIP = -2,      /* a HIP runtime call failed */
  D3D_ERR_NOMEM = -3,    /* metadata arena exhausted */
  D3D_ERR_STATE = -4,    /* grid / rulebook not built (call order) */
  D3D_ERR_UNSUPPORTED = -5
} d3d_status;

const char *d3d_last_error(void);
int d3d_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Metadata: replaces class sparseconvnet.SCN.Metadata_3 (SCN/pybind.cpp:12-32,
 * SCN/Metadata/Metadata.h:36-130).  Owns the per-scale hash grids and the lazily built,
 * cached rulebooks ("plans"), all resident in one HBM arena of `arena_bytes`.
 * Not re-entrant per object (same as the reference: lazy cache mutation).                  */
typedef struct d3d_meta d3d_meta;
int d3d_meta_create(d3d_meta **out, size_t arena_bytes);
int d3d_meta_destroy(d3d_meta *m);
int d3d_meta_clear(d3d_meta *m);                       /* Metadata::clear, Metadata.cpp:30-45 */
int d3d_meta_arena_used(d3d_meta *m, size_t *bytes_host);
/* Two-lane building (no reference counterpart; the reference builds rulebooks on the CPU inside each layer's forward,
 * Metadata.cpp:430-510).  enable != 0: new grids + strided rulebooks (d3d_conv_prepare) may only be BUILT on `stream`
 * from now on (requested on another stream they fail with D3D_ERR_STATE instead of racing), and everything any other
 * stream builds or stages (submanifold / deconvolution rulebooks, partial tiles) comes from a separate third of the
 * slab -- so the caller may run the grid chain of the coarser levels, with its count read-backs, on `stream` while the
 * convolutions of the finer levels run on another one.  The caller orders the two streams with events.  Reset by
 * d3d_meta_clear.                                                                                                    */
int d3d_meta_set_geometry_stream(d3d_meta *m, void *stream, int enable);
/* Measurement / A-B switch: the geometry thread builds the strided grids of a d3d_geometry_async_start chain WITHOUT a
 * read-back between the levels (1, default; grid.hip run_grid_chain) or one d3d_conv_prepare call -- and read-back -- per
 * level (0).  Same results.  -> the previous setting.  Also settable through the environment: D3D_GRID_CHAIN=0.        */
int d3d_grid_chain_enable(int on);
/* Measurement / A-B switch of the fp32 sparse convolutions: 0 (default) = every launch through k_conv; 1 = the large
 * launches of the 64 -> 64 and 128 -> 128 layers through the weight-sharing kernel (conv_ws.hip, measured slower); 2 =
 * those layers' launches of every size (tests).  Same results bit for bit.  mode < 0: query only.  -> the previous setting.
 * Environment: D3D_CONV_WS.                                                                                          */
int d3d_conv_ws_mode(int mode);
/* ... and whether k_conv requests the next step's gathered rows behind the step's first MFMAs (1, default) or ahead of its
 * matrix work (0).  Same results.  on < 0: query only.  -> previous setting.  Environment: D3D_CONV_LATE.              */
int d3d_conv_late_mode(int on);
/* ... and how many of the leading levels form a chain (and read-back) of their own before the chain over the rest (default
 * 1: the first strided grid is wanted long before a chain over all levels ends; 0: one chain).  -> previous setting.   */
int d3d_grid_chain_head(int levels);
/* The chain of strided grids run by a thread of the library (no reference counterpart; every new grid costs one blocking
 * read-back of its site count, during which a caller that builds the chain itself cannot enqueue feature kernels).
 * start: `specs` = n x 13 ints (kind, in_size[3], out_size[3], filter[3], stride[3]), built in order:
 *   kind 1: d3d_conv_prepare (a new grid + strided rulebook) on `stream`, which must be the metadata's geometry stream;
 *   kind 0: d3d_subm_prepare(in_size, filter), kind 2: d3d_deconv_prepare -- views of grids that exist by then, enqueued
 *           on `view_stream` (the metadata's plan stream, d3d_meta_set_plan_stream) behind the newest grid.
 * The caller has already ordered `stream` after whatever built the first input grid and must not build on either stream
 * (or touch their lanes of the arena) until finish.
 * wait: blocks until entry `index` is built (enqueued, for views), returns its output site count (kind 1) and makes
 *       `wait_stream` wait for it.
 * finish: joins the thread (d3d_meta_clear / _destroy do so too) and reports its error, if any.                       */
int d3d_geometry_async_start(d3d_meta *m, const int *specs_host, int n, void *stream, void *view_stream);
int d3d_geometry_async_wait(d3d_meta *m, int index, int *n_out_host, void *wait_stream);
int d3d_geometry_async_finish(d3d_meta *m);
/* Third lane (needs a geometry stream): rulebooks that are views of an existing grid -- submanifold, deconvolution --
 * built on `stream` take their memory from a part of the slab of their own (the upper 3/4 of the feature lane, carved
 * once per scene), so that they can be built while the geometry stream works on the next grid and a third stream
 * convolves.  enable == 0 ends the routing; the rulebooks stay valid until d3d_meta_clear.                            */
int d3d_meta_set_plan_stream(d3d_meta *m, void *stream, int enable);

/* The stable radix sort behind the rulebooks' row order and the input layer's point lists (no reference counterpart: the
 * reference keeps rulebooks in insertion order on the CPU, RuleBookIterator.h:15-32), exported so that it can be tested
 * on its own.  Sorts (key, value) pairs by the low `bits` bits of the key, ascending or descending, stably; keys_out
 * may be null.  `scratch`: d3d_sort_scratch_bytes(n, bits) bytes of device memory.                                     */
size_t d3d_sort_scratch_bytes(int n, int bits);
int d3d_sort_pairs(const uint32_t *keys, const int32_t *vals, int n, int bits, int descending, uint32_t *keys_out,
                   int32_t *vals_out, void *scratch, size_t scratch_bytes, void *stream);

/* a1. data3d/suncg_utils/suncg_dataset.py:97-177: a = xyz*scale in fp64, shift by per-axis min,
 * drop points outside [0, full_scale), trunc -> int64; feats[:,0:3] = a/scale.
 * pcl fp32 [n, nfeat] (xyz first).  coords_out int64 [n,3], feats_out fp32 [n,nfeat]
 * (compacted, input order preserved).  Synchronises; *n_kept_host = rows written.            */
int d3d_voxelize(const float *pcl, int n, int nfeat, double scale, const int *full_scale_host,
                 int64_t *coords_out, float *feats_out, int *n_kept_host, void *scratch,
                 size_t scratch_bytes, void *stream);
size_t d3d_voxelize_scratch_bytes(int n);

/* a2/a3. InputLayer_updateOutput (SCN/sparseconvnet.h:159-163; SCN/Metadata/IOLayersRules.h:19-125;
 * SCN/CPU/IOLayers.cpp:11-47), split into the hash build (sizes) and the feature pass.
 * mode 3 = sum, 4 = mean.  Site ids follow first occurrence in input order (bit-exact with
 * the reference); duplicates are accumulated in input order.                                */
int d3d_input_layer_build(d3d_meta *m, const int64_t *coords, int n, int ncols,
                          const int *spatial_size_host, int batch_size, int mode, void *stream,
                          int *n_active_host);
/* The same, and while the host waits for the site count the neighbour table of the submanifold rulebook
 * (spatial_size, prefetch_filter) -- the one d3d_subm_prepare is asked for first -- is already being probed on
 * `stream`, sized by n and reading the count on the device; d3d_subm_prepare on the same stream picks it up.
 * prefetch_filter NULL: plain d3d_input_layer_build.  (The reference builds that rulebook inside the first
 * convolution's forward, Metadata.cpp:430-443.)                                                              */
int d3d_input_layer_build_prefetch(d3d_meta *m, const int64_t *coords, int n, int ncols,
                                   const int *spatial_size_host, int batch_size, int mode,
                                   const int *prefetch_filter_host, void *stream, int *n_active_host);
/* The per-site point lists d3d_input_layer_forward / _backward read (IOLayersRules.h:19-125 builds them with the
 * grid) are built by the first call that needs them, on ITS stream; this builds them explicitly, e.g. on a side stream
 * while the caller's stream already probes the level-0 rulebook.                                               */
int d3d_input_layer_prepare(d3d_meta *m, void *stream);
int d3d_input_layer_forward(d3d_meta *m, const float *feats, int planes, float *out, void *stream);
/* debug exporter: rule table rows [count, idx0..] (IOLayersRules.h:112-124) as CSR.
 * offsets int32 [n_active+1], idx int32 [n].                                                 */
int d3d_input_layer_export(d3d_meta *m, int32_t *offsets, int32_t *idx, void *stream);

/* Metadata::getNActive / getSpatialLocations (SCN/Metadata/Metadata.cpp:148-168): int64 [n,4]. */
int d3d_get_n_active(d3d_meta *m, const int *spatial_size_host, int *n_host);
int d3d_get_spatial_locations(d3d_meta *m, const int *spatial_size_host, int64_t *out, void *stream);

/* AnchorGenerator.forward for one feature map (modeling/rpn/anchor_generator_sparse3d.py:86-120): for active site i
 * (in getSpatialLocations order) and cell anchor a < A (<= 16): out[(i*A + a)] = (loc_i / voxel_scale * stride, 0,0,0,0)
 * + base[a], fp32 with the reference's operation order.  base_host [A,7], stride_host [3]; out [n_active*A, 7]. */
int d3d_anchors(d3d_meta *m, const int *spatial_size_host, const float *base_host, int A,
                const float *stride_host, float voxel_scale, float *out, void *stream);
/* ... for all selected maps in one launch (the concatenation of AnchorGenerator.forward's list, rpn_sparse3d.py:246):
 * sizes_host [n_maps*3], bases_host [n_maps, A, 7], strides_host [n_maps*3]; n_maps <= 6, A <= 4;
 * out [sum_m n_active_m * A, 7], map after map.  Bit-identical to d3d_anchors per map.                              */
int d3d_anchors_maps(d3d_meta *m, int n_maps, const int *sizes_host, const float *bases_host, int A,
                     const float *strides_host, float voxel_scale, float *out, void *stream);

/* a12. RPN head at inference, SingleConvRPNHead_Sparse3D.forward + cat_scales_obj_reg
 * (modeling/rpn/rpn_sparse3d.py:80-131 and :19-77) for all selected maps in ONE launch:
 *   t = relu(x W1^T + b1);  objectness = t Wc^T + bc  [n, a];  regression = t Wr^T + br  [n, 7a]
 * over the n = sum(rows_host) site rows of the maps laid end to end (scale, site, anchor order).
 * maps_host[m]: device pointer of map m's features [rows_host[m], channels] fp32 (channels 128 or 256);
 * w1_packed [channels/4, channels, 4] with w1_packed[g][co][j] = W1[co][4g+j] (W1 = conv.weight [channels, channels]);
 * w2_packed [channels/4, NOUT, 4] likewise for the rows of (Wc; Wr) = [8a, channels] zero-padded to
 * NOUT = 32*ceil(8a/32) columns; b1 [channels], b2 [8a] = (bc; br).  a = anchors per site x class groups.          */
#define D3D_RPN_MAX_MAPS 6
int d3d_rpn_head(const float *const *maps_host, const int *rows_host, int n_maps, int channels,
                 const float *w1_packed, const float *b1, const float *w2_packed, const float *b2, int a,
                 float *objectness, float *regression, void *stream);
/* a22. The box head behind fc6 at inference (roi_box_feature_extractors.py:110-117, roi_box_predictors.py:33-55) with the
 * same kernel:  t = relu(relu?(x) W1^T + b1)  [rows, channels];  out_a = t Wa^T + ba  [rows, a];  out_7a = t Wb^T + bb
 * [rows, 7a]  (a = classes: cls_score and bbox_pred).  x [rows, channels] fp32, channels 128, 256 or 512; relu_in != 0
 * rectifies x as it is read (fc6's output).  w1_packed / w2_packed / b2 as for d3d_rpn_head; either stage may be left out:
 * w1_packed == NULL -> the rows are t already (the predictor alone), w2_packed == NULL -> only t_out is written (fc7
 * alone).  t_out (may be NULL when stage 2 runs) receives t.  One launch for both stages gives the bits of the two
 * stages launched one after the other.                                                                              */
int d3d_mlp_heads(const float *x, int rows, int channels, int relu_in, const float *w1_packed, const float *b1,
                  float *t_out, const float *w2_packed, const float *b2, int a, float *out_a, float *out_7a, void *stream);

/* a4. Metadata::getSubmanifoldRuleBook (Metadata.cpp:430-443; SubmanifoldConvolutionRules.h:27-45).
 * Builds (or finds cached) the rulebook; *n_rules_host = number of (in,out) pairs.           */
int d3d_subm_prepare(d3d_meta *m, const int *spatial_size_host, const int *filter_host,
                     void *stream, long *n_rules_host);
/* a5. Metadata::getRuleBook (Metadata.cpp:485-510; ConvolutionRules.h:12-34): builds the output
 * grid of spatial size `out_size` and the rulebook.  Output sites are numbered by first touch
 * while visiting input sites in id order (canonical; the reference's order is hash-iteration
 * dependent).                                                                                */
int d3d_conv_prepare(d3d_meta *m, const int *in_size_host, const int *out_size_host,
                     const int *filter_host, const int *stride_host, void *stream,
                     int *n_out_host, long *n_rules_host);
/* Deconvolution view of the strided rulebook of (fine size `out_size`, filter, stride): rows are the
 * fine sites (SCN/CPU/Deconvolution.cpp:17).  Requires the matching d3d_conv_prepare.          */
int d3d_deconv_prepare(d3d_meta *m, const int *in_size_host, const int *out_size_host,
                       const int *filter_host, const int *stride_host, void *stream,
                       long *n_rules_host);
/* debug exporter: rulebook as (in,out,offset) int32 triples in unspecified order; capacity in
 * triples; *n_host receives the count.  kind: 0 submanifold (out_size ignored), 1 strided.    */
int d3d_export_rules(d3d_meta *m, int kind, const int *in_size_host, const int *filter_host,
                     const int *stride_host, int32_t *triples, long capacity, long *n_host,
                     void *stream);

/* debug / measurement: grouping quality of a built rulebook.  *executed_host = 32 x the sum over the 32-row blocks of the
 * number of filter offsets the block runs, *rules_host = (in, out) pairs; executed / rules = 1 when every block's rows
 * share one offset mask.  kind as d3d_export_rules (2 = deconvolution view).  Synchronises.                          */
int d3d_plan_stats(d3d_meta *m, int kind, const int *in_size_host, const int *filter_host, const int *stride_host,
                   long *n_blocks_host, long *executed_host, long *rules_host, void *stream);

/* Measurement hook (bench.py's roofline leg; no reference counterpart): the next sparse-convolution launch made by
 * the calling thread records the two HIP events (hipEvent_t, created with timing) on its stream immediately before
 * and after the k_conv kernel itself -- not the k_conv_reduce of an offset-split launch -- so that the live average
 * is the duration `rocprofv3 --kernel-trace --stats` reports for that kernel name.  NULLs disarm.                */
int d3d_conv_time_next(void *start_event, void *stop_event);
/* Conv weights: reference layout [filter_volume, groups=1, Cin, Cout]
 * (sparseconvnet/submanifoldConvolution.py:24-26).  The kernels read a k-interleaved copy
 * [fv][ceil(Cin/8)*2][Cout][4]; pack once per weight update.                                 */
size_t d3d_packed_weight_floats(int filter_volume, int cin, int cout);
int d3d_pack_conv_weight(const float *w, int filter_volume, int cin, int cout, float *packed,
                         void *stream);

/* a6. SubmanifoldConvolution_updateOutput (SCN/sparseconvnet.h:99-105), Convolution_updateOutput
 * (:85-91), Deconvolution_updateOutput (:147-152): out[r_out] = sum_k in[r_in(k)] @ W[k],
 * bias-free (fpn_net.py builds every conv with bias=False).  `residual` (may be null) is added
 * in the epilogue (fuses sparseconvnet/tables.py AddTable).  *macs_host (may be null)
 * receives rules*Cin*Cout like the reference's return value.                                 */
/* Optional fusion of the PRODUCER's BatchNormalization (+ leaky ReLU) into the gather of a convolution:
 * the conv reads the un-normalised rows and applies y = leaky(x * (invstd*weight) + (bias - mean*invstd*weight))
 * on the fly (same arithmetic as d3d_bn_forward), which removes one full read+write of the tensor.  Host
 * struct with device pointers; pass NULL (or mean == NULL) for none.                                 */
typedef struct {
  const float *mean, *invstd, *weight, *bias; /* [Cin]; weight / bias may be NULL */
  float leakiness;
  /* ... and of the statistics pass of the CONSUMER's BatchNormalization into the epilogue (fp32 storage): when
   * out_stats != NULL the convolution also leaves, per 32-row block (per workgroup of the reduction for an offset-split
   * launch), the fp64 sums and sums of squares of the output columns it stored: out_stats[row][2 * Cout], at most
   * out_stats_cap rows (ceil(n_out / 32) * max(1, Cout / 32) always suffice); *out_stats_rows (host) receives the
   * number of rows written, 0 when nothing was (bf16 storage, capacity too small).  d3d_bn_stats_from_partials turns
   * them into the BatchNorm's statistics in a fixed summation order.  No reference counterpart: SCN/CPU/BatchNormalization.cpp:20-31
   * makes its own pass over the tensor.                                                                             */
  double *out_stats;
  int out_stats_cap;
  int *out_stats_rows;
} d3d_bn_prologue;
int d3d_subm_conv_forward(d3d_meta *m, const int *spatial_size_host, const int *filter_host,
                          const float *in, int cin, const float *packed_w, int cout,
                          const float *residual, float *out, void *stream, double *macs_host,
                          const d3d_bn_prologue *bn_host);
int d3d_conv_forward(d3d_meta *m, const int *in_size_host, const int *out_size_host,
                     const int *filter_host, const int *stride_host, const float *in, int cin,
                     const float *packed_w, int cout, float *out, void *stream, double *macs_host,
                     const d3d_bn_prologue *bn_host);
int d3d_deconv_forward(d3d_meta *m, const int *in_size_host, const int *out_size_host,
                       const int *filter_host, const int *stride_host, const float *in, int cin,
                       const float *packed_w, int cout, const float *residual, float *out,
                       void *stream, double *macs_host, const d3d_bn_prologue *bn_host);

/* ---- bf16 storage (BASELINE.json configs[4]; the reference dispatches on the tensor type in
 * SCN/CUDA/Convolution.cu:444-521 and sparseconvnet_cuda.cpp).  Feature rows, packed weights and outputs are bf16
 * (raw 16-bit words), accumulation is fp32 on v_mfma_f32_32x32x16_bf16, BatchNorm statistics stay fp64/fp32.
 * For D3D_BF16 `cin` is the STORED row width: 16, 32, 64, 128 or 256 channels (narrower inputs are zero padded by
 * the caller; d3d_pack_conv_weight_dt pads the weights to match).  D3D_F32 forwards to the fp32 entry points.      */
typedef enum { D3D_F32 = 0, D3D_BF16 = 1 } d3d_dtype;
size_t d3d_packed_weight_bytes(int filter_volume, int cin, int cout, int dtype);
int d3d_pack_conv_weight_dt(const float *w, int filter_volume, int cin, int cout, void *packed, int dtype,
                            void *stream);
int d3d_subm_conv_forward_dt(d3d_meta *m, const int *spatial_size_host, const int *filter_host, const void *in,
                             int cin, const void *packed_w, int cout, const void *residual, void *out, int dtype,
                             void *stream, double *macs_host, const d3d_bn_prologue *bn_host);
int d3d_conv_forward_dt(d3d_meta *m, const int *in_size_host, const int *out_size_host, const int *filter_host,
                        const int *stride_host, const void *in, int cin, const void *packed_w, int cout, void *out,
                        int dtype, void *stream, double *macs_host, const d3d_bn_prologue *bn_host);
int d3d_deconv_forward_dt(d3d_meta *m, const int *in_size_host, const int *out_size_host, const int *filter_host,
                          const int *stride_host, const void *in, int cin, const void *packed_w, int cout,
                          const void *residual, void *out, int dtype, void *stream, double *macs_host,
                          const d3d_bn_prologue *bn_host);
/* Tuning / test hook of the bf16 convolution (process-wide): `row_blocks` (1, 2 or 4) consecutive 32-row blocks share
 * every weight fetch in launches that keep at least min_waves * row_blocks waves (min_waves < 0: the default).
 * Results do not depend on it beyond the summation grouping of offset-split launches.                           */
int d3d_conv_bf16_tuning(int row_blocks, long min_waves);
/* d3d_bn_batch_stats (want_invstd == 0: mean, unbiased var) / d3d_bn_batch_invstd (want_invstd != 0: mean,
 * powf(var + eps, -0.5)) from the column sums a convolution left per row block (d3d_bn_prologue.out_stats: partial_rows
 * rows of [2 * planes] fp64) instead of from the tensor; `rows` = rows of the tensor.  Same arithmetic after the sums,
 * a fixed summation order (deterministic), one launch.                                                              */
int d3d_bn_stats_from_partials(const double *partials, int partial_rows, int rows, int planes, float eps,
                               int want_invstd, float *mean, float *var_or_invstd, void *scratch, size_t scratch_bytes,
                               void *stream);
/* d3d_bn_batch_invstd / d3d_bn_apply on a tensor of the given storage type (statistics and parameters fp32). */
int d3d_bn_batch_invstd_dt(const void *in, int rows, int planes, float eps, float *mean, float *invstd,
                           void *scratch, size_t scratch_bytes, int dtype, void *stream);
int d3d_bn_apply_dt(const void *in, void *out, int rows, int planes, const float *mean, const float *invstd,
                    const float *weight, const float *bias, float leakiness, int dtype, void *stream);
/* fp32 rows [rows, cin] -> bf16 rows [rows, width], width >= cin a multiple of 8, the channels past cin zero, round to
 * nearest even: the input layer's output as the bf16 backbone stores it (fpn_net.py:150 hands the fp32 means to the first
 * convolution; the bf16 storage of BASELINE configs[4] pads the 9 input channels to 16).  One launch instead of a pad and a
 * cast.                                                                                                             */
int d3d_rows_to_bf16(const float *in, long rows, int cin, int width, void *out, void *stream);

/* a7. Backward (training).  SubmanifoldConvolution_backward / Convolution_backward / Deconvolution_backward
 * (SCN/sparseconvnet.h:92-98,106-111,153-158; SCN/CUDA/Convolution.cu:249-442): d_in is overwritten,
 * d_weight [fv, Cin, Cout] is accumulated into (the caller pre-zeroes it, as the reference's Python does).
 * `packed_wt*` is the transposed packing of d3d_pack_conv_weight_transposed (flip=1 for submanifold).
 * Either gradient pointer may be null to skip it.  dWeight uses fp32 atomics (like the reference).     */
int d3d_pack_conv_weight_transposed(const float *w, int filter_volume, int cin, int cout, int flip,
                                    float *packed, void *stream);   /* size d3d_packed_weight_floats(fv, cout, cin) */
int d3d_subm_conv_backward(d3d_meta *m, const int *spatial_size_host, const int *filter_host,
                           const float *in, int cin, const float *packed_wt_flipped, int cout,
                           const float *d_out, float *d_in, float *d_weight, void *stream);
int d3d_conv_backward(d3d_meta *m, const int *in_size_host, const int *out_size_host,
                      const int *filter_host, const int *stride_host, const float *in, int cin,
                      const float *packed_wt, int cout, const float *d_out, float *d_in,
                      float *d_weight, void *stream);
int d3d_deconv_backward(d3d_meta *m, const int *in_size_host, const int *out_size_host,
                        const int *filter_host, const int *stride_host, const float *in, int cin,
                        const float *packed_wt, int cout, const float *d_out, float *d_in,
                        float *d_weight, void *stream);
/* Process-wide switch of the dWeight accumulation of the three backward entry points above: 0 (default) = fp32 atomics
 * like the reference (SCN/CUDA/Convolution.cu:249-442 accumulates with atomicAdd: the last bits differ from run to run),
 * 1 = a fixed summation order (every workgroup sums its row blocks in order into a partial dWeight held in the
 * metadata's feature lane -- at most 32 x [fv, Cin, Cout] floats --, the partials are added in order): the same bits in
 * every run; the backward pass of a 6c training step at 500 k points takes 15.7 instead of 10.1 ms.  on < 0: query.  -> previous setting.  Environment: D3D_DW_DETERMINISTIC=1.       */
int d3d_conv_dw_deterministic(int on);
/* BatchNormalization_backward (SCN/sparseconvnet.h:27-32; SCN/CPU/BatchNormalization.cpp:62-107). */
int d3d_bn_backward(const float *in, const float *out, const float *d_out, float *d_in, int rows,
                    int planes, const float *save_mean, const float *save_invstd, const float *weight,
                    float *d_weight, float *d_bias, float leakiness, void *scratch,
                    size_t scratch_bytes, void *stream);
size_t d3d_bn_backward_scratch_bytes(int planes);
/* InputLayer_updateGradInput (SCN/sparseconvnet.h:164-167; SCN/CPU/IOLayers.cpp:30-47). */
int d3d_input_layer_backward(d3d_meta *m, const float *d_out, int planes, float *d_in, void *stream);
/* SparseToDense_updateGradInput (SCN/sparseconvnet.h:218-222): d_in [n_active, planes] gathered from the
 * dense gradient [batch, planes, X, Y, Z].                                                          */
int d3d_sparse_to_dense_backward(d3d_meta *m, const int *spatial_size_host, const float *d_out, int planes,
                                 float *d_in, void *stream);
/* _C.roi_align_rotated_3d_backward (maskrcnn_benchmark/csrc/ROIAlignRotated3D.h:28-47): bottom_diff
 * [B,C,H,W,Z] is zeroed and accumulated into with fp32 atomics.                                      */
int d3d_roi_align_rotated_3d_backward(const float *top_diff, int B, int C, int H, int W, int Z,
                                      const float *rois, int K, float spatial_scale, int ph, int pw,
                                      int pz, int sampling_ratio, float *bottom_diff, void *stream);
/* roi_align_rotated_3d_backward restricted to the active sites (csrc/cuda/ROIAlignRotated3D_cuda.cu:238-354
 * followed by SparseToDense_updateGradInput); d_feats [n_active, C] is accumulated into.                */
int d3d_roi_align_rotated_3d_sparse_backward(d3d_meta *m, const int *spatial_size_host,
                                             const float *top_diff, int C, const int *crop_host,
                                             const float *rois, int K, float spatial_scale, int ph,
                                             int pw, int pz, int sampling_ratio, float *d_feats,
                                             void *stream);

/* a8. BatchNormalization_updateOutput (SCN/sparseconvnet.h:21-26; SCN/CPU/BatchNormalization.cpp:12-60).
 * train!=0: batch statistics, running update r = m*r + (1-m)*batch.  train==0: uses
 * running_mean / running_var.  d3d_bn_batch_stats computes mean(0) and the UNBIASED var(0)
 * that sparseconvnet/batchNormalization.py:51-56 feeds in when track_running_stats=False.    */
int d3d_bn_forward(const float *in, float *out, int rows, int planes, float *save_mean,
                   float *save_invstd, float *running_mean, float *running_var,
                   const float *weight, const float *bias, float eps, float momentum, int train,
                   float leakiness, void *scratch, size_t scratch_bytes, void *stream);
int d3d_bn_batch_stats(const float *in, int rows, int planes, float *mean, float *var_unbiased,
                       void *scratch, size_t scratch_bytes, void *stream);
/* mean(0) and invstd = powf(var_unbiased(0) + eps, -0.5): what the eval path with track_running_stats=False
 * feeds the normalisation with; use with d3d_bn_prologue.                                            */
int d3d_bn_batch_invstd(const float *in, int rows, int planes, float eps, float *mean, float *invstd,
                        void *scratch, size_t scratch_bytes, void *stream);
/* the normalisation alone, y = leaky(x * (invstd*gamma) + (beta - mean*invstd*gamma)) (BatchNormalization.cpp:46-59),
 * for a consumer of a deferred BatchNorm that is not a convolution.                                   */
int d3d_bn_apply(const float *in, float *out, int rows, int planes, const float *mean, const float *invstd,
                 const float *weight, const float *bias, float leakiness, void *stream);
/* scratch of d3d_bn_forward / d3d_bn_batch_stats / d3d_bn_batch_invstd: device memory of this many bytes whose
 * first 256 bytes (ticket counters) are ZERO before the first call; the library leaves them zero.  planes must
 * be a multiple of 4 with planes/4 dividing 1024 (planes = 4, 8, ..., 4096).                                     */
size_t d3d_bn_scratch_bytes(int planes);
/* sparseconvnet/utils.py:61-66 add_feature_planes / tables.py AddTable: out = a + b. */
int d3d_add(const float *a, const float *b, float *out, size_t n, void *stream);

/* a20. SparseToDense_updateOutput (SCN/sparseconvnet.h:214-217; SCN/CPU/SparseToDense.cpp:7-60):
 * zero-filled dense [batch, planes, X, Y, Z].                                                */
int d3d_sparse_to_dense_forward(d3d_meta *m, const int *spatial_size_host, const float *in,
                                int planes, int batch, float *out, void *stream);

/* Pooler pre-processing in one launch: convert_metric_to_pixel (roi_box_feature_extractors.py:100), BoxList3D
 * yx_zb -> standard + convert_to_roi_format (modeling/poolers_3d.py:107-124, structures/bounding_box_3d.py:221-242)
 * and LevelMapper (poolers_3d.py:19-54 in its sqrt(max size)/canonical form).  boxes_metric [n,7] yx_zb in metres ->
 * rois [n,8] (example index, x, y, z centre, x, y, z size in full-resolution pixels, yaw in degrees) and, for
 * n_levels > 1, levels[n] = argmin_l |scales[l] - sqrt(max(size_y, size_x)) / canonical_size| (first minimum).
 * batch_ids [n] (device, may be NULL = example 0): the example each box belongs to (poolers_3d.py:112-118).      */
int d3d_roi_prepare(const float *boxes_metric, int n, float voxel_scale, const float *scales_host, int n_levels,
                    float canonical_size, const int32_t *batch_ids, float *rois, int32_t *levels, void *stream);
/* ... for a proposal list padded to n rows whose real length *count_dev (<= n) is still on the device (the survivor
 * count of the RPN's NMS, rpn/inference_3d.py:127-131, not yet read back): rows >= *count_dev get zero RoIs and level
 * -1, so no level's RoIAlign launch pools them.  count_dev NULL: all n rows are real (= d3d_roi_prepare).          */
int d3d_roi_prepare_counted(const float *boxes_metric, int n, const int32_t *count_dev, float voxel_scale,
                            const float *scales_host, int n_levels, float canonical_size, const int32_t *batch_ids,
                            float *rois, int32_t *levels, void *stream);
/* a21. _C.roi_align_rotated_3d_forward (maskrcnn_benchmark/csrc/ROIAlignRotated3D.h:10-26;
 * csrc/cuda/ROIAlignRotated3D_cuda.cu:89-177).  Dense input [B,C,H,W,Z]; rois [K,8].          */
int d3d_roi_align_rotated_3d_forward(const float *input, int B, int C, int H, int W, int Z,
                                     const float *rois, int K, float spatial_scale, int ph, int pw,
                                     int pz, int sampling_ratio, float *out, void *stream);
/* Same result sampled straight from the sparse tensor through the hash grid (no 1 GB dense
 * map): equals sparse_3d_to_dense_2d (sparseconvnet/tools_3d_2d.py:7-48, crop to the occupied
 * extent crop_host[3]; NULL: the extent of the grid itself, found on the device without a host read-back)
 * followed by the dense op.
 * roi_levels (device int32[K], nullable): only RoIs with roi_levels[i] == level are pooled, the other output
 * slots are left untouched, so that the per-level calls of poolers_3d.py:150-168 fill one result tensor
 * without nonzero / index / index_put passes.
 * layout 0: out[slot][C][ph][pw][pz] (the reference's); layout 1: out[slot][ph][pw][C][pz], the row-major
 * operand of the box head's [1,1,pz] convolution (roi_box_feature_extractors.py:68-77) as a GEMM.   */
int d3d_roi_align_rotated_3d_sparse_forward(d3d_meta *m, const int *spatial_size_host,
                                            const float *feats, int C, const int *crop_host,
                                            const float *rois, int K, float spatial_scale, int ph,
                                            int pw, int pz, int sampling_ratio, const int *roi_levels,
                                            int level, int layout, float *out, void *stream);
/* ... all levels of the pooler (poolers_3d.py:150-168) in ONE launch: RoI i is pooled from the map of level
 * roi_levels[i] (0 <= level < n_levels <= 4; any other value, e.g. the -1 of d3d_roi_prepare_counted, leaves its slot
 * untouched).  sizes_host [n_levels*3] spatial sizes of the maps, feats_host[l] device pointer of map l's features
 * [n_active_l, C], scales_host[l] its spatial scale; the occupied extent of every map is found on the device.
 * roi_levels may be NULL only for n_levels == 1.                                                                      */
int d3d_roi_align_rotated_3d_sparse_forward_levels(d3d_meta *m, int n_levels, const int *sizes_host,
                                                   const float *const *feats_host, int C, const float *scales_host,
                                                   const float *rois, int K, int ph, int pw, int pz,
                                                   int sampling_ratio, const int *roi_levels, int layout, float *out,
                                                   void *stream);

/* a17. rotate_iou_gpu_eval (second/core/non_max_suppression/nms_gpu.py:614-664) incl.
 * check_same_boxes: boxes [N,5], query [K,5] -> out [N,K].                                   */
int d3d_rotate_iou_eval(const float *boxes, int N, const float *query, int K, int criterion,
                        float *out, void *stream);
/* a16. boxes_iou_3d (utils3d/rotate_nms_3d_torch.py:23-88): targets [M,7], anchors [N,7] yx_zb,
 * aug_host = {target_Y, target_Z, anchor_Y, anchor_Z}.                                       */
int d3d_boxes_iou_3d(const float *targets, int M, const float *anchors, int N,
                     const float *aug_host, int criterion, int only_xy, float *out, void *stream);
/* a15/a18. rotate_nms_3d_cc (second/core/non_max_suppression/nms_cpu.py:32-44) + spconv's
 * rotate_non_max_suppression_cpu: boxes [n,7] yx_zb ALREADY sorted by descending score
 * (the callers top-k first, box_torch_ops.py:495-499).  keep int32 [n] receives positions in
 * selection order; n_keep (device int32).  The sweep may stop once max_keep (> 0) survivors exist: the
 * callers truncate to post_max_size anyway (box_torch_ops.py:506).  scratch >= d3d_nms_scratch_bytes(n). */
int d3d_rotate_nms_3d_sorted(const float *boxes, int n, float thresh, int max_keep, int32_t *keep,
                             int32_t *n_keep, void *scratch, size_t scratch_bytes, void *stream);
size_t d3d_nms_scratch_bytes(int n);
/* The same sweep for `segments` independent candidate lists in one set of launches (the classes of
 * roi_heads/box_head_3d/inference.py:113-139, or one list with its top-k order): candidate i of segment b is box
 * order[b*stride + i] of boxes [*,7] (b*stride + i when order is NULL), i < counts[b] (device int32, NULL: n_max
 * each, <= 4096), in descending score order.  min_yx / min_z: the NMS_AUG_THICKNESS clamp of boxlist_nms_3d
 * (structures/boxlist_ops_3d.py:18-52) applied for the IoU only.  keep int32 [segments, n_max] receives the BOX
 * indices of the survivors in selection order, n_keep int32 [segments] their number (<= max_keep when > 0).   */
int d3d_rotate_nms_3d_batched(const float *boxes, const int32_t *order, int stride, const int32_t *counts,
                              int segments, int n_max, float thresh, float min_yx, float min_z,
                              int max_keep, int32_t *keep, int32_t *n_keep, void *scratch,
                              size_t scratch_bytes, void *stream);
/* a15. rotate_nms_3d AS THE REFERENCE DEFINES IT (second/pytorch/core/box_torch_ops.py:489-514, called by
 * boxlist_nms_3d, structures/boxlist_ops_3d.py:14-62, with its size clamp): boxes [n,7] yx_zb and scores [n] in any order.
 * Candidates = the pre_max_size best scores (<= 0: all; at most d3d_topk_max()); descending score, and among EQUAL scores
 * the lower index first -- torch.topk / numpy argsort (nms_cpu.py:37) leave that order open, here and in the oracle port
 * it is defined.  Sizes clamped for the IoU only (dy, dx >= aug_yx, dz >= aug_z); greedy rotated NMS at `thresh`; at
 * most post_max_size survivors (<= 0: all).  keep_out int64 [min(n, pre_max_size)] = indices into the input in selection
 * order (the LongTensor the reference returns); n_keep_dev device int32 [1]; n_keep_host (NULL: no read-back) receives
 * the count after one stream synchronisation.  scratch >= d3d_rotate_nms_3d_scratch_bytes(pre_max_size, 0); with
 * d3d_rotate_nms_3d_scratch_bytes(pre_max_size, n) bytes the selection also caches its keys there (faster for large n). */
int d3d_rotate_nms_3d(const float *boxes, const float *scores, int n, int pre_max_size, int post_max_size, float thresh,
                      float aug_yx, float aug_z, int64_t *keep_out, int32_t *n_keep_dev, int *n_keep_host, void *scratch,
                      size_t scratch_bytes, void *stream);
size_t d3d_rotate_nms_3d_scratch_bytes(int pre_max_size, int n);   /* n = input boxes (0: no key scratch, slower selection) */
/* a13. The selection of RPNPostProcessor.forward_for_single_feature_map (modeling/rpn/inference_3d.py:105-123), and of
 * the per-class candidate lists of the box head's post-processing (roi_heads/box_head_3d/inference.py:113-131), for
 * n_examples x n_groups segments in ONE launch.  Segment s = example * n_groups + group reads element i at
 * vals[group * group_stride + i * elem_stride] (apply_sigmoid: 1 / (1 + exp(-x)) first, inference_3d.py:105) over the
 * elements with example[i] == its example (example NULL when n_examples == 1), keeps the k best (k <= d3d_topk_max();
 * descending, equal scores: lower index first) and writes rows [s * k, s * k + min(k, elements)) of the outputs: element
 * indices (idx32_out = i * idx_map[0] + idx_map[1] + group * idx_map[2], idx_map_host NULL: i; idx64_out = i), scores
 * and -- when props_out is given -- the decoded boxes BoxCoder3D.decode(reg[i * reg_stride + 7 group .. + 7],
 * anchors[i]) with unit weights (box_coder_3d.py:38-65, inference_3d.py:123).  counts_out int32 [segments] =
 * min(k, elements of the segment), or with min_value_host the number of kept elements with value > *min_value_host
 * (the score threshold of inference.py:118; a prefix of the sorted list).  Outputs other than counts_out may be NULL. */
int d3d_topk_segments(const float *vals, int n, int elem_stride, int group_stride, int n_groups, const int32_t *example,
                      int n_examples, int k, int apply_sigmoid, const float *min_value_host, const int *idx_map_host,
                      const float *reg, int reg_stride, const float *anchors, float clip, int32_t *idx32_out,
                      int64_t *idx64_out, float *scores_out, float *props_out, int32_t *counts_out, void *scratch,
                      size_t scratch_bytes, void *stream);
/* scratch (may be NULL; 16-byte aligned, >= d3d_topk_scratch_bytes(n, segments)): the selection then evaluates every
 * element once and re-reads cached keys with wide loads in its later passes (several times faster for n ~ 10^5).      */
size_t d3d_topk_scratch_bytes(int n, int segments);
int d3d_topk_max(void);
size_t d3d_nms_batched_scratch_bytes(int segments, int n_max);
/* Box-head post-processing glue around the batched NMS (roi_heads/box_head_3d/inference.py:113-148), one launch each:
 * post_scores: sc[(nc-1), K] = prob[i][j+1] if > thresh else -1 (class-major), counts[nc-1] = candidates per class
 *              (`inds = scores[:, j] > score_thresh`, :118);
 * post_order:  order[j][i] = idx[j][i] * nc + j + 1, the box index of class j+1's i-th best RoI (idx = stable
 *              descending argsort of sc rows) in the [K, nc] box layout -- the segments d3d_rotate_nms_3d_batched takes;
 * post_gather: survivors class-major in selection order: flat[t] = keep[t], scores[t] = prob_flat[keep[t]] for
 *              t % n_max < n_keep[t / n_max], else (0, -1) (:140-148 then picks the top detections_per_img). */
int d3d_post_scores(const float *prob, int K, int nc, float thresh, float *sc, int32_t *counts, void *stream);
int d3d_post_order(const int64_t *idx, int K, int nc, int32_t *order, void *stream);
int d3d_post_gather(const int32_t *keep, const int32_t *n_keep, int segments, int n_max, const float *prob_flat,
                    float *scores, int64_t *flat, void *stream);
/* post_select: post_gather, the cut to `detections` per image and the final gathers of :140-148 in one launch.  Over the
 * candidates t (class-major, selection order) with score s[t] = prob_flat[keep[t]] (survivors) or -1 (padding): thresh =
 * the detections-th largest s when 0 < detections < segments * n_max, clamped to >= 0, else 0; every t with
 * s[t] >= thresh is kept IN t ORDER (ties at the threshold all stay, like `cls_scores >= image_thresh`, :146) and
 * out_boxes [n,7] = boxes[keep[t]], out_scores [n] = s[t], out_labels int64 [n] = keep[t] % nc, *out_n (device) = n.
 * Output capacity: segments * n_max rows.  segments * n_max <= d3d_post_select_max().                                
 * out_n (and any other count this header returns through a device pointer) may be pinned host memory: the kernel's
 * store is then the read-back, visible to the host behind an event recorded after the launch.                          */
int d3d_post_select_max(void);
int d3d_post_select(const int32_t *keep, const int32_t *n_keep, int segments, int n_max, const float *prob_flat,
                    const float *boxes, int nc, int detections, float *out_boxes, float *out_scores, int64_t *out_labels,
                    int32_t *out_n, void *stream);
/* a14. BoxCoder3D.decode (maskrcnn_benchmark/modeling/box_coder_3d.py:38-65). */
int d3d_box_decode(const float *enc, const float *anchors, int n, const float *weights_host,
                   float clip, float *out, void *stream);
/* ... of class-wise encodings (roi_heads/box_head_3d/inference.py:84-87 `box_coder.decode(box_regression.view(sum, -1),
 * concat_boxes)`): enc [n, 7 nc], anchors [n, 7] -> out [n, 7 nc], every class of row i decoded against anchor i. */
int d3d_box_decode_classes(const float *enc, const float *anchors, int n, int nc, const float *weights_host,
                           float clip, float *out, void *stream);
/* ... of the rows the RPN's top-k selected (rpn/inference_3d.py:109-123: `box_regression[topk_idx]`,
 * `concat_anchors[topk_idx]`, then decode): out[i] = decode(enc[rows[i]], anchors[rows[i]]), rows int64 [n] on the device. */
int d3d_box_decode_rows(const float *enc, const float *anchors, const int64_t *rows, int n,
                        const float *weights_host, float clip, float *out, void *stream);
/* The survivors of the RPN's NMS (rpn/inference_3d.py:127-131 `boxlist = boxlist[keep]`) as a list padded to P rows
 * while their count is still on the device: for i < *n_keep_dev, out_boxes[i] = boxes[keep[i]] with the three sizes
 * clamped to >= min_size (BoxList3D.clamp_size, structures/bounding_box_3d.py) and out_scores[i] = scores[keep[i]];
 * rows >= *n_keep_dev repeat candidate 0 (d3d_roi_prepare_counted switches them off).  keep int32 (device).
 * count_out (nullable): receives *n_keep_dev by a system-scope store -- pinned host memory: the host reads it after
 * an event recorded behind this launch, while later launches of the stream (the pooler) are already running.       */
int d3d_gather_kept(const float *boxes, const float *scores, const int32_t *keep, const int32_t *n_keep_dev, int P,
                    float min_size, float *out_boxes, float *out_scores, int32_t *count_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* D3D_HIP_H */
