/*
 * msi_hip.h -- C ABI of libmsi_hip.so: the MI355X (gfx950) native kernels of the
 * multi-sphere-image infer -> render hot path.
 *
 * The reference (brownvc/matryodshka) has no native code and no FFI: its only
 * seam is the Python class matryodshka.msi.MSI called from test.py:127-159.
 * Each entry point below therefore cites the reference *Python function* whose
 * arithmetic it replaces (file:line inside the reference checkout); the Python
 * class matryodshka_amd.msi.MSI keeps the reference's method signatures and is
 * the only caller (INTEGRATION.md shows the ctypes binding).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types cross this boundary;
 *   - the CALLER owns every buffer (device pointers unless the name ends in
 *     `_host`); the library never allocates persistent device memory; scratch
 *     is passed in, sized by the matching `*_workspace_bytes` query;
 *   - every launch is asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream); no hidden synchronisation;
 *   - return value: MSI_OK (0) or a negative MSI_E_* code; never a C++
 *     exception; msi_last_error_string() gives the thread-local detail;
 *   - re-entrant: no mutable global state besides the thread-local error text;
 *   - all tensors are contiguous fp32 in the documented layout, dims int32.
 *
 * Layouts
 *   image        [B,H,W,3]     preprocessed, range [-1,1]
 *   psv          [B,H,W,C]     C = 6*D for the two-source ODS volume; channel
 *                              = src*3D + d*3 + c (projector.py:164-169,
 *                              msi.py:1124-1129)
 *   pred         [B,H,W,2*D]   tanh output of the CNN (blend weights | alphas)
 *   rgba_native  [B,D,H,W,4]   the layer stack, D-major (what msi.py:422
 *                              transposes to before warping); float4 texels
 *   out          [B,H,W,3]
 *   trig         [2W+2H]       cosS[W] sinS[W] cosT[H] sinT[H] of the lat-long
 *                              grid (spherical.py:42-44), host-built
 */
#ifndef MSI_HIP_H
#define MSI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSI_OK 0
#define MSI_E_BADARG (-1)
#define MSI_E_LAUNCH (-2)
#define MSI_E_UNSUPPORTED (-3)
#define MSI_E_WORKSPACE (-4)
#define MSI_E_RANGE (-5) /* msi_net_plan_status: a LayerNorm statistic left the range the kernels resolve */

typedef void *msi_stream_t; /* hipStream_t */

const char *msi_version(void);
/* ABI version of this header: bumped whenever a struct of this file grows or changes layout, an entry point changes
 * its signature, or the PACKED weight blob (msi_net_pack_weights_host) changes format.  A caller compares
 * msi_abi_version() of the library it loaded with the MSI_ABI_VERSION it was compiled against and refuses to run on
 * a mismatch; packed blobs are not portable across versions (re-pack from the parameter blob).
 *   3: msi_layer_info.ln_scale_offset; LayerNorm window doubles in the packed blob (round 3)
 *   4: msi_net_plan_layer_kernel; render status word (msi_render_status_*); sweep volume takes shared poses (round 4)
 *   5: packed blob carries the fp16-split (x2) block; MSI_NET_OPT_F32_SPLIT_F16, MSI_NET_STATUS_F16_SPLIT_RANGE (round 4)
 *   6: msi_net_plan_calibrate; MSI_NET_OPT_X3_TILE8 (round 5)
 *   7: MSI_NET_OPT_X3_ROWPAR (round 5)
 *   8: msi_probe_matrix_rate (round 6) */
#define MSI_ABI_VERSION 8
int32_t msi_abi_version(void);
const char *msi_last_error_string(void);
/* CRC-32C (Castagnoli) of a host buffer, continuing from `crc` (0 for a new message): the per-tensor checksum of
 * the TensorFlow checkpoints test.py:191-202 restores (matryodshka_amd/tf_checkpoint.py verifies it on load). */
uint32_t msi_crc32c_host(const void *data_host, size_t n, uint32_t crc);

/* Measurement aid (round 6; bench.py's `roofline.frac_of_sustained`; no reference counterpart): `num_workgroups` workgroups of four waves issue
 * `iterations` x 12 v_mfma_f32_32x32x16_bf16 back to back per wave -- no memory or LDS traffic -- on operands that change between consecutive instructions
 * (changing_operands = 1) or stay constant (0).  Dense bf16 flops = num_workgroups x 4 x iterations x 12 x 2 x 32 x 32 x 16; time it with events on `stream`.
 * ticks_device (may be NULL): [num_workgroups][2] = per-workgroup deltas of s_memtime (shader-clock cycles) and s_memrealtime (100 MHz).  sink_device: one float. */
int32_t msi_probe_matrix_rate(int32_t changing_operands, int64_t iterations, int32_t num_workgroups, uint64_t *ticks_device, float *sink_device,
                              msi_stream_t stream);

/* ---- geometry tables (host) -------------------------------------------------
 * spherical.lat_long_grid (spherical.py:42-44) is separable; the kernels take
 * cos/sin of its two axes as a table.  tf.linspace fp32 semantics, then the
 * correctly rounded fp32 cos/sin of each fp32 angle (DESIGN.md "trig tables"). */
size_t msi_trig_table_floats(int32_t height, int32_t width);
int msi_build_trig_tables_host(int32_t height, int32_t width, float *out_host);

/* ---- pre / de-process -------------------------------------------------------
 * MSI.preprocess_image (msi.py:1163-1171): uint8 -> x*(1/255), then x*2-1. */
int msi_preprocess_u8_f32(const uint8_t *in, float *out, size_t n, msi_stream_t stream);
int msi_preprocess_f32(const float *in, float *out, size_t n, msi_stream_t stream);
/* both images of a frame (raw_src_image, raw_ref_image of infer_msi, msi.py:73-75) in one launch */
int msi_preprocess_pair_u8_f32(const uint8_t *in0, const uint8_t *in1, float *out0, float *out1, size_t n,
                               msi_stream_t stream);
/* MSI.deprocess_image (msi.py:1173-1181): trunc(((x+1)/2)*255.5) -> uint8;
 * is_depth != 0: MSI.deprocess_depth_image (msi.py:1186-1194): trunc(x*255.5).
 * Values are clamped to [0,255] before the cast. */
int msi_deprocess_f32_u8(const float *in, uint8_t *out, size_t n, int32_t is_depth,
                         msi_stream_t stream);
/* deprocess_image(rgb) and deprocess_depth_image(depth) of test.py:149-159 in one launch (n elements each) */
int msi_deprocess_pair_f32_u8(const float *rgb, const float *depth, uint8_t *out_rgb, uint8_t *out_depth, size_t n,
                              msi_stream_t stream);

/* out[b] = lhs[b] @ rhs[b] for [B,4,4] row-major poses, terms summed k = 0..3 without fma:
 * curr_pose = psv_src_pose @ ref_pose_inv (msi.py:1125), tgt_pose @ interp_pose_inv (msi.py:644-646).
 * One 16-thread-per-sample launch instead of a BLAS call in the frame loop. */
int msi_compose_poses_f32(const float *lhs, const float *rhs, float *out, int32_t batch,
                          msi_stream_t stream);

/* ---- K1: ODS sphere sweep -----------------------------------------------------
 * pj.ods_sphere_sweep -> sweep_one (projector.py:209-211, 129-170) with
 * backproject_spherical (spherical.py:116-129), apply_pose (projector.py:275-291),
 * project_ods (spherical.py:170-233) and the wrap-around bilinear gather
 * sampling.resample (sampling.py:135-197), fused; nothing is materialised but
 * the output.  The part of project_ods that decides its branches (up to the discriminant) is IEEE fp32 op for op;
 * the continuous remainder (root, angles, pixel coordinates) uses 1-ulp primitives (<= 2e-5 px).  Writes channels [channel_offset, channel_offset+3*D) of psv.
 *   pose [B,4,4] (curr_pose = src_pose @ ref_pose_inv, msi.py:1125),
 *   intrinsics [B,3,3] (ODS baseline in [b,0,0], data_loader.py:160),
 *   depths [D], order = +1 (ref) / -1 (src) (msi.py:1127). */
int msi_ods_sphere_sweep_f32(const float *image, const float *pose, const float *intrinsics,
                             const float *depths, const float *trig, int32_t batch,
                             int32_t height, int32_t width, int32_t num_depths, int32_t order,
                             float *psv, int32_t psv_channels, int32_t channel_offset,
                             msi_stream_t stream);

/* Same sweep with a bf16 volume (round to nearest even): the network input of BASELINE
 * configs[2]; everything up to the final store is the fp32 arithmetic above. */
int msi_ods_sphere_sweep_bf16(const float *image, const float *pose, const float *intrinsics,
                              const float *depths, const float *trig, int32_t batch,
                              int32_t height, int32_t width, int32_t num_depths, int32_t order,
                              void *psv_bf16, int32_t psv_channels, int32_t channel_offset,
                              msi_stream_t stream);

/* The whole double volume of MSI.format_network_input (msi.py:1124-1129) in one launch: ref_image with order +1
 * into channels [0,3D), src_image with order -1 into [3D,6D) of psv [B,H,W,6D] (fp32, or bf16 when psv_is_bf16 != 0).
 * The poses are the two curr_pose = pose @ ref_pose_inv (msi_compose_pose_pair_f32).  When they are equal per
 * sample (test path: identity poses) the branch-deciding quadratic of project_ods is evaluated once for both
 * sources; results are identical to two msi_ods_sphere_sweep_* calls. */
int msi_ods_sweep_volume(const float *ref_image, const float *src_image, const float *ref_curr_pose,
                         const float *src_curr_pose, const float *intrinsics, const float *depths, const float *trig,
                         int32_t batch, int32_t height, int32_t width, int32_t num_depths, void *psv, int32_t psv_is_bf16,
                         msi_stream_t stream);
/* out0[b] = lhs0[b] @ rhs[b], out1[b] = lhs1[b] @ rhs[b] (one launch for both curr_pose, msi.py:1125). */
int msi_compose_pose_pair_f32(const float *lhs0, const float *lhs1, const float *rhs, float *out0, float *out1,
                              int32_t batch, msi_stream_t stream);

/* ---- K3: RGBA layer assembly ----------------------------------------------------
 * infer_msi "layer_prediction", which_color_pred = blend_psv (msi.py:130-147):
 * w=(pred[..,d]+1)/2, a=(pred[..,D+d]+1)/2, rgb = w*psv_ref_d + (1-w)*psv_src_d.
 * Writes rgba_native [B,D,H,W,4]; blend_weights / alphas ([B,H,W,D]) may be NULL
 * (the optional extra outputs of msi.py:281-287). */
int msi_assemble_rgba_f32(const float *psv, const float *pred, float *rgba_native,
                          float *blend_weights, float *alphas, int32_t batch, int32_t height,
                          int32_t width, int32_t num_planes, msi_stream_t stream);

/* msi_assemble_rgba_f32 reading the bf16 PSV of the bf16 path (pred, outputs fp32). */
int msi_assemble_rgba_bf16psv_f32(const void *psv_bf16, const float *pred, float *rgba_native,
                                  float *blend_weights, float *alphas, int32_t batch, int32_t height,
                                  int32_t width, int32_t num_planes, msi_stream_t stream);

/* The other colour schemes of infer_msi (FLAGS.which_color_pred, test.py:55-56; msi.py:166-275).  pred holds
 *   MSI_COLOR_BLEND_PSV    [w | alpha]                2D   channels  (= msi_assemble_rgba_f32)
 *   MSI_COLOR_BLEND_BG     [w | alpha | bg rgb]       2D+3           rgb = w psv_ref + (1-w) bg          (msi.py:177-188)
 *   MSI_COLOR_BLEND_BG_PSV [w | alpha | bw | bg rgb]  3D+3           rgb = bw (w psv_ref + (1-w) psv_src) + (1-bw) bg
 *   MSI_COLOR_ALPHA_ONLY   [alpha]                    D              rgb = psv_ref                       (msi.py:258-268)
 * with w, alpha, bw = (x+1)/2 and bg the raw tanh output.  psv [B,H,W,6D] fp32, or bf16 when psv_is_bf16 != 0;
 * blend_weights / alphas / bg_blend_weights [B,H,W,D] may be NULL (msi.py:276-289). */
#define MSI_COLOR_BLEND_PSV 0
#define MSI_COLOR_BLEND_BG 1
#define MSI_COLOR_BLEND_BG_PSV 2
#define MSI_COLOR_ALPHA_ONLY 3
int msi_assemble_rgba_color_f32(const void *psv, int32_t psv_is_bf16, const float *pred, int32_t which_color_pred,
                                float *rgba_native, float *blend_weights, float *alphas, float *bg_blend_weights,
                                int32_t batch, int32_t height, int32_t width, int32_t num_planes, msi_stream_t stream);

/* High-res re-render (test.py:283-394): the per-plane loop there is (a) the high-res sphere
 * sweep (msi_ods_sphere_sweep_f32 at the high resolution), (b) tf.image.resize(BILINEAR,
 * align_corners=True) of the low-res blend weights / alphas (test.py:319-325), (c) the blend of
 * test.py:327-334 and (d) the warp + over-composite of test.py:337-382.  (b) and (c) are: */
int msi_resize_bilinear_f32(const float *in, float *out, int32_t batch, int32_t in_h, int32_t in_w,
                            int32_t channels, int32_t out_h, int32_t out_w, msi_stream_t stream);
/* as msi_assemble_rgba_f32, but `weights_alphas` [B,H,W,2*D] already holds blend weights | alphas
 * in (0,1) (i.e. after the (x+1)/2 of msi.py:132-133). */
int msi_assemble_rgba_scaled_f32(const float *psv, const float *weights_alphas, float *rgba_native,
                                 int32_t batch, int32_t height, int32_t width, int32_t num_planes,
                                 msi_stream_t stream);

/* ---- K4: target-view reprojection + over-composite -------------------------------
 * MSI.msi_render_equirect_view / _depth (msi.py:407-429, 384-405):
 * pj.projective_forward_sphere (projector.py:34-62) = spherical.intersect_sphere
 * (spherical.py:268-326) -> project_spherical (:235-246) -> theta_phi_to_pixels
 * (:54-68) -> sampling.resample per layer, then pj.over_composite
 * (projector.py:246-265) and/or pj.over_composite_depth (:225-244), fused in one
 * pass: neither the pixel coordinates nor the warped layers are materialised,
 * and RGB + depth share the warp.  out_rgb / out_depth may be NULL (not both).
 *   tgt_pose_rt [B,4,4], tgt_pos [B,3], depths [D] (far -> near).
 * status_device (all four render entry points; may be NULL): one int32 in DEVICE memory the kernel ORs
 * MSI_RENDER_STATUS_ORIGIN_OUTSIDE into when a sample's ray origin (pose @ tgt_pos, or the ODS viewing circle) is not
 * strictly inside every sphere it is intersected with.  There spherical.py:316-318 takes the square root of a negative
 * number and the int cast of the NaN pixel coordinate is undefined; the kernel clamps the discriminant (finite pixels,
 * not reference-defined) and reports it here instead of failing silently.  The caller zeroes the word and reads it back
 * when it wants to know (the MSI class checks host-side inputs before the launch and device-side ones through this). */
#define MSI_RENDER_STATUS_ORIGIN_OUTSIDE 1
int msi_render_equirect_f32(const float *rgba_native, const float *tgt_pose_rt,
                            const float *tgt_pos, const float *depths, const float *trig,
                            int32_t batch, int32_t height, int32_t width, int32_t num_planes,
                            float *out_rgb, float *out_depth, int32_t *status_device, msi_stream_t stream);
/* MSI.msi_render_equirect_view_single (msi.py:431-452): the warped, un-composited
 * layers, out_layers [D,B,H,W,4]. */
int msi_project_layers_f32(const float *rgba_native, const float *tgt_pose_rt,
                           const float *tgt_pos, const float *depths, const float *trig,
                           int32_t batch, int32_t height, int32_t width, int32_t num_planes,
                           float *out_layers, int32_t *status_device, msi_stream_t stream);

/* MSI.msi_render_ods_view (msi.py:502-525): pj.projective_forward_ods (projector.py:100-127) =
 * spherical.intersect_ods (spherical.py:328-365; eye rays tangent to the viewing circle of radius
 * intrinsics[b,0,0], order +1 left / -1 right, transformed by pose [B,4,4]) -> resample ->
 * over_composite.  out_rgb [B,H,W,3]. */
int msi_render_ods_f32(const float *rgba_native, const float *pose, const float *intrinsics,
                       const float *depths, const float *trig, int32_t batch, int32_t height,
                       int32_t width, int32_t num_planes, int32_t order, float *out_rgb,
                       int32_t *status_device, msi_stream_t stream);
/* MSI.msi_render_perspective_view (msi.py:475-500): pj.projective_forward_sphere_to_perspective
 * (projector.py:64-98) = spherical.intersect_perspective (spherical.py:367-401; hard-coded
 * 0.1/0.05 intrinsics) -> resample -> over_composite.  `pose` [B,4,4] is the crop rotation the
 * caller builds from viewing_window (projector.py:78-86); out_rgb [B,tgt_height,tgt_width,3]. */
int msi_render_perspective_f32(const float *rgba_native, const float *pose, const float *tgt_pos,
                               const float *depths, int32_t batch, int32_t height, int32_t width,
                               int32_t num_planes, int32_t tgt_height, int32_t tgt_width, float *out_rgb,
                               int32_t *status_device, msi_stream_t stream);

/* ---- PP (perspective cube-face) path, BASELINE configs[4] ------------------------------------
 * pj.perspective_plane_sweep (projector.py:221-223): sweep_one with spherical.uv_grid (:46-48),
 * backproject_planar (:131-149), apply_pose (projector.py:275-291), project_perspective
 * (spherical.py:248-266) and the wrap-around sampler.  intrinsics [B,3,3] = fx,cx / fy,cy. */
int msi_perspective_plane_sweep_f32(const float *image, const float *pose, const float *intrinsics,
                                    const float *depths, int32_t batch, int32_t height, int32_t width,
                                    int32_t num_depths, float *psv, int32_t psv_channels,
                                    int32_t channel_offset, msi_stream_t stream);
/* MSI.mpi_render_view (msi.py:527-548): pj.projective_forward_homography (projector.py:343-373) ->
 * homography.planar_transform (homography.py:35-157) -> tf.contrib.resampler (zero padding) ->
 * pj.over_composite.  intrinsics_inv [B,3,3] replaces the hidden graph input `intrinsics_inv:0`
 * (homography.py:52). */
int msi_mpi_render_f32(const float *rgba_native, const float *tgt_pose, const float *intrinsics,
                       const float *intrinsics_inv, const float *depths, int32_t batch, int32_t height,
                       int32_t width, int32_t num_planes, float *out_rgb, msi_stream_t stream);

/* ---- K2: encoder-decoder CNN -------------------------------------------------------
 * nets.msi_coord_train_net (nets.py:471-515; coord_net=1) and nets.msi_train_net
 * (nets.py:387-450; coord_net=0): 14x conv3x3 (+|sin(lat)| coordinate channel,
 * nets.py:260-270), 3x conv-transpose 4x4 s2, LayerNorm over (H,W,C) + ReLU after
 * each, 1x1 tanh head with bias.  Implicit-GEMM on fp32 MFMA; LayerNorm sums are produced by the conv
 * epilogue -- order-independent 64-bit fixed-point accumulation (one word per sum, every wave's share rounded to one
 * unit) inside a per-layer window whose exponent the packer derives from the weights; a forward whose statistics leave
 * that window reports it through msi_net_plan_status instead of returning silently wrong values -- and applied with
 * the ReLU by the consumer while it stages its input (halo-patch kernels, head) or by one in-place pass per layer.  msi_train_net's conv-transposes are normalised over their uncropped
 * (2H+10) x (2W+10) VALID output, as nets.py:423-435 does, before the [5:-5] crop. */
typedef struct msi_net_desc {
  int32_t batch, height, width; /* height, width multiples of 8 */
  int32_t in_channels;          /* 6*D */
  int32_t num_outputs;          /* 2*D for blend_psv */
  int32_t ngf;                  /* 64 in the reference */
  int32_t coord_net;            /* 1: msi_coord_train_net, 0: msi_train_net */
  int32_t dtype;                /* MSI_DTYPE_F32 (0) or MSI_DTYPE_BF16 (1): operand type of the convolutions */
} msi_net_desc;

#define MSI_DTYPE_F32 0
#define MSI_DTYPE_BF16 1

#define MSI_NET_NUM_LAYERS 18

typedef struct msi_layer_info {
  char name[16];          /* TF scope name: conv1_1 ... conv8_2, color_pred */
  int32_t kind;           /* 0 conv3x3, 1 convT4x4s2, 2 head 1x1 */
  int32_t cin, cout;      /* cin without the coordinate channel */
  int32_t has_coord;
  int32_t stride, rate;
  int32_t in_h, in_w, out_h, out_w;
  uint64_t param_offset;  /* float offset of `weights` in the parameter blob   */
  uint64_t param_floats;  /* weights + gamma + beta (or + biases)              */
  uint64_t raw_offset;    /* BYTE offset of the raw (pre-LayerNorm) output in  */
                          /* the workspace; (uint64)-1 for the head.  fp32     */
                          /* plans: fp32 [B,H,W,cout]; bf16 plans: fp16 of     */
                          /* x * 2^-e, e = the layer's LayerNorm window (below) */
  uint64_t affine_offset; /* BYTE offset of scale[cout] shift[cout] in the ws  */
  uint64_t ln_scale_offset; /* FLOAT offset in the PACKED blob of four doubles */
                          /* {S1, S2, 1/S1, 1/S2}, S1 = 2^(24 - e), S2 =       */
                          /* 2^(16 - 2e): the fixed-point window of the layer's */
                          /* LayerNorm sums; 2^e = 2^24 / S1                    */
} msi_layer_info;

/* Parameter blob ("reference layout"), fp32, layers in graph order; per layer
 *   conv3x3 : weights [3,3,cin+has_coord,cout], gamma [cout], beta [cout]
 *   convT   : weights [4,4,cout,cin],           gamma [cout], beta [cout]
 *   head    : weights [1,1,cin,cout],           biases [cout]
 * i.e. the TF variables net/<name>/weights, .../LayerNorm/gamma, .../LayerNorm/beta,
 * net/color_pred/{weights,biases} (test.py:191-202 restores exactly these). */
int msi_net_layer_info(const msi_net_desc *desc, int32_t layer, msi_layer_info *out);
size_t msi_net_param_floats(const msi_net_desc *desc);
size_t msi_net_packed_floats(const msi_net_desc *desc);
int msi_net_pack_weights_host(const msi_net_desc *desc, const float *params_host,
                              float *packed_host);
size_t msi_net_workspace_bytes(const msi_net_desc *desc);

/* A plan resolves everything about a descriptor that does not depend on the buffers -- layer table, tile
 * choice and work decomposition per layer, workspace layout, the CU count of the current device
 * (hipDeviceProp.multiProcessorCount) -- once, so that msi_net_plan_forward does no allocation and no
 * per-layer set-up arithmetic.  The plan is host memory owned by the caller (create / destroy); forward takes
 * it const and keeps all per-call state on the stack: concurrent forwards on one plan (different streams,
 * different workspaces) are safe.  Options replace what used to be environment variables; they are per plan,
 * never process-global, and changing one re-plans (query msi_net_plan_workspace_bytes again afterwards). */
typedef struct msi_net_plan msi_net_plan;
#define MSI_NET_OPT_FIXUP_KERNEL 0 /* 0 (default): split tiles are summed inside the conv launch (last arriver);  */
                                   /* 1: by a separate conv_fixup_kernel launch (bitwise-identical results)       */
#define MSI_NET_OPT_TAILSPLIT 1    /* cut the tiles of the partial last round along K: 1 (default) whole tiles in multiples of   */
                                   /* the CU count, 2 in multiples of a full residency (5 workgroups x CUs; measured slower), 0 never */
#define MSI_NET_OPT_BIGTILE 2      /* bf16 tile choice: 0 never 128x128 / 128x64, 1 (default) by grid size, 2 always */
#define MSI_NET_OPT_HEAD_FUSE_LN 3 /* 1 (default): the fp32 head applies its source's LayerNorm while loading      */
#define MSI_NET_OPT_NUM_CUS 4      /* CUs the work decomposition balances over (default: the device's count)      */
#define MSI_NET_OPT_F32_TILE 5     /* fp32 tile of the layers in F32_TILE_MASK: 0 = 64x64, 1 = 128x64, 2 = 64x128 (tuning)     */
#define MSI_NET_OPT_F32_TILE_MASK 6 /* bit i = layer i (graph order) uses MSI_NET_OPT_F32_TILE                                 */
#define MSI_NET_OPT_APPLY_AHEAD 7   /* 1: a layer's LayerNorm + ReLU is applied by the first workgroups of its consumer's launch,    */
                                   /* overlapped with that layer's tiles (row counters); 0 (default): one ln_apply launch per layer */
                                   /* -- bit-identical results; measured slower on MI355X (write-through hand-off), see DESIGN.md  */
#define MSI_NET_OPT_HALO 8          /* default 5.  bit 0: the stride-1 3x3 layers run the halo-patch kernels (conv_halo_kernel fp32,   */
                                   /* conv_halo_bf16_kernel: one LDS-stationary halo patch per workgroup and input chunk, the          */
                                   /* producer's LayerNorm applied while staging it) and the bf16 conv-transposes                      */
                                   /* convt_halo_bf16_kernel; bit 1 (measured slower: opt-in): the fp32 SAME conv-transposes run       */
                                   /* convt_halo_kernel (the two classes of one output-row parity per workgroup, either source's       */
                                   /* LayerNorm applied while staging); bit 2: the fp32 stride-2 3x3 layers run conv_halo_s2_kernel    */
                                   /* (four parity-plane patches per 32-channel group, the producer's LayerNorm applied while staging);*/
                                   /* 0: tap-DMA kernel everywhere.  With F32_SPLIT3 on for a layer (the default) ANY non-zero HALO    */
                                   /* value lets that layer take its split halo kernel: the conv-transposes run convt_halo_x3_kernel    */
                                   /* without bit 1 and the stride-2 layers conv_halo_s2_x3_kernel whatever their grid size (the native */
                                   /* stride-2 form is only chosen at >= 3 tiles per CU); set F32_SPLIT3 = 0 to get the bit-by-bit      */
                                   /* selection described above                                                                         */
#define MSI_NET_OPT_HALO_SKIP 9     /* bit i = layer i (graph order) does NOT take a halo kernel although it qualifies (tuning)          */
#define MSI_NET_OPT_UNIFORM_SPLIT 10 /* s >= 2: layers with one to two 64x64 tiles per CU cut EVERY tile into s equal K-ranges (tuning; 0 = default split) */
#define MSI_NET_OPT_BF16_STAGE_RAW 11 /* bf16 plans: bit 0 the 256x64 conv tile (conv8_2), bit 1 the 128x64 conv-transpose tile (conv8_1) read their  */
                                      /* sources RAW (fp16) and apply the producer's LayerNorm while staging (default 1: bit 1 measured slower); 0: from ln_apply's bf16 copies */
#define MSI_NET_OPT_SPLIT_OVERHEAD 12 /* k-steps of prologue + epilogue a K-range visit is charged in the tail-split cost model (tuning; 0 = r01 rule) */
#define MSI_NET_OPT_BF16_WAVES 13 /* waves per workgroup of the bf16 halo-patch conv kernel's 128x128 tile: 8 (default; 4 x 2 waves of 32 pixels x 64 channels: */
                                  /* four waves per SIMD with two workgroups per CU, so that a workgroup's prologue / epilogue has neighbours to hide behind) */
                                  /* or 4 (r02 shape).  The 256x64 tile always runs four (eight measured slower) */
#define MSI_NET_OPT_F32_SPLIT3 14 /* fp32 plans, bit i = layer i (graph order): a 3x3 layer (stride 1 or 2, rate 1 or 2) or conv-transpose (SAME, or msi_train_net's */
                                  /* wrap_pad + VALID form) that qualifies for a halo-patch kernel (MSI_NET_OPT_HALO != 0) computes its fp32 convolution as a 3-way */
                                  /* bf16 split of both operands with SIX products (h.h, h.m, m.h, h.l, l.h, m.m; exact products, fp32 accumulation; */
                                  /* dropped terms <= 2^-26 of a product: fp32-grade, NOT the 3-product TF32-grade split) on the 16x faster bf16 MFMA */
                                  /* (conv_halo_x3_kernel).  Default 0x3ffff (every layer that qualifies); 0 = native fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere */
#define MSI_NET_OPT_F32_SPLIT_F16 15 /* fp32 plans, bit i = layer i: a layer that runs the split (F32_SPLIT3) uses its fp16 form -- x = h + m' 2^-11 with fp16 */
                                     /* parts (22 significand bits), THREE products h.h + (h.m' + m'.h) 2^-11, fp32 accumulation: half the matrix work of the */
                                     /* six-product bf16 form at the same measured error (profiles/r04_split_numerics.txt), but the operands must lie in */
                                     /* the fp16 RANGE: |x| > 65504 (weights or normalised activations) poisons the layer and sets MSI_NET_STATUS_F16_SPLIT_RANGE. */
                                     /* The 22 bits hold for |x| >= 2^-14 only: below that h, and below 2^-25 m' too, are fp16 subnormals (absolute resolution 2^-24 and */
                                     /* 2^-35) -- LayerNorm'd activations and slim-initialised weights sit far above; nothing flags an operand that small.  NaN operands */
                                     /* set the status bit like |x| > 65504. */
                                     /* Default 0 (opt-in: 22-bit operands are narrower than fp32's 24 -- the default fp32 arithmetic stays the six-product bf16 form); */
                                     /* 0x3ffff = every layer that runs the split, msi_train_net's conv-transposes included (until r05 those ignored the bit: the r04 */
                                     /* finding that kept them on the bf16 form is closed, DESIGN.md section 4 "the wobble") */
#define MSI_NET_OPT_X3_TILE8 16 /* fp32 plans, bit i = layer i: a stride-1, rate-1 layer on the six-product split (F32_SPLIT3 without F32_SPLIT_F16) whose input height is a  */
                                /* multiple of 8 runs conv_halo8_x3_kernel: 8 x 16-pixel x 64-channel tiles, two accumulators per wave sharing the weight fragments -- half */
                                /* the weight bytes L2 -> LDS, half the prologues / patch swaps per output, 0.75 fragment reads per MFMA, two workgroups per CU.  Same        */
                                /* arithmetic and per-accumulator summation order as the 4-row tile.  Applied only where the layer's grid stays >= 3 such tiles per CU       */
                                /* (smaller grids are cut into K-ranges either way and lose); bit 30 forces it on every eligible layer (tests).  Default 0x3ffff.            */
                                /* The conv-transposes and the stride-2 layers take the 8 x 16-pixel tile under the same bit and rule (convt_halo8_x3_kernel: four            */
                                /* accumulators per wave; conv_halo8_s2_x3_kernel: 9 x 17-pixel unit patches)                                                               */
#define MSI_NET_OPT_X3_ROWPAR 17 /* fp32 plans, bit i = layer i: a stride-1, RATE-2 layer on the split (F32_SPLIT3) whose input height is a multiple of 8 runs on ROW-PARITY  */
                                 /* tiles: a tile's four rows are every other image row (tile row t = 2 t' + parity -> rows 8 t' + parity + 2 r), so along H a dilation-2 tap */
                                 /* is the NEXT tile row -- a 6 x 20-pixel patch instead of 8 x 20, the two-stage weight ring and three workgroups per CU instead of two.      */
                                 /* Same arithmetic and summation order per output element (bit-identical to the plain rate-2 tile).  Default 0x3ffff                         */
#define MSI_NET_OPT_COUNT 18
int msi_net_plan_create(const msi_net_desc *desc, msi_net_plan **out_plan);
void msi_net_plan_destroy(msi_net_plan *plan);
int msi_net_plan_set_option(msi_net_plan *plan, int32_t option, int32_t value);
size_t msi_net_plan_workspace_bytes(const msi_net_plan *plan);
/* After a forward, does workspace[raw_offset of `layer`] hold the layer's LayerNorm + ReLU'd activation (1) or its raw
 * convolution output (0: every consumer applies the LayerNorm itself while loading -- the head, halo-patch layers)?
 * -1: bad arguments.  (Tests / debugging; the frame loop does not need it.) */
int32_t msi_net_plan_layer_is_normalized(const msi_net_plan *plan, int32_t layer);
/* Which kernel instantiation the plan launches for `layer` (0 .. MSI_NET_NUM_LAYERS-1), written to name[name_bytes] as
 * rocprofv3 spells it without the namespace -- e.g. "conv_halo_s2_kernel<1>", "conv_halo_bf16_kernel<128, 128, 1, 1, 8>",
 * "conv_igemm_kernel<64, 64, 1, 0>" (<BM, BN, MODE 0 conv / 1 conv-transpose / 2 head, BF16>); *nblocks = workgroups of the
 * launch, *nsplit_tiles = tiles cut into K-ranges (both optional).  The choice depends on batch x tiles against the CU
 * count, so a parity test at a bench batch asserts with this that the variants it checked are the ones the bench times
 * (color_pred reports the stand-alone head; msi_net_plan_forward_rgba replaces it by head_assemble_kernel). */
int32_t msi_net_plan_layer_kernel(const msi_net_plan *plan, int32_t layer, char *name, size_t name_bytes, int32_t *nblocks,
                                  int32_t *nsplit_tiles);
/* Health of the LAST forward that ran with `workspace` on `stream` (synchronises the stream: call it after a frame,
 * not inside the frame loop's hot path): MSI_OK, or MSI_E_RANGE with the reason in msi_last_error_string when a
 * LayerNorm sum overflowed its fixed-point window (bit 1 of *status_bits: raw outputs > ~3000x the scale the weights
 * predict, or non-finite input; bf16 plans also set it when a raw output may have left the range of the fp16 it is
 * stored in -- some |x 2^-e - pivot| of a wave's 1 024 values above 32 752, i.e. > ~1000x that scale) or a variance fell below its resolution (bit 2: < ~1e-3 of that scale, or a constant
 * layer); bit 0: an apply-ahead wait timed out.  status_bits may be NULL.  The word is reset by the next forward. */
#define MSI_NET_STATUS_APPLY_AHEAD_TIMEOUT 1
#define MSI_NET_STATUS_LN_OVERFLOW 2
#define MSI_NET_STATUS_LN_UNDERFLOW 4
#define MSI_NET_STATUS_F16_SPLIT_RANGE 8 /* an operand of a layer on the fp16 split (F32_SPLIT_F16) exceeded 65504: rerun with that option 0 */
int32_t msi_net_plan_status(const msi_net_plan *plan, const void *workspace, msi_stream_t stream, int32_t *status_bits);
/* LayerNorm window calibration (round 5).  The fixed-point window of a layer's LayerNorm sums (MSI_NET_STATUS_LN_*) follows an exponent the packer
 * estimates from the weights; a checkpoint whose trained gamma / beta / weights put a layer's raw output far from that estimate would end every forward in
 * MSI_E_RANGE.  msi_net_plan_calibrate measures instead: layer by layer it runs the network on `net_input` (a representative frame, device memory, the
 * forward's layout), moves / centres each layer's window on the measured raw rms and rewrites the four scale doubles at ln_scale_offset of `packed`
 * (DEVICE memory, modified in place; every plan that shares the blob sees the new windows).  Blocking (it synchronises `stream` several times per layer),
 * a one-off of a few dozen forwards; *layers_changed (may be NULL) = layers whose exponent moved.  MSI_E_RANGE if a layer has no finite, non-constant output.
 * ALL OR NOTHING (round 6): on ANY error return every window is written back as it was on entry.  Within a batch the window is centred on the samples it
 * resolves; a constant sample among others is left out (a forward flags it itself). */
int32_t msi_net_plan_calibrate(const msi_net_plan *plan, float *packed, const void *net_input, void *workspace, size_t workspace_bytes,
                               msi_stream_t stream, int32_t *layers_changed);
/* net_input [B,H,W,in_channels] (fp32, or bf16 when desc.dtype = MSI_DTYPE_BF16) -> pred [B,H,W,num_outputs] fp32. */
int msi_net_plan_forward(const msi_net_plan *plan, const float *packed, const void *net_input, float *pred,
                         void *workspace, size_t workspace_bytes, msi_stream_t stream);

/* The network followed by infer_msi's layer_prediction for which_color_pred = blend_psv (msi.py:130-147) with the
 * 1x1 head, its source's LayerNorm and the RGBA assembly fused into ONE kernel: `pred` never exists in HBM.
 * net_input [B,H,W,6D] (fp32, or bf16 for a bf16 plan) is both the network input and the sweep volume the layers are
 * blended from; rgba_native [B,D,H,W,4] fp32; blend_weights / alphas [B,H,W,D] and pred [B,H,W,2D] (the tanh output)
 * are optional (NULL).  fp32 plans: bit-identical to msi_net_plan_forward + msi_assemble_rgba_f32.  bf16 plans: the
 * head runs on the fp32 MFMA over the same bf16-rounded operands (activation rounded where ln_apply rounds it, weights
 * rounded at pack time), so it equals the two-step bf16 path up to the fp32 summation order; conv8_2 then has no
 * ln_apply launch and no bf16 copy.  event_after_convs: optional hipEvent_t recorded on `stream` between the last
 * convolution and the fused tail (bench.py times the MFMA-bound and the HBM-bound part separately).
 * MSI_E_UNSUPPORTED for other colour schemes / ngf > 64 / D > 64 / HEAD_FUSE_LN = 0: use msi_net_plan_forward +
 * msi_assemble_rgba_color_f32 there. */
int msi_net_plan_forward_rgba(const msi_net_plan *plan, const float *packed, const void *net_input, float *rgba_native,
                              float *blend_weights, float *alphas, float *pred, void *workspace, size_t workspace_bytes,
                              msi_stream_t stream, void *event_after_convs);

/* Descriptor-level convenience: the two calls below build a transient plan per call (set-up time, tests); the
 * frame loop uses msi_net_plan_forward.
 * net_input [B,H,W,in_channels] -> pred [B,H,W,num_outputs]. */
int msi_net_forward_f32(const msi_net_desc *desc, const float *packed, const float *net_input,
                        float *pred, void *workspace, size_t workspace_bytes,
                        msi_stream_t stream);
/* BASELINE configs[2] (bf16): desc->dtype = MSI_DTYPE_BF16.  net_input [B,H,W,in_channels] bf16
 * (msi_ods_sphere_sweep_bf16 writes it), weights and activations bf16 (round to nearest even),
 * products accumulated in fp32 on v_mfma_f32_32x32x16_bf16, LayerNorm statistics and affine in
 * fp32 on the fp32 accumulators, pred [B,H,W,num_outputs] fp32.  in_channels and ngf must be
 * multiples of 8.  The packed blob (same size query / packer, keyed by desc->dtype) holds bf16
 * weights and fp32 gamma / beta / bias. */
int msi_net_forward_bf16(const msi_net_desc *desc, const float *packed, const void *net_input_bf16,
                         float *pred, void *workspace, size_t workspace_bytes,
                         msi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MSI_HIP_H */
