/*
 * kpnerf.h — C ABI of libkpnerf_hip.so, the MI355X (gfx950) ray-march library behind
 * KeypointNeRF's render path.
 *
 * The reference (facebookresearch/KeypointNeRF, pure Python) has no FFI/plugin interface; the seam
 * for this path is attribute lookup on its `net` object.  Each entry point below
 * replaces one of those Python callables and cites it (file:line under the reference repo).
 * The binding a maintainer would add on the reference side is a ctypes stub — see INTEGRATION.md
 * and keypointnerf_amd/lib.py, which is exactly that stub.
 *
 * Conventions
 *   - every pointer is DEVICE memory unless its name ends in `_host`;
 *   - all arithmetic is fp32 (the reference sets torch default dtype float32, train.py:16);
 *     index/compaction lists are int32; masks are uint8 (0/1);
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it, never synchronise,
 *     never allocate;
 *   - inputs are never written; outputs and `workspace` are caller-allocated;
 *   - every function returns 0 on success, a negative KPN_E* code otherwise; kpn_last_error()
 *     gives the message (thread-local).  Nothing returns NaN silently for valid inputs.
 */
#ifndef KPNERF_H
#define KPNERF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 6): kpn_render_args gained `stages` at its end (callers built against 2 must be rebuilt: the struct grew);
 * new entry points: kpn_set_density_first / kpn_get_density_first / kpn_density_stats / kpn_density_first_passes, kpn_bwd_profile_* */
#define KPN_ABI_VERSION 3
#define KPN_N_KPT 24      /* configs/zju.json:44 sp_args.n_kpt */
#define KPN_MAX_VIEWS 16

enum { KPN_OK = 0, KPN_EINVAL = -1, KPN_ELAUNCH = -2, KPN_EWORKSPACE = -3 };

int kpn_abi_version(void);
const char* kpn_last_error(void);
/* 1 if the library was built for the device (gfx950), 0 for the host SIMT emulator build (tests only) */
int kpn_is_device_build(void);

/* ---------------------------------------------------------------------------------------------
 * Parameters.  `plain_host`: the weight-norm-folded hot-path parameters as one flat fp32 vector in
 * the order of keypointnerf_amd/synthetic.py:HOTPATH_LAYERS (W row-major (out,in) then bias per
 * layer, raw ani_al last) — i.e. the reference modules mlp_geo (src/utils.py:476-517),
 * ibr_compress_gfeat (src/model.py:577-580) and mlp_tex (src/model.py:1239-1258).
 * kpn_pack_weights re-orders them into the MFMA A-operand streams the kernels read. */
size_t kpn_plain_weight_floats(void);
size_t kpn_packed_weight_floats(void);
int kpn_pack_weights(const float* plain_host, float* packed_host);
/* The same re-ordering on the device (plain_dev -> packed_dev, both device pointers, asynchronous on `stream`): a
 * training loop re-packs after every optimizer step without a host round trip.  The index map is taken once per
 * process from kpn_pack_weights. */
int kpn_pack_weights_device(const float* plain_dev, float* packed_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Scene = what KeypointNeRF.query() receives besides the points: source cameras, keypoints, source
 * images, fg masks and the encoder feature maps in NCHW, as the reference hands them over
 * (src/model.py:690-701, 336-355, 653-680). */
typedef struct kpn_scene_desc {
    int32_t n_views;                 /* V  (<= KPN_MAX_VIEWS) */
    int32_t src_h, src_w;            /* source image size H, W (cam["height"], cam["width"]) */
    int32_t geo0_h, geo0_w;          /* feat_geo[0]: (V,64,geo0_h,geo0_w) */
    int32_t geo1_h, geo1_w;          /* feat_geo[1]: (V, 8,geo1_h,geo1_w) */
    int32_t tex_h, tex_w;            /* feat_tex   : (V, 8,tex_h,tex_w)   */
    int32_t disable_fg_mask;         /* KeypointNeRF.disable_fg_mask, src/model.py:566,734 */
    float znear, zfar;               /* cam["znear"], cam["zfar"] (2.0/5.0, src/model.py:43,345) */
    float nml_scale;                 /* cam["nml_scale"] (100.0) */
    float sigma;                     /* sp_args.sigma (0.1) */
    const float* KRT;                /* (V,4,4) */
    const float* extrin;             /* (V,4,4) */
    const float* kpt3d;              /* (24,3) */
    const float* img;                /* (V,3,H,W) */
    const uint8_t* fg_mask;          /* (V,H,W) bool bytes; ignored if disable_fg_mask */
    const float* geo0;
    const float* geo1;
    const float* tex;
} kpn_scene_desc;

/* bytes of device workspace that kpn_scene_prepare fills (channels-last maps + per-view tables) */
size_t kpn_scene_workspace_bytes(const kpn_scene_desc* desc);
/* Builds the kernel-side scene: NCHW -> channels-last (one 256-B line per 64-channel tap), RGB+mask
 * interleaved, keypoints moved to every camera frame (src/spatial.py:85), source camera centres
 * (inverse(KRT)[:3,3], src/model.py:823-824). */
int kpn_scene_prepare(const kpn_scene_desc* desc, void* scene_ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stage ops = the reference's per-stage callables. */

/* KeypointNeRF.ray_bbox_intersection, src/model.py:1178-1237.
 * bounds (2,3), orig (3), direct (R,3) -> near (R), far (R), hit (R). */
int kpn_ray_bbox_intersection(const float* bounds, const float* orig, const float* direct, int64_t n_rays,
                              float* near_out, float* far_out, uint8_t* hit_out, void* stream);

/* Ray set-up of batch_render_pifu_nerf, src/model.py:1019-1043, for the pixel grid
 * px = x0 + ix*step, py = y0 + iy*step (ix<nx, iy<ny; ray index = iy*nx + ix; eval branch :1019-1022
 * has step = 2^(level-1), (x0,y0) = stride offset (j,i)).  K,RT: target camera (4,4).
 * -> dirs (R,3), cam_pos (3), near (R), far (R) after the AABB clip. */
int kpn_make_rays(const float* K, const float* RT, float znear, float zfar, const float* bounds, int32_t x0,
                  int32_t y0, int32_t step, int32_t nx, int32_t ny, float* dirs, float* cam_pos, float* near_out,
                  float* far_out, void* stream);

/* KeypointNeRF.importance_sample, src/model.py:1110-1148.  contrib (R,D-2), z (R,D-1),
 * u (R,n) or NULL (uniform=True -> linspace(0,1,n)) -> samples (R,n). */
int kpn_importance_sample(const float* contrib, const float* z, const float* u, int64_t n_rays, int32_t d_minus_2,
                          int32_t n_samples, float* samples_out, void* stream);

/* KeypointNeRF.rgba2out, src/model.py:1150-1176.  rgba (R,S,5) [sigma,sdf,r,g,b], z (R,S) ->
 * color (R,3), depth (R), alpha (R), contrib (R,S) (may be NULL), sdf (R). */
int kpn_rgba2out(const float* rgba, const float* z, int64_t n_rays, int32_t n_samples, float* color, float* depth,
                 float* alpha, float* contrib, float* sdf, void* stream);

/* Backward of rgba2out (what torch autograd derives from src/model.py:1162-1174): given the forward inputs and the
 * upstream gradients of color (R,3), depth (R), alpha (R), sdf (R) (any may be NULL = zero), writes d_rgba (R,S,5).
 * z carries no gradient in the reference (sample positions are drawn under no_grad, :1038,1118).  The compositor piece of
 * the training backward; kpn_render_rays_train_backward chains it with the field reverse. */
int kpn_rgba2out_backward(const float* rgba, const float* z, int64_t n_rays, int32_t n_samples, const float* d_color,
                          const float* d_depth, const float* d_alpha, const float* d_sdf, float* d_rgba, void* stream);

/* Backward of the per-(point, source view) geometry rows: MLPUNet.layers1 232->128->128->(+8)120->64
 * (src/utils.py:691-716) and the bilinear gathers of feat_geo[0] / feat_geo[1] that feed it (src/model.py:763-765,
 * src/utils.py:74-89) — what autograd does for ~70 % of the field's arithmetic in `training_step`
 * (src/model.py:128-155).  The forward activations are recomputed (nothing is kept from the forward pass).
 *   pts (N,3); keep_mask = the train-time view dropout bits (bit v = 0: view v off, as in kpn_train_args);
 *   d_x (N, V, 64): d loss / d (layers1 output of point n in view v); rows of points that are masked
 *        (query's validity, src/model.py:725-739) are ignored;
 *   d_plain: += gradient w.r.t. the weight-norm-folded parameters, same flat layout as kpn_pack_weights' input
 *        (kpn_plain_weight_floats() floats; only the four layers1 blocks are touched);
 *   d_geo0 (V, geo0_h, geo0_w, 64), d_geo1 (V, geo1_h, geo1_w, 8): += gradient w.r.t. the feature maps,
 *        CHANNELS-LAST (permute to NCHW to hand it to the encoder's backward).
 * Accumulating: the caller zeroes the three outputs.  Float atomics: the summation order is not deterministic.
 * The layers1 piece of the training backward; pooling / layers2 / colour head: kpn_query_backward. */
size_t kpn_geo_rows_backward_workspace_bytes(int64_t n_points, int32_t n_views);
int kpn_geo_rows_backward(const kpn_scene_desc* desc, const void* scene_ws, const float* packed_weights, int64_t n_points,
                          const float* pts, uint32_t keep_mask, const float* d_x, float* d_plain, float* d_geo0,
                          float* d_geo1, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the field evaluation w.r.t. its two GEOMETRY outputs: query()'s [sdf_raw, rad] (mode 0) or eval_func's
 * [sigma = mask*relu(rad + noise), sdf] (mode 1, src/model.py:981-996) — i.e. loss.backward() of training_step
 * (src/model.py:128-155) through view pooling, MLPUNetFusion.layers2 (src/utils.py:500-518, 612-647, 722-748), and
 * then layers1 + the feat_geo gathers as kpn_geo_rows_backward.  d_out (N,5): columns 0,1 are used; the colour
 * columns 2..4 are ignored by THIS entry point (kpn_query_backward propagates all five).
 * Masked points contribute nothing in mode 1 (mask = 0); in mode 0 their constant layers2(0) output is not
 * differentiated either (training uses mode 1).  noise (N) / noise_std: the density noise added before the relu.
 * Outputs accumulate like kpn_geo_rows_backward's (d_plain: layers1 and layers2 blocks). */
size_t kpn_query_backward_geometry_workspace_bytes(int64_t n_points, int32_t n_views);
int kpn_query_backward_geometry(const kpn_scene_desc* desc, const void* scene_ws, const float* packed_weights,
                                int64_t n_points, const float* pts, int32_t mode, uint32_t keep_mask, const float* noise,
                                float noise_std, const float* d_out, float* d_plain, float* d_geo0, float* d_geo1,
                                void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the WHOLE field evaluation, colour head included: everything above plus
 * ibr_compress_gfeat, IBRRenderingHead (src/model.py:784-843, 1267-1302: ray_encoder, the ani_al blend weights,
 * weighted mean/var over views, base / vis / out layers, softmax blend of the source colours) and the feat_tex gather.
 * view (N,3): the query rays' directions.  d_out (N,5): all five columns are propagated.
 * d_plain: += all hot-path parameters incl. the raw ani_al (last entry); d_tex (V, tex_h, tex_w, 8) channels-last, +=. */
size_t kpn_query_backward_workspace_bytes(int64_t n_points, int32_t n_views);
int kpn_query_backward(const kpn_scene_desc* desc, const void* scene_ws, const float* packed_weights, int64_t n_points,
                       const float* pts, const float* view, int32_t mode, uint32_t keep_mask, const float* noise,
                       float noise_std, const float* d_out, float* d_plain, float* d_geo0, float* d_geo1, float* d_tex,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Rows kernel of the dominant stage (MLPUNet.layers1 per (point, view) row, reference src/utils.py:691-720), all with
 * fp32 accumulation and the same parity bar against the reference goldens:
 *   3 = v_mfma_f32_32x32x16_f16 with every fp32 operand carried as two fp16 pieces (x - fp16(x) is formed exactly by one
 *       v_fma_mix_f32) and three products per term set (hh hl lh: every term above 2^-24 relative); two 32-point tiles per
 *       wavefront, one wavefront per SIMD
 *       (k_geo_rows_f2).  The default.  Operands must stay within fp16's range (a pre-activation beyond 454 in natural units,
 *       a packed weight or a feature-map value beyond 65504 is not representable): the RANGE GUARD below sees to that.
 *   2 = v_mfma_f32_32x32x16_bf16, three bf16 pieces, six products (k_geo_rows_h2): the same structure in fp32's exponent range.
 *   0 = v_mfma_f32_32x32x2_f32 (fp32 operands, k_geo_rows).
 *   (1, an earlier one-tile-per-wavefront kernel, is refused: not part of the shipped library.)
 * Process-wide, default 3; one call may select another kernel through kpn_render_args.rows_kernel (no environment variable:
 * removed in round 5 — one selection mechanism). */
int kpn_set_geo_rows_mode(int32_t mode);
int kpn_get_geo_rows_mode(void);
/* The per-point kernel (MLPUNet.layers2 + ibr_compress_gfeat + IBRRenderingHead, reference src/utils.py:577-587,
 * src/model.py:819,1267-1302): 1 = weights as two fp16 pieces per value, three products (hh lh hl) on v_mfma_f32_32x32x16_f16
 * (k_fuse_color_h, the default: fp32-class results at less than a fifth of the matrix time); 0 = fp32 weights on
 * v_mfma_f32_32x32x2_f32 (k_fuse_color).  Process-wide, default 1; per call: kpn_render_args.fuse_kernel.  With V = 3 and no view
 * dropped mode 1 launches k_fuse_color_h3 (the same arithmetic, the loops over the views unrolled). */
int kpn_set_fuse_mode(int32_t mode);
int kpn_get_fuse_mode(void);
/* DENSITY FIRST (round 6) — reference src/model.py:981-996 (eval_func: sigma = relu(rad)) and :1150-1176 (rgba2out: a sample's
 * weight is T * (1 - exp(-sigma * delta))): a sample with relu(rad) == 0 contributes EXACTLY 0 to every output, whatever its colour.
 * The render passes (kpn_render_rays; fuse mode 1 on the pooled scratch layout, i.e. the defaults) can therefore evaluate the
 * per-point part in two passes per batch: k_density_h (pooled vector -> layers2 -> density, the compress layer, and a compact list
 * of the points with !(rad <= 0)), then k_row_records_live + k_colour_h / k_colour_h3 (gather records and the V-view colour head
 * for the listed points only).  Per point the arithmetic is the fused kernel's: frames are bit-identical either way.
 *   mode 0: never (the fused per-point kernel, whose short path needs a whole 32-point tile dead);
 *   mode 1: always;
 *   mode 2: auto, the default: per render pass, from the dead fraction the earlier passes measured on the device (>= 20 %: density
 *           first; the pair is 0.16 ms per launch slower than the fused kernel when every point is live and 3 % of a frame faster
 *           when 82 % are dead; no host synchronisation: DESIGN.md).
 * kpn_query, the train branch and the fp32-range kernels behind the range guard always use the fused kernel.  Process-wide;
 * KPN_DENSITY_FIRST=0/1/2 sets the initial value. */
int kpn_set_density_first(int32_t mode);
int kpn_get_density_first(void);
/* How many render passes (coarse / fine pass of a kpn_render_rays call) ran density first and how many on the fused kernel since
 * the library was loaded or the counts were reset (host-side counts; which form mode 2 chose). */
int kpn_density_first_passes(int64_t* density_first, int64_t* fused, int32_t reset);
/* Measurement hook: *listed = the points whose density the per-point kernels of the render passes decided on since the last reset
 * (every point inside the visual hull that was sent to the MLPs), *live = those with !(rad <= 0), i.e. the points whose colour can
 * reach the image; 1 - live / listed is the zero-density fraction bench.py reports.  Counted on the device (one addition per
 * wavefront), read here: synchronises `stream`.  reset != 0 clears the counters. */
int kpn_density_stats(void* stream, int64_t* listed, int64_t* live, int32_t reset);
/* *beyond = number of packed weights that fp16 cannot hold (0 = rows mode 3 / fuse mode 1 run on these weights; otherwise the
 * range guard routes every pass to the fp32-range kernels).  Reads four floats back from the device and synchronises `stream`. */
int kpn_packed_f16_range_check(const float* packed_weights_dev, void* stream, int32_t* beyond);
/* RANGE GUARD of the two-fp16-piece kernels (rows mode 3, fuse mode 1) — on by default, no host synchronisation involved:
 *   - what is known before a pass is tested ON THE DEVICE by the kernels themselves: the packers' count of weights beyond fp16's
 *     range (above) and max |value| of the source images / feature maps, which kpn_scene_prepare records in the scene workspace;
 *     when either is out of range the fp16 kernels return at once and the fp32-range kernels (rows mode 2, fuse mode 0),
 *     launched behind them for every batch of rows, do the work;
 *   - activations cannot be known before: an operand beyond fp16's range turns every accumulator it touches into a NaN, no
 *     activation of these kernels turns a NaN back into a number, and the per-point kernel flags a batch in which a point's
 *     [sdf, rad, rgb] is not finite; the fp32-range kernels then evaluate that batch again.
 * Results are never NaN where the reference's are finite.  The cost when nothing is out of range: two launches per batch that
 * return at once.  kpn_set_range_guard(0) (or the environment variable KPN_NO_RANGE_GUARD=1) switches it off for timing
 * comparisons.  kpn_range_guard_count: number of batches the fp32-range kernels evaluated again on the current device since
 * the library was loaded (0 = everything ran on the fp16 kernels); synchronises `stream`. */
int kpn_set_range_guard(int32_t on);
int kpn_get_range_guard(void);
int kpn_range_guard_count(void* stream, int64_t* batches_host);

/* KeypointNeRF.query (+ query_color + IBRRenderingHead), src/model.py:690-843,1239-1302, eval mode.
 * pts (N,3), view (N,3) -> out (N,5), valid (N).
 * mode 0: out = [sdf_raw, rad, r,g,b] exactly as query() returns;
 * mode 1: out = eval_func(query()) = [mask*relu(rad), mask*sdf+(1-mask)*0.1/nml_scale, r,g,b]
 *         (src/model.py:978-997, rand_noise_std = 0).
 * workspace: kpn_query_workspace_bytes(N, V) bytes. */
size_t kpn_query_workspace_bytes(int64_t n_points, int32_t n_views);
int kpn_query(const kpn_scene_desc* desc, const void* scene_ws, const float* packed_weights, int64_t n_points,
              const float* pts, const float* view, int32_t mode, float* out, uint8_t* valid, void* workspace,
              size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Whole path: KeypointNeRF.batch_render_pifu_nerf, eval branch (src/model.py:942-1108), and — with
 * step=1 over the full image — render_pifu_nerf's tile loop + pixel_shuffle (src/model.py:897-940),
 * which visits exactly the same rays.  Outputs are planar like the reference's out dict (B=1):
 * tex_fg (3,ny,nx), depth (ny,nx), alpha (ny,nx), and if fine: tex_fg_fine, depth_fine, alpha_fine,
 * sdf.  Any output pointer may be NULL. */
/* Optional per-sample outputs of a render call, in ray order (ray = iy * nx + ix) — what the reference holds in z_vals / raw
 * between its stages (src/model.py:1058-1085): the depths and eval_func'ed field values [sigma, sdf, r, g, b] of the coarse pass
 * and of the merged, sorted fine pass.  For the CONDITIONAL parity check of tests/parity_gate.py (each stage of the oracle run on
 * the kernel's own inputs); any pointer may be NULL.  rgb of a sample with sigma == 0 is unspecified (never composited). */
typedef struct kpn_render_stages {
    float* z_coarse;       /* (R, n_coarse) */
    float* rgba_coarse;    /* (R, n_coarse, 5) */
    float* z_fine;         /* (R, n_coarse + n_fine) */
    float* rgba_fine;      /* (R, n_coarse + n_fine, 5) */
    float* dirs;           /* (R, 3) the rays' unit directions and */
    float* cam_pos;        /* (3) their common origin, as the call's ray set-up formed them (src/model.py:1019-1043) */
} kpn_render_stages;
typedef struct kpn_render_args {
    const float* K;          /* cam_tar["K"]  (4,4) */
    const float* RT;         /* cam_tar["RT"] (4,4) */
    const float* bounds;     /* (2,3) */
    float znear, zfar;       /* cam_tar znear/zfar */
    int32_t x0, y0, step, nx, ny;
    int32_t n_coarse, n_fine;   /* sample_per_ray_c / _f */
    int32_t fine;               /* dr_kwargs.fine */
    int32_t chunk_rays;         /* rays per internal pass (0 = default) */
    float* tex_fg; float* depth; float* alpha;
    float* tex_fg_fine; float* depth_fine; float* alpha_fine; float* sdf;
    int32_t step_y;             /* 0: rows advance by `step` like the columns (the reference's grids); > 0: py = y0 + iy*step_y —
                                 * one frame's rows dealt round-robin to the ranks of a render job (rank r of W: y0 = r,
                                 * step_y = W), SURVEY 8(e) */
    int32_t rows_kernel;        /* KPN_ROWS_*: the rows kernel of THIS call; 0 = the process-wide selection (kpn_set_geo_rows_mode) */
    int32_t fuse_kernel;        /* KPN_FUSE_*: the per-point kernel of THIS call; 0 = the process-wide selection (kpn_set_fuse_mode).
                                 * kpn_render_rays only: the kpn_render_rays_train* entry points (forward, kept state, backward
                                 * recompute must run the same kernels) refuse a non-zero selection with KPN_EINVAL */
    const kpn_render_stages* stages;   /* NULL (the default): nothing but the images leaves the call (ABI 3) */
} kpn_render_args;
/* per-call kernel selection (kpn_render_args): the same kernels as kpn_set_geo_rows_mode(0 / 2 / 3) and kpn_set_fuse_mode(0 / 1) */
enum { KPN_ROWS_DEFAULT = 0, KPN_ROWS_F32 = 1, KPN_ROWS_BF16X3 = 2, KPN_ROWS_F16X2 = 3 };
enum { KPN_FUSE_DEFAULT = 0, KPN_FUSE_F32 = 1, KPN_FUSE_F16X2 = 2 };

/* Note on the fine pass: z_fine = sort(cat(z_coarse, z_new)) (src/model.py:1076) repeats the coarse samples; their field
 * values are taken from the coarse pass and the field is evaluated at the new samples only — identical points, identical
 * deterministic values, outputs bit-identical to evaluating all of them again (environment variable
 * KPN_NO_COARSE_REUSE=1 does that, for comparison).  kpn_render_rays_train evaluates everything (fresh dropout / noise). */
size_t kpn_render_workspace_bytes(const kpn_scene_desc* desc, const kpn_render_args* args);
int kpn_render_rays(const kpn_scene_desc* desc, const void* scene_ws, const float* packed_weights,
                    const kpn_render_args* args, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SURVEY.md §8(f) "next" rows on the output side of the path (callers of the renderer).
 *
 * kpn_frame_to_rgb8: KeypointNeRFLightningModule._arrange_nerf_images (src/model.py:427-430: clamp to [0,1],
 * CHW -> HWC) + the uint8 quantisation `(img*255.).astype(np.uint8)` of render_novel_views / save_test_image
 * (src/model.py:496,504,281) + the channel flip handed to cv2.imwrite (`[:, :, ::-1]`, src/model.py:222,283).
 * chw (3,H,W) fp32 -> hwc_out (H,W,3) uint8; bgr != 0 writes B,G,R.  Integer result is bit-exact.
 *
 * kpn_mse_psnr: ZJUEvaluator.compute_score's mse / _compute_psnr (src/zju_evaluator.py:16-19,63-64):
 * mse = mean((pred-gt)^2) over all n elements, psnr = -10 ln(mse)/ln(10).  out2 (device, 2 doubles) = [mse, psnr];
 * partial sums are accumulated in fp64.  scratch: 2048 doubles + 1 int (device, 16,392 bytes).
 *
 * kpn_pix_l1_loss: the L1 terms of the training loss - pix_loss(src, tar, {"l1": lambda}) (src/utils.py:164-168) as
 * compute_error_nerf applies it to tex_cal = tex_fg (lambda_l1_c) and tex_cal_fine = tex_fg_fine (lambda_l1), src/utils.py:
 * 128-145, configs/zju.json:109-112 - together with its gradient: loss[0] (device) = lambda * mean|src - tar| over n
 * elements; d_src (n floats, may be NULL) = lambda * sign(src - tar) / n, i.e. what loss.backward() hands to the renderer
 * (the d_tex_fg / d_tex_fg_fine of kpn_render_grads).  scratch: 16,392 bytes as kpn_mse_psnr.  Deterministic. */
int kpn_pix_l1_loss(const float* src, const float* tar, int64_t n, float lambda, float* loss, float* d_src, void* scratch,
                    void* stream);
int kpn_frame_to_rgb8(const float* chw, int32_t height, int32_t width, int32_t bgr, uint8_t* hwc_out, void* stream);
int kpn_mse_psnr(const float* pred, const float* gt, int64_t n, double* out2, void* scratch, void* stream);
/* kpn_ssim: ZJUEvaluator._compute_ssim (src/zju_evaluator.py:21-45) = skimage 0.19 (environment.yml:135)
 * structural_similarity(pred, gt, multichannel=True) on the crop [y0, y0+h) x [x0, x0+w) (cv2.boundingRect of
 * mask_at_box, computed by the caller) of two (3,H,W) fp32 images: 7x7 uniform window, sample covariance, data_range 2
 * (skimage's default for float images), mean over the interior and the channels.  out (device, 1 double).
 * skimage is not installed here: the oracle restates its published algorithm with scipy.ndimage.uniform_filter (the
 * filter skimage itself calls); parity against skimage proper is unpinned. */
size_t kpn_ssim_scratch_bytes(int32_t crop_w, int32_t crop_h);
int kpn_ssim(const float* pred_chw, const float* gt_chw, int32_t height, int32_t width, int32_t x0, int32_t y0, int32_t crop_w,
             int32_t crop_h, double* out, void* scratch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py).  When enabled, every launch of the dominant kernel (k_geo_rows) is
 * bracketed by HIP events recorded on the caller's stream and the number of (point,view) rows it
 * processed is copied back asynchronously.  kpn_profile_collect synchronises the recorded events
 * and returns the totals since kpn_profile_enable(1).  Diagnostic only; off by default. */
int kpn_profile_enable(int32_t on);
int kpn_profile_collect(double* geo_rows_ms_host, int64_t* launches_host, int64_t* rows_host);
/* The row scratch between the two field kernels is capped (kpn_row_scratch_cap_bytes(): 3 GiB by default, environment
 * variable KPN_ROW_SCRATCH_MIB); a pass with more valid (point, view) rows than fit is evaluated in batches that reuse it.
 * The launcher cannot know the number of batches without a sync, so it launches the worst-case number and the surplus
 * launches return at once: kpn_profile_collect2 reports them separately (ms / launches / rows cover the launches that
 * processed rows). */
int kpn_profile_collect2(double* geo_rows_ms_host, int64_t* launches_host, int64_t* rows_host, int64_t* surplus_launches_host);
/* The same plus *clock_ghz_host = the shader clock the chip sustained under the recorded pair-tile rows-kernel launches: shader
 * cycles (s_memtime) between the first workgroup's entry and its last work item, divided by the launches' HIP-event time (a lower
 * bound: the workgroup finishes a little before its launch).  The MI355X clocks to its power budget (≈ 1.9 GHz under this load,
 * 2.4 GHz peak): a roofline fraction against the 2.4 GHz peak understates what the kernel does per cycle.  0 if not measured. */
int kpn_profile_collect3(double* geo_rows_ms_host, int64_t* launches_host, int64_t* rows_host, int64_t* surplus_launches_host,
                         double* clock_ghz_host);
/* The same for the BACKWARD (kpn_query_backward, kpn_render_rays_train_backward[_kept]): HIP events around each kernel group of
 * every pass while enabled.  kpn_bwd_profile_collect: ms5 / launches5 = time and launch groups of [0] k_geo_rows (the forward's rows,
 * only when the pass recomputes them), [1] k_color_bwd, [2] k_fuse_bwd, [3] k_geo_rows_bwd, [4] k_weight_grad<*> + reduce;
 * *rows = valid points x V of the recorded passes (the rows k_geo_rows_bwd and k_weight_grad process), *kept_rows = valid points x
 * views kept by the pass's dropout mask (the rows k_color_bwd processes), *points = valid points.  Synchronises the device. */
int kpn_bwd_profile_enable(int32_t on);
int kpn_bwd_profile_collect(double* ms5, int64_t* launches5, int64_t* rows, int64_t* kept_rows, int64_t* points);
size_t kpn_row_scratch_cap_bytes(void);
/* Process-wide; workspace sizes queried before a change are stale (query kpn_*_workspace_bytes again). */
int kpn_set_row_scratch_cap_bytes(size_t bytes);

/* TRAIN branch of batch_render_pifu_nerf (forward only), every random draw supplied by the caller so that it can
 * be the reference's own: src/model.py:1008-1017 (patch pixels), :1049-1053 (stratified jitter), :993-994 (density
 * noise, coarse and fine eval), :742-748 (per-view dropout of the coarse and of the fine query), :1129 (random
 * importance samples).  `args` as for kpn_render_rays with nx*ny = n rays (x0,y0,step unused); outputs are (C,ny,nx)
 * planar in patch order.  Backward: kpn_render_rays_train_backward. */
typedef struct kpn_train_args {
    const int32_t* pix;          /* (R,2) target pixels (x,y) */
    const float* u_coarse;       /* (R,Sc)        th.rand_like(z) */
    const float* noise_coarse;   /* (R*Sc)        th.randn_like(rad) of the coarse eval_func; NULL iff std == 0 */
    const float* noise_fine;     /* (R*(Sc+Sf))   ... of the fine eval_func */
    const float* u_fine;         /* (R,Sf)        th.rand of importance_sample */
    uint32_t keep_coarse;        /* bit v = source view v kept by the coarse query's dropout */
    uint32_t keep_fine;          /* ... by the fine query's dropout */
    float rand_noise_std;        /* dr_kwargs.rand_noise_std (0.01) */
} kpn_train_args;
int kpn_render_rays_train(const kpn_scene_desc* desc, const void* scene_ws, const float* packed_weights,
                          const kpn_render_args* args, const kpn_train_args* train, void* workspace,
                          size_t workspace_bytes, void* stream);

/* loss.backward() through the train branch (the field part of training_step, src/model.py:128-155): given the
 * gradients of kpn_render_rays_train's seven outputs — planar (C, ny*nx) like the outputs, NULL = zero — accumulate the
 * gradients of every hot-path parameter (d_plain, flat layout of kpn_pack_weights' input, raw ani_al last) and of the
 * three feature maps (channels-last, like kpn_query_backward).  Same `args` / `train` as the forward call (outputs in
 * `args` are ignored).  Nothing is kept from the forward pass: z, rgba and the field activations are recomputed per
 * pass; the sample positions carry no gradient (drawn under no_grad, src/model.py:1038,1118). */
typedef struct kpn_render_grads {
    const float* d_tex_fg; const float* d_depth; const float* d_alpha;
    const float* d_tex_fg_fine; const float* d_depth_fine; const float* d_alpha_fine; const float* d_sdf;
} kpn_render_grads;
size_t kpn_render_rays_train_backward_workspace_bytes(const kpn_scene_desc* desc, const kpn_render_args* args);
int kpn_render_rays_train_backward(const kpn_scene_desc* desc, const void* scene_ws, const float* packed_weights,
                                   const kpn_render_args* args, const kpn_train_args* train, const kpn_render_grads* grads,
                                   float* d_plain, float* d_geo0, float* d_geo1, float* d_tex, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* The same pair with the forward's work KEPT for the backward: kpn_render_rays_train_keep is kpn_render_rays_train (same
 * outputs, bit for bit) that also fills `state` (kpn_render_rays_train_state_bytes: rays, depths, field values and the
 * valid lists + (point, view) rows of both field passes, per chunk of rays — about 1 KB per field evaluation);
 * kpn_render_rays_train_backward_kept is kpn_render_rays_train_backward (same gradients up to summation order: the
 * order of the valid list varies from call to call) that reads them from `state` instead of repeating the forward — the field part of a
 * training iteration then runs the forward once (reference: autograd keeps every activation; here only the pass state is
 * kept, the MLP activations are still recomputed tile by tile inside the backward kernels).  `workspace` is the one of
 * kpn_render_rays_train_backward_workspace_bytes.  The state is valid until the weights, the scene or the draws change. */
size_t kpn_render_rays_train_state_bytes(const kpn_scene_desc* desc, const kpn_render_args* args);
int kpn_render_rays_train_keep(const kpn_scene_desc* desc, const void* scene_ws, const float* packed_weights,
                               const kpn_render_args* args, const kpn_train_args* train, void* state, size_t state_bytes,
                               void* stream);
int kpn_render_rays_train_backward_kept(const kpn_scene_desc* desc, const void* scene_ws, const float* packed_weights,
                                        const kpn_render_args* args, const kpn_train_args* train, const kpn_render_grads* grads,
                                        float* d_plain, float* d_geo0, float* d_geo1, float* d_tex, void* state,
                                        size_t state_bytes, void* workspace, size_t workspace_bytes, void* stream);

/* FLOP / byte model of one field evaluation, for roofline reporting */
double kpn_flops_per_point(int32_t n_views);
/* algorithmic FLOPs of one k_geo_rows row = 2 * 70,080 MACs (MLPUNet layers1) */
double kpn_flops_per_row(void);

/* Device self-test of the MFMA operand/result lane maps the kernels assume (asymmetric operands).
 * Returns 0 if D == A*B for v_mfma_f32_32x32x2_f32 in the assumed layout. */
int kpn_selftest_mfma(float* scratch_256k, void* stream, float* max_err_host);

#ifdef __cplusplus
}
#endif
#endif /* KPNERF_H */
