/*
 * nerfdet_hip.h -- C ABI of libnerfdet_hip.so: the MI355X (gfx950) implementation of the
 * NeRF-Det volumetric hot path.
 *
 * The reference has no FFI layer on this path (it is pure PyTorch Python); the drop-in boundary
 * is the set of module-level Python callables listed in SURVEY.md section 8(b).  Every entry
 * point below states which reference callable (file:line under the reference tree) it replaces.
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the parameter name ends in _host;
 *   - fp32 everywhere on the data path, int64 for counts / indices (as the reference);
 *   - caller allocates every output; the library allocates nothing and keeps no state
 *     between calls (re-entrant; one process per GPU);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); work is only enqueued,
 *     never synchronised;
 *   - return value: 0 on success, negative NDET_E_* otherwise; ndet_last_error() gives a
 *     thread-local message.  The Python mirror turns these into the exceptions the reference
 *     raises (AssertionError / ValueError).
 */
#ifndef NERFDET_HIP_H
#define NERFDET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDET_OK 0
#define NDET_E_INVALID (-1)     /* bad argument (null pointer, non-positive size) */
#define NDET_E_UNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define NDET_E_LAUNCH (-3)      /* hipLaunch / runtime failure */

#define NDET_LAYOUT_CN 0 /* (C, N): the reference's (C,X,Y,Z) contiguous layout */
#define NDET_LAYOUT_NC 1 /* (N, C): channels-last (NDHWC), the native layout of this library */

int ndet_version(void);
const char* ndet_last_error(void);

/* A2. Voxel lower-corner lattice. Replaces get_points(), mmdet3d/models/detectors/nerfdet.py:380-390.
 * points: (3, nx*ny*nz) fp32, z fastest.  voxel_size_host/origin_host: 3 floats each, HOST memory.
 * Same fp32 operation order as the reference: idx*vs + (origin - n/2*vs), un-fused. */
int ndet_get_points(float* points, int nx, int ny, int nz, const float* voxel_size_host,
                    const float* origin_host, void* stream);

/* Layout helper: (n, c, hw) -> (n, hw, c).  The reference keeps feature maps NCHW; the fused kernels
 * want one pixel's channels contiguous (SURVEY.md section 7 "Layout").  No reference counterpart. */
int ndet_nchw_to_nhwc(const float* src, float* dst, int n, int c, int hw, void* stream);

/* Measurement aid: dst[i] = src[i], one non-temporal 16-byte access per thread.  Its rate (2 * 4 * n_floats / time) is the
 * empirical HBM ceiling bench.py prices the gather kernels against, next to the 8 TB/s specification (SURVEY.md 8d "the builder must
 * also report an empirical copy-kernel ceiling"; the reference's harness has no counterpart, tools/benchmark.py:63-89 times the model only).
 * Pointers 16-byte aligned, n_floats % 4 == 0. */
int ndet_hbm_copy(const float* src, float* dst, int64_t n_floats, void* stream);

/* A3 (exact API form). Replaces backproject(), nerfdet.py:393-420 (depth=None).
 * features: element (v,c,y,x) at v*sv + c*sc + y*sy + x*sx (floats) -- any layout.
 * points (3,N); projection (n_views,3,4).  Outputs the reference's materialised tensors:
 * volume (n_views, C, N) fp32 (zero where invalid) and valid (n_views, N) uint8 0/1. */
int ndet_backproject(const float* features, int n_views, int C, int h, int w,
                     int64_t sv, int64_t sc, int64_t sy, int64_t sx,
                     const float* points, int N, const float* projection,
                     float* volume, uint8_t* valid, void* stream);

/* A3+A4 (+A6 gating) fused -- the inference hot kernel.  Replaces backproject() followed by the view
 * aggregation of nerfdet.py:171-176 and, when alpha != NULL, the gating of nerfdet.py:259-261, without
 * materialising the (n_views,C,N) volume.
 * features_nhwc: element (v,y,x,c) at v*view_pitch + y*row_pitch + x*C + c (floats); C % 4 == 0, C <= 1024.
 * alpha: NULL or (N) fp32.  out: mean (or alpha*mean) in `out_layout`; count: (N) int64 view count. */
int ndet_backproject_aggregate(const float* features_nhwc, int n_views, int C, int h, int w,
                               int64_t view_pitch, int64_t row_pitch,
                               const float* points, int N, const float* projection,
                               const float* alpha, float* out, int out_layout, int64_t* count,
                               void* stream);

/* A5. Per-voxel NeRF conditioning vector.  Replaces nerfdet.py:234-253 (image mode) with the algebraic
 * identity mapping(volume)[v] == mapped_map[v,y,x] where the view sees the voxel and == bias where it
 * does not.
 * mapped_nhwc: Linear(C->cm)(features), element (v,y,x,c) at v*mview_pitch + y*mrow_pitch + x*cm + c; cm <= 61.
 * bias: (cm) the Linear's bias.  rgb: de-normalised images, element (v,c,y,x) at v*rsv + c*rsc + y*rsy + x
 * (3 channels, H x W).  projection: stride-4 matrices (validity + count), rgb_projection: stride-1 matrices.
 * global_feat: (N, 2*(3+cm)) fp32, channels INTERLEAVED [mean_0,cov_0,mean_1,cov_1,...] with channel order
 * [rgb0,rgb1,rgb2,m0..m(cm-1)] -- the effective layout of the reference's cat(dim=1)+view (nerfdet.py:251-253). */
int ndet_density_features(const float* mapped_nhwc, int n_views, int cm, int h, int w,
                          int64_t mview_pitch, int64_t mrow_pitch, const float* bias,
                          const float* rgb, int H, int W, int64_t rsv, int64_t rsc, int64_t rsy,
                          const float* points, int N, const float* projection, const float* rgb_projection,
                          float* global_feat, void* stream);

/* A5, packed form (the inference path's kernel): same statement, inputs and output layout as ndet_density_features --
 * nerfdet.py:234-253 -- for cm % 4 == 0 and n_views <= 128: channel quads per lane, 64/(cm/4+1) voxels per wavefront, both
 * projections of a (voxel, view) pair evaluated once, only the views that see the voxel are walked, and the variance uses the
 * shifted one-pass form (equal to the reference's two-pass sum up to fp32 rounding, not bit-equal). */
int ndet_density_features_packed(const float* mapped_nhwc, int n_views, int cm, int h, int w,
                                 int64_t mview_pitch, int64_t mrow_pitch, const float* bias,
                                 const float* rgb, int H, int W, int64_t rsv, int64_t rsc, int64_t rsy,
                                 const float* points, int N, const float* projection,
                                 const float* rgb_projection, float* global_feat, void* stream);

/* A6 (gating only, unfused form). volume = (1-exp(-density)) * mean, 0 where count==0; nerfdet.py:257-261.
 * mean/out in `layout` with C channels. */
int ndet_alpha_gate(const float* mean, const float* density, const int64_t* count, float* out,
                    int C, int N, int layout, void* stream);

/* alpha = 1 - exp(-relu(raw_sigma)) for N voxels (nerf_mlp.py:227 + nerfdet.py:257). */
int ndet_sigma_to_alpha(const float* raw_sigma, float* alpha, int N, void* stream);

/* A6 input assembly: rows [posenc(xyz) (63) | global (F) | zero padding] for the sigma-MLP, nerf_mlp.py:181-197,140.
 * points (3,N) SoA; out (N, out_stride), out_stride >= 63+F (a multiple of 32 when the rows feed the MFMA kernel). */
int ndet_posenc_concat(const float* points, const float* global_feat, int N, int F, int out_stride, float* out,
                       void* stream);

/* A6 tail. sigma = w . [h | x] + b on the re-joined trunk output without materialising the concat (nerf_mlp.py:86,143),
 * alpha = 1 - exp(-relu(sigma)) (nerf_mlp.py:227, nerfdet.py:257).  h (N,Ch); x (N, x_stride) of which the first Cx columns
 * are used; w (Ch+Cx); bias: DEVICE scalar; raw_sigma (N) or NULL; alpha (N). */
int ndet_sigma_head(const float* h, int Ch, const float* x, int Cx, int x_stride, const float* w, const float* bias,
                    int N, float* raw_sigma, float* alpha, void* stream);

/* A15. Greedy class-aware axis-aligned 3D NMS. Replaces aligned_3d_nms(),
 * mmdet3d/core/post_processing/box3d_nms.py:91-138 (pinned by the reference's tests/test_nms.py:5-58).
 * boxes (n,6) x1,y1,z1,x2,y2,z2; scores (n); classes (n) int64; n <= 4096.
 * keep: (n) int64, receives the picked ORIGINAL indices in pick order (highest score first);
 * n_keep: (1) int64 device scalar.  workspace: ndet_nms_workspace_bytes(n) bytes of device memory.
 * Score ties (undefined in the reference: torch.argsort is unstable) are taken higher index first. */
int64_t ndet_nms_workspace_bytes(int n);
int ndet_aligned_3d_nms(const float* boxes, const float* scores, const int64_t* classes, int n, float thresh,
                        int64_t* keep, int64_t* n_keep, void* workspace, void* stream);

/* A14 decode. One pass per head level over the fused head convolution output raw (N, 7 + n_cls) =
 * [centerness logit | 6 reg | n_cls class logits]: best class score = max_k sigmoid(cls_k)*sigmoid(centerness)*valid, its label,
 * and the decoded box (x1,y1,z1,x2,y2,z2) = voxel corner -/+ exp(scale*reg).  Replaces the elementwise chain of
 * mmdet3d/models/dense_heads/imvoxel_head_v2.py:262-271 + :447 + :547-555 (top-k, thresholding and NMS follow).
 * valid (N) uint8; scale: DEVICE pointer to the level's learnable scalar; voxel_size_host already multiplied by 2^level. */
int ndet_head_decode(const float* raw, int n_cls, const uint8_t* valid, const float* scale, int nx, int ny, int nz,
                     const float* voxel_size_host, const float* origin_host, float* best, int64_t* label, float* boxes,
                     void* stream);

/* ndet_level_valid + ndet_head_decode for up to four head levels in ONE launch (imvoxel_head_v2.py:262-271,442-449,547-555; the tail of the
 * one-scene loop is a chain of tiny dependent kernels).  HOST arrays of per-level DEVICE pointers (raw, scale, best, label, boxes), HOST dims
 * (3 ints per level), integer down-scale factors against the (X, Y, Z) float validity volume `valid` (1 or even, dims * factor == X, Y, Z), HOST
 * voxel sizes (3 floats per level, already times 2^level) and origin.  Same arithmetic, level by level, as the two separate calls. */
int ndet_head_decode_levels(int n_levels, const float* const* raw_host, const float* const* scale_host, const int* dims_host,
                            const int* factor_host, const float* voxel_size_host, const float* origin_host, int n_cls, const float* valid, int X,
                            int Y, int Z, float* const* best_host, int64_t* const* label_host, float* const* boxes_host, void* stream);

/* valid mask of an FPN level: F.interpolate(valid, size, mode='trilinear').round().bool() of
 * mmdet3d/models/dense_heads/imvoxel_head_v2.py:442-449 for the integer down-scale `factor` (1 or even) of the level:
 * valid (X,Y,Z) float view counts -> out (X/f,Y/f,Z/f) uint8. */
int ndet_level_valid(const float* valid, int X, int Y, int Z, int factor, uint8_t* out, void* stream);

/* `scores > score_thr` over the concatenated levels (imvoxel_head_v2.py:533-545) as one order-preserving compaction:
 * best/label/boxes are HOST arrays of n_levels device pointers (per-level outputs of ndet_head_decode), n the level sizes;
 * survivors go to out_* in level-then-voxel order, counts (device, n_levels + 1 ints) = per level, then the total. */
int ndet_select_candidates(int n_levels, const float* const* best, const int64_t* const* label, const float* const* boxes,
                           const int* n, float score_thr, float* out_best, int64_t* out_label, float* out_boxes, int* counts,
                           void* stream);

/* The same compaction with the per-level top-``nms_pre`` cut of dense_heads/imvoxel_head_v2.py:272-276 decided on the device (radix
 * select over the score bits): the kept set per level is what ``topk(nms_pre)`` followed by ``scores > score_thr`` keeps.
 * counts (n_levels + 2 ints): per level after the cut, total, survivors before the cut. */
int ndet_select_candidates_topk(int n_levels, const float* const* best, const int64_t* const* label, const float* const* boxes,
                                const int* n, float score_thr, int nms_pre, float* out_best, int64_t* out_label, float* out_boxes,
                                int* counts, void* stream);

/* picked candidates -> detections in pick order: (centre, size) boxes, scores, labels (imvoxel_head_v2.py:546-555). */
int ndet_gather_detections(const int64_t* keep, int n_keep, const float* boxes, const float* scores, const int64_t* labels,
                           float* out_boxes, float* out_scores, int64_t* out_labels, void* stream);

/* The tail of get_bboxes without a host round trip (dense_heads/imvoxel_head_v2.py:216-285,528-555 + core/bbox/transforms.py:49-67):
 * greedy NMS (core/post_processing/box3d_nms.py:91-138) over the candidates ndet_select_candidates compacted -- their count is read
 * on the device from counts[n_levels] -- and the picks packed for ONE device-to-host copy: out_packed = {n_keep, n_candidates,
 * status, range guard} followed by up to k_cap rows [x, y, z_bottom, dx, dy, dz, 0, score, label]; range_guard (may be null): the scene's guard word
 * of ndet_conv_ndhwc_guarded, whose bit 0 is copied into the header so that it reaches the host with the picks.  status != 0 (more than n_cap <= 4096
 * candidates, a level with more than nms_pre survivors, more picks than rows): the caller repeats the scene on the synchronous
 * path.  workspace: ndet_nms_workspace_bytes(n_cap). */
int ndet_nms_pack_detections(const float* cand_boxes, const float* cand_scores, const int64_t* cand_labels, const int* counts,
                             int n_levels, int nms_pre, int n_cap, float thresh, int64_t* keep, int64_t* n_keep, void* workspace,
                             float* out_packed, int k_cap, const unsigned* range_guard, void* stream);

/* A9. Samples along rays. Replaces sample_along_camera_ray(), mmdet3d/models/model_utils/render_ray.py:145-189
 * (inv_uniform=False).  ray_o, ray_d (R,3); t_rand NULL (det=True) or (R,S) uniforms in [0,1) -- the stream the
 * reference draws with torch.rand_like, injectable for parity.  Outputs pts (R,S,3), z_vals (R,S). */
int ndet_sample_along_rays(const float* ray_o, const float* ray_d, int R, int S, float near, float far,
                           const float* t_rand, float* pts, float* z_vals, void* stream);

/* A7+A8 fused. Replaces Projector.compute() (model_utils/projection.py:91-151, grid_sample=True) followed by
 * compute_mask_points() (render_ray.py:71-93) and the concat / pixel mask of render_ray.py:301-303, without
 * materialising the (R,S,n_views,3+d) tensor.
 * pts (P,3) sample points; KE (n_views,3,4) = rows 0..2 of K(4x4) @ E(4x4) per view (render_ray.py:48-69 cameras);
 * img_h,img_w: cameras[:, :2]; rgb: source images, element (v,c,y,x) at v*rsv + c*rsc + y*rsy + x, H x W;
 * feat_nhwc: mapped features, element (v,y,x,c) at v*fview_pitch + y*frow_pitch + x*d + c, hf x wf, d <= 61.
 * global_feat (P, 2*(3+d)) = [mean(3+d) | exp(-var)(3+d)]; pixel_mask (P) uint8 = (views seeing the point > 1);
 * view_count (P) int32 or NULL. */
int ndet_ray_view_stats(const float* pts, int n_points, const float* KE, int n_views, float img_h, float img_w,
                        const float* rgb, int H, int W, int64_t rsv, int64_t rsc, int64_t rsy,
                        const float* feat_nhwc, int d, int hf, int wf, int64_t fview_pitch, int64_t frow_pitch,
                        float* global_feat, uint8_t* pixel_mask, int* view_count, void* stream);

/* Source images as (n_views,H,W,4) fp32 (4th component 0) for the packed sampler below: one image pixel = one 16-byte load.
 * rgb: element (v,c,y,x) at v*rsv + c*rsc + y*rsy + x -- the (n_v,3,H,W) ``denorm_images`` the reference permutes to
 * (1,n_v,H,W,3) before Projector.compute (model_utils/render_ray.py:296-299).  Once per scene. */
int ndet_pack_rgb_nhwc4(const float* rgb, int n_views, int H, int W, int64_t rsv, int64_t rsc, int64_t rsy, float* out_nhwc4,
                        void* stream);

/* A7+A8 fused, packed form (the training / render_testing hot kernel): same outputs as ndet_ray_view_stats -- Projector.compute()
 * (model_utils/projection.py:91-151) + compute_mask_points() (render_ray.py:71-93) + the concat / pixel mask of render_ray.py:301-303 --
 * for d % 4 == 0 and n_views <= 128.  rgb_nhwc4 from ndet_pack_rgb_nhwc4.  Each (sample, view) is projected once; views without a
 * bilinear tap inside a map are counted, not gathered (they sample exactly 0); the variance uses the shifted one-pass form
 * (equal to the reference's two-pass sum up to fp32 rounding, not bit-equal). */
int ndet_ray_view_stats_packed(const float* pts, int n_points, const float* KE, int n_views, float img_h, float img_w,
                               const float* rgb_nhwc4, int H, int W, const float* feat_nhwc, int d, int hf, int wf,
                               int64_t fview_pitch, int64_t frow_pitch, float* global_feat, uint8_t* pixel_mask, int* view_count,
                               void* stream);

/* A7 exact API form. Replaces Projector.compute(), projection.py:91-151: rgb_feat (P, n_views, 3+d) and
 * mask (P, n_views) fp32 0/1, materialised like the reference. Same inputs as ndet_ray_view_stats. */
int ndet_project_sample(const float* pts, int n_points, const float* KE, int n_views, float img_h, float img_w,
                        const float* rgb, int H, int W, int64_t rsv, int64_t rsc, int64_t rsy,
                        const float* feat_nhwc, int d, int hf, int wf, int64_t fview_pitch, int64_t frow_pitch,
                        float* rgb_feat, float* mask, void* stream);

/* A11. Alpha compositing. Replaces raw2outputs(), render_ray.py:196-247.
 * raw (R,S,4) = [rgb, sigma]; z_vals (R,S); pixel_mask (R,S) uint8 or NULL; zminmax: DEVICE pointer to
 * {z_vals.min(), z_vals.max()} (the depth clamp is global, render_ray.py:236).
 * Outputs rgb_map (R,3), depth_map (R), weights/alpha/transparency (R,S), ray_mask (R) uint8 (NULL if no pixel_mask). */
int ndet_composite(const float* raw, const float* z_vals, const uint8_t* pixel_mask, int R, int S, int white_bkgd,
                   const float* zminmax, float* rgb_map, float* depth_map, float* weights, uint8_t* ray_mask,
                   float* alpha, float* transparency, void* stream);

/* A13/A14. 3D convolution, channels-last, fp32 on the matrix cores, fused epilogue.  Replaces nn.Conv3d /
 * nn.ConvTranspose3d(2,2) + BatchNorm3d(eval) + ReLU (+ residual add) of mmdet3d/models/necks/imvoxelnet.py:8-67,
 * 233-260 and the head convs of mmdet3d/models/dense_heads/imvoxel_head_v2.py:45-49,442-449.
 * in (D,H,W,Cin) fp32, Cin % 32 == 0; w_packed (taps, Cout, Cin) with tap = (kd*k + kh)*k + kw
 * [transposed: (8, Cout, Cin), tap = kd*4 + kh*2 + kw, out voxel = 2*in + (kd,kh,kw)];
 * out (OD,OH,OW,Cout), O = (I + 2*(k/2) - k)/stride + 1 [transposed: 2*I].  ksize 3 or 1, stride 1 or 2.
 * Epilogue: v = acc*scale[co] + shift[co] (if scale != NULL); relu == 2: max(v,0); v += residual (if != NULL);
 * relu == 1: max(v,0).  splits > 1: split-K over blockIdx.z into `workspace` (ndet_conv3d_workspace_bytes) and a
 * fixed-order reduction (bitwise reproducible).  tile: 0 auto, 64 or 128. */
int64_t ndet_conv3d_workspace_bytes(int D, int H, int W, int Cin, int Cout, int ksize, int stride, int splits);
int ndet_conv3d_ndhwc(const float* in, const float* w_packed, float* out, int D, int H, int W, int Cin, int Cout,
                      int ksize, int stride, int transposed, const float* scale, const float* shift,
                      const float* residual, int relu, int splits, int tile, void* workspace, void* stream);

/* Generic form of the same kernel: per-axis kernel / stride / zero-pad (3 ints each, HOST memory, order D,H,W).
 * A batch of 2D feature maps (N,H,W,C) is the case D = N, kernel[0] = 1, pad[0] = 0 -- used for the ResNet/FPN convolutions
 * (conv + eval-BatchNorm + ReLU + residual in one pass; third-party mmdet ResNet/FPN, SURVEY.md appendix C, called at
 * mmdet3d/models/detectors/nerfdet.py:140-142).  w_packed (kd*kh*kw, Cout, Cin).  residual_up2 != 0: `residual` is the
 * coarser (OD, ceil(OH/2), ceil(OW/2), Cout) map and is added at (d, h>>1, w>>1) -- FPN's nearest-x2 top-down add fused
 * into the lateral convolution (not combinable with split-K).  Other arguments as ndet_conv3d_ndhwc;
 * split-K workspace = splits * OD*OH*OW*Cout * 4 bytes. */
int ndet_conv_ndhwc(const float* in, const float* w_packed, float* out, int D, int H, int W, int Cin, int Cout,
                    const int* kernel_host, const int* stride_host, const int* pad_host, const float* scale,
                    const float* shift, const float* residual, int residual_up2, int relu, int splits, int tile,
                    void* workspace, void* stream);

/* Packed fp32 weights (taps, Cout, Cin) -> three bf16 planes tiled per 32-channel K step, (taps, Cin/32, 3, Cout, 32), with
 * w = p0 + p1 + p2 exactly (p0 = bf16(w), p1 = bf16(w - p0), p2 = bf16(w - p0 - p1), round-to-nearest-even).  Prepares the
 * weights of ndet_conv_ndhwc_split (the conv weights of necks/imvoxelnet.py:36-67, dense_heads/imvoxel_head_v2.py:45-49). */
int ndet_split_weights_bf16x3(const float* w_packed, int taps, int Cout, int Cin, uint16_t* planes, void* stream);

/* The same planes straight from a torch-layout weight (Cout, Cin, taps) -- training re-packs every step.  adjoint = 0: the layer's
 * own weight, planes (taps, Cin/32, 3, Cout, 32).  adjoint = 1: the weight of the layer's data gradient (autograd of the
 * convolutions of mmdet3d/models/necks/imvoxelnet.py:22-67,233-260), W'[t][ci][co] = W[co][ci][taps-1-t], planes
 * (taps, ceil32(Cout)/32, 3, Cin, 32) with the padded input channels zero. */
int ndet_split_weights_bf16x3_torch(const float* w_torch, int taps, int Cout, int Cin, int adjoint, uint16_t* planes, void* stream);

/* Same contract as ndet_conv_ndhwc / the transposed form of ndet_conv3d_ndhwc, computed on the bf16 matrix cores:
 * weights as tiled bf16 planes (taps, Cin/32, 3, Cout, 32) from ndet_split_weights_bf16x3, activations split on the fly; the six
 * products of order <= 2 are accumulated in fp32 (error at the level of an fp32 FMA chain).  transposed = 1: k2 s2
 * ConvTranspose3d (kernel/stride must be 2, pad 0; 8 taps).  tile (rows x output channels of a workgroup's tile): 0 auto; 64 (64 x 64), 128
 * (128 x 128), 12864 (128 x 64): the unified tiles, LDS-staged epilogue; 100064 / 100128 / 112864: the same tiles storing straight from the
 * accumulators (splits == 1, not transposed, Cout % 32 == 0, output < 4 GB; plain and nearest-x2 residual); 128256: wave-specialised 128 x 256;
 * 129256 / 129257 (eight consumer waves) / 129064 (64-row tiles): its persistent form (plain convolutions, Cout % 16 == 0, <= 32 taps);
 * 3128 / 3256 / 3257 / 3258: halo-stationary 128-voxel patch x 128 / 256 channels (stride 1, odd kernel, same padding, more than one tap;
 * 3257: two consumer waves per SIMD, 3258: eight producer waves).  The staged and direct forms of a unified tile, and the one-shot and persistent
 * forms of the wave-specialised tile, give bit-identical results (tests/test_conv3d_gpu.py).
 * Replaces the same reference modules as ndet_conv3d_ndhwc (necks/imvoxelnet.py:36-67,233-260,
 * dense_heads/imvoxel_head_v2.py:45-49) and the mmdet ResNet/FPN convolutions behind nerfdet.py:140. */
int ndet_conv_ndhwc_split(const float* in, const uint16_t* w_planes, float* out, int D, int H, int W, int Cin, int Cout,
                          const int* kernel, const int* stride, const int* pad, int transposed, const float* scale,
                          const float* shift, const float* residual, int residual_up2, int relu, int splits, int tile,
                          void* workspace, void* stream);

/* The same convolution with both operands rounded to bf16 and ONE MFMA product per multiply (fp32 accumulate, fp32 activations in
 * HBM): the "bf16" arithmetic BASELINE.json's configs 3 and 5 name -- what torch.autocast(bfloat16) would run nn.Conv3d / nn.Conv2d
 * of mmdet3d/models/necks/imvoxelnet.py:22-67,233-260 and the ResNet/FPN layers in.  Same arguments and weight planes as
 * ndet_conv_ndhwc_split (only the first plane is multiplied). */
int ndet_conv_ndhwc_bf16(const float* in, const uint16_t* w_planes, float* out, int D, int H, int W, int Cin, int Cout,
                         const int* kernel, const int* stride, const int* pad, int transposed, const float* scale,
                         const float* shift, const float* residual, int residual_up2, int relu, int splits, int tile,
                         void* workspace, void* stream);

/* Convolution + chained 1x1 convolution in one launch: out = act3(bn3(W3 . relu(bn1(conv(in)))) + residual), the intermediate (Cmid = 64 or
 * 128 channels, ALL of them in one 128-row tile) never leaves the CU.  Replaces the conv2 -> bn2 -> relu -> conv3 -> bn3 -> (+identity) ->
 * relu tail of the ResNet bottlenecks the detector runs as `self.backbone(img)` in mmdet3d/models/detectors/nerfdet.py:140 (mmdet's
 * resnet.py Bottleneck.forward; third-party, restated): in stages 1 / 2 the intermediate is 61 / 31 MB per block at 50 views 240x320.
 * in (D,H,W,Cin) fp32 channels-last (2D: D = batch, kernel[0] = 1); w_planes / w3_planes: bf16 planes of ndet_split_weights_bf16x3 for
 * the (taps, Cmid, Cin) and (1, Cout, Cmid) packed weights; scale1/shift1, scale3/shift3: folded BatchNorm (null = identity); residual
 * (M, Cout) or null; relu3: 0 none, 1 after the residual add, 2 before it; max_order 2: six products (fp32-class), 0: one (bf16).
 * Cin % 32 == 0, Cout % 64 == 0.  Same arithmetic as the two ndet_conv_ndhwc_split launches it replaces except that the intermediate is
 * not rounded through memory (it is the same fp32 value). */
int ndet_conv_chain_split(const float* in, const uint16_t* w_planes, int D, int H, int W, int Cin, int Cmid, const int* kernel,
                          const int* stride, const int* pad, const float* scale1, const float* shift1, const uint16_t* w3_planes,
                          int Cout, const float* scale3, const float* shift3, const float* residual, int relu3, float* out,
                          int max_order, void* stream);

/* A6 fused: [posenc(points) | global_feat] -> 4 x (Linear 256 + ReLU) -> sigma layer over [h | input] -> alpha = 1 - exp(-relu(sigma)), one launch,
 * the 256-wide activations never leave the CU.  Replaces VanillaNeRFRadianceField.query_density (mmdet3d/models/model_utils/nerf_mlp.py:224-227;
 * NerfMLP.query_density :138-144, MLP.forward :80-90 with skip_layer = 3, SinusoidalEncoder.forward :181-197) and the alpha line of
 * detectors/nerfdet.py:254-257 for the shipped architecture (net_depth 4, net_width 256).  points (3, N); global_feat (N, F) or null with F = 0;
 * K0 = 63 + F rounded up to a multiple of 32 (<= 256) = the padded input width the first layer's planes were built for.
 * w_planes_host / w_inv_scale_host / bias_host: HOST arrays of 4 entries -- device pointers to the fp16-pair planes of each hidden layer
 * (ndet_split_weights_f16x2 of the (1, 256, K) packed weight, K = K0 for layer 0 and 256 after it), 1 / their scales, device pointers to the
 * biases (256).  w_sigma (256 + 63 + F) over [h | input], b_sigma (1).  Arithmetic: fp16-pair products, fp32 accumulate, the activation scale
 * taken PER ROW (a row's error is 2^-22 of its own magnitude -- the rows of voxels no view sees hold ~1e9, nerfdet.py:236-243).
 * Outputs: alpha (N); raw_sigma (N, before the ReLU) and h_out (N, 256, the trunk output) may be null. */
int ndet_point_mlp_alpha(const float* points, const float* global_feat, int N, int F, int K0, int hidden, const uint16_t* const* w_planes_host,
                         const float* w_inv_scale_host, const float* const* bias_host, const float* w_sigma, const float* b_sigma,
                         float* raw_sigma, float* alpha, float* h_out, void* stream);

/* fp16-PAIR arithmetic of the same convolutions (arith = 1 below): every fp32 operand, pre-scaled by a power of two so that the tensor's
 * largest magnitude sits in [2^14, 2^15), is written as hi + lo with hi = fp16(x), lo = fp16(x - hi) (2 x 11 significand bits + sign:
 * |x - hi - lo| <= 2^-23 |x| for every element above ~2^-16 of the tensor's maximum), and a*b is accumulated in fp32 as the THREE products
 * hi_a hi_b + hi_a lo_b + lo_a hi_b (each exact in fp32; the dropped lo_a lo_b is <= 2^-22 |ab|) -- half the MFMA work of the six-product
 * bf16x3 scheme, with a measured error against fp64 at or below bf16x3's on every layer shape (three accumulator roundings per K step
 * instead of six; tests/test_conv3d_gpu.py).  Weights: ndet_split_weights_f16x2 builds the planes (taps, Cin/32, 2, Cout, 32) of
 * w * scale once per model (scale = a power of two, chosen by the caller from max |w|); activations: the kernels derive their scale on
 * the device from `in_amax` (max |in|, written by the previous layer's epilogue through its `out_amax`, or by ndet_amax_f32) and undo both
 * scales in the epilogue -- nothing is synchronised with the host.  Same reference modules as ndet_conv_ndhwc_split
 * (mmdet3d/models/necks/imvoxelnet.py:36-67,233-260). */
int ndet_split_weights_f16x2(const float* w_packed, int taps, int Cout, int Cin, float scale, uint16_t* planes, void* stream);

/* max |x| over n floats, atomically maxed (as uint bits) into *slot, which the caller zeroed: the `in_amax` of a fp16-pair convolution
 * whose input no convolution kernel wrote (the voxel volume of nerfdet.py:363-420, the MLP inputs of nerf_mlp.py:200-245). */
int ndet_amax_f32(const float* x, int64_t n, float* slot, void* stream);

/* Floats per amax slot (8 per-XCD sub-slots x 32 floats = 1 KiB): the host allocates slots of exactly this size (nerfdet_amd/conv3d.py asserts it
 * at load time; a narrower slot would make the kernels' sub-slot atomics run into the neighbouring tensor's slot).  Serves the same convolutions as ndet_amax_f32 (mmdet3d/models/necks/imvoxelnet.py:36-67); the reference
 * itself has no counterpart. */
int ndet_amax_slot_floats(void);

/* Measurement knobs of the convolution launchers, set explicitly by profiling scripts (tools/layer_times.py) -- never read from the environment, so
 * a production process cannot pick them up by accident.  "nt_bytes": outputs of at least this many bytes are written with non-temporal stores
 * (default 32 MiB); "order2": 1 / 0 = deal the column tiles of a row tile to one XCD or keep grid order (neither changes a result bit);
 * "deterministic_scatter": 1 = the backward kernels' gradient scatter on 64-bit fixed-point integer atomics (the caller then passes zeroed int64
 * buffers in place of the float ones: order-independent sums, for reproducibility tests; nerfdet_amd/autograd.py::set_deterministic);
 * "wgrad_wide": 0 = ndet_wgrad_split_f16x2 keeps its 128 x 128 tile where it would take 128 x 256 (same sums in another association);
 * "wgrad_xcd": 0 = the weight-gradient kernel's workgroups in grid order instead of one K split per XCD at a time (no result bit changes).
 * HOST string.  No reference counterpart (the reference's harness, tools/benchmark.py:63-89, times the model only). */
int ndet_measurement_knob(const char* name_host, int64_t value);

/* ndet_conv_ndhwc_split / ndet_conv_ndhwc_bf16 (necks/imvoxelnet.py:36-67,233-260, imvoxel_head_v2.py:45-49, the backbone behind
 * nerfdet.py:140) with the arithmetic as an argument: arith 0 = bf16x3 (six products), 1 = fp16 pair (three
 * products; w_planes from ndet_split_weights_f16x2, `in_amax` required, w_inv_scale = 1 / that call's scale), 2 = bf16 (one product).
 * out_amax (any arith, may be null): max |out| is maxed into *out_amax (zeroed by the caller) -- the next layer's in_amax. */
int ndet_conv_ndhwc_arith(const float* in, const uint16_t* w_planes, float* out, int D, int H, int W, int Cin, int Cout,
                          const int* kernel, const int* stride, const int* pad, int transposed, const float* scale,
                          const float* shift, const float* residual, int residual_up2, int relu, int splits, int tile, int arith,
                          const float* in_amax, float w_inv_scale, float* out_amax, void* workspace, void* stream);

/* ndet_conv_ndhwc_arith with the RANGE GUARD of the fp16-pair arithmetic (arith 1; ignored otherwise).  The activation scale of that arithmetic
 * is per tensor, so beside its fp32-class relative error an output carries an absolute floor of at most
 *     2^-39 max|in| * guard_l1,      guard_l1 = max_j |scale_j| (sum_k |w_jk| + max|w| #{k: 0 < |w_jk| < 2^-16 max|w|})      (host, once per pack)
 * whatever the distribution inside the tensor (csrc/conv_common.hpp::conv_guard_check).  max|in| is known on the device at kernel entry, and so is
 * the smallest maximum any workgroup tile of the input committed (word 1 of the amax sub-slots): when the floor exceeds guard_tol AND that tile
 * minimum lies below 2^-16 of max|in| (a part of the tensor really is outside the fp16-pair window; a uniformly large tensor is not), the launch ORs
 * 1 into *guard (device word, zeroed by the caller per scene).  The caller reads the word with the
 * detections (ndet_nms_pack_detections) and repeats such a scene on the six-product bf16x3 arithmetic.  Same reference modules as
 * ndet_conv_ndhwc_split (mmdet3d/models/necks/imvoxelnet.py:36-67,233-260, dense_heads/imvoxel_head_v2.py:45-49, the backbone behind
 * detectors/nerfdet.py:140); the tensor that first needed it: the sigma-MLP rows of nerfdet.py:236-243. */
int ndet_conv_ndhwc_guarded(const float* in, const uint16_t* w_planes, float* out, int D, int H, int W, int Cin, int Cout,
                            const int* kernel, const int* stride, const int* pad, int transposed, const float* scale,
                            const float* shift, const float* residual, int residual_up2, int relu, int splits, int tile, int arith,
                            const float* in_amax, float w_inv_scale, float* out_amax, void* workspace, float guard_l1, float guard_tol,
                            unsigned* guard, void* stream);

/* ndet_conv_ndhwc_guarded (plain stride-1 same-padded convolution, no residual / ReLU / split-K) with a chained 32-channel projection of every output
 * row in the same launch: map_out (M, 32) = out_row . map_w + map_b, map_w (Cout, 32) and map_b (32) with the convolution's own affine folded in by
 * the caller (map_w[c][j] = scale_c Wm[j][c], map_b[j] = sum_c shift_c Wm[j][c] + bm[j]); fp32 FMAs.  The detector's feature mapping
 * (mmdet3d/models/detectors/nerfdet.py:194-197: self.mapping on every FPN level-0 pixel) behind the FPN output convolution (nerfdet.py:140-142): the
 * 276 MB feature map is not read back by a launch of its own.  Only the 256-column halo tiles own whole rows: tile 3256 / 3257 / 3258, Cout = 256. */
int ndet_conv_ndhwc_mapped(const float* in, const uint16_t* w_planes, float* out, int D, int H, int W, int Cin, int Cout, const int* kernel,
                           const int* stride, const int* pad, const float* scale, const float* shift, int tile, int arith, const float* in_amax,
                           float w_inv_scale, float* out_amax, float guard_l1, float guard_tol, unsigned* guard, const float* map_w,
                           const float* map_b, float* map_out, void* stream);

/* ndet_conv_chain_arith with the range guard (see ndet_conv_ndhwc_guarded): guard_l1 belongs to w_planes and max|in|, guard_l1_3 to w3_planes and
 * the chained product (reserved: its operand is scaled by each workgroup's own maximum, which needs no check).  The bottleneck tail of the backbone called at mmdet3d/models/detectors/nerfdet.py:140. */
int ndet_conv_chain_guarded(const float* in, const uint16_t* w_planes, int D, int H, int W, int Cin, int Cmid, const int* kernel,
                            const int* stride, const int* pad, const float* scale1, const float* shift1, const uint16_t* w3_planes,
                            int Cout, const float* scale3, const float* shift3, const float* residual, int relu3, float* out,
                            int arith, const float* in_amax, float w1_inv_scale, float w3_inv_scale, float* out_amax, float guard_l1,
                            float guard_l1_3, float guard_tol, unsigned* guard, void* stream);

/* A whole ResNet bottleneck of stage 1 in one launch, fp16-pair arithmetic: out = relu(bn3(W3 . relu(bn2(conv3x3(relu(bn1(W1 . x)))))) + identity),
 * identity = x (Cin == Cout) or, with wd_planes, bnD(WD . x) (the first block of the stage; Cin = 64) -- mmdet's Bottleneck.forward (style 'pytorch',
 * stride 1) behind mmdet3d/models/detectors/nerfdet.py:140.  x (N, H, W, Cin) channels-last, out (N, H, W, Cout); the intermediate has 64 channels
 * (w1 (1, Cin/32, 2, 64, 32), w2 (9, 2, 2, 64, 32), w3 (1, 2, 2, Cout, 32), wd (1, Cin/32, 2, Cout, 32): ndet_split_weights_f16x2 planes with their
 * inverse scales; scale* / shift*: folded eval-mode BatchNorm).  A workgroup owns a 4 x 16 patch of one map from x to out: conv1 is evaluated on
 * the patch plus its one-pixel halo into LDS, conv2's taps multiply out of that image, conv3 follows as in ndet_conv_chain_arith; the scales of
 * the two intermediates are the workgroup's own maxima.  in_amax: x's slot; out_amax (may be null): max |out|.  guard (may be null): the range
 * guard word (see ndet_conv_ndhwc_guarded), guard_l1_host = 4 HOST floats (w1, w2, w3, wd). */
int ndet_bottleneck_f16x2(const float* x, int N, int H, int W, int Cin, int Cout, const uint16_t* w1_planes, float w1_inv_scale, const float* scale1,
                          const float* shift1, const uint16_t* w2_planes, float w2_inv_scale, const float* scale2, const float* shift2,
                          const uint16_t* w3_planes, float w3_inv_scale, const float* scale3, const float* shift3, const uint16_t* wd_planes,
                          float wd_inv_scale, const float* scale_d, const float* shift_d, const float* in_amax, float* out_amax, float* out,
                          const float* guard_l1_host, float guard_tol, unsigned* guard, void* stream);

/* ndet_conv_chain_split with the arithmetic as an argument (as above; w1_inv_scale / w3_inv_scale belong to w_planes / w3_planes).  In the
 * fp16-pair arithmetic the intermediate's scale is the workgroup's own maximum: it never exists as a whole tensor.  Same reference code
 * as ndet_conv_chain_split: the bottleneck tail of the backbone called at mmdet3d/models/detectors/nerfdet.py:140. */
int ndet_conv_chain_arith(const float* in, const uint16_t* w_planes, int D, int H, int W, int Cin, int Cmid, const int* kernel,
                          const int* stride, const int* pad, const float* scale1, const float* shift1, const uint16_t* w3_planes,
                          int Cout, const float* scale3, const float* shift3, const float* residual, int relu3, float* out,
                          int arith, const float* in_amax, float w1_inv_scale, float w3_inv_scale, float* out_amax, void* stream);

/* ResNet stem tail in one pass: BatchNorm(eval) as per-channel scale/shift + ReLU + MaxPool(3, stride 2, pad 1) on the
 * channels-last stem output x (N,H,W,C), C % 4 == 0 -> out (N, (H-1)/2+1, (W-1)/2+1, C).  Third-party mmdet ResNet stem
 * (SURVEY.md appendix C), called at mmdet3d/models/detectors/nerfdet.py:140. */
int ndet_bn_relu_maxpool_nhwc(const float* x, const float* scale, const float* shift, int N, int H, int W, int C,
                              float* out, void* stream);

/* ResNet stem in ONE launch: Conv2d(3, 64, 7, stride 2, pad 3, no bias) + BatchNorm(eval, as per-channel scale/shift) + ReLU +
 * MaxPool(3, stride 2, pad 1) -- conv1 / bn1 / relu / maxpool of the third-party mmdet ResNet called at
 * mmdet3d/models/detectors/nerfdet.py:140.  images: N views of 3 x H x W fp32 with the given ELEMENT strides (NCHW or channels-last);
 * w_planes: ndet_stem_pack_weights of the (64,3,7,7) weight; out (N, PH, PW, 64) channels-last, PH = ((H-1)/2+1 - 1)/2 + 1.
 * Arithmetic of ndet_conv_ndhwc_split (fp32 operands as exact sums of three bf16 terms, six MFMA products). */
int ndet_stem_pack_weights(const float* w_64x3x7x7, uint16_t* planes /* (3, 64, 176) */, void* stream);
/* ... and its fp16-pair form (two fp16 planes of w * scale, scale a power of two chosen by the caller from max |w|): ndet_stem_conv_bn_relu_maxpool
 * with w_inv_scale = 1 / scale then issues three fp16 MFMA products per multiply, the image patch scaled by the power of two of its own maximum
 * inside the kernel (same conv1 / bn1 / relu / maxpool of mmdet's ResNet behind mmdet3d/models/detectors/nerfdet.py:140). */
int ndet_stem_pack_weights_f16x2(const float* w_64x3x7x7, float scale, uint16_t* planes /* (2, 64, 176) */, void* stream);

/* The launch itself (see above; conv1 / bn1 / relu / maxpool behind mmdet3d/models/detectors/nerfdet.py:140).  w_inv_scale: 0 for the bf16x3 planes of
 * ndet_stem_pack_weights, else 1 / scale of ndet_stem_pack_weights_f16x2.  out_amax (may be null): max |out|
 * into a zeroed amax slot -- the first bottleneck's fp16-pair scale without another pass over the 61 MB. */
int ndet_stem_conv_bn_relu_maxpool(const float* images, int N, int H, int W, int64_t stride_n, int64_t stride_c, int64_t stride_y,
                                   int64_t stride_x, const uint16_t* w_planes, float w_inv_scale, const float* scale, const float* shift, float* out, float* out_amax,
                                   void* stream);

/* ---- input contract: what the data pipeline hands to nerfdet.forward_*, from decoded,
 * resized and padded uint8 BGR frames (n_frames,H,W,3) resident on the device.  mean_rgb / std_rgb are HOST arrays of 3
 * doubles (the config's img_norm_cfg); the image arithmetic is float32 with stdinv = float(1 / std), as mmcv does it.
 *
 * ndet_normalize_views: for each selected frame ids[v]: img (n_sel,3,H,W) = mmcv.imnormalize(to_rgb=True), and
 * denorm (n_sel,3,H,W) = imdenormalize(img, to_bgr=True).astype(uint8) / 255 -- the round trip of
 * mmdet3d/datasets/pipelines/multi_view.py:107-110, channel-first as formating.py:44-52,80-85 stacks them. */
int ndet_normalize_views(const uint8_t* frames_bgr, const int* ids, int n_sel, int H, int W, const double* mean_rgb,
                         const double* std_rgb, float* img, float* denorm, void* stream);

/* ndet_target_rays: the NeRF targets of multi_view.py:117-155 for frames target_ids[t]: rays on the pixel grid
 * [margin, W-margin) x [margin, H-margin), row-major; raydirs (n_t,R,3) = get_dtu_raydir
 * (data_augment_utils.py:410-424) with intrinsic_rows = rows 0,1 of intrinsic[:2]/ratio as a device (2,3) array and
 * camrotc2w (n_frames,3,3); lightpos (n_t,R,3) = cam_lightpos[frame] repeated (formating.py:70-75); gt_images
 * (n_t,R,3) = the de-normalised BGR frame / 255 at the ray's pixel (multi_view.py:147-150). */
int ndet_target_rays(const uint8_t* frames_bgr, const int* target_ids, int n_targets, int H, int W, int margin,
                     const float* intrinsic_rows, const float* camrotc2w, const float* cam_lightpos, const double* mean_rgb,
                     const double* std_rgb, float* raydirs, float* lightpos, float* gt_images, void* stream);

/* Weight-gradient staging for the training-time convolutions (autograd of nn.Conv3d / nn.Conv2d in
 * mmdet3d/models/necks/imvoxelnet.py:22-67,233-260 and of the third-party ResNet/FPN layers): channels-last x (D,H,W,C) -> rows
 * out[t - t0][c][j] = x[stride * o(j) + tap(t) - pad][c] over the flattened OUTPUT grid of the (kernel, stride, pad) convolution, taps
 * t0 .. t0+n_taps-1; every element of out (n_taps, C, lrow) is written (zeros where the tap reads padding and for j past the grid;
 * lrow >= OD*OH*OW).  With these rows (and dY staged by the same call with a 1x1x1 kernel) the weight gradient is one GEMM over j on
 * ndet_conv_ndhwc_split, for any stride (nerfdet_amd/conv_train.py). */
int ndet_wgrad_rows(const float* x_ndhwc, int D, int H, int W, int C, const int* kernel, const int* stride, const int* pad, int t0,
                    int n_taps, int lrow, float* out, void* stream);

/* Weight gradient of a convolution as ONE implicit GEMM (autograd of nn.Conv3d / nn.Conv2d in
 * mmdet3d/models/necks/imvoxelnet.py:22-67,233-260 and of the ResNet / FPN layers): dw_rows[(t, ci)][co] = sum over output voxels j of
 * x[stride * o(j) + tap(t) - pad][ci] * dy[j][co].  x (D,H,W,Cin) channels-last fp32 is read in place (transposed on its way into
 * LDS); dy arrives as the bf16 planes of its channel-major rows: ndet_wgrad_rows(dy, 1x1x1) -> (Cout, lrow), then
 * ndet_split_weights_bf16x3 on that (1, Cout, lrow) "weight".  dw_rows (taps * Cin, Cout) row-major; Cin % 64 == 0, lrow % 32 == 0;
 * splits > 1: workspace of splits * taps * Cin * Cout floats, reduced in a fixed order.  max_order 2: six products, 0: one (bf16). */
int ndet_wgrad_split(const float* x_ndhwc, int D, int H, int W, int Cin, const int* kernel, const int* stride, const int* pad,
                     const uint16_t* dy_planes, int Cout, int lrow, int splits, int max_order, void* workspace, float* dw_rows,
                     void* stream);

/* dY of a convolution (L output voxels x Cout, channels-last fp32) -> the bf16 planes of its channel-major rows in the layout of
 * ndet_split_weights_bf16x3 for a (1, Cout, lrow) "weight": (lrow/32, 3, Cout, 32), zeros past L -- the "weight" operand of the
 * weight-gradient GEMMs (ndet_wgrad_split, or ndet_conv_ndhwc_split on ndet_wgrad_rows' tap copies) in one pass; autograd of nn.Conv3d /
 * nn.Conv2d in mmdet3d/models/necks/imvoxelnet.py:22-67,233-260 and of the ResNet / FPN layers. */
int ndet_wgrad_dy_planes(const float* dy_rows_by_voxel, int L, int Cout, int lrow, uint16_t* planes, void* stream);

/* fp16-pair forms of the two calls above (the training step on the three-product arithmetic): dy's planes are (lrow/32, 2, Cout, 32), two fp16
 * halves of dy * 2^k with 2^k taken ON THE DEVICE from dy's amax slot (ndet_amax_f32; largest magnitude in [2^14, 2^15)); ndet_wgrad_split_f16x2
 * splits x the same way under x_amax and multiplies the sums by the inverse of both scales, read from the same slots.  Same reference as
 * ndet_wgrad_split: autograd of nn.Conv3d / nn.Conv2d in mmdet3d/models/necks/imvoxelnet.py:22-67,233-260 and of the ResNet / FPN layers. */
int ndet_wgrad_dy_planes_f16x2(const float* dy_rows_by_voxel, int L, int Cout, int lrow, const float* dy_amax, uint16_t* planes, void* stream);
int ndet_wgrad_split_f16x2(const float* x_ndhwc, int D, int H, int W, int Cin, const int* kernel, const int* stride, const int* pad,
                           const uint16_t* dy_planes, int Cout, int lrow, int splits, const float* x_amax, const float* dy_amax, void* workspace,
                           float* dw_rows, int keep_partials, void* stream);

/* Both weight packs of one training step in ONE pass over a torch-layout weight (Cout, Cin, taps <= 27): planes = the layer's own
 * (taps, Cin/32, P, Cout, 32), planes_adjoint (may be null) = its data gradient's (taps, ceil32(Cout)/32, P, Cin, 32), W'[t][ci][co] = W[co][ci][taps-1-t]
 * (see ndet_split_weights_bf16x3_torch).  arith 0: P = 3 bf16 planes (w_amax null).  arith 1: P = 2 fp16 planes of w * 2^k, 2^k derived on the
 * device from the weight's amax slot w_amax (ndet_amax_f32) -- the optimizer moves the weights every step
 * (config nerfdet_res50_2x_low_res.py:167-172, AdamW), and no maximum has to reach the host for it.  A workgroup moves a 32 x 32 (co, ci) block with
 * all its taps through LDS: coalesced reads of the torch layout, every (tap, plane) of either pack written as one 2 KB piece. */
int ndet_split_weights_train(const float* w_torch, int taps, int Cout, int Cin, int arith, const float* w_amax, uint16_t* planes,
                             uint16_t* planes_adjoint, void* stream);

/* BatchNorm on BATCH statistics over channels-last rows (N x C fp32, C / 4 a divisor of 1024), with the ReLU and the residual add that follow it in
 * BasicBlock3dV2 / the up and out blocks of FastIndoorImVoxelNeck folded in (mmdet3d/models/necks/imvoxelnet.py:22-67,233-260; training mode of
 * nn.BatchNorm3d): y = relu?((x - mean) / sqrt(var + eps) * gamma + beta (+ residual)), mean / biased var over the N rows; running_mean /
 * running_var (may be null) updated with `momentum` (unbiased variance), save_mean / save_invstd kept for the backward.  y_amax (may be null): a
 * zeroed amax slot that receives max |y|.  workspace: ndet_bn_workspace_floats(N, C) floats.  Three launches (partial sums, finish, apply); the
 * partial sums are added in a fixed order.  The backward: dx = gamma invstd (g - mean(g) - xhat mean(g xhat)) with g = dy [y > 0] when relu (y = the
 * forward's output), d_residual (may be null) = g, dgamma = sum g xhat, dbeta = sum g; dx_amax as y_amax. */
int64_t ndet_bn_workspace_floats(int64_t N, int C);
int ndet_bn_train_forward(const float* x, int64_t N, int C, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          float momentum, float eps, const float* residual, int relu, float* y, float* save_mean, float* save_invstd, float* y_amax,
                          float* workspace, void* stream);
int ndet_bn_train_backward(const float* dy, const float* x, const float* y, int64_t N, int C, const float* gamma, const float* save_mean,
                           const float* save_invstd, int relu, float* dx, float* d_residual, float* dgamma, float* dbeta, float* dx_amax,
                           float* workspace, void* stream);

/* The weight gradient in torch's layout: dw_rows ((tap, ci) rows x Cout floats, what ndet_wgrad_split* and the staged GEMM write) ->
 * dw_torch (Cout, Cin, taps), the layout autograd hands to the optimizer for nn.Conv3d / nn.Conv2d.weight
 * (mmdet3d/models/necks/imvoxelnet.py:22-67,233-260).  32 x 32 x taps blocks through LDS, coalesced on both sides; taps <= 27, Cin % 32 == 0.
 * splits > 1: dw_rows is the split-K WORKSPACE of a launch made with keep_partials = 1 (ndet_conv_ndhwc_train, ndet_wgrad_split_f16x2) -- `splits`
 * partial sums taps * Cin * Cout floats apart, added here in index order (what the separate reduction pass would have done, bit for bit). */
int ndet_wgrad_to_torch(const float* dw_rows, int splits, int taps, int Cout, int Cin, float* dw_torch, void* stream);

/* The fp16-pair convolution launch of the training step (forward and data gradient of the convolutions of
 * mmdet3d/models/necks/imvoxelnet.py:22-67,233-260, dense_heads/imvoxel_head_v2.py:45-58,444-449 and the trainable ResNet / FPN layers behind
 * detectors/nerfdet.py:140-142): ndet_conv_ndhwc_guarded with arith = 1 whose weight planes were scaled on the device -- by ndet_split_weights_train,
 * or, for the weight-gradient GEMM over ndet_wgrad_rows' tap copies, dy's planes from ndet_wgrad_dy_planes_f16x2 -- so 1 / (weight scale) is taken
 * from the slot w_amax instead of a host float.  guard (may be null): the range guard with ||w||_1 bounded by guard_k * max|w|, guard_k = taps * Cin.
 * keep_partials = 1 (weight-gradient GEMMs; no affine / residual / ReLU / out_amax): a split-K launch leaves its partial sums in the workspace for
 * ndet_wgrad_to_torch instead of running the reduction pass; `out` is then not written. */
int ndet_conv_ndhwc_train(const float* in, const uint16_t* w_planes, float* out, int D, int H, int W, int Cin, int Cout, const int* kernel,
                          const int* stride, const int* pad, const float* scale, const float* shift, const float* residual, int relu, int splits,
                          int tile, const float* in_amax, const float* w_amax, float* out_amax, void* workspace, float guard_k, float guard_tol,
                          unsigned* guard, int keep_partials, void* stream);

/* Backward of the fused epilogue y = relu(conv * scale + shift (+ identity)) -- convolution + frozen eval-mode BatchNorm + ReLU (+ the
 * bottleneck's identity) of the trainable ResNet stages (mmdet Bottleneck.forward behind mmdet3d/models/detectors/nerfdet.py:140;
 * config: norm_eval=True, norm_cfg.requires_grad=False): d_identity = dy [y > 0] (null = not wanted), d_conv = d_identity * scale[c],
 * one pass over `rows` channels-last rows of C floats (C % 4 == 0).  relu = 0: no mask (y may be null). */
int ndet_relu_affine_bwd(const float* dy, const float* y, const float* scale, int64_t rows, int C, int relu, float* d_identity,
                         float* d_conv, void* stream);
/* ndet_relu_affine_bwd that also leaves max |d_conv| in the zeroed 1 KiB amax slot d_conv_amax (see ndet_amax_f32): d_conv is the operand of the
 * layer's data and weight gradients, whose fp16-pair launches take their scale from that slot.  Same reference: mmdet Bottleneck.forward behind
 * mmdet3d/models/detectors/nerfdet.py:140 under norm_eval=True. */
int ndet_relu_affine_bwd_amax(const float* dy, const float* y, const float* scale, int64_t rows, int C, int relu, float* d_identity,
                              float* d_conv, float* d_conv_amax, void* stream);

/* ---- backward passes (training).  The reference obtains these from autograd over its materialised tensors; each
 * entry point names the forward statement it differentiates.  Scatter targets must be zero-initialised by the caller;
 * accumulation uses float atomics (order not fixed). ---- */

/* d(features) of the view mean, mmdet3d/models/detectors/nerfdet.py:164-176: grad_features[v,y,x,:] += grad_mean[n,:] /
 * (count_n + 1e-8) for every view v that sees voxel n.  grad_mean in `grad_layout`; grad_features_nhwc as features_nhwc. */
int ndet_backproject_aggregate_bwd(const float* grad_mean, int grad_layout, int n_views, int C, int h, int w,
                                   int64_t view_pitch, int64_t row_pitch, const float* points, int N,
                                   const float* projection, float* grad_features_nhwc, void* stream);

/* d(mapped features), d(bias) of ndet_density_features, nerfdet.py:234-253 (the RGB volume carries no gradient). */
int ndet_density_features_bwd(const float* grad_global_feat, const float* mapped_nhwc, int n_views, int cm, int h, int w,
                              int64_t mview_pitch, int64_t mrow_pitch, const float* bias, const float* points, int N,
                              const float* projection, float* grad_mapped_nhwc, float* grad_bias, void* stream);

/* d(mapped features) of ndet_ray_view_stats: F.grid_sample backward (projection.py:127) through the masked statistics of
 * render_ray.py:83-88.  Sample points and images carry no gradient. */
int ndet_ray_view_stats_bwd(const float* grad_global_feat, const float* pts, int n_points, const float* KE, int n_views,
                            float img_h, float img_w, const float* feat_nhwc, int d, int hf, int wf,
                            int64_t fview_pitch, int64_t frow_pitch, float* grad_feat_nhwc, void* stream);

/* Packed form of ndet_ray_view_stats_bwd (d % 4 == 0, n_views <= 128): F.grid_sample backward (projection.py:127) through the
 * masked statistics of render_ray.py:83-88; grad_feat_nhwc shares feat_nhwc's pitches and must be zero-initialised. */
int ndet_ray_view_stats_packed_bwd(const float* grad_global_feat, const float* pts, int n_points, const float* KE, int n_views,
                                   float img_h, float img_w, const float* feat_nhwc, int d, int hf, int wf, int64_t fview_pitch,
                                   int64_t frow_pitch, float* grad_feat_nhwc, void* stream);

/* d(raw) of ndet_composite (render_ray.py:196-236) from d(rgb_map) (R,3) and d(depth_map) (R) or NULL.
 * transparency: the forward's (R,S) output. */
int ndet_composite_bwd(const float* raw, const float* z_vals, const float* transparency, int R, int S, int white_bkgd,
                       const float* zminmax, const float* grad_rgb, const float* grad_depth, float* grad_raw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFDET_HIP_H */
