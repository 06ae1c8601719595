// Input contract of the hot path: what MultiViewPipeline + DefaultFormatBundle3D hand to
// nerfdet.forward_*, produced on the device from decoded, resized and padded uint8 frames.
//
//   k_normalize_views  mmcv.imnormalize(to_rgb=True) and the reference's own round trip
//                      mmcv.imdenormalize(img, mean, std, to_bgr=True).astype(uint8) / 255 (multi_view.py:107-110), per
//                      selected view, written channel-first as DefaultFormatBundle does (formating.py:44-52,80-85).
//   k_target_rays      the NeRF targets of multi_view.py:117-155: margin-cropped pixel grid, get_dtu_raydir
//                      (data_augment_utils.py:410-424, un-normalised), the camera centre repeated per ray
//                      (formating.py:70-75) and the target colours gathered from the de-normalised frame.
#include "ndet_common.hpp"
#include "spl_common.hpp"

__global__ __launch_bounds__(256) void k_normalize_views(const uint8_t* __restrict__ frames, const int* __restrict__ ids, int n_sel, int H, int W,
                                                         float m0, float m1, float m2, float i0, float i1, float i2, float s0, float s1,
                                                         float s2, float* __restrict__ img, float* __restrict__ denorm) {
    const int64_t hw = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sel * hw) return;
    const int v = (int)(i / hw);
    const int64_t px = i - v * hw;
    const uint8_t* src = frames + ((int64_t)ids[v] * hw + px) * 3;   // B, G, R
    const float mean[3] = {m0, m1, m2}, inv[3] = {i0, i1, i2}, sd[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {   // c indexes RGB
        const float x = ((float)src[2 - c] - mean[c]) * inv[c];
        img[((int64_t)v * 3 + c) * hw + px] = x;
        const float back = x * sd[c] + mean[c];                       // imdenormalize
        const float q = (float)(unsigned char)(int)back;              // .astype(np.uint8): truncation
        denorm[((int64_t)v * 3 + (2 - c)) * hw + px] = q / 255.0f;    // BGR planes
    }
}

__global__ __launch_bounds__(256) void k_target_rays(const uint8_t* __restrict__ frames, const int* __restrict__ tids, int n_t, int H, int W, int margin,
                                                     const float* __restrict__ kn /* (2,3): rows 0,1 of the scaled intrinsics */,
                                                     const float* __restrict__ rot /* (n_frames,3,3) c2w rotations */,
                                                     const float* __restrict__ lpos /* (n_frames,3) */, float m0, float m1, float m2, float i0,
                                                     float i1, float i2, float s0, float s1, float s2, float* __restrict__ raydirs,
                                                     float* __restrict__ lightpos, float* __restrict__ gt) {
    const int rw = W - 2 * margin, rh = H - 2 * margin;
    const int64_t per = (int64_t)rw * rh;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_t * per) return;
    const int t = (int)(i / per);
    const int r = (int)(i - t * per);
    const int px = margin + r % rw, py = margin + r / rw;
    const int f = tids[t];
    const float x = ((float)px + 0.5f - kn[2]) / kn[0];
    const float y = ((float)py + 0.5f - kn[5]) / kn[4];
    const float* R = rot + (int64_t)f * 9;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float d = x * R[j * 3 + 0];
        d = fmaf(y, R[j * 3 + 1], d);
        d = d + R[j * 3 + 2];
        raydirs[i * 3 + j] = d;
        lightpos[i * 3 + j] = lpos[(int64_t)f * 3 + j];
    }
    const uint8_t* src = frames + (((int64_t)f * H + py) * W + px) * 3;
    const float mean[3] = {m0, m1, m2}, inv[3] = {i0, i1, i2}, sd[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float xx = ((float)src[2 - c] - mean[c]) * inv[c];
        const float back = xx * sd[c] + mean[c];
        gt[i * 3 + (2 - c)] = (float)(unsigned char)(int)back / 255.0f;
    }
}

extern "C" int ndet_normalize_views(const uint8_t* frames_bgr, const int* ids, int n_sel, int H, int W, const double* mean_rgb,
                                    const double* std_rgb, float* img, float* denorm, void* stream) {
    const char* fn = "ndet_normalize_views";
    NDET_REQUIRE(frames_bgr && ids && mean_rgb && std_rgb && img && denorm, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_sel > 0 && H > 0 && W > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    float inv[3], mean[3], sd[3];
    for (int c = 0; c < 3; ++c) {
        NDET_REQUIRE(std_rgb[c] > 0.0, NDET_E_INVALID, "%s: std must be positive", fn);
        inv[c] = (float)(1.0 / std_rgb[c]);   // mmcv: stdinv = 1 / np.float64(std), applied to the float32 image
        mean[c] = (float)mean_rgb[c];
        sd[c] = (float)std_rgb[c];
    }
    const int64_t total = (int64_t)n_sel * H * W;
    hipLaunchKernelGGL(k_normalize_views, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, frames_bgr, ids, n_sel, H, W,
                       mean[0], mean[1], mean[2], inv[0], inv[1], inv[2], sd[0], sd[1], sd[2], img, denorm);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

extern "C" int ndet_target_rays(const uint8_t* frames_bgr, const int* target_ids, int n_targets, int H, int W, int margin,
                                const float* intrinsic_rows /* device (2,3) */, const float* camrotc2w, const float* cam_lightpos,
                                const double* mean_rgb, const double* std_rgb, float* raydirs, float* lightpos, float* gt_images, void* stream) {
    const char* fn = "ndet_target_rays";
    NDET_REQUIRE(frames_bgr && target_ids && intrinsic_rows && camrotc2w && cam_lightpos && mean_rgb && std_rgb && raydirs && lightpos && gt_images,
                 NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_targets > 0 && H > 0 && W > 0 && margin >= 0 && 2 * margin < H && 2 * margin < W, NDET_E_INVALID, "%s: bad sizes / margin", fn);
    float inv[3], mean[3], sd[3];
    for (int c = 0; c < 3; ++c) {
        NDET_REQUIRE(std_rgb[c] > 0.0, NDET_E_INVALID, "%s: std must be positive", fn);
        inv[c] = (float)(1.0 / std_rgb[c]);
        mean[c] = (float)mean_rgb[c];
        sd[c] = (float)std_rgb[c];
    }
    const int64_t total = (int64_t)n_targets * (H - 2 * margin) * (W - 2 * margin);
    hipLaunchKernelGGL(k_target_rays, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, frames_bgr, target_ids, n_targets, H, W,
                       margin, intrinsic_rows, camrotc2w, cam_lightpos, mean[0], mean[1], mean[2], inv[0], inv[1], inv[2], sd[0], sd[1], sd[2], raydirs,
                       lightpos, gt_images);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight-gradient staging (nerfdet_amd/conv_train.py): channels-last activations (D,H,W,C) -> channel-major rows over the flattened
// OUTPUT grid (OD,OH,OW) of the convolution, one copy per tap:
//     out[t][c][j] = x[s * o(j) + tap(t) - pad][c]     (0 where that voxel lies outside the input, and for j past the grid)
// so that dW[t] = dY_rows . out[t]^T is a plain GEMM over j for any stride (autograd of nn.Conv3d / nn.Conv2d in
// mmdet3d/models/necks/imvoxelnet.py:22-67,233-260).  dY itself is staged by the same kernel (1x1x1 tap, stride 1: a transpose).
// Every destination element is written exactly once: no separate clear.  64 positions x 64 channels per workgroup through an LDS
// transpose: reads coalesced along C, writes coalesced along j.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wgrad_rows(const float* __restrict__ x, int D, int H, int W, int C, int kd, int kh, int kw, int pd,
                                                    int ph, int pw, int sd, int sh, int sw, int OD, int OH, int OW, int t0, int lrow,
                                                    float* __restrict__ out) {
    __shared__ float tile[64][65];
    __shared__ int src[64];                                  // source voxel of each of the 64 positions, -1 = padding / past the grid
    const int j0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int t = t0 + blockIdx.z;
    const int a = t / (kh * kw), b = (t / kw) % kh, c = t % kw;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (threadIdx.x < 64) {                                  // one index decode per position (integer divisions), not per element
        const int j = j0 + tx;
        int s_ = -1;
        if (j < OD * OH * OW) {
            const int w_ = (j % OW) * sw + c - pw, h_ = ((j / OW) % OH) * sh + b - ph, d_ = (j / (OW * OH)) * sd + a - pd;
            if (w_ >= 0 && w_ < W && h_ >= 0 && h_ < H && d_ >= 0 && d_ < D) s_ = (d_ * H + h_) * W + w_;
        }
        src[tx] = s_;
    }
    __syncthreads();
    const bool cok = c0 + tx < C;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int vj = ty + 4 * i;
        const int s_ = src[vj];
        tile[vj][tx] = (s_ >= 0 && cok) ? x[(int64_t)s_ * C + c0 + tx] : 0.0f;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int ci = ty + 4 * i;
        if (c0 + ci < C && j0 + tx < lrow) out[((int64_t)blockIdx.z * C + c0 + ci) * lrow + j0 + tx] = tile[tx][ci];
    }
}

extern "C" int ndet_wgrad_rows(const float* x_ndhwc, int D, int H, int W, int C, const int* kernel, const int* stride, const int* pad, int t0,
                               int n_taps, int lrow, float* out, void* stream) {
    const char* fn = "ndet_wgrad_rows";
    NDET_REQUIRE(x_ndhwc && out && kernel && stride && pad, NDET_E_INVALID, "%s: null pointer", fn);
    const int kd = kernel[0], kh = kernel[1], kw = kernel[2];
    NDET_REQUIRE(D > 0 && H > 0 && W > 0 && C > 0 && kd > 0 && kh > 0 && kw > 0 && n_taps > 0 && t0 >= 0 && t0 + n_taps <= kd * kh * kw && lrow > 0,
                 NDET_E_INVALID, "%s: bad sizes", fn);
    for (int a = 0; a < 3; ++a)
        NDET_REQUIRE(stride[a] >= 1 && pad[a] >= 0 && pad[a] < kernel[a], NDET_E_INVALID, "%s: bad stride / pad on axis %d", fn, a);
    const int OD = (D + 2 * pad[0] - kd) / stride[0] + 1, OH = (H + 2 * pad[1] - kh) / stride[1] + 1, OW = (W + 2 * pad[2] - kw) / stride[2] + 1;
    NDET_REQUIRE(OD > 0 && OH > 0 && OW > 0 && (int64_t)OD * OH * OW <= lrow && (int64_t)D * H * W < ((int64_t)1 << 31), NDET_E_INVALID,
                 "%s: row shorter than the output grid (%d x %d x %d)", fn, OD, OH, OW);
    NDET_REQUIRE((C + 63) / 64 <= 65535 && n_taps <= 65535, NDET_E_UNSUPPORTED, "%s: grid too large", fn);
    hipLaunchKernelGGL(k_wgrad_rows, dim3((lrow + 63) / 64, (C + 63) / 64, n_taps), dim3(256), 0, (hipStream_t)stream, x_ndhwc, D, H, W, C, kd, kh,
                       kw, pad[0], pad[1], pad[2], stride[0], stride[1], stride[2], OD, OH, OW, t0, lrow, out);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward of the fused epilogue y = relu(conv * scale + shift (+ identity)) of the frozen-BatchNorm training convolutions
// (nerfdet_amd/conv_train.py::ConvAffineAct; mmdet's Bottleneck.forward behind mmdet3d/models/detectors/nerfdet.py:140):
//     gm = dy [y > 0]        the gradient of the identity branch (written when asked for)
//     gs = gm * scale[c]     what the data / weight gradients of the convolution receive
// one pass over channels-last rows instead of three library launches (compare, multiply, multiply).
// ------------------------------------------------------------------------------------------------
// Each workgroup walks ONE contiguous range of float4s (so that, with `amax`, its maximum is a region's: conv_tilemin_read) and, when asked, leaves
// max |gs| in the slot -- gs is the operand of the layer's data and weight gradients, which in the fp16-pair arithmetic would otherwise each start
// with an ndet_amax_f32 pass over it.
__global__ __launch_bounds__(256) void k_relu_affine_bwd(const float4* __restrict__ g, const float4* __restrict__ y, const float4* __restrict__ scale,
                                                        int64_t n4, int c4, int relu, float4* __restrict__ gm, float4* __restrict__ gs, float* __restrict__ amax) {
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    float mx = 0.0f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        float4 v = g[i];
        if (relu) {
            const float4 t = y[i];
            v.x = t.x > 0.f ? v.x : 0.f; v.y = t.y > 0.f ? v.y : 0.f; v.z = t.z > 0.f ? v.z : 0.f; v.w = t.w > 0.f ? v.w : 0.f;
        }
        if (gm) gm[i] = v;
        const float4 s = scale[i % c4];
        const float4 o = make_float4(v.x * s.x, v.y * s.y, v.z * s.z, v.w * s.w);
        gs[i] = o;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    if (amax) conv_amax_commit(amax, mx);
}

static int relu_affine_bwd_entry(const char* fn, const float* dy, const float* y, const float* scale, int64_t rows, int C, int relu, float* d_identity,
                                 float* d_conv, float* d_conv_amax, void* stream) {
    NDET_REQUIRE(dy && scale && d_conv && (y || !relu), NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, NDET_E_UNSUPPORTED, "%s: C=%d must be a positive multiple of 4", fn, C);
    NDET_REQUIRE((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)scale | (uintptr_t)d_identity | (uintptr_t)d_conv) & 15) == 0, NDET_E_UNSUPPORTED,
                 "%s: pointers must be 16-byte aligned", fn);
    const int64_t n4 = rows * (C / 4);
    int64_t blocks = (n4 + 1023) / 1024;            // >= 4 float4s per thread
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_relu_affine_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)dy, (const float4*)y,
                       (const float4*)scale, n4, C / 4, relu, (float4*)d_identity, (float4*)d_conv, d_conv_amax);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

extern "C" int ndet_relu_affine_bwd(const float* dy, const float* y, const float* scale, int64_t rows, int C, int relu, float* d_identity,
                                    float* d_conv, void* stream) {
    return relu_affine_bwd_entry("ndet_relu_affine_bwd", dy, y, scale, rows, C, relu, d_identity, d_conv, nullptr, stream);
}

// the same pass leaving max |d_conv| in a zeroed amax slot (the fp16-pair data / weight gradients read their operand's scale from it)
extern "C" int ndet_relu_affine_bwd_amax(const float* dy, const float* y, const float* scale, int64_t rows, int C, int relu, float* d_identity,
                                         float* d_conv, float* d_conv_amax, void* stream) {
    NDET_REQUIRE(d_conv_amax != nullptr, NDET_E_INVALID, "ndet_relu_affine_bwd_amax: null slot");
    return relu_affine_bwd_entry("ndet_relu_affine_bwd_amax", dy, y, scale, rows, C, relu, d_identity, d_conv, d_conv_amax, stream);
}

// ------------------------------------------------------------------------------------------------
// dY of a convolution, channels-last (L output voxels x Cout), straight to the weight-gradient GEMM's "weight" operand: the three bf16
// planes of its channel-major rows, tiled per K step of 32 voxels -- (lrow/32, 3, Cout, 32), the layout of ndet_split_weights_bf16x3 --
// with zeros past L.  One pass instead of k_wgrad_rows (transpose to fp32 rows) + k_split_weights (read them back, split): autograd of
// nn.Conv3d / nn.Conv2d, mmdet3d/models/necks/imvoxelnet.py:22-67,233-260.  64 voxels x 64 channels per workgroup through an LDS
// transpose; a thread then owns 16 consecutive voxels of one channel: splits them and writes 32 bytes per plane.
// ------------------------------------------------------------------------------------------------
// F16: two fp16 planes of dy * conv_xscale(amax slot) per K step (the fp16-pair weight gradient; 1 / scale is read from the same slot by the GEMM)
template <bool F16>
__global__ __launch_bounds__(256) void k_wgrad_dy_planes(const float* __restrict__ g, int L, int C, int lrow, const float* __restrict__ amax, uint16_t* __restrict__ planes) {
    constexpr int WPL = F16 ? 2 : 3;
    const float xs = F16 ? conv_xscale(amax) : 1.0f;
    __shared__ float tile[64][65];
    const int j0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const bool cok = c0 + tx < C;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int vj = ty + 4 * i;
        tile[vj][tx] = (j0 + vj < L && cok) ? g[(int64_t)(j0 + vj) * C + c0 + tx] : 0.0f;
    }
    __syncthreads();
    const int co = c0 + tx, chunk = ty & 1, half = ty >> 1;                 // 64 channels x 2 K chunks x 2 halves of 16 voxels
    const int kc = (j0 >> 5) + chunk;
    if (!cok || kc * 32 >= lrow) return;
    uint32_t o[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float a = tile[chunk * 32 + half * 16 + 2 * i][tx], b = tile[chunk * 32 + half * 16 + 2 * i + 1][tx];
        spl_split2<F16 ? 1 : 0>(a, b, xs, o[0][i], o[1][i], o[2][i]);
    }
#pragma unroll
    for (int pl = 0; pl < WPL; ++pl) {
        uint4* dst = reinterpret_cast<uint4*>(planes + (((int64_t)kc * WPL + pl) * C + co) * 32 + half * 16);
        dst[0] = make_uint4(o[pl][0], o[pl][1], o[pl][2], o[pl][3]);
        dst[1] = make_uint4(o[pl][4], o[pl][5], o[pl][6], o[pl][7]);
    }
}

extern "C" int ndet_wgrad_dy_planes(const float* dy_rows_by_voxel, int L, int Cout, int lrow, uint16_t* planes, void* stream) {
    const char* fn = "ndet_wgrad_dy_planes";
    NDET_REQUIRE(dy_rows_by_voxel && planes, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(L > 0 && Cout > 0 && lrow >= L && lrow % 32 == 0, NDET_E_INVALID, "%s: bad sizes (L=%d, lrow=%d)", fn, L, lrow);
    NDET_REQUIRE((((uintptr_t)planes) & 15) == 0 && (Cout + 63) / 64 <= 65535, NDET_E_UNSUPPORTED, "%s: planes must be 16-byte aligned", fn);
    hipLaunchKernelGGL(k_wgrad_dy_planes<false>, dim3((lrow + 63) / 64, (Cout + 63) / 64), dim3(256), 0, (hipStream_t)stream, dy_rows_by_voxel, L, Cout, lrow,
                       (const float*)nullptr, planes);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// fp16-pair form: (lrow/32, 2, Cout, 32) planes of dy scaled by conv_xscale of its amax slot (ndet_amax_f32 / a producing epilogue)
extern "C" int ndet_wgrad_dy_planes_f16x2(const float* dy_rows_by_voxel, int L, int Cout, int lrow, const float* dy_amax, uint16_t* planes, void* stream) {
    const char* fn = "ndet_wgrad_dy_planes_f16x2";
    NDET_REQUIRE(dy_rows_by_voxel && planes && dy_amax, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(L > 0 && Cout > 0 && lrow >= L && lrow % 32 == 0, NDET_E_INVALID, "%s: bad sizes (L=%d, lrow=%d)", fn, L, lrow);
    NDET_REQUIRE((((uintptr_t)planes) & 15) == 0 && (Cout + 63) / 64 <= 65535, NDET_E_UNSUPPORTED, "%s: planes must be 16-byte aligned", fn);
    hipLaunchKernelGGL(k_wgrad_dy_planes<true>, dim3((lrow + 63) / 64, (Cout + 63) / 64), dim3(256), 0, (hipStream_t)stream, dy_rows_by_voxel, L, Cout, lrow,
                       dy_amax, planes);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}
