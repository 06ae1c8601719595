// NeRF ray branch of the hot path on gfx950.
//   k_sample_rays        A9   z-values (+ stratified jitter from a caller-supplied uniform stream) and points
//   k_ray_view_stats     A7+A8 fused: project every sample into every source view, bilinear-sample RGB + mapped
//                        features (align_corners=True, zero padding), masked mean / unmasked-sum variance over
//                        views -> (R*S, 2*(3+d)) conditioning rows + the ">1 view" sample mask.  The reference's
//                        (R,S,n_v,35) tensor (917 MB at cfg2 training shapes) is never materialised.
//   k_project_sample     A7 exact API form (materialises rgb_feat and mask like Projector.compute)
//   k_composite          A11  alpha compositing along each ray
// Compiled with -ffp-contract=off.
#include "ndet_common.hpp"

// ------------------------------------------------------------------------------------------------
// A9  sample_along_camera_ray  (render_ray.py:145-189, inv_uniform=False)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sample_rays(const float* __restrict__ ray_o, const float* __restrict__ ray_d, int R, int S,
                                                     float near, float far, const float* __restrict__ t_rand,
                                                     float* __restrict__ pts, float* __restrict__ z_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * S) return;
    const int r = i / S, s = i % S;
    // start + i*step with step = (far - near) / (S - 1), every op rounded on its own
    const float step = (far - near) / (float)(S - 1);
    float z = near + (float)s * step;
    if (t_rand != nullptr) {
        const float zp = near + (float)(s - 1) * step, zn = near + (float)(s + 1) * step;
        const float lower = s == 0 ? z : 0.5f * (z + zp);          // mids = .5 * (z[1:] + z[:-1])
        const float upper = s == S - 1 ? z : 0.5f * (zn + z);
        z = lower + (upper - lower) * t_rand[i];
    }
    z_out[i] = z;
#pragma unroll
    for (int k = 0; k < 3; ++k) pts[(int64_t)i * 3 + k] = z * ray_d[r * 3 + k] + ray_o[r * 3 + k];
}

extern "C" int ndet_sample_along_rays(const float* ray_o, const float* ray_d, int R, int S, float near, float far,
                                      const float* t_rand, float* pts, float* z_vals, void* stream) {
    const char* fn = "ndet_sample_along_rays";
    NDET_REQUIRE(ray_o && ray_d && pts && z_vals, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(R > 0 && S > 1, NDET_E_INVALID, "%s: need R > 0 and S > 1", fn);
    NDET_REQUIRE(near > 0.f && far > 0.f && far > near, NDET_E_INVALID, "%s: need 0 < near < far", fn);  // render_ray.py:161
    const int64_t total = (int64_t)R * S;
    NDET_REQUIRE(total < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: too many samples", fn);
    hipLaunchKernelGGL(k_sample_rays, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ray_o, ray_d, R, S,
                       near, far, t_rand, pts, z_vals);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------------
// shared device pieces of A7
// ------------------------------------------------------------------------------------------------
struct ViewHit {
    float nx, ny;  // normalised coordinates in [-1, 1] (projection.py:37-40)
    bool mask;     // in-image and in front (projection.py:149-150)
};

// projection.py:42-64 + :37-40 + :24-35.  KE = (K @ E) rows 0..2, the same k-ordered FMA chain as K1.
__device__ __forceinline__ ViewHit ray_project(const float* __restrict__ KE, float x, float y, float z, float h, float w) {
    float q0 = KE[0] * x;
    q0 = fmaf(KE[1], y, q0);
    q0 = fmaf(KE[2], z, q0);
    q0 = q0 + KE[3];
    float q1 = KE[4] * x;
    q1 = fmaf(KE[5], y, q1);
    q1 = fmaf(KE[6], z, q1);
    q1 = q1 + KE[7];
    float q2 = KE[8] * x;
    q2 = fmaf(KE[9], y, q2);
    q2 = fmaf(KE[10], z, q2);
    q2 = q2 + KE[11];
    const float den = fmaxf(q2, 1e-8f);                       // torch.clamp(min=1e-8)
    float px = q0 / den, py = q1 / den;
    px = fminf(fmaxf(px, -1e6f), 1e6f);                       // torch.clamp(-1e6, 1e6)
    py = fminf(fmaxf(py, -1e6f), 1e6f);
    ViewHit r;
    r.mask = (px <= w - 1.0f) && (px >= 0.0f) && (py <= h - 1.0f) && (py >= 0.0f) && (q2 > 0.0f);
    r.nx = (2.0f * px) / (w - 1.0f) - 1.0f;
    r.ny = (2.0f * py) / (h - 1.0f) - 1.0f;
    return r;
}

// F.grid_sample(bilinear, zeros, align_corners=True) of ONE channel at normalised (nx, ny).
// `at(yy, xx)` must return the element's address.
template <typename At>
__device__ __forceinline__ float bilinear_zeros(float nx, float ny, int Hs, int Ws, At at) {
    const float ix = ((nx + 1.0f) / 2.0f) * (float)(Ws - 1);
    const float iy = ((ny + 1.0f) / 2.0f) * (float)(Hs - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    // clamp before the int conversion only to keep the cast defined; out-of-range taps are dropped below anyway
    const int x0 = (int)fminf(fmaxf(fx, -2.0f), (float)Ws + 1.0f);
    const int y0 = (int)fminf(fmaxf(fy, -2.0f), (float)Hs + 1.0f);
    const int x1 = x0 + 1, y1 = y0 + 1;
    const float nw = ((fx + 1.0f) - ix) * ((fy + 1.0f) - iy);
    const float ne = (ix - fx) * ((fy + 1.0f) - iy);
    const float sw = ((fx + 1.0f) - ix) * (iy - fy);
    const float se = (ix - fx) * (iy - fy);
    const bool xin0 = x0 >= 0 && x0 < Ws, xin1 = x1 >= 0 && x1 < Ws;
    const bool yin0 = y0 >= 0 && y0 < Hs, yin1 = y1 >= 0 && y1 < Hs;
    const bool finite = (ix == ix) && (iy == iy);
    float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
    if (finite && yin0 && xin0) v00 = *at(y0, x0);
    if (finite && yin0 && xin1) v01 = *at(y0, x1);
    if (finite && yin1 && xin0) v10 = *at(y1, x0);
    if (finite && yin1 && xin1) v11 = *at(y1, x1);
    float out = 0.0f;
    out = out + v00 * nw;
    out = out + v01 * ne;
    out = out + v10 * sw;
    out = out + v11 * se;
    return out;
}

// ------------------------------------------------------------------------------------------------
// K4  fused Projector.compute + compute_mask_points   (projection.py:91-151, render_ray.py:71-93,301-303)
//
// One wavefront per sample point.  Phase 1, lanes over VIEWS: project, normalise, in-image/in-front mask ->
// ballot.  Phase 2, lanes over the 3 + d CHANNELS (lane c < 3: RGB plane c of the full-resolution image, else
// channel c-3 of the channels-last mapped feature map): for every view the wave pulls (nx, ny) out of the lane
// that computed it (v_readlane) and each lane bilinearly samples its own channel; pass A accumulates the masked
// mean, pass B the squared deviations over ALL views (the reference's unmasked-sum convention).  Pass B re-reads
// the same taps from L1/L2.
// ------------------------------------------------------------------------------------------------
#define RAY_SPW 4  // samples per wave (sequential)

__global__ __launch_bounds__(256) void k_ray_view_stats(const float* __restrict__ pts, int n_samples_total,
                                                        const float* __restrict__ KE, int n_views, float img_h, float img_w,
                                                        const float* __restrict__ rgb, int H, int W, int64_t rsv, int64_t rsc, int rsy,
                                                        const float* __restrict__ feat, int d, int hf, int wf, int64_t fview_pitch,
                                                        int frow_pitch, float* __restrict__ glob, uint8_t* __restrict__ pixel_mask,
                                                        int* __restrict__ view_count) {
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // global wave id
    const int nch = 3 + d;
    const bool is_rgb = lane < 3;
    const bool active = lane < nch;
    for (int j = 0; j < RAY_SPW; ++j) {
        const int p = gw * RAY_SPW + j;
        if (p >= n_samples_total) return;  // wave-uniform
        const float x = pts[(int64_t)p * 3 + 0], y = pts[(int64_t)p * 3 + 1], z = pts[(int64_t)p * 3 + 2];
        // ---- count pass (lanes over views) to get sum(mask) first: the mean weights need it ----
        int cnt = 0;
        for (int r0 = 0; r0 < n_views; r0 += 64) {
            const int v = r0 + lane;
            bool m = false;
            if (v < n_views) m = ray_project(KE + v * 12, x, y, z, img_h, img_w).mask;
            cnt += __popcll(__ballot(m));
        }
        const float denom = (float)cnt + 1e-8f;
        const float wgt = 1.0f / denom;  // weight = mask / (sum(mask) + 1e-8), render_ray.py:83
        float mean = 0.0f, var = 0.0f;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            float acc = 0.0f;
            for (int r0 = 0; r0 < n_views; r0 += 64) {
                const int v = r0 + lane;
                ViewHit hit;
                hit.nx = 0.f; hit.ny = 0.f; hit.mask = false;
                if (v < n_views) hit = ray_project(KE + v * 12, x, y, z, img_h, img_w);
                const unsigned long long mbits = __ballot(hit.mask);
                const int nv_here = min(64, n_views - r0);
                for (int b = 0; b < nv_here; ++b) {
                    const float nx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hit.nx), b));
                    const float ny = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hit.ny), b));
                    const bool mv = (mbits >> b) & 1ull;
                    float val = 0.0f;
                    if (active) {
                        if (is_rgb) {
                            const float* base = rgb + (int64_t)(r0 + b) * rsv + (int64_t)lane * rsc;
                            val = bilinear_zeros(nx, ny, H, W, [&](int yy, int xx) { return base + (int64_t)yy * rsy + xx; });
                        } else {
                            const float* base = feat + (int64_t)(r0 + b) * fview_pitch + (lane - 3);
                            val = bilinear_zeros(nx, ny, hf, wf, [&](int yy, int xx) { return base + (int64_t)yy * frow_pitch + xx * d; });
                        }
                    }
                    if (pass == 0) {
                        if (mv) acc = acc + val * wgt;
                    } else {
                        const float dd = val - mean;
                        acc = acc + dd * dd;
                    }
                }
            }
            if (pass == 0) mean = acc;
            else var = acc;
        }
        var = var / denom;
        const float ev = expf(-var);
        if (active) {
            glob[(int64_t)p * 2 * nch + lane] = mean;         // cat([mean, var], dim=-1), render_ray.py:303
            glob[(int64_t)p * 2 * nch + nch + lane] = ev;
        }
        if (lane == 0) {
            pixel_mask[p] = cnt > 1 ? 1 : 0;                 // render_ray.py:301
            if (view_count) view_count[p] = cnt;
        }
    }
}

extern "C" int ndet_ray_view_stats(const float* pts, int n_points, const float* KE, int n_views, float img_h, float img_w,
                                   const float* rgb, int H, int W, int64_t rsv, int64_t rsc, int64_t rsy,
                                   const float* feat_nhwc, int d, int hf, int wf, int64_t fview_pitch, int64_t frow_pitch,
                                   float* global_feat, uint8_t* pixel_mask, int* view_count, void* stream) {
    const char* fn = "ndet_ray_view_stats";
    NDET_REQUIRE(pts && KE && rgb && feat_nhwc && global_feat && pixel_mask, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_points > 0 && n_views > 0 && H > 1 && W > 1 && hf > 1 && wf > 1 && d > 0, NDET_E_INVALID, "%s: bad sizes", fn);
    NDET_REQUIRE(d <= 61, NDET_E_UNSUPPORTED, "%s: d=%d feature channels do not fit one wavefront (max 61)", fn, d);
    NDET_REQUIRE((int64_t)H * rsy < ((int64_t)1 << 31) && (int64_t)hf * frow_pitch < ((int64_t)1 << 31), NDET_E_UNSUPPORTED,
                 "%s: one view exceeds 2^31 floats", fn);
    const int64_t waves = ((int64_t)n_points + RAY_SPW - 1) / RAY_SPW;
    const int64_t blocks = (waves + 3) / 4;
    NDET_REQUIRE(blocks < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: too many points", fn);
    hipLaunchKernelGGL(k_ray_view_stats, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pts, n_points, KE, n_views, img_h,
                       img_w, rgb, H, W, rsv, rsc, (int)rsy, feat_nhwc, d, hf, wf, fview_pitch, (int)frow_pitch, global_feat, pixel_mask,
                       view_count);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------------
// A7 exact API form: rgb_feat (P, n_views, 3+d) and mask (P, n_views) like Projector.compute
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_project_sample(const float* __restrict__ pts, int n_points, const float* __restrict__ KE,
                                                        int n_views, float img_h, float img_w, const float* __restrict__ rgb, int H, int W,
                                                        int64_t rsv, int64_t rsc, int rsy, const float* __restrict__ feat, int d, int hf,
                                                        int wf, int64_t fview_pitch, int frow_pitch, float* __restrict__ rgb_feat,
                                                        float* __restrict__ mask) {
    // one thread per (point, view, channel-slot); channel fastest so stores coalesce
    const int nch = 3 + d;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_points * n_views * nch) return;
    const int c = (int)(i % nch);
    const int v = (int)((i / nch) % n_views);
    const int64_t p = i / ((int64_t)nch * n_views);
    const ViewHit hit = ray_project(KE + v * 12, pts[p * 3], pts[p * 3 + 1], pts[p * 3 + 2], img_h, img_w);
    float val;
    if (c < 3) {
        const float* base = rgb + (int64_t)v * rsv + (int64_t)c * rsc;
        val = bilinear_zeros(hit.nx, hit.ny, H, W, [&](int yy, int xx) { return base + (int64_t)yy * rsy + xx; });
    } else {
        const float* base = feat + (int64_t)v * fview_pitch + (c - 3);
        val = bilinear_zeros(hit.nx, hit.ny, hf, wf, [&](int yy, int xx) { return base + (int64_t)yy * frow_pitch + xx * d; });
    }
    rgb_feat[i] = val;
    if (c == 0) mask[p * n_views + v] = hit.mask ? 1.0f : 0.0f;
}

extern "C" int ndet_project_sample(const float* pts, int n_points, const float* KE, int n_views, float img_h, float img_w,
                                   const float* rgb, int H, int W, int64_t rsv, int64_t rsc, int64_t rsy, const float* feat_nhwc, int d,
                                   int hf, int wf, int64_t fview_pitch, int64_t frow_pitch, float* rgb_feat, float* mask, void* stream) {
    const char* fn = "ndet_project_sample";
    NDET_REQUIRE(pts && KE && rgb && feat_nhwc && rgb_feat && mask, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_points > 0 && n_views > 0 && H > 1 && W > 1 && hf > 1 && wf > 1 && d > 0, NDET_E_INVALID, "%s: bad sizes", fn);
    const int64_t total = (int64_t)n_points * n_views * (3 + d);
    const int64_t blocks = (total + 255) / 256;
    NDET_REQUIRE(blocks < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: too many elements", fn);
    hipLaunchKernelGGL(k_project_sample, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pts, n_points, KE, n_views, img_h,
                       img_w, rgb, H, W, rsv, rsc, (int)rsy, feat_nhwc, d, hf, wf, fview_pitch, (int)frow_pitch, rgb_feat, mask);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------------
// A11  raw2outputs  (render_ray.py:196-247).  One thread per ray, samples walked in order so the transmittance
// is the same left-to-right product torch.cumprod forms.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_composite(const float* __restrict__ raw, const float* __restrict__ z, const uint8_t* __restrict__ pmask,
                                                  int R, int S, int white_bkgd, const float* __restrict__ zminmax, float* __restrict__ rgb_map,
                                                  float* __restrict__ depth_map, float* __restrict__ weights, uint8_t* __restrict__ ray_mask,
                                                  float* __restrict__ alpha_out, float* __restrict__ T_out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float T = 1.0f, c0 = 0.f, c1 = 0.f, c2 = 0.f, wsum = 0.f, wz = 0.f;
    int seen = 0;
    for (int s = 0; s < S; ++s) {
        const int64_t i = (int64_t)r * S + s;
        const float4 q = *reinterpret_cast<const float4*>(raw + i * 4);
        const float a = 1.0f - expf(-q.w);       // sigma2alpha without the interval (render_ray.py:209)
        const float wgt = a * T;
        alpha_out[i] = a;
        T_out[i] = T;                            // exclusive product, T_0 = 1
        weights[i] = wgt;
        c0 = c0 + wgt * q.x;
        c1 = c1 + wgt * q.y;
        c2 = c2 + wgt * q.z;
        wsum = wsum + wgt;
        wz = wz + wgt * z[i];
        T = T * ((1.0f - a) + 1e-10f);           // cumprod(1 - alpha + 1e-10)
        if (pmask) seen += pmask[i] ? 1 : 0;
    }
    if (white_bkgd) {
        const float bg = 1.0f - wsum;
        c0 = c0 + bg; c1 = c1 + bg; c2 = c2 + bg;
    }
    rgb_map[r * 3 + 0] = c0;
    rgb_map[r * 3 + 1] = c1;
    rgb_map[r * 3 + 2] = c2;
    float dep = wz / (wsum + 1e-8f);
    dep = fminf(fmaxf(dep, zminmax[0]), zminmax[1]);  // clamp to the GLOBAL z range (render_ray.py:236)
    depth_map[r] = dep;
    if (ray_mask) ray_mask[r] = seen > 8 ? 1 : 0;      // render_ray.py:230
}

extern "C" int ndet_composite(const float* raw, const float* z_vals, const uint8_t* pixel_mask, int R, int S, int white_bkgd,
                              const float* zminmax, float* rgb_map, float* depth_map, float* weights, uint8_t* ray_mask, float* alpha,
                              float* transparency, void* stream) {
    const char* fn = "ndet_composite";
    NDET_REQUIRE(raw && z_vals && zminmax && rgb_map && depth_map && weights && alpha && transparency, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(R > 0 && S > 0, NDET_E_INVALID, "%s: bad sizes", fn);
    NDET_REQUIRE(((uintptr_t)raw & 15) == 0, NDET_E_UNSUPPORTED, "%s: raw must be 16-byte aligned", fn);
    hipLaunchKernelGGL(k_composite, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, raw, z_vals, pixel_mask, R, S, white_bkgd,
                       zminmax, rgb_map, depth_map, weights, ray_mask, alpha, transparency);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}
