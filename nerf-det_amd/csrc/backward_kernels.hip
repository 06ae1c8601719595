// Backward passes of the hot-path kernels (training, BASELINE configs[2]).  The reference gets these from PyTorch
// autograd through its materialised tensors (nerfdet.py:164-261, projection.py:120-127, render_ray.py:71-93,196-247);
// here each is one kernel that recomputes the cheap geometry and scatters with float atomics
// (global_atomic_add_f32, one dword per lane, 256 contiguous bytes per wave instruction where the layout allows --
// the shape MI355X_MICROARCH.md "Global float atomics" prices at ~1.3 TB/s).  Summation order of the scatter is
// not fixed: gradients are reproducible to fp32 rounding, not bitwise.
#include "ndet_common.hpp"

#define BVOX_PER_TILE 16

// ------------------------------------------------------------------------------------------------
// K1 backward: d(features) from d(mean).   mean[n,:] = sum_{v sees n} feat[v, y, x, :] / (cnt_n + 1e-8)
//   => d feat[v,y,x,:] += g[n,:] / (cnt_n + 1e-8)      for every valid (n, v);   nothing flows where cnt == 0
// Same wave-per-voxel / lanes-over-channels / ballot walk as the forward kernel.
// ------------------------------------------------------------------------------------------------
template <int LAYOUT>
__global__ __launch_bounds__(256) void k_backproject_aggregate_bwd(const float* __restrict__ g, int n_views, int C, int h, int w,
                                                                   int64_t view_pitch, int row_pitch, const float* __restrict__ points, int N,
                                                                   const float* __restrict__ proj, float* __restrict__ dfeat, int n_tiles, int det) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = ndet_xcd_remap(blockIdx.x, n_tiles);
    for (int j = 0; j < BVOX_PER_TILE / 4; ++j) {
        const int n = tile * BVOX_PER_TILE + j * 4 + wave;
        if (n >= N) continue;
        const float px = points[n], py = points[N + n], pz = points[2 * N + n];
        // count first (the scale needs it), keeping each round's mask / offsets for the scatter
        int cnt = 0;
        for (int r0 = 0; r0 < n_views; r0 += 64) {
            const int v = r0 + lane;
            int xi, yi;
            const bool ok = v < n_views && ndet_project(proj + v * 12, px, py, pz, w, h, xi, yi);
            cnt += __popcll(__ballot(ok));
        }
        if (cnt == 0) continue;
        const float inv = 1.0f / ((float)cnt + 1e-8f);
        // this voxel's gradient row, scaled; the first 4 x 64 channels live in registers
        float gv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = lane + 64 * q;
            gv[q] = c0 < C ? (LAYOUT == NDET_LAYOUT_NC ? g[(int64_t)n * C + c0] : g[(int64_t)c0 * N + n]) * inv : 0.0f;
        }
        for (int r0 = 0; r0 < n_views; r0 += 64) {
            const int v = r0 + lane;
            int xi = 0, yi = 0;
            const bool ok = v < n_views && ndet_project(proj + v * 12, px, py, pz, w, h, xi, yi);
            const int off = yi * row_pitch + xi * C;
            unsigned long long m = __ballot(ok);
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1ull;
                const int o = __builtin_amdgcn_readlane(off, b);
                const int64_t base = (int64_t)(r0 + b) * view_pitch + o;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (lane + 64 * q < C) ndet_scatter_add(dfeat, base + lane + 64 * q, gv[q], det);  // 256 contiguous bytes per instruction
                for (int c0 = lane + 256; c0 < C; c0 += 64)
                    ndet_scatter_add(dfeat, base + c0, (LAYOUT == NDET_LAYOUT_NC ? g[(int64_t)n * C + c0] : g[(int64_t)c0 * N + n]) * inv, det);
            }
        }
    }
}

extern "C" int ndet_backproject_aggregate_bwd(const float* grad_mean, int grad_layout, int n_views, int C, int h, int w,
                                              int64_t view_pitch, int64_t row_pitch, const float* points, int N, const float* projection,
                                              float* grad_features_nhwc, void* stream) {
    const char* fn = "ndet_backproject_aggregate_bwd";
    NDET_REQUIRE(grad_mean && points && projection && grad_features_nhwc, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_views > 0 && C > 0 && h > 0 && w > 0 && N > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE(grad_layout == NDET_LAYOUT_CN || grad_layout == NDET_LAYOUT_NC, NDET_E_INVALID, "%s: bad layout", fn);
    NDET_REQUIRE((int64_t)h * row_pitch < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: one view exceeds 2^31 floats", fn);
    const int n_tiles = (N + BVOX_PER_TILE - 1) / BVOX_PER_TILE;
    if (grad_layout == NDET_LAYOUT_NC)
        hipLaunchKernelGGL(k_backproject_aggregate_bwd<NDET_LAYOUT_NC>, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, grad_mean, n_views, C, h, w,
                           view_pitch, (int)row_pitch, points, N, projection, grad_features_nhwc, n_tiles, g_ndet_deterministic_scatter);
    else
        hipLaunchKernelGGL(k_backproject_aggregate_bwd<NDET_LAYOUT_CN>, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, grad_mean, n_views, C, h, w,
                           view_pitch, (int)row_pitch, points, N, projection, grad_features_nhwc, n_tiles, g_ndet_deterministic_scatter);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------------
// K2 backward: d(mapped features), d(bias) from d(global_feat).
// Per voxel and channel c (values val_v = mapped row where the view sees the voxel, else the bias b_c):
//   S = sum_v val_v, mean = S/den, Q = sum_v (val_v-mean)^2, cov = exp(-Q/den), den = cnt + 1e-8
//   dL/dval_v = gm/den + (-gc*cov/den) * ( 2 (val_v - mean) - 2 (S - n_v*mean)/den )        (cov term 0 when cnt == 0)
// RGB channels carry no gradient (images are inputs).  Lanes over the cm mapped channels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_density_features_bwd(const float* __restrict__ gout, const float* __restrict__ mapped, int n_views, int cm,
                                                              int h, int w, int64_t mview_pitch, int mrow_pitch, const float* __restrict__ bias,
                                                              const float* __restrict__ points, int N, const float* __restrict__ proj,
                                                              float* __restrict__ dmapped, float* __restrict__ dbias, int n_tiles, int det) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = ndet_xcd_remap(blockIdx.x, n_tiles);
    const int F = 2 * (3 + cm);
    const bool active = lane < cm;
    const float b = active ? bias[lane] : 0.0f;
    float dbias_acc = 0.0f;
    for (int j = 0; j < BVOX_PER_TILE / 4; ++j) {
        const int n = tile * BVOX_PER_TILE + j * 4 + wave;
        if (n >= N) continue;
        const float px = points[n], py = points[N + n], pz = points[2 * N + n];
        // pass 0: S and cnt; pass 1: Q; pass 2: scatter
        float S = 0.f, Q = 0.f, mean = 0.f, den = 1.f, k1 = 0.f, k2 = 0.f, k3 = 0.f;
        int cnt = 0;
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
            float acc = 0.0f;
            int c_here = 0;
            for (int r0 = 0; r0 < n_views; r0 += 64) {
                const int v = r0 + lane;
                int xi = 0, yi = 0;
                const bool ok = v < n_views && ndet_project(proj + v * 12, px, py, pz, w, h, xi, yi);
                const int off = yi * mrow_pitch + xi * cm;
                unsigned long long m = __ballot(ok);
                c_here += __popcll(m);
                while (m) {
                    const int bb = __builtin_ctzll(m);
                    m &= m - 1ull;
                    const int o = __builtin_amdgcn_readlane(off, bb);
                    const int64_t idx = (int64_t)(r0 + bb) * mview_pitch + o + lane;
                    const float val = active ? mapped[idx] : 0.0f;
                    if (pass == 0) acc = acc + val;
                    else if (pass == 1) acc = acc + (val - mean) * (val - mean);
                    else if (active) ndet_scatter_add(dmapped, idx, k1 + k2 * (2.0f * (val - mean) - k3), det);
                }
            }
            const float n_inv = (float)(n_views - c_here);
            if (pass == 0) {
                cnt = c_here;
                den = (float)cnt + 1e-8f;
                S = acc + n_inv * b;
                mean = S / den;
            } else if (pass == 1) {
                Q = acc + n_inv * ((b - mean) * (b - mean));
                const float gm = active ? gout[(int64_t)n * F + 2 * (3 + lane)] : 0.0f;
                const float gc = active ? gout[(int64_t)n * F + 2 * (3 + lane) + 1] : 0.0f;
                const float cov = cnt == 0 ? 0.0f : expf(-(Q / den));  // forced-variance branch has zero slope
                k1 = gm / den;
                k2 = -gc * cov / den;
                k3 = 2.0f * (S - (float)n_views * mean) / den;
            } else {
                // views that do not see the voxel contributed the bias
                dbias_acc = dbias_acc + n_inv * (k1 + k2 * (2.0f * (b - mean) - k3));
            }
        }
    }
    if (active && dbias_acc != 0.0f) ndet_scatter_add(dbias, lane, dbias_acc, det);
}

extern "C" int ndet_density_features_bwd(const float* grad_global_feat, const float* mapped_nhwc, int n_views, int cm, int h, int w,
                                         int64_t mview_pitch, int64_t mrow_pitch, const float* bias, const float* points, int N,
                                         const float* projection, float* grad_mapped_nhwc, float* grad_bias, void* stream) {
    const char* fn = "ndet_density_features_bwd";
    NDET_REQUIRE(grad_global_feat && mapped_nhwc && bias && points && projection && grad_mapped_nhwc && grad_bias, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_views > 0 && cm > 0 && cm <= 61 && h > 0 && w > 0 && N > 0, NDET_E_INVALID, "%s: bad sizes", fn);
    NDET_REQUIRE((int64_t)h * mrow_pitch < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: one view exceeds 2^31 floats", fn);
    const int n_tiles = (N + BVOX_PER_TILE - 1) / BVOX_PER_TILE;
    hipLaunchKernelGGL(k_density_features_bwd, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, grad_global_feat, mapped_nhwc, n_views, cm, h, w,
                       mview_pitch, (int)mrow_pitch, bias, points, N, projection, grad_mapped_nhwc, grad_bias, n_tiles, g_ndet_deterministic_scatter);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------------
// K4 backward: d(mapped features) from d(globalfeat) of the ray sampler.
//   val_v = bilinear(feat_v), mean = sum_v mask_v wgt val_v, var = sum_v (val_v-mean)^2 / den, ev = exp(-var)
//   dL/dval_v = gm mask_v wgt + (-ge ev/den) (2 (val_v-mean) - 2 mask_v wgt sum_u (val_u - mean))
// then spread over the 4 taps with the bilinear weights.  Lanes over the d feature channels.
// ------------------------------------------------------------------------------------------------
struct RayHit { float nx, ny; bool mask; };

__device__ __forceinline__ RayHit bwd_ray_project(const float* __restrict__ KE, float x, float y, float z, float h, float w) {
    float q0 = KE[0] * x; q0 = fmaf(KE[1], y, q0); q0 = fmaf(KE[2], z, q0); q0 = q0 + KE[3];
    float q1 = KE[4] * x; q1 = fmaf(KE[5], y, q1); q1 = fmaf(KE[6], z, q1); q1 = q1 + KE[7];
    float q2 = KE[8] * x; q2 = fmaf(KE[9], y, q2); q2 = fmaf(KE[10], z, q2); q2 = q2 + KE[11];
    const float den = fmaxf(q2, 1e-8f);
    float px = fminf(fmaxf(q0 / den, -1e6f), 1e6f), py = fminf(fmaxf(q1 / den, -1e6f), 1e6f);
    RayHit r;
    r.mask = (px <= w - 1.0f) && (px >= 0.0f) && (py <= h - 1.0f) && (py >= 0.0f) && (q2 > 0.0f);
    r.nx = (2.0f * px) / (w - 1.0f) - 1.0f;
    r.ny = (2.0f * py) / (h - 1.0f) - 1.0f;
    return r;
}

struct Taps { int o00, o01, o10, o11; float w00, w01, w10, w11; };  // offsets < 0: tap outside the map

__device__ __forceinline__ Taps bilinear_taps(float nx, float ny, int Hs, int Ws, int row_pitch, int d) {
    Taps t;
    const float ix = ((nx + 1.0f) / 2.0f) * (float)(Ws - 1), iy = ((ny + 1.0f) / 2.0f) * (float)(Hs - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fminf(fmaxf(fx, -2.0f), (float)Ws + 1.0f), y0 = (int)fminf(fmaxf(fy, -2.0f), (float)Hs + 1.0f);
    const int x1 = x0 + 1, y1 = y0 + 1;
    t.w00 = ((fx + 1.0f) - ix) * ((fy + 1.0f) - iy);
    t.w01 = (ix - fx) * ((fy + 1.0f) - iy);
    t.w10 = ((fx + 1.0f) - ix) * (iy - fy);
    t.w11 = (ix - fx) * (iy - fy);
    const bool fin = (ix == ix) && (iy == iy);
    const bool xa = x0 >= 0 && x0 < Ws, xb = x1 >= 0 && x1 < Ws, ya = y0 >= 0 && y0 < Hs, yb = y1 >= 0 && y1 < Hs;
    t.o00 = (fin && ya && xa) ? y0 * row_pitch + x0 * d : -1;
    t.o01 = (fin && ya && xb) ? y0 * row_pitch + x1 * d : -1;
    t.o10 = (fin && yb && xa) ? y1 * row_pitch + x0 * d : -1;
    t.o11 = (fin && yb && xb) ? y1 * row_pitch + x1 * d : -1;
    return t;
}

__global__ __launch_bounds__(256) void k_ray_view_stats_bwd(const float* __restrict__ gglob, const float* __restrict__ pts, int n_points,
                                                            const float* __restrict__ KE, int n_views, float img_h, float img_w,
                                                            const float* __restrict__ feat, int d, int hf, int wf, int64_t fview_pitch,
                                                            int frow_pitch, float* __restrict__ dfeat, int det) {
    const int lane = threadIdx.x & 63;
    const int p = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // one wave per sample point
    if (p >= n_points) return;
    const bool active = lane < d;
    const int nch = 3 + d;
    const float x = pts[(int64_t)p * 3], y = pts[(int64_t)p * 3 + 1], z = pts[(int64_t)p * 3 + 2];
    int cnt = 0;
    for (int r0 = 0; r0 < n_views; r0 += 64) {
        const int v = r0 + lane;
        const bool m = v < n_views && bwd_ray_project(KE + v * 12, x, y, z, img_h, img_w).mask;
        cnt += __popcll(__ballot(m));
    }
    const float den = (float)cnt + 1e-8f, wgt = 1.0f / den;
    const float gm = active ? gglob[(int64_t)p * 2 * nch + 3 + lane] : 0.0f;
    const float ge = active ? gglob[(int64_t)p * 2 * nch + nch + 3 + lane] : 0.0f;
    float mean = 0.f, dsum = 0.f, k2 = 0.f;
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        float acc = 0.0f, acc2 = 0.0f;
        for (int r0 = 0; r0 < n_views; r0 += 64) {
            const int v = r0 + lane;
            RayHit hit; hit.nx = 0.f; hit.ny = 0.f; hit.mask = false;
            if (v < n_views) hit = bwd_ray_project(KE + v * 12, x, y, z, img_h, img_w);
            const unsigned long long mbits = __ballot(hit.mask);
            const int nv_here = min(64, n_views - r0);
            for (int b = 0; b < nv_here; ++b) {
                const float nx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hit.nx), b));
                const float ny = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hit.ny), b));
                const bool mv = (mbits >> b) & 1ull;
                const Taps t = bilinear_taps(nx, ny, hf, wf, frow_pitch, d);
                const float* base = feat + (int64_t)(r0 + b) * fview_pitch + lane;
                float val = 0.0f;
                if (active) {
                    if (t.o00 >= 0) val = val + base[t.o00] * t.w00;
                    if (t.o01 >= 0) val = val + base[t.o01] * t.w01;
                    if (t.o10 >= 0) val = val + base[t.o10] * t.w10;
                    if (t.o11 >= 0) val = val + base[t.o11] * t.w11;
                }
                if (pass == 0) {
                    if (mv) acc = acc + val * wgt;
                } else if (pass == 1) {
                    acc = acc + (val - mean) * (val - mean);
                    acc2 = acc2 + (val - mean);
                } else if (active) {
                    const float gv = (mv ? gm * wgt : 0.0f) + k2 * (2.0f * (val - mean) - (mv ? 2.0f * wgt * dsum : 0.0f));
                    const int64_t dbo = (int64_t)(r0 + b) * fview_pitch + lane;     // element index (8-byte elements in the deterministic mode)
                    if (gv != 0.0f) {
                        if (t.o00 >= 0) ndet_scatter_add(dfeat, dbo + t.o00, gv * t.w00, det);
                        if (t.o01 >= 0) ndet_scatter_add(dfeat, dbo + t.o01, gv * t.w01, det);
                        if (t.o10 >= 0) ndet_scatter_add(dfeat, dbo + t.o10, gv * t.w10, det);
                        if (t.o11 >= 0) ndet_scatter_add(dfeat, dbo + t.o11, gv * t.w11, det);
                    }
                }
            }
        }
        if (pass == 0) mean = acc;
        else if (pass == 1) {
            const float ev = expf(-(acc / den));
            k2 = -ge * ev / den;
            dsum = acc2;
        }
    }
}

extern "C" int ndet_ray_view_stats_bwd(const float* grad_global_feat, const float* pts, int n_points, const float* KE, int n_views, float img_h,
                                       float img_w, const float* feat_nhwc, int d, int hf, int wf, int64_t fview_pitch, int64_t frow_pitch,
                                       float* grad_feat_nhwc, void* stream) {
    const char* fn = "ndet_ray_view_stats_bwd";
    NDET_REQUIRE(grad_global_feat && pts && KE && feat_nhwc && grad_feat_nhwc, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_points > 0 && n_views > 0 && hf > 1 && wf > 1 && d > 0 && d <= 61, NDET_E_INVALID, "%s: bad sizes", fn);
    NDET_REQUIRE((int64_t)hf * frow_pitch < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: one view exceeds 2^31 floats", fn);
    const int64_t blocks = ((int64_t)n_points + 3) / 4;
    NDET_REQUIRE(blocks < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: too many points", fn);
    hipLaunchKernelGGL(k_ray_view_stats_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad_global_feat, pts, n_points, KE, n_views,
                       img_h, img_w, feat_nhwc, d, hf, wf, fview_pitch, (int)frow_pitch, grad_feat_nhwc, g_ndet_deterministic_scatter);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------------
// A11 backward: d(raw) from d(rgb_map), d(depth_map).  One thread per ray, reverse walk.
//   w_s = a_s T_s,  T_{s+1} = T_s (1 - a_s + 1e-10),  a_s = 1 - exp(-sigma_s)
//   dL/da_s = gw_s T_s - (sum_{k>s} gw_k w_k) / (1 - a_s + 1e-10),   dL/dsigma_s = dL/da_s (1 - a_s)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_composite_bwd(const float* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ Tfwd,
                                                      int R, int S, int white_bkgd, const float* __restrict__ zminmax,
                                                      const float* __restrict__ g_rgb, const float* __restrict__ g_depth, float* __restrict__ d_raw) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    // forward sums (the transmittance itself comes from the forward pass: dividing it back out is unstable once it underflows)
    float wsum = 0.f, wz = 0.f;
    for (int s = 0; s < S; ++s) {
        const int64_t i = (int64_t)r * S + s;
        const float wgt = (1.0f - expf(-raw[i * 4 + 3])) * Tfwd[i];
        wsum = wsum + wgt;
        wz = wz + wgt * z[i];
    }
    const float gr0 = g_rgb[r * 3], gr1 = g_rgb[r * 3 + 1], gr2 = g_rgb[r * 3 + 2];
    float gd = g_depth ? g_depth[r] : 0.0f;
    const float dn = wsum + 1e-8f, dep = wz / dn;
    if (dep < zminmax[0] || dep > zminmax[1]) gd = 0.0f;  // clamped: zero slope
    const float gbg = white_bkgd ? -(gr0 + gr1 + gr2) : 0.0f;  // rgb += 1 - sum(w)
    float suffix = 0.0f;  // sum_{k>s} gw_k w_k
    for (int s = S - 1; s >= 0; --s) {
        const int64_t i = (int64_t)r * S + s;
        const float4 q = *reinterpret_cast<const float4*>(raw + i * 4);
        const float a = 1.0f - expf(-q.w);
        const float om = (1.0f - a) + 1e-10f;
        const float T = Tfwd[i];
        const float wgt = a * T;
        const float gw = gr0 * q.x + gr1 * q.y + gr2 * q.z + gbg + gd * (z[i] * dn - wz) / (dn * dn);
        const float ga = gw * T - suffix / om;
        float4 o;
        o.x = gr0 * wgt; o.y = gr1 * wgt; o.z = gr2 * wgt;
        o.w = ga * (1.0f - a);
        *reinterpret_cast<float4*>(d_raw + i * 4) = o;
        suffix = suffix + gw * wgt;
    }
}

extern "C" int ndet_composite_bwd(const float* raw, const float* z_vals, const float* transparency, int R, int S, int white_bkgd,
                                  const float* zminmax, const float* grad_rgb, const float* grad_depth, float* grad_raw, void* stream) {
    const char* fn = "ndet_composite_bwd";
    NDET_REQUIRE(raw && z_vals && transparency && zminmax && grad_rgb && grad_raw, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(R > 0 && S > 0, NDET_E_INVALID, "%s: bad sizes", fn);
    NDET_REQUIRE((((uintptr_t)raw | (uintptr_t)grad_raw) & 15) == 0, NDET_E_UNSUPPORTED, "%s: raw / grad_raw must be 16-byte aligned", fn);
    hipLaunchKernelGGL(k_composite_bwd, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, raw, z_vals, transparency, R, S, white_bkgd, zminmax,
                       grad_rgb, grad_depth, grad_raw);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}
