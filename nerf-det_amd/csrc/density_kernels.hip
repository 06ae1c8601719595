// K2, packed form: per-voxel NeRF conditioning rows (mmdet3d/models/detectors/nerfdet.py:234-253) for gfx950.
//
// Same statement as k_density_features in volume_kernels.hip (kept for cm % 4 != 0 and for more than 128 views):
//   value of channel c in view v = mapped[v, y4, x4, c] where the stride-4 projection sees the voxel, else the Linear's bias
//                                = rgb[v, c, y1, x1]    where the stride-1 projection sees it, else 0
//   mean = sum over ALL views / (cnt + 1e-8) (cnt = views seeing it in the stride-4 map; not zeroed at cnt == 0),
//   cov  = exp(-sum over ALL views (value - mean)^2 / (cnt + 1e-8)), 0 where cnt == 0.
// What changed is the mapping onto the machine, the same scheme as the packed ray sampler (ray_stats_kernels.hip):
//   * a lane holds one channel QUAD (one 16-byte load per view instead of four 4-byte ones), one more lane the three colour planes;
//   * both projections of a (voxel, view) pair are evaluated once, lanes over views, and parked in LDS as element offsets;
//   * each lane walks only views that see the voxel in ITS map (set bits of its ballot), eight gathers per trip;
//   * one walk: the sum for the mean and the shifted sums for the variance (pivot = the fill value), see the kernel's comment.
// Compiled with -ffp-contract=off.
#include "ndet_common.hpp"

#define DK_ROUNDS 2  // view rounds of 64 kept in registers: n_views <= 128

// One voxel per wavefront.  The voxel's views are dealt round-robin to SPLIT = 64 / (cm/4 + 1) lane groups (7 at cm = 32), each group
// the cm/4 + 1 channel-quad lanes of the scheme above: a voxel seen by all 50 views is ONE trip of eight gathers per lane instead
// of a chain of seven, and the grid is 25 600 waves instead of 3 658 -- the launch used to last as long as its longest chain
// (3.6 waves per SIMD, all resident at once).  The groups share the pivot (the fill value: the Linear's bias / 0), so their partial
// sums add; they meet in LDS.  With pivot = fill the views that do not see the voxel drop out of the shifted sums:
//     sum_all (v - mean)^2 = sum_seen (v - fill)^2 - 2 (mean - fill) sum_seen (v - fill) + n_views (mean - fill)^2.
template <int DUMMY>
__global__ __launch_bounds__(256) void k_density_features_packed(const float* __restrict__ mapped, int n_views, int cm, int h, int w,
                                                                 int mview_pitch, int mrow_pitch, const float* __restrict__ bias,
                                                                 const float* __restrict__ rgb, int H, int W, int rsv, int rsc, int rsy,
                                                                 const float* __restrict__ points, int N, const float* __restrict__ proj,
                                                                 const float* __restrict__ rgb_proj, float* __restrict__ out, int n_blocks,
                                                                 int nvp) {
    extern __shared__ int2 s_off[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lps = (cm >> 2) + 1;
    const int SPLIT = 64 / lps;
    const int grp = lane / lps, sub = lane - grp * lps;
    const int blk = ndet_xcd_remap(blockIdx.x, n_blocks);      // neighbouring voxel blocks hit the same pixels: keep them on one XCD's L2
    const int n = blk * 4 + wave;                              // this wave's voxel
    int2* rec = s_off + (size_t)wave * nvp;
    float4* part = reinterpret_cast<float4*>(s_off + (size_t)4 * nvp) + (size_t)wave * 64 * 3;   // [lane][acc, q, s1]
    const int rounds = (n_views + 63) >> 6;
    const bool live = n < N;

    // ---- phase 1: lanes over views, both projections of every (voxel, view) pair once ----
    unsigned long long mf[DK_ROUNDS], mr[DK_ROUNDS];
#pragma unroll
    for (int r = 0; r < DK_ROUNDS; ++r) { mf[r] = 0ull; mr[r] = 0ull; }
    if (live) {
        const float px = points[n], py = points[N + n], pz = points[2 * N + n];
#pragma unroll
        for (int r = 0; r < DK_ROUNDS; ++r) {
            if (r < rounds) {
                const int v = r * 64 + lane;
                bool okf = false, okr = false;
                if (v < n_views) {
                    int xf, yf, xr, yr;
                    okf = ndet_project(proj + v * 12, px, py, pz, w, h, xf, yf);
                    okr = ndet_project(rgb_proj + v * 12, px, py, pz, W, H, xr, yr);
                    rec[v] = make_int2(v * mview_pitch + yf * mrow_pitch + xf * cm, v * rsv + yr * rsy + xr);
                }
                mf[r] = __ballot(okf);
                mr[r] = __ballot(okr);
            }
        }
    }
    __syncthreads();

    // ---- phase 2: lanes over (view group, channel quad) ----
    const bool on = live && grp < SPLIT;
    const bool is_rgb = sub == 0;
    const int fq = sub - 1;
    float4 fill = make_float4(0.f, 0.f, 0.f, 0.f);            // what a view that does not see the voxel contributes (nerfdet.py:233)
    if (!is_rgb && on) fill = *reinterpret_cast<const float4*>(bias + 4 * fq);
    const float* fbase = mapped + 4 * max(fq, 0);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), q = acc, s1 = acc;
    auto fetch = [&](int idx) -> float4 {
        const int2 o = rec[idx];
        if (is_rgb) {
            const float* p = rgb + o.y;
            return make_float4(p[0], p[rsc], p[2 * rsc], 0.0f);
        }
        return *reinterpret_cast<const float4*>(fbase + o.x);
    };
    auto take = [&](const float4& v) {
        acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
        const float dx = v.x - fill.x, dy = v.y - fill.y, dz = v.z - fill.z, dw = v.w - fill.w;
        q.x = q.x + dx * dx; q.y = q.y + dy * dy; q.z = q.z + dz * dz; q.w = q.w + dw * dw;
        s1.x = s1.x + dx; s1.y = s1.y + dy; s1.z = s1.z + dz; s1.w = s1.w + dw;
    };
    // the views of group g: bits g, g + SPLIT, g + 2 SPLIT, ... of the round's mask
    unsigned long long stripe = 0ull;
    for (int b = grp; b < 64; b += SPLIT) stripe |= 1ull << b;
#pragma unroll
    for (int r = 0; r < DK_ROUNDS; ++r) {
        if (r >= rounds) break;
        unsigned long long m = on ? ((is_rgb ? mr[r] : mf[r]) & stripe) : 0ull;
        while (__ballot(m != 0ull) != 0ull) {
            bool hh[8];
            int bb[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                hh[k] = m != 0ull;
                bb[k] = hh[k] ? __builtin_ctzll(m) : 0;
                m = hh[k] ? (m & (m - 1ull)) : 0ull;
            }
            float4 vv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                vv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (hh[k]) vv[k] = fetch(r * 64 + bb[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (hh[k]) take(vv[k]);
        }
    }
    // ---- the groups' partial sums meet in LDS; group 0 finishes the voxel ----
    part[lane * 3 + 0] = acc;
    part[lane * 3 + 1] = q;
    part[lane * 3 + 2] = s1;
    __syncthreads();
    if (!(on && grp == 0)) return;
    for (int g2 = 1; g2 < SPLIT; ++g2) {
        const float4 a2 = part[(g2 * lps + sub) * 3 + 0], q2 = part[(g2 * lps + sub) * 3 + 1], s2 = part[(g2 * lps + sub) * 3 + 2];
        acc.x += a2.x; acc.y += a2.y; acc.z += a2.z; acc.w += a2.w;
        q.x += q2.x; q.y += q2.y; q.z += q2.z; q.w += q2.w;
        s1.x += s2.x; s1.y += s2.y; s1.z += s2.z; s1.w += s2.w;
    }
    int cnt = 0, n_mine = 0;
#pragma unroll
    for (int r = 0; r < DK_ROUNDS; ++r) {
        cnt += __popcll(mf[r]);
        n_mine += __popcll(is_rgb ? mr[r] : mf[r]);
    }
    const float denom = (float)cnt + 1e-8f;
    const float nu = (float)(n_views - n_mine), nv = (float)n_views;
    auto finish = [&](float a, float qq, float ss, float fl, float& mean, float& cov) {
        const float sum = a + nu * fl;
        mean = sum / denom;                                   // NOT zeroed at cnt == 0 (nerfdet.py:241)
        const float dm = mean - fl;
        float s = qq - 2.0f * dm * ss + nv * (dm * dm);
        s = fmaxf(s, 0.0f);                                   // a sum of squares: rounding may leave -1 ulp
        float var = s / denom;
        if (cnt == 0) var = 1e6f;                             // nerfdet.py:249
        cov = expf(-var);
    };
    float4 mean, cov;
    finish(acc.x, q.x, s1.x, fill.x, mean.x, cov.x);
    finish(acc.y, q.y, s1.y, fill.y, mean.y, cov.y);
    finish(acc.z, q.z, s1.z, fill.z, mean.z, cov.z);
    finish(acc.w, q.w, s1.w, fill.w, mean.w, cov.w);
    const int F = 2 * (3 + cm);
    float* row = out + (int64_t)n * F + (is_rgb ? 0 : 2 * (3 + 4 * fq));   // interleaved [mean_c, cov_c] (nerfdet.py:251-253)
    *reinterpret_cast<float2*>(row + 0) = make_float2(mean.x, cov.x);
    *reinterpret_cast<float2*>(row + 2) = make_float2(mean.y, cov.y);
    *reinterpret_cast<float2*>(row + 4) = make_float2(mean.z, cov.z);
    if (!is_rgb) *reinterpret_cast<float2*>(row + 6) = make_float2(mean.w, cov.w);
}

extern "C" int ndet_density_features_packed(const float* mapped_nhwc, int n_views, int cm, int h, int w, int64_t mview_pitch,
                                            int64_t mrow_pitch, const float* bias, const float* rgb, int H, int W, int64_t rsv,
                                            int64_t rsc, int64_t rsy, const float* points, int N, const float* projection,
                                            const float* rgb_projection, float* global_feat, void* stream) {
    const char* fn = "ndet_density_features_packed";
    NDET_REQUIRE(mapped_nhwc && bias && rgb && points && projection && rgb_projection && global_feat, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_views > 0 && cm > 0 && h > 0 && w > 0 && H > 0 && W > 0 && N > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE(cm % 4 == 0 && cm <= 128, NDET_E_UNSUPPORTED, "%s: cm=%d must be a multiple of 4, at most 128 (use ndet_density_features)", fn, cm);
    NDET_REQUIRE(n_views <= 64 * DK_ROUNDS, NDET_E_UNSUPPORTED, "%s: %d views exceed %d (use ndet_density_features)", fn, n_views, 64 * DK_ROUNDS);
    NDET_REQUIRE(mrow_pitch >= (int64_t)w * cm && mview_pitch >= (int64_t)h * mrow_pitch && rsy >= W && rsc >= 0 && rsv >= 0, NDET_E_INVALID,
                 "%s: pitches smaller than the maps", fn);
    NDET_REQUIRE((int64_t)n_views * mview_pitch < ((int64_t)1 << 31) && (int64_t)n_views * rsv + 3 * rsc < ((int64_t)1 << 31), NDET_E_UNSUPPORTED,
                 "%s: a source tensor exceeds 2^31 floats", fn);
    NDET_REQUIRE(mview_pitch % 4 == 0 && mrow_pitch % 4 == 0 && (((uintptr_t)mapped_nhwc | (uintptr_t)bias) & 15) == 0, NDET_E_UNSUPPORTED,
                 "%s: mapped features / bias must keep channel quads 16-byte aligned", fn);
    NDET_REQUIRE(((uintptr_t)global_feat & 7) == 0, NDET_E_UNSUPPORTED, "%s: global_feat must be 8-byte aligned", fn);
    const int nvp = ((n_views + 63) / 64) * 64;
    const int lds = 4 * nvp * (int)sizeof(int2) + 4 * 64 * 3 * (int)sizeof(float4);
    const int64_t blocks = ((int64_t)N + 3) / 4;               // one voxel per wavefront
    NDET_REQUIRE(blocks < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: too many voxels", fn);
    hipLaunchKernelGGL(k_density_features_packed<0>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, mapped_nhwc, n_views, cm, h, w,
                       (int)mview_pitch, (int)mrow_pitch, bias, rgb, H, W, (int)rsv, (int)rsc, (int)rsy, points, N, projection, rgb_projection,
                       global_feat, (int)blocks, nvp);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}
