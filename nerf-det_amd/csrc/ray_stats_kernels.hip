// K4, packed form: fused Projector.compute + compute_mask_points of the NeRF ray branch (SURVEY.md 8a rows A7 + A8;
// mmdet3d/models/model_utils/projection.py:91-151, model_utils/render_ray.py:71-93,299-303) and its backward, for gfx950.
//
// Differences from the generic kernel in ray_kernels.hip (kept for d % 4 != 0 and for more than 128 views):
//   * every (sample, view) pair is projected ONCE: lanes over views, the normalised coordinates go to LDS, the in-image/in-front
//     mask and the "near" mask (at least one bilinear tap can fall inside a map) stay in registers as ballots;
//   * views whose taps all fall outside a map contribute exactly 0 to the masked mean and exactly mean^2 to the reference's
//     unmasked variance sum (render_ray.py:86-87): they are counted, never gathered -- only the set bits of the near mask are walked;
//   * the source images are read from an (n_v,H,W,4) copy (ndet_pack_rgb_nhwc4), so an image pixel is one 16-byte load like a
//     feature pixel's channel quad: a sample occupies d/4 feature lanes + 1 image lane, 64/(d/4+1) samples share a wave
//     (7 at d = 32: 63 of 64 lanes busy) and every lane runs the same instruction stream (its map size, pitches and base pointer are
//     per-lane constants);
//   * two views per trip, eight independent 16-byte loads in flight per lane;
//   * the variance is accumulated in the same walk with the first gathered value as pivot:
//         sum_near (v - mean)^2 = sum (v - c)^2 - 2 (mean - c) sum (v - c) + n_near (mean - c)^2,   + n_far mean^2
//     (shifted-data form: conditioned like the two-pass sum when the pivot lies inside the data, no second gather).
// Compiled with -ffp-contract=off.
#include "ndet_common.hpp"

#define RS_ROUNDS 2  // view rounds of 64 kept in registers: n_views <= 128

// ------------------------------------------------------------------------------------------------
// (n_v,3,H,W) with arbitrary view / plane / row strides -> dense (n_v,H,W,4), 4th component 0
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_rgb_nhwc4(const float* __restrict__ rgb, int64_t total, int H, int W, int64_t rsv, int64_t rsc,
                                                        int64_t rsy, float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int64_t v = i / ((int64_t)W * H);
    const float* b = rgb + v * rsv + (int64_t)y * rsy + x;
    out[i] = make_float4(b[0], b[rsc], b[2 * rsc], 0.0f);
}

extern "C" int ndet_pack_rgb_nhwc4(const float* rgb, int n_views, int H, int W, int64_t rsv, int64_t rsc, int64_t rsy, float* out_nhwc4,
                                   void* stream) {
    const char* fn = "ndet_pack_rgb_nhwc4";
    NDET_REQUIRE(rgb && out_nhwc4, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_views > 0 && H > 0 && W > 0, NDET_E_INVALID, "%s: bad sizes", fn);
    NDET_REQUIRE(((uintptr_t)out_nhwc4 & 15) == 0, NDET_E_UNSUPPORTED, "%s: output must be 16-byte aligned", fn);
    const int64_t total = (int64_t)n_views * H * W;
    NDET_REQUIRE((total + 255) / 256 < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: too many pixels", fn);
    hipLaunchKernelGGL(k_pack_rgb_nhwc4, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rgb, total, H, W, rsv, rsc,
                       rsy, reinterpret_cast<float4*>(out_nhwc4));
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------------
// device pieces
// ------------------------------------------------------------------------------------------------
struct RsHit {
    float nx, ny;
    bool mask;
};

// projection.py:42-64,37-40,24-35 with the camera row in registers; same k-ordered FMA chain as ray_project / K1
__device__ __forceinline__ RsHit rs_project(const float (&k)[12], float x, float y, float z, float h, float w) {
    float q0 = k[0] * x;
    q0 = fmaf(k[1], y, q0);
    q0 = fmaf(k[2], z, q0);
    q0 = q0 + k[3];
    float q1 = k[4] * x;
    q1 = fmaf(k[5], y, q1);
    q1 = fmaf(k[6], z, q1);
    q1 = q1 + k[7];
    float q2 = k[8] * x;
    q2 = fmaf(k[9], y, q2);
    q2 = fmaf(k[10], z, q2);
    q2 = q2 + k[11];
    const float den = fmaxf(q2, 1e-8f);
    float px = q0 / den, py = q1 / den;
    px = fminf(fmaxf(px, -1e6f), 1e6f);
    py = fminf(fmaxf(py, -1e6f), 1e6f);
    RsHit r;
    r.mask = (px <= w - 1.0f) && (px >= 0.0f) && (py <= h - 1.0f) && (py >= 0.0f) && (q2 > 0.0f);
    r.nx = (2.0f * px) / (w - 1.0f) - 1.0f;
    r.ny = (2.0f * py) / (h - 1.0f) - 1.0f;
    return r;
}

// F.grid_sample(align_corners=True) source coordinate; a tap can only be inside the map for -1 < i < size (NaN: never)
__device__ __forceinline__ bool rs_near(float nx, float ny, int Hs, int Ws) {
    const float ix = ((nx + 1.0f) / 2.0f) * (float)(Ws - 1);
    const float iy = ((ny + 1.0f) / 2.0f) * (float)(Hs - 1);
    return ix > -1.0f && ix < (float)Ws && iy > -1.0f && iy < (float)Hs;
}

struct RsTaps {
    int o00, o01, o10, o11;      // element offsets inside one view (clamped into the map: always loadable)
    float w00, w01, w10, w11;    // bilinear weights, 0 for a tap outside the map (zero padding)
};

__device__ __forceinline__ RsTaps rs_taps(float nx, float ny, int Hs, int Ws, int row_pitch, int pix) {
    const float ix = ((nx + 1.0f) / 2.0f) * (float)(Ws - 1);
    const float iy = ((ny + 1.0f) / 2.0f) * (float)(Hs - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    // tap validity on the UNCLAMPED coordinates (a view that is near for one map may be far outside the other: the walk covers the
    // union of both near sets); the clamp below only keeps the (then zero-weighted) loads inside the map
    const bool xa = fx >= 0.0f && fx <= (float)(Ws - 1), xb = fx >= -1.0f && fx <= (float)(Ws - 2);
    const bool ya = fy >= 0.0f && fy <= (float)(Hs - 1), yb = fy >= -1.0f && fy <= (float)(Hs - 2);
    const int x0 = (int)fminf(fmaxf(fx, -1.0f), (float)(Ws - 1));
    const int y0 = (int)fminf(fmaxf(fy, -1.0f), (float)(Hs - 1));
    const int x0c = max(x0, 0), x1c = min(x0 + 1, Ws - 1), y0c = max(y0, 0), y1c = min(y0 + 1, Hs - 1);
    RsTaps t;
    t.w00 = (ya && xa) ? ((fx + 1.0f) - ix) * ((fy + 1.0f) - iy) : 0.0f;
    t.w01 = (ya && xb) ? (ix - fx) * ((fy + 1.0f) - iy) : 0.0f;
    t.w10 = (yb && xa) ? ((fx + 1.0f) - ix) * (iy - fy) : 0.0f;
    t.w11 = (yb && xb) ? (ix - fx) * (iy - fy) : 0.0f;
    t.o00 = y0c * row_pitch + x0c * pix;
    t.o01 = y0c * row_pitch + x1c * pix;
    t.o10 = y1c * row_pitch + x0c * pix;
    t.o11 = y1c * row_pitch + x1c * pix;
    return t;
}

__device__ __forceinline__ float4 rs_blend(const float4& a, const float4& b, const float4& c, const float4& d, const RsTaps& t) {
    float4 o;   // the reference's order: nw, ne, sw, se
    o.x = 0.0f + a.x * t.w00; o.x = o.x + b.x * t.w01; o.x = o.x + c.x * t.w10; o.x = o.x + d.x * t.w11;
    o.y = 0.0f + a.y * t.w00; o.y = o.y + b.y * t.w01; o.y = o.y + c.y * t.w10; o.y = o.y + d.y * t.w11;
    o.z = 0.0f + a.z * t.w00; o.z = o.z + b.z * t.w01; o.z = o.z + c.z * t.w10; o.z = o.z + d.z * t.w11;
    o.w = 0.0f + a.w * t.w00; o.w = o.w + b.w * t.w01; o.w = o.w + c.w * t.w10; o.w = o.w + d.w * t.w11;
    return o;
}

struct RsAcc {
    float4 acc, q, s1, piv;   // masked-mean numerator, sum (v-c)^2, sum (v-c), pivot c
    bool have;
};

__device__ __forceinline__ void rs_take(RsAcc& a, const float4& v, bool valid, float wgt) {
    if (valid) {
        a.acc.x = a.acc.x + v.x * wgt; a.acc.y = a.acc.y + v.y * wgt; a.acc.z = a.acc.z + v.z * wgt; a.acc.w = a.acc.w + v.w * wgt;
    }
    if (!a.have) { a.piv = v; a.have = true; }
    const float dx = v.x - a.piv.x, dy = v.y - a.piv.y, dz = v.z - a.piv.z, dw = v.w - a.piv.w;
    a.q.x = a.q.x + dx * dx; a.q.y = a.q.y + dy * dy; a.q.z = a.q.z + dz * dz; a.q.w = a.q.w + dw * dw;
    a.s1.x = a.s1.x + dx; a.s1.y = a.s1.y + dy; a.s1.z = a.s1.z + dz; a.s1.w = a.s1.w + dw;
}

__device__ __forceinline__ float rs_var_sum(float q, float s1, float piv, float mean, float n_near, float n_far) {
    const float dm = mean - piv;
    float s = q - 2.0f * dm * s1 + n_near * (dm * dm);
    s = fmaxf(s, 0.0f);                 // a sum of squares: rounding may leave -1 ulp
    return s + n_far * (mean * mean);   // views with no tap inside a map sample exactly 0
}

// ------------------------------------------------------------------------------------------------
// the kernel.  FWD: lanes of a sample = [image quad | d/4 feature quads]; BWD: d/4 feature quads only (images carry no gradient)
// ------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void k_ray_stats_packed(const float* __restrict__ pts, int n_points, const float* __restrict__ KE, int n_views,
                                                          float img_h, float img_w, const float* __restrict__ rgb4, int H, int W,
                                                          const float* __restrict__ feat, int d, int hf, int wf, int fview_pitch, int frow_pitch,
                                                          float* __restrict__ glob, uint8_t* __restrict__ pixel_mask, int* __restrict__ view_count,
                                                          const float* __restrict__ gglob, float* __restrict__ dfeat, int nvp, int det) {
    extern __shared__ float2 s_rec[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lps = (d >> 2) + (BWD ? 0 : 1);          // lanes per sample
    const int G = 64 / lps;                            // samples per wave
    const int g = lane / lps, sub = lane - g * lps;
    const int p_base = (blockIdx.x * 4 + wave) * G;
    float2* rec = s_rec + (size_t)wave * G * nvp;
    const int rounds = (n_views + 63) >> 6;

    // ---- phase 1: lanes over views.  Camera rows in registers, one projection per (sample, view) ----
    float cam[RS_ROUNDS][12];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int v = r * 64 + lane;
#pragma unroll
        for (int k = 0; k < 12; ++k) cam[r][k] = (r < rounds && v < n_views) ? KE[v * 12 + k] : 0.0f;
    }
    unsigned long long my_valid[RS_ROUNDS], my_near[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) { my_valid[r] = 0ull; my_near[r] = 0ull; }
    for (int gg = 0; gg < G; ++gg) {
        const int p = p_base + gg;
        if (p >= n_points) break;   // wave-uniform
        const float x = pts[(int64_t)p * 3 + 0], y = pts[(int64_t)p * 3 + 1], z = pts[(int64_t)p * 3 + 2];
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; ++r) {
            if (r < rounds) {
                const int v = r * 64 + lane;
                bool m = false, nr = false;
                if (v < n_views) {
                    const RsHit hit = rs_project(cam[r], x, y, z, img_h, img_w);
                    m = hit.mask;
                    nr = rs_near(hit.nx, hit.ny, hf, wf) || (!BWD && rs_near(hit.nx, hit.ny, H, W));
                    rec[gg * nvp + v] = make_float2(hit.nx, hit.ny);
                }
                const unsigned long long vb = __ballot(m), nb = __ballot(nr);
                if (g == gg) { my_valid[r] = vb; my_near[r] = nb; }
            }
        }
    }
    __syncthreads();

    // ---- phase 2: lanes over (sample, channel quad) ----
    const int my_p = p_base + g;
    const bool on = g < G && my_p < n_points;
    const bool is_rgb = !BWD && sub == 0;
    const int fq = BWD ? sub : sub - 1;                       // feature quad of this lane
    const int Hs = is_rgb ? H : hf, Ws = is_rgb ? W : wf;
    const int vpitch = is_rgb ? H * W * 4 : fview_pitch, rpitch = is_rgb ? W * 4 : frow_pitch, pix = is_rgb ? 4 : d;
    const float* base = is_rgb ? rgb4 : feat + 4 * fq;
    int cnt = 0, n_near = 0;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) { cnt += __popcll(my_valid[r]); n_near += __popcll(my_near[r]); }
    const float denom = (float)cnt + 1e-8f;
    const float wgt = 1.0f / denom;                           // mask / (sum(mask) + 1e-8), render_ray.py:83
    RsAcc a;
    a.acc = a.q = a.s1 = a.piv = make_float4(0.f, 0.f, 0.f, 0.f);
    a.have = false;
    const float2* myrec = rec + (on ? g : 0) * nvp;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        if (r >= rounds) break;
        unsigned long long m = on ? my_near[r] : 0ull;
        const unsigned long long vm = my_valid[r];
        while (__ballot(m != 0ull) != 0ull) {                 // until every sample of the wave has walked its near views
            const bool h0 = m != 0ull;
            const int b0 = h0 ? __builtin_ctzll(m) : 0;
            m = h0 ? (m & (m - 1ull)) : 0ull;
            const bool h1 = m != 0ull;
            const int b1 = h1 ? __builtin_ctzll(m) : 0;
            m = h1 ? (m & (m - 1ull)) : 0ull;
            const float2 r0 = h0 ? myrec[r * 64 + b0] : make_float2(0.f, 0.f);
            const float2 r1 = h1 ? myrec[r * 64 + b1] : make_float2(0.f, 0.f);
            const RsTaps t0 = rs_taps(r0.x, r0.y, Hs, Ws, rpitch, pix), t1 = rs_taps(r1.x, r1.y, Hs, Ws, rpitch, pix);
            const float* v0 = base + (r * 64 + b0) * vpitch;
            const float* v1 = base + (r * 64 + b1) * vpitch;
            float4 a00 = make_float4(0.f, 0.f, 0.f, 0.f), a01 = a00, a10 = a00, a11 = a00, c00 = a00, c01 = a00, c10 = a00, c11 = a00;
            if (h0) {
                a00 = *reinterpret_cast<const float4*>(v0 + t0.o00);
                a01 = *reinterpret_cast<const float4*>(v0 + t0.o01);
                a10 = *reinterpret_cast<const float4*>(v0 + t0.o10);
                a11 = *reinterpret_cast<const float4*>(v0 + t0.o11);
            }
            if (h1) {
                c00 = *reinterpret_cast<const float4*>(v1 + t1.o00);
                c01 = *reinterpret_cast<const float4*>(v1 + t1.o01);
                c10 = *reinterpret_cast<const float4*>(v1 + t1.o10);
                c11 = *reinterpret_cast<const float4*>(v1 + t1.o11);
            }
            if (h0) rs_take(a, rs_blend(a00, a01, a10, a11, t0), (vm >> b0) & 1ull, wgt);
            if (h1) rs_take(a, rs_blend(c00, c01, c10, c11, t1), (vm >> b1) & 1ull, wgt);
        }
    }
    const float nn = (float)n_near, nf = (float)(n_views - n_near);
    const float4 mean = a.acc;
    float4 ev = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) {
        ev.x = expf(-(rs_var_sum(a.q.x, a.s1.x, a.piv.x, mean.x, nn, nf) / denom));
        ev.y = expf(-(rs_var_sum(a.q.y, a.s1.y, a.piv.y, mean.y, nn, nf) / denom));
        ev.z = expf(-(rs_var_sum(a.q.z, a.s1.z, a.piv.z, mean.z, nn, nf) / denom));
        ev.w = expf(-(rs_var_sum(a.q.w, a.s1.w, a.piv.w, mean.w, nn, nf) / denom));
    }
    const int nch = 3 + d;
    if (!BWD) {
        if (!on) return;
        float* row = glob + (int64_t)my_p * 2 * nch;          // cat([mean, exp(-var)], dim=-1), render_ray.py:303
        if (is_rgb) {
            row[0] = mean.x; row[1] = mean.y; row[2] = mean.z;
            row[nch + 0] = ev.x; row[nch + 1] = ev.y; row[nch + 2] = ev.z;
            pixel_mask[my_p] = cnt > 1 ? 1 : 0;               // render_ray.py:301
            if (view_count) view_count[my_p] = cnt;
        } else {
            const int c = 3 + 4 * fq;
            row[c + 0] = mean.x; row[c + 1] = mean.y; row[c + 2] = mean.z; row[c + 3] = mean.w;
            row[nch + c + 0] = ev.x; row[nch + c + 1] = ev.y; row[nch + c + 2] = ev.z; row[nch + c + 3] = ev.w;
        }
        return;
    }
    // ---- backward: dL/dval_v = gm mask_v wgt + k2 (2 (val_v - mean) - 2 mask_v wgt sum_u (val_u - mean)),  k2 = -ge ev / den;
    // views without a tap inside the map have no pixel to receive anything.
    // The scatter runs with ONE channel per lane: a (sample, view, tap) then is one atomic instruction over d consecutive floats
    // (a 128-byte run at d = 32), which the L2 atomic units take ~4x faster than the quad layout's 16-byte-strided dwords.  The
    // per-channel constants and the sample's masks change lanes through LDS. ----
    float4* s_const = reinterpret_cast<float4*>(s_rec + (size_t)4 * G * nvp) + (size_t)wave * G * d;   // [g][c] = {mean, k2, gm wgt, 2 wgt dsum}
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(reinterpret_cast<float4*>(s_rec + (size_t)4 * G * nvp) + (size_t)4 * G * d)
                                 + (size_t)wave * G * 4;                                             // [g] = {near0, near1, valid0, valid1}
    if (on) {
        const float* grow = gglob + (int64_t)my_p * 2 * nch + 3 + 4 * fq;
        const float nv = (float)n_views;
        const float m4[4] = {mean.x, mean.y, mean.z, mean.w}, e4[4] = {ev.x, ev.y, ev.z, ev.w};
        const float s4[4] = {a.s1.x, a.s1.y, a.s1.z, a.s1.w}, p4[4] = {a.piv.x, a.piv.y, a.piv.z, a.piv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gm = grow[k], ge = grow[nch + k];
            const float dsum = (s4[k] + nn * p4[k]) - nv * m4[k];          // sum over ALL views of (val - mean); far views sample 0
            s_const[g * d + 4 * fq + k] = make_float4(m4[k], -ge * e4[k] / denom, gm * wgt, 2.0f * wgt * dsum);
        }
        if (sub == 0) {
            s_mask[g * 4 + 0] = my_near[0]; s_mask[g * 4 + 1] = my_near[1];
            s_mask[g * 4 + 2] = my_valid[0]; s_mask[g * 4 + 3] = my_valid[1];
        }
    }
    __syncthreads();
    const int spb = 64 / d;                                   // samples scattered at a time
    const int s2 = lane / d, ch = lane - s2 * d;
    for (int s0 = 0; s0 < G; s0 += spb) {
        const int sg = s0 + s2;
        if (s2 >= spb || sg >= G || p_base + sg >= n_points) continue;
        const float4 kc = s_const[sg * d + ch];
        const float2* srec = rec + sg * nvp;
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; ++r) {
            if (r >= rounds) break;
            unsigned long long m = s_mask[sg * 4 + r];
            const unsigned long long vm = s_mask[sg * 4 + 2 + r];
            while (m != 0ull) {
                const int b = __builtin_ctzll(m);
                m &= m - 1ull;
                const float2 rc = srec[r * 64 + b];
                const RsTaps t = rs_taps(rc.x, rc.y, hf, wf, frow_pitch, d);
                const float* v0 = feat + (r * 64 + b) * fview_pitch + ch;
                float val = 0.0f + v0[t.o00] * t.w00;
                val = val + v0[t.o01] * t.w01;
                val = val + v0[t.o10] * t.w10;
                val = val + v0[t.o11] * t.w11;
                const bool mv = (vm >> b) & 1ull;
                const float gv = (mv ? kc.z : 0.0f) + kc.y * (2.0f * (val - kc.x) - (mv ? kc.w : 0.0f));
                const int64_t dvo = (int64_t)(r * 64 + b) * fview_pitch + ch;      // element index: the deterministic mode's elements are 8 bytes wide
                if (t.w00 != 0.0f) ndet_scatter_add(dfeat, dvo + t.o00, gv * t.w00, det);
                if (t.w01 != 0.0f) ndet_scatter_add(dfeat, dvo + t.o01, gv * t.w01, det);
                if (t.w10 != 0.0f) ndet_scatter_add(dfeat, dvo + t.o10, gv * t.w10, det);
                if (t.w11 != 0.0f) ndet_scatter_add(dfeat, dvo + t.o11, gv * t.w11, det);
            }
        }
    }
}

static int rs_check(const char* fn, int n_points, int n_views, int H, int W, int d, int hf, int wf, int64_t fview_pitch, int64_t frow_pitch,
                    bool bwd, int* lds_bytes, int64_t* blocks, int* nvp_out) {
    NDET_REQUIRE(n_points > 0 && n_views > 0 && H > 1 && W > 1 && hf > 1 && wf > 1 && d > 0, NDET_E_INVALID, "%s: bad sizes", fn);
    NDET_REQUIRE(d % 4 == 0 && d <= 128, NDET_E_UNSUPPORTED, "%s: d=%d must be a multiple of 4, at most 128 (use the generic entry point)", fn, d);
    NDET_REQUIRE(n_views <= 64 * RS_ROUNDS, NDET_E_UNSUPPORTED, "%s: %d views exceed %d (use the generic entry point)", fn, n_views, 64 * RS_ROUNDS);
    NDET_REQUIRE(frow_pitch >= (int64_t)wf * d && fview_pitch >= (int64_t)hf * frow_pitch, NDET_E_INVALID, "%s: pitches smaller than the map", fn);
    NDET_REQUIRE((int64_t)n_views * fview_pitch < ((int64_t)1 << 31) && (int64_t)n_views * H * W * 4 < ((int64_t)1 << 31), NDET_E_UNSUPPORTED,
                 "%s: a source tensor exceeds 2^31 floats", fn);
    NDET_REQUIRE(fview_pitch % 4 == 0 && frow_pitch % 4 == 0, NDET_E_UNSUPPORTED, "%s: pitches must keep pixels 16-byte aligned", fn);
    const int lps = d / 4 + (bwd ? 0 : 1);
    NDET_REQUIRE(lps <= 64, NDET_E_UNSUPPORTED, "%s: d too large", fn);
    const int G = 64 / lps;
    const int nvp = ((n_views + 63) / 64) * 64;
    *nvp_out = nvp;
    *lds_bytes = 4 * G * nvp * (int)sizeof(float2);
    if (bwd) {
        NDET_REQUIRE(d <= 64, NDET_E_UNSUPPORTED, "%s: the scatter phase maps one channel per lane: d <= 64", fn);
        *lds_bytes += 4 * G * d * (int)sizeof(float4) + 4 * G * 4 * (int)sizeof(unsigned long long);
    }
    NDET_REQUIRE(*lds_bytes <= 64 * 1024, NDET_E_UNSUPPORTED, "%s: %d bytes of LDS needed (d too small for this many views)", fn, *lds_bytes);
    *blocks = ((int64_t)n_points + 4 * G - 1) / (4 * G);
    NDET_REQUIRE(*blocks < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: too many points", fn);
    return NDET_OK;
}

extern "C" int ndet_ray_view_stats_packed(const float* pts, int n_points, const float* KE, int n_views, float img_h, float img_w,
                                          const float* rgb_nhwc4, int H, int W, const float* feat_nhwc, int d, int hf, int wf,
                                          int64_t fview_pitch, int64_t frow_pitch, float* global_feat, uint8_t* pixel_mask, int* view_count,
                                          void* stream) {
    const char* fn = "ndet_ray_view_stats_packed";
    NDET_REQUIRE(pts && KE && rgb_nhwc4 && feat_nhwc && global_feat && pixel_mask, NDET_E_INVALID, "%s: null pointer", fn);
    int lds = 0, nvp = 0;
    int64_t blocks = 0;
    const int rc = rs_check(fn, n_points, n_views, H, W, d, hf, wf, fview_pitch, frow_pitch, false, &lds, &blocks, &nvp);
    if (rc != NDET_OK) return rc;
    NDET_REQUIRE((((uintptr_t)rgb_nhwc4 | (uintptr_t)feat_nhwc) & 15) == 0, NDET_E_UNSUPPORTED, "%s: sources must be 16-byte aligned", fn);
    hipLaunchKernelGGL(k_ray_stats_packed<false>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, pts, n_points, KE, n_views, img_h,
                       img_w, rgb_nhwc4, H, W, feat_nhwc, d, hf, wf, (int)fview_pitch, (int)frow_pitch, global_feat, pixel_mask, view_count,
                       (const float*)nullptr, (float*)nullptr, nvp, 0);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

extern "C" int ndet_ray_view_stats_packed_bwd(const float* grad_global_feat, const float* pts, int n_points, const float* KE, int n_views,
                                              float img_h, float img_w, const float* feat_nhwc, int d, int hf, int wf, int64_t fview_pitch,
                                              int64_t frow_pitch, float* grad_feat_nhwc, void* stream) {
    const char* fn = "ndet_ray_view_stats_packed_bwd";
    NDET_REQUIRE(grad_global_feat && pts && KE && feat_nhwc && grad_feat_nhwc, NDET_E_INVALID, "%s: null pointer", fn);
    int lds = 0, nvp = 0;
    int64_t blocks = 0;
    const int rc = rs_check(fn, n_points, n_views, 2, 2, d, hf, wf, fview_pitch, frow_pitch, true, &lds, &blocks, &nvp);
    if (rc != NDET_OK) return rc;
    NDET_REQUIRE(((uintptr_t)feat_nhwc & 15) == 0, NDET_E_UNSUPPORTED, "%s: features must be 16-byte aligned", fn);
    hipLaunchKernelGGL(k_ray_stats_packed<true>, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, pts, n_points, KE, n_views, img_h,
                       img_w, (const float*)nullptr, 2, 2, feat_nhwc, d, hf, wf, (int)fview_pitch, (int)frow_pitch, (float*)nullptr,
                       (uint8_t*)nullptr, (int*)nullptr, grad_global_feat, grad_feat_nhwc, nvp, g_ndet_deterministic_scatter);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}
