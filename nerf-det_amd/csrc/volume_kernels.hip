// Voxel-grid side of the NeRF-Det hot path on gfx950: lattice, projection + nearest gather,
// multi-view aggregation, density conditioning features, alpha gating.
// SURVEY.md section 8a rows A2-A6.  Compiled with -ffp-contract=off: every fused multiply-add in
// here is an explicit fmaf().
#include "ndet_common.hpp"

#include <stdarg.h>

// ------------------------------------------------------------------------------------------
// error plumbing (host)
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void ndet_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int g_ndet_deterministic_scatter = 0;      // ndet_common.hpp::ndet_scatter_add
extern "C" int ndet_version(void) { return 107; }
extern "C" const char* ndet_last_error(void) { return g_err; }

#define VOX_PER_TILE 16  // one workgroup = 4 waves x 4 voxels = 16 consecutive voxels (one z column at Z=16)
// K2 gathers one float per lane and view (a 140-byte row): a batch is cheap in registers and the kernel is latency bound, so it
// keeps many more views in flight than K1's 1-KiB rows allow
#ifndef K2_BATCH
#define K2_BATCH 4
#endif
#ifndef GATHER_BATCH
#define GATHER_BATCH 4  // independent 1-KiB row loads a wave keeps in flight per voxel (tools/tune_k1.py: 4 beats 8/12/16)
#endif
#ifndef K1_MIN_WAVES
#define K1_MIN_WAVES 6  // __launch_bounds__ 2nd argument (waves per SIMD) for K1; tuned with tools/tune_k1.py
#endif

// ------------------------------------------------------------------------------------------
// A2  get_points  (nerfdet.py:380-390)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_get_points(float* __restrict__ pts, int nx, int ny, int nz,
                                                    float vx, float vy, float vz, float ox, float oy, float oz) {
    const int N = nx * ny * nz;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int iz = n % nz;
    const int iy = (n / nz) % ny;
    const int ix = n / (nz * ny);
    // idx * voxel_size, rounded, then + shifted origin, rounded (two torch ops in the reference)
    pts[n] = (float)ix * vx + ox;
    pts[N + n] = (float)iy * vy + oy;
    pts[2 * N + n] = (float)iz * vz + oz;
}

extern "C" int ndet_get_points(float* points, int nx, int ny, int nz, const float* vs, const float* org, void* stream) {
    NDET_REQUIRE(points && vs && org, NDET_E_INVALID, "ndet_get_points: null pointer");
    NDET_REQUIRE(nx > 0 && ny > 0 && nz > 0, NDET_E_INVALID, "ndet_get_points: n_voxels must be positive");
    // new_origin = origin - n_voxels / 2. * voxel_size   (fp32, un-fused; nerfdet.py:388)
    volatile float hx = (float)nx / 2.0f, hy = (float)ny / 2.0f, hz = (float)nz / 2.0f;
    volatile float mx = hx * vs[0], my = hy * vs[1], mz = hz * vs[2];
    const float ox = org[0] - mx, oy = org[1] - my, oz = org[2] - mz;
    const int N = nx * ny * nz;
    hipLaunchKernelGGL(k_get_points, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, points, nx, ny, nz,
                       vs[0], vs[1], vs[2], ox, oy, oz);
    NDET_CHECK_LAUNCH("ndet_get_points");
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------
// layout helper: (n, c, hw) -> (n, hw, c)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ src, float* __restrict__ dst, int c, int hw) {
    __shared__ float tile[32][33];
    const int img = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const float* s = src + (int64_t)img * c * hw;
    float* d = dst + (int64_t)img * c * hw;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int cc = c0 + ty + k, pp = p0 + tx;
        tile[ty + k][tx] = (cc < c && pp < hw) ? s[(int64_t)cc * hw + pp] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int pp = p0 + ty + k, cc = c0 + tx;
        if (pp < hw && cc < c) d[(int64_t)pp * c + cc] = tile[tx][ty + k];
    }
}

extern "C" int ndet_nchw_to_nhwc(const float* src, float* dst, int n, int c, int hw, void* stream) {
    NDET_REQUIRE(src && dst, NDET_E_INVALID, "ndet_nchw_to_nhwc: null pointer");
    NDET_REQUIRE(n > 0 && c > 0 && hw > 0, NDET_E_INVALID, "ndet_nchw_to_nhwc: sizes must be positive");
    NDET_REQUIRE(n <= 65535 && (c + 31) / 32 <= 65535, NDET_E_UNSUPPORTED, "ndet_nchw_to_nhwc: grid too large");
    dim3 grid((hw + 31) / 32, (c + 31) / 32, n);
    hipLaunchKernelGGL(k_nchw_to_nhwc, grid, dim3(256), 0, (hipStream_t)stream, src, dst, c, hw);
    NDET_CHECK_LAUNCH("ndet_nchw_to_nhwc");
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------
// measurement aid: the float4 copy whose rate is the empirical HBM ceiling the gather kernels are priced against
// ------------------------------------------------------------------------------------------
typedef float ndet_f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy_float4(const ndet_f4v* __restrict__ src, ndet_f4v* __restrict__ dst, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

extern "C" int ndet_hbm_copy(const float* src, float* dst, int64_t n_floats, void* stream) {
    NDET_REQUIRE(src && dst, NDET_E_INVALID, "ndet_hbm_copy: null pointer");
    NDET_REQUIRE(n_floats > 0 && n_floats % 4 == 0, NDET_E_INVALID, "ndet_hbm_copy: the length must be a positive multiple of 4 floats");
    NDET_REQUIRE(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, NDET_E_INVALID, "ndet_hbm_copy: pointers must be 16-byte aligned");
    const int64_t n4 = n_floats / 4;
    // one 16-byte element per thread, non-temporal: the best of the shapes swept on MI355X (grid-stride with 1/2/4/8 elements per thread,
    // 1 024 ... n4/256 workgroups, temporal / non-temporal: 5.4 - 6.66 TB/s on 1 GiB buffers)
    const int64_t blocks = (n4 + 255) / 256;
    NDET_REQUIRE(blocks <= 0x7fffffff, NDET_E_UNSUPPORTED, "ndet_hbm_copy: at most 2^39 floats per launch");
    const int grid = (int)blocks;
    hipLaunchKernelGGL(k_copy_float4, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const ndet_f4v*)src, (ndet_f4v*)dst, n4);
    NDET_CHECK_LAUNCH("ndet_hbm_copy");
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------
// A3  backproject, materialising form (exact reference API; not the hot path)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_backproject(const float* __restrict__ feat, int C, int h, int w,
                                                     int64_t sv, int64_t sc, int64_t sy, int64_t sx,
                                                     const float* __restrict__ points, int N,
                                                     const float* __restrict__ proj, float* __restrict__ volume,
                                                     uint8_t* __restrict__ valid) {
    const int v = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int xi, yi;
    const bool ok = ndet_project(proj + v * 12, points[n], points[N + n], points[2 * N + n], w, h, xi, yi);
    valid[(int64_t)v * N + n] = ok ? 1 : 0;
    const float* src = feat + v * sv + yi * sy + xi * sx;
    float* dst = volume + (int64_t)v * C * N + n;
    int c = 0;
    for (; c + 4 <= C; c += 4) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        if (ok) {
            t0 = src[(c + 0) * sc];
            t1 = src[(c + 1) * sc];
            t2 = src[(c + 2) * sc];
            t3 = src[(c + 3) * sc];
        }
        dst[(int64_t)(c + 0) * N] = t0;
        dst[(int64_t)(c + 1) * N] = t1;
        dst[(int64_t)(c + 2) * N] = t2;
        dst[(int64_t)(c + 3) * N] = t3;
    }
    for (; c < C; ++c) dst[(int64_t)c * N] = ok ? src[c * sc] : 0.f;
}

extern "C" int ndet_backproject(const float* features, int n_views, int C, int h, int w, int64_t sv, int64_t sc,
                                int64_t sy, int64_t sx, const float* points, int N, const float* projection,
                                float* volume, uint8_t* valid, void* stream) {
    NDET_REQUIRE(features && points && projection && volume && valid, NDET_E_INVALID, "ndet_backproject: null pointer");
    NDET_REQUIRE(n_views > 0 && C > 0 && h > 0 && w > 0 && N > 0, NDET_E_INVALID, "ndet_backproject: sizes must be positive");
    NDET_REQUIRE(n_views <= 65535, NDET_E_UNSUPPORTED, "ndet_backproject: more than 65535 views");
    dim3 grid((N + 255) / 256, n_views);
    hipLaunchKernelGGL(k_backproject, grid, dim3(256), 0, (hipStream_t)stream, features, C, h, w, sv, sc, sy, sx, points,
                       N, projection, volume, valid);
    NDET_CHECK_LAUNCH("ndet_backproject");
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------
// K1  fused backproject + view mean / count (+ alpha gating)   A3 + A4 (+ A6 gating)
//
// One wavefront per voxel, lanes over channels: with channels-last features one pixel's C floats
// are one contiguous row (1 KiB at C=256 = 64 lanes x float4), so a voxel-view gather is a single
// fully coalesced wave load.  The projection of the voxel into the views is computed with lanes
// over VIEWS (64 views per round), the valid views become a ballot mask, and the wave then walks
// the set bits with scalar code: only views that see the voxel cost a load, and the per-view pixel
// offset comes out of the lane that computed it with v_readlane.  GATHER_BATCH independent row
// loads are kept in flight.  The sum runs in ascending view order.
// Nothing of size (n_views, C, N) is ever written: reads = feature rows actually hit,
// writes = (C + 2) * N * 4 bytes.
// ------------------------------------------------------------------------------------------
template <int NCHUNK, bool GATE, int LAYOUT>
__global__ __launch_bounds__(256, K1_MIN_WAVES) void k_backproject_aggregate(
    const float* __restrict__ feat, int n_views, int C, int h, int w, int64_t view_pitch, int row_pitch,
    const float* __restrict__ points, int N, const float* __restrict__ proj, const float* __restrict__ alpha,
    float* __restrict__ out, int64_t* __restrict__ count, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // LAYOUT_CN only: [16][C + 4]
    constexpr int VPW = VOX_PER_TILE / 4;  // voxels per wave, processed together
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tile = ndet_xcd_remap(blockIdx.x, n_tiles);
    const int n0 = tile * VOX_PER_TILE;
    const int c4 = C >> 2;
    const int ldp = C + 4;

    // the wave's voxels: slots wave, wave+4, wave+8, wave+12 of the tile (the 4 waves of the workgroup work on
    // 4 neighbouring voxels at a time); coordinates of all of them are fetched up front
    float px[VPW], py[VPW], pz[VPW];
    bool live[VPW];
#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        const int n = n0 + j * 4 + wave;
        live[j] = n < N;  // wave-uniform
        const int nn = live[j] ? n : 0;
        px[j] = points[nn];
        py[j] = points[N + nn];
        pz[j] = points[2 * N + nn];
    }
    float4 acc[VPW][NCHUNK];
    int cnt[VPW];
#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        cnt[j] = 0;
#pragma unroll
        for (int q = 0; q < NCHUNK; ++q) acc[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    for (int r0 = 0; r0 < n_views; r0 += 64) {
        // phase A: lanes over views, one camera matrix per lane, all voxels of the wave projected back to back
        const int v = r0 + lane;
        float P[12];
        {
            const float4* pm = reinterpret_cast<const float4*>(proj + (v < n_views ? v : 0) * 12);
            const float4 a = pm[0], b = pm[1], c = pm[2];
            P[0] = a.x; P[1] = a.y; P[2] = a.z; P[3] = a.w;
            P[4] = b.x; P[5] = b.y; P[6] = b.z; P[7] = b.w;
            P[8] = c.x; P[9] = c.y; P[10] = c.z; P[11] = c.w;
        }
        int off[VPW];
        unsigned long long mask[VPW];
#pragma unroll
        for (int j = 0; j < VPW; ++j) {
            int xi = 0, yi = 0;
            const bool ok = (v < n_views) && live[j] && ndet_project(P, px[j], py[j], pz[j], w, h, xi, yi);
            off[j] = yi * row_pitch + xi * C;  // floats inside one view (< 2^31, checked on the host)
            mask[j] = __ballot(ok);
            cnt[j] += __popcll(mask[j]);
        }
        // phase B: walk the set bits; GATHER_BATCH independent row loads in flight
        const float* vbase = feat + (int64_t)r0 * view_pitch;
#pragma unroll
        for (int j = 0; j < VPW; ++j) {
            unsigned long long m = mask[j];
            int b = 0;
            while (m) {
                float4 t[GATHER_BATCH][NCHUNK];
                bool has[GATHER_BATCH];
#pragma unroll
                for (int k = 0; k < GATHER_BATCH; ++k) {
                    has[k] = (m != 0ull);
                    if (has[k]) {
                        b = __builtin_ctzll(m);
                        m &= (m - 1ull);
                    }  // else: re-read the previous row (L1 hit), discarded below
                    const int o = __builtin_amdgcn_readlane(off[j], b);
                    const float4* p = reinterpret_cast<const float4*>(vbase + (int64_t)b * view_pitch + o);
#pragma unroll
                    for (int q = 0; q < NCHUNK; ++q) {
                        const int ci = lane + q * 64;
                        t[k][q] = (ci < c4) ? p[ci] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int k = 0; k < GATHER_BATCH; ++k) {
                    if (has[k]) {
#pragma unroll
                        for (int q = 0; q < NCHUNK; ++q) acc[j][q] = ndet_add4(acc[j][q], t[k][q]);
                    }
                }
            }
        }
    }

    // volume_sum / (valid + 1e-8); zero where no view sees the voxel (nerfdet.py:175-176);
    // optionally alpha * mean, again zeroed at count 0 (nerfdet.py:259-261).
#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        if (!live[j]) continue;
        const int slot = j * 4 + wave;
        const int n = n0 + slot;
        const float denom = (float)cnt[j] + 1e-8f;
        float a = 1.0f;
        if (GATE) a = alpha[n];
#pragma unroll
        for (int q = 0; q < NCHUNK; ++q) {
            float4 mean;
            mean.x = acc[j][q].x / denom;
            mean.y = acc[j][q].y / denom;
            mean.z = acc[j][q].z / denom;
            mean.w = acc[j][q].w / denom;
            if (GATE) {
                mean.x = a * mean.x;
                mean.y = a * mean.y;
                mean.z = a * mean.z;
                mean.w = a * mean.w;
            }
            if (cnt[j] == 0) mean = make_float4(0.f, 0.f, 0.f, 0.f);
            const int ci = lane + q * 64;
            if (ci < c4) {
                if (LAYOUT == NDET_LAYOUT_NC)
                    *reinterpret_cast<float4*>(out + (int64_t)n * C + ci * 4) = mean;
                else
                    *reinterpret_cast<float4*>(smem + slot * ldp + ci * 4) = mean;
            }
        }
        if (lane == 0) count[n] = (int64_t)cnt[j];
    }

    if (LAYOUT == NDET_LAYOUT_CN) {
        // (16 voxels x C) tile -> C segments of 16 consecutive voxels (64 B) in the (C, N) tensor
        __syncthreads();
        for (int idx = threadIdx.x; idx < C * VOX_PER_TILE; idx += 256) {
            const int c = idx >> 4, jv = idx & 15;
            const int n = n0 + jv;
            if (n < N) out[(int64_t)c * N + n] = smem[jv * ldp + c];
        }
    }
}

template <int NCHUNK>
static void launch_k1(bool gate, int layout, dim3 grid, size_t lds, hipStream_t st, const float* feat, int n_views, int C,
                      int h, int w, int64_t view_pitch, int row_pitch, const float* points, int N, const float* proj,
                      const float* alpha, float* out, int64_t* count, int n_tiles) {
#define K1_LAUNCH(G, L)                                                                                              \
    hipLaunchKernelGGL((k_backproject_aggregate<NCHUNK, G, L>), grid, dim3(256), lds, st, feat, n_views, C, h, w,    \
                       view_pitch, row_pitch, points, N, proj, alpha, out, count, n_tiles)
    if (gate) {
        if (layout == NDET_LAYOUT_NC) K1_LAUNCH(true, NDET_LAYOUT_NC);
        else K1_LAUNCH(true, NDET_LAYOUT_CN);
    } else {
        if (layout == NDET_LAYOUT_NC) K1_LAUNCH(false, NDET_LAYOUT_NC);
        else K1_LAUNCH(false, NDET_LAYOUT_CN);
    }
#undef K1_LAUNCH
}

extern "C" int ndet_backproject_aggregate(const float* features_nhwc, int n_views, int C, int h, int w,
                                          int64_t view_pitch, int64_t row_pitch, const float* points, int N,
                                          const float* projection, const float* alpha, float* out, int out_layout,
                                          int64_t* count, void* stream) {
    const char* fn = "ndet_backproject_aggregate";
    NDET_REQUIRE(features_nhwc && points && projection && out && count, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_views > 0 && C > 0 && h > 0 && w > 0 && N > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE(out_layout == NDET_LAYOUT_CN || out_layout == NDET_LAYOUT_NC, NDET_E_INVALID, "%s: bad layout %d", fn, out_layout);
    NDET_REQUIRE(C % 4 == 0 && C <= 1024, NDET_E_UNSUPPORTED, "%s: C=%d must be a multiple of 4 and <= 1024", fn, C);
    NDET_REQUIRE(row_pitch >= (int64_t)w * C && view_pitch >= (int64_t)h * row_pitch, NDET_E_INVALID, "%s: pitches smaller than the image", fn);
    NDET_REQUIRE(row_pitch % 4 == 0 && view_pitch % 4 == 0 && ((uintptr_t)features_nhwc & 15) == 0, NDET_E_UNSUPPORTED,
                 "%s: feature rows must be 16-byte aligned", fn);
    NDET_REQUIRE((int64_t)h * row_pitch < (int64_t)1 << 31, NDET_E_UNSUPPORTED, "%s: one view exceeds 2^31 floats", fn);
    if (out_layout == NDET_LAYOUT_NC) NDET_REQUIRE(((uintptr_t)out & 15) == 0, NDET_E_UNSUPPORTED, "%s: out must be 16-byte aligned", fn);
    const int n_tiles = (N + VOX_PER_TILE - 1) / VOX_PER_TILE;
    const size_t lds = out_layout == NDET_LAYOUT_CN ? (size_t)VOX_PER_TILE * (C + 4) * sizeof(float) : 0;
    const dim3 grid(n_tiles);
    hipStream_t st = (hipStream_t)stream;
    const bool gate = alpha != nullptr;
    if (C <= 256)
        launch_k1<1>(gate, out_layout, grid, lds, st, features_nhwc, n_views, C, h, w, view_pitch, (int)row_pitch, points, N, projection, alpha, out, count, n_tiles);
    else if (C <= 512)
        launch_k1<2>(gate, out_layout, grid, lds, st, features_nhwc, n_views, C, h, w, view_pitch, (int)row_pitch, points, N, projection, alpha, out, count, n_tiles);
    else
        launch_k1<4>(gate, out_layout, grid, lds, st, features_nhwc, n_views, C, h, w, view_pitch, (int)row_pitch, points, N, projection, alpha, out, count, n_tiles);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------
// K2  density conditioning features   A5
//
// One wavefront per voxel, lanes over the 3 + cm channels (35 of 64 lanes at cm = 32): lanes 0-2
// read the three RGB planes at the stride-1 pixel, lanes 3.. read the cm-float row of the mapped
// feature map at the stride-4 pixel, all in one load instruction per view.  A view that does not see
// the voxel contributes the Linear's bias (mapped channels) or 0 (RGB) -- the "0 bias issue" of
// nerfdet.py:233.  Two passes over the views (mean, then squared deviations) as the reference does;
// the second pass re-reads the same few rows from L1/L2.  Views outside the union of the two
// validity masks all contribute the same constant and are folded in as n * const.
// ------------------------------------------------------------------------------------------
// Both validity masks and pixel offsets of one voxel for one 64-view round (lanes over views).
struct DensityRound {
    unsigned long long mf, mr;  // views seeing the voxel in the stride-4 map / in the full-resolution image
    int off_f, off_r;           // per-lane (= per-view) offsets into the mapped map / one RGB plane
    unsigned boff_f, boff_r;    // the same as byte offsets from the tensor base, view included (buffer-load path)
};

__device__ __forceinline__ DensityRound density_project(int v, int n_views, const float* __restrict__ proj, const float* __restrict__ rgb_proj,
                                                        float px, float py, float pz, int w, int h, int W, int H, int mrow_pitch, int cm,
                                                        int rsy, int64_t mview_pitch = 0, int64_t rsv = 0) {
    int xf = 0, yf = 0, xr = 0, yr = 0;
    bool okf = false, okr = false;
    if (v < n_views) {
        okf = ndet_project(proj + v * 12, px, py, pz, w, h, xf, yf);
        okr = ndet_project(rgb_proj + v * 12, px, py, pz, W, H, xr, yr);
    }
    DensityRound r;
    r.off_f = yf * mrow_pitch + xf * cm;
    r.off_r = yr * rsy + xr;
    r.boff_f = (unsigned)((v * mview_pitch + r.off_f) * 4);
    r.boff_r = (unsigned)((v * rsv + r.off_r) * 4);
    r.mf = __ballot(okf);
    r.mr = __ballot(okr);
    return r;
}

// One pass over the union of valid views of a round: PASS 0 accumulates values, PASS 1 squared deviations from `mean`.
template <int PASS>
__device__ __forceinline__ float density_round_pass(const DensityRound& d, int r0, const float* __restrict__ mapped, int64_t mview_pitch,
                                                    const float* __restrict__ rgb, int64_t rsv, int64_t rsc, int lane, int cm, float fill,
                                                    float mean, float acc) {
    const int ch = lane;
    const bool is_rgb = ch < 3;
    const bool active = ch < 3 + cm;
    unsigned long long m = d.mf | d.mr;
    int b = 0;
    while (m) {
        float t[K2_BATCH];
        bool has[K2_BATCH], mine[K2_BATCH];
#pragma unroll
        for (int k = 0; k < K2_BATCH; ++k) {
            has[k] = (m != 0ull);
            if (has[k]) {
                b = __builtin_ctzll(m);
                m &= (m - 1ull);
            }
            const bool vf = (d.mf >> b) & 1ull, vr = (d.mr >> b) & 1ull;
            const int of = __builtin_amdgcn_readlane(d.off_f, b);
            const int orr = __builtin_amdgcn_readlane(d.off_r, b);
            mine[k] = active && (is_rgb ? vr : vf);
            const float* p = is_rgb ? (rgb + (int64_t)(r0 + b) * rsv + (int64_t)ch * rsc + orr)
                                    : (mapped + (int64_t)(r0 + b) * mview_pitch + of + (ch - 3));
            t[k] = mine[k] ? *p : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < K2_BATCH; ++k) {
            if (has[k]) {
                const float val = mine[k] ? t[k] : fill;
                if (PASS == 0) {
                    acc = acc + val;
                } else {
                    const float dd = val - mean;
                    acc = acc + dd * dd;
                }
            }
        }
    }
    return acc;
}

// The same pass with buffer loads: a view's offset (computed by the lane that projected it) becomes the instruction's scalar
// offset, the lane's channel its constant vector offset -- no per-view address arithmetic in the vector unit, which is what
// bounds this kernel (a 140-byte gather per voxel-view: ~30 instructions of pointer math against 2 loads).  Lanes of the other
// kind (and idle lanes) carry an out-of-range offset and read zeros.
#define K2_OOB 0x80000000u
template <int PASS>
__device__ __forceinline__ float density_round_pass_buf(const DensityRound& d, __amdgpu_buffer_rsrc_t mres, __amdgpu_buffer_rsrc_t rres,
                                                        unsigned fvoff, unsigned rvoff, bool is_rgb, float fill, float mean, float acc) {
    unsigned long long m = d.mf | d.mr;
    int b = 0;
    while (m) {
        float tf[K2_BATCH], tr[K2_BATCH];
        bool has[K2_BATCH], vf[K2_BATCH], vr[K2_BATCH];
#pragma unroll
        for (int k = 0; k < K2_BATCH; ++k) {
            has[k] = (m != 0ull);
            if (has[k]) {
                b = __builtin_ctzll(m);
                m &= (m - 1ull);
            }
            vf[k] = has[k] && ((d.mf >> b) & 1ull);
            vr[k] = has[k] && ((d.mr >> b) & 1ull);
            const unsigned of = (unsigned)__builtin_amdgcn_readlane((int)d.boff_f, b);
            const unsigned orr = (unsigned)__builtin_amdgcn_readlane((int)d.boff_r, b);
            tf[k] = vf[k] ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mres, fvoff, of, 0)) : 0.0f;
            tr[k] = vr[k] ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rres, rvoff, orr, 0)) : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < K2_BATCH; ++k) {
            if (has[k]) {
                const float val = is_rgb ? (vr[k] ? tr[k] : fill) : (vf[k] ? tf[k] : fill);
                if (PASS == 0) {
                    acc = acc + val;
                } else {
                    const float dd = val - mean;
                    acc = acc + dd * dd;
                }
            }
        }
    }
    return acc;
}

template <bool BUF>
__global__ __launch_bounds__(256) void k_density_features(const float* __restrict__ mapped, int n_views, int cm, int h, int w,
                                                          int64_t mview_pitch, int mrow_pitch, const float* __restrict__ bias,
                                                          const float* __restrict__ rgb, int H, int W, int64_t rsv, int64_t rsc,
                                                          int rsy, const float* __restrict__ points, int N,
                                                          const float* __restrict__ proj, const float* __restrict__ rgb_proj,
                                                          float* __restrict__ out, int n_tiles) {
    constexpr int VPW = VOX_PER_TILE / 4;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tile = ndet_xcd_remap(blockIdx.x, n_tiles);
    const int F = 2 * (3 + cm);
    const float fill = (lane >= 3 && lane < 3 + cm) ? bias[lane - 3] : 0.0f;
    const bool single_round = n_views <= 64;  // the common case: projections are computed once and reused by both passes
    const bool is_rgb = lane < 3;
    // buffer path: per-lane constant offsets (the lane's channel), out of range for lanes of the other kind
    const __amdgpu_buffer_rsrc_t mres = __builtin_amdgcn_make_buffer_rsrc((void*)mapped, 0, K2_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)rgb, 0, K2_OOB, 0x00020000);
    const unsigned fvoff = (lane >= 3 && lane < 3 + cm) ? (unsigned)((lane - 3) * 4) : K2_OOB;
    const unsigned rvoff = is_rgb ? (unsigned)(lane * rsc * 4) : K2_OOB;

    float px[VPW], py[VPW], pz[VPW];
    bool live[VPW];
#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        const int n = tile * VOX_PER_TILE + j * 4 + wave;
        live[j] = n < N;
        const int nn = live[j] ? n : 0;
        px[j] = points[nn]; py[j] = points[N + nn]; pz[j] = points[2 * N + nn];
    }
    // the wave's 4 voxels are projected back to back (lanes over views) before any gather is issued
    DensityRound first[VPW];
#pragma unroll
    for (int j = 0; j < VPW; ++j) first[j] = density_project(lane, n_views, proj, rgb_proj, px[j], py[j], pz[j], w, h, W, H, mrow_pitch, cm, rsy, mview_pitch, rsv);

#pragma unroll
    for (int j = 0; j < VPW; ++j) {
        if (!live[j]) continue;
        const int n = tile * VOX_PER_TILE + j * 4 + wave;
        float sum = 0.0f;
        int cnt = 0, n_union = 0;
        for (int r0 = 0; r0 < n_views; r0 += 64) {
            const DensityRound d = (r0 == 0) ? first[j]
                                             : density_project(r0 + lane, n_views, proj, rgb_proj, px[j], py[j], pz[j], w, h, W, H, mrow_pitch, cm, rsy, mview_pitch, rsv);
            cnt += __popcll(d.mf);
            n_union += __popcll(d.mf | d.mr);
            sum = BUF ? density_round_pass_buf<0>(d, mres, rres, fvoff, rvoff, is_rgb, fill, 0.0f, sum)
                      : density_round_pass<0>(d, r0, mapped, mview_pitch, rgb, rsv, rsc, lane, cm, fill, 0.0f, sum);
        }
        const float rest = (float)(n_views - n_union);
        sum = sum + rest * fill;
        const float denom = (float)cnt + 1e-8f;
        const float mean = sum / denom;  // NOT zeroed at cnt == 0 (nerfdet.py:241)
        float ss = 0.0f;
        for (int r0 = 0; r0 < n_views; r0 += 64) {
            const DensityRound d = (r0 == 0 || single_round) ? first[j]
                                                             : density_project(r0 + lane, n_views, proj, rgb_proj, px[j], py[j], pz[j], w, h, W, H, mrow_pitch, cm, rsy, mview_pitch, rsv);
            ss = BUF ? density_round_pass_buf<1>(d, mres, rres, fvoff, rvoff, is_rgb, fill, mean, ss)
                     : density_round_pass<1>(d, r0, mapped, mview_pitch, rgb, rsv, rsc, lane, cm, fill, mean, ss);
        }
        const float dd = fill - mean;
        ss = ss + rest * (dd * dd);
        float var = ss / denom;
        if (cnt == 0) var = 1e6f;  // nerfdet.py:249
        const float cov = expf(-var);
        if (lane < 3 + cm) *reinterpret_cast<float2*>(out + (int64_t)n * F + 2 * lane) = make_float2(mean, cov);
    }
}

extern "C" int ndet_density_features(const float* mapped_nhwc, int n_views, int cm, int h, int w, int64_t mview_pitch,
                                     int64_t mrow_pitch, const float* bias, const float* rgb, int H, int W, int64_t rsv,
                                     int64_t rsc, int64_t rsy, const float* points, int N, const float* projection,
                                     const float* rgb_projection, float* global_feat, void* stream) {
    const char* fn = "ndet_density_features";
    NDET_REQUIRE(mapped_nhwc && bias && rgb && points && projection && rgb_projection && global_feat, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_views > 0 && cm > 0 && h > 0 && w > 0 && H > 0 && W > 0 && N > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE(cm <= 61, NDET_E_UNSUPPORTED, "%s: cm=%d mapped channels do not fit one wavefront (max 61)", fn, cm);
    NDET_REQUIRE((int64_t)h * mrow_pitch < (int64_t)1 << 31 && (int64_t)H * rsy < (int64_t)1 << 31, NDET_E_UNSUPPORTED,
                 "%s: one view exceeds 2^31 floats", fn);
    NDET_REQUIRE(((uintptr_t)global_feat & 7) == 0, NDET_E_UNSUPPORTED, "%s: global_feat must be 8-byte aligned", fn);
    const int n_tiles = (N + VOX_PER_TILE - 1) / VOX_PER_TILE;
    // both tensors within 2 GB: gathers as buffer loads with the view offset in the scalar register
    const bool buf = (int64_t)n_views * mview_pitch * 4 < ((int64_t)1 << 31) && (int64_t)n_views * rsv * 4 < ((int64_t)1 << 31) &&
                     mview_pitch >= 0 && rsv >= 0 && rsc >= 0;
    if (buf)
        hipLaunchKernelGGL(k_density_features<true>, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, mapped_nhwc, n_views, cm, h, w,
                           mview_pitch, (int)mrow_pitch, bias, rgb, H, W, rsv, rsc, (int)rsy, points, N, projection, rgb_projection,
                           global_feat, n_tiles);
    else
        hipLaunchKernelGGL(k_density_features<false>, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, mapped_nhwc, n_views, cm, h, w,
                           mview_pitch, (int)mrow_pitch, bias, rgb, H, W, rsv, rsc, (int)rsy, points, N, projection, rgb_projection,
                           global_feat, n_tiles);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------
// A6 pieces: sigma -> alpha, gating (unfused form), MLP input rows
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sigma_to_alpha(const float* __restrict__ raw, float* __restrict__ alpha, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float s = fmaxf(raw[n], 0.0f);  // F.relu, nerf_mlp.py:227
    alpha[n] = 1.0f - expf(-s);            // nerfdet.py:257
}

extern "C" int ndet_sigma_to_alpha(const float* raw_sigma, float* alpha, int N, void* stream) {
    NDET_REQUIRE(raw_sigma && alpha, NDET_E_INVALID, "ndet_sigma_to_alpha: null pointer");
    NDET_REQUIRE(N > 0, NDET_E_INVALID, "ndet_sigma_to_alpha: N must be positive");
    hipLaunchKernelGGL(k_sigma_to_alpha, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw_sigma, alpha, N);
    NDET_CHECK_LAUNCH("ndet_sigma_to_alpha");
    return NDET_OK;
}

__global__ __launch_bounds__(256) void k_alpha_gate(const float* __restrict__ mean, const float* __restrict__ density,
                                                    const int64_t* __restrict__ count, float* __restrict__ out, int C, int N,
                                                    int layout) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)C * N) return;
    const int n = layout == NDET_LAYOUT_NC ? (int)(i / C) : (int)(i % N);
    const float a = 1.0f - expf(-density[n]);
    out[i] = count[n] == 0 ? 0.0f : a * mean[i];
}

extern "C" int ndet_alpha_gate(const float* mean, const float* density, const int64_t* count, float* out, int C, int N,
                               int layout, void* stream) {
    NDET_REQUIRE(mean && density && count && out, NDET_E_INVALID, "ndet_alpha_gate: null pointer");
    NDET_REQUIRE(C > 0 && N > 0, NDET_E_INVALID, "ndet_alpha_gate: sizes must be positive");
    NDET_REQUIRE(layout == NDET_LAYOUT_CN || layout == NDET_LAYOUT_NC, NDET_E_INVALID, "ndet_alpha_gate: bad layout");
    const int64_t total = (int64_t)C * N;
    hipLaunchKernelGGL(k_alpha_gate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mean, density,
                       count, out, C, N, layout);
    NDET_CHECK_LAUNCH("ndet_alpha_gate");
    return NDET_OK;
}

__global__ __launch_bounds__(256) void k_posenc_concat(const float* __restrict__ points, const float* __restrict__ glob, int N,
                                                       int F, int K, float* __restrict__ out) {
    // K = row stride >= 63 + F; columns beyond 63 + F are zero (padding to the MFMA kernel's 32-channel K step)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * K) return;
    const int n = (int)(i / K), col = (int)(i % K);
    float r;
    if (col >= 63 + F) {
        r = 0.0f;
    } else if (col >= 63) {
        r = glob[(int64_t)n * F + (col - 63)];
    } else if (col < 3) {
        r = points[col * N + n];
    } else {
        // latent = sin(cat([xb, xb + pi/2])), xb degree-major / xyz-minor (nerf_mlp.py:190-194)
        const int t = col - 3;
        const int half = t >= 30 ? 1 : 0;
        const int k = (t - 30 * half) / 3, d = (t - 30 * half) % 3;
        float xb = points[d * N + n] * (float)(1 << k);
        if (half) xb = xb + 1.57079632679489661923f;
        r = sinf(xb);
    }
    out[i] = r;
}

extern "C" int ndet_posenc_concat(const float* points, const float* global_feat, int N, int F, int out_stride, float* out, void* stream) {
    NDET_REQUIRE(points && out && (global_feat || F == 0), NDET_E_INVALID, "ndet_posenc_concat: null pointer");
    NDET_REQUIRE(N > 0 && F >= 0 && out_stride >= 63 + F, NDET_E_INVALID, "ndet_posenc_concat: bad sizes");
    const int64_t total = (int64_t)N * out_stride;
    hipLaunchKernelGGL(k_posenc_concat, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, points,
                       global_feat, N, F, out_stride, out);
    NDET_CHECK_LAUNCH("ndet_posenc_concat");
    return NDET_OK;
}


// ------------------------------------------------------------------------------------------------
// A6 tail: sigma = w . [h | x] + b over the re-joined trunk output (nerf_mlp.py:86,143: the skip concat after the last
// hidden layer feeds the 389 -> 1 sigma layer), then alpha = 1 - exp(-relu(sigma)) (nerf_mlp.py:227, nerfdet.py:257).
// One wavefront per row; the concat is never materialised.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sigma_head(const float* __restrict__ h, int Ch, const float* __restrict__ x, int Cx, int x_stride,
                                                    const float* __restrict__ w, const float* __restrict__ bias, int N,
                                                    float* __restrict__ raw_sigma, float* __restrict__ alpha) {
    const int lane = threadIdx.x & 63;
    const int n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (n >= N) return;
    float acc = 0.0f;
    for (int c = lane; c < Ch; c += 64) acc = fmaf(h[(int64_t)n * Ch + c], w[c], acc);
    for (int c = lane; c < Cx; c += 64) acc = fmaf(x[(int64_t)n * x_stride + c], w[Ch + c], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        const float sg = acc + bias[0];
        if (raw_sigma) raw_sigma[n] = sg;
        alpha[n] = 1.0f - expf(-fmaxf(sg, 0.0f));
    }
}

extern "C" int ndet_sigma_head(const float* h, int Ch, const float* x, int Cx, int x_stride, const float* w, const float* bias, int N,
                               float* raw_sigma, float* alpha, void* stream) {
    const char* fn = "ndet_sigma_head";
    NDET_REQUIRE(h && x && w && bias && alpha, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(N > 0 && Ch > 0 && Cx >= 0 && x_stride >= Cx, NDET_E_INVALID, "%s: bad sizes", fn);
    const int64_t blocks = ((int64_t)N + 3) / 4;
    hipLaunchKernelGGL(k_sigma_head, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, h, Ch, x, Cx, x_stride, w, bias, N, raw_sigma,
                       alpha);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}
