// BatchNorm on batch statistics over channels-last rows (N voxels x C channels), forward and backward, with the ReLU and the residual add of
// mmdet3d/models/necks/imvoxelnet.py:22-67 (BasicBlock3dV2: relu(norm1(conv1 x)), relu(norm2(conv2 .) + identity)) and :233-260 (conv - BN - ReLU of the
// up / out blocks) folded into the same passes.  The training step of the 3D neck has 15 such layers; through ATen each is a statistics kernel at
// ~0.6 TB/s, a normalise pass, a separate ReLU, (a separate add,) and the same again backwards -- 1.8 ms + 0.4 ms of ReLU passes per step (profiles/r04_c_train_profile_cfg3_f16x2.txt).
//
//   forward   k_bn_stats      per-workgroup partial sums of (x - s) and (x - s)^2 per channel, s = the tensor's first row (a shift that keeps the
//                             one-pass variance well conditioned); fixed-order partials: deterministic
//             k_bn_finish     partials -> mean, 1 / sqrt(var + eps) (biased variance, as F.batch_norm), running statistics (unbiased variance)
//             k_bn_apply      y = relu?((x - mean) invstd gamma + beta (+ residual)); leaves max |y| in an amax slot for the next convolution
//   backward  k_bn_bwd_stats  partial sums of g and g xhat, g = dy [y > 0]
//             k_bn_bwd_finish partials -> dgamma, dbeta and the two means of the input gradient
//             k_bn_bwd_apply  dx = gamma invstd (g - mean(g) - xhat mean(g xhat)); d_residual = g
#include "conv_common.hpp"

#define BN_THREADS 1024      // row kernels: 16 waves per workgroup, C / 4 lanes per row
#define BN_MAX_WG 256        // partial sums per channel (the finish kernels add them in a fixed order: 8 lanes x <= 32 each)
#define BN_FIN_THREADS 256   // finish kernels: 32 channels x 8 partial lanes

struct BnGeom { int64_t N; int C, CQ, RL; int64_t rows_per_wg; };      // CQ = C / 4 channel quads, RL = 1024 / CQ rows side by side in a workgroup

__device__ __forceinline__ float4 bn_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// two per-channel sums over a workgroup's rows -> partial[(g * 2 + k) * C + c]
__device__ __forceinline__ void bn_reduce_rows(float4 a, float4 b, const BnGeom& G, float* __restrict__ partial) {
    __shared__ float4 red[2][BN_THREADS];
    const int t = threadIdx.x, q = t % G.CQ, rl = t / G.CQ;
    red[0][t] = a; red[1][t] = b;
    __syncthreads();
    if (rl == 0 && t < G.CQ) {
        for (int r = 1; r < G.RL; ++r) {                       // fixed order
            const float4 u = red[0][r * G.CQ + q], v = red[1][r * G.CQ + q];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 0) * G.C + 4 * q) = a;
        *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + 1) * G.C + 4 * q) = b;
    }
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_stats(const float* __restrict__ x, BnGeom G, float* __restrict__ partial) {
    const int t = threadIdx.x, q = t % G.CQ, rl = t / G.CQ;
    const int64_t lo = (int64_t)blockIdx.x * G.rows_per_wg, hi = lo + G.rows_per_wg < G.N ? lo + G.rows_per_wg : G.N;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (rl < G.RL) {
        const float4 sh = bn_ld4(x + 4 * q);
        for (int64_t r = lo + rl; r < hi; r += G.RL) {
            const float4 v = bn_ld4(x + r * G.C + 4 * q);
            const float dx = v.x - sh.x, dy = v.y - sh.y, dz = v.z - sh.z, dw = v.w - sh.w;
            s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
            s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
        }
    }
    bn_reduce_rows(s1, s2, G, partial);
}

// partial[(g * 2 + k) * C + c], g < n_wg -> the two sums of channel c.  32 channels x 8 lanes per workgroup: lane p adds the partials p, p + 8, ...
// (eight loads in flight, index order), the eight lane sums are added in lane order: the same association whatever the launch's timing.
__device__ __forceinline__ void bn_sum_partials(const float* __restrict__ partial, int n_wg, int C, int c, int p, float& o1, float& o2) {
    __shared__ float lane_sum[2][8][32];
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        for (int g0 = p; g0 < n_wg; g0 += 64) {
            float a[8], b[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int g = g0 + 8 * k;
                a[k] = g < n_wg ? partial[((int64_t)g * 2 + 0) * C + c] : 0.f;
                b[k] = g < n_wg ? partial[((int64_t)g * 2 + 1) * C + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { s1 += a[k]; s2 += b[k]; }
        }
    }
    lane_sum[0][p][threadIdx.x & 31] = s1; lane_sum[1][p][threadIdx.x & 31] = s2;
    __syncthreads();
    o1 = 0.f; o2 = 0.f;
    for (int k = 0; k < 8; ++k) { o1 += lane_sum[0][k][threadIdx.x & 31]; o2 += lane_sum[1][k][threadIdx.x & 31]; }
}

__global__ __launch_bounds__(BN_FIN_THREADS) void k_bn_finish(const float* __restrict__ x, const float* __restrict__ partial, int n_wg, int64_t N, int C, float eps,
                                                              float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              float* __restrict__ mean, float* __restrict__ invstd) {
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), p = threadIdx.x >> 5;
    float s1, s2;
    bn_sum_partials(partial, n_wg, C, c, p, s1, s2);
    if (c >= C || p != 0) return;
    const float inv_n = 1.0f / (float)N;
    const float d = s1 * inv_n;                    // mean - shift
    float var = s2 * inv_n - d * d;
    var = var > 0.f ? var : 0.f;
    const float m = x[c] + d;
    mean[c] = m;
    invstd[c] = 1.0f / sqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * m;
    if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (N > 1 ? var * ((float)N / (float)(N - 1)) : var);
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_apply(const float* __restrict__ x, BnGeom G, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ res, int relu,
                                                         float* __restrict__ y, float* __restrict__ amax) {
    const int t = threadIdx.x, q = t % G.CQ, rl = t / G.CQ;
    const int64_t lo = (int64_t)blockIdx.x * G.rows_per_wg, hi = lo + G.rows_per_wg < G.N ? lo + G.rows_per_wg : G.N;
    float mx = 0.f;
    if (rl < G.RL) {
        const float4 m = bn_ld4(mean + 4 * q), is = bn_ld4(invstd + 4 * q), ga = bn_ld4(gamma + 4 * q), be = bn_ld4(beta + 4 * q);
        for (int64_t r = lo + rl; r < hi; r += G.RL) {
            const int64_t o = r * G.C + 4 * q;
            const float4 v = bn_ld4(x + o);
            float4 w = make_float4((v.x - m.x) * is.x * ga.x + be.x, (v.y - m.y) * is.y * ga.y + be.y, (v.z - m.z) * is.z * ga.z + be.z, (v.w - m.w) * is.w * ga.w + be.w);
            if (res) { const float4 u = bn_ld4(res + o); w.x += u.x; w.y += u.y; w.z += u.z; w.w += u.w; }
            if (relu) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
            *reinterpret_cast<float4*>(y + o) = w;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
        }
    }
    if (amax) conv_amax_commit(amax, mx);
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_bwd_stats(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y, BnGeom G,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd, int relu, float* __restrict__ partial) {
    const int t = threadIdx.x, q = t % G.CQ, rl = t / G.CQ;
    const int64_t lo = (int64_t)blockIdx.x * G.rows_per_wg, hi = lo + G.rows_per_wg < G.N ? lo + G.rows_per_wg : G.N;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (rl < G.RL) {
        const float4 m = bn_ld4(mean + 4 * q), is = bn_ld4(invstd + 4 * q);
        for (int64_t r = lo + rl; r < hi; r += G.RL) {
            const int64_t o = r * G.C + 4 * q;
            float4 g = bn_ld4(dy + o);
            if (relu) { const float4 u = bn_ld4(y + o); g.x = u.x > 0.f ? g.x : 0.f; g.y = u.y > 0.f ? g.y : 0.f; g.z = u.z > 0.f ? g.z : 0.f; g.w = u.w > 0.f ? g.w : 0.f; }
            const float4 v = bn_ld4(x + o);
            s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
            s2.x += g.x * ((v.x - m.x) * is.x); s2.y += g.y * ((v.y - m.y) * is.y); s2.z += g.z * ((v.z - m.z) * is.z); s2.w += g.w * ((v.w - m.w) * is.w);
        }
    }
    bn_reduce_rows(s1, s2, G, partial);
}

__global__ __launch_bounds__(BN_FIN_THREADS) void k_bn_bwd_finish(const float* __restrict__ partial, int n_wg, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), p = threadIdx.x >> 5;
    float s1, s2;
    bn_sum_partials(partial, n_wg, C, c, p, s1, s2);
    if (c >= C || p != 0) return;
    dbeta[c] = s1;
    dgamma[c] = s2;
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y, BnGeom G,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ dgamma, const float* __restrict__ dbeta, int relu, float* __restrict__ dx,
                                                             float* __restrict__ dres, float* __restrict__ amax) {
    const int t = threadIdx.x, q = t % G.CQ, rl = t / G.CQ;
    const int64_t lo = (int64_t)blockIdx.x * G.rows_per_wg, hi = lo + G.rows_per_wg < G.N ? lo + G.rows_per_wg : G.N;
    float mx = 0.f;
    if (rl < G.RL) {
        const float inv_n = 1.0f / (float)G.N;
        const float4 m = bn_ld4(mean + 4 * q), is = bn_ld4(invstd + 4 * q), ga = bn_ld4(gamma + 4 * q), dg = bn_ld4(dgamma + 4 * q), db = bn_ld4(dbeta + 4 * q);
        const float4 a = make_float4(ga.x * is.x, ga.y * is.y, ga.z * is.z, ga.w * is.w);
        const float4 b = make_float4(db.x * inv_n, db.y * inv_n, db.z * inv_n, db.w * inv_n), cc = make_float4(dg.x * inv_n, dg.y * inv_n, dg.z * inv_n, dg.w * inv_n);
        for (int64_t r = lo + rl; r < hi; r += G.RL) {
            const int64_t o = r * G.C + 4 * q;
            float4 g = bn_ld4(dy + o);
            if (relu) { const float4 u = bn_ld4(y + o); g.x = u.x > 0.f ? g.x : 0.f; g.y = u.y > 0.f ? g.y : 0.f; g.z = u.z > 0.f ? g.z : 0.f; g.w = u.w > 0.f ? g.w : 0.f; }
            if (dres) *reinterpret_cast<float4*>(dres + o) = g;
            const float4 v = bn_ld4(x + o);
            const float4 w = make_float4(a.x * (g.x - b.x - (v.x - m.x) * is.x * cc.x), a.y * (g.y - b.y - (v.y - m.y) * is.y * cc.y),
                                         a.z * (g.z - b.z - (v.z - m.z) * is.z * cc.z), a.w * (g.w - b.w - (v.w - m.w) * is.w * cc.w));
            *reinterpret_cast<float4*>(dx + o) = w;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
        }
    }
    if (amax) conv_amax_commit(amax, mx);
}

static int bn_geom(const char* fn, int64_t N, int C, BnGeom& G, int& n_wg) {
    NDET_REQUIRE(N > 0 && C >= 4 && C % 4 == 0 && C / 4 <= BN_THREADS && BN_THREADS % (C / 4) == 0, NDET_E_UNSUPPORTED,
                 "%s: C=%d must be 4 x a divisor of %d (C / 4 lanes per row)", fn, C, BN_THREADS);
    G.N = N; G.C = C; G.CQ = C / 4; G.RL = BN_THREADS / G.CQ;
    int64_t rows = 4 * (int64_t)G.RL;                       // >= 4 rows per thread
    while ((N + rows - 1) / rows > BN_MAX_WG) rows *= 2;
    G.rows_per_wg = rows;
    n_wg = (int)((N + rows - 1) / rows);
    return NDET_OK;
}

extern "C" int64_t ndet_bn_workspace_floats(int64_t N, int C) {
    if (N <= 0 || C < 4 || C % 4 || C / 4 > BN_THREADS || BN_THREADS % (C / 4)) return -1;
    BnGeom G; int n_wg = 0;
    bn_geom("ndet_bn_workspace_floats", N, C, G, n_wg);
    return (int64_t)n_wg * 2 * C;
}

extern "C" int ndet_bn_train_forward(const float* x, int64_t N, int C, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     float momentum, float eps, const float* residual, int relu, float* y, float* save_mean, float* save_invstd,
                                     float* y_amax, float* workspace, void* stream) {
    const char* fn = "ndet_bn_train_forward";
    NDET_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && workspace, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)save_mean | (uintptr_t)save_invstd | (uintptr_t)workspace) & 15) == 0,
                 NDET_E_UNSUPPORTED, "%s: pointers must be 16-byte aligned", fn);
    NDET_REQUIRE(eps > 0.0f && momentum >= 0.0f && momentum <= 1.0f, NDET_E_INVALID, "%s: eps > 0, momentum in [0, 1]", fn);
    BnGeom G; int n_wg = 0;
    const int rc = bn_geom(fn, N, C, G, n_wg);
    if (rc != NDET_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bn_stats, dim3(n_wg), dim3(BN_THREADS), 0, st, x, G, workspace);
    hipLaunchKernelGGL(k_bn_finish, dim3((C + 31) / 32), dim3(BN_FIN_THREADS), 0, st, x, (const float*)workspace, n_wg, N, C, eps, momentum, running_mean,
                       running_var, save_mean, save_invstd);
    hipLaunchKernelGGL(k_bn_apply, dim3(n_wg), dim3(BN_THREADS), 0, st, x, G, (const float*)save_mean, (const float*)save_invstd, gamma, beta, residual, relu ? 1 : 0, y, y_amax);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

extern "C" int ndet_bn_train_backward(const float* dy, const float* x, const float* y, int64_t N, int C, const float* gamma, const float* save_mean,
                                      const float* save_invstd, int relu, float* dx, float* d_residual, float* dgamma, float* dbeta, float* dx_amax,
                                      float* workspace, void* stream) {
    const char* fn = "ndet_bn_train_backward";
    NDET_REQUIRE(dy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta && workspace && (y || !relu), NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)d_residual | (uintptr_t)gamma | (uintptr_t)save_mean | (uintptr_t)save_invstd |
                   (uintptr_t)dgamma | (uintptr_t)dbeta | (uintptr_t)workspace) & 15) == 0, NDET_E_UNSUPPORTED, "%s: pointers must be 16-byte aligned", fn);
    BnGeom G; int n_wg = 0;
    const int rc = bn_geom(fn, N, C, G, n_wg);
    if (rc != NDET_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bn_bwd_stats, dim3(n_wg), dim3(BN_THREADS), 0, st, dy, x, y, G, save_mean, save_invstd, relu ? 1 : 0, workspace);
    hipLaunchKernelGGL(k_bn_bwd_finish, dim3((C + 31) / 32), dim3(BN_FIN_THREADS), 0, st, (const float*)workspace, n_wg, C, dgamma, dbeta);
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(n_wg), dim3(BN_THREADS), 0, st, dy, x, y, G, save_mean, save_invstd, gamma, (const float*)dgamma, (const float*)dbeta, relu ? 1 : 0, dx,
                       d_residual, dx_amax);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}
