// A6 in one launch: per-voxel density MLP of the volumetric path, fused end to end for gfx950.
//
//   rows   = [x, sin(2^k x), sin(2^k x + pi/2) (k = 0..9) | conditioning row]          SinusoidalEncoder.forward, nerf_mlp.py:181-197; concat :140
//   h      = 4 x relu(Linear)  (133 -> 256 -> 256 -> 256 -> 256)                       MLP.forward, nerf_mlp.py:80-90 (skip_layer 3: the input re-joins
//   sigma  = w . [h | rows] + b                                                          after the last hidden layer and feeds only the sigma layer, :138-144)
//   alpha  = 1 - exp(-relu(sigma))                                                       nerf_mlp.py:224-227, detectors/nerfdet.py:254-257
//
// Before this kernel the chain was nine launches (posenc/concat, four 1x1 "convolutions" with the 256-wide rows round-tripping HBM between them,
// sigma head, ...: 0.19 ms of the 0.39 ms hot path at cfg2, profiles/r03_d_layer_times_cfg2_f16x2.txt rows 52-56).  Here a workgroup owns 64 rows
// from the encoder to alpha: the activations live in LDS as fp16 (hi, lo) planes, the weights stream from L2 straight into MFMA fragments
// (each wave owns 64 output channels: no weight element is needed by two waves, so staging them in LDS would buy nothing), the
// accumulators never leave registers between a layer's last MFMA and the next layer's operand planes.
//
// Arithmetic: the fp16-pair scheme of conv_split_kernels.hip (three v_mfma_f32_32x32x16_f16 products hi*hi + hi*lo + lo*hi, fp32 accumulate) with
// the activation scale taken PER ROW: every row is pre-scaled by its own power of two (largest magnitude of the row in [2^14, 2^15)), undone
// exactly in the epilogue.  The per-tensor scale of the convolution kernels is not safe here -- the reference's conditioning rows of voxels no
// view sees carry n_views * bias / 1e-8 ~ 1e9 beside O(1) rows (nerfdet.py:236-243) -- a per-row scale is: a row's error is 2^-22 of ITS OWN
// magnitude, whatever its neighbours hold.  Weights: planes of w * 2^e from ndet_split_weights_f16x2, (1, K/32, 2, 256, 32).
//
// The MFMA runs "transposed": A = weights (M = output channel), B = activations (N = row), so a lane of the 32x32 result holds ONE row's
// values for 4 x 4 consecutive channels -- the next layer's operand planes are written as 8-byte pieces (not 2-byte scatters), the row maximum
// is an in-lane reduction plus one cross-half shuffle, and the sigma layer is a register dot product.
#include "conv_common.hpp"

typedef _Float16 pm_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pm_f16x2 __attribute__((ext_vector_type(2)));
typedef float pm_f32x2 __attribute__((ext_vector_type(2)));
typedef float pm_f32x16 __attribute__((ext_vector_type(16)));

#define PM_BM 64       // rows per workgroup
#define PM_HID 256     // hidden width (the shipped architecture; other widths take the layer-by-layer path)
#define PM_LDW 256     // fp16 elements per LDS row: >= the padded input width; a multiple of 64 so that the chunk swizzle stays inside the row

struct PointMlpParams {
    const float* points;    // (3, N)
    const float* glob;      // (N, F) conditioning rows, or null when F == 0
    int N, F, K0;           // K0: input width padded to the K step (multiple of 32, 63 + F <= K0 <= PM_LDW)
    const uint16_t* w[4];   // fp16-pair planes of the four hidden layers
    float winv[4];          // 1 / (their power-of-two scales)
    const float* b[4];      // biases (256 each)
    const float* wsig;      // (256 + 63 + F): sigma layer over [h | rows]
    const float* bsig;      // (1)
    float* raw;             // (N) sigma before the ReLU, or null
    float* alpha;           // (N)
    float* h_out;           // (N, 256) trunk output, or null
};

__device__ __forceinline__ uint32_t pm_pack(float x, float y) {
    const pm_f16x2 v = __builtin_convertvector((pm_f32x2){x, y}, pm_f16x2);
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float pm_lo(uint32_t u) { return (float)__builtin_bit_cast(pm_f16x2, u)[0]; }
__device__ __forceinline__ float pm_hi(uint32_t u) { return (float)__builtin_bit_cast(pm_f16x2, u)[1]; }
// two pre-scaled values -> (hi pair, lo pair); the differences are exact in fp32
__device__ __forceinline__ void pm_split2(float a, float b, uint32_t& h, uint32_t& l) {
    h = pm_pack(a, b);
    l = pm_pack(a - pm_lo(h), b - pm_hi(h));
}

// element `col` of the MLP input row of a point (nerf_mlp.py:190-197: latent = [x | sin(x 2^k) (k-major, xyz-minor) | sin(x 2^k + pi/2)])
__device__ __forceinline__ float pm_input(int col, float px, float py, float pz, const float* __restrict__ grow, int F) {
    if (col < 3) return col == 0 ? px : (col == 1 ? py : pz);
    if (col < 63) {
        const int t = col - 3;
        const int half = t >= 30 ? 1 : 0;
        const int u = t - 30 * half;
        const int k = u / 3, d = u - 3 * k;
        float xb = (d == 0 ? px : (d == 1 ? py : pz)) * (float)(1 << k);
        if (half) xb = xb + 1.57079632679489661923f;
        return sinf(xb);
    }
    return col < 63 + F ? grow[col - 63] : 0.0f;
}

__global__ __launch_bounds__(256, 2) void k_point_mlp(const PointMlpParams p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t pm_lds[];
    constexpr int PL = PM_BM * PM_LDW;                     // one plane, fp16 elements
    uint16_t* act = pm_lds;                                // [2][PM_BM][PM_LDW], 16-byte chunk index XORed with row & 15 (16 rows of a fragment read sweep all 64 banks)
    float* fl = reinterpret_cast<float*>(pm_lds + 2 * PL);
    float* pmax = fl;                                      // [4][PM_BM]  per-wave partial row maxima of the layer being finished
    float* rinv = fl + 4 * PM_BM;                          // [PM_BM]     1 / (scale of the row's current operand planes)
    float* xdot = rinv + PM_BM;                            // [PM_BM]     w_sigma[256:] . rows  (the re-joined input's share of sigma)
    float* sigp = xdot + PM_BM;                            // [4][PM_BM]  per-wave partial sums of w_sigma[:256] . h

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * PM_BM;

    // ---------------- phase 0: encoder + concat + per-row scale + split into the operand planes ----------------
    // Four threads per row.  The conditioning values of a thread's quarter of the row are fetched as ONE batch of independent loads (statically
    // indexed registers): fetched one by one inside a data-dependent loop they were 20 us of dependent L2 round trips per workgroup.
    {
        constexpr int CWMAX = PM_LDW / 4;
        const int r = tid >> 2, q = tid & 3;
        const int n = row0 + r;
        const bool live = n < p.N;
        const int nn = live ? n : 0;
        const float px = live ? p.points[nn] : 0.0f, py = live ? p.points[p.N + nn] : 0.0f, pz = live ? p.points[2 * p.N + nn] : 0.0f;
        const float* grow = p.F ? p.glob + (int64_t)nn * p.F : p.points;
        const int cw = p.K0 >> 2, c0 = q * cw, cend = 63 + p.F;
        float gv[CWMAX];
#pragma unroll
        for (int j = 0; j < CWMAX; ++j) {
            const int c = c0 + j;
            gv[j] = (live && j < cw && c >= 63 && c < cend) ? grow[c - 63] : 0.0f;
        }
        // the row's maximum: coordinates, the encoder's terms (bounded by 1: the bound serves, the scale is a power of two anyway), conditioning
        float mx = fmaxf(fmaxf(fabsf(px), fabsf(py)), fmaxf(fabsf(pz), 1.0f));
#pragma unroll
        for (int j = 0; j < CWMAX; ++j) mx = fmaxf(mx, fabsf(gv[j]));
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        const float s = conv_xscale_of(mx);
        uint16_t* arow = act + r * PM_LDW;
        auto put = [&](int c, float v) {                   // one element -> its (hi, lo) halves at (row r, column c)
            const _Float16 h = (_Float16)(v * s);
            const _Float16 l = (_Float16)(v * s - (float)h);
            uint16_t* dst = arow + ((((c >> 3) ^ (r & 15))) << 3) + (c & 7);
            dst[0] = __builtin_bit_cast(uint16_t, h);
            dst[PL] = __builtin_bit_cast(uint16_t, l);
        };
        float dot = 0.0f;
        for (int c = q; c < 63; c += 4) {                  // the encoder's 63 columns, dealt over the row's four threads
            const float v = live ? pm_input(c, px, py, pz, grow, 0) : 0.0f;
            dot = fmaf(v, p.wsig[PM_HID + c], dot);
            put(c, v);
        }
#pragma unroll
        for (int j = 0; j < CWMAX; ++j) {                  // conditioning columns (and the zero padding up to K0) of this thread's quarter
            const int c = c0 + j;
            if (j < cw && c >= 63) {
                if (c < cend) dot = fmaf(gv[j], p.wsig[PM_HID + c], dot);
                put(c, gv[j]);
            }
        }
        dot += __shfl_xor(dot, 1);
        dot += __shfl_xor(dot, 2);
        if (q == 0) { rinv[r] = conv_xinv_of(mx); xdot[r] = dot; }
    }
    __syncthreads();

    // ---------------- the four hidden layers ----------------
    const int frow = lane & 31, fh = lane >> 5;            // fragment row (output channel of A / row of B) and K half
    const int chb = wave * 64;                             // this wave's 64 output channels
    float sig_part[2] = {0.0f, 0.0f};
#pragma unroll 1
    for (int L = 0; L < 4; ++L) {
        const int nks = (L == 0 ? p.K0 : PM_HID) >> 4;     // 16-wide K slices
        const uint16_t* __restrict__ wl = p.w[L];
        pm_f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
        // weight fragment of K slice ks, plane pl, channel tile mt: (K/32, 2, 256, 32) planes
        auto wptr = [&](int ks, int pl, int mt) -> const pm_f16x8* {
            return reinterpret_cast<const pm_f16x8*>(wl + ((((int64_t)(ks >> 1) * 2 + pl) * PM_HID + chb + mt * 32 + frow) << 5) + (ks & 1) * 16 + fh * 8);
        };
        auto aptr = [&](int ks, int pl, int nt) -> const pm_f16x8* {
            return reinterpret_cast<const pm_f16x8*>(act + pl * PL + (nt * 32 + frow) * PM_LDW + (((2 * ks + fh) ^ (frow & 15)) << 3));
        };
        // Weight fragments come straight from L2 (~1 us away): the slices ks + 1 .. ks + 3 travel while slice ks multiplies (ring of four register
        // sets, static indices through the 4-way unrolled walk); the activation fragments of slice ks + 1 are read from LDS under slice ks's MFMAs.
        pm_f16x8 fw[4][2][2];                               // [ring slot][plane][mt]
        pm_f16x8 fa[2][2][2];                               // [slot][plane][nt]
        auto load_w = [&](int slot, int ks) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) fw[slot][pl][mt] = *wptr(ks, pl, mt);
        };
        auto load_a = [&](int slot, int ks) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) fa[slot][pl][nt] = *aptr(ks, pl, nt);
        };
        load_w(0, 0);
        load_w(1, 1);
        if (2 < nks) load_w(2, 2);
        load_a(0, 0);
#pragma unroll 1
        for (int ks0 = 0; ks0 < nks; ks0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ks = ks0 + u;
                if (ks < nks) {                             // (uniform: nks is 10 or 16 for the shipped widths)
                    if (ks + 3 < nks) load_w((u + 3) & 3, ks + 3);
                    if (ks + 1 < nks) load_a((u + 1) & 1, ks + 1);
                    // smallest terms first: w_hi a_lo, w_lo a_hi, then w_hi a_hi
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[u][0][mt], fa[u & 1][1][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[u][1][mt], fa[u & 1][0][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[u][0][mt], fa[u & 1][0][nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
        // ---- epilogue: undo the scales, bias, ReLU; lane = row (nt, frow), registers = channels chb + 32 mt + 8 g + 4 fh + (0..3) ----
        const float wi = p.winv[L];
        float rmax[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float osc = rinv[nt * 32 + frow] * wi;
            float m = 0.0f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bb = *reinterpret_cast<const float4*>(p.b[L] + chb + mt * 32 + 8 * g + 4 * fh);
                    const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = fmaxf(acc[mt][nt][4 * g + j] * osc + bv[j], 0.0f);
                        acc[mt][nt][4 * g + j] = v;
                        m = fmaxf(m, v);
                    }
                }
            m = fmaxf(m, __shfl_xor(m, 32));
            rmax[nt] = m;
        }
        if (L < 3) {
            if (fh == 0) { pmax[wave * PM_BM + frow] = rmax[0]; pmax[wave * PM_BM + 32 + frow] = rmax[1]; }
            __syncthreads();     // every wave is done reading this layer's operand planes; the partial maxima are visible
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int row = nt * 32 + frow;
                const float m = fmaxf(fmaxf(pmax[row], pmax[PM_BM + row]), fmaxf(pmax[2 * PM_BM + row], pmax[3 * PM_BM + row]));
                const float s = conv_xscale_of(m);
                if (wave == 0 && fh == 0) rinv[row] = conv_xinv_of(m);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint32_t h0, l0, h1, l1;
                        pm_split2(acc[mt][nt][4 * g] * s, acc[mt][nt][4 * g + 1] * s, h0, l0);
                        pm_split2(acc[mt][nt][4 * g + 2] * s, acc[mt][nt][4 * g + 3] * s, h1, l1);
                        const int chunk = ((chb + mt * 32) >> 3) + g;
                        uint16_t* dst = act + row * PM_LDW + ((chunk ^ (row & 15)) << 3) + 4 * fh;
                        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
                        *reinterpret_cast<uint2*>(dst + PL) = make_uint2(l0, l1);
                    }
            }
            __syncthreads();
        } else {
            // ---- sigma layer over the trunk output: this wave's 64 channels of w_sigma[:256] . h, per row ----
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float sp = 0.0f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c0 = chb + mt * 32 + 8 * g + 4 * fh;
                        const float4 ws = *reinterpret_cast<const float4*>(p.wsig + c0);
                        sp = fmaf(acc[mt][nt][4 * g], ws.x, sp);
                        sp = fmaf(acc[mt][nt][4 * g + 1], ws.y, sp);
                        sp = fmaf(acc[mt][nt][4 * g + 2], ws.z, sp);
                        sp = fmaf(acc[mt][nt][4 * g + 3], ws.w, sp);
                        const int n = row0 + nt * 32 + frow;
                        if (p.h_out && n < p.N)
                            *reinterpret_cast<float4*>(p.h_out + (int64_t)n * PM_HID + c0) =
                                make_float4(acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
                    }
                sig_part[nt] = sp + __shfl_xor(sp, 32);
            }
            if (fh == 0) { sigp[wave * PM_BM + frow] = sig_part[0]; sigp[wave * PM_BM + 32 + frow] = sig_part[1]; }
        }
    }
    __syncthreads();
    if (tid < PM_BM) {
        const int n = row0 + tid;
        if (n < p.N) {
            const float sg = (((sigp[tid] + sigp[PM_BM + tid]) + sigp[2 * PM_BM + tid]) + sigp[3 * PM_BM + tid]) + xdot[tid] + p.bsig[0];
            if (p.raw) p.raw[n] = sg;
            p.alpha[n] = 1.0f - expf(-fmaxf(sg, 0.0f));     // F.relu (nerf_mlp.py:227), 1 - exp(-density) (nerfdet.py:257)
        }
    }
}

extern "C" int ndet_point_mlp_alpha(const float* points, const float* global_feat, int N, int F, int K0, int hidden, const uint16_t* const* w_planes_host,
                                    const float* w_inv_scale_host, const float* const* bias_host, const float* w_sigma, const float* b_sigma,
                                    float* raw_sigma, float* alpha, float* h_out, void* stream) {
    const char* fn = "ndet_point_mlp_alpha";
    NDET_REQUIRE(points && (global_feat || F == 0) && w_planes_host && w_inv_scale_host && bias_host && w_sigma && b_sigma && alpha, NDET_E_INVALID,
                 "%s: null pointer", fn);
    NDET_REQUIRE(N > 0 && F >= 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE(hidden == PM_HID, NDET_E_UNSUPPORTED, "%s: hidden width %d (the fused kernel is built for %d; use the layer-by-layer path)", fn, hidden, PM_HID);
    NDET_REQUIRE(K0 % 32 == 0 && K0 >= 63 + F && K0 <= PM_LDW, NDET_E_UNSUPPORTED, "%s: padded input width %d must be a multiple of 32 in [63 + F, %d]", fn, K0, PM_LDW);
    NDET_REQUIRE((int64_t)N * (F > PM_HID ? F : PM_HID) < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: too many rows", fn);
    PointMlpParams p;
    p.points = points; p.glob = global_feat; p.N = N; p.F = F; p.K0 = K0;
    for (int l = 0; l < 4; ++l) {
        NDET_REQUIRE(w_planes_host[l] && bias_host[l], NDET_E_INVALID, "%s: layer %d: null pointer", fn, l);
        NDET_REQUIRE((((uintptr_t)w_planes_host[l] | (uintptr_t)bias_host[l]) & 15) == 0, NDET_E_UNSUPPORTED, "%s: layer %d: planes / bias must be 16-byte aligned", fn, l);
        p.w[l] = w_planes_host[l]; p.winv[l] = w_inv_scale_host[l]; p.b[l] = bias_host[l];
    }
    NDET_REQUIRE(((uintptr_t)w_sigma & 15) == 0 && (!h_out || ((uintptr_t)h_out & 15) == 0), NDET_E_UNSUPPORTED, "%s: w_sigma / h_out must be 16-byte aligned", fn);
    p.wsig = w_sigma; p.bsig = b_sigma; p.raw = raw_sigma; p.alpha = alpha; p.h_out = h_out;
    const size_t lds = (size_t)2 * PM_BM * PM_LDW * sizeof(uint16_t) + (size_t)(4 * PM_BM + PM_BM + PM_BM + 4 * PM_BM) * sizeof(float);
    static int attr_state[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (attr_state[dev] == 0) {
        hipError_t e = hipFuncSetAttribute((const void*)k_point_mlp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        NDET_REQUIRE(e == hipSuccess, NDET_E_LAUNCH, "%s: cannot raise the LDS limit to %zu bytes: %s", fn, lds, hipGetErrorString(e));
        attr_state[dev] = 1;
    }
    const int64_t blocks = ((int64_t)N + PM_BM - 1) / PM_BM;
    hipLaunchKernelGGL(k_point_mlp, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}
