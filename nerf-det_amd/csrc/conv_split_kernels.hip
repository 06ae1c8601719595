// A13/A14 + the 2D backbone convolutions: fp32 convolution on the bf16 matrix cores by operand splitting.
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the fp32-input MFMA (MI355X_MICROARCH.md, Matrix cores).
// Every fp32 operand is written EXACTLY as a sum of three bf16 numbers,
//     x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)       (3 x 8 = 24 significand bits)
// and a*b is accumulated in fp32 as the six products of total order <= 2,
//     a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0),
// each of which is exact in fp32 (8 x 8 significand bits).  The dropped terms (a1b2, a2b1, a2b2) are <= 2^-24 |ab|, i.e.
// below one fp32 rounding of the product: the result carries the same error as an fp32 FMA chain (measured against
// an fp64 convolution in tests/test_conv3d_gpu.py), at 6/16 of the fp32-MFMA issue time.
//
// Layout: weights arrive pre-split as three bf16 planes (3, taps, Cout, Cin) (ndet_split_bf16x3 below, once per model);
// activations stay fp32 channels-last in HBM and are split while they are staged into LDS.  LDS holds three bf16 planes
// per operand, K-contiguous rows of 32 k (64 B) padded to 80 B so that the ds_read_b128 fragment reads (lane -> row
// l & 31, k-octet l >> 5) fall on 16 distinct 16-byte slots per 16 lanes.  One LDS stage (60 KB at 128 x 128) with the
// next K step prefetched into registers: two workgroups fit on a CU and cover each other's barrier / split phases.
// Wave tile 64 x 64 (2 x 2 MFMA tiles of 32 x 32): 12 fragment reads feed 24 MFMAs per 16-k sub-step.
#include "conv_common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define SPL_RS 40  // LDS row stride in bf16 elements (32 k + 8 pad = 80 B)

// two fp32 -> packed bf16 pair (v_cvt_pk_bf16_f32, round to nearest even)
__device__ __forceinline__ uint32_t spl_pack(float x, float y) {
    const bf16x2 v = __builtin_convertvector((f32x2){x, y}, bf16x2);
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float spl_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float spl_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// 8 fp32 -> three planes of 8 bf16 (16 B each)
__device__ __forceinline__ void spl_split8(const float4 lo, const float4 hi, uint4& p0, uint4& p1, uint4& p2) {
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t o0[4], o1[4], o2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        o0[i] = spl_pack(a, b);
        const float ra = a - spl_lo(o0[i]), rb = b - spl_hi(o0[i]);   // exact
        o1[i] = spl_pack(ra, rb);
        const float sa = ra - spl_lo(o1[i]), sb = rb - spl_hi(o1[i]); // exact, <= 8 significant bits left
        o2[i] = spl_pack(sa, sb);
    }
    p0 = make_uint4(o0[0], o0[1], o0[2], o0[3]);
    p1 = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    p2 = make_uint4(o2[0], o2[1], o2[2], o2[3]);
}

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void k_conv_split(const Conv3dParams p, const uint16_t* __restrict__ wsplit, int64_t plane_elems) {
    constexpr int NTHR = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;   // per-wave tile
    constexpr int MT = WM / 32, NT = WN / 32;     // 32x32 MFMA tiles per wave
    constexpr int RPP = NTHR / 4;                 // tile rows staged per pass (4 threads x 8 k cover one 32-k row)
    constexpr int AR = BM / RPP, BR = BN / RPP;
    static_assert(AR >= 1 && BR >= 1 && MT >= 1 && NT >= 1, "tile too small for the thread count");
    constexpr int APL = BM * SPL_RS, BPL = BN * SPL_RS;  // one plane, in bf16 elements
    extern __shared__ __attribute__((aligned(16))) uint16_t lds16[];
    uint16_t* As = lds16;             // [3][BM][SPL_RS]
    uint16_t* Bs = lds16 + 3 * APL;   // [3][BN][SPL_RS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kg = tid & 3;         // which 8-k octet of the 32-k run
    const int srow = tid >> 2;      // 0..RPP-1

    const int cin_steps = p.Cin / CBK;
    const int taps = p.transposed ? 1 : p.kd * p.kh * p.kw;
    const int n_iters_all = taps * cin_steps;
    int it_begin = 0, it_end = n_iters_all;
    int ztap = 0;
    if (p.transposed) {
        ztap = blockIdx.z;
    } else if (p.splits > 1) {
        const int s = blockIdx.z;
        it_begin = (int)((int64_t)n_iters_all * s / p.splits);
        it_end = (int)((int64_t)n_iters_all * (s + 1) / p.splits);
    }

    int vd[AR], vh[AR], vw[AR];
    bool vok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + srow + RPP * i;
        vok[i] = m < p.M;
        const int mm = vok[i] ? m : 0;
        const int ow_ = p.transposed ? p.W : p.OW, oh_ = p.transposed ? p.H : p.OH;
        vw[i] = mm % ow_;
        vh[i] = (mm / ow_) % oh_;
        vd[i] = mm / (ow_ * oh_);
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[AR][2];
    uint4 rb[BR][3];
    const float* arow[AR];
    const uint16_t* brow[BR];
    int cur_tap = -1;
    auto enter_tap = [&](int tap) {
        cur_tap = tap;
        const int kd = tap / (p.kh * p.kw), kh = (tap / p.kw) % p.kh, kw = tap % p.kw;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            int id, ih, iw;
            if (p.transposed) {
                id = vd[i]; ih = vh[i]; iw = vw[i];
            } else {
                id = vd[i] * p.sd + kd - p.pd;
                ih = vh[i] * p.sh + kh - p.ph;
                iw = vw[i] * p.sw + kw - p.pw;
            }
            const bool ok = vok[i] && id >= 0 && id < p.D && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            arow[i] = ok ? p.in + ((int64_t)(id * p.H + ih) * p.W + iw) * p.Cin + kg * 8 : nullptr;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int co = n0 + srow + RPP * i;
            brow[i] = co < p.Cout ? wsplit + ((int64_t)tap * p.Cout + co) * p.Cin + kg * 8 : nullptr;
        }
    };
    auto load_tile = [&](int it) {
        const int tap = p.transposed ? ztap : it / cin_steps;
        if (tap != cur_tap) enter_tap(tap);
        const int ci0 = (p.transposed ? it : it - tap * cin_steps) * CBK;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            ra[i][0] = ra[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (arow[i]) {
                ra[i][0] = *reinterpret_cast<const float4*>(arow[i] + ci0);
                ra[i][1] = *reinterpret_cast<const float4*>(arow[i] + ci0 + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                rb[i][pl] = make_uint4(0u, 0u, 0u, 0u);
                if (brow[i]) rb[i][pl] = *reinterpret_cast<const uint4*>(brow[i] + pl * plane_elems + ci0);
            }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            uint4 s0, s1, s2;
            spl_split8(ra[i][0], ra[i][1], s0, s1, s2);
            uint16_t* dst = As + (srow + RPP * i) * SPL_RS + kg * 8;
            *reinterpret_cast<uint4*>(dst) = s0;
            *reinterpret_cast<uint4*>(dst + APL) = s1;
            *reinterpret_cast<uint4*>(dst + 2 * APL) = s2;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            uint16_t* dst = Bs + (srow + RPP * i) * SPL_RS + kg * 8;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint4*>(dst + pl * BPL) = rb[i][pl];
        }
    };

    if (it_begin < it_end) {
        load_tile(it_begin);
        store_tile();
    }
    __syncthreads();

    // fragment of the 32x32x16 MFMA: lane l holds row (l & 31), k = 8 (l >> 5) + j, j = 0..7 -> one 16-byte read
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    const uint16_t* abase = As + (wm * WM + frow) * SPL_RS + fk;
    const uint16_t* bbase = Bs + (wn * WN + frow) * SPL_RS + fk;
    for (int it = it_begin; it < it_end; ++it) {
        const bool more = it + 1 < it_end;
        if (more) load_tile(it + 1);
#pragma unroll
        for (int ks = 0; ks < CBK / 16; ++ks) {
            bf16x8 fa[3][MT], fb[3][NT];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int t = 0; t < MT; ++t) fa[pl][t] = *reinterpret_cast<const bf16x8*>(abase + pl * APL + t * 32 * SPL_RS + ks * 16);
#pragma unroll
                for (int t = 0; t < NT; ++t) fb[pl][t] = *reinterpret_cast<const bf16x8*>(bbase + pl * BPL + t * 32 * SPL_RS + ks * 16);
            }
            // smallest terms first; the (pa, pb) pairs with pa + pb <= 2
#pragma unroll
            for (int order = 2; order >= 0; --order)
#pragma unroll
                for (int pa = 0; pa <= order; ++pa) {
                    const int pb = order - pa;
#pragma unroll
                    for (int ta = 0; ta < MT; ++ta)
#pragma unroll
                        for (int tb = 0; tb < NT; ++tb)
                            acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa][ta], fb[pb][tb], acc[ta][tb], 0, 0, 0);
                }
        }
        __syncthreads();   // every wave is done reading this K step
        if (more) store_tile();
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS (one wave-row of the tile at a time) -> fused row-wise stores ----
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    constexpr int CLDC = BN + 4;
    float* Cs = reinterpret_cast<float*>(lds16);   // [WM][CLDC] floats <= the operand planes
    for (int h = 0; h < WGM; ++h) {
        if (wm == h) {
#pragma unroll
            for (int ta = 0; ta < MT; ++ta)
#pragma unroll
                for (int tb = 0; tb < NT; ++tb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Cs[(ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CLDC + wn * WN + tb * 32 + (lane & 31)] = acc[ta][tb][r];
        }
        __syncthreads();
        conv_store_rows<BN, NTHR>(p, Cs, CLDC, m0 + h * WM, WM, n0, tid, ztap, blockIdx.z);
        __syncthreads();
    }
}

template <int BM, int BN, int WGM, int WGN>
static int split_launch_tile(const Conv3dParams& p, hipStream_t st, const char* fn) {
    const int zdim = p.transposed ? 8 : p.splits;
    dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, zdim);
    size_t lds = (size_t)3 * (BM + BN) * SPL_RS * sizeof(uint16_t);
    const size_t cs = (size_t)(BM / WGM) * (BN + 4) * sizeof(float);
    if (cs > lds) lds = cs;
    if (lds > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)k_conv_split<BM, BN, WGM, WGN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            NDET_REQUIRE(e == hipSuccess, NDET_E_LAUNCH, "%s: cannot raise the LDS limit: %s", fn, hipGetErrorString(e));
            attr_set = true;
        }
    }
    const int64_t plane = (int64_t)(p.transposed ? 8 : p.kd * p.kh * p.kw) * p.Cout * p.Cin;
    hipLaunchKernelGGL((k_conv_split<BM, BN, WGM, WGN>), grid, dim3(64 * WGM * WGN), lds, st, p, (const uint16_t*)p.w, plane);
    return NDET_OK;
}

int conv_split_launch(Conv3dParams& p, int tile, hipStream_t st, const char* fn) {
    const int64_t big_tiles = (int64_t)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    if (tile == 0) tile = (big_tiles >= 192 && p.Cout >= 128) ? 128 : 64;
    int rc;
    switch (tile) {
        case 64: rc = split_launch_tile<64, 64, 2, 2>(p, st, fn); break;
        case 128: rc = split_launch_tile<128, 128, 2, 2>(p, st, fn); break;
        case 12864: rc = split_launch_tile<128, 64, 2, 2>(p, st, fn); break;
        default: ndet_set_error("%s: unknown tile %d", fn, tile); return NDET_E_INVALID;
    }
    if (rc != NDET_OK) return rc;
    NDET_CHECK_LAUNCH(fn);
    return conv_splitk_reduce_launch(p, st, fn);
}

// fp32 array -> three bf16 planes (plane stride n): x = p0 + p1 + p2 exactly
__global__ __launch_bounds__(256) void k_split_bf16x3(const float* __restrict__ x, int64_t n, uint16_t* __restrict__ planes) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    const float a = x[i], b = (i + 1 < n) ? x[i + 1] : 0.f;
    const uint32_t o0 = spl_pack(a, b);
    const float ra = a - spl_lo(o0), rb = b - spl_hi(o0);
    const uint32_t o1 = spl_pack(ra, rb);
    const uint32_t o2 = spl_pack(ra - spl_lo(o1), rb - spl_hi(o1));
    const uint32_t o[3] = {o0, o1, o2};
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        planes[pl * n + i] = (uint16_t)(o[pl] & 0xffffu);
        if (i + 1 < n) planes[pl * n + i + 1] = (uint16_t)(o[pl] >> 16);
    }
}

extern "C" int ndet_split_bf16x3(const float* x, int64_t n, uint16_t* planes, void* stream) {
    const char* fn = "ndet_split_bf16x3";
    NDET_REQUIRE(x && planes, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n > 0, NDET_E_INVALID, "%s: n must be positive", fn);
    const int64_t work = (n + 1) / 2;
    hipLaunchKernelGGL(k_split_bf16x3, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, planes);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

extern "C" int ndet_conv_ndhwc_split(const float* in, const uint16_t* w_planes, float* out, int D, int H, int W, int Cin, int Cout,
                                     const int* kernel, const int* stride, const int* pad, int transposed, const float* scale,
                                     const float* shift, const float* residual, int residual_up2, int relu, int splits, int tile,
                                     void* workspace, void* stream) {
    const char* fn = "ndet_conv_ndhwc_split";
    NDET_REQUIRE(in && w_planes && out && kernel && stride && pad, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE((scale == nullptr) == (shift == nullptr), NDET_E_INVALID, "%s: scale and shift go together", fn);
    NDET_REQUIRE(relu >= 0 && relu <= 2 && (tile == 0 || tile == 64 || tile == 128 || tile == 12864), NDET_E_INVALID, "%s: bad relu mode / tile", fn);
    NDET_REQUIRE(Cin % CBK == 0, NDET_E_UNSUPPORTED, "%s: Cin=%d must be a multiple of %d", fn, Cin, CBK);
    NDET_REQUIRE((((uintptr_t)in | (uintptr_t)w_planes) & 15) == 0, NDET_E_UNSUPPORTED, "%s: in / weights must be 16-byte aligned", fn);
    Conv3dParams p;
    p.in = in; p.w = reinterpret_cast<const float*>(w_planes); p.out = out; p.scale = scale; p.shift = shift; p.res = residual;
    p.partial = (float*)workspace;
    p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
    if (transposed) {
        for (int a = 0; a < 3; ++a)
            NDET_REQUIRE(kernel[a] == 2 && stride[a] == 2 && pad[a] == 0, NDET_E_UNSUPPORTED, "%s: transposed conv supports kernel 2 stride 2 pad 0 only", fn);
        p.transposed = 1;
        p.kd = p.kh = p.kw = 2; p.sd = p.sh = p.sw = 2; p.pd = p.ph = p.pw = 0;
        p.OD = 2 * D; p.OH = 2 * H; p.OW = 2 * W;
        NDET_REQUIRE((int64_t)p.OD * p.OH * p.OW < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: tensor too large", fn);
        p.M = D * H * W;
        p.splits = 1; p.partial = nullptr;
        p.res_up2 = 0; p.RH = p.RW = 0;
        return conv_split_launch(p, tile, (hipStream_t)stream, fn);
    }
    for (int a = 0; a < 3; ++a)
        NDET_REQUIRE(kernel[a] >= 1 && kernel[a] <= 7 && stride[a] >= 1 && stride[a] <= 4 && pad[a] >= 0 && pad[a] < kernel[a], NDET_E_UNSUPPORTED,
                     "%s: kernel/stride/pad out of range on axis %d", fn, a);
    p.transposed = 0;
    p.kd = kernel[0]; p.kh = kernel[1]; p.kw = kernel[2];
    p.sd = stride[0]; p.sh = stride[1]; p.sw = stride[2];
    p.pd = pad[0]; p.ph = pad[1]; p.pw = pad[2];
    p.OD = (D + 2 * p.pd - p.kd) / p.sd + 1;
    p.OH = (H + 2 * p.ph - p.kh) / p.sh + 1;
    p.OW = (W + 2 * p.pw - p.kw) / p.sw + 1;
    NDET_REQUIRE(p.OD > 0 && p.OH > 0 && p.OW > 0, NDET_E_INVALID, "%s: empty output", fn);
    NDET_REQUIRE((int64_t)p.OD * p.OH * p.OW < ((int64_t)1 << 31) && (int64_t)D * H * W * Cin < ((int64_t)1 << 40), NDET_E_UNSUPPORTED, "%s: tensor too large", fn);
    p.M = p.OD * p.OH * p.OW;
    p.splits = splits < 1 ? 1 : splits;
    p.res_up2 = (residual && residual_up2) ? 1 : 0;
    p.RH = (p.OH + 1) / 2; p.RW = (p.OW + 1) / 2;
    NDET_REQUIRE(!(p.res_up2 && p.splits > 1), NDET_E_UNSUPPORTED, "%s: upsampled residual cannot be combined with split-K", fn);
    const int iters = p.kd * p.kh * p.kw * (Cin / CBK);
    NDET_REQUIRE(p.splits <= iters, NDET_E_INVALID, "%s: splits=%d exceeds the %d K steps", fn, p.splits, iters);
    NDET_REQUIRE(p.splits == 1 || workspace != nullptr, NDET_E_INVALID, "%s: split-K needs a workspace", fn);
    return conv_split_launch(p, tile, (hipStream_t)stream, fn);
}
