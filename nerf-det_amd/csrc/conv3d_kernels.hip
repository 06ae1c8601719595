// A13/A14: 3D convolutions of the voxel neck and head as a direct implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains, no reduced precision), channels-last (NDHWC).
//
//   out[m, co] = epilogue( sum_{tap} sum_{ci} in[nbr(m, tap), ci] * Wp[tap][co][ci] )
//
// GEMM view: M = output voxels, N = Cout, K = taps * Cin.  One workgroup (128 x 128 tile: 8 waves as 4 x 2; 64 x 64
// tile: 4 waves as 2 x 2) computes a BM x BN tile; K is walked tap-major in steps of BK = 32 input channels.  A rows are gathered on the fly (one
// 128-byte channel run of the neighbour voxel per (row, tap), zeros outside the grid) -- nothing like the 27x
// im2col buffer the vendor path materialises.  Both operand tiles sit K-contiguous in LDS ([rows][BK + 4]); a lane
// reads 4 consecutive k with one ds_read_b128 and feeds them to 4 successive MFMAs (the k permutation is the same
// for A and B, so the product is unchanged).  Global -> register -> LDS staging is double buffered: the loads of
// K-step i+1 are in flight while step i is multiplied; one barrier per step.
// Epilogue (fused): per-channel scale/shift (eval-mode BatchNorm as alpha = gamma/sqrt(var+eps), beta = bias -
// mean*alpha, the form ATen's CPU kernel uses), residual add, ReLU.  Split-K variants write raw partial sums to a
// workspace and a small second kernel reduces them in a fixed order (bitwise reproducible) and applies the epilogue.
//
// Replaces nn.Conv3d / nn.ConvTranspose3d(k=2,s=2) + BatchNorm3d(eval) + ReLU of
// mmdet3d/models/necks/imvoxelnet.py:8-67,233-260 and the head convs of dense_heads/imvoxel_head_v2.py:45-49.
#include "conv_common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS row stride in floats.  A lane (row = l & 15, k-group g = l >> 4) reads 16 B at row*LD + 4g; ds_read_b128 is served
// in 16-lane groups {0-3,12-15,20-27},... over 64 banks.  LD = 40 makes the 16 starts of a group hit 16 distinct
// 4-bank slots (conflict-free); LD = 36 leaves 2-way conflicts (measured: a third of the LDS cycles) but lets a
// fourth 64x64 workgroup fit on the CU, which matters more for the small memory-bound layers.
template <int BM> struct LdsStride { static constexpr int value = (BM >= 128) ? CBK + 8 : CBK + 4; };


template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN == 8) ? 4 : 1) void k_conv3d_igemm(const Conv3dParams p) {
    constexpr int NTHR = 64 * WGM * WGN;        // WGM x WGN waves
    constexpr int WM = BM / WGM, WN = BN / WGN; // per-wave tile
    constexpr int MT = WM / 16, NT = WN / 16;   // 16x16 MFMA tiles per wave
    constexpr int RPP = NTHR / 8;               // tile rows staged per pass (8 threads cover one 128-byte K run)
    constexpr int AR = BM / RPP, BR = BN / RPP; // rows each thread stages per operand
    constexpr int CLD = LdsStride<BM>::value;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                       // [2][BM][CLD]
    float* Bs = lds + 2 * BM * CLD;        // [2][BN][CLD]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int chunk = tid & 7;      // which 4-float piece of the 32-float K run
    const int srow = tid >> 3;      // 0..RPP-1

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

    // decode the output (or, transposed, input) voxel of the A rows this thread stages
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

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra[AR], rb[BR];

    // Per-tap state, recomputed only when the K walk enters a new tap: for each staged A row the address of the
    // neighbour voxel's channel run (or null outside the grid), for each staged B row the weight row.
    const float* arow[AR];
    const float* brow[BR];
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
            arow[i] = ok ? p.in + ((int64_t)(id * p.H + ih) * p.W + iw) * p.Cin + chunk * 4 : nullptr;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int co = n0 + srow + RPP * i;
            brow[i] = co < p.Cout ? p.w + ((int64_t)tap * p.Cout + co) * p.Cin + chunk * 4 : nullptr;
        }
    };
    auto load_tile = [&](int it) {
        const int tap = p.transposed ? ztap : it / cin_steps;
        if (tap != cur_tap) enter_tap(tap);
        const int ci0 = (p.transposed ? it : it - tap * cin_steps) * CBK;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (arow[i]) ra[i] = *reinterpret_cast<const float4*>(arow[i] + ci0);
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (brow[i]) rb[i] = *reinterpret_cast<const float4*>(brow[i] + ci0);
        }
    };
    auto store_tile = [&](int buf) {
        float* a = As + buf * BM * CLD;
        float* b = Bs + buf * BN * CLD;
#pragma unroll
        for (int i = 0; i < AR; ++i) *reinterpret_cast<float4*>(a + (srow + RPP * i) * CLD + chunk * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i) *reinterpret_cast<float4*>(b + (srow + RPP * i) * CLD + chunk * 4) = rb[i];
    };

    if (it_begin < it_end) {
        load_tile(it_begin);
        store_tile(0);
    }
    __syncthreads();

    const int frow = lane & 15, fk = (lane >> 4) * 4;
    for (int it = it_begin; it < it_end; ++it) {
        const int buf = (it - it_begin) & 1;
        if (it + 1 < it_end) load_tile(it + 1);
        const float* a = As + buf * BM * CLD + (wm * WM + frow) * CLD + fk;
        const float* b = Bs + buf * BN * CLD + (wn * WN + frow) * CLD + fk;
        // all fragment reads of the K step are issued up front (the second half's LDS latency hides behind the first
        // half's MFMAs; the compiler places counted lgkmcnt waits), then 2 x (MT*NT*4) MFMAs
        constexpr int KH = CBK / 16;
        float4 fa[KH][MT], fb[KH][NT];
#pragma unroll
        for (int hk = 0; hk < KH; ++hk) {
#pragma unroll
            for (int t = 0; t < MT; ++t) fa[hk][t] = *reinterpret_cast<const float4*>(a + t * 16 * CLD + hk * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t) fb[hk][t] = *reinterpret_cast<const float4*>(b + t * 16 * CLD + hk * 16);
        }
#pragma unroll
        for (int hk = 0; hk < KH; ++hk) {
            // j outermost: MT*NT independent accumulators between two dependent MFMAs (40-cycle dependent latency)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ta = 0; ta < MT; ++ta)
#pragma unroll
                    for (int tb = 0; tb < NT; ++tb) {
                        const float av = j == 0 ? fa[hk][ta].x : j == 1 ? fa[hk][ta].y : j == 2 ? fa[hk][ta].z : fa[hk][ta].w;
                        const float bv = j == 0 ? fb[hk][tb].x : j == 1 ? fb[hk][tb].y : j == 2 ? fb[hk][tb].z : fb[hk][tb].w;
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[ta][tb], 0, 0, 0);
                    }
        }
        if (it + 1 < it_end) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ----
    // The accumulators go through LDS (the operand buffers are free now) so that global traffic is whole rows:
    // each thread then owns float4 pieces along Cout -- residual reads and output writes are 16 B per lane, 512 B
    // contiguous per 32 lanes -- instead of the MFMA C layout's 64-byte fragments.  For the 1x1 convolutions of the
    // 2D backbone (K = 64..256) the epilogue IS the kernel: 3.5x faster this way.
    constexpr int CLDC = BN + 4;  // C tile row stride (floats)
    float* Cs = lds;              // [BM][CLDC]  (BM * (BN+4) * 4 B <= the operand buffers)
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int ta = 0; ta < MT; ++ta)
#pragma unroll
        for (int tb = 0; tb < NT; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[(wm * WM + ta * 16 + (lane >> 4) * 4 + r) * CLDC + wn * WN + tb * 16 + (lane & 15)] = acc[ta][tb][r];
    __syncthreads();

    float mx = 0.0f;
    conv_store_rows<BN, NTHR>(p, Cs, CLDC, m0, BM, n0, tid, ztap, blockIdx.z, mx, 1.0f);
    if (p.amax_out && conv_writes_final(p)) conv_amax_commit(p.amax_out, mx);
}

// fixed-order reduction of the split-K partials + epilogue (grid-stride: at most 1024 workgroups, one amax atomic each at most)
__global__ __launch_bounds__(256) void k_conv3d_splitk_reduce(const float* __restrict__ partial, int splits, int64_t MN, int Cout,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ res, int relu, float* __restrict__ out,
                                                              float* __restrict__ amax_out) {
    // VEC = 4 when Cout % 4 == 0: one float4 per thread and pass
    const bool vec = (Cout & 3) == 0;
    // one contiguous range of the output per workgroup (not a grid-stride walk): the maximum a workgroup commits is then the maximum of a region,
    // as the range guard's tile minimum assumes (conv_common.hpp::conv_tilemin_read)
    const int64_t unit = vec ? 4 : 1, units = (MN + unit - 1) / unit;
    const int64_t per = (units + gridDim.x - 1) / gridDim.x;
    const int64_t hi = ((int64_t)blockIdx.x * per + per < units ? (int64_t)blockIdx.x * per + per : units) * unit;
    const int64_t step = (int64_t)blockDim.x * unit;
    float mx = 0.0f;
    for (int64_t i = ((int64_t)blockIdx.x * per + threadIdx.x) * unit; i < hi && i < MN; i += step) {
        if (vec) {
            float4 v = *reinterpret_cast<const float4*>(partial + i);
            for (int s = 1; s < splits; ++s) {
                const float4 t = *reinterpret_cast<const float4*>(partial + (int64_t)s * MN + i);
                v.x = v.x + t.x; v.y = v.y + t.y; v.z = v.z + t.z; v.w = v.w + t.w;
            }
            const int co = (int)(i % Cout);
            if (scale) {
                const float4 sc = *reinterpret_cast<const float4*>(scale + co), sh = *reinterpret_cast<const float4*>(shift + co);
                v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
            }
            if (relu == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (res) {
                const float4 rr = *reinterpret_cast<const float4*>(res + i);
                v.x = v.x + rr.x; v.y = v.y + rr.y; v.z = v.z + rr.z; v.w = v.w + rr.w;
            }
            if (relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(out + i) = v;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        } else {
            float v = partial[i];
            for (int s = 1; s < splits; ++s) v = v + partial[(int64_t)s * MN + i];
            const int co = (int)(i % Cout);
            if (scale) v = v * scale[co] + shift[co];
            if (relu == 2) v = fmaxf(v, 0.0f);
            if (res) v = v + res[i];
            if (relu == 1) v = fmaxf(v, 0.0f);
            out[i] = v;
            mx = fmaxf(mx, fabsf(v));
        }
    }
    if (amax_out) conv_amax_commit(amax_out, mx);
}

template <int BM, int BN, int WGM, int WGN>
static int conv_launch_tile(const Conv3dParams& p, hipStream_t st, const char* fn) {
    const int zdim = p.transposed ? 8 : p.splits;
    dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, zdim);
    const size_t lds = (size_t)2 * (BM + BN) * LdsStride<BM>::value * sizeof(float);
    if (lds > 64 * 1024) {  // above the default dynamic-LDS cap
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)k_conv3d_igemm<BM, BN, WGM, WGN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            NDET_REQUIRE(e == hipSuccess, NDET_E_LAUNCH, "%s: cannot raise the LDS limit: %s", fn, hipGetErrorString(e));
            attr_set = true;
        }
    }
    hipLaunchKernelGGL((k_conv3d_igemm<BM, BN, WGM, WGN>), grid, dim3(64 * WGM * WGN), lds, st, p);
    return NDET_OK;
}

int conv_f32_launch(Conv3dParams& p, int tile, hipStream_t st, const char* fn) {
    const int Cout = p.Cout;
    const int64_t big_tiles = (int64_t)((p.M + 127) / 128) * ((Cout + 127) / 128);
    if (tile == 0) tile = (big_tiles >= 256 && Cout >= 128) ? 128 : 64;
    int rc;
    switch (tile) {
        case 64: rc = conv_launch_tile<64, 64, 2, 2>(p, st, fn); break;
        case 128: rc = conv_launch_tile<128, 128, 4, 2>(p, st, fn); break;  // 8 waves (4 x 2): +3 % over 2 x 2 waves, measured
        default: ndet_set_error("%s: unknown tile %d", fn, tile); return NDET_E_INVALID;
    }
    if (rc != NDET_OK) return rc;
    NDET_CHECK_LAUNCH(fn);
    return conv_splitk_reduce_launch(p, st, fn);
}

int conv_splitk_reduce_launch(const Conv3dParams& p, hipStream_t st, const char* fn) {
    if (p.transposed || p.splits <= 1) return NDET_OK;
    const int64_t mn = (int64_t)p.M * p.Cout;
    const int64_t work = (p.Cout & 3) == 0 ? mn / 4 : mn;
    int64_t blocks = (work + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_conv3d_splitk_reduce, dim3((unsigned)blocks), dim3(256), 0, st, p.partial, p.splits, mn, p.Cout, p.scale,
                       p.shift, p.res, p.relu, p.out, p.amax_out);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

extern "C" int ndet_conv_ndhwc(const float* in, const float* w_packed, float* out, int D, int H, int W, int Cin, int Cout,
                               const int* kernel, const int* stride, const int* pad, const float* scale, const float* shift,
                               const float* residual, int residual_up2, int relu, int splits, int tile, void* workspace, void* stream) {
    const char* fn = "ndet_conv_ndhwc";
    NDET_REQUIRE(in && w_packed && out && kernel && stride && pad, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE((scale == nullptr) == (shift == nullptr), NDET_E_INVALID, "%s: scale and shift go together", fn);
    NDET_REQUIRE(relu >= 0 && relu <= 2 && (tile == 0 || tile == 64 || tile == 128), NDET_E_INVALID, "%s: bad relu mode / tile", fn);
    NDET_REQUIRE(Cin % CBK == 0, NDET_E_UNSUPPORTED, "%s: Cin=%d must be a multiple of %d", fn, Cin, CBK);
    NDET_REQUIRE((((uintptr_t)in | (uintptr_t)w_packed) & 15) == 0, NDET_E_UNSUPPORTED, "%s: in / weights must be 16-byte aligned", fn);
    for (int a = 0; a < 3; ++a)
        NDET_REQUIRE(kernel[a] >= 1 && kernel[a] <= 7 && stride[a] >= 1 && stride[a] <= 4 && pad[a] >= 0 && pad[a] < kernel[a], NDET_E_UNSUPPORTED,
                     "%s: kernel/stride/pad out of range on axis %d", fn, a);
    Conv3dParams p;
    p.max_order = 2;
    p.in = in; p.w = w_packed; p.out = out; p.scale = scale; p.shift = shift; p.res = residual; p.partial = (float*)workspace;
    p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu; p.transposed = 0;
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
    return conv_f32_launch(p, tile, (hipStream_t)stream, fn);
}

extern "C" int64_t ndet_conv3d_workspace_bytes(int D, int H, int W, int Cin, int Cout, int ksize, int stride, int splits) {
    if (splits <= 1) return 0;
    const int pad = ksize / 2;
    const int64_t od = (D + 2 * pad - ksize) / stride + 1, oh = (H + 2 * pad - ksize) / stride + 1, ow = (W + 2 * pad - ksize) / stride + 1;
    return od * oh * ow * (int64_t)Cout * splits * 4;
}

extern "C" int ndet_conv3d_ndhwc(const float* in, const float* w_packed, float* out, int D, int H, int W, int Cin, int Cout, int ksize,
                                 int stride, int transposed, const float* scale, const float* shift, const float* residual, int relu,
                                 int splits, int tile, void* workspace, void* stream) {
    const char* fn = "ndet_conv3d_ndhwc";
    if (!transposed) {
        NDET_REQUIRE((ksize == 3 || ksize == 1) && (stride == 1 || stride == 2), NDET_E_UNSUPPORTED, "%s: kernel %d stride %d unsupported", fn, ksize, stride);
        const int k[3] = {ksize, ksize, ksize}, s[3] = {stride, stride, stride}, pd[3] = {ksize / 2, ksize / 2, ksize / 2};
        return ndet_conv_ndhwc(in, w_packed, out, D, H, W, Cin, Cout, k, s, pd, scale, shift, residual, 0, relu, splits, tile, workspace, stream);
    }
    NDET_REQUIRE(in && w_packed && out, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE((scale == nullptr) == (shift == nullptr), NDET_E_INVALID, "%s: scale and shift go together", fn);
    NDET_REQUIRE(relu >= 0 && relu <= 2 && (tile == 0 || tile == 64 || tile == 128), NDET_E_INVALID, "%s: bad relu mode / tile", fn);
    NDET_REQUIRE(Cin % CBK == 0, NDET_E_UNSUPPORTED, "%s: Cin=%d must be a multiple of %d", fn, Cin, CBK);
    NDET_REQUIRE((((uintptr_t)in | (uintptr_t)w_packed) & 15) == 0, NDET_E_UNSUPPORTED, "%s: in / weights must be 16-byte aligned", fn);
    NDET_REQUIRE(ksize == 2 && stride == 2, NDET_E_UNSUPPORTED, "%s: transposed conv supports kernel 2 stride 2 only", fn);
    Conv3dParams p;
    p.max_order = 2;
    p.in = in; p.w = w_packed; p.out = out; p.scale = scale; p.shift = shift; p.res = residual; p.partial = nullptr;
    p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu; p.transposed = 1;
    p.kd = p.kh = p.kw = 2; p.sd = p.sh = p.sw = 2; p.pd = p.ph = p.pw = 0;
    p.OD = 2 * D; p.OH = 2 * H; p.OW = 2 * W;
    p.M = D * H * W;
    p.splits = 1;
    p.res_up2 = 0; p.RH = p.RW = 0;
    return conv_f32_launch(p, tile, (hipStream_t)stream, fn);
}

// ------------------------------------------------------------------------------------------------
// ResNet stem tail in one pass: eval-mode BatchNorm (scale/shift) + ReLU + 3x3 stride-2 pad-1 max-pool on the 7x7 stem
// convolution's channels-last output (the stem conv itself, Cin = 3, stays on the vendor library).
// Replaces three elementwise passes over the largest activation of the network (245 MB at cfg2).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bn_relu_maxpool(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int N, int H, int W, int C, int OH, int OW, float* __restrict__ out) {
    const int c4n = C >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * OH * OW * c4n) return;
    const int c4 = (int)(i % c4n);
    const int ow = (int)((i / c4n) % OW), oh = (int)((i / ((int64_t)c4n * OW)) % OH), n = (int)(i / ((int64_t)c4n * OW * OH));
    const float4 sc = *reinterpret_cast<const float4*>(scale + c4 * 4), sh = *reinterpret_cast<const float4*>(shift + c4 * 4);
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);  // ReLU output is >= 0 and every window holds at least one real pixel
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int y = 2 * oh + dy;
        if (y < 0 || y >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = 2 * ow + dx;
            if (xx < 0 || xx >= W) continue;
            const float4 v = *reinterpret_cast<const float4*>(x + (((int64_t)n * H + y) * W + xx) * C + c4 * 4);
            m.x = fmaxf(m.x, v.x * sc.x + sh.x);
            m.y = fmaxf(m.y, v.y * sc.y + sh.y);
            m.z = fmaxf(m.z, v.z * sc.z + sh.z);
            m.w = fmaxf(m.w, v.w * sc.w + sh.w);
        }
    }
    *reinterpret_cast<float4*>(out + i * 4) = m;
}

extern "C" int ndet_bn_relu_maxpool_nhwc(const float* x, const float* scale, const float* shift, int N, int H, int W, int C, float* out,
                                         void* stream) {
    const char* fn = "ndet_bn_relu_maxpool_nhwc";
    NDET_REQUIRE(x && scale && shift && out, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE(C % 4 == 0 && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0, NDET_E_UNSUPPORTED,
                 "%s: C must be a multiple of 4 and pointers 16-byte aligned", fn);
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * OH * OW * (C / 4);
    hipLaunchKernelGGL(k_bn_relu_maxpool, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, N, H, W, C,
                       OH, OW, out);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}
