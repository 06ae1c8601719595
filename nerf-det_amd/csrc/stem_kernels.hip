// ResNet stem in one launch: 7x7 stride-2 convolution (3 -> 64 channels) + BatchNorm(eval) + ReLU + MaxPool(3, 2, 1) -- the first four
// modules of the mmdet ResNet the detector runs as `self.backbone(img)` (mmdet3d/models/detectors/nerfdet.py:140; third-party, restated).
// The library path spent 425 us per 50-view scene here: a layout conversion of the images, 86 us of host-side dispatch with an empty
// queue, a 32 us tensor op, the fp32 implicit-GEMM convolution (209 us) writing 245 MB that a second kernel (74 us) read back to pool.
//
// Same arithmetic as conv_split_kernels.hip: fp32 operands as exact sums of three bf16 terms, six MFMA products, fp32 accumulate.
// A persistent workgroup owns tiles of 3 x 8 POOLED pixels = 7 x 17 convolution pixels (119 of the 128 GEMM rows) = a 19 x 39 x 3 input
// patch, split once into LDS planes.  K is laid out as (ky, c, kx) with kx padded 7 -> 8 (176 = 11 slices of 16): an MFMA A fragment --
// 8 consecutive k of one convolution pixel -- is then 8 consecutive input pixels of one patch row, four 4-byte LDS reads.  The weight
// fragments (64 x 176 x 3 planes, 68 KB) live in the waves' registers for the whole launch.  BN + ReLU are applied to the accumulators,
// the 7 x 17 x 64 tile goes through LDS, and the 3 x 8 pooled pixels leave as whole 256-byte channel rows (channels-last output).
#include "conv_common.hpp"      // conv_amax_commit: max |out| for the fp16-pair layer that follows

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define ST_TPY 3
#define ST_TPX 8
#define ST_CY 7            // convolution rows of a tile: 2 TPY + 1
#define ST_CX 17           // convolution columns: 2 TPX + 1
#define ST_IY 19           // input rows: 2 CY + 5
#define ST_IX 39           // input columns: 2 CX + 5
#define ST_PITCH 40        // LDS row pitch of the patch (column 39 stays zero: kx = 7 carries a zero weight)
#define ST_K 176           // (ky 7) x (c 3) x (kx 8) = 168, padded to 11 slices of 16
#define ST_KS 11
#define ST_NE (3 * ST_IY * ST_IX)   // patch elements
#define ST_EPT ((ST_NE + 255) / 256)
#define ST_CLD 68          // pitch of the staged convolution tile (floats)

struct StemParams {
    const float* x;        // images, element strides below (NCHW or channels-last)
    int64_t sn, sc, sy, sx;
    int N, H, W;
    int CH, CW;            // convolution output extent
    int PH, PW;            // pooled output extent
    int tiles_y, tiles_x, tiles;
    const uint16_t* w;     // (3 planes, 64, 176) bf16
    const float* scale;    // (64)
    const float* shift;
    float* out;            // (N, PH, PW, 64)
    float* amax_out;       // max |out| slot (conv_common.hpp) or null
    float winv;            // fp16-pair form: 1 / (the weight planes' power-of-two scale)
};

__device__ __forceinline__ uint32_t st_pack(float x, float y) {
    const bf16x2 v = __builtin_convertvector((f32x2){x, y}, bf16x2);
    return __builtin_bit_cast(uint32_t, v);
}

// SCH 0: three bf16 planes per operand, six products (exact operands).  SCH 1: two fp16 planes, three products (conv_split_kernels.hip's fp16-pair
// scheme): the weights pre-scaled once, the patch by the power of two of ITS OWN maximum (the patch is split, multiplied and discarded inside this
// workgroup: a tile-local scale is legal and needs neither an amax slot of the images nor a range guard); half the matrix work -- the kernel spends
// 5.5 of its 8.2 us per tile multiplying (s_memtime stamps), two workgroups per CU sharing the matrix pipes.
template <int SCH>
__global__ __launch_bounds__(256, 2) void k_stem_conv_pool(const StemParams p) {
    constexpr int NPL = SCH == 1 ? 2 : 3;
    __shared__ __attribute__((aligned(16))) uint16_t P[NPL * 3 * ST_IY * ST_PITCH];   // [plane][c][iy][pitch]
    __shared__ __attribute__((aligned(16))) float Cs[128 * ST_CLD];
    __shared__ float s_pmax[4];
    constexpr int PPL = 3 * ST_IY * ST_PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    for (int i = tid; i < NPL * PPL; i += 256) P[i] = 0;        // the pad column must be a finite zero
    // ---- weight fragments of this wave's 32 output channels: registers for the whole launch ----
    const int frow = lane & 31, fh = lane >> 5;
    bf16x8 fb[NPL][ST_KS];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int ks = 0; ks < ST_KS; ++ks)
            fb[pl][ks] = *reinterpret_cast<const bf16x8*>(p.w + ((int64_t)(pl * 64 + wn * 32 + frow) * ST_K + ks * 16 + fh * 8));
    const float sc = p.scale[wn * 32 + frow], sh = p.shift[wn * 32 + frow];

    // ---- this thread's patch elements (fixed across tiles) ----
    int e_lds[ST_EPT];            // LDS offset (c, iy, ix decode from it); -1: no element
#pragma unroll
    for (int i = 0; i < ST_EPT; ++i) {
        const int e = tid + 256 * i;
        const int c = e / (ST_IY * ST_IX), r = e - c * (ST_IY * ST_IX), iy = r / ST_IX, ix = r - iy * ST_IX;
        e_lds[i] = e < ST_NE ? (c * ST_IY + iy) * ST_PITCH + ix : -1;
    }
    // ---- LDS offsets of this lane's two GEMM rows (convolution pixels) ----
    int aoff[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        int r = wm * 64 + rt * 32 + frow;
        if (r >= ST_CY * ST_CX) r = 0;                               // rows past the tile multiply a real pixel and are never stored
        const int ly = r / ST_CX, lx = r - ly * ST_CX;
        aoff[rt] = (2 * ly) * ST_PITCH + 2 * lx;
    }
    float ra[ST_EPT];
    auto load_patch = [&](int t) {
        const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, n = t / (p.tiles_x * p.tiles_y);
        const int iy0 = 2 * (2 * ty * ST_TPY - 1) - 3, ix0 = 2 * (2 * tx * ST_TPX - 1) - 3;
        const float* base = p.x + (int64_t)n * p.sn;
#pragma unroll
        for (int i = 0; i < ST_EPT; ++i) {
            const int off = e_lds[i] < 0 ? 0 : e_lds[i];
            const int c = off / (ST_IY * ST_PITCH), r = off - c * (ST_IY * ST_PITCH), iy = r / ST_PITCH, ix = r - iy * ST_PITCH;
            const int y = iy0 + iy, x = ix0 + ix;
            const bool ok = e_lds[i] >= 0 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            ra[i] = ok ? base[c * p.sc + y * p.sy + x * p.sx] : 0.0f;
        }
    };
    float omax = 0.0f;
    __syncthreads();
    int t = blockIdx.x;
    if (t < p.tiles) load_patch(t);
    for (; t < p.tiles; t += gridDim.x) {
        // ---- patch -> operand planes in LDS ----
        float xs = 1.0f, xinv = 1.0f;
        if constexpr (SCH == 1) {                                        // the patch's own maximum -> its power-of-two scale
            float m = 0.0f;
#pragma unroll
            for (int i = 0; i < ST_EPT; ++i) m = fmaxf(m, fabsf(ra[i]));
#pragma unroll
            for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            if (lane == 0) s_pmax[wave] = m;
            __syncthreads();                                             // (also: the previous tile's MFMA reads of P are done -- all waves passed (B))
            m = fmaxf(fmaxf(s_pmax[0], s_pmax[1]), fmaxf(s_pmax[2], s_pmax[3]));
            xs = conv_xscale_of(m);
            xinv = conv_xinv_of(m) * p.winv;
        }
#pragma unroll
        for (int i = 0; i < ST_EPT; ++i)
            if (e_lds[i] >= 0) {
                if constexpr (SCH == 1) {
                    const float v = ra[i] * xs;
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    P[e_lds[i]] = __builtin_bit_cast(uint16_t, h); P[PPL + e_lds[i]] = __builtin_bit_cast(uint16_t, l);
                } else {
                    const float v = ra[i];
                    const uint32_t o0 = st_pack(v, 0.0f);
                    const float r1 = v - __uint_as_float(o0 << 16);
                    const uint32_t o1 = st_pack(r1, 0.0f);
                    const uint32_t o2 = st_pack(r1 - __uint_as_float(o1 << 16), 0.0f);
                    P[e_lds[i]] = (uint16_t)o0; P[PPL + e_lds[i]] = (uint16_t)o1; P[2 * PPL + e_lds[i]] = (uint16_t)o2;
                }
            }
        __syncthreads();                                             // (A) patch complete; the previous tile's pooling has read Cs
        const int tn = t + gridDim.x;
        if (tn < p.tiles) load_patch(tn);                            // the next patch travels during the multiplication

        // ---- one 32-row tile at a time (the weight fragments are shared, the accumulator and operand registers are not doubled):
        // multiply, BN + ReLU, stage the 7 x 17 x 64 tile.  C/D layout of the 32x32 MFMA: col = lane & 31,
        // row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma nounroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int ao = aoff[rt];
#pragma unroll
            for (int ks = 0; ks < ST_KS; ++ks) {
                int g = 2 * ks + fh;                                     // (ky, c) group of this lane's 8 k
                if (g > 20) g = 20;                                      // padding slice: zero weights, any finite operand
                const int ky = g / 3, c = g - 3 * ky;
                const int goff = (c * ST_IY + ky) * ST_PITCH;
                bf16x8 fa[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const uint32_t* q = reinterpret_cast<const uint32_t*>(P + pl * PPL + goff + ao);   // 4-byte aligned: the offsets are even
                    const u32x4 v = {q[0], q[1], q[2], q[3]};
                    fa[pl] = __builtin_bit_cast(bf16x8, v);
                }
                if constexpr (SCH == 1) {                                // smallest terms first: hi lo, lo hi, hi hi
                    typedef _Float16 st_f16x8 __attribute__((ext_vector_type(8)));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(st_f16x8, fa[0]), __builtin_bit_cast(st_f16x8, fb[1][ks]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(st_f16x8, fa[1]), __builtin_bit_cast(st_f16x8, fb[0][ks]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(st_f16x8, fa[0]), __builtin_bit_cast(st_f16x8, fb[0][ks]), acc, 0, 0, 0);
                } else {
#pragma unroll
                    for (int order = 2; order >= 0; --order)
#pragma unroll
                        for (int pa = 0; pa <= order; ++pa) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa], fb[order - pa][ks], acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                Cs[row * ST_CLD + wn * 32 + frow] = fmaxf((SCH == 1 ? acc[r] * xinv : acc[r]) * sc + sh, 0.0f);
            }
        }
        __syncthreads();                                             // (B)
        // ---- 3 x 3 / stride 2 maximum over the tile, whole channel rows out ----
        {
            const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, n = t / (p.tiles_x * p.tiles_y);
            const int cy0 = 2 * ty * ST_TPY - 1, cx0 = 2 * tx * ST_TPX - 1;
            for (int w = tid; w < ST_TPY * ST_TPX * 16; w += 256) {
                const int q = w & 15, pp = w >> 4, ppy = pp / ST_TPX, ppx = pp - ppy * ST_TPX;
                const int py = ty * ST_TPY + ppy, px = tx * ST_TPX + ppx;
                if (py >= p.PH || px >= p.PW) continue;
                float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int ly = 2 * ppy + dy, lx = 2 * ppx + dx;
                        if ((unsigned)(cy0 + ly) < (unsigned)p.CH && (unsigned)(cx0 + lx) < (unsigned)p.CW) {
                            const float4 v = *reinterpret_cast<const float4*>(Cs + (ly * ST_CX + lx) * ST_CLD + q * 4);
                            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                        }
                    }
                *reinterpret_cast<float4*>(p.out + (((int64_t)n * p.PH + py) * p.PW + px) * 64 + q * 4) = m;
                omax = fmaxf(fmaxf(omax, fmaxf(m.x, m.y)), fmaxf(m.z, m.w));       // (post-ReLU: non-negative)
            }
        }
        // the next iteration's patch stores touch P only; its barrier (A) orders them against this tile's MFMA reads (all waves passed (B))
        // and orders the next Cs writes against this pooling
    }
    if (p.amax_out) conv_amax_commit(p.amax_out, omax);                  // once per persistent workgroup
}

// stem weight (64, 3, 7, 7) fp32 -> (3 planes, 64, 176) bf16 with k = (ky * 3 + c) * 8 + kx; kx = 7 and k >= 168 are zero
__global__ __launch_bounds__(256) void k_stem_pack_weights(const float* __restrict__ w, uint16_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 64 * ST_K) return;
    const int co = i / ST_K, k = i - co * ST_K;
    const int g = k >> 3, kx = k & 7;
    float v = 0.0f;
    if (g < 21 && kx < 7) {
        const int ky = g / 3, c = g - 3 * ky;
        v = w[((co * 3 + c) * 7 + ky) * 7 + kx];
    }
    const uint32_t o0 = st_pack(v, 0.0f);
    const float r1 = v - __uint_as_float(o0 << 16);
    const uint32_t o1 = st_pack(r1, 0.0f);
    const uint32_t o2 = st_pack(r1 - __uint_as_float(o1 << 16), 0.0f);
    out[i] = (uint16_t)o0; out[64 * ST_K + i] = (uint16_t)o1; out[2 * 64 * ST_K + i] = (uint16_t)o2;
}

// the fp16-pair form: (2 planes, 64, 176) fp16 of w * scale (scale: a power of two putting max |w| in [2^14, 2^15), chosen by the caller)
__global__ __launch_bounds__(256) void k_stem_pack_weights_f16x2(const float* __restrict__ w, float scale, uint16_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 64 * ST_K) return;
    const int co = i / ST_K, k = i - co * ST_K;
    const int g = k >> 3, kx = k & 7;
    float v = 0.0f;
    if (g < 21 && kx < 7) {
        const int ky = g / 3, c = g - 3 * ky;
        v = w[((co * 3 + c) * 7 + ky) * 7 + kx] * scale;
    }
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    out[i] = __builtin_bit_cast(uint16_t, h); out[64 * ST_K + i] = __builtin_bit_cast(uint16_t, l);
}

extern "C" int ndet_stem_pack_weights_f16x2(const float* w_64x3x7x7, float scale, uint16_t* planes, void* stream) {
    const char* fn = "ndet_stem_pack_weights_f16x2";
    NDET_REQUIRE(w_64x3x7x7 && planes && scale > 0.0f, NDET_E_INVALID, "%s: null pointer / non-positive scale", fn);
    hipLaunchKernelGGL(k_stem_pack_weights_f16x2, dim3((64 * ST_K + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_64x3x7x7, scale, planes);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

extern "C" int ndet_stem_pack_weights(const float* w_64x3x7x7, uint16_t* planes, void* stream) {
    const char* fn = "ndet_stem_pack_weights";
    NDET_REQUIRE(w_64x3x7x7 && planes, NDET_E_INVALID, "%s: null pointer", fn);
    hipLaunchKernelGGL(k_stem_pack_weights, dim3((64 * ST_K + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_64x3x7x7, planes);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

extern "C" int ndet_stem_conv_bn_relu_maxpool(const float* images, int N, int H, int W, int64_t stride_n, int64_t stride_c, int64_t stride_y,
                                              int64_t stride_x, const uint16_t* w_planes, float w_inv_scale, const float* scale, const float* shift,
                                              float* out, float* out_amax, void* stream) {
    const char* fn = "ndet_stem_conv_bn_relu_maxpool";
    NDET_REQUIRE(images && w_planes && scale && shift && out, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(N > 0 && H >= 7 && W >= 7, NDET_E_INVALID, "%s: bad sizes", fn);
    NDET_REQUIRE((((uintptr_t)w_planes | (uintptr_t)out) & 15) == 0, NDET_E_UNSUPPORTED, "%s: weights / output must be 16-byte aligned", fn);
    StemParams p;
    p.x = images; p.sn = stride_n; p.sc = stride_c; p.sy = stride_y; p.sx = stride_x;
    p.N = N; p.H = H; p.W = W;
    p.CH = (H + 6 - 7) / 2 + 1; p.CW = (W + 6 - 7) / 2 + 1;
    p.PH = (p.CH + 2 - 3) / 2 + 1; p.PW = (p.CW + 2 - 3) / 2 + 1;
    p.tiles_y = (p.PH + ST_TPY - 1) / ST_TPY; p.tiles_x = (p.PW + ST_TPX - 1) / ST_TPX;
    const int64_t tiles = (int64_t)N * p.tiles_y * p.tiles_x;
    NDET_REQUIRE(tiles < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: too many tiles", fn);
    p.tiles = (int)tiles;
    NDET_REQUIRE(w_inv_scale >= 0.0f, NDET_E_INVALID, "%s: w_inv_scale is 0 (bf16x3 planes) or the fp16-pair planes' inverse scale", fn);
    p.w = w_planes; p.scale = scale; p.shift = shift; p.out = out; p.amax_out = out_amax; p.winv = w_inv_scale;
    const int grid = (int)(tiles < 512 ? tiles : 512);               // two persistent workgroups per CU
    if (w_inv_scale > 0.0f) hipLaunchKernelGGL(k_stem_conv_pool<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(k_stem_conv_pool<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}
