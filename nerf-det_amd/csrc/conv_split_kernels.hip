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
// Layout: weights arrive pre-split as three bf16 planes, tiled per K step: (taps, Cin/32, 3, Cout, 32) -- the B tile of one
// K step is one contiguous run per plane, so every staging load instruction covers 1 KB of whole cache lines
// (ndet_split_weights_bf16x3 below, once per model);
// activations stay fp32 channels-last in HBM and are split while they are staged into LDS (three bf16 planes per operand).
//
// Three kernel families, chosen per layer (nerfdet_amd/conv_tuning.py, measured):
//   k_conv_split<BM,BN>     every wave stages and multiplies (32x32x16 MFMAs, LDS rows padded to 80 B, one stage, next K step
//                           prefetched in registers, two workgroups per CU cover each other) -- small / narrow layers;
//   k_conv_split_ws         128 x 256, wave-specialised: 4 MFMA-only consumer waves (16x16x32), 4 producer waves (activation
//                           split in registers, weight tiles by LDS-DMA), swizzled unpadded LDS rows, one barrier per step;
//   k_conv_split_halo<..>   stride-1 same-padded multi-tap layers: the activation patch + halo is split once per channel
//                           chunk and every tap multiplies out of that LDS image.
#include "conv_common.hpp"

#include "spl_common.hpp"

// set per call by the weight-gradient entry points (ndet_conv_ndhwc_train, ndet_wgrad_split*): the split-K partials stay in the workspace and
// ndet_wgrad_to_torch adds them up (same fixed order) on its way to torch's layout -- one launch and one pass over dW fewer per layer
static thread_local int g_keep_partials = 0;

// K walk of the unified tiles: fills acc (per-wave 32x32 MFMA tiles) for GEMM rows m0.. and channels n0..; returns the transposed-conv tap
// (blockIdx.z) in ztap.  Ends behind a barrier: the LDS operand planes are free for the epilogue.
template <int BM, int BN, int WGM, int WGN, int SCH>
__device__ __forceinline__ void conv_split_mainloop(const Conv3dParams& p, const uint16_t* __restrict__ wsplit, uint16_t* lds16,
                                                    f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32], int& ztap) {
    constexpr int NTHR = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;   // per-wave tile
    constexpr int MT = WM / 32, NT = WN / 32;     // 32x32 MFMA tiles per wave
    constexpr int RPA = NTHR / 8;                 // A rows staged per pass: 8 threads x 4 fp32 cover one 128-byte row
    constexpr int RPB = NTHR / 4;                 // B rows staged per pass: 4 threads x 8 bf16 cover one 64-byte row
    constexpr int AR = BM / RPA, BR = BN / RPB;
    static_assert(AR >= 1 && BR >= 1 && MT >= 1 && NT >= 1, "tile too small for the thread count");
    constexpr int APL = BM * SPL_RS, BPL = BN * SPL_RS;  // one plane, in bf16 elements
    constexpr int NPL = Spl<SCH>::NPL, WPL = Spl<SCH>::WPL;
    const float xs = SCH == 1 ? conv_xscale(p.amax_in) : 1.0f;
    uint16_t* As = lds16;               // [NPL][BM][SPL_RS]
    uint16_t* Bs = lds16 + NPL * APL;   // [NPL][BN][SPL_RS]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const ConvBlock blk = conv_block(p);
    const int m0 = blk.x * BM, n0 = blk.y * BN;
    const int akq = tid & 7, arow_ = tid >> 3;   // A: 4-k quad and first row
    const int bkg = tid & 3, brow_ = tid >> 2;   // B: 8-k octet and first row

    const int cin_steps = p.Cin / CBK;
    const int taps = p.transposed ? 1 : p.kd * p.kh * p.kw;
    const int n_iters_all = taps * cin_steps;
    int it_begin = 0, it_end = n_iters_all;
    ztap = 0;
    if (p.transposed) {
        ztap = blk.z;
    } else if (p.splits > 1) {
        const int s = blk.z;
        it_begin = (int)((int64_t)n_iters_all * s / p.splits);
        it_end = (int)((int64_t)n_iters_all * (s + 1) / p.splits);
    }

    int vd[AR], vh[AR], vw[AR];
    bool vok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + arow_ + RPA * i;
        vok[i] = m < p.M;
        const int mm = vok[i] ? m : 0;
        const int ow_ = p.transposed ? p.W : p.OW, oh_ = p.transposed ? p.H : p.OH;
        vw[i] = mm % ow_;
        vh[i] = (mm / ow_) % oh_;
        vd[i] = mm / (ow_ * oh_);
    }

#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[AR];
    uint4 rb[BR][NPL];
    const int64_t wtile = (int64_t)p.Cout * CBK;   // one plane of one K step, in elements
    bool bok[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) bok[i] = n0 + brow_ + RPB * i < p.Cout;
    const uint16_t* bbase_g = wsplit + (int64_t)(n0 + brow_) * CBK + bkg * 8;
    // K walk: channel chunk outermost, taps innermost -- consecutive steps re-read the same input rows shifted by one
    // tap, so the A working set of a workgroup (its rows + halo, 32 channels) stays cache resident across the taps.
    const int wkd = p.transposed ? 1 : p.kd, wkh = p.transposed ? 1 : p.kh, wkw = p.transposed ? 1 : p.kw;
    int bd[AR], bh[AR], bw[AR];
    const float* rowbase[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        bd[i] = p.transposed ? vd[i] : vd[i] * p.sd - p.pd;
        bh[i] = p.transposed ? vh[i] : vh[i] * p.sh - p.ph;
        bw[i] = p.transposed ? vw[i] : vw[i] * p.sw - p.pw;
        rowbase[i] = p.in + (((int64_t)bd[i] * p.H + bh[i]) * p.W + bw[i]) * p.Cin + akq * 4;  // dereferenced only when in range
    }
    // state of the next tile to load
    int nkd, nkh, nkw, ncs;
    {
        const int t0 = it_begin % taps;
        ncs = it_begin / taps;
        nkd = t0 / (wkh * wkw); nkh = (t0 / wkw) % wkh; nkw = t0 % wkw;
    }
    auto load_tile = [&]() {
        const int tap = p.transposed ? ztap : (nkd * wkh + nkh) * wkw + nkw;
        const int64_t toff = (((int64_t)nkd * p.H + nkh) * p.W + nkw) * p.Cin + ncs * CBK;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const bool ok = vok[i] && (unsigned)(bd[i] + nkd) < (unsigned)p.D && (unsigned)(bh[i] + nkh) < (unsigned)p.H &&
                            (unsigned)(bw[i] + nkw) < (unsigned)p.W;
            const float4 v = *reinterpret_cast<const float4*>(ok ? rowbase[i] + toff : p.in + akq * 4);   // always issued
            ra[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const uint16_t* bt = bbase_g + ((int64_t)tap * cin_steps + ncs) * WPL * wtile;
#pragma unroll
        for (int i = 0; i < BR; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                const uint4 v = *reinterpret_cast<const uint4*>(bok[i] ? bt + pl * wtile + (int64_t)RPB * i * CBK : wsplit + bkg * 8);
                rb[i][pl] = bok[i] ? v : make_uint4(0u, 0u, 0u, 0u);
            }
        if (++nkw == wkw) {
            nkw = 0;
            if (++nkh == wkh) {
                nkh = 0;
                if (++nkd == wkd) { nkd = 0; ++ncs; }
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            uint2 s0, s1, s2;
            spl_split<SCH>(ra[i], xs, s0, s1, s2);
            uint16_t* dst = As + (arow_ + RPA * i) * SPL_RS + akq * 4;
            *reinterpret_cast<uint2*>(dst) = s0;
            if (NPL > 1) *reinterpret_cast<uint2*>(dst + APL) = s1;
            if (NPL > 2) *reinterpret_cast<uint2*>(dst + 2 * APL) = s2;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            uint16_t* dst = Bs + (brow_ + RPB * i) * SPL_RS + bkg * 8;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint4*>(dst + pl * BPL) = rb[i][pl];
        }
    };

    if (it_begin < it_end) {
        load_tile();
        store_tile();
    }
    __syncthreads();

    // fragment of the 32x32x16 MFMA: lane l holds row (l & 31), k = 8 (l >> 5) + j, j = 0..7 -> one 16-byte read
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    const uint16_t* abase = As + (wm * WM + frow) * SPL_RS + fk;
    const uint16_t* bbase = Bs + (wn * WN + frow) * SPL_RS + fk;
    for (int it = it_begin; it < it_end; ++it) {
        const bool more = it + 1 < it_end;
        if (more) load_tile();
#pragma unroll
        for (int ks = 0; ks < CBK / 16; ++ks) {
            bf16x8 fa[NPL][MT], fb[NPL][NT];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {      // in the order the products below need them: (A plane k, B plane NPL-1-k) first -- the first MFMA
                const int pq = NPL - 1 - pl;        // then waits for 2 of the 2 NPL plane reads instead of all of them (LDS returns in order)
#pragma unroll
                for (int t = 0; t < MT; ++t) fa[pl][t] = *reinterpret_cast<const bf16x8*>(abase + pl * APL + t * 32 * SPL_RS + ks * 16);
#pragma unroll
                for (int t = 0; t < NT; ++t) fb[pq][t] = *reinterpret_cast<const bf16x8*>(bbase + pq * BPL + t * 32 * SPL_RS + ks * 16);
            }
            // smallest terms first; the (pa, pb) pairs with pa + pb <= 2
#pragma unroll
            for (int order = NPL - 1; order >= 0; --order)
                if (order <= p.max_order)
#pragma unroll
                for (int pa = 0; pa <= order; ++pa) {
                    const int pb = order - pa;
#pragma unroll
                    for (int ta = 0; ta < MT; ++ta)
#pragma unroll
                        for (int tb = 0; tb < NT; ++tb)
                            acc[ta][tb] = spl_mfma32<SCH>(fa[pa][ta], fb[pb][tb], acc[ta][tb]);
                }
        }
        __syncthreads();   // every wave is done reading this K step
        if (more) store_tile();
        __syncthreads();
    }
}

#ifndef UNI128_F16_WGS
#define UNI128_F16_WGS 3     // fp16 pair, 128 x 128: 40 KB of LDS per workgroup would admit four per CU (1 024 slots for the 944 tiles of the stage-3
                             // conv3 layers, which now take two rounds), but at 128 registers the kernel spills 316 bytes per lane (162 needed): three
#endif
template <int BM, int BN, int SCH> constexpr int uni_min_wgs() { return (SCH == 1 && BM == 128 && BN == 128) ? UNI128_F16_WGS : 2; }
template <int BM, int BN, int WGM, int WGN, int SCH = 0>
__global__ __launch_bounds__(64 * WGM * WGN, (uni_min_wgs<BM, BN, SCH>())) void k_conv_split(const Conv3dParams p, const uint16_t* __restrict__ wsplit) {
    constexpr int NTHR = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, MT = WM / 32, NT = WN / 32;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds16[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const ConvBlock blk = conv_block(p);
    const int m0 = blk.x * BM, n0 = blk.y * BN;
    f32x16 acc[MT][NT];
    int ztap;
    const float amax_in = conv_amax_in(p);
    conv_guard_check(p, amax_in);
    conv_split_mainloop<BM, BN, WGM, WGN, SCH>(p, wsplit, lds16, acc, ztap);
    const float osc = conv_oscale_of(p, amax_in);

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    if (p.direct) {
        float mx = 0.0f;
        // ---- direct epilogue: a register of a 32 x 32 tile is 32 consecutive channels of one row = one 128-byte line; residual reads and
        // stores are buffer operations (lane part of the address in one VGPR, the register's row in the scalar offset, rows past M outside the
        // descriptor).  No LDS staging, no barrier: the workgroup's waves drain independently while the CU's other workgroups multiply ----
        // Upsampled residual (FPN lateral: the coarser map read at (h >> 1, w >> 1)): the rows of a register set are not evenly spaced in the
        // residual, so (w, h, n) of the lane's first row is divided out once per 32 x 32 tile and stepped along the registers' rows (+1 +1 +1 +5).
        const unsigned obytes = (unsigned)((int64_t)p.M * p.Cout * 4);
        const unsigned rbytes = p.res_up2 ? (unsigned)((int64_t)p.OD * p.RH * p.RW * p.Cout * 4) : obytes;
        const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.out), 0, p.res ? rbytes : obytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, obytes, 0x00020000);
#pragma unroll
        for (int tb = 0; tb < NT; ++tb) {
            const int co = n0 + wn * WN + tb * 32 + (lane & 31);
            if (n0 + wn * WN + tb * 32 >= p.Cout) continue;            // Cout % 32 == 0: a 32-column tile is all inside or all outside
            const float sc = p.scale ? p.scale[co] : 1.0f, sh = p.scale ? p.shift[co] : 0.0f;
#pragma unroll
            for (int ta = 0; ta < MT; ++ta) {
                const int mrow = m0 + wm * WM + ta * 32 + 4 * (lane >> 5);
                const unsigned vo = (unsigned)(((int64_t)mrow * p.Cout + co) * 4);
                float rr[16];
                if (p.res && p.res_up2) {
                    int uw = mrow % p.OW, uh = (mrow / p.OW) % p.OH, ud = mrow / (p.OW * p.OH);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (r) {
                            uw += (r & 3) ? 1 : 5;
                            if (p.OW >= 5) { if (uw >= p.OW) { uw -= p.OW; if (++uh == p.OH) { uh = 0; ++ud; } } }     // one wrap at most
                            else while (uw >= p.OW) { uw -= p.OW; if (++uh == p.OH) { uh = 0; ++ud; } }
                        }
                        const unsigned ro = ((((unsigned)ud * p.RH + (uh >> 1)) * p.RW + (uw >> 1)) * p.Cout + co) * 4u;   // < 4 GB (launcher)
                        // rows past M: an offset outside the descriptor (the load returns 0, the store below is dropped the same way)
                        rr[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rres, (mrow + (r & 3) + 8 * (r >> 2) < p.M) ? ro : 0xfffffff0u, 0, 0));
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(((r & 3) + 8 * (r >> 2)) * p.Cout * 4));
                        rr[r] = p.res ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rres, vo, so, 0)) : 0.0f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(((r & 3) + 8 * (r >> 2)) * p.Cout * 4));
                    float v = (acc[ta][tb][r] * osc) * sc + sh;
                    if (p.relu == 2) v = fmaxf(v, 0.f);
                    v += rr[r];
                    if (p.relu == 1) v = fmaxf(v, 0.f);
                    if (p.nt) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rout, vo, so, 2);
                    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rout, vo, so, 0);
                    if (mrow + (r & 3) + 8 * (r >> 2) < p.M) mx = fmaxf(mx, fabsf(v));      // rows past M: dropped stores of values that are not the layer's
                }
            }
        }
        if (p.amax_out) conv_amax_commit(p.amax_out, mx);
        return;
    }
    // ---- epilogue: accumulators -> LDS (one wave-row of the tile at a time) -> fused row-wise stores ----
    constexpr int CLDC = BN + 4;
    float* Cs = reinterpret_cast<float*>(lds16);   // [WM][CLDC] floats <= the operand planes
    float mx = 0.0f;
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
        conv_store_rows<BN, NTHR>(p, Cs, CLDC, m0 + h * WM, WM, n0, tid, ztap, blk.z, mx, osc);
        __syncthreads();
    }
    if (p.amax_out && conv_writes_final(p)) conv_amax_commit(p.amax_out, mx);
}


// ------------------------------------------------------------------------------------------------
// Convolution + chained 1x1 convolution in one launch: out = act3(bn3(W3 . relu(bn1(conv(x)))) + residual) -- conv2 -> conv3 of a
// ResNet bottleneck (mmdet's Bottleneck.forward behind nerfdet.py:140; the 64- / 128-channel intermediate of stages 1 / 2 is 61 MB at
// cfg2 and would be written and read back once per block).  The first convolution runs as above on a 128 x MID tile holding ALL
// of its output channels; its accumulators go through BN + ReLU and the bf16 split straight into LDS in the A-operand layout
// (three planes, 128 rows x MID), and the same four waves multiply that image by W3: every wave owns 32 CTW-column slices of the
// output, keeps the slice's W3 fragments in registers (read from L2 once per workgroup, never staged) and walks the four 32-row
// tiles.  The epilogue works in the MFMA's own C layout: a register of the 32 x 32 tile is 32 consecutive channels of one row = one
// whole 128-byte line for the residual read and the store.
// ------------------------------------------------------------------------------------------------
struct ConvChain {
    const uint16_t* w3;    // (1, MID/32, 3, Cout3, 32) bf16 planes  [fp16 pair: (1, MID/32, 2, Cout3, 32), pre-scaled by 1 / w3inv]
    float w3inv;           // fp16 pair: 1 / (scale of w3)
    float* amax_out;       // optional: max |out| (see Conv3dParams::amax_out)
    const float* scale3;   // (Cout3) or null
    const float* shift3;
    const float* res;      // (M, Cout3) or null
    float* out;            // (M, Cout3)
    int Cout3;
    int relu3;             // 0 none, 1 ReLU last, 2 ReLU before the residual add
    float guard_l1 = 0.0f; // range guard (Conv3dParams::guard_l1) of W3
};

#ifndef CHAIN64_F16_WGS
#define CHAIN64_F16_WGS 4     // fp16 pair: 30 KB of LDS per workgroup; four per CU (128 registers, 16 bytes of scratch) stream 7 % faster than three (HBM-bound layer)
#endif
template <int MID, int SCH>
__global__ __launch_bounds__(256, MID == 64 ? (SCH == 1 ? CHAIN64_F16_WGS : 3) : 2) void k_conv_split_chain(const Conv3dParams p, const uint16_t* __restrict__ wsplit, const ConvChain c) {
    constexpr int BM = 128, BN = MID, WGM = 2, WGN = 2;
    constexpr int WM = BM / WGM, WN = BN / WGN, MT = WM / 32, NT = WN / 32;
    constexpr int NPL = Spl<SCH>::NPL, WPL = Spl<SCH>::WPL;
    constexpr int YRS = MID;              // LDS rows of the intermediate are unpadded; the 16-byte chunk index is XORed with row & 7 (fragment reads
    constexpr int HALVES = MID == 128 ? 2 : 1;   // 128 channels: the intermediate is chained 64 rows at a time (48 KB of LDS either way)
    constexpr int RH = BM / HALVES;
    constexpr int YPL = RH * YRS;         // of 32 consecutive rows at one chunk then cover all banks)
    constexpr int NRT = RH / 32;
    constexpr int KS = MID / 16;          // 16-wide K slices of the chained GEMM
    constexpr int KG = 4, NKG = KS / KG;  // K slices whose W3 fragments are register-resident at a time (KG x NPL x 4 VGPRs)
    extern __shared__ __attribute__((aligned(16))) uint16_t lds16[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM;
    f32x16 acc[MT][NT];
    int ztap;
    const float amax_in = conv_amax_in(p);
    conv_guard_check(p, amax_in);
    conv_split_mainloop<BM, BN, WGM, WGN, SCH>(p, wsplit, lds16, acc, ztap);
    const float osc = conv_oscale_of(p, amax_in);

    // ---- intermediate: BN + ReLU in place.  fp16 pair: its scale comes from the workgroup's own maximum (the tile is multiplied by W3 here and
    // nowhere else, so the scale only has to be the same for the rows and K slices of this workgroup's chained GEMM) ----
    float ys = 1.0f, osc3 = 1.0f;
    if constexpr (SCH == 1) {
        const float osc1 = conv_oscale(p);
        float ymax = 0.0f;
#pragma unroll
        for (int ta = 0; ta < MT; ++ta)
#pragma unroll
            for (int tb = 0; tb < NT; ++tb) {
                const int col = wn * WN + tb * 32 + (lane & 31);
                const float sc = p.scale ? p.scale[col] : 1.0f, sh = p.scale ? p.shift[col] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float a = (acc[ta][tb][r] * osc1) * sc + sh;
                    if (p.relu) a = fmaxf(a, 0.f);
                    acc[ta][tb][r] = a;
                    ymax = fmaxf(ymax, fabsf(a));
                }
            }
        __shared__ float wg_max[4];
#pragma unroll
        for (int o = 32; o; o >>= 1) ymax = fmaxf(ymax, __shfl_xor(ymax, o));
        if (lane == 0) wg_max[wave] = ymax;
        __syncthreads();
        ymax = fmaxf(fmaxf(wg_max[0], wg_max[1]), fmaxf(wg_max[2], wg_max[3]));
        ys = conv_xscale_of(ymax);
        osc3 = conv_xinv_of(ymax) * c.w3inv;
        // (no range guard for the chained GEMM: its operand scale is this workgroup's own maximum -- already the granularity at which the guard
        // looks for parts of a tensor below the fp16-pair window, conv_guard_check)
    }
    float omax = 0.0f;
    // ---- the intermediate's split, into LDS as the A operand of the chained GEMM ----
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    uint16_t* Y = lds16;
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    const int64_t w3pl = (int64_t)c.Cout3 * CBK;   // one plane of one 32-channel chunk, elements
    const unsigned obytes = (unsigned)((int64_t)p.M * c.Cout3 * 4);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(c.res ? c.res : c.out), 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)c.out, 0, obytes, 0x00020000);
    for (int h = 0; h < HALVES; ++h) {
        if (HALVES == 1 || wm == h) {
#pragma unroll
            for (int ta = 0; ta < MT; ++ta)
#pragma unroll
                for (int tb = 0; tb < NT; ++tb) {
                    const int col = wn * WN + tb * 32 + (lane & 31);
                    const float sc = (SCH != 1 && p.scale) ? p.scale[col] : 1.0f, sh = (SCH != 1 && p.scale) ? p.shift[col] : 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        float a = acc[ta][tb][r], b = acc[ta][tb][r + 1];   // rows r and r + 1 of this lane's column
                        if (SCH != 1) {                                      // (fp16 pair: BN + ReLU were applied in place above)
                            a = a * sc + sh; b = b * sc + sh;
                            if (p.relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
                        }
                        const int row = (HALVES == 1 ? wm * WM : 0) + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        uint16_t* da = Y + row * YRS + ((((col >> 3) ^ (row & 7)) << 3) | (col & 7));
                        uint16_t* db = Y + (row + 1) * YRS + ((((col >> 3) ^ ((row + 1) & 7)) << 3) | (col & 7));
                        uint32_t o0, o1 = 0, o2 = 0;
                        spl_split2<SCH>(a, b, ys, o0, o1, o2);
                        da[0] = (uint16_t)o0; db[0] = (uint16_t)(o0 >> 16);
                        if (NPL > 1) { da[YPL] = (uint16_t)o1; db[YPL] = (uint16_t)(o1 >> 16); }
                        if (NPL > 2) { da[2 * YPL] = (uint16_t)o2; db[2 * YPL] = (uint16_t)(o2 >> 16); }
                    }
                }
        }
        __syncthreads();

        // ---- chained GEMM: (RH x MID) . W3 (MID x Cout3), one 32-column tile per wave and pass; the W3 fragments of KG K-slices are
        // register-resident (all of them at MID = 64; at MID = 128 in two groups, with the pass's NRT accumulator tiles kept) ----
        for (int cb = wave * 32; cb < c.Cout3; cb += 4 * 32) {
            const int co = cb + frow;
            const float sc3 = c.scale3 ? c.scale3[co] : 1.0f, sh3 = c.scale3 ? c.shift3[co] : 0.0f;
            constexpr int NC2 = NKG == 1 ? 1 : NRT;
            f32x16 c2[NC2];
#pragma unroll
            for (int rt = 0; rt < NC2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) c2[rt][r] = 0.f;
            auto load_b = [&](bf16x8 (&fb)[NPL][KG], int kg) {
#pragma unroll
                for (int k4 = 0; k4 < KG; ++k4)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                        const int ks = kg * KG + k4;
                        fb[pl][k4] = *reinterpret_cast<const bf16x8*>(c.w3 + ((int64_t)(ks >> 1) * WPL + pl) * w3pl + (int64_t)co * CBK + (ks & 1) * 16 + fk);
                    }
            };
            auto mul = [&](f32x16& cc, const bf16x8 (&fb)[NPL][KG], int rt, int kg) {
                const uint16_t* ya = Y + (rt * 32 + frow) * YRS;
#pragma unroll
                for (int k4 = 0; k4 < KG; ++k4) {
                    bf16x8 fa[NPL];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        fa[pl] = *reinterpret_cast<const bf16x8*>(ya + pl * YPL + (((2 * (kg * KG + k4) + (lane >> 5)) ^ (frow & 7)) << 3));
#pragma unroll
                    for (int order = NPL - 1; order >= 0; --order)
                        if (order <= p.max_order)
#pragma unroll
                            for (int pa = 0; pa <= order; ++pa) cc = spl_mfma32<SCH>(fa[pa], fb[order - pa][k4], cc);
                }
            };
            // residual reads and stores as buffer operations: the lane's part of the address is one VGPR for the whole pass, the row of
            // register r rides in the scalar offset, rows past M fall outside the descriptor (reads return 0, stores are dropped)
            const unsigned vo = (unsigned)(((int64_t)(m0 + h * RH + 4 * (lane >> 5)) * c.Cout3 + co) * 4);
            auto load_res = [&](float (&rr)[16], int rt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((rt * 32 + (r & 3) + 8 * (r >> 2)) * c.Cout3 * 4));
                    rr[r] = c.res ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rres, vo, so, 0)) : 0.0f;
                }
            };
            auto finish = [&](const f32x16& cc, const float (&rr)[16], int rt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((rt * 32 + (r & 3) + 8 * (r >> 2)) * c.Cout3 * 4));
                    float v = (cc[r] * osc3) * sc3 + sh3;
                    if (c.relu3 == 2) v = fmaxf(v, 0.f);
                    v += rr[r];
                    if (c.relu3 == 1) v = fmaxf(v, 0.f);
                    if (p.nt) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rout, vo, so, 2);
                    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rout, vo, so, 0);
                    if (m0 + h * RH + 4 * (lane >> 5) + rt * 32 + (r & 3) + 8 * (r >> 2) < p.M) omax = fmaxf(omax, fabsf(v));
                }
            };
            if constexpr (NKG == 1) {
                bf16x8 fb[NPL][KG];
                load_b(fb, 0);
                float rn[16];                         // the residual of the NEXT row tile travels while this one is multiplied and stored
                load_res(rn, 0);
#pragma nounroll
                for (int rt = 0; rt < NRT; ++rt) {
                    mul(c2[0], fb, rt, 0);
                    float rr[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) rr[r] = rn[r];
                    if (rt + 1 < NRT) load_res(rn, rt + 1);
                    finish(c2[0], rr, rt);
#pragma unroll
                    for (int r = 0; r < 16; ++r) c2[0][r] = 0.f;
                }
            } else {
#pragma nounroll
                for (int kg = 0; kg < NKG; ++kg) {
                    bf16x8 fb[NPL][KG];
                    load_b(fb, kg);
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt) mul(c2[rt], fb, rt, kg);
                }
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) {
                    float rr[16];
                    load_res(rr, rt);
                    finish(c2[rt], rr, rt);
                }
            }
        }
        if (h + 1 < HALVES) __syncthreads();
    }
    if (c.amax_out) conv_amax_commit(c.amax_out, omax);
}


// ------------------------------------------------------------------------------------------------
// Weight gradient as an implicit GEMM, the tap copies of x never materialised:
//     dW[(t, ci)][co] = sum_j x[s o(j) + t - p][ci] . dy[j][co]             (j = output voxel; autograd of nn.Conv3d / nn.Conv2d,
// mmdet3d/models/necks/imvoxelnet.py:22-67,233-260).  GEMM rows = (tap, input channel), columns = output channels, contraction over the
// output voxels.  The B operand is dy staged channel-major and split once (k_wgrad_rows + k_split_weights, 1x its size); the A operand is
// read straight from channels-last x: a K step is 32 output voxels x BM channels of ONE tap -- 32 coalesced BM x 4-byte runs -- and
// is transposed on its way into LDS (a thread holds four channels of one voxel and writes four 2-byte elements per plane).  With
// k_wgrad_rows staging x as well, a 3x3x3 layer wrote and re-read 27 copies of its input (707 MB at 40x40x16x256).
// Split-K over the voxels with the fixed-order reduction of the convolutions.
// ------------------------------------------------------------------------------------------------
struct WgradParams {
    const float* x;        // (D, H, W, Cin) channels-last
    int D, H, W, Cin;
    int kh, kw;            // taps per axis below the depth axis (tap t -> (t / (kh kw), (t / kw) % kh, t % kw))
    int pd, ph, pw, sd, sh, sw;
    int OD, OH, OW;
    int L;                 // output voxels = OD OH OW
    int ksteps;            // K steps of 32 voxels (dy rows are zero-filled past L)
    const float* x_amax = nullptr;    // fp16-pair arithmetic (SCH 1): the amax slots of x (split here, scaled by conv_xscale of its slot) and of dy (whose
    const float* dy_amax = nullptr;   // planes ndet_wgrad_dy_planes_f16x2 scaled the same way); the epilogue multiplies by the inverse of both
    int xcd_order = 0;                // 1 (splits % 8 == 0): the tiles of one K split run on one XCD (see the kernel)
};

template <int BM, int BN, int SCH>
__global__ __launch_bounds__(256, 2) void k_wgrad_split(const WgradParams g, const Conv3dParams p, const uint16_t* __restrict__ gsplit) {
    constexpr int NTHR = 256, WGM = 2, WGN = 2;
    constexpr int WM = BM / WGM, WN = BN / WGN, MT = WM / 32, NT = WN / 32;
    constexpr int CQ = BM / 4;              // channel quads of the A tile
    constexpr int VG = NTHR / CQ;           // voxel groups staged side by side (8 at BM = 128, 16 at BM = 64)
    constexpr int VT = 32 / VG;             // consecutive voxels per thread (4 / 2): a thread transposes a VT x 4 block in registers
    constexpr int RPB = NTHR / 4, BR = BN / RPB;
    static_assert((VT == 4 || VT == 2) && BR >= 1 && MT >= 1 && NT >= 1, "tile too small for the thread count");
    constexpr int NPL = Spl<SCH>::NPL, WPL = Spl<SCH>::WPL;      // SCH 0: three bf16 planes, 1: two fp16 planes of the pre-scaled operands, 2: one bf16 plane
    constexpr int APL = BM * SPL_RS, BPL = BN * SPL_RS;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds16[];
    uint16_t* As = lds16;
    uint16_t* Bs = lds16 + NPL * APL;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    // Which workgroups meet in one XCD's L2 (workgroup b of the dispatch order runs on XCD b % 8).  In grid order the tiles of one K split are dealt
    // round-robin over all eight XCDs, so every L2 sees every split's slice of x and dy and none can keep it (27-tap 256 -> 256 layer: 2.8 GB pulled per
    // launch, mostly across the fabric).  With the splits a multiple of 8, XCD c takes the splits c, c + 8, ... one after the other, all tiles of a
    // split side by side in time: they walk the same voxels, and the slice they share stays in that L2.  Same sums, same partial layout.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.xcd_order) {
        const int gx = gridDim.x, T = gx * (int)gridDim.y;
        const int b = bx + gx * (by + (int)gridDim.y * bz);
        const int xcd = b & 7, slot = b >> 3;
        bz = xcd + 8 * (slot / T);
        const int tile = slot % T;
        bx = tile % gx; by = tile / gx;
    }
    const int m0 = bx * BM, n0 = by * BN;
    const int tap = m0 / g.Cin, ci0 = m0 - tap * g.Cin;    // Cin % BM == 0: a tile never straddles two taps
    const int ta = tap / (g.kh * g.kw), tb = (tap / g.kw) % g.kh, tc = tap % g.kw;
    const int cq = tid % CQ, vg = tid / CQ;
    const int bkg = tid & 3, brow_ = tid >> 2;

    const float xs = SCH == 1 ? conv_xscale(g.x_amax) : 1.0f;
    const float osc = SCH == 1 ? conv_xinv_of(conv_amax_read(g.x_amax)) * conv_xinv_of(conv_amax_read(g.dy_amax)) : 1.0f;
    int it_begin = 0, it_end = g.ksteps;
    if (p.splits > 1) {
        it_begin = (int)((int64_t)g.ksteps * bz / p.splits);
        it_end = (int)((int64_t)g.ksteps * (bz + 1) / p.splits);
    }
    // output voxel of each of this thread's VT slots (consecutive voxels vg*VT .. +VT-1 of the step), advanced by 32 per step
    int od[VT], oh[VT], ow[VT];
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        const int j = it_begin * 32 + vg * VT + i;
        ow[i] = j % g.OW; oh[i] = (j / g.OW) % g.OH; od[i] = j / (g.OW * g.OH);
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[VT];
    uint4 rb[BR][NPL];
    const int64_t wtile = (int64_t)p.Cout * CBK;
    bool bok[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) bok[i] = n0 + brow_ + RPB * i < p.Cout;
    const uint16_t* bbase_g = gsplit + (int64_t)(n0 + brow_) * CBK + bkg * 8;
    const float* xbase = g.x + ci0 + cq * 4;
    int nit = it_begin;
    auto load_tile = [&]() {
#pragma unroll
        for (int i = 0; i < VT; ++i) {
            const int d_ = od[i] * g.sd + ta - g.pd, h_ = oh[i] * g.sh + tb - g.ph, w_ = ow[i] * g.sw + tc - g.pw;
            const bool ok = od[i] < g.OD && (unsigned)d_ < (unsigned)g.D && (unsigned)h_ < (unsigned)g.H && (unsigned)w_ < (unsigned)g.W;
            const float4 v = *reinterpret_cast<const float4*>(ok ? xbase + (((int64_t)d_ * g.H + h_) * g.W + w_) * g.Cin : xbase);   // always issued
            ra[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            ow[i] += 32;                                   // the slot's voxel of the next step
            while (ow[i] >= g.OW) {
                ow[i] -= g.OW;
                if (++oh[i] == g.OH) { oh[i] = 0; ++od[i]; }
            }
        }
        const uint16_t* bt = bbase_g + (int64_t)nit * WPL * wtile;
#pragma unroll
        for (int i = 0; i < BR; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                const uint4 v = *reinterpret_cast<const uint4*>(bok[i] ? bt + pl * wtile + (int64_t)RPB * i * CBK : gsplit + bkg * 8);
                rb[i][pl] = bok[i] ? v : make_uint4(0u, 0u, 0u, 0u);
            }
        ++nit;
    };
    // A tile in LDS: row = channel, 32 voxels of the step along the row (SPL_RS pitch).  The 8-voxel group index is XORed with
    // (row >> 4) & 3: the lanes of one store (rows 4 apart, 16 banks apart) then land in different 4-dword groups.
    auto store_tile = [&]() {
        const float* rf = reinterpret_cast<const float*>(ra);      // rf[4 * i + e]: voxel slot i, channel e of the quad
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = cq * 4 + e;
            const int k0 = vg * VT;                                 // first voxel slot of this thread
            uint16_t* dst = As + row * SPL_RS + ((((k0 >> 3) ^ ((row >> 4) & 3)) << 3) | (k0 & 7));
            if (VT == 4) {
                uint2 s0, s1, s2;
                spl_split<SCH>(make_float4(rf[e], rf[4 + e], rf[8 + e], rf[12 + e]), xs, s0, s1, s2);
                *reinterpret_cast<uint2*>(dst) = s0;
                if (NPL > 1) *reinterpret_cast<uint2*>(dst + APL) = s1;
                if (NPL > 2) *reinterpret_cast<uint2*>(dst + 2 * APL) = s2;
            } else {
                uint32_t o0, o1, o2;
                spl_split2<SCH>(rf[e], rf[4 + e], xs, o0, o1, o2);
                *reinterpret_cast<uint32_t*>(dst) = o0;
                if (NPL > 1) *reinterpret_cast<uint32_t*>(dst + APL) = o1;
                if (NPL > 2) *reinterpret_cast<uint32_t*>(dst + 2 * APL) = o2;
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            uint16_t* dst = Bs + (brow_ + RPB * i) * SPL_RS + bkg * 8;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint4*>(dst + pl * BPL) = rb[i][pl];
        }
    };
    if (it_begin < it_end) {
        load_tile();
        store_tile();
    }
    __syncthreads();
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    const uint16_t* abase = As + (wm * WM + frow) * SPL_RS;
    const uint16_t* bbase = Bs + (wn * WN + frow) * SPL_RS + fk;
    for (int it = it_begin; it < it_end; ++it) {
        const bool more = it + 1 < it_end;
        if (more) load_tile();
#pragma unroll
        for (int ks = 0; ks < CBK / 16; ++ks) {
            bf16x8 fa[NPL][MT], fb[NPL][NT];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    fa[pl][t] = *reinterpret_cast<const bf16x8*>(abase + pl * APL + t * 32 * SPL_RS + (((2 * ks + (lane >> 5)) ^ (((wm * WM + t * 32 + frow) >> 4) & 3)) << 3));
#pragma unroll
                for (int t = 0; t < NT; ++t) fb[pl][t] = *reinterpret_cast<const bf16x8*>(bbase + pl * BPL + t * 32 * SPL_RS + ks * 16);
            }
#pragma unroll
            for (int order = NPL - 1; order >= 0; --order)
                if (order <= p.max_order)
#pragma unroll
                for (int pa = 0; pa <= order; ++pa) {
                    const int pb = order - pa;
#pragma unroll
                    for (int ta_ = 0; ta_ < MT; ++ta_)
#pragma unroll
                        for (int tb_ = 0; tb_ < NT; ++tb_)
                            acc[ta_][tb_] = spl_mfma32<SCH>(fa[pa][ta_], fb[pb][tb_], acc[ta_][tb_]);
                }
        }
        __syncthreads();
        if (more) store_tile();
        __syncthreads();
    }
    constexpr int CLDC = BN + 4;
    float* Cs = reinterpret_cast<float*>(lds16);
    for (int h = 0; h < WGM; ++h) {
        if (wm == h) {
#pragma unroll
            for (int ta_ = 0; ta_ < MT; ++ta_)
#pragma unroll
                for (int tb_ = 0; tb_ < NT; ++tb_)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Cs[(ta_ * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CLDC + wn * WN + tb_ * 32 + (lane & 31)] = acc[ta_][tb_][r];
        }
        __syncthreads();
        float mx_unused = 0.0f;
        conv_store_rows<BN, NTHR>(p, Cs, CLDC, m0 + h * WM, WM, n0, tid, 0, bz, mx_unused, osc);
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------------
// Wave-specialised 128 x 256 tile (the 256-channel layers of the neck and the FPN): 4 consumer waves (2 x 2, wave tile
// 64 x 128: 96 MFMAs per K step, nothing else) + 4 producer waves (one per SIMD: loads, the A split, LDS fills of the
// other of two stages).  Measured on the 128 x 128 tile: staging costs a wave ~2400 cycles per K step on its own (one
// VALU instruction per >= 4 cycles), more than the 1536 MFMA cycles it feeds -- with 256 output channels the same A
// split feeds 3072 MFMA cycles.  Producers keep two K steps in flight in a register ring; loads are buffer loads whose
// out-of-range lanes (padding taps, rows past M, channels past Cout) return zeros in hardware, so a step costs the
// producer one mask test per row and no address arithmetic (the tap / chunk walk lives in the scalar offset).
// LDS rows are unpadded 64-byte runs (2 x 72 KB stages); the 16-byte chunk index is XORed with (row >> 2) & 3, which
// spreads the 16 lanes of every ds_read_b128 group over all 16 slots of the 256-byte bank row.
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
// chunk swizzle of the unpadded 64-byte LDS rows: physical 16-byte chunk = chunk ^ ws_swz(row).  The 16 x 16 x 32 fragment
// read (lane l: row base + (l & 15), chunk l >> 4) is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... --
// rows {0-3,12-15} at chunk c together with rows 4-11 at chunk c ^ 1.  XORing bit 1 of the chunk with bit 2 of the row puts
// those 16 accesses on 16 distinct slots of the 256-byte bank row for ANY base row (exhaustive search over the 4-entry
// tables on (row >> 2) & 3; the halo tiles read at arbitrary row offsets -- with a table that is only conflict-free for
// 16-row-aligned bases half of their LDS cycles were bank conflicts, SQ_LDS_BANK_CONFLICT 323 M -> 162 M on the 849-GFLOP layer).
__device__ __forceinline__ int ws_swz(int row) { return (row >> 1) & 2; }
// 16 bytes per lane, global -> LDS without registers: lane l lands at lds_dst + 16 l (lds_dst wave-uniform); lanes whose
// offset is out of range write zeros.  (A plain function: the address-space cast does not survive the host pass of a
// kernel template.)
__device__ __forceinline__ void spl_dma16(__amdgpu_buffer_rsrc_t rsrc, uint16_t* lds_dst, unsigned voffset, unsigned soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}

#define WS_BM 128
#define WS_BN 256
#define WS_DEPTH 2
#define WS_OOB 0x80000000u

// SCH: the arithmetic scheme (Spl above).
template <int SCH>
__global__ __launch_bounds__(512, 1) void k_conv_split_ws(const Conv3dParams p, const uint16_t* __restrict__ wsplit) {
    constexpr int APL = WS_BM * CBK, BPL = WS_BN * CBK;   // one plane, in bf16 elements
    constexpr int NPL = Spl<SCH>::NPL, WPL = Spl<SCH>::WPL;
    constexpr int STAGE = NPL * (APL + BPL);
    constexpr int AR = 4, BR = 4;                          // rows per producer thread: A 128 / 32, B 256 / 64
    extern __shared__ __attribute__((aligned(16))) uint16_t lds16[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the role branch and everything under it stay wave-uniform
    const bool consumer = wave < 4;
    const int wm = (wave & 3) >> 1, wn = wave & 1;
    const ConvBlock blk = conv_block(p);
    const int m0 = blk.x * WS_BM, n0 = blk.y * WS_BN;
    const float amax_in = conv_amax_in(p);
    conv_guard_check(p, amax_in);
    const int stid = tid & 255;
    const int akq = stid & 7, arow_ = stid >> 3;
    const int bkg = stid & 3, brow_ = stid >> 2;

    const int cin_steps = p.Cin / CBK;
    const int wkd = p.transposed ? 1 : p.kd, wkh = p.transposed ? 1 : p.kh, wkw = p.transposed ? 1 : p.kw;
    const int taps = wkd * wkh * wkw;
    const int n_iters_all = taps * cin_steps;
    int it_begin = 0, it_end = n_iters_all;
    int ztap = 0;
    if (p.transposed) {
        ztap = blk.z;
    } else if (p.splits > 1) {
        const int s = blk.z;
        it_begin = (int)((int64_t)n_iters_all * s / p.splits);
        it_end = (int)((int64_t)n_iters_all * (s + 1) / p.splits);
    }
    const int n_it = it_end - it_begin;
    const int n_round = (n_it + WS_DEPTH - 1) / WS_DEPTH * WS_DEPTH;

    f32x4v acc[4][8];   // 16 x 16 tiles of the wave's 64 x 128
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    if (!consumer) {
        // ---------------- producers ----------------
        // A: buffer base shifted back by the padding so that the tap offset in the scalar register is non-negative
        const int64_t shift = p.transposed ? 0 : (((int64_t)p.pd * p.H + p.ph) * p.W + p.pw) * p.Cin;
        const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in - shift), 0, WS_OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc((void*)wsplit, 0, WS_OOB, 0x00020000);
        unsigned avoff[AR], amask[AR];   // byte offset of the row's own position; bit t CLEAR = tap t reads inside the grid
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int m = m0 + arow_ + 32 * i;
            const bool vok = m < p.M;
            const int mm = vok ? m : 0;
            const int ow_ = p.transposed ? p.W : p.OW, oh_ = p.transposed ? p.H : p.OH;
            const int vw = mm % ow_, vh = (mm / ow_) % oh_, vd = mm / (ow_ * oh_);
            const int sd = p.transposed ? 1 : p.sd, sh = p.transposed ? 1 : p.sh, sw = p.transposed ? 1 : p.sw;
            avoff[i] = (unsigned)(((((int64_t)vd * sd) * p.H + vh * sh) * p.W + vw * sw) * p.Cin * 4 + akq * 16);
            unsigned msk = 0;
            if (vok) {
                if (p.transposed) {
                    msk = 1u;
                } else {
                    for (int t = 0; t < taps; ++t) {
                        const int kd = t / (wkh * wkw), kh = (t / wkw) % wkh, kw = t % wkw;
                        const int id = vd * sd + kd - p.pd, ih = vh * sh + kh - p.ph, iw = vw * sw + kw - p.pw;
                        if ((unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) msk |= 1u << t;
                    }
                }
            }
            amask[i] = ~msk;
        }
        // weight tile by LDS-DMA (no registers, no ds_write): lane (row = brow_ (+ 64 i), physical chunk = bkg) fetches the
        // logical chunk physical ^ swizzle(row) of its row; the LDS image is lane-linear, 1 KB per wave instruction
        unsigned bvoff[BR];
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int row = brow_ + 64 * i, co = n0 + row;
            bvoff[i] = co < p.Cout ? (unsigned)((co * CBK + ((bkg ^ ws_swz(row)) * 8)) * 2) : WS_OOB;
        }
        const int pw4 = wave - 4;
        const float xs = SCH == 1 ? conv_xscale(p.amax_in) : 1.0f;
        const unsigned wtile_b = (unsigned)p.Cout * CBK * 2;   // bytes of one weight plane of one K step
        // LDS destinations (bytes within a stage): the chunk swizzle depends on the row only through (row >> 2) & 3
        unsigned adst[AR];
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int row = arow_ + 32 * i;
            adst[i] = (unsigned)((row * CBK + (((akq >> 1) ^ ws_swz(row)) * 8) + (akq & 1) * 4) * 2);
        }

        int nkd, nkh, nkw, ncs;   // the next tile to load: chunk outermost, taps innermost
        {
            const int t0 = it_begin % taps;
            ncs = it_begin / taps;
            nkd = t0 / (wkh * wkw); nkh = (t0 / wkw) % wkh; nkw = t0 % wkw;
        }
        u32x4 ra[WS_DEPTH][AR];
        unsigned asoff = 0, tapsh = 31;
        int bkd = nkd, bkh = nkh, bkw = nkw, bcs = ncs;   // the weight tiles walk the same sequence on their own clock
        unsigned soff = 0;
        auto dma_b = [&](int u, bool live) {   // weight tile u -> stage u & 1 (past the end: the last tile again, into a stage nobody reads)
            if (live) {
                const int tap = p.transposed ? ztap : (bkd * wkh + bkh) * wkw + bkw;
                soff = __builtin_amdgcn_readfirstlane((unsigned)(((int64_t)tap * cin_steps + bcs) * WPL) * wtile_b);
            }
            uint16_t* stage = lds16 + (u & 1) * STAGE + NPL * APL;
#pragma unroll
            for (int i = 0; i < BR; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    spl_dma16(bres, stage + pl * BPL + (64 * i + pw4 * 16) * CBK, bvoff[i], __builtin_amdgcn_readfirstlane(soff + pl * wtile_b));
            if (live) {
                if (++bkw == wkw) {
                    bkw = 0;
                    if (++bkh == wkh) {
                        bkh = 0;
                        if (++bkd == wkd) { bkd = 0; ++bcs; }
                    }
                }
            }
        };
        auto load_tile = [&](int slot, bool live) {
            // scalar side of the addresses (the buffer instruction takes them from SGPRs).  A tile past the end of this
            // workgroup's K range (`live` false: the last WS_DEPTH steps) re-reads the previous tile -- legal addresses, the
            // copy is staged but never multiplied -- so no per-load select is needed
            if (live) {
                const int tapw = (nkd * wkh + nkh) * wkw + nkw;                    // tap inside the walk (mask bit)
                asoff = __builtin_amdgcn_readfirstlane((unsigned)(((((int64_t)nkd * p.H + nkh) * p.W + nkw) * p.Cin + ncs * CBK) * 4));
                tapsh = 31 - tapw;
            }
#pragma unroll
            for (int i = 0; i < AR; ++i)   // (amask << (31 - tap)) has bit 31 set when this tap reads padding: offset >= 2^31 -> zeros
                ra[slot][i] = __builtin_amdgcn_raw_buffer_load_b128(ares, ((amask[i] << tapsh) & WS_OOB) | avoff[i], asoff, 0);
            if (++nkw == wkw) {
                nkw = 0;
                if (++nkh == wkh) {
                    nkh = 0;
                    if (++nkd == wkd) { nkd = 0; ++ncs; }
                }
            }
        };
        auto store_tile = [&](int slot, int stage) {
            char* base = reinterpret_cast<char*>(lds16) + stage * (STAGE * 2);
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const u32x4 u = ra[slot][i];
                uint2 s0, s1, s2;
                const float4 xv = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
                spl_split<SCH>(xv, xs, s0, s1, s2);
                *reinterpret_cast<uint2*>(base + adst[i]) = s0;
                if (NPL > 1) *reinterpret_cast<uint2*>(base + adst[i] + APL * 2) = s1;
                if (NPL > 2) *reinterpret_cast<uint2*>(base + adst[i] + APL * 4) = s2;
            }
        };

        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int d = 0; d < WS_DEPTH; ++d) load_tile(d, d < n_it);
        store_tile(0, 0);
        dma_b(0, n_it > 0);
        load_tile(0, WS_DEPTH < n_it);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR) : "memory");   // weight tile 0 has landed (the AR younger activation loads may fly)
        __syncthreads();
        for (int j0 = 0; j0 < n_round; j0 += WS_DEPTH) {
#pragma unroll
            for (int d = 0; d < WS_DEPTH; ++d) {   // tile j0 + d is being multiplied; stage tile j0 + d + 1 (ring slot (d + 1) % DEPTH)
                // (the DMA goes after the ds_writes of the split: the compiler drains vmcnt before any LDS store that follows an
                // LDS-DMA in program order)
                store_tile((d + 1) % WS_DEPTH, (j0 + d + 1) & 1);
                dma_b(j0 + d + 1, j0 + d + 1 < n_it);                // its stage was released by the barrier that ended step j0 + d - 1
                load_tile((d + 1) % WS_DEPTH, j0 + d + 1 + WS_DEPTH < n_it);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR) : "memory");   // the weight tile has landed; completion is in order
                __syncthreads();
            }
        }
    } else {
        // ---------------- consumers ----------------
        // v_mfma_f32_16x16x32_bf16: lane l holds row (l & 15), k = 8 (l >> 4) + j -> one 16-byte read covers a K step's
        // quarter row; twice the MFMA count of the 32x32x16 form at half the cycles each (more issue gaps for the
        // producer wave on the same SIMD, and the chip holds a higher clock on this shape)
        const int frow = lane & 15, fc = lane >> 4;
        const int kc = (fc ^ ws_swz(frow)) * 8;                 // (row >> 2) & 3 is the same for every 16-row tile
        const int aoff = (wm * 64 + frow) * CBK + kc;
        const int boff = NPL * APL + (wn * 128 + frow) * CBK + kc;
        __syncthreads();
        for (int j = 0; j < n_round; ++j) {
            if (j < n_it) {
                const uint16_t* st = lds16 + (j & 1) * STAGE;
                bf16x8 fa[NPL][4], fb[NPL][8];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {      // (A plane k, B plane NPL-1-k): the operands of the first products first
                    const int pq = NPL - 1 - pl;
#pragma unroll
                    for (int t = 0; t < 4; ++t) fa[pl][t] = *reinterpret_cast<const bf16x8*>(st + aoff + pl * APL + t * 16 * CBK);
#pragma unroll
                    for (int t = 0; t < 8; ++t) fb[pq][t] = *reinterpret_cast<const bf16x8*>(st + boff + pq * BPL + t * 16 * CBK);
                }
#pragma unroll
                for (int order = NPL - 1; order >= 0; --order)
                    if (order <= p.max_order)
#pragma unroll
                    for (int pa = 0; pa <= order; ++pa) {
                        const int pb = order - pa;
#pragma unroll
                        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                            for (int tb = 0; tb < 8; ++tb)
                                acc[ta][tb] = spl_mfma16<SCH>(fa[pa][ta], fb[pb][tb], acc[ta][tb]);
                    }
            }
            __syncthreads();
        }
    }

    // ---- epilogue (all 8 waves store): one 64-row half of the tile at a time through LDS ----
    constexpr int CLDC = WS_BN + 4;
    float* Cs = reinterpret_cast<float*>(lds16);
    float mx = 0.0f;
    for (int h = 0; h < 2; ++h) {
        if (consumer && wm == h) {
            // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = 0; tb < 8; ++tb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Cs[(ta * 16 + (lane >> 4) * 4 + r) * CLDC + wn * 128 + tb * 16 + (lane & 15)] = acc[ta][tb][r];
        }
        __syncthreads();
        conv_store_rows<WS_BN, 512>(p, Cs, CLDC, m0 + h * 64, 64, n0, tid, ztap, blk.z, mx, conv_oscale_of(p, amax_in));
        __syncthreads();
    }
    if (p.amax_out && conv_writes_final(p)) conv_amax_commit(p.amax_out, mx);
}

// arithmetic scheme of a launch (Spl): max_order 2 = bf16x3, 1 = fp16 pair, 0 = one bf16 product
static inline int conv_scheme(const Conv3dParams& p) { return p.max_order == 0 ? 2 : (p.max_order == 1 ? 1 : 0); }

static int split_launch_ws(const Conv3dParams& p, hipStream_t st, const char* fn) {
    const int taps = p.transposed ? 1 : p.kd * p.kh * p.kw;
    NDET_REQUIRE(taps <= 32, NDET_E_UNSUPPORTED, "%s: the 128x256 tile supports at most 32 taps", fn);
    NDET_REQUIRE((int64_t)p.D * p.H * p.W * p.Cin * 4 < ((int64_t)1 << 31) && (int64_t)(p.transposed ? 8 : taps) * p.Cin * p.Cout * 6 < ((int64_t)1 << 31),
                 NDET_E_UNSUPPORTED, "%s: the 128x256 tile addresses at most 2 GB per operand", fn);
    const int zdim = p.transposed ? 8 : p.splits;
    dim3 grid((p.M + WS_BM - 1) / WS_BM, (p.Cout + WS_BN - 1) / WS_BN, zdim);
    const int sch = conv_scheme(p);
    size_t lds = (size_t)2 * (sch == 2 ? 1 : (sch == 1 ? 2 : 3)) * (WS_BM + WS_BN) * CBK * sizeof(uint16_t);
    const size_t cs = (size_t)64 * (WS_BN + 4) * sizeof(float);   // the epilogue's C staging
    if (cs > lds) lds = cs;
    static bool attr_set[3] = {false, false, false};
    const void* kfn = sch == 2 ? (const void*)k_conv_split_ws<2> : (sch == 1 ? (const void*)k_conv_split_ws<1> : (const void*)k_conv_split_ws<0>);
    if (!attr_set[sch]) {
        hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        NDET_REQUIRE(e == hipSuccess, NDET_E_LAUNCH, "%s: cannot raise the LDS limit: %s", fn, hipGetErrorString(e));
        attr_set[sch] = true;
    }
    if (sch == 2) hipLaunchKernelGGL(k_conv_split_ws<2>, grid, dim3(512), lds, st, p, (const uint16_t*)p.w);
    else if (sch == 1) hipLaunchKernelGGL(k_conv_split_ws<1>, grid, dim3(512), lds, st, p, (const uint16_t*)p.w);
    else hipLaunchKernelGGL(k_conv_split_ws<0>, grid, dim3(512), lds, st, p, (const uint16_t*)p.w);
    return NDET_OK;
}


// ------------------------------------------------------------------------------------------------
// Persistent form of the wave-specialised tile (BM x 256, BM = 128 or 64).  The one-shot kernel above spends a fixed 6 - 8 us per
// workgroup outside its K walk -- row addresses, the first loads' latency, the C tile through LDS, two barriers per half -- with one
// workgroup per CU and nothing else on the CU to use that time: 1/3 of a workgroup's life at K = 256 (8 steps), 1/5 at K = 512.
// Here one workgroup per CU walks a list of tiles:
//   * the CONSUMERS store their accumulators straight from the 16x16 MFMA's C layout (a register = 16 consecutive channels of 4 rows;
//     buffer stores, lane part of the address in one VGPR, the register's row in the scalar offset, rows past M outside the descriptor);
//     no LDS round trip and no barrier in the epilogue, so
//   * the PRODUCERS are free the moment the last K step's barrier falls: they compute the next tile's addresses, fetch its first two
//     K steps, split step 0 into LDS and start its weight DMA WHILE the consumers drain the current tile -- the next tile's first MFMA
//     issues as soon as the last store has.
// Tiles are dealt so that the workgroups of one XCD (blockIdx % 8) hold neighbouring tiles: the column tiles of one row tile (same
// activation rows) and the row tiles of one weight slice meet in the same L2.  Split-K partials, the plain and the nearest-x2 residual,
// both ReLU positions: as above, bit-identical results.  BM = 64 (four consumer waves side by side, 64 x 64 each) doubles the tile
// count of launches that would otherwise fill half of the chip (M = 15 000: 118 row tiles).
// ------------------------------------------------------------------------------------------------
// 8-byte LDS store the compiler does not see as one.  SIInsertWaitcnts drains vmcnt before any LDS store that follows an LDS-DMA in
// program order (it cannot tell that the DMA's destination and the store's do not overlap), which forced the one-shot tile to issue its
// weight DMA AFTER the activation split -- the DMA's whole latency then sat between the split and the barrier, every K step.  Here the
// DMA goes out first and the split runs under its latency.  No result register is involved (DESIGN.md 9.1's rule concerns loads); the
// stores are retired by an explicit lgkmcnt(0) before the barrier that publishes the stage.
template <int OFF>
__device__ __forceinline__ void lds_store64_unseen(unsigned addr, uint2 v) {
    const unsigned long long d = ((unsigned long long)v.y << 32) | v.x;
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(d), "n"(OFF) : "memory");
}

#ifndef WSP_DEPTH
#define WSP_DEPTH 2    // K steps of activation loads in flight per producer wave
#endif
template <int SCH, int BM, int NCONS>
__global__ __launch_bounds__(64 * (NCONS + 4), 1) void k_conv_split_wsp(const Conv3dParams p, const uint16_t* __restrict__ wsplit, int n_mt, int n_nt) {
    static_assert((BM == 128 && (NCONS == 4 || NCONS == 8)) || (BM == 64 && NCONS == 4), "consumer layouts: 2 x 2 of 64 x 128, 2 x 4 of 64 x 64, 1 x 4 of 64 x 64");
    constexpr int CW = NCONS / (BM / 64);                 // consumer waves along N (BM / 64 along M)
    constexpr int WNC = WS_BN / CW, NTB = WNC / 16;       // columns and 16-column MFMA tiles per consumer wave
    constexpr int APL = BM * CBK, BPL = WS_BN * CBK;      // one plane, in bf16 elements
    constexpr int NPL = Spl<SCH>::NPL, WPL = Spl<SCH>::WPL;
    constexpr int STAGE = NPL * (APL + BPL);
    extern __shared__ __attribute__((aligned(16))) uint16_t lds16[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < NCONS;
    const float xs = SCH == 1 ? conv_xscale(p.amax_in) : 1.0f;
    const float osc = conv_oscale(p);
    if (SCH == 1) conv_guard_check(p, conv_amax_in(p));
    float mx = 0.0f;
    const int wm = (wave % NCONS) / CW, wn = (wave % NCONS) % CW;

    const int cin_steps = p.Cin / CBK;
    const int wkd = p.kd, wkh = p.kh, wkw = p.kw;
    const int taps = wkd * wkh * wkw;
    const int n_iters_all = taps * cin_steps;
    const int n_tiles = n_mt * n_nt * p.splits;
    // workgroup b sits on XCD b % 8: give the workgroups of one XCD consecutive tiles
    const int G = gridDim.x;
    const int vb = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;

    for (int tile = vb; tile < n_tiles; tile += G) {
        const int nt = tile % n_nt, mt = (tile / n_nt) % n_mt, zs = tile / (n_nt * n_mt);
        const int m0 = mt * BM, n0 = nt * WS_BN;
        int it_begin = 0, it_end = n_iters_all;
        if (p.splits > 1) {
            it_begin = (int)((int64_t)n_iters_all * zs / p.splits);
            it_end = (int)((int64_t)n_iters_all * (zs + 1) / p.splits);
        }
        const int n_it = it_end - it_begin;
        const int n_round = (n_it + WSP_DEPTH - 1) / WSP_DEPTH * WSP_DEPTH;

        if (!consumer) {
            // ---------------- producers (waves 4-7): a quarter of the activation rows and a quarter of the weight tile each ----------------
            // An LDS-DMA piece (1 KiB) keeps its wave at the issue stage for ~130 cycles on an idle CU and ~250 beside the MFMA stream and the
            // consumers' fragment reads (s_memtime stamps: the 12 pieces of a step = 3 000 cycles, the split of the step's 16 activation
            // elements 1 000, in a 4 700-cycle step whose MFMAs need 3 100).  Issued as one block -- as the one-shot tile must, see
            // lds_store64_unseen -- the two add up and the producers set the step time.  Here every group of weight pieces is followed by the
            // split of one activation piece: the vector unit works while the texture path digests the pieces already issued.
            const int64_t shift = (((int64_t)p.pd * p.H + p.ph) * p.W + p.pw) * p.Cin;
            const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in - shift), 0, WS_OOB, 0x00020000);
            const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc((void*)wsplit, 0, WS_OOB, 0x00020000);
            constexpr int AR = BM / 32, BR = 4;                   // activation rows / weight rows per producer thread
            const int stid = tid - 64 * NCONS;
            const int akq = stid & 7, arow_ = stid >> 3;
            const int bkg = stid & 3, brow_ = stid >> 2;
            unsigned avoff[AR], amask[AR], adst[AR];   // byte offset of the row's own position; bit t CLEAR = tap t reads inside the grid
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int row = arow_ + 32 * i;
                const int m = m0 + row;
                const bool vok = m < p.M;
                const int mm = vok ? m : 0;
                const int vw = mm % p.OW, vh = (mm / p.OW) % p.OH, vd = mm / (p.OW * p.OH);
                avoff[i] = (unsigned)(((((int64_t)vd * p.sd) * p.H + vh * p.sh) * p.W + vw * p.sw) * p.Cin * 4 + akq * 16);
                unsigned msk = 0;
                if (vok) {
                    for (int t = 0; t < taps; ++t) {
                        const int kd = t / (wkh * wkw), kh = (t / wkw) % wkh, kw = t % wkw;
                        const int id = vd * p.sd + kd - p.pd, ih = vh * p.sh + kh - p.ph, iw = vw * p.sw + kw - p.pw;
                        if ((unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) msk |= 1u << t;
                    }
                }
                amask[i] = ~msk;
                adst[i] = (unsigned)((row * CBK + (((akq >> 1) ^ ws_swz(row)) * 8) + (akq & 1) * 4) * 2);
            }
            unsigned bvoff[BR];
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const int row = brow_ + 64 * i, co = n0 + row;
                bvoff[i] = co < p.Cout ? (unsigned)((co * CBK + ((bkg ^ ws_swz(row)) * 8)) * 2) : WS_OOB;
            }
            const int pw4 = wave - NCONS;
            const unsigned wtile_b = (unsigned)p.Cout * CBK * 2;   // bytes of one weight plane of one K step
            const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(char*)lds16;
            int nkd, nkh, nkw, ncs;   // the next tile of the K walk to load: chunk outermost, taps innermost
            {
                const int t0 = it_begin % taps;
                ncs = it_begin / taps;
                nkd = t0 / (wkh * wkw); nkh = (t0 / wkw) % wkh; nkw = t0 % wkw;
            }
            u32x4 ra[WSP_DEPTH][AR];
            unsigned asoff = 0, tapsh = 31;
            int bkd = nkd, bkh = nkh, bkw = nkw, bcs = ncs;   // the weight tiles walk the same sequence on their own clock
            unsigned soff = 0;
            // tile u of the K walk -> stage u & 1: its weight pieces by LDS-DMA and, between them, the split of the activation pieces in ring
            // slot `slot` (past the end of the range: the last tile again, into a stage nobody reads)
            auto stage_tile = [&](int u, int slot, bool live) {
                if (live) {
                    const int tap = (bkd * wkh + bkh) * wkw + bkw;
                    soff = __builtin_amdgcn_readfirstlane((unsigned)(((int64_t)tap * cin_steps + bcs) * WPL) * wtile_b);
                    if (++bkw == wkw) {
                        bkw = 0;
                        if (++bkh == wkh) {
                            bkh = 0;
                            if (++bkd == wkd) { bkd = 0; ++bcs; }
                        }
                    }
                }
                uint16_t* bst = lds16 + (u & 1) * STAGE + NPL * APL;
                const unsigned abase = lds0 + (u & 1) * (STAGE * 2);
                constexpr int BPA = BR / AR;                       // weight row groups issued per activation piece (1 at BM = 128, 2 at BM = 64)
#pragma unroll
                for (int i = 0; i < AR; ++i) {
#pragma unroll
                    for (int g = i * BPA; g < (i + 1) * BPA; ++g)
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl)
                            spl_dma16(bres, bst + pl * BPL + (64 * g + pw4 * 16) * CBK, bvoff[g], __builtin_amdgcn_readfirstlane(soff + pl * wtile_b));
                    const u32x4 uu = ra[slot][i];
                    uint2 s0, s1, s2;
                    const float4 xv = make_float4(__uint_as_float(uu.x), __uint_as_float(uu.y), __uint_as_float(uu.z), __uint_as_float(uu.w));
                    spl_split<SCH>(xv, xs, s0, s1, s2);
                    lds_store64_unseen<0>(abase + adst[i], s0);
                    if (NPL > 1) lds_store64_unseen<APL * 2>(abase + adst[i], s1);
                    if (NPL > 2) lds_store64_unseen<APL * 4>(abase + adst[i], s2);
                }
            };
            auto load_tile = [&](int slot, bool live) {
                // a tile past the end of this K range (`live` false) re-reads the previous one: legal addresses, staged but never multiplied
                if (live) {
                    const int tapw = (nkd * wkh + nkh) * wkw + nkw;
                    asoff = __builtin_amdgcn_readfirstlane((unsigned)(((((int64_t)nkd * p.H + nkh) * p.W + nkw) * p.Cin + ncs * CBK) * 4));
                    tapsh = 31 - tapw;
                    if (++nkw == wkw) {
                        nkw = 0;
                        if (++nkh == wkh) {
                            nkh = 0;
                            if (++nkd == wkd) { nkd = 0; ++ncs; }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < AR; ++i)   // (amask << (31 - tap)) has bit 31 set when this tap reads padding: offset >= 2^31 -> zeros
                    ra[slot][i] = __builtin_amdgcn_raw_buffer_load_b128(ares, ((amask[i] << tapsh) & WS_OOB) | avoff[i], asoff, 0);
            };

            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int d = 0; d < WSP_DEPTH; ++d) load_tile(d, d < n_it);
            stage_tile(0, 0, n_it > 0);
            load_tile(0, WSP_DEPTH < n_it);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR) : "memory");   // weight tile 0 has landed (the AR younger activation loads may fly)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // ... and the split's LDS stores
            __syncthreads();                                              // the consumers arrive here when their epilogue is out
            for (int j0 = 0; j0 < n_round; j0 += WSP_DEPTH) {
#pragma unroll
                for (int d = 0; d < WSP_DEPTH; ++d) {   // tile j0 + d is being multiplied; stage tile j0 + d + 1 (its stage was released by the barrier
                    stage_tile(j0 + d + 1, (d + 1) % WSP_DEPTH, j0 + d + 1 < n_it);
                    load_tile((d + 1) % WSP_DEPTH, j0 + d + 1 + WSP_DEPTH < n_it);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __syncthreads();
                }
            }
        } else {
            // ---------------- consumers ----------------
            f32x4v acc[4][NTB];   // 16 x 16 tiles of the wave's 64 x WNC
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < NTB; ++b) acc[a][b] = (f32x4v){0.f, 0.f, 0.f, 0.f};
            const int frow = lane & 15, fc = lane >> 4;
            const int kc = (fc ^ ws_swz(frow)) * 8;
            const int aoff = (wm * 64 + frow) * CBK + kc;
            const int boff = NPL * APL + (wn * WNC + frow) * CBK + kc;
            __syncthreads();
            for (int j = 0; j < n_round; ++j) {
                if (j < n_it) {
                    const uint16_t* st = lds16 + (j & 1) * STAGE;
                    bf16x8 fa[NPL][4], fb[NPL][NTB];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {      // (A plane k, B plane NPL-1-k): the operands of the first products first
                        const int pq = NPL - 1 - pl;
#pragma unroll
                        for (int t = 0; t < 4; ++t) fa[pl][t] = *reinterpret_cast<const bf16x8*>(st + aoff + pl * APL + t * 16 * CBK);
#pragma unroll
                        for (int t = 0; t < NTB; ++t) fb[pq][t] = *reinterpret_cast<const bf16x8*>(st + boff + pq * BPL + t * 16 * CBK);
                    }
#pragma unroll
                    for (int order = NPL - 1; order >= 0; --order)
                        if (order <= p.max_order)
#pragma unroll
                        for (int pa = 0; pa <= order; ++pa) {
                            const int pb = order - pa;
#pragma unroll
                            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                                for (int tb = 0; tb < NTB; ++tb)
                                    acc[ta][tb] = spl_mfma16<SCH>(fa[pa][ta], fb[pb][tb], acc[ta][tb]);
                        }
                }
                __syncthreads();
            }
            // ---- epilogue, consumers only, straight from the C layout of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg ----
            const bool raw = p.splits > 1;
            float* dst = raw ? p.partial + (int64_t)zs * p.M * p.Cout : p.out;
            const unsigned obytes = (unsigned)((int64_t)p.M * p.Cout * 4);
            const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, obytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.out), 0,
                                                                                  p.res_up2 ? (unsigned)((int64_t)p.OD * p.RH * p.RW * p.Cout * 4) : obytes, 0x00020000);
            const int rowl = m0 + wm * 64 + 4 * (lane >> 4);      // + 16 ta + r
            const bool has_res = !raw && p.res != nullptr;
            // nearest-x2 residual: its row is a function of the output row (4 x 4 of them per lane, the same for every column tile)
            unsigned rrow[4][4];
            if (has_res && p.res_up2) {
#pragma unroll
                for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = rowl + 16 * ta + r;
                        const int ow = m % p.OW, oh = (m / p.OW) % p.OH, od = m / (p.OW * p.OH);
                        rrow[ta][r] = m < p.M ? (unsigned)((((int64_t)od * p.RH + (oh >> 1)) * p.RW + (ow >> 1)) * p.Cout * 4) : WS_OOB;   // < 2^31 (launcher)
                    }
            }
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb) {
                const int cbase = n0 + wn * WNC + tb * 16;
                if (cbase >= p.Cout) continue;                    // Cout % 16 == 0: a column tile is all inside or all outside
                const int co = cbase + (lane & 15);
                const float sc = (!raw && p.scale) ? p.scale[co] : 1.0f, sh = (!raw && p.scale) ? p.shift[co] : 0.0f;
                const unsigned vo = (unsigned)(((int64_t)rowl * p.Cout + co) * 4);
                float rr[4][4];
#pragma unroll
                for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (!has_res) { rr[ta][r] = 0.0f; continue; }
                        if (p.res_up2) {
                            // row offset PLUS column offset (Cout * 4 need not be a power of two: OR-ing them read Cout = 96 from the wrong place); rows past M keep
                            // bit 31 set -- row offsets are below 2^31, so the sum cannot carry out of it -- and fall outside the descriptor
                            rr[ta][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rres, rrow[ta][r] + (unsigned)(co * 4), 0, 0));
                        } else {
                            const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((16 * ta + r) * p.Cout * 4));
                            rr[ta][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rres, vo, so, 0));
                        }
                    }
#pragma unroll
                for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((16 * ta + r) * p.Cout * 4));
                        float v = acc[ta][tb][r] * osc;
                        if (!raw) {
                            if (p.scale) v = v * sc + sh;
                            if (p.relu == 2) v = fmaxf(v, 0.f);
                            if (has_res) v = v + rr[ta][r];
                            if (p.relu == 1) v = fmaxf(v, 0.f);
                            if (rowl + 16 * ta + r < p.M) mx = fmaxf(mx, fabsf(v));
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rout, vo, so, 0);
                    }
            }
        }
    }
    if (p.amax_out && p.splits == 1) conv_amax_commit(p.amax_out, mx);
}

template <int BM, int NCONS>
static int split_launch_wsp(const Conv3dParams& p, hipStream_t st, const char* fn) {
    const int taps = p.kd * p.kh * p.kw;
    NDET_REQUIRE(!p.transposed, NDET_E_UNSUPPORTED, "%s: the persistent 128x256 tile does not take transposed convolutions", fn);
    NDET_REQUIRE(taps <= 32, NDET_E_UNSUPPORTED, "%s: the persistent 128x256 tile supports at most 32 taps", fn);
    NDET_REQUIRE(p.Cout % 16 == 0, NDET_E_UNSUPPORTED, "%s: the persistent 128x256 tile needs Cout %% 16 == 0", fn);
    NDET_REQUIRE((int64_t)p.D * p.H * p.W * p.Cin * 4 < ((int64_t)1 << 31) && (int64_t)taps * p.Cin * p.Cout * 6 < ((int64_t)1 << 31) &&
                     ((int64_t)p.M + 128) * p.Cout * 4 < ((int64_t)1 << 31),
                 NDET_E_UNSUPPORTED, "%s: the persistent 128x256 tile addresses at most 2 GB per operand", fn);
    const int n_mt = (p.M + BM - 1) / BM, n_nt = (p.Cout + WS_BN - 1) / WS_BN;
    const int64_t n_tiles = (int64_t)n_mt * n_nt * p.splits;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) n_cu = 256;
        else n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = (int)(n_tiles < n_cu ? n_tiles : n_cu);
    const int sch = conv_scheme(p);
    const size_t lds = (size_t)2 * (sch == 2 ? 1 : (sch == 1 ? 2 : 3)) * (BM + WS_BN) * CBK * sizeof(uint16_t);
    static bool attr_set[3] = {false, false, false};
    const void* kfn = sch == 2 ? (const void*)k_conv_split_wsp<2, BM, NCONS> : (sch == 1 ? (const void*)k_conv_split_wsp<1, BM, NCONS> : (const void*)k_conv_split_wsp<0, BM, NCONS>);
    if (!attr_set[sch]) {
        hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        NDET_REQUIRE(e == hipSuccess, NDET_E_LAUNCH, "%s: cannot raise the LDS limit: %s", fn, hipGetErrorString(e));
        attr_set[sch] = true;
    }
    if (sch == 2) hipLaunchKernelGGL((k_conv_split_wsp<2, BM, NCONS>), dim3(grid), dim3(64 * (NCONS + 4)), lds, st, p, (const uint16_t*)p.w, n_mt, n_nt);
    else if (sch == 1) hipLaunchKernelGGL((k_conv_split_wsp<1, BM, NCONS>), dim3(grid), dim3(64 * (NCONS + 4)), lds, st, p, (const uint16_t*)p.w, n_mt, n_nt);
    else hipLaunchKernelGGL((k_conv_split_wsp<0, BM, NCONS>), dim3(grid), dim3(64 * (NCONS + 4)), lds, st, p, (const uint16_t*)p.w, n_mt, n_nt);
    return NDET_OK;
}


// ------------------------------------------------------------------------------------------------
// Halo-stationary tile for stride-1 "same" convolutions with more than one tap (3x3 of the backbone / FPN, 3x3x3 of the
// neck).  A partner wave gets about one VALU issue slot per MFMA on its SIMD, and the activation split costs 4.5 VALU
// instructions per element: splitting the A tile anew for every tap (the tiles above) makes the producers the limiter.
// Here the output tile is a TD x TH x TW patch of 128 voxels; for each 32-channel chunk the producers load and split the
// patch plus its halo ONCE ((TD+kd-1)(TH+kh-1)(TW+kw-1) rows, 1.4 - 2.8 x 128) and all taps multiply out of that LDS
// image -- a tap is a constant row offset in the halo grid.  Per tap step the producers only move the weight tile:
// LDS-DMA (buffer_load ... lds, no registers, no ds_write), 2 - 3 stages deep, waited with counted vmcnt.
// Consumers: 4 waves (2 x 2), 16x16x32 MFMAs, wave tile 64 x (16 NT16).
// ------------------------------------------------------------------------------------------------
struct HaloGeom {
    int ltd, lth, ltw;     // log2 of the patch extents (TD TH TW = 128)
    int npd, nph, npw;     // patches per axis
    int HH, HW, NH;        // halo extents along H and W, halo rows in total
    int T;                 // taps
    int KDL;               // 1: every tap reads the one halo image; kd: the depth taps are looped OUTSIDE the halo (the image has
                           // no depth margin and is re-staged, shifted in depth, once per depth tap) -- a third of the LDS rows
    int Tin;               // taps per staged image = T / KDL
};

// tile row -> GEMM row of the output voxel it holds (-1 outside the grid)
struct HaloRowMap {
    int first, ltw, lth, d0, h0, w0, OD, OH, OW;
    __device__ __forceinline__ int operator()(int row) const {
        const int r = first + row;
        const int tw = r & ((1 << ltw) - 1), th = (r >> ltw) & ((1 << lth) - 1), td = r >> (ltw + lth);
        const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
        return (od < OD && oh < OH && ow < OW) ? (od * OH + oh) * OW + ow : -1;
    }
};

// NT16: 16x16 tiles per consumer wave along N; WGN: consumer waves along N (2 along M).  <4,2>: 128 channels, <8,2>: 256 channels with
// one consumer wave per SIMD, <4,4>: 256 channels with two consumer waves per SIMD (each covers the other's LDS latency).
// SCH 0: three bf16 planes, six products (fp32-class to 2^-24).  SCH 1: two fp16 planes, three products (see spl_split4_f16).
// SCH 2: the leading bf16 plane only, one product (bf16 autocast arithmetic); the weights keep their three-plane layout.
// NPROD: producer waves (4, or 8: an LDS-DMA piece holds its wave at the issue stage for ~250 cycles beside the MFMA stream, so the weight tile of a
// 256-column step -- 32 pieces on two planes -- costs four producer waves 2 000 cycles where the step's MFMAs need 1 536; eight waves issue it in half)
template <int NT16, int WGN, int SCH, int NPROD = 4>
__global__ __launch_bounds__(64 * (2 * WGN + NPROD), 1) void k_conv_split_halo(const Conv3dParams p, const uint16_t* __restrict__ wsplit, const HaloGeom g) {
    constexpr int BN = 16 * NT16 * WGN;
    constexpr int NCONS = 2 * WGN, NTHR = 64 * (NCONS + NPROD);
    constexpr int PT = 64 * NPROD;                          // producer threads
    constexpr int HALO_MAX = (BN == 128) ? 400 : 224;
    constexpr int NSTAGE = (BN == 128) ? 3 : 2;
    constexpr int NPIECE = (HALO_MAX * 8 + PT - 1) / PT;
    constexpr int APL = HALO_MAX * CBK, BPL = BN * CBK;   // one plane, elements
    constexpr int NPL = SCH == 1 ? 2 : (SCH == 2 ? 1 : 3);   // operand planes staged and multiplied
    constexpr int WPL = SCH == 1 ? 2 : 3;                    // planes per K step in the weight tensor
    constexpr int BSTAGE = NPL * BPL;
    constexpr int BR = BN / (16 * NPROD);                  // weight rows per producer thread and plane
    static_assert(BR >= 1 && BN % (16 * NPROD) == 0, "every producer wave stages 16 rows of each row group");
    constexpr int NB = NPL * BR;                             // LDS-DMA instructions per producer thread and K step
    extern __shared__ __attribute__((aligned(16))) uint16_t lds16[];
    uint16_t* Bs = lds16 + NPL * APL;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < NCONS;
    const int wm = (wave % NCONS) / WGN, wn = (wave % NCONS) % WGN;
    const ConvBlock blk = conv_block(p);
    const int n0 = blk.y * BN;
    const int TH = 1 << g.lth, TW = 1 << g.ltw;
    int pb = blk.x;
    const int pwi = pb % g.npw; pb /= g.npw;
    const int phi = pb % g.nph;
    const int pdi = pb / g.nph;
    const int d0 = pdi << g.ltd, h0 = phi << g.lth, w0 = pwi << g.ltw;
    const float amax_in = conv_amax_in(p);
    conv_guard_check(p, amax_in);

    const int nch_all = p.Cin / CBK;
    int cb = 0, ce = nch_all;
    if (p.splits > 1) {
        cb = (int)((int64_t)nch_all * blk.z / p.splits);
        ce = (int)((int64_t)nch_all * (blk.z + 1) / p.splits);
    }
    const int nch = ce - cb;
    const int T = g.T, Tin = g.Tin;
    const int S = nch * T;
    const int nphase = nch * g.KDL;   // staged activation images: (chunk, looped depth tap)

    f32x4v acc[4][NT16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NT16; ++b) acc[a][b] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    if (!consumer) {
        // ---------------- producers ----------------
        const int stid = tid - 64 * NCONS;
        const int pw4 = wave - NCONS;
        const float xs = SCH == 1 ? conv_xscale(p.amax_in) : 1.0f;
        // looped depth taps: the depth shift (kd - pd) rides in the scalar offset, the base pointer is moved back by pd slices
        const int64_t slice = (int64_t)p.H * p.W * p.Cin;
        const bool dloop = g.KDL > 1;
        const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in - (dloop ? p.pd * slice : 0)), 0, WS_OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc((void*)wsplit, 0, WS_OOB, 0x00020000);
        unsigned avoff[NPIECE], adst[NPIECE], amask[NPIECE];   // amask bit k SET: looped depth tap k reads outside the grid
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const int idx = stid + PT * i;
            const int row = idx >> 3, q = idx & 7;
            const int hw = row % g.HW, hh = (row / g.HW) % g.HH, hd = row / (g.HW * g.HH);
            const int vd = d0 + hd - (dloop ? 0 : p.pd), vh = h0 - p.ph + hh, vw = w0 - p.pw + hw;
            const bool ok = row < g.NH && (unsigned)vd < (unsigned)p.D && (unsigned)vh < (unsigned)p.H && (unsigned)vw < (unsigned)p.W;
            avoff[i] = ok ? (unsigned)((((int64_t)vd * p.H + vh) * p.W + vw) * p.Cin * 4 + q * 16) : WS_OOB;
            unsigned msk = 0;
            if (dloop)
                for (int k = 0; k < g.KDL; ++k)
                    if ((unsigned)(vd + k - p.pd) >= (unsigned)p.D) msk |= 1u << k;
            amask[i] = msk;
            // rows past the halo are never read: their pieces are not stored (adst = ~0u)
            adst[i] = row < g.NH ? (unsigned)((row * CBK + (((q >> 1) ^ ws_swz(row)) * 8) + (q & 1) * 4) * 2) : ~0u;
        }
        // weight tile by LDS-DMA: lane (row = stid >> 2 (+ PT / 4 i), physical chunk = stid & 3) fetches logical chunk
        // physical ^ swizzle(row) of its row; the LDS image is lane-linear (1 KB per wave instruction)
        unsigned bvoff[BR];
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int row = (stid >> 2) + (PT / 4) * i, co = n0 + row;
            bvoff[i] = co < p.Cout ? (unsigned)((co * CBK + (((stid & 3) ^ ws_swz(row)) * 8)) * 2) : WS_OOB;
        }
        const unsigned wtile_b = (unsigned)p.Cout * CBK * 2;
        int dt = 0, dc = 0;   // tap and chunk (relative to cb) of the next weight tile to fetch
        auto dma_b = [&](int u) {   // tile u -> stage u % NSTAGE
            const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(((int64_t)dt * nch_all + cb + dc) * WPL) * wtile_b);
            uint16_t* stage = Bs + (u % NSTAGE) * BSTAGE;
#pragma unroll
            for (int i = 0; i < BR; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    uint16_t* dst = stage + pl * BPL + ((PT / 4) * i + pw4 * 16) * CBK;   // wave-uniform; the hardware adds lane * 16 B
                    spl_dma16(bres, dst, bvoff[i], __builtin_amdgcn_readfirstlane(soff + pl * wtile_b));
                }
            if (++dt == T) { dt = 0; ++dc; }
        };
        u32x4 ra[NPIECE];
        auto load_a = [&](int a) {   // staged image a = (chunk a / KDL, looped depth tap a % KDL)
            const int c = a / g.KDL, kdl = a - c * g.KDL;
            const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(((cb + c) * CBK + (dloop ? kdl * slice : 0)) * 4));
            const unsigned sh = 31 - kdl;
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(ares, ((amask[i] << sh) & WS_OOB) | avoff[i], soff, 0);
        };
        auto split_store_a = [&]() {
            char* base = reinterpret_cast<char*>(lds16);
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) {
                const u32x4 u = ra[i];
                uint2 s0, s1, s2;
                const float4 xv = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
                spl_split<SCH>(xv, xs, s0, s1, s2);
                if (adst[i] != ~0u) {
                    *reinterpret_cast<uint2*>(base + adst[i]) = s0;
                    if (NPL > 1) *reinterpret_cast<uint2*>(base + adst[i] + APL * 2) = s1;
                    if (NPL > 2) *reinterpret_cast<uint2*>(base + adst[i] + APL * 4) = s2;
                }
            }
        };

        __builtin_amdgcn_s_setprio(3);
        load_a(0);
#pragma unroll
        for (int u = 0; u < NSTAGE - 1; ++u)
            if (u < S) dma_b(u);
        split_store_a();
        if (nphase > 1) load_a(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // weight tile 0 (and whatever else) has landed
        __syncthreads();                                     // P0
        int s = 0;
        for (int ci = 0; ci < nphase; ++ci) {
            for (int t = 0; t < Tin; ++t, ++s) {
                if (s + NSTAGE - 1 < S) {
                    dma_b(s + NSTAGE - 1);
                    // tile s + 1 must have landed before the barrier; the NSTAGE - 2 younger tiles may stay in flight
                    // (completion is in order: right after a chunk boundary the next chunk's activation loads are younger
                    // than tile s + 1 and may stay in flight too)
                    if (NSTAGE == 3) {
                        // exactly the NB pieces just issued may stay in flight.  (Right after a chunk boundary the next chunk's
                        // NPIECE activation loads are younger than tile s + 1 too and could be allowed to fly as well -- vmcnt(NB +
                        // NPIECE) -- but that count is only right on the path through P2, which tools/audit_vmcnt.py cannot tell
                        // from the loop's back edge in the compiled control flow: the provable wait is kept.)
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();                             // P1
            }
            if (ci + 1 < nphase) {
                split_store_a();                             // image ci + 1 (loaded during image ci); the consumers wait at P2
                __syncthreads();                             // P2
                if (ci + 2 < nphase) load_a(ci + 2);
            }
        }
    } else {
        // ---------------- consumers ----------------
        const int frow = lane & 15, fc = lane >> 4;
        int hr0[4];
#pragma unroll
        for (int ta = 0; ta < 4; ++ta) {
            const int r = wm * 64 + ta * 16 + frow;
            const int tw = r & (TW - 1), th = (r >> g.ltw) & (TH - 1), td = r >> (g.ltw + g.lth);
            hr0[ta] = (td * g.HH + th) * g.HW + tw;
        }
        const int boff = (wn * (16 * NT16) + frow) * CBK + ((fc ^ ws_swz(frow)) * 8);
        const int wkh = p.kh, wkw = p.kw;
        __syncthreads();                                     // P0
        int s = 0;
        for (int ci = 0; ci < nphase; ++ci) {
            int kd = 0, kh = 0, kw = 0;   // tap inside the staged image (kd stays 0 when the depth taps are looped outside)
            for (int t = 0; t < Tin; ++t, ++s) {
                const int tapoff = (kd * g.HH + kh) * g.HW + kw;
                const uint16_t* bst = Bs + (s % NSTAGE) * BSTAGE + boff;
                bf16x8 fa[NPL][4], fb[NPL][NT16];
                const uint16_t* ap[4];
#pragma unroll
                for (int ta = 0; ta < 4; ++ta) {
                    const int hr = hr0[ta] + tapoff;
                    ap[ta] = lds16 + hr * CBK + ((fc ^ ws_swz(hr)) * 8);
                }
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {      // (A plane k, B plane NPL-1-k): the operands of the first products first -- LDS returns in
                    const int pq = NPL - 1 - pl;        // order, so the first MFMA then waits for a third / half of the reads instead of all of them
#pragma unroll
                    for (int ta = 0; ta < 4; ++ta) fa[pl][ta] = *reinterpret_cast<const bf16x8*>(ap[ta] + pl * APL);
#pragma unroll
                    for (int tb = 0; tb < NT16; ++tb) fb[pq][tb] = *reinterpret_cast<const bf16x8*>(bst + pq * BPL + tb * 16 * CBK);
                }
#pragma unroll
                for (int order = NPL - 1; order >= 0; --order)
                    if (order <= p.max_order)
#pragma unroll
                    for (int pa = 0; pa <= order; ++pa) {
                        const int pbb = order - pa;
#pragma unroll
                        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                            for (int tb = 0; tb < NT16; ++tb) {
                                acc[ta][tb] = spl_mfma16<SCH>(fa[pa][ta], fb[pbb][tb], acc[ta][tb]);
                            }
                    }
                if (++kw == wkw) {
                    kw = 0;
                    if (++kh == wkh) { kh = 0; ++kd; }
                }
                __syncthreads();                             // P1
            }
            if (ci + 1 < nphase) __syncthreads();            // P2
        }
    }
    __builtin_amdgcn_s_setprio(0);

    // ---- epilogue (all 8 waves store): one 64-row half of the patch at a time through LDS ----
    constexpr int CLDC = BN + 4;
    float* Cs = reinterpret_cast<float*>(lds16);
    float mx = 0.0f;
    for (int h = 0; h < 2; ++h) {
        if (consumer && wm == h) {
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = 0; tb < NT16; ++tb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Cs[(ta * 16 + (lane >> 4) * 4 + r) * CLDC + wn * (16 * NT16) + tb * 16 + (lane & 15)] = acc[ta][tb][r];
        }
        __syncthreads();
        const HaloRowMap rmap{h * 64, g.ltw, g.lth, d0, h0, w0, p.OD, p.OH, p.OW};
        const float osc = conv_oscale_of(p, amax_in);
        float* Wl = Cs + 64 * CLDC;                     // (BN, 32) floats behind the staged rows (the launcher checked the room)
        if (p.map_out && h == 0)                        // every wave has left the K walk (the barrier above): the operand stages are free
            for (int i = tid; i < BN * 8; i += NTHR) reinterpret_cast<float4*>(Wl)[i] = reinterpret_cast<const float4*>(p.map_w)[i];
        conv_store_rows_mapped<BN, NTHR>(p, Cs, CLDC, 64, n0, tid, 0, blk.z, rmap, mx, osc);
        __syncthreads();                                // (and Wl is in place)
        if (p.map_out) conv_map_rows<BN, NTHR>(p, Cs, CLDC, tid, rmap, osc, Wl);      // (uniform branch; barriers inside)
    }
    if (p.amax_out && conv_writes_final(p)) conv_amax_commit(p.amax_out, mx);
}

// patch shape for the halo tile: the power-of-two TD x TH x TW = 128 with the least padded work (weighted by the halo it drags)
static bool halo_geometry(const Conv3dParams& p, int halo_max, HaloGeom& g) {
    double best = 1e30;
    for (int dloop = 0; dloop <= (p.kd > 1 ? 1 : 0); ++dloop)   // depth taps inside the halo image, or looped outside it
        for (int ltd = 0; ltd <= 7; ++ltd)
            for (int lth = 0; lth + ltd <= 7; ++lth) {
                const int ltw = 7 - ltd - lth;
                const int TD = 1 << ltd, TH = 1 << lth, TW = 1 << ltw;
                const int HD = dloop ? TD : TD + p.kd - 1, HH = TH + p.kh - 1, HW = TW + p.kw - 1;
                const int NH = HD * HH * HW;
                if (NH > halo_max) continue;
                const int npd = (p.OD + TD - 1) / TD, nph = (p.OH + TH - 1) / TH, npw = (p.OW + TW - 1) / TW;
                const double waste = (double)npd * TD * nph * TH * npw * TW / ((double)p.OD * p.OH * p.OW);
                const int kdl = dloop ? p.kd : 1;
                const double cost = waste * (1.0 + 0.03 * NH * kdl / 128.0);   // staged rows per chunk, relative to the tile
                if (cost < best) {
                    best = cost;
                    g.ltd = ltd; g.lth = lth; g.ltw = ltw; g.npd = npd; g.nph = nph; g.npw = npw; g.HH = HH; g.HW = HW; g.NH = NH;
                    g.KDL = kdl;
                }
            }
    g.T = p.kd * p.kh * p.kw;
    g.Tin = g.T / (best < 1e29 ? g.KDL : 1);
    return best < 1e29;
}

template <int NT16, int WGN, int SCH = 0, int NPROD = 4>
static int split_launch_halo(const Conv3dParams& p, hipStream_t st, const char* fn) {
    constexpr int NPL = SCH == 1 ? 2 : (SCH == 2 ? 1 : 3);
    constexpr int BN = 16 * NT16 * WGN, HALO_MAX = (BN == 128) ? 400 : 224, NSTAGE = (BN == 128) ? 3 : 2;
    NDET_REQUIRE(!p.transposed && p.sd == 1 && p.sh == 1 && p.sw == 1 && (p.kd & 1) && (p.kh & 1) && (p.kw & 1) && p.pd == p.kd / 2 &&
                     p.ph == p.kh / 2 && p.pw == p.kw / 2,
                 NDET_E_UNSUPPORTED, "%s: the halo tile needs a stride-1 same-padded convolution with odd kernel extents", fn);
    NDET_REQUIRE((int64_t)p.D * p.H * p.W * p.Cin * 4 < ((int64_t)1 << 31) && (int64_t)p.kd * p.kh * p.kw * p.Cin * p.Cout * 6 < ((int64_t)1 << 31),
                 NDET_E_UNSUPPORTED, "%s: the halo tile addresses at most 2 GB per operand", fn);
    NDET_REQUIRE(p.splits <= p.Cin / CBK, NDET_E_INVALID, "%s: the halo tile splits K over the %d channel chunks only", fn, p.Cin / CBK);
    HaloGeom g;
    NDET_REQUIRE(halo_geometry(p, HALO_MAX, g), NDET_E_UNSUPPORTED, "%s: no patch shape fits the halo tile", fn);
    dim3 grid(g.npd * g.nph * g.npw, (p.Cout + BN - 1) / BN, p.splits);
    size_t lds = (size_t)(NPL * HALO_MAX * CBK + NSTAGE * NPL * BN * CBK) * sizeof(uint16_t);
    const size_t cs = (size_t)64 * (BN + 4) * sizeof(float);
    if (cs > lds) lds = cs;
    const size_t lds_map = cs + (size_t)BN * 32 * sizeof(float);     // the staged rows + the chained projection's (BN, 32) weight
    const size_t lds_cap = lds_map > lds ? lds_map : lds;
    if (p.map_out) {
        NDET_REQUIRE(p.Cout == BN && p.splits == 1 && !p.res && p.relu == 0, NDET_E_UNSUPPORTED,
                     "%s: the chained projection needs a tile that owns all %d output channels of its rows, no split-K / residual / ReLU", fn, p.Cout);
        lds = lds_cap;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_conv_split_halo<NT16, WGN, SCH, NPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
        NDET_REQUIRE(e == hipSuccess, NDET_E_LAUNCH, "%s: cannot raise the LDS limit: %s", fn, hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_conv_split_halo<NT16, WGN, SCH, NPROD>), grid, dim3(64 * (2 * WGN + NPROD)), lds, st, p, (const uint16_t*)p.w, g);
    return NDET_OK;
}

template <int BM, int BN, int WGM, int WGN, int SCH>
static int split_launch_tile_sch(const Conv3dParams& p, hipStream_t st, const char* fn) {
    const int zdim = p.transposed ? 8 : p.splits;
    dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, zdim);
    size_t lds = (size_t)Spl<SCH>::NPL * (BM + BN) * SPL_RS * sizeof(uint16_t);
    const size_t cs = (size_t)(BM / WGM) * (BN + 4) * sizeof(float);
    if (cs > lds) lds = cs;
    if (lds > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)k_conv_split<BM, BN, WGM, WGN, SCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            NDET_REQUIRE(e == hipSuccess, NDET_E_LAUNCH, "%s: cannot raise the LDS limit: %s", fn, hipGetErrorString(e));
            attr_set = true;
        }
    }
    hipLaunchKernelGGL((k_conv_split<BM, BN, WGM, WGN, SCH>), grid, dim3(64 * WGM * WGN), lds, st, p, (const uint16_t*)p.w);
    return NDET_OK;
}
template <int BM, int BN, int WGM, int WGN>
static int split_launch_tile(const Conv3dParams& p, hipStream_t st, const char* fn) {
    switch (conv_scheme(p)) {
        case 1: return split_launch_tile_sch<BM, BN, WGM, WGN, 1>(p, st, fn);
        case 2: return split_launch_tile_sch<BM, BN, WGM, WGN, 2>(p, st, fn);
        default: return split_launch_tile_sch<BM, BN, WGM, WGN, 0>(p, st, fn);
    }
}
template <int NT16, int WGN, int NPROD = 4>
static int split_launch_halo_any(const Conv3dParams& p, hipStream_t st, const char* fn) {
    switch (conv_scheme(p)) {
        case 1: return split_launch_halo<NT16, WGN, 1, NPROD>(p, st, fn);
        case 2: return split_launch_halo<NT16, WGN, 2, NPROD>(p, st, fn);
        default: return split_launch_halo<NT16, WGN, 0, NPROD>(p, st, fn);
    }
}

#ifndef NDET_ORDER_DEFAULT
#define NDET_ORDER_DEFAULT -1      // -1: chosen per launch below; 0 / 1 force one order (measurement builds)
#endif
// Outputs of 32 MB and more (eight L2s of 4 MB: nothing of them is re-read from a cache) are written with non-temporal stores: the float4 copy
// kernel of bench.py reaches 6.5 TB/s with them and 5.4 without, and the HBM-bound layers gain 3 - 12 % (l1.conv1 86 -> 75 us; the step 0.05 ms).
// Measurement runs move the threshold (and switch the order-2 deal off) through ndet_measurement_knob -- an explicit call, not the environment:
// a stray variable in a production shell must not change what the production path launches.
static int64_t g_nt_bytes = (int64_t)32 << 20;
static bool g_order2 = true;
static int64_t conv_nt_bytes() { return g_nt_bytes; }
static int g_wgrad_wide = 1;     // measurement knob "wgrad_wide": 0 = the 128 x 128 weight-gradient tile for every layer
static int g_wgrad_xcd = 1;      // measurement knob "wgrad_xcd": 0 = grid order for the weight-gradient kernel's workgroups

extern "C" int ndet_measurement_knob(const char* name, int64_t value) {
    const char* fn = "ndet_measurement_knob";
    NDET_REQUIRE(name, NDET_E_INVALID, "%s: null name", fn);
    if (!strcmp(name, "nt_bytes")) { NDET_REQUIRE(value >= 0, NDET_E_INVALID, "%s: nt_bytes must be >= 0", fn); g_nt_bytes = value; return NDET_OK; }
    if (!strcmp(name, "order2")) { NDET_REQUIRE(value == 0 || value == 1, NDET_E_INVALID, "%s: order2 is 0 or 1", fn); g_order2 = value != 0; return NDET_OK; }
    if (!strcmp(name, "deterministic_scatter")) { NDET_REQUIRE(value == 0 || value == 1, NDET_E_INVALID, "%s: deterministic_scatter is 0 or 1", fn); g_ndet_deterministic_scatter = (int)value; return NDET_OK; }
    if (!strcmp(name, "wgrad_wide")) { NDET_REQUIRE(value == 0 || value == 1, NDET_E_INVALID, "%s: wgrad_wide is 0 or 1", fn); g_wgrad_wide = (int)value; return NDET_OK; }
    if (!strcmp(name, "wgrad_xcd")) { NDET_REQUIRE(value == 0 || value == 1, NDET_E_INVALID, "%s: wgrad_xcd is 0 or 1", fn); g_wgrad_xcd = (int)value; return NDET_OK; }
    ndet_set_error("%s: unknown knob '%s' (nt_bytes, order2, deterministic_scatter, wgrad_wide, wgrad_xcd)", fn, name);
    return NDET_E_INVALID;
}
extern "C" int ndet_amax_slot_floats(void) { return NDET_AMAX_SUB * NDET_AMAX_STRIDE; }

int conv_split_launch(Conv3dParams& p, int tile, hipStream_t st, const char* fn) {
    p.direct = 0;
    p.nt = (p.splits <= 1 || p.transposed) && (int64_t)p.M * p.Cout * 4 * (p.transposed ? 8 : 1) >= conv_nt_bytes() ? 1 : 0;
    if (tile >= 100000 && (tile - 100000 == 64 || tile - 100000 == 128 || tile - 100000 == 12864)) {   // 100064 / 100128 / 112864: direct epilogue
        tile -= 100000;
        NDET_REQUIRE(!p.transposed && p.splits == 1 && p.Cout % 32 == 0 && ((int64_t)p.M + 128) * p.Cout * 4 < ((int64_t)1 << 32),
                     NDET_E_UNSUPPORTED, "%s: the direct epilogue needs splits == 1, no transposition, Cout %% 32 == 0 and an output below 4 GB", fn);
        p.direct = 1;
    }
    const int64_t big_tiles = (int64_t)((p.M + 127) / 128) * ((p.Cout + 127) / 128);
    if (tile == 0) tile = (big_tiles >= 192 && p.Cout >= 128) ? 128 : 64;
    {   // which operand is worth keeping in one XCD's L2 (Conv3dParams::order): the bytes its re-reads would otherwise fetch again
        const int tm = tile == 64 ? 64 : 128, tn = tile == 64 || tile == 12864 ? 64 : (tile == 128 || tile == 3128 ? 128 : 256);
        const int64_t mt = (p.M + tm - 1) / tm, nt = (p.Cout + tn - 1) / tn;
        const int64_t taps = p.transposed ? 1 : (int64_t)p.kd * p.kh * p.kw;
        const int64_t w_bytes = taps * p.Cin * p.Cout * 6, a_bytes = (int64_t)p.M * p.Cin * 4 * (p.sd * p.sh * p.sw);
        const int64_t save_w = (mt - 1) * w_bytes, save_a = (nt - 1) * a_bytes;
        p.order = NDET_ORDER_DEFAULT;
        if (NDET_ORDER_DEFAULT < 0) {
            p.order = 0;
            if (!p.transposed && mt > 1 && nt * p.splits > 1 && save_w > save_a && w_bytes > (8 << 20)) p.order = 1;
            // the mirror case: weights small enough to sit in every L2 (1x1 layers), several column tiles, the rows re-read once per XCD otherwise
            if (g_order2 && p.order == 0 && !p.transposed && nt > 1 && mt >= 16 && w_bytes <= (2 << 20) && save_a >= (8 << 20) &&
                (tile == 64 || tile == 128 || tile == 12864 || tile == 128256)) p.order = 2;
        }
    }
    int rc;
    switch (tile) {
        case 64: rc = split_launch_tile<64, 64, 2, 2>(p, st, fn); break;
        case 128: rc = split_launch_tile<128, 128, 2, 2>(p, st, fn); break;
        case 12864: rc = split_launch_tile<128, 64, 2, 2>(p, st, fn); break;
        case 128256: rc = split_launch_ws(p, st, fn); break;
        case 129256: rc = split_launch_wsp<128, 4>(p, st, fn); break;     // persistent form of 128256
        case 129257: rc = split_launch_wsp<128, 8>(p, st, fn); break;     // ... with eight consumer waves (two per SIMD, 64 x 64 each)
        case 129064: rc = split_launch_wsp<64, 4>(p, st, fn); break;      // ... with 64-row tiles
        case 3128: rc = split_launch_halo_any<4, 2>(p, st, fn); break;
        case 3256: rc = split_launch_halo_any<8, 2>(p, st, fn); break;
        case 3257: rc = split_launch_halo_any<4, 4>(p, st, fn); break;
        case 3258:                                                            // ... with eight producer waves (16 waves: 128 registers each --
            rc = conv_scheme(p) == 1 ? split_launch_halo<4, 4, 1, 8>(p, st, fn)                  // the three-plane form does not fit and keeps four)
                 : conv_scheme(p) == 2 ? split_launch_halo<4, 4, 2, 8>(p, st, fn) : split_launch_halo<4, 4, 0, 4>(p, st, fn);
            break;
        default: ndet_set_error("%s: unknown tile %d", fn, tile); return NDET_E_INVALID;
    }
    if (rc != NDET_OK) return rc;
    NDET_CHECK_LAUNCH(fn);
    if (g_keep_partials && p.splits > 1) return NDET_OK;
    return conv_splitk_reduce_launch(p, st, fn);
}

// packed fp32 weights (taps, Cout, Cin) -> three bf16 planes tiled per K step, (taps, Cin/32, 3, Cout, 32):
// w = p0 + p1 + p2 exactly
__global__ __launch_bounds__(256) void k_split_weights(const float* __restrict__ w, int taps, int Cout, int Cin, uint16_t* __restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;   // element pair along Cin
    const int64_t n = (int64_t)taps * Cout * Cin;
    if (i >= n) return;
    const int ci = (int)(i % Cin), co = (int)((i / Cin) % Cout), tap = (int)(i / ((int64_t)Cin * Cout));
    const float a = w[i], b = w[i + 1];
    const uint32_t o0 = spl_pack(a, b);
    const float ra = a - spl_lo(o0), rb = b - spl_hi(o0);
    const uint32_t o1 = spl_pack(ra, rb);
    const uint32_t o2 = spl_pack(ra - spl_lo(o1), rb - spl_hi(o1));
    const uint32_t o[3] = {o0, o1, o2};
    const int steps = Cin / CBK;
    uint16_t* dst = out + (((int64_t)tap * steps + ci / CBK) * 3 * Cout + co) * CBK + (ci % CBK);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint32_t*>(dst + (int64_t)pl * Cout * CBK) = o[pl];
}

extern "C" int ndet_split_weights_bf16x3(const float* w_packed, int taps, int Cout, int Cin, uint16_t* planes, void* stream) {
    const char* fn = "ndet_split_weights_bf16x3";
    NDET_REQUIRE(w_packed && planes, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(taps > 0 && Cout > 0 && Cin > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE(Cin % CBK == 0, NDET_E_UNSUPPORTED, "%s: Cin=%d must be a multiple of %d", fn, Cin, CBK);
    const int64_t work = (int64_t)taps * Cout * Cin / 2;
    hipLaunchKernelGGL(k_split_weights, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_packed, taps, Cout, Cin, planes);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// packed fp32 weights (taps, Cout, Cin) x scale -> two fp16 planes tiled per K step, (taps, Cin/32, 2, Cout, 32): w*scale = hi + lo to 2^-22
__global__ __launch_bounds__(256) void k_split_weights_f16x2(const float* __restrict__ w, int taps, int Cout, int Cin, float scale, uint16_t* __restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const int64_t n = (int64_t)taps * Cout * Cin;
    if (i >= n) return;
    const int ci = (int)(i % Cin), co = (int)((i / Cin) % Cout), tap = (int)(i / ((int64_t)Cin * Cout));
    const float a = w[i] * scale, b = w[i + 1] * scale;
    const uint32_t o0 = spl_pack_f16(a, b);
    const uint32_t o1 = spl_pack_f16(a - spl_f16_lo(o0), b - spl_f16_hi(o0));
    const int steps = Cin / CBK;
    uint16_t* dst = out + (((int64_t)tap * steps + ci / CBK) * 2 * Cout + co) * CBK + (ci % CBK);
    *reinterpret_cast<uint32_t*>(dst) = o0;
    *reinterpret_cast<uint32_t*>(dst + (int64_t)Cout * CBK) = o1;
}

extern "C" int ndet_split_weights_f16x2(const float* w_packed, int taps, int Cout, int Cin, float scale, uint16_t* planes, void* stream) {
    const char* fn = "ndet_split_weights_f16x2";
    NDET_REQUIRE(w_packed && planes, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(taps > 0 && Cout > 0 && Cin > 0 && scale > 0.0f, NDET_E_INVALID, "%s: sizes and scale must be positive", fn);
    NDET_REQUIRE(Cin % CBK == 0, NDET_E_UNSUPPORTED, "%s: Cin=%d must be a multiple of %d", fn, Cin, CBK);
    const int64_t work = (int64_t)taps * Cout * Cin / 2;
    hipLaunchKernelGGL(k_split_weights_f16x2, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_packed, taps, Cout, Cin, scale,
                       planes);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// The same planes straight from a torch-layout weight (Cout, Cin, taps): `adjoint` = 0 packs the layer's own weight, 1 the weight of
// its data gradient, W'[t][ci][co] = W[co][ci][taps - 1 - t] (tap-flipped, transposed; the new input-channel count Cout is zero-padded
// to a multiple of 32).  Training re-packs every step (the optimizer moves the weights): one launch instead of permute + copy + split.
__global__ __launch_bounds__(256) void k_split_weights_torch(const float* __restrict__ w, int taps, int Cout, int Cin, int adjoint, int No, int Ki,
                                                             uint16_t* __restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;   // element pair along the packed input-channel axis
    const int64_t n = (int64_t)taps * No * Ki;
    if (i >= n) return;
    const int ki = (int)(i % Ki), no = (int)((i / Ki) % No), tap = (int)(i / ((int64_t)Ki * No));
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int k = ki + e;
        if (!adjoint) v[e] = w[((int64_t)no * Cin + k) * taps + tap];
        else v[e] = k < Cout ? w[((int64_t)k * Cin + no) * taps + (taps - 1 - tap)] : 0.0f;
    }
    const uint32_t o0 = spl_pack(v[0], v[1]);
    const float ra = v[0] - spl_lo(o0), rb = v[1] - spl_hi(o0);
    const uint32_t o1 = spl_pack(ra, rb);
    const uint32_t o2 = spl_pack(ra - spl_lo(o1), rb - spl_hi(o1));
    const uint32_t o[3] = {o0, o1, o2};
    const int steps = Ki / CBK;
    uint16_t* dst = out + (((int64_t)tap * steps + ki / CBK) * 3 * No + no) * CBK + (ki % CBK);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint32_t*>(dst + (int64_t)pl * No * CBK) = o[pl];
}

extern "C" int ndet_split_weights_bf16x3_torch(const float* w_torch, int taps, int Cout, int Cin, int adjoint, uint16_t* planes, void* stream) {
    const char* fn = "ndet_split_weights_bf16x3_torch";
    NDET_REQUIRE(w_torch && planes, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(taps > 0 && Cout > 0 && Cin > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    const int No = adjoint ? Cin : Cout, Ki = adjoint ? ((Cout + CBK - 1) / CBK) * CBK : Cin;
    NDET_REQUIRE(Ki % CBK == 0, NDET_E_UNSUPPORTED, "%s: Cin=%d must be a multiple of %d", fn, Cin, CBK);
    const int64_t work = (int64_t)taps * No * Ki / 2;
    hipLaunchKernelGGL(k_split_weights_torch, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_torch, taps, Cout, Cin,
                       adjoint ? 1 : 0, No, Ki, planes);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}


// Both packs of a TRAINING step's weight in one pass over it: the layer's own planes (taps, Cin/32, WPL, Cout, 32) and the planes of its data gradient
// W'[t][ci][co] = W[co][ci][taps - 1 - t], (taps, ceil(Cout/32), WPL, Cin, 32) (Cout zero-padded) -- SCH 0: three bf16 planes; SCH 1: two fp16 planes of
// w * conv_xscale(amax slot), the scale never leaving the device (ndet_conv_ndhwc_train reads its inverse from the same slot).  k_split_weights_torch
// reads the torch layout (Cout, Cin, taps) with a stride of `taps` floats between neighbouring lanes (a 1024 x 1024 x 27 weight: 559 us, 0.2 TB/s of
// reads) and runs once per pack; here a workgroup takes a 32 x 32 (co, ci) block with all its taps -- 32 contiguous runs of 32 taps floats -- through
// LDS, and every (tap, plane) of either pack leaves as one contiguous 2 KB piece.
template <int SCH>
__global__ __launch_bounds__(1024) void k_split_weights_train(const float* __restrict__ w, int taps, int Cout, int Cin, const float* __restrict__ amax,
                                                              uint16_t* __restrict__ fwd, uint16_t* __restrict__ adj) {
    constexpr int WPL = Spl<SCH>::WPL;
    extern __shared__ __attribute__((aligned(16))) float wt_tile[];      // [32 co][32 ci x taps + 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int run = 32 * taps, pitch = run + 1;
    const float xs = SCH == 1 ? conv_xscale(amax) : 1.0f;
    for (int r = wave; r < 32; r += 16) {
        const int co = co0 + r;
        const float* src = w + ((int64_t)co * Cin + ci0) * taps;
        for (int c = lane; c < run; c += 64) wt_tile[r * pitch + c] = co < Cout ? src[c] * xs : 0.0f;
    }
    __syncthreads();
    const int stepsF = Cin / CBK, stepsA = (Cout + CBK - 1) / CBK;
    // one item = an element pair of one (tap, row): 16 pairs per row, 32 rows, taps x 512 items; a wave covers 4 rows = 256 contiguous bytes per plane
    for (int item = tid; item < taps * 512; item += 1024) {
        const int tap = item >> 9, r = (item >> 4) & 31, cp = item & 15;
        uint32_t o[3];
        // the layer's own planes: row = output channel, pair along the input channels
        spl_split2<SCH>(wt_tile[r * pitch + (2 * cp) * taps + tap], wt_tile[r * pitch + (2 * cp + 1) * taps + tap], 1.0f, o[0], o[1], o[2]);
        if (co0 + r < Cout) {
            uint16_t* dst = fwd + (((int64_t)tap * stepsF + blockIdx.x) * WPL * Cout + (co0 + r)) * CBK + 2 * cp;
#pragma unroll
            for (int pl = 0; pl < WPL; ++pl) *reinterpret_cast<uint32_t*>(dst + (int64_t)pl * Cout * CBK) = o[pl];
        }
        // the data gradient's planes: row = input channel, pair along the (padded) output channels, taps flipped
        if (adj) {
            spl_split2<SCH>(wt_tile[(2 * cp) * pitch + r * taps + tap], wt_tile[(2 * cp + 1) * pitch + r * taps + tap], 1.0f, o[0], o[1], o[2]);
            uint16_t* dst = adj + (((int64_t)(taps - 1 - tap) * stepsA + blockIdx.y) * WPL * Cin + (ci0 + r)) * CBK + 2 * cp;
#pragma unroll
            for (int pl = 0; pl < WPL; ++pl) *reinterpret_cast<uint32_t*>(dst + (int64_t)pl * Cin * CBK) = o[pl];
        }
    }
}

// The weight gradient's last step: dw_rows ((tap, ci) rows x Cout, what the GEMMs write) -> torch's (Cout, Cin, taps) layout.  The same 32 x 32 x taps
// block through LDS as above, the other way round: 128-byte row pieces in, one contiguous run of 32 taps floats per output channel out.  (As
// `dw.view(taps, Cin, Cout).permute(2, 1, 0).reshape(...)` this was ATen's generic strided copy: ~2.4 ms of the training step for 108 M parameters.)
// `splits` > 1: `rows` holds that many split-K partials, taps * Cin * Cout floats apart, added here in index order (the order k_conv3d_splitk_reduce uses).
__global__ __launch_bounds__(1024) void k_wgrad_to_torch(const float* __restrict__ rows, int splits, int taps, int Cout, int Cin, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float wt_tile[];      // [32 co][32 ci x taps + 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int run = 32 * taps, pitch = run + 1;
    for (int item = tid; item < taps * 1024; item += 1024) {
        const int tap = item >> 10, r = (item >> 5) & 31, c = item & 31;       // r: input channel, c: output channel
        if (co0 + c < Cout) {
            const float* src = rows + ((int64_t)tap * Cin + ci0 + r) * Cout + co0 + c;
            const int64_t ps = (int64_t)taps * Cin * Cout;
            float v = src[0];
            // eight partials in flight, added in index order (a loop of load-then-add waits out one memory latency per partial: measured 28 us per
            // launch on average against 9 + 9 for the separate reduction and layout passes)
            for (int s0 = 1; s0 < splits; s0 += 8) {
                float t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = s0 + k < splits ? src[(int64_t)(s0 + k) * ps] : 0.0f;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (s0 + k < splits) v = v + t[k];
            }
            wt_tile[c * pitch + r * taps + tap] = v;
        }
    }
    __syncthreads();
    for (int r = wave; r < 32; r += 16) {
        const int co = co0 + r;
        if (co >= Cout) break;
        float* dst = out + ((int64_t)co * Cin + ci0) * taps;
        for (int c = lane; c < run; c += 64) dst[c] = wt_tile[r * pitch + c];
    }
}

extern "C" int ndet_wgrad_to_torch(const float* dw_rows, int splits, int taps, int Cout, int Cin, float* dw_torch, void* stream) {
    const char* fn = "ndet_wgrad_to_torch";
    NDET_REQUIRE(dw_rows && dw_torch && splits >= 1, NDET_E_INVALID, "%s: null pointer / splits < 1", fn);
    NDET_REQUIRE(taps > 0 && taps <= 27 && Cout > 0 && Cin > 0 && Cin % 32 == 0 && (Cout + 31) / 32 <= 65535, NDET_E_UNSUPPORTED, "%s: 1..27 taps, Cin %% 32 == 0", fn);
    const size_t lds = (size_t)32 * (32 * taps + 1) * sizeof(float);
    if (lds > 48 * 1024) {
        static bool attr_set[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            hipError_t e = hipFuncSetAttribute((const void*)k_wgrad_to_torch, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * (32 * 27 + 1) * 4);
            NDET_REQUIRE(e == hipSuccess, NDET_E_LAUNCH, "%s: cannot raise the LDS limit: %s", fn, hipGetErrorString(e));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL(k_wgrad_to_torch, dim3(Cin / 32, (Cout + 31) / 32), dim3(1024), lds, (hipStream_t)stream, dw_rows, splits, taps, Cout, Cin, dw_torch);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

extern "C" int ndet_split_weights_train(const float* w_torch, int taps, int Cout, int Cin, int arith, const float* w_amax, uint16_t* planes,
                                        uint16_t* planes_adjoint, void* stream) {
    const char* fn = "ndet_split_weights_train";
    NDET_REQUIRE(w_torch && planes, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(taps > 0 && taps <= 27 && Cout > 0 && Cin > 0, NDET_E_UNSUPPORTED, "%s: 1..27 taps, positive sizes", fn);
    NDET_REQUIRE(arith == 0 || arith == 1, NDET_E_INVALID, "%s: arith must be 0 (bf16x3) or 1 (fp16 pair)", fn);
    NDET_REQUIRE((arith == 1) == (w_amax != nullptr), NDET_E_INVALID, "%s: the amax slot belongs to the fp16-pair arithmetic", fn);
    NDET_REQUIRE(Cin % CBK == 0 && (Cout + CBK - 1) / CBK <= 65535, NDET_E_UNSUPPORTED, "%s: Cin=%d must be a multiple of %d", fn, Cin, CBK);
    const size_t lds = (size_t)32 * (32 * taps + 1) * sizeof(float);
    const void* kfn = arith == 1 ? (const void*)k_split_weights_train<1> : (const void*)k_split_weights_train<0>;
    if (lds > 48 * 1024) {                           // per device: a process may drive several (one rank per GPU is the rule, not a guarantee)
        static bool attr_set[2][64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_set[arith][dev]) {
            hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * (32 * 27 + 1) * 4);
            NDET_REQUIRE(e == hipSuccess, NDET_E_LAUNCH, "%s: cannot raise the LDS limit: %s", fn, hipGetErrorString(e));
            if (dev >= 0 && dev < 64) attr_set[arith][dev] = true;
        }
    }
    const dim3 grid(Cin / CBK, (Cout + CBK - 1) / CBK);
    if (arith == 1) hipLaunchKernelGGL(k_split_weights_train<1>, grid, dim3(1024), lds, (hipStream_t)stream, w_torch, taps, Cout, Cin, w_amax, planes, planes_adjoint);
    else hipLaunchKernelGGL(k_split_weights_train<0>, grid, dim3(1024), lds, (hipStream_t)stream, w_torch, taps, Cout, Cin, w_amax, planes, planes_adjoint);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

static int conv_split_entry(const char* fn, int max_order, const float* in_amax, float w_inv_scale, float* out_amax, const float* in, const uint16_t* w_planes,
                            float* out, int D, int H, int W, int Cin, int Cout, const int* kernel, const int* stride, const int* pad, int transposed,
                            const float* scale, const float* shift, const float* residual, int residual_up2, int relu, int splits, int tile,
                            void* workspace, void* stream);

extern "C" int ndet_conv_ndhwc_split(const float* in, const uint16_t* w_planes, float* out, int D, int H, int W, int Cin, int Cout,
                                     const int* kernel, const int* stride, const int* pad, int transposed, const float* scale,
                                     const float* shift, const float* residual, int residual_up2, int relu, int splits, int tile,
                                     void* workspace, void* stream) {
    return conv_split_entry("ndet_conv_ndhwc_split", 2, nullptr, 1.0f, nullptr, in, w_planes, out, D, H, W, Cin, Cout, kernel, stride, pad, transposed, scale, shift, residual,
                            residual_up2, relu, splits, tile, workspace, stream);
}

extern "C" int ndet_conv_ndhwc_bf16(const float* in, const uint16_t* w_planes, float* out, int D, int H, int W, int Cin, int Cout,
                                    const int* kernel, const int* stride, const int* pad, int transposed, const float* scale,
                                    const float* shift, const float* residual, int residual_up2, int relu, int splits, int tile,
                                    void* workspace, void* stream) {
    return conv_split_entry("ndet_conv_ndhwc_bf16", 0, nullptr, 1.0f, nullptr, in, w_planes, out, D, H, W, Cin, Cout, kernel, stride, pad, transposed, scale, shift, residual,
                            residual_up2, relu, splits, tile, workspace, stream);
}

struct ConvGuard { unsigned* flag; float l1, l1_3, tol; const float* w_amax; const float* map_w; const float* map_b; float* map_out; };
static thread_local ConvGuard g_guard = {nullptr, 0.0f, 0.0f, 0.0f, nullptr, nullptr, nullptr, nullptr};     // handed from the *_guarded / *_train entry points to the shared argument checks below (per call)

extern "C" int ndet_conv_ndhwc_guarded(const float* in, const uint16_t* w_planes, float* out, int D, int H, int W, int Cin, int Cout,
                                       const int* kernel, const int* stride, const int* pad, int transposed, const float* scale,
                                       const float* shift, const float* residual, int residual_up2, int relu, int splits, int tile, int arith,
                                       const float* in_amax, float w_inv_scale, float* out_amax, void* workspace, float guard_l1, float guard_tol,
                                       unsigned* guard, void* stream) {
    NDET_REQUIRE(!guard || (guard_l1 >= 0.0f && guard_tol > 0.0f), NDET_E_INVALID, "ndet_conv_ndhwc_guarded: the guard needs guard_l1 >= 0 and guard_tol > 0");
    g_guard = ConvGuard{arith == 1 ? guard : nullptr, guard_l1, 0.0f, guard_tol, nullptr};
    const int rc = ndet_conv_ndhwc_arith(in, w_planes, out, D, H, W, Cin, Cout, kernel, stride, pad, transposed, scale, shift, residual, residual_up2, relu, splits,
                                         tile, arith, in_amax, w_inv_scale, out_amax, workspace, stream);
    g_guard = ConvGuard{nullptr, 0.0f, 0.0f, 0.0f, nullptr};
    return rc;
}

// ndet_conv_ndhwc_guarded with a chained 32-channel projection of the output rows in the same launch (see Conv3dParams::map_out): only the
// halo-stationary tiles that own all Cout = 256 channels of their rows take it (tile 3256 / 3257 / 3258), without split-K, residual or ReLU.
extern "C" int ndet_conv_ndhwc_mapped(const float* in, const uint16_t* w_planes, float* out, int D, int H, int W, int Cin, int Cout,
                                      const int* kernel, const int* stride, const int* pad, const float* scale, const float* shift, int tile, int arith,
                                      const float* in_amax, float w_inv_scale, float* out_amax, float guard_l1, float guard_tol, unsigned* guard,
                                      const float* map_w, const float* map_b, float* map_out, void* stream) {
    const char* fn = "ndet_conv_ndhwc_mapped";
    NDET_REQUIRE(map_w && map_b && map_out, NDET_E_INVALID, "%s: null projection pointers", fn);
    NDET_REQUIRE((((uintptr_t)map_w | (uintptr_t)map_b | (uintptr_t)map_out) & 15) == 0, NDET_E_UNSUPPORTED, "%s: projection pointers must be 16-byte aligned", fn);
    NDET_REQUIRE(tile == 3256 || tile == 3257 || tile == 3258, NDET_E_UNSUPPORTED, "%s: tile %d does not own whole rows (3256 / 3257 / 3258 do)", fn, tile);
    NDET_REQUIRE(Cout == 256, NDET_E_UNSUPPORTED, "%s: Cout=%d, the 256-column tiles own whole rows of 256 channels only", fn, Cout);
    NDET_REQUIRE(!guard || (guard_l1 >= 0.0f && guard_tol > 0.0f), NDET_E_INVALID, "%s: the guard needs guard_l1 >= 0 and guard_tol > 0", fn);
    g_guard = ConvGuard{arith == 1 ? guard : nullptr, guard_l1, 0.0f, guard_tol, nullptr, map_w, map_b, map_out};
    const int rc = ndet_conv_ndhwc_arith(in, w_planes, out, D, H, W, Cin, Cout, kernel, stride, pad, 0, scale, shift, nullptr, 0, 0, 1, tile, arith, in_amax, w_inv_scale,
                                         out_amax, nullptr, stream);
    g_guard = ConvGuard{nullptr, 0.0f, 0.0f, 0.0f, nullptr, nullptr, nullptr, nullptr};
    return rc;
}

// The fp16-pair launch of the training step: the weight planes were scaled ON THE DEVICE by conv_xscale of the slot `w_amax` (ndet_split_weights_train:
// the optimizer moves the weights every step; ndet_wgrad_dy_planes_f16x2: the weight gradient's "weight" operand is dy), so the kernels take 1 / scale
// from the same slot.  `guard_k` = the contraction length (taps x Cin): the range guard bounds ||w||_1 by guard_k max|w|; guard null = no check.
extern "C" int ndet_conv_ndhwc_train(const float* in, const uint16_t* w_planes, float* out, int D, int H, int W, int Cin, int Cout,
                                     const int* kernel, const int* stride, const int* pad, const float* scale, const float* shift, const float* residual,
                                     int relu, int splits, int tile, const float* in_amax, const float* w_amax, float* out_amax, void* workspace,
                                     float guard_k, float guard_tol, unsigned* guard, int keep_partials, void* stream) {
    NDET_REQUIRE(w_amax != nullptr, NDET_E_INVALID, "ndet_conv_ndhwc_train: the weight planes' amax slot is required");
    NDET_REQUIRE(!keep_partials || (!scale && !residual && !relu && !out_amax), NDET_E_INVALID, "ndet_conv_ndhwc_train: keep_partials leaves the epilogue to ndet_wgrad_to_torch (no affine / residual / ReLU / amax)");
    NDET_REQUIRE(!guard || (guard_k >= 0.0f && guard_tol > 0.0f), NDET_E_INVALID, "ndet_conv_ndhwc_train: the guard needs guard_k >= 0 and guard_tol > 0");
    g_guard = ConvGuard{guard, guard_k, 0.0f, guard_tol, w_amax};
    g_keep_partials = keep_partials ? 1 : 0;
    const int rc = ndet_conv_ndhwc_arith(in, w_planes, out, D, H, W, Cin, Cout, kernel, stride, pad, 0, scale, shift, residual, 0, relu, splits, tile, 1, in_amax, 1.0f,
                                         out_amax, workspace, stream);
    g_keep_partials = 0;
    g_guard = ConvGuard{nullptr, 0.0f, 0.0f, 0.0f, nullptr};
    return rc;
}

extern "C" int ndet_conv_ndhwc_arith(const float* in, const uint16_t* w_planes, float* out, int D, int H, int W, int Cin, int Cout,
                                     const int* kernel, const int* stride, const int* pad, int transposed, const float* scale,
                                     const float* shift, const float* residual, int residual_up2, int relu, int splits, int tile, int arith,
                                     const float* in_amax, float w_inv_scale, float* out_amax, void* workspace, void* stream) {
    const char* fn = "ndet_conv_ndhwc_arith";
    NDET_REQUIRE(arith >= 0 && arith <= 2, NDET_E_INVALID, "%s: arith must be 0 (bf16x3), 1 (fp16 pair) or 2 (bf16)", fn);
    if (arith == 1) {
        NDET_REQUIRE(in_amax != nullptr && w_inv_scale > 0.0f, NDET_E_INVALID, "%s: the fp16-pair arithmetic needs the input's amax slot and the weight planes' inverse scale", fn);
    } else {
        NDET_REQUIRE(in_amax == nullptr, NDET_E_INVALID, "%s: in_amax belongs to the fp16-pair arithmetic only", fn);
    }
    return conv_split_entry(fn, arith == 0 ? 2 : (arith == 1 ? 1 : 0), in_amax, arith == 1 ? w_inv_scale : 1.0f, out_amax, in, w_planes, out, D, H, W, Cin, Cout,
                            kernel, stride, pad, transposed, scale, shift, residual, residual_up2, relu, splits, tile, workspace, stream);
}

// max |x| of a tensor into a zeroed slot: the amax_in of a fp16-pair convolution whose input was not written by one of the convolution kernels
__global__ __launch_bounds__(256) void k_amax(const float* __restrict__ x, int64_t n4, int64_t n, float* __restrict__ slot) {
    // every workgroup takes ONE contiguous range of the tensor: its maximum is the maximum of a region (a few rows / voxels), which is what the range
    // guard's tile minimum wants to see (conv_tilemin_read) -- a grid-stride walk would make every workgroup's maximum the tensor's
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    float mx = 0.0f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = n4 * 4; i < n; ++i) mx = fmaxf(mx, fabsf(x[i]));
    conv_amax_commit(slot, mx);
}

extern "C" int ndet_amax_f32(const float* x, int64_t n, float* slot, void* stream) {
    const char* fn = "ndet_amax_f32";
    NDET_REQUIRE(x && slot && n > 0, NDET_E_INVALID, "%s: null pointer / empty tensor", fn);
    NDET_REQUIRE(((uintptr_t)x & 15) == 0, NDET_E_UNSUPPORTED, "%s: x must be 16-byte aligned", fn);
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_amax, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n4, n, slot);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

static int conv_split_entry(const char* fn, int max_order, const float* in_amax, float w_inv_scale, float* out_amax, const float* in, const uint16_t* w_planes,
                            float* out, int D, int H, int W, int Cin, int Cout, const int* kernel, const int* stride, const int* pad, int transposed,
                            const float* scale, const float* shift, const float* residual, int residual_up2, int relu, int splits, int tile,
                            void* workspace, void* stream) {
    NDET_REQUIRE(in && w_planes && out && kernel && stride && pad, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE((scale == nullptr) == (shift == nullptr), NDET_E_INVALID, "%s: scale and shift go together", fn);
    NDET_REQUIRE(relu >= 0 && relu <= 2 && (tile == 0 || tile == 64 || tile == 128 || tile == 12864 || tile == 128256 || tile == 129256 || tile == 129257 || tile == 129064 || tile == 3128 || tile == 3256 || tile == 3257 || tile == 3258 ||
                                          tile == 100064 || tile == 100128 || tile == 112864), NDET_E_INVALID, "%s: bad relu mode / tile", fn);
    NDET_REQUIRE(Cin % CBK == 0, NDET_E_UNSUPPORTED, "%s: Cin=%d must be a multiple of %d", fn, Cin, CBK);
    NDET_REQUIRE((((uintptr_t)in | (uintptr_t)w_planes) & 15) == 0, NDET_E_UNSUPPORTED, "%s: in / weights must be 16-byte aligned", fn);
    Conv3dParams p;
    p.in = in; p.w = reinterpret_cast<const float*>(w_planes); p.out = out; p.scale = scale; p.shift = shift; p.res = residual;
    p.partial = (float*)workspace;
    p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
    p.max_order = max_order;
    p.amax_in = in_amax; p.winv = w_inv_scale; p.amax_out = out_amax;
    p.guard = g_guard.flag; p.guard_l1 = g_guard.l1; p.guard_tol = g_guard.tol; p.w_amax = g_guard.w_amax;
    p.map_w = g_guard.map_w; p.map_b = g_guard.map_b; p.map_out = g_guard.map_out;
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

template <int MID, int SCH>
static int chain_launch(const Conv3dParams& p, const ConvChain& c, hipStream_t st, const char* fn) {
    constexpr int NPL = Spl<SCH>::NPL;
    size_t lds = (size_t)NPL * (128 + MID) * SPL_RS * sizeof(uint16_t);
    const size_t y = (size_t)NPL * (MID == 128 ? 64 : 128) * MID * sizeof(uint16_t);
    if (y > lds) lds = y;
    if (lds > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)k_conv_split_chain<MID, SCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            NDET_REQUIRE(e == hipSuccess, NDET_E_LAUNCH, "%s: cannot raise the LDS limit: %s", fn, hipGetErrorString(e));
            attr_set = true;
        }
    }
    hipLaunchKernelGGL((k_conv_split_chain<MID, SCH>), dim3((p.M + 127) / 128), dim3(256), lds, st, p, (const uint16_t*)p.w, c);
    return NDET_OK;
}

extern "C" int ndet_conv_chain_arith(const float* in, const uint16_t* w_planes, int D, int H, int W, int Cin, int Cmid, const int* kernel,
                                     const int* stride, const int* pad, const float* scale1, const float* shift1, const uint16_t* w3_planes,
                                     int Cout, const float* scale3, const float* shift3, const float* residual, int relu3, float* out,
                                     int arith, const float* in_amax, float w1_inv_scale, float w3_inv_scale, float* out_amax, void* stream);

extern "C" int ndet_conv_chain_guarded(const float* in, const uint16_t* w_planes, int D, int H, int W, int Cin, int Cmid, const int* kernel,
                                       const int* stride, const int* pad, const float* scale1, const float* shift1, const uint16_t* w3_planes,
                                       int Cout, const float* scale3, const float* shift3, const float* residual, int relu3, float* out,
                                       int arith, const float* in_amax, float w1_inv_scale, float w3_inv_scale, float* out_amax, float guard_l1,
                                       float guard_l1_3, float guard_tol, unsigned* guard, void* stream) {
    NDET_REQUIRE(!guard || (guard_l1 >= 0.0f && guard_l1_3 >= 0.0f && guard_tol > 0.0f), NDET_E_INVALID, "ndet_conv_chain_guarded: the guard needs l1 >= 0 and tol > 0");
    g_guard = ConvGuard{arith == 1 ? guard : nullptr, guard_l1, guard_l1_3, guard_tol};
    const int rc = ndet_conv_chain_arith(in, w_planes, D, H, W, Cin, Cmid, kernel, stride, pad, scale1, shift1, w3_planes, Cout, scale3, shift3, residual, relu3, out,
                                         arith, in_amax, w1_inv_scale, w3_inv_scale, out_amax, stream);
    g_guard = ConvGuard{nullptr, 0.0f, 0.0f, 0.0f};
    return rc;
}

extern "C" int ndet_conv_chain_split(const float* in, const uint16_t* w_planes, int D, int H, int W, int Cin, int Cmid, const int* kernel,
                                     const int* stride, const int* pad, const float* scale1, const float* shift1, const uint16_t* w3_planes,
                                     int Cout, const float* scale3, const float* shift3, const float* residual, int relu3, float* out,
                                     int max_order, void* stream) {
    NDET_REQUIRE(max_order == 0 || max_order == 2, NDET_E_INVALID, "ndet_conv_chain_split: bad arithmetic");
    return ndet_conv_chain_arith(in, w_planes, D, H, W, Cin, Cmid, kernel, stride, pad, scale1, shift1, w3_planes, Cout, scale3, shift3, residual, relu3, out,
                                 max_order == 0 ? 2 : 0, nullptr, 1.0f, 1.0f, nullptr, stream);
}

extern "C" int ndet_conv_chain_arith(const float* in, const uint16_t* w_planes, int D, int H, int W, int Cin, int Cmid, const int* kernel,
                                     const int* stride, const int* pad, const float* scale1, const float* shift1, const uint16_t* w3_planes,
                                     int Cout, const float* scale3, const float* shift3, const float* residual, int relu3, float* out,
                                     int arith, const float* in_amax, float w1_inv_scale, float w3_inv_scale, float* out_amax, void* stream) {
    const char* fn = "ndet_conv_chain_arith";
    NDET_REQUIRE(arith >= 0 && arith <= 2, NDET_E_INVALID, "%s: arith must be 0 (bf16x3), 1 (fp16 pair) or 2 (bf16)", fn);
    NDET_REQUIRE((arith == 1) == (in_amax != nullptr), NDET_E_INVALID, "%s: in_amax goes with the fp16-pair arithmetic", fn);
    NDET_REQUIRE(arith != 1 || (w1_inv_scale > 0.0f && w3_inv_scale > 0.0f), NDET_E_INVALID, "%s: the fp16-pair arithmetic needs the weight planes' inverse scales", fn);
    const int max_order = arith == 0 ? 2 : (arith == 1 ? 1 : 0);
    NDET_REQUIRE(in && w_planes && w3_planes && out && kernel && stride && pad, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE((scale1 == nullptr) == (shift1 == nullptr) && (scale3 == nullptr) == (shift3 == nullptr), NDET_E_INVALID, "%s: scale and shift go together", fn);
    NDET_REQUIRE(Cmid == 64 || Cmid == 128, NDET_E_UNSUPPORTED, "%s: the intermediate must have 64 or 128 channels (got %d)", fn, Cmid);
    NDET_REQUIRE(Cin % CBK == 0 && Cout % 64 == 0, NDET_E_UNSUPPORTED, "%s: Cin=%d must be a multiple of %d, Cout=%d of 64", fn, Cin, CBK, Cout);
    NDET_REQUIRE(relu3 >= 0 && relu3 <= 2, NDET_E_INVALID, "%s: bad relu mode", fn);
    NDET_REQUIRE((((uintptr_t)in | (uintptr_t)w_planes | (uintptr_t)w3_planes) & 15) == 0, NDET_E_UNSUPPORTED, "%s: in / weights must be 16-byte aligned", fn);
    Conv3dParams p;
    p.in = in; p.w = reinterpret_cast<const float*>(w_planes); p.out = nullptr; p.scale = scale1; p.shift = shift1; p.res = nullptr; p.partial = nullptr;
    p.D = D; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cmid; p.relu = 1; p.max_order = max_order;
    p.amax_in = in_amax; p.winv = w1_inv_scale;
    p.guard = g_guard.flag; p.guard_l1 = g_guard.l1; p.guard_tol = g_guard.tol;
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
    NDET_REQUIRE(((int64_t)p.M + 128) * Cout * 4 < ((int64_t)1 << 32), NDET_E_UNSUPPORTED, "%s: the output is addressed with 32-bit byte offsets (< 4 GB)", fn);
    p.splits = 1; p.res_up2 = 0; p.RH = p.RW = 0;
    ConvChain c;
    c.w3 = w3_planes; c.scale3 = scale3; c.shift3 = shift3; c.res = residual; c.out = out; c.Cout3 = Cout; c.relu3 = relu3;
    c.w3inv = arith == 1 ? w3_inv_scale : 1.0f; c.amax_out = out_amax; c.guard_l1 = g_guard.l1_3;
    p.nt = (int64_t)p.M * Cout * 4 >= conv_nt_bytes() ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (Cmid == 64) rc = arith == 2 ? chain_launch<64, 2>(p, c, st, fn) : (arith == 1 ? chain_launch<64, 1>(p, c, st, fn) : chain_launch<64, 0>(p, c, st, fn));
    else rc = arith == 2 ? chain_launch<128, 2>(p, c, st, fn) : (arith == 1 ? chain_launch<128, 1>(p, c, st, fn) : chain_launch<128, 0>(p, c, st, fn));
    if (rc != NDET_OK) return rc;
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

template <int BM, int BN, int SCH>
static int wgrad_launch(const WgradParams& g, const Conv3dParams& p, const uint16_t* gplanes, hipStream_t st) {
    dim3 grid(p.M / BM, (p.Cout + BN - 1) / BN, p.splits);
    size_t lds = (size_t)Spl<SCH>::NPL * (BM + BN) * SPL_RS * sizeof(uint16_t);
    const size_t cs = (size_t)(BM / 2) * (BN + 4) * sizeof(float);
    if (cs > lds) lds = cs;
    hipLaunchKernelGGL((k_wgrad_split<BM, BN, SCH>), grid, dim3(256), lds, st, g, p, gplanes);
    return NDET_OK;
}

static int wgrad_split_entry(const char* fn, const float* x_ndhwc, int D, int H, int W, int Cin, const int* kernel, const int* stride, const int* pad,
                             const uint16_t* dy_planes, int Cout, int lrow, int splits, int max_order, const float* x_amax, const float* dy_amax, void* workspace,
                             float* dw_rows, int keep_partials, void* stream) {
    NDET_REQUIRE(x_ndhwc && kernel && stride && pad && dy_planes && dw_rows, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && lrow > 0 && lrow % CBK == 0, NDET_E_INVALID, "%s: bad sizes", fn);
    NDET_REQUIRE(Cin % 64 == 0, NDET_E_UNSUPPORTED, "%s: Cin=%d must be a multiple of 64", fn, Cin);
    NDET_REQUIRE(max_order == 0 || max_order == 1 || max_order == 2, NDET_E_INVALID, "%s: bad arithmetic", fn);
    NDET_REQUIRE((max_order == 1) == (x_amax != nullptr && dy_amax != nullptr), NDET_E_INVALID, "%s: the fp16-pair arithmetic needs both amax slots (the others none)", fn);
    NDET_REQUIRE((((uintptr_t)x_ndhwc | (uintptr_t)dy_planes | (uintptr_t)dw_rows) & 15) == 0, NDET_E_UNSUPPORTED, "%s: pointers must be 16-byte aligned", fn);
    WgradParams g;
    g.x = x_ndhwc; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.x_amax = x_amax; g.dy_amax = dy_amax;
    for (int a = 0; a < 3; ++a)
        NDET_REQUIRE(kernel[a] >= 1 && kernel[a] <= 7 && stride[a] >= 1 && stride[a] <= 4 && pad[a] >= 0 && pad[a] < kernel[a], NDET_E_UNSUPPORTED,
                     "%s: kernel/stride/pad out of range on axis %d", fn, a);
    g.kh = kernel[1]; g.kw = kernel[2];
    g.pd = pad[0]; g.ph = pad[1]; g.pw = pad[2]; g.sd = stride[0]; g.sh = stride[1]; g.sw = stride[2];
    g.OD = (D + 2 * pad[0] - kernel[0]) / stride[0] + 1;
    g.OH = (H + 2 * pad[1] - kernel[1]) / stride[1] + 1;
    g.OW = (W + 2 * pad[2] - kernel[2]) / stride[2] + 1;
    NDET_REQUIRE(g.OD > 0 && g.OH > 0 && g.OW > 0 && (int64_t)g.OD * g.OH * g.OW <= lrow && (int64_t)D * H * W * Cin < ((int64_t)1 << 40),
                 NDET_E_INVALID, "%s: dy rows (%d) shorter than the output grid", fn, lrow);
    g.L = g.OD * g.OH * g.OW;
    g.ksteps = lrow / CBK;
    const int taps = kernel[0] * kernel[1] * kernel[2];
    Conv3dParams p;
    p.in = nullptr; p.w = nullptr; p.out = dw_rows; p.scale = nullptr; p.shift = nullptr; p.res = nullptr; p.partial = (float*)workspace;
    p.D = p.H = p.W = 1; p.Cin = lrow; p.Cout = Cout; p.OD = p.OH = 1; p.OW = taps * Cin;
    p.kd = p.kh = p.kw = 1; p.sd = p.sh = p.sw = 1; p.pd = p.ph = p.pw = 0;
    p.relu = 0; p.transposed = 0; p.M = taps * Cin; p.res_up2 = 0; p.RH = p.RW = 0; p.max_order = max_order;
    p.splits = splits < 1 ? 1 : splits;
    NDET_REQUIRE(p.splits <= g.ksteps, NDET_E_INVALID, "%s: splits=%d exceeds the %d K steps", fn, p.splits, g.ksteps);
    NDET_REQUIRE(p.splits == 1 || workspace != nullptr, NDET_E_INVALID, "%s: split-K needs a workspace", fn);
    g.xcd_order = (g_wgrad_xcd && p.splits >= 8 && p.splits % 8 == 0) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const bool wide = Cout > 64, big = Cin % 128 == 0;
    const int sch = max_order == 0 ? 2 : (max_order == 1 ? 1 : 0);
    int rc;
#define NDET_WGRAD_TILE(BM, BN) (sch == 2 ? wgrad_launch<BM, BN, 2>(g, p, dy_planes, st) : (sch == 1 ? wgrad_launch<BM, BN, 1>(g, p, dy_planes, st) : wgrad_launch<BM, BN, 0>(g, p, dy_planes, st)))
    // 256 output channels per workgroup where the layer has them (fp16 pairs; the 3-plane arithmetic's LDS tiles would not leave room for two
    // workgroups per CU): the kernel is bound by what it pulls through L2 -- every (tap, channel tile) re-reads dy, every column tile re-reads x;
    // measured on the neck's 27-tap 256 -> 256 layer at 128 x 128: 2.8 GB per launch, 6.2 TB/s -- and the wider tile halves the x side
    // (layers with fewer than 32 row tiles keep 128 x 128: the FPN's 9-tap 256 -> 256 layer over 192 000 pixels has 18, and with the K splits capped
    // at 32 the wide tile leaves it 576 workgroups -- measured 1 413 us against 1 247 us)
    if (big && sch == 1 && Cout % 256 == 0 && taps * (Cin / 128) >= 32 && g_wgrad_wide) {
        static bool attr_set[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            hipError_t e = hipFuncSetAttribute((const void*)k_wgrad_split<128, 256, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (256 + 4) * 4);
            NDET_REQUIRE(e == hipSuccess, NDET_E_LAUNCH, "%s: cannot raise the LDS limit: %s", fn, hipGetErrorString(e));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
        rc = wgrad_launch<128, 256, 1>(g, p, dy_planes, st);
    }
    else if (big && wide) rc = NDET_WGRAD_TILE(128, 128);
    else if (big) rc = NDET_WGRAD_TILE(128, 64);
    else if (wide) rc = NDET_WGRAD_TILE(64, 128);
    else rc = NDET_WGRAD_TILE(64, 64);
#undef NDET_WGRAD_TILE
    if (rc != NDET_OK) return rc;
    NDET_CHECK_LAUNCH(fn);
    if (keep_partials && p.splits > 1) return NDET_OK;
    return conv_splitk_reduce_launch(p, st, fn);
}

extern "C" int ndet_wgrad_split(const float* x_ndhwc, int D, int H, int W, int Cin, const int* kernel, const int* stride, const int* pad,
                                const uint16_t* dy_planes, int Cout, int lrow, int splits, int max_order, void* workspace, float* dw_rows,
                                void* stream) {
    NDET_REQUIRE(max_order == 0 || max_order == 2, NDET_E_INVALID, "ndet_wgrad_split: max_order 0 (bf16) or 2 (bf16x3); the fp16-pair form is ndet_wgrad_split_f16x2");
    return wgrad_split_entry("ndet_wgrad_split", x_ndhwc, D, H, W, Cin, kernel, stride, pad, dy_planes, Cout, lrow, splits, max_order, nullptr, nullptr, workspace, dw_rows, 0, stream);
}

// The same implicit GEMM in the fp16-pair arithmetic: dy_planes from ndet_wgrad_dy_planes_f16x2 (two planes per K step, scaled by its slot), x split in
// the kernel under the scale of x_amax, three products.
extern "C" int ndet_wgrad_split_f16x2(const float* x_ndhwc, int D, int H, int W, int Cin, const int* kernel, const int* stride, const int* pad,
                                      const uint16_t* dy_planes, int Cout, int lrow, int splits, const float* x_amax, const float* dy_amax, void* workspace,
                                      float* dw_rows, int keep_partials, void* stream) {
    return wgrad_split_entry("ndet_wgrad_split_f16x2", x_ndhwc, D, H, W, Cin, kernel, stride, pad, dy_planes, Cout, lrow, splits, 1, x_amax, dy_amax, workspace, dw_rows,
                             keep_partials, stream);
}

