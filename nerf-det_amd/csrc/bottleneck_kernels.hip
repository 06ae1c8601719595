// A whole ResNet bottleneck of stage 1 in ONE launch (mmdet's Bottleneck.forward behind mmdet3d/models/detectors/nerfdet.py:140, style 'pytorch'):
//
//     out = relu( bn3(W3 . relu(bn2(conv3x3(relu(bn1(W1 . x)))))) + identity ),     identity = x   or   bnD(WD . x)   (first block of the stage)
//
// Before this kernel the block was two launches (the 1x1 reduction conv1, then conv2 -> conv3 chained, k_conv_split_chain) -- three for the first
// block (+ the 1x1 downsample): the 256-channel input was read twice (conv1, residual), the 64-channel intermediate written and read back with
// its halo: 860 MB moved per block at cfg2 where 491 MB are compulsory (the memory-bound convolutions, the largest pool of time in
// the step).  Here a workgroup owns a 4 x 16 patch of output pixels of one view from x to out:
//
//   A  conv1 on the patch PLUS its one-pixel halo (6 x 18 = 108 pixels, 4 MFMA row tiles): x rows -> fp16 (hi, lo) planes in LDS chunk by chunk
//      (the staging scheme of conv_split_mainloop: global -> registers -> split -> LDS, next chunk in flight), BN1 + ReLU, pixels outside the
//      image forced to 0 (conv2 pads ITS input with zeros, not with relu(bn1(0)));
//   B  the 64-channel halo image split into LDS with the workgroup's own power-of-two scale (it is multiplied here and nowhere else);
//   C  conv2: the nine taps multiply out of that image at constant row offsets, W2 fragments stream from L2 one tap ahead;
//   D  BN2 + ReLU, split into LDS (scale: the workgroup's own maximum again);
//   E  conv3 as in k_conv_split_chain: each wave owns 32-column slices, W3 fragments in registers, epilogue in the MFMA's C layout with buffer
//      operations; the identity is read from x (L2: this workgroup fetched those rows in phase A) -- or, for the first block of the stage,
//      computed by a fourth GEMM on the patch's own x rows kept in LDS from phase A (WD: 64 -> 256).
//
// The price: conv1 is evaluated on 128 rows per 64 output pixels (2x; the patch is small so that two workgroups fit a CU: 62 KB of LDS).
// The kernel is a chain of L2 / HBM round trips (8 chunks of x, 9 taps of W2, W3 slices, identity rows: ~1 us each against 0.2 us of MFMA work per
// step -- s_memtime stamps of the first form: 31 us per workgroup, 5x its matrix time), so every phase keeps the next THREE steps' loads in flight
// in statically indexed register rings (static_for: the indices are template constants, nothing the optimiser has to prove), and phase E's weight
// fragments and identity rows are requested before phase D starts.
// fp16-pair arithmetic only (the six-product bf16x3 mode keeps the two-launch path): three v_mfma_f32_32x32x16_f16 products per multiply.
#include "spl_common.hpp"

#define BT_TH 4
#define BT_TW 16
#define BT_HW (BT_TW + 2)
#define BT_HROWS ((BT_TH + 2) * BT_HW)   // 108 halo pixels
#define BT_AROWS 128                     // ... padded to four 32-row MFMA tiles
#define BT_MID 64
#define BT_OROWS (BT_TH * BT_TW)         // 64 output pixels

struct BottleneckParams {
    const float* x;         // (N, H, W, Cin) channels-last
    float* out;             // (N, H, W, Cout)
    int N, H, W, Cin, Cout;
    int tiles_x, tiles_y;
    const uint16_t* w1;     // fp16-pair planes (1, Cin/32, 2, 64, 32)
    const uint16_t* w2;     // (9, 2, 2, 64, 32)
    const uint16_t* w3;     // (1, 2, 2, Cout, 32)
    const uint16_t* wd;     // (1, Cin/32, 2, Cout, 32) or null (identity residual)
    float w1inv, w2inv, w3inv, wdinv;
    const float *s1, *b1, *s2, *b2, *s3, *b3, *sd, *bd;     // folded BatchNorm (eval): per-channel scale / shift
    const float* amax_in;   // x's amax slot
    float* amax_out;        // out's amax slot or null
    unsigned* guard;        // range guard word or null (conv_common.hpp::conv_guard_check)
    float g1, g2, g3, gd, gtol;
    int nt;                 // non-temporal stores for the output
};

#include <type_traits>
template <int I> using ic = std::integral_constant<int, I>;
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(ic<I>{});
        static_for<I + 1, N>(f);
    }
}

#ifdef BT_STAMPS      // diagnostic builds (tools/diag/bottleneck_stamps.py): thread 0 of every workgroup records the 100 MHz clock at the phase boundaries
__device__ unsigned long long* g_bt_stamps = nullptr;
#define STAMP(i) do { if (g_bt_stamps && tid == 0) g_bt_stamps[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
extern "C" int ndet_bt_set_stamps(unsigned long long* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_bt_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -3; }
#else
#define STAMP(i)
#endif

template <bool DS, int NCH>
__global__ __launch_bounds__(256, 2) void k_bottleneck_f16x2(const BottleneckParams p) {
    constexpr int APL = BT_AROWS * SPL_RS, BPL = BT_MID * SPL_RS;      // staging planes, rows padded to 40 elements
    constexpr int STAGE = 2 * (APL + BPL);
    constexpr int Y1PL = BT_AROWS * BT_MID, Y2PL = BT_OROWS * BT_MID;
    constexpr int NCB = 2;                     // 32-column slices of the output per wave: Cout = 4 waves x NCB x 32 = 256
    extern __shared__ __attribute__((aligned(16))) uint16_t lds16[];
    uint16_t* As = lds16;                      // [2][128][40]
    uint16_t* Bs = lds16 + 2 * APL;            // [2][64][40]
    uint16_t* Y1 = lds16 + STAGE;              // [2][128][64]: the halo image of relu(bn1(conv1 x)), 16-byte chunk index XORed with row & 7
    uint16_t* Y2 = lds16;                      // [2][64][64]: relu(bn2(conv2)), over the staging planes (free after phase A)
    uint16_t* XO = Y1 + 2 * Y1PL;              // DS: [2][64][64] the patch's own x rows (Cin = 64), A operand of the downsample GEMM
    __shared__ float wg_max[4];
    __shared__ unsigned char rowok[BT_AROWS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5, fk = fh * 8;
    const int t = ndet_xcd_remap(blockIdx.x, gridDim.x);      // neighbouring patches share halo rows: keep them on one XCD's L2
    const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, n = t / (p.tiles_x * p.tiles_y);
    const int y0 = ty * BT_TH, x0 = tx * BT_TW;
    const float amax_in = conv_amax_read(p.amax_in);
    const float xs = conv_xscale_of(amax_in);
    if (p.guard && blockIdx.x == 0) {          // range guard of conv1 (and the downsample branch) on x: conv_common.hpp::conv_guard_check's two conditions
        const float tmin = conv_tilemin_read(p.amax_in);
        const float g = DS ? fmaxf(p.g1, p.gd) : p.g1;
        if (tid == 0 && amax_in * g * 0x1p-39f > p.gtol && tmin < amax_in * 0x1p-16f) atomicOr(p.guard, 1u);
    }

    if (tid < BT_AROWS) {
        const int hy = tid / BT_HW, hx = tid - hy * BT_HW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        rowok[tid] = (tid < BT_HROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? 1 : 0;
    }
    STAMP(0);

    // ---------------- phase A: conv1 (1x1, Cin -> 64) on the 128 halo rows ----------------
    const int wm = wave >> 1, wn = wave & 1;         // 2 x 2 waves: 64 rows x 32 channels each
    f32x16 acc1[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[a][r] = 0.f;
    {
        const int akq = tid & 7, arow_ = tid >> 3;       // A: 4-channel quad, first row (rows arow_ + 32 i)
        const int bkg = tid & 3, brow_ = tid >> 2;       // B: 8-k octet, row (64 rows)
        const float* rowptr[4];
        bool aok[4];
        int own[4];                                        // DS: index of the row among the patch's own 64 pixels, or -1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hr = arow_ + 32 * i;
            const int hy = hr / BT_HW, hx = hr - hy * BT_HW;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            aok[i] = hr < BT_HROWS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            rowptr[i] = p.x + (((int64_t)n * p.H + (aok[i] ? iy : 0)) * p.W + (aok[i] ? ix : 0)) * p.Cin + akq * 4;
            own[i] = (hr < BT_HROWS && hy >= 1 && hy <= BT_TH && hx >= 1 && hx <= BT_TW) ? (hy - 1) * BT_TW + (hx - 1) : -1;
        }
        const uint16_t* w1p = p.w1 + (int64_t)brow_ * CBK + bkg * 8;
        constexpr int RING = NCH < 4 ? NCH : 4;           // chunk c lives in register set c % RING; RING - 1 chunks travel while one multiplies
        typedef unsigned bt_u32x4 __attribute__((ext_vector_type(4)));
        f32x4_ ra[RING][4];                               // (native vector types: arrays of HIP's struct-wrapped uint4 are left on the stack -- every
        bt_u32x4 rb[RING][2];                             //  access a scratch round trip behind a vmcnt(0) that drains the whole ring)
        auto load_tile = [&](auto S, int ch) {
            constexpr int s = decltype(S)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4_ v = *reinterpret_cast<const f32x4_*>(rowptr[i] + ch * CBK);     // always issued (a legal address): no divergence
                ra[s][i] = aok[i] ? v : (f32x4_){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) rb[s][pl] = *reinterpret_cast<const bt_u32x4*>(w1p + ((int64_t)ch * 2 + pl) * BT_MID * CBK);
        };
        auto store_tile = [&](auto S, int ch) {
            constexpr int s = decltype(S)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint2 s0, s1, s2;
                spl_split<1>(make_float4(ra[s][i].x, ra[s][i].y, ra[s][i].z, ra[s][i].w), xs, s0, s1, s2);
                uint16_t* dst = As + (arow_ + 32 * i) * SPL_RS + akq * 4;
                *reinterpret_cast<uint2*>(dst) = s0;
                *reinterpret_cast<uint2*>(dst + APL) = s1;
                if (DS && own[i] >= 0) {                   // the downsample GEMM's operand: the patch's own rows, all Cin = 64 channels
                    uint16_t* xo = XO + own[i] * BT_MID + ((((ch * 4 + (akq >> 1)) ^ (own[i] & 7))) << 3) + (akq & 1) * 4;
                    *reinterpret_cast<uint2*>(xo) = s0;
                    *reinterpret_cast<uint2*>(xo + Y2PL) = s1;
                }
            }
            uint16_t* dst = Bs + brow_ * SPL_RS + bkg * 8;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<bt_u32x4*>(dst + pl * BPL) = rb[s][pl];
        };
        static_for<0, RING>([&](auto C) { load_tile(C, decltype(C)::value); });
        store_tile(ic<0>{}, 0);
        __syncthreads();
        const uint16_t* abase = As + (wm * 64 + frow) * SPL_RS + fk;
        const uint16_t* bbase = Bs + (wn * 32 + frow) * SPL_RS + fk;
        static_for<0, NCH>([&](auto C) {
            constexpr int ch = decltype(C)::value;
            if constexpr (ch + RING < NCH) load_tile(ic<ch % RING>{}, ch + RING);       // into the set chunk ch just left
#pragma unroll
            for (int ks = 0; ks < CBK / 16; ++ks) {
                bf16x8 fa[2][2], fb[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) fa[pl][a] = *reinterpret_cast<const bf16x8*>(abase + pl * APL + a * 32 * SPL_RS + ks * 16);
                    fb[pl] = *reinterpret_cast<const bf16x8*>(bbase + pl * BPL + ks * 16);
                }
                // smallest terms first: hi lo, lo hi, then hi hi
#pragma unroll
                for (int a = 0; a < 2; ++a) acc1[a] = spl_mfma32<1>(fa[0][a], fb[1], acc1[a]);
#pragma unroll
                for (int a = 0; a < 2; ++a) acc1[a] = spl_mfma32<1>(fa[1][a], fb[0], acc1[a]);
#pragma unroll
                for (int a = 0; a < 2; ++a) acc1[a] = spl_mfma32<1>(fa[0][a], fb[0], acc1[a]);
            }
            __syncthreads();
            if constexpr (ch + 1 < NCH) store_tile(ic<(ch + 1) % RING>{}, ch + 1);
            __syncthreads();
        });
    }
    STAMP(1);

    // conv2's weight fragments (one output channel per lane, 4 K slices x 2 planes per tap) stream from L2 three taps ahead: a ring of four sets.
    // The first three are requested here, before phase B: they land while the halo image is written.
    const int wm2 = wave >> 1, wn2 = wave & 1;       // phases C / D: tap group (C) and finished pixel tile (D), channel half
    bf16x8 fw[4][4][2];                               // [ring set][k slice][plane]
    const uint16_t* w2p = p.w2 + (int64_t)(wn2 * 32 + frow) * CBK + fk;
    auto load_w2 = [&](auto S, int tap) {
        constexpr int s = decltype(S)::value;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                fw[s][ks][pl] = *reinterpret_cast<const bf16x8*>(w2p + ((((int64_t)tap * 2 + (ks >> 1)) * 2 + pl) * BT_MID) * CBK + (ks & 1) * 16);
    };
    static_for<0, 3>([&](auto T) { load_w2(T, wm2 * 5 + decltype(T)::value); });      // the first three taps of this wave's tap group (phase C)

    // workgroup maximum of a wave-level value (uniform result; a barrier inside)
    auto wg_maximum = [&](float v) -> float {
#pragma unroll
        for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
        __syncthreads();                 // the previous round's readers are done with wg_max
        if (lane == 0) wg_max[wave] = v;
        __syncthreads();
        return fmaxf(fmaxf(wg_max[0], wg_max[1]), fmaxf(wg_max[2], wg_max[3]));
    };

    // ---------------- phase B: BN1 + ReLU (0 outside the image), the halo image into LDS with the workgroup's own scale ----------------
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float ymax1;
    {
        const float osc1 = conv_xinv_of(amax_in) * p.w1inv;
        const int col = wn * 32 + (lane & 31);
        const float sc = p.s1[col], sh = p.b1[col];
        float m = 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                float v = fmaxf((acc1[a][r] * osc1) * sc + sh, 0.f);
                v = rowok[row] ? v : 0.0f;
                acc1[a][r] = v;
                m = fmaxf(m, v);
            }
        ymax1 = wg_maximum(m);
        const float ys1 = conv_xscale_of(ymax1);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int row = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                uint16_t* da = Y1 + row * BT_MID + ((((col >> 3) ^ (row & 7)) << 3) | (col & 7));
                uint16_t* db = Y1 + (row + 1) * BT_MID + ((((col >> 3) ^ ((row + 1) & 7)) << 3) | (col & 7));
                uint32_t o0, o1 = 0, o2 = 0;
                spl_split2<1>(acc1[a][r], acc1[a][r + 1], ys1, o0, o1, o2);
                da[0] = (uint16_t)o0; db[0] = (uint16_t)(o0 >> 16);
                da[Y1PL] = (uint16_t)o1; db[Y1PL] = (uint16_t)(o1 >> 16);
            }
    }
    __syncthreads();
    STAMP(2);

    // ---------------- phase C: conv2 (3x3, 64 -> 64) out of the halo image ----------------
    // A wave owns 32 output channels (wn2) and HALF OF THE TAPS (group tg: taps 0-4 or 5-8) for all 64 pixels: every W2 fragment is fetched by
    // exactly one wave (with 32 pixels x all taps per wave each fragment was fetched twice: 295 KB of the 680 KB a patch pulled from L2).  The
    // two waves of a channel half then swap the partial sums of the pixel tile the other one finishes.
    const int tg = wm2;
    f32x16 acc2[2];                                   // the two 32-pixel tiles
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[a][r] = 0.f;
    {
        static_for<0, 5>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const int tap = tg * 5 + i;                               // wave-uniform
            if (tap < 9) {
                if (i + 3 < 5 && tap + 3 < 9) load_w2(ic<(i + 3) % 4>{}, tap + 3);
                const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int po = a * 32 + frow;
                        const int irow = ((po >> 4) + dy) * BT_HW + (po & 15) + dx;
                        bf16x8 fa[2];
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl)
                            fa[pl] = *reinterpret_cast<const bf16x8*>(Y1 + pl * Y1PL + irow * BT_MID + (((2 * ks + fh) ^ (irow & 7)) << 3));
                        acc2[a] = spl_mfma32<1>(fa[0], fw[i % 4][ks][1], acc2[a]);
                        acc2[a] = spl_mfma32<1>(fa[1], fw[i % 4][ks][0], acc2[a]);
                        acc2[a] = spl_mfma32<1>(fa[0], fw[i % 4][ks][0], acc2[a]);
                    }
                }
            }
        });
        // swap: this wave finishes pixel tile tg; its partial sums of the other tile go to the partner (same channel half, other tap group)
        float* xch = reinterpret_cast<float*>(lds16);                  // 4 waves x 16 registers x 64 lanes (16 KB over the staging planes; Y2 comes later)
        const int partner = (1 - tg) * 2 + wn2;
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(partner * 16 + r) * 64 + lane] = tg ? acc2[0][r] : acc2[1][r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[0][r] = (tg ? acc2[1][r] : acc2[0][r]) + xch[(wave * 16 + r) * 64 + lane];
    }
    STAMP(3);

    // phase E's operands, requested now: the W3 (and WD) fragments of both column slices of this wave and the identity rows of the first one
    // travel while BN2 + ReLU + the split of phase D run.
    const unsigned obytes = (unsigned)((int64_t)p.N * p.H * p.W * p.Cout * 4);
    const unsigned xbytes = (unsigned)((int64_t)p.N * p.H * p.W * p.Cin * 4);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, xbytes, 0x00020000);
    bf16x8 f3[NCB][4][2];
    auto load_frags = [&](auto CB, const uint16_t* wp) {
        constexpr int c = decltype(CB)::value;
        const int co = wave * 32 + c * 128 + frow;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                f3[c][ks][pl] = *reinterpret_cast<const bf16x8*>(wp + (((int64_t)(ks >> 1) * 2 + pl) * p.Cout + co) * CBK + (ks & 1) * 16 + fk);
    };
    static_for<0, NCB>([&](auto CB) { load_frags(CB, p.w3); });
    // byte offset of (row tile rt, register r) of this lane's column in slice 0: pixel (y0 + py, x0 + px), channel wave * 32 + frow
    unsigned off[2][16];
    {
        const unsigned vo = (unsigned)(((((int64_t)n * p.H + y0) * p.W + x0 + 4 * fh) * p.Cout + wave * 32 + frow) * 4);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = (r & 3) + 8 * (r >> 2);             // row inside the 32-row tile, without the lane's 4 fh
                const int py = rt * 2 + (q >> 4), px = (q & 15) + 4 * fh;
                const bool ok = y0 + py < p.H && x0 + px < p.W;
                off[rt][r] = ok ? vo + (unsigned)((py * p.W + (q & 15)) * p.Cout * 4) : 0xfffffff0u;      // outside the descriptor: reads 0, stores dropped
            }
    }
    float rr[2][16];                                  // identity rows of the slice about to be finished (Cin == Cout: the same byte offsets in x)
    auto load_identity = [&](int cb) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                rr[rt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rres, off[rt][r] == 0xfffffff0u ? 0xfffffff0u : off[rt][r] + (unsigned)(cb * 128 * 4), 0, 0));
    };
    if (!DS) load_identity(0);

    // ---------------- phase D: BN2 + ReLU, into LDS as the A operand of conv3 ----------------
    float ymax2;
    {
        const float osc2 = conv_xinv_of(ymax1) * p.w2inv;
        const int col = wn2 * 32 + (lane & 31);
        const float sc = p.s2[col], sh = p.b2[col];
        float m = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = fmaxf((acc2[0][r] * osc2) * sc + sh, 0.f);
            acc2[0][r] = v;
            m = fmaxf(m, v);
        }
        ymax2 = wg_maximum(m);           // (its barriers: every wave has left phase A's staging planes, which Y2 overwrites)
        const float ys2 = conv_xscale_of(ymax2);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const int row = wm2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            uint16_t* da = Y2 + row * BT_MID + ((((col >> 3) ^ (row & 7)) << 3) | (col & 7));
            uint16_t* db = Y2 + (row + 1) * BT_MID + ((((col >> 3) ^ ((row + 1) & 7)) << 3) | (col & 7));
            uint32_t o0, o1 = 0, o2 = 0;
            spl_split2<1>(acc2[0][r], acc2[0][r + 1], ys2, o0, o1, o2);
            da[0] = (uint16_t)o0; db[0] = (uint16_t)(o0 >> 16);
            da[Y2PL] = (uint16_t)o1; db[Y2PL] = (uint16_t)(o1 >> 16);
        }
    }
    __syncthreads();
    STAMP(4);

    // ---------------- phase E: conv3 (1x1, 64 -> 256) + BN3 + identity + ReLU, two 32-column slices per wave ----------------
    float omax = 0.0f;
    {
        const float osc3 = conv_xinv_of(ymax2) * p.w3inv;
        const float oscd = DS ? conv_xinv_of(amax_in) * p.wdinv : 0.0f;
        auto gemm = [&](auto CB, f32x16 (&cc)[2], const uint16_t* img) {
            constexpr int c = decltype(CB)::value;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) cc[rt][r] = 0.f;
                const int arow = rt * 32 + frow;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    bf16x8 fa[2];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) fa[pl] = *reinterpret_cast<const bf16x8*>(img + pl * Y2PL + arow * BT_MID + (((2 * ks + fh) ^ (arow & 7)) << 3));
                    cc[rt] = spl_mfma32<1>(fa[0], f3[c][ks][1], cc[rt]);
                    cc[rt] = spl_mfma32<1>(fa[1], f3[c][ks][0], cc[rt]);
                    cc[rt] = spl_mfma32<1>(fa[0], f3[c][ks][0], cc[rt]);
                }
            }
        };
        f32x16 c2[NCB][2];
        static_for<0, NCB>([&](auto CB) { gemm(CB, c2[decltype(CB)::value], Y2); });
        if (DS) static_for<0, NCB>([&](auto CB) { load_frags(CB, p.wd); });      // the identity branch of the stage's first block: WD's fragments in W3's place
        static_for<0, NCB>([&](auto CB) {
            constexpr int c = decltype(CB)::value;
            const int co = wave * 32 + c * 128 + frow;
            const float sc3 = p.s3[co], sh3 = p.b3[co];
            if (DS) {
                const float scd = p.sd[co], shd = p.bd[co];
                f32x16 cd[2];
                gemm(CB, cd, XO);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) rr[rt][r] = (cd[rt][r] * oscd) * scd + shd;
            }
            float vv[2][16];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) vv[rt][r] = fmaxf(((c2[c][rt][r] * osc3) * sc3 + sh3) + rr[rt][r], 0.f);
            if constexpr (c + 1 < NCB) { if (!DS) load_identity(c + 1); }      // (rr is consumed: the next slice's identity rows travel under these stores)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool ok = off[rt][r] != 0xfffffff0u;
                    const unsigned o = ok ? off[rt][r] + (unsigned)(c * 128 * 4) : 0xfffffff0u;
                    if (p.nt) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vv[rt][r]), rout, o, 0, 2);
                    else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vv[rt][r]), rout, o, 0, 0);
                    if (ok) omax = fmaxf(omax, fabsf(vv[rt][r]));
                }
        });
    }
    STAMP(5);
    if (p.amax_out) conv_amax_commit(p.amax_out, omax);
    STAMP(6);
}

extern "C" int ndet_bottleneck_f16x2(const float* x, int N, int H, int W, int Cin, int Cout, const uint16_t* w1_planes, float w1_inv_scale, const float* scale1,
                                     const float* shift1, const uint16_t* w2_planes, float w2_inv_scale, const float* scale2, const float* shift2,
                                     const uint16_t* w3_planes, float w3_inv_scale, const float* scale3, const float* shift3, const uint16_t* wd_planes,
                                     float wd_inv_scale, const float* scale_d, const float* shift_d, const float* in_amax, float* out_amax, float* out,
                                     const float* guard_l1_host, float guard_tol, unsigned* guard, void* stream) {
    const char* fn = "ndet_bottleneck_f16x2";
    NDET_REQUIRE(x && out && w1_planes && w2_planes && w3_planes && scale1 && shift1 && scale2 && shift2 && scale3 && shift3 && in_amax, NDET_E_INVALID,
                 "%s: null pointer", fn);
    NDET_REQUIRE(N > 0 && H > 0 && W > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE((Cin == 64 || Cin == 256) && Cout == 256, NDET_E_UNSUPPORTED, "%s: built for Cin = 64 or 256 and Cout = 256 (got %d, %d)", fn, Cin, Cout);
    NDET_REQUIRE(w1_inv_scale > 0.0f && w2_inv_scale > 0.0f && w3_inv_scale > 0.0f, NDET_E_INVALID, "%s: the weight planes' inverse scales must be positive", fn);
    const bool ds = wd_planes != nullptr;
    if (ds) {
        NDET_REQUIRE(Cin == BT_MID, NDET_E_UNSUPPORTED, "%s: the fused downsample branch takes Cin = %d (got %d)", fn, BT_MID, Cin);
        NDET_REQUIRE(scale_d && shift_d && wd_inv_scale > 0.0f, NDET_E_INVALID, "%s: the downsample branch needs its scale / shift / inverse weight scale", fn);
    } else {
        NDET_REQUIRE(Cin == Cout, NDET_E_INVALID, "%s: an identity residual needs Cin == Cout (%d vs %d)", fn, Cin, Cout);
    }
    NDET_REQUIRE((int64_t)N * H * W * (Cin > Cout ? Cin : Cout) * 4 < ((int64_t)0xfffffff0), NDET_E_UNSUPPORTED, "%s: tensors are addressed with 32-bit byte offsets (< 4 GB)", fn);
    NDET_REQUIRE((((uintptr_t)x | (uintptr_t)w1_planes | (uintptr_t)w2_planes | (uintptr_t)w3_planes | (uintptr_t)wd_planes) & 15) == 0, NDET_E_UNSUPPORTED,
                 "%s: x / weight planes must be 16-byte aligned", fn);
    NDET_REQUIRE(!guard || (guard_l1_host && guard_tol > 0.0f), NDET_E_INVALID, "%s: the guard needs its four l1 constants and a tolerance", fn);
    BottleneckParams p;
    p.x = x; p.out = out; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.tiles_x = (W + BT_TW - 1) / BT_TW; p.tiles_y = (H + BT_TH - 1) / BT_TH;
    p.w1 = w1_planes; p.w2 = w2_planes; p.w3 = w3_planes; p.wd = wd_planes;
    p.w1inv = w1_inv_scale; p.w2inv = w2_inv_scale; p.w3inv = w3_inv_scale; p.wdinv = ds ? wd_inv_scale : 1.0f;
    p.s1 = scale1; p.b1 = shift1; p.s2 = scale2; p.b2 = shift2; p.s3 = scale3; p.b3 = shift3; p.sd = scale_d; p.bd = shift_d;
    p.amax_in = in_amax; p.amax_out = out_amax;
    p.guard = guard;
    p.g1 = guard ? guard_l1_host[0] : 0.0f; p.g2 = guard ? guard_l1_host[1] : 0.0f; p.g3 = guard ? guard_l1_host[2] : 0.0f; p.gd = guard ? guard_l1_host[3] : 0.0f;
    p.gtol = guard_tol;
    p.nt = (int64_t)N * H * W * Cout * 4 >= ((int64_t)32 << 20) ? 1 : 0;
    const int64_t blocks = (int64_t)N * p.tiles_x * p.tiles_y;
    NDET_REQUIRE(blocks < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: too many patches", fn);
    const size_t lds = (size_t)(2 * (BT_AROWS + BT_MID) * SPL_RS + 2 * BT_AROWS * BT_MID + (ds ? 2 * BT_OROWS * BT_MID : 0)) * sizeof(uint16_t);
    const void* kfn = ds ? (const void*)k_bottleneck_f16x2<true, 2> : (Cin == 64 ? (const void*)k_bottleneck_f16x2<false, 2> : (const void*)k_bottleneck_f16x2<false, 8>);
    const int variant = ds ? 0 : (Cin == 64 ? 1 : 2);
    static int attr_state[16][3] = {{0}};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (lds > 64 * 1024 && attr_state[dev][variant] == 0) {
        hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        NDET_REQUIRE(e == hipSuccess, NDET_E_LAUNCH, "%s: cannot raise the LDS limit to %zu bytes: %s", fn, lds, hipGetErrorString(e));
        attr_state[dev][variant] = 1;
    }
    if (ds) hipLaunchKernelGGL((k_bottleneck_f16x2<true, 2>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    else if (Cin == 64) hipLaunchKernelGGL((k_bottleneck_f16x2<false, 2>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((k_bottleneck_f16x2<false, 8>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}
