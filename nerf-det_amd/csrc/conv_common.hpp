// Shared by the two implicit-GEMM convolution kernels (conv3d_kernels.hip: exact fp32 MFMA; conv_split_kernels.hip:
// fp32 operands split into three bf16 terms on the bf16 matrix cores).
#pragma once
#include <type_traits>
#include "ndet_common.hpp"

#define CBK 32          // K step (input channels per step)

struct Conv3dParams {
    const float* in;      // (D, H, W, Cin)
    const float* w;       // packed (taps, Cout, Cin)
    float* out;           // (OD, OH, OW, Cout)   [transposed: (2D, 2H, 2W, Cout)]
    const float* scale;   // (Cout) or null
    const float* shift;   // (Cout) or null
    const float* res;     // same shape as out, or null
    float* partial;       // split-K workspace (splits, M, Cout) or null
    int D, H, W, Cin;
    int OD, OH, OW, Cout;
    int kd, kh, kw;       // kernel extent per axis (2D convolution: D = batch, kd = 1)
    int sd, sh, sw;       // stride per axis
    int pd, ph, pw;       // zero padding per axis
    int relu;             // 0 none, 1 ReLU last (after the residual add), 2 ReLU before the residual add
    int transposed;       // 1: ConvTranspose3d k=2 s=2 (blockIdx.z = tap)
    int splits;           // split-K factor (blockIdx.z = split) when !transposed
    int M;                // GEMM rows: output voxels (input voxels when transposed)
    int res_up2;          // 1: residual is a (OD, ceil(OH/2), ceil(OW/2), Cout) map read at (d, h>>1, w>>1): nearest x2 upsample-add
    int RH, RW;           //    its H and W
    int max_order;        // bf16x3 kernels: products (pa, pb) with pa + pb <= max_order are issued -- 2: all six (fp32-class result),
                          // 0: a0*b0 only = both operands rounded to bf16, fp32 accumulate (the "bf16" arithmetic of BASELINE configs 3/5)
    int direct = 0;       // unified bf16x3 tiles: 1 = the epilogue stores straight from the MFMA's C layout with buffer operations (no LDS staging,
                          // no barriers; needs splits == 1, no transposed mode, Cout % 32 == 0, output < 4 GB)
    int order = 0;        // bf16x3 grid kernels: which workgroups meet in one XCD's L2 (workgroup b is dispatched to XCD b % 8).  0: grid order (row tiles
                          // fastest).  1: the row tiles of one (column tile, K split) weight slice run on one XCD -- layers whose weights outweigh their
                          // activations (the 20x20x8 / 10x10x4 neck levels: 42 - 170 MB of weight planes, re-streamed from HBM by every XCD in grid order)
                          // 2: the column tiles of one row tile run on one XCD, side by side in time -- layers whose activations outweigh their weights
                          // and that have several column tiles (1x1 layers into 256 - 1024 channels: grid order re-reads the rows once per XCD they land on)
    // fp16-pair arithmetic (max_order == 1; conv_split_kernels.hip, SCH 1): both operands are pre-scaled by a power of two so that their largest
    // magnitude sits in [2^14, 2^15) -- the weights once, when their planes are built (`winv` = 1 / that scale), the activations while
    // they are split, by the scale the kernel derives from `amax_in`; the epilogue multiplies the accumulators by the inverse of both.
    const float* amax_in = nullptr;   // device: the input tensor's amax slot (8 per-XCD sub-slots, see conv_amax_read below)
    float winv = 1.0f;                // 1 / (weight scale)
    const float* w_amax = nullptr;    // device, optional: the amax slot the weight planes were scaled by (training: the weights move every step and
                                      // the weight gradient's "weight" operand is dy -- their planes are split on the device by conv_xscale of this
                                      // slot, so 1 / scale is conv_xinv_of the same slot and no maximum ever travels to the host); replaces winv
    int nt = 0;                       // 1: the output is written with non-temporal stores (set by the launcher for outputs that cannot stay in the caches)
    float* amax_out = nullptr;        // any arithmetic, optional: max |out| is atomically maxed into the slot at amax_out (1 KiB, zeroed by the caller
                                      // before the launch) -- the next layer's amax_in without another pass over the tensor
    // chained 1x1 projection of the finished rows (halo tiles that own all Cout channels of their rows; conv_map_rows below): the detector's 256 -> 32
    // feature mapping (mmdet3d/models/detectors/nerfdet.py:194-197) behind the FPN output convolution, so that the 276 MB feature map is not read
    // back by a launch of its own.  map_w: (Cout, 32) floats = scale_c * Wm[j][c]; map_b: (32) = sum_c shift_c Wm[j][c] + bm[j] (the convolution's own
    // affine folded in on the host: no ReLU / residual on such a launch); map_out: (M, 32) floats
    const float* map_w = nullptr;
    const float* map_b = nullptr;
    float* map_out = nullptr;
    // range guard of the fp16-pair arithmetic (conv_guard_check below)
    unsigned* guard = nullptr;        // device word, bit 0 raised when this launch's absolute error floor exceeds guard_tol (null: no check)
    float guard_l1 = 0.0f;            // max over output channels j of |scale_j| (sum_k |w_jk| + wmax #{k: 0 < |w_jk| < 2^-16 wmax})
    float guard_tol = 0.0f;           // absolute tolerance the floor is compared with
};

// ---- range guard of the fp16-pair arithmetic ----
// The activation scale is per TENSOR: an element keeps a relative error of 2^-22 only while it lies within 2^-16 of the tensor's maximum; below
// that its error is absolute, <= 2^-39 amax (fp16's subnormal spacing under the scale).  An output y_j = scale_j sum_k w_jk a_k therefore carries,
// beside the fp32-class relative term 2^-22 sum |w||a|, an ABSOLUTE floor of at most
//         2^-39 amax_in |scale_j| sum_k |w_jk|          (+ the same for the few weights below 2^-16 of the weight maximum: folded into guard_l1)
// however the magnitudes are distributed inside the tensor.  For activations of ordinary size the floor is far below any tolerance (amax 1e3,
// ||w||_1 30: 5e-8); it becomes visible when a tensor's maximum is ~1e6 and more -- the sigma-MLP rows of voxels no view sees (1e9, nerfdet.py:236-243),
// a saturated image region, a BatchNorm-folded outlier channel.  Nothing has to guess which tensor that is: amax_in is on the device at kernel
// entry, so workgroup 0 compares the floor with the tolerance and raises a flag that travels to the host with the detections' status word
// (nerfdet_amd/detector.py re-runs such a scene on the six-product bf16x3 arithmetic, whose operands are exact).  The check itself: conv_guard_check
// below (it needs the slot readers).

// ---- the amax slot of a tensor: 8 sub-slots, one per XCD, each in a 128-byte line of its own (256 floats = 1 KiB per slot) ----
// The L2s of the eight XCDs are kept coherent line by line: a line that workgroups on different XCDs update (atomics or stores) migrates
// between the L2s on every access -- measured: one atomic per wave on ONE address cost a 240 000-row layer 70 us, and a per-XCD hint word
// sharing its line with the other XCDs' hints cost more than it saved.  Here a workgroup touches only the sub-slot of the XCD it runs on
// (hardware register XCC_ID), so the line stays in that L2 and the atomic is an L2-local operation; the reader takes the maximum of the
// eight sub-slots (read-only by then: shared copies, no migration).
#define NDET_AMAX_SUB 8
#define NDET_AMAX_STRIDE 32     // floats between sub-slots (128 bytes)
__device__ __forceinline__ unsigned conv_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }   // HW_REG_XCC_ID[3:0]

// max |x| of the tensor behind `slot` (uniform: every lane reads one sub-slot, an 8-lane butterfly finishes)
__device__ __forceinline__ float conv_amax_read(const float* slot) {
    float v = slot[(threadIdx.x & (NDET_AMAX_SUB - 1)) * NDET_AMAX_STRIDE];
#pragma unroll
    for (int o = NDET_AMAX_SUB / 2; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
}

// Power-of-two activation scale of the fp16-pair kernels and its inverse, from the tensor's max |x| = m 2^(eb-126), m in [1/2, 1):
// x * conv_xscale < 2^15.  Tensors whose maximum is below 2^-97 (eb < 30) are scaled by 2^111: no overflow of the scale itself.
__device__ __forceinline__ float conv_xscale_of(float amax) {
    int eb = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    eb = eb < 30 ? 30 : eb;
    return __uint_as_float((unsigned)(268 - eb) << 23);
}
__device__ __forceinline__ float conv_xinv_of(float amax) {
    int eb = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    eb = eb < 30 ? 30 : eb;
    return __uint_as_float((unsigned)(eb - 14) << 23);
}
__device__ __forceinline__ float conv_xscale(const float* amax) { return conv_xscale_of(conv_amax_read(amax)); }
// what the accumulators are multiplied by before the epilogue's affine: 1 except in the fp16-pair arithmetic (exact: a power of two).
// Every lane of the calling wave must be active (conv_amax_read shuffles).
__device__ __forceinline__ float conv_winv(const Conv3dParams& p) { return p.w_amax ? conv_xinv_of(conv_amax_read(p.w_amax)) : p.winv; }
__device__ __forceinline__ float conv_oscale(const Conv3dParams& p) { return p.amax_in ? conv_xinv_of(conv_amax_read(p.amax_in)) * conv_winv(p) : 1.0f; }
__device__ __forceinline__ float conv_amax_in(const Conv3dParams& p) { return p.amax_in ? conv_amax_read(p.amax_in) : 0.0f; }
__device__ __forceinline__ float conv_oscale_of(const Conv3dParams& p, float amax_in) { return p.amax_in ? conv_xinv_of(amax_in) * conv_winv(p) : 1.0f; }

// max |v| of a WORKGROUP -> one L2-local atomic on this XCD's sub-slot (non-negative floats order like their bit patterns).  Sent without
// looking at the sub-slot first: a read in front of it (to skip the atomic when the sub-slot already holds as much) made every workgroup wait
// for a load behind its own stores -- 0.11 ms of a cfg2 step (tools/diag/ab_variants.sh, tools/diag/amax_commit_variants.patch).  Every
// thread of the workgroup must arrive (there is a barrier inside).
__device__ __forceinline__ void conv_amax_commit(float* slot, float mx) {
    __shared__ float wg_amax[16];
#pragma unroll
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const int nw = ((int)blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) wg_amax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i) mx = fmaxf(mx, wg_amax[i]);
        const unsigned bits = __float_as_uint(mx);
        unsigned* sub = reinterpret_cast<unsigned*>(slot) + conv_xcc_id() * NDET_AMAX_STRIDE;
        if (bits) atomicMax(sub, bits);
        // word 1 of the sub-slot: the SMALLEST non-zero workgroup maximum, kept as the maximum of (bits of +inf - bits) so that the zero fill means
        // "none yet" (conv_tilemin_read; the range guard's second condition).  An all-zero tile is exact in any arithmetic: not recorded.
        if (bits && bits <= 0x7f800000u) atomicMax(sub + 1, 0x7f800000u - bits);
    }
}

// The smallest non-zero maximum any workgroup committed for the tensor behind `slot` (0: none recorded).  Every lane of the calling wave must be
// active (shuffles).
__device__ __forceinline__ float conv_tilemin_read(const float* slot) {
    unsigned v = reinterpret_cast<const unsigned*>(slot)[(threadIdx.x & (NDET_AMAX_SUB - 1)) * NDET_AMAX_STRIDE + 1];
#pragma unroll
    for (int o = NDET_AMAX_SUB / 2; o; o >>= 1) { const unsigned w = __shfl_xor(v, o); v = w > v ? w : v; }
    v = __builtin_amdgcn_readfirstlane(v);
    return v ? __uint_as_float(0x7f800000u - v) : 0.0f;
}

// The range guard's check (see the comment at Conv3dParams::guard): raised when BOTH hold --
//   (1) the launch's absolute error floor, 2^-39 max|in| guard_l1, exceeds the tolerance, and
//   (2) some part of the input really lives below the fp16-pair window: the smallest workgroup-tile maximum recorded for it is below 2^-16 of the
//       tensor's maximum.  Where every region of a tensor is within 2^-16 of its maximum, outputs of that region carry an fp32-class error of
//       ~2^-22 |w||a| >= the floor anyway (a uniformly large tensor -- a deep un-normalised network's activations of 1e6 -- loses nothing to the
//       per-tensor scale; without (2) the ResNet-101 workload tripped on every scene).  Tile granularity: 64 - 128 rows x the tile's channels.
// Every thread of workgroup 0 must call (the slot reads shuffle).
__device__ __forceinline__ void conv_guard_check(const Conv3dParams& p, float amax_in) {
    if (!p.guard || blockIdx.x != 0 || blockIdx.y != 0 || blockIdx.z != 0) return;
    const float tmin = conv_tilemin_read(p.amax_in);
    // weights scaled on the device (w_amax): guard_l1 is then the contraction length K and ||w||_1 <= K max|w| the bound (no host copy of the weights' norm)
    const float l1 = p.w_amax ? p.guard_l1 * conv_amax_read(p.w_amax) : p.guard_l1;
    if (threadIdx.x == 0 && amax_in * l1 * 0x1p-39f > p.guard_tol && tmin < amax_in * 0x1p-16f) atomicOr(p.guard, 1u);
}

// internal launchers (one per kernel family) and the shared split-K reduction
int conv_f32_launch(Conv3dParams& p, int tile, hipStream_t st, const char* fn);
int conv_split_launch(Conv3dParams& p, int tile, hipStream_t st, const char* fn);
int conv_splitk_reduce_launch(const Conv3dParams& p, hipStream_t st, const char* fn);

// Logical (row tile, column tile, split / tap) of this workgroup under p.order; gx = gridDim.x row tiles, gy column tiles, gz splits.
struct ConvBlock { int x, y, z; };
__device__ __forceinline__ ConvBlock conv_block(const Conv3dParams& p) {
    ConvBlock b{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    if (p.order == 1) {
        const int s = gy * gz, l = b.x + gx * (b.y + gy * b.z);
        const int slice = l % s;
        b.x = l / s; b.y = slice % gy; b.z = slice / gy;
    } else if (p.order == 2) {
        // groups of 8 row tiles x gy column tiles: workgroup 8 j + k of a group is (row tile k, column tile j) -- dispatched to XCD k with its
        // gy - 1 siblings 8 workgroups apart, so the activation rows come from HBM once; the gx % 8 last row tiles keep grid order
        const int l = b.x + gx * b.y, full = gx & ~7;
        if (l < full * gy) { const int w = l % (8 * gy); b.x = (l / (8 * gy)) * 8 + (w & 7); b.y = w >> 3; }
        else { const int r = l - full * gy, rem = gx - full; b.x = full + r % rem; b.y = r / rem; }
    }
    return b;
}

// Output row of GEMM row m (identity; the k2 s2 transposed convolution scatters tap `ztap` of input voxel m).
__device__ __forceinline__ int64_t conv_out_row(const Conv3dParams& p, int m, int ztap) {
    if (!p.transposed) return m;
    const int iw = m % p.W, ih = (m / p.W) % p.H, id = m / (p.W * p.H);
    const int kd = ztap >> 2, kh = (ztap >> 1) & 1, kw = ztap & 1;
    return ((int64_t)(2 * id + kd) * p.OH + (2 * ih + kh)) * p.OW + (2 * iw + kw);
}

// Fused epilogue over `rows` rows of a C tile staged in LDS (row stride cld floats, BN columns): per-channel scale/shift,
// ReLU before or after the residual add, residual (optionally a half-resolution map read nearest-upsampled), or raw
// split-K partials.  Threads own float4 pieces along Cout so global traffic is whole rows.
struct ConvLinearRows {
    int first;
    __device__ __forceinline__ int operator()(int row) const { return first + row; }
};

// `mx`: running max |v| of the final values this thread stored (the caller commits it once, conv_amax_commit).  `osc`: conv_oscale_of(p, amax)
// with the slot read at KERNEL ENTRY (conv_amax_in: a scalar register over the K walk instead of a global load in front of the stores).
template <int BN, int NTHR, typename RowMap>
__device__ __forceinline__ void conv_store_rows_mapped(const Conv3dParams& p, const float* Cs, int cld, int rows, int n0, int tid, int ztap,
                                                       int zsplit, RowMap m_of, float& mx, float osc) {
    const bool raw = (!p.transposed && p.splits > 1);
    float* dst = raw ? p.partial + (int64_t)zsplit * p.M * p.Cout : p.out;
    auto res_row = [&](int m, int64_t orow) -> int64_t {
        if (!p.res_up2) return orow;
        const int ow = m % p.OW, oh = (m / p.OW) % p.OH, od = m / (p.OW * p.OH);
        return ((int64_t)od * p.RH + (oh >> 1)) * p.RW + (ow >> 1);
    };
    if ((p.Cout & 3) == 0) {
        constexpr int V = BN / 4;
        // Upsampled residual over consecutive GEMM rows: (ow, oh, od) of the thread's row is divided out once and stepped from then on (a thread's
        // rows are NTHR / V apart) -- three integer divisions per float4 cost the FPN lateral 0 100 us of its 260 (tools/diag/lat0_probe.py).
        constexpr bool kWalk = std::is_same<RowMap, ConvLinearRows>::value && NTHR % V == 0;
        const bool walk = kWalk && !raw && p.res && p.res_up2;
        int uw = 0, uh = 0, ud = 0;
        if (walk) { const int m = m_of(tid / V); uw = m % p.OW; uh = (m / p.OW) % p.OH; ud = m / (p.OW * p.OH); }
        auto step = [&]() {
            if (!walk) return;
            uw += NTHR / V;
            while (uw >= p.OW) { uw -= p.OW; if (++uh == p.OH) { uh = 0; ++ud; } }
        };
        for (int idx = tid; idx < rows * V; idx += NTHR, step()) {
            const int row = idx / V, c4 = idx % V;
            const int m = m_of(row), co = n0 + c4 * 4;   // m < 0: the tile row has no output voxel
            if (m < 0 || m >= p.M || co >= p.Cout) continue;
            const int64_t orow = conv_out_row(p, m, ztap);
            float4 v = *reinterpret_cast<const float4*>(Cs + row * cld + c4 * 4);
            v.x *= osc; v.y *= osc; v.z *= osc; v.w *= osc;
            if (!raw) {
                if (p.scale) {
                    const float4 sc = *reinterpret_cast<const float4*>(p.scale + co), sh = *reinterpret_cast<const float4*>(p.shift + co);
                    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                }
                if (p.relu == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (p.res) {
                    const int64_t rrow = walk ? ((int64_t)ud * p.RH + (uh >> 1)) * p.RW + (uw >> 1) : res_row(m, orow);
                    const float4 rr = *reinterpret_cast<const float4*>(p.res + rrow * p.Cout + co);
                    v.x = v.x + rr.x; v.y = v.y + rr.y; v.z = v.z + rr.z; v.w = v.w + rr.w;
                }
                if (p.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
            if (p.nt) {
                typedef float f32x4nt __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store((f32x4nt){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4nt*>(dst + orow * p.Cout + co));
            }
            else *reinterpret_cast<float4*>(dst + orow * p.Cout + co) = v;
        }
    } else {  // Cout not a multiple of 4 (the fused head conv, Cout = 25): scalar columns
        for (int idx = tid; idx < rows * BN; idx += NTHR) {
            const int row = idx / BN, c = idx % BN;
            const int m = m_of(row), co = n0 + c;
            if (m < 0 || m >= p.M || co >= p.Cout) continue;
            const int64_t orow = conv_out_row(p, m, ztap);
            float v = Cs[row * cld + c] * osc;
            if (!raw) {
                if (p.scale) v = v * p.scale[co] + p.shift[co];
                if (p.relu == 2) v = fmaxf(v, 0.0f);
                if (p.res) v = v + p.res[res_row(m, orow) * p.Cout + co];
                if (p.relu == 1) v = fmaxf(v, 0.0f);
                mx = fmaxf(mx, fabsf(v));
            }
            dst[orow * p.Cout + co] = v;
        }
    }
}

// tile rows are consecutive GEMM rows starting at m_first
template <int BN, int NTHR>
__device__ __forceinline__ void conv_store_rows(const Conv3dParams& p, const float* Cs, int cld, int m_first, int rows, int n0, int tid,
                                                int ztap, int zsplit, float& mx, float osc) {
    conv_store_rows_mapped<BN, NTHR>(p, Cs, cld, rows, n0, tid, ztap, zsplit, ConvLinearRows{m_first}, mx, osc);
}
// Chained 1x1 projection of 64 tile rows staged in LDS (raw accumulators, as conv_store_rows_mapped reads them): out[m][j] = osc * sum_c Cs[row][c] *
// Wl[c][j] + map_b[j], 32 outputs per row, fp32 FMAs (exact-operand arithmetic, whatever the convolution ran in).  Wl: the (BN, 32) weight in LDS.
// A thread owns 4 rows x 4 outputs over a SLICE of the channels (a quarter, or an eighth where the tile has 1 024 threads; 5 LDS reads per 16 FMAs; one output pair per thread over all
// channels was LDS-issue bound: +90 us on the FPN output convolution, as much as the separate launch it replaces); the slices' partial sums per
// output meet in LDS -- in the rows' own staging area, which is dead by then -- and are added in a fixed order.  Three barriers inside: every
// thread of the workgroup must call.
template <int BN, int NTHR, typename RowMap>
__device__ __forceinline__ void conv_map_rows(const Conv3dParams& p, float* Cs, int cld, int tid, RowMap m_of, float osc, const float* Wl) {
    constexpr int KS = NTHR >= 1024 ? 8 : 4;           // channel slices = 128-thread groups at work (all 16 waves of the eight-producer tile)
    static_assert(BN % KS == 0 && NTHR >= 128 * KS, "conv_map_rows: 128 threads per channel slice");
    constexpr int KQ = BN / KS;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const int kq = tid >> 7, rg = (tid >> 3) & 15, jg = tid & 7;     // channel slice, row group (rows rg + 16 i), output quad
    if (tid < 128 * KS) {
        const float* c = Cs + rg * cld + kq * KQ;
        const float* w = Wl + (kq * KQ) * 32 + 4 * jg;
#pragma unroll 2
        for (int k = 0; k < KQ; ++k) {
            const float4 wv = *reinterpret_cast<const float4*>(w + k * 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = c[(16 * i) * cld + k];
                acc[i][0] = fmaf(v, wv.x, acc[i][0]); acc[i][1] = fmaf(v, wv.y, acc[i][1]);
                acc[i][2] = fmaf(v, wv.z, acc[i][2]); acc[i][3] = fmaf(v, wv.w, acc[i][3]);
            }
        }
    }
    __syncthreads();                                   // every reader of the staged rows is done: the partial sums take their place
    float* P = Cs;                                     // [KS slices][64 rows][32]  (KS * 8 KB <= the rows' 65 KB)
    if (tid < 128 * KS) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(P + ((kq * 64 + rg + 16 * i) * 32 + 4 * jg)) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
    __syncthreads();
    for (int o = tid; o < 64 * 8; o += NTHR) {         // a float4 of one row's outputs per thread
        const int row = o >> 3, j4 = o & 7;
        const int m = m_of(row);
        if (m < 0 || m >= p.M) continue;
        float4 t = *reinterpret_cast<const float4*>(P + (row * 32 + 4 * j4));
#pragma unroll
        for (int q = 1; q < KS; ++q) {
            const float4 u = *reinterpret_cast<const float4*>(P + ((q * 64 + row) * 32 + 4 * j4));
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        const float4 b = *reinterpret_cast<const float4*>(p.map_b + 4 * j4);
        *reinterpret_cast<float4*>(p.map_out + (int64_t)m * 32 + 4 * j4) = make_float4(fmaf(t.x, osc, b.x), fmaf(t.y, osc, b.y), fmaf(t.z, osc, b.z), fmaf(t.w, osc, b.w));
    }
    __syncthreads();                                   // the staging area is free for the next half's rows
}

// does this launch's epilogue write the layer's final values (split-K launches write partials: their reduce pass commits the maximum)
__device__ __forceinline__ bool conv_writes_final(const Conv3dParams& p) { return p.transposed || p.splits <= 1; }
