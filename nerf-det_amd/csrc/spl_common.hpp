// Operand splitting shared by the split-family convolution kernels (conv_split_kernels.hip) and the whole-bottleneck kernel
// (bottleneck_kernels.hip): fp32 -> three bf16 planes (exact) or two fp16 planes of the pre-scaled value, and the MFMA wrappers per scheme.
#pragma once
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

// 4 fp32 -> three planes of 4 bf16 (8 B each)
__device__ __forceinline__ void spl_split4(const float4 v, uint2& p0, uint2& p1, uint2& p2) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    uint32_t o0[2], o1[2], o2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        o0[i] = spl_pack(a, b);
        const float ra = a - spl_lo(o0[i]), rb = b - spl_hi(o0[i]);   // exact
        o1[i] = spl_pack(ra, rb);
        const float sa = ra - spl_lo(o1[i]), sb = rb - spl_hi(o1[i]); // exact, <= 8 significant bits left
        o2[i] = spl_pack(sa, sb);
    }
    p0 = make_uint2(o0[0], o0[1]);
    p1 = make_uint2(o1[0], o1[1]);
    p2 = make_uint2(o2[0], o2[1]);
}

// fp16 pair scheme (SCH 1 of the halo tiles): with the tensor pre-scaled by a power of two so that its largest magnitude sits near
// 2^15, x = hi + lo with hi = fp16(x), lo = fp16(x - hi): 2 x 11 significand bits, both halves in fp16's normal range for every
// element above ~4e-6 of the tensor's maximum.  a*b ~ hi_a hi_b + hi_a lo_b + lo_a hi_b (each product exact in fp32); the dropped
// lo_a lo_b and the rounding of the lo halves are <= 3 x 2^-22 |ab| -- below the rounding noise a K >= 64 fp32 accumulation carries.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t spl_pack_f16(float x, float y) {
    const f16x2v v = __builtin_convertvector((f32x2){x, y}, f16x2v);
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float spl_f16_lo(uint32_t u) { return (float)__builtin_bit_cast(f16x2v, u)[0]; }
__device__ __forceinline__ float spl_f16_hi(uint32_t u) { return (float)__builtin_bit_cast(f16x2v, u)[1]; }
__device__ __forceinline__ void spl_split4_f16(const float4 v, float scale, uint2& p0, uint2& p1) {
    const float x[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
    uint32_t o0[2], o1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        o0[i] = spl_pack_f16(a, b);
        o1[i] = spl_pack_f16(a - spl_f16_lo(o0[i]), b - spl_f16_hi(o0[i]));   // the differences are exact in fp32
    }
    p0 = make_uint2(o0[0], o0[1]);
    p1 = make_uint2(o1[0], o1[1]);
}

// Arithmetic scheme of a kernel instantiation.  SCH 0: three bf16 planes, six products (fp32-class to 2^-24).  SCH 1: two fp16 planes of the
// pre-scaled operands, three products (spl_split4_f16).  SCH 2: the leading bf16 plane only, one product (bf16 autocast arithmetic; the
// weight tensor keeps its three-plane layout).
template <int SCH> struct Spl {
    static constexpr int NPL = SCH == 1 ? 2 : (SCH == 2 ? 1 : 3);   // operand planes staged and multiplied
    static constexpr int WPL = SCH == 1 ? 2 : 3;                    // planes per K step in the weight tensor
};
template <int SCH>
__device__ __forceinline__ void spl_split(const float4 v, float xs, uint2& s0, uint2& s1, uint2& s2) {
    if (SCH == 1) spl_split4_f16(v, xs, s0, s1);
    else if (SCH == 2) s0 = make_uint2(spl_pack(v.x, v.y), spl_pack(v.z, v.w));
    else spl_split4(v, s0, s1, s2);
}
// two values -> the scheme's planes, packed pairs (element 0 in the low half)
template <int SCH>
__device__ __forceinline__ void spl_split2(float a, float b, float xs, uint32_t& o0, uint32_t& o1, uint32_t& o2) {
    if (SCH == 1) {
        a *= xs; b *= xs;
        o0 = spl_pack_f16(a, b);
        o1 = spl_pack_f16(a - spl_f16_lo(o0), b - spl_f16_hi(o0));
    } else {
        o0 = spl_pack(a, b);
        if (SCH == 0) {
            const float ra = a - spl_lo(o0), rb = b - spl_hi(o0);
            o1 = spl_pack(ra, rb);
            o2 = spl_pack(ra - spl_lo(o1), rb - spl_hi(o1));
        }
    }
}
typedef float f32x16_ __attribute__((ext_vector_type(16)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));
template <int SCH>
__device__ __forceinline__ f32x16_ spl_mfma32(bf16x8 a, bf16x8 b, f32x16_ c) {
    if (SCH == 1) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int SCH>
__device__ __forceinline__ f32x4_ spl_mfma16(bf16x8 a, bf16x8 b, f32x4_ c) {
    if (SCH == 1) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

