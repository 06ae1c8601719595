// Shared host/device helpers for libnerfdet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/nerfdet_hip.h"

#define NDET_WAVE 64

// ---- host-side error plumbing ------------------------------------------------------------
void ndet_set_error(const char* fmt, ...);

#define NDET_REQUIRE(cond, code, ...)  \
    do {                               \
        if (!(cond)) {                 \
            ndet_set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

#define NDET_CHECK_LAUNCH(name)                                                      \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            ndet_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
            return NDET_E_LAUNCH;                                                    \
        }                                                                            \
    } while (0)

// ---- device helpers ----------------------------------------------------------------------

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one).  Give each XCD a
// contiguous slab of tiles so neighbouring voxel tiles (which hit the same feature pixels)
// share an L2.  Bijective for any n (cdna guide, "XCD swizzle must be bijective").  Speed only.
__device__ __forceinline__ int ndet_xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (b >> 3);
}

// One voxel -> one view, the arithmetic of nerfdet.py:398-403.
//   torch.bmm(P, [p;1]) on the CPU reference evaluates each row as a k-ordered FMA chain
//   acc = p0*x; acc = fma(p1,y,acc); acc = fma(p2,z,acc); acc = acc + p3   (measured: 0 of
//   3.84 M elements differ from MKL sgemm at the cfg2 shapes) -- the same chain is used here,
//   the file is compiled with -ffp-contract=off so nothing else fuses.
//   `/` is IEEE correctly-rounded (hipcc default), rintf = round-half-even = torch.round.
//   Validity is evaluated on the rounded *float* coordinate: equivalent to the reference's
//   int64 cast + compare for every finite value, and NaN / inf / |x| >= 2^63 come out invalid
//   on both sides (x86 cvttss2si yields INT64_MIN there).
__device__ __forceinline__ bool ndet_project(const float* __restrict__ P, float px, float py, float pz,
                                             int w, int h, int& xi, int& yi) {
    float u = P[0] * px;
    u = fmaf(P[1], py, u);
    u = fmaf(P[2], pz, u);
    u = u + P[3];
    float v = P[4] * px;
    v = fmaf(P[5], py, v);
    v = fmaf(P[6], pz, v);
    v = v + P[7];
    float d = P[8] * px;
    d = fmaf(P[9], py, d);
    d = fmaf(P[10], pz, d);
    d = d + P[11];
    const float fx = rintf(u / d);
    const float fy = rintf(v / d);
    const bool ok = (fx >= 0.0f) && (fy >= 0.0f) && (fx < (float)w) && (fy < (float)h) && (d > 0.0f);
    xi = ok ? (int)fx : 0;
    yi = ok ? (int)fy : 0;
    return ok;
}

// Gradient scatter of the backward kernels.  Default: float atomics (global_atomic_add_f32) -- fast, but the ORDER of the adds, and with it the last
// bits of every sum, changes from run to run.  Deterministic mode (tests: ndet_measurement_knob("deterministic_scatter", 1); the caller then hands
// a zeroed buffer of int64 in place of the float buffer): every contribution is rounded to a multiple of 2^-40 and added as a 64-bit INTEGER --
// integer addition is associative, so the sum is the same whatever the order (range +-8.4e6, resolution 9e-13; the caller converts back).
extern int g_ndet_deterministic_scatter;      // host side, set by ndet_measurement_knob, read by the backward launchers
__device__ __forceinline__ void ndet_scatter_add(float* buf, int64_t idx, float v, int det) {
    if (det) atomicAdd(reinterpret_cast<unsigned long long*>(buf) + idx, (unsigned long long)(long long)llrintf(v * 0x1p40f));
    else unsafeAtomicAdd(buf + idx, v);
}

__device__ __forceinline__ float4 ndet_add4(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
