// dist.h -- the angular distances of granne, bit-for-bit, as gfx950 device code.
//
// f32: /root/reference/src/math.rs:5-52 (dot_product_f32) + src/elements/angular.rs:63-74.
//      The reference accumulates into 32 independent FUSED accumulators over 32-wide chunks,
//      then sums the 32 accumulators in order starting from 0.0, then folds the tail with
//      fused multiply-adds. Any other association changes low-order bits, and one flipped
//      comparison changes the ids a walk returns -- so this file keeps that association:
//      one lane evaluates one (candidate, query) pair with 32 VGPR accumulators. The file
//      must be compiled with -ffp-contract=off (every fused op below is an explicit fmaf).
// i8:  src/math.rs:59-89 + src/elements/angular_int.rs:47-60. The three sums are exact in
//      i32, so they may be split over lanes (v_dot4_i32_i8) and reduced in any order; the f32
//      tail (cvt, sqrt, mul, div, sub, clamp) is evaluated in the reference's order with
//      correctly rounded sqrt and divide (-fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace granne_hip {

// max(0, 1 - r) the way cmp::max on NotNan does it (angular.rs:70-72)
__device__ __forceinline__ float angular_from_dot(float r) {
    float d = 1.0f - r;
    return (0.0f <= d) ? d : 0.0f;
}

// x, q: 16-byte aligned LDS (or any) pointers to DIM floats. DIM % 4 == 0.
template <int DIM>
__device__ __forceinline__ float dot_f32_exact(const float* __restrict__ x, const float* __restrict__ q) {
    static_assert(DIM % 4 == 0, "vector path needs dim % 4 == 0");
    constexpr int FULL = (DIM / 32) * 32;
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < FULL; k += 4) {
        float4 a = *reinterpret_cast<const float4*>(x + k);
        float4 b = *reinterpret_cast<const float4*>(q + k);
        acc[(k + 0) & 31] = __builtin_fmaf(a.x, b.x, acc[(k + 0) & 31]);
        acc[(k + 1) & 31] = __builtin_fmaf(a.y, b.y, acc[(k + 1) & 31]);
        acc[(k + 2) & 31] = __builtin_fmaf(a.z, b.z, acc[(k + 2) & 31]);
        acc[(k + 3) & 31] = __builtin_fmaf(a.w, b.w, acc[(k + 3) & 31]);
    }
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) r = r + acc[i];
#pragma unroll
    for (int k = FULL; k < DIM; k += 4) {
        float4 a = *reinterpret_cast<const float4*>(x + k);
        float4 b = *reinterpret_cast<const float4*>(q + k);
        r = __builtin_fmaf(a.x, b.x, r);
        r = __builtin_fmaf(a.y, b.y, r);
        r = __builtin_fmaf(a.z, b.z, r);
        r = __builtin_fmaf(a.w, b.w, r);
    }
    return r;
}

// Same arithmetic with the query held in registers (q[k] are compile-time indexed).
template <int DIM>
__device__ __forceinline__ float dot_f32_exact_qreg(const float* __restrict__ x, const float (&q)[DIM]) {
    static_assert(DIM % 4 == 0, "vector path needs dim % 4 == 0");
    constexpr int FULL = (DIM / 32) * 32;
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int k = 0; k < FULL; k += 4) {
        float4 a = *reinterpret_cast<const float4*>(x + k);
        acc[(k + 0) & 31] = __builtin_fmaf(a.x, q[k + 0], acc[(k + 0) & 31]);
        acc[(k + 1) & 31] = __builtin_fmaf(a.y, q[k + 1], acc[(k + 1) & 31]);
        acc[(k + 2) & 31] = __builtin_fmaf(a.z, q[k + 2], acc[(k + 2) & 31]);
        acc[(k + 3) & 31] = __builtin_fmaf(a.w, q[k + 3], acc[(k + 3) & 31]);
    }
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) r = r + acc[i];
#pragma unroll
    for (int k = FULL; k < DIM; k += 4) {
        float4 a = *reinterpret_cast<const float4*>(x + k);
        r = __builtin_fmaf(a.x, q[k + 0], r);
        r = __builtin_fmaf(a.y, q[k + 1], r);
        r = __builtin_fmaf(a.z, q[k + 2], r);
        r = __builtin_fmaf(a.w, q[k + 3], r);
    }
    return r;
}

// Any dimension, scalar reads (x, q 4-byte aligned).
__device__ __forceinline__ float dot_f32_exact_rt(const float* __restrict__ x, const float* __restrict__ q,
                                                  uint32_t dim) {
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0f;
    uint32_t full = dim & ~31u;
    for (uint32_t base = 0; base < full; base += 32) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_fmaf(x[base + i], q[base + i], acc[i]);
    }
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) r = r + acc[i];
    for (uint32_t k = full; k < dim; ++k) r = __builtin_fmaf(x[k], q[k], r);
    return r;
}

// angular_int.rs:52-58 from the three exact sums
__device__ __forceinline__ float angular_int_from_sums(int r, int dx, int dy) {
    float rf = (float)r, dxf = (float)dx, dyf = (float)dy;
    float q = rf / (__builtin_sqrtf(dxf) * __builtin_sqrtf(dyf));
    if (q != q) q = 0.0f; // NaN (0/0) -> 0.0, angular_int.rs:55
    float d = 1.0f - q;
    return (0.0f <= d) ? d : 0.0f;
}

__device__ __forceinline__ int dot4_i8(uint32_t a, uint32_t b, int c) {
    return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); // v_dot4_i32_i8: signed x signed
}

} // namespace granne_hip
