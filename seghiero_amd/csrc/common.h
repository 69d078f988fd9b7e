// Shared device/host helpers for the seghiero_amd HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/seghiero_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SH_WAVE 64

static inline int sh_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SH_OK : SH_ELAUNCH;
}

static inline int64_t sh_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// wave-wide sum (all 64 lanes end with the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Bijective XCD-aware block remap (guide T1): blocks b and b+8 share an XCD (round-robin dispatch), so
// give each XCD a contiguous chunk of the logical grid => neighbouring tiles hit the same L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u, xcd = bid & 7u, loc = bid >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---- bf16 activation storage (BASELINE configs[4]): raw conv outputs and block outputs may be STORED as bf16 (half the HBM bytes);
// every kernel computes in fp32.  `base` is the tensor's base pointer whatever its element type, `idx` the ELEMENT index
// (row * ld + channel, a multiple of 4), `bf` != 0: bf16 elements (wave-uniform flag from the entry point's act_flags).
typedef unsigned int sh_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 bf16x4_to_f32(sh_u32x2 v) {
    return f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
}
__device__ __forceinline__ unsigned f32_to_bf16_rne(float f) {       // round to nearest even; NaN stays NaN
    const unsigned u = __float_as_uint(f);
    return (f != f) ? ((u >> 16) | 0x40u) : ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ sh_u32x2 f32_to_bf16x4(f32x4 v) {
    return sh_u32x2{f32_to_bf16_rne(v[0]) | (f32_to_bf16_rne(v[1]) << 16), f32_to_bf16_rne(v[2]) | (f32_to_bf16_rne(v[3]) << 16)};
}
__device__ __forceinline__ f32x4 lda4(const float* base, long long idx, int bf) {
    if (bf) return bf16x4_to_f32(*reinterpret_cast<const sh_u32x2*>(reinterpret_cast<const unsigned short*>(base) + idx));
    return ld4(base + idx);
}
// ReLU quad mask: one byte per (pixel, 4 adjacent channels), bit j = (out[c + j] > 0); pixel stride in BYTES (>= C / 4).  Written by
// sh_bn_act next to a residual block's output and read by the backward passes that need only the mask of that output (1/16 of its bytes).
__device__ __forceinline__ unsigned quad_mask_bits(f32x4 v) {
    return (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
}
__device__ __forceinline__ f32x4 quad_mask_load(const void* base, long long byte_idx) {     // -> 1.f / 0.f per channel
    const unsigned b = reinterpret_cast<const unsigned char*>(base)[byte_idx];
    return f32x4{(b & 1u) ? 1.f : 0.f, (b & 2u) ? 1.f : 0.f, (b & 4u) ? 1.f : 0.f, (b & 8u) ? 1.f : 0.f};
}
__device__ __forceinline__ void sta4(float* base, long long idx, f32x4 v, int bf) {
    if (bf) *reinterpret_cast<sh_u32x2*>(reinterpret_cast<unsigned short*>(base) + idx) = f32_to_bf16x4(v);
    else st4(base + idx, v);
}

// eight consecutive channels of one pixel: ONE 16-byte access of a bf16 tensor, two of an fp32 one (idx = element index, multiple of 8).  The
// streaming kernels take this form where their tensors are bf16 (bf16 compute mode): the 4-channel form moves 8 bytes per lane there.
struct sh_f8 { f32x4 lo, hi; };
typedef unsigned int sh_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ sh_f8 lda8(const float* base, long long idx, int bf) {
    sh_f8 r;
    if (bf) {
        const sh_u32x4 v = *reinterpret_cast<const sh_u32x4*>(reinterpret_cast<const unsigned short*>(base) + idx);
        r.lo = bf16x4_to_f32(sh_u32x2{v[0], v[1]}); r.hi = bf16x4_to_f32(sh_u32x2{v[2], v[3]});
    } else { r.lo = ld4(base + idx); r.hi = ld4(base + idx + 4); }
    return r;
}
__device__ __forceinline__ void sta8(float* base, long long idx, const sh_f8& v, int bf) {
    if (bf) {
        const sh_u32x2 a = f32_to_bf16x4(v.lo), b = f32_to_bf16x4(v.hi);
        *reinterpret_cast<sh_u32x4*>(reinterpret_cast<unsigned short*>(base) + idx) = sh_u32x4{a[0], a[1], b[0], b[1]};
    } else { st4(base + idx, v.lo); st4(base + idx + 4, v.hi); }
}
