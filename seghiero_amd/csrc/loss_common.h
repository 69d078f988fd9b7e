// Device helpers shared by the loss kernels (loss.hip, loss3.hip): ATen's bilinear source rule, the gather-form
// contribution range, logits interpolation, softmax-CE with optional gradient, block reduction to partials.
#pragma once
#include "common.h"

#define IGN 255
#define LOSS_PIX_PER_BLOCK 1024

struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lerp lerp_src(int dst, float scale, int in) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    Lerp L;
    L.i0 = (int)s;
    if (L.i0 > in - 1) L.i0 = in - 1;
    L.i1 = L.i0 + (L.i0 < in - 1 ? 1 : 0);
    L.w1 = s - (float)L.i0;
    L.w0 = 1.f - L.w1;
    return L;
}
__device__ __forceinline__ void contrib_range(int i, float inv_scale, int out, int& lo, int& hi) {
    // outputs whose source lies in (i-1, i+1): dst in ((i-0.5)/scale-0.5, (i+1.5)/scale-0.5).  floor / ceil of the open
    // ends is a superset even under f32 rounding (an end that rounds across an integer only drops/keeps a candidate whose
    // weight is ~1 ulp); every candidate re-derives the exact forward weights.
    lo = (int)floorf(((float)i - 0.5f) * inv_scale - 0.5f);
    hi = (int)ceilf(((float)i + 1.5f) * inv_scale - 0.5f);
    if (i == 0) lo = 0;
    lo = lo < 0 ? 0 : lo;
    hi = hi > out - 1 ? out - 1 : hi;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// interpolate C channels of pixel (oy,ox) from the low-res logits of image n
template <int MAXC>
__device__ __forceinline__ void fetch_logits(const float* __restrict__ base, long long ldl, int w, const Lerp& ly, const Lerp& lx,
                                             bool identity, int C, float (&z)[MAXC]) {
    // rows of ldl floats: 16-byte loads when the layout allows it (padded NHWC logits: ldl % 4 == 0, ldl >= round4(C),
    // 16-byte aligned base) -- 4x fewer vector-memory instructions; the per-channel arithmetic is the same either way
    const bool vec = (MAXC % 4 == 0) && (ldl & 3) == 0 && ldl >= ((C + 3) & ~3) && ((uintptr_t)base & 15) == 0;
    if (identity) {
        const float* p = base + ((long long)ly.i0 * w + lx.i0) * ldl;
        if (vec) {
#pragma unroll
            for (int j = 0; j < MAXC; j += 4) {
                if (j < C) {
                    const f32x4 v = ld4(p + j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[j + e] = j + e < C ? v[e] : 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[j + e] = 0.f;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < MAXC; ++j) z[j] = j < C ? p[j] : 0.f;
    } else {
        const float* p00 = base + ((long long)ly.i0 * w + lx.i0) * ldl;
        const float* p01 = base + ((long long)ly.i0 * w + lx.i1) * ldl;
        const float* p10 = base + ((long long)ly.i1 * w + lx.i0) * ldl;
        const float* p11 = base + ((long long)ly.i1 * w + lx.i1) * ldl;
        if (vec) {
#pragma unroll
            for (int j = 0; j < MAXC; j += 4) {
                if (j < C) {
                    const f32x4 a = ld4(p00 + j), b = ld4(p01 + j), c = ld4(p10 + j), d = ld4(p11 + j);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        z[j + e] = j + e < C ? ly.w0 * (lx.w0 * a[e] + lx.w1 * b[e]) + ly.w1 * (lx.w0 * c[e] + lx.w1 * d[e]) : 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[j + e] = 0.f;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < MAXC; ++j)
            z[j] = j < C ? ly.w0 * (lx.w0 * p00[j] + lx.w1 * p01[j]) + ly.w1 * (lx.w0 * p10[j] + lx.w1 * p11[j]) : 0.f;
    }
}

// loss.hip: adjoint of the bilinear resize as a gather from a full-resolution gradient [N*H*W][lddl] (not scaled)
int sh_launch_resize_adjoint_gather(const float* gfull, float* dlogits, int lddl, int N, int h, int w, int H, int W, hipStream_t st);
// ... from a workspace that holds the gradient at unit upstream gradient (written by a forward), scaled by gscale * gscale_dev[0]
int sh_launch_gather_from_grad(const float* gfull, const float* gscale_dev, float gscale, float* dlogits, int lddl, int N, int h, int w,
                               int H, int W, hipStream_t st);

// log-softmax CE of z[off..off+n) against target tgt: returns -log p_tgt; optionally adds coef*(softmax - onehot) to g
template <int MAXC, bool GRAD>
__device__ __forceinline__ float softmax_ce(const float (&z)[MAXC], int off, int n, int tgt, float coef, float (&g)[MAXC]) {
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) if (j >= off && j < off + n) mx = fmaxf(mx, z[j]);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) if (j >= off && j < off + n) se += expf(z[j] - mx);
    const float lse = mx + logf(se);
    float zt = 0.f;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
        if (j >= off && j < off + n) {
            if (j - off == tgt) zt = z[j];
            if (GRAD) g[j] += coef * (expf(z[j] - lse) - (j - off == tgt ? 1.f : 0.f));
        }
    }
    return lse - zt;
}

// block reduce NV floats -> partials[blockIdx.x][8]
template <int NV>
__device__ __forceinline__ void block_reduce_store(float (&v)[NV], float* __restrict__ partials) {
    __shared__ float red[NV][4];
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = wave_sum(v[j]);
    if ((t & 63) == 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) red[j][t >> 6] = v[j];
    }
    __syncthreads();
    if (t < 8) partials[(long long)blockIdx.x * 8 + t] = t < NV ? (red[t][0] + red[t][1]) + (red[t][2] + red[t][3]) : 0.f;
}

