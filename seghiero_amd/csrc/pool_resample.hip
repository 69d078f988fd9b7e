// Max-pool, global average pool / broadcast, bilinear resize (fwd + gather-form bwd), channel L2-normalise.
// NHWC, HBM-bound: one thread per (pixel, 4 channels) with 16-byte accesses; backward passes are written in
// gather form so they are deterministic and need no atomics.
#include "common.h"

static inline unsigned grid_for(long long work_items, int block = 256, int cap = 8192) {
    long long g = sh_cdiv(work_items, block);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}
#define GRID_STRIDE(i, total) for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (total); i += (long long)gridDim.x * 256)

// ------------------------------------------------------------------------------------------ maxpool 3x3 / 2 / pad 1
// reference: nn.MaxPool2d(kernel_size=3, stride=2, padding=1) behind models/backbone/resnet.py:68
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned* __restrict__ argmax,
                                                          int H, int W, int Ho, int Wo, int C4, long long total,
                                                          const float* __restrict__ isc, const float* __restrict__ ish, int af) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        long long q = i / C4;
        const int ow = (int)(q % Wo); q /= Wo;
        const int oh = (int)(q % Ho);
        const long long n = q / Ho;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned idx = 0;                  // per channel one byte: window position kh*3+kw of the FIRST maximum (ATen's rule)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                f32x4 v = lda4(x, (((n * H + ih) * W + iw) * C4 + c) * 4, af & 1);      // af: bit 0 x, 1 y stored as bf16
                if (isc != nullptr) {          // input read through the stem's train-mode BatchNorm + ReLU (sh_bn_act's operation order)
                    v = v * ld4(isc + 4 * c) + ld4(ish + 4 * c);
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (v[j] > m[j] || v[j] != v[j]) { m[j] = v[j]; idx = (idx & ~(0xffu << (8 * j))) | ((unsigned)(kh * 3 + kw) << (8 * j)); }
            }
        }
        sta4(y, i * 4, m, af & 2);
        if (argmax != nullptr) argmax[i] = idx;
    }
}
// backward, gather form: input pixel (ih,iw) receives dy from every window whose recorded first maximum is this pixel
// (each input element is written exactly once: deterministic, no atomics, x is not re-read).
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned* __restrict__ argmax, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int H, int W, int Ho, int Wo, int C4,
                                                          long long total) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        long long q = i / C4;
        const int iw = (int)(q % W); q /= W;
        const int ih = (int)(q % H);
        const long long n = q / H;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int oh = (ih + 1) / 2 - 1; oh <= (ih + 1) / 2; ++oh) {
            if (oh < 0 || oh >= Ho) continue;
            const int kh = ih - (oh * 2 - 1);
            if (kh < 0 || kh > 2) continue;
            for (int ow = (iw + 1) / 2 - 1; ow <= (iw + 1) / 2; ++ow) {
                if (ow < 0 || ow >= Wo) continue;
                const int kw = iw - (ow * 2 - 1);
                if (kw < 0 || kw > 2) continue;
                const long long o = ((n * Ho + oh) * Wo + ow) * C4 + c;
                const unsigned idx = argmax[o], want = (unsigned)(kh * 3 + kw);
                const f32x4 g = ld4(dy + o * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (((idx >> (8 * j)) & 0xffu) == want) acc[j] += g[j];
            }
        }
        st4(dx + i * 4, acc);
    }
}
extern "C" int sh_maxpool_fwd(const float* x, const float* in_scale, const float* in_shift, float* y, uint8_t* argmax, int N, int H, int W,
                              int C, int act_flags, void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || ((uintptr_t)argmax & 3)) return SH_EINVAL;
    if ((in_scale == nullptr) != (in_shift == nullptr) || (((uintptr_t)in_scale | (uintptr_t)in_shift) & 15)) return SH_EINVAL;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * (C / 4);
    maxpool_fwd_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(x, y, reinterpret_cast<unsigned*>(argmax), H, W, Ho, Wo, C / 4, total,
                                                                         in_scale, in_shift, act_flags);
    return sh_launch_status();
}
extern "C" int sh_maxpool_bwd(const uint8_t* argmax, const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    if (!argmax || !dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || ((uintptr_t)argmax & 3)) return SH_EINVAL;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * H * W * (C / 4);
    maxpool_bwd_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const unsigned*>(argmax), dy, dx, H, W, Ho, Wo, C / 4, total);
    return sh_launch_status();
}

// ------------------------------------------------------------------------------------------ global average pool
// x [N][HW][ldx] -> y [N][C].  block = (n, 64-channel chunk); 4 row groups; f32 partial sums, f64 combine.
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, long long ldx, float* __restrict__ y,
                                                          int HW, int C, float inv) {
    __shared__ double red[4][64];
    const int t = threadIdx.x, cl = t & 63, g = t >> 6, c = blockIdx.y * 64 + cl;
    const long long n = blockIdx.x;
    double s = 0;
    if (c < C) {
        // eight rows per pass with their loads issued together (one load per pass and a counter branch made 64 dependent passes of a
        // 16 x 16 map: 28 us for 8 MB); the f32 partial of a pass goes to the f64 sum
        const float* base = x + n * HW * ldx + c;
        int r = g;
        for (; r + 28 < HW; r += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[(long long)(r + 4 * u) * ldx];
            s += (double)(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
        }
        float part = 0.f;
        for (; r < HW; r += 4) part += base[(long long)r * ldx];
        s += part;
    }
    red[g][cl] = s;
    __syncthreads();
    if (t < 64 && c < C) y[n * C + c] = (float)(((red[0][t] + red[1][t]) + (red[2][t] + red[3][t])) * (double)inv);
}
extern "C" int sh_avgpool_fwd(const float* x, int ldx, float* y, int N, int HW, int C, void* stream) {
    if (!x || !y || N <= 0 || HW <= 0 || C <= 0 || ldx < C) return SH_EINVAL;
    dim3 grid(N, (unsigned)sh_cdiv(C, 64));
    avgpool_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, ldx, y, HW, C, 1.0f / (float)HW);
    return sh_launch_status();
}
extern "C" int sh_sum_hw(const float* dy, int lddy, float* dx, int N, int HW, int C, void* stream) {
    if (!dy || !dx || N <= 0 || HW <= 0 || C <= 0 || lddy < C) return SH_EINVAL;
    dim3 grid(N, (unsigned)sh_cdiv(C, 64));
    avgpool_fwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(dy, lddy, dx, HW, C, 1.0f);
    return sh_launch_status();
}
// dx[n][r][c] (+)= dy[n][c] * scale
__global__ __launch_bounds__(256) void bcast_kernel(const float* __restrict__ v, float* __restrict__ out, long long ldo, int HW,
                                                    int C, float scale, int accumulate, long long total) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const long long q = i / C, n = q / HW;
        const float val = v[n * C + c] * scale;
        float* dst = out + q * ldo + c;
        *dst = accumulate ? *dst + val : val;
    }
}
extern "C" int sh_avgpool_bwd(const float* dy, float* dx, int lddx, int N, int HW, int C, float scale, int accumulate, void* stream) {
    if (!dy || !dx || N <= 0 || HW <= 0 || C <= 0 || lddx < C) return SH_EINVAL;
    const long long total = (long long)N * HW * C;
    bcast_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(dy, dx, lddx, HW, C, scale, accumulate, total);
    return sh_launch_status();
}
extern "C" int sh_broadcast_hw(const float* x, float* y, int ldy, int N, int HW, int C, void* stream) {
    return sh_avgpool_bwd(x, y, ldy, N, HW, C, 1.0f, 0, stream);
}

// ------------------------------------------------------------------------------------------ bilinear (align_corners=False)
// ATen's rule: src = scale*(dst+0.5)-0.5 clamped at 0, scale = in/out (f32); i1 = i0 + (i0 < in-1).
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
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, long long ldx, float* __restrict__ y,
                                                           long long ldy, int h, int w, int H, int W, int C4, float sy, float sx,
                                                           long long total, int y_bf) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4) * 4;
        long long q = i / C4;
        const int ox = (int)(q % W); q /= W;
        const int oy = (int)(q % H);
        const long long n = q / H;
        const Lerp ly = lerp_src(oy, sy, h), lx = lerp_src(ox, sx, w);
        const float* b = x + n * h * w * ldx + c;
        const f32x4 v00 = ld4(b + ((long long)ly.i0 * w + lx.i0) * ldx), v01 = ld4(b + ((long long)ly.i0 * w + lx.i1) * ldx);
        const f32x4 v10 = ld4(b + ((long long)ly.i1 * w + lx.i0) * ldx), v11 = ld4(b + ((long long)ly.i1 * w + lx.i1) * ldx);
        const f32x4 r = ly.w0 * (lx.w0 * v00 + lx.w1 * v01) + ly.w1 * (lx.w0 * v10 + lx.w1 * v11);
        sta4(y, ((n * H + oy) * W + ox) * ldy + c, r, y_bf);
    }
}
// gather-form backward: each input pixel scans the output window that can reference it and re-derives the forward weights.
__device__ __forceinline__ void contrib_range(int i, float inv_scale, int out, int& lo, int& hi) {
    // outputs with src in (i-1, i+1): dst in ((i-0.5)/scale-0.5, (i+1.5)/scale-0.5); widen by 1 for rounding, exact test later
    lo = (int)floorf(((float)i - 0.5f) * inv_scale - 0.5f) - 1;
    hi = (int)ceilf(((float)i + 1.5f) * inv_scale - 0.5f) + 1;
    if (i == 0) lo = 0;           // clamped sources all map to 0
    lo = lo < 0 ? 0 : lo;
    hi = hi > out - 1 ? out - 1 : hi;
}
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, long long lddy, float* __restrict__ dx,
                                                           long long lddx, int h, int w, int H, int W, int C4, float sy, float sx,
                                                           long long total) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4) * 4;
        long long q = i / C4;
        const int ix = (int)(q % w); q /= w;
        const int iy = (int)(q % h);
        const long long n = q / h;
        int ylo, yhi, xlo, xhi;
        contrib_range(iy, 1.f / sy, H, ylo, yhi);
        contrib_range(ix, 1.f / sx, W, xlo, xhi);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int oy = ylo; oy <= yhi; ++oy) {
            const Lerp ly = lerp_src(oy, sy, h);
            const float wy = (ly.i0 == iy ? ly.w0 : 0.f) + (ly.i1 == iy ? ly.w1 : 0.f);
            if (wy == 0.f) continue;
            const float* row = dy + ((n * H + oy) * W) * lddy + c;
            for (int ox = xlo; ox <= xhi; ++ox) {
                const Lerp lx = lerp_src(ox, sx, w);
                const float wx = (lx.i0 == ix ? lx.w0 : 0.f) + (lx.i1 == ix ? lx.w1 : 0.f);
                if (wx == 0.f) continue;
                acc += (wy * wx) * ld4(row + (long long)ox * lddy);
            }
        }
        st4(dx + ((n * h + iy) * w + ix) * lddx + c, acc);
    }
}
extern "C" int sh_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int N, int h, int w, int H, int W, int C, int act_flags, void* stream) {
    if (!x || !y || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || ldx < C || ldy < C || (ldx & 3) || (ldy & 3)) return SH_EINVAL;
    const long long total = (long long)N * H * W * (C / 4);
    bilinear_fwd_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(x, ldx, y, ldy, h, w, H, W, C / 4, (float)h / (float)H, (float)w / (float)W, total, act_flags & 1);      // bit 0: y stored as bf16
    return sh_launch_status();
}
// Upsampling backward that reads dy ONCE.  The output rows whose upper source row (i0) is iy form one contiguous band; a block
// = (image, iy, 16 input columns, 64 channels) owns that band: thread = (input column, channel quad) first folds a row horizontally
// (its <= 2*scale contributing columns, 8 loads in flight), then adds the row into TWO accumulators -- weight w0 for input row iy,
// w1 for row iy+1 -- which go to p0[iy] and p1[iy]; dx[iy] = p0[iy] + p1[iy-1] (bilinear_bwd_combine_kernel).  The gather form above
// reads every dy element (2 x 2 =) 4 times (3.8x measured at the x8 decoder resize).
__global__ __launch_bounds__(256) void bilinear_bwd_rows_kernel(const float* __restrict__ dy, long long lddy, float* __restrict__ p0,
                                                                float* __restrict__ p1, int h, int w, int H, int W, int C, float sy, float sx) {
    const int t = threadIdx.x, cq = t & 15, il = t >> 4;
    const unsigned nch = (unsigned)((C + 63) / 64), nxb = (unsigned)((w + 15) / 16);
    unsigned b = blockIdx.x;
    const int c = (int)(b % nch) * 64 + cq * 4; b /= nch;
    const int ix = (int)(b % nxb) * 16 + il; b /= nxb;
    const int iy = (int)(b % (unsigned)h);
    const long long n = b / (unsigned)h;
    if (c >= C || ix >= w) return;                 // no barriers below
    int ylo, yhi, xlo, xhi;
    contrib_range(iy, 1.f / sy, H, ylo, yhi);
    contrib_range(ix, 1.f / sx, W, xlo, xhi);
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    for (int oy = ylo; oy <= yhi; ++oy) {
        const Lerp ly = lerp_src(oy, sy, h);
        if (ly.i0 != iy) continue;
        const float wy0 = ly.w0 + (ly.i1 == iy ? ly.w1 : 0.f), wy1 = ly.i1 != iy ? ly.w1 : 0.f;
        const float* row = dy + ((n * H + oy) * W) * lddy + c;
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        for (int ox0 = xlo; ox0 <= xhi; ox0 += 8) {
            f32x4 v[8];
            float wx[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ox = ox0 + k, oxc = ox <= xhi ? ox : xhi;
                const Lerp lx = lerp_src(oxc, sx, w);
                wx[k] = ox <= xhi ? (lx.i0 == ix ? lx.w0 : 0.f) + (lx.i1 == ix ? lx.w1 : 0.f) : 0.f;
                v[k] = ld4(row + (long long)oxc * lddy);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) r += wx[k] * v[k];
        }
        a0 += wy0 * r;
        a1 += wy1 * r;
    }
    const long long o = ((n * h + iy) * w + ix) * (long long)C + c;
    st4(p0 + o, a0);
    st4(p1 + o, a1);
}
__global__ __launch_bounds__(256) void bilinear_bwd_combine_kernel(const float* __restrict__ p0, const float* __restrict__ p1, float* __restrict__ dx,
                                                                   long long lddx, int h, int w, int C4, long long total) {
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4) * 4;
        const long long pix = i / C4;
        const int iy = (int)((pix / w) % h);
        f32x4 v = ld4(p0 + pix * C4 * 4 + c);
        if (iy > 0) v += ld4(p1 + (pix - w) * C4 * 4 + c);
        st4(dx + pix * lddx + c, v);
    }
}
extern "C" int64_t sh_bilinear_bwd_workspace(int N, int h, int w, int C) {
    if (N <= 0 || h <= 0 || w <= 0 || C <= 0) return SH_EINVAL;
    return (int64_t)2 * N * h * w * C * 4;
}
extern "C" int sh_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int h, int w, int H, int W, int C, float* workspace,
                               int64_t workspace_bytes, void* stream) {
    if (!dy || !dx || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || lddx < C || lddy < C || (lddx & 3) || (lddy & 3)) return SH_EINVAL;
    const long long total = (long long)N * h * w * (C / 4);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const long long nblk = (long long)N * h * sh_cdiv(w, 16) * sh_cdiv(C, 64);
    if (workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= sh_bilinear_bwd_workspace(N, h, w, C) && H >= 2 * h && W >= 2 * w &&
        nblk < (1LL << 31)) {
        float* p0 = workspace;
        float* p1 = workspace + (long long)N * h * w * C;
        bilinear_bwd_rows_kernel<<<(unsigned)nblk, 256, 0, (hipStream_t)stream>>>(dy, lddy, p0, p1, h, w, H, W, C, sy, sx);
        bilinear_bwd_combine_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(p0, p1, dx, lddx, h, w, C / 4, total);
        return sh_launch_status();
    }
    bilinear_bwd_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(dy, lddy, dx, lddx, h, w, H, W, C / 4, sy, sx, total);
    return sh_launch_status();
}

// ------------------------------------------------------------------------------------------ channel L2 normalise
// F.normalize(p=2, dim=1, eps=1e-12): y = x / max(||x||, eps).  One wave per pixel row.
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ norm,
                                                         long long M, int C) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = xr[c]; s += v * v; }
    s = wave_sum(s);
    const float nrm = fmaxf(sqrtf(s), 1e-12f);
    for (int c = lane; c < C; c += 64) y[row * C + c] = xr[c] / nrm;
    if (lane == 0) norm[row] = nrm;
}
// dx = (dy - y * <dy, y>) / norm       (for norm > eps; the clamped branch gives dx = dy / eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ norm, float* __restrict__ dx, long long M, int C) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float d = 0.f;
    for (int c = lane; c < C; c += 64) d += dy[row * C + c] * y[row * C + c];
    d = wave_sum(d);
    const float nrm = norm[row];
    const bool clamped = !(nrm > 1e-12f);
    for (int c = lane; c < C; c += 64) {
        const float g = dy[row * C + c];
        dx[row * C + c] = clamped ? g / nrm : (g - y[row * C + c] * d) / nrm;
    }
}
extern "C" int sh_l2norm_fwd(const float* x, float* y, float* norm, int64_t M, int C, void* stream) {
    if (!x || !y || !norm || M <= 0 || C <= 0) return SH_EINVAL;
    l2norm_fwd_kernel<<<(unsigned)sh_cdiv(M, 4), 256, 0, (hipStream_t)stream>>>(x, y, norm, M, C);
    return sh_launch_status();
}
extern "C" int sh_l2norm_bwd(const float* dy, const float* y, const float* norm, float* dx, int64_t M, int C, void* stream) {
    if (!dy || !y || !norm || !dx || M <= 0 || C <= 0) return SH_EINVAL;
    l2norm_bwd_kernel<<<(unsigned)sh_cdiv(M, 4), 256, 0, (hipStream_t)stream>>>(dy, y, norm, dx, M, C);
    return sh_launch_status();
}

// ---------------------------------------------------------------------------------------------- on-device input pipeline
// dataset/dataloader.py:49-63 after PIL decoding: [hflip] -> ToTensor (u8 / 255) -> Normalize((v - mean) / std), written
// straight into the stem's NHWC4 layout (4th channel 0); masks: F.interpolate(mode="nearest") [+ hflip] -> u8 labels.
__global__ __launch_bounds__(256) void ingest_image_kernel(const unsigned char* __restrict__ rgb, float* __restrict__ out,
                                                           const unsigned char* __restrict__ flip, long long NHW, int HW, int W,
                                                           float m0, float m1, float m2, float s0, float s1, float s2) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < NHW; i += (long long)gridDim.x * 256) {
        const long long n = i / HW;
        const int x = (int)(i % W);
        long long src = i;
        if (flip != nullptr && flip[n]) src = i - x + (W - 1 - x);
        const unsigned char* q = rgb + src * 3;
        f32x4 v;
        v[0] = ((float)q[0] / 255.0f - m0) / s0;
        v[1] = ((float)q[1] / 255.0f - m1) / s1;
        v[2] = ((float)q[2] / 255.0f - m2) / s2;
        v[3] = 0.f;
        st4(out + i * 4, v);
    }
}
extern "C" int sh_ingest_image_u8(const uint8_t* rgb, float* out, const uint8_t* flip, int N, int H, int W, const float* mean3_host,
                                  const float* std3_host, void* stream) {
    if (!rgb || !out || !mean3_host || !std3_host || N <= 0 || H <= 0 || W <= 0) return SH_EINVAL;
    if (std3_host[0] == 0.f || std3_host[1] == 0.f || std3_host[2] == 0.f) return SH_EINVAL;
    const long long nhw = (long long)N * H * W;
    long long g = sh_cdiv(nhw, 256);
    if (g > 16384) g = 16384;
    ingest_image_kernel<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(rgb, out, flip, nhw, H * W, W, mean3_host[0], mean3_host[1], mean3_host[2],
                                                                      std3_host[0], std3_host[1], std3_host[2]);
    return sh_launch_status();
}
template <typename T>
__global__ __launch_bounds__(256) void ingest_mask_kernel(const T* __restrict__ mask, unsigned char* __restrict__ out,
                                                          const unsigned char* __restrict__ flip, int N, int Hs, int Ws, int H, int W) {
    const float sy = (float)Hs / (float)H, sx = (float)Ws / (float)W;            // ATen nearest: src = min(floor(dst * in/out), in-1)
    const long long total = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int x = (int)(i % W);
        const long long r = i / W;
        const int y = (int)(r % H);
        const long long n = r / H;
        if (flip != nullptr && flip[n]) x = W - 1 - x;                           // resize first, then flip (dataloader.py:50-59)
        const int yy = min((int)floorf((float)y * sy), Hs - 1), xx = min((int)floorf((float)x * sx), Ws - 1);
        out[i] = (unsigned char)mask[(n * Hs + yy) * Ws + xx];
    }
}
extern "C" int sh_ingest_mask(const void* mask, int is_i64, uint8_t* out, const uint8_t* flip, int N, int Hs, int Ws, int H, int W,
                              void* stream) {
    if (!mask || !out || N <= 0 || Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0) return SH_EINVAL;
    long long g = sh_cdiv((long long)N * H * W, 256);
    if (g > 16384) g = 16384;
    if (is_i64) ingest_mask_kernel<long long><<<(unsigned)g, 256, 0, (hipStream_t)stream>>>((const long long*)mask, out, flip, N, Hs, Ws, H, W);
    else ingest_mask_kernel<unsigned char><<<(unsigned)g, 256, 0, (hipStream_t)stream>>>((const unsigned char*)mask, out, flip, N, Hs, Ws, H, W);
    return sh_launch_status();
}

// ------------------------------------------------------------------------------------------ PIL bilinear image resize (u8)
// `img.resize(size, Image.BILINEAR)` of dataset/dataloader.py:50.  Pillow's resampler (src/libImaging/Resample.c) is a separable
// triangle filter whose support grows with the downscale factor (antialiasing); each pass rounds to uint8.  The coefficient
// tables are host arithmetic in double, exactly Pillow's operation order (precompute_coeffs + normalize_coeffs_8bpc); the passes
// are integer: acc = 2^21 + sum pixel * k (22 fractional bits), >> 22, clip to 0..255 -- bit-identical to Pillow.
#define SH_RZ_BITS 22
extern "C" int sh_resize_bilinear_coeffs(int in_size, int out_size, int* bounds, int* kk, int kk_capacity) {
    if (in_size <= 0 || out_size <= 0) return SH_EINVAL;
    double scale, filterscale;
    filterscale = scale = (double)in_size / (double)out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    if (!bounds || !kk) return ksize;                         // size query
    if ((long long)ksize * out_size > kk_capacity) return SH_EINVAL;
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int* k = kk + (long long)xx * ksize;
        double w[64];
        if (xmax > 64) return SH_EINVAL;                      // downscale factors beyond 31x are not supported
        for (int x = 0; x < xmax; ++x) {
            double t = (x + xmin - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            w[x] = t < 1.0 ? 1.0 - t : 0.0;
            ww += w[x];
        }
        for (int x = 0; x < ksize; ++x) {
            if (x < xmax) {
                const double v = ww != 0.0 ? w[x] / ww : w[x];
                k[x] = v < 0 ? (int)(-0.5 + v * (1 << SH_RZ_BITS)) : (int)(0.5 + v * (1 << SH_RZ_BITS));
            } else k[x] = 0;
        }
        bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}
// one pass over interleaved RGB u8: AXIS 0 = along x ([N,H,W,3] -> [N,H,Wo,3]), 1 = along y ([N,H,W,3] -> [N,Ho,W,3]); thread = output pixel
template <int AXIS>
__global__ __launch_bounds__(256) void resize_u8_pass_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                             const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                             int H, int W, int Ho, int Wo, long long total) {
    GRID_STRIDE(i, total) {
        const int ox = (int)(i % Wo);
        const long long q = i / Wo;
        const int oy = (int)(q % Ho);
        const long long n = q / Ho;
        const int o = AXIS == 0 ? ox : oy;
        const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
        const int* k = kk + (long long)o * ksize;
        int a0 = 1 << (SH_RZ_BITS - 1), a1 = a0, a2 = a0;
        for (int t = 0; t < cnt; ++t) {
            const long long pix = AXIS == 0 ? (n * H + oy) * W + lo + t : (n * H + lo + t) * W + ox;
            const unsigned char* sp = src + pix * 3;
            const int kv = k[t];
            a0 += sp[0] * kv; a1 += sp[1] * kv; a2 += sp[2] * kv;
        }
        unsigned char* dp = dst + i * 3;
        dp[0] = (unsigned char)min(255, max(0, a0 >> SH_RZ_BITS));
        dp[1] = (unsigned char)min(255, max(0, a1 >> SH_RZ_BITS));
        dp[2] = (unsigned char)min(255, max(0, a2 >> SH_RZ_BITS));
    }
}
// src [N,H,W,3] -> dst [N,Ho,Wo,3]; tmp: N*H*Wo*3 bytes (horizontal pass first, as Pillow); tables from sh_resize_bilinear_coeffs
// uploaded by the caller (device int32): bounds_x [Wo][2], kk_x [Wo][ksize_x], bounds_y [Ho][2], kk_y [Ho][ksize_y].
extern "C" int sh_resize_bilinear_u8(const uint8_t* src, uint8_t* tmp, uint8_t* dst, int N, int H, int W, int Ho, int Wo,
                                     const int* bounds_x, const int* kk_x, int ksize_x, const int* bounds_y, const int* kk_y, int ksize_y,
                                     void* stream) {
    if (!src || !dst || N <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return SH_EINVAL;
    if ((Wo != W && (!bounds_x || !kk_x || ksize_x <= 0)) || (Ho != H && (!bounds_y || !kk_y || ksize_y <= 0))) return SH_EINVAL;
    if (Wo != W && Ho != H && !tmp) return SH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const uint8_t* cur = src;
    if (Wo != W) {
        uint8_t* out = Ho != H ? tmp : dst;
        const long long total = (long long)N * H * Wo;
        resize_u8_pass_kernel<0><<<grid_for(total), 256, 0, st>>>(cur, out, bounds_x, kk_x, ksize_x, H, W, H, Wo, total);
        cur = out;
    }
    if (Ho != H) {
        const long long total = (long long)N * Ho * Wo;
        resize_u8_pass_kernel<1><<<grid_for(total), 256, 0, st>>>(cur, dst, bounds_y, kk_y, ksize_y, H, Wo, Ho, Wo, total);
    } else if (Wo == W) {
        (void)hipMemcpyAsync(dst, src, (size_t)N * H * W * 3, hipMemcpyDeviceToDevice, st);
    }
    return sh_launch_status();
}
