// Depthwise 3x3 convolution (groups = C, stride 1, padding = dilation), NHWC, fp32.
// reference: DepthwiseSeparableConv.depthwise, models/head/sep_aspp_contrast_head.py:43-46,56 (the three ASPP
// branches with dilation 12/24/36 and the two sep_bottleneck convs with dilation 1).
//
// HBM-bound (AI 2.2 FLOP/B): one thread per (pixel, 4 channels), nine 16-byte taps, weights staged once per block
// in LDS as [tap][channel].  Taps that fall outside the image for EVERY pixel (dilation >= H and >= W, the
// centre-tap degeneration of SURVEY A.1) are skipped block-uniformly.  The forward also emits the per-channel
// (sum, sum^2) partials of its output for the following train-mode BatchNorm, so the tensor is not re-read.
#include "common.h"
#include <stdlib.h>

#define DW_PIX 64          // pixels per block
#define DW_CH 64           // channels per block (16 float4 lanes)

// input read through the producer's train-mode BatchNorm + ReLU (in_scale != nullptr): relu(x * scale + shift) in sh_bn_act's own
// operation order, applied to in-image taps only (zero padding stays zero) -- see sh_conv_fprop_x6_aff
__device__ __forceinline__ f32x4 dw_in(const float* p, const float* isc, const float* ish, int c) {
    f32x4 v = ld4(p);
    if (isc != nullptr) {
        v = v * ld4(isc + c) + ld4(ish + c);
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    return v;
}

// dgrad epilogue = front half of the BatchNorm backward of the layer that produced the conv's input (see sh_conv_dgrad_x6_bnb):
// g = relumask(y * scale + shift) * dx is stored and (sum g, sum g * xhat) per 64-pixel block go to `partials`
struct DwBnb { const float* y; long long ldy; const float* mean; const float* invstd; const float* scale; const float* shift; float* partials;
               int y_bf; };          // y_bf: y stored as bf16 (common.h lda4)
struct DwBnbC { f32x4 scale, shift, mean, invstd; };       // the four per-channel coefficients of this thread's channel quad, in registers
__device__ __forceinline__ DwBnbC dw_bnb_coefs(const DwBnb& b, int c, bool cok) {
    DwBnbC k; k.scale = k.shift = k.mean = k.invstd = f32x4{0.f, 0.f, 0.f, 0.f};
    if (b.y != nullptr && cok) { k.scale = ld4(b.scale + c); k.shift = ld4(b.shift + c); k.mean = ld4(b.mean + c); k.invstd = ld4(b.invstd + c); }
    return k;
}
__device__ __forceinline__ f32x4 dw_bnb_apply(const DwBnbC& k, f32x4 yv, f32x4 dx, f32x4& sg, f32x4& sq) {
    const f32x4 a = yv * k.scale + k.shift;                                  // the forward's own arithmetic (bn_act_kernel)
#pragma unroll
    for (int j = 0; j < 4; ++j) if (!(a[j] > 0.f)) dx[j] = 0.f;
    sg += dx;
    sq += dx * ((yv - k.mean) * k.invstd);
    return dx;
}
__device__ __forceinline__ f32x4 dw_bnb_apply(const DwBnb& b, long long m, int c, f32x4 dx, f32x4& sg, f32x4& sq) {
    const f32x4 yv = lda4(b.y, m * b.ldy + c, b.y_bf);
    const f32x4 a = yv * ld4(b.scale + c) + ld4(b.shift + c);                // the forward's own arithmetic (bn_act_kernel)
#pragma unroll
    for (int j = 0; j < 4; ++j) if (!(a[j] > 0.f)) dx[j] = 0.f;
    sg += dx;
    sq += dx * ((yv - ld4(b.mean + c)) * ld4(b.invstd + c));
    return dx;
}
// block-level sum over the 16 pixel lanes of the two statistics -> partials[pidx][2][C]   (red: [16][DW_CH] LDS scratch)
__device__ __forceinline__ void dw_bnb_store(const DwBnb& b, float (*red)[64], f32x4 sg, f32x4 sq, long long pidx, int C, int c0, int t, int cq, int pl) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) red[pl][cq * 4 + j] = pass == 0 ? sg[j] : sq[j];
        __syncthreads();
        if (t < 64 && c0 + t < C) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) a += red[k][t];
            b.partials[(pidx * 2 + pass) * C + c0 + t] = a;
        }
    }
}

extern "C" int sh_dw_partials(int N, int H, int W) { return (int)sh_cdiv((long long)N * H * W, DW_PIX); }
extern "C" int sh_dw_tile_rows(void) { return DW_PIX; }

// block: 256 threads = 16 channel-quads x 16 pixel lanes; each thread loops 4 pixels.
template <int MODE>   // 0 fprop (+stats), 1 dgrad (flipped taps)
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ w,
                                                     float* __restrict__ y, long long ldy, float* __restrict__ partials,
                                                     int H, int W, int C, int dil, long long M, int accumulate,
                                                     const float* __restrict__ isc, const float* __restrict__ ish, const DwBnb bnb) {
    __shared__ float ws[9][DW_CH];
    __shared__ float red[16][DW_CH];
    __shared__ float colmean[DW_CH];
    const int t = threadIdx.x, cq = t & 15, pl = t >> 4;
    const int c0 = blockIdx.y * DW_CH;
    for (int i = t; i < 9 * DW_CH; i += 256) {
        const int tap = i / DW_CH, cc = i % DW_CH;
        const int src_tap = MODE == 0 ? tap : 8 - tap;      // dgrad = correlation with the flipped kernel
        ws[tap][cc] = (c0 + cc < C) ? w[(long long)(c0 + cc) * 9 + src_tap] : 0.f;
    }
    __syncthreads();
    const int c = c0 + cq * 4;
    const bool cok = c < C;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    f32x4 bsg = s, bsq = s;
    f32x4 kept[DW_PIX / 16];
    bool kept_ok[DW_PIX / 16];
    const bool tap_row_ok = dil < H, tap_col_ok = dil < W;   // off-centre taps can touch the image at all?
#pragma unroll
    for (int it = 0; it < DW_PIX / 16; ++it) {
        const long long m = (long long)blockIdx.x * DW_PIX + it * 16 + pl;
        kept_ok[it] = false;
        kept[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < M && cok) {
            const int ow = (int)(m % W);
            const long long r = m / W;
            const int oh = (int)(r % H);
            const long long n = r / H;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                if (kh != 1 && !tap_row_ok) continue;
                const int ih = oh + (kh - 1) * dil;
                if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    if (kw != 1 && !tap_col_ok) continue;
                    const int iw = ow + (kw - 1) * dil;
                    if ((unsigned)iw >= (unsigned)W) continue;
                    const f32x4 v = dw_in(x + ((n * H + ih) * W + iw) * ldx + c, isc, ish, c);
                    const f32x4 wv = ld4(&ws[kh * 3 + kw][cq * 4]);
                    acc += v * wv;
                }
            }
            if (MODE == 1 && accumulate) acc += ld4(y + m * ldy + c);
            if (MODE == 1 && bnb.y != nullptr) acc = dw_bnb_apply(bnb, m, c, acc, bsg, bsq);
            st4(y + m * ldy + c, acc);
            s += acc; kept[it] = acc; kept_ok[it] = true;
        }
    }
    if (MODE == 1 && bnb.y != nullptr) dw_bnb_store(bnb, red, bsg, bsq, blockIdx.x, C, c0, t, cq, pl);
    if (MODE == 0 && partials != nullptr) {
        // per-block partial = (sum, M2 about the block's own mean) -- centred, see conv_gemm.hip
        const long long left = M - (long long)blockIdx.x * DW_PIX;
        const float nvalid = (float)(left < DW_PIX ? left : DW_PIX);
#pragma unroll
        for (int j = 0; j < 4; ++j) red[pl][cq * 4 + j] = s[j];
        __syncthreads();
        float colsum = 0.f;
        if (t < DW_CH) {
#pragma unroll
            for (int k = 0; k < 16; ++k) colsum += red[k][t];
            colmean[t] = colsum / nvalid;
        }
        __syncthreads();
        f32x4 q = {0.f, 0.f, 0.f, 0.f};
        const f32x4 mu = ld4(&colmean[cq * 4]);
#pragma unroll
        for (int it = 0; it < DW_PIX / 16; ++it)
            if (kept_ok[it]) { const f32x4 dv = kept[it] - mu; q += dv * dv; }
#pragma unroll
        for (int j = 0; j < 4; ++j) red[pl][cq * 4 + j] = q[j];
        __syncthreads();
        if (t < DW_CH && c0 + t < C) {
            float m2 = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) m2 += red[k][t];
            partials[((long long)blockIdx.x * 2 + 0) * C + c0 + t] = colsum;
            partials[((long long)blockIdx.x * 2 + 1) * C + c0 + t] = m2;
        }
    }
}

// ---------------------------------------------------------------------------------------------- 2-D tiled form (dilation 1)
// One block = an 8x8 pixel tile x 64 channels.  The 10x10 input halo is staged ONCE in LDS (25.6 KB), so every input
// element crosses L2->CU once instead of up to nine times; used when H and W are multiples of 8 (the decoder's 128x128).
#define DT 8
template <int MODE>   // 0 fprop (+stats), 1 dgrad (flipped taps, optional accumulate)
__global__ __launch_bounds__(256) void dwconv_tile_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ w,
                                                          float* __restrict__ y, long long ldy, float* __restrict__ partials,
                                                          int H, int W, int C, int accumulate,
                                                          const float* __restrict__ isc, const float* __restrict__ ish, const DwBnb bnb) {
    __shared__ __attribute__((aligned(16))) float xs[(DT + 2) * (DT + 2)][DW_CH];
    __shared__ float ws[9][DW_CH];
    __shared__ float red[16][DW_CH];
    __shared__ float colmean[DW_CH];
    const int t = threadIdx.x, cq = t & 15, pl = t >> 4;
    // 1-D grid, channel chunk fastest: the blocks that run together read adjacent 256-byte pieces of the same pixel rows
    const unsigned nch = (unsigned)((C + DW_CH - 1) / DW_CH);
    const unsigned tile = blockIdx.x / nch;
    const int c0 = (int)(blockIdx.x % nch) * DW_CH;
    const int tiles_x = W / DT, tiles_y = H / DT;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y;
    const long long n = tile / (tiles_x * tiles_y);
    for (int i = t; i < 9 * DW_CH; i += 256) {
        const int tap = i / DW_CH, cc = i % DW_CH;
        ws[tap][cc] = (c0 + cc < C) ? w[(long long)(c0 + cc) * 9 + (MODE == 0 ? tap : 8 - tap)] : 0.f;
    }
    const int c = c0 + cq * 4;
    const bool cok = c < C;
    for (int i = pl; i < (DT + 2) * (DT + 2); i += 16) {
        const int hy = i / (DT + 2), hx = i - hy * (DT + 2);
        const int iy = ty * DT + hy - 1, ix = tx * DT + hx - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (cok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = dw_in(x + ((n * H + iy) * W + ix) * ldx + c, isc, ish, c);
        st4(&xs[i][cq * 4], v);
    }
    __syncthreads();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    f32x4 bsg = s, bsq = s;
    f32x4 kept[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int pidx = it * 16 + pl, py = pidx / DT, px = pidx - py * DT;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                acc += ld4(&xs[(py + kh) * (DT + 2) + px + kw][cq * 4]) * ld4(&ws[kh * 3 + kw][cq * 4]);
        kept[it] = acc;
        if (cok) {
            const long long m = (n * H + ty * DT + py) * W + tx * DT + px;
            float* dst = y + m * ldy + c;
            if (MODE == 1 && accumulate) acc += ld4(dst);
            if (MODE == 1 && bnb.y != nullptr) acc = dw_bnb_apply(bnb, m, c, acc, bsg, bsq);
            st4(dst, acc);
        }
        s += kept[it];
    }
    if (MODE == 1 && bnb.y != nullptr) dw_bnb_store(bnb, red, bsg, bsq, tile, C, c0, t, cq, pl);
    if (MODE == 0 && partials != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[pl][cq * 4 + j] = s[j];
        __syncthreads();
        float colsum = 0.f;
        if (t < DW_CH) {
#pragma unroll
            for (int k = 0; k < 16; ++k) colsum += red[k][t];
            colmean[t] = colsum / (float)(DT * DT);
        }
        __syncthreads();
        f32x4 q = {0.f, 0.f, 0.f, 0.f};
        const f32x4 mu = ld4(&colmean[cq * 4]);
#pragma unroll
        for (int it = 0; it < 4; ++it) { const f32x4 dv = kept[it] - mu; q += dv * dv; }
#pragma unroll
        for (int j = 0; j < 4; ++j) red[pl][cq * 4 + j] = q[j];
        __syncthreads();
        if (t < DW_CH && c0 + t < C) {
            float m2 = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) m2 += red[k][t];
            partials[((long long)tile * 2 + 0) * C + c0 + t] = colsum;
            partials[((long long)tile * 2 + 1) * C + c0 + t] = m2;
        }
    }
}
// wgrad, tiled: each block walks tiles (grid-stride), x halo staged in LDS, nine accumulators per thread
__global__ __launch_bounds__(256) void dwconv_wgrad_tile_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ dy,
                                                               long long lddy, float* __restrict__ partials, int H, int W, int C,
                                                               long long ntiles, const float* __restrict__ isc, const float* __restrict__ ish) {
    __shared__ __attribute__((aligned(16))) float xs[(DT + 2) * (DT + 2)][DW_CH];
    __shared__ float red[16][DW_CH];
    const int t = threadIdx.x, cq = t & 15, pl = t >> 4;
    // 1-D grid, channel chunk fastest (see dwconv_tile_kernel): group = blockIdx.x / nch walks the tiles with stride ngroups
    const unsigned nch = (unsigned)((C + DW_CH - 1) / DW_CH), group = blockIdx.x / nch, ngroups = gridDim.x / nch;
    const int c0 = (int)(blockIdx.x % nch) * DW_CH, c = c0 + cq * 4;
    const bool cok = c < C;
    const int tiles_x = W / DT, tiles_y = H / DT;
    f32x4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long tile = group; tile < ntiles; tile += ngroups) {
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y);
        const long long n = tile / (tiles_x * tiles_y);
        __syncthreads();
        for (int i = pl; i < (DT + 2) * (DT + 2); i += 16) {
            const int hy = i / (DT + 2), hx = i - hy * (DT + 2);
            const int iy = ty * DT + hy - 1, ix = tx * DT + hx - 1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = dw_in(x + ((n * H + iy) * W + ix) * ldx + c, isc, ish, c);
            st4(&xs[i][cq * 4], v);
        }
        __syncthreads();
        if (cok) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pidx = it * 16 + pl, py = pidx / DT, px = pidx - py * DT;
                const f32x4 g = ld4(dy + ((n * H + ty * DT + py) * W + tx * DT + px) * lddy + c);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] += g * ld4(&xs[(py + kh) * (DT + 2) + px + kw][cq * 4]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) red[pl][cq * 4 + j] = acc[k][j];
        __syncthreads();
        if (t < DW_CH) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) a += red[r][t];
            if (c0 + t < C) partials[((long long)group * 9 + k) * C + c0 + t] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------- strip walk (dilation 1)
// The 8x8 tiles of one 64-channel chunk, ordered strip by strip (a strip = 8 rows of one image, left to right), are cut into
// contiguous runs, one per block (the launcher makes a run = a strip).  Inside a strip
// the input halo lives in an LDS ring of 10 columns x 10 rows (25.6 KB): a step replaces the 8 columns the finished tile no
// longer needs with the 8 NEW ones (10x8 instead of 10x10 pixel rows per tile), and they are fetched into registers BEFORE the
// current tile is computed, so the loads of tile t+1 are in flight under the LDS reads, FMAs and stores of tile t.  Channel chunk
// is fastest in the grid (8 chunks -> one chunk per XCD), so runs of neighbouring strips share their halo rows in that XCD's L2.
// Weights sit in registers.  Statistics partials keep the per-tile layout of dwconv_tile_kernel (one partial per 64 pixels).
#define DWR 10                    // ring columns: column ix lives in slot (ix + 1) % 10
__device__ __forceinline__ int dw_slot(int base, int k) { const int i = base + k; return i >= DWR ? i - DWR : i; }   // base < 10, k < 10
struct DwWalk {
    int n, ty, c0, c, tiles_x, tiles_y;
    bool cok;
    long long t0, t1;             // this block's run of tiles [t0, t1) of its chunk; tile t = strip * tiles_x + tx
};
__device__ __forceinline__ void dw_walk_init(DwWalk& q, int H, int W, int C, int N, int cq) {
    const unsigned nch = (unsigned)((C + DW_CH - 1) / DW_CH), nb = gridDim.x / nch, k = blockIdx.x / nch;
    q.c0 = (int)(blockIdx.x % nch) * DW_CH; q.c = q.c0 + cq * 4; q.cok = q.c < C;
    q.tiles_x = W / DT; q.tiles_y = H / DT;
    const long long T = (long long)N * q.tiles_y * q.tiles_x;
    q.t0 = T * k / nb; q.t1 = T * (k + 1) / nb;
}
__device__ __forceinline__ void dw_walk_strip(DwWalk& q, long long strip) { q.ty = (int)(strip % q.tiles_y); q.n = (int)(strip / q.tiles_y); }
__device__ __forceinline__ bool dw_walk_inside(const DwWalk& q, int H, int W, int hy, int ix) {
    return q.cok && (unsigned)(q.ty * DT + hy - 1) < (unsigned)H && (unsigned)ix < (unsigned)W;
}
// raw element (zero outside the image); the producer's BatchNorm + ReLU is applied by dw_walk_act when the value goes to LDS,
// i.e. AFTER the arithmetic of the current tile, so the prefetch stays in flight
__device__ __forceinline__ f32x4 dw_walk_fetch(const float* __restrict__ x, long long ldx, const DwWalk& q, int H, int W, int hy, int ix, int bf = 0) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (dw_walk_inside(q, H, W, hy, ix)) v = lda4(x, (((long long)q.n * H + q.ty * DT + hy - 1) * W + ix) * ldx + q.c, bf);
    return v;
}
// deferred BatchNorm-backward apply (dgrad / wgrad of a depthwise conv -> BN layer): the gradient operand is lin(g, y) =
// A*g + B*(y - mean) + D per channel (sh_bn_bwd_finalize's lin[4][C]) of the masked gradient g and the raw conv output y
struct DwLin { const float* y; long long ldy; const float* lin; int y_bf; };
struct DwLinC { f32x4 a, b, mu, d; };
__device__ __forceinline__ DwLinC dw_lin_coefs(const DwLin& L, int C, int c, bool cok) {
    DwLinC k; k.a = k.b = k.mu = k.d = f32x4{0.f, 0.f, 0.f, 0.f};
    if (L.lin != nullptr && cok) { k.a = ld4(L.lin + c); k.b = ld4(L.lin + C + c); k.mu = ld4(L.lin + 2 * C + c); k.d = ld4(L.lin + 3 * C + c); }
    return k;
}
__device__ __forceinline__ f32x4 dw_lin_eval(f32x4 g, f32x4 y, bool inside, const DwLinC& k) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = inside ? fmaf(y[e] - k.mu[e], k.b[e], fmaf(g[e], k.a[e], k.d[e])) : 0.f;      // padding stays zero
    return r;
}
__device__ __forceinline__ f32x4 dw_walk_act(f32x4 v, bool aff, bool inside, const f32x4& sc, const f32x4& sh) {
    if (aff && inside) {            // sh_bn_act's own operation order; zero padding stays zero
        v = v * sc + sh;
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    return v;
}
// columns 8tx-1 .. 8tx+8 of the strip -> ring slots rb .. rb+9 (start of a run / of a strip); 100 items over 16 pixel lanes
template <int LIN, int XBF, int YBF>       // compile-time: deferred-apply loader, element types of x and of the lin stream y
__device__ __forceinline__ void dw_walk_prologue(float (*xs)[DWR][DW_CH], const float* __restrict__ x, long long ldx, bool aff,
                                                 const f32x4& sc, const f32x4& sh, const DwWalk& q, int H, int W, int tx, int rb, int pl, int cq,
                                                 const DwLin* L = nullptr, const DwLinC* lc = nullptr) {
    for (int i = pl; i < (DT + 2) * (DT + 2); i += 16) {
        const int hy = i / (DT + 2), hx = i - hy * (DT + 2), ix = tx * DT + hx - 1;
        f32x4 v = dw_walk_fetch(x, ldx, q, H, W, hy, ix, XBF);
        if constexpr (LIN) v = dw_lin_eval(v, dw_walk_fetch(L->y, L->ldy, q, H, W, hy, ix, YBF), dw_walk_inside(q, H, W, hy, ix), *lc);
        st4(&xs[hy][dw_slot(rb, hx)][cq * 4], dw_walk_act(v, aff, dw_walk_inside(q, H, W, hy, ix), sc, sh));
    }
}
// MODE 0 fprop (+stats), 1 dgrad (flipped taps, optional accumulate / BatchNorm-backward epilogue).  Compile-time so that a tile's
// arithmetic is ONE basic block (runtime flags put a branch at every load and the scheduler then sank the arithmetic below the
// barrier with all 36 LDS operands live): LIN = deferred-apply loader (dgrad), BNB = BatchNorm-backward epilogue (dgrad),
// BF bits = element types -- fprop: bit 0 x, bit 1 y;  dgrad: bit 0 the lin stream y, bit 1 the epilogue's y_prev.
template <int MODE, int LIN, int BNB, int BF>
__global__ __launch_bounds__(256, (LIN && BNB) ? 2 : 3) void dwconv_walk_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ w,
                                                          float* __restrict__ y, long long ldy, float* __restrict__ partials,
                                                          int N, int H, int W, int C, int accumulate,
                                                          const float* __restrict__ isc, const float* __restrict__ ish, const DwBnb bnb,
                                                          const DwLin lin, int af) {      // af (fprop): bit 0 x, bit 1 y stored as bf16
    __shared__ __attribute__((aligned(16))) float xs[DT + 2][DWR][DW_CH];
    __shared__ float red[16][DW_CH];
    __shared__ float colmean[DW_CH];
    static_assert(MODE == 1 || (LIN == 0 && BNB == 0), "the loader / epilogue hooks belong to the dgrad");
    constexpr int x_bf = MODE == 0 ? (BF & 1) : 0, y_bf = MODE == 0 ? (BF & 2) : 0, ly_bf = MODE == 1 ? (BF & 1) : 0, by_bf = MODE == 1 ? (BF & 2) : 0;
    constexpr bool has_lin = LIN != 0;
    (void)af;
    const int t = threadIdx.x, cq = t & 15, pl = t >> 4;
    DwWalk q;
    dw_walk_init(q, H, W, C, N, cq);
    const DwLinC lc = dw_lin_coefs(lin, C, q.c, q.cok);
    const DwBnbC bc = dw_bnb_coefs(bnb, q.c, q.cok);
    f32x4 wr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[k][j] = q.cok ? w[(long long)(q.c + j) * 9 + (MODE == 0 ? k : 8 - k)] : 0.f;
    const bool aff = MODE == 0 && isc != nullptr;      // (the dgrad's operand is a gradient: no producer activation)
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
    if (aff && q.cok) { sc = ld4(isc + q.c); sh = ld4(ish + q.c); }
    int tx = (int)(q.t0 % q.tiles_x), rb = (tx * DT) % DWR;
    for (long long tile = q.t0; tile < q.t1; ++tile) {
        if (tile == q.t0 || tx == 0) {            // the run or a strip begins: the whole 10x10 halo (the loop ends on a barrier)
            dw_walk_strip(q, tile / q.tiles_x);
            dw_walk_prologue<LIN, x_bf, ly_bf>(xs, x, ldx, aff, sc, sh, q, H, W, tx, rb, pl, cq, &lin, &lc);
            __syncthreads();
        }
        // columns 8tx+9 .. 8tx+16 (tile tx+1's new ones): 80 items, 5 per thread, in flight during this tile's arithmetic
        f32x4 nx[5];
        [[maybe_unused]] f32x4 ny[5];
        const bool more = tx + 1 < q.tiles_x && tile + 1 < q.t1;
        if (more) {
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int i = it * 16 + pl;
                nx[it] = dw_walk_fetch(x, ldx, q, H, W, i >> 3, tx * DT + 9 + (i & 7), x_bf);
                if constexpr (has_lin) ny[it] = dw_walk_fetch(lin.y, lin.ldy, q, H, W, i >> 3, tx * DT + 9 + (i & 7), ly_bf);
            }
        }
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        f32x4 bsg = s, bsq = s;
        f32x4 kept[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pidx = it * 16 + pl, py = pidx / DT, px = pidx - py * DT;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    acc += ld4(&xs[py + kh][dw_slot(rb, px + kw)][cq * 4]) * wr[kh * 3 + kw];
            kept[it] = acc;
            if (q.cok) {
                const long long m = (((long long)q.n * H + q.ty * DT + py) * W + tx * DT + px);
                float* dst = y + m * ldy + q.c;
                if (MODE == 1 && accumulate) acc += ld4(dst);
                if constexpr (BNB) acc = dw_bnb_apply(bc, lda4(bnb.y, m * bnb.ldy + q.c, by_bf), acc, bsg, bsq);
                if (MODE == 0) sta4(y, m * ldy + q.c, acc, y_bf);          // (the statistics below use the fp32 values in `kept`)
                else st4(dst, acc);
            }
            s += kept[it];
        }
        if constexpr (BNB) dw_bnb_store(bnb, red, bsg, bsq, tile, C, q.c0, t, cq, pl);
        if (MODE == 0 && partials != nullptr) {
#pragma unroll
            for (int j = 0; j < 4; ++j) red[pl][cq * 4 + j] = s[j];
            __syncthreads();
            float colsum = 0.f;
            if (t < DW_CH) {
#pragma unroll
                for (int k = 0; k < 16; ++k) colsum += red[k][t];
                colmean[t] = colsum / (float)(DT * DT);
            }
            __syncthreads();
            f32x4 qq = {0.f, 0.f, 0.f, 0.f};
            const f32x4 mu = ld4(&colmean[cq * 4]);
#pragma unroll
            for (int it = 0; it < 4; ++it) { const f32x4 dv = kept[it] - mu; qq += dv * dv; }
#pragma unroll
            for (int j = 0; j < 4; ++j) red[pl][cq * 4 + j] = qq[j];
            __syncthreads();
            if (t < DW_CH && q.c0 + t < C) {
                float m2 = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) m2 += red[k][t];
                partials[(tile * 2 + 0) * C + q.c0 + t] = colsum;
                partials[(tile * 2 + 1) * C + q.c0 + t] = m2;
            }
        }
        __syncthreads();                 // every wave is done reading the ring columns the next step overwrites (and `red`)
        if (more) {
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int i = it * 16 + pl;
                const bool inside = dw_walk_inside(q, H, W, i >> 3, tx * DT + 9 + (i & 7));
                f32x4 v = nx[it];
                if constexpr (has_lin) v = dw_lin_eval(v, ny[it], inside, lc);
                st4(&xs[i >> 3][dw_slot(rb, i & 7)][cq * 4], dw_walk_act(v, aff, inside, sc, sh));
            }
            __syncthreads();
        }
        if (++tx == q.tiles_x) { tx = 0; rb = 0; } else rb = rb >= 2 ? rb - 2 : rb + 8;       // rb = (8 tx) % 10
    }
}
// wgrad, strip walk: x through the same ring, dy straight from global (prefetched one tile ahead), nine accumulators per thread;
// one partial [9][C] per block
template <int LIN, int BF>     // LIN: dy = lin(g, y) evaluated here;  BF bit 0: x, bit 1: the lin stream y stored as bf16
__global__ __launch_bounds__(256, 3) void dwconv_wgrad_walk_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ dy,
                                                               long long lddy, float* __restrict__ partials, int N, int H, int W, int C,
                                                               const float* __restrict__ isc, const float* __restrict__ ish, const DwLin lin) {
    constexpr int x_bf = BF & 1, ly_bf = BF & 2;
    __shared__ __attribute__((aligned(16))) float xs[DT + 2][DWR][DW_CH];
    __shared__ float red[16][DW_CH];
    const int t = threadIdx.x, cq = t & 15, pl = t >> 4;
    DwWalk q;
    dw_walk_init(q, H, W, C, N, cq);
    constexpr bool has_lin = LIN != 0;
    const DwLinC lc = dw_lin_coefs(lin, C, q.c, q.cok);
    f32x4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool aff = isc != nullptr;
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
    if (aff && q.cok) { sc = ld4(isc + q.c); sh = ld4(ish + q.c); }
    f32x4 g[4];
    [[maybe_unused]] f32x4 gy[4];         // dy of the next tile (raw g and, deferred apply, y: the linear form is evaluated when the tile is used)
    auto fetch_dy = [&](long long tile) {
        const long long strip = tile / q.tiles_x;
        const int ftx = (int)(tile - strip * q.tiles_x), fty = (int)(strip % q.tiles_y);
        const long long fn = strip / q.tiles_y;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pidx = it * 16 + pl, py = pidx / DT, px = pidx - py * DT;
            const long long m = (fn * H + fty * DT + py) * W + ftx * DT + px;
            g[it] = q.cok ? ld4(dy + m * lddy + q.c) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (has_lin) gy[it] = q.cok ? lda4(lin.y, m * lin.ldy + q.c, ly_bf) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    if (q.t0 < q.t1) fetch_dy(q.t0);
    int tx = (int)(q.t0 % q.tiles_x), rb = (tx * DT) % DWR;
    for (long long tile = q.t0; tile < q.t1; ++tile) {
        if (tile == q.t0 || tx == 0) {
            dw_walk_strip(q, tile / q.tiles_x);
            dw_walk_prologue<0, x_bf, 0>(xs, x, ldx, aff, sc, sh, q, H, W, tx, rb, pl, cq);
            __syncthreads();
        }
        f32x4 nx[5];
        const bool more = tx + 1 < q.tiles_x && tile + 1 < q.t1;
        if (more) {
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int i = it * 16 + pl;
                nx[it] = dw_walk_fetch(x, ldx, q, H, W, i >> 3, tx * DT + 9 + (i & 7), x_bf);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pidx = it * 16 + pl, py = pidx / DT, px = pidx - py * DT;
            f32x4 gv = g[it];
            if constexpr (has_lin) gv = dw_lin_eval(g[it], gy[it], q.cok, lc);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    acc[kh * 3 + kw] += gv * ld4(&xs[py + kh][dw_slot(rb, px + kw)][cq * 4]);
            // pin the nine accumulators here: left alone, the compiler sinks all 4 x 9 multiply-adds below the two barriers that
            // follow and keeps the 36 LDS operands (144 registers) alive until then -- 289-341 VGPRs, one block per CU
#pragma unroll
            for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(acc[k]));
        }
        if (tile + 1 < q.t1) fetch_dy(tile + 1);
        __syncthreads();
        if (more) {
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int i = it * 16 + pl;
                st4(&xs[i >> 3][dw_slot(rb, i & 7)][cq * 4],
                    dw_walk_act(nx[it], aff, dw_walk_inside(q, H, W, i >> 3, tx * DT + 9 + (i & 7)), sc, sh));
            }
            __syncthreads();
        }
        if (++tx == q.tiles_x) { tx = 0; rb = 0; } else rb = rb >= 2 ? rb - 2 : rb + 8;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) red[pl][cq * 4 + j] = acc[k][j];
        __syncthreads();
        if (t < DW_CH) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) a += red[r][t];
            if (q.c0 + t < C) partials[((long long)(blockIdx.x / ((C + DW_CH - 1) / DW_CH)) * 9 + k) * C + q.c0 + t] = a;
        }
    }
}

// wgrad: dw[c][tap] = sum_pix dy[pix][c] * x[pix + tap][c]; per-block partials [P][9][C], then a column reduce.
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ dy,
                                                           long long lddy, float* __restrict__ partials, int H, int W, int C,
                                                           int dil, long long M, const float* __restrict__ isc, const float* __restrict__ ish) {
    __shared__ float red[16][DW_CH];
    const int t = threadIdx.x, cq = t & 15, pl = t >> 4;
    const int c0 = blockIdx.y * DW_CH, c = c0 + cq * 4;
    const bool cok = c < C;
    f32x4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long nchunks = (M + DW_PIX - 1) / DW_PIX;
    for (long long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {      // grid-stride: few, fat partials
#pragma unroll
        for (int it = 0; it < DW_PIX / 16; ++it) {
            const long long m = chunk * DW_PIX + it * 16 + pl;
            if (m < M && cok) {
                const int ow = (int)(m % W);
                const long long r = m / W;
                const int oh = (int)(r % H);
                const long long n = r / H;
                const f32x4 g = ld4(dy + m * lddy + c);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int ih = oh + (kh - 1) * dil;
                    if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int iw = ow + (kw - 1) * dil;
                        if ((unsigned)iw >= (unsigned)W) continue;
                        acc[kh * 3 + kw] += g * dw_in(x + ((n * H + ih) * W + iw) * ldx + c, isc, ish, c);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) red[pl][cq * 4 + j] = acc[k][j];
        __syncthreads();
        if (t < DW_CH) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) a += red[r][t];
            if (c0 + t < C) partials[((long long)blockIdx.x * 9 + k) * C + c0 + t] = a;
        }
    }
}
// dw[c][tap] = sum_p partials[p][tap][c]   (f64 accumulate; one block per 64 (tap,c) columns, 4 row groups)
__global__ __launch_bounds__(256) void dwconv_wgrad_reduce_kernel(const float* __restrict__ partials, float* __restrict__ dw, int P, int C) {
    __shared__ double red[4][64];
    const int t = threadIdx.x, cl = t & 63, g = t >> 6;
    const int col = blockIdx.x * 64 + cl;       // col = tap*C + c
    double s = 0;
    if (col < 9 * C)
        for (int p = g; p < P; p += 4) s += (double)partials[(long long)p * 9 * C + col];
    red[g][cl] = s;
    __syncthreads();
    if (t < 64 && col < 9 * C) {
        const int tap = col / C, c = col % C;
        dw[(long long)c * 9 + tap] = (float)((red[0][t] + red[1][t]) + (red[2][t] + red[3][t]));
    }
}

// runtime flags -> the compile-time instantiations of the strip-walk kernels
template <typename... A>
static void dw_launch_fprop_walk(int bf, unsigned blocks, hipStream_t st, A... a) {
    switch (bf & 3) {
    case 0: dwconv_walk_kernel<0, 0, 0, 0><<<blocks, 256, 0, st>>>(a...); break;
    case 1: dwconv_walk_kernel<0, 0, 0, 1><<<blocks, 256, 0, st>>>(a...); break;
    case 2: dwconv_walk_kernel<0, 0, 0, 2><<<blocks, 256, 0, st>>>(a...); break;
    default: dwconv_walk_kernel<0, 0, 0, 3><<<blocks, 256, 0, st>>>(a...); break;
    }
}
template <int LIN, int BNB, typename... A>
static void dw_launch_dgrad_walk_bf(int bf, unsigned blocks, hipStream_t st, A... a) {
    switch (bf & 3) {
    case 0: dwconv_walk_kernel<1, LIN, BNB, 0><<<blocks, 256, 0, st>>>(a...); break;
    case 1: dwconv_walk_kernel<1, LIN, BNB, 1><<<blocks, 256, 0, st>>>(a...); break;
    case 2: dwconv_walk_kernel<1, LIN, BNB, 2><<<blocks, 256, 0, st>>>(a...); break;
    default: dwconv_walk_kernel<1, LIN, BNB, 3><<<blocks, 256, 0, st>>>(a...); break;
    }
}
template <typename... A>
static void dw_launch_dgrad_walk(bool lin, bool bnb, int bf, unsigned blocks, hipStream_t st, A... a) {
    if (!lin) bf &= ~1;
    if (!bnb) bf &= ~2;
    if (lin && bnb) dw_launch_dgrad_walk_bf<1, 1>(bf, blocks, st, a...);
    else if (lin) dw_launch_dgrad_walk_bf<1, 0>(bf, blocks, st, a...);
    else if (bnb) dw_launch_dgrad_walk_bf<0, 1>(bf, blocks, st, a...);
    else dw_launch_dgrad_walk_bf<0, 0>(bf, blocks, st, a...);
}
template <typename... A>
static void dw_launch_wgrad_walk(bool lin, int bf, unsigned blocks, hipStream_t st, A... a) {
    if (!lin) bf &= ~2;
    if (lin) {
        switch (bf & 3) {
        case 0: dwconv_wgrad_walk_kernel<1, 0><<<blocks, 256, 0, st>>>(a...); break;
        case 1: dwconv_wgrad_walk_kernel<1, 1><<<blocks, 256, 0, st>>>(a...); break;
        case 2: dwconv_wgrad_walk_kernel<1, 2><<<blocks, 256, 0, st>>>(a...); break;
        default: dwconv_wgrad_walk_kernel<1, 3><<<blocks, 256, 0, st>>>(a...); break;
        }
    } else if (bf & 1) dwconv_wgrad_walk_kernel<0, 1><<<blocks, 256, 0, st>>>(a...);
    else dwconv_wgrad_walk_kernel<0, 0><<<blocks, 256, 0, st>>>(a...);
}
// SEGHIERO_DW_WALK=0: the per-tile kernels (A/B timing)
static bool dw_walk_on() {
    static const bool on = [] { const char* e = getenv("SEGHIERO_DW_WALK"); return !(e && e[0] == '0'); }();
    return on;
}
// blocks per channel chunk of the walk kernels = strips.  (Measured: cutting the tile sequence into equal runs, one per resident
// block, is 10 % SLOWER -- runs that start mid-strip fall out of step with the strips above / below, whose halo rows they then miss
// in L2.)
static unsigned dw_walk_blocks(int N, int H, int W, int C) { (void)W; (void)C; return (unsigned)(N * (H / DT)); }
static bool dw_args_ok(const void* a, const void* b, const void* c, int N, int H, int W, int C, int dil, int ld1, int ld2) {
    return a && b && c && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && dil > 0 && ld1 >= C && ld2 >= C && !(ld1 & 3) && !(ld2 & 3);
}
// act_flags (include/seghiero_hip.h "bf16 ACTIVATION STORAGE"): the strip-walk kernels only -- SH_EUNSUPPORTED for other geometries
extern "C" int sh_dwconv_fprop(const float* x, int ldx, const float* in_scale, const float* in_shift, const float* w, float* y, int ldy,
                               float* stat_partials, int N, int H, int W, int C, int dil, int act_flags, void* stream) {
    if (!dw_args_ok(x, w, y, N, H, W, C, dil, ldx, ldy) || (act_flags & ~3)) return SH_EINVAL;
    if (act_flags && !(dil == 1 && H % DT == 0 && W % DT == 0 && dw_walk_on())) return SH_EUNSUPPORTED;
    if ((in_scale == nullptr) != (in_shift == nullptr) || (((uintptr_t)in_scale | (uintptr_t)in_shift) & 15)) return SH_EINVAL;
    const long long M = (long long)N * H * W;
    dim3 grid((unsigned)sh_cdiv(M, DW_PIX), (unsigned)sh_cdiv(C, DW_CH));
    if (dil == 1 && H % DT == 0 && W % DT == 0 && dw_walk_on())       // same partial count: (H/8)*(W/8) tiles of 64 pixels per image
        dw_launch_fprop_walk(act_flags, dw_walk_blocks(N, H, W, C) * grid.y, (hipStream_t)stream, x, (long long)ldx, w, y, (long long)ldy, stat_partials, N, H, W, C, 0, in_scale,
                             in_shift, DwBnb{}, DwLin{}, act_flags);
    else if (dil == 1 && H % DT == 0 && W % DT == 0)
        dwconv_tile_kernel<0><<<grid.x * grid.y, 256, 0, (hipStream_t)stream>>>(x, ldx, w, y, ldy, stat_partials, H, W, C, 0, in_scale, in_shift, DwBnb{});
    else
        dwconv_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(x, ldx, w, y, ldy, stat_partials, H, W, C, dil, M, 0, in_scale, in_shift, DwBnb{});
    return sh_launch_status();
}
static bool dw_lin_args_ok(const float* y_lin, int ldyl, const float* lin, int C) {
    return (lin == nullptr && y_lin == nullptr) || (lin && y_lin && ldyl >= C && !(ldyl & 3) && !(((uintptr_t)y_lin | (uintptr_t)lin) & 15));
}
static int dw_dgrad_any(const float* dy, int lddy, const float* w, float* dx, int lddx, int N, int H, int W, int C, int dil, int accumulate,
                        const DwBnb& bnb, void* stream, const DwLin lin = DwLin{}) {
    const long long M = (long long)N * H * W;
    dim3 grid((unsigned)sh_cdiv(M, DW_PIX), (unsigned)sh_cdiv(C, DW_CH));
    if (dil == 1 && H % DT == 0 && W % DT == 0 && dw_walk_on())
        dw_launch_dgrad_walk(lin.lin != nullptr, bnb.y != nullptr, (lin.y_bf ? 1 : 0) | (bnb.y_bf ? 2 : 0), dw_walk_blocks(N, H, W, C) * grid.y, (hipStream_t)stream,
                             dy, (long long)lddy, w, dx, (long long)lddx, (float*)nullptr, N, H, W, C, accumulate, (const float*)nullptr, (const float*)nullptr, bnb, lin, 0);
    else if (lin.lin != nullptr || bnb.y_bf) return SH_EUNSUPPORTED;          // the deferred-apply loader and bf16 streams exist in the strip-walk kernels only
    else if (dil == 1 && H % DT == 0 && W % DT == 0)
        dwconv_tile_kernel<1><<<grid.x * grid.y, 256, 0, (hipStream_t)stream>>>(dy, lddy, w, dx, lddx, nullptr, H, W, C, accumulate, nullptr, nullptr, bnb);
    else
        dwconv_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(dy, lddy, w, dx, lddx, nullptr, H, W, C, dil, M, accumulate, nullptr, nullptr, bnb);
    return sh_launch_status();
}
// y_lin / lin (optional, both or neither): deferred BatchNorm-backward apply -- `dy` holds the masked gradient g and the loader evaluates
// dy = lin[0]*g + lin[1]*(y_lin - lin[2]) + lin[3] (sh_bn_bwd_finalize's lin[4][C]); dil == 1 and H, W multiples of 8, else SH_EUNSUPPORTED
extern "C" int sh_dwconv_dgrad(const float* dy, int lddy, const float* y_lin, int ldyl, const float* lin, const float* w, float* dx, int lddx,
                               int N, int H, int W, int C, int dil, int accumulate, int act_flags, void* stream) {
    if (!dw_args_ok(dy, w, dx, N, H, W, C, dil, lddy, lddx) || !dw_lin_args_ok(y_lin, ldyl, lin, C) || (act_flags & ~1)) return SH_EINVAL;
    return dw_dgrad_any(dy, lddy, w, dx, lddx, N, H, W, C, dil, accumulate, DwBnb{}, stream, DwLin{y_lin, ldyl, lin, act_flags & 1});
}
// ... with the front half of the producer layer's BatchNorm backward in the epilogue (see sh_conv_dgrad_x6_bnb): g <- relumask * dx,
// stat_partials[sh_dw_partials(N,H,W)][2][C] <- (sum g, sum g * xhat) per 64-pixel block.  y_prev: raw output of the producer conv.
extern "C" int sh_dwconv_dgrad_bnb(const float* dy, int lddy, const float* y_lin, int ldyl, const float* lin, const float* w, float* g, int ldg,
                                   const float* y_prev, int ldyp, const float* mean, const float* invstd, const float* scale, const float* shift,
                                   float* stat_partials, int N, int H, int W, int C, int dil, int act_flags, void* stream) {
    if (!dw_args_ok(dy, w, g, N, H, W, C, dil, lddy, ldg) || !y_prev || !mean || !invstd || !scale || !shift || !stat_partials ||
        !dw_lin_args_ok(y_lin, ldyl, lin, C) || (act_flags & ~3)) return SH_EINVAL;
    if (ldyp < C || (ldyp & 3) || (((uintptr_t)y_prev | (uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)scale | (uintptr_t)shift) & 15)) return SH_EINVAL;
    return dw_dgrad_any(dy, lddy, w, g, ldg, N, H, W, C, dil, 0, DwBnb{y_prev, ldyp, mean, invstd, scale, shift, stat_partials, (act_flags >> 1) & 1}, stream,
                        DwLin{y_lin, ldyl, lin, act_flags & 1});          // act_flags: bit 0 y_lin, bit 1 y_prev stored as bf16
}
extern "C" int sh_dwconv_wgrad(const float* x, int ldx, const float* in_scale, const float* in_shift, const float* dy, int lddy,
                               const float* y_lin, int ldyl, const float* lin, float* dw_partials, float* dw, int N, int H, int W, int C, int dil,
                               int act_flags, void* stream) {
    if (!dw_args_ok(x, dy, dw, N, H, W, C, dil, ldx, lddy) || !dw_partials || !dw_lin_args_ok(y_lin, ldyl, lin, C) || (act_flags & ~3)) return SH_EINVAL;
    if (act_flags && !(dil == 1 && H % DT == 0 && W % DT == 0 && dw_walk_on())) return SH_EUNSUPPORTED;      // bit 0 x, bit 1 y_lin stored as bf16
    if ((in_scale == nullptr) != (in_shift == nullptr) || (((uintptr_t)in_scale | (uintptr_t)in_shift) & 15)) return SH_EINVAL;
    const long long M = (long long)N * H * W;
    int P = (int)sh_cdiv(M, DW_PIX);
    const int cap = (int)sh_cdiv(1024, sh_cdiv(C, DW_CH));          // ~4 blocks per CU over all channel chunks
    if (P > cap) P = cap < 1 ? 1 : cap;
    dim3 grid((unsigned)P, (unsigned)sh_cdiv(C, DW_CH));
    if (dil == 1 && H % DT == 0 && W % DT == 0 && dw_walk_on()) {
        P = (int)dw_walk_blocks(N, H, W, C);                        // one partial per block (<= M/64 rows of the workspace)
        dw_launch_wgrad_walk(lin != nullptr, act_flags & 3, (unsigned)P * grid.y, (hipStream_t)stream, x, (long long)ldx, dy, (long long)lddy, dw_partials, N, H, W, C, in_scale,
                             in_shift, DwLin{y_lin, ldyl, lin, (act_flags >> 1) & 1});
    } else if (lin != nullptr) return SH_EUNSUPPORTED;
    else if (dil == 1 && H % DT == 0 && W % DT == 0)
        dwconv_wgrad_tile_kernel<<<grid.x * grid.y, 256, 0, (hipStream_t)stream>>>(x, ldx, dy, lddy, dw_partials, H, W, C, M / (DT * DT), in_scale, in_shift);
    else
        dwconv_wgrad_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, ldx, dy, lddy, dw_partials, H, W, C, dil, M, in_scale, in_shift);
    int rc = sh_launch_status();
    if (rc != SH_OK) return rc;
    dwconv_wgrad_reduce_kernel<<<(unsigned)sh_cdiv(9 * C, 64), 256, 0, (hipStream_t)stream>>>(dw_partials, dw, P, C);
    return sh_launch_status();
}
