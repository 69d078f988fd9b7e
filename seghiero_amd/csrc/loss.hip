// Fused loss / metric kernels of the SegHiero training step (gfx950).
//
//  * sh_hiera2_loss  : bilinear resize of the 1/4-resolution logits to the label grid (train.py:282-284) fused with
//                      the coarse-target lookup (hiera_triplet_loss.py:11-38), the sigmoid hierarchical BCE (:41-107)
//                      and both all-pixel-mean CE terms (:183-187).  The 218 MB full-resolution logit tensor of the
//                      reference is never materialised: forward = one pass producing 6 sums, backward = one gather-form
//                      pass producing d/d(low-res logits) directly (deterministic, no atomics).
//  * sh_ce_loss      : the same for the aux head's nn.CrossEntropyLoss(ignore_index=255) (train.py:309-313).
//  * sh_triplet_*    : tree-triplet (tree_triplet_loss.py:15-65 / rmi_tree_triplet_loss.py:14-70).
//  * sh_pixel_metrics: fine argmax, pixel accuracy counts (train.py:37-49, 381-385) and a confusion matrix.
// All are HBM/latency-bound reductions: wave shuffles -> LDS -> one partial per block -> f64 finalize.
#include "loss_common.h"

#define MAXB 32       // max coarse buckets
#define MAXF 64       // max fine classes

struct H2Tab {
    int nf, nc;
    int bs[MAXB], be[MAXB];
    signed char bucket_of[MAXF];
};

__device__ __forceinline__ int coarse_of(int f, const H2Tab& T) {
    int c = IGN;
    for (int i = 0; i < T.nc; ++i)
        if (f >= T.bs[i] && f < T.be[i]) c = i;     // later buckets overwrite, like the reference's sequential masked writes
    return c;
}

// Per-pixel 2-level terms.  out[0..3] = bce_fine, bce_coarse, ce_fine, ce_coarse.  With GRAD: g[j] += d/dz_j of
//   af*bce_fine + ac*bce_coarse + b*(ce_fine + ce_coarse).
template <int MAXC, bool GRAD>
__device__ __forceinline__ void hiera2_pixel(const float (&z)[MAXC], int f, int c, const H2Tab& T, float af, float ac, float b,
                                             float (&out)[4], float (&g)[MAXC]) {
    // Register arrays only take compile-time indices, so every label-dependent access is a select chain over the channels.
    // The chains are kept to the few that are needed -- pick(idx) reads p[idx], add_at(idx, v) adds to g[idx] -- and the
    // label-independent part of each term is one static loop; the order of every floating-point sum is the reference's.
    const float eps = 1e-8f;
    const int nf = T.nf, nc = T.nc;
    float p[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) p[j] = j < nf + nc ? sigmoidf_(z[j]) : 0.f;
    auto pick = [&](int idx) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < MAXC; ++j) v = (j == idx) ? p[j] : v;
        return v;
    };
    auto add_at = [&](int idx, float v) {
#pragma unroll
        for (int j = 0; j < MAXC; ++j) g[j] = (j == idx) ? g[j] + v : g[j];
    };
    out[0] = out[1] = out[2] = out[3] = 0.f;
    if (f != IGN) {
        int bi = -1;
#pragma unroll
        for (int k = 0; k < MAXC; ++k) bi = (k == f && k < nf) ? (int)T.bucket_of[k] : bi;
        const int tj = bi >= 0 ? nf + bi : -1;
        const float pf = pick(f), tt = bi >= 0 ? pick(tj) : 2.f;
        const bool s_is_min = (bi < 0) || (pf <= tt);
        const float m = s_is_min ? pf : tt;
        const float term_f = -logf(m + eps);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
            if (k < nf) {
                const float s = p[k], om = 1.f - s + eps;
                const bool isf = k == f;
                acc += isf ? term_f : -logf(om);
                if (GRAD) g[k] += isf ? 0.f : af / om * s * (1.f - s);
            }
        }
        if (GRAD) {
            const float d = -af / (m + eps);
            add_at(s_is_min ? f : tj, s_is_min ? d * pf * (1.f - pf) : d * tt * (1.f - tt));
        }
        out[0] = acc;
        out[2] = softmax_ce<MAXC, GRAD>(z, 0, nf, f, b, g);
    }
    if (c != IGN) {
        float acc = 0.f;
        for (int i = 0; i < nc; ++i) {
            const float ti = pick(nf + i);
            if (i == c) {
                acc += -logf(ti + eps);
                if (GRAD) add_at(nf + i, -ac / (ti + eps) * ti * (1.f - ti));
            } else {
                // max over [fine channels of bucket i ..., coarse channel i], first maximum wins (torch.max tie rule)
                float best = -INFINITY; int bj = -1;
                const int bs = T.bs[i], be = T.be[i];
#pragma unroll
                for (int j = 0; j < MAXC; ++j)
                    if (j < nf && j >= bs && j < be && p[j] > best) { best = p[j]; bj = j; }
                if (ti > best) { best = ti; bj = nf + i; }
                acc += -logf(1.f - best + eps);
                if (GRAD) add_at(bj, ac / (1.f - best + eps) * best * (1.f - best));
            }
        }
        out[1] = acc;
        out[3] = softmax_ce<MAXC, GRAD>(z, nf, nc, c, b, g);
    }
}

// ------------------------------------------------------------------------------------------ 2-level forward
// valid-label counts ahead of a forward that also emits the gradient (its normalisers): cnt[0] = #fine valid, cnt[1] = #coarse valid
__global__ __launch_bounds__(256) void label_counts_kernel(const uint8_t* __restrict__ labels, const H2Tab T, int with_coarse, long long total,
                                                           unsigned long long* __restrict__ cnt) {
    __shared__ unsigned int red[2][4];
    unsigned int a = 0, b = 0;
    // 16 labels per load, four loads per thread in flight (all-255 = ignore words beyond the end); the tail and unaligned maps byte-wise.
    // (One 4-byte load per pass, 16 dependent passes per thread: 27 us for a 4 MB map.)
    const bool al16 = ((uintptr_t)labels & 15) == 0;
    const long long n16 = al16 ? total / 16 : 0, gs = (long long)gridDim.x * 256;
    const long long n4 = 4 * n16;                               // (name kept for the tail loop below: labels covered = 4 * n4)
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n16; i0 += 4 * gs) {
        typedef unsigned int lab16_t __attribute__((ext_vector_type(4)));
        lab16_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = i0 + u * gs;
            v[u] = i < n16 ? reinterpret_cast<const lab16_t*>(labels)[i] : lab16_t{~0u, ~0u, ~0u, ~0u};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int f = (v[u][q] >> (8 * k)) & 255;
                    if (f != IGN) { ++a; if (with_coarse && coarse_of(f, T) != IGN) ++b; }
                }
    }
    for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int f = labels[i];
        if (f != IGN) { ++a; if (with_coarse && coarse_of(f, T) != IGN) ++b; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {          // one atomic per block and counter (8192 per-wave atomics on one address cost 80 us)
        atomicAdd(cnt + 0, (unsigned long long)(red[0][0] + red[0][1] + red[0][2] + red[0][3]));
        if (with_coarse) atomicAdd(cnt + 1, (unsigned long long)(red[1][0] + red[1][1] + red[1][2] + red[1][3]));
    }
}
// GRAD: every pixel's d(loss_out)/d(interpolated logits) goes to gfull [N*H*W][L] in the same pass (the backward is then only
// the adjoint of the resize); cnt = the label counts of label_counts_kernel.
template <int MAXC, bool GRAD>
__global__ __launch_bounds__(256) void hiera2_fwd_kernel(const float* __restrict__ logits, long long ldl, const uint8_t* __restrict__ labels,
                                                         const H2Tab T, float* __restrict__ partials, uint8_t* __restrict__ coarse_out,
                                                         int h, int w, int H, int W, float sy, float sx, long long total,
                                                         const unsigned long long* __restrict__ cnt, float* __restrict__ gfull, int L,
                                                         const long long* __restrict__ norm) {
    const bool identity = (h == H && w == W);
    const int C = T.nf + T.nc;
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float af = 0.f, ac = 0.f, b = 0.f;
    if (GRAD) {       // the coefficients of loss_grad_fullres_kernel at unit upstream gradient (norm: the all-reduced counts, see the entry point)
        const double c0 = norm ? (double)norm[0] : (double)cnt[0], c1 = norm ? (double)norm[1] : (double)cnt[1];
        const double nvf = c0 < 1 ? 1.0 : c0, nvc = c1 < 1 ? 1.0 : c1;
        af = (float)(5.0 / (nvf * T.nf));
        ac = T.nc > 0 ? (float)(5.0 / (nvc * T.nc)) : 0.f;
        b = (float)(1.0 / (norm ? (double)norm[2] : (double)total));
    }
    const long long base = (long long)blockIdx.x * LOSS_PIX_PER_BLOCK;
#pragma unroll 1
    for (int it = 0; it < LOSS_PIX_PER_BLOCK / 256; ++it) {
        const long long i = base + it * 256 + threadIdx.x;
        if (i >= total) break;
        const int f = labels[i];
        const int c = f == IGN ? IGN : coarse_of(f, T);
        if (coarse_out) coarse_out[i] = (uint8_t)c;
        float g[MAXC];
        if (GRAD) {
#pragma unroll
            for (int j = 0; j < MAXC; ++j) g[j] = 0.f;
        }
        if (f != IGN) {       // else coarse is 255 too (255 is in no bucket): nothing contributes
            const int ox = (int)(i % W);
            const long long q = i / W;
            const int oy = (int)(q % H);
            const long long n = q / H;
            const Lerp ly = lerp_src(oy, sy, h), lx = lerp_src(ox, sx, w);
            float z[MAXC], o[4];
            fetch_logits<MAXC>(logits + n * h * w * ldl, ldl, w, ly, lx, identity, C, z);
            hiera2_pixel<MAXC, GRAD>(z, f, c, T, af, ac, b, o, g);
            v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
            v[4] += 1.f; v[5] += (c != IGN) ? 1.f : 0.f;
        }
        if (GRAD) {
            float* dst = gfull + i * L;
#pragma unroll
            for (int j = 0; j < MAXC; j += 4)
                if (j < L) st4(dst + j, f32x4{g[j], g[j + 1], g[j + 2], g[j + 3]});
        }
    }
    block_reduce_store<6>(v, partials);
}
// sums[0..3] = the four sums, sums[4] = n_valid_fine, sums[5] = n_valid_coarse, sums[6] = n_pixels; loss_out = scalar loss
__global__ __launch_bounds__(256) void hiera2_finalize_kernel(const float* __restrict__ partials, int nblk, double npix, int nf, int nc,
                                                              double* __restrict__ sums, float* __restrict__ loss_out,
                                                              const long long* __restrict__ norm) {
    __shared__ double red[8][4];
    const int t = threadIdx.x;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = t; b < nblk; b += 256)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += (double)partials[(long long)b * 8 + j];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = wave_sum_d(a[j]);
    if ((t & 63) == 0)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[j][t >> 6] = a[j];
    __syncthreads();
    if (t == 0) {
        double s[8];
        for (int j = 0; j < 8; ++j) s[j] = (red[j][0] + red[j][1]) + (red[j][2] + red[j][3]);
        s[6] = npix;
        if (norm) { s[4] = (double)norm[0]; s[5] = (double)norm[1]; s[6] = npix = (double)norm[2]; }      // global denominators, local numerators
        for (int j = 0; j < 8; ++j) sums[j] = s[j];
        const double nvf = s[4] < 1.0 ? 1.0 : s[4], nvc = s[5] < 1.0 ? 1.0 : s[5];
        const double lf = s[0] / (nvf * nf), lc = nc > 0 ? s[1] / (nvc * nc) : 0.0;
        // the reference evaluates every term in f32; round each term to f32 before combining like it does
        const float hiera = 5.0f * ((float)lf + (float)lc);
        loss_out[0] = hiera + (float)(s[2] / npix) + (float)(s[3] / npix);
    }
}

// ------------------------------------------------------------------------------------------ 2-level backward (gather form)
template <int MAXC>
__global__ __launch_bounds__(256) void hiera2_bwd_kernel(const float* __restrict__ logits, long long ldl, const uint8_t* __restrict__ labels,
                                                         const H2Tab T, const double* __restrict__ sums, const float* __restrict__ gscale_dev,
                                                         float gscale, float* __restrict__ dlogits, long long lddl, int h, int w, int H,
                                                         int W, float sy, float sx, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const bool identity = (h == H && w == W);
    const int C = T.nf + T.nc;
    const int ix = (int)(i % w);
    const long long q = i / w;
    const int iy = (int)(q % h);
    const long long n = q / h;
    const float gs = gscale * (gscale_dev ? gscale_dev[0] : 1.f);
    const double nvf = sums[4] < 1.0 ? 1.0 : sums[4], nvc = sums[5] < 1.0 ? 1.0 : sums[5];
    const float af = gs * (float)(5.0 / (nvf * T.nf)), ac = T.nc > 0 ? gs * (float)(5.0 / (nvc * T.nc)) : 0.f;
    const float b = gs * (float)(1.0 / sums[6]);
    float acc[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) acc[j] = 0.f;
    int ylo, yhi, xlo, xhi;
    if (identity) { ylo = yhi = iy; xlo = xhi = ix; }
    else { contrib_range(iy, 1.f / sy, H, ylo, yhi); contrib_range(ix, 1.f / sx, W, xlo, xhi); }
    const float* base = logits + n * h * w * ldl;
    for (int oy = ylo; oy <= yhi; ++oy) {
        const Lerp ly = lerp_src(oy, sy, h);
        const float wy = identity ? 1.f : (ly.i0 == iy ? ly.w0 : 0.f) + (ly.i1 == iy ? ly.w1 : 0.f);
        if (wy == 0.f) continue;
        for (int ox = xlo; ox <= xhi; ++ox) {
            const Lerp lx = lerp_src(ox, sx, w);
            const float wx = identity ? 1.f : (lx.i0 == ix ? lx.w0 : 0.f) + (lx.i1 == ix ? lx.w1 : 0.f);
            if (wx == 0.f) continue;
            const int f = labels[(n * H + oy) * W + ox];
            if (f == IGN) continue;
            const int c = coarse_of(f, T);
            float z[MAXC], g[MAXC], o[4];
#pragma unroll
            for (int j = 0; j < MAXC; ++j) g[j] = 0.f;
            fetch_logits<MAXC>(base, ldl, w, ly, lx, identity, C, z);
            hiera2_pixel<MAXC, true>(z, f, c, T, af, ac, b, o, g);
            const float wgt = wy * wx;
#pragma unroll
            for (int j = 0; j < MAXC; ++j) acc[j] += wgt * g[j];
        }
    }
    float* dst = dlogits + ((n * h + iy) * w + ix) * lddl;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) if (j < lddl) dst[j] = j < C ? acc[j] : 0.f;   // padding lanes are zeroed (dgrad reads them)
}

// ------------------------------------------------------------------------------------------ tiled backward (both losses)
// One block = TL x TL low-resolution pixels of one image.  Phase A: every full-resolution pixel that can reference the
// tile (its bilinear footprint) gets its loss gradient computed ONCE into an LDS tile.  Phase B: each (low-res pixel,
// channel) gathers its footprint from LDS with the forward's exact weights, in a fixed order (deterministic, no atomics).
// Halo recompute is (TL*s+s+2)^2/(TL*s)^2 instead of the ~4x of a per-pixel gather, and the logits are read once.
template <int MAXC, int MODE>   // MODE 0: 2-level hiera loss, 1: CE (valid-pixel mean)
__global__ __launch_bounds__(256) void loss_bwd_tile_kernel(const float* __restrict__ logits, long long ldl, const uint8_t* __restrict__ labels,
                                                            const H2Tab T, int C, const double* __restrict__ sums,
                                                            const float* __restrict__ gscale_dev, float gscale, float* __restrict__ dlogits,
                                                            long long lddl, int h, int w, int H, int W, float sy, float sx, int TL,
                                                            int tiles_x, int tiles_y, int lds_cap) {
    extern __shared__ __attribute__((aligned(16))) float gt[];
    const int t = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y;
    const long long n = blockIdx.x / (tiles_x * tiles_y);
    const int i0 = ty * TL, i1 = min(i0 + TL, h) - 1, j0 = tx * TL, j1 = min(j0 + TL, w) - 1;
    const bool identity = (h == H && w == W);
    int ylo, yhi, xlo, xhi, tmp;
    if (identity) { ylo = i0; yhi = i1; xlo = j0; xhi = j1; }
    else {
        contrib_range(i0, 1.f / sy, H, ylo, tmp); contrib_range(i1, 1.f / sy, H, tmp, yhi);
        contrib_range(j0, 1.f / sx, W, xlo, tmp); contrib_range(j1, 1.f / sx, W, tmp, xhi);
    }
    const int RH = yhi - ylo + 1, RW = xhi - xlo + 1;
    if (RH * RW * C + 2 * (RH + RW) > lds_cap) return;       // cannot happen (host sizes lds_cap from the same bound)
    // source index / weight of every full-resolution row and column of the region, computed once per block (both phases
    // would otherwise redo lerp_src per pixel and per gathered tap): tab[k] = {i0, bits(w1)}; i1 = i0 + (i0 < in-1), w0 = 1-w1
    int* tabY = reinterpret_cast<int*>(gt + RH * RW * C);
    int* tabX = tabY + 2 * RH;
    for (int k = t; k < RH + RW; k += 256) {
        const bool isy = k < RH;
        const Lerp l = isy ? lerp_src(ylo + k, sy, h) : lerp_src(xlo + (k - RH), sx, w);
        int* dst = isy ? tabY + 2 * k : tabX + 2 * (k - RH);
        dst[0] = l.i0; dst[1] = __float_as_int(l.w1);
    }
    __syncthreads();
    auto lerp_tab = [&](const int* tab, int k, int in) {
        Lerp l;
        l.i0 = tab[2 * k]; l.w1 = __int_as_float(tab[2 * k + 1]);
        l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0); l.w0 = 1.f - l.w1;
        return l;
    };
    const float gs = gscale * (gscale_dev ? gscale_dev[0] : 1.f);
    float af = 0.f, ac = 0.f, b;
    if (MODE == 0) {
        const double nvf = sums[4] < 1.0 ? 1.0 : sums[4], nvc = sums[5] < 1.0 ? 1.0 : sums[5];
        af = gs * (float)(5.0 / (nvf * T.nf));
        ac = T.nc > 0 ? gs * (float)(5.0 / (nvc * T.nc)) : 0.f;
        b = gs * (float)(1.0 / sums[6]);
    } else {
        b = gs * (float)(1.0 / sums[1]);
    }
    const float* base = logits + n * h * w * ldl;
    // ---- phase A
    for (int idx = t; idx < RH * RW; idx += 256) {
        const int ry = idx / RW, rx = idx - ry * RW, oy = ylo + ry, ox = xlo + rx;
        float g[MAXC];
#pragma unroll
        for (int j = 0; j < MAXC; ++j) g[j] = 0.f;
        const int f = labels[(n * H + oy) * W + ox];
        if (f != IGN) {
            const Lerp ly = lerp_tab(tabY, ry, h), lx = lerp_tab(tabX, rx, w);
            float z[MAXC];
            fetch_logits<MAXC>(base, ldl, w, ly, lx, identity, C, z);
            if (MODE == 0) {
                float o[4];
                hiera2_pixel<MAXC, true>(z, f, coarse_of(f, T), T, af, ac, b, o, g);
            } else {
                softmax_ce<MAXC, true>(z, 0, C, f, b, g);
            }
        }
#pragma unroll
        for (int j = 0; j < MAXC; ++j) if (j < C) gt[idx * C + j] = g[j];
    }
    __syncthreads();
    // ---- phase B
    const int nrows = i1 - i0 + 1, ncols = j1 - j0 + 1, L = (int)lddl;
    for (int item = t; item < nrows * ncols * L; item += 256) {
        const int ch = item % L, pq = item / L;
        const int iy = i0 + pq / ncols, ix = j0 + pq % ncols;
        float acc = 0.f;
        if (ch < C) {
            int cylo, cyhi, cxlo, cxhi;
            if (identity) { cylo = cyhi = iy; cxlo = cxhi = ix; }
            else { contrib_range(iy, 1.f / sy, H, cylo, cyhi); contrib_range(ix, 1.f / sx, W, cxlo, cxhi); }
            for (int oy = cylo; oy <= cyhi; ++oy) {
                const Lerp ly = lerp_tab(tabY, oy - ylo, h);
                const float wy = identity ? 1.f : (ly.i0 == iy ? ly.w0 : 0.f) + (ly.i1 == iy ? ly.w1 : 0.f);
                if (wy == 0.f) continue;
                const float* row = gt + ((oy - ylo) * RW - xlo) * C + ch;
                for (int ox = cxlo; ox <= cxhi; ++ox) {
                    const Lerp lx = lerp_tab(tabX, ox - xlo, w);
                    const float wx = identity ? 1.f : (lx.i0 == ix ? lx.w0 : 0.f) + (lx.i1 == ix ? lx.w1 : 0.f);
                    if (wx != 0.f) acc += (wy * wx) * row[ox * C];
                }
            }
        }
        dlogits[((n * h + iy) * w + ix) * lddl + ch] = acc;          // padding lanes (ch >= C) are written as zeros
    }
}

// ------------------------------------------------------------------------------------------ two-pass backward through a workspace
// The fused tile kernel above recomputes a 39 x 39 full-resolution region per 32 x 32 owned pixels (x4 resize: +48 %), holds 79 KB
// of LDS per block (2 blocks per CU) and took 1.02 ms at the headline shape -- 10x the time its 16.8 MB output and 17.6 MB input
// justify.  With a workspace the same arithmetic runs as two streaming kernels:
//   A  every full-resolution pixel's gradient w.r.t. its (interpolated) logits, once, to workspace [N*H*W][lddl]   (coalesced 64 B / pixel)
//   B  the adjoint of the bilinear resize: each low-resolution (pixel, channel quad) gathers its <= 8 x 8 contributions in the
//      tile kernel's own (row, column) order with the forward's exact weights => bit-identical to the tile kernel.
// 268 MB written + re-read (mostly from L2) instead of recomputation: the full-resolution LOGITS still never exist.
template <int MAXC, int MODE>
__global__ __launch_bounds__(256) void loss_grad_fullres_kernel(const float* __restrict__ logits, long long ldl, const uint8_t* __restrict__ labels,
                                                                const H2Tab T, int C, const double* __restrict__ sums,
                                                                const float* __restrict__ gscale_dev, float gscale, float* __restrict__ gfull,
                                                                int L, int h, int w, int H, int W, float sy, float sx, long long total) {
    const bool identity = (h == H && w == W);
    const float gs = gscale * (gscale_dev ? gscale_dev[0] : 1.f);
    float af = 0.f, ac = 0.f, b;
    if (MODE == 0) {
        const double nvf = sums[4] < 1.0 ? 1.0 : sums[4], nvc = sums[5] < 1.0 ? 1.0 : sums[5];
        af = gs * (float)(5.0 / (nvf * T.nf));
        ac = T.nc > 0 ? gs * (float)(5.0 / (nvc * T.nc)) : 0.f;
        b = gs * (float)(1.0 / sums[6]);
    } else {
        b = gs * (float)(1.0 / sums[1]);
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float g[MAXC];
#pragma unroll
        for (int j = 0; j < MAXC; ++j) g[j] = 0.f;
        const int f = labels[i];
        if (f != IGN) {
            const int ox = (int)(i % W);
            const long long q = i / W;
            const int oy = (int)(q % H);
            const long long n = q / H;
            const Lerp ly = lerp_src(oy, sy, h), lx = lerp_src(ox, sx, w);
            float z[MAXC];
            fetch_logits<MAXC>(logits + n * h * w * ldl, ldl, w, ly, lx, identity, C, z);
            if (MODE == 0) {
                float o[4];
                hiera2_pixel<MAXC, true>(z, f, coarse_of(f, T), T, af, ac, b, o, g);
            } else {
                softmax_ce<MAXC, true>(z, 0, C, f, b, g);
            }
        }
        float* dst = gfull + i * L;
#pragma unroll
        for (int j = 0; j < MAXC; j += 4)
            if (j < L) st4(dst + j, f32x4{g[j], g[j + 1], g[j + 2], g[j + 3]});
    }
}
// adjoint of the resize: thread = (low-res pixel, channel quad).  A row's candidates are taken 8 at a time -- weights first, the 8
// loads issued together, then the sums in (row, column) order -- so a thread has 8 loads in flight instead of one (the serial form
// spent its time in L2 latency: 364 us for 268 MB).  post_scale: the workspace holds the gradient at unit upstream gradient
// (written by the forward), scaled here.
__global__ __launch_bounds__(256) void resize_adjoint_gather_kernel(const float* __restrict__ gfull, float* __restrict__ dlogits, int L, int h, int w,
                                                                    int H, int W, float sy, float sx, long long total,
                                                                    const float* __restrict__ gscale_dev, float gscale, int post_scale) {
    const int LQ = L / 4;
    const float gs = post_scale ? gscale * (gscale_dev ? gscale_dev[0] : 1.f) : 1.f;
    for (long long item = (long long)blockIdx.x * 256 + threadIdx.x; item < total; item += (long long)gridDim.x * 256) {
        const int cq = (int)(item % LQ);
        long long pq = item / LQ;
        const int ix = (int)(pq % w); pq /= w;
        const int iy = (int)(pq % h);
        const long long n = pq / h;
        int cylo, cyhi, cxlo, cxhi;
        contrib_range(iy, 1.f / sy, H, cylo, cyhi);
        contrib_range(ix, 1.f / sx, W, cxlo, cxhi);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int oy = cylo; oy <= cyhi; ++oy) {
            const Lerp ly = lerp_src(oy, sy, h);
            const float wy = (ly.i0 == iy ? ly.w0 : 0.f) + (ly.i1 == iy ? ly.w1 : 0.f);
            if (wy == 0.f) continue;
            const float* row = gfull + ((n * H + oy) * W) * L + 4 * cq;
            for (int ox0 = cxlo; ox0 <= cxhi; ox0 += 8) {
                f32x4 v[8];
                float wx[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int ox = ox0 + k, oxc = ox <= cxhi ? ox : cxhi;
                    const Lerp lx = lerp_src(oxc, sx, w);
                    wx[k] = ox <= cxhi ? (lx.i0 == ix ? lx.w0 : 0.f) + (lx.i1 == ix ? lx.w1 : 0.f) : 0.f;
                    v[k] = ld4(row + (long long)oxc * L);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f32x4 nx = acc + (wy * wx[k]) * v[k];
                    acc = wx[k] != 0.f ? nx : acc;
                }
            }
        }
        if (post_scale) acc = acc * gs;
        st4(dlogits + ((n * h + iy) * w + ix) * L + 4 * cq, acc);
    }
}
int sh_launch_resize_adjoint_gather(const float* gfull, float* dlogits, int lddl, int N, int h, int w, int H, int W, hipStream_t st) {
    const long long low = (long long)N * h * w * (lddl / 4);
    long long gb = sh_cdiv(low, 256);
    if (gb > 16384) gb = 16384;
    resize_adjoint_gather_kernel<<<(unsigned)gb, 256, 0, st>>>(gfull, dlogits, lddl, h, w, H, W, (float)h / (float)H, (float)w / (float)W, low,
                                                               nullptr, 1.f, 0);
    return sh_launch_status();
}
// the backward when the forward already left the per-pixel gradient (unit upstream gradient) in the workspace
static int launch_gather_from_grad(const float* gfull, const float* gscale_dev, float gscale, float* dlogits, int lddl, int N, int h, int w,
                                   int H, int W, hipStream_t st) {
    const long long low = (long long)N * h * w * (lddl / 4);
    long long gb = sh_cdiv(low, 256);
    if (gb > 16384) gb = 16384;
    resize_adjoint_gather_kernel<<<(unsigned)gb, 256, 0, st>>>(gfull, dlogits, lddl, h, w, H, W, (float)h / (float)H, (float)w / (float)W, low,
                                                               gscale_dev, gscale, 1);
    return sh_launch_status();
}
int sh_launch_gather_from_grad(const float* gfull, const float* gscale_dev, float gscale, float* dlogits, int lddl, int N, int h, int w,
                               int H, int W, hipStream_t st) {       // (loss3.hip: the 3-level loss's gather-only backward)
    return launch_gather_from_grad(gfull, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, st);
}
template <int MAXC, int MODE>
static int launch_loss_bwd_two_pass(const float* logits, int ldl, const uint8_t* labels, const H2Tab& T, int C, const double* sums,
                                    const float* gscale_dev, float gscale, float* dlogits, int lddl, int N, int h, int w, int H, int W,
                                    float* workspace, hipStream_t st) {
    const long long full = (long long)N * H * W, low = (long long)N * h * w * (lddl / 4);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    long long ga = sh_cdiv(full, 256), gb = sh_cdiv(low, 256);
    if (ga > 16384) ga = 16384;
    if (gb > 16384) gb = 16384;
    loss_grad_fullres_kernel<MAXC, MODE><<<(unsigned)ga, 256, 0, st>>>(logits, ldl, labels, T, C, sums, gscale_dev, gscale, workspace, lddl, h, w,
                                                                        H, W, sy, sx, full);
    resize_adjoint_gather_kernel<<<(unsigned)gb, 256, 0, st>>>(workspace, dlogits, lddl, h, w, H, W, sy, sx, low, nullptr, 1.f, 0);
    return sh_launch_status();
}
extern "C" int64_t sh_loss_bwd_workspace(int N, int H, int W, int lddl) {
    if (N <= 0 || H <= 0 || W <= 0 || lddl <= 0) return SH_EINVAL;
    return (int64_t)N * H * W * lddl * 4;
}

// host: pick TL (low-res tile side) so the LDS gradient tile stays <= budget; returns 0 if even TL = 1 does not fit
static int pick_tile(int h, int w, int H, int W, int C, int budget_bytes, int& region_elems) {
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
    const float s = sy > sx ? sy : sx;
    int best = 0;
    for (int TL = 1; TL <= 32; ++TL) {
        const int side = (h == H && w == W) ? TL : (int)ceilf(TL * s + s) + 3;
        const long long elems = (long long)side * side * C + 4 * side;      // gradient tile + the row / column lerp tables
        if (elems * 4 > budget_bytes) break;
        best = TL; region_elems = (int)elems;
    }
    return best;
}
template <int MAXC, int MODE>
static int launch_loss_bwd_tile(const float* logits, int ldl, const uint8_t* labels, const H2Tab& T, int C, const double* sums,
                                const float* gscale_dev, float gscale, float* dlogits, int lddl, int N, int h, int w, int H, int W,
                                int TL, int region_elems, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&loss_bwd_tile_kernel<MAXC, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_done = true;
    }
    const int tiles_x = (int)sh_cdiv(w, TL), tiles_y = (int)sh_cdiv(h, TL);
    const unsigned nblk = (unsigned)((long long)N * tiles_x * tiles_y);
    loss_bwd_tile_kernel<MAXC, MODE><<<nblk, 256, (size_t)region_elems * 4, st>>>(
        logits, ldl, labels, T, C, sums, gscale_dev, gscale, dlogits, lddl, h, w, H, W, (float)h / (float)H, (float)w / (float)W, TL,
        tiles_x, tiles_y, region_elems);
    return sh_launch_status();
}

extern "C" int sh_hiera2_partials(int N, int H, int W) { return (int)sh_cdiv((long long)N * H * W, LOSS_PIX_PER_BLOCK); }

static bool make_tab(H2Tab& T, const int* buckets, int nf, int nc) {
    if (nf <= 0 || nf > MAXF || nc < 0 || nc > MAXB || nf + nc > 32 || (nc > 0 && !buckets)) return false;
    T.nf = nf; T.nc = nc;
    for (int k = 0; k < MAXF; ++k) T.bucket_of[k] = -1;
    for (int i = 0; i < MAXB; ++i) { T.bs[i] = 0; T.be[i] = 0; }
    for (int i = 0; i < nc; ++i) {
        T.bs[i] = buckets[2 * i]; T.be[i] = buckets[2 * i + 1];
        for (int k = T.bs[i]; k < T.be[i] && k < nf; ++k) if (k >= 0) T.bucket_of[k] = (signed char)i;
    }
    return true;
}

// The loss normalisers of one shard as integers -- counts[0] = pixels with a valid fine label (hiera_triplet_loss.py:41-107 num_valid, also
// nn.CrossEntropyLoss's denominator), counts[1] = pixels whose coarse label is valid, counts[2] = all pixels (the all-pixel mean of
// models/loss/utils.py:20-21) -- for the EXACT data-parallel mode: the ranks all-reduce (sum) this 24-byte vector and hand the result to
// sh_hiera2_loss_fwd / sh_ce_loss_fwd (norm_counts), which then divide their LOCAL numerators by the GLOBAL denominators: the per-rank
// losses sum to the full-batch loss and the summed gradients are the full-batch gradients.
__global__ void label_counts_finish_kernel(long long* counts, long long total) { counts[2] = total; }
extern "C" int sh_label_counts(const uint8_t* labels, const int* buckets_host, int n_fine, int n_coarse, int64_t total, int64_t* counts,
                               void* stream) {
    H2Tab T;
    if (!labels || !counts || total <= 0 || !make_tab(T, buckets_host, n_fine, n_coarse)) return SH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, 3 * sizeof(int64_t), st) != hipSuccess) return SH_ELAUNCH;
    label_counts_kernel<<<256, 256, 0, st>>>(labels, T, 1, total, reinterpret_cast<unsigned long long*>(counts));
    label_counts_finish_kernel<<<1, 1, 0, st>>>(reinterpret_cast<long long*>(counts), (long long)total);
    return sh_launch_status();
}
// grad_out usable? (forward that also emits the per-pixel gradient for the gather-only backward)
static bool grad_out_ok(const float* grad_out, int64_t bytes, int ldg, int C, int N, int h, int w, int H, int W) {
    return grad_out && ldg >= C && ldg <= 32 && (ldg & 3) == 0 && ((uintptr_t)grad_out & 15) == 0 && bytes >= sh_loss_bwd_workspace(N, H, W, ldg) &&
           (h < H || w < W) && h <= H && w <= W;
}
extern "C" int sh_hiera2_loss_fwd(const float* logits, int ldl, const uint8_t* labels, const int* buckets_host, int n_fine,
                                  int n_coarse, double* sums, float* loss_out, float* partials, uint8_t* coarse_out,
                                  int N, int h, int w, int H, int W, float* grad_out, int64_t grad_out_bytes, int ldg,
                                  const int64_t* norm_counts, void* stream) {
    H2Tab T;
    if (!logits || !labels || !sums || !loss_out || !partials || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SH_EINVAL;
    if (!make_tab(T, buckets_host, n_fine, n_coarse) || ldl < n_fine + n_coarse) return SH_EINVAL;
    const long long* norm = reinterpret_cast<const long long*>(norm_counts);
    const long long total = (long long)N * H * W;
    const int nblk = (int)sh_cdiv(total, LOSS_PIX_PER_BLOCK);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipStream_t st = (hipStream_t)stream;
    const int C = n_fine + n_coarse;
    if (grad_out != nullptr) {
        if (!grad_out_ok(grad_out, grad_out_bytes, ldg, C, N, h, w, H, W)) return SH_EINVAL;
        // the gradient's normalisers are label counts: taken first, parked (as integers) where the finalize kernel later writes the same counts
        unsigned long long* cnt = reinterpret_cast<unsigned long long*>(sums + 4);
        if (norm == nullptr) {
            if (hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned long long), st) != hipSuccess) return SH_ELAUNCH;
            label_counts_kernel<<<256, 256, 0, st>>>(labels, T, 1, total, cnt);
        }
        if (C <= 8 && ldg <= 8) hiera2_fwd_kernel<8, true><<<nblk, 256, 0, st>>>(logits, ldl, labels, T, partials, coarse_out, h, w, H, W, sy, sx, total, cnt, grad_out, ldg, norm);
        else if (C <= 16 && ldg <= 16) hiera2_fwd_kernel<16, true><<<nblk, 256, 0, st>>>(logits, ldl, labels, T, partials, coarse_out, h, w, H, W, sy, sx, total, cnt, grad_out, ldg, norm);
        else hiera2_fwd_kernel<32, true><<<nblk, 256, 0, st>>>(logits, ldl, labels, T, partials, coarse_out, h, w, H, W, sy, sx, total, cnt, grad_out, ldg, norm);
    } else if (C <= 8) hiera2_fwd_kernel<8, false><<<nblk, 256, 0, st>>>(logits, ldl, labels, T, partials, coarse_out, h, w, H, W, sy, sx, total, nullptr, nullptr, 0, norm);
    else if (C <= 16) hiera2_fwd_kernel<16, false><<<nblk, 256, 0, st>>>(logits, ldl, labels, T, partials, coarse_out, h, w, H, W, sy, sx, total, nullptr, nullptr, 0, norm);
    else hiera2_fwd_kernel<32, false><<<nblk, 256, 0, st>>>(logits, ldl, labels, T, partials, coarse_out, h, w, H, W, sy, sx, total, nullptr, nullptr, 0, norm);
    int rc = sh_launch_status();
    if (rc != SH_OK) return rc;
    hiera2_finalize_kernel<<<1, 256, 0, st>>>(partials, nblk, (double)total, n_fine, n_coarse, sums, loss_out, norm);
    return sh_launch_status();
}

extern "C" int sh_hiera2_loss_bwd(const float* logits, int ldl, const uint8_t* labels, const int* buckets_host, int n_fine,
                                  int n_coarse, const double* sums, const float* gscale_dev, float gscale, float* dlogits,
                                  int lddl, int N, int h, int w, int H, int W, float* workspace, int64_t workspace_bytes,
                                  int workspace_has_grad, void* stream) {
    H2Tab T;
    if (!logits || !labels || !sums || !dlogits || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SH_EINVAL;
    if (!make_tab(T, buckets_host, n_fine, n_coarse) || ldl < n_fine + n_coarse || lddl < n_fine + n_coarse || lddl > 32) return SH_EINVAL;
    if (workspace_has_grad) {
        if (!grad_out_ok(workspace, workspace_bytes, lddl, n_fine + n_coarse, N, h, w, H, W) || ((uintptr_t)dlogits & 15)) return SH_EINVAL;
        return launch_gather_from_grad(workspace, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, (hipStream_t)stream);
    }
    const long long total = (long long)N * h * w;
    const unsigned nblk = (unsigned)sh_cdiv(total, 256);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipStream_t st = (hipStream_t)stream;
    if (workspace && workspace_bytes >= sh_loss_bwd_workspace(N, H, W, lddl) && (h < H || w < W) && h <= H && w <= W && (lddl & 3) == 0 &&
        (((uintptr_t)workspace | (uintptr_t)dlogits) & 15) == 0) {
        const int Cr = n_fine + n_coarse;
        if (Cr <= 8 && lddl <= 8) return launch_loss_bwd_two_pass<8, 0>(logits, ldl, labels, T, Cr, sums, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, workspace, st);
        if (Cr <= 16 && lddl <= 16) return launch_loss_bwd_two_pass<16, 0>(logits, ldl, labels, T, Cr, sums, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, workspace, st);
        return launch_loss_bwd_two_pass<32, 0>(logits, ldl, labels, T, Cr, sums, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, workspace, st);
    }
    {
        const int Cr = n_fine + n_coarse;
        int region = 0;
        const int TL = (h <= H && w <= W) ? pick_tile(h, w, H, W, Cr, 80 * 1024, region) : 0;
        if (TL > 0 && TL * TL * lddl >= 256) {
            if (Cr <= 8) return launch_loss_bwd_tile<8, 0>(logits, ldl, labels, T, Cr, sums, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, TL, region, st);
            if (Cr <= 16) return launch_loss_bwd_tile<16, 0>(logits, ldl, labels, T, Cr, sums, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, TL, region, st);
            return launch_loss_bwd_tile<32, 0>(logits, ldl, labels, T, Cr, sums, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, TL, region, st);
        }
    }
    const int C = lddl;   // fallback gather kernel writes lddl lanes (padding zeroed), so size the register arrays for it
    if (C <= 8) hiera2_bwd_kernel<8><<<nblk, 256, 0, st>>>(logits, ldl, labels, T, sums, gscale_dev, gscale, dlogits, lddl, h, w, H, W, sy, sx, total);
    else if (C <= 16) hiera2_bwd_kernel<16><<<nblk, 256, 0, st>>>(logits, ldl, labels, T, sums, gscale_dev, gscale, dlogits, lddl, h, w, H, W, sy, sx, total);
    else hiera2_bwd_kernel<32><<<nblk, 256, 0, st>>>(logits, ldl, labels, T, sums, gscale_dev, gscale, dlogits, lddl, h, w, H, W, sy, sx, total);
    return sh_launch_status();
}

// ------------------------------------------------------------------------------------------ aux CE (valid-pixel mean)
template <int MAXC, bool GRAD>      // GRAD: as hiera2_fwd_kernel (cnt[0] = number of valid pixels)
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, long long ldl, const uint8_t* __restrict__ labels, int C,
                                                     float* __restrict__ partials, int h, int w, int H, int W, float sy, float sx, long long total,
                                                     const unsigned long long* __restrict__ cnt, float* __restrict__ gfull, int L,
                                                     const long long* __restrict__ norm) {
    const bool identity = (h == H && w == W);
    float v[2] = {0.f, 0.f};
    const float b = GRAD ? (float)(1.0 / (norm ? (double)norm[0] : (double)cnt[0])) : 0.f;          // no valid pixel: inf -> NaN gradient, as the separate backward
    const long long base = (long long)blockIdx.x * LOSS_PIX_PER_BLOCK;
#pragma unroll 1
    for (int it = 0; it < LOSS_PIX_PER_BLOCK / 256; ++it) {
        const long long i = base + it * 256 + threadIdx.x;
        if (i >= total) break;
        const int f = labels[i];
        float g[MAXC];
        if (GRAD) {
#pragma unroll
            for (int j = 0; j < MAXC; ++j) g[j] = 0.f;
        }
        if (f != IGN) {
            const int ox = (int)(i % W);
            const long long q = i / W;
            const int oy = (int)(q % H);
            const long long n = q / H;
            const Lerp ly = lerp_src(oy, sy, h), lx = lerp_src(ox, sx, w);
            float z[MAXC];
            fetch_logits<MAXC>(logits + n * h * w * ldl, ldl, w, ly, lx, identity, C, z);
            v[0] += softmax_ce<MAXC, GRAD>(z, 0, C, f, b, g);
            v[1] += 1.f;
        }
        if (GRAD) {
            float* dst = gfull + i * L;
#pragma unroll
            for (int j = 0; j < MAXC; j += 4)
                if (j < L) st4(dst + j, f32x4{g[j], g[j + 1], g[j + 2], g[j + 3]});
        }
    }
    block_reduce_store<2>(v, partials);
}
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* __restrict__ partials, int nblk, double* __restrict__ sums, float* __restrict__ loss_out,
                                                          const long long* __restrict__ norm) {
    __shared__ double red[2][4];
    const int t = threadIdx.x;
    double a0 = 0, a1 = 0;
    for (int b = t; b < nblk; b += 256) { a0 += (double)partials[(long long)b * 8]; a1 += (double)partials[(long long)b * 8 + 1]; }
    a0 = wave_sum_d(a0); a1 = wave_sum_d(a1);
    if ((t & 63) == 0) { red[0][t >> 6] = a0; red[1][t >> 6] = a1; }
    __syncthreads();
    if (t == 0) {
        const double s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const double c = norm ? (double)norm[0] : (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);          // global valid count, local numerator
        sums[0] = s; sums[1] = c;
        loss_out[0] = (float)(s / c);      // 0/0 -> NaN like nn.CrossEntropyLoss on an all-ignored batch
    }
}
template <int MAXC>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long long ldl, const uint8_t* __restrict__ labels, int C,
                                                     const double* __restrict__ sums, const float* __restrict__ gscale_dev, float gscale,
                                                     float* __restrict__ dlogits, long long lddl, int h, int w, int H, int W, float sy, float sx,
                                                     long long total) {
    // one WAVE per low-res pixel: the contributing window (up to (2*16+2)^2 pixels for the x16 aux resize) is strided over lanes
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= total) return;
    const bool identity = (h == H && w == W);
    const int ix = (int)(i % w);
    const long long q = i / w;
    const int iy = (int)(q % h);
    const long long n = q / h;
    const float b = gscale * (gscale_dev ? gscale_dev[0] : 1.f) * (float)(1.0 / sums[1]);
    float acc[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) acc[j] = 0.f;
    int ylo, yhi, xlo, xhi;
    if (identity) { ylo = yhi = iy; xlo = xhi = ix; }
    else { contrib_range(iy, 1.f / sy, H, ylo, yhi); contrib_range(ix, 1.f / sx, W, xlo, xhi); }
    const int nx = xhi - xlo + 1, ncand = (yhi - ylo + 1) * nx;
    const float* base = logits + n * h * w * ldl;
    for (int k = lane; k < ncand; k += 64) {
        const int oy = ylo + k / nx, ox = xlo + k % nx;
        const Lerp ly = lerp_src(oy, sy, h), lx = lerp_src(ox, sx, w);
        const float wy = identity ? 1.f : (ly.i0 == iy ? ly.w0 : 0.f) + (ly.i1 == iy ? ly.w1 : 0.f);
        const float wx = identity ? 1.f : (lx.i0 == ix ? lx.w0 : 0.f) + (lx.i1 == ix ? lx.w1 : 0.f);
        const float wgt = wy * wx;
        if (wgt == 0.f) continue;
        const int f = labels[(n * H + oy) * W + ox];
        if (f == IGN) continue;
        float z[MAXC], g[MAXC];
#pragma unroll
        for (int j = 0; j < MAXC; ++j) g[j] = 0.f;
        fetch_logits<MAXC>(base, ldl, w, ly, lx, identity, C, z);
        softmax_ce<MAXC, true>(z, 0, C, f, b, g);
#pragma unroll
        for (int j = 0; j < MAXC; ++j) acc[j] += wgt * g[j];
    }
#pragma unroll
    for (int j = 0; j < MAXC; ++j) acc[j] = wave_sum(acc[j]);
    if (lane == 0) {
        float* dst = dlogits + ((n * h + iy) * w + ix) * lddl;
#pragma unroll
        for (int j = 0; j < MAXC; ++j) if (j < lddl) dst[j] = j < C ? acc[j] : 0.f;
    }
}
extern "C" int sh_ce_loss_fwd(const float* logits, int ldl, const uint8_t* labels, int C, double* sums, float* loss_out,
                              float* partials, int N, int h, int w, int H, int W, float* grad_out, int64_t grad_out_bytes, int ldg,
                              const int64_t* norm_count, void* stream) {
    if (!logits || !labels || !sums || !loss_out || !partials || C <= 0 || C > 32 || ldl < C || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SH_EINVAL;
    const long long* norm = reinterpret_cast<const long long*>(norm_count);
    const long long total = (long long)N * H * W;
    const int nblk = (int)sh_cdiv(total, LOSS_PIX_PER_BLOCK);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipStream_t st = (hipStream_t)stream;
    if (grad_out != nullptr) {
        if (!grad_out_ok(grad_out, grad_out_bytes, ldg, C, N, h, w, H, W)) return SH_EINVAL;
        unsigned long long* cnt = reinterpret_cast<unsigned long long*>(sums + 1);       // see sh_hiera2_loss_fwd
        if (norm == nullptr) {
            if (hipMemsetAsync(cnt, 0, sizeof(unsigned long long), st) != hipSuccess) return SH_ELAUNCH;
            label_counts_kernel<<<256, 256, 0, st>>>(labels, H2Tab{}, 0, total, cnt);
        }
        if (ldg <= 8) ce_fwd_kernel<8, true><<<nblk, 256, 0, st>>>(logits, ldl, labels, C, partials, h, w, H, W, sy, sx, total, cnt, grad_out, ldg, norm);
        else if (ldg <= 16) ce_fwd_kernel<16, true><<<nblk, 256, 0, st>>>(logits, ldl, labels, C, partials, h, w, H, W, sy, sx, total, cnt, grad_out, ldg, norm);
        else ce_fwd_kernel<32, true><<<nblk, 256, 0, st>>>(logits, ldl, labels, C, partials, h, w, H, W, sy, sx, total, cnt, grad_out, ldg, norm);
    } else if (C <= 8) ce_fwd_kernel<8, false><<<nblk, 256, 0, st>>>(logits, ldl, labels, C, partials, h, w, H, W, sy, sx, total, nullptr, nullptr, 0, norm);
    else if (C <= 16) ce_fwd_kernel<16, false><<<nblk, 256, 0, st>>>(logits, ldl, labels, C, partials, h, w, H, W, sy, sx, total, nullptr, nullptr, 0, norm);
    else ce_fwd_kernel<32, false><<<nblk, 256, 0, st>>>(logits, ldl, labels, C, partials, h, w, H, W, sy, sx, total, nullptr, nullptr, 0, norm);
    int rc = sh_launch_status();
    if (rc != SH_OK) return rc;
    ce_finalize_kernel<<<1, 256, 0, st>>>(partials, nblk, sums, loss_out, norm);
    return sh_launch_status();
}
extern "C" int sh_ce_loss_bwd(const float* logits, int ldl, const uint8_t* labels, int C, const double* sums, const float* gscale_dev,
                              float gscale, float* dlogits, int lddl, int N, int h, int w, int H, int W, float* workspace,
                              int64_t workspace_bytes, int workspace_has_grad, void* stream) {
    if (!logits || !labels || !sums || !dlogits || C <= 0 || C > 32 || ldl < C || lddl < C || lddl > 32 || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SH_EINVAL;
    if (workspace_has_grad) {
        if (!grad_out_ok(workspace, workspace_bytes, lddl, C, N, h, w, H, W) || ((uintptr_t)dlogits & 15)) return SH_EINVAL;
        return launch_gather_from_grad(workspace, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, (hipStream_t)stream);
    }
    const long long total = (long long)N * h * w;
    const unsigned nblk = (unsigned)sh_cdiv(total, 4);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipStream_t st = (hipStream_t)stream;
    if (workspace && workspace_bytes >= sh_loss_bwd_workspace(N, H, W, lddl) && (h < H || w < W) && h <= H && w <= W && (lddl & 3) == 0 &&
        (((uintptr_t)workspace | (uintptr_t)dlogits) & 15) == 0) {
        H2Tab T{};
        if (lddl <= 8) return launch_loss_bwd_two_pass<8, 1>(logits, ldl, labels, T, C, sums, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, workspace, st);
        if (lddl <= 16) return launch_loss_bwd_two_pass<16, 1>(logits, ldl, labels, T, C, sums, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, workspace, st);
        return launch_loss_bwd_two_pass<32, 1>(logits, ldl, labels, T, C, sums, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, workspace, st);
    }
    {
        H2Tab T{};
        int region = 0;
        const int TL = (h <= H && w <= W) ? pick_tile(h, w, H, W, C, 80 * 1024, region) : 0;
        // the tiled form needs enough (pixel, channel) items for its gather phase; at large resize factors (x16 aux head:
        // TL = 1) the wave-per-pixel gather below is the faster one
        if (TL > 0 && TL * TL * lddl >= 256) {
            if (C <= 8) return launch_loss_bwd_tile<8, 1>(logits, ldl, labels, T, C, sums, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, TL, region, st);
            if (C <= 16) return launch_loss_bwd_tile<16, 1>(logits, ldl, labels, T, C, sums, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, TL, region, st);
            return launch_loss_bwd_tile<32, 1>(logits, ldl, labels, T, C, sums, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, TL, region, st);
        }
    }
    if (lddl <= 8) ce_bwd_kernel<8><<<nblk, 256, 0, st>>>(logits, ldl, labels, C, sums, gscale_dev, gscale, dlogits, lddl, h, w, H, W, sy, sx, total);
    else if (lddl <= 16) ce_bwd_kernel<16><<<nblk, 256, 0, st>>>(logits, ldl, labels, C, sums, gscale_dev, gscale, dlogits, lddl, h, w, H, W, sy, sx, total);
    else ce_bwd_kernel<32><<<nblk, 256, 0, st>>>(logits, ldl, labels, C, sums, gscale_dev, gscale, dlogits, lddl, h, w, H, W, sy, sx, total);
    return sh_launch_status();
}

// ------------------------------------------------------------------------------------------ labels
__global__ __launch_bounds__(256) void labels_to_u8_kernel(const long long* __restrict__ in, uint8_t* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = (uint8_t)in[i];
}
extern "C" int sh_labels_to_u8(const int64_t* in, uint8_t* out, int64_t n, void* stream) {
    if (!in || !out || n <= 0) return SH_EINVAL;
    long long g = sh_cdiv(n, 256); if (g > 4096) g = 4096;
    labels_to_u8_kernel<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>((const long long*)in, out, n);
    return sh_launch_status();
}

// ------------------------------------------------------------------------------------------ tree-triplet
// workspace layout (ints): [0..M) u8 labels (packed, M bytes rounded to 4) | per class c in 0..255: rec[c] = {m, pad, pad, pad,
//   idxA[T], idxP[T], idxN[T], active[T]}  with T = TRIP_MAX.   class_loss: float[256] after the records.
#define TRIP_MAX 256
#define TRIP_REC (4 + 4 * TRIP_MAX)
struct TripWs { uint8_t* lab; int* rec; float* closs; };
__host__ __device__ inline TripWs trip_ws(void* ws, long long M) {
    TripWs t;
    t.lab = (uint8_t*)ws;
    const long long off = ((M + 15) / 16) * 16;
    t.rec = (int*)((uint8_t*)ws + off);
    t.closs = (float*)(t.rec + 256 * TRIP_REC);
    return t;
}
extern "C" int64_t sh_triplet_workspace(int64_t M) { return ((M + 15) / 16) * 16 + 256ll * TRIP_REC * 4 + 256 * 4 + 64; }

__global__ __launch_bounds__(256) void trip_labels_kernel(const uint8_t* __restrict__ labels, uint8_t* __restrict__ out, int h, int w, int H,
                                                          int W, float sy, float sx, long long M) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int x = (int)(i % w);
    const long long q = i / w;
    const int y = (int)(q % h);
    const long long n = q / h;
    int iy = (int)floorf((float)y * sy), ix = (int)floorf((float)x * sx);   // ATen nearest: min(floor(dst*scale), in-1)
    iy = iy > H - 1 ? H - 1 : iy; ix = ix > W - 1 ? W - 1 : ix;
    out[i] = labels[(n * H + iy) * W + ix];
}
// masks: [256][2][4] u64 (pos, neg membership bitsets per anchor class); anchor_ok: [4] u64.
__device__ __forceinline__ bool bit256(const unsigned long long* m, int v) { return (m[v >> 6] >> (v & 63)) & 1ull; }

__global__ __launch_bounds__(256) void trip_class_kernel(const float* __restrict__ emb, int D, const unsigned long long* __restrict__ masks,
                                                         const unsigned long long* __restrict__ anchor_ok, int max_triplet, float margin,
                                                         void* ws, long long M) {
    __shared__ int cnt[3];
    __shared__ int wave_off[3][4];
    __shared__ int idx[3][TRIP_MAX];
    __shared__ float wsum[4];
    const int cls = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    TripWs W = trip_ws(ws, M);
    int* rec = W.rec + (long long)cls * TRIP_REC;
    if (!bit256(anchor_ok, cls)) { if (t == 0) { rec[0] = 0; W.closs[cls] = 0.f; } return; }
    const unsigned long long* pm = masks + (long long)cls * 8;
    const unsigned long long* nm = pm + 4;
    if (t < 3) cnt[t] = 0;
    __syncthreads();
    // ordered compaction of the first max_triplet anchor / positive / negative rows (raster order)
    for (long long basei = 0; basei < M; basei += 256) {
        const long long i = basei + t;
        const int lab = i < M ? W.lab[i] : -1;
        bool fl[3];
        fl[0] = lab == cls;
        fl[1] = lab >= 0 && bit256(pm, lab);
        fl[2] = lab >= 0 && bit256(nm, lab);
        int pre[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned long long bal = __ballot(fl[k]);
            pre[k] = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wave_off[k][wv] = __popcll(bal);
        }
        __syncthreads();
        int basec[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int o = cnt[k];
            for (int v = 0; v < wv; ++v) o += wave_off[k][v];
            basec[k] = o;
            const int pos = o + pre[k];
            if (fl[k] && pos < max_triplet) idx[k][pos] = (int)i;
        }
        __syncthreads();
        if (t < 3) cnt[t] = cnt[t] + wave_off[t][0] + wave_off[t][1] + wave_off[t][2] + wave_off[t][3];
        __syncthreads();
        (void)basec;
        if (cnt[0] >= max_triplet && cnt[1] >= max_triplet && cnt[2] >= max_triplet) break;   // block-uniform
    }
    int m = min(min(cnt[0], cnt[1]), min(cnt[2], max_triplet));
    if (m == 0) { if (t == 0) { rec[0] = 0; W.closs[cls] = 0.f; } return; }
    float part = 0.f;
    // a wave takes triplets wv, wv + 4, ...; FOUR of them per pass with all twelve row loads of a 64-column slice issued before the first
    // use (indices clamped instead of branched, so nothing separates the loads): the embedding rows are cold in this XCD's L2 and one
    // triplet at a time paid three dependent misses per slice.  Per-triplet sums and the per-wave accumulation order are unchanged.
    for (int j0 = wv; j0 < m; j0 += 16) {
        const float* a[4]; const float* p[4]; const float* n[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = min(j0 + 4 * u, m - 1);
            a[u] = emb + (long long)idx[0][j] * D; p[u] = emb + (long long)idx[1][j] * D; n[u] = emb + (long long)idx[2][j] * D;
        }
        float dap[4] = {0.f, 0.f, 0.f, 0.f}, dan[4] = {0.f, 0.f, 0.f, 0.f};
        for (int d = lane; d < D; d += 64) {
            float av[4], pv[4], nv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = a[u][d]; pv[u] = p[u][d]; nv[u] = n[u][d]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { dap[u] += av[u] * pv[u]; dan[u] += av[u] * nv[u]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 4 * u;
            const float sp = wave_sum(dap[u]), sn = wave_sum(dan[u]);
            const float tl = (1.f - sp) - (1.f - sn) + margin;
            if (lane == 0 && j < m) {
                rec[4 + j] = idx[0][j]; rec[4 + TRIP_MAX + j] = idx[1][j]; rec[4 + 2 * TRIP_MAX + j] = idx[2][j];
                rec[4 + 3 * TRIP_MAX + j] = tl > 0.f ? 1 : 0;
                part += tl > 0.f ? tl : 0.f;
            }
        }
    }
    if (lane == 0) wsum[wv] = part;
    __syncthreads();
    if (t == 0) { rec[0] = m; W.closs[cls] = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) / (float)m; }
}
__global__ __launch_bounds__(256) void trip_finalize_kernel(void* ws, long long M, float* out) {
    // out[0] = mean over used classes (0 if none), out[1] = class_count.  256 threads fetch the 256 class records at once; thread 0 then
    // sums them in class order from LDS (one thread walking the 4 KB-strided records was 61 us of dependent loads)
    __shared__ float cl[256];
    __shared__ int used[256];
    TripWs W = trip_ws(ws, M);
    const int k = threadIdx.x;
    used[k] = W.rec[(long long)k * TRIP_REC] > 0;
    cl[k] = W.closs[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f; int c = 0;
        for (int q = 0; q < 256; ++q) if (used[q]) { s += cl[q]; ++c; }
        out[0] = c > 0 ? s / (float)c : 0.f;
        out[1] = (float)c;
    }
}
__global__ __launch_bounds__(256) void trip_bwd_kernel(const float* __restrict__ emb, int D, void* ws, long long M, const float* __restrict__ out,
                                                       const float* __restrict__ gscale_dev, float gscale, float* __restrict__ demb) {
    const int cls = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    TripWs W = trip_ws(ws, M);
    const int* rec = W.rec + (long long)cls * TRIP_REC;
    const int m = rec[0];
    if (m == 0) return;
    const float coef = gscale * (gscale_dev ? gscale_dev[0] : 1.f) / ((float)m * out[1]);
    // the wave's triplets are wv, wv + 4, ... (at most 64 with TRIP_MAX = 256): lane l first fetches the record of triplet wv + 4 l, so
    // the loop below reads records from registers (readlane) instead of paying a memory round trip per triplet before the row loads
    const int jl = wv + 4 * lane;
    int r_act = 0, r_a = 0, r_p = 0, r_n = 0;
    if (jl < m) { r_act = rec[4 + 3 * TRIP_MAX + jl]; r_a = rec[4 + jl]; r_p = rec[4 + TRIP_MAX + jl]; r_n = rec[4 + 2 * TRIP_MAX + jl]; }
    const int nl = (m - wv + 3) / 4;                          // triplets of this wave
    for (int l0 = 0; l0 < nl; l0 += 4) {                     // four per pass: their twelve row loads of a slice are in flight together
        long long ia[4], ip[4], in[4];
        float cf[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int l = min(l0 + u, nl - 1);
            ia[u] = __shfl(r_a, l, 64); ip[u] = __shfl(r_p, l, 64); in[u] = __shfl(r_n, l, 64);
            cf[u] = (l0 + u < nl && __shfl(r_act, l, 64)) ? coef : 0.f;          // inactive / padding triplets add exact zeros ... (skipped below)
        }
        for (int d = lane; d < D; d += 64) {
            float av[4], pv[4], nv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { av[u] = emb[ia[u] * D + d]; pv[u] = emb[ip[u] * D + d]; nv[u] = emb[in[u] * D + d]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (cf[u] != 0.f) {                           // wave-uniform
                    atomicAdd(demb + ia[u] * D + d, cf[u] * (nv[u] - pv[u]));
                    atomicAdd(demb + ip[u] * D + d, -cf[u] * av[u]);
                    atomicAdd(demb + in[u] * D + d, cf[u] * av[u]);
                }
            }
        }
    }
}
extern "C" int sh_triplet_fwd(const float* emb, const uint8_t* labels, const uint64_t* masks, const uint64_t* anchor_ok, int max_triplet,
                              float margin, float* out, void* workspace, int N, int h, int w, int D, int H, int W, void* stream) {
    if (!emb || !labels || !masks || !anchor_ok || !out || !workspace || N <= 0 || h <= 0 || w <= 0 || D <= 0 || H <= 0 || W <= 0) return SH_EINVAL;
    if (max_triplet <= 0 || max_triplet > TRIP_MAX) return SH_EINVAL;
    const long long M = (long long)N * h * w;
    hipStream_t st = (hipStream_t)stream;
    TripWs ws = trip_ws(workspace, M);
    trip_labels_kernel<<<(unsigned)sh_cdiv(M, 256), 256, 0, st>>>(labels, ws.lab, h, w, H, W, (float)H / (float)h, (float)W / (float)w, M);
    trip_class_kernel<<<256, 256, 0, st>>>(emb, D, (const unsigned long long*)masks, (const unsigned long long*)anchor_ok, max_triplet, margin, workspace, M);
    trip_finalize_kernel<<<1, 256, 0, st>>>(workspace, M, out);
    return sh_launch_status();
}
extern "C" int sh_triplet_bwd(const float* emb, const void* workspace, const float* out, const float* gscale_dev, float gscale, float* demb,
                              int N, int h, int w, int D, void* stream) {
    if (!emb || !workspace || !out || !demb || N <= 0 || h <= 0 || w <= 0 || D <= 0) return SH_EINVAL;
    const long long M = (long long)N * h * w;
    trip_bwd_kernel<<<256, 256, 0, (hipStream_t)stream>>>(emb, D, const_cast<void*>(workspace), M, out, gscale_dev, gscale, demb);
    return sh_launch_status();
}

// combine: out = (main + (min_count > 0 ? factor * trip : 0)) * loss_weight        (hiera_triplet_loss.py:200-211)
__global__ void combine_loss_kernel(const float* main_loss, const float* trip_out, const float* ready_count, float factor, float lw, float* out) {
    if (threadIdx.x == 0) {
        const float cnt = ready_count ? ready_count[0] : trip_out[1];
        out[0] = (main_loss[0] + (cnt > 0.f ? factor * trip_out[0] : 0.f)) * lw;
    }
}
extern "C" int sh_combine_loss(const float* main_loss, const float* trip_out, const float* ready_count, float factor, float loss_weight,
                               float* out, void* stream) {
    if (!main_loss || !trip_out || !out) return SH_EINVAL;
    combine_loss_kernel<<<1, 64, 0, (hipStream_t)stream>>>(main_loss, trip_out, ready_count, factor, loss_weight, out);
    return sh_launch_status();
}

// ------------------------------------------------------------------------------------------ metrics
template <int MAXC>
__global__ __launch_bounds__(256) void pixel_metrics_kernel(const float* __restrict__ logits, long long ldl, const uint8_t* __restrict__ labels,
                                                            int nf, unsigned long long* __restrict__ counts, int h, int w, int H, int W, float sy,
                                                            float sx, long long total) {
    __shared__ unsigned int hist[2 + 32 * 32];
    const bool identity = (h == H && w == W);
    for (int i = threadIdx.x; i < 2 + nf * nf; i += 256) hist[i] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * LOSS_PIX_PER_BLOCK;
#pragma unroll 1
    for (int it = 0; it < LOSS_PIX_PER_BLOCK / 256; ++it) {
        const long long i = base + it * 256 + threadIdx.x;
        if (i >= total) break;
        const int f = labels[i];
        if (f == IGN) continue;
        const int ox = (int)(i % W);
        const long long q = i / W;
        const int oy = (int)(q % H);
        const long long n = q / H;
        const Lerp ly = lerp_src(oy, sy, h), lx = lerp_src(ox, sx, w);
        float z[MAXC];
        fetch_logits<MAXC>(logits + n * h * w * ldl, ldl, w, ly, lx, identity, nf, z);
        int am = 0; float best = z[0];
#pragma unroll
        for (int j = 1; j < MAXC; ++j) if (j < nf && z[j] > best) { best = z[j]; am = j; }
        atomicAdd(&hist[1], 1u);
        if (am == f) atomicAdd(&hist[0], 1u);
        if (f < nf) atomicAdd(&hist[2 + f * nf + am], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 + nf * nf; i += 256)
        if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}
extern "C" int sh_pixel_metrics(const float* logits, int ldl, const uint8_t* labels, int n_fine, long long* counts, int N, int h, int w,
                                int H, int W, void* stream) {
    if (!logits || !labels || !counts || n_fine <= 0 || n_fine > 32 || ldl < n_fine || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SH_EINVAL;
    const long long total = (long long)N * H * W;
    const int nblk = (int)sh_cdiv(total, LOSS_PIX_PER_BLOCK);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* c = (unsigned long long*)counts;
    if (n_fine <= 8) pixel_metrics_kernel<8><<<nblk, 256, 0, st>>>(logits, ldl, labels, n_fine, c, h, w, H, W, sy, sx, total);
    else if (n_fine <= 16) pixel_metrics_kernel<16><<<nblk, 256, 0, st>>>(logits, ldl, labels, n_fine, c, h, w, H, W, sy, sx, total);
    else pixel_metrics_kernel<32><<<nblk, 256, 0, st>>>(logits, ldl, labels, n_fine, c, h, w, H, W, sy, sx, total);
    return sh_launch_status();
}
