// 3-level (fine -> mid -> high) hierarchical loss with the RMI lower bound -- reference
// models/loss/rmi_hiera_triplet_loss.py:323-546 (math: SURVEY A.6, A.7).
//
//   sh_hiera3_loss_fwd : bilinear resize fused; 3-level sigmoid BCE (eps 1e-6, :352-470), three all-pixel-mean CE
//                        terms (:523-526), valid counts; optionally materialises P = sigmoid(z)*valid + 1e-6 (planar f32)
//                        for the RMI term (:496).
//   sh_rmi_gram        : per (image, channel) the three 9x9 Gram matrices la*la^T, pr*pr^T, la*pr^T over the
//                        (H-2)(W-2) 3x3 windows, accumulated in f64 like the reference's .double() bmm (:498-510),
//                        without ever building the 2 x [B,C,9,N] f64 stacks (3.6 GB each at config 4).
//   sh_rmi_solve       : 9x9 f64 algebra per (image, channel): inverse, Schur complement, Cholesky log-det (:313-317,
//                        :509-513) + the closed-form backward matrices K1, K2.
//   sh_rmi_dprob       : dL/dP at every pixel = 3x3 col2im of (K1*la + K2*pr).
//   sh_hiera3_loss_bwd : tiled gather-form backward (as loss.hip) adding the RMI gradient through sigmoid'.
#include <type_traits>
#include "loss_common.h"

#define MAXF3 64
#define MAXM3 16
#define MAXH3 8

struct H3Tab {
    int nf, nm, nh;
    signed char f2m[MAXF3], f2h[MAXF3];
    unsigned long long fine_of_mid[MAXM3];   // bitmask of fine ids with f2m == m
    unsigned int high_of_mid[MAXM3];         // bitmask of high ids reachable from mid m
    unsigned int mid_of_high[MAXH3];         // bitmask of mid ids under high j
};

// Per-pixel 3-level terms.  out[0..5] = bce_f, bce_m, bce_h, ce_f, ce_m, ce_h.  With GRAD:
//   g[j] += d/dz_j of  a0*bce_f + a1*bce_m + a2*bce_h + b*(ce_f+ce_m+ce_h)
template <int MAXC, bool GRAD>
__device__ __forceinline__ void hiera3_pixel(const float (&z)[MAXC], int tf, const H3Tab& T, float a0, float a1, float a2, float b,
                                             float (&out)[6], float (&g)[MAXC]) {
    const float eps = 1e-6f;
    const int nf = T.nf, nm = T.nm, nh = T.nh;
    float p[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) p[j] = j < nf + nm + nh ? sigmoidf_(z[j]) : 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) out[j] = 0.f;
    if (tf == IGN) return;                     // mid / high targets are void exactly when the fine one is
    const int tm = T.f2m[tf], th = T.f2h[tf];
    auto P = [&](int idx) { float v = 0.f;
#pragma unroll
        for (int j = 0; j < MAXC; ++j) if (j == idx) v = p[j];
        return v; };
    auto addg = [&](int idx, float d) {        // d = dL/dp_idx  -> through sigmoid'
#pragma unroll
        for (int j = 0; j < MAXC; ++j) if (j == idx) g[j] += d * p[j] * (1.f - p[j]); };

    // combined maxima: mcmb[m] = max(max_{f in m} p_f, p_mid_m) with first-max tie rule; remember the argmax channel
    float mcmb[MAXM3]; int amb[MAXM3];
    for (int m = 0; m < nm; ++m) {
        float best = -INFINITY; int bi = -1;
#pragma unroll
        for (int j = 0; j < MAXC; ++j)
            if (j < nf && ((T.fine_of_mid[m] >> j) & 1ull) && p[j] > best) { best = p[j]; bi = j; }
        const float pm = P(nf + m);
        if (bi < 0 || pm > best) { best = pm; bi = nf + m; }
        mcmb[m] = best; amb[m] = bi;
    }
    // ---- fine
    {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
            if (k < nf) {
                const float s = p[k];
                if (k == tf) {
                    const float tt = P(nf + T.f2m[k]);
                    const bool s_min = s <= tt;
                    const float mn = s_min ? s : tt;
                    acc += -logf(mn + eps);
                    if (GRAD) addg(s_min ? k : nf + T.f2m[k], -a0 / (mn + eps));
                } else {
                    acc += -logf(1.f - s + eps);
                    if (GRAD) g[k] += a0 / (1.f - s + eps) * s * (1.f - s);
                }
            }
        }
        out[0] = acc;
    }
    // ---- mid
    {
        float acc = 0.f;
        for (int m = 0; m < nm; ++m) {
            if (m == tm) {
                // mclb = min(min_{h in H(m)} p_high_h, p_mid_m), the high minimum wins ties
                float best = INFINITY; int bi = -1;
                for (int hh = 0; hh < nh; ++hh)
                    if ((T.high_of_mid[m] >> hh) & 1u) { const float v = P(nf + nm + hh); if (v < best) { best = v; bi = nf + nm + hh; } }
                const float pm = P(nf + m);
                if (bi < 0 || pm < best) { best = pm; bi = nf + m; }
                acc += -logf(best + eps);
                if (GRAD) addg(bi, -a1 / (best + eps));
            } else {
                acc += -logf(1.f - mcmb[m] + eps);
                if (GRAD) addg(amb[m], a1 / (1.f - mcmb[m] + eps));
            }
        }
        out[1] = acc;
    }
    // ---- high
    {
        float acc = 0.f;
        for (int j = 0; j < nh; ++j) {
            const float pj = P(nf + nm + j);
            if (j == th) {
                acc += -logf(pj + eps);
                if (GRAD) addg(nf + nm + j, -a2 / (pj + eps));
            } else {
                float best = -INFINITY; int bi = -1;
                for (int m = 0; m < nm; ++m)
                    if (((T.mid_of_high[j] >> m) & 1u) && mcmb[m] > best) { best = mcmb[m]; bi = amb[m]; }
                if (bi < 0 || pj > best) { best = pj; bi = nf + nm + j; }
                acc += -logf(1.f - best + eps);
                if (GRAD) addg(bi, a2 / (1.f - best + eps));
            }
        }
        out[2] = acc;
    }
    out[3] = softmax_ce<MAXC, GRAD>(z, 0, nf, tf, b, g);
    out[4] = softmax_ce<MAXC, GRAD>(z, nf, nm, tm, b, g);
    out[5] = softmax_ce<MAXC, GRAD>(z, nf + nm, nh, th, b, g);
}

// ------------------------------------------------------------------------------------------ forward
// valid-label count ahead of a forward that also emits the gradient (its normaliser n_valid): cnt[0] += #(label != 255)
__global__ __launch_bounds__(256) void valid_count_kernel(const uint8_t* __restrict__ labels, long long total, unsigned long long* __restrict__ cnt) {
    __shared__ unsigned int red[4];
    unsigned int a = 0;
    const long long n4 = (((uintptr_t)labels & 3) == 0) ? total / 4 : 0, gs = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += gs) {
        const unsigned v = reinterpret_cast<const unsigned*>(labels)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) a += ((v >> (8 * k)) & 255u) != (unsigned)IGN;
    }
    for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += gs) a += labels[i] != IGN;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(cnt, (unsigned long long)((red[0] + red[1]) + (red[2] + red[3])));
}
// GRAD (r3): every pixel's d(loss_out)/d(interpolated logits) -- everything but the RMI term, at unit upstream gradient -- goes to
// gfull [N*H*W][L] in the same pass (as hiera2_fwd_kernel); cnt[0] = n_valid from valid_count_kernel.  The backward is then
// hiera3_rmi_add_kernel (the RMI term, a streaming pass) and the scaled adjoint of the resize.
template <int MAXC, bool GRAD>
__global__ __launch_bounds__(256) void hiera3_fwd_kernel(const float* __restrict__ logits, long long ldl, const uint8_t* __restrict__ labels,
                                                         const H3Tab T, float* __restrict__ partials, float* __restrict__ probs,
                                                         uint8_t* __restrict__ mid_out, uint8_t* __restrict__ high_out,
                                                         int h, int w, int H, int W, float sy, float sx, long long total,
                                                         const unsigned long long* __restrict__ cnt, float* __restrict__ gfull, int L) {
    const bool identity = (h == H && w == W);
    const int C = T.nf + T.nm + T.nh;
    float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, b = 0.f;
    if (GRAD) {          // the coefficients of hiera3_grad_fullres_kernel at unit upstream gradient
        const double nv = cnt[0] < 1 ? 1.0 : (double)cnt[0];
        a0 = 0.5f * (float)(5.0 / (nv * T.nf)); a1 = 0.5f * (float)(5.0 / (nv * T.nm)); a2 = 0.5f * (float)(5.0 / (nv * T.nh));
        b = (float)(1.0 / (double)total);
    }
    const long long base = (long long)blockIdx.x * LOSS_PIX_PER_BLOCK;
#pragma unroll 1
    for (int it = 0; it < LOSS_PIX_PER_BLOCK / 256; ++it) {
        const long long i = base + it * 256 + threadIdx.x;
        if (i >= total) break;
        const int f = labels[i];
        if (mid_out) {     // the target maps of _prepare_targets_three_level (:21-63): 255 stays 255, else gather through the maps
            mid_out[i] = f == IGN ? (uint8_t)IGN : (uint8_t)T.f2m[f];
            high_out[i] = f == IGN ? (uint8_t)IGN : (uint8_t)T.f2h[f];
        }
        const int ox = (int)(i % W);
        const long long q = i / W;
        const int oy = (int)(q % H);
        const long long n = q / H;
        if (GRAD && f == IGN) {
#pragma unroll
            for (int j = 0; j < MAXC; j += 4)
                if (j < L) st4(gfull + i * L + j, f32x4{0.f, 0.f, 0.f, 0.f});
        }
        if (f == IGN && probs == nullptr) continue;
        const Lerp ly = lerp_src(oy, sy, h), lx = lerp_src(ox, sx, w);
        float z[MAXC], g[MAXC], o[6];
        fetch_logits<MAXC>(logits + n * h * w * ldl, ldl, w, ly, lx, identity, C, z);
        if (probs) {       // P = sigmoid(z) * valid + 1e-6, planar [n][c][H][W]  (rmi_hiera_triplet_loss.py:496)
#pragma unroll
            for (int j = 0; j < MAXC; ++j)
                if (j < C) probs[((n * C + j) * H + oy) * W + ox] = (f == IGN ? 0.f : sigmoidf_(z[j])) + 1e-6f;
        }
        if (f == IGN) continue;
        if (GRAD) {
#pragma unroll
            for (int j = 0; j < MAXC; ++j) g[j] = 0.f;
        }
        hiera3_pixel<MAXC, GRAD>(z, f, T, a0, a1, a2, b, o, g);
        if (GRAD) {
#pragma unroll
            for (int j = 0; j < MAXC; j += 4)
                if (j < L) st4(gfull + i * L + j, f32x4{g[j], g[j + 1], g[j + 2], g[j + 3]});
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] += o[j];
        v[6] += 1.f;
    }
    block_reduce_store<7>(v, partials);
}
// sums[0..5] = the six sums, sums[6] = n_valid, sums[7] = n_pixels;
// loss_out[0] = 0.5 * 5*(bce_f/(nv*nf) + bce_m/(nv*nm) + bce_h/(nv*nh)) + (ce_f+ce_m+ce_h)/npix      (everything but RMI / triplet)
__global__ __launch_bounds__(256) void hiera3_finalize_kernel(const float* __restrict__ partials, int nblk, double npix, int nf, int nm, int nh,
                                                              double* __restrict__ sums, float* __restrict__ loss_out) {
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
        s[7] = npix;
        for (int j = 0; j < 8; ++j) sums[j] = s[j];
        const double nv = s[6] < 1.0 ? 1.0 : s[6];
        const float hiera = 5.0f * ((float)(s[0] / (nv * nf)) + (float)(s[1] / (nv * nm)) + (float)(s[2] / (nv * nh)));
        loss_out[0] = 0.5f * hiera + (float)(s[3] / npix) + (float)(s[4] / npix) + (float)(s[5] / npix);
    }
}

// ------------------------------------------------------------------------------------------ RMI Gram matrices
// la for channel c at a pixel: one-hot of the level's target with void -> class 0 (NOT masked, :479-483).
__device__ __forceinline__ float la_of(int f, int c, const H3Tab& T) {
    const int nf = T.nf, nm = T.nm;
    if (c < nf) return ((f == IGN ? 0 : f) == c) ? 1.f : 0.f;
    if (c < nf + nm) return ((f == IGN ? 0 : T.f2m[f]) == c - nf) ? 1.f : 0.f;
    return ((f == IGN ? 0 : T.f2h[f]) == c - nf - nm) ? 1.f : 0.f;
}
#define GRAM_ROWS 64
#define GRAM_ENTRIES 171      // pp upper 45 | lp 81 | ll upper 45
// block = (window-column strip of 64, window-row chunk of GRAM_ROWS, image*channel); 4 waves = 4 entry groups.
// Lane = window column x (coalesced loads), each thread slides a 3x3 window down its column.
__global__ __launch_bounds__(256) void rmi_gram_kernel(const float* __restrict__ probs, const uint8_t* __restrict__ labels, const H3Tab T,
                                                       double* __restrict__ partials, int H, int W, int C) {
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int nW = W - 2, nH = H - 2;
    const int x = blockIdx.x * 64 + lane;
    const int y0 = blockIdx.y * GRAM_ROWS, y1 = min(y0 + GRAM_ROWS, nH);
    const int bc = blockIdx.z, n = bc / C, c = bc - n * C;
    // label -> one-hot value of this block's channel, as an LDS table (a per-lane index into the by-value table struct
    // would otherwise live in scratch memory)
    __shared__ float lut[256];
    lut[threadIdx.x] = (threadIdx.x < T.nf || threadIdx.x == IGN) ? la_of((int)threadIdx.x, c, T) : 0.f;
    __syncthreads();
    const float* P = probs + (long long)bc * H * W;
    const uint8_t* L = labels + (long long)n * H * W;
    double acc[45];
#pragma unroll
    for (int k = 0; k < 45; ++k) acc[k] = 0.0;
    const bool active = x < nW;
    // (r3) The 3 x 3 window lives in a ring of three row slots: the loop is unrolled by three so that every step addresses its rows at
    // compile time (no 12 f64 moves per row to slide the window), the next row's six loads are issued before this row's arithmetic,
    // and each product is accumulated with one fused multiply-add (45 v_fma_f64 per row and wave instead of 45 v_mul + 45 v_add; the
    // f64 accumulation order per entry is unchanged).  Measured on configs[3] (12 channels, 512^2, B = 16): 1.38 -> see DESIGN.md.
    double pw[3][3], lw[3][3];      // [ring slot][dx]
    if (active) {
        float pn[3]; uint8_t ln[3];
        auto fetch = [&](int row) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) { pn[dx] = P[(long long)row * W + x + dx]; ln[dx] = L[(long long)row * W + x + dx]; }
        };
        auto park = [&](int slot) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) { pw[slot][dx] = (double)pn[dx]; lw[slot][dx] = (double)lut[ln[dx]]; }
        };
        fetch(y0); park(0);
        fetch(y0 + 1); park(1);
        fetch(y0 + 2);                                   // row y0 + 2 of the first window: parked by the first step
        auto step = [&](auto kk, int y) {
            constexpr int K = decltype(kk)::value;       // window rows top -> bottom = slots K, K+1, K+2 (mod 3); the new row goes to K+2
            park((K + 2) % 3);
            if (y + 1 < y1) fetch(y + 3);                // next window's new row: in flight during this window's arithmetic
            double pr[9], la[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) { pr[3 * dy + dx] = pw[(K + dy) % 3][dx]; la[3 * dy + dx] = lw[(K + dy) % 3][dx]; }
            if (grp == 0) {            // pp upper triangle
                int e = 0;
#pragma unroll
                for (int i = 0; i < 9; ++i)
#pragma unroll
                    for (int j = i; j < 9; ++j) { acc[e] = __builtin_fma(pr[i], pr[j], acc[e]); ++e; }
            } else if (grp == 1) {     // lp rows 0..4
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int j = 0; j < 9; ++j) acc[i * 9 + j] = __builtin_fma(la[i], pr[j], acc[i * 9 + j]);
            } else if (grp == 2) {     // lp rows 5..8
#pragma unroll
                for (int i = 5; i < 9; ++i)
#pragma unroll
                    for (int j = 0; j < 9; ++j) acc[(i - 5) * 9 + j] = __builtin_fma(la[i], pr[j], acc[(i - 5) * 9 + j]);
            } else {                   // ll upper triangle
                int e = 0;
#pragma unroll
                for (int i = 0; i < 9; ++i)
#pragma unroll
                    for (int j = i; j < 9; ++j) { acc[e] = __builtin_fma(la[i], la[j], acc[e]); ++e; }
            }
        };
        int y = y0;
        for (; y + 2 < y1; y += 3) {
            step(std::integral_constant<int, 0>{}, y);
            step(std::integral_constant<int, 1>{}, y + 1);
            step(std::integral_constant<int, 2>{}, y + 2);
        }
        if (y < y1) step(std::integral_constant<int, 0>{}, y);
        if (y + 1 < y1) step(std::integral_constant<int, 1>{}, y + 1);
    }
    const int cnt = grp == 2 ? 36 : 45;
    const int off = grp == 0 ? 0 : grp == 1 ? 45 : grp == 2 ? 90 : 126;
    const long long pidx = ((long long)bc * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#pragma unroll
    for (int k = 0; k < 45; ++k) {
        const double s = wave_sum_d(acc[k]);
        if (lane == 0 && k < cnt) partials[pidx * GRAM_ENTRIES + off + k] = s;
    }
}

// ------------------------------------------------------------------------------------------ 9x9 f64 algebra
__device__ void inv9(const double* A, double* Ai) {      // Gauss-Jordan with partial pivoting
    double a[9][18];
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) { a[i][j] = A[i * 9 + j]; a[i][9 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 9; ++c) {
        int piv = c; double best = fabs(a[c][c]);
        for (int r = c + 1; r < 9; ++r) if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
        if (piv != c) for (int j = 0; j < 18; ++j) { const double tmp = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = tmp; }
        const double d = 1.0 / a[c][c];
        for (int j = 0; j < 18; ++j) a[c][j] *= d;
        for (int r = 0; r < 9; ++r) if (r != c) { const double f = a[r][c]; if (f != 0.0) for (int j = 0; j < 18; ++j) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) Ai[i * 9 + j] = a[i][9 + j];
}
__device__ void mm9(const double* A, const double* B, double* Cm, bool tA, bool tB) {
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) {
            double s = 0;
            for (int k = 0; k < 9; ++k) s += (tA ? A[k * 9 + i] : A[i * 9 + k]) * (tB ? B[j * 9 + k] : B[k * 9 + j]);
            Cm[i * 9 + j] = s;
        }
}
// ---- block-parallel 9x9 algebra in LDS (81+ threads; every element goes through the same operation sequence as the serial
//      inv9 / mm9 above, so the results are bit-identical to a one-thread evaluation)
__device__ void inv9_par(const double* A, double* Ai, double (*a)[18], int* piv_s, int t) {      // a: LDS [9][18]
    if (t < 162) { const int i = t / 18, j = t % 18; a[i][j] = j < 9 ? A[i * 9 + j] : (i == j - 9 ? 1.0 : 0.0); }
    __syncthreads();
    for (int c = 0; c < 9; ++c) {
        if (t == 0) {
            int piv = c; double best = fabs(a[c][c]);
            for (int r = c + 1; r < 9; ++r) if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
            *piv_s = piv;
        }
        __syncthreads();
        const int piv = *piv_s;
        if (piv != c && t < 18) { const double tmp = a[c][t]; a[c][t] = a[piv][t]; a[piv][t] = tmp; }
        __syncthreads();
        const double d = 1.0 / a[c][c];
        __syncthreads();
        if (t < 18) a[c][t] *= d;
        __syncthreads();
        const int r = t / 18, j = t % 18;
        double f = 0.0, v = 0.0, pc = 0.0;
        bool upd = false;
        if (t < 162) { f = a[r][c]; v = a[r][j]; pc = a[c][j]; upd = r != c && f != 0.0; }
        __syncthreads();                                 // every old value is read before any element is rewritten
        if (upd) a[r][j] = v - f * pc;
        __syncthreads();
    }
    if (t < 81) Ai[t] = a[t / 9][9 + t % 9];
    __syncthreads();
}
__device__ void mm9_par(const double* A, const double* B, double* Cm, bool tA, bool tB, int t) {
    if (t < 81) {
        const int i = t / 9, j = t % 9;
        double s = 0;
        for (int k = 0; k < 9; ++k) s += (tA ? A[k * 9 + i] : A[i * 9 + k]) * (tB ? B[j * 9 + k] : B[k * 9 + j]);
        Cm[t] = s;
    }
    __syncthreads();
}
// one block (192 threads) per (image, channel): out rmi[bc], K1[bc][81], K2[bc][81]
__global__ __launch_bounds__(192) void rmi_solve_kernel(const double* __restrict__ partials, int nparts, int BC, double* __restrict__ rmi,
                                                        double* __restrict__ K1, double* __restrict__ K2) {
    __shared__ double e[GRAM_ENTRIES];
    __shared__ double Slp[81], A[81], M[81], T1[81], T2[81], V[81], G0[81], Lc[81];
    __shared__ double aug[9][18];
    __shared__ int piv_s;
    const int bc = blockIdx.x, t = threadIdx.x;
    if (t < GRAM_ENTRIES) {            // Gram partials summed entry-parallel, in the serial order over the parts
        double acc = 0.0;
        for (int pp = 0; pp < nparts; ++pp) acc += partials[((long long)bc * nparts + pp) * GRAM_ENTRIES + t];
        e[t] = acc;
    }
    __syncthreads();
    const double alpha = (double)1e-3f;              // f32 1e-3 promoted to f64, as `diag_eye * _POS_ALPHA` does
    if (t < 81) {
        const int i = t / 9, j = t % 9, lo = i < j ? i : j, hi = i < j ? j : i;
        const int k = lo * 9 - lo * (lo - 1) / 2 + (hi - lo);      // index of (lo, hi) in the packed upper triangle
        A[t] = e[k] + (i == j ? alpha : 0.0);                       // Spp + alpha I
        V[t] = e[126 + k];                                           // Sll (T2 is subtracted below)
        Slp[t] = e[45 + t];
    }
    __syncthreads();
    inv9_par(A, M, aug, &piv_s, t);
    mm9_par(Slp, M, T1, false, false, t);                // Slp * M
    mm9_par(T1, Slp, T2, false, true, t);                // Slp * M * Slp^T
    if (t < 81) V[t] = V[t] - T2[t] + ((t / 9 == t % 9) ? alpha : 0.0);
    __syncthreads();
    if (t == 0) {                                         // Cholesky V = L L^T ; rmi = 0.5 * 2 * sum log(L_ii + 1e-8)
        for (int k = 0; k < 81; ++k) Lc[k] = 0.0;
        double logdet = 0.0;
        for (int j = 0; j < 9; ++j) {
            double s2 = V[j * 9 + j];
            for (int k = 0; k < j; ++k) s2 -= Lc[j * 9 + k] * Lc[j * 9 + k];
            const double d = sqrt(s2);
            Lc[j * 9 + j] = d;
            logdet += log(d + 1e-8);
            for (int i = j + 1; i < 9; ++i) {
                double tt = V[i * 9 + j];
                for (int k = 0; k < j; ++k) tt -= Lc[i * 9 + k] * Lc[j * 9 + k];
                Lc[i * 9 + j] = tt / d;
            }
        }
        rmi[bc] = 0.5 * 2.0 * logdet;
    }
    __syncthreads();
    // backward matrices: G0 = (V_a + alpha I)^-1 ; d(0.5 logdet)/dV_a = 0.5 G0
    //   K1 = -2 M Slp^T (0.5 G0) = -M Slp^T G0 ;  K2 = 2 M Slp^T (0.5 G0) Slp M = M Slp^T G0 Slp M
    inv9_par(V, G0, aug, &piv_s, t);
    mm9_par(M, Slp, T1, false, true, t);                 // M * Slp^T
    mm9_par(T1, G0, T2, false, false, t);                // M Slp^T G0
    if (t < 81) K1[(long long)bc * 81 + t] = -T2[t];
    __syncthreads();
    mm9_par(T2, Slp, T1, false, false, t);               // M Slp^T G0 Slp
    mm9_par(T1, M, T2, false, false, t);
    if (t < 81) K2[(long long)bc * 81 + t] = T2[t];
}
__global__ void rmi_value_kernel(const double* __restrict__ rmi, int B, int C, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float total = 0.f;
        for (int c = 0; c < C; ++c) {
            double s = 0;
            for (int b = 0; b < B; ++b) s += rmi[b * C + c];
            total += (float)(s / B) / 9.0f;
        }
        out[0] = total;
    }
}

// dP[bc][y][x] = sum over the <=9 windows containing (y,x) as element k of (K1[k,:] . la_win + K2[k,:] . pr_win)
#define RMI_DP_ROWS 16
__global__ __launch_bounds__(256) void rmi_dprob_kernel(const float* __restrict__ probs, const uint8_t* __restrict__ labels, const H3Tab T,
                                                        const double* __restrict__ K1, const double* __restrict__ K2, float* __restrict__ dprob,
                                                        int H, int W, int C) {
    __shared__ double k1[81], k2[81];
    __shared__ float lut[256];
    const int bc = blockIdx.z, n = bc / C, c = bc - n * C;
    if (threadIdx.x < 81) { k1[threadIdx.x] = K1[(long long)bc * 81 + threadIdx.x]; k2[threadIdx.x] = K2[(long long)bc * 81 + threadIdx.x]; }
    lut[threadIdx.x] = (threadIdx.x < T.nf || threadIdx.x == IGN) ? la_of((int)threadIdx.x, c, T) : 0.f;
    __syncthreads();
    __shared__ double a1[25], a2[25];
    if (threadIdx.x < 25) {
        const int dy = (int)threadIdx.x / 5 - 2, dx = (int)threadIdx.x % 5 - 2;      // offset of the element from the pixel
        double s1 = 0.0, s2 = 0.0;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int jy = dy + ky, jx = dx + kx;                                // element index inside window k
                if (jy < 0 || jy > 2 || jx < 0 || jx > 2) continue;
                s1 += k1[(ky * 3 + kx) * 9 + jy * 3 + jx]; s2 += k2[(ky * 3 + kx) * 9 + jy * 3 + jx];
            }
        a1[threadIdx.x] = s1; a2[threadIdx.x] = s2;
    }
    __syncthreads();
    // block = 256 columns x RMI_DP_ROWS rows, thread = 4 consecutive pixels of one row in each of the RMI_DP_ROWS / 4 row groups.  The
    // (256+4) x (rows+4) neighbourhood is staged once in LDS; a thread reads its 5 x 8 window with 16-byte LDS reads and each correlation
    // coefficient once for its 4 pixels.  (r3: 16 rows per block instead of 4 -- the halo rows were half of all staged bytes and the
    // per-block set-up (81 + 81 coefficients, the 25 + 25 collapsed ones) ran 49 152 times -- and fused multiply-adds.)
    constexpr int TW = 264;                                  // 256 + 4 halo columns, padded to a multiple of 4 floats
    __shared__ __attribute__((aligned(16))) float tp[RMI_DP_ROWS + 4][TW], tl[RMI_DP_ROWS + 4][TW];
    const int bx0 = blockIdx.x * 256 - 2, by0 = blockIdx.y * RMI_DP_ROWS - 2;
    const float* P = probs + (long long)bc * H * W;
    const uint8_t* L = labels + (long long)n * H * W;
    // staging in chunks of 7 elements per thread: all 14 global loads of a chunk are issued before the first dependent look-up / LDS store
    // (one element at a time the loop was a chain of 21 exposed load latencies per block)
    constexpr int NEL = (RMI_DP_ROWS + 4) * TW, NST = (NEL + 255) / 256, CHK = 7;
    for (int s0 = 0; s0 < NST; s0 += CHK) {
        float pq[CHK]; int lq[CHK]; bool okq[CHK];
#pragma unroll
        for (int u = 0; u < CHK; ++u) {
            const int i = (s0 + u) * 256 + (int)threadIdx.x;
            const int ry = i / TW, rx = i - ry * TW, yy = by0 + ry, xx = bx0 + rx;
            okq[u] = i < NEL && rx < 260 && yy >= 0 && yy < H && xx >= 0 && xx < W;
            pq[u] = okq[u] ? P[(long long)yy * W + xx] : 0.f;
            lq[u] = okq[u] ? (int)L[(long long)yy * W + xx] : 0;
        }
#pragma unroll
        for (int u = 0; u < CHK; ++u) {
            const int i = (s0 + u) * 256 + (int)threadIdx.x;
            if (i < NEL) { const int ry = i / TW, rx = i - ry * TW; tp[ry][rx] = pq[u]; tl[ry][rx] = okq[u] ? lut[lq[u]] : 0.f; }
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & 63;
    const int x0 = blockIdx.x * 256 + 4 * lx;
    if (x0 >= W) return;
    const int nrows = min(RMI_DP_ROWS, H - (int)blockIdx.y * RMI_DP_ROWS);
#pragma unroll 1
    for (int ly = threadIdx.x >> 6; ly < nrows; ly += 4) {
    // the coefficient tables (a1 / a2, k1 / k2: 212 doubles in LDS) are loop-invariant; hoisted out of this loop they took all 512
    // registers and spilled 212 -- the memory clobber keeps their loads inside the iteration
    asm volatile("" ::: "memory");
    const int y = blockIdx.y * RMI_DP_ROWS + ly;
    float pv[5][8], lv[5][8];
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
        const f32x4 p0 = ld4(&tp[ly + dy][4 * lx]), p1 = ld4(&tp[ly + dy][4 * lx + 4]);
        const f32x4 l0 = ld4(&tl[ly + dy][4 * lx]), l1 = ld4(&tl[ly + dy][4 * lx + 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { pv[dy][e] = p0[e]; pv[dy][4 + e] = p1[e]; lv[dy][e] = l0[e]; lv[dy][4 + e] = l1[e]; }
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const bool rows_in = y >= 2 && y < H - 2;
    if (rows_in && x0 >= 2 && x0 + 3 < W - 2) {
        // interior: all nine windows exist, so the double sum over (window k, element j) collapses to two 5x5 correlations
        // with a1[d] = sum_{j-k=d} K1[k][j] (same for K2), built once per block
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                const double c1 = a1[dy * 5 + dx], c2 = a2[dy * 5 + dx];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_fma(c2, (double)pv[dy][q + dx], __builtin_fma(c1, (double)lv[dy][q + dx], acc[q]));
            }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int x = x0 + q;
            if (x >= W) continue;
            if (rows_in && x >= 2 && x < W - 2) {
#pragma unroll
                for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 5; ++dx) acc[q] = __builtin_fma(a2[dy * 5 + dx], (double)pv[dy][q + dx], __builtin_fma(a1[dy * 5 + dx], (double)lv[dy][q + dx], acc[q]));
                continue;
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int wy = y - ky, wx = x - kx;        // window origin for which (y,x) is element (ky,kx)
                    if (wy < 0 || wx < 0 || wy >= H - 2 || wx >= W - 2) continue;
                    const int k = ky * 3 + kx;
#pragma unroll
                    for (int jy = 0; jy < 3; ++jy)
#pragma unroll
                        for (int jx = 0; jx < 3; ++jx) {
                            const int j = jy * 3 + jx;
                            acc[q] += k1[k * 9 + j] * (double)lv[2 - ky + jy][q + 2 - kx + jx] + k2[k * 9 + j] * (double)pv[2 - ky + jy][q + 2 - kx + jx];
                        }
                }
        }
    }
    float* dst = dprob + ((long long)bc * H + y) * W + x0;
    if (x0 + 3 < W && (((uintptr_t)dst) & 15) == 0) {
        st4(dst, f32x4{(float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]});
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (x0 + q < W) dst[q] = (float)acc[q];
    }
    }
}

// ------------------------------------------------------------------------------------------ two-pass backward, pass A
// every full-resolution pixel's gradient w.r.t. its (interpolated) logits, once, to gfull [N*H*W][L] (as loss_grad_fullres_kernel of
// the 2-level loss; pass B = the shared resize-adjoint gather) -- the tile kernel below recomputes a halo per 80 KB LDS tile
template <int MAXC>
__global__ __launch_bounds__(256) void hiera3_grad_fullres_kernel(const float* __restrict__ logits, long long ldl, const uint8_t* __restrict__ labels,
                                                                  const H3Tab T, const double* __restrict__ sums, const float* __restrict__ dprob,
                                                                  float rmi_coef, const float* __restrict__ gscale_dev, float gscale,
                                                                  float* __restrict__ gfull, int L, int h, int w, int H, int W, float sy, float sx,
                                                                  long long total) {
    const int C = T.nf + T.nm + T.nh;
    const bool identity = (h == H && w == W);
    const float gs = gscale * (gscale_dev ? gscale_dev[0] : 1.f);
    const double nv = sums[6] < 1.0 ? 1.0 : sums[6];
    const float a0 = gs * 0.5f * (float)(5.0 / (nv * T.nf)), a1 = gs * 0.5f * (float)(5.0 / (nv * T.nm)), a2 = gs * 0.5f * (float)(5.0 / (nv * T.nh));
    const float b = gs * (float)(1.0 / sums[7]);
    const float rc = gs * rmi_coef;
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
            float z[MAXC], o[6];
            fetch_logits<MAXC>(logits + n * h * w * ldl, ldl, w, ly, lx, identity, C, z);
            hiera3_pixel<MAXC, true>(z, f, T, a0, a1, a2, b, o, g);
            if (dprob != nullptr) {          // RMI: dL/dz = dL/dP * valid * sigmoid'(z)   (the tile kernel's own expression)
#pragma unroll
                for (int j = 0; j < MAXC; ++j)
                    if (j < C) { const float pj = sigmoidf_(z[j]); g[j] += rc * dprob[((n * C + j) * H + oy) * W + ox] * pj * (1.f - pj); }
            }
        }
        float* dst = gfull + i * L;
#pragma unroll
        for (int j = 0; j < MAXC; j += 4)
            if (j < L) st4(dst + j, f32x4{g[j], g[j + 1], g[j + 2], g[j + 3]});
    }
}

// the RMI term of the gradient added to the forward's per-pixel gradient (unit upstream gradient): g[j] += coef * dL/dP_j * sigmoid'(z_j) at
// valid pixels, sigmoid(z_j) recovered from the stored P = sigmoid(z) + 1e-6.  Thread = one pixel: plane reads coalesced along x, its own
// 4 * L bytes of gfull read and written once.
template <int MAXC>
__global__ __launch_bounds__(256) void hiera3_rmi_add_kernel(float* __restrict__ gfull, int L, const float* __restrict__ probs, const float* __restrict__ dprob,
                                                             const uint8_t* __restrict__ labels, float coef, int C, long long HW, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        if (labels[i] == IGN) continue;
        const long long n = i / HW, pix = i - n * HW;
        float pj[MAXC], dj[MAXC];
#pragma unroll
        for (int j = 0; j < MAXC; ++j)
            if (j < C) { pj[j] = probs[(n * C + j) * HW + pix]; dj[j] = dprob[(n * C + j) * HW + pix]; }
        float* dst = gfull + i * L;
#pragma unroll
        for (int j = 0; j < MAXC; j += 4) {
            if (j >= L) break;
            f32x4 g = ld4(dst + j);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (j + e < C) { const float p = pj[j + e] - 1e-6f; g[e] += coef * dj[j + e] * p * (1.f - p); }
            st4(dst + j, g);
        }
    }
}

// ------------------------------------------------------------------------------------------ tiled backward
template <int MAXC>
__global__ __launch_bounds__(256) void hiera3_bwd_tile_kernel(const float* __restrict__ logits, long long ldl, const uint8_t* __restrict__ labels,
                                                              const H3Tab T, const double* __restrict__ sums, const float* __restrict__ dprob,
                                                              float rmi_coef, const float* __restrict__ gscale_dev, float gscale,
                                                              float* __restrict__ dlogits, long long lddl, int h, int w, int H, int W, float sy,
                                                              float sx, int TL, int tiles_x, int tiles_y, int lds_cap) {
    extern __shared__ __attribute__((aligned(16))) float gt[];
    const int t = threadIdx.x;
    const int C = T.nf + T.nm + T.nh;
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
    if (RH * RW * C > lds_cap) return;
    const float gs = gscale * (gscale_dev ? gscale_dev[0] : 1.f);
    const double nv = sums[6] < 1.0 ? 1.0 : sums[6];
    // total = lw * (lambda*rmi + 0.5*hiera + ce...): the caller folds lw into gscale, lambda/(9B) into rmi_coef
    const float a0 = gs * 0.5f * (float)(5.0 / (nv * T.nf)), a1 = gs * 0.5f * (float)(5.0 / (nv * T.nm)), a2 = gs * 0.5f * (float)(5.0 / (nv * T.nh));
    const float b = gs * (float)(1.0 / sums[7]);
    const float rc = gs * rmi_coef;
    const float* base = logits + n * h * w * ldl;
    for (int idx = t; idx < RH * RW; idx += 256) {
        const int ry = idx / RW, rx = idx - ry * RW, oy = ylo + ry, ox = xlo + rx;
        float g[MAXC];
#pragma unroll
        for (int j = 0; j < MAXC; ++j) g[j] = 0.f;
        const int f = labels[(n * H + oy) * W + ox];
        if (f != IGN) {
            const Lerp ly = lerp_src(oy, sy, h), lx = lerp_src(ox, sx, w);
            float z[MAXC], o[6];
            fetch_logits<MAXC>(base, ldl, w, ly, lx, identity, C, z);
            hiera3_pixel<MAXC, true>(z, f, T, a0, a1, a2, b, o, g);
            if (dprob != nullptr) {          // RMI: dL/dz = dL/dP * valid * sigmoid'(z)
#pragma unroll
                for (int j = 0; j < MAXC; ++j)
                    if (j < C) { const float pj = sigmoidf_(z[j]); g[j] += rc * dprob[((n * C + j) * H + oy) * W + ox] * pj * (1.f - pj); }
            }
        }
#pragma unroll
        for (int j = 0; j < MAXC; ++j) if (j < C) gt[idx * C + j] = g[j];
    }
    __syncthreads();
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
                const Lerp ly = lerp_src(oy, sy, h);
                const float wy = identity ? 1.f : (ly.i0 == iy ? ly.w0 : 0.f) + (ly.i1 == iy ? ly.w1 : 0.f);
                if (wy == 0.f) continue;
                const float* row = gt + ((oy - ylo) * RW - xlo) * C + ch;
                for (int ox = cxlo; ox <= cxhi; ++ox) {
                    const Lerp lx = lerp_src(ox, sx, w);
                    const float wx = identity ? 1.f : (lx.i0 == ix ? lx.w0 : 0.f) + (lx.i1 == ix ? lx.w1 : 0.f);
                    if (wx != 0.f) acc += (wy * wx) * row[ox * C];
                }
            }
        }
        dlogits[((n * h + iy) * w + ix) * lddl + ch] = acc;
    }
}

// ------------------------------------------------------------------------------------------ host side
static bool make_tab3(H3Tab& T, const int* f2m, const int* f2h, int nf, int nm, int nh) {
    if (nf <= 0 || nf > MAXF3 || nm <= 0 || nm > MAXM3 || nh <= 0 || nh > MAXH3 || nf + nm + nh > 32 || !f2m || !f2h) return false;
    T.nf = nf; T.nm = nm; T.nh = nh;
    for (int i = 0; i < MAXM3; ++i) { T.fine_of_mid[i] = 0; T.high_of_mid[i] = 0; }
    for (int i = 0; i < MAXH3; ++i) T.mid_of_high[i] = 0;
    for (int f = 0; f < MAXF3; ++f) { T.f2m[f] = 0; T.f2h[f] = 0; }
    for (int f = 0; f < nf; ++f) {
        if (f2m[f] < 0 || f2m[f] >= nm || f2h[f] < 0 || f2h[f] >= nh) return false;
        T.f2m[f] = (signed char)f2m[f]; T.f2h[f] = (signed char)f2h[f];
        T.fine_of_mid[f2m[f]] |= 1ull << f;
        T.high_of_mid[f2m[f]] |= 1u << f2h[f];
        T.mid_of_high[f2h[f]] |= 1u << f2m[f];
    }
    return true;
}

// grad_out (optional, with ldg in {16, 32} >= C and sh_loss_bwd_workspace(N, H, W, ldg) bytes): the per-pixel gradient of everything but the
// RMI term at unit upstream gradient, for sh_hiera3_loss_bwd(..., workspace_has_grad = 1)
extern "C" int sh_hiera3_loss_fwd(const float* logits, int ldl, const uint8_t* labels, const int* f2m_host, const int* f2h_host, int n_fine,
                                  int n_mid, int n_high, double* sums, float* loss_out, float* partials, float* probs, uint8_t* mid_out,
                                  uint8_t* high_out, int N, int h, int w, int H, int W, float* grad_out, int64_t grad_out_bytes, int ldg,
                                  void* stream) {
    H3Tab T;
    if (!logits || !labels || !sums || !loss_out || !partials || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return SH_EINVAL;
    if ((mid_out == nullptr) != (high_out == nullptr)) return SH_EINVAL;
    if (!make_tab3(T, f2m_host, f2h_host, n_fine, n_mid, n_high) || ldl < n_fine + n_mid + n_high) return SH_EINVAL;
    const long long total = (long long)N * H * W;
    const int nblk = (int)sh_cdiv(total, LOSS_PIX_PER_BLOCK);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipStream_t st = (hipStream_t)stream;
    const int C = n_fine + n_mid + n_high;
    if (grad_out != nullptr) {
        if (ldg < C || (ldg != 16 && ldg != 32) || ((uintptr_t)grad_out & 15) || grad_out_bytes < (int64_t)total * ldg * 4 || h > H || w > W) return SH_EINVAL;
        unsigned long long* cnt = reinterpret_cast<unsigned long long*>(sums + 7);     // scratch until hiera3_finalize_kernel writes sums[7]
        if (hipMemsetAsync(cnt, 0, sizeof(unsigned long long), st) != hipSuccess) return SH_ELAUNCH;
        valid_count_kernel<<<256, 256, 0, st>>>(labels, total, cnt);
        if (ldg == 16) hiera3_fwd_kernel<16, true><<<nblk, 256, 0, st>>>(logits, ldl, labels, T, partials, probs, mid_out, high_out, h, w, H, W, sy, sx, total, cnt, grad_out, ldg);
        else hiera3_fwd_kernel<32, true><<<nblk, 256, 0, st>>>(logits, ldl, labels, T, partials, probs, mid_out, high_out, h, w, H, W, sy, sx, total, cnt, grad_out, ldg);
    } else if (C <= 16) hiera3_fwd_kernel<16, false><<<nblk, 256, 0, st>>>(logits, ldl, labels, T, partials, probs, mid_out, high_out, h, w, H, W, sy, sx, total, nullptr, nullptr, 0);
    else hiera3_fwd_kernel<32, false><<<nblk, 256, 0, st>>>(logits, ldl, labels, T, partials, probs, mid_out, high_out, h, w, H, W, sy, sx, total, nullptr, nullptr, 0);
    int rc = sh_launch_status();
    if (rc != SH_OK) return rc;
    hiera3_finalize_kernel<<<1, 256, 0, st>>>(partials, nblk, (double)total, n_fine, n_mid, n_high, sums, loss_out);
    return sh_launch_status();
}

extern "C" int64_t sh_rmi_workspace(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H < 3 || W < 3) return SH_EINVAL;
    const long long parts = sh_cdiv(W - 2, 64) * sh_cdiv(H - 2, GRAM_ROWS);
    return (int64_t)((long long)N * C * parts * GRAM_ENTRIES * 8 + (long long)N * C * (1 + 81 + 81) * 8 + 64);
}
// byte offset, inside the sh_rmi_loss workspace, of the f64 [N][C] values rmi_now = 0.5 * logdet (rmi_hiera_triplet_loss.py:513)
// that sh_rmi_loss leaves behind -- read back by the parity tests against the reference's own per-channel values
extern "C" int64_t sh_rmi_values_offset(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H < 3 || W < 3) return SH_EINVAL;
    const long long parts = sh_cdiv(W - 2, 64) * sh_cdiv(H - 2, GRAM_ROWS);
    return (int64_t)((long long)N * C * parts * GRAM_ENTRIES * 8);
}
// probs: planar [N][C][H][W] from sh_hiera3_loss_fwd.  workspace: sh_rmi_workspace bytes.  rmi_out: device float[1] = RMI term
// (sum_c mean_b 0.5*logdet / 9).  dprob (optional): planar [N][C][H][W] <- d(rmi_out * 9 * N)/dP, i.e. WITHOUT the 1/(9N) factor.
extern "C" int sh_rmi_loss(const float* probs, const uint8_t* labels, const int* f2m_host, const int* f2h_host, int n_fine, int n_mid,
                           int n_high, void* workspace, float* rmi_out, float* dprob, int N, int H, int W, void* stream) {
    H3Tab T;
    if (!probs || !labels || !workspace || !rmi_out || N <= 0 || H < 3 || W < 3) return SH_EINVAL;
    if (!make_tab3(T, f2m_host, f2h_host, n_fine, n_mid, n_high)) return SH_EINVAL;
    const int C = n_fine + n_mid + n_high, BC = N * C;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)sh_cdiv(W - 2, 64), (unsigned)sh_cdiv(H - 2, GRAM_ROWS), (unsigned)BC);
    const int parts = (int)(grid.x * grid.y);
    double* partials = (double*)workspace;
    double* rmi = partials + (long long)BC * parts * GRAM_ENTRIES;
    double* K1 = rmi + BC;
    double* K2 = K1 + (long long)BC * 81;
    rmi_gram_kernel<<<grid, 256, 0, st>>>(probs, labels, T, partials, H, W, C);
    rmi_solve_kernel<<<(unsigned)BC, 192, 0, st>>>(partials, parts, BC, rmi, K1, K2);
    rmi_value_kernel<<<1, 64, 0, st>>>(rmi, N, C, rmi_out);
    if (dprob) {
        dim3 g2((unsigned)sh_cdiv(W, 256), (unsigned)sh_cdiv(H, RMI_DP_ROWS), (unsigned)BC);
        rmi_dprob_kernel<<<g2, 256, 0, st>>>(probs, labels, T, K1, K2, dprob, H, W, C);
    }
    return sh_launch_status();
}

static int pick_tile3(int h, int w, int H, int W, int C, int budget_bytes, int& region_elems) {
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
    const float s = sy > sx ? sy : sx;
    int best = 0;
    for (int TL = 1; TL <= 32; ++TL) {
        const int side = (h == H && w == W) ? TL : (int)ceilf(TL * s + s) + 3;
        if ((long long)side * side * C * 4 > budget_bytes) break;
        best = TL; region_elems = side * side * C;
    }
    return best;
}
// d/dlogits of  gscale*gscale_dev[0] * ( 0.5*hiera3 + ce_f + ce_m + ce_h + rmi_coef * sum_pixels dprob*dP/dz ), into [N,h,w,lddl]
// workspace_has_grad: the workspace is the grad_out of sh_hiera3_loss_fwd (row stride lddl); `probs` (that forward's planar probabilities)
// is then needed with dprob: the RMI term is added to the workspace in one streaming pass, the backward itself is the scaled adjoint of the resize
extern "C" int sh_hiera3_loss_bwd(const float* logits, int ldl, const uint8_t* labels, const int* f2m_host, const int* f2h_host, int n_fine,
                                  int n_mid, int n_high, const double* sums, const float* dprob, float rmi_coef, const float* gscale_dev,
                                  float gscale, float* dlogits, int lddl, int N, int h, int w, int H, int W, float* workspace,
                                  int64_t workspace_bytes, int workspace_has_grad, const float* probs, void* stream) {
    H3Tab T;
    if (!logits || !labels || !sums || !dlogits || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || h > H || w > W) return SH_EINVAL;
    const int C = n_fine + n_mid + n_high;
    if (!make_tab3(T, f2m_host, f2h_host, n_fine, n_mid, n_high) || ldl < C || lddl < C || lddl > 32) return SH_EINVAL;
    if (workspace_has_grad) {
        const long long full = (long long)N * H * W;
        if (!workspace || (lddl != 16 && lddl != 32) || workspace_bytes < full * lddl * 4 || (((uintptr_t)workspace | (uintptr_t)dlogits) & 15) ||
            (dprob && !probs)) return SH_EINVAL;
        hipStream_t st0 = (hipStream_t)stream;
        if (dprob) {
            long long ga = sh_cdiv(full, 256);
            if (ga > 16384) ga = 16384;
            if (lddl == 16) hiera3_rmi_add_kernel<16><<<(unsigned)ga, 256, 0, st0>>>(workspace, lddl, probs, dprob, labels, rmi_coef, C, (long long)H * W, full);
            else hiera3_rmi_add_kernel<32><<<(unsigned)ga, 256, 0, st0>>>(workspace, lddl, probs, dprob, labels, rmi_coef, C, (long long)H * W, full);
            const int rc0 = sh_launch_status();
            if (rc0 != SH_OK) return rc0;
        }
        return sh_launch_gather_from_grad(workspace, gscale_dev, gscale, dlogits, lddl, N, h, w, H, W, st0);
    }
    if (workspace && workspace_bytes >= (int64_t)N * H * W * lddl * 4 && (h < H || w < W) && (lddl & 3) == 0 &&
        (((uintptr_t)workspace | (uintptr_t)dlogits) & 15) == 0) {
        // two streaming passes (see sh_hiera2_loss_bwd): same arithmetic and summation order as the tile kernel => bit-identical
        const long long full = (long long)N * H * W;
        long long ga = sh_cdiv(full, 256);
        if (ga > 16384) ga = 16384;
        const float sy2 = (float)h / (float)H, sx2 = (float)w / (float)W;
        hipStream_t st2 = (hipStream_t)stream;
        if (C <= 16 && lddl <= 16)
            hiera3_grad_fullres_kernel<16><<<(unsigned)ga, 256, 0, st2>>>(logits, ldl, labels, T, sums, dprob, rmi_coef, gscale_dev, gscale, workspace, lddl, h, w, H, W, sy2, sx2, full);
        else
            hiera3_grad_fullres_kernel<32><<<(unsigned)ga, 256, 0, st2>>>(logits, ldl, labels, T, sums, dprob, rmi_coef, gscale_dev, gscale, workspace, lddl, h, w, H, W, sy2, sx2, full);
        const int rc2 = sh_launch_status();
        if (rc2 != SH_OK) return rc2;
        return sh_launch_resize_adjoint_gather(workspace, dlogits, lddl, N, h, w, H, W, st2);
    }
    int region = 0;
    const int TL = pick_tile3(h, w, H, W, C, 80 * 1024, region);
    if (TL <= 0) return SH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int tiles_x = (int)sh_cdiv(w, TL), tiles_y = (int)sh_cdiv(h, TL);
    const unsigned nblk = (unsigned)((long long)N * tiles_x * tiles_y);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    static bool attr16 = false, attr32 = false;
    if (C <= 16) {
        if (!attr16) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hiera3_bwd_tile_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr16 = true; }
        hiera3_bwd_tile_kernel<16><<<nblk, 256, (size_t)region * 4, st>>>(logits, ldl, labels, T, sums, dprob, rmi_coef, gscale_dev, gscale, dlogits, lddl, h, w, H, W, sy, sx, TL, tiles_x, tiles_y, region);
    } else {
        if (!attr32) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&hiera3_bwd_tile_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr32 = true; }
        hiera3_bwd_tile_kernel<32><<<nblk, 256, (size_t)region * 4, st>>>(logits, ldl, labels, T, sums, dprob, rmi_coef, gscale_dev, gscale, dlogits, lddl, h, w, H, W, sy, sx, TL, tiles_x, tiles_y, region);
    }
    return sh_launch_status();
}

__global__ void scalar_axpy_kernel(const float* a, const float* b, float alpha, float* out) {
    if (threadIdx.x == 0) out[0] = a[0] + alpha * b[0];
}
extern "C" int sh_scalar_axpy(const float* a, const float* b, float alpha, float* out, void* stream) {
    if (!a || !b || !out) return SH_EINVAL;
    scalar_axpy_kernel<<<1, 64, 0, (hipStream_t)stream>>>(a, b, alpha, out);
    return sh_launch_status();
}
