// BatchNorm statistics / apply / backward, layout converters and small elementwise helpers (gfx950).
// All HBM-bound: 16-byte loads where the channel count allows, grid-stride loops capped at 2048 blocks,
// wave-shuffle -> LDS -> per-block partials, and f64 for every cross-block combine (deterministic two-stage
// reductions, no float atomics).
#include "common.h"
#include <stdlib.h>

static inline unsigned grid_for(long long work_items, int block = 256, int cap = 4096) {
    long long g = sh_cdiv(work_items, block);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---------------------------------------------------------------------------------------------- layout
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int C, long long HW, int Cpad, long long total) {
    // one thread per output element (n, hw, c); reads are strided by HW but C is tiny (3) for the only user
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % Cpad);
        const long long q = i / Cpad, hw = q % HW, n = q / HW;
        y[i] = c < C ? x[(n * C + c) * HW + hw] : 0.f;
    }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int C, long long HW, int Cpad, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long hw = i % HW, q = i / HW;
        const int c = (int)(q % C);
        const long long n = q / C;
        y[i] = x[(n * HW + hw) * Cpad + c];
    }
}
// the image path (C <= 4 -> Cpad = 4): one thread per pixel, C coalesced plane reads, one 16-byte store
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y, int C, long long HW,
                                                            long long npix) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const long long n = i / HW, hw = i - n * HW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const float* src = x + n * C * HW + hw;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < C) v[c] = src[(long long)c * HW];
        st4(y + i * 4, v);
    }
}
extern "C" int sh_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int Cpad, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C) return SH_EINVAL;
    if (Cpad == 4 && ((uintptr_t)y & 15) == 0) {
        const long long npix = (long long)N * H * W;
        nchw_to_nhwc4_kernel<<<grid_for(npix), 256, 0, (hipStream_t)stream>>>(x, y, C, (long long)H * W, npix);
        return sh_launch_status();
    }
    const long long total = (long long)N * H * W * Cpad;
    nchw_to_nhwc_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(x, y, C, (long long)H * W, Cpad, total);
    return sh_launch_status();
}
extern "C" int sh_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int Cpad, void* stream) {
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C) return SH_EINVAL;
    const long long total = (long long)N * H * W * C;
    nhwc_to_nchw_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(x, y, C, (long long)H * W, Cpad, total);
    return sh_launch_status();
}

// ---------------------------------------------------------------------------------------------- misc
__global__ __launch_bounds__(256) void fill_kernel(float* p, float v, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = v;
}
__global__ __launch_bounds__(256) void axpy_kernel(float* y, const float* x, float a, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] += a * x[i];
}
__global__ __launch_bounds__(256) void axpy4_kernel(float* y, const float* x, float a, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        st4(y + i * 4, ld4(y + i * 4) + a * ld4(x + i * 4));
}
extern "C" int sh_fill(float* p, float v, int64_t n, void* stream) {
    if (!p || n < 0) return SH_EINVAL;
    if (n == 0) return SH_OK;
    fill_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(p, v, n);
    return sh_launch_status();
}
extern "C" int sh_axpy(float* y, const float* x, float a, int64_t n, void* stream) {
    if (!y || !x || n < 0) return SH_EINVAL;
    if (n == 0) return SH_OK;
    if ((n & 3) == 0 && (((uintptr_t)y | (uintptr_t)x) & 15) == 0) axpy4_kernel<<<grid_for(n / 4), 256, 0, (hipStream_t)stream>>>(y, x, a, n / 4);
    else axpy_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(y, x, a, n);
    return sh_launch_status();
}

// ---------------------------------------------------------------------------------------------- stat partial reduce
// partials: [P][2][C] floats.  Block = 4 channels x 256 row-groups; f64 accumulation; LDS tree.
// Writes sums[0][c], sums[1][c] (double) into LDS-resident result then calls the functor on thread < 4.
// CENTRED != 0: partial = (sum, M2 about its own mean) over n_p = min(R, M - p*R) rows -> returns (sum, sum of squares about 0)
// NT threads per block (256, or 1024 for long partial lists: the kernel is a latency-bound strided gather -- 16 bytes per row --
// and the layers with 4096+ partials and 64 channels run only 16 blocks, so more threads per block is the parallelism there is)
// Tried (round 2): blocks of 16 / 32 adjacent channels (64 / 128 contiguous bytes per row, NT/4 or NT/8 rows per pass) -- slower, 17 -> 20 /
// 25 us at P = 4096, C = 64: the time follows the passes per thread, not the bytes pulled through L2.
template <int NT = 256, typename F>
__device__ __forceinline__ void reduce_partials_4ch(const float* __restrict__ partials, int P, int C, int c0, int R, long long M, F&& fin, long long ldp = 0) {
    if (ldp == 0) ldp = C;                 // row length of the partials (> C: this tensor is a column slice of a wider set)
    __shared__ double red[8][NT / 64];   // [stat*4+ch][wave]
    const int t = threadIdx.x;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool vec = (C & 3) == 0 && (ldp & 3) == 0 && ((uintptr_t)partials & 15) == 0;
    auto inv_rows = [&](int p) -> double {          // 1 / rows of partial p (centred partials), 0 for plain sums
        if (R <= 0) return 0.0;
        const long long left = M - (long long)p * R;
        return 1.0 / (double)(left < R ? left : R);
    };
    if (vec) {
        // the loads of successive partials are independent: four rows (eight 16-byte loads) in flight per thread.  (With the
        // vec / scalar branch inside the loop the compiler waited for each row before it issued the next: 17 us at P = 4096.)
        int p = t;
        for (; p + 3 * NT < P; p += 4 * NT) {
            f32x4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* r0 = partials + ((long long)(p + u * NT) * 2) * ldp + c0;
                a[u] = ld4(r0); b[u] = ld4(r0 + ldp);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double inv_n = inv_rows(p + u * NT);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[j] += (double)a[u][j]; acc[4 + j] += (double)b[u][j] + (double)a[u][j] * (double)a[u][j] * inv_n; }
            }
        }
        for (; p < P; p += NT) {
            const float* r0 = partials + ((long long)p * 2) * ldp + c0;
            const f32x4 a = ld4(r0), b = ld4(r0 + ldp);
            const double inv_n = inv_rows(p);
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[j] += (double)a[j]; acc[4 + j] += (double)b[j] + (double)a[j] * (double)a[j] * inv_n; }
        }
    } else {
        for (int p = t; p < P; p += NT) {
            const float* r0 = partials + ((long long)p * 2 + 0) * ldp + c0;
            const float* r1 = partials + ((long long)p * 2 + 1) * ldp + c0;
            const double inv_n = inv_rows(p);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < C) { acc[j] += (double)r0[j]; acc[4 + j] += (double)r1[j] + (double)r0[j] * (double)r0[j] * inv_n; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = wave_sum_d(acc[j]);
    if ((t & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[j][t >> 6] = acc[j];
    }
    __syncthreads();
    if (t < 4 && c0 + t < C) {
        double s = 0, q = 0;
#pragma unroll
        for (int wv = 0; wv < NT / 64; wv += 4) {
            s += (red[t][wv] + red[t][wv + 1]) + (red[t][wv + 2] + red[t][wv + 3]);
            q += (red[4 + t][wv] + red[4 + t][wv + 1]) + (red[4 + t][wv + 2] + red[4 + t][wv + 3]);
        }
        fin(c0 + t, s, q);
    }
}

// Long partial lists (P >= 4096: the 128x128 layers and the stem) fold first: `chunk` consecutive partials -> one partial of chunk * R
// rows, in the SAME format (centred (sum, M2) pairs when R > 0, plain sums when R == 0), so the finalize kernels above then run on
// P / chunk rows.  The one-stage reduce owns 4 channels per block -- 16 of a row's bytes out of every 128-byte line it pulls, and at
// C = 64 only 16 CUs take part (17-31 us at P = 4096, 51 us for the stem).  Here a block owns 32 adjacent channels (whole lines: 8
// float4 lanes x 32 row lanes) of one chunk, so the grid is (C / 32) x (P / chunk) blocks over the whole chip.
// Centred merge in f64: M2 = sum M2_p + sum (s_p^2 / n_p) - S^2 / n with S = sum s_p, n = sum n_p.
__global__ __launch_bounds__(256) void bn_fold_partials_kernel(const float* __restrict__ partials, int P, int C, long long M, int R, int chunk,
                                                               float* __restrict__ out) {
    __shared__ double red[4][8][12];
    const int t = threadIdx.x, l = t & 7, rr = t >> 3;
    const int c = blockIdx.x * 32 + 4 * l, s = blockIdx.y;
    const int p0 = s * chunk, p1 = p0 + chunk < P ? p0 + chunk : P;
    double acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};          // [0..3] sum s_p, [4..7] sum M2_p (or the second plain sum), [8..11] sum s_p^2 / n_p
    if (c < C) {
        for (int p = p0 + rr; p < p1; p += 32) {
            const float* r0 = partials + ((long long)p * 2) * C + c;
            const f32x4 a = ld4(r0), b = ld4(r0 + C);
            double inv_n = 0.0;
            if (R > 0) { const long long left = M - (long long)p * R; inv_n = 1.0 / (double)(left < R ? left : R); }
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[j] += (double)a[j]; acc[4 + j] += (double)b[j]; acc[8 + j] += (double)a[j] * (double)a[j] * inv_n; }
        }
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) {
#pragma unroll
        for (int o = 32; o >= 8; o >>= 1) acc[j] += __shfl_xor(acc[j], o, 64);
    }
    if ((t & 63) < 8) {
#pragma unroll
        for (int j = 0; j < 12; ++j) red[t >> 6][l][j] = acc[j];
    }
    __syncthreads();
    if (t < 8 && c < C) {
        double v[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) v[j] = (red[0][t][j] + red[1][t][j]) + (red[2][t][j] + red[3][t][j]);
        f32x4 o0, o1;
        double n = 0.0;
        if (R > 0) { const long long left = M - (long long)p0 * R, full = (long long)chunk * R; n = (double)(left < full ? left : full); }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o0[j] = (float)v[j];
            double m2 = v[4 + j];
            if (R > 0) { m2 += v[8 + j] - v[j] * v[j] / n; if (m2 < 0) m2 = 0; }
            o1[j] = (float)m2;
        }
        st4(out + ((long long)s * 2) * C + c, o0);
        st4(out + ((long long)s * 2 + 1) * C + c, o1);
    }
}
// out: [ceil(n_partials / chunk)][2][C].  rows_per_partial > 0: centred partials of that many rows each (the last one short: count rows
// in all); 0: plain sums.  C % 4 == 0 and 16-byte aligned buffers, else SH_EUNSUPPORTED (the caller finalizes the list as it is).
extern "C" int sh_bn_fold_partials(const float* partials, int n_partials, int C, double count, int rows_per_partial, int chunk, float* out,
                                   void* stream) {
    if (!partials || !out || n_partials <= 0 || C <= 0 || chunk < 2 || rows_per_partial < 0 || (rows_per_partial > 0 && count <= 0)) return SH_EINVAL;
    if (rows_per_partial > 0 && (long long)n_partials != sh_cdiv((long long)count, rows_per_partial)) return SH_EINVAL;
    if ((C & 3) || (((uintptr_t)partials | (uintptr_t)out) & 15)) return SH_EUNSUPPORTED;
    dim3 grid((unsigned)sh_cdiv(C, 32), (unsigned)sh_cdiv(n_partials, chunk));
    bn_fold_partials_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(partials, n_partials, C, (long long)count, rows_per_partial, chunk, out);
    return sh_launch_status();
}

// wmul != nullptr (per-channel multiplier wmul[c * wstride]): the statistics are those of x but the BatchNorm normalises y = w * x
// (a depthwise conv whose off-centre taps never touch the image, SURVEY A.1: dilation >= H, W) -- mean_y = w mean_x, var_y = w^2 var_x.
// The coefficients are then emitted in the x domain so that every consumer can work on x itself:
//   mean = mean_x, invstd = w * invstd_y  (=> xhat_y = (x - mean) * invstd, and gamma * invstd * (...) is already d/dx),
//   scale = w * gamma * invstd_y, shift = beta - mean_y * gamma * invstd_y  (=> y_bn = x * scale + shift);  isy[c] = invstd_y.
template <int NT>
__global__ __launch_bounds__(NT) void bn_finalize_kernel(const float* __restrict__ partials, int P, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, float* running_mean,
                                                          float* running_var, float* mean, float* invstd, float* scale,
                                                          float* shift, int R, long long ldp, const float* __restrict__ wmul, int wstride,
                                                          float* __restrict__ isy) {
    // the per-channel operands of the epilogue are fetched BEFORE the reduction (threads 0..3 own channels c0..c0+3 there): the kernel is a
    // chain of dependent memory round trips on a few blocks, and this takes one of them off the critical path
    const int cpre = blockIdx.x * 4 + threadIdx.x;
    float g = 1.f, b = 0.f, w = 1.f, rm0 = 0.f, rv0 = 0.f;
    if (threadIdx.x < 4 && cpre < C) {
        if (gamma) g = gamma[cpre];
        if (beta) b = beta[cpre];
        if (wmul) w = wmul[(long long)cpre * wstride];
        if (running_mean) { rm0 = running_mean[cpre]; rv0 = running_var[cpre]; }
    }
    reduce_partials_4ch<NT>(partials, P, C, blockIdx.x * 4, R, (long long)count, [&](int c, double s, double q) {
        const double mu = s / count;
        double var = q / count - mu * mu;
        if (var < 0) var = 0;
        const float fmu = (float)mu;
        const float fmu_y = wmul ? w * fmu : fmu;
        const double var_y = wmul ? (double)w * (double)w * var : var;
        const float is = 1.0f / sqrtf((float)var_y + eps);
        mean[c] = fmu;
        invstd[c] = wmul ? w * is : is;
        const float sc = g * is;
        scale[c] = wmul ? w * sc : sc;
        shift[c] = b - fmu_y * sc;
        if (isy) isy[c] = is;
        if (running_mean) {
            const double unbiased = count > 1 ? var_y * (count / (count - 1.0)) : var_y;
            running_mean[c] = (1.f - momentum) * rm0 + momentum * fmu_y;
            running_var[c] = (1.f - momentum) * rv0 + momentum * (float)unbiased;
        }
    }, ldp);
}
extern "C" int sh_bn_finalize(const float* partials, int n_partials, int C, double count, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                              float* mean, float* invstd, float* scale, float* shift, int rows_per_partial, int partials_ld, void* stream) {
    if (!partials || n_partials <= 0 || C <= 0 || count <= 0 || !mean || !invstd || !scale || !shift || rows_per_partial <= 0) return SH_EINVAL;
    if ((long long)n_partials != sh_cdiv((long long)count, rows_per_partial)) return SH_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr) || (partials_ld != 0 && partials_ld < C)) return SH_EINVAL;
    if (n_partials >= 1024)
        bn_finalize_kernel<1024><<<(unsigned)sh_cdiv(C, 4), 1024, 0, (hipStream_t)stream>>>(
            partials, n_partials, C, count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, rows_per_partial,
            partials_ld, nullptr, 0, nullptr);
    else
        bn_finalize_kernel<256><<<(unsigned)sh_cdiv(C, 4), 256, 0, (hipStream_t)stream>>>(
            partials, n_partials, C, count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, rows_per_partial,
            partials_ld, nullptr, 0, nullptr);
    return sh_launch_status();
}
// Several BatchNorm layers of C channels each finalized in ONE launch (the grouped ASPP unit: its four pointwise BatchNorms share one
// partials tensor, layer k's columns start at partials_col[k]; its centre-tap depthwise BatchNorms share the statistics of c4 with
// a per-layer channel multiplier).  blockIdx.y = layer.
#define SH_BN_MULTI 8
struct BnMultiTab {
    const float* gamma[SH_BN_MULTI]; const float* beta[SH_BN_MULTI];
    float* rm[SH_BN_MULTI]; float* rv[SH_BN_MULTI];
    float* mean[SH_BN_MULTI]; float* invstd[SH_BN_MULTI]; float* scale[SH_BN_MULTI]; float* shift[SH_BN_MULTI];
    const float* wmul[SH_BN_MULTI]; float* isy[SH_BN_MULTI];
    int col[SH_BN_MULTI];
};
__global__ __launch_bounds__(256) void bn_finalize_multi_kernel(const float* __restrict__ partials, int P, int C, double count, float eps,
                                                                float momentum, int R, long long ldp, int wstride, const BnMultiTab T) {
    const int k = blockIdx.y;
    const float* gamma = T.gamma[k]; const float* beta = T.beta[k]; const float* wmul = T.wmul[k];
    float* running_mean = T.rm[k]; float* running_var = T.rv[k];
    float* mean = T.mean[k]; float* invstd = T.invstd[k]; float* scale = T.scale[k]; float* shift = T.shift[k]; float* isy = T.isy[k];
    reduce_partials_4ch(partials + T.col[k], P, C, blockIdx.x * 4, R, (long long)count, [&](int c, double s, double q) {
        const double mu = s / count;
        double var = q / count - mu * mu;
        if (var < 0) var = 0;
        const float w = wmul ? wmul[(long long)c * wstride] : 1.f;
        const float fmu = (float)mu;
        const float fmu_y = wmul ? w * fmu : fmu;
        const double var_y = wmul ? (double)w * (double)w * var : var;
        const float is = 1.0f / sqrtf((float)var_y + eps);
        mean[c] = fmu;
        invstd[c] = wmul ? w * is : is;
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        const float sc = g * is;
        scale[c] = wmul ? w * sc : sc;
        shift[c] = b - fmu_y * sc;
        if (isy) isy[c] = is;
        if (running_mean) {
            const double unbiased = count > 1 ? var_y * (count / (count - 1.0)) : var_y;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * fmu_y;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }, ldp);
}
extern "C" int sh_bn_finalize_multi(int k, const float* partials, int n_partials, int C, double count, int rows_per_partial, int partials_ld,
                                    const int* partials_col, const float* const* gamma, const float* const* beta, float* const* running_mean,
                                    float* const* running_var, float* const* mean, float* const* invstd, float* const* scale,
                                    float* const* shift, const float* const* chan_mul, int chan_stride, float* const* isy, float eps,
                                    float momentum, void* stream) {
    if (k < 1 || k > SH_BN_MULTI || !partials || n_partials <= 0 || C <= 0 || count <= 0 || rows_per_partial <= 0 || !partials_col || !gamma ||
        !beta || !running_mean || !running_var || !mean || !invstd || !scale || !shift) return SH_EINVAL;
    if ((long long)n_partials != sh_cdiv((long long)count, rows_per_partial) || (partials_ld != 0 && partials_ld < C)) return SH_EINVAL;
    BnMultiTab T;
    for (int i = 0; i < k; ++i) {
        if (!mean[i] || !invstd[i] || !scale[i] || !shift[i] || partials_col[i] < 0 || (running_mean[i] == nullptr) != (running_var[i] == nullptr))
            return SH_EINVAL;
        T.gamma[i] = gamma[i]; T.beta[i] = beta[i]; T.rm[i] = running_mean[i]; T.rv[i] = running_var[i];
        T.mean[i] = mean[i]; T.invstd[i] = invstd[i]; T.scale[i] = scale[i]; T.shift[i] = shift[i];
        T.wmul[i] = chan_mul ? chan_mul[i] : nullptr; T.isy[i] = isy ? isy[i] : nullptr; T.col[i] = partials_col[i];
        if (T.wmul[i] && chan_stride <= 0) return SH_EINVAL;
    }
    dim3 grid((unsigned)sh_cdiv(C, 4), (unsigned)k);
    bn_finalize_multi_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(partials, n_partials, C, count, eps, momentum, rows_per_partial, partials_ld,
                                                                   chan_stride, T);
    return sh_launch_status();
}
// BatchNorm of y = w[c] * x from the statistics partials of x (see bn_finalize_kernel): the depthwise + BN of an ASPP branch whose
// dilation exceeds the feature map (sep_aspp_contrast_head.py:125-131 at stride 32) without ever forming y.  chan_mul: the depthwise
// weight's centre taps = weight + 4 with chan_stride 9.  isy[C] <- 1/sqrt(var_y + eps) (for sh_dw_center_wgrad).
extern "C" int sh_bn_finalize_scaled(const float* partials, int n_partials, int C, double count, const float* chan_mul, int chan_stride,
                                     const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                     float* running_var, float* mean, float* invstd, float* scale, float* shift, float* isy,
                                     int rows_per_partial, void* stream) {
    if (!partials || n_partials <= 0 || C <= 0 || count <= 0 || !chan_mul || chan_stride <= 0 || !mean || !invstd || !scale || !shift || !isy ||
        rows_per_partial <= 0) return SH_EINVAL;
    if ((long long)n_partials != sh_cdiv((long long)count, rows_per_partial)) return SH_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return SH_EINVAL;
    bn_finalize_kernel<256><<<(unsigned)sh_cdiv(C, 4), 256, 0, (hipStream_t)stream>>>(
        partials, n_partials, C, count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, rows_per_partial,
        0, chan_mul, chan_stride, isy);
    return sh_launch_status();
}
// Weight gradient of such a centre-tap depthwise conv, in closed form from the BatchNorm-backward sums: with y = w x the loss
// depends on w only through eps -- dL/dw_centre = gamma * dgamma * eps * invstd_y^2 / w (derivation in DESIGN.md) -- and the eight
// off-centre taps never touch the image: exact zeros.  dw: [C][9].
// A centre tap that is EXACTLY zero (zero-initialised or pruned weights) carries no information in dgamma (= w * isy * S with
// S = sum g * (x - mean_x)): that channel's block forms S itself from the masked gradient g and the input x, dL/dw = gamma * isy * S
// (the same closed form with eps * isy^2 = 1 at w = 0).  One block of 64 threads per channel; only zero-weight channels loop.
__global__ __launch_bounds__(64) void dw_center_wgrad_kernel(const float* __restrict__ dgamma, const float* __restrict__ gamma,
                                                             const float* __restrict__ isy, const float* __restrict__ w, float eps,
                                                             float* __restrict__ dw, int C, const float* __restrict__ g, long long ldg,
                                                             const float* __restrict__ x, long long ldx, const float* __restrict__ mean_x,
                                                             long long M) {
    const int c = blockIdx.x, t = threadIdx.x;
    const float wc = w[(long long)c * 9 + 4];
    const float gm = gamma ? gamma[c] : 1.f;
    float v = 0.f;
    if (wc != 0.f) v = gm * dgamma[c] * eps * isy[c] * isy[c] / wc;
    else if (g != nullptr) {                                   // block-uniform branch
        const float mu = mean_x[c];
        double s = 0.0;
        for (long long m = t; m < M; m += 64) s += (double)g[m * ldg + c] * (double)(x[m * ldx + c] - mu);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        v = (float)((double)gm * (double)isy[c] * s);
    }
    if (t < 9) dw[(long long)c * 9 + t] = t == 4 ? v : 0.f;
}
extern "C" int sh_dw_center_wgrad(const float* dgamma, const float* gamma, const float* isy, const float* w, float eps, float* dw, int C,
                                  const float* g, int ldg, const float* x, int ldx, const float* mean_x, int64_t M, void* stream) {
    if (!dgamma || !isy || !w || !dw || C <= 0) return SH_EINVAL;
    if (g && (!x || !mean_x || ldg < C || ldx < C || M <= 0)) return SH_EINVAL;
    dw_center_wgrad_kernel<<<(unsigned)C, 64, 0, (hipStream_t)stream>>>(dgamma, gamma, isy, w, eps, dw, C, g, ldg, x, ldx, mean_x, M);
    return sh_launch_status();
}

__global__ __launch_bounds__(256) void bn_eval_coefs_kernel(const float* gamma, const float* beta, const float* rm,
                                                            const float* rv, float eps, int C, float* scale, float* shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float is = 1.0f / sqrtf(rv[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}
extern "C" int sh_bn_eval_coefs(const float* gamma, const float* beta, const float* running_mean,
                                const float* running_var, float eps, int C, float* scale, float* shift, void* stream) {
    if (!running_mean || !running_var || !scale || !shift || C <= 0) return SH_EINVAL;
    bn_eval_coefs_kernel<<<(unsigned)sh_cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(gamma, beta, running_mean, running_var, eps, C, scale, shift);
    return sh_launch_status();
}

// ---------------------------------------------------------------------------------------------- generic channel stats
// partial p covers rows [256p, 256p+256); block = (row chunk, 64-channel chunk); thread = (channel, row group of 4).
#define STAT_ROWS 256
template <int KIND>   // 0: (sum y, sum y^2)   1: BN backward (sum g, sum g*xhat)
__global__ __launch_bounds__(256) void channel_partials_kernel(const float* __restrict__ y, long long ldy,
                                                               const float* __restrict__ dout, long long lddo,
                                                               const float* __restrict__ out, long long ldo,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               float* __restrict__ partials, long long M, int C, int relu) {
    __shared__ float red[2][4][64];
    const int t = threadIdx.x, cl = t & 63, g = t >> 6;
    const int c = blockIdx.y * 64 + cl;
    const long long rbeg = (long long)blockIdx.x * STAT_ROWS;
    const long long rend = rbeg + STAT_ROWS < M ? rbeg + STAT_ROWS : M;
    float s = 0.f, q = 0.f;
    if (c < C) {
        float mu = 0.f, is = 0.f, sc = 0.f, sh = 0.f;
        if (KIND == 1) { mu = mean[c]; is = invstd[c]; if (relu == 2) { sc = scale[c]; sh = shift[c]; } }
        for (long long r = rbeg + g; r < rend; r += 4) {
            if (KIND == 0) {
                const float v = y[r * ldy + c];
                s += v; q += v * v;
            } else {
                float gv = dout[r * lddo + c];
                const float yv = y[r * ldy + c];
                if (relu == 1 && !(out[r * ldo + c] > 0.f)) gv = 0.f;
                if (relu == 2 && !(yv * sc + sh > 0.f)) gv = 0.f;       // the forward's own arithmetic (bn_act_kernel)
                const float xh = (yv - mu) * is;
                s += gv; q += gv * xh;
            }
        }
    }
    if (KIND == 0) {
        // centred form: per-thread (n, sum, M2) over its rows, Chan-merged over the 4 row groups
        float n = 0.f, mean = 0.f, m2 = 0.f;
        if (c < C)
            for (long long r = rbeg + g; r < rend; r += 4) {
                const float v = y[r * ldy + c];
                n += 1.f;
                const float d = v - mean;
                mean += d / n;
                m2 += d * (v - mean);
            }
        red[0][g][cl] = n * mean; red[1][g][cl] = m2;
        __shared__ float cnt[4][64];
        cnt[g][cl] = n;
        __syncthreads();
        if (t < 64) {
            const int ch = blockIdx.y * 64 + t;
            if (ch < C) {
                float N = cnt[0][t], S = red[0][0][t], Q = red[1][0][t];
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    const float nb = cnt[k][t], sb = red[0][k][t];
                    if (nb > 0.f) {
                        if (N > 0.f) { const float dm = S / N - sb / nb; Q += red[1][k][t] + dm * dm * N * nb / (N + nb); }
                        else Q = red[1][k][t];
                        N += nb; S += sb;
                    }
                }
                partials[((long long)blockIdx.x * 2 + 0) * C + ch] = S;
                partials[((long long)blockIdx.x * 2 + 1) * C + ch] = Q;
            }
        }
        return;
    }
    red[0][g][cl] = s; red[1][g][cl] = q;
    __syncthreads();
    if (t < 128) {
        const int st = t >> 6, cc = t & 63, ch = blockIdx.y * 64 + cc;
        if (ch < C)
            partials[((long long)blockIdx.x * 2 + st) * C + ch] = (red[st][0][cc] + red[st][1][cc]) + (red[st][2][cc] + red[st][3][cc]);
    }
}
// float4 form of the plain statistics: block = 256 rows x 64 channels, thread = (channel quad, row lane) holding its 16 rows in
// registers: one pass over y, block sum -> block mean through LDS, then M2 about that mean (centred partial, as the conv epilogues)
__global__ __launch_bounds__(256) void channel_stats_v4_kernel(const float* __restrict__ y, long long ldy, float* __restrict__ partials,
                                                               long long M, int C) {
    __shared__ float red[16][64];
    __shared__ float colmean[64];
    const int t = threadIdx.x, cq = t & 15, rl = t >> 4;
    const unsigned nch = (unsigned)((C + 63) / 64), rb = blockIdx.x / nch, cb = blockIdx.x % nch;
    const int c = cb * 64 + cq * 4;
    const long long rbeg = (long long)rb * STAT_ROWS;
    const long long left = M - rbeg;
    const float nvalid = (float)(left < STAT_ROWS ? left : STAT_ROWS);
    f32x4 v[STAT_ROWS / 16];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < STAT_ROWS / 16; ++k) {
        const long long r = rbeg + rl + 16 * k;
        v[k] = (c < C && r < M) ? ld4(y + r * ldy + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += v[k];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][cq * 4 + j] = s[j];
    __syncthreads();
    float colsum = 0.f;
    if (t < 64) {
#pragma unroll
        for (int k = 0; k < 16; ++k) colsum += red[k][t];
        colmean[t] = colsum / nvalid;
    }
    __syncthreads();
    const f32x4 mu = ld4(&colmean[cq * 4]);
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < STAT_ROWS / 16; ++k)
        if (rbeg + rl + 16 * k < M) { const f32x4 d = v[k] - mu; q += d * d; }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][cq * 4 + j] = q[j];
    __syncthreads();
    if (t < 64 && cb * 64 + t < C) {
        float m2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) m2 += red[k][t];
        partials[((long long)rb * 2 + 0) * C + cb * 64 + t] = colsum;
        partials[((long long)rb * 2 + 1) * C + cb * 64 + t] = m2;
    }
}
extern "C" int sh_stats_partials_count(int64_t M) { return (int)sh_cdiv(M, STAT_ROWS); }
extern "C" int sh_stats_tile_rows(void) { return STAT_ROWS; }
extern "C" int sh_channel_stats(const float* y, int ldy, int64_t M, int C, float* partials, void* stream) {
    if (!y || !partials || M <= 0 || C <= 0 || ldy < C) return SH_EINVAL;
    dim3 grid((unsigned)sh_cdiv(M, STAT_ROWS), (unsigned)sh_cdiv(C, 64));
    if ((C & 3) == 0 && (ldy & 3) == 0 && ((uintptr_t)y & 15) == 0)
        channel_stats_v4_kernel<<<grid.x * grid.y, 256, 0, (hipStream_t)stream>>>(y, ldy, partials, M, C);
    else
        channel_partials_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(y, ldy, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, partials, M, C, 0);
    return sh_launch_status();
}
// float4 variant of the BN-backward partials: block = 256 rows x 64 channels, thread = (channel quad, row lane),
// 16 rows per thread with 4 rows (12 x 16-byte loads) in flight.
__global__ __launch_bounds__(256) void bn_bwd_partials_v4_kernel(const float* __restrict__ y, long long ldy,
                                                                 const float* __restrict__ dout, long long lddo,
                                                                 const float* __restrict__ out, long long ldo,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 float* __restrict__ partials, long long M, int C, int relu,
                                                                 float* __restrict__ g_out, long long ldg, int af) {
    __shared__ float red[2][16][64];
    const int t = threadIdx.x, cq = t & 15, rl = t >> 4;
    // 1-D grid, channel chunk fastest: blocks that run together read adjacent 256-byte pieces of the same rows
    const unsigned nch = (unsigned)((C + 63) / 64), rb = blockIdx.x / nch, cb = blockIdx.x % nch;
    const int c = cb * 64 + cq * 4;
    const long long rbeg = (long long)rb * STAT_ROWS;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const f32x4 mu = ld4(mean + c), is = ld4(invstd + c);
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
        if (relu == 2) { sc = ld4(scale + c); sh = ld4(shift + c); }
#pragma unroll 4
        for (int k = 0; k < STAT_ROWS / 16; ++k) {
            const long long r = rbeg + rl + 16 * k;
            if (r < M) {
                f32x4 g = lda4(dout, r * lddo + c, af & 4);                 // af: bit 0 y, 1 out, 2 dout, 3 g_out stored as bf16
                const f32x4 yv = lda4(y, r * ldy + c, af & 1);
                if (relu == 1) {
                    const f32x4 o = lda4(out, r * ldo + c, af & 2);
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
                } else if (relu == 2) {                 // no residual: the mask is the sign of the forward's y*scale+shift
                    const f32x4 o = yv * sc + sh;
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
                } else if (relu == 3) {                 // `out` is the ReLU quad mask sh_bn_act wrote (ldo bytes per pixel)
                    const f32x4 o = quad_mask_load(out, r * ldo + (c >> 2));
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
                }
                s += g;
                q += g * ((yv - mu) * is);
                if (g_out) sta4(g_out, r * ldg + c, g, af & 8);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][rl][cq * 4 + j] = s[j]; red[1][rl][cq * 4 + j] = q[j]; }
    __syncthreads();
    if (t < 128) {
        const int st = t >> 6, cc = t & 63, ch = cb * 64 + cc;
        if (ch < C) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) a += red[st][k][cc];
            partials[((long long)rb * 2 + st) * C + ch] = a;
        }
    }
}
extern "C" int sh_bn_bwd_reduce(const float* dout, int lddo, const float* out, int ldo, const float* y, int ldy,
                                const float* mean, const float* invstd, const float* scale, const float* shift, float* partials,
                                int64_t M, int C, int relu, float* g_out, int ldg, int act_flags, void* stream) {
    if (!dout || !y || !mean || !invstd || !partials || M <= 0 || C <= 0 || lddo < C || ldy < C) return SH_EINVAL;
    if (g_out && (ldg < C || (ldg & 3) || ((uintptr_t)g_out & 15))) return SH_EINVAL;
    if (relu < 0 || relu > 3 || (relu == 1 && (!out || ldo < C)) || (relu == 2 && (!scale || !shift)) || (relu == 3 && (!out || ldo * 4 < C))) return SH_EINVAL;
    dim3 grid((unsigned)sh_cdiv(M, STAT_ROWS), (unsigned)sh_cdiv(C, 64));
    const bool v4 = (C & 3) == 0 && ((lddo | ldy | (relu == 1 ? ldo : 0)) & 3) == 0 && ((uintptr_t)dout & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                    (relu != 1 || ((uintptr_t)out & 15) == 0) && ((uintptr_t)mean & 15) == 0 && ((uintptr_t)invstd & 15) == 0 &&
                    (relu != 2 || ((((uintptr_t)scale | (uintptr_t)shift) & 15) == 0));
    if (v4) bn_bwd_partials_v4_kernel<<<grid.x * grid.y, 256, 0, (hipStream_t)stream>>>(y, ldy, dout, lddo, out, ldo, mean, invstd, scale, shift, partials, M, C, relu, g_out, ldg, act_flags);
    else if (g_out || act_flags || relu == 3) return SH_EUNSUPPORTED;          // masked-gradient output, bf16 tensors, quad masks: the 16-byte kernel only
    else channel_partials_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(y, ldy, dout, lddo, out, ldo, mean, invstd, scale, shift, partials, M, C, relu);
    return sh_launch_status();
}

// The second half of the BatchNorm backward as ONE per-channel linear form in the two tensors it reads,
//   dy = gamma*invstd*(g - c1 - xhat*c2) = A*g + B*(y - mean) + D,   A = gamma*invstd, B = -A*invstd*c2, D = -A*c1,
// lin[0..3][C] = (A, B, mean, D): the 1x1 consumers of dy (sh_conv_dgrad_x6_lin, sh_conv_wgrad_x6_lin) evaluate it in their
// loaders as fma(y - mean, B, fma(g, A, D)) and the sh_bn_bwd_apply pass with its dy tensor disappears.
__device__ __forceinline__ void bn_bwd_lin(float* lin, int C, int c, float ga, float is, float mu, float c1, float c2) {
    const float A = ga * is;
    lin[c] = A; lin[C + c] = -(A * is) * c2; lin[2 * C + c] = mu; lin[3 * C + c] = -A * c1;
}
template <int NT>
__global__ __launch_bounds__(NT) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int P, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                              double count, float* dgamma, float* dbeta, float* c1, float* c2,
                                                              const float* __restrict__ mean, float* __restrict__ lin) {
    const int cpre = blockIdx.x * 4 + threadIdx.x;          // epilogue operands fetched before the reduction (see bn_finalize_kernel)
    float ga = 1.f, is = 0.f, mu = 0.f;
    if (lin && threadIdx.x < 4 && cpre < C) { if (gamma) ga = gamma[cpre]; is = invstd[cpre]; mu = mean[cpre]; }
    reduce_partials_4ch<NT>(partials, P, C, blockIdx.x * 4, 0, 0, [&](int c, double s, double q) {
        if (dbeta) dbeta[c] = (float)s;
        if (dgamma) dgamma[c] = (float)q;
        const float f1 = (float)(s / count), f2 = (float)(q / count);
        c1[c] = f1;
        c2[c] = f2;
        if (lin) bn_bwd_lin(lin, C, c, ga, is, mu, f1, f2);
    });
}
extern "C" int sh_bn_bwd_finalize(const float* partials, int n_partials, int C, const float* gamma, const float* invstd,
                                  double count, float* dgamma, float* dbeta, float* c1, float* c2, const float* mean, float* lin,
                                  void* stream) {
    if (!partials || n_partials <= 0 || C <= 0 || count <= 0 || !c1 || !c2 || (lin && (!mean || !invstd))) return SH_EINVAL;
    if (n_partials >= 1024)
        bn_bwd_finalize_kernel<1024><<<(unsigned)sh_cdiv(C, 4), 1024, 0, (hipStream_t)stream>>>(partials, n_partials, C, gamma, invstd, count, dgamma, dbeta, c1, c2, mean, lin);
    else
        bn_bwd_finalize_kernel<256><<<(unsigned)sh_cdiv(C, 4), 256, 0, (hipStream_t)stream>>>(partials, n_partials, C, gamma, invstd, count, dgamma, dbeta, c1, c2, mean, lin);
    return sh_launch_status();
}

#define EW_ROWS 16
// ---------------------------------------------------------------------------------------------- BN apply (+residual, +ReLU)
template <int V>
__global__ __launch_bounds__(256) void bn_act_kernel(const float* __restrict__ y, long long ldy, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const float* __restrict__ res, long long ldr,
                                                     float* __restrict__ out, long long ldo, long long M, int C, int relu,
                                                     const float* __restrict__ rscale, const float* __restrict__ rshift, int af,
                                                     unsigned char* __restrict__ qmask) {
    const int cv = C / V;
    // chunks of EW_ROWS rows per block iteration; 32-bit (row, column) split inside a chunk
    for (long long m0 = (long long)blockIdx.x * EW_ROWS; m0 < M; m0 += (long long)gridDim.x * EW_ROWS)
    for (int e = threadIdx.x, tot = (int)((M - m0 < EW_ROWS ? M - m0 : EW_ROWS)) * cv; e < tot; e += 256) {
        const int r = e / cv;
        const long long m = m0 + r;
        const int c = (e - r * cv) * V;
        if (V == 8) {          // 16-byte accesses of bf16 tensors (two 4-channel halves, each in the 4-channel form's own arithmetic)
            const sh_f8 yv = lda8(y, m * ldy + c, af & 1);
            sh_f8 v;
            v.lo = yv.lo * ld4(scale + c) + ld4(shift + c); v.hi = yv.hi * ld4(scale + c + 4) + ld4(shift + c + 4);
            if (res) {
                sh_f8 r = lda8(res, m * ldr + c, af & 2);
                if (rscale) { r.lo = r.lo * ld4(rscale + c) + ld4(rshift + c); r.hi = r.hi * ld4(rscale + c + 4) + ld4(rshift + c + 4); }
                v.lo += r.lo; v.hi += r.hi;
            }
            if (relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { v.lo[j] = fmaxf(v.lo[j], 0.f); v.hi[j] = fmaxf(v.hi[j], 0.f); }
            }
            sta8(out, m * ldo + c, v, af & 4);
            if (qmask) *reinterpret_cast<unsigned short*>(qmask + m * (C / 4) + (c >> 2)) = (unsigned short)(quad_mask_bits(v.lo) | (quad_mask_bits(v.hi) << 8));
        } else if (V == 4) {
            f32x4 v = lda4(y, m * ldy + c, af & 1) * ld4(scale + c) + ld4(shift + c);        // af: bit 0 y, 1 residual, 2 out stored as bf16
            if (res) {
                f32x4 r = lda4(res, m * ldr + c, af & 2);
                if (rscale) r = r * ld4(rscale + c) + ld4(rshift + c);      // the downsample path's BatchNorm applied on the fly
                v += r;
            }
            if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            sta4(out, m * ldo + c, v, af & 4);
            if (qmask) qmask[m * cv + (c >> 2)] = (unsigned char)quad_mask_bits(v);           // (common.h: ReLU quad mask)
        } else {
            float v = y[m * ldy + c] * scale[c] + shift[c];
            if (res) v += rscale ? res[m * ldr + c] * rscale[c] + rshift[c] : res[m * ldr + c];
            if (relu) v = fmaxf(v, 0.f);
            out[m * ldo + c] = v;
        }
    }
}
static inline unsigned rows_grid(long long M, int cv) {
    (void)cv;
    long long g = sh_cdiv(M, EW_ROWS);
    if (g > 8192) g = 8192;
    return (unsigned)(g < 1 ? 1 : g);
}
static inline bool vec4_ok(int C, long long a, long long b = 0, long long c = 0, long long d = 0, long long e = 0) {
    return (C & 3) == 0 && ((a | b | c | d | e) & 3) == 0;
}
static inline bool ptr16(const void* p) { return ((uintptr_t)p & 15) == 0; }
// SEGHIERO_EW8=0: the 4-channel form also for bf16 tensors (A/B and the bit-equality test of the 8-channel form)
static inline bool ew8_on() { const char* e = getenv("SEGHIERO_EW8"); return !(e && e[0] == '0'); }

extern "C" int sh_bn_act(const float* y, int ldy, const float* scale, const float* shift, const float* residual, int ldr,
                         const float* res_scale, const float* res_shift, float* out, int ldo, int64_t M, int C, int relu, uint8_t* relu_mask,
                         int act_flags, void* stream) {
    if (!y || !scale || !shift || !out || M <= 0 || C <= 0 || ldy < C || ldo < C) return SH_EINVAL;
    if (residual && ldr < C) return SH_EINVAL;
    if ((res_scale == nullptr) != (res_shift == nullptr) || (res_scale && !residual)) return SH_EINVAL;
    // (bf16 tensors: 8-byte accesses -- base pointers 8-byte aligned suffice, 16 is what the allocator gives anyway)
    const bool v4 = vec4_ok(C, ldy, ldo, residual ? ldr : 0) && ptr16(y) && ptr16(out) && ptr16(scale) && ptr16(shift) && (!residual || ptr16(residual)) &&
                    (!res_scale || (ptr16(res_scale) && ptr16(res_shift)));
    if ((act_flags || relu_mask) && !v4) return SH_EUNSUPPORTED;
    // bf16 tensors with 16-byte rows: eight channels per lane (one 16-byte access per tensor instead of 8 bytes)
    const bool v8 = v4 && ew8_on() && (act_flags & 1) && (C & 7) == 0 && ((ldy | ldo | (residual ? ldr : 0)) & 7) == 0 && (!relu_mask || ((uintptr_t)relu_mask & 1) == 0);
    if (v8) bn_act_kernel<8><<<rows_grid(M, C / 8), 256, 0, (hipStream_t)stream>>>(y, ldy, scale, shift, residual, ldr, out, ldo, M, C, relu, res_scale, res_shift, act_flags, relu_mask);
    else if (v4) bn_act_kernel<4><<<rows_grid(M, C / 4), 256, 0, (hipStream_t)stream>>>(y, ldy, scale, shift, residual, ldr, out, ldo, M, C, relu, res_scale, res_shift, act_flags, relu_mask);
    else bn_act_kernel<1><<<rows_grid(M, C), 256, 0, (hipStream_t)stream>>>(y, ldy, scale, shift, residual, ldr, out, ldo, M, C, relu, res_scale, res_shift, 0, nullptr);
    return sh_launch_status();
}

template <int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, long long lddo, const float* __restrict__ out,
                                                           long long ldo, const float* __restrict__ y, long long ldy,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ gamma, const float* __restrict__ c1,
                                                           const float* __restrict__ c2, float* __restrict__ dy, long long lddy,
                                                           float* __restrict__ dres, long long lddres, long long M, int C, int relu, int af) {
    const int cv = C / V;
    // chunks of EW_ROWS rows per block iteration; 32-bit (row, column) split inside a chunk
    for (long long m0 = (long long)blockIdx.x * EW_ROWS; m0 < M; m0 += (long long)gridDim.x * EW_ROWS)
    for (int e = threadIdx.x, tot = (int)((M - m0 < EW_ROWS ? M - m0 : EW_ROWS)) * cv; e < tot; e += 256) {
        const int r = e / cv;
        const long long m = m0 + r;
        const int c = (e - r * cv) * V;
        if (V == 8) {          // 16-byte accesses of bf16 tensors: the 4-channel arithmetic on two halves
            sh_f8 g8 = lda8(dout, m * lddo + c, af & 4);
            const sh_f8 y8 = lda8(y, m * ldy + c, af & 1);
            sh_f8 o8 = sh_f8{f32x4{1.f, 1.f, 1.f, 1.f}, f32x4{1.f, 1.f, 1.f, 1.f}};
            if (relu == 1) o8 = lda8(out, m * ldo + c, af & 2);
            else if (relu == 2) { o8.lo = y8.lo * ld4(scale + c) + ld4(shift + c); o8.hi = y8.hi * ld4(scale + c + 4) + ld4(shift + c + 4); }
            else if (relu == 3) { o8.lo = quad_mask_load(out, m * ldo + (c >> 2)); o8.hi = quad_mask_load(out, m * ldo + (c >> 2) + 1); }
            if (relu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { if (!(o8.lo[j] > 0.f)) g8.lo[j] = 0.f; if (!(o8.hi[j] > 0.f)) g8.hi[j] = 0.f; }
            }
            sh_f8 r8;
            {
                const f32x4 is = ld4(invstd + c), xh = (y8.lo - ld4(mean + c)) * is;
                f32x4 ga = {1.f, 1.f, 1.f, 1.f};
                if (gamma) ga = ld4(gamma + c);
                r8.lo = ga * is * (g8.lo - ld4(c1 + c) - xh * ld4(c2 + c));
            }
            {
                const f32x4 is = ld4(invstd + c + 4), xh = (y8.hi - ld4(mean + c + 4)) * is;
                f32x4 ga = {1.f, 1.f, 1.f, 1.f};
                if (gamma) ga = ld4(gamma + c + 4);
                r8.hi = ga * is * (g8.hi - ld4(c1 + c + 4) - xh * ld4(c2 + c + 4));
            }
            sta8(dy, m * lddy + c, r8, af & 8);
            if (dres) sta8(dres, m * lddres + c, g8, af & 16);
        } else if (V == 4) {
            f32x4 g = lda4(dout, m * lddo + c, af & 4);                     // af: bit 0 y, 1 out, 2 dout, 3 dy, 4 dres stored as bf16
            const f32x4 yv = lda4(y, m * ldy + c, af & 1);
            if (relu == 1) {
                const f32x4 o = lda4(out, m * ldo + c, af & 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) if (!(o[j] > 0.f)) g[j] = 0.f;
            } else if (relu == 2) {
                const f32x4 o = yv * ld4(scale + c) + ld4(shift + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) if (!(o[j] > 0.f)) g[j] = 0.f;
            } else if (relu == 3) {
                const f32x4 o = quad_mask_load(out, m * ldo + (c >> 2));
#pragma unroll
                for (int j = 0; j < 4; ++j) if (!(o[j] > 0.f)) g[j] = 0.f;
            }
            const f32x4 is = ld4(invstd + c);
            const f32x4 xh = (yv - ld4(mean + c)) * is;
            f32x4 ga = {1.f, 1.f, 1.f, 1.f};
            if (gamma) ga = ld4(gamma + c);
            const f32x4 r = ga * is * (g - ld4(c1 + c) - xh * ld4(c2 + c));
            sta4(dy, m * lddy + c, r, af & 8);
            if (dres) sta4(dres, m * lddres + c, g, af & 16);
        } else {
            float g = dout[m * lddo + c];
            const float yv = y[m * ldy + c];
            if (relu == 1 && !(out[m * ldo + c] > 0.f)) g = 0.f;
            if (relu == 2 && !(yv * scale[c] + shift[c] > 0.f)) g = 0.f;
            const float is = invstd[c];
            const float xh = (yv - mean[c]) * is;
            dy[m * lddy + c] = (gamma ? gamma[c] : 1.f) * is * (g - c1[c] - xh * c2[c]);
            if (dres) dres[m * lddres + c] = g;
        }
    }
}
extern "C" int sh_bn_bwd_apply(const float* dout, int lddo, const float* out, int ldo, const float* y, int ldy,
                               const float* mean, const float* invstd, const float* scale, const float* shift, const float* gamma,
                               const float* c1, const float* c2, float* dy, int lddy, float* dres, int lddres, int64_t M, int C, int relu,
                               int act_flags, void* stream) {
    if (!dout || !y || !mean || !invstd || !c1 || !c2 || !dy || M <= 0 || C <= 0 || lddo < C || ldy < C || lddy < C) return SH_EINVAL;
    if (relu < 0 || relu > 3 || (relu == 1 && (!out || ldo < C)) || (relu == 2 && (!scale || !shift)) || (relu == 3 && (!out || ldo * 4 < C))) return SH_EINVAL;
    if (dres && lddres < C) return SH_EINVAL;
    const bool v4 = vec4_ok(C, lddo, ldy, lddy, relu == 1 ? ldo : 0, dres ? lddres : 0) && ptr16(dout) && ptr16(y) && ptr16(dy) &&
                    ptr16(mean) && ptr16(invstd) && ptr16(c1) && ptr16(c2) && (!gamma || ptr16(gamma)) && (relu != 1 || ptr16(out)) &&
                    (relu != 2 || (ptr16(scale) && ptr16(shift))) && (!dres || ptr16(dres));
    if ((act_flags || relu == 3) && !v4) return SH_EUNSUPPORTED;
    const bool v8 = v4 && ew8_on() && (act_flags & 1) && (C & 7) == 0 && ((lddo | ldy | lddy | (relu == 1 ? ldo : 0) | (dres ? lddres : 0)) & 7) == 0 && (relu != 3 || (ldo & 1) == 0);
    if (v8) bn_bwd_apply_kernel<8><<<rows_grid(M, C / 8), 256, 0, (hipStream_t)stream>>>(dout, lddo, out, ldo, y, ldy, mean, invstd, scale, shift, gamma, c1, c2, dy, lddy, dres, lddres, M, C, relu, act_flags);
    else if (v4) bn_bwd_apply_kernel<4><<<rows_grid(M, C / 4), 256, 0, (hipStream_t)stream>>>(dout, lddo, out, ldo, y, ldy, mean, invstd, scale, shift, gamma, c1, c2, dy, lddy, dres, lddres, M, C, relu, act_flags);
    else bn_bwd_apply_kernel<1><<<rows_grid(M, C), 256, 0, (hipStream_t)stream>>>(dout, lddo, out, ldo, y, ldy, mean, invstd, scale, shift, gamma, c1, c2, dy, lddy, dres, lddres, M, C, relu, 0);
    return sh_launch_status();
}

// ---------------------------------------------------------------------------------------------- SyncBN building blocks
// Cross-GPU BatchNorm (new functionality): each rank reduces its partials to f64 per-channel sums, the host
// side all-reduces them over RCCL, and the finalize kernels run on the global sums.
//   forward : sq[0][c] = sum x, sq[1][c] = sum x^2 (assembled from the centred partials in f64)
//   backward: sq[0][c] = sum g, sq[1][c] = sum g*xhat
//   sq[2*C] = this rank's pixel count: after the all-reduce it is the global count the finalize kernels divide by, so ranks
//   may hold different numbers of pixels (torch.nn.SyncBatchNorm all-gathers the counts for the same reason)
__global__ __launch_bounds__(256) void bn_reduce_partials_kernel(const float* __restrict__ partials, int P, int C, int R, long long M,
                                                                 double* __restrict__ sq) {
    if (blockIdx.x == 0 && threadIdx.x == 0) sq[2 * C] = (double)M;      // the local count travels with the sums
    reduce_partials_4ch(partials, P, C, blockIdx.x * 4, R, M, [&](int c, double s, double q) { sq[c] = s; sq[C + c] = q; });
}
extern "C" int sh_bn_reduce_partials(const float* partials, int n_partials, int C, double count, int rows_per_partial, double* sq,
                                     void* stream) {
    if (!partials || !sq || n_partials <= 0 || C <= 0 || rows_per_partial < 0) return SH_EINVAL;
    bn_reduce_partials_kernel<<<(unsigned)sh_cdiv(C, 4), 256, 0, (hipStream_t)stream>>>(partials, n_partials, C, rows_per_partial,
                                                                                       (long long)count, sq);
    return sh_launch_status();
}
__global__ __launch_bounds__(256) void bn_finalize_sq_kernel(const double* __restrict__ sq, int C, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, float momentum, float* running_mean,
                                                             float* running_var, float* mean, float* invstd, float* scale, float* shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double count = sq[2 * C];
    const double mu = sq[c] / count;
    double var = sq[C + c] / count - mu * mu;
    if (var < 0) var = 0;
    const float fmu = (float)mu, is = 1.0f / sqrtf((float)var + eps);
    mean[c] = fmu; invstd[c] = is;
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    scale[c] = sc; shift[c] = (beta ? beta[c] : 0.f) - fmu * sc;
    if (running_mean) {
        const double unbiased = count > 1 ? var * (count / (count - 1.0)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * fmu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}
extern "C" int sh_bn_finalize_sq(const double* sq, int C, const float* gamma, const float* beta, float eps, float momentum,
                                 float* running_mean, float* running_var, float* mean, float* invstd, float* scale, float* shift,
                                 void* stream) {
    if (!sq || C <= 0 || !mean || !invstd || !scale || !shift) return SH_EINVAL;
    bn_finalize_sq_kernel<<<(unsigned)sh_cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(sq, C, gamma, beta, eps, momentum, running_mean,
                                                                                      running_var, mean, invstd, scale, shift);
    return sh_launch_status();
}
// dgamma / dbeta from the LOCAL sums (DDP sums them over ranks later), c1 / c2 from the GLOBAL sums and count
__global__ __launch_bounds__(256) void bn_bwd_finalize_sq_kernel(const double* __restrict__ local_sq, const double* __restrict__ global_sq, int C,
                                                                 float* dgamma, float* dbeta, float* c1, float* c2, const float* gamma,
                                                                 const float* invstd, const float* mean, float* lin) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double count = global_sq[2 * C];
    if (dbeta) dbeta[c] = (float)local_sq[c];
    if (dgamma) dgamma[c] = (float)local_sq[C + c];
    const float f1 = (float)(global_sq[c] / count), f2 = (float)(global_sq[C + c] / count);
    c1[c] = f1;
    c2[c] = f2;
    if (lin) bn_bwd_lin(lin, C, c, gamma ? gamma[c] : 1.f, invstd[c], mean[c], f1, f2);
}
extern "C" int sh_bn_bwd_finalize_sq(const double* local_sq, const double* global_sq, int C, float* dgamma, float* dbeta,
                                     float* c1, float* c2, const float* gamma, const float* invstd, const float* mean, float* lin, void* stream) {
    if (!local_sq || !global_sq || C <= 0 || !c1 || !c2 || (lin && (!mean || !invstd))) return SH_EINVAL;
    bn_bwd_finalize_sq_kernel<<<(unsigned)sh_cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(local_sq, global_sq, C, dgamma, dbeta, c1, c2, gamma, invstd, mean, lin);
    return sh_launch_status();
}
