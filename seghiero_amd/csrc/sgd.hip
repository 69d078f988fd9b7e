// Fused multi-tensor SGD with momentum and weight decay -- torch.optim.SGD(lr, momentum=0.9, weight_decay=1e-4).step()
// of reference train.py:239-246,317 (math: SURVEY A.9).  One launch updates up to SH_SGD_MAX tensors; the tensor table
// travels in the kernel arguments (no device-side pointer table to maintain).  HBM-bound: reads w,g,v, writes w,v.
#include "common.h"

struct SgdTab {
    float* w[SH_SGD_MAX];
    const float* g[SH_SGD_MAX];
    float* v[SH_SGD_MAX];
    long long start[SH_SGD_MAX + 1];   // prefix sums of ceil(numel/4) "quads" per tensor
    long long numel[SH_SGD_MAX];
    int n;
};

__global__ __launch_bounds__(256) void sgd_kernel(const SgdTab T, float lr, float mom, float wd, int first, float gscale) {
    const long long total = T.start[T.n];
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        int k = 0;                                   // binary search the owning tensor
        int lo = 0, hi = T.n - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (T.start[mid] <= q) lo = mid; else hi = mid - 1; }
        k = lo;
        const long long e = (q - T.start[k]) * 4, n = T.numel[k];
        float* w = T.w[k]; const float* g = T.g[k]; float* v = T.v[k];
        if (e + 4 <= n && (((uintptr_t)w | (uintptr_t)g | (uintptr_t)v) & 15) == 0) {
            f32x4 wv = ld4(w + e), gv = ld4(g + e) * gscale + wd * wv;
            f32x4 vv = first ? gv : mom * ld4(v + e) + gv;
            st4(v + e, vv);
            st4(w + e, wv - lr * vv);
        } else {
            for (long long j = e; j < n && j < e + 4; ++j) {
                const float wj = w[j], gj = g[j] * gscale + wd * wj;
                const float vj = first ? gj : mom * v[j] + gj;
                v[j] = vj; w[j] = wj - lr * vj;
            }
        }
    }
}

extern "C" int sh_sgd_step(int n_tensors, float* const* w, const float* const* g, float* const* v, const int64_t* numel,
                           float lr, float momentum, float weight_decay, int first_step, float gscale, void* stream) {
    if (n_tensors <= 0 || n_tensors > SH_SGD_MAX || !w || !g || !v || !numel) return SH_EINVAL;
    SgdTab T;
    T.n = n_tensors;
    long long acc = 0;
    for (int i = 0; i < n_tensors; ++i) {
        if (!w[i] || !g[i] || !v[i] || numel[i] <= 0) return SH_EINVAL;
        T.w[i] = w[i]; T.g[i] = g[i]; T.v[i] = v[i]; T.numel[i] = numel[i];
        T.start[i] = acc;
        acc += (numel[i] + 3) / 4;
    }
    T.start[n_tensors] = acc;
    long long grid = sh_cdiv(acc, 256);
    if (grid > 2048) grid = 2048;
    sgd_kernel<<<(unsigned)grid, 256, 0, (hipStream_t)stream>>>(T, lr, momentum, weight_decay, first_step, gscale);
    return sh_launch_status();
}

extern "C" int sh_copy(void* dst, const void* src, int64_t bytes, void* stream) {
    if (!dst || !src || bytes < 0) return SH_EINVAL;
    if (bytes == 0) return SH_OK;
    return hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? SH_OK : SH_ELAUNCH;
}
extern "C" int sh_abi_version(void) { return 1; }
