// Shared definitions of the fp32-accurate "6 x bf16 split" convolution kernels (conv_bf16x6.hip, conv_x6p.hip).
#pragma once
#include "common.h"
#include <stdlib.h>

enum { FPROP = 0, DGRAD = 1 };

struct ConvQ {
    const float* a;
    const float* b;
    float* c;
    const float* extra;     // fprop: bias[Cout] ; dgrad: addend[M][ldadd]
    float* partials;
    long long lda, ldb, ldc, ldadd;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, dil;
    int M, Nn, K, Kc;       // Kc = channels per tap along K (Cin for fprop, padded Cout for dgrad)
    int Kp;                 // row length (in k) of the pre-split B planes (multiple of 32)
    int Kreal;              // channels actually present per tap in the A rows (dgrad: pad4(Cout); fprop: Cin)
    long long bplane;       // elements per B plane
    int scatter, sH, sW, sstride;
    int parity;             // dgrad of a stride-2 KxK conv: blockIdx.y = input-pixel parity class (only its taps are non-zero)
    int kchunk;
    int tiles_m, tiles_n, n_partials;
    const float* act_scale; // ACT instantiations: per-output-channel scale / shift, optional residual (stride ldadd), ReLU flag
    const float* act_shift;
    const float* act_res;
    int act_relu;
    int ksplit;             // fprop / dgrad split-K: blockIdx.y = K slice, raw accumulators go to slab[ksplit][M][ldslab]
    float* slab;
    long long ldslab;
    // ---- fused BatchNorm hooks of the pipelined kernels (conv_x6p.hip)
    const float* aff_scale; // A operand (fprop) is relu(x * aff_scale[c] + aff_shift[c]) of the stored tensor: the producer's train-mode
    const float* aff_shift; // BatchNorm + ReLU applied in the loader (zero padding stays exact zero); nullptr = plain x
    const float* bnb_y;     // dgrad epilogue = front half of the BatchNorm backward of the layer that produced the conv's input:
    long long bnb_ldy;      //   g = relu-mask(y*scale+shift) * dx is stored instead of dx and (sum g, sum g*xhat) per 64 rows go to
    const float* bnb_mean;  //   `partials`; y = that layer's raw conv output [M][bnb_ldy], coefficients per channel
    const float* bnb_invstd;
    const float* bnb_scale;
    const float* bnb_shift;
    int bnb_relu;
    const float* bnb_out;   // optional: ReLU mask from this tensor (> 0) instead of y*scale+shift -- residual blocks, out = relu(bn(y) + identity)
    long long bnb_ldo;
    // ---- grouped 1x1 fprop (conv_x6p.hip, GRP): ngroups convolutions of the same geometry whose outputs are consecutive column slices of
    // one buffer; group g = output column / group_n reads its own input (through its own BatchNorm + ReLU, or plain: gfloor = -inf,
    // scale 1, shift 0) and its own weights
    int ngroups, group_n;
    const float* ga[6]; unsigned ga_bytes[6]; long long glda[6];
    const float* gb[6];
    const float* gsc[6]; const float* gsh[6];
    // ---- deferred BatchNorm-backward apply (AFF == 2 dgrad loader, LIN wgrad loader): the gradient operand is not stored; it is
    // lin(g, y) = lin[0][c]*g + lin[1][c]*(y - lin[2][c]) + lin[3][c] of the masked gradient g (p.a) and the raw conv output y (a2)
    const float* a2; long long lda2; unsigned a2_bytes;
    const float* lin;       // [4][Kc (dgrad) / M (wgrad)]
    // ---- bf16 activation storage (common.h): bit 0 the fprop input x, 1 the fprop output y, 2 bnb_y, 3 bnb_out, 4 the lin y stream (a2),
    // 5 the wgrad X operand are stored as bf16 (pixel strides count elements).  Loads inside a scheduled conv phase are selected by a
    // template parameter (ABF / XBF), epilogue accesses by this mask at run time.
    // bit 6 (64): bnb_out is the ReLU quad mask written by sh_bn_act (common.h; bnb_ldo = bytes per pixel), not the output tensor.
    int act;
    int vec_epi;            // output / addend / bnb_y rows are 16-byte addressable: row-major float4 epilogue through LDS
    unsigned a_bytes, b_bytes;  // extents of the A / B operands for the buffer descriptors (bytes, < 2^31)
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// exact 3-way split of 4 floats -> three packed bf16x4 (8 bytes each)
// (16 and/sub + 6 perm.  Tried: the residual subtractions on element pairs, v_pk_add_f32 -- 18 VALU instead of 22, same box 33.4-33.5
// vs 32.7-32.9 ms per step: the packed op is no cheaper to issue and its 64-bit-aligned operand pairs cost the scheduler freedom.)
__device__ __forceinline__ void split4(const f32x4 v, u32x2& p1, u32x2& p2, u32x2& p3) {
    unsigned h1[4], h2[4], h3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned u = __float_as_uint(v[j]);
        const float r1 = v[j] - __uint_as_float(u & 0xffff0000u);
        const unsigned u2 = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u2 & 0xffff0000u);
        h1[j] = u; h2[j] = u2; h3[j] = __float_as_uint(r2);
    }
    // perm(S0, S1, 0x07060302) = (S0 & 0xffff0000) | (S1 >> 16): element j in the low half, j+1 in the high half
    p1[0] = __builtin_amdgcn_perm(h1[1], h1[0], 0x07060302u); p1[1] = __builtin_amdgcn_perm(h1[3], h1[2], 0x07060302u);
    p2[0] = __builtin_amdgcn_perm(h2[1], h2[0], 0x07060302u); p2[1] = __builtin_amdgcn_perm(h2[3], h2[2], 0x07060302u);
    p3[0] = __builtin_amdgcn_perm(h3[1], h3[0], 0x07060302u); p3[1] = __builtin_amdgcn_perm(h3[3], h3[2], 0x07060302u);
}

// LDS rows of a K-contiguous plane are 64 bytes (32 bf16) with the 16-byte k-chunk XOR-swizzled by (row >> 2) & 3: the 16
// lanes of every ds_read_b128 lane group ({0-3,12-15,20-27}, ...) then cover all 64 banks once, and the ds_write_b64 of 16
// consecutive lanes (2 rows) covers the 32 store banks once -- conflict-free without padding (20 % less LDS than 80-byte rows).
#define ROWB 64
#define ROWB_G 80          // gemm_x6_kernel keeps padded rows (64 data + 16 pad)

// six-product accumulate of one 32x32 tile over K=16
__device__ __forceinline__ f32x16 mma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
    c = mfma_bf16(a[2], b[0], c);      // smallest terms first
    c = mfma_bf16(a[0], b[2], c);
    c = mfma_bf16(a[1], b[1], c);
    c = mfma_bf16(a[1], b[0], c);
    c = mfma_bf16(a[0], b[1], c);
    c = mfma_bf16(a[0], b[0], c);
    return c;
}


// three-product accumulate: one operand is an exact bf16 value (its mid / lo planes are zero)
__device__ __forceinline__ f32x16 mma3(const bf16x8 a0, const bf16x8 (&b)[3], f32x16 c) {
    c = mfma_bf16(a0, b[2], c);        // smallest terms first
    c = mfma_bf16(a0, b[1], c);
    c = mfma_bf16(a0, b[0], c);
    return c;
}

__device__ __forceinline__ f32x16 mma3b(const bf16x8 (&a)[3], const bf16x8 b0, f32x16 c) {
    c = mfma_bf16(a[2], b0, c);
    c = mfma_bf16(a[1], b0, c);
    c = mfma_bf16(a[0], b0, c);
    return c;
}

// conv_x6p.hip: software-pipelined fprop / dgrad with the fused BatchNorm hooks; SH_X6P_NO = shape not handled (fall back)
#define SH_X6P_NO (-100)
int sh_x6p_launch(int mode, ConvQ& p, hipStream_t st);
int sh_x6p_grouped_launch(ConvQ& p, hipStream_t st);
int sh_x6p_wgrad_launch(ConvQ& p, int wgm, int wgn, int splits, hipStream_t st);
// conv_b16.hip: bf16 COMPUTE mode (one MFMA product per tile, operands rounded once to bf16); SH_X6P_NO = shape not handled
int sh_b16_launch(int mode, ConvQ& p, int a32, hipStream_t st);
int sh_b16_grouped_launch(ConvQ& p, hipStream_t st);
int sh_b16_wgrad_launch(ConvQ& p, int dy32, int wgm, int wgn, int splits, hipStream_t st);
// conv_bf16x6.hip: sum of the split-K slabs + bias / addend / BN statistics / BN-backward front half (p.slab, p.ksplit set)
int sh_x6_splitk_reduce(const ConvQ& p, int mode, hipStream_t st);
