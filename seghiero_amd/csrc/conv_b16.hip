// bf16 COMPUTE mode of the convolution path (BASELINE configs[4]: "bf16 storage / f32 accumulate").
//
// The fp32-accurate kernels (conv_x6p.hip) split every fp32 operand into three bf16 terms and issue six MFMA products per tile.  Here
// an operand is rounded ONCE to bf16 (round to nearest even) on its way to LDS -- after the producer's BatchNorm + ReLU where the
// activation is deferred to the loader -- and a tile costs ONE v_mfma_f32_32x32x16_bf16 per K = 16: fp32 accumulators, BatchNorm
// statistics from the fp32 accumulators, fp32 weight gradients, fp32 master weights and SGD.  What the kernel moves is bf16:
// activations (raw conv outputs, block outputs) are stored as bf16, the weights come as a bf16 copy made once per step
// (sh_weights_to_bf16), gradients w.r.t. activations are fp32 or bf16 tensors (A32).
//
// With one product per tile most layers of the step are HBM-bound (AI of a 1x1 conv at 128^2: 50-130 flop/B against a bf16 ridge of
// 312), so the loop is built for bytes: K = 64 per tile, every 128-byte line of a gathered im2col row is fetched whole by 8 adjacent
// lanes of ONE 16-byte raw buffer load (the fp32 kernels take a 128-byte line in four phases), two tiles of loads in flight per block
// (two register sets), 64 KB of LDS (two buffers of 128 x 128-byte rows per operand) => 2 blocks per CU.
//
//     phase j :  store tile j+1 (registers -> LDS buffer (j+1)&1: BatchNorm + ReLU / lin(g, y) / plain copy, one ds_write_b128 per chunk)
//                issue the raw buffer loads of tile j+3 (-> the register set just freed)
//                ds_read_b128 fragments of tile j (buffer j&1), TM*TN*4 MFMAs
//                ONE barrier
//
// LDS image: rows of 128 bytes (64 bf16), the 16-byte chunk index XORed with (row >> 1) & 7: the 16 lanes of every ds_read_b128 lane
// group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) then cover the 16 slots of the 256-byte bank row once (rows of equal parity in a
// group have distinct (row >> 1) & 7), and a ds_write_b128 group of 8 lanes writes one whole row -- conflict-free both ways.
//
// The fused BatchNorm hooks are those of conv_x6p.hip (same ConvQ fields, same C-ABI semantics):
//   AFF 1 : A = relu(x * scale[c] + shift[c])  (fprop: the producer's train-mode BatchNorm + ReLU in the loader)
//   AFF 2 : A = lin(g, y) = A[c]*g + B[c]*(y - mean[c]) + D[c]  (1x1 dgrad: the deferred second half of the conv's own BatchNorm backward)
//   EPI 1 : fprop epilogue emits centred (sum, M2) BatchNorm partials per 64 rows from the fp32 accumulators
//   EPI 2 : dgrad epilogue = front half of the producer's BatchNorm backward (g = relumask * dx, (sum g, sum g * xhat) per 64 rows)
//   GRP   : grouped 1x1 fprop (the ASPP branches), SK : K slices into fp32 slabs + sh_x6_splitk_reduce
// Geometry: channel counts per tap (Kc) multiples of 8 and 16-byte aligned rows (else SH_X6P_NO: the caller falls back), K x K taps
// need Kc % 64 == 0 (every 3 x 3 conv of the model); stride-2 K x K dgrads (parity classes) and the strided 1x1 scatter stay on the
// fp32-accurate kernels.
#include "conv_x6.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {          // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ int swz8(int row) { return (row >> 1) & 7; }

// eight consecutive channels of one pixel: one 16-byte access of a bf16 tensor, two of an fp32 one (idx = element index, multiple of 8)
struct F8 { f32x4 lo, hi; };
__device__ __forceinline__ F8 f8_zero() { return F8{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ F8 ld8(const float* base, long long idx, int bf) {
    F8 r;
    if (bf) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(base) + idx);
        r.lo = f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
        r.hi = f32x4{__uint_as_float(v[2] << 16), __uint_as_float(v[2] & 0xffff0000u), __uint_as_float(v[3] << 16), __uint_as_float(v[3] & 0xffff0000u)};
    } else { r.lo = ld4(base + idx); r.hi = ld4(base + idx + 4); }
    return r;
}
__device__ __forceinline__ void st8(float* base, long long idx, const F8& v, int bf) {
    if (bf) *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(base) + idx) =
        u32x4{pk_bf16(v.lo[0], v.lo[1]), pk_bf16(v.lo[2], v.lo[3]), pk_bf16(v.hi[0], v.hi[1]), pk_bf16(v.hi[2], v.hi[3])};
    else { st4(base + idx, v.lo); st4(base + idx + 4, v.hi); }
}
__device__ __forceinline__ F8 quad_mask_load8(const void* base, long long byte_idx) {      // two ReLU quad-mask bytes -> 1.f / 0.f per channel
    const unsigned b = *reinterpret_cast<const unsigned short*>(reinterpret_cast<const unsigned char*>(base) + byte_idx);
    F8 r;
    r.lo = f32x4{(b & 1u) ? 1.f : 0.f, (b & 2u) ? 1.f : 0.f, (b & 4u) ? 1.f : 0.f, (b & 8u) ? 1.f : 0.f};
    r.hi = f32x4{(b & 0x100u) ? 1.f : 0.f, (b & 0x200u) ? 1.f : 0.f, (b & 0x400u) ? 1.f : 0.f, (b & 0x800u) ? 1.f : 0.f};
    return r;
}

// act bits used here (conv_x6.h): 1 fprop x bf16 (always set in this mode), 2 fprop y bf16, 4 bnb_y bf16, 8 bnb_out bf16, 16 lin y bf16,
// 64 bnb_out = ReLU quad mask, 128 dgrad output (dx / g) bf16, 256 dgrad addend bf16
// BatchNorm + ReLU of two bf16 values held in one 32-bit word (the loaders' transform): widen, ONE packed fused multiply-add, round both
// to bf16, ReLU as a packed 16-bit max on the rounded pair -- 6 vector instructions per pair instead of 8 (the loaders run ~8 VALU
// instructions per MFMA in these kernels, as many issue cycles as the MFMA itself takes).  Rounding and max(., floor) commute (both are
// monotonic, floor is 0 or "none"); the 16-bit max only has to order a value against +0 / against the most negative pattern, which the
// sign bit decides for every bf16 pattern of magnitude < 2^121 (beyond that the fp16 reading is a NaN).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned aff_pair(unsigned raw, f32x2_t sc, f32x2_t sh, unsigned floor_bits) {
    const f32x2_t x = {__uint_as_float(raw << 16), __uint_as_float(raw & 0xffff0000u)};
    const f32x2_t r = __builtin_elementwise_fma(x, sc, sh);
    const unsigned pk = pk_bf16(r[0], r[1]);
    const h16x2_t m = __builtin_elementwise_max(__builtin_bit_cast(h16x2_t, pk), __builtin_bit_cast(h16x2_t, floor_bits));
    return __builtin_bit_cast(unsigned, m);
}
#define B16_OUT_BF 128
#define B16_ADD_BF 256

// W8: the 128 x 128 tile on EIGHT waves (2 x 4, a wave owns 64 x 32) instead of four (2 x 2, 64 x 64): half the accumulators and staging
// registers per wave (<= 128 VGPRs), so two blocks = 16 waves per CU, four per SIMD.  Measured on the 512 -> 512 pointwise conv at 128^2
// (PMC, 4 waves): the matrix pipe busy 25 %, each wave spends 1,680 VALU instructions on its prologue / epilogue against 128 MFMAs, and
// with two waves per SIMD nothing covers a wave that sits in its epilogue or waits for its first loads.
// RM (input gradients of strided convs, plain epilogue): 1 = stride-2 KxK by input-parity class (blockIdx.y = class: an input pixel only sees
// the taps of its parity, conv_x6p_kernel's plan), 2 = 1x1 strided conv: the GEMM runs on the Ho x Wo grid and its rows are ADDED to the
// rows of dx at the strided pixels (the downsample branch's gradient joins the block input's gradient)
template <int MODE, int TN, int AFF, int EPI, int SK, int TAP, int GRP, int A32, int W8 = 0, int RM = 0>
__global__ __launch_bounds__(W8 ? 512 : 256, W8 ? 4 : 2) void conv_b16_kernel(const ConvQ p) {
    constexpr int TM = 2, WGN = W8 ? 4 : 2, TNW = W8 ? 1 : TN, NT = 128 * WGN, BM = 128, BN = 32 * TNW * WGN;
    static_assert(!W8 || TN == 2, "the eight-wave form is the 128 x 128 tile");
    static_assert(!RM || (MODE == DGRAD && AFF == 0 && EPI == 0 && !SK && !GRP && !W8 && TAP == (RM == 1 ? 1 : 0)), "row maps: plain strided input gradients");
    constexpr int A_BUF = BM * 128, B_BUF = BN * 128;                // bytes per LDS buffer
    constexpr int RPP = NT / 8, NA = BM / RPP, NB = BN / RPP;        // loader: 8 lanes x 16 bytes per row, NT / 8 rows per pass
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const As = smem;
    unsigned char* const Bs = smem + 2 * A_BUF;
    float* const coef = reinterpret_cast<float*>(smem + 2 * (A_BUF + B_BUF));      // AFF 1: scale[Kc], shift[Kc]; AFF 2: lin[4][Kc]
    static_assert(AFF != 2 || (MODE == DGRAD && TAP == 0 && !GRP), "the deferred BatchNorm-backward loader exists for 1x1 dgrads");
    static_assert(!A32 || (MODE == DGRAD && AFF == 0), "fp32 A streams are plain gradients (the lin loader takes a bf16 g)");

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WGN, wn = wave % WGN, l31 = lane & 31, h = lane >> 5;
    const unsigned nblk = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
    const unsigned bid = xcd_remap(blockIdx.x, nblk);
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int grp = GRP ? n0 / p.group_n : 0;
    const float* const a_ptr = GRP ? p.ga[grp] : p.a;
    const float* const b_ptr = GRP ? p.gb[grp] : p.b;
    const unsigned a_bytes = GRP ? p.ga_bytes[grp] : p.a_bytes;
    const int lda = (int)(GRP ? p.glda[grp] : p.lda);
    const float* const sc_ptr = GRP ? p.gsc[grp] : p.aff_scale;
    const float* const sh_ptr = GRP ? p.gsh[grp] : p.aff_shift;
    const float aff_floor = (GRP && sc_ptr == nullptr) ? -INFINITY : 0.f;
    int Mc = p.M, Kt = p.K;
    [[maybe_unused]] int cy = 0, cx = 0, oy0 = 0, ox0 = 0, Hc = p.H, Wc = p.W, ntw = p.KW;
    if constexpr (RM == 1) {
        cy = blockIdx.y >> 1; cx = blockIdx.y & 1;
        oy0 = (cy + p.pad) & 1; ox0 = (cx + p.pad) & 1;
        Hc = (p.H - oy0 + 1) >> 1; Wc = (p.W - ox0 + 1) >> 1;
        ntw = (p.KW - cx + 1) >> 1;
        Mc = p.N * Hc * Wc; Kt = ((p.KH - cy + 1) >> 1) * ntw * p.Kc;
        if (m0 >= Mc) return;                                    // block-uniform, before any barrier
    }
    int q_begin = 0, q_end = (Kt + 63) >> 6;                         // tiles of K = 64
    if constexpr (SK) {
        const int per = (q_end + p.ksplit - 1) / p.ksplit;
        q_begin = min(q_end, (int)blockIdx.y * per); q_end = min(q_end, q_begin + per);
    }
    const int klim = min(Kt, 64 * q_end);
    const int nq = q_end - q_begin;
    constexpr bool single_tap = TAP == 0;

    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_ptr), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b_ptr), 0, p.b_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc_a2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(AFF == 2 ? p.a2 : a_ptr), 0, AFF == 2 ? p.a2_bytes : a_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int kc = t & 7, r0 = t >> 3;
    int a_y[NA], a_x[NA], a_nb[NA];
    int b_row[NB];
    unsigned b_ok[NB];
    int cur_tap = -1;
    [[maybe_unused]] int cur_wtap = 0;                           // RM == 1: the tap's index in the weight tensor
    // 1x1, stride 1, no padding (every pointwise conv of the model): pixel m of the output is row m of the operand -- no coordinate
    // arithmetic (eight integer divisions per thread otherwise: ~300 VALU instructions of a kernel that issues 128 MFMAs per wave)
    const bool direct = TAP == 0 && p.stride == 1 && p.pad == 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + r0 + RPP * i;
        if (direct) { a_y[i] = m < Mc ? 0 : -(1 << 28); a_x[i] = 0; a_nb[i] = m < Mc ? m : 0; }
        else if (m < Mc) {
            if constexpr (MODE == FPROP) {
                const int ow = m % p.Wo, q = m / p.Wo, oh = q % p.Ho, n = q / p.Ho;
                a_y[i] = oh * p.stride - p.pad; a_x[i] = ow * p.stride - p.pad; a_nb[i] = n * p.H * p.W;
            } else if constexpr (RM == 1) {
                const int iwc = m % Wc, q = m / Wc, ihc = q % Hc, n = q / Hc;
                a_y[i] = 2 * ihc + oy0 + p.pad; a_x[i] = 2 * iwc + ox0 + p.pad; a_nb[i] = n * p.Ho * p.Wo;
            } else {
                const int iw = m % p.W, q = m / p.W, ih = q % p.H, n = q / p.H;
                a_y[i] = ih + p.pad; a_x[i] = iw + p.pad; a_nb[i] = n * p.Ho * p.Wo;
            }
        } else { a_y[i] = -(1 << 28); a_x[i] = 0; a_nb[i] = 0; }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int j = n0 + r0 + RPP * i;
        b_ok[i] = j < p.Nn ? ~0u : 0u;
        b_row[i] = (GRP ? j - grp * p.group_n : j) * (MODE == FPROP ? p.K : p.Kc);
    }
    int a_off[NA];
    [[maybe_unused]] int a_off2[NA];
    unsigned a_ok[NA];
    auto set_tap = [&](int tap) {
        if (direct) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                a_ok[i] = a_y[i] == 0 ? ~0u : 0u;
                a_off[i] = a_nb[i] * lda;
                if constexpr (AFF == 2) a_off2[i] = a_nb[i] * (int)p.lda2;
            }
            return;
        }
        int kh = tap / p.KW, kw = tap - kh * p.KW;
        if constexpr (RM == 1) { const int ty = tap / ntw; kh = cy + 2 * ty; kw = cx + 2 * (tap - ty * ntw); cur_wtap = kh * p.KW + kw; }
        const int dh = kh * p.dil, dw = kw * p.dil;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if constexpr (RM == 1) {                             // a_y - kh is even by construction of the class (dil == 1, host check)
                const int t2h = a_y[i] - dh, t2w = a_x[i] - dw, th = t2h >> 1, tw = t2w >> 1;
                a_ok[i] = (t2h >= 0 && t2w >= 0 && th < p.Ho && tw < p.Wo) ? ~0u : 0u;
                a_off[i] = (a_nb[i] + th * p.Wo + tw) * lda;
            } else
            if constexpr (MODE == FPROP) {
                const int ih = a_y[i] + dh, iw = a_x[i] + dw;
                a_ok[i] = ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? ~0u : 0u;
                a_off[i] = (a_nb[i] + ih * p.W + iw) * lda;
            } else {                                             // stride 1 (host check)
                const int th = a_y[i] - dh, tw = a_x[i] - dw;
                a_ok[i] = (th >= 0 && tw >= 0 && th < p.Ho && tw < p.Wo) ? ~0u : 0u;
                a_off[i] = (a_nb[i] + th * p.Wo + tw) * lda;
                if constexpr (AFF == 2) a_off2[i] = (a_nb[i] + th * p.Wo + tw) * (int)p.lda2;
            }
        }
    };
    auto prepare_tap = [&](int q) {
        if constexpr (TAP == 1) {
            const int tap = (64 * q) / p.Kc;                     // Kc % 64 == 0: a tile never straddles taps
            if (tap != cur_tap) { set_tap(tap); cur_tap = tap; }
        }
    };
    if constexpr (TAP == 0) { set_tap(0); cur_tap = 0; }
    struct Regs { u32x4 a[NA][A32 ? 2 : 1]; u32x4 a2[AFF == 2 ? NA : 1]; u32x4 b[NB]; int cc; unsigned okm; };
    auto load_tile = [&](int q, Regs& R) {
        const int k = 64 * q + 8 * kc;
        int tap, cc;
        if constexpr (single_tap) { tap = 0; cc = k; }
        else { tap = cur_tap; cc = k - tap * p.Kc; }
        const unsigned kok = (unsigned)((k - klim) >> 31);
        R.cc = cc; R.okm = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned ok = kok & a_ok[i];
            const unsigned e = (unsigned)(a_off[i] + cc);
            if constexpr (A32) {
                const unsigned voff = ((e * 4u) & ok) | (OOB & ~ok);
                R.a[i][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff, 0, 0));
                R.a[i][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff + 16u, 0, 0));
            } else {
                const unsigned voff = ((e * 2u) & ok) | (OOB & ~ok);
                R.a[i][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff, 0, 0));
            }
            if constexpr (AFF == 2) {
                const unsigned v2 = (((unsigned)(a_off2[i] + cc) * 2u) & ok) | (OOB & ~ok);
                R.a2[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a2, v2, 0, 0));
            }
            if (AFF) R.okm |= ok & (1u << i);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int off;
            if constexpr (MODE == FPROP) off = b_row[i] + k;
            else off = (RM == 1 ? cur_wtap : tap) * p.Cin * p.Kc + b_row[i] + cc;
            const unsigned okb = kok & b_ok[i];
            const unsigned voff = (((unsigned)off * 2u) & okb) | (OOB & ~okb);
            R.b[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, voff, 0, 0));
        }
    };
    const int swz_w = swz8(r0);                                  // rows r0 + 32 i share (row >> 1) & 7
    const int st_off = r0 * 128 + ((kc ^ swz_w) << 4);
    auto store_tile = [&](Regs& R, int buf) {
        unsigned char* const ad = As + buf * A_BUF + st_off;
        if constexpr (AFF == 0 && !A32) {
#pragma unroll
            for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(ad + RPP * i * 128) = R.a[i][0];
        } else {
            float c0[8], c1[8];
            [[maybe_unused]] float c2[8], c3[8];
            if constexpr (AFF) {
                *reinterpret_cast<f32x4*>(c0) = *reinterpret_cast<const f32x4*>(coef + R.cc);
                *reinterpret_cast<f32x4*>(c0 + 4) = *reinterpret_cast<const f32x4*>(coef + R.cc + 4);
                *reinterpret_cast<f32x4*>(c1) = *reinterpret_cast<const f32x4*>(coef + p.Kc + R.cc);
                *reinterpret_cast<f32x4*>(c1 + 4) = *reinterpret_cast<const f32x4*>(coef + p.Kc + R.cc + 4);
            }
            if constexpr (AFF == 2) {
                *reinterpret_cast<f32x4*>(c2) = *reinterpret_cast<const f32x4*>(coef + 2 * p.Kc + R.cc);
                *reinterpret_cast<f32x4*>(c2 + 4) = *reinterpret_cast<const f32x4*>(coef + 2 * p.Kc + R.cc + 4);
                *reinterpret_cast<f32x4*>(c3) = *reinterpret_cast<const f32x4*>(coef + 3 * p.Kc + R.cc);
                *reinterpret_cast<f32x4*>(c3 + 4) = *reinterpret_cast<const f32x4*>(coef + 3 * p.Kc + R.cc + 4);
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                float v[8];
                if constexpr (A32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(R.a[i][0][e]); v[4 + e] = __uint_as_float(R.a[i][A32 ? 1 : 0][e]); }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(R.a[i][0][e] << 16); v[2 * e + 1] = __uint_as_float(R.a[i][0][e] & 0xffff0000u); }
                }
                if constexpr (AFF == 2) {
                    // dy = A*g + B*(y - mean) + D (c0 = A, c1 = B, c2 = mean, c3 = D): rows beyond M / the K tail must stay exact zeros
                    const bool ok = (R.okm >> i) & 1u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y0 = __uint_as_float(R.a2[i][e] << 16), y1 = __uint_as_float(R.a2[i][e] & 0xffff0000u);
                        const float w0 = fmaf(y0 - c2[2 * e], c1[2 * e], fmaf(v[2 * e], c0[2 * e], c3[2 * e]));
                        const float w1 = fmaf(y1 - c2[2 * e + 1], c1[2 * e + 1], fmaf(v[2 * e + 1], c0[2 * e + 1], c3[2 * e + 1]));
                        v[2 * e] = ok ? w0 : 0.f; v[2 * e + 1] = ok ? w1 : 0.f;
                    }
                } else if constexpr (AFF) {
                    // one fused multiply-add per element (the value is rounded to bf16 next; the fp32-accurate kernels keep sh_bn_act's
                    // unfused order for bit-equality with the separate pass); rows beyond M / padded taps are zeroed AFTER the pack
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], c0[e], c1[e]), aff_floor);
                }
                u32x4 o;
                if constexpr (AFF == 1 && !A32) {      // (the unpacked v[] above is dead code in this case)
                    const unsigned okw = ((R.okm >> i) & 1u) ? ~0u : 0u, fl = aff_floor == 0.f ? 0u : 0xfc00fc00u;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[e] = aff_pair(R.a[i][0][e], f32x2_t{c0[2 * e], c0[2 * e + 1]}, f32x2_t{c1[2 * e], c1[2 * e + 1]}, fl) & okw;
                } else {
                o = u32x4{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
                if constexpr (AFF == 1) { const unsigned okw = ((R.okm >> i) & 1u) ? ~0u : 0u; o[0] &= okw; o[1] &= okw; o[2] &= okw; o[3] &= okw; }
                }
                *reinterpret_cast<u32x4*>(ad + RPP * i * 128) = o;
            }
        }
        unsigned char* const bd = Bs + buf * B_BUF + st_off;
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<u32x4*>(bd + RPP * i * 128) = R.b[i];
    };

    f32x16 acc[TM][TNW];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int arow = wm * 64 + l31, brow = wn * 32 * TNW + l31;
    const int swz_r = swz8(l31);                                 // rows l31 + 32 k share (row >> 1) & 7
    auto compute_tile = [&](int buf) {
        const unsigned char* const ab = As + buf * A_BUF + arow * 128;
        const unsigned char* const bb = Bs + buf * B_BUF + brow * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int rd = ((2 * ks + h) ^ swz_r) << 4;
            bf16x8 bfr[TNW];
#pragma unroll
            for (int j = 0; j < TNW; ++j) bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bb + 32 * j * 128 + rd));
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ab + 32 * i * 128 + rd));
#pragma unroll
                for (int j = 0; j < TNW; ++j) acc[i][j] = mfma_bf16(af, bfr[j], acc[i][j]);
            }
        }
    };

    // ------------------------------------------------------------------ prologue
    if constexpr (AFF == 2) {
        for (int c = t; c < 4 * p.Kc; c += NT) coef[c] = p.lin[c];
        __syncthreads();
    } else if constexpr (AFF) {
        for (int c = t; c < p.Kc; c += NT) { coef[c] = sc_ptr ? sc_ptr[c] : 1.f; coef[p.Kc + c] = sh_ptr ? sh_ptr[c] : 0.f; }
        __syncthreads();
    }
    Regs R0, R1;
    prepare_tap(q_begin);     load_tile(q_begin, R0);
    prepare_tap(q_begin + 1); load_tile(q_begin + 1, R1);
    store_tile(R0, 0);
    prepare_tap(q_begin + 2); load_tile(q_begin + 2, R0);
    __syncthreads();
    // ------------------------------------------------------------------ main loop: two tiles per iteration (tiles beyond the K range load zeros)
    for (int j = 0; j < nq; j += 2) {
        prepare_tap(q_begin + j + 3);
        store_tile(R1, 1);
        load_tile(q_begin + j + 3, R1);
        compute_tile(0);
        __syncthreads();
        prepare_tap(q_begin + j + 4);
        store_tile(R0, 0);
        load_tile(q_begin + j + 4, R0);
        if (j + 1 < nq) compute_tile(1);                         // block-uniform: an odd tile count skips the zero tile
        __syncthreads();
    }

    if constexpr (SK) {
        float* slab = p.slab + (long long)blockIdx.y * p.M * p.ldslab;
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int n = n0 + wn * 32 * TNW + 32 * j + l31;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < p.M && n < p.Nn) slab[(long long)m * p.ldslab + n] = acc[i][j][r];
                }
        }
        return;
    }

    // ------------------------------------------------------------------ epilogue: row-major through LDS, 8 columns (16 bytes of bf16) per lane
    // An accumulator register holds one column per lane; each wave parks 32 rows of its 64-row tile in ITS OWN LDS strip and reads them
    // back with a lane owning 8 consecutive columns of a row, so that every global access of a bf16 tensor (output, addend, the
    // BatchNorm-backward y tile) is one 16-byte access -- 128 contiguous bytes per 8 lanes -- and an fp32 tensor takes two.  The strip is
    // private to the wave (LDS operations of one wave complete in order), the tile buffers it overlays were released by the main loop's
    // last barrier: no block-wide barrier in here, the four waves drain their tiles independently.
    {
        constexpr int WC = 32 * TNW, RS = WC + 4, LPR = WC / 8, RPI = 64 / LPR, NRD = 32 / RPI;
        float* const stage = reinterpret_cast<float*>(smem) + wave * (32 * RS);
        const int rr = lane / LPR, c8 = (lane % LPR) * 8;
        const int ncv = n0 + wn * WC + c8;
        const bool nokv = ncv < p.Nn;                                 // Nn % 8 == 0 (host check)
        F8 vb = f8_zero(), v_mu = f8_zero(), v_is = f8_zero(), v_sc = f8_zero(), v_sh = f8_zero();
        if (MODE == FPROP && p.extra != nullptr && nokv) vb = ld8(p.extra, ncv, 0);
        if constexpr (EPI == 2) { if (nokv) { v_mu = ld8(p.bnb_mean, ncv, 0); v_is = ld8(p.bnb_invstd, ncv, 0); v_sc = ld8(p.bnb_scale, ncv, 0); v_sh = ld8(p.bnb_shift, ncv, 0); } }
        [[maybe_unused]] F8 pend_gs = f8_zero(), pend_gq = f8_zero();
        const int out_bf = MODE == FPROP ? (p.act & 2) : (p.act & B16_OUT_BF);
        // GEMM row -> row of the output tensor (and of the addend: with RM == 2 that is the output itself)
        auto orow = [&](long long m) -> long long {
            if constexpr (RM == 1) {
                const int mi = (int)m, iwc = mi % Wc, q = mi / Wc, ihc = q % Hc, nb = q / Hc;
                return ((long long)nb * p.H + 2 * ihc + oy0) * p.W + 2 * iwc + ox0;
            } else if constexpr (RM == 2) {
                const int mi = (int)m, ow = mi % p.W, q = mi / p.W, oh = q % p.H, nb = q / p.H;
                return ((long long)nb * p.sH + oh * p.sstride) * p.sW + ow * p.sstride;
            } else return m;
        };
        // Input-gradient epilogues whose tensors are all bf16 (the bf16-gradient step): every load of the wave's 64 x 64 tile -- addend, the
        // BatchNorm-backward y tile, the ReLU quad mask -- is issued up front into the registers the staging sets no longer need (16-byte
        // raw words, widened when used): ONE memory round trip per wave instead of one per two row groups (measured on the 1024 -> 256
        // layer-3 gradient with residual mask and identity addend: 80 us for a 25 us byte count, the four dependent trips of the chunked form)
        // (the eight-wave form of the input gradient is launched for all-bf16 epilogues only -- sh_b16_launch checks -- and compiles just
        // this path: the generic one needs more than its 128 registers)
        constexpr bool FAST_ONLY = W8 && MODE == DGRAD;
        bool fast = FAST_ONLY;
        if constexpr (MODE == DGRAD && !FAST_ONLY) {
            fast = (p.act & B16_OUT_BF) && (p.extra == nullptr || (p.act & B16_ADD_BF)) &&
                   (EPI != 2 || ((p.act & 4) && (p.bnb_out == nullptr || (p.act & 64))));
        }
        if (fast) {
            if constexpr (MODE == DGRAD) {
                constexpr int NL = TM * NRD;
                u32x4 rad[NL];
                [[maybe_unused]] u32x4 ryv[EPI == 2 ? NL : 1];
                [[maybe_unused]] unsigned rmk[EPI == 2 ? NL : 1];
                const bool has_add = p.extra != nullptr, has_mask = EPI == 2 && p.bnb_out != nullptr;
#pragma unroll
                for (int q = 0; q < NL; ++q) {
                    const long long m = m0 + wm * 64 + 32 * (q / NRD) + (q % NRD) * RPI + rr;
                    const bool ok = m < Mc && nokv;
                    rad[q] = u32x4{0u, 0u, 0u, 0u};
                    if (ok && has_add) rad[q] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(p.extra) + (RM ? orow(m) : m) * p.ldadd + ncv);
                    if constexpr (EPI == 2) {
                        ryv[q] = u32x4{0u, 0u, 0u, 0u}; rmk[q] = 0xffffu;
                        if (ok) {
                            ryv[q] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(p.bnb_y) + m * p.bnb_ldy + ncv);
                            if (has_mask) rmk[q] = *reinterpret_cast<const unsigned short*>(reinterpret_cast<const unsigned char*>(p.bnb_out) + m * p.bnb_ldo + (ncv >> 2));
                        }
                    }
                }
                auto widen = [](const u32x4& v) {
                    F8 r;
                    r.lo = f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
                    r.hi = f32x4{__uint_as_float(v[2] << 16), __uint_as_float(v[2] & 0xffff0000u), __uint_as_float(v[3] << 16), __uint_as_float(v[3] & 0xffff0000u)};
                    return r;
                };
                [[maybe_unused]] F8 pgs = f8_zero(), pgq = f8_zero();
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int j = 0; j < TNW; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * RS + 32 * j + l31] = acc[i][j][r];
                    [[maybe_unused]] F8 gs = f8_zero(), gq = f8_zero();
#pragma unroll
                    for (int k = 0; k < NRD; ++k) {
                        const int q = i * NRD + k, row = k * RPI + rr;
                        const long long m = m0 + wm * 64 + 32 * i + row;
                        F8 o;
                        o.lo = *reinterpret_cast<const f32x4*>(stage + row * RS + c8);
                        o.hi = *reinterpret_cast<const f32x4*>(stage + row * RS + c8 + 4);
                        if (m < Mc && nokv) {
                            const F8 a8 = widen(rad[q]);
                            o.lo += a8.lo; o.hi += a8.hi;
                            if constexpr (EPI == 2) {
                                const F8 y8 = widen(ryv[q]);
                                if (p.bnb_relu) {
                                    if (has_mask) {
#pragma unroll
                                        for (int e = 0; e < 4; ++e) { if (!((rmk[q] >> e) & 1u)) o.lo[e] = 0.f; if (!((rmk[q] >> (8 + e)) & 1u)) o.hi[e] = 0.f; }
                                    } else {
                                        const f32x4 al = y8.lo * v_sc.lo + v_sh.lo, ah = y8.hi * v_sc.hi + v_sh.hi;
#pragma unroll
                                        for (int e = 0; e < 4; ++e) { if (!(al[e] > 0.f)) o.lo[e] = 0.f; if (!(ah[e] > 0.f)) o.hi[e] = 0.f; }
                                    }
                                }
                                gs.lo += o.lo; gs.hi += o.hi;
                                gq.lo += o.lo * ((y8.lo - v_mu.lo) * v_is.lo); gq.hi += o.hi * ((y8.hi - v_mu.hi) * v_is.hi);
                            }
                            st8(p.c, (RM ? orow(m) : m) * p.ldc + ncv, o, 1);
                        }
                    }
                    if constexpr (EPI == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
#pragma unroll
                            for (int o2 = LPR; o2 < 64; o2 <<= 1) {
                                gs.lo[e] += __shfl_xor(gs.lo[e], o2, 64); gs.hi[e] += __shfl_xor(gs.hi[e], o2, 64);
                                gq.lo[e] += __shfl_xor(gq.lo[e], o2, 64); gq.hi[e] += __shfl_xor(gq.hi[e], o2, 64);
                            }
                        }
                        if ((i & 1) == 0) { pgs = gs; pgq = gq; }
                        else {
                            const int pidx = tile_m * (BM / 64) + wm;
                            if (lane < LPR && pidx < p.n_partials && nokv) {
                                float* const ps = p.partials + ((long long)pidx * 2 + 0) * p.Nn + ncv;
                                float* const pq = p.partials + ((long long)pidx * 2 + 1) * p.Nn + ncv;
                                st4(ps, pgs.lo + gs.lo); st4(ps + 4, pgs.hi + gs.hi);
                                st4(pq, pgq.lo + gq.lo); st4(pq + 4, pgq.hi + gq.hi);
                            }
                        }
                    }
                }
            }
        } else if constexpr (!FAST_ONLY)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * RS + 32 * j + l31] = acc[i][j][r];
            constexpr int CH = NRD > 2 ? 2 : NRD;
            [[maybe_unused]] F8 gs = f8_zero(), gq = f8_zero();
#pragma unroll
            for (int k0 = 0; k0 < NRD; k0 += CH) {
                F8 v[CH];
                [[maybe_unused]] F8 ad[CH], yv[CH], ov[CH];
                long long mrow[CH];
#pragma unroll
                for (int kk = 0; kk < CH; ++kk) {
                    const int row = (k0 + kk) * RPI + rr;
                    mrow[kk] = m0 + wm * 64 + 32 * i + row;
                    v[kk].lo = *reinterpret_cast<const f32x4*>(stage + row * RS + c8);
                    v[kk].hi = *reinterpret_cast<const f32x4*>(stage + row * RS + c8 + 4);
                    if constexpr (MODE == DGRAD) {
                        ad[kk] = f8_zero();
                        if (mrow[kk] < Mc && nokv) {
                            if (p.extra != nullptr) ad[kk] = ld8(p.extra, (RM ? orow(mrow[kk]) : mrow[kk]) * p.ldadd + ncv, p.act & B16_ADD_BF);
                            if constexpr (EPI == 2) {
                                yv[kk] = ld8(p.bnb_y, mrow[kk] * p.bnb_ldy + ncv, p.act & 4);
                                if (p.bnb_out != nullptr)
                                    ov[kk] = (p.act & 64) ? quad_mask_load8(p.bnb_out, mrow[kk] * p.bnb_ldo + (ncv >> 2)) : ld8(p.bnb_out, mrow[kk] * p.bnb_ldo + ncv, p.act & 8);
                            }
                        }
                    }
                }
#pragma unroll
                for (int kk = 0; kk < CH; ++kk) {
                    if (mrow[kk] < Mc && nokv) {
                        F8 o = v[kk];
                        if constexpr (MODE == FPROP) { o.lo += vb.lo; o.hi += vb.hi; }
                        else {
                            o.lo += ad[kk].lo; o.hi += ad[kk].hi;
                            if constexpr (EPI == 2) {
                                if (p.bnb_relu) {
                                    F8 a;
                                    if (p.bnb_out != nullptr) a = ov[kk];
                                    else { a.lo = yv[kk].lo * v_sc.lo + v_sh.lo; a.hi = yv[kk].hi * v_sc.hi + v_sh.hi; }
#pragma unroll
                                    for (int e = 0; e < 4; ++e) { if (!(a.lo[e] > 0.f)) o.lo[e] = 0.f; if (!(a.hi[e] > 0.f)) o.hi[e] = 0.f; }
                                }
                                gs.lo += o.lo; gs.hi += o.hi;
                                gq.lo += o.lo * ((yv[kk].lo - v_mu.lo) * v_is.lo); gq.hi += o.hi * ((yv[kk].hi - v_mu.hi) * v_is.hi);
                            }
                        }
                        st8(p.c, (RM ? orow(mrow[kk]) : mrow[kk]) * p.ldc + ncv, o, out_bf);
                    }
                }
            }
            if constexpr (EPI == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int o2 = LPR; o2 < 64; o2 <<= 1) {
                        gs.lo[e] += __shfl_xor(gs.lo[e], o2, 64); gs.hi[e] += __shfl_xor(gs.hi[e], o2, 64);
                        gq.lo[e] += __shfl_xor(gq.lo[e], o2, 64); gq.hi[e] += __shfl_xor(gq.hi[e], o2, 64);
                    }
                }
                if ((i & 1) == 0) { pend_gs = gs; pend_gq = gq; }
                else {
                    const int pidx = tile_m * (BM / 64) + wm;
                    if (lane < LPR && pidx < p.n_partials && nokv) {
                        float* const ps = p.partials + ((long long)pidx * 2 + 0) * p.Nn + ncv;
                        float* const pq = p.partials + ((long long)pidx * 2 + 1) * p.Nn + ncv;
                        st4(ps, pend_gs.lo + gs.lo); st4(ps + 4, pend_gs.hi + gs.hi);
                        st4(pq, pend_gq.lo + gq.lo); st4(pq + 4, pend_gq.hi + gq.hi);
                    }
                }
            }
        }
    }
    if constexpr (MODE == FPROP && EPI == 1) {
        if (p.partials != nullptr) {                                // centred (sum, M2) of the wave's 64 rows, from the fp32 accumulators
            const int wrow0 = m0 + wm * 64;
            const int npr = max(0, min(64, p.M - wrow0));
            const int pidx = tile_m * (BM / 64) + wm;
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                float ss = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ss += acc[i][j][r];
                ss += __shfl_xor(ss, 32, 64);
                const float mean = npr > 0 ? ss / (float)npr : 0.f;
                float qq = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wrow0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float dv = acc[i][j][r] - mean;
                        qq += (npr == 64 || row < p.M) ? dv * dv : 0.f;
                    }
                qq += __shfl_xor(qq, 32, 64);
                const int n = n0 + wn * 32 * TNW + 32 * j + l31;
                if (h == 0 && pidx < p.n_partials && n < p.Nn) {
                    p.partials[((long long)pidx * 2 + 0) * p.Nn + n] = ss;
                    p.partials[((long long)pidx * 2 + 1) * p.Nn + n] = qq;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- host side
static int b16_w8() { static int v = -2; if (v == -2) { const char* e = getenv("SEGHIERO_B16_W8"); v = e ? atoi(e) : 1; } return v; }
template <int MODE, int TN, int AFF, int EPI, int SK, int TAP, int GRP, int A32, int W8 = 0, int RM = 0>
static int launch_b16(ConvQ& p, hipStream_t st) {
    // (the lin loader needs more than the 128 registers of the eight-wave form; the input gradient's eight-wave form has the all-bf16
    // epilogue only)
    if constexpr (TN == 2 && !W8 && !A32 && AFF != 2 && !RM) {
        bool ok8 = b16_w8() != 0;
        // input gradients: opt-in (SEGHIERO_B16_W8=2).  Measured on the configs[4] step, same box, alternating rounds of 20 steps: the
        // four-wave form 27.16 / 27.06 / 27.39 / 27.13 ms, the eight-wave form 27.64 / 35.07 / 29.03 / 26.80 ms -- sixteen 128-register waves
        // per CU leave the concurrent weight-gradient stream nothing, and the step time becomes erratic
        if constexpr (MODE == DGRAD)
            ok8 = b16_w8() == 2 && (p.act & B16_OUT_BF) && (p.extra == nullptr || (p.act & B16_ADD_BF)) &&
                  (EPI != 2 || ((p.act & 4) && (p.bnb_out == nullptr || (p.act & 64))));
        if (ok8) return launch_b16<MODE, TN, AFF, EPI, SK, TAP, GRP, A32, 1>(p, st);
    }
    constexpr int BM = 128, BN = 64 * TN;
    const size_t lds = 2 * (size_t)(BM + BN) * 128 + (AFF == 2 ? 16 : AFF ? 8 : 0) * (size_t)p.Kc;
    // 64 KB of tiles + the coefficient table: two blocks per CU up to Kc = 1024 (lin loader) / 2048 (BatchNorm + ReLU loader); beyond
    // that one (layer4's 2048-channel conv3 gradient: 128 tiles, less than one block per CU anyway)
    if (lds > 160 * 1024) return SH_X6P_NO;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_b16_kernel<MODE, TN, AFF, EPI, SK, TAP, GRP, A32, W8, RM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        attr_done = true;
    }
    p.tiles_m = (int)sh_cdiv(p.M, BM);                      // (RM == 1: p.M is the largest parity class)
    p.tiles_n = (int)sh_cdiv(p.Nn, BN);
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), RM == 1 ? 4u : (SK ? (unsigned)p.ksplit : 1u));
    conv_b16_kernel<MODE, TN, AFF, EPI, SK, TAP, GRP, A32, W8, RM><<<grid, W8 ? 512 : 256, lds, st>>>(p);
    return sh_launch_status();
}
template <int MODE, int AFF, int EPI, int A32>
static int pick_b16(ConvQ& p, hipStream_t st) {
    const bool one = p.KH * p.KW == 1;
    if (!one && (p.Kc & 63)) return SH_X6P_NO;
    if constexpr (AFF == 2) { if (!one) return SH_X6P_NO; }
    if (p.ksplit > 1) {
        int rc;
        if (one) rc = launch_b16<MODE, 2, AFF, 0, 1, 0, 0, A32>(p, st);
        else if constexpr (AFF != 2) rc = launch_b16<MODE, 2, AFF, 0, 1, 1, 0, A32>(p, st);
        else rc = SH_X6P_NO;
        return rc == SH_OK ? sh_x6_splitk_reduce(p, MODE, st) : rc;
    }
    if (one) return p.Nn > 64 ? launch_b16<MODE, 2, AFF, EPI, 0, 0, 0, A32>(p, st) : launch_b16<MODE, 1, AFF, EPI, 0, 0, 0, A32>(p, st);
    if constexpr (AFF != 2) return p.Nn > 64 ? launch_b16<MODE, 2, AFF, EPI, 0, 1, 0, A32>(p, st) : launch_b16<MODE, 1, AFF, EPI, 0, 1, 0, A32>(p, st);
    return SH_X6P_NO;
}

// Entry used by sh_conv_fprop_b16 / sh_conv_dgrad_b16 (conv_bf16x6.hip fills the ConvQ exactly as for the fp32-accurate kernels).
// a32: the dgrad's gradient stream (dy, or g of lin(g, y)) is an fp32 tensor.  SH_X6P_NO = no instantiation for this geometry.
int sh_b16_launch(int mode, ConvQ& p, int a32, hipStream_t st) {
    const bool aff = p.aff_scale != nullptr, bnb = p.bnb_y != nullptr, lin = p.lin != nullptr;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if ((p.Kc & 7) || (p.lda & 7) || !al16(p.a) || !al16(p.b)) return SH_X6P_NO;
    const int rm = p.parity ? 1 : p.scatter ? 2 : 0;
    if (rm && (mode != DGRAD || aff || bnb || lin || p.ksplit > 1)) return SH_X6P_NO;
    if (!rm && p.stride != 1 && mode == DGRAD) return SH_X6P_NO;
    if (a32 && ((p.lda & 3) != 0)) return SH_X6P_NO;
    const int obf = mode == FPROP ? (p.act & 2) : (p.act & B16_OUT_BF);
    // the epilogue moves 8 columns per lane: 16-byte rows of bf16 tensors, 2 x 16 bytes of fp32 ones
    if ((p.Nn & 7) || (p.ldc & (obf ? 7 : 3)) || !al16(p.c)) return SH_X6P_NO;
    if (p.extra != nullptr) {
        if (mode == FPROP) { if (!al16(p.extra)) return SH_X6P_NO; }
        else if ((p.ldadd & ((p.act & B16_ADD_BF) ? 7 : 3)) || !al16(p.extra)) return SH_X6P_NO;
    }
    if (bnb) {
        if ((p.bnb_ldy & ((p.act & 4) ? 7 : 3)) || !al16(p.bnb_y) || !al16(p.bnb_mean) || !al16(p.bnb_invstd) || !al16(p.bnb_scale) || !al16(p.bnb_shift) ||
            !al16(p.partials)) return SH_X6P_NO;
        if (p.bnb_out != nullptr && !(p.act & 64) && ((p.bnb_ldo & ((p.act & 8) ? 7 : 3)) || !al16(p.bnb_out))) return SH_X6P_NO;
        if (p.bnb_out != nullptr && (p.act & 64) && (((uintptr_t)p.bnb_out | (uintptr_t)p.bnb_ldo) & 1)) return SH_X6P_NO;      // two mask bytes per access
    }
    if (lin && ((p.lda2 & 7) || !al16(p.a2) || !al16(p.lin))) return SH_X6P_NO;
    p.vec_epi = 1;
    if (mode == FPROP) {
        if (bnb || lin || a32) return SH_EINVAL;
        return aff ? pick_b16<FPROP, 1, 1, 0>(p, st) : pick_b16<FPROP, 0, 1, 0>(p, st);
    }
    if (aff) return SH_EINVAL;
    if (rm == 1) {                // stride-2 KxK by parity class: no addend, whole K = 64 tiles per tap
        if (p.extra != nullptr || p.dil != 1 || p.KH * p.KW == 1 || (p.Kc & 63)) return SH_X6P_NO;
        if (p.Nn > 64) return a32 ? launch_b16<DGRAD, 2, 0, 0, 0, 1, 0, 1, 0, 1>(p, st) : launch_b16<DGRAD, 2, 0, 0, 0, 1, 0, 0, 0, 1>(p, st);
        return a32 ? launch_b16<DGRAD, 1, 0, 0, 0, 1, 0, 1, 0, 1>(p, st) : launch_b16<DGRAD, 1, 0, 0, 0, 1, 0, 0, 0, 1>(p, st);
    }
    if (rm == 2) {                // 1x1 strided conv: rows of the Ho x Wo GEMM are added to dx's rows at the strided pixels
        if (p.extra != nullptr || p.KH * p.KW != 1) return SH_X6P_NO;
        p.extra = p.c; p.ldadd = p.ldc;
        p.act = (p.act & ~B16_ADD_BF) | ((p.act & B16_OUT_BF) ? B16_ADD_BF : 0);
        if (p.Nn > 64) return a32 ? launch_b16<DGRAD, 2, 0, 0, 0, 0, 0, 1, 0, 2>(p, st) : launch_b16<DGRAD, 2, 0, 0, 0, 0, 0, 0, 0, 2>(p, st);
        return a32 ? launch_b16<DGRAD, 1, 0, 0, 0, 0, 0, 1, 0, 2>(p, st) : launch_b16<DGRAD, 1, 0, 0, 0, 0, 0, 0, 0, 2>(p, st);
    }
    if (a32) {
        if (lin) return SH_X6P_NO;                                // lin(g, y): g is a bf16 tensor in this mode
        return bnb ? pick_b16<DGRAD, 0, 2, 1>(p, st) : pick_b16<DGRAD, 0, 0, 1>(p, st);
    }
    if (lin) return bnb ? pick_b16<DGRAD, 2, 2, 0>(p, st) : pick_b16<DGRAD, 2, 0, 0>(p, st);
    return bnb ? pick_b16<DGRAD, 0, 2, 0>(p, st) : pick_b16<DGRAD, 0, 0, 0>(p, st);
}

// grouped 1x1 fprop (the ASPP branches): see sh_x6p_grouped_launch
int sh_b16_grouped_launch(ConvQ& p, hipStream_t st) {
    if (p.ngroups < 1 || p.ngroups > 6 || (p.group_n & 127) || p.KH * p.KW != 1 || (p.Kc & 7)) return SH_X6P_NO;
    const int obf = p.act & 2;
    if ((p.ldc & (obf ? 7 : 3)) || ((uintptr_t)p.c & 15)) return SH_X6P_NO;
    for (int g = 0; g < p.ngroups; ++g)
        if ((p.glda[g] & 7) || ((uintptr_t)p.ga[g] & 15) || ((uintptr_t)p.gb[g] & 15)) return SH_X6P_NO;
    p.vec_epi = 1;
    return launch_b16<FPROP, 2, 1, 1, 0, 0, 1, 0>(p, st);
}

// ============================================================================================ WGRAD, bf16 compute
// dW[co][n'] = sum_pix dY[pix][co] * im2col(X)[pix][n'] with ONE product per tile: K = 64 pixels per tile, [k][row] LDS planes read with
// the transposing ds_read_b64_tr_b16 exactly as conv_wgrad_x6p_kernel, two buffers, one register set (64 pixels of loads in flight per
// block, two blocks per CU).  Loader: thread = (pixel row, 16-byte chunk of 8 channels) -- a thread keeps ITS 8 channels for the whole
// kernel, so the BatchNorm coefficients of AFF (X = relu(x * scale + shift)) and of LIN (dY = lin(g, y)) sit in registers.
// DY32: the gradient stream (dy, or g of lin) is an fp32 tensor; ONE: 1x1 stride-1 conv (pixel index = row of X, no coordinate math).
template <int WGM, int WGN, int AFF, int LIN, int DY32, int ONE>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void conv_wgrad_b16_kernel(const ConvQ p) {
    constexpr int NT = 64 * WGM * WGN, BM = 64 * WGM, BN = 64 * WGN;
    constexpr int AROWB = 2 * BM + 64, BROWB = 2 * BN + 64;           // bytes per k-row (as the x6 planes)
    constexpr int APLANE = 64 * AROWB, BPLANE = 64 * BROWB;           // one buffer = 64 k-rows
    constexpr int ACPR = BM / 8, BCPR = BN / 8;                       // 16-byte chunks per k-row
    constexpr int AKPP = NT / ACPR, BKPP = NT / BCPR;                 // k-rows per loader pass
    constexpr int NA = 64 / AKPP, NB = 64 / BKPP;
    static_assert(NA >= 1 && NB >= 1 && AKPP * NA == 64 && BKPP * NB == 64, "loader geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const As = smem;
    unsigned char* const Bs = smem + 2 * APLANE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WGN, wn = wave % WGN, l31 = lane & 31, h = lane >> 5;
    const unsigned nblk = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
    unsigned slice, bid;
    if (p.scatter) { const unsigned kq = blockIdx.x >> 3; slice = (kq / nblk) * 8u + (blockIdx.x & 7u); bid = kq % nblk; }
    else { slice = blockIdx.x / nblk; bid = xcd_remap(blockIdx.x % nblk, nblk); }
    if (slice >= (unsigned)p.ksplit) return;
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = slice * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    const int nq = (kend - kbeg + 63) >> 6;

    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, p.b_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc_a2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(LIN ? p.a2 : p.a), 0, LIN ? p.a2_bytes : p.a_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int arc = t % ACPR, ak0 = t / ACPR;
    const int brc = t % BCPR, bk0 = t / BCPR;
    const int co = m0 + 8 * arc;
    const int nn = n0 + 8 * brc;
    const unsigned a_col_ok = co < p.M ? ~0u : 0u, b_col_ok = nn < p.Nn ? ~0u : 0u;       // Nn a multiple of 8; M: rows of pad8(M), zero padding
    const int tap = nn / p.Cin, wg_ci = nn - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int wg_dh = kh * p.dil - p.pad, wg_dw = kw * p.dil - p.pad;
    // BatchNorm coefficients of this block's rows / columns live in LDS behind the planes: lin[4][BM] (A, B, mean, D per output channel),
    // aff[2][BN] (scale, shift per im2col column); a thread re-reads ITS 8 of each per store pass (in registers for the whole kernel
    // they cost 48 VGPRs next to 64-100 staging registers: spills)
    float* const lin_c = reinterpret_cast<float*>(smem + 2 * (APLANE + BPLANE));
    float* const aff_c = lin_c + (LIN ? 4 * BM : 0);
    static_assert(!LIN || !DY32, "lin(g, y): g is a bf16 tensor in this mode");
    if constexpr (LIN) {
        for (int c = t; c < 4 * BM; c += NT) { const int r = c / BM, m = m0 + c % BM; lin_c[c] = m < p.M ? p.lin[(long long)r * p.M + m] : 0.f; }
    }
    if constexpr (AFF) {
        for (int c = t; c < BN; c += NT) {
            const int col = n0 + c;
            const bool okc = col < p.Nn;
            const int ci = okc ? col % p.Cin : 0;
            aff_c[c] = okc ? p.aff_scale[ci] : 0.f; aff_c[BN + c] = okc ? p.aff_shift[ci] : 0.f;
        }
    }
    if constexpr (LIN || AFF) __syncthreads();
    // AFF without LIN: a thread's 8 im2col columns (one tap, 8 channels) are the same for every tile, so its 16 coefficients live in
    // registers (162-165 of the 256 available at two waves per SIMD) instead of four LDS reads per stored chunk -- the table reads were
    // 2-way bank conflicts (19 % of the kernel's LDS cycles, profiles/r03a_pmc_b16_sep1pw.json)
    [[maybe_unused]] float aff_sc[8], aff_sh[8];
    if constexpr (AFF && !LIN) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { aff_sc[e] = aff_c[8 * brc + e]; aff_sh[e] = aff_c[BN + 8 * brc + e]; }
    }
    // pixel coordinates of this thread's B k-rows for the NEXT tile, advanced by 64 pixels per tile: (ow, oh, n) += (64 % Wo, 64 / Wo, 0)
    // with one carry each (host check: 64 / Wo + 1 < Ho, so a step crosses at most one image boundary)
    [[maybe_unused]] int px_ow[NB], px_oh[NB], px_n[NB];
    const int step_ow = ONE ? 0 : 64 % p.Wo, step_oh = ONE ? 0 : 64 / p.Wo;
    if constexpr (!ONE) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int pix = kbeg + bk0 + BKPP * i;
            px_ow[i] = pix % p.Wo; const int q2 = pix / p.Wo; px_oh[i] = q2 % p.Ho; px_n[i] = q2 / p.Ho;
        }
    }
    struct Regs { u32x4 a[NA][DY32 ? 2 : 1]; u32x4 a2[LIN ? NA : 1]; u32x4 b[NB]; unsigned okm, okl; };
    auto load_tile = [&](int q, Regs& R) {
        const int kbase = kbeg + 64 * q;
        R.okm = 0; R.okl = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int pix = kbase + ak0 + AKPP * i;
            const unsigned ok = (unsigned)((pix - kend) >> 31) & a_col_ok;
            const unsigned e = (unsigned)(pix * (int)p.lda + co);
            if constexpr (DY32) {
                const unsigned voff = ((e * 4u) & ok) | (OOB & ~ok);
                R.a[i][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff, 0, 0));
                R.a[i][DY32 ? 1 : 0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff + 16u, 0, 0));
            } else {
                const unsigned voff = ((e * 2u) & ok) | (OOB & ~ok);
                R.a[i][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff, 0, 0));
            }
            if constexpr (LIN) {
                const unsigned v2 = (((unsigned)(pix * (int)p.lda2 + co) * 2u) & ok) | (OOB & ~ok);
                R.a2[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a2, v2, 0, 0));
                R.okl |= ok & (1u << i);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int pix = kbase + bk0 + BKPP * i;
            unsigned ok = (unsigned)((pix - kend) >> 31) & b_col_ok;
            unsigned e;
            if constexpr (ONE) e = (unsigned)(pix * (int)p.ldb + wg_ci);
            else {
                const int ih = px_oh[i] * p.stride + wg_dh, iw = px_ow[i] * p.stride + wg_dw;
                ok &= ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? ~0u : 0u;
                e = (unsigned)(((px_n[i] * p.H + ih) * p.W + iw) * (int)p.ldb + wg_ci);
                int ow = px_ow[i] + step_ow, oh = px_oh[i] + step_oh, n = px_n[i];
                const bool c1 = ow >= p.Wo; ow -= c1 ? p.Wo : 0; oh += c1 ? 1 : 0;
                const bool c2 = oh >= p.Ho; oh -= c2 ? p.Ho : 0; n += c2 ? 1 : 0;
                px_ow[i] = ow; px_oh[i] = oh; px_n[i] = n;
            }
            const unsigned voff = ((e * 2u) & ok) | (OOB & ~ok);
            R.b[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, voff, 0, 0));
            if (AFF) R.okm |= ok & (1u << i);
        }
    };
    auto store_tile = [&](Regs& R, int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            unsigned char* const dst = As + buf * APLANE + (ak0 + AKPP * i) * AROWB + arc * 16;
            if constexpr (!DY32 && !LIN) { *reinterpret_cast<u32x4*>(dst) = R.a[i][0]; continue; }
            float v[8];
            if constexpr (DY32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(R.a[i][0][e]); v[4 + e] = __uint_as_float(R.a[i][DY32 ? 1 : 0][e]); }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(R.a[i][0][e] << 16); v[2 * e + 1] = __uint_as_float(R.a[i][0][e] & 0xffff0000u); }
            }
            if constexpr (LIN) {          // pixels beyond the K slice stay exact zeros (D != 0)
                const bool ok = (R.okl >> i) & 1u;
                float lA[8], lB[8], lM[8], lD[8];
#pragma unroll
                for (int e = 0; e < 8; e += 4) {
                    *reinterpret_cast<f32x4*>(lA + e) = *reinterpret_cast<const f32x4*>(lin_c + 8 * arc + e);
                    *reinterpret_cast<f32x4*>(lB + e) = *reinterpret_cast<const f32x4*>(lin_c + BM + 8 * arc + e);
                    *reinterpret_cast<f32x4*>(lM + e) = *reinterpret_cast<const f32x4*>(lin_c + 2 * BM + 8 * arc + e);
                    *reinterpret_cast<f32x4*>(lD + e) = *reinterpret_cast<const f32x4*>(lin_c + 3 * BM + 8 * arc + e);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y0 = __uint_as_float(R.a2[i][e] << 16), y1 = __uint_as_float(R.a2[i][e] & 0xffff0000u);
                    const float w0 = fmaf(y0 - lM[2 * e], lB[2 * e], fmaf(v[2 * e], lA[2 * e], lD[2 * e]));
                    const float w1 = fmaf(y1 - lM[2 * e + 1], lB[2 * e + 1], fmaf(v[2 * e + 1], lA[2 * e + 1], lD[2 * e + 1]));
                    v[2 * e] = ok ? w0 : 0.f; v[2 * e + 1] = ok ? w1 : 0.f;
                }
            }
            *reinterpret_cast<u32x4*>(dst) = u32x4{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            unsigned char* const dst = Bs + buf * BPLANE + (bk0 + BKPP * i) * BROWB + brc * 16;
            if constexpr (!AFF) { *reinterpret_cast<u32x4*>(dst) = R.b[i]; continue; }
            const bool ok = (R.okm >> i) & 1u;
            float v[8], sc[8], sh[8];
            if constexpr (LIN) {          // (with the lin table in play the coefficients stay in LDS: 48 more registers spilled)
#pragma unroll
                for (int e = 0; e < 8; e += 4) {
                    *reinterpret_cast<f32x4*>(sc + e) = *reinterpret_cast<const f32x4*>(aff_c + 8 * brc + e);
                    *reinterpret_cast<f32x4*>(sh + e) = *reinterpret_cast<const f32x4*>(aff_c + BN + 8 * brc + e);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { sc[e] = aff_sc[e]; sh[e] = aff_sh[e]; }
            }
            const unsigned okw = ok ? ~0u : 0u;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e)          // (as conv_b16_kernel's loader: packed, fused, masked after the pack)
                o[e] = aff_pair(R.b[i][e], f32x2_t{sc[2 * e], sc[2 * e + 1]}, f32x2_t{sh[2 * e], sh[2 * e + 1]}, 0u) & okw;
            (void)v;
            *reinterpret_cast<u32x4*>(dst) = o;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // transpose-read addressing (conv_wgrad_x6p_kernel): 16-lane group g -> (row half g&1, k half g>>1); lane 4q+pp supplies row q, columns 4pp..4pp+3
    const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
    const int krow = (g >> 1) * 8 + qq, coff = (16 * (g & 1) + 4 * pp) * 2;
    auto frag = [&](const unsigned char* plane, int rowb, int ks, int rowbase) -> bf16x8 {
        const unsigned char* a0 = plane + (krow + ks * 16) * rowb + coff + rowbase * 2;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 4 * rowb));
        s16x8 r;
        r[0] = v0[0]; r[1] = v0[1]; r[2] = v0[2]; r[3] = v0[3]; r[4] = v1[0]; r[5] = v1[1]; r[6] = v1[2]; r[7] = v1[3];
        return __builtin_bit_cast(bf16x8, r);
    };
    auto compute_tile = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = frag(As + buf * APLANE, AROWB, ks, wm * 64 + 32 * i);
                bfr[i] = frag(Bs + buf * BPLANE, BROWB, ks, wn * 64 + 32 * i);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(af[i], bfr[j], acc[i][j]);
        }
    };
    Regs R0;
    load_tile(0, R0);
    store_tile(R0, 0);
    load_tile(1, R0);
    __syncthreads();
    for (int j = 0; j < nq; j += 2) {
        store_tile(R0, 1);
        load_tile(j + 2, R0);
        compute_tile(0);
        __syncthreads();
        store_tile(R0, 0);
        load_tile(j + 3, R0);
        if (j + 1 < nq) compute_tile(1);
        __syncthreads();
    }
    float* slab = p.c + (long long)slice * p.M * p.Nn;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.M && n < p.Nn) slab[(long long)m * p.Nn + n] = acc[i][j][r];
            }
        }
}
template <int WGM, int WGN, int AFF, int LIN, int DY32, int ONE>
static int launch_wgrad_b16(ConvQ& p, int splits, hipStream_t st) {
    constexpr size_t lds = 2 * 64 * (size_t)((2 * 64 * WGM + 64) + (2 * 64 * WGN + 64)) + (LIN ? 16 * 64 * WGM : 0) + (AFF ? 8 * 64 * WGN : 0);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_b16_kernel<WGM, WGN, AFF, LIN, DY32, ONE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    p.tiles_m = (int)sh_cdiv(p.M, 64 * WGM); p.tiles_n = (int)sh_cdiv(p.Nn, 64 * WGN);
    p.ksplit = splits;
    const unsigned grid = (unsigned)(p.tiles_m * p.tiles_n) * (unsigned)(p.scatter ? sh_cdiv(splits, 8) * 8 : splits);
    conv_wgrad_b16_kernel<WGM, WGN, AFF, LIN, DY32, ONE><<<grid, 64 * WGM * WGN, lds, st>>>(p);
    return sh_launch_status();
}
template <int AFF, int LIN, int DY32, int ONE>
static int pick_wgrad_b16(ConvQ& p, int wgm, int wgn, int splits, hipStream_t st) {
    if (wgm == 4 && wgn == 1) return launch_wgrad_b16<4, 1, AFF, LIN, DY32, ONE>(p, splits, st);
    if (wgm == 2 && wgn == 1) return launch_wgrad_b16<2, 1, AFF, LIN, DY32, ONE>(p, splits, st);
    if (wgm == 2 && wgn == 4) return launch_wgrad_b16<2, 4, AFF, LIN, DY32, ONE>(p, splits, st);
    if (wgm == 1 && wgn == 4) return launch_wgrad_b16<1, 4, AFF, LIN, DY32, ONE>(p, splits, st);
    if (wgm == 1 && wgn == 2) return launch_wgrad_b16<1, 2, AFF, LIN, DY32, ONE>(p, splits, st);
    return launch_wgrad_b16<2, 2, AFF, LIN, DY32, ONE>(p, splits, st);
}
template <int AFF, int LIN>
static int pick_wgrad_b16_src(ConvQ& p, int dy32, bool one, int wgm, int wgn, int splits, hipStream_t st) {
    if constexpr (LIN) { if (dy32 || !one) return SH_X6P_NO; return pick_wgrad_b16<AFF, 1, 0, 1>(p, wgm, wgn, splits, st); }
    else {
        if (dy32) return one ? pick_wgrad_b16<AFF, 0, 1, 1>(p, wgm, wgn, splits, st) : pick_wgrad_b16<AFF, 0, 1, 0>(p, wgm, wgn, splits, st);
        return one ? pick_wgrad_b16<AFF, 0, 0, 1>(p, wgm, wgn, splits, st) : pick_wgrad_b16<AFF, 0, 0, 0>(p, wgm, wgn, splits, st);
    }
}
// Entry used by sh_conv_wgrad_b16 (same tile / K-slice plan as the fp32-accurate kernels; the slab reduce stays with the caller).
// X (p.b) and the lin y stream (p.a2) are bf16 tensors; dy32: the gradient stream p.a is fp32.
int sh_b16_wgrad_launch(ConvQ& p, int dy32, int wgm, int wgn, int splits, hipStream_t st) {
    const bool aff = p.aff_scale != nullptr, lin = p.lin != nullptr;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    // (M = Cout need not be a multiple of 8: the caller guarantees dy rows of pad8(Cout) with zeroed padding lanes -- sh_conv_wgrad_b16)
    if ((lin && (p.M & 7)) || (p.Cin & 7) || (p.lda & 7) || (p.ldb & 7) || !al16(p.a) || !al16(p.b)) return SH_X6P_NO;
    if (lin && ((p.lda2 & 7) || !al16(p.a2))) return SH_X6P_NO;
    const bool one = p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0;
    if (!one && (64 / p.Wo + 1 >= p.Ho)) return SH_X6P_NO;          // the 64-pixel coordinate step carries at most one image boundary
    if (aff) return lin ? pick_wgrad_b16_src<1, 1>(p, dy32, one, wgm, wgn, splits, st) : pick_wgrad_b16_src<1, 0>(p, dy32, one, wgm, wgn, splits, st);
    return lin ? pick_wgrad_b16_src<0, 1>(p, dy32, one, wgm, wgn, splits, st) : pick_wgrad_b16_src<0, 0>(p, dy32, one, wgm, wgn, splits, st);
}
