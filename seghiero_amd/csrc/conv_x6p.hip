// Software-pipelined fprop / dgrad kernels of the "6 x bf16 split" convolution path (see conv_bf16x6.hip for the arithmetic).
//
// Why a second main loop.  conv_x6_kernel stages one K=32 tile at a time through a single LDS buffer: every wave computes,
// barrier, every wave waits for its global loads, splits fp32 -> 3 x bf16 and stores, barrier.  Measured with s_memtime
// (round 1): the matrix pipe idles 38 % of the K loop -- all waves of the block are in the split / store phase (pure
// VALU + LDS writes) at the same time, and both barriers and the load wait are exposed.  Here a K=32 tile is handled as two
// K=16 half-tiles that live in the two k-halves of the SAME LDS image (same 64-byte rows, same swizzle family, no extra LDS):
//
//     phase j :  split + store half-tile j+1 (registers -> LDS half (j+1)&1)      VALU + ds_write
//                issue the global loads of half-tile j+3 (-> the register set just freed)   VMEM
//                ds_read fragments of half-tile j (LDS half j&1), TM*TN*6 MFMAs            LDS + MFMA
//                ONE barrier
//
// The three streams of a phase are independent, so they sit in one basic block and interleave: the split of the next half-tile
// runs in the issue shadow of this half-tile's MFMAs (an MFMA holds the vector issue port 8 of its 32 cycles), global loads
// have two full phases to land, and there is one barrier per K=16 instead of two per K=32.
// Loader mapping: thread = (row, 16-byte k chunk of a half-tile); the two half-tiles of a 128-byte line are fetched by the
// same thread in consecutive phases.
//
// Tried and dropped (round 2): B fragments straight from pre-split, fragment-ordered bf16 weight planes in global memory (no split, no
// LDS image, no staging registers for B; bit-identical results).  Each wave then fetches its own 6 x 1 KiB of fragments per phase --
// 3x the L2 -> CU bytes of the fp32 tile staged once per block -- and the vector-memory path becomes the limit: fprop 512->512
// @16x128^2 1016 vs 714 us, step 34.7 vs 33.8 ms.  The same planes staged through LDS like the fp32 tile (three 16-byte loads and
// ds_write_b128 per thread instead of two loads, the split and six ds_write_b64; bit-identical): dgrad -3 %, fprop +3.7 %, step
// 34.2 vs 33.9 ms with the per-step pack kernel -- also dropped.  (A timing build that merely skipped the B split, with the fp32
// loads unchanged, ran fprop / dgrad 6 % faster: the loop is balanced across VALU, LDS and vector memory, and moving work from one
// to another does not shorten it.)
//
// Fused BatchNorm hooks (the reason the loaders go through registers at all):
//   AFF : A operand = relu(x * scale[c] + shift[c]) applied in the loader -- the producer's train-mode BatchNorm + ReLU
//         (reference: every conv -> BN -> ReLU -> conv chain, models/backbone/resnet.py:65-73, sep_aspp_contrast_head.py:56-61),
//         so the activated tensor is never written or re-read; zero padding stays an exact zero AFTER the activation.
//   AFF == 2 (1x1 dgrad): A operand = lin(g, y) = A[c]*g + B[c]*(y - mean[c]) + D[c] -- the second half of the BatchNorm backward of
//         the conv's OWN BatchNorm (sh_bn_bwd_apply) evaluated in the loader from the masked gradient g and the raw conv output y,
//         so the dy tensor is never written or re-read (coefficients: sh_bn_bwd_finalize's lin).
//   EPI == 1 : fprop epilogue emits the centred BatchNorm (sum, M2) partials per 64 rows (as conv_x6_kernel).
//   EPI == 2 : dgrad epilogue is the front half of the BatchNorm backward of the producer layer: g = relumask * dx is stored and
//              (sum g, sum g * xhat) per 64 rows are emitted, so no separate statistics pass reads dx and y again.
#include "conv_x6.h"

// 16-byte k-chunk swizzle of a 64-byte LDS row: (row>>2)&3 keeps the 16 lanes of every ds_read_b128 lane group on distinct
// banks (as in conv_x6_kernel); the extra (row & 2) term swaps the two k-halves on every other row pair so that the 16 lanes of
// a ds_write_b64 group -- 4 consecutive rows x the 4 chunks-halves of ONE k-half -- cover all 32 store banks once.
__device__ __forceinline__ int swz_row(int row) { return ((row >> 2) & 3) ^ (row & 2); }

#ifndef SH_W1          // blocks per CU (launch bounds) of the 128 x 128 / 128 x 64 instantiations; -D overrides for experiments
#define SH_W1(AFF, EPI) ((AFF) == 2 ? 2 : 3)
#define SH_W2(AFF, EPI) (((AFF) == 2 || (EPI) == 2) ? 3 : 4)
#endif
// TAP: 0 = 1x1 (one tap), 1 = KxK with Kc % 16 == 0 (a half-tile never straddles taps: scalar tap math, switched by a block-uniform
// branch at the start of a phase), 2 = general (per-lane tap; stem 7x7 with 4 channels)
// ABF: the activation stream of the loader is stored as bf16 (fprop: the A operand x; AFF == 2 dgrad: the y stream of lin(g, y)) --
// 8-byte loads of 4 elements, widened to fp32 when the half-tile goes to LDS; everything downstream is unchanged
// This file compiles as ONE translation unit (SH_X6P_PART undefined or 0) or as six (build.sh: -DSH_X6P_PART=1 fprop + grouped launch,
// 2 dgrad, 3-6 the wgrad families by (lin loader, bf16 streams) -- the instantiations of one part each, so they compile side by side:
// 4m45 of wall time for the one unit, about 1m10 for the slowest part).
#ifndef SH_X6P_PART
#define SH_X6P_PART 0
#endif
#define SH_PART(n) (SH_X6P_PART == 0 || SH_X6P_PART == (n))
int sh_x6p_launch_dgrad(ConvQ& p, hipStream_t st, int force);
int sh_x6p_wgrad_l0b0(ConvQ& p, bool aff, int wgm, int wgn, int splits, hipStream_t st);
int sh_x6p_wgrad_l1b0(ConvQ& p, bool aff, int wgm, int wgn, int splits, hipStream_t st);
int sh_x6p_wgrad_l0b1(ConvQ& p, bool aff, int wgm, int wgn, int splits, hipStream_t st);
int sh_x6p_wgrad_l1b1(ConvQ& p, bool aff, int wgm, int wgn, int splits, hipStream_t st);
static int x6p_mode() { static int v = -2; if (v == -2) { const char* e = getenv("SEGHIERO_X6P"); v = e ? atoi(e) : 1; } return v; }
[[maybe_unused]] static int x6p_ws() { static int v = -2; if (v == -2) { const char* e = getenv("SEGHIERO_X6P_WS"); v = e ? atoi(e) : 0; } return v; }
static int x6p_tile() { static int v = -2; if (v == -2) { const char* e = getenv("SEGHIERO_X6P_TILE"); v = e ? atoi(e) : 0; } return v; }
#if SH_PART(1) || SH_PART(2) || defined(SH_X6P_EXP)
// WS (r3): wave-specialised form.  The block carries a second set of WGM * WGN waves; waves 0 .. WGM*WGN-1 ("consumers") only read
// fragments and issue MFMAs, the others ("producers") only load, apply the BatchNorm hooks, split and store -- one consumer and one
// producer of the block on every SIMD, so the matrix pipe and the vector ALU are fed from two instruction streams the hardware arbitrates
// cycle by cycle instead of from one in-order stream whose interleave is the compiler's (which clusters the MFMAs at the end of a phase
// whatever the sched_group_barrier pattern asks for -- see DESIGN.md section 7).  Same LDS image, same phases, same barrier count.
// MEASURED (r3, same box, alternating processes): bit-identical results, 126 VGPRs, no spills -- and no faster: fprop 165.0 -> 163.9 TF,
// dgrad 91.6 -> 88.6 TF over the step's shapes, step 32.0 -> 32.5 ms.  Co-issue is therefore not what limits the loop (the vector
// port needs ~150 VALU x 4 cycles per 768 MFMA cycles of a phase); instantiated only in experiment builds (-DSH_X6P_WS).
template <int MODE, int TM, int TN, int WGM, int WGN, int WPS, int AFF, int EPI, int SK, int TAP, int GRP = 0, int ABF = 0, int WS = 0>
__global__ __launch_bounds__(64 * WGM * WGN * (WS ? 2 : 1), WPS) void conv_x6p_kernel(const ConvQ p) {
    constexpr int NT = 64 * WGM * WGN;                           // loader threads = consumer threads
    constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
    constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;      // bytes
    constexpr int RPP = NT / 4;                                  // rows covered per loader pass (4 lanes x 16 B per half-tile row)
    constexpr int NA = BM / RPP, NB = BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0 && RPP % 64 == 0, "tile / thread-count mismatch");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const As = smem;
    unsigned char* const Bs = smem + 3 * A_PLANE;
    float* const coef = reinterpret_cast<float*>(smem + 3 * (A_PLANE + B_PLANE));     // AFF 1: scale[Kc], shift[Kc]; AFF 2: lin[4][Kc]
    static_assert(AFF != 2 || (MODE == DGRAD && TAP == 0 && !GRP), "the deferred BatchNorm-backward loader exists for 1x1 dgrads");
    static_assert(!ABF || (MODE == FPROP && !GRP) || AFF == 2, "bf16 loader streams: the fprop input, or the y stream of the lin loader");
    constexpr bool A_BF = ABF && MODE == FPROP, A2_BF = ABF && AFF == 2;
    // A3: the A operand is a STORED bf16 tensor read as is (no activation in the loader): x = x_hi exactly, its mid / lo planes are zero,
    // so three of the six products vanish -- half the MFMA work and no split of A (the conv1 / downsample convs of a bf16-stored trunk)
    constexpr bool A3 = A_BF && AFF == 0;
    auto load_a4 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned elem_off, unsigned ok, bool bf) -> f32x4 {
        // 4 consecutive elements: fp32 -> one 16-byte load; bf16 -> one 8-byte load parked in the first two lanes of the f32x4 (widened
        // by widen_a4 when the half-tile is stored); offsets beyond the buffer (masked rows / taps / K tail) read as zeros
        if (bf) {
            const unsigned voff = ((elem_off * 2u) & ok) | (0x80000000u & ~ok);
            const u32x2 r = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0));
            return f32x4{__uint_as_float(r[0]), __uint_as_float(r[1]), 0.f, 0.f};
        }
        const unsigned voff = ((elem_off * 4u) & ok) | (0x80000000u & ~ok);
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
    };
    auto widen_a4 = [&](f32x4 v, bool bf) -> f32x4 {
        return bf ? bf16x4_to_f32(sh_u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}) : v;
    };

    const int t_all = threadIdx.x, t = WS ? (t_all & (NT - 1)) : t_all, lane = t & 63, wave = t >> 6;
    [[maybe_unused]] const bool producer = WS && t_all >= NT;    // wave-uniform
    static_assert(!WS || (NT & (NT - 1)) == 0, "wave-specialised form: power-of-two wave count");
    const int wm = wave / WGN, wn = wave % WGN, l31 = lane & 31, h = lane >> 5;
    const unsigned nblk = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
    const unsigned bid = xcd_remap(blockIdx.x, nblk);
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // GRP: the block's N tile lies inside one group (group_n is a multiple of BN): that group's input, weights and coefficients
    const int grp = GRP ? n0 / p.group_n : 0;
    const float* const a_ptr = GRP ? p.ga[grp] : p.a;
    const float* const b_ptr = GRP ? p.gb[grp] : p.b;
    const unsigned a_bytes = GRP ? p.ga_bytes[grp] : p.a_bytes;
    const int lda = (int)(GRP ? p.glda[grp] : p.lda);
    const float* const sc_ptr = GRP ? p.gsc[grp] : p.aff_scale;
    const float* const sh_ptr = GRP ? p.gsh[grp] : p.aff_shift;
    const float aff_floor = (GRP && sc_ptr == nullptr) ? -INFINITY : 0.f;      // plain input: max(x * 1 + 0, -inf) = x exactly
    int cy = 0, cx = 0, Hc = p.H, Wc = p.W, oy0 = 0, ox0 = 0, ntw = p.KW, Mc = p.M, Kt = p.K;
    if constexpr (MODE == DGRAD) {
        if (p.parity) {                                              // stride-2 KxK dgrad by input-parity class (see conv_x6_kernel)
            cy = blockIdx.y >> 1; cx = blockIdx.y & 1;
            oy0 = (cy + p.pad) & 1; ox0 = (cx + p.pad) & 1;
            Hc = (p.H - oy0 + 1) >> 1; Wc = (p.W - ox0 + 1) >> 1;
            const int nth = (p.KH - cy + 1) >> 1;
            ntw = (p.KW - cx + 1) >> 1;
            Mc = p.N * Hc * Wc; Kt = nth * ntw * p.Kc;
            if (m0 >= Mc) return;                                    // block-uniform, before any barrier
        }
    }
    int q_begin = 0, q_end = (Kt + 15) >> 4;                         // half-tiles of K=16
    if constexpr (SK) {
        const int per = ((q_end + p.ksplit - 1) / p.ksplit + 1) & ~1;            // whole K=32 tiles per slice
        q_begin = min(q_end, (int)blockIdx.y * per); q_end = min(q_end, q_begin + per);
    }
    const int klim = min(Kt, 16 * q_end);                            // loads at k >= klim return zeros (per-lane compare: no scalar branch)
    const int nq = (q_end - q_begin + 1) & ~1;                       // whole K=32 tiles: an odd tail half-tile is followed by a zero one
    constexpr bool single_tap = TAP == 0, tap_uniform = TAP <= 1;

    // Operands are fetched with raw buffer loads: 32-bit byte offsets against a wave-uniform descriptor, and an offset beyond
    // num_records (padded taps, rows >= M, the K tail) returns zeros in hardware -- no branch and no select around any load, so
    // a whole phase stays one basic block.  (The host side routes tensors of 2 GiB or more to conv_x6_kernel.)
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_ptr), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b_ptr), 0, p.b_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc_a2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(AFF == 2 ? p.a2 : a_ptr), 0, AFF == 2 ? p.a2_bytes : a_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int kc = t & 3, r0 = t >> 2;
    int a_y[NA], a_x[NA], a_nb[NA];
    int b_row[NB];                                               // element offset of the B row (< 2^29)
    unsigned b_ok[NB];                                           // validity as all-ones / zero masks: offsets are formed with
    int cur_tap = -1;                                            // AND / OR only, nothing the compiler can turn into a branch
    // (r3) 1x1, stride 1, no padding: output pixel m is row m of the operand -- no integer divisions in the prologue (measured on the bf16
    // kernels: ~300 of a wave's 1 300 prologue / epilogue VALU instructions)
    const bool direct = TAP == 0 && p.stride == 1 && p.pad == 0 && !(MODE == DGRAD && (p.parity || p.scatter));
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + r0 + RPP * i;
        if (direct) { a_y[i] = m < Mc ? 0 : -(1 << 28); a_x[i] = 0; a_nb[i] = m < Mc ? m : 0; }
        else if (m < Mc) {
            if constexpr (MODE == FPROP) {
                const int ow = m % p.Wo, q = m / p.Wo, oh = q % p.Ho, n = q / p.Ho;
                a_y[i] = oh * p.stride - p.pad; a_x[i] = ow * p.stride - p.pad; a_nb[i] = n * p.H * p.W;
            } else {
                const int iwc = m % Wc, q = m / Wc, ihc = q % Hc, n = q / Hc;
                const int ih = p.parity ? 2 * ihc + oy0 : ihc, iw = p.parity ? 2 * iwc + ox0 : iwc;
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
    int a_off[NA];                                               // element offset of the A row for the current tap
    [[maybe_unused]] int a_off2[NA];                             // ... and of the y row (AFF == 2: its own pixel stride)
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
        int kh, kw;
        if (MODE == DGRAD && p.parity) { const int ty = tap / ntw; kh = cy + 2 * ty; kw = cx + 2 * (tap - ty * ntw); }
        else { kh = tap / p.KW; kw = tap - kh * p.KW; }
        const int dh = kh * p.dil, dw = kw * p.dil;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if constexpr (MODE == FPROP) {
                const int ih = a_y[i] + dh, iw = a_x[i] + dw;
                a_ok[i] = ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? ~0u : 0u;
                a_off[i] = (a_nb[i] + ih * p.W + iw) * lda;
            } else {
                int th = a_y[i] - dh, tw = a_x[i] - dw;
                bool ok = th >= 0 && tw >= 0;
                if (p.stride > 1) {
                    ok = ok && (th % p.stride == 0) && (tw % p.stride == 0);
                    th /= p.stride; tw /= p.stride;
                }
                a_ok[i] = (ok && th < p.Ho && tw < p.Wo) ? ~0u : 0u;
                a_off[i] = (a_nb[i] + th * p.Wo + tw) * lda;
                if constexpr (AFF == 2) a_off2[i] = (a_nb[i] + th * p.Wo + tw) * (int)p.lda2;
            }
        }
    };
    // tap of the half-tile whose loads are issued next (scalar when tap_uniform); switching taps is a (block-uniform) branch
    // that is taken at the START of a phase so that the rest of the phase stays one basic block
    auto prepare_tap = [&](int q) {
        if constexpr (TAP == 1) {
            const int tap = (16 * q) / p.Kc;
            if (tap != cur_tap) { set_tap(tap); cur_tap = tap; }
        }
    };
    if constexpr (TAP == 0) { set_tap(0); cur_tap = 0; }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    struct Regs { f32x4 a[NA]; f32x4 a2[AFF == 2 ? NA : 1]; f32x4 b[NB]; int cc; unsigned okm; };
    auto load_half = [&](int q, Regs& R) {
        const int kbase = 16 * q, k = kbase + 4 * kc;
        int tap, cc;
        if constexpr (single_tap) { tap = 0; cc = k; }
        else if constexpr (tap_uniform) { tap = cur_tap; cc = k - tap * p.Kc; }
        else { tap = k / p.Kc; cc = k - tap * p.Kc; set_tap(tap); }
        const unsigned kok = (unsigned)((k - klim) >> 31);          // all ones while k < klim
        int wtap = tap;
        if (MODE == DGRAD && p.parity) { const int ty = tap / ntw; wtap = (cy + 2 * ty) * p.KW + cx + 2 * (tap - ty * ntw); }
        R.cc = cc; R.okm = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned ok = kok & a_ok[i];
            R.a[i] = load_a4(rsrc_a, (unsigned)(a_off[i] + cc), ok, A_BF);
            if constexpr (AFF == 2) R.a2[i] = load_a4(rsrc_a2, (unsigned)(a_off2[i] + cc), ok, A2_BF);
            if (AFF) R.okm |= ok & (1u << i);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int off;
            if constexpr (MODE == FPROP) off = b_row[i] + k;
            else off = wtap * p.Cin * p.Kc + b_row[i] + cc;
            const unsigned okb = kok & b_ok[i];
            const unsigned voff = (((unsigned)off * 4u) & okb) | (OOB & ~okb);
            R.b[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, voff, 0, 0));
        }
    };
    const int swz_w = swz_row(r0);                               // RPP is a multiple of 64: every row of this thread has r0's low bits
    auto store_half = [&](Regs& R, int hb) {
        const int off0 = r0 * ROWB + ((((2 * hb) | (kc >> 1)) ^ swz_w) << 4) + ((kc & 1) << 3);
        f32x4 sc = zero4, sh = zero4;
        [[maybe_unused]] f32x4 lmu = zero4, ld_ = zero4;
        if constexpr (AFF) { sc = *reinterpret_cast<const f32x4*>(coef + R.cc); sh = *reinterpret_cast<const f32x4*>(coef + p.Kc + R.cc); }
        if constexpr (AFF == 2) { lmu = *reinterpret_cast<const f32x4*>(coef + 2 * p.Kc + R.cc); ld_ = *reinterpret_cast<const f32x4*>(coef + 3 * p.Kc + R.cc); }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if constexpr (A3) {          // the 8 loaded bytes ARE the hi plane's four elements (same element order as split4's packing)
                *reinterpret_cast<u32x2*>(As + off0 + RPP * i * ROWB) = u32x2{__float_as_uint(R.a[i][0]), __float_as_uint(R.a[i][1])};
                continue;
            }
            f32x4 v = widen_a4(R.a[i], A_BF);
            if constexpr (AFF == 2) {
                // dy = A*g + B*(y - mean) + D (sc = A, sh = B): rows beyond M / the K tail must stay exact zeros (D != 0)
                const bool ok = (R.okm >> i) & 1u;
                const f32x4 y2 = widen_a4(R.a2[i], A2_BF);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float w = fmaf(y2[e] - lmu[e], sh[e], fmaf(v[e], sc[e], ld_[e])); v[e] = ok ? w : 0.f; }
            } else if constexpr (AFF) {
                // bn_act_kernel's own operation order (y * scale + shift, then max with 0): the loader sees exactly the values the
                // separate BatchNorm + ReLU pass would have written; padded taps / rows beyond M stay exact zeros
                const bool ok = (R.okm >> i) & 1u;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float w = fmaxf(v[e] * sc[e] + sh[e], aff_floor); v[e] = ok ? w : 0.f; }
            }
            u32x2 q1, q2, q3;
            split4(v, q1, q2, q3);
            const int off = off0 + RPP * i * ROWB;
            *reinterpret_cast<u32x2*>(As + off) = q1;
            *reinterpret_cast<u32x2*>(As + A_PLANE + off) = q2;
            *reinterpret_cast<u32x2*>(As + 2 * A_PLANE + off) = q3;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            u32x2 q1, q2, q3;
#ifdef SH_ABLATE_BSPLIT      // timing-only build (wrong numerics): what the B-operand split costs -- the ceiling of pre-split weight planes
            q1[0] = __float_as_uint(R.b[i][0]); q1[1] = __float_as_uint(R.b[i][1]); q2 = q1; q3[0] = __float_as_uint(R.b[i][2]); q3[1] = __float_as_uint(R.b[i][3]);
#else
            split4(R.b[i], q1, q2, q3);
#endif
            const int off = off0 + RPP * i * ROWB;
            *reinterpret_cast<u32x2*>(Bs + off) = q1;
            *reinterpret_cast<u32x2*>(Bs + B_PLANE + off) = q2;
            *reinterpret_cast<u32x2*>(Bs + 2 * B_PLANE + off) = q3;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int arow = wm * 32 * TM + l31, brow = wn * 32 * TN + l31;
    const int swz_r = swz_row(l31);
    auto compute_half = [&](int hb) {
        const int rd = (((2 * hb) | h) ^ swz_r) << 4;
        bf16x8 bfr[TN][3];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bfr[j][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bs + pl * B_PLANE + (brow + 32 * j) * ROWB + rd));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (A3) {
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(As + (arow + 32 * i) * ROWB + rd));
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mma3(a0, bfr[j], acc[i][j]);
            } else {
            bf16x8 af[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                af[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(As + pl * A_PLANE + (arow + 32 * i) * ROWB + rd));
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mma6(af, bfr[j], acc[i][j]);
            }
        }
    };

    // Issue order of one phase (a compile-time directive to the scheduler, guide T19): the B fragments and the first A fragment
    // first, then every MFMA carries a few VALU instructions of the split in its issue shadow; the LDS stores, the global loads
    // and the remaining A-fragment reads are spread over the MFMA stream.
    auto phase_schedule = [&]() {
        constexpr int PA = A3 ? 1 : 3, PP = A3 ? 3 : 6;                          // A planes, products per 32 x 32 x 16 tile
        constexpr int NMFMA = TM * TN * PP, NVALU = 30 * ((A3 ? 0 : NA) + NB) + (AFF == 2 ? 16 : 8 * AFF) * NA, NDSW = PA * NA + 3 * NB,
                      NVM = NA + NB + (AFF == 2 ? NA : 0);
        constexpr int VPM = (NVALU + NMFMA - 1) / NMFMA;
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * TN + PA, 0);            // DS read: B fragments + A fragment 0
#pragma unroll
        for (int g = 0; g < NMFMA; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);                // VALU of the split
            if (g * NDSW / NMFMA != (g + 1) * NDSW / NMFMA) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // DS write
            if (g * NVM / NMFMA != (g + 1) * NVM / NMFMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // VMEM read
            if (g % (PP * TN) == 2 && g / (PP * TN) + 1 < TM) __builtin_amdgcn_sched_group_barrier(0x100, PA, 0);   // next A fragment
        }
    };
    // ------------------------------------------------------------------ prologue
    if constexpr (AFF == 2) {
        for (int c = t_all; c < 4 * p.Kc; c += NT * (WS ? 2 : 1)) coef[c] = p.lin[c];
        __syncthreads();
    } else if constexpr (AFF) {
        for (int c = t_all; c < p.Kc; c += NT * (WS ? 2 : 1)) { coef[c] = sc_ptr ? sc_ptr[c] : 1.f; coef[p.Kc + c] = sh_ptr ? sh_ptr[c] : 0.f; }
        __syncthreads();
    }
    if constexpr (WS) {
        // every wave executes the same number of barriers (the branch is wave-uniform); producers leave after the loop -- a finished
        // wave no longer counts towards the block's barrier, and the epilogue has none
        if (producer) {
            Regs R0, R1;
            prepare_tap(q_begin);     load_half(q_begin, R0);
            prepare_tap(q_begin + 1); load_half(q_begin + 1, R1);
            store_half(R0, 0);
            prepare_tap(q_begin + 2); load_half(q_begin + 2, R0);
            __syncthreads();
            for (int j = 0; j < nq; j += 2) {
                prepare_tap(q_begin + j + 3);
                store_half(R1, 1);
                load_half(q_begin + j + 3, R1);
                __syncthreads();
                prepare_tap(q_begin + j + 4);
                store_half(R0, 0);
                load_half(q_begin + j + 4, R0);
                __syncthreads();
            }
            return;
        }
        __syncthreads();
        for (int j = 0; j < nq; j += 2) {
            compute_half(0);
            __syncthreads();
            compute_half(1);
            __syncthreads();
        }
    } else {
    Regs R0, R1;
    prepare_tap(q_begin);     load_half(q_begin, R0);
    prepare_tap(q_begin + 1); load_half(q_begin + 1, R1);
    store_half(R0, 0);
    prepare_tap(q_begin + 2); load_half(q_begin + 2, R0);
    __syncthreads();
    // ------------------------------------------------------------------ main loop: two phases (= one K=32 tile) per iteration
    for (int j = 0; j < nq; j += 2) {
        prepare_tap(q_begin + j + 3);
        store_half(R1, 1);                       // half-tile j+1 (zeros beyond the K range)
        load_half(q_begin + j + 3, R1);
        compute_half(0);                         // half-tile j
        phase_schedule();
        __syncthreads();
        prepare_tap(q_begin + j + 4);
        store_half(R0, 0);                       // half-tile j+2
        load_half(q_begin + j + 4, R0);
        compute_half(1);                         // half-tile j+1
        phase_schedule();
        __syncthreads();
    }
    }

    if constexpr (SK) {          // split-K: raw partial sums; bias / addend / BN statistics are applied by the reduce kernel
        float* slab = p.slab + (long long)blockIdx.y * p.M * p.ldslab;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 32 * TN + 32 * j + l31;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 32 * TM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < p.M && n < p.Nn) slab[(long long)m * p.ldslab + n] = acc[i][j][r];
                }
        }
        return;
    }

    // ------------------------------------------------------------------ epilogue
    static_assert(EPI == 0 || TM % 2 == 0, "statistics epilogues need whole 64-row groups per wave");
    int ncol[TN];
    bool nok[TN];
    float bias[TN];
    [[maybe_unused]] float b_mu[TN], b_is[TN], b_sc[TN], b_sh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        ncol[j] = n0 + wn * 32 * TN + 32 * j + l31;
        nok[j] = ncol[j] < p.Nn;
        bias[j] = (MODE == FPROP && p.extra != nullptr && nok[j]) ? p.extra[ncol[j]] : 0.f;
        if constexpr (EPI == 2) {
            b_mu[j] = b_is[j] = b_sc[j] = b_sh[j] = 0.f;
            if (nok[j]) { b_mu[j] = p.bnb_mean[ncol[j]]; b_is[j] = p.bnb_invstd[ncol[j]]; b_sc[j] = p.bnb_scale[ncol[j]]; b_sh[j] = p.bnb_shift[ncol[j]]; }
        }
    }
    // ---- row-major epilogue through LDS (the planes are free after the loop's last barrier).  An accumulator register holds ONE
    // column per lane, so a direct epilogue moves 4 bytes per lane and instruction (64 loads + 64 stores per lane for a 64 x 64
    // wave tile, more with an addend or the BatchNorm-backward y tile).  Each wave instead parks 32 rows of its tile in its own LDS
    // strip, reads them back as one float4 of a row per lane, and every global access (output, addend, y) is a 16-byte load /
    // store, 256 contiguous bytes per 16 lanes: a quarter of the memory instructions, all loads of a strip issued before its stores.
    if (EPI == 2 || p.vec_epi) {       // (the BatchNorm-backward epilogue exists in this form only)
        constexpr int WC = 32 * TN, RS = WC + 4, LPR = WC / 4, RPI = 64 / LPR, NRD = 32 / RPI;
        float* const stage = reinterpret_cast<float*>(smem) + wave * (32 * RS);
        const int rr = lane / LPR, c4 = (lane % LPR) * 4;
        const int ncv = n0 + wn * WC + c4;
        const bool nokv = ncv < p.Nn;
        f32x4 vb = zero4, v_mu = zero4, v_is = zero4, v_sc = zero4, v_sh = zero4;
        if (MODE == FPROP && p.extra != nullptr && nokv) vb = ld4(p.extra + ncv);
        if constexpr (EPI == 2) { if (nokv) { v_mu = ld4(p.bnb_mean + ncv); v_is = ld4(p.bnb_invstd + ncv); v_sc = ld4(p.bnb_scale + ncv); v_sh = ld4(p.bnb_shift + ncv); } }
        [[maybe_unused]] f32x4 pend_gs = zero4, pend_gq = zero4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * h) * RS + 32 * j + l31] = acc[i][j][r];
            // (r3) no block-wide barrier: the strip is private to the wave (LDS operations of one wave complete in order) and the planes it
            // overlays were released by the main loop's last barrier
            // the strip's NRD row groups in chunks: the BatchNorm-backward epilogue holds four tiles per row group (accumulator strip,
            // addend, y, out) next to the 64 accumulator registers -- all 8 groups at once spilled 85 registers
            constexpr int CH = (EPI == 2 && NRD > 4) ? 4 : NRD;
            [[maybe_unused]] f32x4 gs = zero4, gq = zero4;
#pragma unroll
            for (int k0 = 0; k0 < NRD; k0 += CH) {
            f32x4 v[CH];
            [[maybe_unused]] f32x4 ad[CH], yv[CH], ov[CH];
            long long mrow[CH];
#pragma unroll
            for (int kk = 0; kk < CH; ++kk) {
                const int row = (k0 + kk) * RPI + rr;
                mrow[kk] = m0 + wm * 32 * TM + 32 * i + row;
                v[kk] = *reinterpret_cast<const f32x4*>(stage + row * RS + c4);
                if constexpr (MODE == DGRAD) {
                    ad[kk] = zero4;
                    if (mrow[kk] < Mc && nokv) {
                        if (p.extra != nullptr) ad[kk] = ld4(p.extra + mrow[kk] * p.ldadd + ncv);
                        if constexpr (EPI == 2) {
                            yv[kk] = lda4(p.bnb_y, mrow[kk] * p.bnb_ldy + ncv, p.act & 4);
                            if (p.bnb_out != nullptr)
                                ov[kk] = (p.act & 64) ? quad_mask_load(p.bnb_out, mrow[kk] * p.bnb_ldo + (ncv >> 2)) : lda4(p.bnb_out, mrow[kk] * p.bnb_ldo + ncv, p.act & 8);
                        }
                    }
                }
            }
#pragma unroll
            for (int kk = 0; kk < CH; ++kk) {
                if (mrow[kk] < Mc && nokv) {
                    f32x4 o = v[kk];
                    if constexpr (MODE == FPROP) o += vb;
                    else {
                        o += ad[kk];
                        if constexpr (EPI == 2) {
                            if (p.bnb_relu) {
                                // the forward's own arithmetic (bn_act_kernel), or the stored block output where a residual was added
                                const f32x4 a = p.bnb_out != nullptr ? ov[kk] : yv[kk] * v_sc + v_sh;
#pragma unroll
                                for (int e = 0; e < 4; ++e) if (!(a[e] > 0.f)) o[e] = 0.f;
                            }
                            gs += o; gq += o * ((yv[kk] - v_mu) * v_is);
                        }
                    }
                    sta4(p.c, mrow[kk] * p.ldc + ncv, o, MODE == FPROP ? (p.act & 2) : 0);
                }
            }
            }
            if constexpr (EPI == 2) {      // column sums over the strip's 32 rows: across the RPI row groups of the wave
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int o2 = LPR; o2 < 64; o2 <<= 1) { gs[e] += __shfl_xor(gs[e], o2, 64); gq[e] += __shfl_xor(gq[e], o2, 64); }
                }
                if ((i & 1) == 0) { pend_gs = gs; pend_gq = gq; }
                else {
                    const int pidx = tile_m * (BM / 64) + wm * (TM / 2) + (i >> 1);
                    if (lane < LPR && pidx < p.n_partials && nokv) {
                        st4(p.partials + ((long long)pidx * 2 + 0) * p.Nn + ncv, pend_gs + gs);
                        st4(p.partials + ((long long)pidx * 2 + 1) * p.Nn + ncv, pend_gq + gq);
                    }
                }
            }
        }
    } else if constexpr (EPI != 2) {
    [[maybe_unused]] float pend_s[TN], pend_q[TN];          // EPI == 2: sums of the even tile of a 64-row group
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        [[maybe_unused]] float gs[TN], gq[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) { gs[j] = 0.f; gq[j] = 0.f; }
        // Two passes per 32-row tile: FIRST every load of the tile (addend, BatchNorm-backward y), THEN arithmetic and stores.  In one
        // fused loop each load would have to wait for the previous element's store (the output may alias the inputs as far as the
        // compiler can tell), i.e. 64 dependent round trips per lane -- measured: +28 % on the 512->512 @128^2 dgrad.
        long long orow[16];
        bool mok[16];
        [[maybe_unused]] float ad[16][TN], yv[16][TN];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 * TM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
            mok[r] = m < Mc;
            orow[r] = m;                                      // row of the output (and of addend / bnb_y) this element belongs to
            if constexpr (MODE == DGRAD) {
                if (p.parity) {
                    const int iwc = m % Wc, q2 = m / Wc, ihc = q2 % Hc, nb2 = q2 / Hc;
                    orow[r] = ((long long)nb2 * p.H + 2 * ihc + oy0) * p.W + 2 * iwc + ox0;
                } else if (p.scatter) {
                    const int ow = m % p.W, q = m / p.W, oh = q % p.H, nb = q / p.H;
                    orow[r] = (long long)(nb * p.sH + oh * p.sstride) * p.sW + ow * p.sstride;
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    ad[r][j] = 0.f;
                    if (mok[r] && nok[j]) {
                        if (p.scatter) ad[r][j] = p.c[orow[r] * p.ldc + ncol[j]];                 // accumulate into the existing gradient
                        else if (p.extra != nullptr) ad[r][j] = p.extra[orow[r] * p.ldadd + ncol[j]];
                        if constexpr (EPI == 2) yv[r][j] = p.bnb_y[orow[r] * p.bnb_ldy + ncol[j]];
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (mok[r] && nok[j]) {
                    float v = acc[i][j][r];
                    float* dst = p.c + orow[r] * p.ldc + ncol[j];
                    if constexpr (MODE == FPROP) {
                        *dst = v + bias[j];
                    } else {
                        v += ad[r][j];
                        if constexpr (EPI == 2) {
                            if (p.bnb_relu && !(yv[r][j] * b_sc[j] + b_sh[j] > 0.f)) v = 0.f;      // the forward's own arithmetic (bn_act_kernel)
                            gs[j] += v; gq[j] += v * ((yv[r][j] - b_mu[j]) * b_is[j]);
                        }
                        *dst = v;
                    }
                }
            }
        }
        if constexpr (EPI == 2) {      // (sum g, sum g*xhat) of this wave's 64-row groups, tiles summed in tile order
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float s1 = gs[j] + __shfl_xor(gs[j], 32, 64), q1 = gq[j] + __shfl_xor(gq[j], 32, 64);
                if ((i & 1) == 0) { pend_s[j] = s1; pend_q[j] = q1; }
                else {
                    const int pidx = tile_m * (BM / 64) + wm * (TM / 2) + (i >> 1);
                    if (h == 0 && pidx < p.n_partials && nok[j]) {
                        p.partials[((long long)pidx * 2 + 0) * p.Nn + ncol[j]] = pend_s[j] + s1;
                        p.partials[((long long)pidx * 2 + 1) * p.Nn + ncol[j]] = pend_q[j] + q1;
                    }
                }
            }
        }
    }
    }      // scalar epilogue
    if constexpr (MODE == FPROP && EPI == 1) {
        if (p.partials != nullptr) {
            const int wrow0 = m0 + wm * 32 * TM;
#pragma unroll
            for (int pr = 0; pr < TM / 2; ++pr) {
                const int prow0 = wrow0 + 64 * pr;
                const int npr = max(0, min(64, p.M - prow0));
                const int pidx = tile_m * (BM / 64) + wm * (TM / 2) + pr;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float ss = 0.f;
#pragma unroll
                    for (int i = 2 * pr; i < 2 * pr + 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ss += acc[i][j][r];      // rows >= M are exact zeros
                    ss += __shfl_xor(ss, 32, 64);
                    const float mean = npr > 0 ? ss / (float)npr : 0.f;
                    float qq = 0.f;
#pragma unroll
                    for (int i = 2 * pr; i < 2 * pr + 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = wrow0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                            const float dv = acc[i][j][r] - mean;
                            qq += (npr == 64 || row < p.M) ? dv * dv : 0.f;
                        }
                    qq += __shfl_xor(qq, 32, 64);
                    const int n = n0 + wn * 32 * TN + 32 * j + l31;
                    if (h == 0 && pidx < p.n_partials && n < p.Nn) {
                        p.partials[((long long)pidx * 2 + 0) * p.Nn + n] = ss;
                        p.partials[((long long)pidx * 2 + 1) * p.Nn + n] = qq;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- host side
template <int MODE, int TM, int TN, int WGM, int WGN, int WPS, int AFF, int EPI, int SK, int TAP, int GRP = 0, int ABF = 0, int WS = 0>
static int launch_x6p(ConvQ& p, hipStream_t st) {
    constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
    const size_t lds = 3 * (size_t)(BM + BN) * ROWB + (AFF == 2 ? 16 : AFF ? 8 : 0) * (size_t)p.Kc;
    if (lds > 160 * 1024) return SH_X6P_NO;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x6p_kernel<MODE, TM, TN, WGM, WGN, WPS, AFF, EPI, SK, TAP, GRP, ABF, WS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        attr_lds = 160 * 1024;
    }
    p.tiles_m = (int)sh_cdiv(p.M, BM);
    p.tiles_n = (int)sh_cdiv(p.Nn, BN);
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), p.parity ? 4u : (SK ? (unsigned)p.ksplit : 1u));
    // (Wave quantisation -- 1024 or 2048 blocks on 768 resident slots -- was tried against TWO blocks per CU, enforced by a larger dynamic-LDS
    // request, so that those grids run 2 or 4 full rounds: slower for every grid-size threshold, step 31.15 -> 31.16 / 31.3 / 31.4 ms.)
    conv_x6p_kernel<MODE, TM, TN, WGM, WGN, WPS, AFF, EPI, SK, TAP, GRP, ABF, WS><<<grid, 64 * WGM * WGN * (WS ? 2 : 1), lds, st>>>(p);
    return sh_launch_status();
}

// Tile choice for one (mode, hooks, tap mode) combination.  Measured on the step's shapes (tools/bench_conv.py, round 2): 128 x 128
// tiles on 4 waves (164 VGPRs, 48 KB LDS => 3 blocks per CU, which also overlaps one block's epilogue stores with its neighbours'
// MFMAs) match or beat 256 x 256 on 8 waves and 256 x 128 everywhere (sep1.pw 740 vs 743 vs 770 us; l2.conv3 65 vs 77 vs 80 us),
// so only two shapes are instantiated: 128 x 128, and 128 x 64 for N <= 64.
template <int MODE, int AFF, int EPI, int TAP, int ABF = 0>
static int pick_tile_x6p(ConvQ& p, hipStream_t st, int force) {
    (void)force;
    // AFF == 2 carries a second A-operand stream (16 more prefetch registers): at 3 blocks per CU (168 VGPRs) it spills 15 registers into
    // the main loop -- measured 7.65 vs 6.27 ms per step at 2 blocks per CU.  (Plain kernels run equally fast at 2 and 3 blocks per CU.)
    // The BatchNorm-backward epilogue (EPI == 2) spilled 85 registers until its row groups were processed in chunks; 128 x 64 tiles with
    // it still need 3 instead of 4 blocks per CU (18 spills at 128 VGPRs).
    constexpr int W1 = SH_W1(AFF, EPI), W2 = SH_W2(AFF, EPI);
#ifdef SH_X6P_WS       // experiment builds only (-DSH_X6P_WS, then SEGHIERO_X6P_WS=1): 4 consumer + 4 producer waves, 2 blocks per CU
    if (p.Nn > 64 && x6p_ws()) return launch_x6p<MODE, 2, 2, 2, 2, 2, AFF, EPI, 0, TAP, 0, ABF, 1>(p, st);
#endif
    if (p.Nn > 64) return launch_x6p<MODE, 2, 2, 2, 2, W1, AFF, EPI, 0, TAP, 0, ABF>(p, st);     // 128 x 128, 4 waves of 64 x 64, 3 blocks per CU
    return launch_x6p<MODE, 2, 1, 2, 2, W2, AFF, EPI, 0, TAP, 0, ABF>(p, st);                    // 128 x 64, 4 blocks per CU
}

template <int MODE, int AFF, int EPI, int ABF = 0>
static int pick_x6p(ConvQ& p, hipStream_t st, int force) {
    if constexpr (AFF == 2) {    // deferred BatchNorm-backward loader: 1x1 only
        if (p.KH * p.KW != 1) return SH_X6P_NO;
        if (p.ksplit > 1) {
            const int rc = launch_x6p<MODE, 2, 2, 2, 2, 2, 2, 0, 1, 0, 0, ABF>(p, st);
            return rc == SH_OK ? sh_x6_splitk_reduce(p, MODE, st) : rc;
        }
        return pick_tile_x6p<MODE, 2, EPI, 0, ABF>(p, st, force);
    } else {
    if (p.ksplit > 1) {          // K slices (mid-network shapes with an under-filled grid): 128 x 128 tiles into slabs, then the reduce
        int rc;
        if (p.KH * p.KW == 1) rc = launch_x6p<MODE, 2, 2, 2, 2, 3, AFF, 0, 1, 0, 0, ABF>(p, st);
        else if ((p.Kc & 15) == 0) rc = launch_x6p<MODE, 2, 2, 2, 2, 3, AFF, 0, 1, 1, 0, ABF>(p, st);
        else return SH_X6P_NO;
        return rc == SH_OK ? sh_x6_splitk_reduce(p, MODE, st) : rc;
    }
    if (p.KH * p.KW == 1) return pick_tile_x6p<MODE, AFF, EPI, 0, ABF>(p, st, force);
    if ((p.Kc & 15) == 0) return pick_tile_x6p<MODE, AFF, EPI, 1, ABF>(p, st, force);
    if (AFF || EPI == 2 || ABF) return SH_X6P_NO;
    return pick_tile_x6p<MODE, 0, EPI, 2>(p, st, force);
    }
}

#endif
#if SH_PART(1)

// ngroups 1x1 convolutions of one geometry in ONE launch (p.ngroups, p.group_n, p.ga / gb / gsc / gsh set by the caller): the ASPP
// branches of sep_aspp_contrast_head.py:100-131 -- grid = tiles_m x (ngroups * group_n / 128), so four under-filled 128-tile GEMMs
// become one 512-tile launch without K slices, each reading its own input through its own BatchNorm + ReLU
int sh_x6p_grouped_launch(ConvQ& p, hipStream_t st) {
    if (p.ngroups < 1 || p.ngroups > 6 || (p.group_n & 127) || p.KH * p.KW != 1) return SH_X6P_NO;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    p.vec_epi = (p.ldc & 3) == 0 && al16(p.c);
    return launch_x6p<FPROP, 2, 2, 2, 2, 3, 1, 1, 0, 0, 1>(p, st);
}

// Entry used by sh_conv_fprop_x6 / sh_conv_dgrad_x6: SH_X6P_NO = shape not handled here (the caller falls back to conv_x6_kernel)
int sh_x6p_launch(int mode, ConvQ& p, hipStream_t st) {
    const bool aff = p.aff_scale != nullptr, bnb = p.bnb_y != nullptr;
    if (!x6p_mode() && !aff && !bnb && !p.lin) return SH_X6P_NO;
    const int force = x6p_tile();
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    p.vec_epi = !p.parity && !p.scatter && (p.Nn & 3) == 0 && (p.ldc & 3) == 0 && al16(p.c) &&
                (p.extra == nullptr || (al16(p.extra) && (mode == FPROP || (p.ldadd & 3) == 0))) &&
                (!bnb || ((p.bnb_ldy & 3) == 0 && al16(p.bnb_y) && (p.bnb_out == nullptr || (p.act & 64) || ((p.bnb_ldo & 3) == 0 && al16(p.bnb_out))) && al16(p.bnb_mean) && al16(p.bnb_invstd) && al16(p.bnb_scale) &&
                          al16(p.bnb_shift) && al16(p.partials)));
    if (bnb && !p.vec_epi) return SH_X6P_NO;
    { static int v = -2; if (v == -2) { const char* e = getenv("SEGHIERO_X6P_VEC"); v = e ? atoi(e) : 1; } if (!v && !bnb) p.vec_epi = 0; }
    if ((p.act & 2) && !p.vec_epi) return SH_X6P_NO;          // a bf16 output is written by the row-major epilogue only
    if (mode == FPROP) {
        if (bnb) return SH_EINVAL;
        if (p.act & 1) return aff ? pick_x6p<FPROP, 1, 1, 1>(p, st, force) : pick_x6p<FPROP, 0, 1, 1>(p, st, force);      // bf16 input
        if (aff) return pick_x6p<FPROP, 1, 1>(p, st, force);
        return pick_x6p<FPROP, 0, 1>(p, st, force);
    }
    if (aff) return SH_EINVAL;
    return sh_x6p_launch_dgrad(p, st, force);
}
#endif
#if SH_PART(2)
int sh_x6p_launch_dgrad(ConvQ& p, hipStream_t st, int force) {
    const bool bnb = p.bnb_y != nullptr;
    if (p.lin != nullptr && (p.act & 16)) return bnb ? pick_x6p<DGRAD, 2, 2, 1>(p, st, force) : pick_x6p<DGRAD, 2, 0, 1>(p, st, force);
    if (p.lin != nullptr) return bnb ? pick_x6p<DGRAD, 2, 2>(p, st, force) : pick_x6p<DGRAD, 2, 0>(p, st, force);
    if (bnb) return pick_x6p<DGRAD, 0, 2>(p, st, force);
    return pick_x6p<DGRAD, 0, 0>(p, st, force);
}
#endif

#if SH_X6P_PART == 0 || SH_X6P_PART >= 3
// ============================================================================================ WGRAD, pipelined
// dW[co][n'] = sum_pix dY[pix][co] * im2col(X)[pix][n'] with the same two-half-tile pipeline as above: a K=32 tile of pixels is
// two K=16 halves = k-rows 0..15 / 16..31 of the [k][row] LDS planes of conv_wgrad_x6_kernel (transposing ds_read_b64_tr_b16
// fragments, (2*rows+64)-byte k-rows); per phase: split + store half-tile j+1, issue the loads of half-tile j+3, MFMAs of
// half-tile j, one barrier.  AFF: the X operand is relu(x * scale[c] + shift[c]) of the stored tensor (the producer's train-mode
// BatchNorm + ReLU, applied in the loader; see AFF above) -- each thread owns one 4-channel chunk, so its coefficients sit in
// registers for the whole kernel.
// LIN: the dY operand is lin(g, y) = A[co]*g + B[co]*(y - mean[co]) + D[co] of the masked gradient g (p.a) and the raw conv output y
// (p.a2) -- the deferred second half of this conv's own BatchNorm backward (see AFF == 2 above); a thread owns one 4-channel chunk
// of co for the whole kernel, so the four coefficient vectors sit in registers.
// XBF: the stored activations this kernel reads -- the X operand and, with LIN, the y stream -- are bf16 (see conv_x6p_kernel's ABF)
template <int WGM, int WGN, int AFF, int LIN = 0, int XBF = 0>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void conv_wgrad_x6p_kernel(const ConvQ p) {
    constexpr int NT = 64 * WGM * WGN, BM = 64 * WGM, BN = 64 * WGN;
    constexpr bool X3 = XBF && AFF == 0;          // X is a stored bf16 tensor read as is: three products (see conv_x6p_kernel's A3)
    constexpr int AROWB = 2 * BM + 64, BROWB = 2 * BN + 64;
    constexpr int APLANE = 32 * AROWB, BPLANE = 32 * BROWB;
    constexpr int ACPR = BM / 4, BCPR = BN / 4;              // float4 chunks per k-row
    constexpr int AKPP = NT / ACPR, BKPP = NT / BCPR;        // k-rows per loader pass
    constexpr int NA = 16 / AKPP, NB = 16 / BKPP;            // passes per half-tile
    static_assert(NA >= 1 && NB >= 1, "a loader pass must not straddle the two halves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const As = smem;
    unsigned char* const Bs = smem + 3 * APLANE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WGN, wn = wave % WGN, l31 = lane & 31, h = lane >> 5;
    const unsigned nblk = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
    unsigned slice, bid;                                     // block -> (K slice, tile): see conv_wgrad_x6_kernel
    if (p.scatter) { const unsigned kq = blockIdx.x >> 3; slice = (kq / nblk) * 8u + (blockIdx.x & 7u); bid = kq % nblk; }
    else { slice = blockIdx.x / nblk; bid = xcd_remap(blockIdx.x % nblk, nblk); }
    if (slice >= (unsigned)p.ksplit) return;
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = slice * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    const int nq = (((kend - kbeg + 15) >> 4) + 1) & ~1;    // half-tiles, rounded up to whole K=32 tiles

    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, p.b_bytes, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc_a2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(LIN ? p.a2 : p.a), 0, LIN ? p.a2_bytes : p.a_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int arc = t % ACPR, ak0 = t / ACPR;
    const int brc = t % BCPR, bk0 = t / BCPR;
    const int co = m0 + 4 * arc;
    const int nn = n0 + 4 * brc;
    const unsigned a_col_ok = co < p.M ? ~0u : 0u, b_col_ok = nn < p.Nn ? ~0u : 0u;
    const int tap = nn / p.Cin, wg_ci = nn - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int wg_dh = kh * p.dil - p.pad, wg_dw = kw * p.dil - p.pad;
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
    if constexpr (AFF) { if (b_col_ok) { sc = ld4(p.aff_scale + wg_ci); sh = ld4(p.aff_shift + wg_ci); } }
    [[maybe_unused]] f32x4 lA = sc, lB = sc, lM = sc, lD = sc;
    if constexpr (LIN) { lA = lB = lM = lD = f32x4{0.f, 0.f, 0.f, 0.f}; if (a_col_ok) { lA = ld4(p.lin + co); lB = ld4(p.lin + p.M + co); lM = ld4(p.lin + 2 * p.M + co); lD = ld4(p.lin + 3 * p.M + co); } }
    // pixel coordinates of this thread's B k-rows for the NEXT half-tile to load, advanced by 16 pixels per load (Wo >= 16 on
    // this path, so a step wraps at most one image row and one image: compare + select, no loops, no divisions)
    int px_ow[NB], px_oh[NB], px_n[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int pix = kbeg + bk0 + BKPP * i;
        px_ow[i] = pix % p.Wo; const int q2 = pix / p.Wo; px_oh[i] = q2 % p.Ho; px_n[i] = q2 / p.Ho;
    }
    auto wg_load4 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned elem_off, unsigned ok, bool bf) -> f32x4 {
        if (bf) {
            const unsigned voff = ((elem_off * 2u) & ok) | (OOB & ~ok);
            const u32x2 r = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0));
            return f32x4{__uint_as_float(r[0]), __uint_as_float(r[1]), 0.f, 0.f};
        }
        const unsigned voff = ((elem_off * 4u) & ok) | (OOB & ~ok);
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
    };
    auto wg_widen = [&](f32x4 v, bool bf) -> f32x4 { return bf ? bf16x4_to_f32(sh_u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}) : v; };
    struct Regs { f32x4 a[NA]; f32x4 a2[LIN ? NA : 1]; f32x4 b[NB]; unsigned okm, okl; };
    auto load_half = [&](int q, Regs& R) {
        const int kbase = kbeg + 16 * q;
        R.okm = 0; R.okl = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int pix = kbase + ak0 + AKPP * i;
            const unsigned ok = (unsigned)((pix - kend) >> 31) & a_col_ok;
            const unsigned voff = (((unsigned)(pix * (int)p.lda + co) * 4u) & ok) | (OOB & ~ok);
            R.a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff, 0, 0));
            if constexpr (LIN) {
                R.a2[i] = wg_load4(rsrc_a2, (unsigned)(pix * (int)p.lda2 + co), ok, XBF);
                R.okl |= ok & (1u << i);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int pix = kbase + bk0 + BKPP * i;
            const int ih = px_oh[i] * p.stride + wg_dh, iw = px_ow[i] * p.stride + wg_dw;
            const unsigned inimg = ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? ~0u : 0u;
            const unsigned ok = (unsigned)((pix - kend) >> 31) & b_col_ok & inimg;
            R.b[i] = wg_load4(rsrc_b, (unsigned)(((px_n[i] * p.H + ih) * p.W + iw) * (int)p.ldb + wg_ci), ok, XBF);
            if (AFF) R.okm |= ok & (1u << i);
            int ow = px_ow[i] + 16, oh = px_oh[i], n = px_n[i];
            const bool c1 = ow >= p.Wo; ow -= c1 ? p.Wo : 0; oh += c1 ? 1 : 0;
            const bool c2 = oh >= p.Ho; oh -= c2 ? p.Ho : 0; n += c2 ? 1 : 0;
            px_ow[i] = ow; px_oh[i] = oh; px_n[i] = n;
        }
    };
    auto store_half = [&](Regs& R, int hb) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int off = (16 * hb + ak0 + AKPP * i) * AROWB + arc * 8;
            u32x2 q1, q2, q3;
            f32x4 va = R.a[i];
            if constexpr (LIN) {      // pixels beyond the K slice stay exact zeros (D != 0)
                const bool ok = (R.okl >> i) & 1u;
                const f32x4 y2 = wg_widen(R.a2[i], XBF);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float w = fmaf(y2[e] - lM[e], lB[e], fmaf(va[e], lA[e], lD[e])); va[e] = ok ? w : 0.f; }
            }
            split4(va, q1, q2, q3);
            *reinterpret_cast<u32x2*>(As + off) = q1;
            *reinterpret_cast<u32x2*>(As + APLANE + off) = q2;
            *reinterpret_cast<u32x2*>(As + 2 * APLANE + off) = q3;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if constexpr (X3) {          // stored bf16 X read as is: the 8 loaded bytes are the hi plane's four elements, mid / lo planes are zero
                *reinterpret_cast<u32x2*>(Bs + (16 * hb + bk0 + BKPP * i) * BROWB + brc * 8) = u32x2{__float_as_uint(R.b[i][0]), __float_as_uint(R.b[i][1])};
                continue;
            }
            f32x4 v = wg_widen(R.b[i], XBF);
            if constexpr (AFF) {
                const bool ok = (R.okm >> i) & 1u;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float w = fmaxf(v[e] * sc[e] + sh[e], 0.f); v[e] = ok ? w : 0.f; }
            }
            const int off = (16 * hb + bk0 + BKPP * i) * BROWB + brc * 8;
            u32x2 q1, q2, q3;
            split4(v, q1, q2, q3);
            *reinterpret_cast<u32x2*>(Bs + off) = q1;
            *reinterpret_cast<u32x2*>(Bs + BPLANE + off) = q2;
            *reinterpret_cast<u32x2*>(Bs + 2 * BPLANE + off) = q3;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // transpose-read addressing: 16-lane group g -> (row half g&1, k half g>>1); lane 4q+pp supplies row q, columns 4pp..4pp+3
    const int g = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
    const int krow = (g >> 1) * 8 + qq, coff = (16 * (g & 1) + 4 * pp) * 2;
    auto frag = [&](const unsigned char* plane, int rowb, int hb, int rowbase) -> bf16x8 {
        const unsigned char* a0 = plane + (krow + hb * 16) * rowb + coff + rowbase * 2;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 4 * rowb));
        s16x8 r;
        r[0] = v0[0]; r[1] = v0[1]; r[2] = v0[2]; r[3] = v0[3]; r[4] = v1[0]; r[5] = v1[1]; r[6] = v1[2]; r[7] = v1[3];
        return __builtin_bit_cast(bf16x8, r);
    };
    auto compute_half = [&](int hb) {
        bf16x8 af[2][3], bfr[2][X3 ? 1 : 3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                af[i][pl] = frag(As + pl * APLANE, AROWB, hb, wm * 64 + 32 * i);
                if (!X3 || pl == 0) bfr[i][X3 ? 0 : pl] = frag(Bs + pl * BPLANE, BROWB, hb, wn * 64 + 32 * i);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (X3) acc[i][j] = mma3b(af[i], bfr[j][0], acc[i][j]);
                else acc[i][j] = mma6(af[i], bfr[j], acc[i][j]);
            }
    };
    auto phase_schedule = [&]() {
        constexpr int NMF = X3 ? 12 : 24, NDSR = X3 ? 8 : 12;                   // MFMAs per phase; fragment reads per batch (two batches)
        constexpr int NVALU = (X3 ? 2 : 30 + 8 * AFF) * NB + (30 + 16 * LIN) * NA + 8 * NB, NDSW = 3 * NA + (X3 ? 1 : 3) * NB, NVM = NA + NB + LIN * NA;
        constexpr int VPM = (NVALU + NMF - 1) / NMF;
        __builtin_amdgcn_sched_group_barrier(0x100, NDSR, 0);                   // DS read: first A / B fragment pairs
#pragma unroll
        for (int gq = 0; gq < NMF; ++gq) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);                // VALU of the split / addressing
            if (gq * NDSW / NMF != (gq + 1) * NDSW / NMF) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // DS write
            if (gq * NVM / NMF != (gq + 1) * NVM / NMF) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // VMEM read
            if (gq == 2) __builtin_amdgcn_sched_group_barrier(0x100, NDSR, 0);  // remaining fragment reads
        }
    };
    Regs R0, R1;
    load_half(0, R0);
    load_half(1, R1);
    store_half(R0, 0);
    load_half(2, R0);
    __syncthreads();
    for (int j = 0; j < nq; j += 2) {
        store_half(R1, 1);
        load_half(j + 3, R1);
        compute_half(0);
        phase_schedule();
        __syncthreads();
        store_half(R0, 0);
        load_half(j + 4, R0);
        compute_half(1);
        phase_schedule();
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
template <int WGM, int WGN, int AFF, int LIN = 0, int XBF = 0>
static int launch_wgrad_x6p(ConvQ& p, int splits, hipStream_t st) {
    constexpr size_t lds = 3 * 32 * (size_t)((2 * 64 * WGM + 64) + (2 * 64 * WGN + 64));
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_x6p_kernel<WGM, WGN, AFF, LIN, XBF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    p.tiles_m = (int)sh_cdiv(p.M, 64 * WGM); p.tiles_n = (int)sh_cdiv(p.Nn, 64 * WGN);
    p.ksplit = splits;
    const unsigned grid = (unsigned)(p.tiles_m * p.tiles_n) * (unsigned)(p.scatter ? sh_cdiv(splits, 8) * 8 : splits);
    conv_wgrad_x6p_kernel<WGM, WGN, AFF, LIN, XBF><<<grid, 64 * WGM * WGN, lds, st>>>(p);
    return sh_launch_status();
}
template <int AFF, int LIN = 0, int XBF = 0>
static int pick_wgrad_x6p(ConvQ& p, int wgm, int wgn, int splits, hipStream_t st) {
    if (wgm == 4 && wgn == 1) return launch_wgrad_x6p<4, 1, AFF, LIN, XBF>(p, splits, st);
    if (wgm == 2 && wgn == 1) return launch_wgrad_x6p<2, 1, AFF, LIN, XBF>(p, splits, st);
    if (wgm == 2 && wgn == 4) return launch_wgrad_x6p<2, 4, AFF, LIN, XBF>(p, splits, st);
    if (wgm == 1 && wgn == 4) return launch_wgrad_x6p<1, 4, AFF, LIN, XBF>(p, splits, st);
    if (wgm == 1 && wgn == 2) return launch_wgrad_x6p<1, 2, AFF, LIN, XBF>(p, splits, st);
    return launch_wgrad_x6p<2, 2, AFF, LIN, XBF>(p, splits, st);
}
// the four (LIN, XBF) families of instantiations, one translation unit each (parts 3-6)
#if SH_PART(3)
int sh_x6p_wgrad_l0b0(ConvQ& p, bool aff, int wgm, int wgn, int splits, hipStream_t st) {
    return aff ? pick_wgrad_x6p<1>(p, wgm, wgn, splits, st) : pick_wgrad_x6p<0>(p, wgm, wgn, splits, st);
}
// Entry used by sh_conv_wgrad_x6 (same tile / K-slice plan as conv_wgrad_x6_kernel; the slab reduce stays with the caller)
int sh_x6p_wgrad_launch(ConvQ& p, int wgm, int wgn, int splits, hipStream_t st) {
    const bool aff = p.aff_scale != nullptr;
    if (!x6p_mode() && !aff && !p.lin) return SH_X6P_NO;
    if (p.Wo < 16 || (p.Cin & 3)) return SH_X6P_NO;
    if (p.act & 32) {          // bf16 X operand; a lin y stream must then be bf16 too (the trunk: both are stored conv outputs)
        if (p.lin != nullptr && !(p.act & 16)) return SH_X6P_NO;
        return p.lin != nullptr ? sh_x6p_wgrad_l1b1(p, aff, wgm, wgn, splits, st) : sh_x6p_wgrad_l0b1(p, aff, wgm, wgn, splits, st);
    }
    if (p.lin != nullptr && (p.act & 16)) return SH_X6P_NO;
    return p.lin != nullptr ? sh_x6p_wgrad_l1b0(p, aff, wgm, wgn, splits, st) : sh_x6p_wgrad_l0b0(p, aff, wgm, wgn, splits, st);
}
#endif
#if SH_PART(4)
int sh_x6p_wgrad_l1b0(ConvQ& p, bool aff, int wgm, int wgn, int splits, hipStream_t st) {
    return aff ? pick_wgrad_x6p<1, 1>(p, wgm, wgn, splits, st) : pick_wgrad_x6p<0, 1>(p, wgm, wgn, splits, st);
}
#endif
#if SH_PART(5)
int sh_x6p_wgrad_l0b1(ConvQ& p, bool aff, int wgm, int wgn, int splits, hipStream_t st) {
    return aff ? pick_wgrad_x6p<1, 0, 1>(p, wgm, wgn, splits, st) : pick_wgrad_x6p<0, 0, 1>(p, wgm, wgn, splits, st);
}
#endif
#if SH_PART(6)
int sh_x6p_wgrad_l1b1(ConvQ& p, bool aff, int wgm, int wgn, int splits, hipStream_t st) {
    return aff ? pick_wgrad_x6p<1, 1, 1>(p, wgm, wgn, splits, st) : pick_wgrad_x6p<0, 1, 1>(p, wgm, wgn, splits, st);
}
#endif
#endif

#ifdef SH_X6P_EXP      // schedule experiments: ONE instantiation, device assembly only (hipcc -S --cuda-device-only -DSH_X6P_PART=-1 -DSH_X6P_EXP=...)
template __global__ void conv_x6p_kernel<SH_X6P_EXP>(const ConvQ);
#endif
