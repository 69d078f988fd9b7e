// fp32-accurate convolution GEMMs on the bf16 matrix cores ("6 x bf16 split").
//
// gfx950's f32 MFMA runs at the vector rate (157 TF); its bf16 MFMA is 16x faster.  Every fp32 operand x is split
// exactly into three bf16 terms x = x1 + x2 + x3 (8 significand bits each, by truncation: x1 = hi16(x), x2 = hi16(x-x1),
// x3 = hi16(x-x1-x2); the subtractions are exact in fp32), and a product is evaluated as
//     a*b ~= a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1          (dropped terms <= 3 * 2^-24 |ab|)
// with six v_mfma_f32_32x32x16_bf16 accumulating in fp32: 192 MFMA cycles per K=16 instead of 512 for the f32 MFMA,
// at fp32-class accuracy (validated by the same parity tests as the f32 kernels, tests/test_ops_gpu.py).
//
// Structure: same implicit-GEMM tiling / loaders / epilogues as conv_gemm.hip.  Global fp32 -> registers -> split ->
// three bf16 LDS planes per operand ([row][32 k], 64-byte rows with an XOR-swizzled 16-byte chunk: conflict-free reads).  FPROP and
// DGRAD have K-contiguous operands (DGRAD reads a per-step transposed weight copy [tap][ci][co]); WGRAD's operands are
// pixel-major, so its planes are stored [k][m] and the fragments are read with ds_read_b64_tr_b16 (hardware transpose).
#include "conv_x6.h"

// ============================================================================================ FPROP / DGRAD
// Block = WGM x WGN waves, each wave TM x TN MFMA tiles of 32x32: block tile (32*TM*WGM) x (32*TN*WGN).
// Measured (tools/bench_conv.py + PMC): with 128x128 tiles these kernels are bound by the CU's L1 line-access rate
// (every K tile re-reads (BM+BN)*128 B through the vector L1), not by MFMA or VALU issue -- so the big shapes use
// 256x256 tiles on 1024 threads (half the L1 lines per flop), with both operands kept fp32 in memory (4 B/element) and
// split to bf16 in registers on their way to LDS.
// PROF: instrumented build for tools/bench_conv.py (SEGHIERO_X6_PROF=1): per-wave s_memtime totals of the main-loop phases
// OCC: waves per SIMD the register budget is held to (0 = one resident block's worth, at least 2)
// ACT: inference epilogue out = [relu](acc * scale[n] + shift[n] [+ residual]) -- eval-mode BatchNorm (+ residual add + ReLU)
//      fused into the convolution, in bn_act_kernel's own operation order (bit-identical to the unfused eval path)
template <int MODE, int TM, int TN, int WGM, int WGN, int OCC = 0, int SK = 0, int PROF = 0, int ACT = 0>      // SK: split-K instantiation (K slice = blockIdx.y)
__global__ __launch_bounds__(64 * WGM * WGN, OCC ? OCC : ((WGM * WGN) / 4 < 2 ? 2 : (WGM * WGN) / 4)) void conv_x6_kernel(const ConvQ p) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN, BK = 32;
    constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;      // bytes
    constexpr int RPP = NT / 8;                                  // rows covered per loader pass (8 lanes x 16 B per row)
    constexpr int NA = (BM + RPP - 1) / RPP, NB = (BN + RPP - 1) / RPP;      // the last pass may be partial (12-wave blocks)
    static_assert(BM >= RPP && BN >= RPP, "tile too small for the thread count");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WGN, wn = wave % WGN, l31 = lane & 31, h = lane >> 5;
    unsigned long long tk0 = 0;
    if constexpr (PROF) tk0 = __builtin_amdgcn_s_memtime();
    const unsigned nblk = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
    const unsigned bid = xcd_remap(blockIdx.x, nblk);
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // stride-2 dgrad by parity class: an input pixel (ih, iw) only sees taps with kh == (ih+pad) mod 2 (same for kw), so the
    // four classes run as separate GEMMs over a quarter of the pixels with 1/2/2/4 of the 9 taps instead of 9 zero-filled ones
    int cy = 0, cx = 0, Hc = p.H, Wc = p.W, oy0 = 0, ox0 = 0, nth = p.KH, ntw = p.KW, Mc = p.M, Kt = p.K;
    if constexpr (MODE == DGRAD) {
        if (p.parity) {
            cy = blockIdx.y >> 1; cx = blockIdx.y & 1;
            oy0 = (cy + p.pad) & 1; ox0 = (cx + p.pad) & 1;          // first input row / column with (i + pad) % 2 == c
            Hc = (p.H - oy0 + 1) >> 1; Wc = (p.W - ox0 + 1) >> 1;
            nth = (p.KH - cy + 1) >> 1; ntw = (p.KW - cx + 1) >> 1;
            Mc = p.N * Hc * Wc; Kt = nth * ntw * p.Kc;
            if (m0 >= Mc) return;                                    // block-uniform, before any barrier
        }
    }
    const int nkt = (Kt + BK - 1) / BK;
    int kt_begin = 0, kt_end = nkt;
    if constexpr (SK) {                                              // never combined with the parity classes (both use blockIdx.y)
        const int per = (nkt + p.ksplit - 1) / p.ksplit;
        kt_begin = min(nkt, (int)blockIdx.y * per); kt_end = min(nkt, kt_begin + per);
    }
    const bool single_tap = p.KH * p.KW == 1;
    const bool tap_uniform = single_tap || (p.Kc & 31) == 0;     // a K tile never straddles taps => tap math is scalar

    const int kc = t & 7, r0 = t >> 3;
    int a_y[NA], a_x[NA], a_nb[NA];
    long long b_row[NB];             // B rows: fprop W[co][K] (k linear over taps); dgrad Wt[tap][ci][Kc] (per-tap rows of Kc floats)
    bool b_ok[NB];
    int cur_tap = -1;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + r0 + RPP * i;
        if (m < Mc && (BM % RPP == 0 || r0 + RPP * i < BM)) {
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
        b_ok[i] = j < p.Nn && (BN % RPP == 0 || r0 + RPP * i < BN);
        b_row[i] = (long long)j * (MODE == FPROP ? p.K : p.Kc);
    }
    long long a_off[NA];             // element offset of each A row for the current tap (never recomputed for 1x1 convs)
    bool a_ok[NA];
    auto set_tap = [&](int tap) {
        int kh, kw;
        if (MODE == DGRAD && p.parity) { const int ty = tap / ntw; kh = cy + 2 * ty; kw = cx + 2 * (tap - ty * ntw); }
        else { kh = tap / p.KW; kw = tap - kh * p.KW; }
        const int dh = kh * p.dil, dw = kw * p.dil;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if constexpr (MODE == FPROP) {
                const int ih = a_y[i] + dh, iw = a_x[i] + dw;
                a_ok[i] = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                a_off[i] = (long long)(a_nb[i] + ih * p.W + iw) * p.lda;
            } else {
                int th = a_y[i] - dh, tw = a_x[i] - dw;
                bool ok = th >= 0 && tw >= 0;
                if (p.stride > 1) {
                    ok = ok && (th % p.stride == 0) && (tw % p.stride == 0);
                    th /= p.stride; tw /= p.stride;
                }
                a_ok[i] = ok && th < p.Ho && tw < p.Wo;
                a_off[i] = (long long)(a_nb[i] + th * p.Wo + tw) * p.lda;
            }
        }
    };
    f32x4 ra[NA], rb[NB];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_tile = [&](int kt) {
        const int kbase = kt * BK;
        int tap, cc;
        if (single_tap) { tap = 0; cc = kbase + 4 * kc; }
        else if (tap_uniform) { tap = kbase / p.Kc; cc = kbase - tap * p.Kc + 4 * kc; }
        else { const int k = kbase + 4 * kc; tap = k / p.Kc; cc = k - tap * p.Kc; }
        const bool kok = (kbase + 4 * kc) < Kt;
        if (!tap_uniform || tap != cur_tap) { set_tap(tap); cur_tap = tap; }
        int wtap = tap;                                           // tap index into the weight tensor
        if (MODE == DGRAD && p.parity) { const int ty = tap / ntw; wtap = (cy + 2 * ty) * p.KW + cx + 2 * (tap - ty * ntw); }
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = (kok && a_ok[i]) ? ld4(p.a + a_off[i] + cc) : zero4;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            long long off;
            if constexpr (MODE == FPROP) off = b_row[i] + kbase + 4 * kc;
            else off = (long long)wtap * p.Cin * p.Kc + b_row[i] + cc;
            rb[i] = (kok && b_ok[i]) ? ld4(p.b + off) : zero4;
        }
    };
    const int wr_off = ((((kc >> 1) ^ ((t >> 5) & 3))) << 4) + ((kc & 1) << 3);      // swizzled chunk + half of this thread's 8 bytes
    auto store_tile = [&]() {
        unsigned char* As = smem;
        unsigned char* Bs = As + 3 * A_PLANE;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (BM % RPP != 0 && r0 + RPP * i >= BM) continue;
            u32x2 q1, q2, q3;
            split4(ra[i], q1, q2, q3);
            const int off = (r0 + RPP * i) * ROWB + wr_off;
            *reinterpret_cast<u32x2*>(As + off) = q1;
            *reinterpret_cast<u32x2*>(As + A_PLANE + off) = q2;
            *reinterpret_cast<u32x2*>(As + 2 * A_PLANE + off) = q3;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (BN % RPP != 0 && r0 + RPP * i >= BN) continue;
            u32x2 q1, q2, q3;
            split4(rb[i], q1, q2, q3);
            const int off = (r0 + RPP * i) * ROWB + wr_off;
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
    const int rd_off[2] = {((h ^ ((l31 >> 2) & 3)) << 4), (((2 + h) ^ ((l31 >> 2) & 3)) << 4)};   // swizzled k-chunk for s = 0, 1
    auto compute = [&]() {
        const unsigned char* As = smem;
        const unsigned char* Bs = As + 3 * A_PLANE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 bfr[TN][3];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    bfr[j][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bs + pl * B_PLANE + (brow + 32 * j) * ROWB + rd_off[s]));
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                bf16x8 af[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    af[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(As + pl * A_PLANE + (arow + 32 * i) * ROWB + rd_off[s]));
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mma6(af, bfr[j], acc[i][j]);
            }
        }
    };
    unsigned long long tc = 0, tb1 = 0, tv = 0, ts = 0, tb2 = 0, tloop = 0, tpro = 0;      // PROF only
    if (!SK || kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile();
    }
    __syncthreads();
    if constexpr (PROF) {
        const unsigned long long tstart = __builtin_amdgcn_s_memtime();
        tpro = tstart - tk0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            if (kt + 1 < kt_end) load_tile(kt + 1);
            compute();
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            __syncthreads();
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            tc += t1 - t0; tb1 += t2 - t1;
            if (kt + 1 < kt_end) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long t3 = __builtin_amdgcn_s_memtime();
                store_tile();
                const unsigned long long t4 = __builtin_amdgcn_s_memtime();
                __syncthreads();
                const unsigned long long t5 = __builtin_amdgcn_s_memtime();
                tv += t3 - t2; ts += t4 - t3; tb2 += t5 - t4;
            }
        }
        tloop = __builtin_amdgcn_s_memtime() - tstart;
    } else
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if (kt + 1 < kt_end) load_tile(kt + 1);
        compute();
        __syncthreads();                      // every wave is done reading the LDS planes
        if (kt + 1 < kt_end) {
            store_tile();
            __syncthreads();
        }
    }

    if constexpr (SK) {          // split-K: raw partial sums; bias / addend / BN statistics are applied by splitk_reduce_kernel
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

    // ------------------------------------------------------------------ epilogue (same as conv_gemm.hip)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * 32 * TN + 32 * j + l31;
        const bool nok = n < p.Nn;
        float bias = 0.f, a_sc = 1.f, a_sh = 0.f;
        if constexpr (MODE == FPROP) bias = (p.extra != nullptr && nok) ? p.extra[n] : 0.f;
        if constexpr (ACT) { if (nok) { a_sc = p.act_scale[n]; a_sh = p.act_shift[n]; } }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * TM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < Mc && nok) {
                    if constexpr (MODE == FPROP && ACT) {
                        float v = acc[i][j][r] * a_sc + a_sh;
                        if (p.act_res != nullptr) v += p.act_res[(long long)m * p.ldadd + n];
                        if (p.act_relu) v = fmaxf(v, 0.f);
                        p.c[(long long)m * p.ldc + n] = v;
                    } else if constexpr (MODE == FPROP) {
                        p.c[(long long)m * p.ldc + n] = acc[i][j][r] + bias;
                    } else {
                        float v = acc[i][j][r];
                        long long mo = m;                         // output pixel row
                        if (p.parity) {
                            const int iwc = m % Wc, q2 = m / Wc, ihc = q2 % Hc, nb2 = q2 / Hc;
                            mo = ((long long)nb2 * p.H + 2 * ihc + oy0) * p.W + 2 * iwc + ox0;
                        }
                        if (p.extra != nullptr) v += p.extra[mo * p.ldadd + n];
                        if (p.parity) { p.c[mo * p.ldc + n] = v; continue; }
                        if (p.scatter) {
                            const int ow = m % p.W, q = m / p.W, oh = q % p.H, nb = q / p.H;
                            float* dst = p.c + ((long long)(nb * p.sH + oh * p.sstride) * p.sW + ow * p.sstride) * p.ldc + n;
                            *dst += v;
                        } else {
                            p.c[(long long)m * p.ldc + n] = v;
                        }
                    }
                }
            }
    }
    if constexpr (MODE == FPROP) {
        if (p.partials != nullptr) {
            const int wrow0 = m0 + wm * 32 * TM;
            const int nw = max(0, min(32 * TM, p.M - wrow0));
            float s[TN], q[TN];
            if constexpr (TM % 2 == 0) {      // a wave covers TM/2 whole 64-row partials
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
            } else {                          // TM == 1 (only instantiated with WGM == 2): two waves share the 64-row partial
                static_assert(TM % 2 == 0 || WGM == 2, "TM == 1 variants use a 2-wave M grid");
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float ss = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) ss += acc[0][j][r];
                    ss += __shfl_xor(ss, 32, 64);
                    const float mean = nw > 0 ? ss / (float)nw : 0.f;
                    float qq = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float dv = acc[0][j][r] - mean;
                        qq += (nw == 32 || row < p.M) ? dv * dv : 0.f;
                    }
                    qq += __shfl_xor(qq, 32, 64);
                    s[j] = ss; q[j] = qq;
                }
                float* red = reinterpret_cast<float*>(smem);      // main loop ended with a barrier
                if (h == 0) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int c = wn * 32 * TN + 32 * j + l31;
                        red[(wm * 2 + 0) * BN + c] = s[j];
                        red[(wm * 2 + 1) * BN + c] = q[j];
                    }
                }
                __syncthreads();
                if (t < BN) {
                    const int n = n0 + t;
                    if (n < p.Nn) {
                        const float na = (float)max(0, min(32, p.M - m0)), nb = (float)max(0, min(32, p.M - m0 - 32));
                        const float sa = red[t], sb = red[2 * BN + t];
                        float m2 = red[BN + t] + red[3 * BN + t];
                        if (nb > 0.f) { const float dm = sa / na - sb / nb; m2 += dm * dm * na * nb / (na + nb); }
                        p.partials[((long long)tile_m * 2 + 0) * p.Nn + n] = sa + sb;
                        p.partials[((long long)tile_m * 2 + 1) * p.Nn + n] = m2;
                    }
                }
            }
        }
    }
    if constexpr (PROF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tk1 = __builtin_amdgcn_s_memtime();
        if (lane == 0 && bid < 64) {
            unsigned long long* d = reinterpret_cast<unsigned long long*>(p.slab) + ((long long)bid * (WGM * WGN) + wave) * 8;
            d[0] = tc; d[1] = tb1; d[2] = tv; d[3] = ts; d[4] = tb2; d[5] = tloop; d[6] = (unsigned long long)(kt_end - kt_begin); d[7] = tpro;
            reinterpret_cast<unsigned long long*>(p.slab)[65536 + (long long)bid * (WGM * WGN) + wave] = tk1 - tk0;
        }
    }
}

// ============================================================================================ 1x1 / stride-1 fast path
// C[M][N] = A[M][K] * B[N][K]^T for pointwise convolutions (fprop: A = x, B = W; dgrad: A = dy, B = Wt) -- ~60 % of the
// conv flops of the step.  Software-pipelined, one block (4 waves, 128x128 tile) per CU with two LDS stages:
//   iteration kt:  read the fragments of tile kt from stage kt&1 -> registers;
//                  48 MFMAs interleaved (sched_group_barrier) with { split tile kt+1 (registers) -> other stage,
//                  issue the global loads of tile kt+2 };   one barrier.
// The steady-state loop body is ONE basic block (branch-free clamped loads, no tap arithmetic) so that hipcc can place
// the VALU / LDS-write / VMEM work in the shadow of the MFMAs.
template <int MODE>
__global__ __launch_bounds__(256, 1) void gemm_x6_kernel(const ConvQ p) {
    constexpr int BM = 128, BN = 128, BK = 32;
    constexpr int A_PLANE = BM * ROWB_G, B_PLANE = BN * ROWB_G, STAGE = 3 * (A_PLANE + B_PLANE);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const unsigned nblk = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
    const unsigned bid = xcd_remap(blockIdx.x, nblk);
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nkt = (p.K + BK - 1) / BK;
    const int kc = t & 7, r0 = t >> 3;

    // row pointers (clamped to the last valid row: out-of-range rows compute garbage that the epilogue never stores)
    const float* ap[4];
    const float* bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = min(m0 + r0 + 32 * i, p.M - 1), n = min(n0 + r0 + 32 * i, p.Nn - 1);
        ap[i] = p.a + (long long)m * p.lda + 4 * kc;
        bp[i] = p.b + (long long)n * p.ldb + 4 * kc;
    }
    f32x4 ra[4], rb[4];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_full = [&]() {                 // tile completely inside K
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra[i] = ld4(ap[i]); rb[i] = ld4(bp[i]); ap[i] += BK; bp[i] += BK; }
    };
    auto load_tail = [&](int kt) {           // last tile: mask k >= K (no out-of-bounds reads)
        const bool ok = kt * BK + 4 * kc < p.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra[i] = ok ? ld4(ap[i]) : zero4; rb[i] = ok ? ld4(bp[i]) : zero4; ap[i] += BK; bp[i] += BK; }
    };
    auto load_any = [&](int kt) { if ((kt + 1) * BK <= p.K) load_full(); else load_tail(kt); };
    auto store_tile = [&](int sb) {
        unsigned char* As = smem + sb * STAGE;
        unsigned char* Bs = As + 3 * A_PLANE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = (r0 + 32 * i) * ROWB_G + kc * 8;
            u32x2 q1, q2, q3;
            split4(ra[i], q1, q2, q3);
            *reinterpret_cast<u32x2*>(As + off) = q1;
            *reinterpret_cast<u32x2*>(As + A_PLANE + off) = q2;
            *reinterpret_cast<u32x2*>(As + 2 * A_PLANE + off) = q3;
            split4(rb[i], q1, q2, q3);
            *reinterpret_cast<u32x2*>(Bs + off) = q1;
            *reinterpret_cast<u32x2*>(Bs + B_PLANE + off) = q2;
            *reinterpret_cast<u32x2*>(Bs + 2 * B_PLANE + off) = q3;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int arow = wm * 64 + l31, brow = wn * 64 + l31;
    bf16x8 af[2][2][3], bfr[2][2][3];
    auto read_frags = [&](int sb) {
        const unsigned char* Ac = smem + sb * STAGE;
        const unsigned char* Bc = Ac + 3 * A_PLANE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    af[s2][i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ac + pl * A_PLANE + (arow + 32 * i) * ROWB_G + (2 * s2 + h) * 16));
                    bfr[s2][i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bc + pl * B_PLANE + (brow + 32 * i) * ROWB_G + (2 * s2 + h) * 16));
                }
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma6(af[s2][i], bfr[s2][j], acc[i][j]);
    };

    load_any(0);
    store_tile(0);
    if (nkt > 1) load_any(1);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nkt && (kt + 3) * BK <= p.K; ++kt) {       // steady state: tiles kt+1 (registers) and kt+2 (full) exist
        const int cur = kt & 1;
        read_frags(cur);
        store_tile(cur ^ 1);
        load_full();
        mfmas();
#pragma unroll
        for (int gq = 0; gq < 48; ++gq) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);      // 5 VALU
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // 1 LDS write
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
        }
        __syncthreads();
    }
    for (; kt < nkt; ++kt) {                                    // drain (and the masked K tail)
        const int cur = kt & 1;
        read_frags(cur);
        if (kt + 1 < nkt) {
            store_tile(cur ^ 1);
            if (kt + 2 < nkt) load_any(kt + 2);
        }
        mfmas();
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + 32 * j + l31;
        const bool nok = n < p.Nn;
        float bias = 0.f;
        if constexpr (MODE == FPROP) bias = (p.extra != nullptr && nok) ? p.extra[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.M && nok) {
                    float v = acc[i][j][r];
                    if constexpr (MODE == FPROP) v += bias;
                    else if (p.extra != nullptr) v += p.extra[(long long)m * p.ldadd + n];
                    p.c[(long long)m * p.ldc + n] = v;
                }
            }
    }
    if constexpr (MODE == FPROP) {
        if (p.partials != nullptr) {
            const int wrow0 = m0 + wm * 64;
            const int nw = max(0, min(64, p.M - wrow0));
            const int pidx = tile_m * 2 + wm;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float ss = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wrow0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                        ss += (nw == 64 || row < p.M) ? acc[i][j][r] : 0.f;          // clamped rows hold garbage: mask them
                    }
                ss += __shfl_xor(ss, 32, 64);
                const float mean = nw > 0 ? ss / (float)nw : 0.f;
                float qq = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wrow0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float dv = acc[i][j][r] - mean;
                        qq += (nw == 64 || row < p.M) ? dv * dv : 0.f;
                    }
                qq += __shfl_xor(qq, 32, 64);
                const int n = n0 + wn * 64 + 32 * j + l31;
                if (h == 0 && pidx < p.n_partials && n < p.Nn) {
                    p.partials[((long long)pidx * 2 + 0) * p.Nn + n] = ss;
                    p.partials[((long long)pidx * 2 + 1) * p.Nn + n] = qq;
                }
            }
        }
    }
}

// ============================================================================================ WGRAD (transposing LDS reads)
// dW[co][n'] = sum_pix dY[pix][co] * im2col(X)[pix][n'];  planes stored [k = pixel][row] with (2*rows + 64)-byte rows
// (=> the four k-rows of a ds_read_b64_tr_b16 block land on disjoint banks).  Block = WGM x WGN waves of 64x64.
template <int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN) / 4 < 2 ? 2 : (WGM * WGN) / 4) void conv_wgrad_x6_kernel(const ConvQ p) {
    constexpr int NT = 64 * WGM * WGN, BM = 64 * WGM, BN = 64 * WGN, BK = 32;
    constexpr int AROWB = 2 * BM + 64, BROWB = 2 * BN + 64;
    constexpr int APLANE = BK * AROWB, BPLANE = BK * BROWB;
    constexpr int ACPR = BM / 4, BCPR = BN / 4;              // float4 chunks per k-row
    constexpr int AKPP = NT / ACPR, BKPP = NT / BCPR;        // k-rows per loader pass
    constexpr int NA = BK / AKPP, NB = BK / BKPP;
    static_assert(NA >= 1 && NB >= 1, "tile too narrow for the thread count");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Bs = smem + 3 * APLANE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WGN, wn = wave % WGN, l31 = lane & 31, h = lane >> 5;
    // Block -> (K slice, tile): all tiles of one K slice read the same pixel range of x and dY, so a slice lives on ONE XCD
    // (hardware block b runs on XCD b % 8): slice = 8 * (k / tiles) + b % 8 with k = b / 8, tile = k % tiles.  Eight slices are in
    // flight on the eight L2s; the tiles of a slice start together and walk K in step, so x / dY come from HBM once per slice
    // instead of once per XCD (measured by PMC before this mapping: 21.9 GB fetched per step for ~12.5 GB of operands).
    // Shapes with many tiles (p.scatter == 0 here) keep tiles spread over the XCDs: slice = b / tiles.
    const unsigned nblk = (unsigned)p.tiles_m * (unsigned)p.tiles_n;
    unsigned slice, bid;
    if (p.scatter) { const unsigned kq = blockIdx.x >> 3; slice = (kq / nblk) * 8u + (blockIdx.x & 7u); bid = kq % nblk; }
    else { slice = blockIdx.x / nblk; bid = xcd_remap(blockIdx.x % nblk, nblk); }
    if (slice >= (unsigned)p.ksplit) return;                 // the per-XCD grid is padded to a multiple of 8 slices (block-uniform exit)
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = slice * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    const int nkt = (kend - kbeg + BK - 1) / BK;

    const int arc = t % ACPR, ak0 = t / ACPR;        // A: float4 column chunk, first k-row
    const int brc = t % BCPR, bk0 = t / BCPR;
    const int co = m0 + 4 * arc;
    const int nn = n0 + 4 * brc;
    const bool wg_ok = nn < p.Nn;
    const int tap = nn / p.Cin, wg_ci = nn - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int wg_dh = kh * p.dil - p.pad, wg_dw = kw * p.dil - p.pad;

    f32x4 ra[NA], rb[NB];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // pixel coordinates of this thread's B k-rows, advanced by 32 pixels per K tile (no divisions in the loop)
    int px_ow[NB], px_oh[NB], px_n[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int pix = kbeg + bk0 + BKPP * i;
        px_ow[i] = pix % p.Wo; const int q2 = pix / p.Wo; px_oh[i] = q2 % p.Ho; px_n[i] = q2 / p.Ho;
    }
    auto load_tile = [&](int kt) {
        const int kbase = kbeg + kt * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int pix = kbase + ak0 + AKPP * i;
            ra[i] = (pix < kend && co < p.M) ? ld4(p.a + (long long)pix * p.lda + co) : zero4;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int pix = kbase + bk0 + BKPP * i;
            const int ih = px_oh[i] * p.stride + wg_dh, iw = px_ow[i] * p.stride + wg_dw;
            const bool ok = pix < kend && wg_ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            rb[i] = ok ? ld4(p.b + ((long long)(px_n[i] * p.H + ih) * p.W + iw) * p.ldb + wg_ci) : zero4;
            px_ow[i] += BK;
            while (px_ow[i] >= p.Wo) { px_ow[i] -= p.Wo; ++px_oh[i]; }
            while (px_oh[i] >= p.Ho) { px_oh[i] -= p.Ho; ++px_n[i]; }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int off = (ak0 + AKPP * i) * AROWB + arc * 8;
            u32x2 q1, q2, q3;
            split4(ra[i], q1, q2, q3);
            *reinterpret_cast<u32x2*>(As + off) = q1;
            *reinterpret_cast<u32x2*>(As + APLANE + off) = q2;
            *reinterpret_cast<u32x2*>(As + 2 * APLANE + off) = q3;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int off = (bk0 + BKPP * i) * BROWB + brc * 8;
            u32x2 q1, q2, q3;
            split4(rb[i], q1, q2, q3);
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
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int krow = (g >> 1) * 8 + q, coff = (16 * (g & 1) + 4 * pp) * 2;
    auto frag = [&](const unsigned char* plane, int rowb, int s, int rowbase) -> bf16x8 {
        const unsigned char* a0 = plane + (krow + s * 16) * rowb + coff + rowbase * 2;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 4 * rowb));
        s16x8 r;
        r[0] = v0[0]; r[1] = v0[1]; r[2] = v0[2]; r[3] = v0[3]; r[4] = v1[0]; r[5] = v1[1]; r[6] = v1[2]; r[7] = v1[3];
        return __builtin_bit_cast(bf16x8, r);
    };

    if (nkt > 0) { load_tile(0); store_tile(); }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[2][3], bfr[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    af[i][pl] = frag(As + pl * APLANE, AROWB, s, wm * 64 + 32 * i);
                    bfr[i][pl] = frag(Bs + pl * BPLANE, BROWB, s, wn * 64 + 32 * i);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma6(af[i], bfr[j], acc[i][j]);
        }
        __syncthreads();
        if (kt + 1 < nkt) { store_tile(); __syncthreads(); }
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

// slab reduce (same as conv_gemm.hip)
__global__ __launch_bounds__(256) void slab_reduce_x6_kernel(const float* __restrict__ slab, float* __restrict__ dw, long long n4, long long n, int S) {
    __shared__ f32x4 red[4][64];
    const int t = threadIdx.x, cl = t & 63, g = t >> 6;
    const long long i = (long long)blockIdx.x * 64 + cl;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
#pragma unroll 4
        for (int k = g; k < S; k += 4) s += ld4(slab + (long long)k * n + 4 * i);
    }
    red[g][cl] = s;
    __syncthreads();
    if (t < 64 && i < n4) st4(dw + 4 * i, (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]));
}

// Wt[tap][ci][CoutP] <- W[co][tap][ci]   (zero padded to CoutP = pad4(Cout))
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int T, int Cin, int CoutP) {
    const long long total = (long long)T * Cin * CoutP;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int co = (int)(i % CoutP);
        const long long q = i / CoutP;
        const int ci = (int)(q % Cin), tap = (int)(q / Cin);
        wt[i] = co < Cout ? w[((long long)co * T + tap) * Cin + ci] : 0.f;
    }
}
extern "C" int sh_weight_transpose(const float* w, float* wt, int Cout, int KH, int KW, int Cin, void* stream) {
    if (!w || !wt || Cout <= 0 || KH <= 0 || KW <= 0 || Cin <= 0) return SH_EINVAL;
    const int CoutP = (Cout + 3) & ~3;
    long long total = (long long)KH * KW * Cin * CoutP, g = sh_cdiv(total, 256);
    if (g > 4096) g = 4096;
    weight_transpose_kernel<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(w, wt, Cout, KH * KW, Cin, CoutP);
    return sh_launch_status();
}

// The same for up to SH_WT_MAX weights in one launch (the step transposes all ~65 conv weights once, before backward).
struct WtTab {
    const float* w[SH_WT_MAX];
    float* wt[SH_WT_MAX];
    int cout[SH_WT_MAX], taps[SH_WT_MAX], cin[SH_WT_MAX];
    long long start[SH_WT_MAX + 1];          // prefix sums of the 32x32 tile counts
    int n;
};
// one block = one 32 (co) x 32 (ci) tile of one tap of one weight, staged through LDS so that both the reads (ci fastest)
// and the writes (co fastest) are coalesced; T.start[] counts tiles
__global__ __launch_bounds__(256) void weight_transpose_multi_kernel(const WtTab T) {
    __shared__ float tile[32][33];
    const long long b = blockIdx.x;
    int lo = 0, hi = T.n - 1;                // binary search the owning tensor
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (T.start[mid] <= b) lo = mid; else hi = mid - 1; }
    const int Cout = T.cout[lo], Cin = T.cin[lo], taps = T.taps[lo], CoutP = (Cout + 3) & ~3;
    const int tco = (CoutP + 31) / 32, tci = (Cin + 31) / 32;
    long long e = b - T.start[lo];
    const int ci0 = (int)(e % tci) * 32; e /= tci;
    const int co0 = (int)(e % tco) * 32;
    const int tap = (int)(e / tco);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* w = T.w[lo];
    float* wt = T.wt[lo];
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < Cout && ci < Cin) ? w[((long long)co * taps + tap) * Cin + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < CoutP) wt[((long long)tap * Cin + ci) * CoutP + co] = tile[tx][r];
    }
}
extern "C" int sh_weight_transpose_multi(int n, const float* const* w, float* const* wt, const int* cout, const int* taps, const int* cin,
                                         void* stream) {
    if (n <= 0 || n > SH_WT_MAX || !w || !wt || !cout || !taps || !cin) return SH_EINVAL;
    WtTab T;
    T.n = n;
    long long acc = 0;
    for (int i = 0; i < n; ++i) {
        if (!w[i] || !wt[i] || cout[i] <= 0 || taps[i] <= 0 || cin[i] <= 0) return SH_EINVAL;
        T.w[i] = w[i]; T.wt[i] = wt[i]; T.cout[i] = cout[i]; T.taps[i] = taps[i]; T.cin[i] = cin[i];
        T.start[i] = acc;
        acc += (long long)taps[i] * sh_cdiv((cout[i] + 3) & ~3, 32) * sh_cdiv(cin[i], 32);
    }
    T.start[n] = acc;
    if (acc >= (1ll << 31)) return SH_EINVAL;
    weight_transpose_multi_kernel<<<(unsigned)acc, 256, 0, (hipStream_t)stream>>>(T);
    return sh_launch_status();
}

// Split-K reduce for fprop / dgrad: out[m][n] = sum_s slab[s][m][n] (+ bias[n] | + addend[m][n]); for fprop also the BN
// statistics of the conv epilogue (centred (sum, M2) per 64 rows); for dgrad optionally the front half of the producer layer's
// BatchNorm backward (BNB: g = relumask * dx stored, (sum g, sum g*xhat) per 64 rows -> partials, see conv_x6p.hip).
// Block = 64 rows x 64 columns, thread = 4 rows x 4 columns.
struct BnbQ { const float* y; long long ldy; const float* mean; const float* invstd; const float* scale; const float* shift; int relu; const float* out; long long ldo;
              int act; int out_bf, add_bf; };      // act: ConvQ::act (bit 2 y, 3 out stored as bf16); out_bf / add_bf: the output / addend are bf16 tensors
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, long long ldslab, int S, long long M, int Nn,
                                                            const float* __restrict__ bias, const float* __restrict__ addend, long long ldadd,
                                                            float* __restrict__ out, long long ldc, float* __restrict__ partials, const BnbQ bnb) {
    __shared__ float red[16][64];
    __shared__ float colmean[64];
    const int t = threadIdx.x, cq = t & 15, rl = t >> 4;
    const int n = blockIdx.y * 64 + cq * 4;
    const long long row0 = (long long)blockIdx.x * 64;
    const int npr = (int)(M - row0 < 64 ? M - row0 : 64);
    f32x4 v[4];
    const bool nok = n < Nn;                 // Nn % 4 == 0 on this path
    f32x4 gq = {0.f, 0.f, 0.f, 0.f};
    f32x4 b_mu = gq, b_is = gq, b_sc = gq, b_sh = gq;
    if (bnb.y != nullptr && nok) { b_mu = ld4(bnb.mean + n); b_is = ld4(bnb.invstd + n); b_sc = ld4(bnb.scale + n); b_sh = ld4(bnb.shift + n); }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long m = row0 + rl + 16 * k;
        v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (nok && m < M) {
            for (int s2 = 0; s2 < S; ++s2) v[k] += ld4(slab + ((long long)s2 * M + m) * ldslab + n);
            f32x4 o = v[k];
            if (bias) o += ld4(bias + n);
            if (addend) o += lda4(addend, m * ldadd + n, bnb.add_bf);
            if (bnb.y != nullptr) {
                const f32x4 yv = lda4(bnb.y, m * bnb.ldy + n, bnb.act & 4);
                if (bnb.relu) {
                    // the forward's own arithmetic (bn_act_kernel), or the stored block output where a residual was added
                    const f32x4 a = bnb.out == nullptr ? yv * b_sc + b_sh
                                    : ((bnb.act & 64) ? quad_mask_load(bnb.out, m * bnb.ldo + (n >> 2)) : lda4(bnb.out, m * bnb.ldo + n, bnb.act & 8));
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (!(a[j] > 0.f)) o[j] = 0.f;
                }
                v[k] = o;                                                // statistics of g, not of the raw sum
                gq += o * ((yv - b_mu) * b_is);
            }
            sta4(out, m * ldc + n, o, bnb.out_bf);       // (the statistics below use the fp32 sums)
        }
    }
    if (partials == nullptr) return;         // block-uniform
    f32x4 ssum = (v[0] + v[1]) + (v[2] + v[3]);           // rows >= M contribute exact zeros
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][cq * 4 + j] = ssum[j];
    __syncthreads();
    if (t < 64) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += red[k][t];
        colmean[t] = a / (float)npr;
        if (blockIdx.y * 64 + t < Nn) partials[((long long)blockIdx.x * 2 + 0) * Nn + blockIdx.y * 64 + t] = a;
    }
    __syncthreads();
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    if (bnb.y != nullptr) q = gq;                              // BNB: plain sum of g * xhat
    else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (row0 + rl + 16 * k < M) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = v[k][j] - colmean[cq * 4 + j]; q[j] += d * d; }
            }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][cq * 4 + j] = q[j];
    __syncthreads();
    if (t < 64 && blockIdx.y * 64 + t < Nn) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += red[k][t];
        partials[((long long)blockIdx.x * 2 + 1) * Nn + blockIdx.y * 64 + t] = a;
    }
}
// reduce launch shared with conv_x6p.hip
int sh_x6_splitk_reduce(const ConvQ& p, int mode, hipStream_t st) {
    dim3 rg((unsigned)sh_cdiv(p.M, 64), (unsigned)sh_cdiv(p.Nn, 64));
    BnbQ bnb{p.bnb_y, p.bnb_ldy, p.bnb_mean, p.bnb_invstd, p.bnb_scale, p.bnb_shift, p.bnb_relu, p.bnb_out, p.bnb_ldo, mode == FPROP ? p.act : (p.act & ~2),
             mode == FPROP ? (p.act & 2) : (p.act & 128), mode == DGRAD ? (p.act & 256) : 0};
    splitk_reduce_kernel<<<rg, 256, 0, st>>>(p.slab, p.ldslab, p.ksplit, p.M, p.Nn, mode == FPROP ? p.extra : nullptr,
                                             mode == DGRAD ? p.extra : nullptr, p.ldadd, p.c, p.ldc,
                                             (mode == FPROP || p.bnb_y != nullptr) ? p.partials : nullptr, bnb);
    return sh_launch_status();
}
// K slices for an under-filled grid: the mid-network shapes (M*N small, K long) give < 2 blocks per CU with 128x128 tiles.
static int splitk_plan(long long M, long long N, long long K, int parity, int scatter, int b16 = 0) {
    if (parity || scatter || (N & 3)) return 1;
    const long long tiles = sh_cdiv(M, 128) * sh_cdiv(N, 128), nkt = sh_cdiv(K, 32);
    static int target6 = 0, minkt6 = 0, target1 = 0, minkt1 = 0;
    if (!target6) {
        const char* e = getenv("SEGHIERO_SK_TARGET"); target6 = e ? atoi(e) : 512; e = getenv("SEGHIERO_SK_MINKT"); minkt6 = e ? atoi(e) : 16;
        // bf16 compute kernels (one MFMA product: the loop is 3-6x shorter, the fp32 slabs and their reduce cost the same)
        // -- measured on the configs[4] shape and on the headline shape in bf16 mode (same box, alternating): without K slices the
        // under-filled mid-network grids leave CUs to the weight gradients on the side stream, with them 2 ms of slab reduces per step:
        // 26.6-28.7 vs 27.7 ms and 19.06 vs 19.27 ms -- off by default (SEGHIERO_B16_SK_TARGET=512 restores the fp32-accurate plan)
        e = getenv("SEGHIERO_B16_SK_TARGET"); target1 = e ? atoi(e) : 1; e = getenv("SEGHIERO_B16_SK_MINKT"); minkt1 = e ? atoi(e) : 16;
    }
    const int target = b16 ? target1 : target6, minkt = b16 ? minkt1 : minkt6;
    if (tiles >= (target * 3) / 4 || nkt < 2 * minkt) return 1;
    long long S = sh_cdiv(target, tiles);
    if (S > nkt / minkt) S = nkt / minkt;    // >= minkt K tiles per slice
    if (S > 8) S = 8;
    return S < 2 ? 1 : (int)S;
}
// ---------------------------------------------------------------------------------------- host side
template <int MODE, int TM, int TN, int WGM, int WGN, int OCC = 0, int SK = 0, int PROF = 0, int ACT = 0>
static int launch_x6(ConvQ& p, hipStream_t st) {
    constexpr int BM = 32 * TM * WGM, BN = 32 * TN * WGN;
    constexpr size_t lds = 3 * (size_t)(BM + BN) * ROWB;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x6_kernel<MODE, TM, TN, WGM, WGN, OCC, SK, PROF, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    p.tiles_m = (int)sh_cdiv(p.M, BM);
    p.tiles_n = (int)sh_cdiv(p.Nn, BN);
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n), p.parity ? 4u : (SK ? (unsigned)p.ksplit : 1u));
    conv_x6_kernel<MODE, TM, TN, WGM, WGN, OCC, SK, PROF, ACT><<<grid, 64 * WGM * WGN, lds, st>>>(p);
    if (SK) return sh_x6_splitk_reduce(p, MODE, st);
    return sh_launch_status();
}
static int x6_variant() { static int v = -1; if (v < 0) { const char* e = getenv("SEGHIERO_X6_VARIANT"); v = e ? atoi(e) : 0; } return v; }
template <int MODE>
static int launch_gemm_x6(ConvQ& p, hipStream_t st) {
    constexpr size_t lds = 2 * 3 * (size_t)(128 + 128) * ROWB_G;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x6_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    p.tiles_m = (int)sh_cdiv(p.M, 128);
    p.tiles_n = (int)sh_cdiv(p.Nn, 128);
    gemm_x6_kernel<MODE><<<(unsigned)(p.tiles_m * p.tiles_n), 256, lds, st>>>(p);
    return sh_launch_status();
}
template <int MODE>
static int launch_conv_x6(ConvQ& p, hipStream_t st) {
    const long long M = p.M, N = p.Nn;
    if (p.ksplit > 1) return launch_x6<MODE, 2, 2, 2, 2, 0, 1>(p, st);
    // pointwise, stride 1 (a plain NT GEMM) with a long K: software-pipelined kernel (1 block per CU).  Measured
    // (tools/bench_conv.py): it wins for K >= 1024 (1024->256 @32^2: 56 vs 76 us) and loses for short K, where its
    // un-overlapped prologue / epilogue dominate (64->256 @128^2: 221 vs 136 us), so short-K shapes keep the 2-blocks/CU form.
    if (x6_variant() != 3 && p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0 && !p.scatter && N >= 96 && p.K >= 1024 &&
        sh_cdiv(M, 128) * sh_cdiv(N, 128) >= 192) {
        p.ldb = (MODE == FPROP) ? p.K : p.Kc;
        return launch_gemm_x6<MODE>(p, st);
    }
    // big tiles while they still give >= 2 blocks per CU (they halve the L1 traffic per flop), then the 128/64 family
    // (measured: 8 waves of 128x64 per 256x256 block -- half the LDS fragment reads per MFMA -- run at the same speed as
    //  16 waves of 64x64, and its dgrad instantiation spills; the 16-wave form is used for both)
    if (N > 128 && sh_cdiv(M, 256) * sh_cdiv(N, 256) >= 512) {
        // a 192-wide tile when 256 would pad N by > 10 % more (N = 560, the decoder's concat width: 576 vs 768 columns)
        if (sh_cdiv(N, 192) * 192 * 10 < sh_cdiv(N, 256) * 256 * 9) return launch_x6<MODE, 2, 2, 4, 3>(p, st);
        // (measured: 12 waves of 192x256 instead: dgrad +8 % on 512->512 @128^2, fprop equal or worse -- not used)
        if (MODE == FPROP && x6_variant() == 7 && p.slab) return launch_x6<FPROP, 2, 2, 4, 4, 0, 0, 1>(p, st);      // instrumented
        // (measured: two 256x128 blocks per CU at 128 VGPRs are 20 % slower -- spills; three 128x128 blocks per CU: no change;
        //  a persistent tile loop that requests the next tile's first K tile before the epilogue: slower main loop, more spills;
        //  single-tap (1x1) and 32-bit-offset instantiations of this 128-VGPR kernel: no fewer in-loop scratch reloads, slower;
        //  a 24-register fragment schedule (one A + one B triple live, five triple loads per K half): 4-7 % slower;
        //  timing-only swap of each 32x32x16 MFMA for two 16x16x32: +21-26 % here but <= 5 % on the spill-free 128x128 kernel,
        //  and a real 16x16 tiling needs twice the LDS fragment reads per flop -- not pursued)
        return launch_x6<MODE, 2, 2, 4, 4>(p, st);
    }
    if (N > 64 && sh_cdiv(M, 256) * sh_cdiv(N, 128) >= 512) return launch_x6<MODE, 2, 2, 4, 2>(p, st);
    // >= 2 blocks per CU where the shape allows it (measured: layer3/4 shapes gain 10-20 % over 1 block per CU)
    int TN = N <= 64 ? 1 : 2, TM = 2;
    if (sh_cdiv(M, 128) * sh_cdiv(N, 64 * TN) < 512) {
        TM = 1;
        if (TN == 2 && sh_cdiv(M, 64) * sh_cdiv(N, 128) < 512) TN = 1;
    }
    if (M <= 64) TM = 1;
    if (MODE == FPROP && TM == 2 && TN == 2 && x6_variant() == 7 && p.slab) return launch_x6<FPROP, 2, 2, 2, 2, 0, 0, 1>(p, st);
    if (TM == 2 && TN == 2) return launch_x6<MODE, 2, 2, 2, 2>(p, st);
    if (TM == 2 && TN == 1) return launch_x6<MODE, 2, 1, 2, 2>(p, st);
    if (TM == 1 && TN == 2) return launch_x6<MODE, 1, 2, 2, 2>(p, st);
    return launch_x6<MODE, 1, 1, 2, 2>(p, st);
}
// inference forward with the fused BN / residual / ReLU epilogue: the same tile rule on a reduced family (no split-K)
static int launch_conv_x6_act(ConvQ& p, hipStream_t st) {
    const long long M = p.M, N = p.Nn;
    if (N > 128 && sh_cdiv(M, 256) * sh_cdiv(N, 256) >= 512) return launch_x6<FPROP, 2, 2, 4, 4, 0, 0, 0, 1>(p, st);
    int TN = N <= 64 ? 1 : 2, TM = 2;
    if (sh_cdiv(M, 128) * sh_cdiv(N, 64 * TN) < 512) {
        TM = 1;
        if (TN == 2 && sh_cdiv(M, 64) * sh_cdiv(N, 128) < 512) TN = 1;
    }
    if (M <= 64) TM = 1;
    if (TM == 2 && TN == 2) return launch_x6<FPROP, 2, 2, 2, 2, 0, 0, 0, 1>(p, st);
    if (TM == 2 && TN == 1) return launch_x6<FPROP, 2, 1, 2, 2, 0, 0, 0, 1>(p, st);
    if (TM == 1 && TN == 2) return launch_x6<FPROP, 1, 2, 2, 2, 0, 0, 0, 1>(p, st);
    return launch_x6<FPROP, 1, 1, 2, 2, 0, 0, 0, 1>(p, st);
}
static bool geom(ConvQ& p, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || dil <= 0 || pad < 0) return false;
    if (Cin % 4 != 0) return false;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
    p.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
    p.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
    if (p.Ho <= 0 || p.Wo <= 0) return false;
    if ((long long)N * H * W >= (1ll << 31) || (long long)N * p.Ho * p.Wo >= (1ll << 31)) return false;
    if ((long long)KH * KW * (long long)(Cin > Cout ? Cin : Cout + 3) >= (1ll << 30)) return false;
    return true;
}

// bytes of split-K workspace sh_conv_fprop_x6 (which = 0) / sh_conv_dgrad_x6 (which = 1, with its `mode`) can use; 0 = no split
extern "C" int64_t sh_conv_x6_workspace(int which, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, int mode) {
    ConvQ p{};
    if (!geom(p, N, H, W, Cin, Cout, KH, KW, stride, pad, dil)) return -1;
    long long M, Nn, K;
    int parity = 0;
    if (which == 0) { M = (long long)N * p.Ho * p.Wo; Nn = Cout; K = (long long)KH * KW * Cin; }
    else {
        if (mode != 0) return 0;
        M = (long long)N * H * W; Nn = Cin; K = (long long)KH * KW * ((Cout + 3) & ~3);
        parity = stride == 2 && dil == 1 && KH * KW > 1;
    }
    const int S = splitk_plan(M, Nn, K, parity, 0);
    return S > 1 ? (int64_t)S * M * Nn * 4 : 0;
}
static void use_splitk(ConvQ& p, float* workspace, int64_t workspace_bytes, int b16 = 0) {
    if (x6_variant() == 7 && workspace && workspace_bytes >= (1 << 20)) { p.slab = workspace; return; }      // phase-timing buffer
    const int S = splitk_plan(p.M, p.Nn, p.K, p.parity, p.scatter, b16);
    if (S > 1 && workspace && workspace_bytes >= (int64_t)S * p.M * p.Nn * 4) { p.ksplit = S; p.slab = workspace; p.ldslab = p.Nn; }
}
// byte extents of the operands for the buffer descriptors of the pipelined kernels; false = too large for 32-bit offsets
static bool operand_extents(ConvQ& p, int mode) {
    long long a, b;
    if (mode == FPROP) { a = ((long long)p.N * p.H * p.W - 1) * p.lda + p.Cin; b = (long long)p.Cout * p.K; }
    else { a = ((long long)p.N * p.Ho * p.Wo - 1) * p.lda + p.Kc; b = (long long)p.KH * p.KW * p.Cin * p.Kc; }
    if (a * 4 >= (1ll << 31) || b * 4 >= (1ll << 31)) return false;
    p.a_bytes = (unsigned)(a * ((mode == FPROP && (p.act & 1)) ? 2 : 4)); p.b_bytes = (unsigned)(b * 4);
    return true;
}
// pipelined kernel (conv_x6p.hip) where it has an instantiation for the shape, else conv_x6_kernel
static int launch_fprop_any(ConvQ& p, hipStream_t st) {
    if (operand_extents(p, FPROP)) { const int rc = sh_x6p_launch(FPROP, p, st); if (rc != SH_X6P_NO) return rc; }
    if (p.act) return SH_EUNSUPPORTED;          // bf16 tensors: the pipelined kernels only
    return launch_conv_x6<FPROP>(p, st);
}
static int launch_dgrad_any(ConvQ& p, hipStream_t st) {
    if (operand_extents(p, DGRAD)) { const int rc = sh_x6p_launch(DGRAD, p, st); if (rc != SH_X6P_NO) return rc; }
    return launch_conv_x6<DGRAD>(p, st);
}
extern "C" int sh_conv_fprop_x6(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, float* stat_partials,
                                int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil,
                                float* workspace, int64_t workspace_bytes, int act_flags, void* stream) {
    ConvQ p{};
    if (!x || !w || !y || !geom(p, N, H, W, Cin, Cout, KH, KW, stride, pad, dil)) return SH_EINVAL;
    if (ldx < Cin || ldy < Cout || (ldx & 3) || (act_flags & ~3)) return SH_EINVAL;
    p.act = act_flags;                                   // bit 0: x, bit 1: y stored as bf16
    p.a = x; p.b = w; p.c = y; p.extra = bias; p.partials = stat_partials; p.lda = ldx; p.ldc = ldy;
    p.M = N * p.Ho * p.Wo; p.Nn = Cout; p.K = KH * KW * Cin; p.Kc = Cin;
    p.n_partials = (int)sh_cdiv(p.M, 64);
    if ((ldy & 3) == 0 && ((uintptr_t)y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0)) use_splitk(p, workspace, workspace_bytes);
    return launch_fprop_any(p, (hipStream_t)stream);
}
// The same convolution reading its input through the producer's BatchNorm + ReLU: x holds the RAW output of the previous
// convolution and the loader applies relu(x * in_scale[c] + in_shift[c]) on the way to LDS (zero padding stays zero), so the
// activated tensor of conv -> BN -> ReLU -> conv chains (models/backbone/resnet.py:65-73, sep_aspp_contrast_head.py:56-61,180-184)
// is never materialised.  SH_EUNSUPPORTED: this geometry has no fused instantiation (caller applies sh_bn_act first).
extern "C" int sh_conv_fprop_x6_aff(const float* x, int ldx, const float* in_scale, const float* in_shift, const float* w, const float* bias,
                                    float* y, int ldy, float* stat_partials, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                    int stride, int pad, int dil, float* workspace, int64_t workspace_bytes, int act_flags, void* stream) {
    ConvQ p{};
    if (!x || !w || !y || !in_scale || !in_shift || !geom(p, N, H, W, Cin, Cout, KH, KW, stride, pad, dil)) return SH_EINVAL;
    if (ldx < Cin || ldy < Cout || (ldx & 3) || (((uintptr_t)in_scale | (uintptr_t)in_shift) & 15) || (act_flags & ~3)) return SH_EINVAL;
    p.act = act_flags;                                   // bit 0: x, bit 1: y stored as bf16
    p.a = x; p.b = w; p.c = y; p.extra = bias; p.partials = stat_partials; p.lda = ldx; p.ldc = ldy;
    p.M = N * p.Ho * p.Wo; p.Nn = Cout; p.K = KH * KW * Cin; p.Kc = Cin;
    p.n_partials = (int)sh_cdiv(p.M, 64);
    p.aff_scale = in_scale; p.aff_shift = in_shift;
    if ((ldy & 3) == 0 && ((uintptr_t)y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0)) use_splitk(p, workspace, workspace_bytes);
    if (!operand_extents(p, FPROP)) return SH_EUNSUPPORTED;
    const int rc = sh_x6p_launch(FPROP, p, (hipStream_t)stream);
    return rc == SH_X6P_NO ? SH_EUNSUPPORTED : rc;
}
extern "C" int sh_conv_fprop_x6_act(const float* x, int ldx, const float* w, const float* scale, const float* shift, const float* residual,
                                    int ldr, int relu, float* out, int ldo, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                    int stride, int pad, int dil, void* stream) {
    ConvQ p{};
    if (!x || !w || !out || !scale || !shift || !geom(p, N, H, W, Cin, Cout, KH, KW, stride, pad, dil)) return SH_EINVAL;
    if (ldx < Cin || ldo < Cout || (ldx & 3) || (residual && ldr < Cout)) return SH_EINVAL;
    p.a = x; p.b = w; p.c = out; p.lda = ldx; p.ldc = ldo;
    p.act_scale = scale; p.act_shift = shift; p.act_res = residual; p.ldadd = ldr; p.act_relu = relu;
    p.M = N * p.Ho * p.Wo; p.Nn = Cout; p.K = KH * KW * Cin; p.Kc = Cin;
    return launch_conv_x6_act(p, (hipStream_t)stream);
}
// ngroups (<= 6) pointwise convolutions of ONE geometry in one launch: group g computes y[:, g*Cout : (g+1)*Cout] = conv1x1(in_g, w[g])
// where in_g = relu(x[g] * in_scale[g] + in_shift[g]) (the producer's train-mode BatchNorm + ReLU in the loader) or plain x[g] when
// in_scale[g] is NULL.  The ASPP branches of models/head/sep_aspp_contrast_head.py:100-131 (1x1 branch + the three depthwise-
// separable branches' pointwise convs) are four 4096 x 512 x 2048 GEMMs that each fill half the chip; grouped they are one 512-tile
// launch that needs no K slices.  y: first group's column (ldy >= ngroups*Cout); stat_partials: [ceil(M/64)][2][ngroups*Cout].
extern "C" int sh_conv1x1_grouped_fprop_x6(int ngroups, const float* const* x, const int* ldx, const float* const* in_scale,
                                           const float* const* in_shift, const float* const* w, float* y, int ldy, float* stat_partials,
                                           int N, int H, int W, int Cin, int Cout, void* stream) {
    ConvQ p{};
    if (ngroups < 1 || ngroups > 6 || !x || !ldx || !in_scale || !in_shift || !w || !y || !geom(p, N, H, W, Cin, Cout, 1, 1, 1, 0, 1)) return SH_EINVAL;
    if ((Cout & 127) || (Cin & 15) || ldy < ngroups * Cout) return SH_EUNSUPPORTED;
    const long long M = (long long)N * H * W;
    for (int g = 0; g < ngroups; ++g) {
        if (!x[g] || !w[g] || ldx[g] < Cin || (ldx[g] & 3) || ((in_scale[g] == nullptr) != (in_shift[g] == nullptr))) return SH_EINVAL;
        if ((((uintptr_t)in_scale[g] | (uintptr_t)in_shift[g]) & 15)) return SH_EINVAL;
        const long long ab = ((M - 1) * ldx[g] + Cin) * 4;
        if (ab >= (1ll << 31)) return SH_EUNSUPPORTED;
        p.ga[g] = x[g]; p.ga_bytes[g] = (unsigned)ab; p.glda[g] = ldx[g]; p.gb[g] = w[g]; p.gsc[g] = in_scale[g]; p.gsh[g] = in_shift[g];
    }
    p.ngroups = ngroups; p.group_n = Cout;
    p.c = y; p.ldc = ldy; p.partials = stat_partials;
    p.M = (int)M; p.Nn = ngroups * Cout; p.K = Cin; p.Kc = Cin;
    p.b_bytes = (unsigned)((long long)Cout * Cin * 4);
    p.n_partials = (int)sh_cdiv(p.M, 64);
    const int rc = sh_x6p_grouped_launch(p, (hipStream_t)stream);
    return rc == SH_X6P_NO ? SH_EUNSUPPORTED : rc;
}
// wt = sh_weight_transpose(w): fp32 [KH*KW][Cin][pad4(Cout)]
extern "C" int sh_conv_dgrad_x6(const float* dy, int lddy, const float* wt, const float* addend, int ldadd, float* dx, int lddx,
                                int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, int mode,
                                float* workspace, int64_t workspace_bytes, void* stream) {
    ConvQ p{};
    if (!dy || !wt || !dx || !geom(p, N, H, W, Cin, Cout, KH, KW, stride, pad, dil)) return SH_EINVAL;
    const int CoutP = (Cout + 3) & ~3;
    if (lddy < CoutP || (lddy & 3) || lddx < Cin) return SH_EINVAL;
    if (addend && ldadd < Cin) return SH_EINVAL;
    p.a = dy; p.b = wt; p.c = dx; p.extra = addend; p.ldadd = ldadd; p.lda = lddy; p.ldc = lddx;
    p.Nn = Cin; p.Kc = CoutP; p.K = KH * KW * CoutP;
    if (mode == 1) {
        if (KH != 1 || KW != 1 || pad != 0) return SH_EINVAL;
        p.scatter = 1; p.sH = H; p.sW = W; p.sstride = stride;
        p.H = p.Ho; p.W = p.Wo; p.stride = 1;
        p.M = N * p.Ho * p.Wo;
    } else if (mode == 0) {
        p.M = N * H * W;
        if (stride == 2 && dil == 1 && KH * KW > 1) {      // parity classes: tile over the largest class
            p.parity = 1;
            p.M = N * ((H + 1) / 2) * ((W + 1) / 2);
        }
    } else return SH_EINVAL;
    if ((lddx & 3) == 0 && ((uintptr_t)dx & 15) == 0 && (!addend || ((ldadd & 3) == 0 && ((uintptr_t)addend & 15) == 0)))
        use_splitk(p, workspace, workspace_bytes);
    return launch_dgrad_any(p, (hipStream_t)stream);
}
// Input gradient + the front half of the BatchNorm backward of the layer that PRODUCED the conv's input (conv -> BN [-> ReLU] -> this
// conv): instead of dx the kernel stores g = relumask(y_prev * scale + shift) * (dx [+ addend]) and emits (sum g, sum g * xhat) per
// 64 rows into stat_partials[ceil(M/64)][2][Cin] -- the statistics pass of that BatchNorm's backward (sh_bn_bwd_reduce) and its
// re-read of dx and y disappear; sh_bn_bwd_finalize + sh_bn_bwd_apply(relu = 0) on g finish the job.
// y_prev: raw output of the producer conv [N*H*W][ldyp]; mean / invstd / scale / shift: its BatchNorm coefficients [Cin].
// SH_EUNSUPPORTED: no fused instantiation for this geometry (strided KxK / scatter); the caller runs the unfused sequence.
extern "C" int sh_conv_dgrad_x6_bnb(const float* dy, int lddy, const float* wt, const float* addend, int ldadd, float* g, int ldg,
                                    const float* y_prev, int ldyp, const float* out_prev, int ldop, const float* mean, const float* invstd,
                                    const float* scale, const float* shift, int relu, float* stat_partials, int N, int H, int W, int Cin, int Cout,
                                    int KH, int KW, int stride, int pad, int dil, float* workspace, int64_t workspace_bytes, int act_flags,
                                    void* stream) {
    ConvQ p{};
    if ((act_flags & ~7) || (act_flags & 6) == 6) return SH_EINVAL;
    // bit 0: y_prev, bit 1: out_prev stored as bf16; bit 2: out_prev is the ReLU quad mask sh_bn_act wrote (ldop = bytes per pixel >= Cin / 4)
    p.act = ((act_flags & 1) ? 4 : 0) | ((act_flags & 2) ? 8 : 0) | ((act_flags & 4) ? 64 : 0);
    const bool qmask = (act_flags & 4) != 0;
    if (!dy || !wt || !g || !y_prev || !mean || !invstd || !scale || !shift || !stat_partials ||
        !geom(p, N, H, W, Cin, Cout, KH, KW, stride, pad, dil)) return SH_EINVAL;
    const int CoutP = (Cout + 3) & ~3;
    if (lddy < CoutP || (lddy & 3) || ldg < Cin || ldyp < Cin || (addend && ldadd < Cin) || (out_prev && (qmask ? ldop * 4 : ldop) < Cin) ||
        (qmask && (!out_prev || (Cin & 3)))) return SH_EINVAL;
    if (stride != 1) return SH_EUNSUPPORTED;
    p.a = dy; p.b = wt; p.c = g; p.extra = addend; p.ldadd = ldadd; p.lda = lddy; p.ldc = ldg;
    p.Nn = Cin; p.Kc = CoutP; p.K = KH * KW * CoutP; p.M = N * H * W;
    p.partials = stat_partials; p.n_partials = (int)sh_cdiv(p.M, 64);
    p.bnb_y = y_prev; p.bnb_ldy = ldyp; p.bnb_mean = mean; p.bnb_invstd = invstd; p.bnb_scale = scale; p.bnb_shift = shift; p.bnb_relu = relu;
    p.bnb_out = out_prev; p.bnb_ldo = ldop;
    if (out_prev && !qmask && ((ldop & 3) || ((uintptr_t)out_prev & 15))) return SH_EUNSUPPORTED;
    const bool al = ((ldg | ldyp) & 3) == 0 && (((uintptr_t)g | (uintptr_t)y_prev | (uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)scale |
                                                 (uintptr_t)shift) & 15) == 0 && (!addend || ((ldadd & 3) == 0 && ((uintptr_t)addend & 15) == 0));
    if (al) use_splitk(p, workspace, workspace_bytes);
    if (!operand_extents(p, DGRAD)) return SH_EUNSUPPORTED;
    const int rc = sh_x6p_launch(DGRAD, p, (hipStream_t)stream);
    return rc == SH_X6P_NO ? SH_EUNSUPPORTED : rc;
}

// 1x1 stride-1 input gradient whose dy operand is NOT stored: dy = lin[0]*g + lin[1]*(y - lin[2]) + lin[3] per output channel (the
// deferred second half of the conv's own BatchNorm backward, sh_bn_bwd_finalize's lin[4][Cout]) is evaluated in the loader from
// the masked gradient g and the raw conv output y -- sh_bn_bwd_apply and the dy tensor disappear (Bottleneck conv1 / conv3,
// models/backbone/resnet.py via torchvision; the pointwise convs of sep_aspp_contrast_head.py:47-61).  Optional BatchNorm-backward
// epilogue for the PRODUCER of the conv's input exactly as sh_conv_dgrad_x6_bnb (y_prev == NULL: plain dx [+ addend]).
// SH_EUNSUPPORTED: no instantiation (Cout % 4, extents >= 2 GiB); the caller materialises dy with sh_bn_bwd_apply.
extern "C" int sh_conv_dgrad_x6_lin(const float* g, int ldg, const float* y, int ldy, const float* lin, const float* wt, const float* addend,
                                    int ldadd, float* dx, int lddx, const float* y_prev, int ldyp, const float* mean, const float* invstd,
                                    const float* scale, const float* shift, int relu, float* stat_partials, int N, int H, int W, int Cin,
                                    int Cout, float* workspace, int64_t workspace_bytes, int act_flags, void* stream) {
    ConvQ p{};
    if (!g || !y || !lin || !wt || !dx || !geom(p, N, H, W, Cin, Cout, 1, 1, 1, 0, 1) || (act_flags & ~3)) return SH_EINVAL;
    p.act = ((act_flags & 1) ? 16 : 0) | ((act_flags & 2) ? 4 : 0);         // bit 0: y (the lin stream), bit 1: y_prev stored as bf16
    if (ldg < Cout || ldy < Cout || ((ldg | ldy) & 3) || lddx < Cin || (addend && ldadd < Cin)) return SH_EINVAL;
    if (y_prev && (!mean || !invstd || !scale || !shift || !stat_partials || ldyp < Cin)) return SH_EINVAL;
    if ((Cout & 3) || (((uintptr_t)g | (uintptr_t)y | (uintptr_t)lin) & 15)) return SH_EUNSUPPORTED;
    p.a = g; p.lda = ldg; p.a2 = y; p.lda2 = ldy; p.lin = lin;
    p.b = wt; p.c = dx; p.extra = addend; p.ldadd = ldadd; p.ldc = lddx;
    p.Nn = Cin; p.Kc = Cout; p.K = Cout; p.M = N * H * W;
    bool al = (lddx & 3) == 0 && ((uintptr_t)dx & 15) == 0 && (!addend || ((ldadd & 3) == 0 && ((uintptr_t)addend & 15) == 0));
    if (y_prev) {
        p.partials = stat_partials; p.n_partials = (int)sh_cdiv(p.M, 64);
        p.bnb_y = y_prev; p.bnb_ldy = ldyp; p.bnb_mean = mean; p.bnb_invstd = invstd; p.bnb_scale = scale; p.bnb_shift = shift; p.bnb_relu = relu;
        al = al && (ldyp & 3) == 0 && (((uintptr_t)y_prev | (uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0;
    }
    if (al) use_splitk(p, workspace, workspace_bytes);
    if (!operand_extents(p, DGRAD)) return SH_EUNSUPPORTED;
    const long long a2 = ((long long)p.M - 1) * ldy + Cout;
    if (a2 * 4 >= (1ll << 31)) return SH_EUNSUPPORTED;
    p.a2_bytes = (unsigned)(a2 * ((p.act & 16) ? 2 : 4));
    const int rc = sh_x6p_launch(DGRAD, p, (hipStream_t)stream);
    return rc == SH_X6P_NO ? SH_EUNSUPPORTED : rc;
}

struct WgX6Plan { int wgm, wgn, splits, kchunk, per_xcd; };
static WgX6Plan wgrad_plan_x6(int Cout, long long Nn, long long npix, double rate = 0.29e12) {      // rate: flop/s of the main loop (bf16 compute: 1.2e12)
    WgX6Plan g;
    // measured (tools/bench_conv.py): 128x256 helps the narrow-Cout shapes, 256x256 on 1024 threads does not help wgrad
    g.wgm = Cout <= 64 ? 1 : 2;
    g.wgn = (Nn >= 256 && Cout <= 128) ? 4 : 2;
    // the stem (Cout 64, 7 x 7 x 4 = 196 columns): one 64 x 256 tile on 4 waves instead of two 64 x 128 tiles on 2 -- with 512 blocks the
    // 2-wave blocks left every SIMD with ONE wave (381 us); padding 60 columns costs less than that
    if (Cout <= 64 && Nn > 128 && Nn < 256) g.wgn = 4;
    if (Nn <= 64 && Cout >= 128) { g.wgn = 1; g.wgm = Cout >= 256 ? 4 : 2; }      // layer1's 64-channel inputs: 64-wide N tile
    const long long tiles = sh_cdiv(Cout, 64 * g.wgm) * sh_cdiv(Nn, 64 * g.wgn);
    // K slices: the grid should be a whole number of "rounds" of 512 resident blocks (2 per CU) -- measured
    // (tools/wg_target_sweep.sh): 640 blocks cost as much as 1024, e.g. 512->512 @128^2: 1143 us at 640 vs 878 us at 512 --
    // and each slice writes + re-reads one fp32 copy of dW, which prices many short slices out for the small layers.
    //   cost(r) = r * (npix / s_r) * t_pixel + s_r * dW_bytes * 2 / HBM,   s_r = floor(512 r / tiles)
    const long long maxs = sh_cdiv(npix, 256);
    const double t_pixel = 2.0 * (64 * g.wgm) * (64 * g.wgn) / rate, dwb = 4.0 * Cout * (double)Nn;
    // resident blocks per CU: the kernels are bounded to 2 waves per SIMD (8 waves per CU) and their [k][row] planes take
    // 96 * ((128 wgm + 64) + (128 wgn + 64)) bytes of the 160 KB LDS -- 2 for the 256-thread tiles, 1 for 128 x 256 on 512 threads
    const long long lds_b = 96ll * ((128 * g.wgm + 64) + (128 * g.wgn + 64));
    long long bpc = 8 / (g.wgm * g.wgn);
    if (bpc > 160 * 1024 / lds_b) bpc = 160 * 1024 / lds_b;
    if (bpc < 1) bpc = 1;
    static int slot_fix = -1;
    if (slot_fix < 0) { const char* e = getenv("SEGHIERO_WG_SLOTS"); slot_fix = e ? atoi(e) : 0; }
    const long long slots = slot_fix > 0 ? slot_fix : 256 * bpc;
    static int xcd_tiles = -1;
    if (xcd_tiles < 0) { const char* e = getenv("SEGHIERO_WG_XCD_TILES"); xcd_tiles = e ? atoi(e) : 32; }
    // few tiles per slice: one slice per XCD (see the kernel), slices in multiples of 8, slots/8 resident blocks per XCD
    g.per_xcd = tiles <= xcd_tiles && maxs >= 8;
    long long s = 1;
    double best = 1e30;
    for (int r = 1; r <= 4; ++r) {
        long long sr;
        double rounds;
        if (g.per_xcd) {
            long long j = (slots / 8) * r / tiles;
            if (j < 1) j = 1;
            if (8 * j > maxs) j = maxs / 8;
            sr = 8 * j;
            rounds = (double)sh_cdiv(tiles * j, slots / 8);
        } else {
            sr = slots * r / tiles;
            if (sr < 1) sr = 1;
            if (sr > maxs) sr = maxs;
            rounds = (double)sh_cdiv(tiles * sr, slots);
        }
        const double cost = rounds * sh_cdiv(npix, sr) * t_pixel + sr * dwb * 2.0 / 4e12;
        if (cost < best * 0.97) { best = cost; s = sr; }          // more rounds only for a clear (> 3 %) gain
    }
    const long long chunk = sh_cdiv(sh_cdiv(npix, s), 32) * 32;
    g.kchunk = (int)chunk;
    g.splits = (int)sh_cdiv(npix, chunk);
    return g;
}
extern "C" int64_t sh_conv_wgrad_x6_workspace(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil) {
    ConvQ p{};
    if (!geom(p, N, H, W, Cin, Cout, KH, KW, stride, pad, dil)) return SH_EINVAL;
    const WgX6Plan g = wgrad_plan_x6(Cout, (long long)KH * KW * Cin, (long long)N * p.Ho * p.Wo);
    return (int64_t)g.splits * Cout * KH * KW * Cin * (int64_t)sizeof(float);
}
template <int WGM, int WGN>
static int launch_wgrad_x6(ConvQ& p, int splits, hipStream_t st) {
    constexpr size_t lds = 3 * 32 * (size_t)((2 * 64 * WGM + 64) + (2 * 64 * WGN + 64));
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_x6_kernel<WGM, WGN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    p.tiles_m = (int)sh_cdiv(p.M, 64 * WGM); p.tiles_n = (int)sh_cdiv(p.Nn, 64 * WGN);
    p.ksplit = splits;
    const unsigned grid = (unsigned)(p.tiles_m * p.tiles_n) * (unsigned)(p.scatter ? sh_cdiv(splits, 8) * 8 : splits);
    conv_wgrad_x6_kernel<WGM, WGN><<<grid, 64 * WGM * WGN, lds, st>>>(p);
    return sh_launch_status();
}
static int wgrad_x6_any(const float* x, int ldx, const float* in_scale, const float* in_shift, const float* dy, int lddy, float* dw,
                        float* workspace, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, void* stream,
                        const float* y_lin = nullptr, int ldyl = 0, const float* lin = nullptr, int act_flags = 0) {
    ConvQ p{};
    if (act_flags & ~3) return SH_EINVAL;
    p.act = ((act_flags & 1) ? 32 : 0) | ((act_flags & 2) ? 16 : 0);        // bit 0: x, bit 1: the lin y stream stored as bf16
    if (!x || !dy || !dw || !workspace || !geom(p, N, H, W, Cin, Cout, KH, KW, stride, pad, dil)) return SH_EINVAL;
    if (lddy < ((Cout + 3) & ~3) || (lddy & 3) || ldx < Cin || (ldx & 3)) return SH_EINVAL;
    p.a = dy; p.b = x; p.c = workspace; p.lda = lddy; p.ldb = ldx;
    if (lin) {
        const long long a2 = (((long long)N * p.Ho * p.Wo) - 1) * ldyl + Cout;
        if ((Cout & 3) || a2 * 4 >= (1ll << 31)) return SH_EUNSUPPORTED;
        p.a2 = y_lin; p.lda2 = ldyl; p.a2_bytes = (unsigned)(a2 * ((p.act & 16) ? 2 : 4)); p.lin = lin;
    }
    p.M = Cout; p.Nn = KH * KW * Cin; p.K = N * p.Ho * p.Wo;
    p.aff_scale = in_scale; p.aff_shift = in_shift;
    const WgX6Plan g = wgrad_plan_x6(Cout, p.Nn, p.K);
    p.kchunk = g.kchunk;
    p.scatter = g.per_xcd;              // (field reused) block -> (slice, tile) mapping, see the kernel
    hipStream_t st = (hipStream_t)stream;
    int rc = SH_X6P_NO;
    const long long ab = (((long long)p.K - 1) * lddy + ((Cout + 3) & ~3)) * 4, bb = (((long long)N * H * W - 1) * ldx + Cin) * 4;
    if (ab < (1ll << 31) && bb < (1ll << 31)) {
        p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)((p.act & 32) ? bb / 2 : bb);
        rc = sh_x6p_wgrad_launch(p, g.wgm, g.wgn, g.splits, st);
    }
    if (rc == SH_X6P_NO) {
        if (in_scale || lin || p.act) return SH_EUNSUPPORTED;
        if (g.wgm == 4 && g.wgn == 1) rc = launch_wgrad_x6<4, 1>(p, g.splits, st);
        else if (g.wgm == 2 && g.wgn == 1) rc = launch_wgrad_x6<2, 1>(p, g.splits, st);
        else if (g.wgm == 2 && g.wgn == 4) rc = launch_wgrad_x6<2, 4>(p, g.splits, st);
        else if (g.wgm == 1 && g.wgn == 4) rc = launch_wgrad_x6<1, 4>(p, g.splits, st);
        else if (g.wgm == 1 && g.wgn == 2) rc = launch_wgrad_x6<1, 2>(p, g.splits, st);
        else rc = launch_wgrad_x6<2, 2>(p, g.splits, st);
    }
    if (rc != SH_OK) return rc;
    const long long n = (long long)Cout * p.Nn, n4 = n / 4;
    slab_reduce_x6_kernel<<<(unsigned)sh_cdiv(n4, 64), 256, 0, st>>>(workspace, dw, n4, n, g.splits);
    return sh_launch_status();
}
extern "C" int sh_conv_wgrad_x6(const float* x, int ldx, const float* dy, int lddy, float* dw, float* workspace, int N, int H, int W,
                                int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, int act_flags, void* stream) {
    return wgrad_x6_any(x, ldx, nullptr, nullptr, dy, lddy, dw, workspace, N, H, W, Cin, Cout, KH, KW, stride, pad, dil, stream, nullptr, 0, nullptr,
                        act_flags & 1);
}
// Weight gradient of a conv whose input is read THROUGH the producer's BatchNorm + ReLU (see sh_conv_fprop_x6_aff): x holds the raw
// output of the previous convolution, the loader applies relu(x * in_scale[c] + in_shift[c]).  SH_EUNSUPPORTED: output width < 16.
extern "C" int sh_conv_wgrad_x6_aff(const float* x, int ldx, const float* in_scale, const float* in_shift, const float* dy, int lddy,
                                    float* dw, float* workspace, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                    int pad, int dil, int act_flags, void* stream) {
    if (!in_scale || !in_shift || (((uintptr_t)in_scale | (uintptr_t)in_shift) & 15)) return SH_EINVAL;
    return wgrad_x6_any(x, ldx, in_scale, in_shift, dy, lddy, dw, workspace, N, H, W, Cin, Cout, KH, KW, stride, pad, dil, stream, nullptr, 0, nullptr,
                        act_flags & 1);
}
// Weight gradient whose dY operand is lin(g, y) (see sh_conv_dgrad_x6_lin), optionally with x read through the producer's BatchNorm +
// ReLU (in_scale / in_shift, as sh_conv_wgrad_x6_aff).  SH_EUNSUPPORTED: output width < 16, Cout % 4, extents >= 2 GiB.
extern "C" int sh_conv_wgrad_x6_lin(const float* x, int ldx, const float* in_scale, const float* in_shift, const float* g, int ldg,
                                    const float* y, int ldy, const float* lin, float* dw, float* workspace, int N, int H, int W, int Cin,
                                    int Cout, int KH, int KW, int stride, int pad, int dil, int act_flags, void* stream) {
    if (!y || !lin || ldy < Cout || (ldy & 3) || (((uintptr_t)y | (uintptr_t)lin) & 15)) return SH_EINVAL;
    if ((in_scale == nullptr) != (in_shift == nullptr) || (((uintptr_t)in_scale | (uintptr_t)in_shift) & 15)) return SH_EINVAL;
    return wgrad_x6_any(x, ldx, in_scale, in_shift, g, ldg, dw, workspace, N, H, W, Cin, Cout, KH, KW, stride, pad, dil, stream, y, ldy, lin, act_flags);
}


// ============================================================================================ bf16 COMPUTE mode (conv_b16.hip)
// bf16 copies of the dense conv weights, made once per step from the fp32 masters: wb = the forward operand in the parameter's own OHWI
// layout [Cout][taps][Cin], wtb = the dgrad operand [taps][Cin][pad8(Cout)] (K-contiguous, zero padded); either may be NULL.
struct WbTab {
    const float* w[SH_WT_MAX]; unsigned short* wb[SH_WT_MAX]; unsigned short* wtb[SH_WT_MAX];
    int cout[SH_WT_MAX], taps[SH_WT_MAX], cin[SH_WT_MAX];
    long long start[SH_WT_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void weights_to_bf16_multi_kernel(const WbTab T) {
    __shared__ float tile[32][33];
    const long long b = blockIdx.x;
    int lo = 0, hi = T.n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (T.start[mid] <= b) lo = mid; else hi = mid - 1; }
    const int Cout = T.cout[lo], Cin = T.cin[lo], taps = T.taps[lo], CoutP = (Cout + 7) & ~7;
    const int tco = (CoutP + 31) / 32, tci = (Cin + 31) / 32;
    long long e = b - T.start[lo];
    const int ci0 = (int)(e % tci) * 32; e /= tci;
    const int co0 = (int)(e % tco) * 32;
    const int tap = (int)(e / tco);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* w = T.w[lo];
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        const float v = (co < Cout && ci < Cin) ? w[((long long)co * taps + tap) * Cin + ci] : 0.f;
        tile[r][tx] = v;
        if (T.wb[lo] && co < Cout && ci < Cin) T.wb[lo][((long long)co * taps + tap) * Cin + ci] = (unsigned short)f32_to_bf16_rne(v);
    }
    __syncthreads();
    if (T.wtb[lo] == nullptr) return;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < CoutP) T.wtb[lo][((long long)tap * Cin + ci) * CoutP + co] = (unsigned short)f32_to_bf16_rne(tile[tx][r]);
    }
}
extern "C" int sh_weights_to_bf16_multi(int n, const float* const* w, void* const* wb, void* const* wtb, const int* cout, const int* taps,
                                        const int* cin, void* stream) {
    if (n <= 0 || n > SH_WT_MAX || !w || !wb || !wtb || !cout || !taps || !cin) return SH_EINVAL;
    WbTab T;
    T.n = n;
    long long acc = 0;
    for (int i = 0; i < n; ++i) {
        if (!w[i] || cout[i] <= 0 || taps[i] <= 0 || cin[i] <= 0) return SH_EINVAL;
        T.w[i] = w[i]; T.wb[i] = (unsigned short*)wb[i]; T.wtb[i] = (unsigned short*)wtb[i]; T.cout[i] = cout[i]; T.taps[i] = taps[i]; T.cin[i] = cin[i];
        T.start[i] = acc;
        acc += (long long)taps[i] * sh_cdiv((cout[i] + 7) & ~7, 32) * sh_cdiv(cin[i], 32);
    }
    T.start[n] = acc;
    if (acc >= (1ll << 31)) return SH_EINVAL;
    weights_to_bf16_multi_kernel<<<(unsigned)acc, 256, 0, (hipStream_t)stream>>>(T);
    return sh_launch_status();
}

// y = conv(x', w) in bf16 compute mode: x is a bf16 tensor read as is or through the producer's BatchNorm + ReLU (in_scale / in_shift, as
// sh_conv_fprop_x6_aff), w_bf16 = sh_weights_to_bf16_multi's forward copy; one MFMA product per tile, fp32 accumulate, BatchNorm
// statistics partials from the fp32 accumulators; y is stored as bf16 (act_flags bit 1) or fp32.  SH_EUNSUPPORTED: no instantiation
// (channel counts not multiples of 8, unaligned rows, K x K taps with Cin % 64 != 0, extents >= 2 GiB): run the fp32-accurate entry point.
extern "C" int sh_conv_fprop_b16(const void* x, int ldx, const float* in_scale, const float* in_shift, const void* w_bf16, const float* bias,
                                 void* y, int ldy, float* stat_partials, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                 int pad, int dil, float* workspace, int64_t workspace_bytes, int act_flags, void* stream) {
    ConvQ p{};
    if (!x || !w_bf16 || !y || !geom(p, N, H, W, Cin, Cout, KH, KW, stride, pad, dil) || (act_flags & ~2)) return SH_EINVAL;
    if ((in_scale == nullptr) != (in_shift == nullptr) || (((uintptr_t)in_scale | (uintptr_t)in_shift) & 15)) return SH_EINVAL;
    if (ldx < Cin || ldy < Cout) return SH_EINVAL;
    // Cout % 8 != 0 (the classifier): the tile's columns run to pad8(Cout) -- the weight rows beyond Cout lie past the operand's extent and
    // read as zeros, so y's padding lanes receive bias[c] (the caller's bias then holds pad8(Cout) floats, zeros in the padding)
    const int CoutP8 = (Cout + 7) & ~7;
    if ((Cout & 7) && (stat_partials || ldy < CoutP8)) return SH_EUNSUPPORTED;
    p.act = 1 | (act_flags & 2);
    p.a = (const float*)x; p.b = (const float*)w_bf16; p.c = (float*)y; p.extra = bias; p.partials = stat_partials; p.lda = ldx; p.ldc = ldy;
    p.M = N * p.Ho * p.Wo; p.Nn = CoutP8; p.K = KH * KW * Cin; p.Kc = Cin;
    p.n_partials = (int)sh_cdiv(p.M, 64);
    p.aff_scale = in_scale; p.aff_shift = in_shift;
    if ((ldy & 3) == 0 && ((uintptr_t)y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0)) use_splitk(p, workspace, workspace_bytes, 1);
    const long long a = ((long long)N * H * W - 1) * ldx + Cin, b = (long long)Cout * p.K;
    if (a * 2 >= (1ll << 31) || b * 2 >= (1ll << 31)) return SH_EUNSUPPORTED;
    p.a_bytes = (unsigned)(a * 2); p.b_bytes = (unsigned)(b * 2);
    const int rc = sh_b16_launch(FPROP, p, 0, (hipStream_t)stream);
    return rc == SH_X6P_NO ? SH_EUNSUPPORTED : rc;
}
// the grouped ASPP launch (sh_conv1x1_grouped_fprop_x6) in bf16 compute mode: x[g] bf16 tensors, w[g] bf16 copies, y bf16 (bit 1) or fp32
extern "C" int sh_conv1x1_grouped_fprop_b16(int ngroups, const void* const* x, const int* ldx, const float* const* in_scale,
                                            const float* const* in_shift, const void* const* w_bf16, void* y, int ldy, float* stat_partials,
                                            int N, int H, int W, int Cin, int Cout, int act_flags, void* stream) {
    ConvQ p{};
    if (ngroups < 1 || ngroups > 6 || !x || !ldx || !in_scale || !in_shift || !w_bf16 || !y || !geom(p, N, H, W, Cin, Cout, 1, 1, 1, 0, 1) ||
        (act_flags & ~2)) return SH_EINVAL;
    if ((Cout & 127) || (Cin & 15) || ldy < ngroups * Cout) return SH_EUNSUPPORTED;
    const long long M = (long long)N * H * W;
    for (int g = 0; g < ngroups; ++g) {
        if (!x[g] || !w_bf16[g] || ldx[g] < Cin || ((in_scale[g] == nullptr) != (in_shift[g] == nullptr))) return SH_EINVAL;
        if ((((uintptr_t)in_scale[g] | (uintptr_t)in_shift[g]) & 15)) return SH_EINVAL;
        const long long ab = ((M - 1) * ldx[g] + Cin) * 2;
        if (ab >= (1ll << 31)) return SH_EUNSUPPORTED;
        p.ga[g] = (const float*)x[g]; p.ga_bytes[g] = (unsigned)ab; p.glda[g] = ldx[g]; p.gb[g] = (const float*)w_bf16[g]; p.gsc[g] = in_scale[g]; p.gsh[g] = in_shift[g];
    }
    p.act = 1 | (act_flags & 2);
    p.ngroups = ngroups; p.group_n = Cout;
    p.c = (float*)y; p.ldc = ldy; p.partials = stat_partials;
    p.M = (int)M; p.Nn = ngroups * Cout; p.K = Cin; p.Kc = Cin;
    p.b_bytes = (unsigned)((long long)Cout * Cin * 2);
    p.n_partials = (int)sh_cdiv(p.M, 64);
    const int rc = sh_b16_grouped_launch(p, (hipStream_t)stream);
    return rc == SH_X6P_NO ? SH_EUNSUPPORTED : rc;
}
// Input gradient in bf16 compute mode, every hook of the fp32-accurate entry points optional:
//   dy: the gradient operand [N*Ho*Wo][lddy] (lddy >= pad8(Cout), padding lanes zero), bf16 (act_flags bit 0) or fp32; with
//   y_lin / lin (1x1 convs) it is the masked gradient g (bf16 only) and the operand is lin(g, y_lin) as sh_conv_dgrad_x6_lin;
//   wt_bf16: sh_weights_to_bf16_multi's transposed copy; addend (bit 2: bf16) summed into the result; dx stored bf16 (bit 1) or fp32;
//   y_prev != NULL: BatchNorm-backward epilogue as sh_conv_dgrad_x6_bnb (bit 3: y_prev bf16, bit 4: out_prev bf16, bit 5: out_prev is the
//   ReLU quad mask).  Strided geometries without hooks: stride-2 KxK (input-parity classes), and -- act_flags bit 6 -- 1x1 strided convs
//   whose result is ADDED to dx at the strided pixels (sh_conv_dgrad_x6's mode 1).  SH_EUNSUPPORTED: run the fp32-accurate entry point.
extern "C" int sh_conv_dgrad_b16(const void* dy, int lddy, const void* y_lin, int ldyl, const float* lin, const void* wt_bf16, const void* addend,
                                 int ldadd, void* dx, int lddx, const void* y_prev, int ldyp, const void* out_prev, int ldop, const float* mean,
                                 const float* invstd, const float* scale, const float* shift, int relu, float* stat_partials, int N, int H,
                                 int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, float* workspace,
                                 int64_t workspace_bytes, int act_flags, void* stream) {
    ConvQ p{};
    if (!dy || !wt_bf16 || !dx || !geom(p, N, H, W, Cin, Cout, KH, KW, stride, pad, dil) || (act_flags & ~127)) return SH_EINVAL;
    const int CoutP = (Cout + 7) & ~7;
    const bool dy_bf = act_flags & 1, qmask = (act_flags & 32) != 0, scatter = (act_flags & 64) != 0;
    if (lddy < CoutP || lddx < Cin || (addend && ldadd < Cin)) return SH_EINVAL;
    if ((y_lin == nullptr) != (lin == nullptr) || (y_lin && (ldyl < Cout || KH * KW != 1))) return SH_EINVAL;
    if (y_prev && (!mean || !invstd || !scale || !shift || !stat_partials || ldyp < Cin)) return SH_EINVAL;
    if (out_prev && (!y_prev || (qmask ? ldop * 4 : ldop) < Cin || (qmask && (Cin & 3)))) return SH_EINVAL;
    if (y_lin && !dy_bf) return SH_EUNSUPPORTED;
    const bool hooks = y_lin || y_prev || addend;
    if (scatter) {               // 1x1 strided conv (the downsample branch): the result is ADDED to dx at the strided pixels
        if (KH != 1 || KW != 1 || pad != 0 || hooks) return SH_EINVAL;
    } else if (stride != 1) {    // stride-2 KxK: by input-parity class, plain epilogue
        if (stride != 2 || dil != 1 || KH * KW == 1 || hooks) return SH_EUNSUPPORTED;
    }
    p.act = (y_lin ? 16 : 0) | ((act_flags & 2) ? 128 : 0) | ((act_flags & 4) ? 256 : 0) | ((act_flags & 8) ? 4 : 0) | ((act_flags & 16) ? 8 : 0) | (qmask ? 64 : 0);
    p.a = (const float*)dy; p.lda = lddy; p.a2 = (const float*)y_lin; p.lda2 = ldyl; p.lin = lin;
    p.b = (const float*)wt_bf16; p.c = (float*)dx; p.extra = (const float*)addend; p.ldadd = ldadd; p.ldc = lddx;
    p.Nn = Cin; p.Kc = CoutP; p.K = KH * KW * CoutP; p.M = N * H * W;
    const long long dy_rows = (long long)N * p.Ho * p.Wo;
    if (scatter) {
        p.scatter = 1; p.sH = H; p.sW = W; p.sstride = stride;
        p.H = p.Ho; p.W = p.Wo; p.stride = 1;
        p.M = N * p.Ho * p.Wo;
    } else if (stride == 2) {
        p.parity = 1;
        p.M = N * ((H + 1) / 2) * ((W + 1) / 2);        // tiles over the largest class
    }
    bool al = (lddx & 3) == 0 && ((uintptr_t)dx & 15) == 0 && (!addend || ((ldadd & 3) == 0 && ((uintptr_t)addend & 15) == 0));
    if (y_prev) {
        p.partials = stat_partials; p.n_partials = (int)sh_cdiv(p.M, 64);
        p.bnb_y = (const float*)y_prev; p.bnb_ldy = ldyp; p.bnb_mean = mean; p.bnb_invstd = invstd; p.bnb_scale = scale; p.bnb_shift = shift; p.bnb_relu = relu;
        p.bnb_out = (const float*)out_prev; p.bnb_ldo = ldop;
        al = al && (ldyp & 3) == 0 && (((uintptr_t)y_prev | (uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0 &&
             (!out_prev || qmask || ((ldop & 3) == 0 && ((uintptr_t)out_prev & 15) == 0));
    }
    if (al) use_splitk(p, workspace, workspace_bytes, 1);
    const long long a = (dy_rows - 1) * lddy + CoutP, b = (long long)KH * KW * Cin * CoutP;
    if (a * 4 >= (1ll << 31) || b * 2 >= (1ll << 31)) return SH_EUNSUPPORTED;
    p.a_bytes = (unsigned)(a * (dy_bf ? 2 : 4)); p.b_bytes = (unsigned)(b * 2);
    if (y_lin) {
        const long long a2 = ((long long)p.M - 1) * ldyl + Cout;
        if (a2 * 2 >= (1ll << 31)) return SH_EUNSUPPORTED;
        p.a2_bytes = (unsigned)(a2 * 2);
    }
    const int rc = sh_b16_launch(DGRAD, p, dy_bf ? 0 : 1, (hipStream_t)stream);
    return rc == SH_X6P_NO ? SH_EUNSUPPORTED : rc;
}
// Weight gradient (fp32) in bf16 compute mode: x a bf16 tensor read as is or through the producer's BatchNorm + ReLU (in_scale /
// in_shift); dy bf16 (act_flags bit 0) or fp32, or -- 1x1 stride-1 convs -- the masked gradient g (bf16) with y_lin / lin as
// sh_conv_wgrad_x6_lin.  workspace: sh_conv_wgrad_x6_workspace bytes.  SH_EUNSUPPORTED: run the fp32-accurate entry point.
extern "C" int sh_conv_wgrad_b16(const void* x, int ldx, const float* in_scale, const float* in_shift, const void* dy, int lddy, const void* y_lin,
                                 int ldyl, const float* lin, float* dw, float* workspace, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                 int stride, int pad, int dil, int act_flags, void* stream) {
    ConvQ p{};
    if (!x || !dy || !dw || !workspace || !geom(p, N, H, W, Cin, Cout, KH, KW, stride, pad, dil) || (act_flags & ~1)) return SH_EINVAL;
    if ((in_scale == nullptr) != (in_shift == nullptr) || (((uintptr_t)in_scale | (uintptr_t)in_shift | (uintptr_t)lin) & 15)) return SH_EINVAL;
    if ((y_lin == nullptr) != (lin == nullptr) || (y_lin && ldyl < Cout)) return SH_EINVAL;
    if (lddy < Cout || ldx < Cin) return SH_EINVAL;
    const bool dy_bf = act_flags & 1;
    const int CoutP8 = (Cout + 7) & ~7;                    // the loader reads 8 channels of dy at a time: rows of pad8(Cout), padding lanes zero
    if (lddy < CoutP8 || (y_lin && ((Cout & 7) || !dy_bf)) || ((uintptr_t)dw & 15)) return SH_EUNSUPPORTED;
    p.a = (const float*)dy; p.b = (const float*)x; p.c = workspace; p.lda = lddy; p.ldb = ldx;
    p.M = Cout; p.Nn = KH * KW * Cin; p.K = N * p.Ho * p.Wo;
    p.aff_scale = in_scale; p.aff_shift = in_shift;
    if (y_lin) {
        const long long a2 = ((long long)p.K - 1) * ldyl + Cout;
        if (a2 * 2 >= (1ll << 31)) return SH_EUNSUPPORTED;
        p.a2 = (const float*)y_lin; p.lda2 = ldyl; p.a2_bytes = (unsigned)(a2 * 2); p.lin = lin;
    }
    // K slices priced at the bf16 loop's rate: a slice costs its fp32 slab (written, then read by the reduce) whatever the loop's speed,
    // so the one-product loop takes fewer, longer slices than the six-product one (never more: the fp32-accurate workspace size covers it)
    WgX6Plan g = wgrad_plan_x6(Cout, p.Nn, p.K, 1.2e12);
    { const WgX6Plan g6 = wgrad_plan_x6(Cout, p.Nn, p.K); if (g.splits > g6.splits) g = g6; }
    p.kchunk = g.kchunk;
    p.scatter = g.per_xcd;
    hipStream_t st = (hipStream_t)stream;
    const long long ab = ((long long)p.K - 1) * lddy + CoutP8, bb = ((long long)N * H * W - 1) * ldx + Cin;
    if (ab * 4 >= (1ll << 31) || bb * 2 >= (1ll << 31)) return SH_EUNSUPPORTED;
    p.a_bytes = (unsigned)(ab * (dy_bf ? 2 : 4)); p.b_bytes = (unsigned)(bb * 2);
    int rc = sh_b16_wgrad_launch(p, dy_bf ? 0 : 1, g.wgm, g.wgn, g.splits, st);
    if (rc == SH_X6P_NO) return SH_EUNSUPPORTED;
    if (rc != SH_OK) return rc;
    const long long n = (long long)Cout * p.Nn, n4 = n / 4;
    slab_reduce_x6_kernel<<<(unsigned)sh_cdiv(n4, 64), 256, 0, st>>>(workspace, dw, n4, n, g.splits);
    return sh_launch_status();
}
