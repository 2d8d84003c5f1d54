// Fused decoder kernel, 8-wave variant: TWO waves per SIMD, ONE 16-row batch tile per wave.
//
// Same mathematics, LDS images, staging buffers and partial-block layout as dec_kernel<.., MODE_FUSED> in vpc_dec.hip
// (4 waves x 2 batch tiles, one wave per SIMD); what changes is who hides latency.  In the 4-wave kernel everything a
// wave waits for (LDS fragment reads, the VALU of the loss behind the MFMAs of an output tile) is exposed unless the
// compiler interleaves it inside that one wave, and hipcc can only do that by keeping both tiles' state live, which
// spills.  Here the second wave of the SIMD fills those gaps in hardware; the price is one MFMA chain per wave
// (dependent-issue latency) and half the registers per wave (256), which the smaller per-wave state fits:
//   wgrad accumulators 48 (dW6 tile w x 7, dW5 tile w < 7 x 4, dW4 tile w < 4), g1 16, g2 28, dpre 32, latent 16.
// wgrad staging keeps the 64-column buffers: two rounds per phase, waves 4r..4r+3 stage their tile in round r and all
// 8 waves multiply.  Partial blocks are written in the 4-wave layout (out tile mt of dW6 lives at wave mt & 3,
// registers 28 (mt >> 2) + ...), so the host-side index tables and the reduction do not change.
#include "vpc_abi_internal.h"
#include "vpc_device.h"
#include "vpc_bf16.h"
#include "vpc_dec_args.h"

namespace vpc {

constexpr int DEC8_WAVES = 8, DEC8_THREADS = 512;

// Single-chain tile products with a 2-deep fragment pipeline: the second wave of the SIMD hides LDS latency here, so
// only two A fragments are in flight (8 registers) instead of the whole tile's (up to 32) as in the 4-wave kernel.
#ifdef VPC_ABLATE
#define ABL(bit) VPC_DBG(bit)   // timing experiments (diagnostic build): 1 no staging writes, 2 no barriers, 4 no wgrad MFMAs
#else
#define ABL(bit) false
#endif

template <int KT, int S, int NK = 4 * KT>  // NK: k-steps to run (the rest multiply padding zeros)
__device__ __forceinline__ f32x4 tile_fwd_p2(const float* W, int mt, const f32x4 (&in)[KT], int m, int q) {
    constexpr int MASK = (S / 4 - 1) & 15;
    const float* rowp = W + (16 * mt + m) * S;
    f32x4 acc = zero4();
    f32x4 fa = *reinterpret_cast<const f32x4*>(rowp + 4 * ((0 + q) ^ (m & MASK)));
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int kn = kt + 1 < KT ? kt + 1 : kt;
        const f32x4 fn = *reinterpret_cast<const f32x4*>(rowp + 4 * ((4 * kn + q) ^ (m & MASK)));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * kt + j < NK) acc = VPC_MFMA(fa[j], in[kt][j], acc);
        fa = fn;
    }
    return acc;
}
template <int KT, int S, int NK = 4 * KT>
__device__ __forceinline__ f32x4 tile_T_p2(const float* W, int mt, const f32x4 (&in)[KT], int m, int q) {
    constexpr int MASK = (S / 4 - 1) & 15;
    const int col = 16 * mt + m;
    const int cs = col >> 2, cl = col & 3;
    auto rd = [&](int kt) {
        f32x4 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 4 * q + j;
            f[j] = (4 * kt + j < NK) ? W[(16 * kt + r) * S + (((cs ^ (r & MASK)) << 2) | cl)] : 0.f;
        }
        return f;
    };
    f32x4 acc = zero4();
    f32x4 fa = rd(0);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const f32x4 fn = rd(kt + 1 < KT ? kt + 1 : kt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * kt + j < NK) acc = VPC_MFMA(fa[j], in[kt][j], acc);
        fa = fn;
    }
    return acc;
}

// PREC != PREC_F32: the bf16 engine of vpc_bf16.h on the bf16 decoder image (DecImgBf: W4 rows are 32 dwords).
template <int DT, bool VEC, int PREC = PREC_F32>
__global__ __launch_bounds__(DEC8_THREADS) void dec8_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef VPC_ABLATE
    unsigned long long T[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
    constexpr int CH = DEC_CH;
    constexpr int NA = (16 * DT > H1P ? 16 * DT : H1P);
    constexpr bool BF = PREC != PREC_F32;
    constexpr int S4K = BF ? 32 : S4;  // row pitch of the W4 image
    const DecImg im(DT, S4K);
    load_image<13>(lds, a.img, im.total);  // one round of loads for 512 threads
    const float* W4 = lds + im.oW4;
    const float* W5 = lds + im.oW5;
    const float* W6 = lds + im.oW6;
    float* stA = lds + im.total;   // [NA][CH]   A operands of wgrad (dY)
    float* stB = stA + NA * CH;    // [112][CH]  B operands of wgrad (activations)
    // [8][LOSS_TERMS] for the end-of-kernel loss reduction; with the (4 KB larger) bf16 image the kernel is at the 160 KB
    // LDS limit, so there it aliases the staging buffer (only used after the last staging round)
    float* red = BF ? stA : stB + H1P * CH;
    // bf16 engine: wgrad operands are staged as bf16, row-major (vpc_bf16.h, bf_stage_*): A rows of 8 tile slots, B rows of 7.
    // Plain bf16 fits all 128 batch rows in the buffers (ONE staging round per wgrad), the split form 64 rows x (hi, lo).
    constexpr int ROUNDS = PREC == PREC_BF16 ? 1 : 2, SROWS = TILE_ROWS / ROUNDS, SKB = SROWS / 32;
    float* sAh = stA;
    float* sAl = stA + SROWS * 64;
    float* sBh = stB;
    float* sBl = stB + SROWS * 56;
    const int sw16 = 16 * (int)(ROUNDS == 1 ? (threadIdx.x >> 6) : ((threadIdx.x >> 6) & 3));  // first staged row of this wave
    __syncthreads();
    VPC_STAMP(0);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int round_w = w >> 2;                 // staging round in which this wave writes
    int sb[4];
    stage_bases<CH>(sb, 16 * (w & 3), c, q);
    const float inv_s2 = expf(-a.x_logvar), half_lv = 0.5f * a.x_logvar;
    constexpr float HL2PI = 0.91893853320467274f;
    const bool own6 = w < DT, own5 = w < H1T, own4 = w < H2T;

    // latent-width ([B][16]) arrays go through range-checked buffer descriptors over the rows [row0, B): a row past B
    // reads 0 / is not written, and an absent optional array (NULL) gets an empty descriptor and reads 0 - no address
    // clamps, no masking VALU, no exec-masked store blocks
    const int lrow = (threadIdx.x >> 6) * 16 + (threadIdx.x & 15);
    auto ld_lat = [&](const float* base, long row0) -> f32x4 {
        return ld_rows(rows_rsrc(base ? base : a.mean[0], row0, base ? a.B : row0, 16), lrow, 16, 4 * q);
    };
    auto st_lat = [&](float* base, long row0, f32x4 v) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (4 * q + j < a.L) ? v[j] : 0.f;
        st_rows(rows_rsrc(base, row0, a.B, 16), lrow, 16, 4 * q, v);
    };

    f32x4 acc6[H1T], acc5[H2T], acc4 = zero4();
#pragma unroll
    for (int t = 0; t < H1T; ++t) acc6[t] = zero4();
#pragma unroll
    for (int t = 0; t < H2T; ++t) acc5[t] = zero4();
    float S_A0 = 0.f, S_E0 = 0.f, S_A1 = 0.f, S_kl0q = 0.f, S_kl0p = 0.f, S_klr = 0.f, S_zll = 0.f;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const long row0 = (long)tile * TILE_ROWS;
        const long row = row0 + w * 16 + c;
        const bool ok = row < a.B;
        for (int p = 0; p < a.npass; ++p) {
            asm volatile("" ::: "memory");  // keep LDS weight reads inside the pass (see vpc_enc.hip)
            int cc = c, qq = q;
            launder(cc, qq);
            // ---------------- latent: z = mean + eps * exp(logvar / 2).  The KL terms and their seeds are computed at the END
            // of the pass from a second read of the (L2-resident) statistics: nothing of them is live - or parked in
            // memory - while the decoder runs (this kernel has 256 registers per wave)
            const bool two = a.npass == 2;
            f32x4 z[1][1];
            {
                const f32x4 mu = ld_lat(a.mean[p], row0);
                const f32x4 lv = ld_lat(a.logvar[p], row0);
                const f32x4 e = ld_lat(a.eps[p], row0);
#pragma unroll
                for (int j = 0; j < 4; ++j)  // padded eps rows hold noise
                    z[0][0][j] = mu[j] + ((4 * q + j < a.L) ? e[j] : 0.f) * __expf(0.5f * lv[j]);
            }
            f32x4 mu, lv, e, mo, lo, e3;  // pass-end operands
            auto fetch_stats = [&]() {
                mu = ld_lat(a.mean[p], row0);
                lv = ld_lat(a.logvar[p], row0);
                e = ld_lat(a.eps[p], row0);
                mo = ld_lat(two ? a.mean[1 - p] : nullptr, row0);
                lo = ld_lat(two ? a.logvar[1 - p] : nullptr, row0);
                e3 = ld_lat(a.eps_ml, row0);
            };
            VPC_STAMP(1);
            VPC_CUT();
            const bool skip_dec = a.cA[p] == 0.f && a.cE[p] == 0.f;
            f32x4 dzt[1] = {zero4()};
            if (skip_dec) fetch_stats();
            if (!skip_dec) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * q + j == a.L) z[0][0][j] = 1.f;  // constant feature that drives the bias chain
                // ---------------- decoder forward
                f32x4 g1[1][H2T], g2[1][H1T];
                BfOp zb[1];
                if (BF) zb[0] = bf_pack<PREC>(z[0][0], zero4());
#pragma unroll
                for (int mt = 0; mt < H2T; ++mt) {
                    if (BF) {
                        g1[0][mt] = relu4(bf_tile_fwd<PREC, 1, S4K>(W4, mt, zb, zero4(), cc, qq));
                    } else {
                        f32x4 acc[1] = {zero4()};
                        tile_fwd_nb<1, S4, 1>(W4, mt, z, acc, cc, qq);
                        g1[0][mt] = relu4(acc[0]);
                    }
                }
                VPC_CUT();
                launder(cc, qq);
                BfOp g1b[2];
                if (BF) bf_acts<PREC, H2T>(g1[0], g1b);
                if (BF) {
                    bf_layer_fwd<PREC, 2, 64, H1T, 2>(W5, g1b, cc, qq,
                                                                         [&](int mt, f32x4 acc) { g2[0][mt] = relu4(acc); });
                } else {
#pragma unroll
                    for (int mt = 0; mt < H1T; ++mt) {
                        __builtin_amdgcn_sched_barrier(0);
                        g2[0][mt] = relu4(tile_fwd_p2<H2T, 64, NK2>(W5, mt, g1[0], cc, qq));
                    }
                }
                // bf16 engine: from here on the PACKED operands are the currency - they feed the next layer's MFMAs, the dgrad
                // and the wgrad staging writes (one conversion per value); the fp32 tiles die as soon as they are packed
                BfOp g2b[4];
                uint32_t gm2_bf = 0;
                if (BF) {
                    bf_acts<PREC, H1T>(g2[0], g2b);
                    gm2_bf = relu_bits<H1T>(g2[0]);
                }
                launder(cc, qq);
                VPC_STAMP(2);
                VPC_CUT();
                // ---------------- output tiles: forward, loss terms, d/d pre-activation
                f32x4 dpre[1][DT];
                BfOp dpreb[(DT + 1) / 2];
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                constexpr float NLOG2E = -1.4426950408889634f;
                f32x2 sa2 = {0.f, 0.f}, se2 = {0.f, 0.f};
                const bool hasB = a.mB[p] != nullptr;
                const float kA = a.cA[p] * inv_s2 * a.inv_B, kE = a.cE[p] * inv_s2 * a.inv_B, hinv_s2 = 0.5f * inv_s2;
                // x / mask words of tile mt + 1 are requested before tile mt's MFMAs and consumed after the next tile's: a
                // wave never sits on a vmcnt wait in front of its MFMAs.  Range-checked buffer loads relative to the
                // tile's first row: a row past B reads 0 (zero mask = zero weight); out-of-range columns (possible in the
                // last DT / 2 tiles only, by dt_for) read column 0 and have their mask words cleared below.
                const __amdgpu_buffer_rsrc_t rx = rows_rsrc(a.x, row0, a.B, a.d);
                const long mrem = (a.B - row0) * (long)a.d;
                const uint32_t mrec = mrem > 0xffffffffL ? 0xffffffffu : (uint32_t)mrem;
                const __amdgpu_buffer_rsrc_t rmA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.mA[p]) + row0 * a.d,
                                                                                     0, mrec, 0x00020000);
                const __amdgpu_buffer_rsrc_t rmB = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<uint8_t*>(hasB ? a.mB[p] : a.mA[p]) + row0 * a.d, 0, mrec, 0x00020000);
                const int vo = lrow * a.d + ((VEC && 4 * q + 3 < a.d) ? 4 * q : 0);  // element offset inside the tile's rows
                auto fetch = [&](int mt, f32x4& xv, uint32_t& ua, uint32_t& ub) {
                    const int f0 = 16 * mt + 4 * q;
                    if (VEC) {
                        // d > 16 * DT / 2 (dt_for): the first DT / 2 tiles have no out-of-range columns and are immediate offsets
                        const int fo = (mt < DT / 2 || f0 + 3 < a.d) ? 16 * mt : 0;
                        xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, 4 * (vo + fo), 0, 0));
                        ua = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rmA, vo + fo, 0, 0);
                        ub = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rmB, vo + fo, 0, 0);
                    } else {
                        xv = ld_tile<false>(a.x, row, a.d, f0, a.d, ok);
                        ua = ld_mask_raw<false>(a.mA[p], row, a.d, f0, a.d, ok);
                        ub = hasB ? ld_mask_raw<false>(a.mB[p], row, a.d, f0, a.d, ok) : ua;
                    }
                };
                f32x4 xv_n;
                uint32_t ua_n, ub_n;
                fetch(0, xv_n, ua_n, ub_n);
                // plain bf16: the W6 fragments of tile mt + 1 are requested before tile mt's MFMAs and loss math (4 MFMAs of 16
                // cycles do not cover an LDS round trip; in the split form the 12-MFMA chain nearly does, and registers are short)
                constexpr bool WPF = PREC == PREC_BF16;
                BfOp w6n[4];
                if (WPF) {
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) w6n[kb] = bf_wfrag<PREC, 128>(W6, 0, kb, cc, qq);
                }
#pragma unroll
                for (int mt = 0; mt < DT; ++mt) {
                    VPC_CUT();
                    const f32x4 xv = xv_n;
                    uint32_t ua = ua_n, ub = ub_n;
                    if (mt + 1 < DT) fetch(mt + 1, xv_n, ua_n, ub_n);
                    f32x4 pre[1];
                    if (WPF) {
                        BfOp w6c[4];
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) w6c[kb] = w6n[kb];
                        if (mt + 1 < DT) {
#pragma unroll
                            for (int kb = 0; kb < 4; ++kb) w6n[kb] = bf_wfrag<PREC, 128>(W6, mt + 1, kb, cc, qq);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        pre[0] = zero4();
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) pre[0] = bf_mma<PREC>(w6c[kb], g2b[kb], pre[0]);
                    } else if (BF) pre[0] = bf_tile_fwd<PREC, 4, 128>(W6, mt, g2b, zero4(), cc, qq);
                    else pre[0] = tile_fwd_p2<H1T, 128, NK1>(W6, mt, g2[0], cc, qq);
                    if (VEC && mt >= DT / 2) {
                        const uint32_t vm = opaque_mask(16 * mt + 4 * q + 3 < a.d);
                        ua &= vm;
                        ub &= vm;
                    }
                    const f32x4 mA = mask_to_f32(ua);
                    // without a second mask the B words alias the A words (fetch): mA (1 - mA) = 0, no extra factor needed
                    const f32x4 mE = mA * (1.f - mask_to_f32(ub));
                    // every VALU instruction here costs MFMA time (fp32 MFMA and VALU do not overlap on a SIMD, see
                    // tools/microbench/mfma_valu_overlap.hip): constants are folded per pass, not per element
                    // two elements per instruction (v_pk_mul / v_pk_fma / v_pk_add_f32) wherever the math is not transcendental
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 p2 = {pre[0][2 * h], pre[0][2 * h + 1]}, x2 = {xv[2 * h], xv[2 * h + 1]};
                        const f32x2 a2 = {mA[2 * h], mA[2 * h + 1]}, e2 = {mE[2 * h], mE[2 * h + 1]};
                        const f32x2 en = p2 * NLOG2E;
                        const f32x2 den = f32x2{__builtin_amdgcn_exp2f(en[0]), __builtin_amdgcn_exp2f(en[1])} + 1.f;
                        const f32x2 xh = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
                        const f32x2 diff = xh - x2;
                        const f32x2 t = __builtin_elementwise_fma(diff * diff, f32x2{hinv_s2, hinv_s2}, f32x2{half_lv, half_lv});
                        sa2 = __builtin_elementwise_fma(a2, t, sa2);
                        se2 = __builtin_elementwise_fma(e2, t, se2);
                        const f32x2 wgt = __builtin_elementwise_fma(f32x2{kE, kE}, e2, a2 * kA);  // (cA mA + cE mE) / (sigma^2 B)
                        const f32x2 dp = (wgt * diff) * __builtin_elementwise_fma(-xh, xh, xh);
                        dpre[0][mt][2 * h] = dp[0];
                        dpre[0][mt][2 * h + 1] = dp[1];
                    }
                    // pin this tile's VALU here: without a volatile use hipcc sinks the sigmoid / loss code of ALL tiles
                    // below the loop (next to the first use of dpre), keeping 8 tiles of pre / x / masks live
                    asm volatile("" : "+v"(dpre[0][mt][0]), "+v"(dpre[0][mt][1]), "+v"(dpre[0][mt][2]), "+v"(dpre[0][mt][3]),
                                      "+v"(sa2), "+v"(se2));
                    if (BF && ((mt & 1) || mt + 1 == DT))
                        dpreb[mt >> 1] = bf_pack<PREC>(dpre[0][mt & ~1], (mt & 1) ? dpre[0][mt] : zero4());
                }
                VPC_STAMP(3);
                VPC_CUT();
                const float sa = sa2[0] + sa2[1], se = se2[0] + se2[1];
                if (p == 0) { S_A0 += sa; S_E0 += se; } else { S_A1 += sa; }
                const uint32_t gm2 = BF ? gm2_bf : relu_bits<H1T>(g2[0]);
                // ---------------- dW6~ += dpre * g2^T   (owner: wave w -> out tile w; all 7 in tiles)
                VPC_CUT();
                launder(cc, qq);
#pragma unroll
                for (int r = 0; r < (BF ? ROUNDS : 2); ++r) {
                    if (!ABL(2)) lds_barrier();
                    if (BF) {
                        if ((ROUNDS == 1 || round_w == r) && !ABL(1)) {
#pragma unroll
                            for (int kb = 0; kb < (DT + 1) / 2; ++kb) bf_stage_write_op<PREC, 8, DT>(sAh, sAl, sw16 + cc, kb, qq, dpreb[kb]);
#pragma unroll
                            for (int kb = 0; kb < 4; ++kb) bf_stage_write_op<PREC, 7, H1T>(sBh, sBl, sw16 + cc, kb, qq, g2b[kb]);
                        }
                    } else if (round_w == r && !ABL(1)) {
#pragma unroll
                        for (int t = 0; t < DT; ++t) stage_write_b<CH>(stA, t, dpre[0][t], sb);
#pragma unroll
                        for (int t = 0; t < H1T; ++t) stage_write_b<CH>(stB, t, g2[0][t], sb);
                    }
                    if (!ABL(2)) lds_barrier();
                    if (BF) {
                        if (own6 && !ABL(4)) {
#pragma unroll
                            for (int kb = 0; kb < SKB; ++kb) {
                                __builtin_amdgcn_sched_barrier(0);
                                const BfOp fa = bf_stage_frag<PREC, 8>(sAh, sAl, w, kb, 16 * qq + cc);
                                BfOp fb = bf_stage_frag<PREC, 7>(sBh, sBl, 0, kb, 16 * qq + cc);
#pragma unroll
                                for (int nt = 0; nt < H1T; ++nt) {  // two fragments in flight, not seven (registers)
                                    const BfOp fn = bf_stage_frag<PREC, 7>(sBh, sBl, nt + 1 < H1T ? nt + 1 : nt, kb, 16 * qq + cc);
                                    __builtin_amdgcn_sched_barrier(0);
                                    acc6[nt] = bf_mma<PREC>(fa, fb, acc6[nt]);
                                    fb = fn;
                                }
                            }
                        }
                    } else if (own6 && !ABL(4)) {
#pragma unroll
                        for (int s = 0; s < CH / 16; ++s) {
                            __builtin_amdgcn_sched_barrier(0);
                            const f32x4 fa = stage_frag<CH>(stA, w, s, cc, qq);
                            f32x4 fb_cur = stage_frag<CH>(stB, 0, s, cc, qq);
#pragma unroll
                            for (int nt = 0; nt < H1T; ++nt) {
                                const f32x4 fb_nxt = stage_frag<CH>(stB, nt + 1 < H1T ? nt + 1 : nt, s, cc, qq);
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc6[nt] = VPC_MFMA(fa[j], fb_cur[j], acc6[nt]);
                                fb_cur = fb_nxt;
                            }
                        }
                    }
                }
                // ---------------- dg2 = relu'(g2) * (W6~^T dpre)
                VPC_STAMP(4);
                VPC_CUT();
                launder(cc, qq);
                f32x4 dg2[1][H1T];
                if (BF) {
                    bf_layer_T<PREC, (DT + 1) / 2, 128, H1T, DT, PREC == PREC_BF16 ? (DT + 1) / 2 : 2>(
                        W6, dpreb, 16 * qq + cc, [&](int mt, f32x4 acc) { dg2[0][mt] = gate_bits(acc, gm2, mt); });
                } else {
#pragma unroll
                    for (int mt = 0; mt < H1T; ++mt) {
                        __builtin_amdgcn_sched_barrier(0);
                        VPC_CUT();
                        dg2[0][mt] = gate_bits(tile_T_p2<DT, 128>(W6, mt, dpre[0], cc, qq), gm2, mt);
                    }
                }
                // ---------------- dW5~ += dg2 * g1^T   (28 tiles; owners below)
                VPC_STAMP(5);
                VPC_CUT();
                launder(cc, qq);
                // g1 is RECOMPUTED here (16 MFMAs from z) instead of being kept live since the forward pass: 16 registers
                f32x4 g1r[1][H2T];
#pragma unroll
                for (int mt = 0; mt < H2T; ++mt) {
                    if (BF) {
                        g1r[0][mt] = relu4(bf_tile_fwd<PREC, 1, S4K>(W4, mt, zb, zero4(), cc, qq));
                    } else {
                        f32x4 acc[1] = {zero4()};
                        tile_fwd_nb<1, S4, 1>(W4, mt, z, acc, cc, qq);
                        g1r[0][mt] = relu4(acc[0]);
                    }
                }
                const uint32_t gm1 = relu_bits<H2T>(g1r[0]);
                const int nt5 = w & 3, mt5 = 4 * (w >> 2);
                BfOp dg2b[4], g1rb[2];
                if (BF) {
                    bf_acts<PREC, H1T>(dg2[0], dg2b);
                    bf_acts<PREC, H2T>(g1r[0], g1rb);
                }
#pragma unroll
                for (int r = 0; r < (BF ? ROUNDS : 2); ++r) {
                    if (!ABL(2)) lds_barrier();
                    if (BF) {
                        if ((ROUNDS == 1 || round_w == r) && !ABL(1)) {
#pragma unroll
                            for (int kb = 0; kb < 4; ++kb) bf_stage_write_op<PREC, 8, H1T>(sAh, sAl, sw16 + cc, kb, qq, dg2b[kb]);
#pragma unroll
                            for (int kb = 0; kb < 2; ++kb) bf_stage_write_op<PREC, 7, H2T>(sBh, sBl, sw16 + cc, kb, qq, g1rb[kb]);
                        }
                    } else if (round_w == r && !ABL(1)) {
#pragma unroll
                        for (int t = 0; t < H1T; ++t) stage_write_b<CH>(stA, t, dg2[0][t], sb);
#pragma unroll
                        for (int t = 0; t < H2T; ++t) stage_write_b<CH>(stB, t, g1r[0][t], sb);
                    }
                    if (!ABL(2)) lds_barrier();
                    // owner: wave w -> in tile w & 3 of out tiles 4 (w >> 2) .. + 3; out tile 7 does not exist, so waves 4..7
                    // run three: 7 tiles on every SIMD (waves s and s + 4) instead of 8 / 8 / 8 / 4 with one wave per out tile
                    if (BF) {
                        if (!ABL(4)) {
#pragma unroll
                            for (int kb = 0; kb < SKB; ++kb) {
                                __builtin_amdgcn_sched_barrier(0);
                                const BfOp fb = bf_stage_frag<PREC, 7>(sBh, sBl, nt5, kb, 16 * qq + cc);
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    if (i < 3 || w < 4) {
                                        const BfOp fa = bf_stage_frag<PREC, 8>(sAh, sAl, mt5 + i, kb, 16 * qq + cc);
                                        acc5[i] = bf_mma<PREC>(fa, fb, acc5[i]);
                                    }
                                    if (PREC == PREC_BF16X3) __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        }
                    } else if (!ABL(4)) {
#pragma unroll
                        for (int s = 0; s < CH / 16; ++s) {
                            __builtin_amdgcn_sched_barrier(0);
                            const f32x4 fb = stage_frag<CH>(stB, nt5, s, cc, qq);
                            f32x4 fa_cur = stage_frag<CH>(stA, mt5, s, cc, qq);
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                const f32x4 fa_nxt = stage_frag<CH>(stA, mt5 + i + 1, s, cc, qq);  // mt5 + 3 = 7: unused rows of stA
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc5[i] = VPC_MFMA(fa_cur[j], fb[j], acc5[i]);
                                fa_cur = fa_nxt;
                            }
                            if (w < 4) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc5[3] = VPC_MFMA(fa_cur[j], fb[j], acc5[3]);
                            }
                        }
                    }
                }
                // ---------------- dg1 = relu'(g1) * (W5~^T dg2)
                VPC_STAMP(6);
                VPC_CUT();
                launder(cc, qq);
                f32x4 dg1[1][H2T];
                if (BF) {
                    bf_layer_T<PREC, 4, 64, H2T, H1T, PREC == PREC_BF16 ? 4 : 2>(
                        W5, dg2b, 16 * qq + cc, [&](int mt, f32x4 acc) { dg1[0][mt] = gate_bits(acc, gm1, mt); });
                } else {
#pragma unroll
                    for (int mt = 0; mt < H2T; ++mt) {
                        __builtin_amdgcn_sched_barrier(0);
                        VPC_CUT();
                        dg1[0][mt] = gate_bits(tile_T_p2<H1T, 64, NK1>(W5, mt, dg2[0], cc, qq), gm1, mt);
                    }
                }
                // ---------------- dW4~ += dg1 * z^T   (owner: wave w < 4 -> out tile w)
                VPC_STAMP(7);
                VPC_CUT();
                launder(cc, qq);
                fetch_stats();  // the pass-end operands come in under this phase's MFMAs
                BfOp dg1b[2];
                if (BF) bf_acts<PREC, H2T>(dg1[0], dg1b);
#pragma unroll
                for (int r = 0; r < (BF ? ROUNDS : 2); ++r) {
                    if (!ABL(2)) lds_barrier();
                    if (BF) {
                        if ((ROUNDS == 1 || round_w == r) && !ABL(1)) {
#pragma unroll
                            for (int kb = 0; kb < 2; ++kb) bf_stage_write_op<PREC, 8, H2T>(sAh, sAl, sw16 + cc, kb, qq, dg1b[kb]);
                            bf_stage_write_op<PREC, 7, 1>(sBh, sBl, sw16 + cc, 0, qq, zb[0]);
                        }
                    } else if (round_w == r && !ABL(1)) {
#pragma unroll
                        for (int t = 0; t < H2T; ++t) stage_write_b<CH>(stA, t, dg1[0][t], sb);
                        stage_write_b<CH>(stB, 0, z[0][0], sb);
                    }
                    if (!ABL(2)) lds_barrier();
                    if (BF) {
                        if (own4 && !ABL(4)) {
#pragma unroll
                            for (int kb = 0; kb < SKB; ++kb) {
                                const BfOp fa = bf_stage_frag<PREC, 8>(sAh, sAl, w, kb, 16 * qq + cc);
                                const BfOp fb = bf_stage_frag<PREC, 7>(sBh, sBl, 0, kb, 16 * qq + cc);
                                acc4 = bf_mma<PREC>(fa, fb, acc4);
                            }
                        }
                    } else if (own4 && !ABL(4)) {
#pragma unroll
                        for (int s = 0; s < CH / 16; ++s) {
                            const f32x4 fa = stage_frag<CH>(stA, w, s, cc, qq);
                            const f32x4 fb = stage_frag<CH>(stB, 0, s, cc, qq);
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc4 = VPC_MFMA(fa[j], fb[j], acc4);
                        }
                    }
                }
                launder(cc, qq);
                if (BF) {
                    dzt[0] = bf_tile_T<PREC, 2, S4K>(W4, 0, dg1b, zero4(), 16 * qq + cc);
                } else {
                    tile_T_nb_k<H2T, S4, 1, NK2>(W4, 0, dg1, dzt, cc, qq);
                }
            }
            VPC_STAMP(8);
            // ---------------- KL terms, their seeds, and the total seeds on the encoder outputs (KL part + reparameterisation)
            {
                f32x4 dmu_kl, dlv_kl;
                const float b0 = (p == 0) ? a.bq : a.bp;
                const float sgn = (p == 0) ? 1.f : -1.f;
                const float crr = two ? a.cr : 0.f;
                float kl0 = 0.f, klr = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float elv = __expf(lv[j]);
                    kl0 += 0.5f * (elv + mu[j] * mu[j] - 1.f - lv[j]);
                    const float mq = (p == 0) ? mu[j] : mo[j], lq = (p == 0) ? lv[j] : lo[j];
                    const float mp = (p == 0) ? mo[j] : mu[j], lp = (p == 0) ? lo[j] : lv[j];
                    const float diff = mq - mp, eip = __expf(-lp), r = __expf(lq - lp);
                    klr += 0.5f * (r + diff * diff * eip - 1.f - (lq - lp));
                    const float dm = b0 * mu[j] + sgn * crr * diff * eip;
                    const float dl = b0 * 0.5f * (elv - 1.f) +
                                     crr * 0.5f * ((p == 0) ? (r - 1.f) : (1.f - r - diff * diff * eip));
                    dmu_kl[j] = dm * a.inv_B;
                    dlv_kl[j] = dl * a.inv_B;
                }
                if (p == 0) { S_kl0q += kl0; if (two) S_klr += klr; } else { S_kl0p += kl0; }
                if (two && a.wml != 0.f) {  // ml_reg: extra rsample z' of q scored under p (VAE.py:435-440)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool live = ok && 4 * q + j < a.L;
                        const float e3j = (4 * q + j < a.L) ? e3[j] : 0.f;
                        const float mq = (p == 0) ? mu[j] : mo[j], lq = (p == 0) ? lv[j] : lo[j];
                        const float mp = (p == 0) ? mo[j] : mu[j], lp = (p == 0) ? lo[j] : lv[j];
                        const float sq = __expf(0.5f * lq), eip = __expf(-lp);
                        const float dlt = mq + e3j * sq - mp;
                        const float g = a.wml * dlt * eip * a.inv_B;
                        if (p == 0) {
                            if (live) S_zll += -HL2PI - 0.5f * lp - 0.5f * dlt * dlt * eip;
                            dmu_kl[j] += g;
                            dlv_kl[j] += g * e3j * 0.5f * sq;
                        } else {
                            dmu_kl[j] -= g;
                            dlv_kl[j] += live ? a.wml * (0.5f - 0.5f * dlt * dlt * eip) * a.inv_B : 0.f;
                        }
                    }
                }
                if (skip_dec) {
                    st_lat(a.dmean[p], row0, dmu_kl);
                    st_lat(a.dlogvar[p], row0, dlv_kl);
                } else {
                    f32x4 ef;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ef[j] = (4 * q + j < a.L) ? e[j] * 0.5f * __expf(0.5f * lv[j]) : 0.f;
                    st_lat(a.dmean[p], row0, dmu_kl + dzt[0]);
                    st_lat(a.dlogvar[p], row0, dlv_kl + dzt[0] * ef);
                }
            }
        }
    }
    VPC_STAMP(9);
    // ---- partial block in the layout of the 4-wave kernel (vpc_layout.h): out tile mt of dW6 -> wave mt & 3, regs
    // 28 (mt >> 2) + 4 nt + j;  dW5 tile mt -> wave mt & 3, regs 56 + 16 (mt >> 2) + 4 nt + j;  dW4 tile mt -> wave mt, 88 + j
    float* part = a.part + (long)blockIdx.x * DEC_PART + (long)(w & 3) * DEC_GREGS * 64 + lane;
    const int hi = w >> 2;
    if (own6) {
#pragma unroll
        for (int nt = 0; nt < H1T; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(28 * hi + 4 * nt + j) * 64] = acc6[nt][j];
    } else {  // out tiles this model does not have (DT < 8): the slot still has to hold zeros for the reduction
#pragma unroll
        for (int nt = 0; nt < H1T; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(28 * hi + 4 * nt + j) * 64] = 0.f;
    }
    // dW5 tile (mt = 4 hi + i, nt = w & 3) sits in acc5[i] (see the wgrad-5 loop) and belongs to wave slot mt & 3 = i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* p5 = a.part + (long)blockIdx.x * DEC_PART + (long)i * DEC_GREGS * 64 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) p5[(56 + 16 * hi + 4 * (w & 3) + j) * 64] = (i < 3 || w < 4) ? acc5[i][j] : 0.f;
    }
    if (own4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) part[(88 + j) * 64] = acc4[j];
    }
    const float s[LOSS_TERMS] = {S_A0, S_E0, S_A1, S_kl0q, S_kl0p, S_klr, S_zll, 0.f};
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LOSS_TERMS; ++i) {
        const float v = wave_sum_dpp(s[i]);  // DPP adds: the shuffle form is 6 dependent ds_bpermute round trips per term
        if (lane == 0) red[w * LOSS_TERMS + i] = v;
    }
    __syncthreads();
    if (threadIdx.x < LOSS_TERMS) {
        double t = 0.0;
        for (int k = 0; k < DEC8_WAVES; ++k) t += (double)red[k * LOSS_TERMS + threadIdx.x];
        a.loss_part[(long)blockIdx.x * LOSS_TERMS + threadIdx.x] = t;
    }
#ifdef VPC_ABLATE
    VPC_STAMP(10);
    if (VPC_DBG(64) && blockIdx.x == 100 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) % 3 == 0)
        printf("dec8 blk %d wave %d: prologue %llu latent %llu g1g2 %llu out %llu w6 %llu dg2 %llu w5 %llu dg1 %llu w4+dz %llu store %llu epi %llu\n",
               blockIdx.x, (int)(threadIdx.x >> 6), T[0], T[1], T[2], T[3], T[4], T[5], T[6], T[7], T[8], T[9], T[10]);
#endif
}

size_t dec8_lds(int DT, int prec) {
    const DecImg im(DT, prec == PREC_F32 ? S4 : 32);
    const int na = 16 * DT > H1P ? 16 * DT : H1P;
    return sizeof(float) * (im.total + na * DEC_CH + H1P * DEC_CH + (prec == PREC_F32 ? DEC8_WAVES * LOSS_TERMS : 0));
}

// FUSED mode through the 8-wave kernel; returns VPC_ERR_SHAPE when this variant does not cover the shape
int dec8_dispatch(const DecArgs& a, bool vec, int grid, int prec, hipStream_t s) {
    const int DT = dt_for(a.d);
    const size_t lds = dec8_lds(DT, prec);
    // vector layout (d % 4 == 0, 16-byte aligned rows) only: the scalar-load instantiation of this kernel spilled 468 B
    // per lane at 256 registers; such shapes run the 4-wave kernel of vpc_dec.hip, which has 512 (the caller falls back)
    if (!vec) return VPC_ERR_SHAPE;
#define VPC_CASE8(T)                                                                                         \
    case T: {                                                                                                \
        auto kern = prec == PREC_BF16X3 ? dec8_kernel<T, true, PREC_BF16X3>                                   \
                    : prec == PREC_BF16 ? dec8_kernel<T, true, PREC_BF16>                                     \
                                        : dec8_kernel<T, true, PREC_F32>;                                     \
        if (!lds_attr_done(reinterpret_cast<const void*>(kern), lds)) return VPC_ERR_HIP;                    \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(DEC8_THREADS), lds, s, a);                                 \
        return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;                                       \
    }
    switch (DT) { VPC_CASE8(8) }
#undef VPC_CASE8
    return VPC_ERR_SHAPE;
}

}  // namespace vpc
