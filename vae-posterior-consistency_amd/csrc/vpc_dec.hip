// Decoder kernels (gfx950): p(x|z) MLP  L -> 50 -> 100 -> d (sigmoid), in three modes that share one body:
//   MODE_FWD    z -> xhat                                   (Reg_VAE.decoder, src/models/VAE.py:397-401)
//   MODE_BWD    (z, d loss/d xhat) -> dz + decoder dW partials       (autograd of the above)
//   MODE_FUSED  reparameterise + decoder forward + ELBO / consistency loss + backward seeds + decoder backward
//               in one pass over the rows: nothing of size B x d is written.  Loss maths:
//               Reg_VAE.loss / vanilla_VAE.loss, src/models/VAE.py:403-467, 1171-1208, helpers :469-494
//               (closed form in SURVEY.md Appendix A).
// The loss is evaluated in the generic form
//   loss*B = sum_p [ cA_p * NLL(A_p, xhat_p) + cE_p * NLL(A_p & ~B_p, xhat_p) ]
//            + bq * KL0(q) + bp * KL0(p) + cr * KL(q||p) - wml * loglik(z'; mu_p, lv_p)
// (kl_reg: pass q has A=mask, B=mask_p, cA=1-alpha, cE=alpha; pass p has A=mask_p, cA=alpha; bq=(1-alpha)b',
//  bp=alpha b', cr=alpha.  ml_reg: cA_q=1, bq=b', wml=(epoch/2800) alpha.  vanilla: one pass, cA=1, bq=b'.)
// The additive constant 0.5*log(2 pi) per element of every NLL term is added on the host.
#include "vpc_device.h"
#include "vpc_bf16.h"
#include "vpc_abi_internal.h"
#include "vpc_dec_args.h"
#include <cstdlib>

namespace vpc {

// One wave per SIMD (4 waves, up to 512 registers each), every wave owns NB = 2 batch tiles of 16 rows:
// the whole per-pass live set (activations of both tiles + 92 wgrad accumulators) stays in registers, each
// weight fragment read from LDS feeds two independent MFMA chains, and barriers involve 4 waves only.
// NB = 1: the small-batch shape (64-row workgroup tiles, the passes spread over blockIdx.y; see tile_shape in
// vpc_abi_internal.h) - same phases, staging and partial-block layout with a single MFMA chain per weight fragment.
// PREC != PREC_F32 (fused mode): the bf16 engine of vpc_bf16.h, as in dec8_kernel.
template <int DT, bool VEC, int MODE, int NB, int PREC = PREC_F32>
__global__ __launch_bounds__(DEC_THREADS) void dec_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef VPC_ABLATE
    unsigned long long T[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
    constexpr int CH = DEC_CH, TILE_ROWS = DEC_WAVES * 16 * NB;  // shadows vpc::TILE_ROWS
    constexpr int NA = (16 * DT > H1P ? 16 * DT : H1P);
    constexpr int I6 = (DT + 3) / 4;  // dW6 out tiles per wave (mt = w + 4i)
    constexpr bool BF = PREC != PREC_F32;
    constexpr int S4K = BF ? 32 : S4;  // row pitch of the W4 image
    const DecImg im(DT, S4K);
    const float* W4 = lds + im.oW4;
    const float* W5 = lds + im.oW5;
    const float* W6 = lds + im.oW6;
    float* stA = lds + im.total;   // [NA][CH]   A operands of wgrad (dY)
    float* stB = stA + NA * CH;    // [112][CH]  B operands of wgrad (activations)
    float* red = BF ? stA : stB + H1P * CH;   // [DEC_WAVES][8] (bf16 image: at the LDS limit, aliases the staging buffer)
    // bf16 engine: the wgrad operands of the 64 staged rows as bf16 blocks, hi plane + lo plane (vpc_bf16.h, bf_stage_*);
    // FTA tile slots per A row (dpre has DT tiles, dg2 7, dg1 4), 7 per B row - the buffers keep their fp32 sizes
    constexpr int FTA = DT == 8 ? 8 : 7, SKB = CH / 32;
    float* sAh = stA;
    float* sAl = stA + CH * 8 * FTA;
    float* sBh = stB;
    float* sBl = stB + CH * 56;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int colbase = 16 * w;
    int sb[4];  // per-lane element offsets of the wgrad staging writes (tile 0); tiles add a compile-time constant
    stage_bases<DEC_CH>(sb, colbase, c, q);
    const float inv_s2 = expf(-a.x_logvar), half_lv = 0.5f * a.x_logvar;
    constexpr float HL2PI = 0.91893853320467274f;

    // latent arrays: 16-byte vector access when padded to 16 floats per row (pad entries are zero / ignored)
    // (the fused mode always works on the padded workspaces, the API modes on dense [B][L] tensors)
    constexpr bool PAD = (MODE == MODE_FUSED);
    auto ld_lat = [&](const float* base, long r, bool rok) -> f32x4 {
        if (PAD) return ld_tile_o<true>(base, r, 16, 4 * q, 16, rok);
        return ld_tile_o<false>(base, r, a.L, 4 * q, a.L, rok);
    };
    auto st_lat = [&](float* base, long r, bool rok, f32x4 v) {
        if (PAD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (4 * q + j < a.L) ? v[j] : 0.f;
            st_tile<true>(base, r, 16, 4 * q, 16, rok, v);
        } else {
            st_tile<false>(base, r, a.L, 4 * q, a.L, rok, v);
        }
    };
    const int p_lo = a.psplit ? (int)blockIdx.y : 0, p_hi = a.psplit ? p_lo + 1 : a.npass;
    // Latent statistics of one (tile, pass): this pass's mean / logvar / eps and the other pass's mean / logvar (the KL
    // coupling).  Optional arrays are aliased to a valid one and and-ed away (no branch: all loads are issued together
    // instead of one exposed latency per CFG join).  The FIRST (tile, pass) of a workgroup is requested before the weight
    // image is loaded - in the small-batch shape every workgroup has exactly one, and the two latencies (~2 500 cycles
    // each, stamps of r02) otherwise add up.
    struct LatIn { f32x4 mu, lv, e, mo, lo; };
    auto fetch_lat = [&](int tile, int p, LatIn (&Lt)[NB]) {
        const bool two = a.npass == 2;
        const uint32_t has_o = opaque_mask(two), has_e = opaque_mask(a.eps[p] != nullptr);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const long r = (long)tile * TILE_ROWS + w * 16 * NB + nb * 16 + c;
            const bool rok = r < a.B;
            Lt[nb].mu = ld_lat(a.mean[p], r, rok);
            Lt[nb].lv = ld_lat(a.logvar[p], r, rok);
            Lt[nb].e = and4(ld_lat(a.eps[p] ? a.eps[p] : a.mean[p], r, rok), has_e);
            Lt[nb].mo = and4(ld_lat(two ? a.mean[1 - p] : a.mean[p], r, rok), has_o);
            Lt[nb].lo = and4(ld_lat(two ? a.logvar[1 - p] : a.logvar[p], r, rok), has_o);
        }
    };
    constexpr bool HOIST = MODE == MODE_FUSED && NB == 1;  // (the two-tile shape has no registers to spare)
    LatIn Lpre[NB];
    bool have_pre = false;
    if (HOIST && (int)blockIdx.x < a.ntiles) {
        fetch_lat(blockIdx.x, p_lo, Lpre);
        have_pre = true;
    }
    load_image<25>(lds, a.img, im.total);  // <= 25 600 floats at DT = 8: one round of loads for 256 threads
    __syncthreads();
    VPC_STAMP(0);

    f32x4 acc6[2][H1T], acc5[2][H2T], acc4 = zero4();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int t = 0; t < H1T; ++t) acc6[i][t] = zero4();
#pragma unroll
        for (int t = 0; t < H2T; ++t) acc5[i][t] = zero4();
    }
    float S_A0 = 0.f, S_E0 = 0.f, S_A1 = 0.f, S_kl0q = 0.f, S_kl0p = 0.f, S_klr = 0.f, S_zll = 0.f;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        long row[NB];
        bool ok[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            row[nb] = (long)tile * TILE_ROWS + w * 16 * NB + nb * 16 + c;
            ok[nb] = row[nb] < a.B;
        }
        for (int p = p_lo; p < p_hi; ++p) {
            asm volatile("" ::: "memory");  // keep LDS weight reads inside the pass (see vpc_enc.hip)
            int cc = c, qq = q;
            launder(cc, qq);
            // ---------------- latent: z = mean + eps * exp(logvar / 2), KL terms and their seeds
            f32x4 z[NB][1], epsfac[NB], dmu_kl[NB], dlv_kl[NB];
            LatIn Lc[NB];
            if (MODE == MODE_FUSED) {
                if (HOIST && have_pre) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) Lc[nb] = Lpre[nb];
                } else {
                    fetch_lat(tile, p, Lc);
                }
                have_pre = false;
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                epsfac[nb] = zero4(); dmu_kl[nb] = zero4(); dlv_kl[nb] = zero4();
                if (MODE != MODE_FUSED) {  // API modes: z is given
                    z[nb][0] = ld_lat(a.z_in[p], row[nb], ok[nb]);
                    continue;
                }
                const f32x4 mu = Lc[nb].mu, lv = Lc[nb].lv;
                f32x4 e = Lc[nb].e;
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = (4 * q + j < a.L) ? e[j] : 0.f;  // padded eps rows hold noise
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float sig = __expf(0.5f * lv[j]);
                    z[nb][0][j] = mu[j] + e[j] * sig;
                    epsfac[nb][j] = e[j] * 0.5f * sig;
                }
                if (MODE == MODE_FUSED) {
                    // Out-of-range lanes (row >= B or feature >= L) hold mu = lv = 0 for both passes, for which
                    // every KL term and seed below is exactly 0, so no per-lane branch is needed (only the
                    // ml_reg log-likelihood has a non-zero value at 0 and is masked explicitly).
                    const bool two = a.npass == 2;
                    const f32x4 mo = Lc[nb].mo, lo = Lc[nb].lo;
                    const float b0 = (p == 0) ? a.bq : a.bp;
                    const float sgn = (p == 0) ? 1.f : -1.f;  // d KL(q||p) / d mu_q = -d / d mu_p
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
                        dmu_kl[nb][j] = dm * a.inv_B;
                        dlv_kl[nb][j] = dl * a.inv_B;
                    }
                    if (p == 0) { S_kl0q += kl0; if (two) S_klr += klr; } else { S_kl0p += kl0; }
                    if (two && a.wml != 0.f) {  // ml_reg: extra rsample z' of q scored under p (VAE.py:435-440)
                        f32x4 e3 = ld_lat(a.eps_ml, row[nb], ok[nb]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) e3[j] = (4 * q + j < a.L) ? e3[j] : 0.f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool live = ok[nb] && 4 * q + j < a.L;
                            const float mq = (p == 0) ? mu[j] : mo[j], lq = (p == 0) ? lv[j] : lo[j];
                            const float mp = (p == 0) ? mo[j] : mu[j], lp = (p == 0) ? lo[j] : lv[j];
                            const float sq = __expf(0.5f * lq), eip = __expf(-lp);
                            const float dlt = mq + e3[j] * sq - mp;
                            const float g = a.wml * dlt * eip * a.inv_B;
                            if (p == 0) {
                                if (live) S_zll += -HL2PI - 0.5f * lp - 0.5f * dlt * dlt * eip;
                                dmu_kl[nb][j] += g;
                                dlv_kl[nb][j] += g * e3[j] * 0.5f * sq;
                            } else {
                                dmu_kl[nb][j] -= g;
                                dlv_kl[nb][j] += live ? a.wml * (0.5f - 0.5f * dlt * dlt * eip) * a.inv_B : 0.f;
                            }
                        }
                    }
                }
            }
            VPC_STAMP(1);
            const bool skip_dec = (MODE == MODE_FUSED) && a.cA[p] == 0.f && a.cE[p] == 0.f;
            f32x4 dzt[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) dzt[nb] = zero4();
            if (!skip_dec) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * q + j == a.L) z[nb][0][j] = 1.f;  // constant feature that drives the bias chain
                // ---------------- decoder forward
                f32x4 g1[NB][H2T], g2[NB][H1T];
#pragma unroll
                for (int mt = 0; mt < H2T; ++mt) {
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 acc[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb] = zero4();
                    if (BF) {
                        BfOp zb[NB][1];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) zb[nb][0] = bf_pack<PREC>(z[nb][0], zero4());
                        bf_tile_fwd_nb<PREC, 1, S4K, NB>(W4, mt, zb, acc, cc, qq);
                    } else
                    tile_fwd_nb<1, S4, NB>(W4, mt, z, acc, cc, qq);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) g1[nb][mt] = relu4(acc[nb]);
                }
                launder(cc, qq);
                BfOp g1b[NB][2];
                if (BF) bf_acts_nb<PREC, H2T, NB>(g1, g1b);
#pragma unroll
                for (int mt = 0; mt < H1T; ++mt) {
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 acc[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb] = zero4();
                    if (BF) bf_tile_fwd_nb<PREC, 2, 64, NB>(W5, mt, g1b, acc, cc, qq);
                    else
                    tile_fwd_nb<H2T, 64, NB, NK2>(W5, mt, g1, acc, cc, qq);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) g2[nb][mt] = relu4(acc[nb]);
                }
                launder(cc, qq);
                VPC_STAMP(2);
                f32x4 dpre[NB][DT];
                float sa = 0.f, se = 0.f;
                // Software pipeline over the DT output tiles: while the MFMAs of tile mt+1 run, the VALU work of
                // tile mt (sigmoid, NLL terms, d/dxhat) is done, and the x / mask loads of tile mt+1 are in flight.
                f32x4 pre_cur[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) pre_cur[nb] = zero4();
                f32x4 xv_cur[NB];
                uint32_t ua_cur[NB], ub_cur[NB];
                const bool hasB = (MODE == MODE_FUSED) && a.mB[p] != nullptr;
                // Fused + vector mode: per-lane row addresses are formed ONCE per pass (rows / columns out of range are
                // clamped to a valid address and their values multiplied / and-ed away).  No select sits on a loaded
                // value and the second mask is always read (aliased to the first when absent): hipcc turns
                // `ok ? load : 0` and `if (hasB) load` into exec-masked branches whose joins carry s_waitcnt vmcnt(0),
                // i.e. two to three fully exposed HBM latencies per output tile (seen in the ISA of the r01 build).
                constexpr bool FAST = (MODE == MODE_FUSED) && VEC;
                const float* xl[NB];
                const uint32_t* mal[NB];
                const uint32_t* mbl[NB];
                const float hasBf = hasB ? 1.f : 0.f;
                if (FAST) {
                    const int cq = (4 * q + 3 < a.d) ? 4 * q : 0;
                    const uint8_t* mbp = hasB ? a.mB[p] : a.mA[p];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const long ro = (ok[nb] ? row[nb] : 0) * a.d + cq;
                        xl[nb] = a.x + ro;
                        mal[nb] = reinterpret_cast<const uint32_t*>(a.mA[p] + ro);
                        mbl[nb] = reinterpret_cast<const uint32_t*>(mbp + ro);
                    }
                }
                auto fetch = [&](int mt, f32x4 (&xv)[NB], uint32_t (&ua)[NB], uint32_t (&ub)[NB]) {
                    const int f0 = 16 * mt + 4 * q;
                    if (FAST) {
                        const bool colok = f0 + 3 < a.d;
                        const int fo = colok ? 16 * mt : 0;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const uint32_t vm = opaque_mask(ok[nb] && colok);
                            xv[nb] = and4(*reinterpret_cast<const f32x4*>(xl[nb] + fo), vm);
                            ua[nb] = mal[nb][fo >> 2] & vm;
                            ub[nb] = mbl[nb][fo >> 2] & vm;
                        }
                        return;
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        ua[nb] = 0u; ub[nb] = 0u;
                        if (MODE == MODE_FUSED) {
                            xv[nb] = ld_tile<VEC>(a.x, row[nb], a.d, f0, a.d, ok[nb]);
                            ua[nb] = ld_mask_raw<VEC>(a.mA[p], row[nb], a.d, f0, a.d, ok[nb]);
                            if (hasB) ub[nb] = ld_mask_raw<VEC>(a.mB[p], row[nb], a.d, f0, a.d, ok[nb]);
                        } else if (MODE == MODE_BWD) {
                            xv[nb] = ld_tile<VEC>(a.dxhat[p], row[nb], a.d, f0, a.d, ok[nb]);
                        } else {
                            xv[nb] = zero4();
                        }
                    }
                };
                fetch(0, xv_cur, ua_cur, ub_cur);
                BfOp g2b[NB][4];
                if (BF) {
                    bf_acts_nb<PREC, H1T, NB>(g2, g2b);
                    bf_tile_fwd_nb<PREC, 4, 128, NB>(W6, 0, g2b, pre_cur, cc, qq);
                } else
                tile_fwd_nb<H1T, 128, NB, NK1>(W6, 0, g2, pre_cur, cc, qq);
#pragma unroll
                for (int mt = 0; mt < DT; ++mt) {
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 pre_nxt[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) pre_nxt[nb] = zero4();
                    f32x4 xv_nxt[NB];
                    uint32_t ua_nxt[NB], ub_nxt[NB];
                    if (mt + 1 < DT) {
                        fetch(mt + 1, xv_nxt, ua_nxt, ub_nxt);
                        if (BF) bf_tile_fwd_nb<PREC, 4, 128, NB>(W6, mt + 1, g2b, pre_nxt, cc, qq);
                        else
                        tile_fwd_nb<H1T, 128, NB, NK1>(W6, mt + 1, g2, pre_nxt, cc, qq);
                    }
                    const int f0 = 16 * mt + 4 * q;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        f32x4 xh;
#pragma unroll
                        for (int j = 0; j < 4; ++j) xh[j] = fast_sigmoid(pre_cur[nb][j]);
                        if (MODE == MODE_FWD) {
                            st_tile<VEC>(a.xhat[p], row[nb], a.d, f0, a.d, ok[nb], xh);
                            continue;
                        }
                        f32x4 dxh;
                        if (MODE == MODE_FUSED && VPC_DBG(8)) {
                            dxh = pre_cur[nb] * a.inv_B;
                        } else if (MODE == MODE_FUSED) {
                            const f32x4 mA = mask_to_f32(ua_cur[nb]);
                            const f32x4 mE = FAST ? mA * (1.f - mask_to_f32(ub_cur[nb])) * hasBf
                                                  : (hasB ? mA * (1.f - mask_to_f32(ub_cur[nb])) : zero4());
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float diff = xh[j] - xv_cur[nb][j];
                                const float t = half_lv + 0.5f * diff * diff * inv_s2;
                                sa += mA[j] * t;
                                se += mE[j] * t;
                                dxh[j] = (a.cA[p] * mA[j] + a.cE[p] * mE[j]) * diff * inv_s2 * a.inv_B;
                            }
                        } else {
                            dxh = xv_cur[nb];
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) dpre[nb][mt][j] = dxh[j] * xh[j] * (1.f - xh[j]);
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        pre_cur[nb] = pre_nxt[nb]; xv_cur[nb] = xv_nxt[nb]; ua_cur[nb] = ua_nxt[nb]; ub_cur[nb] = ub_nxt[nb];
                    }
                }
                VPC_STAMP(3);
                if (MODE != MODE_FWD) {
                    if (p == 0) { S_A0 += sa; S_E0 += se; } else { S_A1 += sa; }
                    uint32_t gm2[NB], gm1[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) { gm2[nb] = relu_bits<H1T>(g2[nb]); gm1[nb] = relu_bits<H2T>(g1[nb]); }
                    // ---------------- dW6~ += dpre * g2^T   (owner: wave w -> out tiles w, w+4; all 7 in tiles)
                    launder(cc, qq);
#pragma unroll
                    for (int ch = 0; ch < NB; ++ch) {
                        if (VPC_DBG(2)) continue;
                        lds_barrier();
                        if (BF) {
#pragma unroll
                            for (int t = 0; t < DT; ++t) bf_stage_write<PREC, FTA>(sAh, sAl, 16 * w + cc, t, qq, dpre[ch][t]);
#pragma unroll
                            for (int t = 0; t < H1T; ++t) bf_stage_write<PREC, 7>(sBh, sBl, 16 * w + cc, t, qq, g2[ch][t]);
                        } else {
#pragma unroll
                            for (int t = 0; t < DT; ++t) stage_write_b<CH>(stA, t, dpre[ch][t], sb);
#pragma unroll
                            for (int t = 0; t < H1T; ++t) stage_write_b<CH>(stB, t, g2[ch][t], sb);
                        }
                        lds_barrier();
                        if (VPC_DBG(1)) continue;
                        if (BF) {
#pragma unroll
                            for (int sb2 = 0; sb2 < SKB; ++sb2) {
                                __builtin_amdgcn_sched_barrier(0);
                                BfOp fa[I6];
#pragma unroll
                                for (int i = 0; i < I6; ++i) fa[i] = bf_stage_frag<PREC, FTA>(sAh, sAl, (w + 4 * i) % DT, sb2, 16 * qq + cc);
#pragma unroll
                                for (int nt = 0; nt < H1T; ++nt) {
                                    const BfOp fb = bf_stage_frag<PREC, 7>(sBh, sBl, nt, sb2, 16 * qq + cc);
#pragma unroll
                                    for (int i = 0; i < I6; ++i)
                                        if (w + 4 * i < DT) acc6[i][nt] = bf_mma<PREC>(fa[i], fb, acc6[i][nt]);
                                }
                            }
                            continue;
                        }
#pragma unroll
                        for (int s = 0; s < CH / 16; ++s) {
                            __builtin_amdgcn_sched_barrier(0);
                            f32x4 fa[I6];
#pragma unroll
                            for (int i = 0; i < I6; ++i) fa[i] = stage_frag<CH>(stA, (w + 4 * i) % DT, s, cc, qq);
                            // B fragments are double-buffered by hand: hipcc sinks every LDS read to just before its
                            // first use (read -> s_waitcnt lgkmcnt(0) -> 8 MFMAs, one exposed LDS latency per 8 MFMAs in
                            // the r01 ISA); the sched_barrier pins the read of fragment nt+1 above the MFMAs of nt
                            constexpr bool ALL6 = 4 * (I6 - 1) + 3 < DT;  // every wave owns all of its I6 tiles
                            const bool own_all = ALL6 || w + 4 * (I6 - 1) < DT, own_0 = w < DT;
                            f32x4 fb_cur = stage_frag<CH>(stB, 0, s, cc, qq);
#pragma unroll
                            for (int nt = 0; nt < H1T; ++nt) {
                                const f32x4 fb_nxt = stage_frag<CH>(stB, nt + 1 < H1T ? nt + 1 : nt, s, cc, qq);
                                __builtin_amdgcn_sched_barrier(0);
                                // ownership (out tile w + 4i < DT) is decided per fragment, never per MFMA
                                if (own_all) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
#pragma unroll
                                        for (int i = 0; i < I6; ++i) acc6[i][nt] = VPC_MFMA(fa[i][j], fb_cur[j], acc6[i][nt]);
                                } else if (own_0) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) acc6[0][nt] = VPC_MFMA(fa[0][j], fb_cur[j], acc6[0][nt]);
                                }
                                fb_cur = fb_nxt;
                            }
                        }
                    }
                    // ---------------- dg2 = relu'(g2) * (W6~^T dpre)
                    VPC_STAMP(4);
                    launder(cc, qq);
                    f32x4 dg2[NB][H1T];
                    BfOp dpreb[NB][(DT + 1) / 2];
                    if (BF) bf_acts_nb<PREC, DT, NB>(dpre, dpreb);
#pragma unroll
                    for (int mt = 0; mt < H1T; ++mt) {
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4 acc[NB];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[nb] = VPC_DBG(4) ? dpre[nb][mt % DT] : zero4();
                        if (BF) bf_tile_T_nb<PREC, (DT + 1) / 2, 128, NB, DT>(W6, mt, dpreb, acc, 16 * qq + cc);
                        else
                        if (!VPC_DBG(4)) tile_T_nb<DT, 128, NB>(W6, mt, dpre, acc, cc, qq);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) dg2[nb][mt] = gate_bits(acc[nb], gm2[nb], mt);
                    }
                    // ---------------- dW5~ += dg2 * g1^T   (owner: wave w -> out tiles w, w+4 (<7); 4 in tiles)
                    VPC_STAMP(5);
                    launder(cc, qq);
#pragma unroll
                    for (int ch = 0; ch < NB; ++ch) {
                        if (VPC_DBG(2)) continue;
                        lds_barrier();
                        if (BF) {
#pragma unroll
                            for (int t = 0; t < H1T; ++t) bf_stage_write<PREC, FTA>(sAh, sAl, 16 * w + cc, t, qq, dg2[ch][t]);
#pragma unroll
                            for (int t = 0; t < H2T; ++t) bf_stage_write<PREC, 7>(sBh, sBl, 16 * w + cc, t, qq, g1[ch][t]);
                        } else {
#pragma unroll
                            for (int t = 0; t < H1T; ++t) stage_write_b<CH>(stA, t, dg2[ch][t], sb);
#pragma unroll
                            for (int t = 0; t < H2T; ++t) stage_write_b<CH>(stB, t, g1[ch][t], sb);
                        }
                        lds_barrier();
                        if (VPC_DBG(1)) continue;
                        if (BF) {
#pragma unroll
                            for (int sb2 = 0; sb2 < SKB; ++sb2) {
                                __builtin_amdgcn_sched_barrier(0);
                                const BfOp fa0 = bf_stage_frag<PREC, FTA>(sAh, sAl, w, sb2, 16 * qq + cc);
                                const BfOp fa1 = bf_stage_frag<PREC, FTA>(sAh, sAl, (w + 4) % H1T, sb2, 16 * qq + cc);
#pragma unroll
                                for (int nt = 0; nt < H2T; ++nt) {
                                    const BfOp fb = bf_stage_frag<PREC, 7>(sBh, sBl, nt, sb2, 16 * qq + cc);
                                    acc5[0][nt] = bf_mma<PREC>(fa0, fb, acc5[0][nt]);
                                    if (w + 4 < H1T) acc5[1][nt] = bf_mma<PREC>(fa1, fb, acc5[1][nt]);
                                }
                            }
                            continue;
                        }
#pragma unroll
                        for (int s = 0; s < CH / 16; ++s) {
                            __builtin_amdgcn_sched_barrier(0);
                            f32x4 fa[2];
                            fa[0] = stage_frag<CH>(stA, w, s, cc, qq);
                            fa[1] = stage_frag<CH>(stA, (w + 4) % H1T, s, cc, qq);
                            const bool two5 = w + 4 < H1T;  // waves 0..2 own two out tiles, wave 3 one
                            f32x4 fb_cur = stage_frag<CH>(stB, 0, s, cc, qq);
#pragma unroll
                            for (int nt = 0; nt < H2T; ++nt) {
                                const f32x4 fb_nxt = stage_frag<CH>(stB, nt + 1 < H2T ? nt + 1 : nt, s, cc, qq);
                                __builtin_amdgcn_sched_barrier(0);
                                if (two5) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        acc5[0][nt] = VPC_MFMA(fa[0][j], fb_cur[j], acc5[0][nt]);
                                        acc5[1][nt] = VPC_MFMA(fa[1][j], fb_cur[j], acc5[1][nt]);
                                    }
                                } else {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) acc5[0][nt] = VPC_MFMA(fa[0][j], fb_cur[j], acc5[0][nt]);
                                }
                                fb_cur = fb_nxt;
                            }
                        }
                    }
                    // ---------------- dg1 = relu'(g1) * (W5~^T dg2)
                    VPC_STAMP(6);
                    launder(cc, qq);
                    f32x4 dg1[NB][H2T];
                    BfOp dg2b[NB][4];
                    if (BF) bf_acts_nb<PREC, H1T, NB>(dg2, dg2b);
#pragma unroll
                    for (int mt = 0; mt < H2T; ++mt) {
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4 acc[NB];
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[nb] = VPC_DBG(4) ? dg2[nb][mt] : zero4();
                        if (BF) bf_tile_T_nb<PREC, 4, 64, NB, H1T>(W5, mt, dg2b, acc, 16 * qq + cc);
                        else
                        if (!VPC_DBG(4)) tile_T_nb_k<H1T, 64, NB, NK1>(W5, mt, dg2, acc, cc, qq);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) dg1[nb][mt] = gate_bits(acc[nb], gm1[nb], mt);
                    }
                    // ---------------- dW4~ += dg1 * z^T   (owner: wave w -> out tile w)
                    VPC_STAMP(7);
                    launder(cc, qq);
#pragma unroll
                    for (int ch = 0; ch < NB; ++ch) {
                        if (VPC_DBG(2)) continue;
                        lds_barrier();
                        if (BF) {
#pragma unroll
                            for (int t = 0; t < H2T; ++t) bf_stage_write<PREC, FTA>(sAh, sAl, 16 * w + cc, t, qq, dg1[ch][t]);
                            bf_stage_write<PREC, 7>(sBh, sBl, 16 * w + cc, 0, qq, z[ch][0]);
                        } else {
#pragma unroll
                            for (int t = 0; t < H2T; ++t) stage_write_b<CH>(stA, t, dg1[ch][t], sb);
                            stage_write_b<CH>(stB, 0, z[ch][0], sb);
                        }
                        lds_barrier();
                        if (BF) {
#pragma unroll
                            for (int sb2 = 0; sb2 < SKB; ++sb2) {
                                const BfOp fa = bf_stage_frag<PREC, FTA>(sAh, sAl, w, sb2, 16 * qq + cc);
                                const BfOp fb = bf_stage_frag<PREC, 7>(sBh, sBl, 0, sb2, 16 * qq + cc);
                                acc4 = bf_mma<PREC>(fa, fb, acc4);
                            }
                            continue;
                        }
#pragma unroll
                        for (int s = 0; s < CH / 16; ++s) {
                            const f32x4 fa = stage_frag<CH>(stA, w, s, cc, qq);
                            const f32x4 fb = stage_frag<CH>(stB, 0, s, cc, qq);
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc4 = VPC_MFMA(fa[j], fb[j], acc4);
                        }
                    }
                    launder(cc, qq);
                    if (BF) {
                        BfOp dg1b[NB][2];
                        bf_acts_nb<PREC, H2T, NB>(dg1, dg1b);
                        bf_tile_T_nb<PREC, 2, S4K, NB>(W4, 0, dg1b, dzt, 16 * qq + cc);
                    } else
                    tile_T_nb_k<H2T, S4, NB, NK2>(W4, 0, dg1, dzt, cc, qq);
                }
            }
            VPC_STAMP(8);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if (MODE == MODE_FUSED) {
                    // total seeds on the encoder outputs: KL part + reparameterisation path
                    st_lat(a.dmean[p], row[nb], ok[nb], dmu_kl[nb] + dzt[nb]);
                    st_lat(a.dlogvar[p], row[nb], ok[nb], dlv_kl[nb] + dzt[nb] * epsfac[nb]);
                } else if (MODE == MODE_BWD) {
                    st_lat(a.dz[p], row[nb], ok[nb], dzt[nb]);
                }
            }
        }
    }
    VPC_STAMP(9);
    if (MODE == MODE_FWD) return;
    const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
    float* part = a.part + blk * DEC_PART + (long)w * DEC_GREGS * 64 + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int nt = 0; nt < H1T; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(28 * i + 4 * nt + j) * 64] = acc6[i][nt][j];
#pragma unroll
        for (int nt = 0; nt < H2T; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(56 + 16 * i + 4 * nt + j) * 64] = acc5[i][nt][j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) part[(88 + j) * 64] = acc4[j];
    if (MODE == MODE_FUSED) {
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
            for (int k = 0; k < DEC_WAVES; ++k) t += (double)red[k * LOSS_TERMS + threadIdx.x];
            a.loss_part[blk * LOSS_TERMS + threadIdx.x] = t;
        }
    }
#ifdef VPC_ABLATE
    VPC_STAMP(10);
    if (VPC_DBG(64) && (blockIdx.x == 0 || blockIdx.x == 100) && (threadIdx.x & 63) == 0)
        printf("blk %d wave %d ticks(100MHz): prologue %llu latent %llu g1g2 %llu out %llu w6 %llu dg2 %llu w5 %llu dg1 %llu "
               "w4+dz %llu store %llu | tail9 %llu epi %llu\n", blockIdx.x, (int)(threadIdx.x >> 6), T[0], T[1], T[2], T[3], T[4], T[5], T[6], T[7], T[8], T[9] , T[9], T[10]);
#endif
}

static size_t dec_lds(int DT, int mode, int prec = 0) {
    const DecImg im(DT, prec ? 32 : S4);
    if (mode == MODE_FWD) return sizeof(float) * im.total;
    const int na = 16 * DT > H1P ? 16 * DT : H1P;
    return sizeof(float) * (im.total + na * DEC_CH + H1P * DEC_CH + (prec ? 0 : DEC_WAVES * LOSS_TERMS));
}

template <typename K>
static int launch(K kern, const DecArgs& args, const TileShape& ts, size_t lds, hipStream_t stream) {
    if (!lds_attr_done(reinterpret_cast<const void*>(kern), lds)) return VPC_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(ts.grid_x, ts.grid_y), dim3(DEC_THREADS), lds, stream, args);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

// bf16 / bf16x3 (fused mode, vector layout): the 4-wave kernel in either shape - every d % 4 == 0 (the 8-wave kernel covers
// d in (64, 128] in the throughput shape)
static int dispatch_bf(const DecArgs& a, const TileShape& ts, int prec, hipStream_t s) {
    const int DT = dt_for(a.d);
    const size_t lds = dec_lds(DT, MODE_FUSED, prec);
#define VPC_CASE(T)                                                                                                   \
    case T:                                                                                                           \
        if (ts.small)                                                                                                 \
            return prec == PREC_BF16X3 ? launch(dec_kernel<T, true, MODE_FUSED, 1, PREC_BF16X3>, a, ts, lds, s)        \
                                       : launch(dec_kernel<T, true, MODE_FUSED, 1, PREC_BF16>, a, ts, lds, s);         \
        return prec == PREC_BF16X3 ? launch(dec_kernel<T, true, MODE_FUSED, DEC_NB, PREC_BF16X3>, a, ts, lds, s)       \
                                   : launch(dec_kernel<T, true, MODE_FUSED, DEC_NB, PREC_BF16>, a, ts, lds, s);
    switch (DT) { VPC_CASE(1) VPC_CASE(2) VPC_CASE(4) VPC_CASE(8) }
#undef VPC_CASE
    return VPC_ERR_SHAPE;
}

template <int MODE>
static int dispatch(const DecArgs& a, bool vec, const TileShape& ts, hipStream_t s) {
    const int DT = dt_for(a.d);
    const size_t lds = dec_lds(DT, MODE);
#define VPC_CASE(T)                                                                                   \
    case T:                                                                                           \
        if (ts.small)                                                                                 \
            return vec ? launch(dec_kernel<T, true, MODE, 1>, a, ts, lds, s)                          \
                       : launch(dec_kernel<T, false, MODE, 1>, a, ts, lds, s);                        \
        return vec ? launch(dec_kernel<T, true, MODE, DEC_NB>, a, ts, lds, s)                         \
                   : launch(dec_kernel<T, false, MODE, DEC_NB>, a, ts, lds, s);
    switch (DT) { VPC_CASE(1) VPC_CASE(2) VPC_CASE(4) VPC_CASE(8) }
#undef VPC_CASE
    return VPC_ERR_SHAPE;
}

static int check_common(long B, int d, int L, int npass) {
    if (npass < 1 || npass > 2 || B <= 0) return VPC_ERR_ARG;
    if (d < 1 || d > MAX_D || L < 1 || L > MAX_L) return VPC_ERR_SHAPE;
    return VPC_OK;
}

}  // namespace vpc

using namespace vpc;

extern "C" int vpc_decoder_fwd(const float* z, const float* dec_img, float* xhat, long B, int d, int L,
                               void* stream) {
    if (!z || !dec_img || !xhat) return VPC_ERR_ARG;
    if (int e = check_common(B, d, L, 1)) return e;
    DecArgs a{};
    a.img = dec_img; a.z_in[0] = z; a.xhat[0] = xhat; a.B = B; a.d = d; a.L = L; a.npass = 1; a.lp = L;
    const TileShape ts = tile_shape(B, 1);
    a.ntiles = ts.ntiles; a.psplit = ts.small;
    const bool vec = (d % 4 == 0) && aligned16(xhat);
    return dispatch<MODE_FWD>(a, vec, ts, (hipStream_t)stream);
}

extern "C" int vpc_decoder_bwd(const float* z, const float* dxhat, const float* dec_img, float* dz,
                               float* partials, int* nblocks_out, long B, int d, int L, void* stream) {
    if (!z || !dxhat || !dec_img || !dz || !partials) return VPC_ERR_ARG;
    if (int e = check_common(B, d, L, 1)) return e;
    DecArgs a{};
    a.img = dec_img; a.z_in[0] = z; a.dxhat[0] = dxhat; a.dz[0] = dz; a.part = partials;
    a.B = B; a.d = d; a.L = L; a.npass = 1; a.lp = L;
    const TileShape ts = tile_shape(B, 1);
    a.ntiles = ts.ntiles; a.psplit = ts.small;
    if (nblocks_out) *nblocks_out = ts.nblocks;
    const bool vec = (d % 4 == 0) && aligned16(dxhat);
    return dispatch<MODE_BWD>(a, vec, ts, (hipStream_t)stream);
}

extern "C" int vpc_decoder_fused(const float* x, const float* dec_img, int npass, const uint8_t* const* maskA,
                                 const uint8_t* const* maskB, const float* cA, const float* cE,
                                 const float* const* mean, const float* const* logvar, const float* const* eps,
                                 const float* eps_ml, float bq, float bp, float cr, float wml, float inv_B,
                                 float x_logvar, float* const* dmean, float* const* dlogvar, int lat_pitch,
                                 int precision, float* partials, double* loss_partials, int* nblocks_out, long B, int d,
                                 int L, void* stream) {
    if (!x || !dec_img || !maskA || !cA || !cE || !mean || !logvar || !dmean || !dlogvar || !partials ||
        !loss_partials)
        return VPC_ERR_ARG;
    if (int e = check_common(B, d, L, npass)) return e;
    if (precision < 0 || precision > 2) return VPC_ERR_ARG;
    DecArgs a{};
    a.x = x; a.img = dec_img; a.part = partials; a.loss_part = loss_partials; a.eps_ml = eps_ml;
    a.bq = bq; a.bp = bp; a.cr = cr; a.wml = wml; a.inv_B = inv_B; a.x_logvar = x_logvar;
    if (lat_pitch != 16) return VPC_ERR_ARG;  // the fused kernel works on padded [B][16] latent workspaces only
    a.B = B; a.d = d; a.L = L; a.npass = npass; a.lp = lat_pitch;
    const TileShape ts = tile_shape(B, npass);
    a.ntiles = ts.ntiles; a.psplit = ts.small;
#ifdef VPC_ABLATE
    if (const char* e = getenv("VPC_DEBUG")) a.dbg = atoi(e);
#endif
    bool vec = (d % 4 == 0) && aligned16(x);
    for (int p = 0; p < npass; ++p) {
        if (!maskA[p] || !mean[p] || !logvar[p] || !dmean[p] || !dlogvar[p]) return VPC_ERR_ARG;
        a.mA[p] = maskA[p]; a.mB[p] = maskB ? maskB[p] : nullptr; a.cA[p] = cA[p]; a.cE[p] = cE[p];
        a.mean[p] = mean[p]; a.logvar[p] = logvar[p]; a.eps[p] = eps ? eps[p] : nullptr;
        a.dmean[p] = dmean[p]; a.dlogvar[p] = dlogvar[p];
        vec = vec && ((uintptr_t)a.mA[p] % 4 == 0) && (!a.mB[p] || (uintptr_t)a.mB[p] % 4 == 0);
    }
    if (wml != 0.f && !eps_ml) return VPC_ERR_ARG;
    if (nblocks_out) *nblocks_out = ts.nblocks;
    // throughput shape, d in (64, 128]: the 8-wave / one-tile-per-wave kernel (vpc_dec8.hip); VPC_DEC8=0 selects the
    // 4-wave / two-tiles-per-wave kernel of this file instead (same arguments, same partial-block layout; kept for A/B
    // runs and for d <= 64).  Small-batch shape: always the 4-wave kernel with one tile per wave.
    const char* e8 = getenv("VPC_DEC8");
    if (precision != 0) {
        if (!vec) return VPC_ERR_SHAPE;  // the bf16 variants cover the vector layout (d % 4 == 0) only
        if (!ts.small && dt_for(d) == 8) return dec8_dispatch(a, vec, ts.grid_x, precision, (hipStream_t)stream);
        return dispatch_bf(a, ts, precision, (hipStream_t)stream);
    }
    if (vec && !ts.small && dt_for(d) == 8 && !(e8 && atoi(e8) == 0)) return dec8_dispatch(a, vec, ts.grid_x, 0, (hipStream_t)stream);
    return dispatch<MODE_FUSED>(a, vec, ts, (hipStream_t)stream);
}
