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
#include "vpc_abi_internal.h"

namespace vpc {

enum { MODE_FWD = 0, MODE_FUSED = 1, MODE_BWD = 2 };

struct DecArgs {
    const float* x;
    const float* img;
    const uint8_t* mA[2];
    const uint8_t* mB[2];
    float cA[2], cE[2];
    const float* mean[2];
    const float* logvar[2];
    const float* eps[2];
    const float* eps_ml;
    const float* z_in[2];
    const float* dxhat[2];
    float* xhat[2];
    float* dmean[2];
    float* dlogvar[2];
    float* dz[2];
    float* part;
    double* loss_part;
    float bq, bp, cr, wml, inv_B, x_logvar;
    long B;
    int d, L, npass, ntiles;
};

constexpr int DEC_CH = 32;

template <int DT, bool VEC, int MODE>
__global__ __launch_bounds__(THREADS, 2) void dec_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CH = DEC_CH;
    constexpr int WPC = CH / 16;
    constexpr int NA = (16 * DT > H1P ? 16 * DT : H1P);
    const DecImg im(DT);
    load_image(lds, a.img, im.total);
    const float* W4 = lds + im.oW4;
    const float* W5 = lds + im.oW5;
    const float* W6 = lds + im.oW6;
    float* stA = lds + im.total;   // [NA][CH]   A operands of wgrad (dY)
    float* stB = stA + NA * CH;    // [112][CH]  B operands of wgrad (activations)
    float* red = stB + H1P * CH;   // [WAVES][8]
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int colbase = 16 * (w % WPC);
    const float inv_s2 = expf(-a.x_logvar), half_lv = 0.5f * a.x_logvar;
    constexpr float HL2PI = 0.91893853320467274f;

    f32x4 acc6[H1T], acc5[H2T], acc4 = zero4();
#pragma unroll
    for (int i = 0; i < H1T; ++i) acc6[i] = zero4();
#pragma unroll
    for (int i = 0; i < H2T; ++i) acc5[i] = zero4();
    float S_A0 = 0.f, S_E0 = 0.f, S_A1 = 0.f, S_kl0q = 0.f, S_kl0p = 0.f, S_klr = 0.f, S_zll = 0.f;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const long row = (long)tile * TILE_ROWS + w * 16 + c;
        const bool ok = row < a.B;
        for (int p = 0; p < a.npass; ++p) {
            asm volatile("" ::: "memory");  // keep LDS weight reads inside the pass (see vpc_enc.hip)
            int cc = c, qq = q;
            launder(cc, qq);
            // ---------------- latent: z = mean + eps * exp(logvar / 2), KL terms and their seeds
            f32x4 z[1], epsfac = zero4(), dmu_kl = zero4(), dlv_kl = zero4();
            if (a.z_in[p]) {
                z[0] = ld_tile<false>(a.z_in[p], row, a.L, 4 * q, a.L, ok);
            } else {
                const f32x4 mu = ld_tile<false>(a.mean[p], row, a.L, 4 * q, a.L, ok);
                const f32x4 lv = ld_tile<false>(a.logvar[p], row, a.L, 4 * q, a.L, ok);
                f32x4 e = zero4();
                if (a.eps[p]) e = ld_tile<false>(a.eps[p], row, a.L, 4 * q, a.L, ok);
                f32x4 sig;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sig[j] = expf(0.5f * lv[j]);
                    z[0][j] = mu[j] + e[j] * sig[j];
                    epsfac[j] = e[j] * 0.5f * sig[j];
                }
                if (MODE == MODE_FUSED) {
                    // Out-of-range lanes (row >= B or feature >= L) hold mu = lv = 0 for both passes, for which
                    // every KL term and seed below is exactly 0, so no per-lane branch is needed (only the
                    // ml_reg log-likelihood has a non-zero value at 0 and is masked explicitly).
                    const bool two = a.npass == 2;
                    f32x4 mo = zero4(), lo = zero4();
                    if (two) {
                        mo = ld_tile<false>(a.mean[1 - p], row, a.L, 4 * q, a.L, ok);
                        lo = ld_tile<false>(a.logvar[1 - p], row, a.L, 4 * q, a.L, ok);
                    }
                    const float b0 = (p == 0) ? a.bq : a.bp;
                    const float sgn = (p == 0) ? 1.f : -1.f;   // d KL(q||p) / d mu_q = -d / d mu_p
                    const float crr = two ? a.cr : 0.f;
                    float kl0 = 0.f, klr = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float elv = expf(lv[j]);
                        kl0 += 0.5f * (elv + mu[j] * mu[j] - 1.f - lv[j]);
                        // q / p roles: (mq, lq) is the q pass, (mp, lp) the p pass
                        const float mq = (p == 0) ? mu[j] : mo[j], lq = (p == 0) ? lv[j] : lo[j];
                        const float mp = (p == 0) ? mo[j] : mu[j], lp = (p == 0) ? lo[j] : lv[j];
                        const float diff = mq - mp, eip = expf(-lp), r = expf(lq - lp);
                        klr += 0.5f * (r + diff * diff * eip - 1.f - (lq - lp));
                        const float dm = b0 * mu[j] + sgn * crr * diff * eip;
                        const float dl = b0 * 0.5f * (elv - 1.f) +
                                         crr * 0.5f * ((p == 0) ? (r - 1.f) : (1.f - r - diff * diff * eip));
                        dmu_kl[j] = dm * a.inv_B;
                        dlv_kl[j] = dl * a.inv_B;
                    }
                    if (p == 0) { S_kl0q += kl0; if (two) S_klr += klr; } else { S_kl0p += kl0; }
                    if (two && a.wml != 0.f) {  // ml_reg: extra rsample z' of q, scored under p (VAE.py:435-440)
                        const f32x4 e3 = ld_tile<false>(a.eps_ml, row, a.L, 4 * q, a.L, ok);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float mq = (p == 0) ? mu[j] : mo[j], lq = (p == 0) ? lv[j] : lo[j];
                            const float mp = (p == 0) ? mo[j] : mu[j], lp = (p == 0) ? lo[j] : lv[j];
                            const float sq = expf(0.5f * lq), eip = expf(-lp);
                            const float dlt = mq + e3[j] * sq - mp;
                            const float g = a.wml * dlt * eip * a.inv_B;
                            if (p == 0) {
                                if (ok && 4 * q + j < a.L) S_zll += -HL2PI - 0.5f * lp - 0.5f * dlt * dlt * eip;
                                dmu_kl[j] += g;
                                dlv_kl[j] += g * e3[j] * 0.5f * sq;
                            } else {
                                dmu_kl[j] -= g;
                                dlv_kl[j] += (ok && 4 * q + j < a.L) ? a.wml * (0.5f - 0.5f * dlt * dlt * eip) * a.inv_B : 0.f;
                            }
                        }
                    }
                }
            }
            const bool skip_dec = (MODE == MODE_FUSED) && a.cA[p] == 0.f && a.cE[p] == 0.f;
            f32x4 dzt = zero4();
            if (!skip_dec) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * q + j == a.L) z[0][j] = 1.f;  // constant feature that drives the bias chain
                // ---------------- decoder forward
                f32x4 g1[H2T], g2[H1T];
#pragma unroll
                for (int mt = 0; mt < H2T; ++mt) g1[mt] = relu4(tile_fwd<1, 64>(W4, mt, z, zero4(), cc, qq));
#pragma unroll
                for (int mt = 0; mt < H1T; ++mt) g2[mt] = relu4(tile_fwd<H2T, 64>(W5, mt, g1, zero4(), cc, qq));
                launder(cc, qq);
                f32x4 dpre[DT];
                float sa = 0.f, se = 0.f;
#pragma unroll
                for (int mt = 0; mt < DT; ++mt) {
                    __builtin_amdgcn_sched_barrier(0);  // one output tile at a time: x / mask loads stay local
                    const f32x4 pre = tile_fwd<H1T, 128>(W6, mt, g2, zero4(), cc, qq);
                    f32x4 xh;
#pragma unroll
                    for (int j = 0; j < 4; ++j) xh[j] = 1.f / (1.f + expf(-pre[j]));
                    const int f0 = 16 * mt + 4 * q;
                    if (MODE == MODE_FWD) {
                        st_tile<VEC>(a.xhat[p], row, a.d, f0, a.d, ok, xh);
                        continue;
                    }
                    f32x4 dxh;
                    if (MODE == MODE_FUSED) {
                        const f32x4 xv = ld_tile<VEC>(a.x, row, a.d, f0, a.d, ok);
                        const f32x4 mA = ld_mask<VEC>(a.mA[p], row, a.d, f0, a.d, ok);
                        f32x4 mE = zero4();
                        if (a.mB[p]) {
                            const f32x4 mB = ld_mask<VEC>(a.mB[p], row, a.d, f0, a.d, ok);
                            mE = mA * (1.f - mB);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float diff = xh[j] - xv[j];
                            const float t = half_lv + 0.5f * diff * diff * inv_s2;
                            sa += mA[j] * t;
                            se += mE[j] * t;
                            dxh[j] = (a.cA[p] * mA[j] + a.cE[p] * mE[j]) * diff * inv_s2 * a.inv_B;
                        }
                    } else {
                        dxh = ld_tile<VEC>(a.dxhat[p], row, a.d, f0, a.d, ok);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) dpre[mt][j] = dxh[j] * xh[j] * (1.f - xh[j]);
                }
                if (MODE != MODE_FWD) {
                    if (p == 0) { S_A0 += sa; S_E0 += se; } else { S_A1 += sa; }
                    const uint32_t gm2 = relu_bits<H1T>(g2), gm1 = relu_bits<H2T>(g1);
                    // ---------------- dW6~ += dpre * g2^T   (owner: wave w<DT -> out tile w, 7 in tiles)
                    launder(cc, qq);
                    for (int ch = 0; ch < TILE_ROWS / CH; ++ch) {
                        __syncthreads();
                        if (w / WPC == ch) {
#pragma unroll
                            for (int t = 0; t < DT; ++t) stage_write<CH>(stA, t, dpre[t], colbase, cc, qq);
#pragma unroll
                            for (int t = 0; t < H1T; ++t) stage_write<CH>(stB, t, g2[t], colbase, cc, qq);
                        }
                        __syncthreads();
                        if (w < DT) {
#pragma unroll
                            for (int s = 0; s < WPC; ++s) {
                                const f32x4 fa = stage_frag<CH>(stA, w, s, cc, qq);
#pragma unroll
                                for (int nt = 0; nt < H1T; ++nt) {
                                    const f32x4 fb = stage_frag<CH>(stB, nt, s, cc, qq);
#pragma unroll
                                    for (int j = 0; j < 4; ++j) acc6[nt] = VPC_MFMA(fa[j], fb[j], acc6[nt]);
                                }
                            }
                        }
                    }
                    // ---------------- dg2 = relu'(g2) * (W6~^T dpre)
                    launder(cc, qq);
                    f32x4 dg2[H1T];
#pragma unroll
                    for (int mt = 0; mt < H1T; ++mt)
                        dg2[mt] = gate_bits(tile_T<DT, 128>(W6, mt, dpre, zero4(), cc, qq), gm2, mt);
                    // ---------------- dW5~ += dg2 * g1^T   (owner: wave w<7 -> out tile w, 4 in tiles)
                    launder(cc, qq);
                    for (int ch = 0; ch < TILE_ROWS / CH; ++ch) {
                        __syncthreads();
                        if (w / WPC == ch) {
#pragma unroll
                            for (int t = 0; t < H1T; ++t) stage_write<CH>(stA, t, dg2[t], colbase, cc, qq);
#pragma unroll
                            for (int t = 0; t < H2T; ++t) stage_write<CH>(stB, t, g1[t], colbase, cc, qq);
                        }
                        __syncthreads();
                        if (w < H1T) {
#pragma unroll
                            for (int s = 0; s < WPC; ++s) {
                                const f32x4 fa = stage_frag<CH>(stA, w, s, cc, qq);
#pragma unroll
                                for (int nt = 0; nt < H2T; ++nt) {
                                    const f32x4 fb = stage_frag<CH>(stB, nt, s, cc, qq);
#pragma unroll
                                    for (int j = 0; j < 4; ++j) acc5[nt] = VPC_MFMA(fa[j], fb[j], acc5[nt]);
                                }
                            }
                        }
                    }
                    // ---------------- dg1 = relu'(g1) * (W5~^T dg2)
                    launder(cc, qq);
                    f32x4 dg1[H2T];
#pragma unroll
                    for (int mt = 0; mt < H2T; ++mt)
                        dg1[mt] = gate_bits(tile_T<H1T, 64>(W5, mt, dg2, zero4(), cc, qq), gm1, mt);
                    // ---------------- dW4~ += dg1 * z^T   (owner: wave w<4 -> out tile w)
                    launder(cc, qq);
                    for (int ch = 0; ch < TILE_ROWS / CH; ++ch) {
                        __syncthreads();
                        if (w / WPC == ch) {
#pragma unroll
                            for (int t = 0; t < H2T; ++t) stage_write<CH>(stA, t, dg1[t], colbase, cc, qq);
                            stage_write<CH>(stB, 0, z[0], colbase, cc, qq);
                        }
                        __syncthreads();
                        if (w < H2T) {
#pragma unroll
                            for (int s = 0; s < WPC; ++s) {
                                const f32x4 fa = stage_frag<CH>(stA, w, s, cc, qq);
                                const f32x4 fb = stage_frag<CH>(stB, 0, s, cc, qq);
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc4 = VPC_MFMA(fa[j], fb[j], acc4);
                            }
                        }
                    }
                    dzt = tile_T<H2T, 64>(W4, 0, dg1, zero4(), cc, qq);
                }
            }
            if (MODE == MODE_FUSED) {
                // total seeds on the encoder outputs: KL part + reparameterisation path
                st_tile<false>(a.dmean[p], row, a.L, 4 * q, a.L, ok, dmu_kl + dzt);
                st_tile<false>(a.dlogvar[p], row, a.L, 4 * q, a.L, ok, dlv_kl + dzt * epsfac);
            } else if (MODE == MODE_BWD) {
                st_tile<false>(a.dz[p], row, a.L, 4 * q, a.L, ok, dzt);
            }
        }
    }
    if (MODE == MODE_FWD) return;
    float* part = a.part + (long)blockIdx.x * DEC_PART + (long)w * GREGS * 64 + lane;
#pragma unroll
    for (int nt = 0; nt < H1T; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) part[(4 * nt + j) * 64] = acc6[nt][j];
#pragma unroll
    for (int nt = 0; nt < H2T; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) part[(28 + 4 * nt + j) * 64] = acc5[nt][j];
#pragma unroll
    for (int j = 0; j < 4; ++j) part[(44 + j) * 64] = acc4[j];
    if (MODE == MODE_FUSED) {
        const float s[LOSS_TERMS] = {S_A0, S_E0, S_A1, S_kl0q, S_kl0p, S_klr, S_zll, 0.f};
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LOSS_TERMS; ++i) {
            const float v = wave_sum(s[i]);
            if (lane == 0) red[w * LOSS_TERMS + i] = v;
        }
        __syncthreads();
        if (threadIdx.x < LOSS_TERMS) {
            double t = 0.0;
            for (int k = 0; k < WAVES; ++k) t += (double)red[k * LOSS_TERMS + threadIdx.x];
            a.loss_part[(long)blockIdx.x * LOSS_TERMS + threadIdx.x] = t;
        }
    }
}

static size_t dec_lds(int DT, int mode) {
    const DecImg im(DT);
    if (mode == MODE_FWD) return sizeof(float) * im.total;
    const int na = 16 * DT > H1P ? 16 * DT : H1P;
    return sizeof(float) * (im.total + na * DEC_CH + H1P * DEC_CH + WAVES * LOSS_TERMS);
}

template <typename K>
static int launch(K kern, const DecArgs& args, size_t lds, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return VPC_ERR_HIP;
    const int grid = args.ntiles < num_cus() ? args.ntiles : num_cus();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, stream, args);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

template <int MODE>
static int dispatch(const DecArgs& a, bool vec, hipStream_t s) {
    const int DT = dt_for(a.d);
    const size_t lds = dec_lds(DT, MODE);
#define VPC_CASE(T)                                                            \
    case T:                                                                    \
        return vec ? launch(dec_kernel<T, true, MODE>, a, lds, s)              \
                   : launch(dec_kernel<T, false, MODE>, a, lds, s);
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
    a.img = dec_img; a.z_in[0] = z; a.xhat[0] = xhat; a.B = B; a.d = d; a.L = L; a.npass = 1;
    a.ntiles = (int)((B + TILE_ROWS - 1) / TILE_ROWS);
    const bool vec = (d % 4 == 0) && aligned16(xhat);
    return dispatch<MODE_FWD>(a, vec, (hipStream_t)stream);
}

extern "C" int vpc_decoder_bwd(const float* z, const float* dxhat, const float* dec_img, float* dz,
                               float* partials, int* nblocks_out, long B, int d, int L, void* stream) {
    if (!z || !dxhat || !dec_img || !dz || !partials) return VPC_ERR_ARG;
    if (int e = check_common(B, d, L, 1)) return e;
    DecArgs a{};
    a.img = dec_img; a.z_in[0] = z; a.dxhat[0] = dxhat; a.dz[0] = dz; a.part = partials;
    a.B = B; a.d = d; a.L = L; a.npass = 1;
    a.ntiles = (int)((B + TILE_ROWS - 1) / TILE_ROWS);
    if (nblocks_out) *nblocks_out = a.ntiles < num_cus() ? a.ntiles : num_cus();
    const bool vec = (d % 4 == 0) && aligned16(dxhat);
    return dispatch<MODE_BWD>(a, vec, (hipStream_t)stream);
}

extern "C" int vpc_decoder_fused(const float* x, const float* dec_img, int npass, const uint8_t* const* maskA,
                                 const uint8_t* const* maskB, const float* cA, const float* cE,
                                 const float* const* mean, const float* const* logvar, const float* const* eps,
                                 const float* eps_ml, float bq, float bp, float cr, float wml, float inv_B,
                                 float x_logvar, float* const* dmean, float* const* dlogvar, float* partials,
                                 double* loss_partials, int* nblocks_out, long B, int d, int L, void* stream) {
    if (!x || !dec_img || !maskA || !cA || !cE || !mean || !logvar || !dmean || !dlogvar || !partials ||
        !loss_partials)
        return VPC_ERR_ARG;
    if (int e = check_common(B, d, L, npass)) return e;
    DecArgs a{};
    a.x = x; a.img = dec_img; a.part = partials; a.loss_part = loss_partials; a.eps_ml = eps_ml;
    a.bq = bq; a.bp = bp; a.cr = cr; a.wml = wml; a.inv_B = inv_B; a.x_logvar = x_logvar;
    a.B = B; a.d = d; a.L = L; a.npass = npass;
    a.ntiles = (int)((B + TILE_ROWS - 1) / TILE_ROWS);
    bool vec = (d % 4 == 0) && aligned16(x);
    for (int p = 0; p < npass; ++p) {
        if (!maskA[p] || !mean[p] || !logvar[p] || !dmean[p] || !dlogvar[p]) return VPC_ERR_ARG;
        a.mA[p] = maskA[p]; a.mB[p] = maskB ? maskB[p] : nullptr; a.cA[p] = cA[p]; a.cE[p] = cE[p];
        a.mean[p] = mean[p]; a.logvar[p] = logvar[p]; a.eps[p] = eps ? eps[p] : nullptr;
        a.dmean[p] = dmean[p]; a.dlogvar[p] = dlogvar[p];
        vec = vec && ((uintptr_t)a.mA[p] % 4 == 0) && (!a.mB[p] || (uintptr_t)a.mB[p] % 4 == 0);
    }
    if (wml != 0.f && !eps_ml) return VPC_ERR_ARG;
    if (nblocks_out) *nblocks_out = a.ntiles < num_cus() ? a.ntiles : num_cus();
    return dispatch<MODE_FUSED>(a, vec, (hipStream_t)stream);
}
