// Elementwise / reduction kernels of the MNAR path (reference src/models/VAE.py: REG_notMIWAE_v2 :2327-2505,
// notMIWAE_myversion :2691-2847).  HBM-bound fp32 work: coalesced row reads, wave-level reductions, no atomics
// (every cross-row sum goes through fixed-order partials => bit-reproducible).
//
//   nm_sample      z[b,k,:] = mean[b] + eps[b,k,:] * exp(logvar[b] / 2)              (encoder :2385-2389 / :2758-2763)
//   nm_sample_bwd  d heads from dz (sum over the K replicas) + the direct mean/logvar gradients
//   nm_loss        importance-weighted bound with the self-masking missingness model, forward AND backward in one
//                  launch: one wave per data row walks its K samples twice (pass 1: l_w and the log-sum-exp,
//                  pass 2: softmax weights x per-element derivatives), nothing but the inputs and the gradients
//                  touches HBM (loss :2398-2471 / :2774-2823)
//   nm_finalize    fixed-order reduction of the per-row statistics and the per-wave dW / db partials
#include "vpc_abi_internal.h"
#include <map>
#include <mutex>
#include <utility>
#include "vpc_device.h"
#include "../../include/vpc.h"

namespace vpc {

constexpr float NM_HALF_LOG_2PI = 0.91893853320467274f;

struct NMLossArgs {
    const float* x; const float* m; const float* mp;        // [B][d]; mp = nullptr for the un-regularised model
    const float* xm_q; const float* xl_q; long ld_q;         // decoder heads of the q pass, rows b*K+k
    const float* xm_p; const float* xl_p; long ld_p;         // p pass (regularised only)
    const float* hq; const float* hp; long ldh;              // encoder heads [B][mean L | logvar L]
    const float* W; const float* b;                          // missingness model, [d]
    const float* eps_kl;                                     // [B][K][L], un-regularised only (MC KL draw)
    float* g_xm_q; float* g_xl_q; long ldg_q;                // gradients (nullptr: forward only)
    float* g_xm_p; float* g_xl_p; long ldg_p;
    float* g_hq; float* g_hp; long ldgh;                     // [B][2L]
    float* gwb_part;                                         // [n_blocks][2][d]
    float* xm_imp;                                           // [B][d] self-normalised imputation (llh_eval), or nullptr
    double* stat_part;                                       // [n_blocks][NM_STATS]
    int B, K, d, L;
    float oq, op, oe, cr;                                    // gradient weights: (1-a)/B, a/B, a/(BK), a/(BL)
    float kq, kp;                                            // weights of the analytic KL gradients: (1-a)/B, a/B
    int gated;                                               // gradients w.r.t. the head PRE-activations
};
constexpr int NM_STATS = 5;  // sums over rows of: lse_q, lse_p, sum_k RE_e, sum_l kl_el, sum_k RE_q

// hardware exp / log / rcp (v_exp_f32, v_log_f32, v_rcp_f32: <= 1-2 ulp) for the per-element work of the loss kernel;
// the once-per-workgroup W transforms and the scalar log-sum-exp keep the IEEE routines
__device__ __forceinline__ float fexp(float v) { return __expf(v); }
__device__ __forceinline__ float frcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float softplus_f(float v) { return v > 20.f ? v : log1pf(expf(v)); }
__device__ __forceinline__ float sigmoid_f(float v) {
    const float e = expf(-fabsf(v));
    return v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}
struct Lse {  // online log-sum-exp
    float mx = -INFINITY, s = 0.f;
    __device__ __forceinline__ void add(float v) {
        if (v > mx) { s = s * fexp(mx - v) + 1.f; mx = v; } else s += fexp(v - mx);
    }
    __device__ __forceinline__ float value() const { return mx + logf(s); }
};

// PF: fetch sample k+1 while sample k is reduced (small batches: one row per wave, nothing else hides the latency;
// at large batches the extra registers cost occupancy and the other waves hide it anyway)
// KS = 4 (small batches): the four waves of a workgroup share ONE data row, wave s takes the replicas k = s, s + 4, ...; the row's
// l_w values meet in LDS for the log-sum-exp.  (One wave per row walks its K replicas twice, one dependent HBM round trip each: 38.8 us
// at batch 128, K = 20 - the longest kernel of the fp32 MNAR step there.)
constexpr int NM_KMAX = 64;  // most replicas the KS = 4 form handles
template <int T, bool REG, bool PF, int KS = 1>
__global__ __launch_bounds__(256) void nm_loss_kernel(NMLossArgs a) {
    extern __shared__ __align__(8) float lds[];
    double* stat_sh = reinterpret_cast<double*>(lds);   // [4 waves][NM_STATS]
    float* spW = lds + 2 * 4 * NM_STATS;                 // softplus(W)
    float* sgW = spW + a.d;                              // sigmoid(W) = d softplus
    float* bb = sgW + a.d;
    float* gsh = bb + a.d;                               // [4 waves][2 d] dW | db
    float* lw_sh = gsh + 4 * 2 * a.d;                    // KS > 1: [2][NM_KMAX] l_w of the q / p pass
    float* dml_sh = lw_sh + 2 * NM_KMAX;                 // KS > 1, un-regularised: [4 waves][2][64] d mean | d logvar
    for (int j = threadIdx.x; j < a.d; j += blockDim.x) {
        const float w = a.W[j];
        spW[j] = softplus_f(w); sgW[j] = sigmoid_f(w); bb[j] = a.b[j];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int gwave = KS == 1 ? (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6) : (int)blockIdx.x;
    const int nwaves = KS == 1 ? (int)((gridDim.x * blockDim.x) >> 6) : (int)gridDim.x;
    const int ks = KS == 1 ? 0 : wv;              // this wave's replicas: k = ks, ks + KS, ...
    const bool lead = KS == 1 || wv == 0;         // the wave that writes what a data row has once
    const int d = a.d, K = a.K, L = a.L;
    const bool grad = a.g_xm_q != nullptr;
    float gW[T], gB[T], sp[T], sg[T], bj[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int j = lane + 64 * t;
        gW[t] = gB[t] = 0.f;
        sp[t] = j < d ? spW[j] : 0.f; sg[t] = j < d ? sgW[j] : 0.f; bj[t] = j < d ? bb[j] : 0.f;
    }
    const float cd = NM_HALF_LOG_2PI * (float)d;
    double st[NM_STATS] = {0.0, 0.0, 0.0, 0.0, 0.0};

    for (int b = gwave; b < a.B; b += nwaves) {
        if (KS > 1) __syncthreads();  // the previous row's exchange buffers are read
        float x[T], m[T], mp[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int j = lane + 64 * t;
            const bool ok = j < d;
            x[t] = ok ? a.x[(long)b * d + j] : 0.f;
            m[t] = ok ? a.m[(long)b * d + j] : 0.f;
            mp[t] = (REG && ok) ? a.mp[(long)b * d + j] : 0.f;
        }
        // ---- latent statistics (lanes over L)
        float mu_q = 0.f, lv_q = 0.f, mu_p = 0.f, lv_p = 0.f, sd_q = 0.f;
        const bool lok = lane < L;
        if (lok) { mu_q = a.hq[(long)b * a.ldh + lane]; lv_q = a.hq[(long)b * a.ldh + L + lane]; }
        float KLq = 0.f, KLp = 0.f, klel = 0.f;
        if (REG) {
            if (lok) { mu_p = a.hp[(long)b * a.ldh + lane]; lv_p = a.hp[(long)b * a.ldh + L + lane]; }
            const float eq = expf(lv_q), ep = expf(lv_p), ivp = expf(-lv_p), ratio = expf(lv_q - lv_p);
            const float dm = mu_q - mu_p;
            KLq = wave_sum_dpp(lok ? 0.5f * (eq + mu_q * mu_q - 1.f - lv_q) : 0.f);
            KLp = wave_sum_dpp(lok ? 0.5f * (ep + mu_p * mu_p - 1.f - lv_p) : 0.f);
            klel = wave_sum_dpp(lok ? 0.5f * (ratio + dm * dm * ivp - 1.f - (lv_q - lv_p)) : 0.f);
            if (grad && lok && lead) {
                float* gq = a.g_hq + (long)b * a.ldgh;
                float* gp = a.g_hp + (long)b * a.ldgh;
                gq[lane] = a.kq * mu_q + a.cr * dm * ivp;
                gq[L + lane] = a.kq * 0.5f * (eq - 1.f) + a.cr * 0.5f * (ratio - 1.f);
                gp[lane] = a.kp * mu_p - a.cr * dm * ivp;
                gp[L + lane] = a.kp * 0.5f * (ep - 1.f) + a.cr * 0.5f * (1.f - ratio - dm * dm * ivp);
            }
        } else {
            sd_q = expf(0.5f * lv_q);
        }

        // the decoder outputs of sample k, fetched one sample ahead of their use (the loads of k+1 fly while k is
        // reduced: a wave has nothing else to overlap its HBM latency with)
        struct Fetch { float xm[T], xl[T], xmp[T], xlp[T], e; };
        auto fetch = [&](int k, Fetch& f) {
            const long row = (long)b * K + k;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int j = lane + 64 * t;
                const bool ok = j < d;
                f.xm[t] = ok ? a.xm_q[row * a.ld_q + j] : 0.f;
                f.xl[t] = ok ? a.xl_q[row * a.ld_q + j] : 0.f;
                if (REG) {
                    f.xmp[t] = ok ? a.xm_p[row * a.ld_p + j] : 0.f;
                    f.xlp[t] = ok ? a.xl_p[row * a.ld_p + j] : 0.f;
                }
            }
            if (!REG) f.e = lok ? a.eps_kl[row * L + lane] : 0.f;
        };
        // per-(b,k) terms; returns l_w_q and l_w_p, leaves the element-wise pieces in the out arrays
        auto terms = [&](const Fetch& f, float& lwq, float& lwp, float& re_q_out, float& re_e_out, float (&riv)[T],
                         float (&dn)[T], float (&rivp)[T], float& z_out, float& e_out) {
            float s_req = 0.f, s_nlp = 0.f, s_ree = 0.f, s_rep = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int j = lane + 64 * t;
                const bool ok = j < d;
                const float xm = f.xm[t];
                const float xl = f.xl[t];
                const float iv = fexp(-xl), r = x[t] - xm;
                riv[t] = r * iv;                       // (x - xm) / var
                const float el = 0.5f * xl + 0.5f * r * riv[t];
                s_req += m[t] * el;
                if (REG) s_ree += m[t] * (1.f - mp[t]) * el;
                const float mixv = xm * (1.f - m[t]) + x[t] * m[t];
                const float lg = -sp[t] * (mixv - bj[t]);
                const float el2 = fexp(-fabsf(lg)), rc = frcp(1.f + el2);
                s_nlp += ok ? fmaxf(lg, 0.f) - lg * m[t] + 0.6931471805599453f * __builtin_amdgcn_logf(1.f + el2) : 0.f;  // (argument in (1, 2]: v_log_f32)
                dn[t] = (lg >= 0.f ? rc : el2 * rc) - m[t];
                if (REG) {
                    const float xmp = f.xmp[t];
                    const float xlp = f.xlp[t];
                    const float ivp = fexp(-xlp), rp = x[t] - xmp;
                    rivp[t] = rp * ivp;
                    s_rep += mp[t] * (0.5f * xlp + 0.5f * rp * rivp[t]);
                }
            }
            float KL = KLq;
            if (!REG) {
                const float e = f.e;
                const float z = mu_q + e * sd_q;
                z_out = z; e_out = e;
                KL = wave_sum_dpp(lok ? -0.5f * e * e - 0.5f * lv_q + 0.5f * z * z : 0.f);
            }
            const float RE_q = wave_sum_dpp(s_req) + cd;
            const float nlp = wave_sum_dpp(s_nlp);
            lwq = RE_q + KL + nlp;
            re_q_out = RE_q;
            if (REG) {
                re_e_out = wave_sum_dpp(s_ree) + cd;
                lwp = wave_sum_dpp(s_rep) + cd + KLp;
            }
        };

        // ---- pass 1: log-sum-exp of +l_w (both passes) and of -l_w (imputation weights)
        Lse lq, lp, ln;
        float sum_ree = 0.f, sum_req = 0.f;
        float riv[T], dn[T], rivp[T];
        Fetch cur, nxt;
        if (PF && ks < K) fetch(ks, cur);
        for (int k = ks; k < K; k += KS) {
            if (PF) { if (k + KS < K) fetch(k + KS, nxt); } else fetch(k, cur);
            float lwq, lwp = 0.f, req, ree = 0.f, z, e;
            terms(cur, lwq, lwp, req, ree, riv, dn, rivp, z, e);
            if (PF) cur = nxt;
            if (KS == 1) {
                lq.add(lwq); ln.add(-lwq);
                if (REG) lp.add(lwp);
            } else if (lane == 0) {
                lw_sh[k] = lwq;
                lw_sh[NM_KMAX + k] = lwp;
            }
            if (REG) sum_ree += ree;
            sum_req += req;
        }
        if (KS > 1) {  // every wave: the log-sum-exp over all K replicas, in replica order
            __syncthreads();
            for (int k = 0; k < K; ++k) {
                const float lwq = lw_sh[k];
                lq.add(lwq); ln.add(-lwq);
                if (REG) lp.add(lw_sh[NM_KMAX + k]);
            }
        }
        const float lse_q = lq.value(), lse_p = REG ? lp.value() : 0.f, lse_n = ln.value();
        if (lead) { st[0] += lse_q; st[1] += lse_p; st[3] += klel; }
        st[2] += sum_ree; st[4] += sum_req;
        if (!grad && !a.xm_imp) continue;

        // ---- pass 2: softmax weights x element derivatives
        float imp[T], dmu = 0.f, dlv = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) imp[t] = 0.f;
        if (PF && ks < K) fetch(ks, cur);
        for (int k = ks; k < K; k += KS) {
            if (PF) { if (k + KS < K) fetch(k + KS, nxt); } else fetch(k, cur);
            float lwq, lwp = 0.f, req, ree = 0.f, z = 0.f, e = 0.f;
            terms(cur, lwq, lwp, req, ree, riv, dn, rivp, z, e);
            const long row = (long)b * K + k;
            if (a.xm_imp) {
                const float wi = fexp(-lwq - lse_n);
#pragma unroll
                for (int t = 0; t < T; ++t) imp[t] += wi * cur.xm[t];
            }
            if (!grad) { if (PF) cur = nxt; continue; }
            const float wq = a.oq * fexp(lwq - lse_q);
            const float wp = REG ? a.op * fexp(lwp - lse_p) : 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int j = lane + 64 * t;
                if (j >= d) continue;
                const float ee = REG ? a.oe * m[t] * (1.f - mp[t]) : 0.f;
                // recomputed rather than carried out of terms(): registers decide the occupancy of this kernel
                const float hq2 = 0.5f - 0.5f * (x[t] - cur.xm[t]) * riv[t];  // d/dxl of the element NLL
                const float mixv = cur.xm[t] * (1.f - m[t]) + x[t] * m[t];
                float gxm = wq * (-m[t] * riv[t] - dn[t] * sp[t] * (1.f - m[t])) - ee * riv[t];
                float gxl = (wq * m[t] + ee) * hq2;
                if (a.gated) {  // through Sigmoid / Hardtanh(-10, 0): the backward GEMMs then need no gate pass
                    gxm *= cur.xm[t] * (1.f - cur.xm[t]);
                    gxl = (cur.xl[t] > -10.f && cur.xl[t] < 0.f) ? gxl : 0.f;
                }
                a.g_xm_q[row * a.ldg_q + j] = gxm;
                a.g_xl_q[row * a.ldg_q + j] = gxl;
                gW[t] -= wq * dn[t] * sg[t] * (mixv - bj[t]);
                gB[t] += wq * dn[t] * sp[t];
                if (REG) {
                    const float hp2 = 0.5f - 0.5f * (x[t] - cur.xmp[t]) * rivp[t];
                    float gxmp = -wp * mp[t] * rivp[t], gxlp = wp * mp[t] * hp2;
                    if (a.gated) {
                        gxmp *= cur.xmp[t] * (1.f - cur.xmp[t]);
                        gxlp = (cur.xlp[t] > -10.f && cur.xlp[t] < 0.f) ? gxlp : 0.f;
                    }
                    a.g_xm_p[row * a.ldg_p + j] = gxmp;
                    a.g_xl_p[row * a.ldg_p + j] = gxlp;
                }
            }
            if (!REG) { dmu += wq * z; dlv += wq * (-0.5f + 0.5f * z * e * sd_q); }
            if (PF) cur = nxt;
        }
        if (a.xm_imp) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int j = lane + 64 * t;
                if (j < d) a.xm_imp[(long)b * d + j] = imp[t];
            }
        }
        if (grad && !REG) {
            if (KS == 1) {
                if (lok) {
                    a.g_hq[(long)b * a.ldgh + lane] = dmu;
                    a.g_hq[(long)b * a.ldgh + L + lane] = dlv;
                }
            } else {  // the waves' shares of the sum over the replicas, added in wave order
                dml_sh[(wv * 2 + 0) * 64 + lane] = dmu;
                dml_sh[(wv * 2 + 1) * 64 + lane] = dlv;
                __syncthreads();
                if (wv == 0 && lok) {
                    float sm = dml_sh[lane], sl = dml_sh[64 + lane];
#pragma unroll
                    for (int ww = 1; ww < 4; ++ww) { sm += dml_sh[(ww * 2) * 64 + lane]; sl += dml_sh[(ww * 2 + 1) * 64 + lane]; }
                    a.g_hq[(long)b * a.ldgh + lane] = sm;
                    a.g_hq[(long)b * a.ldgh + L + lane] = sl;
                }
            }
        }
    }
    // ---- workgroup partials, combined in wave order (fixed => deterministic)
    const int wave = wv;
    if (lane == 0)
        for (int i = 0; i < NM_STATS; ++i) stat_sh[wave * NM_STATS + i] = st[i];
    if (grad) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int j = lane + 64 * t;
            if (j < d) { gsh[wave * 2 * d + j] = gW[t]; gsh[wave * 2 * d + d + j] = gB[t]; }
        }
    }
    __syncthreads();
    if (threadIdx.x < NM_STATS)
        a.stat_part[(long)blockIdx.x * NM_STATS + threadIdx.x] =
            (stat_sh[threadIdx.x] + stat_sh[NM_STATS + threadIdx.x]) +
            (stat_sh[2 * NM_STATS + threadIdx.x] + stat_sh[3 * NM_STATS + threadIdx.x]);
    if (grad)
        for (int j = threadIdx.x; j < 2 * d; j += blockDim.x)
            a.gwb_part[(long)blockIdx.x * 2 * d + j] =
                (gsh[j] + gsh[2 * d + j]) + (gsh[4 * d + j] + gsh[6 * d + j]);
}

struct NMFinArgs {
    const double* stat_part; const float* gwb_part; int n_blocks;
    int B, K, d, L; double alpha; int reg;
    double* out;            // [8]: loss, loss_q, loss_p, KL_reg, NLL_E, RE_q mean, sum lse_q, sum lse_p
    float* gW; float* gb;   // [d] each (nullptr: skip)
    float* loss_f32;        // optional: out[0] as a float (slot of the data-parallel bucket)
    float* accum;           // optional: accum[0] += loss (epoch total, train.py:117, without a host sync)
    long long* state; long long rng_inc;  // optional per-step counters of a replayed graph
    int accumulate;
    double inv_B;           // 1 / (rows the means run over): B, or the global batch under data parallelism
};
// block 0: the loss terms; block c >= 1: columns [64 (c-1), 64 c) of the dW | db partials, 4 row groups per column
__global__ __launch_bounds__(256) void nm_finalize_kernel(NMFinArgs a) {
    __shared__ double red[256][NM_STATS];
    if (blockIdx.x > 0) {
        float (*gs)[64] = reinterpret_cast<float (*)[64]>(&red[0][0]);
        const int col = 64 * (blockIdx.x - 1) + (threadIdx.x & 63), grp = threadIdx.x >> 6, C = 2 * a.d;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (col < C) {
            int w = grp;
            for (; w + 12 < a.n_blocks; w += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] += a.gwb_part[(long)(w + 4 * u) * C + col];
            }
            for (; w < a.n_blocks; w += 4) acc[0] += a.gwb_part[(long)w * C + col];
        }
        gs[grp][threadIdx.x & 63] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        __syncthreads();
        if (threadIdx.x < 64 && col < C) {
            const float v = (gs[0][threadIdx.x] + gs[1][threadIdx.x]) + (gs[2][threadIdx.x] + gs[3][threadIdx.x]);
            float* dst = col < a.d ? a.gW + col : a.gb + (col - a.d);
            *dst = a.accumulate ? *dst + v : v;
        }
        return;
    }
    double s[NM_STATS] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < a.n_blocks; b += 256)
        for (int i = 0; i < NM_STATS; ++i) s[i] += a.stat_part[(long)b * NM_STATS + i];
    for (int i = 0; i < NM_STATS; ++i) red[threadIdx.x][i] = s[i];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int i = 0; i < NM_STATS; ++i) red[threadIdx.x][i] += red[threadIdx.x + o][i];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double logK = log((double)a.K);
        const double loss_q = red[0][0] * a.inv_B - logK * (a.B * a.inv_B);
        const double loss_p = a.reg ? red[0][1] * a.inv_B - logK * (a.B * a.inv_B) : 0.0;
        const double nll_e = red[0][2] * a.inv_B / a.K;
        const double kl_reg = red[0][3] * a.inv_B / a.L;
        a.out[0] = a.reg ? loss_q + a.alpha * (kl_reg - loss_q + loss_p + nll_e) : loss_q;
        a.out[1] = loss_q; a.out[2] = loss_p; a.out[3] = kl_reg; a.out[4] = nll_e;
        a.out[5] = red[0][4] * a.inv_B / a.K;
        a.out[6] = red[0][0]; a.out[7] = red[0][1];
        if (a.loss_f32) a.loss_f32[0] = (float)a.out[0];
        if (a.accum) a.accum[0] += (float)a.out[0];
        if (a.state) { a.state[0] += 1; a.state[1] += a.rng_inc; }
    }
}

// z[b,k,l] = mean[b,l] + eps[b,k,l] * exp(logvar[b,l] / 2)
__global__ void nm_sample_kernel(const float* __restrict__ heads, long ldh, const float* __restrict__ eps,
                                 float* __restrict__ z, long ldz, long B, int K, int L) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = B * K * L;
    if (i >= n) return;
    const int l = (int)(i % L);
    const long row = i / L, b = row / K;
    const float mu = heads[b * ldh + l], lv = heads[b * ldh + L + l];
    z[row * ldz + l] = eps ? mu + eps[i] * expf(0.5f * lv) : mu;
}
// d heads[b] = g_heads[b] + ( sum_k dz[b,k,:] | sum_k dz[b,k,:] * eps * exp(logvar/2) / 2 )
__global__ void nm_sample_bwd_kernel(const float* __restrict__ dz, long lddz, const float* __restrict__ eps,
                                     const float* __restrict__ heads, long ldh, const float* __restrict__ g_heads,
                                     long ldg, float* __restrict__ out, long ldo, long B, int K, int L) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * L) return;
    const int l = (int)(i % L);
    const long b = i / L;
    const float hs = 0.5f * expf(0.5f * heads[b * ldh + L + l]);
    // four replicas in flight (a serial walk over K = 20 was 20 dependent round trips: 9 us at batch 128); fixed association
    float sm[4] = {0.f, 0.f, 0.f, 0.f}, sl[4] = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 3 < K; k += 4) {
        float g[4], e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            g[u] = dz[(b * K + k + u) * lddz + l];
            e[u] = eps ? eps[(b * K + k + u) * L + l] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { sm[u] += g[u]; sl[u] += g[u] * e[u] * hs; }
    }
    for (; k < K; ++k) {
        const float g = dz[(b * K + k) * lddz + l];
        sm[0] += g;
        if (eps) sl[0] += g * eps[(b * K + k) * L + l] * hs;
    }
    const float tm = (sm[0] + sm[1]) + (sm[2] + sm[3]), tl = (sl[0] + sl[1]) + (sl[2] + sl[3]);
    out[b * ldo + l] = tm + (g_heads ? g_heads[b * ldg + l] : 0.f);
    out[b * ldo + L + l] = tl + (g_heads ? g_heads[b * ldg + L + l] : 0.f);
}
// out = x * mask  (encoder input, VAE.py:2379 / :2750)
__global__ void nm_mul_kernel(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ o, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = x[i] * m[i];
}

// The rows are walked by a grid-stride loop, so the grid is sized to ONE resident round of the variant actually
// launched (its register count decides 2 - 6 workgroups per CU): a grid of 4 workgroups per CU on a kernel that fits
// 3 runs two rounds and doubles the time at mid-size batches.
template <int T, bool REG, bool PF, int KS = 1>
static int launch_variant(const NMLossArgs& a, int* n_blocks, size_t lds, hipStream_t st) {
    // occupancy of this variant, per (device, dynamic-LDS size): obs_dim decides the LDS footprint
    static std::mutex mu;
    static std::map<std::pair<int, size_t>, int> cache;
    int dev = 0, occ = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find({dev, lds * 8 + KS});
        if (it != cache.end()) occ = it->second;
    }
    if (occ == 0) {
        int o = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, nm_loss_kernel<T, REG, PF, KS>, 256, lds) != hipSuccess || o < 1)
            o = 1;
        occ = o;
        std::lock_guard<std::mutex> g(mu);
        cache[{dev, lds * 8 + KS}] = o;
    }
    const long cap = (long)occ * num_cus();
    if (*n_blocks > cap) *n_blocks = (int)cap;
    hipLaunchKernelGGL((nm_loss_kernel<T, REG, PF, KS>), dim3(*n_blocks), dim3(256), lds, st, a);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}
constexpr long NM_KS_ROWS = 1024;  // batches up to this many rows: four waves per data row (nm_loss_kernel, KS = 4)
template <int T>
static int launch_loss(const NMLossArgs& a, int reg, int* n_blocks, hipStream_t st) {
    const size_t lds = (2 * 4 * NM_STATS + 3 * (size_t)a.d + 4 * 2 * (size_t)a.d + 2 * NM_KMAX + 4 * 2 * 64) * sizeof(float);
    if ((long)a.B <= NM_KS_ROWS && a.K <= NM_KMAX && !a.xm_imp) {  // one workgroup per data row
        *n_blocks = a.B;
        return reg ? launch_variant<T, true, true, 4>(a, n_blocks, lds, st) : launch_variant<T, false, true, 4>(a, n_blocks, lds, st);
    }
    if (*n_blocks > (a.B + 3) / 4) *n_blocks = (a.B + 3) / 4;  // one row per wave
    const bool pf = (long)a.B < 16L * num_cus();  // fewer rows than resident waves
    if (reg && pf) return launch_variant<T, true, true>(a, n_blocks, lds, st);
    if (reg) return launch_variant<T, true, false>(a, n_blocks, lds, st);
    if (pf) return launch_variant<T, false, true>(a, n_blocks, lds, st);
    return launch_variant<T, false, false>(a, n_blocks, lds, st);
}

}  // namespace vpc

using namespace vpc;

extern "C" {

int vpc_nm_loss_blocks(long B) {
    long blocks = B <= NM_KS_ROWS ? B : (B + 3) / 4;  // (small batches: one workgroup per data row)
    const long cap = 8L * num_cus();  // upper bound of one resident round (the launch trims it to the variant's occupancy)
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

long vpc_nm_loss_scratch(long B, int d) {  // bytes: per-workgroup statistics (doubles) + dW | db partials (floats)
    if (B <= 0 || d <= 0) return 0;
    const long blocks = vpc_nm_loss_blocks(B);
    return blocks * NM_STATS * 8 + blocks * 2 * d * 4;
}

int vpc_nm_loss(const float* x, const float* mask, const float* mask_p, const float* xm_q, const float* xl_q, long ld_q,
                const float* xm_p, const float* xl_p, long ld_p, const float* heads_q, const float* heads_p, long ldh,
                const float* W, const float* b, const float* eps_kl, float* g_xm_q, float* g_xl_q, long ldg_q,
                float* g_xm_p, float* g_xl_p, long ldg_p, float* g_heads_q, float* g_heads_p, long ldgh, float* gW,
                float* gb, int accumulate_wb, float* xm_imp, void* scratch, long scratch_bytes, double* out8,
                float* loss_f32, float* accum, long long* state, long long rng_inc, int gated, long B, long B_global,
                int K, int d, int L, double alpha, void* stream) {
    const int reg = mask_p != nullptr;
    if (!x || !mask || !xm_q || !xl_q || !heads_q || !W || !b || !scratch || !out8) return VPC_ERR_ARG;
    if (B <= 0 || K <= 0 || B * (long)K > 0x7fffff00L || B_global < B) return VPC_ERR_ARG;
    if (d <= 0 || d > 256 || L <= 0 || L > 64) return VPC_ERR_SHAPE;
    if (reg && (!xm_p || !xl_p || !heads_p)) return VPC_ERR_ARG;
    if (!reg && !eps_kl) return VPC_ERR_ARG;
    const bool grad = g_xm_q != nullptr;
    if (grad && (!g_xl_q || !g_heads_q || !gW || !gb || (reg && (!g_xm_p || !g_xl_p || !g_heads_p)))) return VPC_ERR_ARG;
    if (scratch_bytes < vpc_nm_loss_scratch(B, d) || (reinterpret_cast<uintptr_t>(scratch) & 7)) return VPC_ERR_ARG;
    int blocks = vpc_nm_loss_blocks(B);  // upper bound (sizes the scratch); the launch may use fewer
    NMLossArgs a{};
    a.x = x; a.m = mask; a.mp = mask_p; a.xm_q = xm_q; a.xl_q = xl_q; a.ld_q = ld_q; a.xm_p = xm_p; a.xl_p = xl_p;
    a.ld_p = ld_p; a.hq = heads_q; a.hp = heads_p; a.ldh = ldh; a.W = W; a.b = b; a.eps_kl = eps_kl;
    a.g_xm_q = g_xm_q; a.g_xl_q = g_xl_q; a.ldg_q = ldg_q; a.g_xm_p = g_xm_p; a.g_xl_p = g_xl_p; a.ldg_p = ldg_p;
    a.g_hq = g_heads_q; a.g_hp = g_heads_p; a.ldgh = ldgh; a.xm_imp = xm_imp;
    a.stat_part = reinterpret_cast<double*>(scratch);
    a.gwb_part = reinterpret_cast<float*>(a.stat_part + (long)blocks * NM_STATS);
    a.B = (int)B; a.K = K; a.d = d; a.L = L;
    const double al = reg ? alpha : 0.0, Bg = (double)B_global;
    a.oq = (float)((1.0 - al) / Bg); a.op = (float)(al / Bg); a.oe = (float)(al / (Bg * K)); a.cr = (float)(al / (Bg * L));
    a.kq = a.oq; a.kp = a.op; a.gated = gated;
    hipStream_t st = (hipStream_t)stream;
    int rc = d <= 64 ? launch_loss<1>(a, reg, &blocks, st) : d <= 128 ? launch_loss<2>(a, reg, &blocks, st)
                                                                       : launch_loss<4>(a, reg, &blocks, st);
    if (rc != VPC_OK) return rc;
    NMFinArgs f{};
    f.stat_part = a.stat_part; f.gwb_part = a.gwb_part; f.n_blocks = blocks; f.B = (int)B; f.K = K; f.d = d; f.L = L;
    f.alpha = al; f.reg = reg; f.out = out8; f.gW = grad ? gW : nullptr; f.gb = gb; f.accumulate = accumulate_wb;
    f.loss_f32 = loss_f32; f.accum = accum; f.state = state; f.rng_inc = rng_inc;
    f.inv_B = 1.0 / Bg;
    hipLaunchKernelGGL(nm_finalize_kernel, dim3(grad ? 1 + (2 * d + 63) / 64 : 1), dim3(256), 0, st, f);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

int vpc_nm_sample(const float* heads, long ldh, const float* eps, float* z, long ldz, long B, int K, int L,
                  void* stream) {
    if (!heads || !z || B <= 0 || K <= 0 || L <= 0 || ldh < 2 * L || ldz < L) return VPC_ERR_ARG;
    const long n = B * K * L;
    hipLaunchKernelGGL(nm_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, heads,
                       ldh, eps, z, ldz, B, K, L);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

int vpc_nm_sample_bwd(const float* dz, long lddz, const float* eps, const float* heads, long ldh, const float* g_heads,
                      long ldg, float* out, long ldo, long B, int K, int L, void* stream) {
    if (!dz || !heads || !out || B <= 0 || K <= 0 || L <= 0 || lddz < L || ldh < 2 * L || ldo < 2 * L)
        return VPC_ERR_ARG;
    const long n = B * L;
    hipLaunchKernelGGL(nm_sample_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dz,
                       lddz, eps, heads, ldh, g_heads, ldg, out, ldo, B, K, L);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

int vpc_nm_mul(const float* x, const float* mask, float* out, long n, void* stream) {
    if (!x || !mask || !out || n <= 0) return VPC_ERR_ARG;
    hipLaunchKernelGGL(nm_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mask,
                       out, n);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

}  // extern "C"
