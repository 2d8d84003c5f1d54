// Small-batch whole-step kernel, fp32 (gfx950): one workgroup of 8 waves per 16-ROW tile, the FEATURE tiles of every layer
// split over the waves ("N-split"), activations handed from layer to layer through LDS.
//
// Reference semantics: the Reg_VAE / vanilla_VAE training step of src/experiment_main/train.py:87-115 (src/models/VAE.py:496-507
// forward, :403-467 loss, autograd backward) - the same mathematics, fp32 v_mfma_f32_16x16x4_f32 products, weight images, loss
// coefficients and gradient partial-block layouts as vpc_encoder_fwd -> vpc_decoder_fused -> vpc_encoder_bwd with precision 0.
//
// Why.  The row-tiled kernels give a wave 16 batch rows and the whole MLP: a wave's life is the serial chain of all layers
// (~2 000 dependent MFMAs, 13 + 29 + 17 us for the three kernels at batch 64 - the reference's own batch size,
// Data/imputation_args.json - whatever the batch, with 8 of the chip's 1 024 SIMDs busy).  Here the 7 / 4 / 2 / 4 / 7 / 8 output
// tiles of a layer belong to different waves, so a layer is ~30 MFMAs deep instead of ~220 and the step is ONE launch; the price
// is a workgroup barrier per layer and that the weights are not LDS-resident: every wave reads the A fragments of its tile
// straight from the fp32 image in global memory (L2-resident, 193 KB; the row-tiled kernels' own image format, so Adam's
// re-pack serves both), requested one layer ahead.
// LDS: activations and their gradients of the tile as [16 rows][144] fp32 (features contiguous: the C-layout store of a tile
// is one ds_write_b128 per lane, the B operand of the next layer one ds_read_b128 per k-tile; the pitch of 144 dwords puts
// the four 4-row groups of a wgrad's dword reads on different banks).  wgrad contracts over the tile's 16 rows: four MFMAs
// per 16 x 16 gradient tile, operands read as dwords down the columns.
// Gradient accumulators live in registers across the tiles of a workgroup in the 8-wave slot layout of vpc_layout.h (the
// partial blocks are what vpc_reduce_step(_adam) expects).
#include "vpc_abi_internal.h"
#include "vpc_device.h"
#include "vpc_rng.h"
#include "vpc_dec_args.h"

namespace vpc {

constexpr int SP = 144;             // row pitch of the LDS activation buffers (dwords)
constexpr int SBUF = 16 * SP;       // one buffer: 16 rows
enum { B_XQ = 0, B_XP, B_H1Q, B_H1P, B_H2Q, B_H2P, B_ML, B_Z, B_G1, B_G2, B_DP, B_DG2, B_DG1, B_DMLQ, B_DMLP, B_DH2, B_DH1, B_COUNT };
constexpr int SMALL_LDS = (B_COUNT * SBUF + WAVES * LOSS_TERMS) * 4;
static_assert(SMALL_LDS <= 163840, "LDS budget");

struct SmallArgs {
    const float* x;
    const float* enc_img;
    const float* dec_img;
    const uint8_t* m[2];
    const uint8_t* mB[2];
    float cA[2], cE[2];
    const float* eps[2];
    const float* eps_ml;
    float* partE;
    float* partD;
    double* loss_part;
    float bq, bp, cr, wml, inv_B, x_logvar;
    long B;
    int d, L, npass, ntiles;
    // optional in-kernel draws of the step (vpc_step_small_draw_f32): the workgroup draws the mask_p bytes and the eps values of ITS
    // 16 rows with the counters vpc_draw_step would use, stores them where the step reads them (m[1], eps[...]) and goes on
    int draw;                    // 0: inputs are given; 1: draw eps (and mask_p when mask_in != NULL)
    const uint8_t* mask_in;      // mask that mask_p thins (NULL: no mask draw - vanilla_VAE)
    float keep_prob;
    float* eps_out; long n_eps;  // [planes][B][16] planes of normals, n_eps floats in all
    unsigned long long seed, off_mask, off_eps;
    const long long* state;
    long mask_elem_lo;
    EpsShard shard;
};

// C-layout tile <-> [row][feature] buffer
__device__ __forceinline__ f32x4 ld_act(const float* buf, int t, int c, int q) {
    return *reinterpret_cast<const f32x4*>(buf + c * SP + 16 * t + 4 * q);
}
__device__ __forceinline__ void st_act(float* buf, int t, int c, int q, f32x4 v) {
    *reinterpret_cast<f32x4*>(buf + c * SP + 16 * t + 4 * q) = v;
}
// gradient tile (out tile of A-buffer `da`, in tile of B-buffer `xb`) over the 16 rows: acc[m][n] += sum_row da[row][16 ta + m] * xb[row][16 tb + n]
__device__ __forceinline__ f32x4 wgrad16(const float* da, int ta, const float* xb, int tb, f32x4 acc, int m, int kq) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = VPC_MFMA(da[(4 * s + kq) * SP + 16 * ta + m], xb[(4 * s + kq) * SP + 16 * tb + m], acc);
    return acc;
}

// Weight fragments straight from the fp32 image in global memory (the layout of vpc_layout.h: row pitch S dwords, 16-byte slot
// XOR-swizzled with the row) - REQUESTED one stage ahead of their use (ldW / ldWT), consumed by mmW.  Forward: rows 16 mt + m,
// one 16-byte load per k-tile; transposed (dgrad): column 16 mt + m, four dword loads per k-tile (tile_fwd / tile_T of
// vpc_device.h, split into request and use).
template <int KT, int S>
__device__ __forceinline__ void ldW(const float* W, int mt, int m, int q, f32x4 (&a)[KT]) {
    constexpr int MASK = (S / 4 - 1) & 15;
    const float* rowp = W + (16 * mt + m) * S;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) a[kt] = *reinterpret_cast<const f32x4*>(rowp + 4 * ((4 * kt + q) ^ (m & MASK)));
}
template <int KT, int S, int NK = 4 * KT>
__device__ __forceinline__ void ldWT(const float* W, int mt, int m, int q, f32x4 (&a)[KT]) {
    constexpr int MASK = (S / 4 - 1) & 15;
    const int col = 16 * mt + m, cs = col >> 2, cl = col & 3;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 4 * q + j;
            a[kt][j] = (4 * kt + j < NK) ? W[(16 * kt + r) * S + (((cs ^ (r & MASK)) << 2) | cl)] : 0.f;
        }
}
template <int KT, int NK = 4 * KT>
__device__ __forceinline__ f32x4 mmW(const f32x4 (&a)[KT], const f32x4 (&in)[KT], f32x4 acc) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * kt + j < NK) acc = VPC_MFMA(a[kt][j], in[kt][j], acc);
    return acc;
}

template <int DT>
__global__ __launch_bounds__(THREADS) void step_small_kernel(SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    auto buf = [&](int b) { return lds + b * SBUF; };
    float* red = lds + B_COUNT * SBUF;
    constexpr int S1 = s_for_tiles(DT);
    const EncImg ei(DT);
    const DecImg di(DT);
    // (re-derived from opaque base pointers in every tile / pass: the weights never change during the launch, and hipcc
    // otherwise hoists the global fragment loads of ALL layers out of the loops - 900 bytes of scratch per lane)
    const float *W1, *b1, *W2, *W3, *W4, *W5, *W6;
    auto weights = [&]() {
        const float* e_ = a.enc_img;
        const float* d_ = a.dec_img;
        asm volatile("" : "+s"(e_), "+s"(d_)::"memory");
        W1 = e_ + ei.oW1; b1 = e_ + ei.ob1; W2 = e_ + ei.oW2; W3 = e_ + ei.oW3;
        W4 = d_ + di.oW4; W5 = d_ + di.oW5; W6 = d_ + di.oW6;
    };
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const bool two = a.npass == 2;
    const float inv_s2 = expf(-a.x_logvar), half_lv = 0.5f * a.x_logvar;
    constexpr float HL2PI = 0.91893853320467274f;

    // ONE 16-row tile per workgroup (the host launches one workgroup per tile): the decoder-side gradient accumulators live
    // through the decoder phases of both passes, are written out, and only then the encoder-side ones come to life for the
    // encoder backward of both passes - everything those need (x * mask, h1, h2 and the seeds of both passes) is still in LDS.
    // (All 100 accumulators beside two stages' worth of weight fragments do not fit 256 registers.)
    f32x4 acc6[H1T], acc5[H2T], acc4 = zero4();
#pragma unroll
    for (int t = 0; t < H1T; ++t) acc6[t] = zero4();
#pragma unroll
    for (int t = 0; t < H2T; ++t) acc5[t] = zero4();
    float S_A0 = 0.f, S_E0 = 0.f, S_A1 = 0.f, S_kl0q = 0.f, S_kl0p = 0.f, S_klr = 0.f, S_zll = 0.f;

    if (a.draw) {
        // ---- the step's draws for this tile's rows (same Philox counters and values as vpc_draw_step: vpc_rng.h)
        const long row0 = (long)blockIdx.x * 16;
        const long nrow = a.B - row0 < 16 ? a.B - row0 : 16;
        uint64_t off_m = a.off_mask, off_e = a.off_eps;
        if (a.state) { off_m += (uint64_t)a.state[1]; off_e += (uint64_t)a.state[1]; }
        if (a.mask_in) {  // mask_p bytes [row0 d, (row0 + nrow) d): every 8-byte Philox group that touches them (a group on a
                          // tile boundary is written by both neighbours - the same bytes)
            const long lo = row0 * a.d + (a.mask_elem_lo & 7), hi = (row0 + nrow) * a.d + (a.mask_elem_lo & 7);
            const long g0 = lo / MASK_PER_CALL, g1 = (hi + MASK_PER_CALL - 1) / MASK_PER_CALL;
            for (long g = g0 + threadIdx.x; g < g1; g += THREADS)
                draw_mask_body(a.mask_in, const_cast<uint8_t*>(a.m[1]), a.B * (long)a.d, a.keep_prob, a.seed, off_m, g, a.mask_elem_lo);
        }
        const long plane = a.B * 16, nplanes = a.n_eps / plane;
        for (long i = threadIdx.x; i < nplanes * nrow * 4; i += THREADS) {  // 4 groups of 4 normals per row and plane
            const long pl = i / (nrow * 4), rem = i - pl * nrow * 4;
            fill_normal_body(a.eps_out, a.n_eps, a.seed, off_e, (pl * plane + row0 * 16) / 4 + rem, a.shard);
        }
        __syncthreads();  // (a fence: the stores above are visible to the loads below)
    }
    {
        const int tile = blockIdx.x;
        const long row0 = (long)tile * 16;
        const bool ok = row0 + c < a.B;
        // ---- this wave's column tile of x and of the mask words of both passes (range-checked: rows past B read 0; the last
        // tile's columns past d read column 0 and have their mask words cleared)
        const bool colok = 16 * w + 4 * q + 3 < a.d;
        f32x4 xv = zero4();
        uint32_t mwq = 0, mwp = 0;
        if (w < DT) {
            const int vo = c * a.d + (colok ? 16 * w + 4 * q : 0);
            xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rows_rsrc(a.x, row0, a.B, a.d), 4 * vo, 0, 0));
            const long rem = (a.B - row0) * (long)a.d;
            const uint32_t rec = rem > 0xffffffffL ? 0xffffffffu : (uint32_t)rem;
            mwq = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.m[0]) + row0 * a.d, 0, rec, 0x00020000), vo, 0, 0);
            if (two)
                mwp = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.m[1]) + row0 * a.d, 0, rec, 0x00020000), vo, 0, 0);
            if (!colok) { mwq = 0; mwp = 0; }
        }
        auto ld_lat = [&](const float* base) -> f32x4 {  // [B][16] padded latent-width array; NULL reads 0
            return ld_rows(rows_rsrc(base ? base : a.x, row0, base ? a.B : row0, 16), c, 16, 4 * q);
        };
        int cc = c, qq = q;
        launder(cc, qq);
        // ================================================================ E: encoder forward of both passes
        // (every stage requests the weight fragments of the NEXT stage before it computes: a stage is ~30 MFMAs, an L2 round
        // trip as long as that; lds_barrier() leaves those loads in flight)
        for (int p = 0; p < a.npass; ++p) {
            weights();
            float* X = buf(p == 0 ? B_XQ : B_XP);
            float* H1b = buf(p == 0 ? B_H1Q : B_H1P);
            float* H2b = buf(p == 0 ? B_H2Q : B_H2P);
            f32x4 fW1[DT], fW2[H1T], fW3[H2T], bias1 = zero4();
            if (w < H1T) {
                ldW<DT, S1>(W1, w, cc, qq, fW1);
                bias1 = *reinterpret_cast<const f32x4*>(b1 + 16 * w + 4 * qq);
            }
            if (w < DT) st_act(X, w, cc, qq, xv * mask_to_f32(p == 0 ? mwq : mwp));  // x.float() * mask  (VAE.py:388)
            if (w < H2T) ldW<H1T, 128>(W2, w, cc, qq, fW2);
            lds_barrier();
            launder(cc, qq);
            if (w < H1T) {
                f32x4 in[DT];
#pragma unroll
                for (int t = 0; t < DT; ++t) in[t] = ld_act(X, t, cc, qq);
                st_act(H1b, w, cc, qq, relu4(mmW<DT>(fW1, in, bias1)));
            }
            if (w < 2) ldW<H2T, 64>(W3, w, cc, qq, fW3);
            lds_barrier();
            launder(cc, qq);
            if (w < H2T) {
                f32x4 in[H1T];
#pragma unroll
                for (int t = 0; t < H1T; ++t) in[t] = ld_act(H1b, t, cc, qq);
                st_act(H2b, w, cc, qq, relu4(mmW<H1T, NK1>(fW2, in, zero4())));
            }
            lds_barrier();
            launder(cc, qq);
            if (w < 2) {  // wave 0: mean tile, wave 1: logvar tile -> ML tiles 2 p, 2 p + 1
                f32x4 in[H2T];
#pragma unroll
                for (int t = 0; t < H2T; ++t) in[t] = ld_act(H2b, t, cc, qq);
                f32x4 o = mmW<H2T, NK2>(fW3, in, zero4());
                if (!ok) o = zero4();  // rows past B: statistics 0
                st_act(buf(B_ML), 2 * p + w, cc, qq, o);
            }
        }
        lds_barrier();
        launder(cc, qq);
        // ================================================================ per pass: decoder, loss, all backward
        for (int p = 0; p < a.npass; ++p) {
            weights();
            const float* X = buf(p == 0 ? B_XQ : B_XP);
            const float* H1b = buf(p == 0 ? B_H1Q : B_H1P);
            const float* H2b = buf(p == 0 ? B_H2Q : B_H2P);
            f32x4 fW4[1], fW5[H2T], fW6[H1T], fT6[DT], fT5[H1T], fT4[H2T];
            float* DML = buf(p == 0 ? B_DMLQ : B_DMLP);
            if (w < H2T) ldW<1, S4>(W4, w, cc, qq, fW4);
            if (w < H1T) ldW<H2T, 64>(W5, w, cc, qq, fW5);
            const f32x4 mu = ld_act(buf(B_ML), 2 * p, cc, qq), lv = ld_act(buf(B_ML), 2 * p + 1, cc, qq);
            const f32x4 e = ld_lat(a.eps[p]);
            if (w == 0) {  // z = mean + eps * exp(logvar / 2); z[L] = 1 drives the bias chain
                f32x4 z;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    z[j] = mu[j] + ((4 * qq + j < a.L) ? e[j] : 0.f) * __expf(0.5f * lv[j]);
                    if (4 * qq + j == a.L) z[j] = 1.f;
                }
                st_act(buf(B_Z), 0, cc, qq, z);
            }
            lds_barrier();
            launder(cc, qq);
            if (w < H2T) {
                const f32x4 in[1] = {ld_act(buf(B_Z), 0, cc, qq)};
                st_act(buf(B_G1), w, cc, qq, relu4(mmW<1>(fW4, in, zero4())));
            }
            if (w < DT) ldW<H1T, 128>(W6, w, cc, qq, fW6);
            lds_barrier();
            launder(cc, qq);
            if (w < H1T) {
                f32x4 in[H2T];
#pragma unroll
                for (int t = 0; t < H2T; ++t) in[t] = ld_act(buf(B_G1), t, cc, qq);
                st_act(buf(B_G2), w, cc, qq, relu4(mmW<H2T, NK2>(fW5, in, zero4())));
            }
            lds_barrier();
            launder(cc, qq);
            if (w < DT) {  // output tile w: forward, loss terms, d / d pre-activation
                f32x4 in[H1T];
#pragma unroll
                for (int t = 0; t < H1T; ++t) in[t] = ld_act(buf(B_G2), t, cc, qq);
                const f32x4 pre = mmW<H1T, NK1>(fW6, in, zero4());
                const uint32_t ua = p == 0 ? mwq : mwp;
                const uint32_t ub = a.mB[p] ? (p == 0 ? mwp : mwq) : ua;  // (host: the second mask is the other pass's)
                const f32x4 mA = mask_to_f32(ua), mE = mask_to_f32(ua & ~ub);
                const float kA = a.cA[p] * inv_s2 * a.inv_B, kE = a.cE[p] * inv_s2 * a.inv_B, hinv_s2 = 0.5f * inv_s2;
                f32x4 dp;
                float sa = 0.f, se = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = fast_sigmoid(pre[j]);
                    const float diff = xh - xv[j];
                    const float t = diff * diff * hinv_s2 + half_lv;
                    sa += mA[j] * t;
                    se += mE[j] * t;
                    dp[j] = (kA * mA[j] + kE * mE[j]) * diff * (xh - xh * xh);
                }
                if (p == 0) { S_A0 += sa; S_E0 += se; } else { S_A1 += sa; }
                st_act(buf(B_DP), w, cc, qq, dp);
            }
            if (w < H1T) ldWT<DT, 128>(W6, w, cc, qq, fT6);  // dg2's fragments
            lds_barrier();
            launder(cc, qq);
            // ---- dW6~ (wave w: out tile w, 7 in tiles)  |  dg2 = relu'(g2) * (W6~^T dpre) (waves 0-6: tile w)
            if (w < DT) {
#pragma unroll
                for (int nt = 0; nt < H1T; ++nt) acc6[nt] = wgrad16(buf(B_DP), w, buf(B_G2), nt, acc6[nt], cc, qq);
            }
            if (w < H1T) {
                f32x4 in[DT];
#pragma unroll
                for (int t = 0; t < DT; ++t) in[t] = ld_act(buf(B_DP), t, cc, qq);
                st_act(buf(B_DG2), w, cc, qq, gate4(mmW<DT>(fT6, in, zero4()), ld_act(buf(B_G2), w, cc, qq)));
            }
            if (w < H2T) ldWT<H1T, 64, NK1>(W5, w, cc, qq, fT5);  // dg1's fragments
            lds_barrier();
            launder(cc, qq);
            // ---- dW5~ (wave w: in tile w & 3 of out tiles 4 (w >> 2) .. + 3)  |  dg1 (waves 0-3)
            {
                const int nt5 = w & 3, mt5 = 4 * (w >> 2);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < 3 || w < 4) acc5[i] = wgrad16(buf(B_DG2), mt5 + i, buf(B_G1), nt5, acc5[i], cc, qq);
            }
            if (w < H2T) {
                f32x4 in[H1T];
#pragma unroll
                for (int t = 0; t < H1T; ++t) in[t] = ld_act(buf(B_DG2), t, cc, qq);
                st_act(buf(B_DG1), w, cc, qq, gate4(mmW<H1T, NK1>(fT5, in, zero4()), ld_act(buf(B_G1), w, cc, qq)));
            }
            if (w == 4) ldWT<H2T, S4, NK2>(W4, 0, cc, qq, fT4);  // dz's fragments
            lds_barrier();
            launder(cc, qq);
            // ---- dW4~ (waves 0-3: out tile w)  |  wave 4: dz, KL terms, seeds on (mean | logvar) -> DML
            if (w < H2T) acc4 = wgrad16(buf(B_DG1), w, buf(B_Z), 0, acc4, cc, qq);
            if (w == 4) {
                f32x4 in[H2T];
#pragma unroll
                for (int t = 0; t < H2T; ++t) in[t] = ld_act(buf(B_DG1), t, cc, qq);
                const f32x4 dz = mmW<H2T, NK2>(fT4, in, zero4());
                const f32x4 mo = two ? ld_act(buf(B_ML), 2 * (1 - p), cc, qq) : zero4();
                const f32x4 lo = two ? ld_act(buf(B_ML), 2 * (1 - p) + 1, cc, qq) : zero4();
                f32x4 dmu, dlv;
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
                    const float dl = b0 * 0.5f * (elv - 1.f) + crr * 0.5f * ((p == 0) ? (r - 1.f) : (1.f - r - diff * diff * eip));
                    dmu[j] = dm * a.inv_B;
                    dlv[j] = dl * a.inv_B;
                }
                if (p == 0) { S_kl0q += kl0; if (two) S_klr += klr; } else { S_kl0p += kl0; }
                if (two && a.wml != 0.f) {  // ml_reg: extra rsample z' of q scored under p (VAE.py:435-440)
                    const f32x4 e3 = ld_lat(a.eps_ml);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool live = ok && 4 * qq + j < a.L;
                        const float e3j = (4 * qq + j < a.L) ? e3[j] : 0.f;
                        const float mq = (p == 0) ? mu[j] : mo[j], lq = (p == 0) ? lv[j] : lo[j];
                        const float mp = (p == 0) ? mo[j] : mu[j], lp = (p == 0) ? lo[j] : lv[j];
                        const float sq = __expf(0.5f * lq), eip = __expf(-lp);
                        const float dlt = mq + e3j * sq - mp;
                        const float g = a.wml * dlt * eip * a.inv_B;
                        if (p == 0) {
                            if (live) S_zll += -HL2PI - 0.5f * lp - 0.5f * dlt * dlt * eip;
                            dmu[j] += g;
                            dlv[j] += g * e3j * 0.5f * sq;
                        } else {
                            dmu[j] -= g;
                            dlv[j] += live ? a.wml * (0.5f - 0.5f * dlt * dlt * eip) * a.inv_B : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ef = (4 * qq + j < a.L) ? e[j] * 0.5f * __expf(0.5f * lv[j]) : 0.f;
                    dmu[j] = (4 * qq + j < a.L) ? dmu[j] + dz[j] : 0.f;  // columns >= L carry no gradient (dz's column L is db4)
                    dlv[j] = (4 * qq + j < a.L) ? dlv[j] + dz[j] * ef : 0.f;
                }
                st_act(DML, 0, cc, qq, dmu);
                st_act(DML, 1, cc, qq, dlv);
            }
            lds_barrier();
            launder(cc, qq);
        }
        // ================================================================ decoder partial block and the loss terms
        {
            float* part = a.partD + (long)blockIdx.x * DEC_PART + (long)(w & 3) * DEC_GREGS * 64 + lane;
            const int hi = w >> 2;
#pragma unroll
            for (int nt = 0; nt < H1T; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) part[(28 * hi + 4 * nt + j) * 64] = (w < DT) ? acc6[nt][j] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* p5 = a.partD + (long)blockIdx.x * DEC_PART + (long)i * DEC_GREGS * 64 + lane;
#pragma unroll
                for (int j = 0; j < 4; ++j) p5[(56 + 16 * hi + 4 * (w & 3) + j) * 64] = (i < 3 || w < 4) ? acc5[i][j] : 0.f;
            }
            if (w < H2T) {
#pragma unroll
                for (int j = 0; j < 4; ++j) part[(88 + j) * 64] = acc4[j];
            }
            const float s[LOSS_TERMS] = {S_A0, S_E0, S_A1, S_kl0q, S_kl0p, S_klr, S_zll, 0.f};
#pragma unroll
            for (int i = 0; i < LOSS_TERMS; ++i) {
                const float v = wave_sum_dpp(s[i]);
                if (lane == 0) red[w * LOSS_TERMS + i] = v;
            }
            lds_barrier();
            if (threadIdx.x < LOSS_TERMS) {
                double t = 0.0;
                for (int k = 0; k < WAVES; ++k) t += (double)red[k * LOSS_TERMS + threadIdx.x];
                a.loss_part[(long)blockIdx.x * LOSS_TERMS + threadIdx.x] = t;
            }
        }
        // ================================================================ encoder backward of both passes
        f32x4 acc1[H1T], acc2[H2T], acc3 = zero4(), dbacc = zero4();
#pragma unroll
        for (int t = 0; t < H1T; ++t) acc1[t] = zero4();
#pragma unroll
        for (int t = 0; t < H2T; ++t) acc2[t] = zero4();
        for (int p = 0; p < a.npass; ++p) {
            weights();
            const float* X = buf(p == 0 ? B_XQ : B_XP);
            const float* H1b = buf(p == 0 ? B_H1Q : B_H1P);
            const float* H2b = buf(p == 0 ? B_H2Q : B_H2P);
            const float* DML = buf(p == 0 ? B_DMLQ : B_DMLP);
            f32x4 fT3[2], fT2[H2T];
            if (w < H2T) ldWT<2, 64>(W3, w, cc, qq, fT3);    // dh2's fragments
            if (w < H1T) ldWT<H2T, 128, NK2>(W2, w, cc, qq, fT2);  // dh1's fragments
            // ---- dW3~ (wave w: out tile w >> 2, in tile w & 3)  |  dh2 (waves 0-3)
            acc3 = wgrad16(DML, w >> 2, H2b, w & 3, acc3, cc, qq);
            if (w < H2T) {
                const f32x4 in[2] = {ld_act(DML, 0, cc, qq), ld_act(DML, 1, cc, qq)};
                st_act(buf(B_DH2), w, cc, qq, gate4(mmW<2>(fT3, in, zero4()), ld_act(H2b, w, cc, qq)));
            }
            lds_barrier();
            launder(cc, qq);
            // ---- dW2~ (waves 0-6: in tile w, 4 out tiles)  |  dh1 (waves 0-6), db1 += column sums of dh1
            if (w < H1T) {
#pragma unroll
                for (int mt = 0; mt < H2T; ++mt) acc2[mt] = wgrad16(buf(B_DH2), mt, H1b, w, acc2[mt], cc, qq);
                f32x4 in[H2T];
#pragma unroll
                for (int t = 0; t < H2T; ++t) in[t] = ld_act(buf(B_DH2), t, cc, qq);
                const f32x4 dh1 = gate4(mmW<H2T, NK2>(fT2, in, zero4()), ld_act(H1b, w, cc, qq));
                st_act(buf(B_DH1), w, cc, qq, dh1);
                dbacc += dh1;  // per-lane (row c) running sums; the sum over the rows happens once, at the end
            }
            lds_barrier();
            launder(cc, qq);
            // ---- dW1 (wave w < DT: in tile w, 7 out tiles)
            if (w < DT) {
#pragma unroll
                for (int mt = 0; mt < H1T; ++mt) acc1[mt] = wgrad16(buf(B_DH1), mt, X, w, acc1[mt], cc, qq);
            }
            lds_barrier();
            launder(cc, qq);
        }
        // ================================================================ encoder partial block
        {
            float* part = a.partE + (long)blockIdx.x * ENC_PART + (long)w * GREGS * 64 + lane;
#pragma unroll
            for (int mt = 0; mt < H1T; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) part[(4 * mt + j) * 64] = (w < DT) ? acc1[mt][j] : 0.f;
#pragma unroll
            for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) part[(28 + 4 * mt + j) * 64] = (w < H1T) ? acc2[mt][j] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(44 + j) * 64] = acc3[j];
            // db1[16 w + 4 q + j] = sum over the 16 rows (lanes c) of dbacc: DPP butterfly inside each 16-lane row
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = dbacc[j];
                v += dpp_mov<0xB1>(v);
                v += dpp_mov<0x4E>(v);
                v += dpp_mov<0x141>(v);
                v += dpp_mov<0x140>(v);
                if (c == 0 && w < H1T) a.partE[(long)blockIdx.x * ENC_PART + WAVES * GREGS * 64 + 16 * w + 4 * q + j] = v;
            }
            if (w == 7 && lane < 16) a.partE[(long)blockIdx.x * ENC_PART + WAVES * GREGS * 64 + 112 + lane] = 0.f;
        }
    }
}

}  // namespace vpc

using namespace vpc;

// Rows up to which the library runs an fp32 step through vpc_step_small_f32 (0 = never): by default one round of workgroups
// (16 rows x CUs).  The kernel takes up to 2 x CUs tiles (the partial-block buffers' size).  VPC_TILE=16 forces this path for every batch, VPC_TILE=64 / 128 (the
// row-tiled workgroup shapes) switch it off, VPC_STEP_SMALL=n sets the row limit (A/B runs, tests).
extern "C" long vpc_step_small_max_rows(void) {
    long lim = 16L * num_cus();
    if (const char* e = getenv("VPC_TILE")) {
        const int v = atoi(e);
        if (v == 16) lim = 32L * num_cus();
        if (v == 64 || v == 128) lim = 0;
    }
    if (const char* e = getenv("VPC_STEP_SMALL")) lim = atol(e);
    return lim;
}

namespace vpc {
struct SmallDraw {
    int on; const uint8_t* mask_in; float keep_prob; float* eps_out; long n_eps; unsigned long long seed, off_mask, off_eps;
    const long long* state; long mask_elem_lo; EpsShard shard;
};
static int step_small_launch(const float* x, const float* enc_img, const float* dec_img, int npass,
                             const uint8_t* const* mask, const uint8_t* const* maskB, const float* cA, const float* cE,
                             const float* const* eps, const float* eps_ml, float bq, float bp, float cr, float wml,
                             float inv_B, float x_logvar, float* partE, float* partD, double* loss_partials,
                             int* nblocks_out, long B, int d, int L, const SmallDraw& dr, void* stream) {
    if (!x || !enc_img || !dec_img || !mask || !cA || !cE || !eps || !partE || !partD || !loss_partials) return VPC_ERR_ARG;
    if (npass < 1 || npass > 2 || B <= 0) return VPC_ERR_ARG;
    if (d < 4 || d > MAX_D || d % 4 || L < 1 || L > MAX_L) return VPC_ERR_SHAPE;
    if (!aligned16(x) || !aligned16(enc_img) || !aligned16(dec_img)) return VPC_ERR_ARG;
    if (wml != 0.f && !eps_ml) return VPC_ERR_ARG;
    SmallArgs a{};
    a.x = x; a.enc_img = enc_img; a.dec_img = dec_img; a.eps_ml = eps_ml; a.partE = partE; a.partD = partD;
    a.loss_part = loss_partials;
    a.bq = bq; a.bp = bp; a.cr = cr; a.wml = wml; a.inv_B = inv_B; a.x_logvar = x_logvar;
    a.B = B; a.d = d; a.L = L; a.npass = npass;
    for (int p = 0; p < npass; ++p) {
        if (!mask[p] || !eps[p]) return VPC_ERR_ARG;
        a.m[p] = mask[p]; a.mB[p] = maskB ? maskB[p] : nullptr; a.cA[p] = cA[p]; a.cE[p] = cE[p]; a.eps[p] = eps[p];
        if ((uintptr_t)a.m[p] % 4 || !aligned16(a.eps[p])) return VPC_ERR_ARG;
    }
    for (int p = 0; p < npass; ++p)  // the second loss mask of a pass must be the other pass's mask (as vpc_step_fused_bf16)
        if (a.mB[p] && (npass != 2 || a.mB[p] != a.m[1 - p])) return VPC_ERR_ARG;
    if ((B + 15) / 16 > 2L * num_cus()) return VPC_ERR_SHAPE;  // one workgroup per 16-row tile, at most 2 x CUs partial blocks
    if (dr.on) {
        // eps_out must be the planes the step reads: plane p = eps[p] (and eps_ml the next one), [B][16] each
        if (!dr.eps_out || dr.n_eps < B * 16 || dr.n_eps % (B * 16) || dr.eps_out != eps[0]) return VPC_ERR_ARG;
        if (npass == 2 && (eps[1] != dr.eps_out + B * 16 || dr.n_eps < 2 * B * 16)) return VPC_ERR_ARG;
        if (eps_ml && (eps_ml != dr.eps_out + 2 * B * 16 || dr.n_eps < 3 * B * 16)) return VPC_ERR_ARG;
        if (dr.mask_in && npass != 2) return VPC_ERR_ARG;
        if (dr.mask_elem_lo < 0) return VPC_ERR_ARG;
        a.draw = 1; a.mask_in = dr.mask_in; a.keep_prob = dr.keep_prob; a.eps_out = dr.eps_out; a.n_eps = dr.n_eps;
        a.seed = dr.seed; a.off_mask = dr.off_mask; a.off_eps = dr.off_eps; a.state = dr.state;
        a.mask_elem_lo = dr.mask_elem_lo; a.shard = dr.shard;
    }
    a.ntiles = (int)((B + 15) / 16);
    const int grid = a.ntiles;
    if (nblocks_out) *nblocks_out = grid;
    hipStream_t s = (hipStream_t)stream;
#define VPC_CASE(T)                                                                                        \
    case T: {                                                                                              \
        auto kern = step_small_kernel<T>;                                                                  \
        if (!lds_attr_done(reinterpret_cast<const void*>(kern), SMALL_LDS)) return VPC_ERR_HIP;            \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), SMALL_LDS, s, a);                              \
        return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;                                     \
    }
    switch (dt_for(d)) { VPC_CASE(1) VPC_CASE(2) VPC_CASE(4) VPC_CASE(8) }
#undef VPC_CASE
    return VPC_ERR_SHAPE;
}
}  // namespace vpc

extern "C" int vpc_step_small_f32(const float* x, const float* enc_img, const float* dec_img, int npass,
                                  const uint8_t* const* mask, const uint8_t* const* maskB, const float* cA, const float* cE,
                                  const float* const* eps, const float* eps_ml, float bq, float bp, float cr, float wml,
                                  float inv_B, float x_logvar, float* partE, float* partD, double* loss_partials,
                                  int* nblocks_out, long B, int d, int L, void* stream) {
    return step_small_launch(x, enc_img, dec_img, npass, mask, maskB, cA, cE, eps, eps_ml, bq, bp, cr, wml, inv_B, x_logvar, partE,
                             partD, loss_partials, nblocks_out, B, d, L, SmallDraw{}, stream);
}

// vpc_draw_step + vpc_step_small_f32 in ONE launch: every workgroup draws the mask_p bytes (mask[1] = mask_in & keep, when mask_in !=
// NULL) and the normals of ITS 16 rows - eps_out = eps[0], planes [B][16]: eps[1] and eps_ml follow it; n_eps floats in all - with the
// Philox counters vpc_draw_step would use (seed, offsets, state, mask_elem_lo and the eps_* shard description as there), stores them
// where the step reads them, and runs the step.  Same draws, one launch less at the batch sizes where a launch is a fifth of the step.
extern "C" int vpc_step_small_draw_f32(const float* x, const float* enc_img, const float* dec_img, int npass,
                                       const uint8_t* const* mask, const uint8_t* const* maskB, const float* cA, const float* cE,
                                       const float* const* eps, const float* eps_ml, float bq, float bp, float cr, float wml,
                                       float inv_B, float x_logvar, float* partE, float* partD, double* loss_partials,
                                       int* nblocks_out, long B, int d, int L, const uint8_t* mask_in, float keep_prob,
                                       float* eps_out, long n_eps, unsigned long long seed, unsigned long long offset_mask,
                                       unsigned long long offset_eps, const long long* state, long mask_elem_lo,
                                       long eps_rows_local, long eps_rows_global, long eps_row_lo, int eps_pitch, void* stream) {
    if (eps_rows_local < 0 || (eps_rows_local > 0 && (eps_pitch != 16 || eps_rows_local != B || eps_row_lo < 0 ||
                                                      eps_row_lo + eps_rows_local > eps_rows_global)))
        return VPC_ERR_ARG;
    SmallDraw dr{1, mask_in, keep_prob, eps_out, n_eps, seed, offset_mask, offset_eps, state, mask_elem_lo,
                 EpsShard{eps_rows_local, eps_rows_global, eps_row_lo, eps_pitch}};
    return step_small_launch(x, enc_img, dec_img, npass, mask, maskB, cA, cE, eps, eps_ml, bq, bp, cr, wml, inv_B, x_logvar, partE,
                             partD, loss_partials, nblocks_out, B, d, L, dr, stream);
}
