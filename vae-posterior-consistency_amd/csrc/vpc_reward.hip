// Active-variable-selection reward (BASELINE config 5), gfx950.
//
// Reference: R_lindley_chain / chaini_I / chaini_II, src/experiment_main/evaluate.py:514-634, driven by the
// candidate loop of active_learning_func (evaluate.py:424-433):  for every row n, candidate feature u (not yet
// observed) and MC imputation m
//     R[n][u] = 1/M sum_m [ KL_I(n,u,m) - KL_II(n,u,m) ],
//     KL_*  = 0.5 sum_L ( (mu_b - mu_a)^2 / exp(lv_a / 2) + exp(lv_b) / exp(lv_a) - 1 - lv_b + lv_a )      (sic: std)
// where (a, b) are two encoder calls that differ by revealing feature u (and, for KL_II, the target column in
// both).  The reference issues 4 * (d-1) * M encoder calls per acquisition step; here
//   * the "a" encodings do not depend on u      -> computed once per (n, m)   (mode A of the chain kernel)
//   * every encoding differs from the row's base encoding by a rank-1 update of the first layer
//       h1pre = W1 (x * mask) + b1 + W1[:,u] * im[m][n][u] (+ W1[:,T] * delta_T)
//     so layer 1 is an FMA per hidden unit and only layers 2-3 (100 -> 50 -> 2L) run on the matrix cores,
//     register-chained exactly like the training kernels (vpc_device.h), 16 MC samples per MFMA column tile,
//     two chains (I and II) per weight fragment.
#include "vpc_device.h"
#include "vpc_abi_internal.h"

namespace vpc {

constexpr int RW_WAVES = 4, RW_THREADS = RW_WAVES * 64;
constexpr int STAT = 64;  // floats per (n, m): [chain I | chain II] x [mean tile 16 | logvar tile 16]

// ---- prep: base first-layer pre-activations per (n, m) and chain, and W1^T for the rank-1 updates
//   pre[n][m][chain][112]: chain 0 (I) = base + W1[:,T] * mask_T * (xT_carry(m) - x_T),   xT_carry(0) = x_T,
//                                         xT_carry(m) = im[m-1][n][T]   (temp_x[loc,-1] is not reset, evaluate.py:531-536)
//                          chain 1 (II) = base + W1[:,T] * (im[m][n][T] - x_T * mask_T)
//   hidden unit f sits at position pos1_full(f) of the 112-wide rows (vpc_layout.h); unit 100 is the constant 1 of the
//   bias chain, the padding positions are 0.
__global__ __launch_bounds__(128) void reward_prep_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                          const float* __restrict__ im, const float* __restrict__ W1,
                                                          const float* __restrict__ b1, float* __restrict__ pre,
                                                          float* __restrict__ W1T, int* __restrict__ cand,
                                                          float* __restrict__ R, int n, int d, int M, int Mp) {
    if (blockIdx.x == 0 && threadIdx.x == 0) cand[(long)n * d] = 0;  // work counter of reward_chain_kernel<1>
    const int f = threadIdx.x;  // 0..127: hidden unit (100 = constant, 101..111 = padding), >= 112 idle
    const int pf = f < H1P ? pos1_full(f) : 0;
    if (blockIdx.x >= (unsigned)n) {  // trailing blocks: W1T[u][f] = W1[f][u], 8 input columns per block
        const int u0 = 8 * ((int)blockIdx.x - n);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int u = u0 + k;
            if (u < d && f < H1P) W1T[u * H1P + pf] = f < H1 ? W1[f * d + u] : 0.f;
        }
        return;
    }
    const int r = blockIdx.x, T = d - 1;
    // the row's candidates: cand[r][0] = their number, cand[r][1 + k] = the k-th feature u < d - 1 that is not observed yet, in
    // ascending order; observed features get the reference's R = -1e4 here (evaluate.py:424-433)
    {
        __shared__ int cnt0;
        const int u = threadIdx.x;
        const bool isc = u < T && !mask[(long)r * d + u];
        const unsigned long long bal = __ballot(isc);
        if (threadIdx.x == 0) cnt0 = __popcll(bal);
        __syncthreads();
        const int before = __popcll(bal & ((1ull << (threadIdx.x & 63)) - 1ull)) + (threadIdx.x >= 64 ? cnt0 : 0);
        int* cr = cand + (long)r * d;
        if (isc) cr[1 + before] = u;
        if (u < T && !isc) R[(long)r * T + u] = -1e4f;
        if (threadIdx.x == 127) cr[0] = before + (isc ? 1 : 0);
        __syncthreads();
    }
    // base[f] = b1[f] + sum_i W1[f][i] x[r][i] mask[r][i]: each wave takes every second unit, lanes over i (the rows of W1 are read
    // coalesced; one thread per unit walked its row with a stride of d floats: 128 dependent loads, 35 us for this launch)
    __shared__ float base_sh[H1P];
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        float xm[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = lane + 64 * k;
            xm[k] = i < d ? x[(long)r * d + i] * (mask[(long)r * d + i] ? 1.f : 0.f) : 0.f;
        }
        for (int u = wv; u < H1; u += 2) {
            float sacc = 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = lane + 64 * k;
                if (i < d) sacc += W1[u * d + i] * xm[k];
            }
            sacc = wave_sum_dpp(sacc);
            if (lane == 0) base_sh[u] = b1[u] + sacc;
        }
    }
    __syncthreads();
    if (f >= H1P) return;
    float base = 0.f, wT = 0.f;
    if (f < H1) {
        base = base_sh[f];
        wT = W1[f * d + T];
    }
    const float xT = x[(long)r * d + T], mT = mask[(long)r * d + T] ? 1.f : 0.f;
    for (int m = 0; m < Mp; ++m) {
        float p1 = 0.f, p2 = 0.f;
        if (m < M) {
            const float carry = m == 0 ? xT : im[((long)(m - 1) * n + r) * d + T];
            p1 = base + wT * mT * (carry - xT);
            p2 = base + wT * (im[((long)m * n + r) * d + T] - xT * mT);
        }
        if (f == H1) p1 = p2 = 1.f;
        if (f > H1) p1 = p2 = 0.f;
        float* o = pre + (((long)r * Mp + m) * 2) * H1P;
        o[pf] = p1;
        o[H1P + pf] = p2;
    }
}

struct RewardArgs {
    const float* img;     // encoder image (W2, W3 are used)
    const float* pre;     // [n][Mp][2][112]
    const float* W1T;     // [d][112]
    const float* im;      // [M][n][d]
    const uint8_t* mask;  // [n][d]
    const int* cand;      // [n][d]: count, then the row's candidate features (reward_prep_kernel)
    int* next_item;       // MODE 1 work counter (zeroed by reward_prep_kernel)
    float* stat;          // [n][Mp][64]
    float* R;             // [n][d-1]
    int n, d, L, M, Mp;
};

// MODE 0 (A): items = rows; writes stat[n][m] = {mean_I, logvar_I, mean_II, logvar_II} (16-float tiles)
// MODE 1 (B): items = (row, chunk of RW_CH candidates); reads stat, writes R.  A chunk's (candidate, sample) pairs are FLATTENED
// into 16-column MFMA tiles: column c of tile t is pair f = 16 t + c -> candidate f / M, sample f % M.  (Per candidate the M
// samples padded to a multiple of 16 issued 64 columns for M = 50 - 22 % of the matrix work on padding, 50 % at M = 8; a full
// chunk of 8 candidates x 50 samples is exactly 25 tiles.)
constexpr int RW_CH = 4;
template <int MODE>
__global__ __launch_bounds__(RW_THREADS) void reward_chain_kernel(RewardArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const EncImg imd(dt_for(a.d));
    const int nW = imd.total - imd.oW2;
    load_image(lds, a.img + imd.oW2, nW);
    const float* W2 = lds;
    const float* W3 = lds + (imd.oW3 - imd.oW2);
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int nch = (a.d - 1 + RW_CH - 1) / RW_CH;  // chunks per row (upper bound)
    const long nitems = MODE == 0 ? a.n : (long)a.n * nch;
    const float invM = 1.f / (float)a.M;

    long item = (long)blockIdx.x * RW_WAVES + w;
    auto next = [&]() {
        if (MODE == 0) { item += (long)gridDim.x * RW_WAVES; return; }
        int v = 0;
        if (lane == 0) v = atomicAdd(a.next_item, 1);
        item = (long)gridDim.x * RW_WAVES + __builtin_amdgcn_readfirstlane(v);  // (the first round is the launch's own grid)
    };
    for (; item < nitems; next()) {
        const int r = MODE == 0 ? (int)item : (int)(item / nch);
        const int ch = MODE == 0 ? 0 : (int)(item % nch);
        int ncand = 0;
        const int* cl = a.cand + (long)r * a.d + 1 + RW_CH * ch;
        if (MODE == 1) {
            ncand = a.cand[(long)r * a.d] - RW_CH * ch;
            if (ncand <= 0) continue;
            if (ncand > RW_CH) ncand = RW_CH;
        }
        const int total = MODE == 0 ? a.Mp : ncand * a.M;  // columns of this item
        float acc[RW_CH];
#pragma unroll
        for (int k = 0; k < RW_CH; ++k) acc[k] = 0.f;
        for (int t0 = 0; t0 < total; t0 += 16) {
            asm volatile("" ::: "memory");
            int cc = c, qq = q;
            launder(cc, qq);
            const int f = t0 + c;
            const bool live = MODE == 0 ? f < a.M : f < total;
            int kk = 0, m = f;
            if (MODE == 1) {
                kk = live ? f / a.M : 0;
                m = live ? f - kk * a.M : 0;
            }
            const int u = MODE == 1 ? cl[kk] : 0;
            const float imu = (MODE == 1 && live) ? a.im[((long)m * a.n + r) * a.d + u] : 0.f;
            const float* pp = a.pre + (((long)r * a.Mp + m) * 2) * H1P + 4 * q;
            f32x4 h1[2][H1T];
#pragma unroll
            for (int chn = 0; chn < 2; ++chn)
#pragma unroll
                for (int t = 0; t < H1T; ++t) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(pp + chn * H1P + 16 * t);
                    if (MODE == 1) v += *reinterpret_cast<const f32x4*>(a.W1T + (long)u * H1P + 16 * t + 4 * q) * imu;
                    h1[chn][t] = relu4(v);
                }
            f32x4 h2[2][H2T];
#pragma unroll
            for (int t = 0; t < H2T; ++t) {
                f32x4 o[2] = {zero4(), zero4()};
                tile_fwd_nb<H1T, 128, 2, NK1>(W2, t, h1, o, cc, qq);
                h2[0][t] = relu4(o[0]);
                h2[1][t] = relu4(o[1]);
            }
            f32x4 mu[2] = {zero4(), zero4()}, lv[2] = {zero4(), zero4()};
            tile_fwd_nb<H2T, 64, 2, NK2>(W3, 0, h2, mu, cc, qq);
            tile_fwd_nb<H2T, 64, 2, NK2>(W3, 1, h2, lv, cc, qq);
            float* st = a.stat + ((long)r * a.Mp + m) * STAT + 4 * q;
            if (MODE == 0) {
                *reinterpret_cast<f32x4*>(st) = mu[0];
                *reinterpret_cast<f32x4*>(st + 16) = lv[0];
                *reinterpret_cast<f32x4*>(st + 32) = mu[1];
                *reinterpret_cast<f32x4*>(st + 48) = lv[1];
            } else {
                float val = 0.f;
#pragma unroll
                for (int chn = 0; chn < 2; ++chn) {
                    const f32x4 ma = *reinterpret_cast<const f32x4*>(st + 32 * chn);
                    const f32x4 la = *reinterpret_cast<const f32x4*>(st + 32 * chn + 16);
                    float kl = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dm = mu[chn][j] - ma[j];
                        const float t = dm * dm * expf(-0.5f * la[j]) + expf(lv[chn][j] - la[j]) - 1.f - lv[chn][j] + la[j];
                        kl += (4 * q + j < a.L) ? t : 0.f;
                    }
                    kl = live ? 0.5f * kl : 0.f;
                    val += chn == 0 ? kl : -kl;
                }
#pragma unroll
                for (int k = 0; k < RW_CH; ++k) acc[k] += kk == k ? val : 0.f;
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < RW_CH; ++k) {
                const float sacc = wave_sum(acc[k]);
                if (lane == 0 && k < ncand) a.R[(long)r * (a.d - 1) + cl[k]] = sacc * invM;
            }
        }
    }
}

}  // namespace vpc

using namespace vpc;

// Scratch sizes (floats) the caller must provide for vpc_reward_matrix.
extern "C" int vpc_reward_scratch(int n, int d, int M, long* pre_floats, long* stat_floats, long* w1t_floats) {
    if (n <= 0 || d < 2 || d > MAX_D || M <= 0) return VPC_ERR_ARG;
    const long Mp = (M + 15) / 16 * 16;
    if (pre_floats) *pre_floats = (long)n * Mp * 2 * H1P;
    if (stat_floats) *stat_floats = (long)n * Mp * STAT;
    if (w1t_floats) *w1t_floats = (long)d * H1P + (long)n * d + 4;  // W1^T, the rows' candidate lists (ints), a work counter
    return VPC_OK;
}

extern "C" int vpc_reward_matrix(const float* x, const uint8_t* mask, const float* im, const float* W1, const float* b1,
                                 const float* enc_img, float* pre, float* stat, float* w1t, float* R, int n, int d, int L,
                                 int M, void* stream) {
    if (!x || !mask || !im || !W1 || !b1 || !enc_img || !pre || !stat || !w1t || !R) return VPC_ERR_ARG;
    if (n <= 0 || M <= 0) return VPC_ERR_ARG;
    if (d < 2 || d > MAX_D || L < 1 || L > MAX_L) return VPC_ERR_SHAPE;
    if (!aligned16(pre) || !aligned16(stat) || !aligned16(w1t)) return VPC_ERR_ARG;
    const int Mp = (M + 15) / 16 * 16;
    hipStream_t s = (hipStream_t)stream;
    int* cand = reinterpret_cast<int*>(w1t + (long)d * H1P);
    hipLaunchKernelGGL(reward_prep_kernel, dim3(n + (d + 7) / 8), dim3(128), 0, s, x, mask, im, W1, b1, pre, w1t, cand, R, n, d, M, Mp);
    RewardArgs a{enc_img, pre, w1t, im, mask, cand, cand + (long)n * d, stat, R, n, d, L, M, Mp};
    const EncImg imd(dt_for(d));
    const size_t lds = sizeof(float) * (imd.total - imd.oW2);
    const int cap = num_cus() * 3;  // (2 and 4 - 6 resident workgroups per CU measured slower)
    int gA = (n + RW_WAVES - 1) / RW_WAVES;
    if (gA > cap) gA = cap;
    long itemsB = (long)n * ((d - 1 + RW_CH - 1) / RW_CH);
    int gB = (int)((itemsB + RW_WAVES - 1) / RW_WAVES < cap ? (itemsB + RW_WAVES - 1) / RW_WAVES : cap);
    if (!lds_attr_done(reinterpret_cast<const void*>(reward_chain_kernel<0>), lds)) return VPC_ERR_HIP;
    if (!lds_attr_done(reinterpret_cast<const void*>(reward_chain_kernel<1>), lds)) return VPC_ERR_HIP;
    hipLaunchKernelGGL(reward_chain_kernel<0>, dim3(gA), dim3(RW_THREADS), lds, s, a);
    hipLaunchKernelGGL(reward_chain_kernel<1>, dim3(gB), dim3(RW_THREADS), lds, s, a);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}
