// Point-net ("PNP") encoder front-end of Reg_EDDI / vanilla_EDDI (reference src/models/VAE.py:719-733, 903-917):
//     h[b][j] = relu( W [x_bj, x_bj * E_j, t_j] + c )   (Linear(2+K -> K) on a [B*d, 2+K] tensor in the reference)
//     agg[b]  = sum_j mask[b][j] * h[b][j]              -> [B][K], the input of pnp_encoder2 (K -> 100 -> 50 -> 2L)
// The layer is linear in x_bj before the ReLU, so it folds per feature:  pre[b][j] = x_bj * A_j + C_j  with
//     A_j = w_x + W_E E_j,   C_j = w_t t_j + c          (W = [w_x | W_E | w_t], [K][2+K])
// and nothing of size B*d*(2+K) is ever materialised: HBM traffic is x, mask (B*d) in and agg (B*K) out.
// Backward: dA_j = sum_b g[b][j] x_bj, dC_j = sum_b g[b][j] with g = mask * 1[pre > 0] * dagg[b]; per-lane register
// accumulators over a grid-stride loop of rows, per-workgroup partials, fixed-order reduction (deterministic), then
// the chain rule back to (E, t, W, c) in one small workgroup.
#include "vpc_abi_internal.h"
#include "vpc_device.h"
#include "../../include/vpc.h"

namespace vpc {

constexpr int EDDI_MAX_K = 32;

// AC[0][k][j] = A_j[k], AC[1][k][j] = C_j[k]   (k-major: a lane that owns feature j reads consecutive banks)
__global__ void eddi_fold_kernel(const float* __restrict__ E, const float* __restrict__ tb, const float* __restrict__ Wp,
                                 const float* __restrict__ cp, float* __restrict__ AC, int d, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d * K) return;
    const int k = i / d, j = i % d;
    const float* w = Wp + (long)k * (2 + K);
    float a = w[0];
    for (int e = 0; e < K; ++e) a += w[1 + e] * E[(long)j * K + e];
    AC[(long)k * d + j] = a;
    AC[(long)(K + k) * d + j] = w[1 + K] * tb[j] + cp[k];
}

// rows r = pass * B + b: the passes of one step (mask, mask_p) are stacked, x is shared
template <int T>
__global__ __launch_bounds__(256) void eddi_front_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ m0,
                                                             const uint8_t* __restrict__ m1,
                                                             const float* __restrict__ AC, float* __restrict__ agg,
                                                             int B, int npass, int d, int K) {
    extern __shared__ float lds[];  // [2][K][d]
    for (int i = threadIdx.x; i < 2 * K * d; i += blockDim.x) lds[i] = AC[i];
    __syncthreads();
    const float* sA = lds;
    const float* sC = lds + K * d;
    const int lane = threadIdx.x & 63;
    const int gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = gwave; r < npass * B; r += nwaves) {
        const int b = r < B ? r : r - B;
        const uint8_t* __restrict__ m = r < B ? m0 : m1;
        float xv[T], mv[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int j = lane + 64 * t;
            const bool ok = j < d;
            xv[t] = ok ? x[(long)b * d + j] : 0.f;
            mv[t] = (ok && m[(long)b * d + j]) ? 1.f : 0.f;
        }
        float out = 0.f;
        for (int k = 0; k < K; ++k) {
            float v = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int j = lane + 64 * t;
                if (j < d) v += mv[t] * fmaxf(xv[t] * sA[k * d + j] + sC[k * d + j], 0.f);
            }
            const float s = wave_sum_dpp(v);
            if (lane == k) out = s;
        }
        if (lane < K) agg[(long)r * K + lane] = out;
    }
}

template <int T, int KP>
__global__ __launch_bounds__(256) void eddi_front_bwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ m0,
                                                             const uint8_t* __restrict__ m1,
                                                             const float* __restrict__ AC,
                                                             const float* __restrict__ dagg, float* __restrict__ part,
                                                             int B, int npass, int d, int K) {
    extern __shared__ float lds[];  // [2][K][d] images, then the cross-wave combine stage [2][K][d]
    for (int i = threadIdx.x; i < 2 * K * d; i += blockDim.x) lds[i] = AC[i];
    __syncthreads();
    const float* sA = lds;
    const float* sC = lds + K * d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const int R = npass * B;
    float dA[KP][T], dC[KP][T];
#pragma unroll
    for (int k = 0; k < KP; ++k)
#pragma unroll
        for (int t = 0; t < T; ++t) dA[k][t] = dC[k][t] = 0.f;
    // one row ahead: the loads of row r + nwaves are in flight while row r is accumulated (a wave has 2-3 dependent
    // global loads per row and nothing else to overlap them with)
    struct Row { float xv[T], mv[T], dg; };
    auto fetch = [&](int r, Row& o) {
        const int b = r < B ? r : r - B;
        const uint8_t* __restrict__ m = r < B ? m0 : m1;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int j = lane + 64 * t;
            const bool ok = j < d;
            o.xv[t] = ok ? x[(long)b * d + j] : 0.f;
            o.mv[t] = (ok && m[(long)b * d + j]) ? 1.f : 0.f;
        }
        o.dg = lane < K ? dagg[(long)r * K + lane] : 0.f;
    };
    Row cur, nxt;
    if (gwave < R) fetch(gwave, cur);
    for (int r = gwave; r < R; r += nwaves) {
        if (r + nwaves < R) fetch(r + nwaves, nxt);
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            if (k < K) {  // wave-uniform; no `break`: the loop must unroll fully to keep dA / dC in registers
                const float dg = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur.dg), k));
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int j = lane + 64 * t;
                    if (j < d) {
                        const float pre = cur.xv[t] * sA[k * d + j] + sC[k * d + j];
                        const float g = pre > 0.f ? cur.mv[t] * dg : 0.f;
                        dA[k][t] += g * cur.xv[t];
                        dC[k][t] += g;
                    }
                }
            }
        }
        cur = nxt;
    }
    // ---- combine the 4 waves through ONE stage of 2*K*d floats, wave after wave (fixed order => deterministic).  A
    // stage per wave (4x the LDS) limited the kernel to one workgroup per CU, i.e. one wave per SIMD.
    float* st = lds + 2 * K * d;
    for (int ww = 0; ww < 4; ++ww) {
        __syncthreads();
        if (wave == ww) {
#pragma unroll
            for (int k = 0; k < KP; ++k) {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int j = lane + 64 * t;
                    if (k < K && j < d) {
                        float* pa = st + k * d + j;
                        float* pc = st + (K + k) * d + j;
                        *pa = (ww == 0 ? 0.f : *pa) + dA[k][t];
                        *pc = (ww == 0 ? 0.f : *pc) + dC[k][t];
                    }
                }
            }
        }
    }
    __syncthreads();
    const int n = 2 * K * d;
    for (int i = threadIdx.x; i < n; i += blockDim.x) part[(long)blockIdx.x * n + i] = st[i];
}

// dAC[i] = sum over the workgroup partials (sum_partials_16x16: fixed order)
__global__ __launch_bounds__(256) void eddi_reduce_kernel(const float* __restrict__ part, int G, int n, float* __restrict__ dAC) {
    __shared__ float sh[16][16];
    const int i = blockIdx.x * 16 + (threadIdx.x & 15);
    const bool valid = i < n;
    const float s = sum_partials_16x16(part + (valid ? i : 0), n, G, valid, sh);
    if (threadIdx.x < 16 && valid) dAC[i] = s;
}

// chain rule from (dA, dC) [K][d] to the four parameter tensors.  dE / dt (d K + d outputs, K-term dot products): one thread per
// output (workgroups [0, gA)); dW / dc (K (2 + K) + K outputs, d-term dot products): one WAVE per output, lanes over j, DPP sum -
// as one thread per output these were 128 dependent-latency loads in a row on 13 workgroups: 36 us of a 0.3 ms step at d = 128
__global__ __launch_bounds__(256) void eddi_param_bwd_kernel(const float* __restrict__ dAC, const float* __restrict__ E,
                                                             const float* __restrict__ tb, const float* __restrict__ Wp,
                                                             float* __restrict__ gE, float* __restrict__ gtb,
                                                             float* __restrict__ gWp, float* __restrict__ gcp, int d,
                                                             int K, int accumulate, int gA) {
    const float* dA = dAC;
    const float* dC = dAC + (long)K * d;
    auto put = [&](float* p, float v) { *p = accumulate ? *p + v : v; };
    if ((int)blockIdx.x < gA) {
        int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i < d * K) {  // dE[j][e] = sum_k W_E[k][e] dA[k][j]
            const int j = i / K, e = i % K;
            float s = 0.f;
            for (int k = 0; k < K; ++k) s += Wp[(long)k * (2 + K) + 1 + e] * dA[(long)k * d + j];
            put(gE + i, s);
            return;
        }
        i -= d * K;
        if (i < d) {  // dt[j] = sum_k w_t[k] dC[k][j]
            float s = 0.f;
            for (int k = 0; k < K; ++k) s += Wp[(long)k * (2 + K) + 1 + K] * dC[(long)k * d + i];
            put(gtb + i, s);
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    int i = ((int)blockIdx.x - gA) * 4 + (threadIdx.x >> 6);  // one wave per output
    float s = 0.f;
    if (i < K * (2 + K)) {  // dW[k][0] = sum_j dA; dW[k][1+e] = sum_j dA E[j][e]; dW[k][1+K] = sum_j dC t[j]
        const int k = i / (2 + K), c = i % (2 + K);
        for (int j = lane; j < d; j += 64)
            s += c == 0 ? dA[(long)k * d + j] : c == 1 + K ? dC[(long)k * d + j] * tb[j] : dA[(long)k * d + j] * E[(long)j * K + (c - 1)];
        s = wave_sum_dpp(s);
        if (lane == 0) put(gWp + i, s);
        return;
    }
    i -= K * (2 + K);
    if (i < K) {  // dc[k] = sum_j dC[k][j]
        for (int j = lane; j < d; j += 64) s += dC[(long)i * d + j];
        s = wave_sum_dpp(s);
        if (lane == 0) put(gcp + i, s);
    }
}


static int eddi_blocks(long B) {
    long blocks = (B + 3) / 4;
    const long cap = 3L * num_cus();  // the backward kernel fits 3 workgroups per CU (registers)
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace vpc

using namespace vpc;

extern "C" {

int vpc_eddi_fold(const float* E, const float* tb, const float* Wp, const float* cp, float* AC, int d, int K,
                  void* stream) {
    if (!E || !tb || !Wp || !cp || !AC) return VPC_ERR_ARG;
    if (d <= 0 || d > 128 || K <= 0 || K > EDDI_MAX_K) return VPC_ERR_SHAPE;
    hipLaunchKernelGGL(eddi_fold_kernel, dim3((d * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, E, tb, Wp, cp, AC, d,
                       K);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

int vpc_eddi_front_fwd(const float* x, const uint8_t* mask, const uint8_t* mask2, const float* AC, float* agg, long B,
                       int d, int K, void* stream) {
    if (!x || !mask || !AC || !agg || B <= 0 || 2 * B > 0x7fffff00L) return VPC_ERR_ARG;
    if (d <= 0 || d > 128 || K <= 0 || K > EDDI_MAX_K) return VPC_ERR_SHAPE;
    const int npass = mask2 ? 2 : 1;
    const size_t lds = 2 * (size_t)K * d * sizeof(float);
    const int blocks = eddi_blocks(npass * B) * 2;
    hipStream_t st = (hipStream_t)stream;
    if (d <= 64) hipLaunchKernelGGL((eddi_front_fwd_kernel<1>), dim3(blocks), dim3(256), lds, st, x, mask, mask2, AC, agg, (int)B, npass, d, K);
    else hipLaunchKernelGGL((eddi_front_fwd_kernel<2>), dim3(blocks), dim3(256), lds, st, x, mask, mask2, AC, agg, (int)B, npass, d, K);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

long vpc_eddi_front_scratch(long B, int d, int K) {  // floats: per-workgroup partials + reduced (dA | dC); B = all rows
    if (B <= 0 || d <= 0 || K <= 0) return 0;
    return (long)(eddi_blocks(B) + 1) * 2 * K * d;
}

int vpc_eddi_front_bwd(const float* x, const uint8_t* mask, const uint8_t* mask2, const float* AC, const float* dagg,
                       const float* E, const float* tb, const float* Wp, float* scratch, long scratch_floats, float* gE,
                       float* gtb, float* gWp, float* gcp, int accumulate, long B, int d, int K, void* stream) {
    if (!x || !mask || !AC || !dagg || !E || !tb || !Wp || !scratch || !gE || !gtb || !gWp || !gcp || B <= 0 ||
        2 * B > 0x7fffff00L)
        return VPC_ERR_ARG;
    if (d <= 0 || d > 128 || K <= 0 || K > EDDI_MAX_K) return VPC_ERR_SHAPE;
    const int npass = mask2 ? 2 : 1;
    if (scratch_floats < vpc_eddi_front_scratch(npass * B, d, K)) return VPC_ERR_ARG;
    const int G = eddi_blocks(npass * B), n = 2 * K * d;
    const size_t lds = (size_t)4 * K * d * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    float* part = scratch;
    float* dAC = scratch + (long)G * n;
#define VPC_EDDI_BWD(T, KP)                                                                                          \
    do {                                                                                                             \
        if (!lds_attr_done(reinterpret_cast<const void*>(&eddi_front_bwd_kernel<T, KP>), lds)) return VPC_ERR_HIP;   \
        hipLaunchKernelGGL((eddi_front_bwd_kernel<T, KP>), dim3(G), dim3(256), lds, st, x, mask, mask2, AC, dagg, part, \
                           (int)B, npass, d, K);                                                                     \
    } while (0)
    // register accumulators are sized by K: exact variants for the reference's K = 10 / 20 (imputation_args.json)
    if (d <= 64) {
        if (K <= 10) VPC_EDDI_BWD(1, 10); else if (K <= 16) VPC_EDDI_BWD(1, 16);
        else if (K <= 20) VPC_EDDI_BWD(1, 20); else VPC_EDDI_BWD(1, 32);
    } else {
        if (K <= 10) VPC_EDDI_BWD(2, 10); else if (K <= 16) VPC_EDDI_BWD(2, 16);
        else if (K <= 20) VPC_EDDI_BWD(2, 20); else VPC_EDDI_BWD(2, 32);
    }
#undef VPC_EDDI_BWD
    if (hipGetLastError() != hipSuccess) return VPC_ERR_HIP;
    hipLaunchKernelGGL(eddi_reduce_kernel, dim3((n + 15) / 16), dim3(256), 0, st, part, G, n, dAC);
    const int gA = (d * K + d + 255) / 256, gB = (K * (2 + K) + K + 3) / 4;
    hipLaunchKernelGGL(eddi_param_bwd_kernel, dim3(gA + gB), dim3(256), 0, st, dAC, E, tb, Wp, gE, gtb, gWp, gcp, d, K,
                       accumulate, gA);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

}  // extern "C"
