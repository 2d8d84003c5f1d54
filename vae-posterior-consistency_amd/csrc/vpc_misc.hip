// Small kernels and host helpers around the encoder / decoder kernels (gfx950):
//   layout index builders (host), weight pack, gradient-partial reduction, flat Adam, loss finalisation,
//   the stand-alone fused loss kernel (K4: mask-apply / KL / consistency distance + backward seeds),
//   Philox4x32-10 Bernoulli keep-mask and N(0,1) generators.
#include "vpc_device.h"
#include "vpc_rng.h"
#include "vpc_bf16.h"
#include "vpc_adam.h"
#include "vpc_abi_internal.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>

namespace vpc {

// CU count of the CURRENT device (one process may drive several devices: cached per device id)
int num_cus() {
    constexpr int MAXDEV = 64;
    static std::atomic<int> cache[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < MAXDEV) {
        const int c = cache[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    if (dev >= 0 && dev < MAXDEV) cache[dev].store(v, std::memory_order_relaxed);
    return v;
}

TileShape tile_shape(long B, int npass, bool force_big) {
    const int ncu = num_cus();
    const long t64 = (B + 63) / 64, t128 = (B + 127) / 128;
    bool small = t64 * npass <= 2L * ncu;
    if (const char* e = getenv("VPC_TILE")) {
        const int v = atoi(e);
        if (v == 64) small = true;
        if (v == 128) small = false;
    }
    if (force_big) small = false;
    TileShape t;
    t.small = small ? 1 : 0;
    if (small) {
        t.ntiles = (int)t64;
        t.grid_x = (int)(t64 < ncu ? t64 : ncu);
        t.grid_y = npass;
    } else {
        t.ntiles = (int)t128;
        t.grid_x = (int)(t128 < ncu ? t128 : ncu);
        t.grid_y = 1;
    }
    t.nblocks = t.grid_x * t.grid_y;
    return t;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device function attribute: remembered per (device, kernel)
bool lds_attr_done(const void* kern, size_t lds) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> seen;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> g(mu);
    const auto key = std::make_pair(dev, kern);
    auto it = seen.find(key);
    if (it != seen.end() && it->second >= lds) return true;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    seen[key] = lds;
    return true;
}

// packed row of encoder layer 3: mean rows -> tile 0, logvar rows -> tile 1
static inline int row3(int o, int L) { return o < L ? o : 16 + (o - L); }

// ------------------------------------------------------------------------------------------------
// weight pack:  img[pack_idx[i]] = flat[i]
__global__ void pack_kernel(const float* __restrict__ flat, const int* __restrict__ idx, float* __restrict__ img,
                            int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) img[idx[i]] = flat[i];
}

// bf16 images (vpc_bf16.h): idx >= 0 is the u16 index of the hi half (lo 8 u16 further), idx < 0 encodes the dword
// index -(idx + 1) of a value that stays fp32 (the explicit layer-1 bias)
__global__ void pack_bf16_kernel(const float* __restrict__ flat, const int* __restrict__ idx, float* __restrict__ img,
                                 int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = flat[i];
    const int e = idx[i];
    if (e < 0) { img[-(e + 1)] = v; return; }
    unsigned short* u = reinterpret_cast<unsigned short*>(img);
    const uint32_t hi = pk_bf16(v, 0.f) & 0xffffu;
    u[e] = (unsigned short)hi;
    u[e + 8] = (unsigned short)(pk_bf16(v - __uint_as_float(hi << 16), 0.f) & 0xffffu);
}

// out[i] = scale * sum_b part[b * stride + idx[i]].  A 256-thread block handles 32 parameters x 8 block
// groups (thread (pi, bg) sums blocks bg, bg+8, ...), then the 8 group sums are added in a fixed order:
// bitwise reproducible, and 8x more loads in flight than one thread per parameter.
__device__ __forceinline__ void reduce_body(const float* __restrict__ part, int nblocks, long stride,
                                            const int* __restrict__ idx, float* __restrict__ out, int n, float scale,
                                            int blk, float (*sh)[32], const AdamFuse* adam = nullptr, int base = 0) {
    const int pi = threadIdx.x & 31, bg = threadIdx.x >> 5;
    const int i = blk * 32 + pi;
    float s0 = 0.f, s1 = 0.f;
    if (i < n) {
        const float* p = part + idx[i];
        int b = bg;
        for (; b + 8 < nblocks; b += 16) {
            s0 += p[(long)b * stride];
            s1 += p[(long)(b + 8) * stride];
        }
        if (b < nblocks) s0 += p[(long)b * stride];
    }
    sh[bg][pi] = s0 + s1;
    __syncthreads();
    if (bg == 0 && i < n) {
        float t = sh[0][pi];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sh[k][pi];
        out[i] = scale * t;
        if (adam && adam->param) adam_apply(*adam, base + i, scale * t);
    }
}
__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ part, int nblocks, long stride,
                                                     const int* __restrict__ idx, float* __restrict__ out, int n,
                                                     float scale) {
    __shared__ float sh[8][32];
    reduce_body(part, nblocks, stride, idx, out, n, scale, blockIdx.x, sh);
}

// torch.optim.Adam (no weight decay / amsgrad), src/experiment_main/train.py:21,116; optional re-pack
__global__ void adam_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m,
                            float* __restrict__ v, int n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, const int* __restrict__ pack_idx, float* __restrict__ img,
                            const long long* __restrict__ step_dev, const float* __restrict__ loss_in,
                            float* __restrict__ accum) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && accum) accum[0] += loss_in[0];  // epoch total of the (all-reduced) step loss, train.py:117
    if (i >= n) return;
    if (step_dev) {  // graph replay: the step count lives on the device (kernel arguments are frozen)
        const double t = (double)step_dev[0];
        bc1 = (float)(1.0 - pow((double)b1, t));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
    }
    adam_apply(AdamFuse{param, m, v, pack_idx, img, lr, b1, b2, eps, bc1, bc2_sqrt, 0}, i, grad[i]);
}

// loss_part[nblocks][8] doubles -> out[9] floats; out[0] = train loss (already / B), out[1..8] = raw sums;
// accum += loss.  256 threads: thread (term, g) sums blocks g, g+32, ...; fixed-order combine.
struct LossCoef {
    float cA0, cE0, cA1, bq, bp, cr, wml;
    double nll_const, inv_B;
};
__device__ __forceinline__ void finalize_body(const double* __restrict__ lp, int nblocks, const LossCoef& k,
                                              float* __restrict__ out, float* __restrict__ accum,
                                              double (*sh)[LOSS_TERMS], double* s) {
    const int term = threadIdx.x & 7, g = threadIdx.x >> 3;
    double t = 0.0;
    for (int b = g; b < nblocks; b += 32) t += lp[(long)b * LOSS_TERMS + term];
    sh[g][term] = t;
    __syncthreads();
    if (threadIdx.x < LOSS_TERMS) {
        double u = 0.0;
        for (int j = 0; j < 32; ++j) u += sh[j][threadIdx.x];
        s[threadIdx.x] = u;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double loss = k.cA0 * (s[0] + k.nll_const) + k.cE0 * (s[1] + k.nll_const) + k.cA1 * (s[2] + k.nll_const) +
                            k.bq * s[3] + k.bp * s[4] + k.cr * s[5] - k.wml * s[6];
        out[0] = (float)(loss * k.inv_B);
        for (int i = 0; i < LOSS_TERMS; ++i) out[1 + i] = (float)s[i];
        if (accum) accum[0] += (float)(loss * k.inv_B);
    }
}
__global__ __launch_bounds__(256) void loss_finalize_kernel(const double* __restrict__ lp, int nblocks, LossCoef k,
                                                            float* __restrict__ out, float* __restrict__ accum) {
    __shared__ double sh[32][LOSS_TERMS];
    __shared__ double s[LOSS_TERMS];
    finalize_body(lp, nblocks, k, out, accum, sh, s);
}

// one launch for the whole post-backward reduction of the fused step: encoder partial blocks -> grad[0, n_enc),
// decoder partial blocks -> grad[n_enc, n), loss partials -> out9 (+ accum); the last block does the loss
__global__ __launch_bounds__(256) void reduce_step_kernel(const float* __restrict__ partE, int nbE, long strideE,
                                                          const float* __restrict__ partD, int nbD, long strideD,
                                                          const int* __restrict__ idx, float* __restrict__ grad,
                                                          int n_enc, int n, const double* __restrict__ lp, int nbL,
                                                          LossCoef k, float* __restrict__ out9,
                                                          float* __restrict__ accum, long long* __restrict__ state,
                                                          long long rng_inc, AdamFuse adam) {
    __shared__ float shf[8][32];
    __shared__ double shd[32][LOSS_TERMS];
    __shared__ double s[LOSS_TERMS];
    const int gE = (n_enc + 31) / 32, gD = (n - n_enc + 31) / 32;
    const int b = blockIdx.x;
    if (b < gE) reduce_body(partE, nbE, strideE, idx, grad, n_enc, 1.f, b, shf, &adam, 0);
    else if (b < gE + gD)
        reduce_body(partD, nbD, strideD, idx + n_enc, grad + n_enc, n - n_enc, 1.f, b - gE, shf, &adam, n_enc);
    else {
        finalize_body(lp, nbL, k, out9, accum, shd, s);
        if (state && threadIdx.x == 0) {  // device-side step / RNG counters for graph replay
            state[0] += 1;
            state[1] += rng_inc;
        }
    }
}

// ---- the same reduction walking the partial blocks in LAYOUT order: each thread sums one 16-byte group of block
// positions (fully coalesced 16 B / lane instead of 4 B / lane gathers through grad_idx), then scatters the four sums
// through the inverse map position -> parameter (-1 = padding).  Same per-element summation order as reduce_body
// -1 = padding).  Summation order: blocks g, g + 32, ... per group, then groups 0..31 (fixed: reproducible).
__device__ __forceinline__ void reduce_body_v2(const float* __restrict__ part, int nblocks, long stride,
                                               const int* __restrict__ inv, float* __restrict__ out, int n4, int blk,
                                               f32x4 (*sh)[8], const AdamFuse* adam) {
    // 8 positions x 32 block groups per workgroup: with 256 blocks every thread has its 8 loads in flight at once (one
    // memory latency for the whole reduction), a wave touches full 128-byte lines, and the launch has ~6 workgroups/CU
    const int pi = threadIdx.x & 7, bg = threadIdx.x >> 3;
    const int p4 = blk * 8 + pi;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
    if (p4 < n4) {
        const f32x4* p = reinterpret_cast<const f32x4*>(part) + p4;
        const long st4 = stride >> 2;
#pragma unroll 8
        for (int b = bg; b < nblocks; b += 32) s0 += p[(long)b * st4];
    }
    sh[bg][pi] = s0;
    __syncthreads();
    if (threadIdx.x < 32) {  // one thread per element: fixed-order combine of the 32 groups, then the optimiser step
        const int pos = threadIdx.x >> 2, k = threadIdx.x & 3;
        const int e = 4 * (blk * 8 + pos) + k;
        if (blk * 8 + pos < n4) {
            float t = sh[0][pos][k];
#pragma unroll
            for (int g = 1; g < 32; ++g) t += sh[g][pos][k];
            const int id = inv[e];
            if (id >= 0) {
                out[id] = t;
                if (adam && adam->param) adam_apply(*adam, id, t);
            }
        }
    }
}
__global__ __launch_bounds__(256) void reduce_step_v2_kernel(const float* __restrict__ partE, int nbE, long strideE,
                                                             const float* __restrict__ partD, int nbD, long strideD,
                                                             const int* __restrict__ invE, const int* __restrict__ invD,
                                                             float* __restrict__ grad, const double* __restrict__ lp,
                                                             int nbL, LossCoef k, float* __restrict__ out9,
                                                             float* __restrict__ accum, long long* __restrict__ state,
                                                             long long rng_inc, AdamFuse adam) {
    __shared__ f32x4 shf[32][8];
    __shared__ double shd[32][LOSS_TERMS];
    __shared__ double s[LOSS_TERMS];
    const int nE4 = (int)(strideE >> 2), nD4 = (int)(strideD >> 2);
    const int gE = (nE4 + 7) / 8, gD = (nD4 + 7) / 8;
    const int b = blockIdx.x;
    if (b < gE) reduce_body_v2(partE, nbE, strideE, invE, grad, nE4, b, shf, &adam);
    else if (b < gE + gD) reduce_body_v2(partD, nbD, strideD, invD, grad, nD4, b - gE, shf, &adam);
    else {
        finalize_body(lp, nbL, k, out9, accum, shd, s);
        if (state && threadIdx.x == 0) {
            state[0] += 1;
            state[1] += rng_inc;
        }
    }
}
// inverse maps (block position -> flat parameter index, -1 for padding): built by the explicit entry point
// vpc_build_inverse_maps into a caller-owned buffer [enc_stride | dec_stride] ints (the library keeps no state)
__global__ void inv_fill_kernel(int* __restrict__ inv, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[i] = -1;
}
__global__ void inv_scatter_kernel(const int* __restrict__ idx, int* __restrict__ invE, int* __restrict__ invD, int n_enc,
                                   int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i < n_enc) invE[idx[i]] = i; else invD[idx[i]] = i;
}
// the layout-order reduction needs 16-byte vector access to the blocks
static bool inv_usable(const int* inv, const float* pe, long sE, const float* pd, long sD) {
    return inv && !(sE & 3) && !(sD & 3) && aligned16(pe) && aligned16(pd);
}

// ------------------------------------------------------------------------------------------------
// K4: stand-alone fused loss (API path: model.loss(...) on materialised tensors).
// One pass over x / xhat_q / xhat_p / masks / latent stats; per-block double partials; optional seeds.
struct LossArgs {
    const float* x;
    const float* xh[2];
    const uint8_t* mA[2];
    const uint8_t* mB[2];
    float cA[2], cE[2];
    const float* mean[2];
    const float* logvar[2];
    const float* eps_ml;
    float* dxh[2];
    float* dmean[2];
    float* dlogvar[2];
    double* loss_part;
    float bq, bp, cr, wml, inv_B, x_logvar;
    long n_el, n_lat;
    int npass, want_grad, vec;
};

__global__ __launch_bounds__(256) void loss_kernel(LossArgs a) {
    const float inv_s2 = expf(-a.x_logvar), half_lv = 0.5f * a.x_logvar;
    constexpr float HL2PI = 0.91893853320467274f;
    float S[LOSS_TERMS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
    // ---- B x d part: 16-byte vector path when everything is aligned and B*d % 4 == 0, else scalar
    const bool vec = a.vec != 0;
    const long n4 = vec ? a.n_el / 4 : 0;
    for (long i = tid; i < n4; i += nth) {
        const f32x4 xv = reinterpret_cast<const f32x4*>(a.x)[i];
        for (int p = 0; p < a.npass; ++p) {
            const f32x4 xh = reinterpret_cast<const f32x4*>(a.xh[p])[i];
            const uint32_t ua = reinterpret_cast<const uint32_t*>(a.mA[p])[i];
            const uint32_t ub = a.mB[p] ? reinterpret_cast<const uint32_t*>(a.mB[p])[i] : 0u;
            f32x4 g;
            float sa = 0.f, se = 0.f, sn = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float mA = ((ua >> (8 * j)) & 0xffu) ? 1.f : 0.f;
                const float mB = ((ub >> (8 * j)) & 0xffu) ? 1.f : 0.f;
                const float mE = a.mB[p] ? mA * (1.f - mB) : 0.f;
                const float diff = xh[j] - xv[j];
                const float t = half_lv + 0.5f * diff * diff * inv_s2;
                sa += mA * t;
                se += mE * t;
                sn += (1.f - mA) * t;
                g[j] = (a.cA[p] * mA + a.cE[p] * mE) * diff * inv_s2 * a.inv_B;
            }
            if (p == 0) { S[0] += sa; S[1] += se; S[7] += sn; } else { S[2] += sa; }
            if (a.want_grad && a.dxh[p]) reinterpret_cast<f32x4*>(a.dxh[p])[i] = g;
        }
    }
    if (!vec) {
        for (long i = tid; i < a.n_el; i += nth) {
            const float xv = a.x[i];
            for (int p = 0; p < a.npass; ++p) {
                const float mA = a.mA[p][i] ? 1.f : 0.f;
                const float mE = a.mB[p] ? mA * (a.mB[p][i] ? 0.f : 1.f) : 0.f;
                const float diff = a.xh[p][i] - xv;
                const float t = half_lv + 0.5f * diff * diff * inv_s2;
                if (p == 0) { S[0] += mA * t; S[1] += mE * t; S[7] += (1.f - mA) * t; } else { S[2] += mA * t; }
                if (a.want_grad && a.dxh[p])
                    a.dxh[p][i] = (a.cA[p] * mA + a.cE[p] * mE) * diff * inv_s2 * a.inv_B;
            }
        }
    }
    // ---- B x L part
    for (long i = tid; i < a.n_lat; i += nth) {
        const float mq = a.mean[0][i], lq = a.logvar[0][i];
        const float elq = expf(lq);
        S[3] += 0.5f * (elq + mq * mq - 1.f - lq);
        float dmq = a.bq * mq, dlq = a.bq * 0.5f * (elq - 1.f);
        if (a.npass == 2) {
            const float mp = a.mean[1][i], lp = a.logvar[1][i];
            const float elp = expf(lp), eip = expf(-lp), r = expf(lq - lp), diff = mq - mp;
            S[4] += 0.5f * (elp + mp * mp - 1.f - lp);
            S[5] += 0.5f * (r + diff * diff * eip - 1.f - (lq - lp));
            float dmp = a.bp * mp - a.cr * diff * eip;
            float dlp = a.bp * 0.5f * (elp - 1.f) + a.cr * 0.5f * (1.f - r - diff * diff * eip);
            dmq += a.cr * diff * eip;
            dlq += a.cr * 0.5f * (r - 1.f);
            if (a.wml != 0.f) {
                const float e3 = a.eps_ml[i], sq = expf(0.5f * lq);
                const float dlt = mq + e3 * sq - mp;
                S[6] += -HL2PI - 0.5f * lp - 0.5f * dlt * dlt * eip;
                dmq += a.wml * dlt * eip;
                dlq += a.wml * dlt * eip * e3 * 0.5f * sq;
                dmp -= a.wml * dlt * eip;
                dlp += a.wml * (0.5f - 0.5f * dlt * dlt * eip);
            }
            if (a.want_grad) { a.dmean[1][i] = dmp * a.inv_B; a.dlogvar[1][i] = dlp * a.inv_B; }
        }
        if (a.want_grad) { a.dmean[0][i] = dmq * a.inv_B; a.dlogvar[0][i] = dlq * a.inv_B; }
    }
    __shared__ float red[4][LOSS_TERMS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < LOSS_TERMS; ++i) {
        const float v = wave_sum(S[i]);
        if (lane == 0) red[w][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < LOSS_TERMS) {
        double t = 0.0;
        for (int k = 0; k < 4; ++k) t += (double)red[k][threadIdx.x];
        a.loss_part[(long)blockIdx.x * LOSS_TERMS + threadIdx.x] = t;
    }
}

// (Philox4x32-10, draw_mask_body, fill_normal_body: vpc_rng.h)
__global__ void draw_mask_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long n,
                                 float keep_prob, uint64_t seed, uint64_t offset, long elem_lo) {
    draw_mask_body(in, out, n, keep_prob, seed, offset, (long)blockIdx.x * blockDim.x + threadIdx.x, elem_lo);
}

__global__ void fill_normal_kernel(float* __restrict__ out, long n, uint64_t seed, uint64_t offset,
                                   const long long* __restrict__ state, EpsShard sh) {
    if (state) offset += (uint64_t)state[1];
    fill_normal_body(out, n, seed, offset, (long)blockIdx.x * blockDim.x + threadIdx.x, sh);
}
// both per-step draws of the fused step in one launch: blocks [0, gm) draw the keep-mask, the rest the normals
__global__ void draw_step_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ mout, long nm, float keep_prob,
                                 float* __restrict__ eout, long ne, uint64_t seed, uint64_t off_mask,
                                 uint64_t off_eps, unsigned gm, const long long* __restrict__ state, long elem_lo,
                                 EpsShard sh) {
    if (state) { off_mask += (uint64_t)state[1]; off_eps += (uint64_t)state[1]; }
    if (blockIdx.x < gm)
        draw_mask_body(in, mout, nm, keep_prob, seed, off_mask, (long)blockIdx.x * blockDim.x + threadIdx.x, elem_lo);
    else fill_normal_body(eout, ne, seed, off_eps, (long)(blockIdx.x - gm) * blockDim.x + threadIdx.x, sh);
}

// fused MNAR step: float mask_p draw + the stacked encoder input [x*mask ; x*mask_p] (blocks [0, gm)) and the
// step's normal draws (remaining blocks), one launch
__global__ void nm_prep_kernel(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ mp,
                               float* __restrict__ xin, long n, float keep_prob, float* __restrict__ eps, long n_eps,
                               uint64_t seed, uint64_t offset, uint64_t offset_eps, unsigned gm,
                               const long long* __restrict__ state, long elem_lo, EpsShard sh) {
    if (state) { offset += (uint64_t)state[1]; offset_eps += (uint64_t)state[1]; }
    if (blockIdx.x >= gm) {
        fill_normal_body(eps, n_eps, seed, offset_eps, (long)(blockIdx.x - gm) * blockDim.x + threadIdx.x, sh);
        return;
    }
    offset += (uint64_t)(elem_lo >> 2);  // counter of a mask group = its index in the GLOBAL [B_global][d] array
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long i0 = g * 4;
    if (i0 >= n) return;
    U4 r{0, 0, 0, 0};
    if (mp) r = philox((uint64_t)g + offset, 0u, seed);
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
    for (int j = 0; j < 4 && i0 + j < n; ++j) {
        const float xv = x[i0 + j], mv = m[i0 + j];
        xin[i0 + j] = xv * mv;
        if (mp) {
            const float pv = u01(rr[j]) < keep_prob ? mv : 0.f;
            mp[i0 + j] = pv;
            xin[n + i0 + j] = xv * pv;
        }
    }
}

}  // namespace vpc

using namespace vpc;

// ================================================================================================
// host-side layout queries / index builders (no GPU needed)
extern "C" int vpc_layout_sizes(int d, int L, int mask_augm, int* enc_img_floats, int* dec_img_floats,
                                int* n_enc_params, int* n_params, int* enc_part_floats, int* dec_part_floats,
                                int* loss_terms, int* tile_rows) {
    const int din = mask_augm ? 2 * d : d;
    if (d < 1 || din > MAX_D || L < 1 || L > MAX_L) return VPC_ERR_SHAPE;
    const int DT = dt_for(d);
    const ParamOffsets po(d, L, din);
    if (enc_img_floats) *enc_img_floats = EncImg(dt_for(din)).total;
    if (dec_img_floats) *dec_img_floats = DecImg(DT).total;
    if (n_enc_params) *n_enc_params = po.n_enc;
    if (n_params) *n_params = po.total;
    if (enc_part_floats) *enc_part_floats = ENC_PART;
    if (dec_part_floats) *dec_part_floats = DEC_PART;
    if (loss_terms) *loss_terms = LOSS_TERMS;
    if (tile_rows) *tile_rows = TILE_ROWS;
    return VPC_OK;
}

// pack_idx[i]: offset of flat parameter i inside the combined image buffer [enc image | dec image].
// grad_idx[i]: offset of d loss / d param_i inside the encoder (i < n_enc) or decoder partial block.
// img_template: combined image with zeros and the constant ones of the bias chain.
extern "C" int vpc_build_indices(int d, int L, int mask_augm, int* pack_idx, int* grad_idx, float* img_template) {
    const int din = mask_augm ? 2 * d : d;
    if (d < 1 || din > MAX_D || L < 1 || L > MAX_L) return VPC_ERR_SHAPE;
    if (!pack_idx || !grad_idx || !img_template) return VPC_ERR_ARG;
    const int DT = dt_for(d);
    const EncImg ei(dt_for(din));
    const DecImg di(DT);
    const ParamOffsets po(d, L, din);
    const int S1 = ei.S1, DB = ei.total;  // decoder image base inside the combined buffer
    std::memset(img_template, 0, sizeof(float) * (size_t)(ei.total + di.total));
    // hidden units sit at pos1(u) / pos2(u) of their padded width (vpc_layout.h): u -> rows of the producing layer,
    // columns of the consuming layer, positions of the h1 / h2 / g1 / g2 workspaces
    // ---- encoder layer 1: explicit bias; b1[pos1(100)] = 1 seeds the constant chain
    for (int o = 0; o < H1; ++o) {
        const int r = pos1(o);
        for (int i = 0; i < din; ++i) {
            pack_idx[po.w1 + o * din + i] = ei.oW1 + r * S1 + swz(i, r);
            grad_idx[po.w1 + o * din + i] = part_off(i >> 4, 4 * (r >> 4), r & 15, i & 15);
        }
        pack_idx[po.b1 + o] = ei.ob1 + r;
        grad_idx[po.b1 + o] = WAVES * GREGS * 64 + r;
    }
    img_template[ei.ob1 + pos1(H1)] = 1.f;
    // ---- encoder layer 2: bias in the column of h1's constant unit; a fake row forwards the constant to h2
    for (int o = 0; o < H2; ++o) {
        const int r = pos2(o);
        for (int i = 0; i <= H1; ++i) {
            const int flat = (i < H1) ? po.w2 + o * H1 + i : po.b2 + o;
            const int cI = pos1(i);
            pack_idx[flat] = ei.oW2 + r * 128 + swz(cI, r);
            grad_idx[flat] = part_off(cI >> 4, 28 + 4 * (r >> 4), r & 15, cI & 15);
        }
    }
    img_template[ei.oW2 + pos2(H2) * 128 + swz(pos1(H1), pos2(H2))] = 1.f;
    // ---- encoder layer 3: mean rows -> tile 0, logvar rows -> tile 1; bias in the column of h2's constant unit
    for (int o = 0; o < 2 * L; ++o) {
        const int pr = row3(o, L);
        for (int i = 0; i <= H2; ++i) {
            const int flat = (i < H2) ? po.w3 + o * H2 + i : po.b3 + o;
            const int cI = pos2(i);
            pack_idx[flat] = ei.oW3 + pr * 64 + swz(cI, pr);
            grad_idx[flat] = part_off((pr >> 4) * 4 + (cI >> 4), 44, pr & 15, cI & 15);
        }
    }
    // ---- decoder layer 4: bias in column L (z[L] == 1); a fake row forwards the constant to g1
    for (int o = 0; o < H2; ++o) {
        const int r = pos2(o);
        for (int i = 0; i <= L; ++i) {
            const int flat = (i < L) ? po.w4 + o * L + i : po.b4 + o;
            pack_idx[flat] = DB + di.oW4 + r * S4 + swz(i, r, S4);
            grad_idx[flat] = part_off(r >> 4, 88, r & 15, i & 15, DEC_GREGS);
        }
    }
    img_template[DB + di.oW4 + pos2(H2) * S4 + swz(L, pos2(H2), S4)] = 1.f;
    // ---- decoder layer 5: bias in the column of g1's constant unit; a fake row forwards the constant to g2
    for (int o = 0; o < H1; ++o) {
        const int r = pos1(o);
        for (int i = 0; i <= H2; ++i) {
            const int flat = (i < H2) ? po.w5 + o * H2 + i : po.b5 + o;
            const int cI = pos2(i);
            pack_idx[flat] = DB + di.oW5 + r * 64 + swz(cI, r);
            grad_idx[flat] = part_off((r >> 4) & 3, 56 + 16 * (r >> 6) + 4 * (cI >> 4), r & 15, cI & 15, DEC_GREGS);
        }
    }
    img_template[DB + di.oW5 + pos1(H1) * 64 + swz(pos2(H2), pos1(H1))] = 1.f;
    // ---- decoder layer 6: bias in the column of g2's constant unit
    for (int o = 0; o < d; ++o) {
        for (int i = 0; i <= H1; ++i) {
            const int flat = (i < H1) ? po.w6 + o * H1 + i : po.b6 + o;
            const int cI = pos1(i);
            pack_idx[flat] = DB + di.oW6 + o * 128 + swz(cI, o);
            grad_idx[flat] = part_off((o >> 4) & 3, 28 * (o >> 6) + 4 * (cI >> 4), o & 15, cI & 15, DEC_GREGS);
        }
    }
    return VPC_OK;
}

// ---- bf16 images (PREC_BF16X3 / PREC_BF16): same row geometry as the fp32 images (W4 rows 32 dwords instead of 16),
// columns in the k-slot order of vpc_bf16.h.  pack_idx_bf[i] as pack_bf16_kernel reads it; img_template_bf = zeros plus
// the constant ones of the bias chain (bf16 1.0 = 0x3F80, the layer-1 seed stays fp32).
extern "C" int vpc_layout_sizes_bf16(int d, int L, int mask_augm, int* enc_img_floats, int* dec_img_floats) {
    const int din = mask_augm ? 2 * d : d;
    if (d < 1 || din > MAX_D || L < 1 || L > MAX_L) return VPC_ERR_SHAPE;
    if (enc_img_floats) *enc_img_floats = EncImg(dt_for(din)).total;
    if (dec_img_floats) *dec_img_floats = DecImg(dt_for(d), 32).total;
    return VPC_OK;
}
extern "C" int vpc_build_indices_bf16(int d, int L, int mask_augm, int* pack_idx_bf, float* img_template_bf) {
    const int din = mask_augm ? 2 * d : d;
    if (d < 1 || din > MAX_D || L < 1 || L > MAX_L) return VPC_ERR_SHAPE;
    if (!pack_idx_bf || !img_template_bf) return VPC_ERR_ARG;
    const EncImg ei(dt_for(din));
    const DecImg di(dt_for(d), 32);
    const ParamOffsets po(d, L, din);
    const int S1 = ei.S1, DB = ei.total;
    std::memset(img_template_bf, 0, sizeof(float) * (size_t)(ei.total + di.total));
    unsigned short* u = reinterpret_cast<unsigned short*>(img_template_bf);
    const unsigned short ONE = 0x3F80;
    auto at = [](int base_dw, int row, int f, int kp) { return 2 * base_dw + bf_elem(row, f, kp); };
    for (int o = 0; o < H1; ++o) {
        const int r = pos1(o);
        for (int i = 0; i < din; ++i) pack_idx_bf[po.w1 + o * din + i] = at(ei.oW1, r, i, S1);
        pack_idx_bf[po.b1 + o] = -(ei.ob1 + r + 1);
    }
    img_template_bf[ei.ob1 + pos1(H1)] = 1.f;
    for (int o = 0; o < H2; ++o) {
        const int r = pos2(o);
        for (int i = 0; i <= H1; ++i) pack_idx_bf[(i < H1) ? po.w2 + o * H1 + i : po.b2 + o] = at(ei.oW2, r, pos1(i), 128);
    }
    u[at(ei.oW2, pos2(H2), pos1(H1), 128)] = ONE;
    for (int o = 0; o < 2 * L; ++o) {
        const int pr = row3(o, L);
        for (int i = 0; i <= H2; ++i) pack_idx_bf[(i < H2) ? po.w3 + o * H2 + i : po.b3 + o] = at(ei.oW3, pr, pos2(i), 64);
    }
    for (int o = 0; o < H2; ++o) {
        const int r = pos2(o);
        for (int i = 0; i <= L; ++i) pack_idx_bf[(i < L) ? po.w4 + o * L + i : po.b4 + o] = at(DB + di.oW4, r, i, 32);
    }
    u[at(DB + di.oW4, pos2(H2), L, 32)] = ONE;
    for (int o = 0; o < H1; ++o) {
        const int r = pos1(o);
        for (int i = 0; i <= H2; ++i) pack_idx_bf[(i < H2) ? po.w5 + o * H2 + i : po.b5 + o] = at(DB + di.oW5, r, pos2(i), 64);
    }
    u[at(DB + di.oW5, pos1(H1), pos2(H2), 64)] = ONE;
    for (int o = 0; o < d; ++o)
        for (int i = 0; i <= H1; ++i) pack_idx_bf[(i < H1) ? po.w6 + o * H1 + i : po.b6 + o] = at(DB + di.oW6, o, pos1(i), 128);
    return VPC_OK;
}

extern "C" int vpc_num_cus(void) { return num_cus(); }
extern "C" int vpc_max_partial_blocks(void) { return 2 * num_cus(); }

// ================================================================================================
extern "C" int vpc_pack_weights(const float* flat_params, const int* pack_idx, float* img, int n, void* stream) {
    if (!flat_params || !pack_idx || !img || n <= 0) return VPC_ERR_ARG;
    hipLaunchKernelGGL(pack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, flat_params, pack_idx,
                       img, n);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_pack_weights_bf16(const float* flat_params, const int* pack_idx_bf, float* img_bf, int n, void* stream) {
    if (!flat_params || !pack_idx_bf || !img_bf || n <= 0) return VPC_ERR_ARG;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, flat_params,
                       pack_idx_bf, img_bf, n);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_reduce_partials(const float* partials, int nblocks, long block_stride, const int* grad_idx,
                                   float* grad_out, int n, float scale, void* stream) {
    if (!partials || !grad_idx || !grad_out || n <= 0 || nblocks <= 0) return VPC_ERR_ARG;
    hipLaunchKernelGGL(reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, (hipStream_t)stream, partials, nblocks,
                       block_stride, grad_idx, grad_out, n, scale);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, float lr,
                             float beta1, float beta2, float eps, long step, const long long* step_dev,
                             const int* pack_idx, float* img, const float* loss_in, float* accum, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || n <= 0 || (step < 1 && !step_dev)) return VPC_ERR_ARG;
    if (accum && !loss_in) return VPC_ERR_ARG;
    if (step < 1) step = 1;
    if ((pack_idx == nullptr) != (img == nullptr)) return VPC_ERR_ARG;
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, params, grads,
                       exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, (float)bc1, (float)std::sqrt(bc2), pack_idx,
                       img, step_dev, loss_in, accum);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_loss_finalize(const double* loss_partials, int nblocks, float cA0, float cE0, float cA1, float bq,
                                 float bp, float cr, float wml, long B_local, long B_global, int d, float* out9,
                                 float* accum,
                                 void* stream) {
    if (!loss_partials || !out9 || nblocks <= 0 || B_local <= 0 || B_global <= 0) return VPC_ERR_ARG;
    const LossCoef k{cA0, cE0, cA1, bq, bp, cr, wml, 0.91893853320467274178 * (double)B_local * (double)d,
                     1.0 / (double)B_global};
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_partials, nblocks, k,
                       out9, accum);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_build_inverse_maps(const int* grad_idx, int n_enc, int n, long enc_stride, long dec_stride,
                                      int* inv_out, void* stream) {
    if (!grad_idx || !inv_out || n_enc <= 0 || n <= n_enc || enc_stride <= 0 || dec_stride <= 0) return VPC_ERR_ARG;
    const long tot = enc_stride + dec_stride;
    hipLaunchKernelGGL(inv_fill_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, inv_out, tot);
    hipLaunchKernelGGL(inv_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, grad_idx, inv_out,
                       inv_out + enc_stride, n_enc, n);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_reduce_step(const float* enc_partials, int enc_blocks, long enc_stride, const float* dec_partials,
                               int dec_blocks, long dec_stride, const int* grad_idx, const int* inv_maps,
                               float* grad_out, int n_enc, int n,
                               const double* loss_partials, int loss_blocks, float cA0, float cE0, float cA1, float bq,
                               float bp, float cr, float wml, long B_local, long B_global, int d, float* out9,
                               float* accum, long long* state, long long rng_inc, void* stream) {
    if (!enc_partials || !dec_partials || !grad_idx || !grad_out || !loss_partials || !out9) return VPC_ERR_ARG;
    if (enc_blocks <= 0 || dec_blocks <= 0 || loss_blocks <= 0 || n_enc <= 0 || n <= n_enc || B_local <= 0 ||
        B_global <= 0)
        return VPC_ERR_ARG;
    const LossCoef k{cA0, cE0, cA1, bq, bp, cr, wml, 0.91893853320467274178 * (double)B_local * (double)d,
                     1.0 / (double)B_global};
    if (inv_usable(inv_maps, enc_partials, enc_stride, dec_partials, dec_stride)) {
        const int *invE = inv_maps, *invD = inv_maps + enc_stride;
        const int grid2 = (int)((enc_stride / 4 + 7) / 8 + (dec_stride / 4 + 7) / 8 + 1);
        hipLaunchKernelGGL(reduce_step_v2_kernel, dim3(grid2), dim3(256), 0, (hipStream_t)stream, enc_partials, enc_blocks,
                           enc_stride, dec_partials, dec_blocks, dec_stride, invE, invD, grad_out, loss_partials,
                           loss_blocks, k, out9, accum, state, rng_inc, AdamFuse{});
        return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
    }
    const int grid = (n_enc + 31) / 32 + (n - n_enc + 31) / 32 + 1;
    hipLaunchKernelGGL(reduce_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, enc_partials, enc_blocks,
                       enc_stride, dec_partials, dec_blocks, dec_stride, grad_idx, grad_out, n_enc, n, loss_partials,
                       loss_blocks, k, out9, accum, state, rng_inc, AdamFuse{});
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_reduce_step_adam(const float* enc_partials, int enc_blocks, long enc_stride,
                                    const float* dec_partials, int dec_blocks, long dec_stride, const int* grad_idx,
                                    const int* inv_maps, float* grad_out, int n_enc, int n,
                                    const double* loss_partials, int loss_blocks,
                                    float cA0, float cE0, float cA1, float bq, float bp, float cr, float wml,
                                    long B_local, long B_global, int d, float* out9, float* accum, float* params,
                                    float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                                    long step, const int* pack_idx, float* img, void* stream) {
    if (!enc_partials || !dec_partials || !grad_idx || !grad_out || !loss_partials || !out9) return VPC_ERR_ARG;
    if (enc_blocks <= 0 || dec_blocks <= 0 || loss_blocks <= 0 || n_enc <= 0 || n <= n_enc) return VPC_ERR_ARG;
    if (!params || !exp_avg || !exp_avg_sq || step < 1 || (pack_idx == nullptr) != (img == nullptr)) return VPC_ERR_ARG;
    const LossCoef k{cA0, cE0, cA1, bq, bp, cr, wml, 0.91893853320467274178 * (double)B_local * (double)d,
                     1.0 / (double)B_global};
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    const AdamFuse A{params, exp_avg, exp_avg_sq, pack_idx, img, lr, beta1, beta2, eps, (float)bc1, (float)std::sqrt(bc2), 0};
    if (inv_usable(inv_maps, enc_partials, enc_stride, dec_partials, dec_stride)) {
        const int *invE = inv_maps, *invD = inv_maps + enc_stride;
        const int grid2 = (int)((enc_stride / 4 + 7) / 8 + (dec_stride / 4 + 7) / 8 + 1);
        hipLaunchKernelGGL(reduce_step_v2_kernel, dim3(grid2), dim3(256), 0, (hipStream_t)stream, enc_partials, enc_blocks,
                           enc_stride, dec_partials, dec_blocks, dec_stride, invE, invD, grad_out, loss_partials,
                           loss_blocks, k, out9, accum, (long long*)nullptr, 0LL, A);
        return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
    }
    const int grid = (n_enc + 31) / 32 + (n - n_enc + 31) / 32 + 1;
    hipLaunchKernelGGL(reduce_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, enc_partials, enc_blocks,
                       enc_stride, dec_partials, dec_blocks, dec_stride, grad_idx, grad_out, n_enc, n, loss_partials,
                       loss_blocks, k, out9, accum, (long long*)nullptr, 0LL, A);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

// The same with the compact bf16 image of the whole-step kernel as the re-pack target (pack_idx_c / img_c of
// vpc_step_build_indices_bf16): a bf16 step then needs no separate pack launch.
extern "C" int vpc_reduce_step_adam_bf16c(const float* enc_partials, int enc_blocks, long enc_stride,
                                    const float* dec_partials, int dec_blocks, long dec_stride, const int* grad_idx,
                                    const int* inv_maps, float* grad_out, int n_enc, int n,
                                    const double* loss_partials, int loss_blocks,
                                    float cA0, float cE0, float cA1, float bq, float bp, float cr, float wml,
                                    long B_local, long B_global, int d, float* out9, float* accum, float* params,
                                          float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                                    long step, const int* pack_idx, float* img, void* stream) {
    if (!enc_partials || !dec_partials || !grad_idx || !grad_out || !loss_partials || !out9) return VPC_ERR_ARG;
    if (enc_blocks <= 0 || dec_blocks <= 0 || loss_blocks <= 0 || n_enc <= 0 || n <= n_enc) return VPC_ERR_ARG;
    if (!params || !exp_avg || !exp_avg_sq || step < 1 || (pack_idx == nullptr) != (img == nullptr)) return VPC_ERR_ARG;
    const LossCoef k{cA0, cE0, cA1, bq, bp, cr, wml, 0.91893853320467274178 * (double)B_local * (double)d,
                     1.0 / (double)B_global};
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    const AdamFuse A{params, exp_avg, exp_avg_sq, pack_idx, img, lr, beta1, beta2, eps, (float)bc1, (float)std::sqrt(bc2), 1};
    if (inv_usable(inv_maps, enc_partials, enc_stride, dec_partials, dec_stride)) {
        const int *invE = inv_maps, *invD = inv_maps + enc_stride;
        const int grid2 = (int)((enc_stride / 4 + 7) / 8 + (dec_stride / 4 + 7) / 8 + 1);
        hipLaunchKernelGGL(reduce_step_v2_kernel, dim3(grid2), dim3(256), 0, (hipStream_t)stream, enc_partials, enc_blocks,
                           enc_stride, dec_partials, dec_blocks, dec_stride, invE, invD, grad_out, loss_partials,
                           loss_blocks, k, out9, accum, (long long*)nullptr, 0LL, A);
        return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
    }
    const int grid = (n_enc + 31) / 32 + (n - n_enc + 31) / 32 + 1;
    hipLaunchKernelGGL(reduce_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, enc_partials, enc_blocks,
                       enc_stride, dec_partials, dec_blocks, dec_stride, grad_idx, grad_out, n_enc, n, loss_partials,
                       loss_blocks, k, out9, accum, (long long*)nullptr, 0LL, A);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_loss_fwd_bwd(const float* x, int npass, const float* const* xhat, const uint8_t* const* maskA,
                                const uint8_t* const* maskB, const float* cA, const float* cE,
                                const float* const* mean, const float* const* logvar, const float* eps_ml, float bq,
                                float bp, float cr, float wml, float inv_B, float x_logvar, float* const* dxhat,
                                float* const* dmean, float* const* dlogvar, double* loss_partials, int max_blocks,
                                int* nblocks_out, long B, int d, int L, void* stream) {
    if (!x || !xhat || !maskA || !cA || !cE || !mean || !logvar || !loss_partials || !nblocks_out) return VPC_ERR_ARG;
    if (npass < 1 || npass > 2 || B <= 0 || d < 1 || L < 1 || max_blocks < 1) return VPC_ERR_ARG;
    if (wml != 0.f && !eps_ml) return VPC_ERR_ARG;
    LossArgs a{};
    a.x = x; a.eps_ml = eps_ml; a.loss_part = loss_partials;
    a.bq = bq; a.bp = bp; a.cr = cr; a.wml = wml; a.inv_B = inv_B; a.x_logvar = x_logvar;
    a.n_el = B * (long)d; a.n_lat = B * (long)L; a.npass = npass;
    a.want_grad = (dxhat && dmean && dlogvar) ? 1 : 0;
    bool al = aligned16(x);
    for (int p = 0; p < npass; ++p) {
        if (!xhat[p] || !maskA[p] || !mean[p] || !logvar[p]) return VPC_ERR_ARG;
        a.xh[p] = xhat[p]; a.mA[p] = maskA[p]; a.mB[p] = maskB ? maskB[p] : nullptr; a.cA[p] = cA[p]; a.cE[p] = cE[p];
        a.mean[p] = mean[p]; a.logvar[p] = logvar[p];
        if (a.want_grad) {
            if (!dmean[p] || !dlogvar[p]) return VPC_ERR_ARG;
            a.dxh[p] = dxhat[p]; a.dmean[p] = dmean[p]; a.dlogvar[p] = dlogvar[p];
            al = al && (!dxhat[p] || aligned16(dxhat[p]));
        }
        al = al && aligned16(xhat[p]) && ((uintptr_t)a.mA[p] % 4 == 0) && (!a.mB[p] || (uintptr_t)a.mB[p] % 4 == 0);
    }
    a.vec = (al && a.n_el % 4 == 0) ? 1 : 0;
    long work = (B * (long)d + 3) / 4;
    int grid = (int)((work + 255) / 256);
    const int cap = num_cus() * 8;
    if (grid > cap) grid = cap;
    if (grid > max_blocks) grid = max_blocks;
    if (grid < 1) grid = 1;
    *nblocks_out = grid;
    hipLaunchKernelGGL(loss_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_draw_mask(const uint8_t* mask_in, uint8_t* mask_out, long n, float keep_prob,
                             unsigned long long seed, unsigned long long offset, long elem_lo, void* stream) {
    if (!mask_out || n <= 0 || elem_lo < 0) return VPC_ERR_ARG;
    const long groups = ((elem_lo & 7) + n + MASK_PER_CALL - 1) / MASK_PER_CALL;
    hipLaunchKernelGGL(draw_mask_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       mask_in, mask_out, n, keep_prob, (uint64_t)seed, (uint64_t)offset, elem_lo);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_draw_step(const uint8_t* mask_in, uint8_t* mask_out, long n_mask, float keep_prob, float* eps_out,
                             long n_eps, unsigned long long seed, unsigned long long offset_mask,
                             unsigned long long offset_eps, const long long* state, long mask_elem_lo,
                             long eps_rows_local, long eps_rows_global, long eps_row_lo, int eps_pitch, void* stream) {
    if (!mask_out || !eps_out || n_mask <= 0 || n_eps <= 0 || mask_elem_lo < 0) return VPC_ERR_ARG;
    if (eps_rows_local < 0 || (eps_rows_local > 0 && (eps_pitch <= 0 || (eps_pitch & 3) || eps_row_lo < 0 ||
                                                      eps_row_lo + eps_rows_local > eps_rows_global)))
        return VPC_ERR_ARG;
    const unsigned gm = (unsigned)((((mask_elem_lo & 7) + n_mask + MASK_PER_CALL - 1) / MASK_PER_CALL + 255) / 256),
                   ge = (unsigned)(((n_eps + 3) / 4 + 255) / 256);
    hipLaunchKernelGGL(draw_step_kernel, dim3(gm + ge), dim3(256), 0, (hipStream_t)stream, mask_in, mask_out, n_mask,
                       keep_prob, eps_out, n_eps, (uint64_t)seed, (uint64_t)offset_mask, (uint64_t)offset_eps, gm, state,
                       mask_elem_lo, EpsShard{eps_rows_local, eps_rows_global, eps_row_lo, eps_pitch});
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_fill_normal(float* out, long n, unsigned long long seed, unsigned long long offset,
                               const long long* state, long rows_local, long rows_global, long row_lo, int pitch,
                               void* stream) {
    if (!out || n <= 0) return VPC_ERR_ARG;
    if (rows_local < 0 || (rows_local > 0 && (pitch <= 0 || (pitch & 3) || row_lo < 0 || row_lo + rows_local > rows_global)))
        return VPC_ERR_ARG;
    const long groups = (n + 3) / 4;
    hipLaunchKernelGGL(fill_normal_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       out, n, (uint64_t)seed, (uint64_t)offset, state, EpsShard{rows_local, rows_global, row_lo, pitch});
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_nm_prep(const float* x, const float* mask, float* mask_p_out, float* xin, long B, int d,
                           float keep_prob, float* eps_out, long n_eps, unsigned long long seed,
                           unsigned long long offset, unsigned long long offset_eps, const long long* state,
                           long elem_lo, long eps_rows_local, long eps_rows_global, long eps_row_lo, int eps_pitch,
                           void* stream) {
    if (!x || !mask || !xin || B <= 0 || d <= 0 || n_eps < 0 || (n_eps > 0 && !eps_out)) return VPC_ERR_ARG;
    if (elem_lo < 0 || (elem_lo & 3)) return VPC_ERR_ARG;
    if (eps_rows_local < 0 || (eps_rows_local > 0 && (eps_pitch <= 0 || (eps_pitch & 3) || eps_row_lo < 0 ||
                                                      eps_row_lo + eps_rows_local > eps_rows_global)))
        return VPC_ERR_ARG;
    const long n = B * d, groups = (n + 3) / 4, ge = (n_eps + 3) / 4;
    const unsigned gm = (unsigned)((groups + 255) / 256), gn = (unsigned)((ge + 255) / 256);
    hipLaunchKernelGGL(nm_prep_kernel, dim3(gm + gn), dim3(256), 0, (hipStream_t)stream, x, mask, mask_p_out, xin, n,
                       keep_prob, eps_out, n_eps, (uint64_t)seed, (uint64_t)offset, (uint64_t)offset_eps, gm, state,
                       elem_lo, EpsShard{eps_rows_local, eps_rows_global, eps_row_lo, eps_pitch});
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}
