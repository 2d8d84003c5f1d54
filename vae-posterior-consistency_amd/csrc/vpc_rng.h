// Philox counter RNG and the two per-step draws of the fused steps (device code shared by vpc_misc.hip - the draw kernels - and
// vpc_small.hip, whose small-batch step kernel draws its own tile's mask_p and eps: same counters, same values).
#pragma once
#include "vpc_device.h"

namespace vpc {

// Philox4x32-10 counter RNG
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox(uint64_t ctr, uint32_t stream, uint64_t seed) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = stream, c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(uint32_t u) { return ((u >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// mask_out = mask_in & (U < keep_prob)   (create_missing_uci * mask, utils.py:36-39 + train.py:54-55)
// One Philox call serves 8 mask bytes: each 32-bit word gives two 16-bit uniforms (keep probability resolved to
// 2^-16, far below the sampling noise of any batch).
constexpr int MASK_PER_CALL = 8;
// `elem_lo` = index of this call's first element inside the GLOBAL array (data parallel: shard_lo * d): the Philox
// counter of an element is its global index / 8 + offset, so the drawn mask does not depend on how rows are sharded
__device__ __forceinline__ void draw_mask_body(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long n,
                                               float keep_prob, uint64_t seed, uint64_t offset, long g, long elem_lo) {
    const long G = (elem_lo >> 3) + g;           // global group
    const long i0 = G * MASK_PER_CALL - elem_lo;  // local index of its first element (negative: group starts in the previous shard)
    if (i0 >= n) return;
    const U4 r = philox((uint64_t)G + offset, 0u, seed);
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
    const uint32_t thr = (uint32_t)(keep_prob * 65536.f + 0.5f);
    if (i0 >= 0 && i0 + MASK_PER_CALL - 1 < n && ((((uintptr_t)in + (uintptr_t)i0) | ((uintptr_t)out + (uintptr_t)i0)) & 7u) == 0) {
        uint32_t u[2] = {0x01010101u, 0x01010101u}, o[2] = {0u, 0u};
        if (in) {
            const uint2 v = *reinterpret_cast<const uint2*>(in + i0);
            u[0] = v.x; u[1] = v.y;
        }
#pragma unroll
        for (int j = 0; j < MASK_PER_CALL; ++j) {
            const uint32_t h = (rr[j >> 1] >> (16 * (j & 1))) & 0xffffu;
            if (((u[j >> 2] >> (8 * (j & 3))) & 0xffu) && h < thr) o[j >> 2] |= 1u << (8 * (j & 3));
        }
        *reinterpret_cast<uint2*>(out + i0) = make_uint2(o[0], o[1]);
    } else {
        for (int j = 0; j < MASK_PER_CALL && i0 + j < n; ++j) {
            if (i0 + j < 0) continue;
            const uint32_t h = (rr[j >> 1] >> (16 * (j & 1))) & 0xffffu;
            out[i0 + j] = ((in ? in[i0 + j] : 1) && h < thr) ? 1 : 0;
        }
    }
}

// Row-sharded normal draws: `out` is [planes][rows_local][pitch] (pitch % 4 == 0), the local shard of a global
// [planes][rows_global][pitch] array starting at row row_lo; the Philox counter of a 4-float group is its GLOBAL group
// index + offset, so every row gets the same eps whatever the sharding.  rows_local == 0: flat array, counter = g + offset.
struct EpsShard { long rows_local, rows_global, row_lo; int pitch; };
__device__ __forceinline__ uint64_t eps_counter(const EpsShard& sh, long g) {
    if (sh.rows_local == 0) return (uint64_t)g;
    const long gpr = sh.pitch >> 2, gpp = sh.rows_local * gpr;  // groups per row / per local plane
    const long k = g / gpp, rem = g - k * gpp;
    return (uint64_t)((k * sh.rows_global + sh.row_lo) * gpr + rem);
}

__device__ __forceinline__ void fill_normal_body(float* __restrict__ out, long n, uint64_t seed, uint64_t offset,
                                                 long g, const EpsShard& sh) {
    const long i0 = g * 4;
    if (i0 >= n) return;
    const U4 r = philox(eps_counter(sh, g) + offset, 1u, seed);
    // Box-Muller on the hardware transcendentals: v_log_f32 (log2), v_sqrt_f32 and v_sin / v_cos_f32, whose argument is in
    // revolutions - exactly the uniform.  (-2 ln u = -2 ln2 log2 u.)
    const float r0 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r.x)));
    const float r1 = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01(r.z)));
    const float t0 = u01(r.y), t1 = u01(r.w);
    const float s0 = __builtin_amdgcn_sinf(t0), c0 = __builtin_amdgcn_cosf(t0);
    const float s1 = __builtin_amdgcn_sinf(t1), c1 = __builtin_amdgcn_cosf(t1);
    const float v[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
    for (int j = 0; j < 4 && i0 + j < n; ++j) out[i0 + j] = v[j];
}

}  // namespace vpc
