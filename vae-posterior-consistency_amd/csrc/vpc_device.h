// Device-side building blocks: register-chained f32 MFMA MLP tiles (gfx950 / CDNA4 only).
//
// Orientation.  A wave owns 16 batch rows.  Activations are kept TRANSPOSED, feature-major:
// an activation tile is 16 features x 16 batch rows held as one f32x4 per lane in the MFMA C/D layout
//     lane l:  c = l & 15 (batch row inside the wave's 16),  q = l >> 4,   v[j] = act[feature 16t + 4q + j][row c]
// which is at the same time the B-operand layout of v_mfma_f32_16x16x4_f32 for k-step j (B[k=q][n=c]).  So
//     Y^T[out][rows] = W[out][in] * X^T[in][rows]
// chains layer to layer entirely in registers: the weights are the A operand (from the swizzled LDS image),
// the previous layer's accumulators are the B operand, no LDS round trip for activations.  dgrad
// (dX^T = W^T dY^T) uses the same image through the transposed read.  Only wgrad, which contracts over the
// batch index that sits on l & 15, needs a transpose; it goes through a small LDS staging buffer shared by
// the 8 waves of the workgroup, each wave owning a slice of the dW accumulators for the whole kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "vpc_layout.h"

namespace vpc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define VPC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
// sigmoid through v_exp_f32 / v_rcp_f32 (<= 2 ulp each; |rel err| < 1e-6 for |x| < 16).  The IEEE expf + divide
// expansion costs ~45 VALU instructions per element, this one ~6.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
}
// gate(dy, act) = dy where act > 0 else 0  (ReLU backward)
__device__ __forceinline__ f32x4 gate4(f32x4 dy, f32x4 act) {
    return f32x4{act[0] > 0.f ? dy[0] : 0.f, act[1] > 0.f ? dy[1] : 0.f, act[2] > 0.f ? dy[2] : 0.f,
                 act[3] > 0.f ? dy[3] : 0.f};
}

// ReLU gate as a bit mask (bit 4t+j = act tile t element j > 0): lets the activation registers die early
template <int T>
__device__ __forceinline__ uint32_t relu_bits(const f32x4 (&act)[T]) {
    uint32_t b = 0;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) b |= (act[t][j] > 0.f ? 1u : 0u) << (4 * t + j);
    return b;
}
// (v_bfe_i32 of one bit gives 0 / -1: two VALU per element instead of shift-and, compare, select)
__device__ __forceinline__ f32x4 gate_bits(f32x4 dy, uint32_t bits, int t) {
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        r[j] = __uint_as_float(__float_as_uint(dy[j]) & (uint32_t)__builtin_amdgcn_sbfe((int)bits, 4 * t + j, 1));
    return r;
}

// ---- NB batch tiles per wave (decoder kernel, one wave per SIMD): one A fragment feeds NB independent
// accumulator chains, which hides the 40-cycle dependent-MFMA latency without a second wave on the SIMD.
template <int KT, int S, int NB, int NK = 4 * KT>  // NK: k-steps to run (the rest multiply padding zeros)
__device__ __forceinline__ void tile_fwd_nb(const float* W, int mt, const f32x4 (&in)[NB][KT], f32x4 (&acc)[NB],
                                            int m, int q) {
    constexpr int MASK = (S / 4 - 1) & 15;
    const float* rowp = W + (16 * mt + m) * S;
    // all A fragments of the tile are requested before the first MFMA: with one wave per SIMD nothing else
    // hides the LDS latency, and a read issued right before its use costs ~100 cycles per 8 MFMAs
    f32x4 a[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) a[kt] = *reinterpret_cast<const f32x4*>(rowp + 4 * ((4 * kt + q) ^ (m & MASK)));
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                if (4 * kt + j < NK) acc[nb] = VPC_MFMA(a[kt][j], in[nb][kt][j], acc[nb]);
}
template <int KT, int S, int NB, int NK>  // NK: k-steps to run (the rest multiply padding zeros)
__device__ __forceinline__ void tile_T_nb_k(const float* W, int mt, const f32x4 (&in)[NB][KT], f32x4 (&acc)[NB], int m,
                                            int q) {
    constexpr int MASK = (S / 4 - 1) & 15;
    const int col = 16 * mt + m;
    const int cs = col >> 2, cl = col & 3;
    float a[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 4 * q + j;
            if (4 * kt + j < NK) a[kt][j] = W[(16 * kt + r) * S + (((cs ^ (r & MASK)) << 2) | cl)];
        }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                if (4 * kt + j < NK) acc[nb] = VPC_MFMA(a[kt][j], in[nb][kt][j], acc[nb]);
}

template <int KT, int S, int NB>
__device__ __forceinline__ void tile_T_nb(const float* W, int mt, const f32x4 (&in)[NB][KT], f32x4 (&acc)[NB], int m,
                                          int q) {
    tile_T_nb_k<KT, S, NB, 4 * KT>(W, mt, in, acc, m, q);
}

// out tile mt of  W[out][in] * in   (A fragments: one ds_read_b128 per 4 MFMAs, all requested up front)
template <int KT, int S, int NK = 4 * KT>  // NK: k-steps to run (the rest multiply padding zeros)
__device__ __forceinline__ f32x4 tile_fwd(const float* W, int mt, const f32x4 (&in)[KT], f32x4 acc, int m,
                                          int q) {
    constexpr int MASK = (S / 4 - 1) & 15;
    const float* rowp = W + (16 * mt + m) * S;
    f32x4 a[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) a[kt] = *reinterpret_cast<const f32x4*>(rowp + 4 * ((4 * kt + q) ^ (m & MASK)));
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * kt + j < NK) acc = VPC_MFMA(a[kt][j], in[kt][j], acc);
    }
    return acc;
}

// out tile mt of  W^T[in][out] * in  where `in` has KT tiles over W's ROW index (A fragment: 4 x ds_read_b32)
template <int KT, int S, int NK = 4 * KT>
__device__ __forceinline__ f32x4 tile_T(const float* W, int mt, const f32x4 (&in)[KT], f32x4 acc, int m,
                                        int q) {
    constexpr int MASK = (S / 4 - 1) & 15;
    const int col = 16 * mt + m;
    const int cs = col >> 2, cl = col & 3;
    float a[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 4 * q + j;  // row & 15
            if (4 * kt + j < NK) a[kt][j] = W[(16 * kt + r) * S + (((cs ^ (r & MASK)) << 2) | cl)];
        }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * kt + j < NK) acc = VPC_MFMA(a[kt][j], in[kt][j], acc);
    return acc;
}

// ---- wgrad staging: stage[feat][CH] fp32, 16-byte slot XOR-swizzled with feat & (CH/4 - 1)
template <int CH>
__device__ __forceinline__ void stage_write(float* st, int t, f32x4 v, int colbase, int c, int q) {
    const int col = colbase + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = 16 * t + 4 * q + j;
        st[f * CH + ((((col >> 2) ^ (f & (CH / 4 - 1))) << 2) | (col & 3))] = v[j];
    }
}
// stage_write with the lane-dependent part of the address precomputed (sb[j] = element offset of feature 4q+j in
// tile 0): the tile index t only adds the compile-time constant 16 * t * CH, which folds into the ds_write immediate
// offset - no address VALU per write (the plain form costs ~3 VALU per ds_write_b32, 60 writes per staging round).
// the XOR term uses at most the low 4 bits of the row (f & 15 = 4 q + j: it must not depend on the tile index), which
// is enough to spread a wave's 16 rows over 16 different 16-byte slots for any CH >= 64
template <int CH>
__device__ __forceinline__ constexpr int stage_swz_mask() { return CH / 4 - 1 < 15 ? CH / 4 - 1 : 15; }
template <int CH>
__device__ __forceinline__ void stage_bases(int (&sb)[4], int colbase, int c, int q) {
    const int col = colbase + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = 4 * q + j;
        sb[j] = f * CH + ((((col >> 2) ^ (f & stage_swz_mask<CH>())) << 2) | (col & 3));
    }
}
template <int CH>
__device__ __forceinline__ void stage_write_b(float* st, int t, f32x4 v, const int (&sb)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) st[16 * t * CH + sb[j]] = v[j];
}
template <int CH>
__device__ __forceinline__ f32x4 stage_frag(const float* st, int t, int s, int m, int q) {
    const int f = 16 * t + m;
    return *reinterpret_cast<const f32x4*>(st + f * CH + 4 * ((4 * s + q) ^ (f & stage_swz_mask<CH>())));
}

// ---- row-major [rows][ld] global <-> C-layout tile.  Loads are branch-free: an out-of-range element reads
// element 0 of the array (always valid) and is then replaced by 0, so the kernels stay straight-line code.
template <bool VEC>
__device__ __forceinline__ f32x4 ld_tile(const float* base, long row, int ld, int f0, int nvalid, bool row_ok) {
    const long off = row * ld + f0;
    f32x4 v;
    if (VEC) {  // ld % 4 == 0 and base 16-byte aligned: a 4-group is either fully valid or fully invalid
        const bool ok = row_ok && f0 + 3 < nvalid;
        const f32x4 t = *reinterpret_cast<const f32x4*>(base + (ok ? off : 0));
        v = ok ? t : zero4();
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = row_ok && f0 + j < nvalid;
            const float t = base[ok ? off + j : 0];
            v[j] = ok ? t : 0.f;
        }
    }
    return v;
}
// Opaque all-ones / zero mask: hipcc rewrites both `ok ? loaded : 0` and `loaded & (ok ? ~0 : 0)` into an exec-masked
// branch around the load, and the join of that branch carries `s_waitcnt vmcnt(0)`, so every such load exposes a
// full HBM latency (r01 ISA of the decoder kernel: 2-3 per output tile, ~10 in the latent prologue of each pass).
// The and-with-opaque-mask form keeps the load unconditional.  It costs a few VGPRs; the encoder kernels, which sit
// at their register limit, keep the select form below (with it enc_bwd spills: 120 -> 136 us).
__device__ __forceinline__ uint32_t opaque_mask(bool ok) {
    uint32_t m = ok ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));
    return m;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 and4(f32x4 v, uint32_t m) {
    const u32x4 b = __builtin_bit_cast(u32x4, v) & m;
    return __builtin_bit_cast(f32x4, b);
}
// ld_tile with the opaque mask (decoder kernel)
template <bool VEC>
__device__ __forceinline__ f32x4 ld_tile_o(const float* base, long row, int ld, int f0, int nvalid, bool row_ok) {
    const long off = row * ld + f0;
    f32x4 v;
    if (VEC) {
        const bool ok = row_ok && f0 + 3 < nvalid;
        v = and4(*reinterpret_cast<const f32x4*>(base + (ok ? off : 0)), opaque_mask(ok));
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = row_ok && f0 + j < nvalid;
            v[j] = __uint_as_float(__float_as_uint(base[ok ? off + j : 0]) & opaque_mask(ok));
        }
    }
    return v;
}
// ---- raw buffer access for row-tiled workspaces: the descriptor covers the rows [row0, B) of a [B][pitch] fp32 array, so a
// lane whose row is past B is out of range and its store is dropped (its load returns 0) by the hardware's range check -
// no `if (row_ok)` around the access, hence no exec-masked basic block per store and no vmcnt(0) join per load.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const float* base, long row0, long B, int pitch) {
    const long rem = (B - row0) * (long)pitch * 4;  // bytes from row0 to the end of the array
    const uint32_t rec = rem <= 0 ? 0u : (rem > 0xffffffffL ? 0xffffffffu : (uint32_t)rem);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base) + row0 * pitch, 0, rec, 0x00020000);
}
__device__ __forceinline__ void st_rows(__amdgpu_buffer_rsrc_t r, int local_row, int pitch, int f0, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, (local_row * pitch + f0) * 4, 0, 0);
}
__device__ __forceinline__ f32x4 ld_rows(__amdgpu_buffer_rsrc_t r, int local_row, int pitch, int f0) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (local_row * pitch + f0) * 4, 0, 0));
}
template <bool VEC>
__device__ __forceinline__ void st_tile(float* base, long row, int ld, int f0, int nvalid, bool row_ok, f32x4 v) {
    float* p = base + row * ld + f0;
    if (VEC) {
        if (row_ok && f0 + 3 < nvalid) *reinterpret_cast<f32x4*>(p) = v;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (row_ok && f0 + j < nvalid) p[j] = v[j];
    }
}
// mask bytes -> 0/1 floats for 4 consecutive features
template <bool VEC>
__device__ __forceinline__ f32x4 ld_mask(const uint8_t* base, long row, int ld, int f0, int nvalid, bool row_ok) {
    const long off = row * ld + f0;
    f32x4 v;
    if (VEC) {
        const bool ok = row_ok && f0 + 3 < nvalid;
        uint32_t u = *reinterpret_cast<const uint32_t*>(base + (ok ? off : 0));
        u = ok ? u : 0u;
        v = f32x4{(u & 0xffu) ? 1.f : 0.f, (u & 0xff00u) ? 1.f : 0.f, (u & 0xff0000u) ? 1.f : 0.f,
                  (u & 0xff000000u) ? 1.f : 0.f};
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = row_ok && f0 + j < nvalid;
            const uint8_t t = base[ok ? off + j : 0];
            v[j] = (ok && t) ? 1.f : 0.f;
        }
    }
    return v;
}

// raw 4 mask bytes (one per feature) as a u32; out-of-range bytes read as 0
template <bool VEC>
__device__ __forceinline__ uint32_t ld_mask_raw(const uint8_t* base, long row, int ld, int f0, int nvalid, bool row_ok) {
    const long off = row * ld + f0;
    uint32_t u = 0;
    if (VEC) {
        const bool ok = row_ok && f0 + 3 < nvalid;
        u = *reinterpret_cast<const uint32_t*>(base + (ok ? off : 0));
        u = ok ? u : 0u;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = row_ok && f0 + j < nvalid;
            const uint32_t t = base[ok ? off + j : 0];
            u |= (ok ? t : 0u) << (8 * j);
        }
    }
    return u;
}
// Four mask bytes -> four floats.  The ABI's mask contract is 0 / 1 bytes (include/vpc.h; the Python host normalises
// whatever it is given), which makes this one v_cvt_f32_ubyteN per element instead of and + compare + select: every VALU
// instruction in the MFMA kernels costs matrix-pipe time (fp32 MFMA and VALU do not overlap on a SIMD).
__device__ __forceinline__ f32x4 mask_to_f32(uint32_t u) {
    return f32x4{(float)(u & 0xffu), (float)((u >> 8) & 0xffu), (float)((u >> 16) & 0xffu), (float)(u >> 24)};
}

// copy a packed image global -> LDS (16-byte granules; n is a multiple of 4).  U loads are kept in flight per
// thread: a naive load->store loop pays one full memory latency per 16 bytes per thread (tens of us for a
// 100 KB image, once per workgroup).  With U = 8 a 97 KB image takes a 256-thread workgroup three dependent rounds,
// ~7 500 cycles (stamps, r02): a quarter of the small-batch encoder kernel's life.  The kernels pass U so that the
// whole image is ONE round of loads (13 per thread at 512 threads, 25 at 256): ~2 500 cycles.
template <int U = 8>
__device__ __forceinline__ void load_image(float* lds, const float* img, int n) {
    const int step = blockDim.x * 4;
    for (int base = threadIdx.x * 4; base < n; base += step * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * step;
            v[u] = *reinterpret_cast<const f32x4*>(img + (i < n ? i : 0));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * step;
            if (i < n) *reinterpret_cast<f32x4*>(lds + i) = v[u];
        }
    }
}

// Make a lane-id derived value opaque to the optimiser.  All LDS addresses are cheap functions of (c, q);
// without this LLVM hoists ~130 precomputed address VGPRs out of the persistent loops and spills.  Calling
// it at the head of a phase keeps address arithmetic (a few VALU ops per ds_read) next to its use.
__device__ __forceinline__ void launder(int& c, int& q) {
    asm volatile("" : "+v"(c), "+v"(q));
    __builtin_amdgcn_sched_barrier(0);  // phases are scheduled separately: bounds live ranges at 256 VGPRs
}

// Workgroup barrier for LDS hand-offs only: the wave's own LDS operations are complete (lgkmcnt(0)), global memory operations
// stay in flight.  __syncthreads() is a fence as well - hipcc puts s_waitcnt vmcnt(0) in front of s_barrier - so every
// staging round also waited for whatever had been requested from global memory ahead of its use (next tile's x / mask words,
// pass-end operands, the B fragments read straight from global memory in the encoder backward).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 64-lane sum without LDS traffic: four DPP adds give every lane its 16-lane row sum, four v_readlane + adds the
// total (uniform).  ~12 VALU instructions against 6 dependent ds_bpermute round trips for the shuffle version.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    const int i = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
}

// Sum of S partial blocks for 16 consecutive outputs per 256-thread workgroup: thread (o = tid & 15, g = tid >> 4) adds the blocks
// g, g + 16, ... of its output (four interleaved chains), the 16 group sums are combined in group order - a fixed association
// (reproducible).  One thread per output walking ALL the blocks made the reductions of small layers behind a large batch
// (hundreds of blocks, a few thousand outputs) 60 us launches of 20-30 workgroups.  The total comes back in the threads g == 0.
__device__ __forceinline__ float sum_partials_16x16(const float* p, long stride, int S, bool valid, float (*sh)[16]) {
    const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
        int sp = g;
        for (; sp + 48 < S; sp += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += p[(long)(sp + 16 * u) * stride];
        }
        for (; sp < S; sp += 16) acc[0] += p[(long)sp * stride];
    }
    sh[g][o] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    float t = 0.f;
    if (g == 0) {
        t = sh[0][o];
#pragma unroll
        for (int gg = 1; gg < 16; ++gg) t += sh[gg][o];
    }
    return t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace vpc
