// Layer-fused K-fold decoder of the MNAR step, plain bf16 MFMA inputs (gfx950): reparameterisation -> decoder (two ELU layers +
// the Sigmoid | Hardtanh heads) -> importance-weighted bound with the self-masking missingness model -> decoder backward (every
// dgrad and wgrad) -> reduction of dz over the K replicas, for the K replicas of a few data rows per workgroup tile, in ONE
// launch.  Nothing of size B * K crosses HBM (the GEMM form of the same step moves 35 GB of activations at B = 65 536).
//
// Reference semantics: REG_notMIWAE_v2 src/models/VAE.py:2382-2396 (K-fold rsample, decoder), :2398-2471 (loss) and the autograd of
// src/experiment_main/train.py:115.  Same mathematics, gradient weights and rounding points (oracle/notmiwae_oracle.py,
// rounded_linear("bf16") + elu_gate_rounded) as  vpc_nm_sample -> 3 x vpc_linear_fwd -> vpc_nm_loss -> 3 x (vpc_linear_wgrad,
// vpc_linear_dgrad) -> vpc_nm_sample_bwd  with precision = 2, which it replaces for the regularised model at obs_dim = 128.
//
// Workgroup: 4 waves (one per SIMD, 512 registers each), one 16-row slice of the 64-row tile per wave.  A tile holds the K replicas
// of nb = 64 / K data rows (rows r = bl * K + k; the last 64 - nb K rows are padding: K = 20 -> 60 of 64 rows carry work); the
// stacked passes (q rows, then p rows) are tiled separately, so a tile's pass is uniform.
// LDS (162 976 of 163 840 bytes): one bf16 image of the three layers in the compact layout of vpc_step.hip (c_elem<KP>: W1 128 x 32,
// W2 128 x 128, Wx 256 x 128) + fp32 biases + W, b of the missingness model and their softplus / sigmoid (111 KB), 25 staging
// slots of [64 rows x 16 features] bf16 for the wgrad operands (50 KB, bf_stage layout of vpc_bf16.h), 128 floats for the K-coupling.
// Registers: 204 gradient accumulators per lane (dWx 128, dW2 64, dW1 8 and ONE tile for every column sum: the three bias gradients
// and dW | db of the missingness model come from the same staged operands as MFMAs against a selector-column B operand).
// Staging rounds per tile (write - barrier - transposed reads + MFMA - barrier), slots in brackets:
//   R1a dWx[xm rows] = Gxm^T g2, dbx, db(miss) = e1^T 1   [0-7 | 8-15 | 16-23]
//   R1b dWx[xl rows] = Gxl^T g2, dbx, dW(miss) = e2^T 1   [0-7 | (8-15 kept) | 16-23]
//   R2  dW2 = dg2^T g1, db2 [0-7 | 8-15] (+ z -> 24)        R3  dW1 = dg1^T z, db1 [16-23 | 24]
// and two exchanges through LDS: the rows' bound terms l_w (softmax over the K replicas of a data row sits in 2-3 different waves)
// and dz (summed over K by one lane per (data row, latent), which also adds the analytic KL gradients).
#include "vpc_abi_internal.h"
#include "vpc_device.h"
#include "vpc_bf16.h"
#include "vpc_adam.h"
#include "vpc_dec_args.h"
#include <climits>
#include <cmath>
#include <cstring>
#include <type_traits>

namespace vpc {

// compact image helpers (same layout as vpc_step.hip; kept local: that file's are tied to its layer constants)
template <int KP>
VPC_HD constexpr int nd_key(int row) {
    return KP == 128 ? ((((row >> 1) & 3) << 2) | (((row >> 3) & 1) << 1) | (row & 1))
                     : ((row / (128 / KP)) & (KP / 8 - 1));
}
template <int KP>
VPC_HD constexpr int nd_elem(int row, int f) {  // u16 index of (row, input feature f) inside a layer image
    return row * KP + (((4 * (f >> 5) + ((f >> 2) & 3)) ^ nd_key<KP>(row)) << 3) + 4 * ((f >> 4) & 1) + (f & 3);
}

constexpr int ND_WAVES = 4, ND_THREADS = 256, ND_ROWS = 64, ND_HID = 128, ND_HT = 8;
constexpr int ND_FT = 25;                                  // staging slots per row
constexpr int ND_ST_DW = (ND_ROWS / 8) * ND_FT * 64;       // 12 800 dwords
constexpr int ND_RB_DW = ND_FT * 64;                       // one 8-row block of the staging area
constexpr int ND_NSTAT = 5;
// a tile's inputs, staged once per tile by the whole workgroup in the (then idle) staging area: x | mask | mask_p rows of its data
// rows, the eps rows of its replicas (padded to 16), the (mean | logvar) rows of its data rows (16 | 16)
// (from the third 8-row block on: the first two hold the dz exchange of the previous tile while the next one's inputs land)
constexpr int ND_NBMAX = 8, ND_XROW = 5 * 128;  // data rows per tile (K >= 8); floats per staged data row (below)
constexpr int ND_XIN = 2 * ND_RB_DW, ND_EPS = ND_XIN + ND_NBMAX * ND_XROW, ND_HD = ND_EPS + ND_ROWS * 16, ND_EPK = ND_HD + ND_NBMAX * 32,
              ND_IN_DW = ND_EPK + ND_ROWS * 16;  // (ND_EPK: the second draw of the un-regularised class's Monte-Carlo KL)
static_assert(ND_IN_DW <= ND_ST_DW, "tile inputs alias the staging area");
// dz exchange [64 rows][mean 16 | logvar 16]: 32 rows per 8-row block of the staging area, in the dwords of its slots 0-15
__host__ __device__ constexpr int nd_dzx(int row) { return (row >> 5) * ND_RB_DW + (row & 31) * 32; }
constexpr int ND_LWB = 128;  // l_w exchange: data-row groups of (K + 3) & ~3 floats, unused entries stay -inf
// dword offsets inside the image (global and LDS): bf16 layers, fp32 biases, raw W / b of the missingness model; LDS only: their
// softplus / sigmoid
struct NdImg {
    static constexpr int oW1 = 0, oW2 = oW1 + ND_HID * 16, oWx = oW2 + ND_HID * 64, ob1 = oWx + 256 * 64, ob2 = ob1 + ND_HID,
                         obx = ob2 + ND_HID, oWm = obx + 256, oBm = oWm + 128, total = oBm + 128, oSP = total, oSG = oSP + 128,
                         oISP = oSG + 128, lds_total = oISP + 128;
};
constexpr int ND_LDS = (NdImg::lds_total + ND_ST_DW + ND_LWB + 2 * ND_WAVES * ND_NSTAT) * 4;
static_assert(ND_LDS <= 163840, "LDS budget");
// partial block: [204 accumulator registers + 4 of the missingness model][256 threads]
constexpr int ND_REGS = 208, ND_PART = ND_REGS * ND_THREADS;
constexpr int R_X = 0, R_2 = 128, R_1 = 192, R_B = 200;
// columns of a wave's ONE bias accumulator tile (accb, see the kernel): bx of its four head tiles, b2, b1 of its two hidden tiles each
constexpr int NB_BX = 0, NB_B2 = 4, NB_B1 = 6;
constexpr int R_WB = 204;  // 4 more registers of wave 0's threads: dW | db of the missingness model, summed over the waves

typedef bf16x8 Op;

struct NmdArgs {
    const float* img;
    const float* x; const float* m; const float* mp;   // [B][d]
    const float* heads; long ldh;                      // [2 B][mean L | logvar L]: q rows, then p rows
    const float* eps;                                  // [2 B K][L]
    float* dht;                                        // [2 B][2 L]: gradient w.r.t. the encoder heads
    float* part;                                       // [blocks][ND_PART]
    double* stat_part;                                 // [blocks][ND_NSTAT]
    int B, K, d, L, nb, tiles_per_pass, ntiles;
    int reg;                                           // 1: REG_notMIWAE_v2 (two passes); 0: notMIWAE_myversion (one pass, Monte-Carlo KL)
    float oq, op, oe, cr, kq, kp, cd;
    int dbg;
};

__device__ __forceinline__ Op nd_pack2(f32x4 t0, f32x4 t1) {
    const u32x4 h = {pk_bf16(t0[0], t0[1]), pk_bf16(t0[2], t0[3]), pk_bf16(t1[0], t1[1]), pk_bf16(t1[2], t1[3])};
    return __builtin_bit_cast(Op, h);
}
template <int KP>
__device__ __forceinline__ Op nd_wfrag(const float* W, int mt, int kb, int m, int q) {
    return __builtin_bit_cast(Op, *reinterpret_cast<const f32x4*>(W + (16 * mt + m) * (KP / 2) + 4 * ((4 * kb + q) ^ nd_key<KP>(m))));
}
template <int KP>
__device__ __forceinline__ Op nd_wfrag_T(const float* W, int mt, int kb, int lane) {
    const int q = lane >> 4, rr = (lane >> 2) & 3, pp = lane & 3;
    const int r0 = 32 * kb + 4 * q + rr, r1 = r0 + 16;
    const int pi = 4 * (mt >> 1) + pp, e = 2 * (mt & 1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x4 h0 = ds_tr16(W + r0 * (KP / 2) + 4 * (pi ^ nd_key<KP>(r0)) + e);
    const s16x4 h1 = ds_tr16(W + r1 * (KP / 2) + 4 * (pi ^ nd_key<KP>(r1)) + e);
    const s16x8 h = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    return __builtin_bit_cast(Op, h);
}
// forward layer, TWO out tiles per step (two independent MFMA chains: one wave per SIMD has nothing else to cover the dependent
// latency with); the fragments of the next pair are requested behind the MFMAs of this one
// (bias: the lane's part of the layer's fp32 bias, bias + 4 q - the accumulators START from it, so the sink has no add to do; nullptr:
// they start from zero)
template <int KP, int KB, int NT, typename F>
__device__ __forceinline__ void nd_layer_fwd(const float* W, const Op (&in)[KB], int m, int q, F&& sink, const float* bias = nullptr) {
    static_assert(NT % 2 == 0, "");
    Op c0[KB], c1[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) { c0[kb] = nd_wfrag<KP>(W, 0, kb, m, q); c1[kb] = nd_wfrag<KP>(W, 1, kb, m, q); }
#pragma unroll
    for (int mt = 0; mt < NT; mt += 2) {
        f32x4 a0 = bias ? *reinterpret_cast<const f32x4*>(bias + 16 * mt) : zero4();
        f32x4 a1 = bias ? *reinterpret_cast<const f32x4*>(bias + 16 * mt + 16) : zero4();
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) { a0 = VPC_MFMA_BF(c0[kb], in[kb], a0); a1 = VPC_MFMA_BF(c1[kb], in[kb], a1); }
        if (mt + 2 < NT) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) { c0[kb] = nd_wfrag<KP>(W, mt + 2, kb, m, q); c1[kb] = nd_wfrag<KP>(W, mt + 3, kb, m, q); }
        }
        asm volatile("" : "+v"(a0), "+v"(a1));
        __builtin_amdgcn_sched_barrier(0);
        sink(mt, a0, a1);
    }
}
// dgrad layer: NT in-feature tiles (two per step; one per step when the fragments of two tiles would be 64 registers), KB k-blocks
// over the image's rows
template <int KP, int KB, int NT, typename F>
__device__ __forceinline__ void nd_layer_T(const float* W, const Op (&in)[KB], int lane, F&& sink) {
    if constexpr (NT == 1) {
        f32x4 a0 = zero4();
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) a0 = VPC_MFMA_BF(nd_wfrag_T<KP>(W, 0, kb, lane), in[kb], a0);
        sink(0, a0, a0);
    } else if constexpr (KB > 4) {
        // two accumulator chains over the k-blocks of ONE tile pair, fragments fetched in two halves of KB / 2 blocks
        constexpr int HB = KB / 2;
#pragma unroll
        for (int mt = 0; mt < NT; mt += 2) {
            f32x4 a0 = zero4(), a1 = zero4();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Op c0[HB], c1[HB];
#pragma unroll
                for (int kb = 0; kb < HB; ++kb) {
                    c0[kb] = nd_wfrag_T<KP>(W, mt, h * HB + kb, lane);
                    c1[kb] = nd_wfrag_T<KP>(W, mt + 1, h * HB + kb, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kb = 0; kb < HB; ++kb) {
                    a0 = VPC_MFMA_BF(c0[kb], in[h * HB + kb], a0);
                    a1 = VPC_MFMA_BF(c1[kb], in[h * HB + kb], a1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            sink(mt, a0, a1);
        }
    } else {
        Op c0[KB], c1[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) { c0[kb] = nd_wfrag_T<KP>(W, 0, kb, lane); c1[kb] = nd_wfrag_T<KP>(W, 1, kb, lane); }
#pragma unroll
        for (int mt = 0; mt < NT; mt += 2) {
            f32x4 a0 = zero4(), a1 = zero4();
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) { a0 = VPC_MFMA_BF(c0[kb], in[kb], a0); a1 = VPC_MFMA_BF(c1[kb], in[kb], a1); }
            if (mt + 2 < NT) {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) { c0[kb] = nd_wfrag_T<KP>(W, mt + 2, kb, lane); c1[kb] = nd_wfrag_T<KP>(W, mt + 3, kb, lane); }
            }
            __builtin_amdgcn_sched_barrier(0);
            sink(mt, a0, a1);
        }
    }
}
// the two heads, tile by tile: out tiles t (mean head) and DT + t (log-variance head) as two MFMA chains; pre(t) runs BEFORE the
// tile's MFMAs (LDS reads issued there arrive under them), sink(t, mean tile, logvar tile) after; the next tile's weight
// fragments are requested behind the MFMAs
// (bm / bl: the lane's part of the two heads' biases - the accumulators start from them)
template <int DT, typename P, typename F>
__device__ __forceinline__ void nd_heads(const float* W, const Op (&in)[4], int m, int q, P&& pre, F&& sink, const float* bm,
                                         const float* bl) {
    Op c0[4], c1[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) { c0[kb] = nd_wfrag<128>(W, 0, kb, m, q); c1[kb] = nd_wfrag<128>(W, DT, kb, m, q); }
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        pre(t);
        f32x4 a0 = *reinterpret_cast<const f32x4*>(bm + 16 * t), a1 = *reinterpret_cast<const f32x4*>(bl + 16 * t);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) { a0 = VPC_MFMA_BF(c0[kb], in[kb], a0); a1 = VPC_MFMA_BF(c1[kb], in[kb], a1); }
        if (t + 1 < DT) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) { c0[kb] = nd_wfrag<128>(W, t + 1, kb, m, q); c1[kb] = nd_wfrag<128>(W, DT + t + 1, kb, m, q); }
        }
        asm volatile("" : "+v"(a0), "+v"(a1));
        __builtin_amdgcn_sched_barrier(0);
        sink(t, a0, a1);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// staging: a packed operand (tiles 2 kb, 2 kb + 1 of the lane's row) into slots slot0 + 2 kb (+ 1)
template <bool BOTH = true, int FT = ND_FT>
__device__ __forceinline__ void nd_st_op(float* st, int row, int slot0, int kb, int q, Op op) {
    const u32x4 h = __builtin_bit_cast(u32x4, op);
    const int o0 = bf_stage_off<FT>(row, slot0 + 2 * kb, q);
    *reinterpret_cast<u32x2*>(st + o0) = u32x2{h[0], h[1]};
    if (BOTH) *reinterpret_cast<u32x2*>(st + o0 + 64) = u32x2{h[2], h[3]};
}
template <int FT = ND_FT>
__device__ __forceinline__ Op nd_st_frag(const float* st, int slot, int kb, int lane) {
    const int g = lane >> 4, rr = (lane >> 2) & 3, pp = lane & 3;
    const int off = bf_stage_off<FT>(32 * kb + 4 * g + rr, slot, pp);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x4 h0 = ds_tr16(st + off), h1 = ds_tr16(st + off + 128 * FT);
    const s16x8 h = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    return __builtin_bit_cast(Op, h);
}
__device__ __forceinline__ f32x4 elu4(f32x4 v) {  // ELU(alpha = 1), hardware exp as the GEMM epilogue (vpc_gemm.hip)
    return f32x4{v[0] > 0.f ? v[0] : __expf(v[0]) - 1.f, v[1] > 0.f ? v[1] : __expf(v[1]) - 1.f,
                 v[2] > 0.f ? v[2] : __expf(v[2]) - 1.f, v[3] > 0.f ? v[3] : __expf(v[3]) - 1.f};
}
// dy * ELU'(pre) from the PACKED (bf16-rounded) activation h = ELU(pre): 1 where h > 0, else h + 1  (the rounding point the
// emulating oracle mirrors: notmiwae_oracle.NMTorchPort(elu_gate_rounded=True))
__device__ __forceinline__ f32x4 elu_gate(f32x4 dy, Op act, int second) {
    const u32x4 h = __builtin_bit_cast(u32x4, act);
    const uint32_t w0 = second ? h[2] : h[0], w1 = second ? h[3] : h[1];
    const float a0 = __uint_as_float(w0 << 16), a1 = __uint_as_float(w0 & 0xffff0000u);
    const float a2 = __uint_as_float(w1 << 16), a3 = __uint_as_float(w1 & 0xffff0000u);
    // ELU' = 1 where the activation is positive, else activation + 1 (<= 1 there): min(a + 1, 1) - add, min, multiply instead of add,
    // multiply, compare, select; the same value in both branches
    return f32x4{dy[0] * fminf(a0 + 1.f, 1.f), dy[1] * fminf(a1 + 1.f, 1.f), dy[2] * fminf(a2 + 1.f, 1.f), dy[3] * fminf(a3 + 1.f, 1.f)};
}

// sum over the 16 lanes of a DPP row (the 16 batch rows of a lane group): every lane gets it.  The DPP operand is written as part of
// the add (v_add_f32_dpp) - through the update_dpp builtin hipcc emitted a v_mov_b32_dpp AND an add per step, 256 extra VALU
// instructions per q tile.
__device__ __forceinline__ float row_sum_dpp(float v) {
    float t;
    // (s_nop 1 in front of every DPP read: a VGPR written by the previous VALU instruction needs two wait states before a DPP
    // operand may read it, and the hazard recogniser does not look inside inline assembly)
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(t)
        : "v"(v));
    return t;
}

// four such sums at once, interleaved: every DPP add reads a register written three instructions earlier, so the two wait states
// the hazard asks for are filled with the other sums' adds instead of s_nop (one s_nop 1 in front of the first round only)
__device__ __forceinline__ void row_sum_dpp4(float (&v)[4]) {
    float t0, t1, t2, t3;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    v[0] = t0; v[1] = t1; v[2] = t2; v[3] = t3;
}

#ifdef VPC_ABLATE
#define ND_BARRIER() do { if (!(a.dbg & 2)) lds_barrier(); } while (0)
#define NSTP(i) VPC_STAMP(i)
#else
#define ND_BARRIER() lds_barrier()
#define NSTP(i) do {} while (0)
#endif

template <int DT, bool REG>
__global__ __launch_bounds__(ND_THREADS) void nmdec_kernel(NmdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef VPC_ABLATE
    unsigned long long T[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
    const float* W1 = lds + NdImg::oW1;
    const float* W2 = lds + NdImg::oW2;
    const float* Wx = lds + NdImg::oWx;
    const float* b1 = lds + NdImg::ob1;
    const float* b2 = lds + NdImg::ob2;
    const float* bx = lds + NdImg::obx;
    const float* Bm = lds + NdImg::oBm;
    float* SP = lds + NdImg::oSP;
    float* SG = lds + NdImg::oSG;
    float* ISP = lds + NdImg::oISP;
    float* st = lds + NdImg::lds_total;
    float* lwbuf = st + ND_ST_DW;
    double* red = reinterpret_cast<double*>(lwbuf + ND_LWB);
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = lane & 15, q = lane >> 4;
    const int K = a.K, L = a.L, d = a.d;
    constexpr int YT = 2 * DT;  // head tiles: xm 0 .. DT - 1, xl DT .. 2 DT - 1

    // the lane's row inside a tile: replica k of tile-local data row bl
    const int KP4 = (K + 3) & ~3;

    // Column sums over the rows (the three bias gradients, dW | db of the missingness model) come from the same staged operands as
    // the weight gradients: an MFMA of the staged dY^T tile against a B operand whose column n is all ones and the others zero adds
    // the tile's 16 column sums into column n of ONE accumulator tile - 4 registers for the wave's twelve such tiles instead of
    // 4 each (with 48 more accumulators the kernel needs more than 512 registers and hipcc evicts accumulators to scratch).
    f32x4 accx[4][8], acc2[2][8], acc1[2], accb = zero4();
    // dW | db of the missingness model: row sums of e1 = w dn and e2 = e1 (mix - b) in fp32.  Value e = 4 t + j of a lane group (tile
    // t, register j: feature 16 t + 4 q + j) is summed over the group's 16 rows by four DPP adds and kept by lane c = e & 15 in
    // accumulator e >> 4: two registers per lane and array for the 32 values.
    float acc_e1[2] = {0.f, 0.f}, acc_e2[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) accx[i][j] = zero4();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        acc1[i] = zero4();
#pragma unroll
        for (int j = 0; j < 8; ++j) acc2[i][j] = zero4();
    }
    float S[ND_NSTAT] = {0.f, 0.f, 0.f, 0.f, 0.f};  // lse_q (+ KL_q), lse_p (+ KL_p), sum_k RE_e, sum_l kl_el, sum_k RE_q
    auto sel_col = [&](int n, int cc) {  // B operand: column n (lanes with c == n) = eight bf16 ones, every other column zero
        const uint32_t v = cc == n ? 0x3F803F80u : 0u;
        return __builtin_bit_cast(Op, u32x4{v, v, v, v});
    };

    // A tile's inputs in flight, per thread: x | mask | mask_p of one (data row, 4 features) item - nb * 32 items, nb <= 8 -, the eps
    // values of one (row, quad) and the statistics of one (data row, mean | logvar, quad).  What every replica of a data row would
    // otherwise re-derive per element is formed ONCE when the item is stored (store_inputs): with om = 1 - m,
    //   q pass: [x | mA = m | mE = m (1 - mp) | A = -softplus(W) om | C = -softplus(W) (x m - b)]   so that logits = xm A + C
    //   p pass: [x | mA = mp]
    f32x4 pfx[3], pfe, pfh, pfk = zero4();
    auto tile_origin = [&](int tile, int& pass, int& b0) {
        pass = tile / a.tiles_per_pass;
        b0 = (tile - pass * a.tiles_per_pass) * a.nb;
    };
    // (the thread id is made opaque in each of these: their address arithmetic - the division by K included - is otherwise
    // hoisted out of the tile loop, and what is hoisted lives in scratch: ~20 reloads per tile)
    auto opaque_tid = [&]() {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        return t;
    };
    auto request_inputs = [&](int tile) {
        int pass, b0;
        tile_origin(tile, pass, b0);
        const long m0 = ((long)pass * a.B + b0) * K;  // first decoder row (eps row) of the tile
        const int tid = opaque_tid();
        {
            const int it = tid < a.nb * 32 ? tid : 0;
            const int row = it >> 5, c4 = it & 31;
            const int br = b0 + row < a.B ? b0 + row : a.B - 1;
            const long o = (long)br * d + 4 * c4;
            pfx[0] = *reinterpret_cast<const f32x4*>(a.x + o);
            pfx[1] = *reinterpret_cast<const f32x4*>(a.m + o);
            pfx[2] = REG ? *reinterpret_cast<const f32x4*>(a.mp + o) : zero4();
        }
        {
            const int row = tid >> 2, qd = tid & 3;
            const int rb = row / K;
            const bool rok = row < a.nb * K && b0 + rb < a.B;
#pragma unroll
            for (int j = 0; j < 4; ++j) pfe[j] = (rok && 4 * qd + j < L) ? a.eps[(m0 + row) * L + 4 * qd + j] : 0.f;
            if (!REG) {  // eps_kl: the second [B K][L] array behind the first
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    pfk[j] = (rok && 4 * qd + j < L) ? a.eps[((long)a.B * K + m0 + row) * L + 4 * qd + j] : 0.f;
            }
        }
        pfh = zero4();
        if (tid < a.nb * 8) {
            const int row = tid >> 3, which = (tid >> 2) & 1, qd = tid & 3;
            const int br = b0 + row < a.B ? b0 + row : a.B - 1;
            const float* src = a.heads + ((long)pass * a.B + br) * a.ldh + which * L;
#pragma unroll
            for (int j = 0; j < 4; ++j) pfh[j] = (4 * qd + j < L) ? src[4 * qd + j] : 0.f;
        }
    };
    auto store_inputs = [&](bool qpass) {
        const int tid = opaque_tid();
        if (tid < a.nb * 32) {
            const int row = tid >> 5, c4 = tid & 31;
            float* dst = st + ND_XIN + row * ND_XROW + 4 * c4;
            const f32x4 x4 = pfx[0], m4 = pfx[1], p4 = pfx[2];
            *reinterpret_cast<f32x4*>(dst) = x4;
            if (qpass) {
                const f32x4 sp = *reinterpret_cast<const f32x4*>(SP + 4 * c4), bj = *reinterpret_cast<const f32x4*>(Bm + 4 * c4);
                *reinterpret_cast<f32x4*>(dst + 128) = m4;
                *reinterpret_cast<f32x4*>(dst + 256) = m4 * (1.f - p4);
                *reinterpret_cast<f32x4*>(dst + 384) = -sp * (1.f - m4);
                *reinterpret_cast<f32x4*>(dst + 512) = -sp * (x4 * m4 - bj);
            } else {
                *reinterpret_cast<f32x4*>(dst + 128) = p4;
                *reinterpret_cast<f32x4*>(dst + 256) = zero4();
                *reinterpret_cast<f32x4*>(dst + 384) = zero4();
                *reinterpret_cast<f32x4*>(dst + 512) = zero4();
            }
        }
        *reinterpret_cast<f32x4*>(st + ND_EPS + (tid >> 2) * 16 + 4 * (tid & 3)) = pfe;
        if (!REG) *reinterpret_cast<f32x4*>(st + ND_EPK + (tid >> 2) * 16 + 4 * (tid & 3)) = pfk;
        if (tid < a.nb * 8) *reinterpret_cast<f32x4*>(st + ND_HD + (tid >> 3) * 32 + ((tid >> 2) & 1) * 16 + 4 * (tid & 3)) = pfh;
    };
    if ((int)blockIdx.x < a.ntiles) request_inputs(blockIdx.x);
    // (the first tile's inputs are on their way while the weight image is copied: one tile per workgroup at the reference's batch 128)
    load_image<27>(lds, a.img, NdImg::total);
    __syncthreads();
    if (threadIdx.x < ND_LWB) lwbuf[threadIdx.x] = -INFINITY;
    if (threadIdx.x < 128) {
        const float wv = lds[NdImg::oWm + threadIdx.x];
        const float e = expf(-fabsf(wv));
        const float spv = wv > 20.f ? wv : log1pf(expf(wv));
        SP[threadIdx.x] = spv;
        ISP[threadIdx.x] = 1.f / spv;
        SG[threadIdx.x] = wv >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    }
    __syncthreads();

    NSTP(0);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int cc = c, qq = q;
        launder(cc, qq);
        // the lane's row inside the tile: replica k of tile-local data row bl - re-derived per tile from the laundered lane id (kept
        // across the loop these live in scratch)
        const int r = 16 * w + cc;
        const int bl = r / K, k = r - bl * K;
        const bool rvalid = r < a.nb * K;
        const int blc = rvalid ? bl : 0;  // (padding rows read the first data row's inputs and l_w; their weights are zero)
        const int pass = tile / a.tiles_per_pass;
        const bool qpass = pass == 0;
        const int b0 = (tile - pass * a.tiles_per_pass) * a.nb;
        const int b = b0 + bl;
        const bool valid = rvalid && b < a.B;
        // ---------------- the tile's inputs -> LDS, once per tile (every replica of a data row reads the same x / masks / statistics);
        // they were requested from global memory while the previous tile's staging rounds ran (request_inputs below)
        store_inputs(qpass);
        ND_BARRIER();  // B0
        // ---------------- reparameterisation (VAE.py:2385-2389): z = mean + eps * exp(logvar / 2)   (columns >= L are staged as 0)
        f32x4 z, ehs;  // ehs = eps * exp(logvar / 2) / 2: d z / d logvar
        f32x4 zk = zero4(), hk = zero4();  // un-regularised class: z' of the KL draw and eps_kl sd / 2
        float klmc = 0.f;
        {
            const f32x4 mu = *reinterpret_cast<const f32x4*>(st + ND_HD + blc * 32 + 4 * qq);
            const f32x4 lv = *reinterpret_cast<const f32x4*>(st + ND_HD + blc * 32 + 16 + 4 * qq);
            const f32x4 e = *reinterpret_cast<const f32x4*>(st + ND_EPS + r * 16 + 4 * qq);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sd = __expf(0.5f * lv[j]);
                const float esd = e[j] * sd;
                z[j] = mu[j] + esd;
                ehs[j] = 0.5f * esd;
                if (!REG) {  // notMIWAE_myversion: KL from a second draw z' = mean + eps_kl sd (VAE.py:2786-2791), per replica
                    const float ek = *(st + ND_EPK + r * 16 + 4 * qq + j);
                    zk[j] = mu[j] + ek * sd;
                    hk[j] = 0.5f * ek * sd;
                    klmc += -0.5f * ek * ek - 0.5f * lv[j] + 0.5f * zk[j] * zk[j];
                }
            }
        }
        if (!REG) { klmc += __shfl_xor(klmc, 16, 64); klmc += __shfl_xor(klmc, 32, 64); }
        const Op zb = nd_pack2(z, zero4());
        NSTP(1);
        // ---------------- decoder forward
        // g1 = ELU(W1 z + b1) is formed here for the forward and AGAIN in front of R2 (8 MFMAs + the ELUs instead of 16 registers
        // held across the loss passes and R1, the phases with the most live state)
        const Op zin[1] = {zb};
        auto make_g1 = [&](Op (&g1b)[4]) {
            nd_layer_fwd<32, 1, ND_HT>(W1, zin, cc, qq, [&](int mt, f32x4 a0, f32x4 a1) {
                g1b[mt >> 1] = nd_pack2(elu4(a0), elu4(a1));
            }, b1 + 4 * qq);
        };
        Op g2b[4];
        {
            Op g1b[4];
            make_g1(g1b);
            launder(cc, qq);
            nd_layer_fwd<128, 4, ND_HT>(W2, g1b, cc, qq, [&](int mt, f32x4 a0, f32x4 a1) {
                g2b[mt >> 1] = nd_pack2(elu4(a0), elu4(a1));
            }, b2 + 4 * qq);
        }
        launder(cc, qq);
        VPC_CUT();
        NSTP(2);
        // ---------------- heads + bound terms of the lane's row, pass 1: sums over the features (VAE.py:2393-2396, 2405-2440).
        // The head outputs are NOT kept for pass 2 (64 registers across the exchange, beside its 64 registers of results): each
        // pass forms them tile by tile on the matrix pipe - 64 bf16 MFMAs again instead of scratch traffic.
        const float* xin = st + ND_XIN + blc * ND_XROW + 4 * qq;  // x at + 16 t; mA, mE, A, C at + 128, 256, 384, 512
        struct Elems { f32x4 xv, mA, mE, A, C; };
        auto fetch = [&](int t) {
            Elems e;
            e.xv = *reinterpret_cast<const f32x4*>(xin + 16 * t);
            e.mA = *reinterpret_cast<const f32x4*>(xin + 128 + 16 * t);
            // (unconditional: a p-pass tile stores zeros there - a branch here costs 12 register copies per tile at its join)
            e.mE = *reinterpret_cast<const f32x4*>(xin + 256 + 16 * t);
            e.A = *reinterpret_cast<const f32x4*>(xin + 384 + 16 * t);
            e.C = *reinterpret_cast<const f32x4*>(xin + 512 + 16 * t);
            return e;
        };
        // xm = sigmoid(.), xl = hardtanh(., -10, 0) of the lane's 4 features of tile t
        // (a0 / a1 arrive with the heads' biases in them: nd_heads starts its accumulators from bx)
        auto heads_act = [&](int t, f32x4 p0, f32x4 p1, f32x4& xm4, f32x4& xl4) {
            xm4 = f32x4{fast_sigmoid(p0[0]), fast_sigmoid(p0[1]), fast_sigmoid(p0[2]), fast_sigmoid(p0[3])};
            // Hardtanh(-10, 0) as ONE v_med3_f32 (fminf(fmaxf()) is four instructions with the canonicalisation of its NaN rule)
            xl4 = f32x4{__builtin_amdgcn_fmed3f(p1[0], -10.f, 0.f), __builtin_amdgcn_fmed3f(p1[1], -10.f, 0.f),
                        __builtin_amdgcn_fmed3f(p1[2], -10.f, 0.f), __builtin_amdgcn_fmed3f(p1[3], -10.f, 0.f)};
        };
        float sA = 0.f, sE = 0.f, sN = 0.f;
        {
            Elems cur;
            nd_heads<DT>(Wx, g2b, cc, qq, [&](int t) { cur = fetch(t); }, [&](int t, f32x4 a0, f32x4 a1) {
                f32x4 xm4, xl4;
                heads_act(t, a0, a1, xm4, xl4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xm = xm4[j], xl = xl4[j];
                    const float rr = cur.xv[j] - xm, riv = rr * __expf(-xl);
                    const float el = fmaf(rr, riv, xl);  // 2 x the element NLL (without its constant): the halves are taken once, below
                    sA = fmaf(cur.mA[j], el, sA);
                    if (qpass) {
                        sE = fmaf(cur.mE[j], el, sE);
                        const float lg = xm * cur.A[j] + cur.C[j];  // -softplus(W) (xm (1 - m) + x m - b)
                        // softplus(lg) - lg m; the log's argument is in (1, 2]: v_log_f32 directly (__logf is the full-range
                        // expansion, 12 instructions with its denormal scaling)
                        sN += fmaxf(lg, 0.f) - lg * cur.mA[j] + 0.6931471805599453f * __builtin_amdgcn_logf(1.f + __expf(-fabsf(lg)));
                    }
                }
            }, bx + 4 * qq, bx + 16 * DT + 4 * qq);
        }
        launder(cc, qq);
        VPC_CUT();
        NSTP(3);
        // (pass 2 must not be merged with pass 1 by common-subexpression elimination - everything it would carry across the
        // exchange ends up in scratch: the asm makes its operand a new value)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) asm volatile("" : "+v"(g2b[kb]));
        sA *= 0.5f; sE *= 0.5f;
        sA += __shfl_xor(sA, 16, 64); sA += __shfl_xor(sA, 32, 64);
        if (qpass) {
            sE += __shfl_xor(sE, 16, 64); sE += __shfl_xor(sE, 32, 64);
            sN += __shfl_xor(sN, 16, 64); sN += __shfl_xor(sN, 32, 64);
        }
        const float RE = sA + a.cd;
        // (regularised class: the analytic KL of the data row is the same for its K replicas and cancels in the softmax; the
        // Monte-Carlo KL of the other class differs per replica)
        const float lw = RE + sN + klmc;
        if (qq == 0 && rvalid) lwbuf[bl * KP4 + k] = lw;
        NSTP(4);
        ND_BARRIER();  // B1
        float wgt;     // softmax weight of the replica x the gradient weight of its pass
        {
            const float* lwp = lwbuf + blc * KP4;
            float mx = -INFINITY, s = 0.f;
            if (KP4 == 20) {  // config 3's K: all five chunks in flight at once
                f32x4 v[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) v[i] = *reinterpret_cast<const f32x4*>(lwp + 4 * i);
#pragma unroll
                for (int i = 0; i < 5; ++i) mx = fmaxf(mx, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    s += (__expf(v[i][0] - mx) + __expf(v[i][1] - mx)) + (__expf(v[i][2] - mx) + __expf(v[i][3] - mx));
            } else {
                for (int i = 0; i < KP4; i += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(lwp + i);
                    mx = fmaxf(mx, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
                }
                for (int i = 0; i < KP4; i += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(lwp + i);
                    s += (__expf(v[0] - mx) + __expf(v[1] - mx)) + (__expf(v[2] - mx) + __expf(v[3] - mx));
                }
            }
            const float lse = mx + 0.6931471805599453f * __builtin_amdgcn_logf(s);  // (s in [1, K])
            wgt = valid ? (qpass ? a.oq : a.op) * __expf(lw - lse) : 0.f;
            if (valid && qq == 0) {
                if (k == 0) S[qpass ? 0 : 1] += lse;
                if (qpass) { if (REG) S[2] += sE + a.cd; S[4] += RE; }
            }
        }
        NSTP(5);
        // ---------------- pass 2: gradients w.r.t. the head PRE-activations (through Sigmoid / Hardtanh), packed as they are made
        // (each tile's results are packed at once - 2 registers per tile and array; carried as fp32 until the partner tile of a
        // 32-feature operand is done they are 16 more live registers in the phase that has the fewest to spare)
        u32x2 gmh[DT], glh[DT];
        {
            const float oe = valid ? a.oe : 0.f;
            Elems cur;
            nd_heads<DT>(Wx, g2b, cc, qq, [&](int t) { cur = fetch(t); }, [&](int t, f32x4 a0, f32x4 a1) {
                f32x4 xm4, xl4;
                heads_act(t, a0, a1, xm4, xl4);
                const f32x4 isp = *reinterpret_cast<const f32x4*>(ISP + 16 * t + 4 * qq);
                // the missingness model's part (q pass only) in ONE uniform branch, row sums included: only e1A (its term of the
                // gradient w.r.t. xm) leaves it - values defined on one side of a branch and used behind its join cost a register
                // copy each, and the common part below is the same expression in both passes (a p tile has mE = A = 0)
                f32x4 gm, gl, e1A = zero4();
                if (qpass) {
                    float e1[4], e2[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float lg = xm4[j] * cur.A[j] + cur.C[j];
                        const float el2 = __expf(-fabsf(lg)), rc = __builtin_amdgcn_rcpf(1.f + el2);
                        const float dn = (lg >= 0.f ? rc : el2 * rc) - cur.mA[j];
                        e1[j] = wgt * dn;
                        e2[j] = -e1[j] * lg * isp[j];  // e1 (xm (1 - m) + x m - b): the logit divided back by -softplus(W)
                        e1A[j] = e1[j] * cur.A[j];     // (-dn softplus(W) (1 - m) = dn A)
                    }
                    row_sum_dpp4(e1);
                    row_sum_dpp4(e2);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * t + j;
                        if (cc == (e & 15)) { acc_e1[e >> 4] += e1[j]; acc_e2[e >> 4] += e2[j]; }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xm = xm4[j], xl = xl4[j];
                    const float rr = cur.xv[j] - xm, riv = rr * __expf(-xl);
                    const float h2 = 0.5f - 0.5f * rr * riv;  // d / d xl of the element NLL
                    const float wE = wgt * cur.mA[j] + oe * cur.mE[j];
                    const float gxm = e1A[j] - wE * riv, gxl = wE * h2;
                    gm[j] = gxm * fmaf(-xm, xm, xm);  // Sigmoid' = xm (1 - xm)
                    gl[j] = (xl > -10.f && xl < 0.f) ? gxl : 0.f;
                }
                gmh[t] = u32x2{pk_bf16(gm[0], gm[1]), pk_bf16(gm[2], gm[3])};
                glh[t] = u32x2{pk_bf16(gl[0], gl[1]), pk_bf16(gl[2], gl[3])};
            }, bx + 4 * qq, bx + 16 * DT + 4 * qq);
        }
        Op Gb[DT];
#pragma unroll
        for (int kb = 0; kb < DT / 2; ++kb) {
            Gb[kb] = __builtin_bit_cast(Op, u32x4{gmh[2 * kb][0], gmh[2 * kb][1], gmh[2 * kb + 1][0], gmh[2 * kb + 1][1]});
            Gb[DT / 2 + kb] = __builtin_bit_cast(Op, u32x4{glh[2 * kb][0], glh[2 * kb][1], glh[2 * kb + 1][0], glh[2 * kb + 1][1]});
        }
        launder(cc, qq);
        VPC_CUT();
        NSTP(6);
        // ---------------- R1a: dWx rows of the mean head, their bias, db of the missingness model
        const int fl = 16 * qq + cc;
        ND_BARRIER();  // B1': every wave is past its reads of the tile inputs, which the staged operands overwrite
        if (tile + (int)gridDim.x < a.ntiles) request_inputs(tile + gridDim.x);  // (arrive under the staging rounds)
#pragma unroll
        for (int kb = 0; kb < DT / 2; ++kb) nd_st_op(st, r, 0, kb, qq, Gb[kb]);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) nd_st_op(st, r, 8, kb, qq, g2b[kb]);
        ND_BARRIER();  // B2
        auto round_x = [&](auto half_c) {
            constexpr int half = decltype(half_c)::value;  // owner: wave w -> head tiles w and w + 4 of this half (DT = 8: both exist)
#pragma unroll
            for (int kb = 0; kb < ND_ROWS / 32; ++kb) {
                Op fb[8];
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) fb[nt] = nd_st_frag(st, 8 + nt, kb, fl);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (DT < 8 && w + 4 * i >= DT) continue;
                    const Op fa = nd_st_frag(st, w + 4 * i, kb, fl);
                    const int ai = 2 * half + i;
#pragma unroll
                    for (int nt = 0; nt < 8; ++nt) accx[ai][nt] = VPC_MFMA_BF(fa, fb[nt], accx[ai][nt]);
                    accb = VPC_MFMA_BF(fa, sel_col(NB_BX + ai, cc), accb);
                }
            }
        };
        round_x(std::integral_constant<int, 0>{});
        ND_BARRIER();  // B3
        // ---------------- R1b: the log-variance head (g2 stays in slots 8-15)
#pragma unroll
        for (int kb = 0; kb < DT / 2; ++kb) nd_st_op(st, r, 0, kb, qq, Gb[DT / 2 + kb]);
        ND_BARRIER();  // B4
        round_x(std::integral_constant<int, 1>{});
        launder(cc, qq);
        VPC_CUT();
        NSTP(7);
        // ---------------- dg2 = ELU'(g2) * (Wx^T G)
        Op dg2b[4];
        nd_layer_T<128, DT, ND_HT>(Wx, Gb, fl, [&](int mt, f32x4 a0, f32x4 a1) {
            dg2b[mt >> 1] = nd_pack2(elu_gate(a0, g2b[mt >> 1], 0), elu_gate(a1, g2b[mt >> 1], 1));
        });
        launder(cc, qq);
        VPC_CUT();
        NSTP(8);
        Op g1b[4];
        make_g1(g1b);
        launder(cc, qq);
        ND_BARRIER();  // B5: every wave is past the reads of R1b
        // ---------------- R2: dW2 = dg2^T g1, db2   (owner: wave w -> out tiles 2 w, 2 w + 1)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) { nd_st_op(st, r, 0, kb, qq, dg2b[kb]); nd_st_op(st, r, 8, kb, qq, g1b[kb]); }
        nd_st_op<false>(st, r, 24, 0, qq, zb);  // (R3's second operand: slot 24 is read by nobody until then)
        ND_BARRIER();  // B6
#pragma unroll
        for (int kb = 0; kb < ND_ROWS / 32; ++kb) {
            Op fb[8];
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) fb[nt] = nd_st_frag(st, 8 + nt, kb, fl);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const Op fa = nd_st_frag(st, 2 * w + i, kb, fl);
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) acc2[i][nt] = VPC_MFMA_BF(fa, fb[nt], acc2[i][nt]);
                accb = VPC_MFMA_BF(fa, sel_col(NB_B2 + i, cc), accb);
            }
        }
        launder(cc, qq);
        VPC_CUT();
        NSTP(9);
        // ---------------- dg1 = ELU'(g1) * (W2^T dg2);  dz = W1^T dg1
        Op dg1b[4];
        nd_layer_T<128, 4, ND_HT>(W2, dg2b, fl, [&](int mt, f32x4 a0, f32x4 a1) {
            dg1b[mt >> 1] = nd_pack2(elu_gate(a0, g1b[mt >> 1], 0), elu_gate(a1, g1b[mt >> 1], 1));
        });
        f32x4 dz = zero4();
        nd_layer_T<32, 4, 1>(W1, dg1b, fl, [&](int, f32x4 a0, f32x4) { dz = a0; });
        launder(cc, qq);
        NSTP(10);
        // ---------------- R3: dW1 = dg1^T z, db1.  dg1 goes to slots 16-23, which R2 does not read: no barrier in front of the writes
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) nd_st_op(st, r, 16, kb, qq, dg1b[kb]);
        ND_BARRIER();  // B8
#pragma unroll
        for (int kb = 0; kb < ND_ROWS / 32; ++kb) {
            const Op fb = nd_st_frag(st, 24, kb, fl);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const Op fa = nd_st_frag(st, 16 + 2 * w + i, kb, fl);
                acc1[i] = VPC_MFMA_BF(fa, fb, acc1[i]);
                accb = VPC_MFMA_BF(fa, sel_col(NB_B1 + i, cc), accb);
            }
        }
        NSTP(11);
        // the dz exchange lives in the dwords of slots 0-15 (all reads of R2 are behind B8): no barrier in front of the writes either
        {
            float* dzx = st + nd_dzx(r);
            f32x4 dm = valid ? dz : zero4();
            f32x4 dl = valid ? dz * ehs : zero4();
            if (!REG) {  // d KL_mc / d (mean | logvar) of the replica, weighted like the rest of its row (VAE.py:2786-2791)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dm[j] += wgt * zk[j];
                    dl[j] += wgt * (zk[j] * hk[j] - 0.5f);
                }
            }
            *reinterpret_cast<f32x4*>(dzx + 4 * qq) = dm;
            *reinterpret_cast<f32x4*>(dzx + 16 + 4 * qq) = dl;
        }
        ND_BARRIER();  // B10
        // ---------------- sum over the K replicas (nm_sample_bwd) + the analytic KL terms and their gradients (VAE.py:2441-2452):
        // wave w takes the data rows w, w + 4, ...; lane = (half of the replicas, mean | logvar, latent)
        for (int ebl = w; ebl < a.nb; ebl += ND_WAVES) {
            const int eb = b0 + ebl;
            const int col = lane & 31, part = lane >> 5, l = col & 15;
            float sm = 0.f;
            for (int kk = part; kk < K; kk += 2) sm += st[nd_dzx(ebl * K + kk) + col];
            sm += __shfl_xor(sm, 32, 64);
            if (!REG) {  // un-regularised class: no analytic terms - the Monte-Carlo KL's gradients came through the exchange
                if (lane < 32 && eb < a.B && l < L) a.dht[(long)eb * (2 * L) + (col < 16 ? l : L + l)] = sm;
                continue;
            }
            if (lane < 32 && eb < a.B && l < L) {
                const float* hq = a.heads + (long)eb * a.ldh;
                const float* hp = a.heads + ((long)a.B + eb) * a.ldh;
                const float mu_q = hq[l], lv_q = hq[L + l], mu_p = hp[l], lv_p = hp[L + l];
                const float eq = __expf(lv_q), ep = __expf(lv_p), ivp = __expf(-lv_p), ratio = __expf(lv_q - lv_p);  // (v_exp_f32: 2 ulp)
                const float dm = mu_q - mu_p;
                const bool first = col < 16;  // mean lanes also carry the statistics
                float g;
                if (qpass) {
                    g = first ? a.kq * mu_q + a.cr * dm * ivp : a.kq * 0.5f * (eq - 1.f) + a.cr * 0.5f * (ratio - 1.f);
                    if (first) {
                        S[0] += 0.5f * (eq + mu_q * mu_q - 1.f - lv_q);
                        S[3] += 0.5f * (ratio + dm * dm * ivp - 1.f - (lv_q - lv_p));
                    }
                } else {
                    g = first ? a.kp * mu_p - a.cr * dm * ivp : a.kp * 0.5f * (ep - 1.f) + a.cr * 0.5f * (1.f - ratio - dm * dm * ivp);
                    if (first) S[1] += 0.5f * (ep + mu_p * mu_p - 1.f - lv_p);
                }
                a.dht[((long)pass * a.B + eb) * (2 * L) + (first ? l : L + l)] = g + sm;
            }
        }
        // (no barrier: the next tile's inputs land behind the exchange area, and whoever stores them is past B10)
        NSTP(12);
    }

    // ---------------- partial block of the workgroup (register-major, coalesced) and its statistics
    float* part = a.part + (long)blockIdx.x * ND_PART + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(R_X + (8 * i + nt) * 4 + j) * ND_THREADS] = accx[i][nt][j];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(R_2 + (8 * i + nt) * 4 + j) * ND_THREADS] = acc2[i][nt][j];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) part[(R_1 + 4 * i + j) * ND_THREADS] = acc1[i][j];
    // the bias tile: lane column c = NB_* + index, rows 4 q + j = the feature inside its tile
#pragma unroll
    for (int j = 0; j < 4; ++j) part[(R_B + j) * ND_THREADS] = accb[j];
    // missingness model: lane (c, q), register g holds the row sums of feature f = 16 ((c + 16 g) >> 2) + 4 q + ((c + 16 g) & 3) over
    // the wave's rows; the four waves are added in wave order by wave 0 (through the idle staging area), then
    // db = softplus(W) * sum e1, dW = -sigmoid(W) * sum e2   (VAE.py:2424-2431 through autograd)
    __syncthreads();
    {
        float* xw = st;  // [4 waves][64 lanes][4]
        *reinterpret_cast<f32x4*>(xw + (w * 64 + lane) * 4) = f32x4{acc_e1[0], acc_e1[1], acc_e2[0], acc_e2[1]};
        __syncthreads();
        if (w == 0) {
            f32x4 t = *reinterpret_cast<const f32x4*>(xw + lane * 4);
#pragma unroll
            for (int ww = 1; ww < ND_WAVES; ++ww) t += *reinterpret_cast<const f32x4*>(xw + (ww * 64 + lane) * 4);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int e = c + 16 * g, f = (16 * (e >> 2) + 4 * q + (e & 3)) & 127;
                part[(R_WB + g) * ND_THREADS] = SP[f] * t[g];           // db
                part[(R_WB + 2 + g) * ND_THREADS] = -SG[f] * t[2 + g];  // dW
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ND_NSTAT; ++i) {
        const float v = wave_sum_dpp(S[i]);
        if (lane == 0) red[w * ND_NSTAT + i] = (double)v;
    }
    __syncthreads();
    if (threadIdx.x < ND_NSTAT)
        a.stat_part[(long)blockIdx.x * ND_NSTAT + threadIdx.x] =
            (red[threadIdx.x] + red[ND_NSTAT + threadIdx.x]) + (red[2 * ND_NSTAT + threadIdx.x] + red[3 * ND_NSTAT + threadIdx.x]);
#ifdef VPC_ABLATE
    NSTP(13);
    if ((a.dbg & 64) && blockIdx.x == 7 && lane == 0 && (w == 0 || w == 3))
        printf("nmdec blk %d wave %d: prologue %llu | load+z %llu g1g2 %llu Y %llu pass1 %llu (lw write) B1+lse %llu pass2 %llu R1a+R1b %llu dg2 %llu "
               "B5+R2 %llu dg1+dz %llu B7+R3 %llu B9..epi %llu | partials %llu\n",
               blockIdx.x, w, T[0], T[1], T[2], T[3], T[4], T[5], T[6], T[7], T[8], T[9], T[10], T[11], T[12], T[13]);
#endif
}

// ------------------------------------------------------------------------------------------------ encoder forward
// The encoder of the same step (seq_encoder + the two heads, VAE.py:2378-2384 / :2749-2756) as ONE launch instead of three GEMMs:
// d = 128 -> 128 (ELU) -> 128 (ELU) -> (mean | logvar), a 16-row tile per wave, register-chained on the bf16 MFMA with its weights
// in LDS (72 KB: compact images of We1, We2, Wh behind the decoder's in the same image buffer, packed by the same launch).  Same
// rounding points as vpc_linear_fwd with precision 2 (bf16 operands, fp32 bias and accumulation); h1 / h2 leave as fp32 - the
// backward GEMMs read them.  At the reference's batch 128 (256 stacked rows) three dependent 5.5 us launches become one.
struct NeImg {
    static constexpr int oW1 = 0, oW2 = oW1 + ND_HID * 64, oWh = oW2 + ND_HID * 64, ob1 = oWh + 32 * 64, ob2 = ob1 + ND_HID,
                         obh = ob2 + ND_HID, total = obh + 32;
};
constexpr int NE_LDS = NeImg::total * 4;
struct NmeArgs {
    const float* img;   // encoder part of the image buffer
    const float* xin;   // [R][128] = x * mask of the stacked passes
    float* h1; float* h2; float* heads;  // [R][128], [R][128], [R][2 L]
    long R; int L;
};
__global__ __launch_bounds__(ND_THREADS) void nmenc_fwd_kernel(NmeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    load_image<19>(lds, a.img, NeImg::total);
    __syncthreads();
    const float* W1 = lds + NeImg::oW1;
    const float* W2 = lds + NeImg::oW2;
    const float* Wh = lds + NeImg::oWh;
    const float* b1 = lds + NeImg::ob1;
    const float* b2 = lds + NeImg::ob2;
    const float* bh = lds + NeImg::obh;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const long ntiles = (a.R + 15) / 16;
    for (long tile = (long)blockIdx.x * ND_WAVES + w; tile < ntiles; tile += (long)gridDim.x * ND_WAVES) {
        const long row = tile * 16 + c;
        const bool ok = row < a.R;
        const long rc = ok ? row : a.R - 1;
        Op xb[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(a.xin + rc * 128 + 32 * kb + 4 * q);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(a.xin + rc * 128 + 32 * kb + 16 + 4 * q);
            xb[kb] = nd_pack2(t0, t1);
        }
        Op h1b[4], h2b[4];
        nd_layer_fwd<128, 4, ND_HT>(W1, xb, c, q, [&](int mt, f32x4 a0, f32x4 a1) {
            const f32x4 v0 = elu4(a0 + *reinterpret_cast<const f32x4*>(b1 + 16 * mt + 4 * q));
            const f32x4 v1 = elu4(a1 + *reinterpret_cast<const f32x4*>(b1 + 16 * mt + 16 + 4 * q));
            if (ok) {
                *reinterpret_cast<f32x4*>(a.h1 + row * 128 + 16 * mt + 4 * q) = v0;
                *reinterpret_cast<f32x4*>(a.h1 + row * 128 + 16 * mt + 16 + 4 * q) = v1;
            }
            h1b[mt >> 1] = nd_pack2(v0, v1);
        });
        nd_layer_fwd<128, 4, ND_HT>(W2, h1b, c, q, [&](int mt, f32x4 a0, f32x4 a1) {
            const f32x4 v0 = elu4(a0 + *reinterpret_cast<const f32x4*>(b2 + 16 * mt + 4 * q));
            const f32x4 v1 = elu4(a1 + *reinterpret_cast<const f32x4*>(b2 + 16 * mt + 16 + 4 * q));
            if (ok) {
                *reinterpret_cast<f32x4*>(a.h2 + row * 128 + 16 * mt + 4 * q) = v0;
                *reinterpret_cast<f32x4*>(a.h2 + row * 128 + 16 * mt + 16 + 4 * q) = v1;
            }
            h2b[mt >> 1] = nd_pack2(v0, v1);
        });
        nd_layer_fwd<128, 4, 2>(Wh, h2b, c, q, [&](int, f32x4 a0, f32x4 a1) {  // out rows 0 .. 2 L - 1 (two 16-row tiles; the rest is 0)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 v = (h ? a1 : a0) + *reinterpret_cast<const f32x4*>(bh + 16 * h + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = 16 * h + 4 * q + j;
                    if (ok && o < 2 * a.L) a.heads[row * (2 * a.L) + o] = v[j];
                }
            }
        });
    }
}

// ------------------------------------------------------------------------------------------------ encoder backward
// The encoder's backward of the same step (autograd of VAE.py:2378-2384) as ONE launch + the reduction of its partial blocks, instead
// of 3 wgrad + 2 dgrad GEMMs + a reduction: dht [R][2 L] -> dWh, dbh -> dh2 = ELU'(h2) (Wh^T dht) -> dW2, db2 -> dh1 -> dW1, db1.
// 64-row tiles (4 waves x 16 rows), the images of W2 and Wh in LDS for the two dgrads (40 KB; W1 is not needed: no gradient goes to
// x), 16 staging slots for the three wgrad rounds (32 KB).  148 accumulators per lane: dW1, dW2 64 each
// (wave w: out tiles 2 w, 2 w + 1), dWh 16 (in tiles 2 w, 2 w + 1 of both out tiles), one selector-column tile for the biases.
// ELU' comes from the fp32 h1 / h2 the forward stored - as the GEMM dgrads take it -, the bias gradients from the staged bf16 dY.
constexpr int NE_FT = 16, NE_ST_DW = (ND_ROWS / 8) * NE_FT * 64;
constexpr int NEB_LDS = (ND_HID * 64 + 32 * 64 + NE_ST_DW) * 4;  // W2, Wh, staging
constexpr int NEB_REGS = 148, NEB_PART = NEB_REGS * ND_THREADS;
constexpr int RE_1 = 0, RE_2 = 64, RE_H = 128, RE_B = 144;
constexpr int NBE_B1 = 0, NBE_B2 = 2, NBE_BH = 4;  // columns of the bias tile: b1 / b2 of the wave's two tiles; bh tile w (waves 0, 1)
struct NmebArgs {
    const float* img;   // encoder part of the image buffer
    const float* xin; const float* h1; const float* h2; const float* dht;
    float* part;        // [blocks][NEB_PART]
    long R; int L;
};
__device__ __forceinline__ f32x4 elu_gate_f32(f32x4 dy, f32x4 h) {  // dy * ELU'(pre) from the fp32 output h = ELU(pre)
    return f32x4{h[0] > 0.f ? dy[0] : dy[0] * (h[0] + 1.f), h[1] > 0.f ? dy[1] : dy[1] * (h[1] + 1.f),
                 h[2] > 0.f ? dy[2] : dy[2] * (h[2] + 1.f), h[3] > 0.f ? dy[3] : dy[3] * (h[3] + 1.f)};
}
__global__ __launch_bounds__(ND_THREADS) void nmenc_bwd_kernel(NmebArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* W2 = lds;
    float* Wh = lds + ND_HID * 64;
    float* st = Wh + 32 * 64;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = lane & 15, q = lane >> 4;
    const long ntiles = (a.R + ND_ROWS - 1) / ND_ROWS;
    // the lane's row of a tile: dht (two tiles of 16 head outputs), h2, h1, x * mask (fp32, features 16 t + 4 q + j).  Every array
    // is requested for the NEXT tile right behind its last use in this one, so its latency hides under the rest of the tile
    // (one wave per SIMD has nothing else to hide it with); the first tile's requests fly under the image load.
    f32x4 d0, d1, hf2[8], hf1[8], hfx[8];
    auto req_d = [&](long tile) {
        const long row = tile * ND_ROWS + 16 * w + c;
        const bool ok = row < a.R;
        d0 = d1 = zero4();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o0 = 4 * q + j, o1 = 16 + 4 * q + j;
            if (ok && o0 < 2 * a.L) d0[j] = a.dht[row * (2 * a.L) + o0];
            if (ok && o1 < 2 * a.L) d1[j] = a.dht[row * (2 * a.L) + o1];
        }
    };
    auto req = [&](f32x4 (&h)[8], const float* src, long tile) {
        const long row = tile * ND_ROWS + 16 * w + c;
        const long rc = row < a.R ? row : a.R - 1;  // (rows past the end: dht is 0 there, whatever h holds)
#pragma unroll
        for (int t = 0; t < 8; ++t) h[t] = *reinterpret_cast<const f32x4*>(src + rc * 128 + 16 * t + 4 * q);
    };
    if ((long)blockIdx.x < ntiles) { req_d(blockIdx.x); req(hf2, a.h2, blockIdx.x); req(hf1, a.h1, blockIdx.x); req(hfx, a.xin, blockIdx.x); }
    {   // W2 and Wh of the encoder image (W1 is skipped)
        const float* src = a.img + NeImg::oW2;
        for (int i = threadIdx.x * 4; i < (ND_HID + 32) * 64; i += ND_THREADS * 4)
            *reinterpret_cast<f32x4*>(lds + i) = *reinterpret_cast<const f32x4*>(src + i);
    }
    __syncthreads();
    f32x4 acc1[2][8], acc2[2][8], acch[2][2], accb = zero4();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        acch[i][0] = acch[i][1] = zero4();
#pragma unroll
        for (int j = 0; j < 8; ++j) acc1[i][j] = acc2[i][j] = zero4();
    }
    auto sel_col = [&](int n, int cc) {
        const uint32_t v = cc == n ? 0x3F803F80u : 0u;
        return __builtin_bit_cast(Op, u32x4{v, v, v, v});
    };
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int cc = c, qq = q;
        launder(cc, qq);
        const int r = 16 * w + cc, fl = 16 * qq + cc;
        const long nxt = tile + gridDim.x;
        const Op dhb[1] = {nd_pack2(d0, d1)};
        lds_barrier();  // (the previous tile's last round is read)
        // ---- Rh: dWh = dht^T h2, dbh   [dht 0-1 | h2 8-15]
        nd_st_op<true, NE_FT>(st, r, 0, 0, qq, dhb[0]);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) nd_st_op<true, NE_FT>(st, r, 8, kb, qq, nd_pack2(hf2[2 * kb], hf2[2 * kb + 1]));
        lds_barrier();
#pragma unroll
        for (int kb = 0; kb < ND_ROWS / 32; ++kb) {
            const Op fa0 = nd_st_frag<NE_FT>(st, 0, kb, fl), fa1 = nd_st_frag<NE_FT>(st, 1, kb, fl);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const Op fb = nd_st_frag<NE_FT>(st, 8 + 2 * w + i, kb, fl);
                acch[i][0] = VPC_MFMA_BF(fa0, fb, acch[i][0]);
                acch[i][1] = VPC_MFMA_BF(fa1, fb, acch[i][1]);
            }
            if (w < 2) accb = VPC_MFMA_BF(w == 0 ? fa0 : fa1, sel_col(NBE_BH, cc), accb);
        }
        // ---- dh2 = ELU'(h2) (Wh^T dht)
        Op dh2b[4];
        nd_layer_T<128, 1, ND_HT>(Wh, dhb, fl, [&](int mt, f32x4 a0, f32x4 a1) {
            dh2b[mt >> 1] = nd_pack2(elu_gate_f32(a0, hf2[mt]), elu_gate_f32(a1, hf2[mt + 1]));
        });
        launder(cc, qq);
        if (nxt < ntiles) { req_d(nxt); req(hf2, a.h2, nxt); }
        lds_barrier();  // (Rh is read)
        // ---- R2: dW2 = dh2^T h1, db2   [dh2 0-7 | h1 8-15]
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            nd_st_op<true, NE_FT>(st, r, 0, kb, qq, dh2b[kb]);
            nd_st_op<true, NE_FT>(st, r, 8, kb, qq, nd_pack2(hf1[2 * kb], hf1[2 * kb + 1]));
        }
        lds_barrier();
#pragma unroll
        for (int kb = 0; kb < ND_ROWS / 32; ++kb) {
            Op fb[8];
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) fb[nt] = nd_st_frag<NE_FT>(st, 8 + nt, kb, fl);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const Op fa = nd_st_frag<NE_FT>(st, 2 * w + i, kb, fl);
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) acc2[i][nt] = VPC_MFMA_BF(fa, fb[nt], acc2[i][nt]);
                accb = VPC_MFMA_BF(fa, sel_col(NBE_B2 + i, cc), accb);
            }
        }
        // ---- dh1 = ELU'(h1) (W2^T dh2)
        Op dh1b[4];
        nd_layer_T<128, 4, ND_HT>(W2, dh2b, fl, [&](int mt, f32x4 a0, f32x4 a1) {
            dh1b[mt >> 1] = nd_pack2(elu_gate_f32(a0, hf1[mt]), elu_gate_f32(a1, hf1[mt + 1]));
        });
        launder(cc, qq);
        if (nxt < ntiles) req(hf1, a.h1, nxt);
        lds_barrier();  // (R2 is read)
        // ---- R1: dW1 = dh1^T x, db1   [dh1 0-7 | x * mask 8-15]
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            nd_st_op<true, NE_FT>(st, r, 0, kb, qq, dh1b[kb]);
            nd_st_op<true, NE_FT>(st, r, 8, kb, qq, nd_pack2(hfx[2 * kb], hfx[2 * kb + 1]));
        }
        if (nxt < ntiles) req(hfx, a.xin, nxt);
        lds_barrier();
#pragma unroll
        for (int kb = 0; kb < ND_ROWS / 32; ++kb) {
            Op fb[8];
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) fb[nt] = nd_st_frag<NE_FT>(st, 8 + nt, kb, fl);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const Op fa = nd_st_frag<NE_FT>(st, 2 * w + i, kb, fl);
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) acc1[i][nt] = VPC_MFMA_BF(fa, fb[nt], acc1[i][nt]);
                accb = VPC_MFMA_BF(fa, sel_col(NBE_B1 + i, cc), accb);
            }
        }
    }
    float* part = a.part + (long)blockIdx.x * NEB_PART + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part[(RE_1 + (8 * i + nt) * 4 + j) * ND_THREADS] = acc1[i][nt][j];
                part[(RE_2 + (8 * i + nt) * 4 + j) * ND_THREADS] = acc2[i][nt][j];
            }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(RE_H + (2 * i + m2) * 4 + j) * ND_THREADS] = acch[i][m2][j];
#pragma unroll
    for (int j = 0; j < 4; ++j) part[(RE_B + j) * ND_THREADS] = accb[j];
}

// fixed-order reduction of partial blocks read in LAYOUT order (16 bytes per lane, a wave reads full 128-byte lines): a workgroup
// takes 8 positions of 4 floats x 32 block groups; every thread has its loads in flight at once; summation order: blocks g, g + 32,
// ... per group, then groups 0 .. 31 (fixed: reproducible).  sink(parameter index, sum) for the positions inv_idx names.
template <typename F>
__device__ __forceinline__ void layout_sum(const float* part, int part4, int n_blocks, const int* inv_idx, int blk, F&& sink) {
    __shared__ f32x4 sh[32][8];
    const int pi = threadIdx.x & 7, bg = threadIdx.x >> 3;
    const int p4 = blk * 8 + pi;
    f32x4 s0 = zero4();
    if (p4 < part4) {
        const f32x4* p = reinterpret_cast<const f32x4*>(part) + p4;
#pragma unroll 4
        for (int b = bg; b < n_blocks; b += 32) s0 += p[(long)b * part4];
    }
    sh[bg][pi] = s0;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int pos = threadIdx.x >> 2, k = threadIdx.x & 3;
        if (blk * 8 + pos < part4) {
            float t = sh[0][pos][k];
#pragma unroll
            for (int g = 1; g < 32; ++g) t += sh[g][pos][k];
            const int id = inv_idx[4 * (blk * 8 + pos) + k];
            if (id >= 0) sink(id, t);
        }
    }
}
constexpr int ND_FIN_BLOCKS = (ND_PART / 4 + 7) / 8, NE_FIN_BLOCKS = (NEB_PART / 4 + 7) / 8;

// the encoder's partial blocks alone (vpc_nmenc_bwd)
__global__ __launch_bounds__(256) void nmenc_reduce_kernel(const float* part, int n_blocks, const int* inv_idx, float* grad) {
    layout_sum(part, NEB_PART / 4, n_blocks, inv_idx, blockIdx.x, [&](int id, float t) { grad[id] = t; });
}

// fixed-order reduction of the partial blocks into the flat gradient (W | b of the missingness model and the decoder segment)
// + the loss terms (block 0), as nm_finalize_kernel of vpc_nm.hip.  With part_e: the encoder-backward kernel's blocks too (workgroups
// behind the decoder's), and with adam.param: Adam on every finished gradient + the re-pack of its place in the bf16 image - the whole
// tail of a single-device step in one launch (vpc_nm_fused_bwd_step).
struct NmdFinArgs {
    const float* part; const double* stat_part; int n_blocks;
    const int* grad_idx; float* grad; int n;   // grad[i] = sum over blocks of part[block][grad_idx[i]] where grad_idx[i] >= 0
    const int* inv_idx;                        // optional [ND_PART]: parameter of a block position (-1: none) - the blocks are then
                                               // read in layout order (layout_sum)
    int B, K, L, reg; double alpha, inv_B;
    double* out; float* loss_f32; float* accum; long long* state; long long rng_inc;
    const float* part_e; int n_blocks_e; const int* inv_idx_e;
    AdamFuse adam;
};
__global__ __launch_bounds__(256) void nmdec_finalize_kernel(NmdFinArgs a) {
    if (blockIdx.x > 0 && a.inv_idx) {
        const int blk = blockIdx.x - 1;
        auto sink = [&](int id, float t) {
            a.grad[id] = t;
            if (a.adam.param) adam_apply(a.adam, id, t);
        };
        if (blk < ND_FIN_BLOCKS) layout_sum(a.part, ND_PART / 4, a.n_blocks, a.inv_idx, blk, sink);
        else layout_sum(a.part_e, NEB_PART / 4, a.n_blocks_e, a.inv_idx_e, blk - ND_FIN_BLOCKS, sink);
        return;
    }
    if (blockIdx.x > 0) {
        const int i = (blockIdx.x - 1) * 256 + threadIdx.x;
        if (i >= a.n) return;
        const int gi = a.grad_idx[i];
        if (gi < 0) return;
        // (bound by its 64-byte gather transactions, not by latency: 16 loads in flight per thread instead of 4 measured 12.5 us
        // against 10.9 us at 86 blocks)
        const float* p = a.part + gi;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int blk = 0;
        for (; blk + 3 < a.n_blocks; blk += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += p[(long)(blk + u) * ND_PART];
        }
        for (; blk < a.n_blocks; ++blk) acc[0] += p[(long)blk * ND_PART];
        a.grad[i] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        return;
    }
    __shared__ double red[256][ND_NSTAT];
    double s[ND_NSTAT] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < a.n_blocks; b += 256)
        for (int i = 0; i < ND_NSTAT; ++i) s[i] += a.stat_part[(long)b * ND_NSTAT + i];
    for (int i = 0; i < ND_NSTAT; ++i) red[threadIdx.x][i] = s[i];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int i = 0; i < ND_NSTAT; ++i) red[threadIdx.x][i] += red[threadIdx.x + o][i];
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

static inline bool nmdec_shape_ok(int K, int d, int L) { return d == 128 && L >= 1 && L <= 15 && K >= 8 && K <= ND_ROWS; }

}  // namespace vpc

using namespace vpc;

extern "C" {

// 1 when vpc_nmdec_step covers the shape (the regularised model only; obs_dim = 128, latent_dim <= 15, 8 <= K <= 64);
// VPC_NMDEC=0 in the environment keeps the GEMM chain (A/B runs)
int vpc_nmdec_applicable(long B, int K, int d, int L) {
    if (B <= 0 || !nmdec_shape_ok(K, d, L)) return 0;
    if (const char* e = getenv("VPC_NMDEC")) {
        if (atoi(e) == 0) return 0;
    }
    return 1;
}

// sizes the caller allocates: the image (floats), one partial block (floats) and the most workgroups a launch uses
int vpc_nmdec_layout(long B, int K, int d, int L, int* img_floats, long* part_floats, int* max_blocks) {  // (max_blocks: of the two-pass form)
    if (B <= 0 || !nmdec_shape_ok(K, d, L)) return VPC_ERR_SHAPE;
    if (img_floats) *img_floats = NdImg::total + NeImg::total;  // the decoder kernel's image, then the encoder forward kernel's
    if (part_floats) *part_floats = ND_PART;
    if (max_blocks) {
        const int nb = ND_ROWS / K;
        const long tiles = 2 * ((B + nb - 1) / nb);
        const long cap = num_cus();
        *max_blocks = (int)(tiles < cap ? tiles : cap);
    }
    return VPC_OK;
}

// index tables over the model's flat parameter buffer [W b | We1 be1 We2 be2 Wmu Wls bmu bls | Wd1 bd1 Wd2 bd2 Wxm Wxl bxm bxl]
// (notmiwae.py _flat_order; n = its length):  pack_idx[i] as vpc_step_pack_weights_bf16 reads it (u16 position inside the image,
// -(dword + 1) for values that stay fp32; the encoder's parameters sit behind the decoder kernel's image, for vpc_nmenc_fwd),
// grad_idx[i] = position of parameter i's gradient inside a partial block (-1: the encoder's parameters, whose gradients the
// GEMM chain writes)
int vpc_nmdec_build_indices(int d, int L, int hid, int* pack_idx, int* grad_idx, int n) {
    if (hid != ND_HID || !nmdec_shape_ok(8, d, L)) return VPC_ERR_SHAPE;
    if (!pack_idx || !grad_idx) return VPC_ERR_ARG;
    const int n_enc = hid * d + hid + hid * hid + hid + 2 * L * hid + 2 * L;
    const int n_dec = hid * L + hid + hid * hid + hid + 2 * d * hid + 2 * d;
    if (n != 2 * d + n_enc + n_dec) return VPC_ERR_ARG;
    for (int i = 0; i < n; ++i) { pack_idx[i] = INT_MIN; grad_idx[i] = -1; }
    // position inside a partial block of accumulator register `reg` of thread (wave w, lane 16 q + c)
    auto pos = [](int reg, int w, int q, int c) { return reg * ND_THREADS + 64 * w + 16 * q + c; };
    // missingness model: wave 0's lane (c, q), register R_WB + g (db) / R_WB + 2 + g (dW): feature 16 ((c + 16 g) >> 2) + 4 q + ((c + 16 g) & 3)
    for (int f = 0; f < d; ++f) {
        const int t = f >> 4, q = (f >> 2) & 3, j = f & 3, e = 4 * t + j, c = e & 15, g = e >> 4;
        pack_idx[f] = -(NdImg::oWm + f + 1);
        pack_idx[d + f] = -(NdImg::oBm + f + 1);
        grad_idx[f] = pos(R_WB + 2 + g, 0, q, c);
        grad_idx[d + f] = pos(R_WB + g, 0, q, c);
    }
    {  // encoder (vpc_nmenc_fwd): We1 be1 We2 be2 [Wmu ; Wls] [bmu ; bls] behind the decoder's image
        const int e0 = 2 * d, oWe1 = e0, obe1 = oWe1 + hid * d, oWe2 = obe1 + hid, obe2 = oWe2 + hid * hid, oWh = obe2 + hid,
                  obh = oWh + 2 * L * hid, I0 = NdImg::total;
        for (int r = 0; r < hid; ++r) {
            for (int f = 0; f < d; ++f) pack_idx[oWe1 + r * d + f] = 2 * (I0 + NeImg::oW1) + nd_elem<128>(r, f);
            pack_idx[obe1 + r] = -(I0 + NeImg::ob1 + r + 1);
            for (int f = 0; f < hid; ++f) pack_idx[oWe2 + r * hid + f] = 2 * (I0 + NeImg::oW2) + nd_elem<128>(r, f);
            pack_idx[obe2 + r] = -(I0 + NeImg::ob2 + r + 1);
        }
        for (int r = 0; r < 2 * L; ++r) {
            for (int f = 0; f < hid; ++f) pack_idx[oWh + r * hid + f] = 2 * (I0 + NeImg::oWh) + nd_elem<128>(r, f);
            pack_idx[obh + r] = -(I0 + NeImg::obh + r + 1);
        }
    }
    int o = 2 * d + n_enc;
    const int oWd1 = o, obd1 = oWd1 + hid * L, oWd2 = obd1 + hid, obd2 = oWd2 + hid * hid, oWx = obd2 + hid,
              obx = oWx + 2 * d * hid;
    for (int r = 0; r < hid; ++r) {  // out feature r of the two hidden layers: tile mt = r >> 4 owned by wave mt >> 1
        const int mt = r >> 4, w = mt >> 1, i = mt & 1, q = (r >> 2) & 3, j = r & 3;
        for (int f = 0; f < L; ++f) {
            pack_idx[oWd1 + r * L + f] = 2 * NdImg::oW1 + nd_elem<32>(r, f);
            grad_idx[oWd1 + r * L + f] = pos(R_1 + 4 * i + j, w, q, f);
        }
        pack_idx[obd1 + r] = -(NdImg::ob1 + r + 1);
        grad_idx[obd1 + r] = pos(R_B + j, w, q, NB_B1 + i);
        for (int f = 0; f < hid; ++f) {
            pack_idx[oWd2 + r * hid + f] = 2 * NdImg::oW2 + nd_elem<128>(r, f);
            grad_idx[oWd2 + r * hid + f] = pos(R_2 + (8 * i + (f >> 4)) * 4 + j, w, q, f & 15);
        }
        pack_idx[obd2 + r] = -(NdImg::ob2 + r + 1);
        grad_idx[obd2 + r] = pos(R_B + j, w, q, NB_B2 + i);
    }
    for (int r = 0; r < 2 * d; ++r) {  // head rows: tile t = r >> 4; half = t / 8, owner wave (t & 7) & 3, slot i = (t & 7) >> 2
        const int t = r >> 4, half = t >> 3, tt = t & 7, w = tt & 3, i = tt >> 2, ai = 2 * half + i, q = (r >> 2) & 3, j = r & 3;
        for (int f = 0; f < hid; ++f) {
            pack_idx[oWx + r * hid + f] = 2 * NdImg::oWx + nd_elem<128>(r, f);
            grad_idx[oWx + r * hid + f] = pos(R_X + (8 * ai + (f >> 4)) * 4 + j, w, q, f & 15);
        }
        pack_idx[obx + r] = -(NdImg::obx + r + 1);
        grad_idx[obx + r] = pos(R_B + j, w, q, NB_BX + ai);
    }
    return VPC_OK;
}

// launches the tile kernel of vpc_nmdec_step; *blocks_out = the partial blocks it writes
static int nmdec_launch_tiles(const float* img, const float* x, const float* mask, const float* mask_p, const float* heads, long ldh,
                              const float* eps, float* dht, float* part, double* stat_part, long B, long B_global, int K, int d, int L,
                              double alpha, hipStream_t st, int* blocks_out) {
    if (!img || !x || !mask || !heads || !eps || !dht || !part || !stat_part) return VPC_ERR_ARG;
    const int reg = mask_p != nullptr;  // NULL: notMIWAE_myversion (one pass; eps = [B K][L] draws, then the [B K][L] draws of its KL)
    if (!reg) alpha = 0.0;
    if (B <= 0 || B_global < B || ldh < 2 * L || B * (long)K > 0x3fffff00L) return VPC_ERR_ARG;
    if (!nmdec_shape_ok(K, d, L)) return VPC_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(mask_p) |
         reinterpret_cast<uintptr_t>(img)) & 15)  // (a NULL mask_p is aligned)
        return VPC_ERR_ARG;
    NmdArgs a{};
    a.img = img; a.x = x; a.m = mask; a.mp = mask_p; a.heads = heads; a.ldh = ldh; a.eps = eps; a.dht = dht; a.part = part;
    a.stat_part = stat_part; a.B = (int)B; a.K = K; a.d = d; a.L = L;
    a.nb = ND_ROWS / K;
    a.tiles_per_pass = (int)((B + a.nb - 1) / a.nb);
    a.reg = reg;
    a.ntiles = (reg ? 2 : 1) * a.tiles_per_pass;
    const double Bg = (double)B_global;
    a.oq = (float)((1.0 - alpha) / Bg); a.op = (float)(alpha / Bg); a.oe = (float)(alpha / (Bg * K));
    a.cr = (float)(alpha / (Bg * L)); a.kq = a.oq; a.kp = a.op;
    a.cd = 0.91893853320467274f * (float)d;
#ifdef VPC_ABLATE
    if (const char* e = getenv("VPC_DEBUG")) a.dbg = atoi(e);
#endif
    const int cap = num_cus();
    const int blocks = a.ntiles < cap ? a.ntiles : cap;
    if (reg) {
        if (!lds_attr_done(reinterpret_cast<const void*>(nmdec_kernel<8, true>), ND_LDS)) return VPC_ERR_HIP;
        hipLaunchKernelGGL((nmdec_kernel<8, true>), dim3(blocks), dim3(ND_THREADS), ND_LDS, st, a);
    } else {
        if (!lds_attr_done(reinterpret_cast<const void*>(nmdec_kernel<8, false>), ND_LDS)) return VPC_ERR_HIP;
        hipLaunchKernelGGL((nmdec_kernel<8, false>), dim3(blocks), dim3(ND_THREADS), ND_LDS, st, a);
    }
    *blocks_out = blocks;
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}
// launches the tile kernel of vpc_nmenc_bwd
static int nmenc_launch_bwd(const float* img, const float* xin, const float* h1, const float* h2, const float* dht, float* part,
                            long part_floats, long R, int d, int L, hipStream_t st, int* blocks_out) {
    if (!img || !xin || !h1 || !h2 || !dht || !part || R <= 0) return VPC_ERR_ARG;
    if (!nmdec_shape_ok(8, d, L)) return VPC_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(xin) | reinterpret_cast<uintptr_t>(h1) |
         reinterpret_cast<uintptr_t>(h2) | reinterpret_cast<uintptr_t>(part)) & 15)
        return VPC_ERR_ARG;
    const long tiles = (R + ND_ROWS - 1) / ND_ROWS;
    long blocks = num_cus();  // (478 registers per lane: one workgroup per CU)
    if (tiles < blocks) blocks = tiles;
    if (part_floats / NEB_PART < blocks) blocks = part_floats / NEB_PART;
    if (blocks < 1) return VPC_ERR_ARG;
    NmebArgs a{img + NdImg::total, xin, h1, h2, dht, part, R, L};
    if (!lds_attr_done(reinterpret_cast<const void*>(nmenc_bwd_kernel), NEB_LDS)) return VPC_ERR_HIP;
    hipLaunchKernelGGL(nmenc_bwd_kernel, dim3((unsigned)blocks), dim3(ND_THREADS), NEB_LDS, st, a);
    *blocks_out = (int)blocks;
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

// The fused decoder + loss + decoder backward of one regularised MNAR step (two launches: the tile kernel and the fixed-order
// reduction of its partial blocks).  heads [2 B][ldh] = the encoder's (mean | logvar) rows of the q pass, then of the p pass;
// eps [2 B K][L]; dht [2 B][2 L] receives the gradient w.r.t. heads (K-fold sum of dz + the analytic KL gradients);
// grad (the model's flat gradient buffer, n entries) receives the entries grad_idx names (inv_idx, optional: the inverse table
// [part_floats] position -> parameter or -1, for the layout-order reduction); out8 / loss_f32 / accum / state as vpc_nm_loss.  part: max_blocks x part_floats floats, stat_part: max_blocks x 5 doubles (vpc_nmdec_layout).
int vpc_nmdec_step(const float* img, const float* x, const float* mask, const float* mask_p, const float* heads, long ldh,
                   const float* eps, float* dht, float* part, double* stat_part, const int* grad_idx, const int* inv_idx, float* grad,
                   int n, double* out8, float* loss_f32, float* accum, long long* state, long long rng_inc, long B, long B_global,
                   int K, int d, int L, double alpha, void* stream) {
    if (!grad_idx || !grad || !out8) return VPC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int blocks = 0;
    const int rc = nmdec_launch_tiles(img, x, mask, mask_p, heads, ldh, eps, dht, part, stat_part, B, B_global, K, d, L, alpha, st, &blocks);
    if (rc != VPC_OK) return rc;
    const int reg = mask_p != nullptr;
    NmdFinArgs f{};
    f.part = part; f.stat_part = stat_part; f.n_blocks = blocks; f.grad_idx = grad_idx; f.inv_idx = inv_idx; f.grad = grad; f.n = n;
    f.B = (int)B; f.K = K; f.L = L; f.reg = reg; f.alpha = reg ? alpha : 0.0; f.inv_B = 1.0 / (double)B_global;
    f.out = out8; f.loss_f32 = loss_f32; f.accum = accum; f.state = state; f.rng_inc = rng_inc;
    const int fin_grid = 1 + (inv_idx ? ND_FIN_BLOCKS : (n + 255) / 256);
    hipLaunchKernelGGL(nmdec_finalize_kernel, dim3(fin_grid), dim3(256), 0, st, f);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

// Everything behind the encoder forward of a single-device MNAR step in THREE launches: the decoder tile kernel (vpc_nmdec_step's),
// the encoder-backward tile kernel (vpc_nmenc_bwd's) and ONE tail launch that sums both sets of partial blocks into grad, applies Adam
// (torch.optim.Adam as vpc_adam_step: train.py:21,116) to every gradient it finishes, re-packs the parameter's place in the bf16
// image (pack_idx of vpc_nmdec_build_indices) and writes the loss terms.  Arguments as vpc_nmdec_step + vpc_nmenc_bwd; part_e:
// max_blocks x the encoder's part_floats (vpc_nmenc_build_indices), its own buffer here: both sets are live until the tail.
int vpc_nm_fused_bwd_step(float* img, const float* x, const float* mask, const float* mask_p, const float* xin, const float* h1,
                          const float* h2, const float* heads, long ldh, const float* eps, float* dht, float* part,
                          double* stat_part, float* part_e, long part_e_floats, const int* inv_idx, const int* inv_idx_e,
                          float* grad, int n, double* out8, float* loss_f32, float* accum, long B, long B_global, int K, int d,
                          int L, double alpha, float* params, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                          float beta2, float eps_adam, long step, const int* pack_idx, void* stream) {
    if (!inv_idx || !inv_idx_e || !grad || !out8 || !params || !exp_avg || !exp_avg_sq || !pack_idx || step < 1) return VPC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int reg = mask_p != nullptr;
    int blocks = 0, blocks_e = 0;
    int rc = nmdec_launch_tiles(img, x, mask, mask_p, heads, ldh, eps, dht, part, stat_part, B, B_global, K, d, L, alpha, st, &blocks);
    if (rc != VPC_OK) return rc;
    rc = nmenc_launch_bwd(img, xin, h1, h2, dht, part_e, part_e_floats, (reg ? 2 : 1) * B, d, L, st, &blocks_e);
    if (rc != VPC_OK) return rc;
    NmdFinArgs f{};
    f.part = part; f.stat_part = stat_part; f.n_blocks = blocks; f.inv_idx = inv_idx; f.grad = grad; f.n = n;
    f.B = (int)B; f.K = K; f.L = L; f.reg = reg; f.alpha = reg ? alpha : 0.0; f.inv_B = 1.0 / (double)B_global;
    f.out = out8; f.loss_f32 = loss_f32; f.accum = accum;
    f.part_e = part_e; f.n_blocks_e = blocks_e; f.inv_idx_e = inv_idx_e;
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
    f.adam = AdamFuse{params, exp_avg, exp_avg_sq, pack_idx, img, lr, beta1, beta2, eps_adam, (float)bc1, (float)std::sqrt(bc2), 1};
    hipLaunchKernelGGL(nmdec_finalize_kernel, dim3(1 + ND_FIN_BLOCKS + NE_FIN_BLOCKS), dim3(256), 0, st, f);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

// Encoder forward of the stacked passes in one launch (img = the image buffer of vpc_nmdec_build_indices: the encoder's part is
// behind the decoder's): xin [R][128] -> h1, h2 [R][128] (fp32, ELU applied), heads [R][2 L].  Shapes as vpc_nmdec_step.
int vpc_nmenc_fwd(const float* img, const float* xin, float* h1, float* h2, float* heads, long R, int d, int L, void* stream) {
    if (!img || !xin || !h1 || !h2 || !heads || R <= 0) return VPC_ERR_ARG;
    if (!nmdec_shape_ok(8, d, L)) return VPC_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(xin) | reinterpret_cast<uintptr_t>(h1) |
         reinterpret_cast<uintptr_t>(h2)) & 15)
        return VPC_ERR_ARG;
    NmeArgs a{img + NdImg::total, xin, h1, h2, heads, R, L};
    const long tiles = (R + 15) / 16, wgs = (tiles + ND_WAVES - 1) / ND_WAVES;
    const long cap = 2L * num_cus();
    if (!lds_attr_done(reinterpret_cast<const void*>(nmenc_fwd_kernel), NE_LDS)) return VPC_ERR_HIP;
    hipLaunchKernelGGL(nmenc_fwd_kernel, dim3((unsigned)(wgs < cap ? wgs : cap)), dim3(ND_THREADS), NE_LDS, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

// Encoder backward of the stacked passes (autograd of VAE.py:2378-2384 behind d loss / d heads) in one launch + the fixed-order
// reduction of its partial blocks: dht [R][2 L], h1 / h2 [R][128] (fp32, as vpc_nmenc_fwd stored them), xin [R][128] -> the
// gradients of We1 be1 We2 be2 [Wmu ; Wls] [bmu ; bls] inside grad (the model's flat gradient buffer), at the positions
// inv_idx names (vpc_nmenc_build_indices).  part: scratch of part_floats floats (the decoder's partial blocks, already reduced,
// serve: vpc_nmdec_layout's max_blocks x part_floats is always enough for one block per 64-row tile or one per CU).
int vpc_nmenc_bwd(const float* img, const float* xin, const float* h1, const float* h2, const float* dht, float* part,
                  long part_floats, const int* inv_idx, float* grad, long R, int d, int L, void* stream) {
    if (!inv_idx || !grad) return VPC_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int blocks = 0;
    const int rc = nmenc_launch_bwd(img, xin, h1, h2, dht, part, part_floats, R, d, L, st, &blocks);
    if (rc != VPC_OK) return rc;
    hipLaunchKernelGGL(nmenc_reduce_kernel, dim3(NE_FIN_BLOCKS), dim3(256), 0, st, part, blocks, inv_idx, grad);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

// inv_idx [*part_floats] for vpc_nmenc_bwd: the flat parameter index (layout of vpc_nmdec_build_indices, n entries) whose gradient a
// position of an encoder partial block holds, -1 where none.  inv_idx == NULL: only *part_floats is written.
int vpc_nmenc_build_indices(int d, int L, int hid, int* inv_idx, long* part_floats, int n) {
    if (hid != ND_HID || !nmdec_shape_ok(8, d, L)) return VPC_ERR_SHAPE;
    if (part_floats) *part_floats = NEB_PART;
    if (!inv_idx) return VPC_OK;
    const int n_enc = hid * d + hid + hid * hid + hid + 2 * L * hid + 2 * L;
    const int n_dec = hid * L + hid + hid * hid + hid + 2 * d * hid + 2 * d;
    if (n != 2 * d + n_enc + n_dec) return VPC_ERR_ARG;
    for (int i = 0; i < NEB_PART; ++i) inv_idx[i] = -1;
    auto pos = [](int reg, int w, int q, int c) { return reg * ND_THREADS + 64 * w + 16 * q + c; };
    const int e0 = 2 * d, oWe1 = e0, obe1 = oWe1 + hid * d, oWe2 = obe1 + hid, obe2 = oWe2 + hid * hid, oWh = obe2 + hid,
              obh = oWh + 2 * L * hid;
    for (int r = 0; r < hid; ++r) {  // out feature r of the two hidden layers: tile mt = r >> 4 owned by wave mt >> 1
        const int mt = r >> 4, w = mt >> 1, i = mt & 1, q = (r >> 2) & 3, j = r & 3;
        for (int f = 0; f < d; ++f) inv_idx[pos(RE_1 + (8 * i + (f >> 4)) * 4 + j, w, q, f & 15)] = oWe1 + r * d + f;
        inv_idx[pos(RE_B + j, w, q, NBE_B1 + i)] = obe1 + r;
        for (int f = 0; f < hid; ++f) inv_idx[pos(RE_2 + (8 * i + (f >> 4)) * 4 + j, w, q, f & 15)] = oWe2 + r * hid + f;
        inv_idx[pos(RE_B + j, w, q, NBE_B2 + i)] = obe2 + r;
    }
    for (int r = 0; r < 2 * L; ++r) {  // head rows: out tile m2 = r >> 4; in feature f: tile f >> 4 owned by wave (f >> 4) >> 1
        const int m2 = r >> 4, q = (r >> 2) & 3, j = r & 3;
        for (int f = 0; f < hid; ++f) {
            const int ft = f >> 4, w = ft >> 1, i = ft & 1;
            inv_idx[pos(RE_H + (2 * i + m2) * 4 + j, w, q, f & 15)] = oWh + r * hid + f;
        }
        inv_idx[pos(RE_B + j, m2, q, NBE_BH)] = obh + r;
    }
    return VPC_OK;
}

}  // extern "C"
