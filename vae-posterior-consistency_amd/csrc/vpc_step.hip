// Whole-step kernel, plain bf16 (gfx950): encoder forward + reparameterisation + decoder + ELBO / consistency loss + decoder
// backward + encoder backward of BOTH passes of a Reg_VAE step (one pass for vanilla_VAE) in ONE launch per 128-row tile.
//
// Reference semantics: src/models/VAE.py:496-507 (forward: encoder -> decoder -> encoder -> decoder), :403-467 (loss), and
// the autograd of src/experiment_main/train.py:115.  Same mathematics, loss coefficients (include/vpc.h), rounding points
// (oracle/vae_oracle.py GemmModel "bf16") and partial-block layouts as vpc_encoder_fwd -> vpc_decoder_fused ->
// vpc_encoder_bwd with precision = 2, which it replaces for obs_dim in (64, 128] in the throughput shape: what changes is
// what crosses HBM.  The three-kernel form moves 633 MB per step at B = 65 536 (h1 / h2 round trip 212 MB, x and the masks
// read by all three kernels, latent statistics and their seeds) against 50 MB of compulsory input; here a tile's x / mask
// rows are read from HBM once (re-reads hit L2), h1 / h2 / mean / logvar / dmean / dlogvar never leave the chip, and the only
// outputs are the two gradient partial blocks and the loss terms of the workgroup.
//
// LDS (162 304 of 163 840 bytes): ONE compact bf16 image of all six layers (98.5 KB: vpc_step_layout below - the pair-slot
// image of vpc_bf16.h carries an unused lo half in plain bf16) + 15 staging slots of [128 rows x 16 features] bf16 for the
// wgrad operands (bf_stage layout of vpc_bf16.h with FT = 15).
// Registers (256 per wave, two waves per SIMD): 96 wgrad accumulators + 4 for db1, the latent statistics of both passes
// (16), and per pass the packed h1 / h2 operands (24) across the decoder phase.  Pass p's statistics are needed by pass q's
// KL(q || p) seeds before pass p runs, so the order is  E(p: statistics only) -> E(q) D(q) B(q) -> E(p) again, D(p), B(p):
// three encoder forwards per tile instead of parking 24 registers per lane.
// Staging rounds per pass (write - barrier - transposed reads + MFMA), slots in brackets:
//   R1 dW6 = dpre^T g2 [0-7 | 8-14]   R2 dW5 = dg2^T g1 [0-6 | 8-11]   R3 dW4 = dg1^T z, dW3 = dml^T h2 [0-3, 4-5 | 8, 9-12]
//   R4 dW2 = dh2^T h1 [0-3 | 8-14]    R5 dW1 = dh1^T (x*mask), db1 = dh1^T 1 [0-6 | 7-14]
// (dz and the KL seeds need no barrier - dgrads read only the weight image - so dml is known before R3 is staged.)
#include "vpc_abi_internal.h"
#include "vpc_device.h"
#include "vpc_bf16.h"
#include "vpc_dec_args.h"
#include <climits>
#include <cstring>
#include <type_traits>

namespace vpc {

// ------------------------------------------------------------------------------------------------ compact bf16 image
// Layer image: `rows` rows of KP bf16 (KP = inputs padded to 32); the 16-byte slot pi = 4 kb + q of a row holds the k-slots
// (kb, q, 0..7) = input features 32 kb + 16 (j >> 2) + 4 q + (j & 3) (vpc_bf16.h), pi XOR-swizzled with a key of the row.
// The key makes the forward fragment read (ds_read_b128: 16 rows x one slot per lane group) conflict-free for every row
// pitch - rows of 128 / 64 bytes share a 256-byte bank row in pairs / fours, so the key is taken from the row bits above
// that - and, for 256-byte rows, spreads the 8 consecutive rows of a transposed read (ds_read_b64_tr_b16) over four slot
// groups (2-way instead of 4-way conflicts).
template <int KP>
VPC_HD constexpr int c_key(int row) {
    return KP == 128 ? ((((row >> 1) & 3) << 2) | (((row >> 3) & 1) << 1) | (row & 1))
                     : ((row / (128 / KP)) & (KP / 8 - 1));
}
template <int KP>
VPC_HD constexpr int c_elem(int row, int f) {  // u16 index of (row, input feature f) inside the layer image
    return row * KP + (((4 * (f >> 5) + ((f >> 2) & 3)) ^ c_key<KP>(row)) << 3) + 4 * ((f >> 4) & 1) + (f & 3);
}
// dword offsets of the layers inside the image: [W1 112 x 128][b1 128 fp32][W2 64 x 128][W3 32 x 64][W4 64 x 32][W5 112 x 64]
// [W6 128 x 128]
struct StepImg {
    static constexpr int oW1 = 0, ob1 = oW1 + H1P * 64, oW2 = ob1 + 128, oW3 = oW2 + H2P * 64, oW4 = oW3 + 32 * 32,
                         oW5 = oW4 + H2P * 16, oW6 = oW5 + H1P * 32, total = oW6 + 128 * 64;
};
constexpr int ST_FT = 15;                              // staging slots (16-feature tiles) per row
constexpr int ST_DW = (TILE_ROWS / 8) * ST_FT * 64;    // 15 360 dwords
constexpr int STEP_LDS = (StepImg::total + ST_DW + WAVES * LOSS_TERMS) * 4;
// second staging area of sweep 2 (6 slots: dml 0-1, h2 2-5) in the LDS of the decoder layers' image, which that sweep does not read
constexpr int ST2_FT = 6;
static_assert((TILE_ROWS / 8) * ST2_FT * 64 <= StepImg::total - StepImg::oW4, "second staging area");
static_assert(STEP_LDS <= 163840, "LDS budget");

typedef bf16x8 Op;  // one MFMA operand: 8 k-slots per lane

#ifndef VPC_STEP_PREFETCH
#define VPC_STEP_PREFETCH 0  // 1: request a tile's inputs one tile ahead (see request_tile); measured, see profiles/r03_notes.md
#endif
constexpr bool PREFETCH = VPC_STEP_PREFETCH != 0;
#ifndef VPC_STEP_AHEAD
#define VPC_STEP_AHEAD 0  // staged tile inputs: the first column half of a tile is requested one tile ahead (28 registers)
#endif
constexpr bool AHEAD = VPC_STEP_AHEAD != 0;

// (lds_barrier(), vpc_device.h: s_waitcnt lgkmcnt(0); s_barrier - no vmcnt(0) as __syncthreads() carries)

__device__ __forceinline__ Op pack2(f32x4 t0, f32x4 t1) {
    const u32x4 h = {pk_bf16(t0[0], t0[1]), pk_bf16(t0[2], t0[3]), pk_bf16(t1[0], t1[1]), pk_bf16(t1[2], t1[3])};
    return __builtin_bit_cast(Op, h);
}
// forward A fragment: weight rows 16 mt + m, k-block kb
template <int KP>
__device__ __forceinline__ Op c_wfrag(const float* W, int mt, int kb, int m, int q) {
    return __builtin_bit_cast(Op, *reinterpret_cast<const f32x4*>(W + (16 * mt + m) * (KP / 2) + 4 * ((4 * kb + q) ^ c_key<KP>(m))));
}
// transposed A fragment (dgrad): in-feature tile mt, k-block kb of the layer's OUT features (rows of the image)
template <int KP, bool SECOND>
__device__ __forceinline__ Op c_wfrag_T(const float* W, int mt, int kb, int lane) {
    const int q = lane >> 4, rr = (lane >> 2) & 3, pp = lane & 3;
    const int r0 = 32 * kb + 4 * q + rr, r1 = r0 + 16;
    const int pi = 4 * (mt >> 1) + pp, e = 2 * (mt & 1);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x4 zz = {0, 0, 0, 0};
    const s16x4 h0 = ds_tr16(W + r0 * (KP / 2) + 4 * (pi ^ c_key<KP>(r0)) + e);
    const s16x4 h1 = SECOND ? ds_tr16(W + r1 * (KP / 2) + 4 * (pi ^ c_key<KP>(r1)) + e) : zz;
    const s16x8 h = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    return __builtin_bit_cast(Op, h);
}
// forward layer: NT out tiles, KB k-blocks.  ONE set of fragment registers: the fragments of tile mt + 1 are requested right
// behind the MFMAs of tile mt (which have read theirs at issue) and arrive under the sink's VALU and the partner wave's work.
// (A second set - next tile requested before the MFMAs - costs 4 KB registers per layer call, which this kernel does not have.)
template <int KP, int KB, int NT, bool DB = false, typename F>
__device__ __forceinline__ void c_layer_fwd(const float* W, const Op (&in)[KB], int m, int q, F&& sink) {
    Op cur[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) cur[kb] = c_wfrag<KP>(W, 0, kb, m, q);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        Op nxt[DB ? KB : 1];  // DB: a second set, requested BEFORE this tile's MFMAs (where the registers are there)
        if (DB && mt + 1 < NT) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) nxt[DB ? kb : 0] = c_wfrag<KP>(W, mt + 1, kb, m, q);
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 acc = zero4();
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) acc = VPC_MFMA_BF(cur[kb], in[kb], acc);
        if (mt + 1 < NT) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) cur[kb] = DB ? nxt[DB ? kb : 0] : c_wfrag<KP>(W, mt + 1, kb, m, q);
        }
        __builtin_amdgcn_sched_barrier(0);
        sink(mt, acc);
    }
}
// the same for TWO inputs (the two passes of a tile): every weight fragment feeds two independent MFMA chains - half the LDS
// fragment reads per product and a second chain in flight while the first one's result is on its way
template <int KP, int KB, int NT, typename F>
__device__ __forceinline__ void c_layer_fwd2(const float* W, const Op (&in0)[KB], const Op (&in1)[KB], int m, int q, F&& sink) {
    Op cur[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) cur[kb] = c_wfrag<KP>(W, 0, kb, m, q);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            acc0 = VPC_MFMA_BF(cur[kb], in0[kb], acc0);
            acc1 = VPC_MFMA_BF(cur[kb], in1[kb], acc1);
        }
        if (mt + 1 < NT) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) cur[kb] = c_wfrag<KP>(W, mt + 1, kb, m, q);
        }
        __builtin_amdgcn_sched_barrier(0);
        sink(mt, acc0, acc1);
    }
}
// dgrad layer: NT in-feature tiles, KB k-blocks over the image's ROWT 16-row tiles
template <int KP, int KB, int NT, int ROWT, typename F>
__device__ __forceinline__ void c_layer_T(const float* W, const Op (&in)[KB], int lane, F&& sink) {
    auto frag = [&](int mt, int kb) {
        return (2 * kb + 1 < ROWT) ? c_wfrag_T<KP, true>(W, mt, kb, lane) : c_wfrag_T<KP, false>(W, mt, kb, lane);
    };
    Op cur[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) cur[kb] = frag(0, kb);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        f32x4 acc = zero4();
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) acc = VPC_MFMA_BF(cur[kb], in[kb], acc);
        if (mt + 1 < NT) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) cur[kb] = frag(mt + 1, kb);
        }
        __builtin_amdgcn_sched_barrier(0);
        sink(mt, acc);
    }
}
// ---- staging (bf_stage layout of vpc_bf16.h, FT = 15): operand of NT tiles whose first tile sits in slot `slot0`
template <int NT, int FT = ST_FT>
__device__ __forceinline__ void st_op(float* st, int row, int slot0, int kb, int q, Op op) {
    const u32x4 h = __builtin_bit_cast(u32x4, op);
    const int o0 = bf_stage_off<FT>(row, slot0 + 2 * kb, q);
    *reinterpret_cast<u32x2*>(st + o0) = u32x2{h[0], h[1]};
    if (2 * kb + 1 < NT) *reinterpret_cast<u32x2*>(st + o0 + 64) = u32x2{h[2], h[3]};
}
template <int FT = ST_FT>
__device__ __forceinline__ Op st_frag(const float* st, int slot, int kb, int lane) {
    const int g = lane >> 4, rr = (lane >> 2) & 3, pp = lane & 3;
    const int off = bf_stage_off<FT>(32 * kb + 4 * g + rr, slot, pp);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x4 h0 = ds_tr16(st + off), h1 = ds_tr16(st + off + 128 * FT);
    const s16x8 h = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    return __builtin_bit_cast(Op, h);
}

struct StepArgs {
    const float* x;
    const float* img;
    const uint8_t* m[2];    // encoder mask of pass p = first loss mask mA[p]
    const uint8_t* mB[2];   // optional second loss mask: mE = mA * (1 - mB)
    float cA[2], cE[2];
    const float* eps[2];
    const float* eps_ml;
    float* partE;
    float* partD;
    double* loss_part;
    float* ws;              // packed (dmean | dlogvar) seeds between the sweeps: [tiles][2][512 threads] x 16 bytes
    float bq, bp, cr, wml, inv_B, x_logvar;
    long B;
    int d, L, npass, ntiles;
    int stagger;  // start delay of workgroup group (blockIdx.x / 8) % 8, in units of 64 clocks per group (0 = none)
    int dbg;  // diagnostic build only (-DVPC_ABLATE): 64 = print the phase stamps of workgroup 100
};

#ifdef VPC_ABLATE
#define STP(i) VPC_STAMP(i)
#define LDS_BARRIER() do { if (!(a.dbg & 2)) lds_barrier(); } while (0)   // VPC_DEBUG & 2: timing without barriers (wrong results)
#else
#define STP(i) do {} while (0)
#define LDS_BARRIER() lds_barrier()
#endif

// Two sweeps over the workgroup's tiles, so that only HALF of the gradient accumulators is live at any time (all 100 of them
// beside the working set of the decoder phase do not fit 256 registers: hipcc then parks the accumulators in scratch and
// reloads them around every staging round - measured 206 us against 148 us for a variant that merely spilled less):
//   sweep 1 (decoder accumulators dW6, dW5, dW4 + dW3: 52 registers): per tile  E(q), E(p) - one x read, statistics and the
//           packed h2 of both passes kept (32 registers) - then per pass  D: reparameterise, decoder, loss, R1, R2, dz, KL seeds,
//           R3 (dW4, dW3); the packed seeds (dmean | dlogvar: 16 bytes per lane) go to a small workspace (8 MB at B = 65 536)
//   sweep 2 (encoder accumulators dW1, dW2, db1: 48 registers): per tile and pass  E again (h1, h2: the recompute costs ~60
//           bf16 MFMAs per wave), seeds back from the workspace, dh2, R4, dh1, R5.
template <int DT, bool STAGED>
__global__ __launch_bounds__(THREADS) void step_bf16_kernel(StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef VPC_ABLATE
    unsigned long long T[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
    const float* W1 = lds + StepImg::oW1;
    const float* b1 = lds + StepImg::ob1;
    const float* W2 = lds + StepImg::oW2;
    const float* W3 = lds + StepImg::oW3;
    const float* W4 = lds + StepImg::oW4;
    const float* W5 = lds + StepImg::oW5;
    const float* W6 = lds + StepImg::oW6;
    float* st = lds + StepImg::total;
    float* red = st + ST_DW;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int lrow = w * 16 + c;
    const bool two = a.npass == 2;
    constexpr int KB1 = (DT + 1) / 2;
    // Staggered start: all workgroups of a launch otherwise run in lockstep - every CU requests its 100 KB tile at the same
    // moment, then every CU computes while HBM idles.  Group g = (blockIdx.x / 8) % 8 (four CUs of every XCD) starts g * stagger
    // * 64 clocks late, which spreads the bursts of the whole launch over time.
    if (a.stagger > 0) {
        const int g = ((int)blockIdx.x >> 3) & 7;
        for (int i = 0; i < g * a.stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }

    auto mask_rsrc = [&](const uint8_t* mp, long row0) {
        const long rem = (a.B - row0) * (long)a.d;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(mp) + row0 * a.d, 0,
                                                 rem > 0xffffffffL ? 0xffffffffu : (uint32_t)rem, 0x00020000);
    };
    const bool own6 = w < DT, own2 = w < H1T, own4 = w < H2T;

    // x rows of a tile in C layout (range-checked: rows past B read 0; columns past d read column 0 and are cleared through the
    // mask word), and the packed layer-1 operand x * mask of one pass
    auto load_x = [&](long row0, f32x4 (&xr)[DT], int cc, int qq) {
        const __amdgpu_buffer_rsrc_t rx = rows_rsrc(a.x, row0, a.B, a.d);
        const int vo = (w * 16 + cc) * a.d + ((4 * qq + 3 < a.d) ? 4 * qq : 0);
#pragma unroll
        for (int t = 0; t < DT; ++t)
            xr[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rx, 4 * (vo + ((t < DT / 2 || 16 * t + 4 * qq + 3 < a.d) ? 16 * t : 0)), 0, 0));
    };
    auto load_m = [&](const uint8_t* mp, long row0, uint32_t (&mw)[DT], int cc, int qq) {
        const __amdgpu_buffer_rsrc_t rm = mask_rsrc(mp, row0);
        const int vo = (w * 16 + cc) * a.d + ((4 * qq + 3 < a.d) ? 4 * qq : 0);
#pragma unroll
        for (int t = 0; t < DT; ++t)
            mw[t] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rm, vo + ((t < DT / 2 || 16 * t + 4 * qq + 3 < a.d) ? 16 * t : 0), 0, 0);
    };
    // mask words of columns past d (possible in the last DT / 2 tiles only) -> 0: such a column is never observed
    auto clear_cols = [&](uint32_t (&mw)[DT], int qq) {
#pragma unroll
        for (int t = DT / 2; t < DT; ++t) mw[t] &= opaque_mask(16 * t + 4 * qq + 3 < a.d);
    };
    auto make_xb = [&](const f32x4 (&xr)[DT], const uint32_t (&mw)[DT], Op (&xb)[KB1], int qq) {
        f32x4 xprev = zero4();
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const f32x4 xi = xr[t] * mask_to_f32(mw[t]);  // x.float() * mask  (VAE.py:388)
            if (t & 1) xb[t >> 1] = pack2(xprev, xi);
            else if (t + 1 == DT) xb[t >> 1] = pack2(xi, zero4());
            xprev = xi;
        }
    };
    // encoder forward of one pass from its packed input: h1 / h2 as packed operands, (mean | logvar) tiles
    auto enc_fwd = [&](auto db, const Op (&xb)[KB1], Op (&h1b)[4], Op (&h2b)[2], f32x4& mu, f32x4& lv, int cc, int qq, bool ok) {
        constexpr bool DB = decltype(db)::value;
        f32x4 hprev = zero4();
        c_layer_fwd<128, KB1, H1T, DB>(W1, xb, cc, qq, [&](int mt, f32x4 acc) {
            const f32x4 h = relu4(acc + *reinterpret_cast<const f32x4*>(b1 + 16 * mt + 4 * qq));
            if (mt & 1) h1b[mt >> 1] = pack2(hprev, h);
            else if (mt + 1 == H1T) h1b[mt >> 1] = pack2(h, zero4());
            hprev = h;
        });
        c_layer_fwd<128, 4, H2T, DB>(W2, h1b, cc, qq, [&](int mt, f32x4 acc) {
            const f32x4 h = relu4(acc);
            if (mt & 1) h2b[mt >> 1] = pack2(hprev, h);
            hprev = h;
        });
        f32x4 ml[2];
        c_layer_fwd<64, 2, 2, DB>(W3, h2b, cc, qq, [&](int mt, f32x4 acc) { ml[mt] = acc; });
        const uint32_t okm = opaque_mask(ok);  // rows past B: statistics 0 (as the range-checked workspace loads gave)
        mu = and4(ml[0], okm);
        lv = and4(ml[1], okm);
    };
    // both passes of a tile at once (sweep 1: only the statistics are wanted): two MFMA chains per weight fragment
    auto enc_fwd2 = [&](const Op (&xb0)[KB1], const Op (&xb1)[KB1], f32x4& mu0, f32x4& lv0, f32x4& mu1, f32x4& lv1, int cc, int qq,
                        bool ok) {
        Op h1a[4], h1c[4], h2a[2], h2c[2];
        f32x4 pa = zero4(), pc = zero4();
        c_layer_fwd2<128, KB1, H1T>(W1, xb0, xb1, cc, qq, [&](int mt, f32x4 acc0, f32x4 acc1) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(b1 + 16 * mt + 4 * qq);
            const f32x4 ha = relu4(acc0 + bias), hc = relu4(acc1 + bias);
            if (mt & 1) { h1a[mt >> 1] = pack2(pa, ha); h1c[mt >> 1] = pack2(pc, hc); }
            else if (mt + 1 == H1T) { h1a[mt >> 1] = pack2(ha, zero4()); h1c[mt >> 1] = pack2(hc, zero4()); }
            pa = ha; pc = hc;
        });
        c_layer_fwd2<128, 4, H2T>(W2, h1a, h1c, cc, qq, [&](int mt, f32x4 acc0, f32x4 acc1) {
            const f32x4 ha = relu4(acc0), hc = relu4(acc1);
            if (mt & 1) { h2a[mt >> 1] = pack2(pa, ha); h2c[mt >> 1] = pack2(pc, hc); }
            pa = ha; pc = hc;
        });
        f32x4 ml0[2], ml1[2];
        c_layer_fwd2<64, 2, 2>(W3, h2a, h2c, cc, qq, [&](int mt, f32x4 acc0, f32x4 acc1) { ml0[mt] = acc0; ml1[mt] = acc1; });
        const uint32_t okm = opaque_mask(ok);
        mu0 = and4(ml0[0], okm); lv0 = and4(ml0[1], okm);
        mu1 = and4(ml1[0], okm); lv1 = and4(ml1[1], okm);
    };
    // workspace of the packed seeds: [tile][pass][thread] 16 bytes
    auto ws_ptr = [&](int tile, int p) { return reinterpret_cast<u32x4*>(a.ws) + ((long)tile * 2 + p) * THREADS + threadIdx.x; };
    auto ld_lat = [&](const float* base, long row0) -> f32x4 {  // [B][16] padded latent-width array; NULL reads 0
        return ld_rows(rows_rsrc(base ? base : a.x, row0, base ? a.B : row0, 16), lrow, 16, 4 * q);
    };
    // The inputs of a tile are REQUESTED one tile ahead, into the registers that hold them: x, the mask words of both passes
    // (held as (current, other) and swapped at the end of a pass, so the request made during the last pass of the previous tile
    // goes to the swapped places) and, sweep 1, eps of pass 0 / sweep 2, the packed seeds.  The first tile of sweep 1 is requested
    // before the weight image, the first tile of sweep 2 before sweep 1's partial-block stores; without this every tile started
    // with all 8 waves waiting on an HBM burst of ~100 KB per CU that every CU issues at the same time.
    f32x4 xr[DT];
    uint32_t mw0[DT], mw1[DT];
    auto request_tile = [&](int tile, bool swapped) {
        const long r0 = (long)tile * TILE_ROWS;
        load_x(r0, xr, c, q);
        if (!two) load_m(a.m[0], r0, mw0, c, q);
        else if (swapped) { load_m(a.m[0], r0, mw1, c, q); load_m(a.m[1], r0, mw0, c, q); }
        else { load_m(a.m[0], r0, mw0, c, q); load_m(a.m[1], r0, mw1, c, q); }
    };
    // ---- tile inputs THROUGH LDS (obs_dim = 128): fully coalesced global loads -> the staging area (free at the start of a
    // tile) -> the C-layout registers.  Loading straight into C layout (request_tile) makes every wave instruction touch 16
    // rows: a tile is ~3 300 cache-line lookups per CU for 100 KB, 256 of every wave's 416 for the 4 KB of mask bytes, and
    // the loads took 16-20 k cycles per tile whether they came from HBM or from cache (stamps, profiles/r03_notes.md): a fifth
    // of the kernel with all 8 waves waiting.  Coalesced, a wave moves the same bytes in 14 instructions of full lines.
    // Two COLUMN halves (features 0-63, then 64-127) of all 128 rows, 56 KB each: x 32 KB (16 granules of 16 bytes per row),
    // the two masks 8 KB each (4 granules per row), eps of the first pass 8 KB (first half only); after a barrier every wave
    // reads its rows' four tiles of that half - the same code for all waves.  Granules are XOR-swizzled by row (x: key row & 15,
    // masks: key (row >> 1) & 3) so that the C-layout reads spread over the banks (x conflict-free, masks 2-way).
    constexpr int SX = 0, SM0 = 8192, SM1 = 10240, SE0 = 12288;  // dword offsets inside the staging area
    static_assert(SE0 + 2048 <= ST_DW, "staged tile inputs");
    constexpr bool staged_in = STAGED;  // (instantiated for obs_dim == 128; other widths load straight into C layout)
    f32x4 gx[4], gm0 = zero4(), gm1 = zero4(), ge = zero4();  // one column half of a tile on its way from global memory
    auto stage_issue = [&](int tile, int hc, bool with_eps) {
        const long r0 = (long)tile * TILE_ROWS;
        const __amdgpu_buffer_rsrc_t rx = rows_rsrc(a.x, r0, a.B, 128);
        const __amdgpu_buffer_rsrc_t rm0 = mask_rsrc(a.m[0], r0), rm1 = mask_rsrc(a.m[two ? 1 : 0], r0);
#ifdef VPC_ABLATE
        if (a.dbg & 16) {  // timing without the tile loads (wrong results)
#pragma unroll
            for (int k = 0; k < 4; ++k) gx[k] = zero4();
            gm0 = zero4();
            return;
        }
#endif
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gi = (4 * w + k) * 64 + lane;  // row gi >> 4, granule gi & 15 of this column half
            gx[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (gi >> 4) * 512 + hc * 256 + (gi & 15) * 16, 0, 0));
        }
        const int gi = w * 64 + lane;  // row gi >> 2, granule gi & 3
        gm0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm0, (gi >> 2) * 128 + hc * 64 + (gi & 3) * 16, 0, 0));
        if (two) gm1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm1, (gi >> 2) * 128 + hc * 64 + (gi & 3) * 16, 0, 0));
        if (with_eps && hc == 0)
            ge = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rows_rsrc(a.eps[0], r0, a.B, 16), gi * 16, 0, 0));
    };
    // (the first half is already requested when AHEAD: by the prologue / the previous tile / the end of the other sweep)
    auto stage_in = [&](int tile, bool with_eps, f32x4& ev) {
        if (!AHEAD) stage_issue(tile, 0, with_eps);
#pragma unroll
        for (int hc = 0; hc < 2; ++hc) {
            LDS_BARRIER();  // the staging area is free: every wave is past its reads of the previous round / half
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int gi = (4 * w + k) * 64 + lane, row = gi >> 4, gc = gi & 15;
                *reinterpret_cast<f32x4*>(st + SX + 4 * (row * 16 + (gc ^ (row & 15)))) = gx[k];
            }
            {
                const int gi = w * 64 + lane, row = gi >> 2, gc = gi & 3;
                const int o = 4 * (row * 4 + (gc ^ ((row >> 1) & 3)));
                *reinterpret_cast<f32x4*>(st + SM0 + o) = gm0;
                if (two) *reinterpret_cast<f32x4*>(st + SM1 + o) = gm1;
                if (with_eps && hc == 0) *reinterpret_cast<f32x4*>(st + SE0 + 4 * gi) = ge;
            }
#ifdef VPC_ABLATE
            if (hc == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); STP(15); }
#endif
            if (hc == 0) stage_issue(tile, 1, with_eps);  // the second half's loads fly while the first half is read
            LDS_BARRIER();
            const int lr = 16 * w + c;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xr[4 * hc + t] = *reinterpret_cast<const f32x4*>(st + SX + 4 * (lr * 16 + ((4 * t + q) ^ c)));
                const int mo = 4 * (lr * 4 + (t ^ ((c >> 1) & 3))) + q;
                mw0[4 * hc + t] = __float_as_uint(st[SM0 + mo]);
                if (two) mw1[4 * hc + t] = __float_as_uint(st[SM1 + mo]);
            }
            if (with_eps && hc == 0) ev = *reinterpret_cast<const f32x4*>(st + SE0 + 4 * (lr * 4 + q));
        }
    };
    // prologue: the weight image's loads are issued first (one round: 13 x 16 bytes per thread), then the first tile's inputs;
    // the image is written to LDS as soon as ITS loads have returned (vmcnt counts in issue order: the younger tile requests
    // stay in flight)
    // One dword per 128-byte line of a tile's x / mask / eps rows, result unused: the next tile of sweep 1 is pulled into L2 / the
    // Infinity Cache under the current tile, so that its real loads are cache hits instead of a cold HBM burst of ~100 KB per CU
    // that every CU issues at the same moment (measured: ~9 k cycles per tile with all 8 waves waiting).  Unlike requests into
    // registers (request_tile ahead of time) this costs two registers and two loads per thread.
    auto touch = [&](int tl) -> uint32_t {
        const long r0 = (long)tl * TILE_ROWS;
        const int t = threadIdx.x;
        uint32_t v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rows_rsrc(a.x, r0, a.B, a.d), t * 128 < TILE_ROWS * a.d * 4 ? t * 128 : 0, 0, 0);
        const int which = t >> 7, ln = (t & 127) * 128;
        if (which < 2) {
            if (which < a.npass && ln < TILE_ROWS * a.d) v |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(mask_rsrc(a.m[which], r0), ln, 0, 0);
        } else if (which - 2 < a.npass && ln < TILE_ROWS * 64) {
            v |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rows_rsrc(a.eps[which - 2], r0, a.B, 16), ln, 0, 0);
        }
        return v;
    };
    f32x4 e = zero4();
    {
        constexpr int U = 13;
        static_assert(U * THREADS * 4 >= StepImg::total, "one round of image loads");
        f32x4 iv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = (u * THREADS + (int)threadIdx.x) * 4;
            iv[u] = *reinterpret_cast<const f32x4*>(a.img + (i < StepImg::total ? i : 0));
        }
        if (PREFETCH && (int)blockIdx.x < a.ntiles) {
            request_tile(blockIdx.x, false);
            e = ld_lat(a.eps[0], (long)blockIdx.x * TILE_ROWS);
        }
        if (staged_in && AHEAD && (int)blockIdx.x < a.ntiles) stage_issue(blockIdx.x, 0, true);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = (u * THREADS + (int)threadIdx.x) * 4;
            if (i < StepImg::total) *reinterpret_cast<f32x4*>(lds + i) = iv[u];
        }
    }
    if (!two) {
#pragma unroll
        for (int t = 0; t < DT; ++t) mw1[t] = 0u;
    }
    lds_barrier();
    STP(0);

    // ============================================================================================================ sweep 1
    {
        f32x4 acc6[H1T], acc5[H2T], acc4 = zero4();
#pragma unroll
        for (int t = 0; t < H1T; ++t) acc6[t] = zero4();
#pragma unroll
        for (int t = 0; t < H2T; ++t) acc5[t] = zero4();
        float S_A0 = 0.f, S_E0 = 0.f, S_A1 = 0.f, S_kl0q = 0.f, S_kl0p = 0.f, S_klr = 0.f, S_zll = 0.f;
        const float inv_s2 = expf(-a.x_logvar), half_lv = 0.5f * a.x_logvar;
        constexpr float HL2PI = 0.91893853320467274f;

        for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
            const long row0 = (long)tile * TILE_ROWS;
            const bool ok = row0 + lrow < a.B;
            asm volatile("" ::: "memory");  // the weight image never changes: keep LDS weight reads inside the tile
            int cc = c, qq = q;
            launder(cc, qq);
            // Everything that differs between the passes is held as a (current, other) pair and SWAPPED at the end of a pass:
            // selecting per pass (p == 0 ? q : p) makes a second live copy of whatever is selected.  x (32 registers), the mask
            // words of both passes (16) and eps (4) were requested one tile ahead (request_tile) and stay in registers for both
            // passes of this sweep.  (Re-reading them per phase from L2 does not work: 32 workgroups per XCD stream 100 KB each
            // through a 4 MB L2 beside partial blocks - measured 390 MB fetched per launch with per-phase re-reads.)
            if (staged_in) {
                stage_in(tile, true, e);
            } else if (!PREFETCH) {
                request_tile(tile, false);
                e = ld_lat(a.eps[0], row0);
            }
            clear_cols(mw0, qq);
            if (two) clear_cols(mw1, qq);
            // ---------------- E: statistics of both passes
            f32x4 muQ, lvQ, muP = zero4(), lvP = zero4();
            {
                Op xb[KB1];
                make_xb(xr, mw0, xb, qq);
#ifdef VPC_ABLATE
                asm volatile("" ::"v"(xb[0]), "v"(xb[1]), "v"(xb[2]), "v"(xb[3]));
                STP(14);
#endif
                if (two) {
                    Op xbp[KB1];
                    make_xb(xr, mw1, xbp, qq);
                    enc_fwd2(xb, xbp, muQ, lvQ, muP, lvP, cc, qq, ok);
                } else {
                    Op h1b[4], h2b[2];
                    enc_fwd(std::false_type{}, xb, h1b, h2b, muQ, lvQ, cc, qq, ok);
                }
            }
            uint32_t tv = 0;
#ifdef VPC_ABLATE
            if (!(a.dbg & 8))
#endif
            if (!PREFETCH && !(staged_in && AHEAD) && tile + (int)gridDim.x < a.ntiles) tv = touch(tile + (int)gridDim.x);
            STP(1);
            for (int p = 0; p < a.npass; ++p) {
                asm volatile("" ::: "memory");
                launder(cc, qq);
                // (current, other) = (q, p) in pass 0 and (p, q) in pass 1: see the swaps at the end of the pass
                f32x4 &mu = muQ, &lv = lvQ, &mo = muP, &lo = lvP;
                f32x4 z;
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = mu[j] + ((4 * q + j < a.L) ? e[j] : 0.f) * __expf(0.5f * lv[j]);
                // (a pass whose reconstruction terms carry no weight - cA = cE = 0: the p pass of ml_reg - still runs the decoder:
                // its seeds are exact zeros, so it adds nothing; a branch around the whole decoder phase cost 130 bytes of
                // scratch per lane in every configuration)
                f32x4 dz = zero4();
                VPC_CUT();
                {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * q + j == a.L) z[j] = 1.f;  // constant feature that drives the bias chain
                    const Op zb = pack2(z, zero4());
                    Op g2b[4];
                    const Op zin[1] = {zb};
                    // g1 = relu(W4~ z): formed here for the forward and AGAIN in front of R2 (4 MFMAs instead of 8 registers held
                    // across the output-tile phase)
                    auto make_g1 = [&](Op (&g1b)[2]) {
                        f32x4 hprev = zero4();
                        c_layer_fwd<32, 1, H2T>(W4, zin, cc, qq, [&](int mt, f32x4 acc) {
                            const f32x4 h = relu4(acc);
                            if (mt & 1) g1b[mt >> 1] = pack2(hprev, h);
                            hprev = h;
                        });
                    };
                    {
                        Op g1b[2];
                        make_g1(g1b);
                        launder(cc, qq);
                        f32x4 hprev = zero4();
                        c_layer_fwd<64, 2, H1T>(W5, g1b, cc, qq, [&](int mt, f32x4 acc) {
                            const f32x4 h = relu4(acc);
                            if (mt & 1) g2b[mt >> 1] = pack2(hprev, h);
                            else if (mt + 1 == H1T) g2b[mt >> 1] = pack2(h, zero4());
                            hprev = h;
                        });
                    }
                    launder(cc, qq);
                    STP(2);
                    VPC_CUT();
                    // ---------------- output tiles: forward, loss terms, d / d pre-activation.  dpre goes straight into its R1
                    // staging slots (tile mt -> slot mt, 8 bytes per lane and tile) instead of growing to 16 registers across
                    // the loop; the dgrad below reads the lane's own chunks back.  The barrier: every wave is past the reads of
                    // the previous staging round.
                    LDS_BARRIER();
                    {
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        constexpr float NLOG2E = -1.4426950408889634f;
                        f32x2 sa2 = {0.f, 0.f}, se2 = {0.f, 0.f};
                        const bool hasB = a.mB[p] != nullptr;  // (the host checked: the second mask IS the other pass's mask)
                        const float kA = a.cA[p] * inv_s2 * a.inv_B, kE = a.cE[p] * inv_s2 * a.inv_B, hinv_s2 = 0.5f * inv_s2;
                        Op w6f[4];
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) w6f[kb] = c_wfrag<128>(W6, 0, kb, cc, qq);
                        const int so = bf_stage_off<ST_FT>(16 * w + cc, 0, qq);  // slot mt: + 64 mt dwords
#pragma unroll
                        for (int mt = 0; mt < DT; ++mt) {
                            VPC_CUT();
                            const f32x4 xv = xr[mt];
                            const uint32_t ua = mw0[mt];
                            const uint32_t ub = hasB ? mw1[mt] : ua;
                            f32x4 pre = zero4();
#pragma unroll
                            for (int kb = 0; kb < 4; ++kb) pre = VPC_MFMA_BF(w6f[kb], g2b[kb], pre);
                            // the next tile's fragments are requested behind this tile's MFMAs and arrive under its loss math
                            if (mt + 1 < DT) {
#pragma unroll
                                for (int kb = 0; kb < 4; ++kb) w6f[kb] = c_wfrag<128>(W6, mt + 1, kb, cc, qq);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            const f32x4 mA = mask_to_f32(ua);
                            const f32x4 mE = mask_to_f32(ua & ~ub);  // mA (1 - mB) on the 0 / 1 bytes; no second mask: ub aliases ua -> 0
                            f32x4 dp4;
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const f32x2 p2 = {pre[2 * h], pre[2 * h + 1]}, x2 = {xv[2 * h], xv[2 * h + 1]};
                                const f32x2 a2 = {mA[2 * h], mA[2 * h + 1]}, e2 = {mE[2 * h], mE[2 * h + 1]};
                                const f32x2 en = p2 * NLOG2E;
                                const f32x2 den = f32x2{__builtin_amdgcn_exp2f(en[0]), __builtin_amdgcn_exp2f(en[1])} + 1.f;
                                const f32x2 xh = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
                                const f32x2 diff = xh - x2;
                                const f32x2 t = __builtin_elementwise_fma(diff * diff, f32x2{hinv_s2, hinv_s2}, f32x2{half_lv, half_lv});
                                sa2 = __builtin_elementwise_fma(a2, t, sa2);
                                se2 = __builtin_elementwise_fma(e2, t, se2);
                                const f32x2 wgt = __builtin_elementwise_fma(f32x2{kE, kE}, e2, a2 * kA);
                                const f32x2 dp = (wgt * diff) * __builtin_elementwise_fma(-xh, xh, xh);
                                dp4[2 * h] = dp[0];
                                dp4[2 * h + 1] = dp[1];
                            }
                            asm volatile("" : "+v"(dp4[0]), "+v"(dp4[1]), "+v"(dp4[2]), "+v"(dp4[3]), "+v"(sa2), "+v"(se2));
                            *reinterpret_cast<u32x2*>(st + so + 64 * mt) = u32x2{pk_bf16(dp4[0], dp4[1]), pk_bf16(dp4[2], dp4[3])};
                        }
                        const float sa = sa2[0] + sa2[1], se = se2[0] + se2[1];
                        if (p == 0) { S_A0 += sa; S_E0 += se; } else { S_A1 += sa; }
                    }
                    STP(3);
                    VPC_CUT();
                    launder(cc, qq);
                    // x and the mask words are dead after the last pass's output tiles: the next tile's are requested now
                    if (PREFETCH && p + 1 == a.npass && tile + (int)gridDim.x < a.ntiles) request_tile(tile + (int)gridDim.x, two);
                    if (staged_in && AHEAD && p + 1 == a.npass && tile + (int)gridDim.x < a.ntiles) stage_issue(tile + (int)gridDim.x, 0, true);
                    // ---------------- R1: dW6~ += dpre^T g2   (owner: wave w -> out tile w, all 7 in tiles)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) st_op<H1T>(st, lrow, 8, kb, qq, g2b[kb]);
                    LDS_BARRIER();
                    // (all fragment reads of a k-block are in flight before its first MFMA: issued one by one in front of their
                    // MFMA every product waits a full LDS round trip - the rounds were 4-6 k cycles of that)
                    if (own6) {
#pragma unroll
                        for (int kb = 0; kb < TILE_ROWS / 32; ++kb) {
                            __builtin_amdgcn_sched_barrier(0);
                            const Op fa = st_frag(st, w, kb, 16 * qq + cc);
                            Op fb[4];  // in tiles 0-3, then 4-6: two batches of reads (registers)
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt) fb[nt] = st_frag(st, 8 + nt, kb, 16 * qq + cc);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt) acc6[nt] = VPC_MFMA_BF(fa, fb[nt], acc6[nt]);
#pragma unroll
                            for (int nt = 4; nt < H1T; ++nt) fb[nt - 4] = st_frag(st, 8 + nt, kb, 16 * qq + cc);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int nt = 4; nt < H1T; ++nt) acc6[nt] = VPC_MFMA_BF(fa, fb[nt - 4], acc6[nt]);
                        }
                    }
                    STP(4);
                    VPC_CUT();
                    launder(cc, qq);
                    // ---------------- dg2 = relu'(g2) * (W6~^T dpre): B operands = the lane's own dpre chunks, back from their
                    // staging slots (own writes: no barrier)
                    Op dpreb[KB1], g2r[4];  // g2r: the lane's own packed g2 (slots 8-14), the ReLU gates of this dgrad
                    {
                        const int so = bf_stage_off<ST_FT>(16 * w + cc, 0, qq);
#pragma unroll
                        for (int kb = 0; kb < KB1; ++kb) {
                            const u32x2 lo2 = *reinterpret_cast<const u32x2*>(st + so + 128 * kb);
                            const u32x2 hi2 = (2 * kb + 1 < DT) ? *reinterpret_cast<const u32x2*>(st + so + 128 * kb + 64) : u32x2{0u, 0u};
                            dpreb[kb] = __builtin_bit_cast(Op, u32x4{lo2[0], lo2[1], hi2[0], hi2[1]});
                        }
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) {
                            const u32x2 lo2 = *reinterpret_cast<const u32x2*>(st + so + 64 * 8 + 128 * kb);
                            const u32x2 hi2 = (2 * kb + 1 < H1T) ? *reinterpret_cast<const u32x2*>(st + so + 64 * 8 + 128 * kb + 64) : u32x2{0u, 0u};
                            g2r[kb] = __builtin_bit_cast(Op, u32x4{lo2[0], lo2[1], hi2[0], hi2[1]});
                        }
                    }
                    Op dg2b[4];
                    {
                        f32x4 hprev = zero4();
                        c_layer_T<128, KB1, H1T, DT>(W6, dpreb, 16 * qq + cc, [&](int mt, f32x4 acc) {
                            const BfOp gp = {g2r[mt >> 1], g2r[mt >> 1]};
                            const f32x4 h = bf_gate(acc, gp, mt & 1);
                            if (mt & 1) dg2b[mt >> 1] = pack2(hprev, h);
                            else if (mt + 1 == H1T) dg2b[mt >> 1] = pack2(h, zero4());
                            hprev = h;
                        });
                    }
                    STP(5);
                    VPC_CUT();
                    launder(cc, qq);
                    // ---------------- R2: dW5~ += dg2^T g1   (owner: wave w -> in tile w & 3 of out tiles 4 (w >> 2) .. + 3)
                    Op g1b[2];  // (also the ReLU gate of dg1 below: a packed relu output is non-zero where the unit is active)
                    make_g1(g1b);
                    LDS_BARRIER();
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) st_op<H1T>(st, lrow, 0, kb, qq, dg2b[kb]);
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) st_op<H2T>(st, lrow, 8, kb, qq, g1b[kb]);
                    LDS_BARRIER();
                    {
                        const int nt5 = w & 3, mt5 = 4 * (w >> 2);
#pragma unroll
                        for (int kb = 0; kb < TILE_ROWS / 32; ++kb) {
                            __builtin_amdgcn_sched_barrier(0);
                            const Op fb = st_frag(st, 8 + nt5, kb, 16 * qq + cc);
                            Op fa[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) fa[i] = st_frag(st, (i < 3 || w < 4) ? mt5 + i : mt5, kb, 16 * qq + cc);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < 3 || w < 4) acc5[i] = VPC_MFMA_BF(fa[i], fb, acc5[i]);
                        }
                    }
                    STP(6);
                    VPC_CUT();
                    launder(cc, qq);
                    // ---------------- dg1 = relu'(g1) * (W5~^T dg2);  dz = W4~^T dg1
                    Op dg1b[2];
                    {
                        f32x4 hprev = zero4();
                        c_layer_T<64, 4, H2T, H1T>(W5, dg2b, 16 * qq + cc, [&](int mt, f32x4 acc) {
                            const BfOp gp = {g1b[mt >> 1], g1b[mt >> 1]};
                            const f32x4 h = bf_gate(acc, gp, mt & 1);
                            if (mt & 1) dg1b[mt >> 1] = pack2(hprev, h);
                            hprev = h;
                        });
                        c_layer_T<32, 2, 1, H2T>(W4, dg1b, 16 * qq + cc, [&](int, f32x4 acc) { dz = acc; });
                    }
                    // R3's decoder operands (the seeds and h2 follow below)
                    LDS_BARRIER();
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) st_op<H2T>(st, lrow, 0, kb, qq, dg1b[kb]);
                    st_op<1>(st, lrow, 8, 0, qq, zb);
                }
                STP(7);
                // ---------------- KL terms, their seeds, total seeds on (mean | logvar)
                {
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
                        const f32x4 e3 = ld_lat(a.eps_ml, row0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool live = ok && 4 * q + j < a.L;
                            const float e3j = (4 * q + j < a.L) ? e3[j] : 0.f;
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
                        const float ef = (4 * q + j < a.L) ? e[j] * 0.5f * __expf(0.5f * lv[j]) : 0.f;
                        dmu[j] += dz[j];
                        dlv[j] += dz[j] * ef;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {  // columns >= L of the latent tiles carry no gradient (dz's column L is db4)
                        dmu[j] = (4 * q + j < a.L) ? dmu[j] : 0.f;
                        dlv[j] = (4 * q + j < a.L) ? dlv[j] : 0.f;
                    }
                    *ws_ptr(tile, p) = __builtin_bit_cast(u32x4, pack2(dmu, dlv));  // sweep 2 continues from here
                }
                // ---------------- R3: dW4~ += dg1^T z (owner: wave w < 4 -> out tile w)
                launder(cc, qq);
                // the next pass's (or the next tile's first pass's) eps arrives under this round
                if (two && p == 0) e = ld_lat(a.eps[1], row0);  // (a 4-line read per wave; the staged copy is gone by now)
                else if (PREFETCH && tile + (int)gridDim.x < a.ntiles) e = ld_lat(a.eps[0], row0 + (long)gridDim.x * TILE_ROWS);
                LDS_BARRIER();
                if (own4) {
                    Op fa4[TILE_ROWS / 32], fb4[TILE_ROWS / 32];
#pragma unroll
                    for (int kb = 0; kb < TILE_ROWS / 32; ++kb) {
                        fa4[kb] = st_frag(st, w, kb, 16 * qq + cc);
                        fb4[kb] = st_frag(st, 8, kb, 16 * qq + cc);
                    }
#pragma unroll
                    for (int kb = 0; kb < TILE_ROWS / 32; ++kb) acc4 = VPC_MFMA_BF(fa4[kb], fb4[kb], acc4);
                }
                if (p + 1 == a.npass) asm volatile("" ::"v"(tv));  // the touch loads retire here, long after they were issued
                if (two) {  // (current, other) <- (other, current)
                    f32x4 t4;
                    t4 = muQ; muQ = muP; muP = t4;
                    t4 = lvQ; lvQ = lvP; lvP = t4;
#pragma unroll
                    for (int t = 0; t < DT; ++t) { const uint32_t u = mw0[t]; mw0[t] = mw1[t]; mw1[t] = u; }
                }
                STP(8);
            }
        }
        // sweep 2's first tile is requested before this sweep's partial-block stores
        if (PREFETCH && (int)blockIdx.x < a.ntiles) request_tile(blockIdx.x, false);
        if (staged_in && AHEAD && (int)blockIdx.x < a.ntiles) stage_issue(blockIdx.x, 0, false);
        // ---------------- decoder partial block, dW3 and the loss terms
        {
            float* part = a.partD + (long)blockIdx.x * DEC_PART + (long)(w & 3) * DEC_GREGS * 64 + lane;
            const int hi = w >> 2;
#pragma unroll
            for (int nt = 0; nt < H1T; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) part[(28 * hi + 4 * nt + j) * 64] = own6 ? acc6[nt][j] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* p5 = a.partD + (long)blockIdx.x * DEC_PART + (long)i * DEC_GREGS * 64 + lane;
#pragma unroll
                for (int j = 0; j < 4; ++j) p5[(56 + 16 * hi + 4 * (w & 3) + j) * 64] = (i < 3 || w < 4) ? acc5[i][j] : 0.f;
            }
            if (own4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) part[(88 + j) * 64] = acc4[j];
            }
        }
        const float s[LOSS_TERMS] = {S_A0, S_E0, S_A1, S_kl0q, S_kl0p, S_klr, S_zll, 0.f};
#pragma unroll
        for (int i = 0; i < LOSS_TERMS; ++i) {
            const float v = wave_sum_dpp(s[i]);
            if (lane == 0) red[w * LOSS_TERMS + i] = v;
        }
        LDS_BARRIER();
        if (threadIdx.x < LOSS_TERMS) {
            double t = 0.0;
            for (int k = 0; k < WAVES; ++k) t += (double)red[k * LOSS_TERMS + threadIdx.x];
            a.loss_part[(long)blockIdx.x * LOSS_TERMS + threadIdx.x] = t;
        }
    }
    STP(9);
    // ============================================================================================================ sweep 2
    {
        f32x4 acc1[H1T], acc2[H2T], accb = zero4(), acc3 = zero4();
        float* st2 = lds + StepImg::oW4;  // (every wave is past sweep 1's last read of W4 .. W6: the barrier of the loss reduction)
#pragma unroll
        for (int t = 0; t < H1T; ++t) acc1[t] = zero4();
#pragma unroll
        for (int t = 0; t < H2T; ++t) acc2[t] = zero4();
        const u32x4 ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
        const Op ones = __builtin_bit_cast(Op, ones_u);
        u32x4 s0 = {0u, 0u, 0u, 0u}, s1 = {0u, 0u, 0u, 0u};
        if (PREFETCH && (int)blockIdx.x < a.ntiles) {
            s0 = *ws_ptr(blockIdx.x, 0);
            if (two) s1 = *ws_ptr(blockIdx.x, 1);
        }
        for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
            const long row0 = (long)tile * TILE_ROWS;
            const bool ok = row0 + lrow < a.B;
            asm volatile("" ::: "memory");
            int cc = c, qq = q;
            launder(cc, qq);
            if (staged_in || !PREFETCH) {
                s0 = *ws_ptr(tile, 0);
                if (two) s1 = *ws_ptr(tile, 1);
            }
            if (staged_in) {
                f32x4 unused;
                stage_in(tile, false, unused);
            } else if (!PREFETCH) {
                request_tile(tile, false);
            }
            // x and the mask words were requested one tile ahead (request_tile); the packed seeds of both passes come back from
            // the workspace (sweep 1 stored them from this very thread: same address, same lane)
            clear_cols(mw0, qq);
            if (two) clear_cols(mw1, qq);
            for (int p = 0; p < a.npass; ++p) {
                asm volatile("" ::: "memory");
                launder(cc, qq);
                const Op dmlb = __builtin_bit_cast(Op, s0);  // (current, other): swapped at the end of the pass
                uint32_t (&mw)[DT] = mw0;
                Op xb[KB1], h1b[4], h2b[2];
                {
                    f32x4 mu, lv;
                    make_xb(xr, mw, xb, qq);
                    enc_fwd(std::true_type{}, xb, h1b, h2b, mu, lv, cc, qq, ok);
                }
                // x, the mask words and the seed registers are dead from here in the last pass (xb is kept for R5): the next
                // tile's inputs are requested now and arrive under the two staging rounds
                if (staged_in && AHEAD && p + 1 == a.npass && tile + (int)gridDim.x < a.ntiles) stage_issue(tile + (int)gridDim.x, 0, false);
                if (PREFETCH && p + 1 == a.npass && tile + (int)gridDim.x < a.ntiles) {
                    const int nt = tile + (int)gridDim.x;
                    request_tile(nt, two);
                    if (two) { s0 = *ws_ptr(nt, 1); s1 = *ws_ptr(nt, 0); } else { s0 = *ws_ptr(nt, 0); }
                }
                STP(10);
                launder(cc, qq);
                // ---------------- dh2 = relu'(h2) * (W3~^T dml)
                Op dh2b[2];
                {
                    f32x4 hprev = zero4();
                    const Op din[1] = {dmlb};
                    c_layer_T<64, 1, H2T, 2>(W3, din, 16 * qq + cc, [&](int mt, f32x4 acc) {
                        const BfOp hp = {h2b[mt >> 1], h2b[mt >> 1]};
                        const f32x4 h = bf_gate(acc, hp, mt & 1);
                        if (mt & 1) dh2b[mt >> 1] = pack2(hprev, h);
                        hprev = h;
                    });
                }
                // ---------------- R4: dW2~ += dh2^T h1   (owner: wave w < 7 -> in tile w, all 4 out tiles);  dW3~ += dml^T h2 (wave w ->
                // out tile w >> 2, in tile w & 3) from the second staging area
                LDS_BARRIER();
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) st_op<H2T>(st, lrow, 0, kb, qq, dh2b[kb]);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) st_op<H1T>(st, lrow, 8, kb, qq, h1b[kb]);
                st_op<2, ST2_FT>(st2, lrow, 0, 0, qq, dmlb);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) st_op<H2T, ST2_FT>(st2, lrow, 2, kb, qq, h2b[kb]);
                LDS_BARRIER();
                {
                    Op fa3[TILE_ROWS / 32], fb3[TILE_ROWS / 32];
#pragma unroll
                    for (int kb = 0; kb < TILE_ROWS / 32; ++kb) {
                        fa3[kb] = st_frag<ST2_FT>(st2, w >> 2, kb, 16 * qq + cc);
                        fb3[kb] = st_frag<ST2_FT>(st2, 2 + (w & 3), kb, 16 * qq + cc);
                    }
#pragma unroll
                    for (int kb = 0; kb < TILE_ROWS / 32; ++kb) acc3 = VPC_MFMA_BF(fa3[kb], fb3[kb], acc3);
                }
                if (own2) {
#pragma unroll
                    for (int kb = 0; kb < TILE_ROWS / 32; ++kb) {
                        __builtin_amdgcn_sched_barrier(0);
                        const Op fb = st_frag(st, 8 + w, kb, 16 * qq + cc);
                        Op fa[H2T];
#pragma unroll
                        for (int mt = 0; mt < H2T; ++mt) fa[mt] = st_frag(st, mt, kb, 16 * qq + cc);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int mt = 0; mt < H2T; ++mt) acc2[mt] = VPC_MFMA_BF(fa[mt], fb, acc2[mt]);
                    }
                }
                STP(11);
                VPC_CUT();
                launder(cc, qq);
                // ---------------- dh1 = relu'(h1) * (W2~^T dh2)
                Op dh1b[4];
                {
                    f32x4 hprev = zero4();
                    c_layer_T<128, 2, H1T, H2T>(W2, dh2b, 16 * qq + cc, [&](int mt, f32x4 acc) {
                        const BfOp hp = {h1b[mt >> 1], h1b[mt >> 1]};
                        const f32x4 h = bf_gate(acc, hp, mt & 1);
                        if (mt & 1) dh1b[mt >> 1] = pack2(hprev, h);
                        else if (mt + 1 == H1T) dh1b[mt >> 1] = pack2(h, zero4());
                        hprev = h;
                    });
                }
                // ---------------- R5: dW1 += dh1^T (x * mask)  (owner: wave w -> in tile w, all 7 out tiles);  db1 += dh1^T 1
                // (wave w < 7 -> out tile w)
                LDS_BARRIER();
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) st_op<H1T>(st, lrow, 0, kb, qq, dh1b[kb]);
#pragma unroll
                for (int kb = 0; kb < KB1; ++kb) st_op<DT>(st, lrow, 7, kb, qq, xb[kb]);
                LDS_BARRIER();
#pragma unroll
                for (int kb = 0; kb < TILE_ROWS / 32; ++kb) {
                    __builtin_amdgcn_sched_barrier(0);
                    const Op fb = st_frag(st, 7 + w, kb, 16 * qq + cc);
                    const Op fw = st_frag(st, own2 ? w : 0, kb, 16 * qq + cc);  // dh1 tile w once more, for db1
                    Op fa[H1T];
#pragma unroll
                    for (int mt = 0; mt < H1T; ++mt) fa[mt] = st_frag(st, mt, kb, 16 * qq + cc);
                    __builtin_amdgcn_sched_barrier(0);
                    if (own2) accb = VPC_MFMA_BF(fw, ones, accb);
#pragma unroll
                    for (int mt = 0; mt < H1T; ++mt) acc1[mt] = VPC_MFMA_BF(fa[mt], fb, acc1[mt]);
                }
                if (two) {
                    const u32x4 t4 = s0; s0 = s1; s1 = t4;
#pragma unroll
                    for (int t = 0; t < DT; ++t) { const uint32_t u = mw0[t]; mw0[t] = mw1[t]; mw1[t] = u; }
                }
                STP(12);
            }
        }
        float* part = a.partE + (long)blockIdx.x * ENC_PART + (long)w * GREGS * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < H1T; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(4 * mt + j) * 64] = own6 ? acc1[mt][j] : 0.f;
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(28 + 4 * mt + j) * 64] = own2 ? acc2[mt][j] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) part[(44 + j) * 64] = acc3[j];
        // db1[16 w + 4 q + j]: every column of the ones product holds the sum; lane c == 0 writes it
        if (c == 0 && own2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a.partE[(long)blockIdx.x * ENC_PART + WAVES * GREGS * 64 + 16 * w + 4 * q + j] = accb[j];
        }
        if (w == 7 && lane < 16) a.partE[(long)blockIdx.x * ENC_PART + WAVES * GREGS * 64 + 112 + lane] = 0.f;
    }
#ifdef VPC_ABLATE
    STP(13);
    if ((a.dbg & 64) && blockIdx.x == 100 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) % 3 == 0)
        printf("step blk %d wave %d: half0 arrived %llu xwait(rest) %llu | prologue %llu E1 %llu g1g2 %llu out %llu R1 %llu dg2 %llu R2 %llu dg1+dz %llu KL+R3 %llu epi1 %llu | E2 %llu dh2+R4 %llu dh1+R5 %llu epi2 %llu\n",
               blockIdx.x, (int)(threadIdx.x >> 6), T[15], T[14], T[0], T[1], T[2], T[3], T[4], T[5], T[6], T[7], T[8], T[9], T[10], T[11], T[12], T[13]);
#endif
}

static inline int row3c(int o, int L) { return o < L ? o : 16 + (o - L); }

}  // namespace vpc

using namespace vpc;

// 1 when vpc_step_fused_bf16 is the form the library runs for this shape: obs_dim in (64, 128], obs_dim % 4 == 0, any batch (the
// caller routes the smallest batches to vpc_step_small_f32 first).  Mid-size batches too: one 128-row tile per workgroup in ONE launch
// beats the three small-shape launches (B = 8 192: 55 us against 69, B = 16 384: 58 against 97; profiles/r03_notes.md).
// VPC_TILE=64 or VPC_STEP_FUSED=0 in the environment keep the three-kernel form (A/B runs, tests of those kernels).
extern "C" int vpc_step_fused_applicable(long B, int d, int L, int npass) {
    if (d <= 64 || d > MAX_D || d % 4 || L < 1 || L > MAX_L || npass < 1 || npass > 2 || B <= 0) return 0;
    if (const char* e = getenv("VPC_STEP_FUSED")) {
        if (atoi(e) == 0) return 0;
    }
    if (const char* e = getenv("VPC_TILE")) {
        if (atoi(e) == 64) return 0;
    }
    return 1;
}

// floats of the caller-owned workspace vpc_step_fused_bf16 needs for a batch of B rows (the packed seeds between its sweeps)
extern "C" long vpc_step_workspace_floats(long B) {
    return B <= 0 ? 0 : ((B + TILE_ROWS - 1) / TILE_ROWS) * 2 * THREADS * 4;
}

extern "C" int vpc_step_layout_bf16(int d, int L, int* img_floats, int* lds_bytes) {
    if (d <= 64 || d > MAX_D || d % 4 || L < 1 || L > MAX_L) return VPC_ERR_SHAPE;
    if (img_floats) *img_floats = StepImg::total;
    if (lds_bytes) *lds_bytes = STEP_LDS;
    return VPC_OK;
}

// pack_idx_c[i] >= 0: u16 index of flat parameter i inside the compact image; < 0: dword index -(idx + 1) of a value that stays
// fp32 (the layer-1 bias).  img_template_c: zeros + the constant ones of the bias chain.
extern "C" int vpc_step_build_indices_bf16(int d, int L, int* pack_idx_c, float* img_template_c) {
    if (d <= 64 || d > MAX_D || d % 4 || L < 1 || L > MAX_L) return VPC_ERR_SHAPE;
    if (!pack_idx_c || !img_template_c) return VPC_ERR_ARG;
    const ParamOffsets po(d, L, d);
    std::memset(img_template_c, 0, sizeof(float) * (size_t)StepImg::total);
    unsigned short* u = reinterpret_cast<unsigned short*>(img_template_c);
    const unsigned short ONE = 0x3F80;
    for (int o = 0; o < H1; ++o) {
        const int r = pos1(o);
        for (int i = 0; i < d; ++i) pack_idx_c[po.w1 + o * d + i] = 2 * StepImg::oW1 + c_elem<128>(r, i);
        pack_idx_c[po.b1 + o] = -(StepImg::ob1 + r + 1);
    }
    img_template_c[StepImg::ob1 + pos1(H1)] = 1.f;
    for (int o = 0; o < H2; ++o) {
        const int r = pos2(o);
        for (int i = 0; i <= H1; ++i)
            pack_idx_c[(i < H1) ? po.w2 + o * H1 + i : po.b2 + o] = 2 * StepImg::oW2 + c_elem<128>(r, pos1(i));
    }
    u[2 * StepImg::oW2 + c_elem<128>(pos2(H2), pos1(H1))] = ONE;
    for (int o = 0; o < 2 * L; ++o) {
        const int pr = row3c(o, L);
        for (int i = 0; i <= H2; ++i)
            pack_idx_c[(i < H2) ? po.w3 + o * H2 + i : po.b3 + o] = 2 * StepImg::oW3 + c_elem<64>(pr, pos2(i));
    }
    for (int o = 0; o < H2; ++o) {
        const int r = pos2(o);
        for (int i = 0; i <= L; ++i) pack_idx_c[(i < L) ? po.w4 + o * L + i : po.b4 + o] = 2 * StepImg::oW4 + c_elem<32>(r, i);
    }
    u[2 * StepImg::oW4 + c_elem<32>(pos2(H2), L)] = ONE;
    for (int o = 0; o < H1; ++o) {
        const int r = pos1(o);
        for (int i = 0; i <= H2; ++i)
            pack_idx_c[(i < H2) ? po.w5 + o * H2 + i : po.b5 + o] = 2 * StepImg::oW5 + c_elem<64>(r, pos2(i));
    }
    u[2 * StepImg::oW5 + c_elem<64>(pos1(H1), pos2(H2))] = ONE;
    for (int o = 0; o < d; ++o)
        for (int i = 0; i <= H1; ++i)
            pack_idx_c[(i < H1) ? po.w6 + o * H1 + i : po.b6 + o] = 2 * StepImg::oW6 + c_elem<128>(o, pos1(i));
    return VPC_OK;
}

namespace vpc {
__global__ void pack_bf16c_kernel(const float* __restrict__ flat, const int* __restrict__ idx, float* __restrict__ img, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = flat[i];
    const int e = idx[i];
    if (e == INT_MIN) return;  // a parameter that is not part of this image (vpc_nmdec_build_indices: the encoder's)
    if (e < 0) { img[-(e + 1)] = v; return; }
    reinterpret_cast<unsigned short*>(img)[e] = (unsigned short)(pk_bf16(v, 0.f) & 0xffffu);
}
}  // namespace vpc

extern "C" int vpc_step_pack_weights_bf16(const float* flat_params, const int* pack_idx_c, float* img_c, int n, void* stream) {
    if (!flat_params || !pack_idx_c || !img_c || n <= 0) return VPC_ERR_ARG;
    hipLaunchKernelGGL(pack_bf16c_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, flat_params, pack_idx_c,
                       img_c, n);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_step_fused_bf16(const float* x, const float* img_c, int npass, const uint8_t* const* mask,
                                   const uint8_t* const* maskB, const float* cA, const float* cE, const float* const* eps,
                                   const float* eps_ml, float bq, float bp, float cr, float wml, float inv_B, float x_logvar,
                                   float* partE, float* partD, double* loss_partials, float* workspace, int* nblocks_out,
                                   long B, int d, int L, void* stream) {
    if (!x || !img_c || !mask || !cA || !cE || !eps || !partE || !partD || !loss_partials || !workspace) return VPC_ERR_ARG;
    if (!aligned16(workspace)) return VPC_ERR_ARG;
    if (npass < 1 || npass > 2 || B <= 0) return VPC_ERR_ARG;
    if (d <= 64 || d > MAX_D || d % 4 || L < 1 || L > MAX_L) return VPC_ERR_SHAPE;
    if (!aligned16(x) || !aligned16(img_c)) return VPC_ERR_ARG;
    if (wml != 0.f && !eps_ml) return VPC_ERR_ARG;
    StepArgs a{};
    a.x = x; a.img = img_c; a.eps_ml = eps_ml; a.partE = partE; a.partD = partD; a.loss_part = loss_partials; a.ws = workspace;
    a.bq = bq; a.bp = bp; a.cr = cr; a.wml = wml; a.inv_B = inv_B; a.x_logvar = x_logvar;
    a.B = B; a.d = d; a.L = L; a.npass = npass;
    for (int p = 0; p < npass; ++p) {
        if (!mask[p] || !eps[p]) return VPC_ERR_ARG;
        a.m[p] = mask[p]; a.mB[p] = maskB ? maskB[p] : nullptr; a.cA[p] = cA[p]; a.cE[p] = cE[p]; a.eps[p] = eps[p];
        if ((uintptr_t)a.m[p] % 4 || !aligned16(a.eps[p])) return VPC_ERR_ARG;
    }
    // the second loss mask of a pass (mE = mA (1 - mB)) must be the OTHER pass's mask - what the consistency term
    // NLL(mask & ~mask_p) of src/models/VAE.py:444-446 needs: the kernel keeps both passes' mask words in registers and reads no third
    for (int p = 0; p < npass; ++p)
        if (a.mB[p] && (npass != 2 || a.mB[p] != a.m[1 - p])) return VPC_ERR_ARG;
    a.ntiles = (int)((B + TILE_ROWS - 1) / TILE_ROWS);
    if (const char* e = getenv("VPC_STEP_STAGGER")) a.stagger = atoi(e);
#ifdef VPC_ABLATE
    if (const char* e = getenv("VPC_DEBUG")) a.dbg = atoi(e);
#endif
    const int grid = a.ntiles < num_cus() ? a.ntiles : num_cus();
    if (nblocks_out) *nblocks_out = grid;
    auto kern = (d == 128 && !PREFETCH) ? step_bf16_kernel<8, true> : step_bf16_kernel<8, false>;
    if (!lds_attr_done(reinterpret_cast<const void*>(kern), STEP_LDS)) return VPC_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), STEP_LDS, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}
