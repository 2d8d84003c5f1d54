// bf16-input MFMA engine for the register-chained MLP kernels (gfx950: v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// Two precisions behind one code path (PREC template parameter of the kernels):
//   PREC_BF16X3  "split bf16": every fp32 operand v is carried as hi = bf16(v), lo = bf16(v - hi) and a product is
//                hi*hi + hi*lo + lo*hi (three MFMAs into one fp32 accumulator; the dropped lo*lo term is ~2^-18
//                relative).  ~fp32 accuracy (the 1e-4 loss tolerance holds, tests/test_bf16.py) at 16/3 of the fp32 MFMA rate.
//   PREC_BF16    plain bf16 inputs (one MFMA per product, 16 x the fp32 MFMA rate), fp32 accumulation, fp32 loss math.
//
// How the register chaining survives the wider K.  One 16x16x32 MFMA contracts 32 input features.  Its B operand wants
// lane (c = l & 15, q = l >> 4) to hold the 8 features of k-slots (q, j), j = 0..7.  A wave holds an activation tile in
// the MFMA C/D layout: lane (c, q) has features 16 t + 4 q + r (r = 0..3) of tile t for batch row c.  Two adjacent tiles
// (2 kb, 2 kb + 1) therefore ARE one B operand if k-slot (q, j) is taken to mean
//        feature(kb, q, j) = 32 kb + 16 (j >> 2) + 4 q + (j & 3)
// - no lane movement, only four v_cvt_pk_bf16_f32 per operand - and the weight image is simply stored with its columns
// in that order (host-side index table, vpc_build_indices_bf16).
//
// bf16 weight image of one layer: `rows` rows of KP dwords (KP = inputs padded to 32).  The 32-byte "pair slot"
// pi = 4 kb + q of a row holds [hi: 8 x bf16 | lo: 8 x bf16] for k-slots (kb, q, 0..7); pi is XOR-swizzled with
// row & min(15, KP / 8 - 1).  The image has the row pitch (in dwords) and size of the fp32 image with S = KP.
//   forward  A fragment (row 16 mt + m, block kb, lane group q): ds_read_b128 (hi) + ds_read_b128 (lo)
//   dgrad    A fragment of W^T (in feature 16 mt + m, block kb of OUT features, lane group q): the same image read with
//            ds_read_b64_tr_b16 (hardware transpose of a 4-row x 16-column block of 16-bit elements): 2 reads for hi, 2 for lo
//   wgrad    contracts over batch rows: the operands are staged in LDS as bf16, row-major [batch row][feature], by the
//            owner of the row (one conversion per value - the packed operands of the MFMA chain are written as they are),
//            and read back transposed with ds_read_b64_tr_b16 (bf_stage_* below); the 4-wave decoder kernel still
//            converts fp32 staging fragments in registers
#pragma once
#include "vpc_device.h"

namespace vpc {

enum { PREC_F32 = 0, PREC_BF16X3 = 1, PREC_BF16 = 2 };

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define VPC_MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// ---- host + device: geometry of the bf16 image of a layer with `k_in` inputs
VPC_HD constexpr int bf_kp(int k_in) { return (k_in + 31) / 32 * 32; }          // row pitch in dwords
VPC_HD constexpr int bf_mask(int kp) { return (kp / 8 - 1) < 15 ? (kp / 8 - 1) : 15; }
// u16 index of the hi element of (row, input feature f) inside a layer image with pitch kp dwords; lo is 8 u16 later
VPC_HD inline int bf_elem(int row, int f, int kp) {
    const int kb = f >> 5, h = (f >> 4) & 1, q = (f >> 2) & 3, i = f & 3;
    const int pi = (4 * kb + q) ^ (row & bf_mask(kp));
    return row * kp * 2 + pi * 16 + 4 * h + i;
}

#if defined(__HIPCC__)
// ---- fp32 -> bf16 pairs (round to nearest even: v_cvt_pk_bf16_f32) and the split residual
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t pk_bf16_lo(float a, float b, uint32_t hi) {
    const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xffff0000u);
    return pk_bf16(a - ah, b - bh);
}
struct BfOp {  // one MFMA operand (8 k-slots per lane): hi part and, for the split form, the residual
    bf16x8 hi, lo;
};
template <int PREC>
__device__ __forceinline__ BfOp bf_pack(f32x4 t0, f32x4 t1) {
    u32x4 h = {pk_bf16(t0[0], t0[1]), pk_bf16(t0[2], t0[3]), pk_bf16(t1[0], t1[1]), pk_bf16(t1[2], t1[3])};
    BfOp o;
    o.hi = __builtin_bit_cast(bf16x8, h);
    if (PREC == PREC_BF16X3) {
        u32x4 l = {pk_bf16_lo(t0[0], t0[1], h[0]), pk_bf16_lo(t0[2], t0[3], h[1]), pk_bf16_lo(t1[0], t1[1], h[2]),
                   pk_bf16_lo(t1[2], t1[3], h[3])};
        o.lo = __builtin_bit_cast(bf16x8, l);
    } else {
        o.lo = o.hi;
    }
    return o;
}
// acc += A * B in the chosen precision
template <int PREC>
__device__ __forceinline__ f32x4 bf_mma(const BfOp& a, const BfOp& b, f32x4 acc) {
    if (PREC == PREC_BF16X3) {
        acc = VPC_MFMA_BF(a.lo, b.hi, acc);
        acc = VPC_MFMA_BF(a.hi, b.lo, acc);
    }
    return VPC_MFMA_BF(a.hi, b.hi, acc);
}

// ReLU backward from a PACKED activation operand: element j of tile `second` (0: tile 2 kb, 1: tile 2 kb + 1) passes where the
// stored bf16 is not zero (a relu output is >= +0, and bf16 rounding keeps a positive normal positive)
__device__ __forceinline__ f32x4 bf_gate(f32x4 dy, const BfOp& act, int second) {
    const u32x4 h = __builtin_bit_cast(u32x4, act.hi);
    const uint32_t w0 = second ? h[2] : h[0], w1 = second ? h[3] : h[1];
    return f32x4{(w0 & 0xffffu) ? dy[0] : 0.f, (w0 >> 16) ? dy[1] : 0.f, (w1 & 0xffffu) ? dy[2] : 0.f, (w1 >> 16) ? dy[3] : 0.f};
}
// activations of a layer as MFMA B operands: KT fp32 tiles -> (KT + 1) / 2 blocks of 32 features
template <int PREC, int KT>
__device__ __forceinline__ void bf_acts(const f32x4 (&in)[KT], BfOp (&out)[(KT + 1) / 2]) {
#pragma unroll
    for (int kb = 0; kb < (KT + 1) / 2; ++kb)
        out[kb] = bf_pack<PREC>(in[2 * kb], 2 * kb + 1 < KT ? in[2 * kb + 1] : zero4());
}

// ---- forward A fragment of weight rows 16 mt + m, block kb
template <int PREC, int KP>
__device__ __forceinline__ BfOp bf_wfrag(const float* W, int mt, int kb, int m, int q) {
    constexpr int MASK = bf_mask(KP);
    const float* p = W + (16 * mt + m) * KP + 8 * ((4 * kb + q) ^ (m & MASK));
    BfOp o;
    o.hi = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(p));
    if (PREC == PREC_BF16X3) o.lo = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(p + 4));
    else o.lo = o.hi;
    return o;
}
// out tile mt of  W[out][in] * in  (in: KB blocks; NKB of them hold data)
template <int PREC, int KB, int KP, int NKB = KB>
__device__ __forceinline__ f32x4 bf_tile_fwd(const float* W, int mt, const BfOp (&in)[KB], f32x4 acc, int m, int q) {
    BfOp a[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) a[kb] = bf_wfrag<PREC, KP>(W, mt, kb, m, q);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) acc = bf_mma<PREC>(a[kb], in[kb], acc);
    return acc;
}

// ---- transposed A fragment: in-feature tile mt (lane m = in feature 16 mt + m), block kb of the layer's OUT features.
// ds_read_b64_tr_b16: lane 4 rr + pp of 16-lane group g supplies the address of block row rr, columns 4 pp .. 4 pp + 3;
// lane i of the group receives column i of the 4 rows (row rr in element rr).  Group g = q reads rows 32 kb + 4 q + rr
// (k-slots j = rr) and, second read, 16 rows further (j = 4 + rr); its 16 columns are the in features of tile mt, which
// sit in pair slot 4 (mt >> 1) + pp at element offset 4 (mt & 1) of each row.
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;
__device__ __forceinline__ s16x4 ds_tr16(const float* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(const_cast<float*>(p)));
}
// SECOND = false: the image has no rows for the block's second 16-row half (odd number of 16-row tiles: W6 at d <= 16, W5's
// 7 tiles) - those k-slots are zero instead of whatever lies behind the image (uninitialised LDS may hold NaN patterns,
// and 0 x NaN is not 0).
template <int PREC, int KP, bool SECOND = true>
__device__ __forceinline__ BfOp bf_wfrag_T(const float* W, int mt, int kb, int lane) {
    constexpr int MASK = bf_mask(KP);
    const int q = lane >> 4, rr = (lane >> 2) & 3, pp = lane & 3;
    const int r0 = 32 * kb + 4 * q + rr, r1 = r0 + 16;
    const int pi = 4 * (mt >> 1) + pp, e = 2 * (mt & 1);  // element offset 4 (mt & 1) u16 = 2 (mt & 1) dwords
    const float* p0 = W + r0 * KP + 8 * (pi ^ (r0 & MASK)) + e;
    const float* p1 = W + r1 * KP + 8 * (pi ^ (r1 & MASK)) + e;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    BfOp o;
    const s16x4 zz = {0, 0, 0, 0};
    const s16x4 h0 = ds_tr16(p0), h1 = SECOND ? ds_tr16(p1) : zz;
    const s16x8 h = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    o.hi = __builtin_bit_cast(bf16x8, h);
    if (PREC == PREC_BF16X3) {
        const s16x4 l0 = ds_tr16(p0 + 4), l1 = SECOND ? ds_tr16(p1 + 4) : zz;
        const s16x8 l = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        o.lo = __builtin_bit_cast(bf16x8, l);
    } else {
        o.lo = o.hi;
    }
    return o;
}
// out tile mt of  W^T[in][out] * in  where `in` has KB blocks over W's ROW (out-feature) index; ROWT = 16-row tiles the
// image really has (2 KB, or 2 KB - 1)
template <int PREC, int KB, int KP, int ROWT = 2 * KB>
__device__ __forceinline__ f32x4 bf_tile_T(const float* W, int mt, const BfOp (&in)[KB], f32x4 acc, int lane) {
    constexpr int NKB = KB;
    BfOp a[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
        a[kb] = (2 * kb + 1 < ROWT) ? bf_wfrag_T<PREC, KP, true>(W, mt, kb, lane) : bf_wfrag_T<PREC, KP, false>(W, mt, kb, lane);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) acc = bf_mma<PREC>(a[kb], in[kb], acc);
    return acc;
}
// ---- a dgrad layer with the fragments of tile mt + 1 requested before the MFMAs of tile mt.  With the 16-cycle bf16 MFMAs a
// tile's chain (KB MFMAs) is shorter than the LDS round trip of its fragments: tile by tile, each wave waits ~150 cycles per
// ~64 cycles of MFMA issue (profiles/r02_notes.md, "dg2" / "dg1" stamps).  PF = fragments (k-blocks) of the next tile kept
// in flight: all KB in the plain form (hi only), fewer in the split form (registers).
template <int PREC, int KB, int KP, int NT, int ROWT, int PF, typename F>
__device__ __forceinline__ void bf_layer_T(const float* W, const BfOp (&in)[KB], int lane, F&& sink) {
    static_assert(PF >= 1 && PF <= KB, "");
    auto frag = [&](int mt, int kb) {
        return (2 * kb + 1 < ROWT) ? bf_wfrag_T<PREC, KP, true>(W, mt, kb, lane) : bf_wfrag_T<PREC, KP, false>(W, mt, kb, lane);
    };
    BfOp cur[PF];
#pragma unroll
    for (int kb = 0; kb < PF; ++kb) cur[kb] = frag(0, kb);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        BfOp rest[KB - PF + 1];  // (+1: no zero-length array)
#pragma unroll
        for (int kb = PF; kb < KB; ++kb) rest[kb - PF] = frag(mt, kb);
        BfOp nxt[PF];
#pragma unroll
        for (int kb = 0; kb < PF; ++kb) nxt[kb] = frag(mt + 1 < NT ? mt + 1 : mt, kb);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc = zero4();
#pragma unroll
        for (int kb = 0; kb < PF; ++kb) acc = bf_mma<PREC>(cur[kb], in[kb], acc);
#pragma unroll
        for (int kb = PF; kb < KB; ++kb) acc = bf_mma<PREC>(rest[kb - PF], in[kb], acc);
        sink(mt, acc);
#pragma unroll
        for (int kb = 0; kb < PF; ++kb) cur[kb] = nxt[kb];
    }
}

// the same for a forward layer (row fragments, ds_read_b128)
template <int PREC, int KB, int KP, int NT, int PF, typename F>
__device__ __forceinline__ void bf_layer_fwd(const float* W, const BfOp (&in)[KB], int m, int q, F&& sink) {
    static_assert(PF >= 1 && PF <= KB, "");
    BfOp cur[PF];
#pragma unroll
    for (int kb = 0; kb < PF; ++kb) cur[kb] = bf_wfrag<PREC, KP>(W, 0, kb, m, q);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        BfOp rest[KB - PF + 1];
#pragma unroll
        for (int kb = PF; kb < KB; ++kb) rest[kb - PF] = bf_wfrag<PREC, KP>(W, mt, kb, m, q);
        BfOp nxt[PF];
#pragma unroll
        for (int kb = 0; kb < PF; ++kb) nxt[kb] = bf_wfrag<PREC, KP>(W, mt + 1 < NT ? mt + 1 : mt, kb, m, q);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc = zero4();
#pragma unroll
        for (int kb = 0; kb < PF; ++kb) acc = bf_mma<PREC>(cur[kb], in[kb], acc);
#pragma unroll
        for (int kb = PF; kb < KB; ++kb) acc = bf_mma<PREC>(rest[kb - PF], in[kb], acc);
        sink(mt, acc);
#pragma unroll
        for (int kb = 0; kb < PF; ++kb) cur[kb] = nxt[kb];
    }
}

// ---- wgrad operands through LDS.  A wgrad contracts over BATCH rows, so both operands must have rows along the k-slots of
// a lane, while the registers hold them with rows along the lanes (C/D layout).  The fp32 kernels transpose at write time
// (scattered ds_write_b32 into feature-major buffers) and the first bf16 build converted those fp32 fragments in every
// reading wave.  Here every value is converted ONCE, by its owner, and stored row-major [batch row][feature] as bf16 (one
// ds_write_b64 per tile: the lane's 4 features), hi plane and - split form - lo plane; the readers get their fragments with
// the hardware transpose ds_read_b64_tr_b16 (block rows = batch rows = k-slots, block columns = the 16 features of a tile),
// two reads per operand and plane, no VALU.  k-slot j of lane group q in k-block kb is batch row 32 kb + 16 (j >> 2) + 4 q
// + (j & 3) - for both operands, and for a B operand formed in registers from 16-row slices (bf_pack(slice 2 kb, 2 kb + 1)).
// Layout.  The unit is the 8-byte chunk (row r, tile t, quad p) = the 4 features 16 t + 4 p .. + 3 of batch row r: one lane of
// a write, one lane address of a transposed read.  Chunks of 8 consecutive rows x one tile form a 256-byte block (one LDS
// bank row); inside it the chunk sits at 32 (r & 7) + 8 (p ^ ((r >> 2) & 3)) bytes; blocks are ordered (r >> 3) * FT + t
// (FT = tiles per row of the buffer).  Banking (MI355X_MICROARCH.md "LDS"): a transposed read is served per 32-lane half =
// 8 consecutive rows x 4 quads of one tile = exactly one block, all 64 banks once; a ds_write_b64 is served per 16
// consecutive lanes = 16 rows of one (tile, quad) over 32 banks: rows r & 3 pick the 32-byte group, the quad XOR
// (r >> 2) & 3 the chunk inside it - 16 different chunks mod 128 bytes.  (Plain row-major rows were 4-way conflicted on
// the write side: 2 k cycles per wgrad round in the decoder, profiles/r02_notes.md.)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int FT>
__device__ __forceinline__ int bf_stage_off(int row, int t, int p) {  // dword offset of chunk (row, tile t, quad p)
    return 64 * ((row >> 3) * FT + t) + 8 * (row & 7) + 2 * (p ^ ((row >> 2) & 3));
}
// lane (row = batch row inside the staged rows, q) writes its 4 features 16 t + 4 q .. + 3
template <int PREC, int FT>
__device__ __forceinline__ void bf_stage_write(float* hi, float* lo, int row, int t, int q, f32x4 v) {
    const int off = bf_stage_off<FT>(row, t, q);
    const u32x2 h = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3])};
    *reinterpret_cast<u32x2*>(hi + off) = h;
    if (PREC == PREC_BF16X3) {
        const u32x2 l = {pk_bf16_lo(v[0], v[1], h[0]), pk_bf16_lo(v[2], v[3], h[1])};
        *reinterpret_cast<u32x2*>(lo + off) = l;
    }
}
// the same from an already packed operand (bf_acts / bf_pack: tiles 2 kb and 2 kb + 1 of the lane's row): no conversion at all.
// NT = tiles the staged activation has (an odd count leaves the second half of the last operand unwritten).
template <int PREC, int FT, int NT>
__device__ __forceinline__ void bf_stage_write_op(float* hi, float* lo, int row, int kb, int q, const BfOp& op) {
    const u32x4 h = __builtin_bit_cast(u32x4, op.hi);
    const int o0 = bf_stage_off<FT>(row, 2 * kb, q);  // the next tile is the next block: + 64 dwords
    *reinterpret_cast<u32x2*>(hi + o0) = u32x2{h[0], h[1]};
    if (2 * kb + 1 < NT) *reinterpret_cast<u32x2*>(hi + o0 + 64) = u32x2{h[2], h[3]};
    if (PREC == PREC_BF16X3) {
        const u32x4 l = __builtin_bit_cast(u32x4, op.lo);
        *reinterpret_cast<u32x2*>(lo + o0) = u32x2{l[0], l[1]};
        if (2 * kb + 1 < NT) *reinterpret_cast<u32x2*>(lo + o0 + 64) = u32x2{l[2], l[3]};
    }
}
// fragment (A or B operand) of feature tile t, k-block kb (staged rows 32 kb .. 32 kb + 31)
template <int PREC, int FT>
__device__ __forceinline__ BfOp bf_stage_frag(const float* hi, const float* lo, int t, int kb, int lane) {
    const int g = lane >> 4, rr = (lane >> 2) & 3, pp = lane & 3;
    const int r0 = 32 * kb + 4 * g + rr;
    const int off = bf_stage_off<FT>(r0, t, pp);  // row r0 + 16: two 8-row block rows further, same place inside the block
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    BfOp o;
    const s16x4 h0 = ds_tr16(hi + off), h1 = ds_tr16(hi + off + 128 * FT);
    const s16x8 h = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    o.hi = __builtin_bit_cast(bf16x8, h);
    if (PREC == PREC_BF16X3) {
        const s16x4 l0 = ds_tr16(lo + off), l1 = ds_tr16(lo + off + 128 * FT);
        const s16x8 l = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        o.lo = __builtin_bit_cast(bf16x8, l);
    } else {
        o.lo = o.hi;
    }
    return o;
}

// ---- NB batch tiles per wave (the 4-wave decoder kernel): one weight fragment feeds NB accumulator chains
template <int PREC, int KB, int KP, int NB, int NKB = KB>
__device__ __forceinline__ void bf_tile_fwd_nb(const float* W, int mt, const BfOp (&in)[NB][KB], f32x4 (&acc)[NB], int m,
                                               int q) {
    BfOp a[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) a[kb] = bf_wfrag<PREC, KP>(W, mt, kb, m, q);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = bf_mma<PREC>(a[kb], in[nb][kb], acc[nb]);
}
template <int PREC, int KB, int KP, int NB, int ROWT = 2 * KB>
__device__ __forceinline__ void bf_tile_T_nb(const float* W, int mt, const BfOp (&in)[NB][KB], f32x4 (&acc)[NB], int lane) {
    constexpr int NKB = KB;
    BfOp a[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
        a[kb] = (2 * kb + 1 < ROWT) ? bf_wfrag_T<PREC, KP, true>(W, mt, kb, lane) : bf_wfrag_T<PREC, KP, false>(W, mt, kb, lane);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = bf_mma<PREC>(a[kb], in[nb][kb], acc[nb]);
}
template <int PREC, int KT, int NB>
__device__ __forceinline__ void bf_acts_nb(const f32x4 (&in)[NB][KT], BfOp (&out)[NB][(KT + 1) / 2]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bf_acts<PREC, KT>(in[nb], out[nb]);
}
#endif  // __HIPCC__

}  // namespace vpc
