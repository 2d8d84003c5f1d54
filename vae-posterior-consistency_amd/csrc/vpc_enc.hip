// Encoder kernels (gfx950): q(z|x) MLP  d -> 100 -> 50 -> (mean | logvar), forward and backward.
//
// Reference semantics: Reg_VAE.encoder / vanilla_VAE.encoder, src/models/VAE.py:387-395, 1155-1163
//   h = relu(W1 (x*mask) + b1); h = relu(W2 h + b2); mean, logvar = chunk(W3 h + b3); z = mean + eps*exp(logvar/2)
// and its autograd (src/experiment_main/train.py:115).  See vpc_device.h for the register-chained design.
#include "vpc_device.h"
#include "vpc_bf16.h"
#include "vpc_abi_internal.h"

namespace vpc {

#ifdef VPC_ABLATE
#define VPC_STAMP(i)                                        \
    do {                                                    \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        T[i] += t_ - tlast;                                 \
        tlast = t_;                                         \
    } while (0)
#else
#define VPC_STAMP(i) do {} while (0)
#endif

// layer-1 input tile t of one row in C layout.  AUG = mask-augmented encoder input [x*mask | mask] of width 2d
// (Reg_VAE_mask / vanilla_VAE_mask, src/models/VAE.py:545-548): element f < d is x_f * m_f, d <= f < 2d is m_{f-d}.
template <bool VEC, bool AUG>
__device__ __forceinline__ f32x4 ld_input(const float* x, const uint8_t* mask, long row, int d, int t, int q, bool ok) {
    if (!AUG) {
        const f32x4 xv = ld_tile<VEC>(x, row, d, 16 * t + 4 * q, d, ok);
        const f32x4 mk = ld_mask<VEC>(mask, row, d, 16 * t + 4 * q, d, ok);
        return xv * mk;  // x.float() * mask  (VAE.py:388)
    }
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = 16 * t + 4 * q + j;
        const bool isx = ok && f < d, ism = ok && f >= d && f < 2 * d;
        const long ix = isx ? row * d + f : 0, im = (isx || ism) ? row * d + (isx ? f : f - d) : 0;
        const float m = mask[im] ? 1.f : 0.f;
        v[j] = isx ? x[ix] * m : (ism ? m : 0.f);
    }
    return v;
}

struct EncFwdArgs {
    const float* x;
    const float* img;
    const uint8_t* mask[2];
    const float* eps[2];
    float* h1[2];
    float* h2[2];
    float* mean[2];
    float* logvar[2];
    float* z[2];
    long B;
    int d, L, npass, ntiles, lp;
    int psplit;  // 1: the passes are spread over blockIdx.y (small batches), 0: every workgroup loops over them
};

// NW = waves per workgroup = 16-row batch tiles per workgroup iteration.  8 (two waves per SIMD, 128-row tiles): the
// throughput shape.  4 (one wave per SIMD, 64-row tiles, passes spread over blockIdx.y): the small-batch shape - a batch
// of 8 192 rows then occupies 256 workgroups x 4 waves = every SIMD of the chip with ONE tile-pass each, instead of 64
// workgroups that each run two passes with two waves per SIMD (the step is latency-bound there, profiles/r01_notes.md).
// PREC (vpc_bf16.h): PREC_F32 = v_mfma_f32_16x16x4_f32 on the fp32 image; PREC_BF16X3 / PREC_BF16 = v_mfma_f32_16x16x32_bf16
// on the bf16 image (same geometry: same row pitches, b1 stays fp32), activations converted tile pair by tile pair.
template <int DT, bool VEC, bool AUG, int NW, int PREC = PREC_F32>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void enc_fwd_kernel(EncFwdArgs a) {
    constexpr int TILE_ROWS = 16 * NW;  // shadows vpc::TILE_ROWS (the 8-wave value)
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef VPC_ABLATE
    unsigned long long T[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = tlast, r_begin = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int S1 = s_for_tiles(DT);
    const EncImg im(DT);
    const float* W1 = lds + im.oW1;
    const float* b1 = lds + im.ob1;
    const float* W2 = lds + im.oW2;
    const float* W3 = lds + im.oW3;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;

    // PIPE (the vectorised fast path): x is read once per tile (both passes see the same rows, only the mask differs)
    // and the mask words of the next pass - or x and the mask words of the next tile - are requested before this
    // pass's MFMAs, with branch-free loads (rows past B read row 0 and are never stored; columns past d are cleared).
    constexpr bool PIPE = VEC && !AUG;
    f32x4 xraw[DT];
    uint32_t mw[DT];
    const int cq = (4 * q + 3 < a.d) ? 4 * q : 0;
    // range-checked buffer loads relative to a tile's first row t0 (rows past B read 0)
    const int vo = (w * 16 + c) * a.d + cq;  // element offset of this lane's row / column group inside the tile's rows
    auto fetch_x = [&](long t0) {
        const __amdgpu_buffer_rsrc_t rx = rows_rsrc(a.x, t0, a.B, a.d);
#pragma unroll
        for (int t = 0; t < DT; ++t)
            xraw[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rx, 4 * (vo + ((t < DT / 2 || 16 * t + 4 * q + 3 < a.d) ? 16 * t : 0)), 0, 0));
    };
    auto fetch_m = [&](const uint8_t* m, long t0) {
        const long rem = (a.B - t0) * (long)a.d;
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint8_t*>(m) + t0 * a.d, 0, rem > 0xffffffffL ? 0xffffffffu : (uint32_t)rem, 0x00020000);
#pragma unroll
        for (int t = 0; t < DT; ++t)
            mw[t] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
                rm, vo + ((t < DT / 2 || 16 * t + 4 * q + 3 < a.d) ? 16 * t : 0), 0, 0);
    };
    const int p_lo = a.psplit ? (int)blockIdx.y : 0, p_hi = a.psplit ? p_lo + 1 : a.npass;
    // the first tile's x / mask words are requested BEFORE the weight image: both latencies overlap
    if (PIPE && (int)blockIdx.x < a.ntiles) {
        fetch_x((long)blockIdx.x * TILE_ROWS);
        fetch_m(a.mask[p_lo], (long)blockIdx.x * TILE_ROWS);
    }
    load_image<(NW == 8 ? 13 : 25)>(lds, a.img, im.total);
    __syncthreads();
    VPC_STAMP(0);

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const long row = (long)tile * TILE_ROWS + w * 16 + c;
        const bool ok = row < a.B;
        for (int p = p_lo; p < p_hi; ++p) {
            // the weight image never changes after the prologue: without this compiler barrier LICM hoists
            // every LDS weight read out of the pass loop and spills ~1 KB/lane of it to scratch
            asm volatile("" ::: "memory");
            int cc = c, qq = q;
            launder(cc, qq);
            // workspace stores go through range-checked buffer descriptors (rows past B are dropped by the hardware)
            const long row0 = (long)tile * TILE_ROWS;
            const int lrow = w * 16 + c;
            const __amdgpu_buffer_rsrc_t rh1 = rows_rsrc(a.h1[p], row0, a.B, H1P), rh2 = rows_rsrc(a.h2[p], row0, a.B, H2P);
            f32x4 xin[DT];
            if (PIPE) {
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const uint32_t vm = opaque_mask(t < DT / 2 || 16 * t + 4 * q + 3 < a.d);
                    xin[t] = xraw[t] * mask_to_f32(mw[t] & vm);  // x.float() * mask  (VAE.py:388)
                }
                if (p + 1 < p_hi) {
                    fetch_m(a.mask[p + 1], row0);
                } else if (tile + (int)gridDim.x < a.ntiles) {
                    const long tn = row0 + (long)gridDim.x * TILE_ROWS;
                    fetch_x(tn);
                    fetch_m(a.mask[p_lo], tn);
                }
            } else {
#pragma unroll
                for (int t = 0; t < DT; ++t) xin[t] = ld_input<VEC, AUG>(a.x, a.mask[p], row, a.d, t, q, ok);
            }
            VPC_STAMP(1);
            f32x4 h1[H1T], h2[H2T], mu, lv;
            if (PREC == PREC_F32) {
#pragma unroll
                for (int mt = 0; mt < H1T; ++mt) {
                    f32x4 acc = *reinterpret_cast<const f32x4*>(b1 + 16 * mt + 4 * q);
                    acc = tile_fwd<DT, S1>(W1, mt, xin, acc, cc, qq);
                    h1[mt] = relu4(acc);
                    st_rows(rh1, lrow, H1P, 16 * mt + 4 * q, h1[mt]);
                }
                VPC_STAMP(2);
                launder(cc, qq);
#pragma unroll
                for (int mt = 0; mt < H2T; ++mt) {
                    h2[mt] = relu4(tile_fwd<H1T, 128, NK1>(W2, mt, h1, zero4(), cc, qq));
                    st_rows(rh2, lrow, H2P, 16 * mt + 4 * q, h2[mt]);
                }
                VPC_STAMP(3);
                mu = tile_fwd<H2T, 64, NK2>(W3, 0, h2, zero4(), cc, qq);
                lv = tile_fwd<H2T, 64, NK2>(W3, 1, h2, zero4(), cc, qq);
            } else {
                constexpr int KB1 = (DT + 1) / 2;
                BfOp xb[KB1];
                bf_acts<PREC, DT>(xin, xb);
                // (fragments of tile mt + 1 in flight during tile mt: vpc_bf16.h, bf_layer_fwd)
                // plain bf16: the h1 / h2 workspaces hold the PACKED operands (what the backward kernel's MFMAs consume
                // anyway): rows of [k-block][q][8 x bf16], 256 / 128 bytes per row inside the same allocation - half the
                // bytes of the fp32 rows in both directions (enc_fwd is HBM-bound in this form)
                constexpr bool HPK = PREC == PREC_BF16;
                bf_layer_fwd<PREC, KB1, S1, H1T, PREC == PREC_BF16 ? KB1 : (KB1 < 2 ? KB1 : 2)>(W1, xb, cc, qq, [&](int mt, f32x4 acc) {
                    h1[mt] = relu4(acc + *reinterpret_cast<const f32x4*>(b1 + 16 * mt + 4 * q));
                    if (!HPK) st_rows(rh1, lrow, H1P, 16 * mt + 4 * q, h1[mt]);
                });
                VPC_STAMP(2);
                launder(cc, qq);
                BfOp h1b[4];
                bf_acts<PREC, H1T>(h1, h1b);
                if (HPK) {
                    const __amdgpu_buffer_rsrc_t rp = rows_rsrc(a.h1[p], row0, a.B, 64);
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) st_rows(rp, lrow, 64, 16 * kb + 4 * q, __builtin_bit_cast(f32x4, h1b[kb].hi));
                }
                bf_layer_fwd<PREC, 4, 128, H2T, PREC == PREC_BF16 ? 4 : 2>(W2, h1b, cc, qq, [&](int mt, f32x4 acc) {
                    h2[mt] = relu4(acc);
                    if (!HPK) st_rows(rh2, lrow, H2P, 16 * mt + 4 * q, h2[mt]);
                });
                VPC_STAMP(3);
                BfOp h2b[2];
                bf_acts<PREC, H2T>(h2, h2b);
                if (HPK) {
                    const __amdgpu_buffer_rsrc_t rp = rows_rsrc(a.h2[p], row0, a.B, 32);
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) st_rows(rp, lrow, 32, 16 * kb + 4 * q, __builtin_bit_cast(f32x4, h2b[kb].hi));
                }
                mu = bf_tile_fwd<PREC, 2, 64>(W3, 0, h2b, zero4(), cc, qq);
                lv = bf_tile_fwd<PREC, 2, 64>(W3, 1, h2b, zero4(), cc, qq);
            }
            if (a.lp == 16) {  // padded workspaces: rows are 16 floats, features >= L are exact zeros
                st_rows(rows_rsrc(a.mean[p], row0, a.B, 16), lrow, 16, 4 * q, mu);
                st_rows(rows_rsrc(a.logvar[p], row0, a.B, 16), lrow, 16, 4 * q, lv);
            } else {
                st_tile<false>(a.mean[p], row, a.L, 4 * q, a.L, ok, mu);
                st_tile<false>(a.logvar[p], row, a.L, 4 * q, a.L, ok, lv);
            }
            if (a.z[p]) {
                f32x4 z = mu;
                if (a.eps[p]) {
                    const f32x4 e = ld_tile<false>(a.eps[p], row, a.L, 4 * q, a.L, ok);
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[j] = mu[j] + e[j] * expf(lv[j] * 0.5f);  // rsample
                }
                st_tile<false>(a.z[p], row, a.L, 4 * q, a.L, ok, z);
            }
            VPC_STAMP(4);
        }
    }
#ifdef VPC_ABLATE
    if ((blockIdx.x == 0 || blockIdx.x == 100) && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 2)
        printf("enc_fwd blk %d wave %d cycles: prologue %llu loadx %llu L1 %llu L2 %llu L3+st %llu | total %llu cycles in %llu x 10 ns\n",
               blockIdx.x, (int)(threadIdx.x >> 6), T[0], T[1], T[2], T[3], T[4],
               (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin),
               (unsigned long long)(__builtin_amdgcn_s_memrealtime() - r_begin));
#endif
}

struct EncBwdArgs {
    const float* x;
    const float* img;
    const uint8_t* mask[2];
    const float* h1[2];
    const float* h2[2];
    const float* dmean[2];
    const float* dlogvar[2];
    float* part;
    long B;
    int d, L, npass, ntiles, lp, dbg;
    int psplit;  // as EncFwdArgs
};

#ifdef VPC_ABLATE
#define ABLE(bit) ((a.dbg & (bit)) != 0)  // timing experiments (diagnostic build): 1 no staging writes, 2 no barriers, 4 no wgrad 1/2 MFMAs
#else
#define ABLE(bit) false
#endif

// NW as in enc_fwd_kernel.  Every wave owns OWN = 8 / NW slices of each wgrad (the 8-wave kernel: one): in tiles
// w + NW i of dW1 / dW2 and tiles w + NW i of dW3, written to the partial block in the slots of the 8-wave layout
// (vpc_layout.h), so the gradient reduction does not care which shape ran.
template <int DT, bool VEC, bool AUG, int NW, int PREC = PREC_F32>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) void enc_bwd_kernel(EncBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef VPC_ABLATE
    unsigned long long T[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
    // Staging is full width (all batch rows of the workgroup tile at once): ONE write + barrier + read round per
    // wgrad round (two rounds: layer 2, then layers 1 + 3 together), 4 barriers per pass instead of 12 (barriers were 16 %
    // of this kernel, `profiles/r01_notes.md`).  That
    // fits in LDS because the B operand of the layer-1 wgrad - x * mask, the only use of x in this kernel - never
    // goes through LDS: a B fragment wants batch rows along the register index and features along the lanes, which
    // is how row-major x lies in memory, so the tile's owner wave reads it from global memory directly.
    constexpr int TILE_ROWS = 16 * NW, CH = TILE_ROWS, NTHR = 64 * NW, OWN = 8 / NW;
    constexpr bool PIPE = VEC && !AUG;
    const EncImg im(DT);
    const int nW = im.total - im.oW2;  // only W2, W3 are needed (layer 1 has no dgrad)
    float* W2 = lds;
    float* W3 = lds + (im.oW3 - im.oW2);
    float* stA = lds + nW;             // [112][CH]  dY operands (dml, dh2, dh1)
    float* stB = stA + H1P * CH;       // [112][CH]  activations (h2, h1)
    float* db1s = stB + H1P * CH;      // [NW][128]: per-wave bias-gradient sums (no atomics across waves: bit-reproducible)
    // bf16 engine: the same buffers hold the operands as bf16, row-major [batch row][7 tile slots], hi plane + lo plane
    // (vpc_bf16.h, bf_stage_*): written once by the owner of the row, read back with the transposing LDS read
    constexpr bool BF = PREC != PREC_F32;
    constexpr int SKB = CH / 32;
    float* sAh = stA;
    float* sAl = stA + 56 * CH;
    float* sBh = stB;
    float* sBl = stB + 56 * CH;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int p_lo = a.psplit ? (int)blockIdx.y : 0, p_hi = a.psplit ? p_lo + 1 : a.npass;
    // row-layout operands of one (tile, pass) through range-checked buffer descriptors: rows past B read 0 - zero seeds make
    // dh2, dh1 and every wgrad contribution of such a row exactly zero.  (hipcc turns `ok ? load : 0` into an exec-masked
    // branch with a vmcnt(0) wait at the join.)  The first (tile, pass) of the workgroup is requested before the weight
    // image: in the small-batch shape there is only one, and the two latencies otherwise add up.
    // plain bf16: h1 / h2 arrive as the packed operands enc_fwd stored (4 + 2 k-blocks of 8 bf16 per lane)
    constexpr bool HPK = PREC == PREC_BF16;
    struct RowIn { f32x4 dml[2], h2[HPK ? 1 : H2T], h1[HPK ? 1 : H1T]; BfOp h2p[HPK ? 2 : 1], h1p[HPK ? 4 : 1]; };
    auto fetch_rows = [&](int tile, int p, RowIn& R) {
        const long row0 = (long)tile * TILE_ROWS;
        const int lrow = w * 16 + c;
        if (a.lp == 16) {
            R.dml[0] = ld_rows(rows_rsrc(a.dmean[p], row0, a.B, 16), lrow, 16, 4 * q);
            R.dml[1] = ld_rows(rows_rsrc(a.dlogvar[p], row0, a.B, 16), lrow, 16, 4 * q);
        } else {
            const long row = row0 + lrow;
            R.dml[0] = ld_tile<false>(a.dmean[p], row, a.L, 4 * q, a.L, row < a.B);
            R.dml[1] = ld_tile<false>(a.dlogvar[p], row, a.L, 4 * q, a.L, row < a.B);
        }
        if (HPK) {
            const __amdgpu_buffer_rsrc_t rp2 = rows_rsrc(a.h2[p], row0, a.B, 32), rp1 = rows_rsrc(a.h1[p], row0, a.B, 64);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                R.h2p[kb].hi = __builtin_bit_cast(bf16x8, ld_rows(rp2, lrow, 32, 16 * kb + 4 * q));
                R.h2p[kb].lo = R.h2p[kb].hi;
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                R.h1p[kb].hi = __builtin_bit_cast(bf16x8, ld_rows(rp1, lrow, 64, 16 * kb + 4 * q));
                R.h1p[kb].lo = R.h1p[kb].hi;
            }
            return;
        }
        const __amdgpu_buffer_rsrc_t rh2 = rows_rsrc(a.h2[p], row0, a.B, H2P), rh1 = rows_rsrc(a.h1[p], row0, a.B, H1P);
#pragma unroll
        for (int t = 0; t < H2T; ++t) R.h2[t] = ld_rows(rh2, lrow, H2P, 16 * t + 4 * q);
#pragma unroll
        for (int t = 0; t < H1T; ++t) R.h1[t] = ld_rows(rh1, lrow, H1P, 16 * t + 4 * q);
    };
    // (small-batch shape only: the 8-wave kernel sits at its 256-register limit and would spill the extra copy)
    constexpr bool HOIST = NW == 4;
    RowIn Rpre;
    bool have_pre = false;
    if (HOIST && (int)blockIdx.x < a.ntiles) {
        fetch_rows(blockIdx.x, p_lo, Rpre);
        have_pre = true;
    }
    load_image<(NW == 8 ? 5 : 10)>(lds, a.img + im.oW2, nW);
    for (int i = threadIdx.x; i < NW * 128; i += NTHR) db1s[i] = 0.f;
    __syncthreads();
    int sb[4];  // per-lane element offsets of the staging writes (tile 0); tiles add a compile-time constant
    stage_bases<CH>(sb, 16 * w, c, q);

    f32x4 acc1[OWN][H1T], acc2[OWN][H2T], acc3[OWN], dbacc[H1T];
#pragma unroll
    for (int i = 0; i < H1T; ++i) dbacc[i] = zero4();
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
        acc3[o] = zero4();
#pragma unroll
        for (int i = 0; i < H1T; ++i) acc1[o][i] = zero4();
#pragma unroll
        for (int i = 0; i < H2T; ++i) acc2[o][i] = zero4();
    }

    // layer-1 input features of this lane as B-fragment columns (owned in tiles w + NW o)
    const int din = AUG ? 2 * a.d : a.d;
    bool fx[OWN], fm[OWN];
    int colB[OWN];
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
        const int fB = 16 * (w + NW * o) + c;
        fx[o] = fB < a.d; fm[o] = AUG && fB >= a.d && fB < din;  // x * mask column / appended mask column
        colB[o] = fx[o] ? fB : (fm[o] ? fB - a.d : 0);
    }

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const long row0 = (long)tile * TILE_ROWS;
        for (int p = p_lo; p < p_hi; ++p) {
            int cc = c, qq = q;
            launder(cc, qq);
            VPC_STAMP(0);
            // B fragments of slice s (batch rows row0 + 16 s + 4 q + j): raw x and mask bytes; rows past B read row 0
            // (their dh1 is exactly zero), columns past the input width are cleared when the fragment is formed
            // range-checked buffer loads relative to the tile's first row: a row past B is out of range and reads 0 (x and
            // mask byte), so no clamp; per load one 32-bit add of the slice offset to a per-lane base (the range check
            // covers voffset only, so the slice offset must not go into soffset).  Formed from the laundered lane id:
            // hipcc otherwise precomputes all 32 addresses per tile and spills them.
            const __amdgpu_buffer_rsrc_t rx = rows_rsrc(a.x, row0, a.B, a.d);
            const long mrem = (a.B - row0) * (long)a.d;
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t*>(a.mask[p]) + row0 * a.d, 0, mrem > 0xffffffffL ? 0xffffffffu : (uint32_t)mrem, 0x00020000);
            auto ld_xb = [&](int o, int sl, f32x4& xv, uint32_t& mb) {
                const int vo0 = 4 * qq * a.d + colB[o];  // element offset of (row 4 q, this lane's column)
                mb = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int off = vo0 + (16 * sl + j) * a.d;
                    xv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, 4 * off, 0, 0));
                    mb |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rm, off, 0, 0) << (8 * j);
                }
            };
            auto mk_fb = [&](int o, const f32x4& xv, uint32_t mb) -> f32x4 {
                if (!AUG) return xv * mask_to_f32(mb & (fx[o] ? 0xffffffffu : 0u));  // columns past the input width: mask 0
                const f32x4 m = mask_to_f32(mb);
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fx[o] ? xv[j] * m[j] : (fm[o] ? m[j] : 0.f);
                return v;
            };
            RowIn R;
            if (HOIST && have_pre) R = Rpre; else fetch_rows(tile, p, R);
            have_pre = false;
            f32x4 (&dml)[2] = R.dml;
            auto& h2 = R.h2;
            auto& h1 = R.h1;
            // ReLU gate of tile mt from the fp32 tile or, packed form, from the bf16 halves of its operand (relu output >= +0:
            // positive <=> the 16 bits are not zero)
            auto gate_h2 = [&](int mt, f32x4 acc) { return HPK ? bf_gate(acc, R.h2p[HPK ? mt >> 1 : 0], mt & 1) : gate4(acc, h2[HPK ? 0 : mt]); };
            auto gate_h1 = [&](int mt, f32x4 acc) { return HPK ? bf_gate(acc, R.h1p[HPK ? mt >> 1 : 0], mt & 1) : gate4(acc, h1[HPK ? 0 : mt]); };
            // (dW3~ += dml * h2^T is staged and computed together with the layer-1 wgrad below: the B staging buffer is
            // free in that round because x never goes through LDS - 4 instead of 6 barriers per pass)
            VPC_STAMP(1);
            // ---- dh2 = relu'(h2) * (W3~^T dml)
            launder(cc, qq);
            f32x4 dh2[H2T];
            BfOp dmlb[1], dh2b[2];  // bf16 engine: packed once, used by the dgrad MFMAs AND the wgrad staging writes
            if (BF) dmlb[0] = bf_pack<PREC>(dml[0], dml[1]);
#pragma unroll
            for (int mt = 0; mt < H2T; ++mt) {
                if (PREC == PREC_F32) dh2[mt] = gate4(tile_T<2, 64>(W3, mt, dml, zero4(), cc, qq), h2[mt]);
                else dh2[mt] = gate_h2(mt, bf_tile_T<PREC, 1, 64>(W3, mt, dmlb, zero4(), 16 * qq + cc));
            }
            if (BF) bf_acts<PREC, H2T>(dh2, dh2b);
            VPC_STAMP(2);
            // ---- dW2~ += dh2 * h1^T   (owner: wave w -> in tiles w + NW o < 7, all 4 out tiles)
            launder(cc, qq);
            if (!ABLE(2)) lds_barrier();
            if (BF && !ABLE(1)) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) bf_stage_write_op<PREC, 7, H2T>(sAh, sAl, 16 * w + cc, kb, qq, dh2b[kb]);
                if (HPK) {
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) bf_stage_write_op<PREC, 7, H1T>(sBh, sBl, 16 * w + cc, kb, qq, R.h1p[kb]);
                } else {
#pragma unroll
                    for (int t = 0; t < H1T; ++t) bf_stage_write<PREC, 7>(sBh, sBl, 16 * w + cc, t, qq, h1[HPK ? 0 : t]);
                }
            } else if (!ABLE(1)) {
#pragma unroll
                for (int t = 0; t < H2T; ++t) stage_write_b<CH>(stA, t, dh2[t], sb);
#pragma unroll
                for (int t = 0; t < H1T; ++t) stage_write_b<CH>(stB, t, h1[t], sb);
            }
            if (!ABLE(2)) lds_barrier();
#pragma unroll
            for (int o = 0; o < OWN; ++o) {
                if (PREC != PREC_F32) {
                    if (w + NW * o < H1T && !ABLE(4)) {
#pragma unroll
                        for (int kb = 0; kb < SKB; ++kb) {
                            asm volatile("" ::: "memory");
                            const BfOp fb = bf_stage_frag<PREC, 7>(sBh, sBl, w + NW * o, kb, 16 * qq + cc);
#pragma unroll
                            for (int mt = 0; mt < H2T; ++mt) {
                                const BfOp fa = bf_stage_frag<PREC, 7>(sAh, sAl, mt, kb, 16 * qq + cc);
                                acc2[o][mt] = bf_mma<PREC>(fa, fb, acc2[o][mt]);
                            }
                        }
                    }
                } else if (w + NW * o < H1T && !ABLE(4)) {
#pragma unroll
                    for (int s = 0; s < CH / 16; ++s) {
                        asm volatile("" ::: "memory");
                        const f32x4 fb = stage_frag<CH>(stB, w + NW * o, s, cc, qq);
                        // A fragments double-buffered by hand (hipcc sinks each LDS read to its first use: one exposed
                        // LDS latency per 4 MFMAs); the sched_barrier pins the read of mt+1 above the MFMAs of mt
                        f32x4 fa = stage_frag<CH>(stA, 0, s, cc, qq);
#pragma unroll
                        for (int mt = 0; mt < H2T; ++mt) {
                            const f32x4 fn = stage_frag<CH>(stA, mt + 1 < H2T ? mt + 1 : mt, s, cc, qq);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc2[o][mt] = VPC_MFMA(fa[j], fb[j], acc2[o][mt]);
                            fa = fn;
                        }
                    }
                }
            }
            VPC_STAMP(3);
            // the first two B slices of the layer-1 wgrad come in under the dh1 MFMAs
            constexpr int NS = CH / 16;
            f32x4 xb[OWN][3];  // rotating: slices s, s + 1, s + 2
            uint32_t mbb[OWN][3];
#pragma unroll
            for (int o = 0; o < OWN; ++o)
                if (w + NW * o < DT) {
                    ld_xb(o, 0, xb[o][0], mbb[o][0]);
                    ld_xb(o, 1, xb[o][1], mbb[o][1]);
                    if (PREC != PREC_F32) ld_xb(o, 2, xb[o][2], mbb[o][2]);  // bf16: slices are consumed in pairs
                }
            // ---- dh1 = relu'(h1) * (W2~^T dh2);  db1 += dh1
            launder(cc, qq);
            f32x4 dh1[H1T];
            // db1: per-lane running sums over all tile-passes; the cross-lane reduction happens ONCE, after the
            // loops (it used to be 4 DPP adds + a predicated ds_add per value and pass: ~170 VALU and 28 exec-masked
            // basic blocks in the middle of the dgrad MFMA stream)
            if (BF) {
                bf_layer_T<PREC, 2, 128, H1T, 4, 2>(W2, dh2b, 16 * qq + cc, [&](int mt, f32x4 acc) {
                    dh1[mt] = gate_h1(mt, acc);
                    dbacc[mt] += dh1[mt];
                });
            } else {
#pragma unroll
                for (int mt = 0; mt < H1T; ++mt) {
                    dh1[mt] = gate4(tile_T<H2T, 128, NK2>(W2, mt, dh2, zero4(), cc, qq), h1[mt]);
                    dbacc[mt] += dh1[mt];
                }
            }
            VPC_STAMP(4);
            // ---- dW1 += dh1 * (x*mask)^T   (owner: wave w -> in tiles w + NW o < DT, all 7 out tiles; B straight from global)
            launder(cc, qq);
            if (!ABLE(2)) lds_barrier();
            if (BF && !ABLE(1)) {
#pragma unroll
                for (int t = 0; t < H1T; ++t) bf_stage_write<PREC, 7>(sAh, sAl, 16 * w + cc, t, qq, dh1[t]);
                if (HPK) {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) bf_stage_write_op<PREC, 7, H2T>(sBh, sBl, 16 * w + cc, kb, qq, R.h2p[kb]);
                } else {
#pragma unroll
                    for (int t = 0; t < H2T; ++t) bf_stage_write<PREC, 7>(sBh, sBl, 16 * w + cc, t, qq, h2[HPK ? 0 : t]);
                }
                bf_stage_write_op<PREC, 7, 6>(sBh, sBl, 16 * w + cc, H2T / 2, qq, dmlb[0]);  // tiles 4, 5
            } else if (!ABLE(1)) {
#pragma unroll
                for (int t = 0; t < H1T; ++t) stage_write_b<CH>(stA, t, dh1[t], sb);
                // operands of dW3~ (tile t8 = w + NW o -> out tile t8 >> 2, in tile t8 & 3): h2 -> stB tiles 0..3, dml -> stB tiles 4, 5
#pragma unroll
                for (int t = 0; t < H2T; ++t) stage_write_b<CH>(stB, t, h2[t], sb);
                stage_write_b<CH>(stB, H2T, dml[0], sb);
                stage_write_b<CH>(stB, H2T + 1, dml[1], sb);
            }
            if (!ABLE(2)) lds_barrier();
#pragma unroll
            for (int o = 0; o < OWN; ++o) {
                const int t8 = w + NW * o;
                if (PREC == PREC_F32) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const f32x4 fa = stage_frag<CH>(stB, H2T + (t8 >> 2), s, cc, qq);
                        const f32x4 fb = stage_frag<CH>(stB, t8 & 3, s, cc, qq);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc3[o] = VPC_MFMA(fa[j], fb[j], acc3[o]);
                    }
                } else {
#pragma unroll
                    for (int kb = 0; kb < SKB; ++kb) {
                        const BfOp fa = bf_stage_frag<PREC, 7>(sBh, sBl, H2T + (t8 >> 2), kb, 16 * qq + cc);
                        const BfOp fb = bf_stage_frag<PREC, 7>(sBh, sBl, t8 & 3, kb, 16 * qq + cc);
                        acc3[o] = bf_mma<PREC>(fa, fb, acc3[o]);
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < OWN; ++o) {
                if (PREC != PREC_F32) {
                    if (w + NW * o < DT) {
                        // slices in pairs (one 32-row k-block per MFMA); ring of 4 slices: block sb2 + 1 in flight
                        f32x4 xq[4] = {xb[o][0], xb[o][1], xb[o][2], zero4()};
                        uint32_t mq[4] = {mbb[o][0], mbb[o][1], mbb[o][2], 0u};
                        if (3 < NS) ld_xb(o, 3, xq[3], mq[3]);
#pragma unroll
                        for (int sb2 = 0; sb2 < NS / 2; ++sb2) {
                            asm volatile("" ::: "memory");
                            const BfOp fb = bf_pack<PREC>(mk_fb(o, xq[(2 * sb2) & 3], mq[(2 * sb2) & 3]),
                                                          mk_fb(o, xq[(2 * sb2 + 1) & 3], mq[(2 * sb2 + 1) & 3]));
                            if (2 * sb2 + 4 < NS) ld_xb(o, 2 * sb2 + 4, xq[(2 * sb2) & 3], mq[(2 * sb2) & 3]);
                            if (2 * sb2 + 5 < NS) ld_xb(o, 2 * sb2 + 5, xq[(2 * sb2 + 1) & 3], mq[(2 * sb2 + 1) & 3]);
#pragma unroll
                            for (int mt = 0; mt < H1T; ++mt) {
                                const BfOp fa = bf_stage_frag<PREC, 7>(sAh, sAl, mt, sb2, 16 * qq + cc);
                                acc1[o][mt] = bf_mma<PREC>(fa, fb, acc1[o][mt]);
                            }
                        }
                    }
                } else if (w + NW * o < DT && !ABLE(4)) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        asm volatile("" ::: "memory");  // keep each slice's loads in its slice (hipcc hoists all 8 otherwise)
                        if (s + 2 < NS) ld_xb(o, s + 2, xb[o][(s + 2) % 3], mbb[o][(s + 2) % 3]);
                        const f32x4 fb = mk_fb(o, xb[o][s % 3], mbb[o][s % 3]);
                        f32x4 fa = stage_frag<CH>(stA, 0, s, cc, qq);
#pragma unroll
                        for (int mt = 0; mt < H1T; ++mt) {
                            const f32x4 fn = stage_frag<CH>(stA, mt + 1 < H1T ? mt + 1 : mt, s, cc, qq);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc1[o][mt] = VPC_MFMA(fa[j], fb[j], acc1[o][mt]);
                            fa = fn;
                        }
                    }
                }
            }
            VPC_STAMP(5);
        }
    }
    // ---- db1 = sum of dh1 over this wave's 16 rows (lanes c): DPP butterfly inside each 16-lane row, then the c == 0
    // lanes own one (wave, feature) slot each - plain stores, fixed order, bit-reproducible
#pragma unroll
    for (int mt = 0; mt < H1T; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = dbacc[mt][j];
            v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
            v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
            v += dpp_mov<0x141>(v);  // row_half_mirror
            v += dpp_mov<0x140>(v);  // row_mirror: every lane of the 16-lane row holds the row sum
            if (c == 0) db1s[w * 128 + 16 * mt + 4 * q + j] = v;
        }
    // ---- write this workgroup's gradient partial block (8-wave slot layout: owned slice o is "wave" w + NW o)
    const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
        float* part = a.part + blk * ENC_PART + (long)(w + NW * o) * GREGS * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < H1T; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(4 * mt + j) * 64] = acc1[o][mt][j];
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(28 + 4 * mt + j) * 64] = acc2[o][mt][j];
#pragma unroll
        for (int j = 0; j < 4; ++j) part[(44 + j) * 64] = acc3[o][j];
    }
    __syncthreads();  // every wave's db1s row is complete
    if (threadIdx.x < 128) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) t += db1s[k * 128 + threadIdx.x];
        a.part[blk * ENC_PART + WAVES * GREGS * 64 + threadIdx.x] = t;
    }
#ifdef VPC_ABLATE
    VPC_STAMP(6);
    if (ABLE(64) && blockIdx.x == 100 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) % 3 == 0)
        printf("enc_bwd blk %d wave %d cycles: top %llu loads %llu dh2 %llu dW2 %llu dh1 %llu dW1+dW3 %llu epilogue %llu\n",
               blockIdx.x, (int)(threadIdx.x >> 6), T[0], T[1], T[2], T[3], T[4], T[5], T[6]);
#endif
}

static size_t enc_fwd_lds(int DT) { return sizeof(float) * EncImg(DT).total; }
static size_t enc_bwd_lds(int DT, int nw) {
    const EncImg im(DT);
    return sizeof(float) * ((im.total - im.oW2) + 2 * H1P * 16 * nw + nw * 128);
}

template <typename K, typename A>
static int launch(K kern, const A& args, int grid_x, int grid_y, int nw, size_t lds, hipStream_t stream) {
    if (!lds_attr_done(reinterpret_cast<const void*>(kern), lds)) return VPC_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(grid_x, grid_y), dim3(64 * nw), lds, stream, args);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

}  // namespace vpc

using namespace vpc;

extern "C" int vpc_encoder_fwd(const float* x, const float* enc_img, int npass, const uint8_t* const* mask,
                               const float* const* eps, float* const* h1, float* const* h2, float* const* mean,
                               float* const* logvar, float* const* z, int lat_pitch, int mask_augm, int precision,
                               long B, int d, int L, void* stream) {
    if (!x || !enc_img || !mask || !h1 || !h2 || !mean || !logvar) return VPC_ERR_ARG;
    if (npass < 1 || npass > 2 || B <= 0 || precision < 0 || precision > 2) return VPC_ERR_ARG;
    if (d < 1 || (mask_augm ? 2 * d : d) > MAX_D || L < 1 || L > MAX_L) return VPC_ERR_SHAPE;
    if (lat_pitch != L && lat_pitch != 16) return VPC_ERR_ARG;
    if (lat_pitch != L && z) return VPC_ERR_ARG;  // z is only produced in the dense [B][L] layout
    EncFwdArgs a{};
    a.x = x; a.img = enc_img; a.B = B; a.d = d; a.L = L; a.npass = npass; a.lp = lat_pitch;
    const TileShape ts = tile_shape(B, npass);
    a.ntiles = ts.ntiles; a.psplit = ts.small;
    bool vec = (d % 4 == 0) && aligned16(x);
    for (int p = 0; p < npass; ++p) {
        if (!mask[p] || !h1[p] || !h2[p] || !mean[p] || !logvar[p]) return VPC_ERR_ARG;
        a.mask[p] = mask[p]; a.eps[p] = eps ? eps[p] : nullptr; a.h1[p] = h1[p]; a.h2[p] = h2[p];
        a.mean[p] = mean[p]; a.logvar[p] = logvar[p]; a.z[p] = z ? z[p] : nullptr;
        vec = vec && ((uintptr_t)mask[p] % 4 == 0);
        if (!aligned16(h1[p]) || !aligned16(h2[p])) return VPC_ERR_ARG;
    }
    const int DT = dt_for(mask_augm ? 2 * d : d);
    const size_t lds = enc_fwd_lds(DT);
    hipStream_t s = (hipStream_t)stream;
    if (precision != PREC_F32) {  // bf16 variants: throughput shape, vector layout (d % 4 == 0), plain encoder input
        if (!vec || mask_augm) return VPC_ERR_SHAPE;
#define VPC_CASE(T)                                                                                                    \
    case T:                                                                                                            \
        if (ts.small)                                                                                                  \
            return precision == PREC_BF16X3                                                                            \
                       ? launch(enc_fwd_kernel<T, true, false, 4, PREC_BF16X3>, a, ts.grid_x, ts.grid_y, 4, lds, s)    \
                       : launch(enc_fwd_kernel<T, true, false, 4, PREC_BF16>, a, ts.grid_x, ts.grid_y, 4, lds, s);     \
        return precision == PREC_BF16X3                                                                                \
                   ? launch(enc_fwd_kernel<T, true, false, 8, PREC_BF16X3>, a, ts.grid_x, ts.grid_y, 8, lds, s)        \
                   : launch(enc_fwd_kernel<T, true, false, 8, PREC_BF16>, a, ts.grid_x, ts.grid_y, 8, lds, s);
        switch (DT) { VPC_CASE(1) VPC_CASE(2) VPC_CASE(4) VPC_CASE(8) }
#undef VPC_CASE
        return VPC_ERR_SHAPE;
    }
#define VPC_LAUNCH(T, NW)                                                                                   \
    (mask_augm ? launch(enc_fwd_kernel<T, false, true, NW>, a, ts.grid_x, ts.grid_y, NW, lds, s)            \
     : vec     ? launch(enc_fwd_kernel<T, true, false, NW>, a, ts.grid_x, ts.grid_y, NW, lds, s)            \
               : launch(enc_fwd_kernel<T, false, false, NW>, a, ts.grid_x, ts.grid_y, NW, lds, s))
#define VPC_CASE(T) \
    case T: return ts.small ? VPC_LAUNCH(T, 4) : VPC_LAUNCH(T, 8);
    switch (DT) { VPC_CASE(1) VPC_CASE(2) VPC_CASE(4) VPC_CASE(8) }
#undef VPC_CASE
#undef VPC_LAUNCH
    return VPC_ERR_SHAPE;
}

extern "C" int vpc_encoder_bwd(const float* x, const float* enc_img, int npass, const uint8_t* const* mask,
                               const float* const* h1, const float* const* h2, const float* const* dmean,
                               const float* const* dlogvar, int lat_pitch, int mask_augm, int precision,
                               float* partials, int* nblocks_out, long B, int d, int L, void* stream) {
    if (!x || !enc_img || !mask || !h1 || !h2 || !dmean || !dlogvar || !partials) return VPC_ERR_ARG;
    if (npass < 1 || npass > 2 || B <= 0 || precision < 0 || precision > 2) return VPC_ERR_ARG;
    if (d < 1 || (mask_augm ? 2 * d : d) > MAX_D || L < 1 || L > MAX_L) return VPC_ERR_SHAPE;
    if (lat_pitch != L && lat_pitch != 16) return VPC_ERR_ARG;
    EncBwdArgs a{};
    a.x = x; a.img = enc_img; a.part = partials; a.B = B; a.d = d; a.L = L; a.npass = npass; a.lp = lat_pitch;
#ifdef VPC_ABLATE
    if (const char* e = getenv("VPC_DEBUG_ENC")) a.dbg = atoi(e);
#endif
    const TileShape ts = tile_shape(B, npass);
    a.ntiles = ts.ntiles; a.psplit = ts.small;
    bool vec = (d % 4 == 0) && aligned16(x);
    for (int p = 0; p < npass; ++p) {
        if (!mask[p] || !h1[p] || !h2[p] || !dmean[p] || !dlogvar[p]) return VPC_ERR_ARG;
        a.mask[p] = mask[p]; a.h1[p] = h1[p]; a.h2[p] = h2[p]; a.dmean[p] = dmean[p]; a.dlogvar[p] = dlogvar[p];
        vec = vec && ((uintptr_t)mask[p] % 4 == 0);
        if (!aligned16(h1[p]) || !aligned16(h2[p])) return VPC_ERR_ARG;
    }
    if (nblocks_out) *nblocks_out = ts.nblocks;
    const int DT = dt_for(mask_augm ? 2 * d : d);
    hipStream_t s = (hipStream_t)stream;
    if (precision != PREC_F32) {
        if (!vec || mask_augm) return VPC_ERR_SHAPE;
#define VPC_CASE(T)                                                                                                              \
    case T:                                                                                                                      \
        if (ts.small)                                                                                                            \
            return precision == PREC_BF16X3                                                                                      \
                       ? launch(enc_bwd_kernel<T, true, false, 4, PREC_BF16X3>, a, ts.grid_x, ts.grid_y, 4, enc_bwd_lds(T, 4), s) \
                       : launch(enc_bwd_kernel<T, true, false, 4, PREC_BF16>, a, ts.grid_x, ts.grid_y, 4, enc_bwd_lds(T, 4), s); \
        return precision == PREC_BF16X3                                                                                          \
                   ? launch(enc_bwd_kernel<T, true, false, 8, PREC_BF16X3>, a, ts.grid_x, ts.grid_y, 8, enc_bwd_lds(T, 8), s)    \
                   : launch(enc_bwd_kernel<T, true, false, 8, PREC_BF16>, a, ts.grid_x, ts.grid_y, 8, enc_bwd_lds(T, 8), s);
        switch (DT) { VPC_CASE(1) VPC_CASE(2) VPC_CASE(4) VPC_CASE(8) }
#undef VPC_CASE
        return VPC_ERR_SHAPE;
    }
#define VPC_LAUNCH(T, NW)                                                                                                  \
    (mask_augm ? launch(enc_bwd_kernel<T, false, true, NW>, a, ts.grid_x, ts.grid_y, NW, enc_bwd_lds(T, NW), s)            \
     : vec     ? launch(enc_bwd_kernel<T, true, false, NW>, a, ts.grid_x, ts.grid_y, NW, enc_bwd_lds(T, NW), s)            \
               : launch(enc_bwd_kernel<T, false, false, NW>, a, ts.grid_x, ts.grid_y, NW, enc_bwd_lds(T, NW), s))
#define VPC_CASE(T) \
    case T: return ts.small ? VPC_LAUNCH(T, 4) : VPC_LAUNCH(T, 8);
    switch (DT) { VPC_CASE(1) VPC_CASE(2) VPC_CASE(4) VPC_CASE(8) }
#undef VPC_CASE
#undef VPC_LAUNCH
    return VPC_ERR_SHAPE;
}
