// Dense affine layers of the MNAR path (REG_notMIWAE_v2 / notMIWAE_myversion, reference src/models/VAE.py:2343-2363,
// 2378-2397): fp32 MFMA-tiled GEMM + bias + activation, with LDS-staged, XOR-swizzled operand tiles.
//
//   forward   Y[m][n]  = act( sum_k X[m][k] W[n][k] + b[n] )                       (nn.Linear + ELU / Sigmoid / Hardtanh)
//   dgrad     dX[m][k] = ( sum_n dY~[m][n] W[n][k] ) * act'(Xout[m][k])            (autograd of the layer, data side)
//   wgrad     dW[n][k] = sum_m dY~[m][n] X[m][k],  db[n] = sum_m dY~[m][n]         (weight side; split over m, fixed-order
//                                                                                    reduction => deterministic)
// dY~ = dY * act'(Y) can be formed on load (head layers, whose dY arrives from the loss kernel un-gated).
//
// One kernel template, three modes.  The MFMA is v_mfma_f32_16x16x4_f32 used "transposed" as everywhere in this
// library: the D tile is [feature 4q+j][row c], so a lane owns 4 consecutive features of one row and the epilogue
// moves float4s.  A workgroup (8 waves, 4 x 2, each 32 x 64 outputs) computes a 128 x 128 output tile; the contraction runs in chunks of
// 64 through two 32 KB LDS tiles:
//     row-read tile   [128][64]  (operand indexed [output][contraction]): one ds_read_b128 feeds 4 MFMAs
//     transposed tile [64][128]  (operand indexed [contraction][output]): 4 ds_read_b32 feed 4 MFMAs
// both with the 16-byte slot index XOR-ed with (row & 15), which makes either read conflict-free
// (tools/lds_conflicts.py).  Global loads of chunk i+1 are in flight while chunk i is multiplied.
#include "vpc_abi_internal.h"
#include "vpc_device.h"
#include "vpc_bf16.h"
#include "../../include/vpc.h"

namespace vpc {

enum { LIN_FWD = 0, LIN_DGRAD = 1, LIN_WGRAD = 2 };
enum { ACT_NONE = 0, ACT_ELU = 1, ACT_SIGMOID_HARDTANH = 2, ACT_RELU = 3 };

struct LinArgs {
    const float* A; long lda;      // fwd: W [N][K]       dgrad: W [N][K]       wgrad: dY [M][N]
    const float* B; long ldb;      // fwd: X [M][K]       dgrad: dY [M][N]      wgrad: X  [M][K]
    const float* Yg; long ldy;     // optional: outputs of the layer, to gate dY on load (dgrad: B, wgrad: A)
    float* C; long ldc;            // fwd: Y [M][N]       dgrad: dX [M][K]      wgrad: partials [S][N][K]
    const float* bias;             // fwd
    const float* aux; long ldaux;  // dgrad: layer input (= previous layer's output) for act'
    float* bias_part;              // wgrad: [S][N]
    int M, N, K;
    int act, split;                // fwd: output activation; dgrad: activation of the PREVIOUS layer (for aux)
    int gate, gate_split;          // activation of THIS layer when Yg is given
    int rows_per_split;            // wgrad
    int vecA, vecB, vecC, vecX;    // 16-byte path usable for the operand / output / aux
};

// The activation code is a kernel ARGUMENT; it is dispatched ONCE per tile (act_dispatch) into code specialised on it,
// never per element: a per-element `switch` costs a branch forest per value (the first build of the epilogue spent as
// long in it as in the MFMAs of a K = 128 layer).
template <int ACT>
__device__ __forceinline__ float act_fwd(float v, bool second) {
    // hardware exp / rcp (v_exp_f32, v_rcp_f32): |abs err| < 2e-7 on outputs in (-1, 1)
    if (ACT == ACT_ELU) return v > 0.f ? v : __expf(v) - 1.f;
    if (ACT == ACT_SIGMOID_HARDTANH) return second ? __builtin_amdgcn_fmed3f(v, -10.f, 0.f) : __builtin_amdgcn_rcpf(1.f + __expf(-v));
    if (ACT == ACT_RELU) return fmaxf(v, 0.f);
    return v;
}
// derivative of the activation expressed through its OUTPUT y
template <int ACT>
__device__ __forceinline__ float act_grad(float y, bool second) {
    if (ACT == ACT_ELU) return y > 0.f ? 1.f : y + 1.f;
    if (ACT == ACT_SIGMOID_HARDTANH) return second ? ((y > -10.f && y < 0.f) ? 1.f : 0.f) : y * (1.f - y);
    if (ACT == ACT_RELU) return y > 0.f ? 1.f : 0.f;
    return 1.f;
}
template <int V> struct ActC { static constexpr int value = V; };
template <typename F>
__device__ __forceinline__ void act_dispatch(int act, F&& f) {
    switch (act) {
        case ACT_ELU: f(ActC<ACT_ELU>{}); break;
        case ACT_SIGMOID_HARDTANH: f(ActC<ACT_SIGMOID_HARDTANH>{}); break;
        case ACT_RELU: f(ActC<ACT_RELU>{}); break;
        default: f(ActC<ACT_NONE>{}); break;
    }
}

constexpr int LIN_THREADS = 512;  // 8 waves: 4 (A-operand index) x 2 (B-operand index); a wave owns 32 x 64 outputs
constexpr int LIN_TILE = 8192;    // floats per (full) LDS tile
constexpr int IT = 2;             // 16 x 16 MFMA tiles per wave along the A-operand index
// JT = tiles per wave along the B-operand index: 4 (128 B-rows per workgroup) or, for the forward / dgrad of SMALL
// batches, 1 (32 batch rows per workgroup, 4x the workgroups: at the reference's batch 128 x K 20 a 128-row tiling
// would occupy 40 of the 256 CUs)

// ---- global -> registers: a ROWS x COLS tile (ROWS * COLS / 4 / 512 float4 per thread), zero-filled outside
// [rlim) x [clim).  Interior tiles with 16-byte access take a path without any per-load predicate; the gate (dY *= act'(Y),
// head layers on the API path) is applied in a second, activation-specialised sweep.
template <int COLS>
__device__ __forceinline__ f32x4 gload_one(int i, bool interior, const float* __restrict__ base, long ld, int r0, int c0,
                                           int rlim, int clim, int vec) {
    constexpr int C4 = COLS / 4;
    const int idx = threadIdx.x + LIN_THREADS * i;
    const int row = r0 + idx / C4, col = c0 + 4 * (idx % C4);
    const float* p = base + (long)row * ld + col;
    if (interior) return *reinterpret_cast<const f32x4*>(p);
    f32x4 v = zero4();
    if (row < rlim && col < clim) {
        if (vec && col + 3 < clim) {
            v = *reinterpret_cast<const f32x4*>(p);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < clim) v[e] = p[e];
        }
    }
    return v;
}
template <int COLS, int NLD>
__device__ __forceinline__ void gload(f32x4 (&r)[NLD], const float* __restrict__ base, long ld, int r0, int c0,
                                      int rlim, int clim, int vec, const float* __restrict__ yg, long ldy, int gate,
                                      int gate_split) {
    constexpr int C4 = COLS / 4;
    constexpr int ROWS = NLD * LIN_THREADS / C4;
    const bool interior = vec && r0 + ROWS <= rlim && c0 + COLS <= clim;  // workgroup-uniform
#pragma unroll
    for (int i = 0; i < NLD; ++i) r[i] = gload_one<COLS>(i, interior, base, ld, r0, c0, rlim, clim, vec);
    if (!yg) return;
    act_dispatch(gate, [&](auto actc) {
        constexpr int G = decltype(actc)::value;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const f32x4 y = gload_one<COLS>(i, interior, yg, ldy, r0, c0, rlim, clim, vec);
            const int col = c0 + 4 * ((threadIdx.x + LIN_THREADS * i) % C4);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[i][e] *= act_grad<G>(y[e], col + e >= gate_split);
        }
    });
}
// ---- registers -> swizzled LDS tile
template <int COLS, int NLD>
__device__ __forceinline__ void sstore(float* __restrict__ s, const f32x4 (&r)[NLD]) {
    constexpr int C4 = COLS / 4;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = threadIdx.x + LIN_THREADS * i;
        const int row = idx / C4, sl = idx % C4;
        *reinterpret_cast<f32x4*>(s + row * COLS + ((sl ^ (row & 15)) << 2)) = r[i];
    }
}
// A/B fragments for output block `ob` (16 outputs) and contraction sub-block kk (16 values): f[j] feeds MFMA k-step j
__device__ __forceinline__ f32x4 frag_row(const float* __restrict__ s, int ob, int kk, int c, int q) {
    return *reinterpret_cast<const f32x4*>(s + (16 * ob + c) * 64 + (((4 * kk + q) ^ c) << 2));
}
__device__ __forceinline__ f32x4 frag_T(const float* __restrict__ s, int ob, int kk, int c, int q) {
    f32x4 f;
    const int cs = 4 * ob + (c >> 2), cl = c & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 4 * q + j;
        f[j] = s[(16 * kk + r) * 128 + (((cs ^ r) << 2) | cl)];
    }
    return f;
}

// RAGGED: some 16-wide tile of the layer is empty (L = 10, 2L = 20, d = 14 ...): skip those MFMAs (wave-uniform).
// PREC (vpc_bf16.h): PREC_F32 = v_mfma_f32_16x16x4_f32.  PREC_BF16X3 / PREC_BF16: the same fp32 LDS tiles and fragment
// reads; two 16-wide contraction sub-blocks are converted to one bf16 operand in registers (both operands go through the
// same fragment functions, so any k order is consistent) and multiplied on v_mfma_f32_16x16x32_bf16 - one MFMA (three for
// the split form) where the fp32 path issues eight.  Zero fill outside the matrices makes padded sub-blocks harmless.
template <int MODE, bool RAGGED, int JT, int PREC = PREC_F32>
__global__ __launch_bounds__(LIN_THREADS, 2) void linear_kernel(LinArgs a) {
    static_assert(JT == 4 || MODE != LIN_WGRAD, "the narrow tiling is for forward / dgrad only");
    constexpr int BROWS = 32 * JT;                          // B-operand rows (batch rows) per workgroup
    constexpr int NLA = LIN_TILE / 4 / LIN_THREADS;         // float4 loads per thread: A tile (4)
    constexpr int NLB = BROWS * 64 / 4 / LIN_THREADS;       //                          B tile (4 or 1)
    extern __shared__ __align__(16) float lds[];
    float* sA = lds;
    float* sB = lds + LIN_TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    constexpr bool AT = MODE != LIN_FWD;    // A operand read transposed
    constexpr bool BT = MODE == LIN_WGRAD;  // B operand read transposed

    // output tile origin (i0 along the A-operand index, j0 along the B-operand index) and contraction range
    int i0, j0, k_begin, k_end, ilim, jlim;
    if (MODE == LIN_FWD) {
        j0 = blockIdx.x * BROWS; i0 = blockIdx.y * 128; k_begin = 0; k_end = a.K; ilim = a.N; jlim = a.M;
    } else if (MODE == LIN_DGRAD) {
        j0 = blockIdx.x * BROWS; i0 = blockIdx.y * 128; k_begin = 0; k_end = a.N; ilim = a.K; jlim = a.M;
    } else {
        i0 = blockIdx.y * 128; j0 = blockIdx.z * 128; ilim = a.N; jlim = a.K;
        k_begin = blockIdx.x * a.rows_per_split;
        k_end = min(a.M, k_begin + a.rows_per_split);
    }
    const int n_it = RAGGED ? min(IT, max(0, (ilim - i0 - 16 * IT * wr + 15) / 16)) : IT;
    const int n_jt = RAGGED ? min(JT, max(0, (jlim - j0 - 16 * JT * wc + 15) / 16)) : JT;

    f32x4 acc[IT][JT];
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[i][j] = zero4();
    float bsum = 0.f;  // wgrad: column sum of dY~ (bias gradient), threads 0..127 of the k-tile-0 workgroups

    f32x4 ra[NLA], rb[NLB];
    auto load_chunk = [&](int k0) {
        if (MODE == LIN_FWD) {
            gload<64, NLA>(ra, a.A, a.lda, i0, k0, a.N, k_end, a.vecA, nullptr, 0, 0, 0);
            gload<64, NLB>(rb, a.B, a.ldb, j0, k0, a.M, k_end, a.vecB, nullptr, 0, 0, 0);
        } else if (MODE == LIN_DGRAD) {
            gload<128, NLA>(ra, a.A, a.lda, k0, i0, k_end, a.K, a.vecA, nullptr, 0, 0, 0);
            gload<64, NLB>(rb, a.B, a.ldb, j0, k0, a.M, k_end, a.vecB, a.Yg, a.ldy, a.gate, a.gate_split);
        } else {
            gload<128, NLA>(ra, a.A, a.lda, k0, i0, k_end, a.N, a.vecA, a.Yg, a.ldy, a.gate, a.gate_split);
            gload<128, NLB>(rb, a.B, a.ldb, k0, j0, k_end, a.K, a.vecB, nullptr, 0, 0, 0);
        }
    };

    if (k_begin < k_end) load_chunk(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += 64) {
        __syncthreads();
        sstore<AT ? 128 : 64, NLA>(sA, ra);
        sstore<BT ? 128 : 64, NLB>(sB, rb);
        __syncthreads();
        if (k0 + 64 < k_end) load_chunk(k0 + 64);
        if (MODE == LIN_WGRAD && blockIdx.z == 0 && threadIdx.x < 128) {
            const int col = threadIdx.x, cs = col >> 2, cl = col & 3;
#pragma unroll 8
            for (int r = 0; r < 64; ++r) bsum += sA[r * 128 + (((cs ^ (r & 15)) << 2) | cl)];
        }
        const int n_kk = RAGGED ? min(4, (k_end - k0 + 15) / 16) : 4;  // contraction sub-blocks with real data
        if (RAGGED && (n_it == 0 || n_jt == 0)) continue;
        if (PREC != PREC_F32) {
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                if (RAGGED && 2 * kp >= n_kk) break;
                BfOp fa[IT], fb[JT];
#pragma unroll
                for (int t = 0; t < IT; ++t) {
                    if (RAGGED && t >= n_it) continue;
                    fa[t] = AT ? bf_pack<PREC>(frag_T(sA, IT * wr + t, 2 * kp, c, q), frag_T(sA, IT * wr + t, 2 * kp + 1, c, q))
                               : bf_pack<PREC>(frag_row(sA, IT * wr + t, 2 * kp, c, q), frag_row(sA, IT * wr + t, 2 * kp + 1, c, q));
                }
#pragma unroll
                for (int t = 0; t < JT; ++t) {
                    if (RAGGED && t >= n_jt) continue;
                    fb[t] = BT ? bf_pack<PREC>(frag_T(sB, JT * wc + t, 2 * kp, c, q), frag_T(sB, JT * wc + t, 2 * kp + 1, c, q))
                               : bf_pack<PREC>(frag_row(sB, JT * wc + t, 2 * kp, c, q), frag_row(sB, JT * wc + t, 2 * kp + 1, c, q));
                }
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    if (RAGGED && it >= n_it) break;
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        if (RAGGED && jt >= n_jt) break;
                        acc[it][jt] = bf_mma<PREC>(fa[it], fb[jt], acc[it][jt]);
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (RAGGED && kk >= n_kk) break;
            f32x4 fa[IT], fb[JT];
#pragma unroll
            for (int t = 0; t < IT; ++t)
                fa[t] = (!RAGGED || t < n_it)
                            ? (AT ? frag_T(sA, IT * wr + t, kk, c, q) : frag_row(sA, IT * wr + t, kk, c, q)) : zero4();
#pragma unroll
            for (int t = 0; t < JT; ++t)
                fb[t] = (!RAGGED || t < n_jt)
                            ? (BT ? frag_T(sB, JT * wc + t, kk, c, q) : frag_row(sB, JT * wc + t, kk, c, q)) : zero4();
            if (!RAGGED) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int it = 0; it < IT; ++it)
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt) acc[it][jt] = VPC_MFMA(fa[it][j], fb[jt][j], acc[it][jt]);
            } else {
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    if (it >= n_it) break;
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        if (jt >= n_jt) break;
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[it][jt] = VPC_MFMA(fa[it][j], fb[jt][j], acc[it][jt]);
                    }
                }
            }
        }
    }

    // ---- epilogue.  acc[it][jt][e] = D[i0 + 32 wr + 16 it + 4 q + e][j0 + 64 wc + 16 jt + c]
    if (MODE == LIN_WGRAD) {
        float* P = a.C + (long)blockIdx.x * a.N * a.ldc;
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const int col = j0 + 16 * JT * wc + 16 * jt + c;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = i0 + 16 * IT * wr + 16 * it + 4 * q + e;
                    if (row < ilim && col < jlim) P[(long)row * a.ldc + col] = acc[it][jt][e];
                }
            }
        if (blockIdx.z == 0 && threadIdx.x < 128 && i0 + (int)threadIdx.x < a.N)
            a.bias_part[(long)blockIdx.x * a.N + i0 + threadIdx.x] = bsum;
        return;
    }
    act_dispatch(a.act, [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int f = i0 + 16 * IT * wr + 16 * it + 4 * q;  // first of 4 consecutive output features
            if (f >= ilim) continue;
            const bool full4 = f + 3 < ilim;
            f32x4 bv = zero4();
            if (MODE == LIN_FWD && a.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (f + e < ilim) bv[e] = a.bias[f + e];
            }
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const int row = j0 + 16 * JT * wc + 16 * jt + c;
                if (row >= jlim) continue;
                f32x4 v = acc[it][jt];
                if (MODE == LIN_FWD) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fwd<ACT>(v[e] + bv[e], f + e >= a.split);
                } else if (a.aux) {
                    const float* px = a.aux + (long)row * a.ldaux + f;
                    f32x4 xo = zero4();
                    if (a.vecX && full4) {
                        xo = *reinterpret_cast<const f32x4*>(px);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (f + e < ilim) xo[e] = px[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= act_grad<ACT>(xo[e], f + e >= a.split);
                }
                float* pc = a.C + (long)row * a.ldc + f;
                if (a.vecC && full4) {
                    *reinterpret_cast<f32x4*>(pc) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (f + e < ilim) pc[e] = v[e];
                }
            }
        }
    });
}

// sum the per-split partial blocks (sum_partials_16x16: fixed association, deterministic); accumulate != 0 adds to dW / db
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias_part, int S,
                                                           int N, int K, float* __restrict__ dW, float* __restrict__ db,
                                                           int accumulate) {
    __shared__ float sh[16][16];
    const long n_w = (long)N * K;
    const long i = (long)blockIdx.x * 16 + (threadIdx.x & 15);
    const bool valid = i < n_w + N, is_w = i < n_w;
    const float* src = is_w ? part + i : bias_part + (i - n_w);
    const float s = sum_partials_16x16(valid ? src : part, is_w ? n_w : N, S, valid, sh);
    if (threadIdx.x >= 16 || !valid) return;
    float* dst = is_w ? dW + i : (db ? db + (i - n_w) : nullptr);
    if (dst) *dst = accumulate ? *dst + s : s;
}

// the same for up to WGRAD_MANY layers in ONE launch (the fused MNAR step defers its six reductions to the end of the
// backward pass: at the reference's batch sizes every launch is a dependent ~7 us dispatch)
constexpr int WGRAD_MANY = 8;
struct WgradMany {
    const float* part[WGRAD_MANY]; float* dW[WGRAD_MANY]; float* db[WGRAD_MANY];
    int S[WGRAD_MANY], N[WGRAD_MANY], K[WGRAD_MANY], accumulate[WGRAD_MANY], first_block[WGRAD_MANY + 1];
    int n;
};
__global__ __launch_bounds__(256) void wgrad_reduce_many_kernel(WgradMany a) {
    __shared__ float sh[16][16];
    int l = 0;
#pragma unroll
    for (int i = 1; i < WGRAD_MANY; ++i)
        if (i < a.n && (int)blockIdx.x >= a.first_block[i]) l = i;
    const long n_w = (long)a.N[l] * a.K[l];
    const long i = (long)(blockIdx.x - a.first_block[l]) * 16 + (threadIdx.x & 15);
    const bool valid = i < n_w + a.N[l], is_w = i < n_w;
    const float* src = is_w ? a.part[l] + i : a.part[l] + (long)a.S[l] * n_w + (i - n_w);
    // (the association of wgrad_reduce_kernel: bit-identical results)
    const float s = sum_partials_16x16(valid ? src : a.part[l], is_w ? n_w : a.N[l], a.S[l], valid, sh);
    if (threadIdx.x >= 16 || !valid) return;
    float* dst = is_w ? a.dW[l] + i : (a.db[l] ? a.db[l] + (i - n_w) : nullptr);
    if (dst) *dst = a.accumulate[l] ? *dst + s : s;
}

static bool vec_ok(const void* p, long ld) { return aligned16(p) && (ld % 4) == 0; }

template <int MODE, bool RAGGED, int JT, int PREC>
static int launch_prec(const LinArgs& a, dim3 grid, hipStream_t st) {
    constexpr size_t LDS = (LIN_TILE + (MODE == LIN_WGRAD ? LIN_TILE : 32 * JT * 64)) * sizeof(float);
    if (!lds_attr_done(reinterpret_cast<const void*>(&linear_kernel<MODE, RAGGED, JT, PREC>), LDS)) return VPC_ERR_HIP;
    hipLaunchKernelGGL((linear_kernel<MODE, RAGGED, JT, PREC>), grid, dim3(LIN_THREADS), LDS, st, a);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}
template <int MODE, bool RAGGED, int JT>
static int launch_one(const LinArgs& a, dim3 grid, int prec, hipStream_t st) {
    if (prec == PREC_BF16X3) return launch_prec<MODE, RAGGED, JT, PREC_BF16X3>(a, grid, st);
    if (prec == PREC_BF16) return launch_prec<MODE, RAGGED, JT, PREC_BF16>(a, grid, st);
    return launch_prec<MODE, RAGGED, JT, PREC_F32>(a, grid, st);
}
// forward / dgrad: grid.x over batch rows (128 or 32 per workgroup), grid.y over 128 output features
template <int MODE>
static int launch_rows(const LinArgs& a, int out_features, int prec, hipStream_t st) {
    // feature dimensions that leave whole 16-wide tiles / contraction sub-blocks empty take the tile-skipping build
    const bool ragged = (a.N % 64) != 0 || (a.K % 64) != 0;
    const unsigned gy = (unsigned)((out_features + 127) / 128);
    const bool narrow = (long)((a.M + 127) / 128) * gy < 2L * num_cus();
    const unsigned gx = (unsigned)((a.M + (narrow ? 31 : 127)) / (narrow ? 32 : 128));
    if (narrow)
        return ragged ? launch_one<MODE, true, 1>(a, dim3(gx, gy), prec, st) : launch_one<MODE, false, 1>(a, dim3(gx, gy), prec, st);
    return ragged ? launch_one<MODE, true, 4>(a, dim3(gx, gy), prec, st) : launch_one<MODE, false, 4>(a, dim3(gx, gy), prec, st);
}

}  // namespace vpc

using namespace vpc;

extern "C" {

int vpc_linear_fwd(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, long M, int N,
                   int K, int act, int act_split, int precision, void* stream) {
    if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldy < N || M > 0x7fffff00L) return VPC_ERR_ARG;
    if (precision < 0 || precision > 2) return VPC_ERR_ARG;
    if (act < ACT_NONE || act > ACT_RELU) return VPC_ERR_ARG;
    LinArgs a{};
    a.A = w; a.lda = K; a.B = x; a.ldb = ldx; a.C = y; a.ldc = ldy; a.bias = bias;
    a.M = (int)M; a.N = N; a.K = K; a.act = act; a.split = act == ACT_SIGMOID_HARDTANH ? act_split : N;
    a.vecA = vec_ok(w, K); a.vecB = vec_ok(x, ldx); a.vecC = vec_ok(y, ldy);
    return launch_rows<LIN_FWD>(a, N, precision, (hipStream_t)stream);
}

int vpc_linear_dgrad(const float* dy, long lddy, const float* y_gate, long ldyg, int gate, int gate_split,
                     const float* w, const float* x_out, long ldx, int act_prev, float* dx, long lddx, long M, int N,
                     int K, int precision, void* stream) {
    if (!dy || !w || !dx || M <= 0 || N <= 0 || K <= 0 || lddy < N || lddx < K || M > 0x7fffff00L) return VPC_ERR_ARG;
    if (precision < 0 || precision > 2) return VPC_ERR_ARG;
    if ((y_gate && ldyg < N) || (x_out && ldx < K)) return VPC_ERR_ARG;
    LinArgs a{};
    a.A = w; a.lda = K; a.B = dy; a.ldb = lddy; a.Yg = y_gate; a.ldy = ldyg; a.gate = gate;
    a.gate_split = gate == ACT_SIGMOID_HARDTANH ? gate_split : N;
    a.C = dx; a.ldc = lddx; a.aux = x_out; a.ldaux = ldx; a.act = act_prev; a.split = K;
    a.M = (int)M; a.N = N; a.K = K;
    a.vecA = vec_ok(w, K); a.vecB = vec_ok(dy, lddy) && (!y_gate || vec_ok(y_gate, ldyg)); a.vecC = vec_ok(dx, lddx);
    a.vecX = x_out && vec_ok(x_out, ldx);
    return launch_rows<LIN_DGRAD>(a, K, precision, (hipStream_t)stream);
}

long vpc_linear_wgrad_scratch(long M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const long chunks = (M + 63) / 64;
    const long tiles = (long)((N + 127) / 128) * ((K + 127) / 128);
    long S = (2L * num_cus() + tiles - 1) / tiles;  // ~2 workgroups per CU in total
    if (S > chunks) S = chunks;
    if (S < 1) S = 1;
    return S * ((long)N * K + N);
}

int vpc_linear_wgrad(const float* dy, long lddy, const float* y_gate, long ldyg, int gate, int gate_split,
                     const float* x, long ldx, float* dw, float* db, float* scratch, long scratch_floats, long M, int N,
                     int K, int accumulate, int precision, void* stream) {
    if (!dy || !x || !scratch || M <= 0 || N <= 0 || K <= 0 || lddy < N || ldx < K || M > 0x7fffff00L)
        return VPC_ERR_ARG;
    if (precision < 0 || precision > 2) return VPC_ERR_ARG;
    if (y_gate && ldyg < N) return VPC_ERR_ARG;
    const long need = vpc_linear_wgrad_scratch(M, N, K);
    if (scratch_floats < need) return VPC_ERR_ARG;
    const long S = need / ((long)N * K + N);
    const long chunks = (M + 63) / 64;
    const long rows_per_split = ((chunks + S - 1) / S) * 64;
    LinArgs a{};
    a.A = dy; a.lda = lddy; a.Yg = y_gate; a.ldy = ldyg; a.gate = gate;
    a.gate_split = gate == ACT_SIGMOID_HARDTANH ? gate_split : N;
    a.B = x; a.ldb = ldx; a.C = scratch; a.ldc = K; a.bias_part = scratch + S * (long)N * K;
    a.M = (int)M; a.N = N; a.K = K; a.rows_per_split = (int)rows_per_split;
    a.vecA = vec_ok(dy, lddy) && (!y_gate || vec_ok(y_gate, ldyg)); a.vecB = vec_ok(x, ldx);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)S, (unsigned)((N + 127) / 128), (unsigned)((K + 127) / 128));
    const bool ragged = (N % 64) != 0 || (K % 64) != 0;
    int rc = ragged ? launch_one<LIN_WGRAD, true, 4>(a, grid, precision, st) : launch_one<LIN_WGRAD, false, 4>(a, grid, precision, st);
    if (rc != VPC_OK || !dw) return rc;  // dw == NULL: partials only, summed later by vpc_linear_wgrad_reduce
    const long n = (long)N * K + N;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, scratch,
                       a.bias_part, (int)S, N, K, dw, db, accumulate);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

int vpc_linear_wgrad_reduce(int n_layers, const float* const* scratch, const long* M, const int* N, const int* K,
                            float* const* dw, float* const* db, const int* accumulate, void* stream) {
    if (n_layers < 1 || n_layers > WGRAD_MANY || !scratch || !M || !N || !K || !dw || !accumulate) return VPC_ERR_ARG;
    WgradMany a{};
    a.n = n_layers;
    int blocks = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (!scratch[l] || !dw[l] || M[l] <= 0 || N[l] <= 0 || K[l] <= 0) return VPC_ERR_ARG;
        const long per = (long)N[l] * K[l] + N[l];
        a.part[l] = scratch[l]; a.dW[l] = dw[l]; a.db[l] = db ? db[l] : nullptr;
        a.S[l] = (int)(vpc_linear_wgrad_scratch(M[l], N[l], K[l]) / per);  // the split count vpc_linear_wgrad used
        a.N[l] = N[l]; a.K[l] = K[l]; a.accumulate[l] = accumulate[l];
        a.first_block[l] = blocks;
        blocks += (int)((per + 15) / 16);
    }
    a.first_block[n_layers] = blocks;
    hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? VPC_OK : VPC_ERR_HIP;
}

}  // extern "C"
