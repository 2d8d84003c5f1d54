// Layout of the packed weight images, staging buffers and gradient-partial blocks.
// Single source of truth shared by the kernels (device) and the index builders (host).
//
// Model (reference src/models/VAE.py:366-376): encoder d->100->50->2L, decoder L->50->100->d.
// Everything is tiled in 16-feature tiles, the M/N extent of v_mfma_f32_16x16x4_f32:
//   d   -> DT tiles (DT = 1, 2, 4 or 8, i.e. d <= 128)
//   100 -> 7 tiles (112),  50 -> 4 tiles (64),  2L -> 2 tiles (mean tile | logvar tile),  L -> 1 tile
//
// A weight image is W~[out_pad][S] fp32 with S = 64 or 128 dwords (16 for the tiny decoder layer 4) and the
// 16-byte slot index XOR-swizzled with row & min(15, S/4 - 1): phys_col = (((col>>2) ^ (row&15)) << 2) | (col&3)
// for S >= 64.  That makes both the
// forward A-fragment read (ds_read_b128, lane (m,q) -> row m, cols 16kt+4q..+3) and the transposed
// A-fragment read used by dgrad (ds_read_b32, lane (m,q) -> row 16kt+4q+j, col m) conflict-free
// (tools/lds_conflicts.py).
//
// Bias handling ("ones trick"): for layers 2..6 the bias is the extra input column `in_real` of W~ and the
// producing layer emits a constant 1 at that feature (a fake unit whose only non-zero weight is 1 on the
// previous constant), so forward needs no bias add and wgrad delivers db as column `in_real` of dW~.
// Layer 1 (input x has no spare column when d % 16 == 0) keeps an explicit bias vector b1[112]; b1[100] = 1
// seeds the constant chain.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define VPC_HD __host__ __device__
#else
#define VPC_HD
#endif

namespace vpc {

constexpr int H1 = 100, H2 = 50;       // hidden widths (hard-coded in the reference)
constexpr int H1T = 7, H2T = 4;        // tiles
constexpr int H1P = 112, H2P = 64;     // padded widths (also the row pitch of the h1/h2/g1/g2 workspaces)
// 4-wide MFMA k-steps that can hold non-zero data when a hidden layer is the K dimension: its real units plus the
// constant-1 unit of the bias chain; the k-steps behind them multiply exact zeros of the padding and are skipped
constexpr int NK1 = (H1 + 1 + 3) / 4, NK2 = (H2 + 1 + 3) / 4;  // 26 of 28, 13 of 16
// Position of hidden unit u (u == H: the constant-1 unit) inside the padded width.  One MFMA k-step (tile t, register j)
// covers the four positions 16 t + j + 4 q, q = 0..3 - NOT four neighbours - so the units of the last, partly filled
// tile are laid out j-major: 96, 100, 104, 108, 97 for H1 (k-steps j = 2, 3 of tile 6 hold only padding) and
// 48, 52, 56 for H2 (j = 1, 2, 3 of tile 3).  Only vpc_build_indices knows this; the kernels see opaque positions.
VPC_HD constexpr int pos1(int u) { return u < 96 ? u : (u < 100 ? 96 + 4 * (u - 96) : 97); }
VPC_HD constexpr int pos2(int u) { return u < 48 ? u : 48 + 4 * (u - 48); }
// bijection of 0..111 that extends pos1: indices past the constant unit go to the padding positions
VPC_HD inline int pos1_full(int f) {
    const int pad[H1P - H1 - 1] = {98, 99, 101, 102, 103, 105, 106, 107, 109, 110, 111};
    return f <= H1 ? pos1(f) : pad[f - H1 - 1];
}
constexpr int WAVES = 8;               // waves per workgroup
constexpr int THREADS = WAVES * 64;
constexpr int TILE_ROWS = WAVES * 16;  // batch rows per workgroup iteration
constexpr int MAX_D = 128, MAX_L = 15;

VPC_HD constexpr int dt_for(int d) { return d <= 16 ? 1 : d <= 32 ? 2 : d <= 64 ? 4 : 8; }
VPC_HD constexpr int s_for_tiles(int t) { return t > 4 ? 128 : 64; }
VPC_HD inline int swz(int col, int row, int S = 64) {
    return ((((col >> 2) ^ (row & 15 & (S / 4 - 1))) << 2) | (col & 3));
}

// ---- encoder image: [W1: 112 x S1][b1: 128][W2: 64 x 128][W3: 32 x 64]
struct EncImg {
    int DT, S1, oW1, ob1, oW2, oW3, total;
    VPC_HD explicit EncImg(int dt) {
        DT = dt; S1 = s_for_tiles(dt);
        oW1 = 0; ob1 = oW1 + H1P * S1; oW2 = ob1 + 128; oW3 = oW2 + H2P * 128; total = oW3 + 32 * 64;
    }
};
// ---- decoder image: [W4: 64 x 16][W5: 112 x 64][W6: 16*DT x 128]
constexpr int S4 = 16;
// (s4 = row pitch of W4 in dwords: S4 for the fp32 image, 32 for the bf16 image - one 32-wide MFMA k-block)
struct DecImg {
    int DT, oW4, oW5, oW6, total;
    VPC_HD explicit DecImg(int dt, int s4 = S4) {
        DT = dt; oW4 = 0; oW5 = oW4 + H2P * s4; oW6 = oW5 + H1P * 64; total = oW6 + 16 * dt * 128;
    }
};

// ---- gradient partial block written by each workgroup (floats).  Register r of lane l of wave w lives at
// (w * REGS + r) * 64 + l; a 16x16 dW tile in C layout: element (row 4q+j, col c) = reg j of lane 16q+c.
// encoder kernel (8 waves x 48 regs): dW1 tile (mt<7, nt=w) -> regs 4mt..; dW2 tile (mt<4, nt=w<7) -> 28+4mt..;
//                 dW3 tile (mt=w>>2 (<2), nt=w&3) -> 44..47;  db1[112] appended after the 8 wave blocks.
// decoder kernel (4 waves x 92 regs, one wave per SIMD, 2 batch tiles per wave):
//                 dW6 tile (mt = w + 4i, nt<7) -> regs 28i + 4nt..;  dW5 tile (mt = w + 4i < 7, nt<4) ->
//                 56 + 16i + 4nt..;  dW4 tile (mt = w) -> 88..91
constexpr int GREGS = 48;
constexpr int ENC_PART = WAVES * GREGS * 64 + 128;    // + db1
constexpr int DEC_WAVES = 4, DEC_NB = 2, DEC_THREADS = DEC_WAVES * 64;
constexpr int DEC_GREGS = 92;
constexpr int DEC_PART = DEC_WAVES * DEC_GREGS * 64;  // 23552 floats
constexpr int LOSS_TERMS = 8;                          // doubles per workgroup

VPC_HD inline int part_off(int wave, int reg, int row_in_tile, int col_in_tile, int gregs = GREGS) {
    return (wave * gregs + reg + (row_in_tile & 3)) * 64 + ((row_in_tile >> 2) * 16 + col_in_tile);
}

// flat parameter order == state_dict order of the 12 trainable tensors
struct ParamOffsets {
    int w1, b1, w2, b2, w3, b3, w4, b4, w5, b5, w6, b6, n_enc, total;
    // d_in = width of the encoder input: d, or 2d for the mask-augmented variants ([x*mask | mask],
    // Reg_VAE_mask / vanilla_VAE_mask, src/models/VAE.py:545-548, 1031-1033)
    VPC_HD ParamOffsets(int d, int L, int d_in = 0) {
        if (d_in == 0) d_in = d;
        w1 = 0; b1 = w1 + H1 * d_in; w2 = b1 + H1; b2 = w2 + H2 * H1; w3 = b2 + H2; b3 = w3 + 2 * L * H2;
        n_enc = b3 + 2 * L;
        w4 = n_enc; b4 = w4 + H2 * L; w5 = b4 + H2; b5 = w5 + H1 * H2; w6 = b5 + H1; b6 = w6 + d * H1;
        total = b6 + d;
    }
};

}  // namespace vpc
