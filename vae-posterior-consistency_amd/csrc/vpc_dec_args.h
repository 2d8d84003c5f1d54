// Argument block and shared constants of the decoder kernels (vpc_dec.hip: 4 waves x 2 batch tiles, all three modes;
// vpc_dec8.hip: 8 waves x 1 batch tile, fused mode).
#pragma once
#include <cstdint>
#include "vpc_layout.h"

namespace vpc {

enum { MODE_FWD = 0, MODE_FUSED = 1, MODE_BWD = 2 };

struct DecArgs {
    const float* x;
    const float* img;
    const uint8_t* mA[2];
    const uint8_t* mB[2];
    float cA[2], cE[2];
    const float* mean[2];
    const float* logvar[2];
    const float* eps[2];
    const float* eps_ml;
    const float* z_in[2];
    const float* dxhat[2];
    float* xhat[2];
    float* dmean[2];
    float* dlogvar[2];
    float* dz[2];
    float* part;
    double* loss_part;
    float bq, bp, cr, wml, inv_B, x_logvar;
    long B;
    int d, L, npass, ntiles;
    int lp;   // row pitch of the [B][.] latent arrays: L (dense, API tensors) or 16 (padded workspaces)
    int psplit;  // 1: the passes are spread over blockIdx.y (small-batch shape), 0: every workgroup loops over them
    int dbg;  // ablation mask, only honoured by the diagnostic build (-DVPC_ABLATE); 0 in the product build
};

constexpr int DEC_CH = 64;  // batch rows per wgrad staging chunk

// VPC_DBG(bit) guards the ablation switches of the diagnostic build (tools/ablate.sh).  In the product build it
// is an always-false test of a value the optimiser cannot see through: the switches cost one s_cbranch each
// and, more importantly, cut the kernel's 17k-line straight-line body into scheduling regions.  hipcc's
// scheduler otherwise hoists loads across the whole body, overshoots the 512-register budget and spills
// (measured on MI355X: 326 us without the region cuts, 252 us with them).
#ifdef VPC_ABLATE
#define VPC_DBG(bit) ((a.dbg & (bit)) != 0)
// phase timing (diagnostic build only): accumulate s_memtime deltas per phase, printed by block 0 / thread 0
#define VPC_STAMP(i)                                        \
    do {                                                    \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        T[i] += t_ - tlast;                                 \
        tlast = t_;                                         \
    } while (0)
#else
#define VPC_STAMP(i) do {} while (0)
__device__ __forceinline__ int opaque_zero() {
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
}
#define VPC_DBG(bit) ((opaque_zero() & (bit)) != 0)
#endif
// scheduling-region cut with no other effect (a never-taken branch around an empty asm)
#define VPC_CUT()                                              \
    do {                                                       \
        if (VPC_DBG(0x4000)) asm volatile("s_nop 0");          \
    } while (0)




size_t dec8_lds(int DT, int prec);
int dec8_dispatch(const DecArgs& a, bool vec, int grid, int prec, hipStream_t s);

}  // namespace vpc
