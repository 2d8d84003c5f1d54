// Adam update of ONE parameter, shared by the stand-alone Adam launch and the reductions that apply it to the gradient they have
// just finished (vpc_misc.hip, vpc_nmdec.hip): one rounding sequence wherever it is inlined.
#pragma once
#include <climits>
#include "vpc_device.h"
#include "vpc_bf16.h"

namespace vpc {

// optional optimiser update fused into the gradient reduction (Adam is elementwise: the thread that finishes
// gradient i owns parameter i); param == nullptr disables it
struct AdamFuse {
    float* param; float* m; float* v; const int* pack_idx; float* img;
    float lr, b1, b2, eps, bc1, bc2_sqrt;
    int bf16c;  // 1: (pack_idx, img) are the compact bf16 image tables of the whole-step kernel (vpc_step_build_indices_bf16)
};
__device__ __forceinline__ void adam_apply(const AdamFuse& A, int i, float g) {
#pragma clang fp contract(off)  // one rounding sequence wherever this is inlined (stand-alone Adam == fused Adam, bitwise)
    const float mi = A.b1 * A.m[i] + (1.f - A.b1) * g;
    const float vi = A.b2 * A.v[i] + (1.f - A.b2) * g * g;
    A.m[i] = mi;
    A.v[i] = vi;
    const float denom = sqrtf(vi) / A.bc2_sqrt + A.eps;
    const float pnew = A.param[i] - (A.lr / A.bc1) * (mi / denom);
    A.param[i] = pnew;
    if (A.pack_idx) {
        const int e = A.pack_idx[i];
        if (!A.bf16c) A.img[e] = pnew;
        else if (e == INT_MIN) {}                // not part of this image
        else if (e < 0) A.img[-(e + 1)] = pnew;  // values that stay fp32 (biases, the missingness model)
        else reinterpret_cast<unsigned short*>(A.img)[e] = (unsigned short)(pk_bf16(pnew, 0.f) & 0xffffu);
    }
}

}  // namespace vpc
