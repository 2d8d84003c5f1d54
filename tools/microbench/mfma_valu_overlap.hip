// Does VALU work of one wave overlap with MFMA work of another wave on the same SIMD (gfx950)?
// 8 waves per workgroup = 2 per SIMD, one workgroup per CU.  Waves 0-3 run chained fp32 MFMAs (two independent
// accumulators), waves 4-7 run dependent v_fma chains (or also MFMAs).  Times: MFMA only, VALU only, both.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE_A, int MODE_B>  // per half of the workgroup: 0 idle, 1 MFMA (2 chains), 2 VALU, 3 MFMA (1 chain)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int w = threadIdx.x >> 6;
    const int mode = w < 4 ? MODE_A : MODE_B;
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f, y = 1.0001f, v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3;
    if (mode == 1) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
            }
        }
    } else if (mode == 3) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        }
    } else if (mode == 2) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) {  // 256 VALU per iteration, 4 independent chains
                v0 = __builtin_fmaf(v0, y, x);
                v1 = __builtin_fmaf(v1, y, x);
                v2 = __builtin_fmaf(v2, y, x);
                v3 = __builtin_fmaf(v3, y, x);
            }
        }
    }
    else if (mode == 4) {  // integer VALU
        unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, m = blockIdx.x | 1;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                u0 = (u0 ^ m) + u1; u1 = (u1 ^ m) + u2; u2 = (u2 ^ m) + u3; u3 = (u3 ^ m) + u0;
            }
        }
        v0 = __uint_as_float((u0 ^ u1 ^ u2 ^ u3) & 0x3fffffffu);
    } else if (mode == 5) {  // transcendental unit
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                v0 = __builtin_amdgcn_exp2f(v0); v1 = __builtin_amdgcn_exp2f(v1);
                v2 = __builtin_amdgcn_exp2f(v2); v3 = __builtin_amdgcn_exp2f(v3);
            }
        }
    } else if (mode == 6) {  // LDS reads
        __shared__ float sh[512 * 4];
        sh[threadIdx.x] = x;
        const volatile float* p = sh;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) v0 += p[(threadIdx.x + u * 64) & 2047];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + v0 + v1 + v2 + v3;
}

template <int A, int B>
float run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int it = 2000;  // per wave: 64 000 MFMAs (32 cycles each) or 512 000 VALU (4 cycles each)
    printf("one wave/SIMD  MFMA 2 chains        : %8.1f us\n", run<1, 0>(d, it));
    printf("one wave/SIMD  MFMA 1 chain         : %8.1f us\n", run<3, 0>(d, it));
    printf("one wave/SIMD  VALU                 : %8.1f us\n", run<0, 2>(d, it));
    printf("two waves/SIMD MFMA(2ch) + VALU     : %8.1f us\n", run<1, 2>(d, it));
    printf("two waves/SIMD MFMA(1ch) + VALU     : %8.1f us\n", run<3, 2>(d, it));
    printf("two waves/SIMD MFMA(1ch) + MFMA(1ch): %8.1f us\n", run<3, 3>(d, it));
    printf("two waves/SIMD MFMA(2ch) + MFMA(2ch): %8.1f us\n", run<1, 1>(d, it));
    printf("two waves/SIMD VALU + VALU          : %8.1f us\n", run<2, 2>(d, it));
    printf("one wave/SIMD  int VALU             : %8.1f us\n", run<0, 4>(d, it));
    printf("two waves/SIMD MFMA(1ch) + int VALU : %8.1f us\n", run<3, 4>(d, it));
    printf("one wave/SIMD  v_exp_f32            : %8.1f us\n", run<0, 5>(d, it));
    printf("two waves/SIMD MFMA(1ch) + v_exp_f32: %8.1f us\n", run<3, 5>(d, it));
    printf("one wave/SIMD  LDS reads            : %8.1f us\n", run<0, 6>(d, it));
    printf("two waves/SIMD MFMA(1ch) + LDS reads: %8.1f us\n", run<3, 6>(d, it));
    return 0;
}
