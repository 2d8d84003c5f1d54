// Pins the bf16 MFMA engine of csrc/vpc_bf16.h on the GPU: weight image layout (K permutation, swizzle), forward
// fragments, transposed fragments through ds_read_b64_tr_b16, activation packing - against a CPU double reference.
//   hipcc -O3 --offload-arch=gfx950 -I vae-posterior-consistency_amd/csrc tools/microbench/bf16_engine_test.hip -o tools/microbench/bf16_engine_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "vpc_bf16.h"
using namespace vpc;

constexpr int OUT = 64, IN = 128, KP = 128;  // one layer: Y^T[OUT][16 rows] = W[OUT][IN] X^T[IN][16 rows]

template <int PREC>
__global__ void k(const float* img, const float* xT /*[IN][16]*/, const float* dyT /*[OUT][16]*/, float* yT, float* dxT) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < OUT * KP; i += 64) lds[i] = img[i];
    __syncthreads();
    const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    f32x4 xin[IN / 16], dy[OUT / 16];
    for (int t = 0; t < IN / 16; ++t)
        for (int j = 0; j < 4; ++j) xin[t][j] = xT[(16 * t + 4 * q + j) * 16 + c];
    for (int t = 0; t < OUT / 16; ++t)
        for (int j = 0; j < 4; ++j) dy[t][j] = dyT[(16 * t + 4 * q + j) * 16 + c];
    BfOp xb[IN / 32], dyb[OUT / 32];
    bf_acts<PREC, IN / 16>(xin, xb);
    bf_acts<PREC, OUT / 16>(dy, dyb);
    for (int mt = 0; mt < OUT / 16; ++mt) {
        const f32x4 acc = bf_tile_fwd<PREC, IN / 32, KP>(lds, mt, xb, zero4(), c, q);
        for (int j = 0; j < 4; ++j) yT[(16 * mt + 4 * q + j) * 16 + c] = acc[j];
    }
    for (int mt = 0; mt < IN / 16; ++mt) {
        const f32x4 acc = bf_tile_T<PREC, OUT / 32, KP>(lds, mt, dyb, zero4(), lane);
        for (int j = 0; j < 4; ++j) dxT[(16 * mt + 4 * q + j) * 16 + c] = acc[j];
    }
}

static unsigned short f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    std::vector<float> W(OUT * IN), xT(IN * 16), dyT(OUT * 16);
    srand(1);
    for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
    for (auto& v : xT) v = rand() / (float)RAND_MAX;
    for (auto& v : dyT) v = rand() / (float)RAND_MAX - 0.5f;
    std::vector<unsigned short> img(OUT * KP * 2, 0);
    for (int r = 0; r < OUT; ++r)
        for (int f = 0; f < IN; ++f) {
            const int e = bf_elem(r, f, KP);
            const unsigned short hi = f2bf(W[r * IN + f]);
            img[e] = hi;
            img[e + 8] = f2bf(W[r * IN + f] - bf2f(hi));
        }
    float *dimg, *dx, *ddy, *dy_, *ddx;
    hipMalloc(&dimg, OUT * KP * 4); hipMalloc(&dx, IN * 16 * 4); hipMalloc(&ddy, OUT * 16 * 4);
    hipMalloc(&dy_, OUT * 16 * 4); hipMalloc(&ddx, IN * 16 * 4);
    hipMemcpy(dimg, img.data(), OUT * KP * 4, hipMemcpyHostToDevice);
    hipMemcpy(dx, xT.data(), IN * 16 * 4, hipMemcpyHostToDevice);
    hipMemcpy(ddy, dyT.data(), OUT * 16 * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int prec = 1; prec <= 2; ++prec) {
        if (prec == 1) hipLaunchKernelGGL(k<PREC_BF16X3>, dim3(1), dim3(64), OUT * KP * 4, 0, dimg, dx, ddy, dy_, ddx);
        else hipLaunchKernelGGL(k<PREC_BF16>, dim3(1), dim3(64), OUT * KP * 4, 0, dimg, dx, ddy, dy_, ddx);
        std::vector<float> y(OUT * 16), dxo(IN * 16);
        hipMemcpy(y.data(), dy_, OUT * 16 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(dxo.data(), ddx, IN * 16 * 4, hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0, s1 = 0, s2 = 0;
        for (int o = 0; o < OUT; ++o)
            for (int r = 0; r < 16; ++r) {
                double ref = 0;
                for (int f = 0; f < IN; ++f) ref += (double)W[o * IN + f] * xT[f * 16 + r];
                e1 = fmax(e1, fabs(ref - y[o * 16 + r])); s1 = fmax(s1, fabs(ref));
            }
        for (int f = 0; f < IN; ++f)
            for (int r = 0; r < 16; ++r) {
                double ref = 0;
                for (int o = 0; o < OUT; ++o) ref += (double)W[o * IN + f] * dyT[o * 16 + r];
                e2 = fmax(e2, fabs(ref - dxo[f * 16 + r])); s2 = fmax(s2, fabs(ref));
            }
        const double tol = prec == 1 ? 2e-5 : 1e-2;
        printf("prec %d: fwd max err %.3e (scale %.3e)  transposed max err %.3e (scale %.3e)  tol %.0e rel\n", prec, e1, s1, e2, s2, tol);
        if (e1 > tol * s1 || e2 > tol * s2) bad = 1;
    }
    printf(bad ? "FAIL\n" : "PASS\n");
    return bad;
}
