// Shared by the translation units behind the C ABI (include/vpc.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
// every translation unit that defines a vpc_* entry point sees its declaration: hipcc rejects a definition whose
// signature drifted from the header (conflicting types for an extern "C" function)
#include "../../include/vpc.h"

#define VPC_OK 0
#define VPC_ERR_ARG 1     // null / misaligned pointer, bad count
#define VPC_ERR_SHAPE 2   // unsupported d / L
#define VPC_ERR_HIP 3     // HIP runtime reported an error

namespace vpc {
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
int num_cus();  // CUs of the current device (cached)
// Workgroup shape of the row-tiled kernels (encoder forward / backward, decoder) for a batch of B rows and npass passes.
//   throughput shape: 128-row tiles (8 waves x 16 rows, or 4 waves x 2 x 16), every workgroup loops over its tiles and
//                     over the passes; grid = min(tiles, CUs)
//   small-batch shape: 64-row tiles (4 waves x 16 rows, ONE wave per SIMD), the passes spread over blockIdx.y - chosen
//                     when all (tile, pass) pairs fit in two rounds of workgroups (B <= 16 384 for two passes on 256
//                     CUs): the step is then bound by the serial latency of one tile-pass per wave, and this shape
//                     gives every SIMD one tile-pass instead of giving a quarter of the CUs four.
// VPC_TILE=64 / 128 in the environment forces a shape (A/B runs, tests).  nblocks <= 2 * num_cus() always.
struct TileShape { int small, ntiles, grid_x, grid_y, nblocks; };
TileShape tile_shape(long B, int npass, bool force_big = false);  // force_big: kernels that exist in the throughput shape only
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, size); false on a HIP error
bool lds_attr_done(const void* kern, size_t lds);
}  // namespace vpc
