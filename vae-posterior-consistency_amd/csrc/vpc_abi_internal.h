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
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, size); false on a HIP error
bool lds_attr_done(const void* kern, size_t lds);
}  // namespace vpc
