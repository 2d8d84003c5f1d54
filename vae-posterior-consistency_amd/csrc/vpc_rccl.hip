// The step's single collective on the CALLER'S stream: ncclAllReduce (RCCL over xGMI) of the flat fp32 bucket
// [gradients | loss terms] (one 150 KB all-reduce per step, latency-bound).
//
// torch.distributed's NCCL backend runs collectives on an internal stream and synchronises it with the compute stream
// through events on both sides; with RCCL bound here the all-reduce is an ordinary node of the compute stream: no stream
// hop, and reduce_step -> all-reduce -> adam_step can be captured into ONE HIP graph.  librccl is opened lazily
// (dlopen), so libvpc_hip.so has no load-time dependency on it and single-GPU use never touches it.
// The reference has no counterpart (it has no distributed code at all).
#include "vpc_abi_internal.h"
#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace vpc {

constexpr int UID_BYTES = 128;  // NCCL_UNIQUE_ID_BYTES (rccl.h)
struct UniqueId { char internal[UID_BYTES]; };
typedef int (*get_uid_fn)(UniqueId*);
typedef int (*comm_init_fn)(void**, int, UniqueId, int);
typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*comm_destroy_fn)(void*);

struct Rccl {
    get_uid_fn get_uid = nullptr;
    comm_init_fn comm_init = nullptr;
    allreduce_fn allreduce = nullptr;
    comm_destroy_fn comm_destroy = nullptr;
    bool ok = false;
};
static Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        r.get_uid = (get_uid_fn)dlsym(h, "ncclGetUniqueId");
        r.comm_init = (comm_init_fn)dlsym(h, "ncclCommInitRank");
        r.allreduce = (allreduce_fn)dlsym(h, "ncclAllReduce");
        r.comm_destroy = (comm_destroy_fn)dlsym(h, "ncclCommDestroy");
        r.ok = r.get_uid && r.comm_init && r.allreduce && r.comm_destroy;
    });
    return r;
}

}  // namespace vpc

using namespace vpc;

extern "C" int vpc_rccl_unique_id(void* id128_host) {
    if (!id128_host) return VPC_ERR_ARG;
    if (!rccl().ok) return VPC_ERR_HIP;
    UniqueId id;
    if (rccl().get_uid(&id) != 0) return VPC_ERR_HIP;
    std::memcpy(id128_host, id.internal, UID_BYTES);
    return VPC_OK;
}

extern "C" int vpc_rccl_comm_init(const void* id128_host, int nranks, int rank, void** comm_out) {
    if (!id128_host || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) return VPC_ERR_ARG;
    if (!rccl().ok) return VPC_ERR_HIP;
    UniqueId id;
    std::memcpy(id.internal, id128_host, UID_BYTES);
    void* comm = nullptr;
    if (rccl().comm_init(&comm, nranks, id, rank) != 0 || !comm) return VPC_ERR_HIP;
    *comm_out = comm;
    return VPC_OK;
}

extern "C" int vpc_allreduce_flat(void* comm, float* bucket, long count, void* stream) {
    if (!comm || !bucket || count <= 0) return VPC_ERR_ARG;
    if (!rccl().ok) return VPC_ERR_HIP;
    // in place, ncclFloat32 (7), ncclSum (0), on the caller's stream
    return rccl().allreduce(bucket, bucket, (size_t)count, 7, 0, comm, (hipStream_t)stream) == 0 ? VPC_OK : VPC_ERR_HIP;
}

extern "C" int vpc_rccl_comm_destroy(void* comm) {
    if (!comm) return VPC_ERR_ARG;
    if (!rccl().ok) return VPC_ERR_HIP;
    return rccl().comm_destroy(comm) == 0 ? VPC_OK : VPC_ERR_HIP;
}
