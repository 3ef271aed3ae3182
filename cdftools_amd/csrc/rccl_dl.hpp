// rccl_dl.hpp -- RCCL for the sharded plans, looked up at run time.
//
// The library has no link dependency on librccl (a 570 MB shared object that every tool would map at start-up for the sake
// of one option): the first sharded plan made while the collective mode is CDFT_COLLECTIVE_RCCL opens it (dlopen of
// librccl.so.1, or the copy a host program such as PyTorch has already mapped) and takes the entry points it uses.
// One process drives all the devices of a ShardSet, so the communicators come from ncclCommInitAll; a collective is issued
// either from the per-block host threads of shard::for_blocks (one thread per device, no group needed) or, where one thread
// talks to two devices (the carry hop of cdfpsi), inside ncclGroupStart / ncclGroupEnd.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace cdft {
namespace rccl {

typedef void *comm_t;                                    // ncclComm_t
enum { kDouble = 8 };                                    // ncclDouble / ncclFloat64 (rccl.h)

struct Api {
    int (*CommInitAll)(comm_t *, int, const int *);
    int (*CommDestroy)(comm_t);
    int (*CommAbort)(comm_t);
    int (*AllGather)(const void *, void *, size_t, int, comm_t, hipStream_t);
    int (*Broadcast)(const void *, void *, size_t, int, int, comm_t, hipStream_t);
    int (*Send)(const void *, size_t, int, int, comm_t, hipStream_t);
    int (*Recv)(void *, size_t, int, int, comm_t, hipStream_t);
    int (*GroupStart)();
    int (*GroupEnd)();
    const char *(*GetErrorString)(int);
};

// the entry points, or nullptr with the reason in cdft_last_error()
const Api *api();

}  // namespace rccl
}  // namespace cdft

#define CDFT_RCCL(call)                                                                               \
    do {                                                                                              \
        int r_ = (call);                                                                              \
        if (r_ != 0)                                                                                  \
            return cdft::fail(CDFT_ERR_HIP, "%s:%d: %s -> RCCL: %s", __FILE__, __LINE__, #call,       \
                              cdft::rccl::api() ? cdft::rccl::api()->GetErrorString(r_) : "?");       \
    } while (0)
