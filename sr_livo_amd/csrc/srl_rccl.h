// srl_rccl.h -- the ONE RCCL instance of the process, resolved at run time.
//
// libsrlivo_hip.so does not link librccl.  The first communicator call looks the RCCL entry points up with
// dlsym(RTLD_DEFAULT): if the host program has already loaded an RCCL (a PyTorch process carries torch/lib/librccl.so),
// that very instance is used -- never a second copy with its own version, its own bootstrap threads and its own idea of
// the devices.  Only when the process has none, librccl.so.1 is dlopen'ed (system search path, then /opt/rocm/lib).
// Round 1 linked /opt/rocm's RCCL 2.27.7 while the driver's bench process resolved torch's 2.26.6 by SONAME: built
// against one, running on the other.  A single-GPU program never touches RCCL at all.
#pragma once
#include <rccl/rccl.h>   // types and prototypes only

struct SrlRccl {
    ncclResult_t (*GetVersion)(int *);
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int *);      // optional (null when the instance lacks it): srl_comm_info
    int version;            // ncclGetVersion of the instance in use
    char origin[512];       // path of the shared object the entry points live in (dladdr)
    bool preloaded;         // true: an instance the process already had; false: dlopen'ed by this library
};
// nullptr when no RCCL can be found (srl_rccl_error() says why)
const SrlRccl *srl_rccl();
const char *srl_rccl_error();
// Resolve the entry points from this shared object instead (before the first communicator call; false afterwards): the test-only
// stand-in tests/fake_rccl/libfake_rccl.so, through srl_comm_set_library -- an explicit call, never an environment variable.
bool srl_rccl_set_library(const char *path);
