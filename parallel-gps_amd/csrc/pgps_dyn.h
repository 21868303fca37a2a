// pgps_dyn.h -- RCCL and roctx are loaded on first use, not linked: a single-GPU user (or one without torch) does not need
// either library to load libpgps.so, and a process that already carries a copy of RCCL -- torch.distributed's nccl backend
// bundles its own librccl.so.1 -- keeps exactly that one (RTLD_NOLOAD first: two copies in one process interpose each
// other's symbols).  Types come from the headers; no symbol of either library is referenced at link time.
#pragma once

#include <rccl/rccl.h>

#include <string>

namespace pgps {
namespace dyn {

struct Rccl {
    bool ok = false;
    std::string err;                            // why it is not there
    std::string path;                           // what was loaded (diagnostics: pgps_comm_library)
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
const Rccl& rccl();                             // thread-safe, loads once

// named ranges: no-ops when the roctx library is not in reach
void range_push(const char* name);
void range_pop();

}  // namespace dyn
}  // namespace pgps
