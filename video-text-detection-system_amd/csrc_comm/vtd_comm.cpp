// libvtd_comm.so: the detections all-gather of the frame-sharded path over RCCL (include/vtd_comm.h).  Kept out of libvtd_hip.so
// so that the compute library carries no communication dependency; nothing here launches a kernel of its own.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstring>
#include <new>

#include "../../include/vtd_comm.h"

struct vtd_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

static int nccl_rc(ncclResult_t r) { return r == ncclSuccess ? 0 : -3000 - (int)r; }

extern "C" {

int vtd_comm_unique_id(void* id_out) {
    static_assert(sizeof(ncclUniqueId) <= VTD_COMM_ID_BYTES, "id size");
    if (!id_out) return -1;
    ncclUniqueId id;
    const int rc = nccl_rc(ncclGetUniqueId(&id));
    if (rc) return rc;
    std::memset(id_out, 0, VTD_COMM_ID_BYTES);
    std::memcpy(id_out, &id, sizeof id);
    return 0;
}

int vtd_comm_create(const void* unique_id, int rank, int world_size, vtd_comm** out) {
    if (!unique_id || !out || world_size < 1 || rank < 0 || rank >= world_size) return -1;
    vtd_comm* c = new (std::nothrow) vtd_comm;
    if (!c) return -2;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    const int rc = nccl_rc(ncclCommInitRank(&c->comm, world_size, id, rank));
    if (rc) {
        delete c;
        return rc;
    }
    c->rank = rank;
    c->world = world_size;
    *out = c;
    return 0;
}

int vtd_comm_rank(const vtd_comm* c) { return c ? c->rank : -1; }
int vtd_comm_world_size(const vtd_comm* c) { return c ? c->world : 0; }

int vtd_gather(vtd_comm* c, const int32_t* local_dev, int64_t count, int32_t* out_dev, void* stream) {
    if (!c || !local_dev || !out_dev || count <= 0) return -1;
    return nccl_rc(ncclAllGather(local_dev, out_dev, (size_t)count, ncclInt32, c->comm, (hipStream_t)stream));
}

void vtd_comm_destroy(vtd_comm* c) {
    if (!c) return;
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
}

}  // extern "C"
