/* vtd_comm.h -- the one exchange step of the path as a C entry point (libvtd_comm.so), for hosts that bind include/vtd.h
 * without going through torch.distributed.
 *
 * The reference has no multi-GPU path; BASELINE.json's north star shards decoded frames over the GPUs of a node and gathers
 * the detections once ("RCCL over xGMI only for the final gather of detections").  The Python product does that gather through
 * torch.distributed (backend "nccl" = RCCL; vtd_amd/shard.py).  A C / C++ / Go / Java host does it here: every rank pushes its
 * frames through vtd_detector_* / vtd_postproc_run, which leaves a block of `vtd_detection` records (int32, include/vtd.h) plus
 * counts in device memory; vtd_gather all-gathers such a block -- padded to the same element count on every rank -- into
 * [world_size][count] on every rank, on the stream that produced it.
 *
 * One process per GPU, one communicator per process.  Rank 0 makes the id (vtd_comm_unique_id) and ships its 128 bytes to the
 * other ranks by whatever transport the host has (a file, a socket, MPI); every rank then calls vtd_comm_create after selecting
 * its device.  Status: 0 = ok, negative = -(ncclResult_t) - 3000 or a HIP error as in vtd.h.
 */
#ifndef VTD_COMM_H
#define VTD_COMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VTD_COMM_ID_BYTES 128

typedef struct vtd_comm vtd_comm;

int vtd_comm_unique_id(void* id_out /* VTD_COMM_ID_BYTES */);
int vtd_comm_create(const void* unique_id, int rank, int world_size, vtd_comm** out);
int vtd_comm_rank(const vtd_comm* c);
int vtd_comm_world_size(const vtd_comm* c);
/* all-gather of `count` int32 per rank: out_dev[r * count + i] = rank r's local_dev[i]; asynchronous on `stream` (a hipStream_t) */
int vtd_gather(vtd_comm* c, const int32_t* local_dev, int64_t count, int32_t* out_dev, void* stream);
void vtd_comm_destroy(vtd_comm* c);

#ifdef __cplusplus
}
#endif
#endif
