// Library-owned gradient all-reduce over RCCL (xGMI on an MI355X node).  The reference has no collective at all
// (single device, pb_sed/experiments/weak_label_crnn/training.py:284); data-parallel training is new in the build
// (SURVEY.md 8(e)): one sum-all-reduce of the flat fp32 gradient per step, issued per bucket from inside backward.
//
// A communicator owns ONE extra HIP stream.  pbsed_allreduce_begin orders the collective after everything enqueued so
// far on the producer (compute) stream by an event, runs it on the communicator's stream - so it overlaps whatever the
// compute stream does next (the remaining conv dgrad / wgrad launches) - and records a done-event;
// pbsed_allreduce_finish makes a consumer stream wait for every collective begun since the last finish.  No host sync.
#include <rccl/rccl.h>

#include <cstring>
#include <vector>

#include "common.h"

namespace pbsed {

struct Comm {
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;             // producer stream -> comm stream
    std::vector<hipEvent_t> done;           // one per collective begun since the last finish (recycled)
    size_t in_flight = 0;
    int rank = 0, world = 1, device = 0;
};

#define PBSED_NCCL_TRY(expr, what)                                              \
    do {                                                                        \
        const ncclResult_t pbsed_r_ = (expr);                                   \
        if (pbsed_r_ != ncclSuccess) {                                          \
            set_error("%s: %s", what, ncclGetErrorString(pbsed_r_));            \
            return PBSED_E_HIP;                                                 \
        }                                                                       \
    } while (0)

}  // namespace pbsed

using namespace pbsed;

extern "C" {

int pbsed_comm_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int pbsed_comm_unique_id(void* out) {
    if (!out) { set_error("comm_unique_id: null"); return PBSED_E_ARG; }
    ncclUniqueId id;
    PBSED_NCCL_TRY(ncclGetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out, &id, sizeof(id));
    return PBSED_OK;
}

int pbsed_comm_create(const void* unique_id, int rank, int world, void** comm_out) {
    if (!unique_id || !comm_out || world < 1 || rank < 0 || rank >= world) { set_error("comm_create: bad arguments"); return PBSED_E_ARG; }
    Comm* c = new Comm();
    c->rank = rank; c->world = world;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    hipError_t e = hipGetDevice(&c->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
    if (e != hipSuccess) { set_error("comm_create: %s", hipGetErrorString(e)); delete c; return PBSED_E_HIP; }
    const ncclResult_t r = ncclCommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank: %s", ncclGetErrorString(r));
        hipEventDestroy(c->ready); hipStreamDestroy(c->stream); delete c;
        return PBSED_E_HIP;
    }
    *comm_out = c;
    return PBSED_OK;
}

int pbsed_comm_destroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return PBSED_OK;
    hipStreamSynchronize(c->stream);
    if (c->nccl) ncclCommDestroy(c->nccl);
    for (hipEvent_t ev : c->done) hipEventDestroy(ev);
    hipEventDestroy(c->ready);
    hipStreamDestroy(c->stream);
    delete c;
    return PBSED_OK;
}

// In-place sum over ranks of buf[0..n) (fp32), ordered after the work already enqueued on producer_stream.
int pbsed_allreduce_begin(void* comm, float* buf, size_t n, void* producer_stream) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c || !buf) { set_error("allreduce_begin: null argument"); return PBSED_E_ARG; }
    if (n == 0) return PBSED_OK;
    PBSED_HIP_TRY(hipEventRecord(c->ready, (hipStream_t)producer_stream), "hipEventRecord");
    PBSED_HIP_TRY(hipStreamWaitEvent(c->stream, c->ready, 0), "hipStreamWaitEvent");
    PBSED_NCCL_TRY(ncclAllReduce(buf, buf, n, ncclFloat32, ncclSum, c->nccl, c->stream), "ncclAllReduce");
    if (c->in_flight == c->done.size()) {
        hipEvent_t ev;
        PBSED_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
        c->done.push_back(ev);
    }
    PBSED_HIP_TRY(hipEventRecord(c->done[c->in_flight], c->stream), "hipEventRecord");
    ++c->in_flight;
    return PBSED_OK;
}

// consumer_stream waits (on the device) for every collective begun since the last finish.
int pbsed_allreduce_finish(void* comm, void* consumer_stream) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) { set_error("allreduce_finish: null communicator"); return PBSED_E_ARG; }
    for (size_t i = 0; i < c->in_flight; ++i)
        PBSED_HIP_TRY(hipStreamWaitEvent((hipStream_t)consumer_stream, c->done[i], 0), "hipStreamWaitEvent");
    c->in_flight = 0;
    return PBSED_OK;
}

}  // extern "C"
