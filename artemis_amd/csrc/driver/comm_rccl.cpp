// Native inter-GPU transport of the host driver: artemis_comm_t on RCCL (xGMI inside a node).
//
// This is the C++ counterpart of Parthenon's MPI boundary communication for the path
// (AddBoundaryExchangeTasks at artemis_driver.cpp:258, the dt reduction of EstimateTimestep at
// :279-297, the history reductions of utils/history.hpp:29-100): one process per GPU, one RCCL
// communicator, everything stream-ordered.
//   exchange_start     one ncclGroupStart/End of ncclSend / ncclRecv per ghost exchange, posted on the
//                      stream the driver hands in (its comm stream), in tag order on both sides --
//                      RCCL matches point-to-point operations between two ranks by posting order
//   exchange_finish    nothing: unpacks queued on the same stream are ordered behind the group
//   allreduce_min_dev  ncclAllReduce(min) in place on the device scalar, on the caller's stream
//   allreduce_min/sum  host values through a small device scratch on a private stream (synchronous)
// No HIP calls here: device memory, copies and streams come from artemis_rt.h like everywhere else in
// the driver; RCCL only sees the stream handles.
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "artemis_driver.h"
#include "artemis_hip.h"
#include "artemis_rt.h"

namespace {

thread_local std::string g_comm_err;

struct RcclCtx {
  artemis_comm_t iface; // first member: the handle the driver sees
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  void *stream = nullptr;   // private stream for the host-value reductions
  double *scratch = nullptr; // device, kScratch doubles
  static constexpr int kScratch = 256;
  // ordering of the communicator's operations across streams: the last group / device reduction posted on a caller's
  // stream leaves an event; a host-value reduction makes the private stream wait for it before it posts its own
  void *last_op = nullptr;
  bool last_op_set = false;
  void posted_on(void *s) {
    if (last_op && artemis_rt_event_record(last_op, s) == 0) last_op_set = true;
  }
};

bool ok(ncclResult_t r, const char *what) {
  if (r == ncclSuccess) return true;
  g_comm_err = std::string(what) + ": " + ncclGetErrorString(r);
  return false;
}

int exchange_start(void *vctx, int nmsg, const artemis_msg_t *msgs, void *stream) {
  RcclCtx *c = static_cast<RcclCtx *>(vctx);
  // deterministic order on both sides of every pair: by tag, the receive of a tag before its send
  // (only a message to self carries both)
  std::vector<const artemis_msg_t *> order(nmsg);
  for (int q = 0; q < nmsg; ++q) order[q] = msgs + q;
  std::stable_sort(order.begin(), order.end(), [](const artemis_msg_t *a, const artemis_msg_t *b) {
    if (a->tag != b->tag) return a->tag < b->tag;
    return (a->recv != nullptr) > (b->recv != nullptr);
  });
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!ok(ncclGroupStart(), "ncclGroupStart")) return 1;
  bool good = true;
  for (const artemis_msg_t *m : order) {
    if (m->recv) good = good && ok(ncclRecv(m->recv, static_cast<size_t>(m->count), ncclDouble, m->peer, c->comm, s), "ncclRecv");
    if (m->send) good = good && ok(ncclSend(m->send, static_cast<size_t>(m->count), ncclDouble, m->peer, c->comm, s), "ncclSend");
  }
  const bool ended = ok(ncclGroupEnd(), "ncclGroupEnd");
  c->posted_on(stream);
  return (good && ended) ? 0 : 1;
}

int exchange_finish(void *, void *) { return 0; } // stream order does it

int allreduce_min_dev(void *vctx, double *dev_value, void *stream) {
  RcclCtx *c = static_cast<RcclCtx *>(vctx);
  const bool good = ok(ncclAllReduce(dev_value, dev_value, 1, ncclDouble, ncclMin, c->comm, static_cast<hipStream_t>(stream)),
                       "ncclAllReduce(min)");
  c->posted_on(stream);
  return good ? 0 : 1;
}

// Host-value reductions run on the context's private stream.  RCCL orders the operations of one communicator by issue
// order on every rank, whatever stream they are on; every caller (SetGlobalTimeStep's host min in the synchronising
// loop, history / error sums, the refinement tags, remesh) reaches the call at the same point of the cycle on every
// rank.  That the communicator's previous operation -- an exchange group or the device-resident dt reduction on one of
// the driver's streams -- is also BEHIND this one on the device is enforced here, not assumed: the private stream
// waits for the event that operation left (RcclCtx::posted_on) before the reduction is posted.
int host_allreduce(RcclCtx *c, double *values, int n, ncclRedOp_t op) {
  if (c->last_op_set) {
    if (artemis_rt_stream_wait_event(c->stream, c->last_op)) return 1;
    c->last_op_set = false;
  }
  for (int done = 0; done < n; done += RcclCtx::kScratch) {
    const int m = std::min(RcclCtx::kScratch, n - done);
    if (artemis_rt_memcpy_h2d(c->scratch, values + done, m * sizeof(double), c->stream)) return 1;
    if (!ok(ncclAllReduce(c->scratch, c->scratch, m, ncclDouble, op, c->comm, static_cast<hipStream_t>(c->stream)),
            "ncclAllReduce"))
      return 1;
    if (artemis_rt_memcpy_d2h(values + done, c->scratch, m * sizeof(double), c->stream)) return 1;
    if (artemis_rt_stream_sync(c->stream)) return 1;
  }
  return 0;
}
int allreduce_min(void *vctx, double *value) { return host_allreduce(static_cast<RcclCtx *>(vctx), value, 1, ncclMin); }
int allreduce_sum(void *vctx, double *values, int n) {
  return host_allreduce(static_cast<RcclCtx *>(vctx), values, n, ncclSum);
}

} // namespace

extern "C" {

const char *artemis_comm_rccl_last_error(void) { return g_comm_err.c_str(); }

int artemis_comm_rccl_unique_id(char *out, int capacity) {
  if (!out || capacity < static_cast<int>(sizeof(ncclUniqueId))) {
    g_comm_err = "unique id buffer too small";
    return 1;
  }
  ncclUniqueId id;
  if (!ok(ncclGetUniqueId(&id), "ncclGetUniqueId")) return 1;
  std::memcpy(out, &id, sizeof id);
  return 0;
}
int artemis_comm_rccl_unique_id_bytes(void) { return static_cast<int>(sizeof(ncclUniqueId)); }

artemis_comm_t *artemis_comm_rccl_create(const char *unique_id, int rank, int nranks) {
  if (!unique_id || nranks < 1 || rank < 0 || rank >= nranks) {
    g_comm_err = "bad arguments";
    return nullptr;
  }
  RcclCtx *c = new (std::nothrow) RcclCtx();
  if (!c) return nullptr;
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  if (!ok(ncclCommInitRank(&c->comm, nranks, id, rank), "ncclCommInitRank")) {
    delete c;
    return nullptr;
  }
  int count = 0;
  if (!ok(ncclCommCount(c->comm, &count), "ncclCommCount") || count != nranks) {
    if (count != nranks) g_comm_err = "communicator size differs from nranks";
    ncclCommDestroy(c->comm);
    delete c;
    return nullptr;
  }
  c->rank = rank, c->nranks = nranks;
  c->stream = artemis_rt_stream_create();
  c->scratch = static_cast<double *>(artemis_rt_malloc(RcclCtx::kScratch * sizeof(double)));
  c->last_op = artemis_rt_event_create();
  if (!c->stream || !c->scratch || !c->last_op) {
    g_comm_err = std::string("device resources: ") + artemis_hip_last_error();
    ncclCommDestroy(c->comm);
    if (c->scratch) artemis_rt_free(c->scratch); // whichever of them was obtained
    if (c->stream) artemis_rt_stream_destroy(c->stream);
    if (c->last_op) artemis_rt_event_destroy(c->last_op);
    delete c;
    return nullptr;
  }
  c->iface.ctx = c, c->iface.rank = rank, c->iface.nranks = nranks;
  c->iface.exchange_start = exchange_start, c->iface.exchange_finish = exchange_finish;
  c->iface.allreduce_min = allreduce_min, c->iface.allreduce_min_dev = allreduce_min_dev;
  c->iface.allreduce_sum = allreduce_sum;
  return &c->iface;
}

int artemis_comm_rccl_count(const artemis_comm_t *comm) {
  if (!comm || !comm->ctx) return 0;
  int count = 0;
  if (!ok(ncclCommCount(static_cast<RcclCtx *>(comm->ctx)->comm, &count), "ncclCommCount")) return 0;
  return count;
}

int artemis_comm_rccl_barrier(artemis_comm_t *comm) {
  if (!comm || !comm->ctx) return 1;
  double one = 1.0;
  return allreduce_sum(comm->ctx, &one, 1);
}

void artemis_comm_rccl_destroy(artemis_comm_t *comm) {
  if (!comm || !comm->ctx) return;
  RcclCtx *c = static_cast<RcclCtx *>(comm->ctx);
  artemis_rt_stream_sync(c->stream);
  ncclCommDestroy(c->comm);
  artemis_rt_free(c->scratch);
  artemis_rt_stream_destroy(c->stream);
  artemis_rt_event_destroy(c->last_op);
  delete c;
}

} // extern "C"
