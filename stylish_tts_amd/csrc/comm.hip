// sty_comm_*: the gradient exchange of the data-parallel step UNDER the library (sty_comm_init /
// sty_comm_allreduce_bucket / sty_comm_destroy; the reference gets its exchange from accelerate's DDP wrappers,
// train/train_context.py:94-104).  One communicator per process (one process per GPU), RCCL underneath, on a HIP stream the
// LIBRARY owns -- so that which hardware queue the collective's kernels share with the step's four compute streams is this
// library's decision, not a side effect of torch.distributed's stream pool (with GPU_MAX_HW_QUEUES=3 a
// fifth active stream landed on the weight-gradient stream's queue and the step went from 46 to 70 ms).
//
// A bucket's sum over ranks runs as ncclReduceScatter + ncclAllGather in place (xGMI is point-to-point: seven links per GPU;
// the two halves of the ring all-reduce, each moving (world - 1) / world of the bucket, issued as two collectives so that a
// later version can put the optimizer step of a rank's shard between them); buckets whose length is not a multiple of
// world x 4 floats take ncclAllReduce.  Ordering is by events: the communicator's stream waits for an event recorded on the
// producer's stream when the bucket is handed over, and sty_comm_wait makes a consumer stream wait for everything handed over
// so far.  RCCL is resolved at run time (dlopen): the library loads without it, and a process that already holds a copy (torch's)
// shares that copy.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "sty_common.h"

namespace sty {

// (declared here instead of including <rccl/rccl.h>: nothing of RCCL is needed at build or load time)
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId {
  char internal[128];
};
typedef int ncclResult_t;
constexpr int kNcclFloat = 7, kNcclSum = 0;  // ncclFloat32, ncclSum (rccl.h:448,466)

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl.h) return STY_OK;
  const char* names[] = {getenv("STY_RCCL_LIB"), "librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    if (!n || !*n) continue;
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    set_error("sty_comm: librccl.so not found (%s); set STY_RCCL_LIB to its path", dlerror());
    return STY_ESTATE;
  }
  Rccl r;
  r.h = h;
#define STY_SYM(field, name)                                                  \
  *reinterpret_cast<void**>(&r.field) = dlsym(h, name);                       \
  if (!r.field) {                                                             \
    set_error("sty_comm: %s missing from the RCCL library", name);            \
    return STY_ESTATE;                                                        \
  }
  STY_SYM(GetUniqueId, "ncclGetUniqueId")
  STY_SYM(CommInitRank, "ncclCommInitRank")
  STY_SYM(CommDestroy, "ncclCommDestroy")
  STY_SYM(GetErrorString, "ncclGetErrorString")
  STY_SYM(AllReduce, "ncclAllReduce")
  STY_SYM(ReduceScatter, "ncclReduceScatter")
  STY_SYM(AllGather, "ncclAllGather")
#undef STY_SYM
  g_rccl = r;
  return STY_OK;
}

static int nccl_fail(ncclResult_t e, const char* what) {
  set_error("RCCL error in %s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?");
  return STY_EHIP;
}
#define STY_NCCL(expr)                                 \
  do {                                                 \
    ncclResult_t _e = (expr);                          \
    if (_e != 0) return sty::nccl_fail(_e, #expr);     \
  } while (0)

}  // namespace sty

using namespace sty;

struct sty_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  bool own_stream = true;
  hipStream_t stream = nullptr;  // the communicator's own stream (or the caller's: sty_comm_set_stream)
  hipEvent_t handed = nullptr;   // recorded on the producer's stream at every hand-over
  hipEvent_t done = nullptr;     // recorded on `stream` behind the last collective started
  uint64_t buckets = 0, rs_ag = 0;
  double bytes = 0.0;
};

extern "C" {

int sty_comm_unique_id(void* id128) {
  if (!id128) {
    set_error("sty_comm_unique_id: null argument");
    return STY_EINVAL;
  }
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  STY_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128, id.internal, 128);
  return STY_OK;
}

int sty_comm_init(const void* id128, int rank, int world, int stream_priority, sty_comm** out) {
  if (!id128 || !out || world <= 0 || rank < 0 || rank >= world) {
    set_error("sty_comm_init: bad argument");
    return STY_EINVAL;
  }
  int rc = rccl_load();
  if (rc) return rc;
  sty_comm* c = new sty_comm;
  c->rank = rank;
  c->world = world;
  ncclUniqueId id;
  memcpy(id.internal, id128, 128);
  ncclResult_t e = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (e != 0) {
    delete c;
    return nccl_fail(e, "ncclCommInitRank");
  }
  int lo = 0, hi = 0;  // (numerically lower = higher priority)
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  int pr = stream_priority < 0 ? hi : (stream_priority > 0 ? lo : 0);
  hipError_t he = hipStreamCreateWithPriority(&c->stream, getenv("STY_COMM_BLOCKING_STREAM") ? hipStreamDefault : hipStreamNonBlocking, pr);
  if (he == hipSuccess) he = hipEventCreateWithFlags(&c->handed, hipEventDisableTiming);
  if (he == hipSuccess) he = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
  if (he != hipSuccess) {
    (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return hip_fail(he, "sty_comm_init: stream / events");
  }
  *out = c;
  return STY_OK;
}

int sty_comm_allreduce_bucket(sty_comm* c, float* buf, size_t n, void* producer_stream) {
  if (!c || !buf || n == 0) {
    set_error("sty_comm_allreduce_bucket: bad argument");
    return STY_EINVAL;
  }
  // the bucket is complete at this point of the producer's stream
  STY_HIP(hipEventRecord(c->handed, static_cast<hipStream_t>(producer_stream)));
  STY_HIP(hipStreamWaitEvent(c->stream, c->handed, 0));
  const size_t w = (size_t)c->world;
  if (n % (4 * w) == 0 && !getenv("STY_COMM_ALLREDUCE")) {
    const size_t per = n / w;
    float* mine = buf + (size_t)c->rank * per;  // in place: recvbuff = sendbuff + rank * recvcount (rccl.h:655-681)
    STY_NCCL(g_rccl.ReduceScatter(buf, mine, per, kNcclFloat, kNcclSum, c->comm, c->stream));
    STY_NCCL(g_rccl.AllGather(mine, buf, per, kNcclFloat, c->comm, c->stream));
    c->rs_ag += 1;
  } else {
    STY_NCCL(g_rccl.AllReduce(buf, buf, n, kNcclFloat, kNcclSum, c->comm, c->stream));
  }
  STY_HIP(hipEventRecord(c->done, c->stream));
  c->buckets += 1;
  c->bytes += 4.0 * (double)n;
  return STY_OK;
}

int sty_comm_wait(sty_comm* c, void* consumer_stream) {
  if (!c) {
    set_error("sty_comm_wait: null communicator");
    return STY_EINVAL;
  }
  if (c->buckets) STY_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), c->done, 0));
  return STY_OK;
}

int sty_comm_set_stream(sty_comm* c, void* stream) {
  if (!c) {
    set_error("sty_comm_set_stream: null communicator");
    return STY_EINVAL;
  }
  if (c->stream && c->own_stream) {
    STY_HIP(hipStreamSynchronize(c->stream));
    (void)hipStreamDestroy(c->stream);
  }
  c->stream = static_cast<hipStream_t>(stream);
  c->own_stream = false;
  return STY_OK;
}

int sty_comm_stats(sty_comm* c, uint64_t* buckets, uint64_t* reduce_scatter_all_gather, double* bytes) {
  if (!c) {
    set_error("sty_comm_stats: null communicator");
    return STY_EINVAL;
  }
  if (buckets) *buckets = c->buckets;
  if (reduce_scatter_all_gather) *reduce_scatter_all_gather = c->rs_ag;
  if (bytes) *bytes = c->bytes;
  return STY_OK;
}

int sty_comm_destroy(sty_comm* c) {
  if (!c) return STY_OK;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  if (c->handed) (void)hipEventDestroy(c->handed);
  if (c->done) (void)hipEventDestroy(c->done);
  if (c->stream && c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return STY_OK;
}

}  // extern "C"
