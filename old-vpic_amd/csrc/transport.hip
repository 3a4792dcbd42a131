// transport.hip -- the GPU-aware transport of the multi-domain hosts: RCCL point-to-point transfers of DEVICE buffers on
// a communication stream of its own, ordered against the engine's stream with events; the host never waits for a message.
//
// What it stands in for: the reference's port layer (src/util/mp/dmp/mp_dmp.c:241-266 mp_begin_send / mp_begin_recv /
// mp_end_*; src/grid/grid_comm.c:7-78 begin_send_port / begin_recv_port) -- a non-blocking send and receive per shared
// face, posted together, waited for where the data is needed (advance_e.c:114,153,191-197; boundary_p.c:341-384).  A rank
// whose neighbour across a periodic axis is itself sends to itself, as the reference does (grid_comm.c:17-19,49:
// MPI_Issend to its own rank): a one-rank communicator is a legitimate configuration.
//
// RCCL is opened with dlopen when the first communicator is made: the library has no link-time dependency on it (a
// process that never makes one -- the single-domain engine, the Python binding beside torch's own RCCL -- never loads it).
#include "engine.h"
#include <dlfcn.h>
#include <cstring>
#include <cstdlib>
#include <rccl/rccl.h>
#include <vector>

namespace vpichip {
namespace {

struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  const char *(*GetLastError)(ncclComm_t) = nullptr;
};
RcclApi g_rccl;

int load_rccl() {
  if (g_rccl.handle) return 0;
  const char *names[] = {getenv("VPIC_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) VH_FAIL("the RCCL transport needs librccl.so (%s)", dlerror());
#define SYM(field, name) do { *(void **)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) { dlclose(h); VH_FAIL("librccl.so has no %s", name); } } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
  SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  *(void **)(&g_rccl.GetLastError) = dlsym(h, "ncclGetLastError");      // (optional)
  g_rccl.handle = h;
  return 0;
}

}  // namespace
}  // namespace vpichip

using namespace vpichip;

struct vpic_hip_comm {
  vpic_hip_engine_t *e = nullptr;
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0;
  hipStream_t stream = nullptr;                    // the communication stream (high priority: above the push kernels in the hardware queues)
  std::vector<hipEvent_t> done;                    // a ring of events: token k <-> done[k % size]
  hipEvent_t packed = nullptr;
  unsigned turn = 0;
  int64_t messages = 0, bytes = 0;                 // sent since creation (vpic_hip_comm_stats)
  // vpic_hip_comm_timing: how long the exchanges took on the communication stream and how long the engine's stream stood
  // still for them (events with timing around every exchange and every wait; read back and summed by _timing)
  bool timed = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> t_xfer, t_wait;
  double xfer_ms = 0, wait_ms = 0; int64_t n_timed = 0;
};
static int timed_pair(std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, hipEvent_t *a, hipEvent_t *b) {
  hipEvent_t x = nullptr, y = nullptr;
  if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) return 1;
  v.push_back({x, y}); *a = x; *b = y;
  return 0;
}
static int collect_timing(vpic_hip_comm *c) {
  for (int kind = 0; kind < 2; kind++) {
    auto &v = kind ? c->t_wait : c->t_xfer;
    for (auto &pr : v) {
      float ms = 0;
      VH_CHECK(hipEventSynchronize(pr.second));
      VH_CHECK(hipEventElapsedTime(&ms, pr.first, pr.second));
      (kind ? c->wait_ms : c->xfer_ms) += ms;
      if (!kind) c->n_timed++;
      (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
    }
    v.clear();
  }
  return 0;
}

#define NCCL_CHECK(c, expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) { \
    const char *_d = (g_rccl.GetLastError && (c)) ? g_rccl.GetLastError((c)->comm) : ""; \
    VH_FAIL("%s failed: %s %s(%s:%d)", #expr, g_rccl.GetErrorString(_r), _d ? _d : "", __FILE__, __LINE__); } } while (0)

extern "C" {

int vpic_hip_comm_unique_id(void *id128) {
  if (!id128) VH_FAIL("Bad id");
  if (load_rccl()) return 1;
  static_assert(sizeof(ncclUniqueId) == VPIC_HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCL_CHECK((vpic_hip_comm *)nullptr, g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}

int vpic_hip_comm_create(vpic_hip_comm_t **out, vpic_hip_engine_t *e, const void *id128, int nranks, int rank) {
  if (!out || !e || !id128 || nranks < 1 || rank < 0 || rank >= nranks) VH_FAIL("Bad communicator arguments");
  *out = nullptr;
  if (load_rccl()) return 1;
  VH_CHECK(hipSetDevice(e->device));
  vpic_hip_comm *c = new vpic_hip_comm;
  c->e = e; c->nranks = nranks; c->rank = rank;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  const ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    // (the usual reason on a development box: several ranks on ONE device, which RCCL refuses)
    set_error("ncclCommInitRank(%d ranks, rank %d, device %d) failed: %s -- one rank per GPU is required; a host that must share a "
              "device between ranks has to ask for its host-staged transport", nranks, rank, e->device, g_rccl.GetErrorString(r));
    delete c;
    return 1;
  }
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);                              // (hi = the numerically lowest = highest priority)
  if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi) != hipSuccess) VH_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->done.resize(64);
  for (auto &ev : c->done) VH_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  VH_CHECK(hipEventCreateWithFlags(&c->packed, hipEventDisableTiming));
  *out = c;
  return 0;
}

int vpic_hip_comm_destroy(vpic_hip_comm_t *c) {
  if (!c) return 0;
  (void)hipSetDevice(c->e->device);
  (void)hipStreamSynchronize(c->stream);
  (void)collect_timing(c);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  for (auto ev : c->done) (void)hipEventDestroy(ev);
  if (c->packed) (void)hipEventDestroy(c->packed);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

// One exchange: every message in ONE group (all shared faces posted together, like the reference's begin_send_port /
// begin_recv_port pairs), on the communication stream, behind everything the engine's stream has been given so far (the
// packing kernels).  Messages between one pair of ranks match in the order given: the callers list sends and receives by
// direction 0..5 on every rank.  Nothing here waits on the host.
int vpic_hip_comm_start(vpic_hip_comm_t *c, int n_send, const void *const *sbuf, const size_t *sbytes, const int *speer,
                        int n_recv, void *const *rbuf, const size_t *rbytes, const int *rpeer, int *token) {
  if (!c || !token || n_send < 0 || n_recv < 0) VH_FAIL("Bad exchange");
  VH_CHECK(hipSetDevice(c->e->device));
  for (int k = 0; k < n_send; k++) if (!sbuf[k] || speer[k] < 0 || speer[k] >= c->nranks) VH_FAIL("Bad send %d (peer %d of %d)", k, speer[k], c->nranks);
  for (int k = 0; k < n_recv; k++) if (!rbuf[k] || rpeer[k] < 0 || rpeer[k] >= c->nranks) VH_FAIL("Bad receive %d (peer %d of %d)", k, rpeer[k], c->nranks);
  const unsigned t = c->turn++;
  hipEvent_t done = c->done[t % c->done.size()];
  VH_CHECK(hipEventRecord(c->packed, c->e->stream));
  VH_CHECK(hipStreamWaitEvent(c->stream, c->packed, 0));
  hipEvent_t ta = nullptr, tb = nullptr;
  if (c->timed) { if (timed_pair(c->t_xfer, &ta, &tb)) VH_FAIL("out of events"); VH_CHECK(hipEventRecord(ta, c->stream)); }
  NCCL_CHECK(c, g_rccl.GroupStart());
  for (int k = 0; k < n_send; k++) {
    NCCL_CHECK(c, g_rccl.Send(sbuf[k], sbytes[k], ncclChar, speer[k], c->comm, c->stream));
    c->messages++; c->bytes += (int64_t)sbytes[k];
  }
  for (int k = 0; k < n_recv; k++) NCCL_CHECK(c, g_rccl.Recv(rbuf[k], rbytes[k], ncclChar, rpeer[k], c->comm, c->stream));
  NCCL_CHECK(c, g_rccl.GroupEnd());
  if (tb) VH_CHECK(hipEventRecord(tb, c->stream));
  VH_CHECK(hipEventRecord(done, c->stream));
  *token = (int)(t % c->done.size());
  return 0;
}

// what the engine's stream is given next waits for that exchange (device-side; the host goes on)
int vpic_hip_comm_finish(vpic_hip_comm_t *c, int token) {
  if (!c || token < 0 || token >= (int)c->done.size()) VH_FAIL("Bad token");
  VH_CHECK(hipSetDevice(c->e->device));
  hipEvent_t ta = nullptr, tb = nullptr;
  if (c->timed) { if (timed_pair(c->t_wait, &ta, &tb)) VH_FAIL("out of events"); VH_CHECK(hipEventRecord(ta, c->e->stream)); }
  VH_CHECK(hipStreamWaitEvent(c->e->stream, c->done[(size_t)token], 0));
  if (tb) VH_CHECK(hipEventRecord(tb, c->e->stream));
  return 0;
}

// timing of the exchanges (bench.py): on = 1 starts (and clears), on = 0 stops; the sums so far (call when the device is idle):
// the time the exchanges took on the communication stream, the time the engine's stream waited for them, how many there were
int vpic_hip_comm_timing(vpic_hip_comm_t *c, int on, double *exchange_ms, double *exposed_ms, int64_t *exchanges) {
  if (!c) VH_FAIL("Bad communicator");
  VH_CHECK(hipSetDevice(c->e->device));
  if (collect_timing(c)) return 1;
  if (exchange_ms) *exchange_ms = c->xfer_ms;
  if (exposed_ms) *exposed_ms = c->wait_ms;
  if (exchanges) *exchanges = c->n_timed;
  if (on >= 0) { c->timed = on != 0; if (on) { c->xfer_ms = c->wait_ms = 0; c->n_timed = 0; } }
  return 0;
}

int vpic_hip_comm_stats(vpic_hip_comm_t *c, int64_t *messages, int64_t *bytes) {
  if (!c) VH_FAIL("Bad communicator");
  if (messages) *messages = c->messages;
  if (bytes) *bytes = c->bytes;
  return 0;
}

}  // extern "C"
