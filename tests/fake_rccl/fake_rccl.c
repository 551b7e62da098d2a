/* fake_rccl.c -- a TEST DOUBLE for the eight RCCL entry points ecsimd_amd/csrc/group.hip binds (group.hip rccl_api).
 *
 * Test infrastructure, never shipped: it lets a ONE-GPU box execute the RCCL branch of ecsimd_hip_group_scalar_mult
 * (grouped ncclSend / ncclRecv between members) by accepting several "ranks" on one device, which the real RCCL refuses.
 *
 * What it models (and therefore what a green test proves about group.hip): which communicator each call is made on, the peer
 * numbers, counts and datatype of every send / receive, their pairing order, the byte offsets they land at, and stream order --
 * a pair executes at ncclGroupEnd as a copy on the RECEIVER's stream that waits for an event on the SENDER's stream, and the
 * sender's stream then waits for the copy (a send "completes" when its buffer may be reused).  What it does not model: RCCL's
 * transport, its kernels, its channel / proxy machinery, or anything about different physical devices.
 *
 * Stricter than RCCL where that catches bugs: an unmatched send or receive, a count or datatype mismatch, a call outside a
 * group, or a peer out of range is an ERROR here (the real library would hang or corrupt).
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
typedef int ncclDataType_t;

struct ncclComm { int rank, nranks, device; unsigned magic; };
typedef struct ncclComm* ncclComm_t;
#define MAGIC 0xfa4ec0deu

typedef struct { int is_send; const void* buf; size_t count; ncclDataType_t type; int peer; ncclComm_t comm; hipStream_t stream; int matched; } op_t;
#define MAX_OPS 4096
static __thread op_t ops[MAX_OPS];
static __thread int n_ops = 0, depth = 0;
static unsigned long long stat_sends = 0, stat_recvs = 0, stat_groups = 0, stat_bytes = 0, stat_comms = 0;

static size_t type_size(ncclDataType_t t) {
  switch (t) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; default: return 0; }
}

ncclResult_t ncclGetVersion(int* v) { if (!v) return ncclInvalidArgument; *v = 99999; return ncclSuccess; }    /* no real RCCL answers 9.99.99 */
const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) { case 0: return "no error"; case 1: return "fake rccl: HIP error"; case 4: return "fake rccl: invalid argument";
               case 5: return "fake rccl: invalid usage (unmatched or mismatched send/recv, or a call outside a group)"; default: return "fake rccl: error"; }
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devs) {
  if (!comms || n < 1) return ncclInvalidArgument;
  for (int i = 0; i < n; ++i) {
    comms[i] = (ncclComm_t)calloc(1, sizeof(struct ncclComm));
    if (!comms[i]) return ncclSystemError;
    comms[i]->rank = i; comms[i]->nranks = n; comms[i]->device = devs ? devs[i] : i; comms[i]->magic = MAGIC;
  }
  stat_comms += (unsigned long long)n;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { if (!c || c->magic != MAGIC) return ncclInvalidArgument; c->magic = 0; free(c); return ncclSuccess; }
ncclResult_t ncclGroupStart(void) { if (depth == 0) n_ops = 0; ++depth; return ncclSuccess; }

static ncclResult_t enqueue(int is_send, const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
  if (!c || c->magic != MAGIC || !buf || type_size(t) == 0 || peer < 0 || peer >= c->nranks) return ncclInvalidArgument;
  if (depth == 0 || n_ops >= MAX_OPS) return ncclInvalidUsage;          /* group.hip always groups its exchange */
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != c->device) return ncclInvalidUsage;   /* RCCL wants the communicator's device current */
  op_t* o = &ops[n_ops++];
  o->is_send = is_send; o->buf = buf; o->count = count; o->type = t; o->peer = peer; o->comm = c; o->stream = s; o->matched = 0;
  if (is_send) ++stat_sends; else ++stat_recvs;
  return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) { return enqueue(1, buf, count, t, peer, c, s); }
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) { return enqueue(0, buf, count, t, peer, c, s); }

ncclResult_t ncclGroupEnd(void) {
  if (depth <= 0) return ncclInvalidUsage;
  if (--depth > 0) return ncclSuccess;
  ++stat_groups;
  int prev = -1;
  (void)hipGetDevice(&prev);
  ncclResult_t rc = ncclSuccess;
  for (int r = 0; r < n_ops && rc == ncclSuccess; ++r) {
    if (ops[r].is_send) continue;
    int s = -1;                                               /* the first unmatched send from that peer to this rank: NCCL's pairing order */
    for (int j = 0; j < n_ops; ++j)
      if (ops[j].is_send && !ops[j].matched && ops[j].comm->rank == ops[r].peer && ops[j].peer == ops[r].comm->rank) { s = j; break; }
    if (s < 0 || ops[s].count != ops[r].count || ops[s].type != ops[r].type) { rc = ncclInvalidUsage; break; }
    ops[s].matched = ops[r].matched = 1;
    const size_t bytes = ops[r].count * type_size(ops[r].type);
    hipEvent_t sent = NULL, done = NULL;
    hipError_t e = hipSetDevice(ops[s].comm->device);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sent, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(sent, ops[s].stream);                 /* everything the sender enqueued before its ncclSend */
    if (e == hipSuccess) e = hipSetDevice(ops[r].comm->device);
    if (e == hipSuccess) e = hipStreamWaitEvent(ops[r].stream, sent, 0);
    if (e == hipSuccess) e = hipMemcpyAsync((void*)ops[r].buf, ops[s].buf, bytes, hipMemcpyDeviceToDevice, ops[r].stream);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(done, ops[r].stream);
    if (e == hipSuccess) e = hipSetDevice(ops[s].comm->device);
    if (e == hipSuccess) e = hipStreamWaitEvent(ops[s].stream, done, 0);          /* the send buffer is free again only after the copy */
    if (sent) (void)hipEventDestroy(sent);                                        /* destruction is deferred until the event has completed */
    if (done) (void)hipEventDestroy(done);
    if (e != hipSuccess) rc = ncclUnhandledCudaError;
    stat_bytes += bytes;
  }
  for (int j = 0; j < n_ops && rc == ncclSuccess; ++j) if (!ops[j].matched) rc = ncclInvalidUsage;      /* a send nobody receives */
  if (prev >= 0) (void)hipSetDevice(prev);
  n_ops = 0;
  return rc;
}

/* for the test: how much of the RCCL branch ran */
void fake_rccl_stats(unsigned long long out[5]) { out[0] = stat_sends; out[1] = stat_recvs; out[2] = stat_groups; out[3] = stat_bytes; out[4] = stat_comms; }
