// group.hip -- a batch over SEVERAL GPUs behind the C ABI (include/ecsimd_hip.h "device groups").
//
// The reference has nothing to cite here (single thread, no collectives); the contract is
// SURVEY.md 8(e) / BASELINE.json north_star: scalar multiplications are independent units, member m of a group of
// G devices owns the contiguous slice shard_range(n, m, G) of the batch, no collective touches the data path, and
// the result shards are gathered to member 0 by ONE exchange over xGMI: grouped ncclSend / ncclRecv -- RCCL's gather --
// straight into the slices of member 0's output arrays.  One host thread drives all members: every launch, copy and
// RCCL call is asynchronous on the member's stream, and a step is ~50 ms of kernel time per launch.
//
// RCCL is loaded at run time (the copy the process already has, e.g. torch's -- looked up by its soname with RTLD_NOLOAD
// first -- else ROCm's) so that the library has neither a link-time nor a BUILD-time dependency on it: the seven entry
// points used are declared below from RCCL's public, NCCL-compatible ABI (rccl.h is not included).  Members that share a
// device (a group may list a device twice: that is how a one-GPU box exercises the two-member bookkeeping) exchange by
// stream-ordered device copies.
//
// Streams: member m computes on its own stream; the gather runs on a SECOND stream of member 0's device (ncclRecv /
// the device copies), so member 0's next ladder overlaps the previous step's exchange; each other member's ncclSend is
// on that member's own stream, behind its ladder.  Every entry point leaves the caller's current HIP device as it found it.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <set>
#include <vector>

#include "../../include/ecsimd_hip.h"

namespace {
// RCCL's public ABI (rccl.h: NCCL 2.x compatible), restated so that this file builds without the RCCL development headers.
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;                       // enum in rccl.h; ncclSuccess = 0
constexpr ncclResult_t ncclSuccess = 0;
typedef int ncclDataType_t;                     // enum in rccl.h; ncclUint64 = 5
constexpr ncclDataType_t ncclUint64 = 5;
struct rccl_api {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  bool load() {
    // ECSIMD_HIP_RCCL_LIB=<path>: the RCCL build to use instead of the search below (a deployment's pinned copy; the test double
    // tests/fake_rccl/libfake_rccl.so).  Nothing else is tried when it is set: a typo must not fall back silently.
    // secure_getenv: ignored in a set-user-ID / set-group-ID / capability-raised process (this library handles private keys: ecdsa_sign).
    if (const char* forced = secure_getenv("ECSIMD_HIP_RCCL_LIB")) {
      if (forced[0]) { lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL); if (!lib) return false; }
    }
    // the copy this process already has (torch ships its own librccl.so, soname librccl.so.1): a second RCCL in one process
    // would bring a second set of communicators' global state with it
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (lib) break;
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!lib) return false;
#define SYM(f) f = reinterpret_cast<decltype(f)>(dlsym(lib, "nccl" #f)); if (!f) return false
    SYM(CommInitAll); SYM(CommDestroy); SYM(GroupStart); SYM(GroupEnd); SYM(Send); SYM(Recv); SYM(GetErrorString); SYM(GetVersion);
#undef SYM
    return true;
  }
};
// every entry point leaves the caller's current device alone
struct device_guard {
  int prev = -1;
  device_guard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~device_guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct member {
  int device = -1;
  ecsimd_hip_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;          // this member's results are complete (consumed by the gather stream)
  hipEvent_t c0 = nullptr, c1 = nullptr;   // timing events around the member's last ladder launch
  bool timed = false;
  uint64_t* stage = nullptr;          // grow-only: inputs and outputs of the host-array form, outputs of the device form
  size_t stage_bytes = 0;
  bool stage_read_pending = false;    // the device-copy gather read this member's staging: its next ladder waits for group.copied
};
}  // namespace

struct ecsimd_hip_group {
  std::vector<member> m;
  rccl_api rccl;
  std::vector<ncclComm_t> comm;       // one per member; empty when RCCL is not in use (one member, or shared devices)
  hipStream_t gstream = nullptr;      // member 0's device: the gather (ncclRecv / device copies) runs beside member 0's next ladder
  hipEvent_t g0 = nullptr, g1 = nullptr;   // on gstream, around the last gather
  hipEvent_t copied = nullptr;        // on gstream, after the device-copy gather has read the members' staging
  bool timed = false;
  char err[256];
};

namespace {
int gfail(ecsimd_hip_group* g, int code, const char* what, const char* detail = "") {
  if (g) snprintf(g->err, sizeof g->err, "%s%s%s", what, detail[0] ? ": " : "", detail);
  return code;
}
// Growing a staging block frees the old one, which this member's stream (RCCL send) or the gather stream (device copies)
// may still read: both are drained first.  Happens on the first call and when a batch is larger than any before; a call
// that does not grow anything stays asynchronous.
int ensure_stage(ecsimd_hip_group* g, member& mb, size_t bytes) {
  if (mb.stage_bytes >= bytes) return ECSIMD_HIP_OK;
  hipError_t e = hipSetDevice(g->m[0].device);
  if (e == hipSuccess) e = hipStreamSynchronize(g->gstream);
  if (e == hipSuccess) e = hipSetDevice(mb.device);
  if (e == hipSuccess) e = hipStreamSynchronize(mb.stream);
  if (e == hipSuccess && mb.stage) e = hipFree(mb.stage);
  mb.stage = nullptr; mb.stage_bytes = 0;
  if (e == hipSuccess) e = hipMalloc(&mb.stage, bytes);
  if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "group staging hipMalloc", hipGetErrorString(e));
  mb.stage_bytes = bytes;
  return ECSIMD_HIP_OK;
}
}  // namespace

extern "C" {

int ecsimd_hip_shard_range(size_t n_total, int member_index, int members, size_t* first, size_t* count) {
  if (members < 1 || member_index < 0 || member_index >= members || !first || !count) return ECSIMD_HIP_ERR_BAD_ARG;
  const size_t base = n_total / (size_t)members, extra = n_total % (size_t)members, r = (size_t)member_index;
  *count = base + (r < extra ? 1 : 0);
  *first = r * base + (r < extra ? r : extra);
  return ECSIMD_HIP_OK;
}

int ecsimd_hip_group_destroy(ecsimd_hip_group* g) {
  if (!g) return ECSIMD_HIP_ERR_BAD_ARG;
  device_guard keep;
  if (!g->m.empty() && g->m[0].device >= 0 && g->gstream) { (void)hipSetDevice(g->m[0].device); (void)hipStreamSynchronize(g->gstream); }
  for (auto& mb : g->m) {                                     // every stream first: the gather stream may still read another member's staging
    if (mb.device >= 0) (void)hipSetDevice(mb.device);
    if (mb.stream) (void)hipStreamSynchronize(mb.stream);
  }
  for (size_t i = 0; i < g->comm.size(); ++i) if (g->comm[i]) (void)g->rccl.CommDestroy(g->comm[i]);
  for (auto& mb : g->m) {
    if (mb.device >= 0) (void)hipSetDevice(mb.device);
    if (mb.ctx) (void)ecsimd_hip_destroy(mb.ctx);          // the context does not own the group's stream
    if (mb.stage) (void)hipFree(mb.stage);
    if (mb.done) (void)hipEventDestroy(mb.done);
    if (mb.c0) (void)hipEventDestroy(mb.c0);
    if (mb.c1) (void)hipEventDestroy(mb.c1);
    if (mb.stream) (void)hipStreamDestroy(mb.stream);
  }
  if (!g->m.empty() && g->m[0].device >= 0) (void)hipSetDevice(g->m[0].device);
  if (g->g0) (void)hipEventDestroy(g->g0);
  if (g->g1) (void)hipEventDestroy(g->g1);
  if (g->copied) (void)hipEventDestroy(g->copied);
  if (g->gstream) (void)hipStreamDestroy(g->gstream);
  if (g->rccl.lib) (void)dlclose(g->rccl.lib);
  delete g;
  return ECSIMD_HIP_OK;
}

int ecsimd_hip_group_init(const int* devices, int n_devices, ecsimd_hip_group** out) {
  if (!out) return ECSIMD_HIP_ERR_BAD_ARG;
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > 64) return ECSIMD_HIP_ERR_BAD_ARG;
  device_guard keep;
  ecsimd_hip_group* g = new (std::nothrow) ecsimd_hip_group();
  if (!g) return ECSIMD_HIP_ERR_HIP;
  g->err[0] = 0;
  g->m.resize((size_t)n_devices);
  for (int i = 0; i < n_devices; ++i) {
    member& mb = g->m[(size_t)i];
    mb.device = devices[i];
    int rc = ecsimd_hip_init(mb.device, &mb.ctx);            // refuses anything but a gfx950 device
    if (rc != ECSIMD_HIP_OK) { mb.device = -1; (void)ecsimd_hip_group_destroy(g); return rc; }
    hipError_t e = hipSetDevice(mb.device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&mb.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&mb.done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&mb.c0);
    if (e == hipSuccess) e = hipEventCreate(&mb.c1);
    if (e == hipSuccess && ecsimd_hip_set_stream(mb.ctx, mb.stream) != ECSIMD_HIP_OK) e = hipErrorUnknown;
    if (e != hipSuccess) { (void)ecsimd_hip_group_destroy(g); return ECSIMD_HIP_ERR_HIP; }
  }
  hipError_t e = hipSetDevice(g->m[0].device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->gstream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&g->g0);
  if (e == hipSuccess) e = hipEventCreate(&g->g1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&g->copied, hipEventDisableTiming);
  if (e != hipSuccess) { (void)ecsimd_hip_group_destroy(g); return ECSIMD_HIP_ERR_HIP; }
  // RCCL communicators: one per member, only when every member has its own device (RCCL refuses two ranks on one GPU)
  // ECSIMD_HIP_GROUP_FORCE_RCCL=1 (tests only): take the RCCL branch although members share a device.  The real RCCL refuses such a
  // communicator ("Duplicate GPU detected") and group_init then fails; the test double accepts it, which is how a one-GPU box
  // executes the grouped ncclSend / ncclRecv bookkeeping below (tests/test_gpu_parity.py::test_device_group_rccl_branch_on_a_double).
  std::set<int> distinct(devices, devices + n_devices);
  // The hook is honoured only when the library that got loaded IS the test double (ncclGetVersion 99999; no RCCL release answers that).
  const char* force = secure_getenv("ECSIMD_HIP_GROUP_FORCE_RCCL");
  const bool forced = n_devices > 1 && (int)distinct.size() != n_devices && force && force[0] == '1';
  if (n_devices > 1 && ((int)distinct.size() == n_devices || forced)) {
    if (!g->rccl.load()) { (void)ecsimd_hip_group_destroy(g); return ECSIMD_HIP_ERR_HIP; }
    if (forced) { int v = 0; if (g->rccl.GetVersion(&v) != ncclSuccess || v != 99999) { (void)ecsimd_hip_group_destroy(g); return ECSIMD_HIP_ERR_BAD_ARG; } }
    g->comm.assign((size_t)n_devices, nullptr);
    ncclResult_t r = g->rccl.CommInitAll(g->comm.data(), n_devices, devices);
    if (r != ncclSuccess) { g->comm.clear(); (void)ecsimd_hip_group_destroy(g); return ECSIMD_HIP_ERR_HIP; }
  }
  *out = g;
  return ECSIMD_HIP_OK;
}

int ecsimd_hip_group_size(const ecsimd_hip_group* g) { return g ? (int)g->m.size() : ECSIMD_HIP_ERR_BAD_ARG; }
ecsimd_hip_ctx* ecsimd_hip_group_context(ecsimd_hip_group* g, int member_index) {
  return (g && member_index >= 0 && member_index < (int)g->m.size()) ? g->m[(size_t)member_index].ctx : nullptr;
}
int ecsimd_hip_group_uses_rccl(const ecsimd_hip_group* g) { return g ? (g->comm.empty() ? 0 : 1) : ECSIMD_HIP_ERR_BAD_ARG; }
const char* ecsimd_hip_group_last_error(const ecsimd_hip_group* g) { return g ? g->err : "null group"; }

// ncclGetVersion of the RCCL this group's gather goes through (e.g. 22606 = 2.26.6); 0 when the group does not use RCCL.
int ecsimd_hip_group_rccl_version(const ecsimd_hip_group* g) {
  if (!g) return ECSIMD_HIP_ERR_BAD_ARG;
  int v = 0;
  if (g->comm.empty() || !g->rccl.GetVersion || g->rccl.GetVersion(&v) != ncclSuccess) return 0;
  return v;
}

int ecsimd_hip_group_sync(ecsimd_hip_group* g, double* last_gather_ms) {
  if (!g) return ECSIMD_HIP_ERR_BAD_ARG;
  device_guard keep;
  for (auto& mb : g->m) {
    hipError_t e = hipSetDevice(mb.device);
    if (e == hipSuccess) e = hipStreamSynchronize(mb.stream);
    if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "group sync", hipGetErrorString(e));
  }
  hipError_t e = hipSetDevice(g->m[0].device);
  if (e == hipSuccess) e = hipStreamSynchronize(g->gstream);
  if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "group sync (gather stream)", hipGetErrorString(e));
  if (last_gather_ms) {
    float ms = 0.f;
    *last_gather_ms = (g->timed && hipEventElapsedTime(&ms, g->g0, g->g1) == hipSuccess) ? (double)ms : -1.0;
  }
  return ECSIMD_HIP_OK;
}

// Device time of member `member_index`'s LAST ladder launch (HIP events on that member's stream); call after group_sync.
int ecsimd_hip_group_member_ms(ecsimd_hip_group* g, int member_index, double* ms) {
  if (!g || !ms || member_index < 0 || member_index >= (int)g->m.size()) return ECSIMD_HIP_ERR_BAD_ARG;
  member& mb = g->m[(size_t)member_index];
  device_guard keep;
  float t = 0.f;
  if (!mb.timed) { *ms = -1.0; return ECSIMD_HIP_OK; }
  hipError_t e = hipSetDevice(mb.device);
  if (e == hipSuccess) e = hipEventElapsedTime(&t, mb.c0, mb.c1);
  if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "member timing", hipGetErrorString(e));
  *ms = (double)t;
  return ECSIMD_HIP_OK;
}

// What a one-GPU machine can check of the RCCL side: load the library the way ecsimd_hip_group_init does, create a ONE-rank
// communicator on member 0's device and move `elements` 256-bit elements from one buffer to another with the very
// calls the gather uses (grouped ncclSend + ncclRecv, ncclUint64 counts of 4 per element, the gather stream).  Returns 0 when
// the bytes arrived intact.  The exchange between DIFFERENT devices is what only a multi-GPU node can run.
int ecsimd_hip_group_rccl_selftest(ecsimd_hip_group* g, size_t elements) {
  if (!g || elements == 0) return ECSIMD_HIP_ERR_BAD_ARG;
  device_guard keep;
  member& root = g->m[0];
  rccl_api api;
  if (!api.load()) return gfail(g, ECSIMD_HIP_ERR_HIP, "RCCL selftest", "librccl.so.1 could not be loaded");
  int rc = ECSIMD_HIP_OK;
  ncclComm_t comm = nullptr;
  uint64_t* buf = nullptr;
  hipStream_t s = g->gstream;
  hipError_t e = hipSetDevice(root.device);
  if (e == hipSuccess) e = hipMalloc(&buf, 2 * elements * 32);
  std::vector<uint64_t> h(elements * 4), back(elements * 4, 0);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x9e3779b97f4a7c15ull * (i + 1);
  if (e == hipSuccess) e = hipMemcpyAsync(buf, h.data(), elements * 32, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemsetAsync(buf + elements * 4, 0, elements * 32, s);
  if (e != hipSuccess) rc = gfail(g, ECSIMD_HIP_ERR_HIP, "RCCL selftest setup", hipGetErrorString(e));
  if (rc == ECSIMD_HIP_OK) {
    const int dev = root.device;
    ncclResult_t r = api.CommInitAll(&comm, 1, &dev);
    if (r == ncclSuccess) r = api.GroupStart();
    if (r == ncclSuccess) r = api.Recv(buf + elements * 4, elements * 4, ncclUint64, 0, comm, s);
    if (r == ncclSuccess) r = api.Send(buf, elements * 4, ncclUint64, 0, comm, s);
    const ncclResult_t r2 = (r == ncclSuccess) ? api.GroupEnd() : r;
    if (r2 != ncclSuccess) rc = gfail(g, ECSIMD_HIP_ERR_HIP, "RCCL selftest", api.GetErrorString(r2));
  }
  if (rc == ECSIMD_HIP_OK) {
    e = hipMemcpyAsync(back.data(), buf + elements * 4, elements * 32, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) rc = gfail(g, ECSIMD_HIP_ERR_HIP, "RCCL selftest read-back", hipGetErrorString(e));
    else if (back != h) rc = gfail(g, ECSIMD_HIP_ERR_HIP, "RCCL selftest", "the received bytes differ from the sent ones");
  }
  (void)hipStreamSynchronize(s);
  if (comm) (void)api.CommDestroy(comm);
  if (buf) (void)hipFree(buf);
  (void)dlclose(api.lib);
  return rc;
}

// Device-resident form.  k[m], x[m], y[m]: member m's shard (shard_range(n, m, G) elements) in ITS device memory.
// ox, oy, oz: n elements each in member 0's device memory (with ECSIMD_HIP_OUT_AFFINE oz may be NULL, and oy too: x only).
// Asynchronous: ecsimd_hip_group_sync() waits and reports how long the gather took on the gather stream.
int ecsimd_hip_group_scalar_mult(ecsimd_hip_group* g, int curve, const uint64_t* const* k, const uint64_t* const* x, const uint64_t* const* y,
                                 uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags) {
  if (!g) return ECSIMD_HIP_ERR_BAD_ARG;
  if (!k || !x || !y || !ox) return gfail(g, ECSIMD_HIP_ERR_BAD_ARG, "bad argument: null pointer");
  const bool affine = (flags & ECSIMD_HIP_OUT_AFFINE) != 0;
  const bool gather = (flags & ECSIMD_HIP_GROUP_NO_GATHER) == 0;
  flags &= ~ECSIMD_HIP_GROUP_NO_GATHER;
  if (!affine && (!oy || !oz)) return gfail(g, ECSIMD_HIP_ERR_BAD_ARG, "bad argument: a Jacobian result needs ox, oy and oz");
  device_guard keep;
  const int G = (int)g->m.size(), outs = affine ? (oy ? 2 : 1) : 3;
  uint64_t* root_out[3] = {ox, oy, oz};
  // ---- compute: every member on its own slice; member 0 straight into its slice of the result arrays
  for (int i = 0; i < G; ++i) {
    member& mb = g->m[(size_t)i];
    size_t first, count; (void)ecsimd_hip_shard_range(n, i, G, &first, &count);
    if (count == 0) continue;
    uint64_t* o[3] = {nullptr, nullptr, nullptr};
    if (i == 0) { for (int c = 0; c < outs; ++c) o[c] = root_out[c] + 4 * first; }
    else {
      int rc = ensure_stage(g, mb, (size_t)outs * count * 32);
      if (rc != ECSIMD_HIP_OK) return rc;
      for (int c = 0; c < outs; ++c) o[c] = mb.stage + (size_t)c * count * 4;
    }
    if (!k[i] || !x[i] || !y[i]) return gfail(g, ECSIMD_HIP_ERR_BAD_ARG, "bad argument: a member's shard pointer is null");
    hipError_t e = hipSetDevice(mb.device);
    // the previous device-copy gather may still be reading this member's staging on the gather stream
    // (per member: one that sits a small batch out still waits the next time it writes its staging; the event's latest record is
    // behind every earlier copy on the in-order gather stream)
    if (e == hipSuccess && i != 0 && mb.stage_read_pending) { e = hipStreamWaitEvent(mb.stream, g->copied, 0); mb.stage_read_pending = false; }
    if (e == hipSuccess) e = hipEventRecord(mb.c0, mb.stream);
    if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "member launch", hipGetErrorString(e));
    int rc = ecsimd_hip_scalar_mult(mb.ctx, curve, k[i], x[i], y[i], o[0], o[1], o[2], count, flags);
    if (rc != ECSIMD_HIP_OK) return gfail(g, rc, "member scalar_mult", ecsimd_hip_last_error(mb.ctx));
    e = hipSetDevice(mb.device);
    if (e == hipSuccess) e = hipEventRecord(mb.c1, mb.stream);
    if (e == hipSuccess) e = hipEventRecord(mb.done, mb.stream);
    if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "member launch", hipGetErrorString(e));
    mb.timed = true;
  }
  if (!gather || G == 1) return ECSIMD_HIP_OK;
  // ---- the one exchange: every other member's shard into member 0's arrays, on member 0's gather stream
  // The gather stream first waits for EVERY sending member's ladder (cross-device event waits are legal), and only then is g0
  // recorded: without the waits RCCL's receive kernel would start at once on member 0's device and spin on its CUs beside member
  // 0's ladder for the whole step, and g0..g1 would time ladder + transfer instead of the transfer.
  member& root = g->m[0];
  hipError_t e = hipSetDevice(root.device);
  for (int i = 1; i < G && e == hipSuccess; ++i) {
    size_t first, count; (void)ecsimd_hip_shard_range(n, i, G, &first, &count);
    if (count != 0) e = hipStreamWaitEvent(g->gstream, g->m[(size_t)i].done, 0);
  }
  if (e == hipSuccess) e = hipEventRecord(g->g0, g->gstream);
  if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "gather", hipGetErrorString(e));
  if (!g->comm.empty()) {
    ncclResult_t r = g->rccl.GroupStart();
    for (int i = 1; i < G && r == ncclSuccess; ++i) {
      size_t first, count; (void)ecsimd_hip_shard_range(n, i, G, &first, &count);
      if (count == 0) continue;
      member& mb = g->m[(size_t)i];
      for (int c = 0; c < outs && r == ncclSuccess; ++c) {
        (void)hipSetDevice(root.device);
        r = g->rccl.Recv(root_out[c] + 4 * first, count * 4, ncclUint64, i, g->comm[0], g->gstream);
        if (r != ncclSuccess) break;
        (void)hipSetDevice(mb.device);
        r = g->rccl.Send(mb.stage + (size_t)c * count * 4, count * 4, ncclUint64, 0, g->comm[(size_t)i], mb.stream);   // behind the member's ladder
      }
    }
    const ncclResult_t r2 = g->rccl.GroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "RCCL gather", g->rccl.GetErrorString(r != ncclSuccess ? r : r2));
    e = hipSetDevice(root.device);
  } else {
    for (int i = 1; i < G; ++i) {
      size_t first, count; (void)ecsimd_hip_shard_range(n, i, G, &first, &count);
      if (count == 0) continue;
      member& mb = g->m[(size_t)i];
      e = hipSetDevice(root.device);                       // (the gather stream already waits for mb.done, above)
      for (int c = 0; c < outs && e == hipSuccess; ++c)
        e = hipMemcpyAsync(root_out[c] + 4 * first, mb.stage + (size_t)c * count * 4, count * 32, hipMemcpyDeviceToDevice, g->gstream);
      mb.stage_read_pending = true;
      if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "gather copy", hipGetErrorString(e));
    }
    // the members' next ladders overwrite their staging: they wait for these copies (see the compute loop)
    e = hipEventRecord(g->copied, g->gstream);
  }
  if (e == hipSuccess) e = hipEventRecord(g->g1, g->gstream);
  if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "gather", hipGetErrorString(e));
  g->timed = true;
  return ECSIMD_HIP_OK;
}

// Host-array form: k, x, y (n elements each) and the result arrays live in HOST memory.  Each member copies its slice
// in, computes, the shards are gathered to member 0 and copied out.  Synchronous (returns with the results in place).
// A CONVENIENCE path: the caller's arrays are pageable as far as this library knows, so the copies are staged by the
// HIP runtime and do not overlap the ladders (1.5 GB each way at config-4 size, ~0.1 s at PCIe rates); a caller who needs
// the overlap keeps pinned buffers and device-resident shards and uses ecsimd_hip_group_scalar_mult (tools/pcie_rate.py
// shows the three-stream pipeline on one device).
int ecsimd_hip_group_scalar_mult_host(ecsimd_hip_group* g, int curve, const uint64_t* k, const uint64_t* x, const uint64_t* y,
                                      uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags) {
  if (!g) return ECSIMD_HIP_ERR_BAD_ARG;
  if (!k || !x || !y || !ox) return gfail(g, ECSIMD_HIP_ERR_BAD_ARG, "bad argument: null pointer");
  const bool affine = (flags & ECSIMD_HIP_OUT_AFFINE) != 0;
  if (!affine && (!oy || !oz)) return gfail(g, ECSIMD_HIP_ERR_BAD_ARG, "bad argument: a Jacobian result needs ox, oy and oz");
  if (flags & ECSIMD_HIP_GROUP_NO_GATHER) return gfail(g, ECSIMD_HIP_ERR_BAD_ARG, "bad argument: the host-array form always gathers");
  if (n == 0) return ECSIMD_HIP_OK;
  device_guard keep;
  const int G = (int)g->m.size(), outs = affine ? (oy ? 2 : 1) : 3;
  // staging layout per member: [outputs of the device form: outs x count][k, x, y: 3 x count]; member 0 adds the n-element results
  std::vector<const uint64_t*> dk((size_t)G, nullptr), dx((size_t)G, nullptr), dy((size_t)G, nullptr);
  uint64_t* res[3] = {nullptr, nullptr, nullptr};
  for (int i = 0; i < G; ++i) {
    member& mb = g->m[(size_t)i];
    size_t first, count; (void)ecsimd_hip_shard_range(n, i, G, &first, &count);
    const size_t words = (size_t)(outs + 3) * count * 4 + (i == 0 ? (size_t)outs * n * 4 : 0);
    int rc = ensure_stage(g, mb, words * 8 + 32);
    if (rc != ECSIMD_HIP_OK) return rc;
    uint64_t* in = mb.stage + (size_t)outs * count * 4;
    if (i == 0) for (int c = 0; c < outs; ++c) res[c] = mb.stage + (size_t)(outs + 3) * count * 4 + (size_t)c * n * 4;
    if (count == 0) continue;
    hipError_t e = hipSetDevice(mb.device);
    const uint64_t* src[3] = {k, x, y};
    for (int c = 0; c < 3 && e == hipSuccess; ++c)
      e = hipMemcpyAsync(in + (size_t)c * count * 4, src[c] + 4 * first, count * 32, hipMemcpyHostToDevice, mb.stream);
    if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "group h2d", hipGetErrorString(e));
    dk[(size_t)i] = in; dx[(size_t)i] = in + count * 4; dy[(size_t)i] = in + 2 * count * 4;
  }
  int rc = ecsimd_hip_group_scalar_mult(g, curve, dk.data(), dx.data(), dy.data(), res[0], res[1], res[2], n, flags);
  if (rc != ECSIMD_HIP_OK) return rc;
  rc = ecsimd_hip_group_sync(g, nullptr);                 // member 0's ladder (its stream) and the gather (the gather stream)
  if (rc != ECSIMD_HIP_OK) return rc;
  member& root = g->m[0];
  hipError_t e = hipSetDevice(root.device);
  uint64_t* dst[3] = {ox, oy, oz};
  for (int c = 0; c < outs && e == hipSuccess; ++c) e = hipMemcpyAsync(dst[c], res[c], n * 32, hipMemcpyDeviceToHost, root.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(root.stream);
  if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "group d2h", hipGetErrorString(e));
  return ECSIMD_HIP_OK;
}

}  // extern "C"
