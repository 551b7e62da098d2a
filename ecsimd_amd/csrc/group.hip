// group.hip -- a batch over SEVERAL GPUs behind the C ABI (include/ecsimd_hip.h "device groups").
//
// The reference has nothing to cite here (single thread, no collectives); the contract is
// SURVEY.md 8(e) / BASELINE.json north_star: scalar multiplications are independent units, member m of a group of
// G devices owns the contiguous slice shard_range(n, m, G) of the batch, no collective touches the data path, and
// the result shards are gathered to member 0 by ONE exchange over xGMI: grouped ncclSend / ncclRecv -- RCCL's gather --
// straight into the slices of member 0's output arrays.  One host thread drives all members: every launch, copy and
// RCCL call is asynchronous on the member's stream, and a step is ~50 ms of kernel time per launch.
//
// RCCL is loaded at run time (dlopen of the library the process already has, e.g. torch's, or ROCm's) so that the
// single-GPU library keeps no link-time dependency on it.  Members that share a device (a group may list a device
// twice: that is how a one-GPU box exercises the two-member bookkeeping) exchange by stream-ordered device copies.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <set>
#include <vector>

#include "../../include/ecsimd_hip.h"

namespace {
struct rccl_api {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load() {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return false;
#define SYM(f) f = reinterpret_cast<decltype(f)>(dlsym(lib, "nccl" #f)); if (!f) return false
    SYM(CommInitAll); SYM(CommDestroy); SYM(GroupStart); SYM(GroupEnd); SYM(Send); SYM(Recv); SYM(GetErrorString);
#undef SYM
    return true;
  }
};

struct member {
  int device = -1;
  ecsimd_hip_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;          // this member's results are complete (consumed by member 0's stream)
  uint64_t* stage = nullptr;          // grow-only: inputs and outputs of the host-array form, outputs of the device form
  size_t stage_bytes = 0;
};
}  // namespace

struct ecsimd_hip_group {
  std::vector<member> m;
  rccl_api rccl;
  std::vector<ncclComm_t> comm;       // one per member; empty when RCCL is not in use (one member, or shared devices)
  hipEvent_t g0 = nullptr, g1 = nullptr;   // on member 0's stream, around the last gather
  bool timed = false;
  char err[256];
};

namespace {
int gfail(ecsimd_hip_group* g, int code, const char* what, const char* detail = "") {
  if (g) snprintf(g->err, sizeof g->err, "%s%s%s", what, detail[0] ? ": " : "", detail);
  return code;
}
int ensure_stage(ecsimd_hip_group* g, member& mb, size_t bytes) {
  if (mb.stage_bytes >= bytes) return ECSIMD_HIP_OK;
  // the block may still be read by this member's stream (RCCL send) or by member 0's (the device-copy gather of a shared device)
  hipError_t e = hipSetDevice(g->m[0].device);
  if (e == hipSuccess) e = hipStreamSynchronize(g->m[0].stream);
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
  for (auto& mb : g->m) {                                     // every stream first: member 0's may still read another member's staging
    if (mb.device >= 0) (void)hipSetDevice(mb.device);
    if (mb.stream) (void)hipStreamSynchronize(mb.stream);
  }
  for (size_t i = 0; i < g->comm.size(); ++i) if (g->comm[i]) (void)g->rccl.CommDestroy(g->comm[i]);
  for (auto& mb : g->m) {
    if (mb.device >= 0) (void)hipSetDevice(mb.device);
    if (mb.ctx) (void)ecsimd_hip_destroy(mb.ctx);          // the context does not own the group's stream
    if (mb.stage) (void)hipFree(mb.stage);
    if (mb.done) (void)hipEventDestroy(mb.done);
    if (mb.stream) (void)hipStreamDestroy(mb.stream);
  }
  if (!g->m.empty()) (void)hipSetDevice(g->m[0].device);
  if (g->g0) (void)hipEventDestroy(g->g0);
  if (g->g1) (void)hipEventDestroy(g->g1);
  if (g->rccl.lib) (void)dlclose(g->rccl.lib);
  delete g;
  return ECSIMD_HIP_OK;
}

int ecsimd_hip_group_init(const int* devices, int n_devices, ecsimd_hip_group** out) {
  if (!out) return ECSIMD_HIP_ERR_BAD_ARG;
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > 64) return ECSIMD_HIP_ERR_BAD_ARG;
  ecsimd_hip_group* g = new (std::nothrow) ecsimd_hip_group();
  if (!g) return ECSIMD_HIP_ERR_HIP;
  g->err[0] = 0;
  g->m.resize((size_t)n_devices);
  for (int i = 0; i < n_devices; ++i) {
    member& mb = g->m[(size_t)i];
    mb.device = devices[i];
    int rc = ecsimd_hip_init(mb.device, &mb.ctx);            // refuses anything but a gfx950 device
    if (rc != ECSIMD_HIP_OK) { (void)ecsimd_hip_group_destroy(g); return rc; }
    hipError_t e = hipSetDevice(mb.device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&mb.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&mb.done, hipEventDisableTiming);
    if (e == hipSuccess && ecsimd_hip_set_stream(mb.ctx, mb.stream) != ECSIMD_HIP_OK) e = hipErrorUnknown;
    if (e != hipSuccess) { (void)ecsimd_hip_group_destroy(g); return ECSIMD_HIP_ERR_HIP; }
  }
  (void)hipSetDevice(g->m[0].device);
  if (hipEventCreate(&g->g0) != hipSuccess || hipEventCreate(&g->g1) != hipSuccess) { (void)ecsimd_hip_group_destroy(g); return ECSIMD_HIP_ERR_HIP; }
  // RCCL communicators: one per member, only when every member has its own device (RCCL refuses two ranks on one GPU)
  std::set<int> distinct(devices, devices + n_devices);
  if (n_devices > 1 && (int)distinct.size() == n_devices) {
    if (!g->rccl.load()) { (void)ecsimd_hip_group_destroy(g); return ECSIMD_HIP_ERR_HIP; }
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

int ecsimd_hip_group_sync(ecsimd_hip_group* g, double* last_gather_ms) {
  if (!g) return ECSIMD_HIP_ERR_BAD_ARG;
  for (auto& mb : g->m) {
    hipError_t e = hipSetDevice(mb.device);
    if (e == hipSuccess) e = hipStreamSynchronize(mb.stream);
    if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "group sync", hipGetErrorString(e));
  }
  if (last_gather_ms) {
    float ms = 0.f;
    *last_gather_ms = (g->timed && hipEventElapsedTime(&ms, g->g0, g->g1) == hipSuccess) ? (double)ms : -1.0;
  }
  return ECSIMD_HIP_OK;
}

// What a one-GPU machine can check of the RCCL side: load the library the way ecsimd_hip_group_init does, create a ONE-rank
// communicator on member 0's device and move `elements` 256-bit elements from one buffer to another with the very
// calls the gather uses (grouped ncclSend + ncclRecv, ncclUint64 counts of 4 per element, member 0's stream).  Returns 0 when
// the bytes arrived intact.  The exchange between DIFFERENT devices is what only a multi-GPU node can run.
int ecsimd_hip_group_rccl_selftest(ecsimd_hip_group* g, size_t elements) {
  if (!g || elements == 0) return ECSIMD_HIP_ERR_BAD_ARG;
  member& root = g->m[0];
  rccl_api api;
  if (!api.load()) return gfail(g, ECSIMD_HIP_ERR_HIP, "RCCL selftest", "librccl.so could not be loaded");
  int rc = ECSIMD_HIP_OK;
  ncclComm_t comm = nullptr;
  uint64_t* buf = nullptr;
  hipError_t e = hipSetDevice(root.device);
  if (e == hipSuccess) e = hipMalloc(&buf, 2 * elements * 32);
  std::vector<uint64_t> h(elements * 4), back(elements * 4, 0);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x9e3779b97f4a7c15ull * (i + 1);
  if (e == hipSuccess) e = hipMemcpyAsync(buf, h.data(), elements * 32, hipMemcpyHostToDevice, root.stream);
  if (e == hipSuccess) e = hipMemsetAsync(buf + elements * 4, 0, elements * 32, root.stream);
  if (e != hipSuccess) rc = gfail(g, ECSIMD_HIP_ERR_HIP, "RCCL selftest setup", hipGetErrorString(e));
  if (rc == ECSIMD_HIP_OK) {
    const int dev = root.device;
    ncclResult_t r = api.CommInitAll(&comm, 1, &dev);
    if (r == ncclSuccess) r = api.GroupStart();
    if (r == ncclSuccess) r = api.Recv(buf + elements * 4, elements * 4, ncclUint64, 0, comm, root.stream);
    if (r == ncclSuccess) r = api.Send(buf, elements * 4, ncclUint64, 0, comm, root.stream);
    const ncclResult_t r2 = (r == ncclSuccess) ? api.GroupEnd() : r;
    if (r2 != ncclSuccess) rc = gfail(g, ECSIMD_HIP_ERR_HIP, "RCCL selftest", api.GetErrorString(r2));
  }
  if (rc == ECSIMD_HIP_OK) {
    e = hipMemcpyAsync(back.data(), buf + elements * 4, elements * 32, hipMemcpyDeviceToHost, root.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(root.stream);
    if (e != hipSuccess) rc = gfail(g, ECSIMD_HIP_ERR_HIP, "RCCL selftest read-back", hipGetErrorString(e));
    else if (back != h) rc = gfail(g, ECSIMD_HIP_ERR_HIP, "RCCL selftest", "the received bytes differ from the sent ones");
  }
  (void)hipStreamSynchronize(root.stream);
  if (comm) (void)api.CommDestroy(comm);
  if (buf) (void)hipFree(buf);
  (void)dlclose(api.lib);
  return rc;
}

// Device-resident form.  k[m], x[m], y[m]: member m's shard (shard_range(n, m, G) elements) in ITS device memory.
// ox, oy, oz: n elements each in member 0's device memory (oz may be NULL with ECSIMD_HIP_OUT_AFFINE).  Asynchronous:
// ecsimd_hip_group_sync() waits and reports how long the gather took on member 0's stream.
int ecsimd_hip_group_scalar_mult(ecsimd_hip_group* g, int curve, const uint64_t* const* k, const uint64_t* const* x, const uint64_t* const* y,
                                 uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags) {
  if (!g) return ECSIMD_HIP_ERR_BAD_ARG;
  if (!k || !x || !y || !ox || !oy) return gfail(g, ECSIMD_HIP_ERR_BAD_ARG, "bad argument: null pointer");
  const bool affine = (flags & ECSIMD_HIP_OUT_AFFINE) != 0;
  if (!affine && !oz) return gfail(g, ECSIMD_HIP_ERR_BAD_ARG, "bad argument: oz is null");
  const int G = (int)g->m.size(), outs = affine ? 2 : 3;
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
    int rc = ecsimd_hip_scalar_mult(mb.ctx, curve, k[i], x[i], y[i], o[0], o[1], o[2], count, flags);
    if (rc != ECSIMD_HIP_OK) return gfail(g, rc, "member scalar_mult", ecsimd_hip_last_error(mb.ctx));
  }
  // ---- the one exchange: every other member's shard into member 0's arrays
  member& root = g->m[0];
  hipError_t e = hipSetDevice(root.device);
  if (e == hipSuccess) e = hipEventRecord(g->g0, root.stream);
  if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "gather", hipGetErrorString(e));
  if (!g->comm.empty()) {
    ncclResult_t r = g->rccl.GroupStart();
    for (int i = 1; i < G && r == ncclSuccess; ++i) {
      size_t first, count; (void)ecsimd_hip_shard_range(n, i, G, &first, &count);
      if (count == 0) continue;
      member& mb = g->m[(size_t)i];
      for (int c = 0; c < outs && r == ncclSuccess; ++c) {
        (void)hipSetDevice(root.device);
        r = g->rccl.Recv(root_out[c] + 4 * first, count * 4, ncclUint64, i, g->comm[0], root.stream);
        if (r != ncclSuccess) break;
        (void)hipSetDevice(mb.device);
        r = g->rccl.Send(mb.stage + (size_t)c * count * 4, count * 4, ncclUint64, 0, g->comm[(size_t)i], mb.stream);
      }
    }
    const ncclResult_t r2 = g->rccl.GroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "RCCL gather", g->rccl.GetErrorString(r != ncclSuccess ? r : r2));
  } else {
    for (int i = 1; i < G; ++i) {
      size_t first, count; (void)ecsimd_hip_shard_range(n, i, G, &first, &count);
      if (count == 0) continue;
      member& mb = g->m[(size_t)i];
      e = hipSetDevice(mb.device);
      if (e == hipSuccess) e = hipEventRecord(mb.done, mb.stream);
      if (e == hipSuccess) e = hipSetDevice(root.device);
      if (e == hipSuccess) e = hipStreamWaitEvent(root.stream, mb.done, 0);
      for (int c = 0; c < outs && e == hipSuccess; ++c)
        e = hipMemcpyAsync(root_out[c] + 4 * first, mb.stage + (size_t)c * count * 4, count * 32, hipMemcpyDeviceToDevice, root.stream);
      if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "gather copy", hipGetErrorString(e));
    }
  }
  e = hipSetDevice(root.device);
  if (e == hipSuccess) e = hipEventRecord(g->g1, root.stream);
  if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "gather", hipGetErrorString(e));
  g->timed = true;
  return ECSIMD_HIP_OK;
}

// Host-array form: k, x, y (n elements each) and the result arrays live in HOST memory.  Each member copies its slice
// in, computes, the shards are gathered to member 0 and copied out.  Synchronous (returns with the results in place).
int ecsimd_hip_group_scalar_mult_host(ecsimd_hip_group* g, int curve, const uint64_t* k, const uint64_t* x, const uint64_t* y,
                                      uint64_t* ox, uint64_t* oy, uint64_t* oz, size_t n, int flags) {
  if (!g) return ECSIMD_HIP_ERR_BAD_ARG;
  if (!k || !x || !y || !ox || !oy) return gfail(g, ECSIMD_HIP_ERR_BAD_ARG, "bad argument: null pointer");
  const bool affine = (flags & ECSIMD_HIP_OUT_AFFINE) != 0;
  if (!affine && !oz) return gfail(g, ECSIMD_HIP_ERR_BAD_ARG, "bad argument: oz is null");
  if (n == 0) return ECSIMD_HIP_OK;
  const int G = (int)g->m.size(), outs = affine ? 2 : 3;
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
  member& root = g->m[0];
  hipError_t e = hipSetDevice(root.device);
  uint64_t* dst[3] = {ox, oy, oz};
  for (int c = 0; c < outs && e == hipSuccess; ++c) e = hipMemcpyAsync(dst[c], res[c], n * 32, hipMemcpyDeviceToHost, root.stream);
  if (e != hipSuccess) return gfail(g, ECSIMD_HIP_ERR_HIP, "group d2h", hipGetErrorString(e));
  return ecsimd_hip_group_sync(g, nullptr);
}

}  // extern "C"
