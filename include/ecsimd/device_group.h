// ecsimd/device_group.h -- one batch over several GPUs from the C++ host API (C ABI: ecsimd_hip_group_*).
// The reference is single-threaded and has nothing of the kind; this is SURVEY.md 8(e) /
// BASELINE.json north_star behind the repo's own C++: contiguous shards, one context per device, no collective on
// the data path, the result shards gathered to the first device by one RCCL exchange.
#ifndef ECSIMD_DEVICE_GROUP_H
#define ECSIMD_DEVICE_GROUP_H
#include <ecsimd/hip_runtime.h>

#include <utility>
#include <vector>

namespace ecsimd {
namespace hip {
class device_group {
 public:
  // devices: HIP device indices, one member each (a device may be listed twice: those members exchange by device copies)
  explicit device_group(std::vector<int> const& devices) {
    const int rc = ecsimd_hip_group_init(devices.data(), (int)devices.size(), &g_);
    if (rc != ECSIMD_HIP_OK) throw error("ecsimd_hip_group_init failed (" + std::to_string(rc) + "): every member needs a gfx950 device; there is no CPU fallback");
  }
  device_group(device_group const&) = delete;
  device_group& operator=(device_group const&) = delete;
  ~device_group() { if (g_) ecsimd_hip_group_destroy(g_); }
  int size() const { return ecsimd_hip_group_size(g_); }
  bool uses_rccl() const { return ecsimd_hip_group_uses_rccl(g_) == 1; }
  int rccl_version() const { return ecsimd_hip_group_rccl_version(g_); }     // e.g. 22606; 0 when the gather needs no RCCL
  ecsimd_hip_group* handle() const { return g_; }
  // (first index, count) of member m: the partition every group entry point uses
  static std::pair<size_t, size_t> shard_range(size_t n, int member, int members) {
    size_t first = 0, count = 0;
    if (ecsimd_hip_shard_range(n, member, members, &first, &count) != ECSIMD_HIP_OK) throw error("ecsimd_hip_shard_range: bad member index");
    return {first, count};
  }
  void check(int rc, const char* what) const {
    if (rc != ECSIMD_HIP_OK) throw error(std::string(what) + " failed (" + std::to_string(rc) + "): " + ecsimd_hip_group_last_error(g_));
  }
 private:
  ecsimd_hip_group* g_ = nullptr;
};
}  // namespace hip
}  // namespace ecsimd
#endif
