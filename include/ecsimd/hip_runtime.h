// ecsimd/hip_runtime.h -- host-side plumbing shared by the C++ API headers: the process-wide
// engine context and a reference-counted device buffer.  Everything here sits on the C ABI of
// <ecsimd_hip.h>; there is no CPU implementation behind these headers -- if the HIP library or a
// gfx950 device is missing, the first use throws.
#ifndef ECSIMD_HIP_RUNTIME_H
#define ECSIMD_HIP_RUNTIME_H

#include <ecsimd_hip.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace ecsimd {
namespace hip {

struct error : std::runtime_error { using std::runtime_error::runtime_error; };

// One context per process (device from ECSIMD_HIP_DEVICE, default 0), created on first use.
inline ecsimd_hip_ctx* context() {
  struct holder {
    ecsimd_hip_ctx* ctx = nullptr;
    holder() {
      const char* d = std::getenv("ECSIMD_HIP_DEVICE");
      const int rc = ecsimd_hip_init(d ? std::atoi(d) : 0, &ctx);
      if (rc != ECSIMD_HIP_OK)
        throw error("ecsimd_hip_init failed (" + std::to_string(rc) + "): no gfx950 device or HIP runtime; this library has no CPU fallback");
    }
    ~holder() { if (ctx) ecsimd_hip_destroy(ctx); }
  };
  static holder h;
  return h.ctx;
}

inline void check(int rc, const char* what) {
  if (rc != ECSIMD_HIP_OK) throw error(std::string(what) + " failed (" + std::to_string(rc) + "): " + ecsimd_hip_last_error(context()));
}
inline void sync() { check(ecsimd_hip_sync(context()), "ecsimd_hip_sync"); }

// Freed device blocks are kept for reuse by size (hipMalloc / hipFree synchronise and cost ~0.1 ms each, more than an
// element-wise kernel over a million elements).  Reuse is safe without events: every kernel and copy of this
// API runs on the context's one stream, so a block's next user is ordered after its last.  At most `cap` bytes
// are parked; beyond that blocks go back to the driver.
class block_pool {
 public:
  static block_pool& instance() { static block_pool p; return p; }
  void* take(size_t bytes) {
    auto it = parked_.find(bytes);
    if (it != parked_.end()) { void* p = it->second; parked_.erase(it); held_ -= bytes; return p; }
    void* p = nullptr;
    check(ecsimd_hip_malloc(context(), &p, bytes), "ecsimd_hip_malloc");
    return p;
  }
  void give(void* p, size_t bytes) {
    if (held_ + bytes > cap) { ecsimd_hip_free(context(), p); return; }
    parked_.emplace(bytes, p); held_ += bytes;
  }
  ~block_pool() { for (auto& kv : parked_) ecsimd_hip_free(context(), kv.second); }
 private:
  block_pool() { (void)context(); }                    // the context outlives the pool (constructed first)
  static constexpr size_t cap = (size_t)8 << 30;
  std::multimap<size_t, void*> parked_;
  size_t held_ = 0;
};

// Device array of 64-bit words, shared between copies of a wide value (values are immutable once
// produced, like registers; in-place updates clone first -- see wide_bignum::unshare()).
class buffer {
 public:
  buffer() = default;
  explicit buffer(size_t words) : words_(words) {
    const size_t bytes = (words ? words : 2) * sizeof(uint64_t);
    void* p = block_pool::instance().take(bytes);
    mem_ = std::shared_ptr<uint64_t>(static_cast<uint64_t*>(p), [bytes](uint64_t* q) { block_pool::instance().give(q, bytes); });
  }
  uint64_t* data() const { return mem_.get(); }
  size_t words() const { return words_; }
  bool shared() const { return mem_.use_count() > 1; }
  void upload(const uint64_t* src) { check(ecsimd_hip_memcpy_h2d(context(), mem_.get(), src, words_ * 8), "h2d"); }
  void download(uint64_t* dst) const { check(ecsimd_hip_memcpy_d2h(context(), dst, mem_.get(), words_ * 8), "d2h"); }
  buffer clone() const {
    buffer b(words_);
    check(ecsimd_hip_memcpy_d2d(context(), b.mem_.get(), mem_.get(), words_ * 8), "d2d");
    return b;
  }
 private:
  std::shared_ptr<uint64_t> mem_;
  size_t words_ = 0;
};

// One flag per lane (the reference's eve::logical lane mask, bignum.h:136-137).
class mask {
 public:
  mask() = default;
  explicit mask(size_t n) : n_(n) {
    const size_t bytes = (n + 15) / 16 * 16 + 16;
    void* p = block_pool::instance().take(bytes);
    mem_ = std::shared_ptr<uint8_t>(static_cast<uint8_t*>(p), [bytes](uint8_t* q) { block_pool::instance().give(q, bytes); });
  }
  mask(std::initializer_list<bool> v) : mask(v.size()) {
    std::vector<uint8_t> h; for (bool b : v) h.push_back(b ? 1 : 0);
    check(ecsimd_hip_memcpy_h2d(context(), mem_.get(), h.data(), h.size()), "h2d");
  }
  static mask filled(size_t n, bool v) {
    mask m(n); std::vector<uint8_t> h(n, v ? 1 : 0);
    check(ecsimd_hip_memcpy_h2d(context(), m.mem_.get(), h.data(), n), "h2d"); return m;
  }
  uint8_t* data() const { return mem_.get(); }
  size_t size() const { return n_; }
  std::vector<uint8_t> host() const {
    std::vector<uint8_t> h(n_);
    if (n_) check(ecsimd_hip_memcpy_d2h(context(), h.data(), mem_.get(), n_), "d2h");
    return h;
  }
  bool get(size_t i) const { return host().at(i) != 0; }
  size_t count() const {                                 // lanes set (one reduction kernel + an 8-byte read-back)
    size_t c = 0; check(ecsimd_hip_mask_count(context(), mem_.get(), n_, &c), "ecsimd_hip_mask_count"); return c;
  }
  mask operator!() const { return combine(ECSIMD_HIP_MASK_NOT, *this, *this); }
  friend mask operator==(mask const& a, mask const& b) { return combine(ECSIMD_HIP_MASK_EQ, a, b); }
  friend mask operator&&(mask const& a, mask const& b) { return combine(ECSIMD_HIP_MASK_AND, a, b); }
  friend mask operator||(mask const& a, mask const& b) { return combine(ECSIMD_HIP_MASK_OR, a, b); }
 private:
  static mask combine(int op, mask const& a, mask const& b) {
    if (a.n_ != b.n_) throw error("ecsimd: lane masks of different length");
    mask m(a.n_);
    check(ecsimd_hip_mask_op(context(), op, a.mem_.get(), b.mem_.get(), m.mem_.get(), a.n_), "ecsimd_hip_mask_op");
    return m;
  }
  std::shared_ptr<uint8_t> mem_;
  size_t n_ = 0;
};

}  // namespace hip

// eve::all / eve::any / eve::none over a lane mask (the reference's tests use exactly these).
inline bool all(hip::mask const& m) { return m.count() == m.size(); }
inline bool any(hip::mask const& m) { return m.count() != 0; }
inline bool none(hip::mask const& m) { return !any(m); }

}  // namespace ecsimd
#endif
