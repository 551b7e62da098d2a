// ecsimd/serialization.h -- big-endian byte strings <-> little-endian limb order, host side, one value at a time
// (the device-side batch codecs are in sec1.h).  Contract of the reference's serialization.h:12-48: the most
// significant limb comes first in the byte string, each limb most significant byte first.
#ifndef ECSIMD_SERIALIZATION_H
#define ECSIMD_SERIALIZATION_H
#include <ecsimd/bignum.h>
#include <array>

namespace ecsimd {
namespace detail {
// byte k of a big-endian string of `total` bytes carries bits [8 (total-1-k), +8) of the integer
constexpr size_t limb_of_byte(size_t total, size_t k) { return (total - 1 - k) / 8; }
constexpr unsigned shift_of_byte(size_t total, size_t k) { return 8u * unsigned((total - 1 - k) % 8); }
}  // namespace detail

template <class Bignum> constexpr Bignum bn_from_bytes_BE(const uint8_t* bytes) {
  constexpr size_t total = Bignum::nlimbs * sizeof(uint64_t);
  Bignum out{};
  for (size_t k = 0; k < total; ++k)
    out.limbs[detail::limb_of_byte(total, k)] |= uint64_t(bytes[k]) << detail::shift_of_byte(total, k);
  return out;
}
template <class Bignum> constexpr Bignum bn_from_bytes_BE(std::array<uint8_t, Bignum::nlimbs * 8> const& bytes) {
  return bn_from_bytes_BE<Bignum>(bytes.data());
}
template <class Bignum> void bn_to_bytes_BE(uint8_t* out, Bignum const& v) {
  constexpr size_t total = Bignum::nlimbs * sizeof(uint64_t);
  for (size_t k = 0; k < total; ++k)
    out[k] = uint8_t(v.limbs[detail::limb_of_byte(total, k)] >> detail::shift_of_byte(total, k));
}
template <class Bignum> std::array<uint8_t, Bignum::nlimbs * 8> bn_to_bytes_BE(Bignum const& v) {
  std::array<uint8_t, Bignum::nlimbs * 8> bytes{};
  bn_to_bytes_BE(bytes.data(), v);
  return bytes;
}
}  // namespace ecsimd
#endif
