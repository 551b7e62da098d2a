// ecsimd/serialization.h -- 32 big-endian bytes <-> little-endian limb order.
// Same contract as the reference (serialization.h:12-48): limb i is read from
// bytes[(nlimbs-1-i)*8 .. +8) big-endian.
#ifndef ECSIMD_SERIALIZATION_H
#define ECSIMD_SERIALIZATION_H
#include <ecsimd/bignum.h>
#include <array>

namespace ecsimd {
template <class Bignum> constexpr Bignum bn_from_bytes_BE(const uint8_t* bytes) {
  Bignum r;
  for (size_t i = 0; i < Bignum::nlimbs; ++i) {
    uint64_t v = 0;
    for (size_t b = 0; b < 8; ++b) v = (v << 8) | bytes[(Bignum::nlimbs - 1 - i) * 8 + b];
    r.limbs[i] = v;
  }
  return r;
}
template <class Bignum> constexpr Bignum bn_from_bytes_BE(std::array<uint8_t, Bignum::nlimbs * 8> const& bytes) { return bn_from_bytes_BE<Bignum>(bytes.data()); }
template <class Bignum> void bn_to_bytes_BE(uint8_t* out, Bignum const& v) {
  for (size_t i = 0; i < Bignum::nlimbs; ++i)
    for (size_t b = 0; b < 8; ++b) out[(Bignum::nlimbs - 1 - i) * 8 + b] = uint8_t(v.limbs[i] >> (8 * (7 - b)));
}
template <class Bignum> std::array<uint8_t, Bignum::nlimbs * 8> bn_to_bytes_BE(Bignum const& v) {
  std::array<uint8_t, Bignum::nlimbs * 8> r{}; bn_to_bytes_BE(r.data(), v); return r;
}
}  // namespace ecsimd
#endif
