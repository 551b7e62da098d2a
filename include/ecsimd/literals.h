// ecsimd/literals.h -- "…"_hex compile-time hex literal -> std::array<uint8_t, N/2>.
// Same spelling as the reference (literals.h:28-43) so test vectors paste unchanged.
#ifndef ECSIMD_LITERALS_H
#define ECSIMD_LITERALS_H
#include <array>
#include <cstddef>
#include <cstdint>

namespace ecsimd {
namespace literals {
namespace detail {
constexpr uint8_t nibble(char c) {
  return (c >= '0' && c <= '9') ? uint8_t(c - '0') : (c >= 'a' && c <= 'f') ? uint8_t(c - 'a' + 10) : (c >= 'A' && c <= 'F') ? uint8_t(c - 'A' + 10)
         : throw "invalid hex digit";
}
template <size_t N> struct hex_text {
  char s[N]{};
  constexpr hex_text(const char (&str)[N]) { for (size_t i = 0; i < N; ++i) s[i] = str[i]; }
};
}  // namespace detail
template <detail::hex_text T> constexpr auto operator""_hex() {
  constexpr size_t digits = sizeof(T.s) - 1;
  static_assert(digits % 2 == 0, "_hex needs an even number of digits");
  std::array<uint8_t, digits / 2> out{};
  for (size_t i = 0; i < digits / 2; ++i) out[i] = uint8_t(detail::nibble(T.s[2 * i]) << 4 | detail::nibble(T.s[2 * i + 1]));
  return out;
}
}  // namespace literals
}  // namespace ecsimd
#endif
