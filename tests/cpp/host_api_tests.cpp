// host_api_tests.cpp -- the reference's known-answer scenarios (tests/ops.cpp, mgry.cpp, curve_point.cpp,
// curve_group.cpp of aguinet/ecsimd) driven through THIS repo's include/ecsimd headers, i.e. through
// the C ABI and the HIP kernels.  The hex vectors are the reference's (cited per test); the harness is
// this repo's mini_test.h.  Built and run by tests/test_cpp_host_api.py on the GPU box.
#include <ecsimd/ecsimd.h>
#include "mini_test.h"

using namespace ecsimd;
using namespace ecsimd::literals;

namespace {
template <class WBN, size_t N> WBN splat(std::array<uint8_t, N> const& be) {                 // tests/tests.h:10-14 wide_bignum_set1
  static_assert(N == sizeof(uint64_t) * WBN::nlimbs);
  return WBN{bn_from_bytes_BE<typename WBN::value_type>(be)};
}
template <class WBN, size_t N> WBN lanes(std::array<uint8_t, N> const& l0, std::array<uint8_t, N> const& l1, std::array<uint8_t, N> const& l2, std::array<uint8_t, N> const& l3) {
  const std::array<uint8_t, N> be[4] = {l0, l1, l2, l3};
  return WBN{[&](size_t i, size_t) { return bn_from_bytes_BE<typename WBN::value_type>(be[i % 4]); }};
}
using W128 = wide_bignum<bignum_128>;
using W256 = wide_bignum<bignum_256>;
using W512 = wide_bignum<bignum_512>;
struct K1P { static constexpr auto value = bn_from_bytes_BE<bignum_256>("FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F"_hex); };   // tests/mgry.cpp:25-27
}  // namespace

// ------------------------------------------------------------------ tests/ops.cpp:71-153
TEST(Ops128, AddSub) {
  EXPECT_TRUE(all(add_no_carry(splat<W128>("00000000000000000000000500000005"_hex), splat<W128>("0000000000000000FFFFFFFFFFFFFFFF"_hex)) == splat<W128>("00000000000000010000000500000004"_hex)));
  EXPECT_TRUE(all(add_no_carry(splat<W128>("909680e1f399ca5916134a18b816399b"_hex), splat<W128>("0e36dfecf5e7f74363c453efc1cbc153"_hex)) == splat<W128>("9ecd60cee981c19c79d79e0879e1faee"_hex)));
  EXPECT_TRUE(all(sub_no_carry(splat<W128>("00000000000000000000000500000005"_hex), splat<W128>("0000000000000000FFFFFFFFFFFFFFFF"_hex)) == splat<W128>("ffffffffffffffff0000000500000006"_hex)));
}
TEST(Ops128, SubIfAbove) {
  const auto p = splat<W128>("F0000000000000000000000000000004"_hex);
  EXPECT_TRUE(all(sub_if_above(splat<W128>("F0000000000000000000000000000005"_hex), p) == splat<W128>("00000000000000000000000000000001"_hex)));
  EXPECT_TRUE(all(sub_if_above(p, p) == splat<W128>("00000000000000000000000000000000"_hex)));
  EXPECT_TRUE(all(sub_if_above(splat<W128>("F0000000000000000000000000000003"_hex), p) == splat<W128>("F0000000000000000000000000000003"_hex)));
  // lane-distinct (tests/ops.cpp:93-119)
  const auto a = lanes<W128>("F0000000000000000000000000000005"_hex, "F0000000000000000000000000000004"_hex, "F0000000000000000000000000000003"_hex, "F0000000000000000000000000000002"_hex);
  const auto r = lanes<W128>("00000000000000000000000000000001"_hex, "00000000000000000000000000000000"_hex, "F0000000000000000000000000000003"_hex, "F0000000000000000000000000000002"_hex);
  EXPECT_TRUE(all(sub_if_above(a, p) == r));
}
TEST(Ops128, MulSquare) {
  EXPECT_TRUE(all(mul(splat<W128>("ffffffffffffffffffffffffffffffff"_hex), splat<W128>("eeeeeeeeeeeeeeeeeeeeeeeeeeeeeeee"_hex)) ==
                  splat<W256>("EEEEEEEEEEEEEEEEEEEEEEEEEEEEEEED11111111111111111111111111111112"_hex)));
  using W192 = wide_bignum<bignum<uint64_t, 3>>;
  EXPECT_TRUE(all(limb_mul(splat<W128>("e43aba669166dad6a334ad6bb13a2c9c"_hex), {198769}) == splat<W192>("000000000002b436c2f33005f5c13775b7eefdc191e690dc"_hex)));
  EXPECT_TRUE(all(square(splat<W128>("00000000000000000000000000000004"_hex)) == splat<W256>("0000000000000000000000000000000000000000000000000000000000000010"_hex)));
  EXPECT_TRUE(all(square(splat<W128>("ffffffffffffffffffffffffffffffff"_hex)) == splat<W256>("fffffffffffffffffffffffffffffffe00000000000000000000000000000001"_hex)));
  EXPECT_TRUE(all(square(splat<W128>("b59edca51009bb15c309b23171c102da"_hex)) == splat<W256>("80da06968299ac8e1bc23ef95d49c1469d01bb136df7c96b75ba357dc0bc21a4"_hex)));
}
TEST(Ops128, ZextTruncAndLimbShifts) {
  // tests/ops.cpp:122-127: trunc_u64x32(zext_u32x64(v)) == v
  const auto v = splat<W128>("b59edca51009bb15c309b23171c102da"_hex);
  const auto z = zext_u32x64(v);
  EXPECT_TRUE(all(z == splat<W256>("00000000b59edca5000000001009bb1500000000c309b2310000000071c102da"_hex)));
  EXPECT_TRUE(all(trunc_u64x32(z) == v));
  const auto q = lanes<W256>("80000000800000008000000080000000ffffffffffffffff0123456789abcdef"_hex, "0000000000000000000000000000000000000000000000000000000000000001"_hex,
                             "ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff"_hex, "00000000000000010000000000000002000000000000000300000000000000f4"_hex);
  EXPECT_TRUE(all(trunc_u64x32(zext_u32x64(q)) == q));
  // shift.h:53-96 (what mgry_reduce's zero-extended shifts are built from in the reference)
  using W192 = wide_bignum<bignum<uint64_t, 3>>;
  EXPECT_TRUE(all(limb_shift_left<4, 1>(v) == splat<W256>("0000000000000000b59edca51009bb15c309b23171c102da0000000000000000"_hex)));
  EXPECT_TRUE(all(limb_shift_left<2, 1>(v) == splat<W128>("c309b23171c102da0000000000000000"_hex)));                       // the high limb does not fit: dropped
  EXPECT_TRUE(all(limb_shift_left<3, 1>(v) == splat<W192>("b59edca51009bb15c309b23171c102da0000000000000000"_hex)));
  EXPECT_TRUE(all(limb_shift_left<3, 2>(v) == W192{bignum<uint64_t, 3>{}}));                                               // ShiftBy >= the operand's limbs: zero (shift.h:60-62)
  EXPECT_TRUE(all(limb_shift_left<2, 2>(v) == W128{bignum_128{}}));
  EXPECT_TRUE(all(limb_shift_right<1>(q) == lanes<W192>("80000000800000008000000080000000ffffffffffffffff"_hex, "000000000000000000000000000000000000000000000000"_hex,
                                                        "ffffffffffffffffffffffffffffffffffffffffffffffff"_hex, "000000000000000100000000000000020000000000000003"_hex)));
  // utility.h:36-43 wide_uasr: arithmetic shift of one limb of every lane
  const auto sh = wide_uasr(q, 3, 63);
  EXPECT_TRUE(sh[0] == ~0ull && sh[1] == 0 && sh[2] == ~0ull && sh[3] == 0);
  EXPECT_TRUE(wide_uasr(q, 0, 4)[3] == 0xfull && wide_uasr(q, 3, 4)[0] == 0xf800000008000000ull);
  // bignum.h:61-67 cbn() / from()
  constexpr auto one = bignum_256::from(bignum_256::cbn_type{1, 0, 0, 0});
  static_assert(one.cbn()[0] == 1 && one == bignum_256::from(1));
}
TEST(Ops128, Compare) {
  const auto lo = splat<W128>("AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA"_hex), hi = splat<W128>("BAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA"_hex);
  EXPECT_TRUE(all(lo < hi)); EXPECT_TRUE(all(lo <= hi)); EXPECT_TRUE(all(lo <= lo));
  EXPECT_TRUE(all(hi > lo)); EXPECT_TRUE(all(hi >= lo)); EXPECT_TRUE(all(lo >= lo));
  EXPECT_TRUE(none(splat<W128>("AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAB"_hex) < lo));
}
// tests/ops.cpp:155-177 (the carry mask is what the reference pins; the shifted value is checked here
// against the arithmetically correct a << 1, the reference's own value check being vacuous)
TEST(Ops128, Shifts) {
  const auto a = lanes<W128>("80000000800000008000000080000000"_hex, "70000000800000001000000000000001"_hex, "00000000000000000000000000000001"_hex, "FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF"_hex);
  auto [shifted, carry] = shift_left_one(a);
  EXPECT_TRUE(all(carry == cmp_res_t<W128>{true, false, false, true}));
  EXPECT_TRUE(all(shifted == lanes<W128>("00000001000000010000000100000000"_hex, "E0000001000000002000000000000002"_hex, "00000000000000000000000000000002"_hex, "FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFE"_hex)));
}
TEST(Ops128, Swap) {                                                                          // tests/ops.cpp:179-208
  const auto a = lanes<W128>("00000001000000010000000100000000"_hex, "80000001000000002000000000000002"_hex, "00000000000000000000000000000002"_hex, "FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFE"_hex);
  const auto b = lanes<W128>("FFFEEEE100AAAAA100DDDDD10FFFAAAA"_hex, "8BBBB001000FFF002000AAAAAAA00002"_hex, "DDDDDDDAAAAAAAFFFFFFF24566660002"_hex, "0000000111111144444555555FFFFFFE"_hex);
  const auto Z = hip::mask::filled(4, false);
  auto aa = a, bb = b;
  swap_if(Z, aa, bb);
  EXPECT_TRUE(all(aa == a)); EXPECT_TRUE(all(bb == b));
  swap_if(!Z, aa, bb);
  EXPECT_TRUE(all(aa == b)); EXPECT_TRUE(all(bb == a));
  EXPECT_TRUE(all(a == lanes<W128>("00000001000000010000000100000000"_hex, "80000001000000002000000000000002"_hex, "00000000000000000000000000000002"_hex, "FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFE"_hex)));   // the originals are untouched
}
TEST(Ops256, Mul) {                                                                           // tests/ops.cpp:210-219
  EXPECT_TRUE(all(mul(splat<W256>("ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff"_hex), splat<W256>("eeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeee"_hex)) ==
                  splat<W512>("EEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEED1111111111111111111111111111111111111111111111111111111111111112"_hex)));
}
TEST(Ops256, Mod) {                                                                           // tests/ops.cpp:221-252
  const auto p = splat<W256>("FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F"_hex);
  EXPECT_TRUE(all(mod_add(splat<W256>("FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2E"_hex), splat<W256>("0000000000000000000000000000000000000000000000000000000000000002"_hex), p) ==
                  splat<W256>("0000000000000000000000000000000000000000000000000000000000000001"_hex)));
  const auto a = splat<W256>("fffffffffffffffffffffffffffffffffffffffffffffffffffffff000000000"_hex);
  const auto b = splat<W256>("ffeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeee"_hex);
  EXPECT_TRUE(all(mod_add(a, b, p) == splat<W256>("ffeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeedfeeeef2bf"_hex)));
  EXPECT_TRUE(all(mod_sub(a, b, p) == splat<W256>("0011111111111111111111111111111111111111111111111111110111111112"_hex)));
  EXPECT_TRUE(all(mod_shift_left_one(a, p) == splat<W256>("ffffffffffffffffffffffffffffffffffffffffffffffffffffffe1000003d1"_hex)));
}

// ------------------------------------------------------------------ tests/mgry.cpp
TEST(Mgry, FromTo) {                                                                          // :32-50
  using WMBN = wide_mgry_bignum<W256, K1P>;
  for (auto const& v : {"eeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeee"_hex, "0168db3a8eca3fd7d4d08943182e189aef318068ba8853d77cb49c17bae00c0e"_hex,
                        "2714dac0b974321b75d6ef64e7c3b118adb2801bf674282df5712cd2af390f79"_hex, "a3fc64fece6f3e1effab4045a9a54faa49a228f787025f0ecb761145755cb2d0"_hex,
                        "3af178b78710adae9cc096188ed09c210078aaa7e965ef83d22a91f21fec4eb5"_hex, "688c743cde3987e299d2b028038ddc12dc02e7033c9d3c8f4d20edf9544232aa"_hex,
                        "45e29166c6441f0fd27e3b85a205f1e102b025cc8e8ea158ab4885a22ed68905"_hex}) {
    const auto a = splat<W256>(v);
    EXPECT_TRUE(all(WMBN::from_classical(a).to_classical() == a));
  }
}
TEST(Mgry, ConstexprToMgry) {                                                                 // mgry.h:18-26
  using WMBN = wide_mgry_bignum<W256, K1P>;
  constexpr auto four = bn_from_bytes_BE<bignum_256>("FFFFFFFFFFFFFFFFFFFFFF000000000000000000000000000000000000000004"_hex);
  constexpr auto m = to_mgry<K1P>(four);                                                        // a compile-time constant, like the reference's Am / Bm
  static_assert(to_mgry<K1P>(bignum_256::from(1)) == bn_from_bytes_BE<bignum_256>("00000000000000000000000000000000000000000000000000000001000003d1"_hex));   // R mod p
  EXPECT_TRUE(all(WMBN::from_classical(W256{four}).wbn() == W256{m}));
  EXPECT_TRUE(all(WMBN::R().wbn() == W256{to_mgry<K1P>(bignum_256::from(1))}));                 // mgry.h:43
  using P256P = curve_nist_p256::P;
  static_assert(to_mgry<P256P>(bignum_256::from(1)) == bn_from_bytes_BE<bignum_256>("00000000fffffffeffffffffffffffffffffffff000000000000000000000001"_hex));
  EXPECT_TRUE(all(W256{to_mgry<P256P>(curve_nist_p256::Gx::value)} == curve_group<curve_nist_p256>::WJG().x().wbn()));
  // a value >= p is reduced first
  constexpr auto big = bn_from_bytes_BE<bignum_256>("FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF"_hex);
  EXPECT_TRUE(all(wide_mgry_bignum<W256, K1P>::from_classical(W256{big}).wbn() == W256{to_mgry<K1P>(big)}));
}
TEST(Mgry, Reduce) {                                                                          // :52-76: reduce(mul(a,b)) == a*b*R^-1 == mgry_mul
  using WMBN = wide_mgry_bignum<W256, K1P>;
  const auto a = splat<W256>("00000000000AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA"_hex), b = splat<W256>("00000000000BBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBBB"_hex);
  EXPECT_TRUE(all(details::mgry_reduce<K1P>(mul(a, b)) == mgry_mul(WMBN{a}, WMBN{b}).wbn()));
  const auto four = splat<W256>("0000000000000000000000000000000000000000000000000000000000000004"_hex), five = splat<W256>("0000000000000000000000000000000000000000000000000000000000000005"_hex);
  // 4*5*R^-1, brought back with one from_classical (x R), is 20
  const auto r = details::mgry_reduce<K1P>(mul(four, five));
  EXPECT_TRUE(all(WMBN::from_classical(r).wbn() == splat<W256>("0000000000000000000000000000000000000000000000000000000000000014"_hex)));
}
TEST(Mgry, Ops) {                                                                             // :78-120
  using WMBN = wide_mgry_bignum<W256, K1P>;
  const auto ma = WMBN::from_classical(splat<W256>("FFFFFFFFFFFFFFFFFFFFFF000000000000000000000000000000000000000004"_hex));
  const auto mb = WMBN::from_classical(splat<W256>("FFFFFFFFFFFFFFFFFFFFFF000000000000000000000000000000000000000005"_hex));
  EXPECT_TRUE(all((ma + mb).to_classical() == splat<W256>("fffffffffffffffffffffe0000000000000000000000000000000001000003da"_hex)));
  EXPECT_TRUE(all((ma - mb).to_classical() == splat<W256>("fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2e"_hex)));
  EXPECT_TRUE(all((mb - ma).to_classical() == splat<W256>("0000000000000000000000000000000000000000000000000000000000000001"_hex)));
  struct { std::array<uint8_t, 32> e, r; } pows[] = {
    {"FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2D"_hex, "DC1B98237FD316F9AEE7342E6DC7629A75A99A9E9EF591170282CE3E1D8E26ED"_hex},
    {"0000000000000000000000000000000000000000000000000000000000000002"_hex, "fffffffffffffdfffff85600000000000001000003d10001000007a9000eab68"_hex},
    {"00000000000F0000000000000000000000000000000000000000000000000001"_hex, "a51e978903ca7fcd788382ff283366ad7457d27c7aac417127a8723626773516"_hex},
    {"0000000000000000000000000000000000000000000000000000000000000000"_hex, "0000000000000000000000000000000000000000000000000000000000000001"_hex}};
  for (auto const& c : pows) EXPECT_TRUE(all(mgry_pow(ma, bn_from_bytes_BE<bignum_256>(c.e)).to_classical() == splat<W256>(c.r)));
}
TEST(Mgry, Gfp) {                                                                             // :122-150
  using GFP = GFp<W256, K1P>;
  EXPECT_TRUE(all(GFP::from_classical(splat<W256>("FFFFFFFFFFFFFFFFFFFFFF000000000000000000000000000000000000000004"_hex)).inverse().to_classical() ==
                  splat<W256>("DC1B98237FD316F9AEE7342E6DC7629A75A99A9E9EF591170282CE3E1D8E26ED"_hex)));
  const auto ma = GFP::from_classical(splat<W256>("b560fd7b259468b53c3a1623f35786a491fcb1fcdfbb0165da4dccce1f185b60"_hex));
  const auto root = ma.sqrt();
  EXPECT_TRUE(root.has_value());
  if (root) EXPECT_TRUE(all(root->to_classical() == splat<W256>("a59f1be7c1f892ff2adf14187e9cff7666112af579bc1a11b63e248098567e71"_hex)));
  const auto Z = ma + ma.opposite();
  EXPECT_TRUE(all(Z.wbn() == W256{bignum_256{}}));
}

// ------------------------------------------------------------------ the field layer is generic in P (mgry_mul.h:84-121, gfp.h:17-115)
namespace {
struct P192 { static constexpr auto value = bn_from_bytes_BE<bignum_256>("0000000000000000FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFFFFFFFFFFFF"_hex); };   // P-192's prime: a modulus of no curve here
}
TEST(Mgry, AnyModulus) {
  using GFP = GFp<W256, P192>;                                                                  // = 3 mod 4, as gfp.h:84 wants
  const auto a = GFP::from_classical(splat<W256>("0000000000000000123456789abcdef0fedcba9876543210aabbccddeeff0011"_hex));
  const auto b = GFP::from_classical(splat<W256>("00000000000000000f1e2d3c4b5a69788796a5b4c3d2e1f00112233445566778"_hex));
  EXPECT_TRUE(all((a * b).to_classical() == splat<W256>("00000000000000002caf945261384f2410d8a623fcdc6a3f7a4f03ba95b7c40c"_hex)));
  EXPECT_TRUE(all(a.inverse().to_classical() == splat<W256>("00000000000000001145f09c0965e40d3a322db8613aa048e7e16376a1b07a1e"_hex)));
  EXPECT_TRUE(all(a.opposite().to_classical() == splat<W256>("0000000000000000edcba9876543210f0123456789abcdee554433221100ffee"_hex)));
  const auto root = a.sqr().sqrt();
  EXPECT_TRUE(root.has_value());
  if (root) EXPECT_TRUE(all(root->sqr().wbn() == a.sqr().wbn()));
  EXPECT_TRUE(all((a + a.opposite()).wbn() == W256{bignum_256{}}));
  // compile-time to_mgry for a modulus below 2^255, against the engine's from_classical
  constexpr auto am = to_mgry<P192>(bn_from_bytes_BE<bignum_256>("0000000000000000123456789abcdef0fedcba9876543210aabbccddeeff0011"_hex));
  EXPECT_TRUE(all(a.wbn() == W256{am}));
  EXPECT_TRUE((mgry_constants<P192>::R_p() == to_mgry<P192>(bignum_256::from(1))));
}
TEST(Mgry, GroupOrderArithmetic) {
  // u1 = e / s, u2 = r / s modulo the order of P-256 with the reference's own vocabulary (RFC 6979 A.2.5, SHA-256, "sample")
  using Fn = GFp<W256, p256_order>;
  const auto e = splat<W256>("AF2BDBE1AA9B6EC1E2ADE1D694F41FC71A831D0268E9891562113D8A62ADD1BF"_hex);
  const auto r = splat<W256>("EFD48B2AACB6A8FD1140DD9CD45E81D69D2C877B56AAF991C34D0EA84EAF3716"_hex);
  const auto s = splat<W256>("F7CB1C942D657C41D436C7A1B6E29F65F3E900DBB9AFF4064DC4AB2F843ACDA8"_hex);
  const auto w = Fn::from_classical(s).inverse();
  EXPECT_TRUE(all((Fn::from_classical(e) * w).to_classical() == splat<W256>("a9cceaf9beeb5f3ef17670f8eb7f810b486952f78536ee77f31cff76caae5841"_hex)));
  EXPECT_TRUE(all((Fn::from_classical(r) * w).to_classical() == splat<W256>("48dc5acda3b1ad61b01f62f0ec7e692d6b6ca086e80a10b4241298ec71e7211d"_hex)));
}
TEST(Ecdsa, Rfc6979Vectors) {
  // RFC 6979 A.2.5 (P-256, SHA-256): "sample" and "test" under one key; then each input disturbed in turn
  using Curve = curve_nist_p256; using CG = curve_group<Curve>; using WCP = wide_curve_point<Curve>;
  const auto qx = "60FED4BA255A9D31C961EB74C6356D68C049B8923B61FA6CE669622E60F29FB6"_hex, qy = "7903FE1008B8BC99A41AE9E95628BC64F2F1B20C2D7E9F5177A3C294D4462299"_hex;
  const auto e1 = "AF2BDBE1AA9B6EC1E2ADE1D694F41FC71A831D0268E9891562113D8A62ADD1BF"_hex, e2 = "9F86D081884C7D659A2FEAA0C55AD015A3BF4F1B2B0B822CD15D6C15B0F00A08"_hex;
  const auto r1 = "EFD48B2AACB6A8FD1140DD9CD45E81D69D2C877B56AAF991C34D0EA84EAF3716"_hex, s1 = "F7CB1C942D657C41D436C7A1B6E29F65F3E900DBB9AFF4064DC4AB2F843ACDA8"_hex;
  const auto r2 = "F1ABB023518351CD71D881567B1EA663ED3EFCF6C5132B354F28D3B0B7D38367"_hex, s2 = "019F4113742A2B14BD25926B49C649155F267E60D3814B4C0CC84250E46F0083"_hex;
  const auto zero = "0000000000000000000000000000000000000000000000000000000000000000"_hex;
  const WCP Q{splat<W256>(qx), splat<W256>(qy)};
  //                                  valid  valid  wrong message  r of the other  s = 0
  const auto e = lanes<W256>(e1, e2, e2, e1), r = lanes<W256>(r1, r2, r1, r2), s = lanes<W256>(s1, s2, s1, zero);
  const auto ok = CG::ecdsa_verify(e, r, s, Q).host();
  EXPECT_TRUE(ok[0] == 1 && ok[1] == 1 && ok[2] == 0 && ok[3] == 0);
  const WCP off{splat<W256>(qx), splat<W256>(qx)};                                              // not a curve point
  EXPECT_TRUE(CG::ecdsa_verify(e, r, s, off).count() == 0);
  // signing with the RFC's own nonces (A.2.5: k for "sample" and "test") reproduces its signatures; a zero nonce is refused
  const auto x = "C9AFA9D845BA75166B5C215767B1D6934E50C3DB36E89B127B8A622B120F6721"_hex;
  const auto k1 = "A6E3C57DD01ABE90086538398355DD4C3B17AA873382B0F24D6129493D8AAD60"_hex, k2 = "D16B6AE827F17175E040871A1C7EC3500192C4C92677336EC2537ACAEE0008E0"_hex;
  hip::mask signed_ok;
  const auto sig = CG::ecdsa_sign(lanes<W256>(e1, e2, e1, e2), splat<W256>(x), lanes<W256>(k1, k2, zero, k1), signed_ok);
  const auto so = signed_ok.host();
  EXPECT_TRUE(so[0] == 1 && so[1] == 1 && so[2] == 0 && so[3] == 1);
  EXPECT_TRUE(sig.first.get(0) == bn_from_bytes_BE<bignum_256>(r1) && sig.second.get(0) == bn_from_bytes_BE<bignum_256>(s1));
  EXPECT_TRUE(sig.first.get(1) == bn_from_bytes_BE<bignum_256>(r2) && sig.second.get(1) == bn_from_bytes_BE<bignum_256>(s2));
  EXPECT_TRUE(sig.first.get(2) == bignum_256{} && sig.second.get(2) == bignum_256{});
  const auto again = CG::ecdsa_verify(lanes<W256>(e1, e2, e1, e2), sig.first, sig.second, Q).host();
  EXPECT_TRUE(again[0] == 1 && again[1] == 1 && again[2] == 0 && again[3] == 1);                // lane 3: "test" signed with the other nonce: valid all the same
}

// ------------------------------------------------------------------ tests/curve_point.cpp
TEST(CurvePoint, FromXAndRoundTrip) {
  using Curve = curve_nist_p256; using WCP = wide_curve_point<Curve>; using WJCP = wide_jacobian_curve_point<Curve>;
  const auto x = splat<W256>("ce11d601ec0e947529e66021a0cd3d57518d58d0d5f2eb7ed75805d78c986e60"_hex);
  const auto pt = WCP::from_x(x);
  EXPECT_TRUE(pt.has_value());
  if (!pt) return;
  EXPECT_TRUE(all(pt->y() == splat<W256>("f2a40cfbb248ae2c7749c76641b51b7137ccad8916931adf83b857e418fad591"_hex)));
  EXPECT_TRUE(all(WJCP::from_affine(*pt).to_affine() == *pt));
  EXPECT_FALSE(WCP::from_x(splat<W256>("0000000000000000000000000000000000000000000000000000000000000005"_hex)).has_value() &&
               WCP::from_x(splat<W256>("0000000000000000000000000000000000000000000000000000000000000006"_hex)).has_value() &&
               WCP::from_x(splat<W256>("0000000000000000000000000000000000000000000000000000000000000007"_hex)).has_value());   // not every x is on the curve
}

// ------------------------------------------------------------------ tests/curve_group.cpp
namespace {
using Curve = curve_nist_p256; using CG = curve_group<Curve>;
const auto G2x = "7cf27b188d034f7e8a52380304b51ac3c08969e277f21b35a60b48fc47669978"_hex, G2y = "07775510db8ed040293d9ac69f7430dbba7dade63ce982299e04b79d227873d1"_hex;
const auto G3x = "5ecbe4d1a6330a44c8f7ef951d4bf165e6c6b721efada985fb41661bc6e7fd6c"_hex, G3y = "8734640c4998ff7e374b06ce1a64a2ecd82ab036384fb83d9a79b127a27d5032"_hex;
const auto G5x = "51590b7a515140d2d784c85608668fdfef8c82fd1f5be52421554a0dc3d033ed"_hex, G5y = "e0c17da8904a727d8ae1bf36bf8a79260d012f00d4d80888d1d0bb44fda16da4"_hex;
template <class WJ, class A> bool affine_is(WJ const& J, A const& x, A const& y) { const auto p = J.to_affine(); return all(p.x() == splat<W256>(x)) && all(p.y() == splat<W256>(y)); }
}
// per-lane decompression and membership (beyond the reference's all-or-nothing from_x), both curves
template <class K> static void lanes_membership() {
  using KG = curve_group<K>; using WCP = wide_curve_point<K>;
  const size_t n = 64;
  W256 s(n, [](size_t i, size_t) { bignum_256 b; b.limbs = {i * 0x9e3779b97f4a7c15ull + 11, i, 5, 0x1000000000000000ull + i}; return b; });
  const auto P = KG::scalar_mult(s, KG::WJG(n)).to_affine();                       // points on the curve
  EXPECT_TRUE(all(P.on_curve()));
  hip::mask valid;
  const auto Q = WCP::from_x_lanes(P.x(), valid);
  EXPECT_TRUE(all(valid)); EXPECT_TRUE(all(Q.on_curve()));
  EXPECT_TRUE(all(Q.x() == P.x()));
  auto bad = P.y().host(); bad[3].limbs[0] ^= 1; bad[40].limbs[2] ^= 0x10;          // two lanes leave the curve
  const auto flags = WCP{P.x(), W256(bad)}.on_curve().host();
  for (size_t i = 0; i < n; ++i) EXPECT_TRUE((flags[i] != 0) == (i != 3 && i != 40));
  W256 xs(n, [](size_t i, size_t) { bignum_256 b; b.limbs = {i + 1, 0, 0, 0}; return b; });   // small x: about half are abscissae
  const auto R = WCP::from_x_lanes(xs, valid);
  const auto v = valid.host(); const auto on = R.on_curve().host();
  size_t good = 0;
  for (size_t i = 0; i < n; ++i) { good += v[i] != 0; EXPECT_TRUE((v[i] != 0) == (on[i] != 0)); }
  EXPECT_TRUE(good > n / 4 && good < 3 * n / 4);
}
TEST(CurvePoint, PerLaneDecompressionAndMembership) {
  lanes_membership<curve_nist_p256>();
  lanes_membership<curve_secp256k1>();
}

TEST(CurveGroup, DBLU) {                                                                      // :38-52
  auto G = CG::WJG();
  const auto D = CG::DBLU(G);
  EXPECT_TRUE(all(G.z().wbn() == D.z().wbn()));
  EXPECT_TRUE(all(G.to_affine() == CG::WG()));
  EXPECT_TRUE(affine_is(D, G2x, G2y));
}
TEST(CurveGroup, ZADDU_TRPLU) {                                                               // :54-76
  auto G = CG::WJG();
  const auto D = CG::DBLU(G);
  const auto T = CG::ZADDU(G, D);
  EXPECT_TRUE(all(T.z().wbn() == G.z().wbn()));
  EXPECT_TRUE(affine_is(T, G3x, G3y));
  G = CG::WJG();
  EXPECT_TRUE(affine_is(CG::TRPLU(G), G3x, G3y));
}
TEST(CurveGroup, ZDAU) {                                                                      // :78-94
  auto G = CG::WJG();
  auto D = CG::DBLU(G);
  const auto F = CG::ZDAU(D, G);
  EXPECT_TRUE(all(F.z().wbn() == G.z().wbn()));
  EXPECT_TRUE(affine_is(F, G5x, G5y));
}
TEST(CurveGroup, SharedZKeepsValueSemantics) {
  // The co-Z formulas hand out ONE Z array for the result and the rewritten operand (written once on the device).  A later
  // in-place update of either must not show through the other: D stays 2G over the Z it was born with.
  auto G = CG::WJG();
  auto D = CG::DBLU(G);                                   // D.z shares G.z
  const auto Dz = D.z().wbn().host();
  const auto T = CG::ZADDU(G, D);                         // rewrites G (new Z) -- D's Z must survive
  EXPECT_TRUE(D.z().wbn().host() == Dz);
  EXPECT_TRUE(affine_is(D, G2x, G2y) && affine_is(T, G3x, G3y));
  auto G2 = CG::WJG();
  auto D2 = CG::DBLU(G2);
  const auto D2z = D2.z().wbn().host();
  const auto F = CG::ZDAU(D2, G2);                        // rewrites G2 (shares its old Z with D2): D2 is const
  EXPECT_TRUE(D2.z().wbn().host() == D2z && affine_is(D2, G2x, G2y) && affine_is(F, G5x, G5y));
  EXPECT_TRUE(all(F.z().wbn() == G2.z().wbn()));
}
TEST(CurveGroup, Swap) {                                                                      // :96-115
  auto G = CG::WJG();
  auto D = CG::DBLU(G);
  const auto Z = hip::mask::filled(4, false);
  auto a = G, b = D;
  swap_if(Z, a, b);
  EXPECT_TRUE(all(a == G)); EXPECT_TRUE(all(b == D));
  swap_if(!Z, a, b);
  EXPECT_TRUE(all(a == D)); EXPECT_TRUE(all(b == G));
}
TEST(CurveGroup, IfElse) {                                                                    // ifelse.h:15-49 (what scalar_mult's even-k correction uses, curve_group.h:217)
  auto G = CG::WJG();
  auto D = CG::DBLU(G);
  const hip::mask m{true, false, false, true};
  const auto S = if_else(m, G, D);                       // lanes 0, 3 from G; lanes 1, 2 from D
  const auto Gx = G.x().wbn().host(), Dx = D.x().wbn().host(), Sx = S.x().wbn().host();
  EXPECT_TRUE(Sx[0] == Gx[0] && Sx[1] == Dx[1] && Sx[2] == Dx[2] && Sx[3] == Gx[3]);
  EXPECT_TRUE(all(if_else(hip::mask::filled(4, true), G, D) == G) && all(if_else(hip::mask::filled(4, false), G, D) == D));
  const auto a = lanes<W256>("0000000000000000000000000000000000000000000000000000000000000001"_hex, "0000000000000000000000000000000000000000000000000000000000000002"_hex,
                             "0000000000000000000000000000000000000000000000000000000000000003"_hex, "0000000000000000000000000000000000000000000000000000000000000004"_hex);
  const auto b = splat<W256>("00000000000000000000000000000000000000000000000000000000000000ff"_hex);
  EXPECT_TRUE(all(if_else(m, a, b) == lanes<W256>("0000000000000000000000000000000000000000000000000000000000000001"_hex, "00000000000000000000000000000000000000000000000000000000000000ff"_hex,
                                                  "00000000000000000000000000000000000000000000000000000000000000ff"_hex, "0000000000000000000000000000000000000000000000000000000000000004"_hex)));
  bool threw = false;
  try { (void)if_else(hip::mask::filled(3, true), a, b); } catch (hip::error const&) { threw = true; }
  EXPECT_TRUE(threw);
}
TEST(CurveGroup, ScalarMult) {                                                                // :117-173
  const auto G = CG::WJG();
  struct { std::array<uint8_t, 32> k, x, y; } cases[] = {
    {"0000000000000000000000000000000000000000000000000000000000000005"_hex, G5x, G5y},
    {"0bc1b1f28709decb543d9677d2cc9942348f6b984deff409430740942ff38827"_hex, "1b7721565b2c4a9f203bbccc6b531df2789fde0d135c76db71e4a7bbab9e85b2"_hex, "393655bcc30f67f3a4e257b39685657d7c8df7b2a132b49c848003e300c8dcd1"_hex},
    {"0a891cecc2bf13b0aca744434a9c9f4bd7bf5c8ed86e2f76e7df72bad813bd80"_hex, "f411d79e2997b2954975046d23b0e4a69ce580a4a81e1bed18fef6fd9ea4a912"_hex, "43895f527937e816c3d7c0a2370002796d3cd4860cb034df86cbe7da227d9113"_hex}};
  for (auto const& c : cases) {
    const auto xs = bn_from_bytes_BE<bignum_256>(c.k);
    EXPECT_TRUE(affine_is(CG::scalar_mult(W256{xs}, G), c.x, c.y));
    EXPECT_TRUE(affine_is(CG::scalar_mult_1s(xs, G), c.x, c.y));
    EXPECT_TRUE(affine_is(scalar_mult_p256(W256{xs}, G), c.x, c.y));                          // lib/scalar_mult_p256.cpp:10-12
    EXPECT_TRUE(all(CG::scalar_mult(W256{xs}, G) == CG::scalar_mult_1s(xs, G)));             // Jacobian level too
  }
}
// Beyond the reference's tests: runtime-length batches and the second curve through the same API.
TEST(Batch, RuntimeLengthAndSecp256k1) {
  using K = curve_secp256k1; using KG = curve_group<K>;
  const size_t n = 1000;
  W256 k(n, [](size_t i, size_t) { bignum_256 b; b.limbs = {0x9e3779b97f4a7c15ull * (i + 1), i, ~i, 0x0123456789abcdefull ^ (i << 20)}; return b; });
  const auto J = KG::scalar_mult(k, KG::WJG(n));
  const auto J1 = KG::scalar_mult(W256(n, k.get(17)), KG::WJG(n));
  EXPECT_TRUE(J.size() == n);
  EXPECT_TRUE(J.x().wbn().get(17) == J1.x().wbn().get(0) && J.z().wbn().get(17) == J1.z().wbn().get(999));
  const auto five = KG::scalar_mult(W256(n, bignum_256::from(5)), KG::WJG(n)).to_affine();   // 5G on secp256k1
  EXPECT_TRUE(all(five.x() == W256(n, bn_from_bytes_BE<bignum_256>("2f8bde4d1a07209355b4a7250a5c5128e88b84bddc619ab7cba8d569b240efe4"_hex))));
  EXPECT_TRUE(all(five.y() == W256(n, bn_from_bytes_BE<bignum_256>("d8ac222636e5e3d6d4dba9dda6c9c426f788271bab0d6840dca87d3aa6ac62d6"_hex))));
}

// Round 5: curve_group<Curve> for a curve that is NOT one of the engine's two (the reference's template takes any type with bn_type, P, A, B, Gx, Gy:
// curve.h:12-15).  brainpoolP256r1 (RFC 5639 3.4) described with the same members; the expected points are the reference's own answers for this curve
// (tests/golden/ref_curves_vectors.json, minted from the reference instantiated with these parameters).  First use registers the curve with the engine.
struct curve_brainpoolp256r1 {
  using bn_type = bignum_256;
  using P  = bn256_constant<0xa9fb57dba1eea9bcull, 0x3e660a909d838d72ull, 0x6e3bf623d5262028ull, 0x2013481d1f6e5377ull>;
  using A  = bn256_constant<0x7d5a0975fc2c3057ull, 0xeef67530417affe7ull, 0xfb8055c126dc5c6cull, 0xe94a4b44f330b5d9ull>;
  using B  = bn256_constant<0x26dc5c6ce94a4b44ull, 0xf330b5d9bbd77cbfull, 0x958416295cf7e1ceull, 0x6bccdc18ff8c07b6ull>;
  using Gx = bn256_constant<0x8bd2aeb9cb7e57cbull, 0x2c4b482ffc81b7afull, 0xb9de27e1e3bd23c2ull, 0x3a4453bd9ace3262ull>;
  using Gy = bn256_constant<0x547ef835c3dac4fdull, 0x97f8461a14611dc9ull, 0xc27745132ded8e54ull, 0x5c1d54c72f046997ull>;
};
TEST(Curves, AnyCurveThroughCurveGroup) {
  using K = curve_brainpoolp256r1; using KG = curve_group<K>; using KJ = wide_jacobian_curve_point<K>;
  EXPECT_TRUE(KG::curve_id() >= ECSIMD_HIP_FIRST_REGISTERED_CURVE);
  EXPECT_TRUE(curve_group<curve_nist_p256>::curve_id() == ECSIMD_HIP_P256 && curve_group<curve_secp256k1>::curve_id() == ECSIMD_HIP_SECP256K1);
  EXPECT_TRUE(KG::Am() == bn_from_bytes_BE<bignum_256>("1e4676abd666bc1795ec1e5e6398556ea68123f1c1d20c64d5d18edf69696261"_hex));     // to_mgry(A), curve_group.h:32
  EXPECT_TRUE(KG::Bm() == bn_from_bytes_BE<bignum_256>("1634f57646a3c93e64ca989357f2e9d90ac34a49cc51bf5905d24d72c0c0f36f"_hex));
  const auto G = KG::WJG();
  struct { std::array<uint8_t, 32> k, x, y; } cases[] = {
    {"0000000000000000000000000000000000000000000000000000000000000005"_hex, "855433a3a4c8e334a5f863e8b69fc1477cf41589c0d8c3fb32f95f7c85fe101d"_hex, "a50c95efc2ad06c4d7e172e40350d911097082129591c88bef9e224a5fd8814c"_hex},
    {"0bc1b1f28709decb543d9677d2cc9942348f6b984deff409430740942ff38827"_hex, "9531554560f0e4bb5bac426b8e8001bf95592d3b79d265bda2b1de28a474579a"_hex, "35b9ac2a8a75dfdce0eb7aa0127d8b244cf315c89f0c2c04409d2caf717f9798"_hex},
    {"0a891cecc2bf13b0aca744434a9c9f4bd7bf5c8ed86e2f76e7df72bad813bd80"_hex, "9bcd00f871cc765404f41bcce735aa940d55eb38974f2227e6244a3552d17020"_hex, "291c97f4d254ca5b2c19803a4d5ea365ee5aa6698436278eb1a6db2be8f81ac0"_hex}};
  for (auto const& c : cases) {
    const auto xs = bn_from_bytes_BE<bignum_256>(c.k);
    const auto A1 = KG::scalar_mult(W256{xs}, G).to_affine(), A2 = KG::scalar_mult_1s(xs, G).to_affine();
    EXPECT_TRUE(all(A1.x() == W256{bn_from_bytes_BE<bignum_256>(c.x)}) && all(A1.y() == W256{bn_from_bytes_BE<bignum_256>(c.y)}));
    EXPECT_TRUE(all(A2.x() == A1.x()) && all(A2.y() == A1.y()));
  }
  // the co-Z formulas compose as on any curve: 2G (DBLU) + G (ZADDU) = 3G = TRPLU; 2 * 3G + G (ZDAU) + G (ADD_Z2_1) = 8G = scalar_mult(8)
  auto P = G; const auto D = KG::DBLU(P); const auto T = KG::ZADDU(P, D);
  auto P2 = G; const auto T2 = KG::TRPLU(P2);
  EXPECT_TRUE(all(T == T2) && all(P == P2));
  auto Q = P2; const auto S = KG::ZDAU(T2, Q);                                                   // 2 * 3G + G = 7G
  const auto E = KG::ADD_Z2_1(S, G).to_affine(), E8 = KG::scalar_mult(W256{bignum_256::from(8)}, G).to_affine();
  EXPECT_TRUE(all(E.x() == E8.x()) && all(E.y() == E8.y()));
  const auto y = KG::compute_y(E8.x());                                                          // y^2 = x^3 + a x + b with THIS curve's a
  EXPECT_TRUE(y.has_value() && (all(*y == E8.y()) || all(*y == (KJ::from_affine(E8).opposite().to_affine().y()))));
  const size_t n = 333;                                                                          // a runtime-length batch on the registered curve
  W256 k(n, [](size_t i, size_t) { bignum_256 b; b.limbs = {0x9e3779b97f4a7c15ull * (i + 1), i, ~i, 0x0123456789abcdefull ^ (i << 20)}; return b; });
  const auto J = KG::scalar_mult(k, KG::WJG(n)), J1 = KG::scalar_mult(W256(n, k.get(17)), KG::WJG(n));
  EXPECT_TRUE(J.x().wbn().get(17) == J1.x().wbn().get(0) && J.z().wbn().get(17) == J1.z().wbn().get(n - 1));
}

// The first application on such a curve: a Curve type that also names its order n gets u1 G + u2 Q, ECDSA and the SEC1 codecs on top of the reference's
// ladder, and the table-driven algorithms a registered curve has (its generator's comb, the variable-base window loop).  Expected values: textbook affine arithmetic on Python integers (tests/test_gpu_curves.py's model).
struct curve_brainpoolp256r1_n : curve_brainpoolp256r1 {
  using N = bn256_constant<0xa9fb57dba1eea9bcull, 0x3e660a909d838d71ull, 0x8c397aa3b561a6f7ull, 0x901e0e82974856a7ull>;
};
struct curve_sm2_no_order {
  using bn_type = bignum_256;
  using P  = bn256_constant<0xfffffffeffffffffull, 0xffffffffffffffffull, 0xffffffff00000000ull, 0xffffffffffffffffull>;
  using A  = bn256_constant<0xfffffffeffffffffull, 0xffffffffffffffffull, 0xffffffff00000000ull, 0xfffffffffffffffcull>;
  using B  = bn256_constant<0x28e9fa9e9d9f5e34ull, 0x4d5a9e4bcf6509a7ull, 0xf39789f515ab8f92ull, 0xddbcbd414d940e93ull>;
  using Gx = bn256_constant<0x32c4ae2c1f198119ull, 0x5f9904466a39c994ull, 0x8fe30bbff2660be1ull, 0x715a4589334c74c7ull>;
  using Gy = bn256_constant<0xbc3736a2f4f6779cull, 0x59bdcee36b692153ull, 0xd0a9877cc62a4740ull, 0x02df32e52139f0a0ull>;
};
TEST(Curves, EcdsaAndSec1OnARegisteredCurve) {
  using K = curve_brainpoolp256r1_n; using KG = curve_group<K>; using WCP = wide_curve_point<K>;
  EXPECT_TRUE(KG::curve_id() != curve_group<curve_brainpoolp256r1>::curve_id());                 // the order is part of the registration's key: the id without it keeps the reference's layers only
  const auto e = "AF2BDBE1AA9B6EC1E2ADE1D694F41FC71A831D0268E9891562113D8A62ADD1BF"_hex, d = "0bc1b1f28709decb543d9677d2cc9942348f6b984deff409430740942ff38827"_hex;
  const auto k1 = "0a891cecc2bf13b0aca744434a9c9f4bd7bf5c8ed86e2f76e7df72bad813bd80"_hex, k2 = "a9fb57dba1eea9bc3e660a909d838d718c397aa3b561a6f7901e0e82974856a6"_hex;   // k2 = n - 1: the ladder alone is wrong there
  const auto zero = "0000000000000000000000000000000000000000000000000000000000000000"_hex;
  hip::mask signed_ok;
  const auto sig = KG::ecdsa_sign(splat<W256>(e), splat<W256>(d), lanes<W256>(k1, k2, zero, k1), signed_ok);
  const auto so = signed_ok.host();
  EXPECT_TRUE(so[0] == 1 && so[1] == 1 && so[2] == 0 && so[3] == 1);
  EXPECT_TRUE(sig.first.get(0) == bn_from_bytes_BE<bignum_256>("9bcd00f871cc765404f41bcce735aa940d55eb38974f2227e6244a3552d17020"_hex));
  EXPECT_TRUE(sig.second.get(0) == bn_from_bytes_BE<bignum_256>("1e00473ff879c5454c3e134bb3efb639303ee530dc5a62d255c95c8a8c1fec63"_hex));
  EXPECT_TRUE(sig.first.get(1) == bn_from_bytes_BE<bignum_256>("8bd2aeb9cb7e57cb2c4b482ffc81b7afb9de27e1e3bd23c23a4453bd9ace3262"_hex));   // x(-G) = x(G)
  EXPECT_TRUE(sig.second.get(1) == bn_from_bytes_BE<bignum_256>("620eb7b2e74ab24a7e468eb910295d5d45c8bcfe966b106b265cf33a2c88b80a"_hex));
  const WCP Q{splat<W256>("9531554560f0e4bb5bac426b8e8001bf95592d3b79d265bda2b1de28a474579a"_hex), splat<W256>("35b9ac2a8a75dfdce0eb7aa0127d8b244cf315c89f0c2c04409d2caf717f9798"_hex)};
  const auto ok = KG::ecdsa_verify(splat<W256>(e), sig.first, sig.second, Q).host();
  EXPECT_TRUE(ok[0] == 1 && ok[1] == 1 && ok[2] == 0 && ok[3] == 1);
  EXPECT_TRUE(KG::ecdsa_verify(splat<W256>(d), sig.first, sig.second, Q).count() == 0);            // another message
  hip::mask fin;
  const auto sum = KG::double_scalar_mult(lanes<W256>(k2, zero, k1, k1), lanes<W256>(zero, k2, zero, zero), Q, fin);   // (n - 1) G = -G; (n - 1) Q = -Q
  const auto fh = fin.host();
  EXPECT_TRUE(fh[0] == 1 && fh[1] == 1 && sum.x().get(0) == K::Gx::value && sum.x().get(1) == Q.x().get(1) && !(sum.y().get(0) == K::Gy::value));
  const W256 ks(300, [](size_t i, size_t) { bignum_256 b; b.limbs = {0x9e3779b97f4a7c15ull * (i + 1), i * 77, ~i, 0x0123456789abcdefull ^ (i << 20)}; return b; });
  const auto ladder = KG::scalar_mult(ks, KG::WJG(300)).to_affine();                              // the generator's comb (plain and constant-time) = the ladder's points
  EXPECT_TRUE(all(KG::scalar_mult_base_affine(ks) == ladder) && all(KG::scalar_mult_base_affine_secret(ks) == ladder));
  // a variable base: the lane's own window table (k_gvarwin.hip) = the ladder + to_affine, lane for lane; the curve without its order keeps the ladder
  const W256 ks2(300, [](size_t i, size_t) { bignum_256 b; b.limbs = {0xd1342543de82ef95ull * (i + 3), i * 131, ~(i << 7), 0xfedcba9876543210ull ^ (i << 33)}; return b; });
  EXPECT_TRUE(all(KG::scalar_mult_affine(ks2, ladder) == KG::scalar_mult_affine(ks2, ladder, false)));
  EXPECT_TRUE(all(KG::scalar_mult_affine_secret(ks2, ladder) == KG::scalar_mult_affine(ks2, ladder, false)));   // ECDH on such a curve: every table entry read in every window
  // the same curve WITHOUT its order has no tables: scalar_mult_affine serves the same points from the ladder
  using K0 = curve_brainpoolp256r1; using KG0 = curve_group<K0>;
  EXPECT_TRUE(KG::can(ECSIMD_HIP_CURVE_WINDOW_VARIABLE_BASE) && KG::can(ECSIMD_HIP_CURVE_ECDSA) && !KG0::can(ECSIMD_HIP_CURVE_WINDOW_VARIABLE_BASE) && !KG0::can(ECSIMD_HIP_CURVE_HAS_ORDER));
  const auto plain = KG0::scalar_mult_affine(ks2, wide_curve_point<K0>{ladder.x(), ladder.y()});
  const auto want = KG::scalar_mult_affine(ks2, ladder, false);
  EXPECT_TRUE(all(plain.x() == want.x()) && all(plain.y() == want.y()));
  hip::mask dec_ok;
  const auto back = sec1_decode<K>(sec1_encode<K>(Q, true), true, dec_ok);                        // compressed: the square root with THIS curve's a and b
  EXPECT_TRUE(dec_ok.count() == 4 && all(back == Q));
  // the curve WITHOUT its order has the reference's layers only
  bool refused = false;
  try { (void)curve_group<curve_sm2_no_order>::ecdsa_verify(splat<W256>(e), sig.first, sig.second, wide_curve_point<curve_sm2_no_order>{splat<W256>(e), splat<W256>(e)}); }
  catch (std::exception const&) { refused = true; }
  EXPECT_TRUE(refused);
}

// The curve structs' constants equal the engine's own table (ecsimd_hip_get_constant: 0 p, 1 a, 2 b, 3 Gx, 4 Gy) and
// the hexadecimal literals of the standards documents.
template <class K> static void constants_match_engine() {
  using M = mgry_constants<typename K::P>;
  EXPECT_TRUE(K::P::value == M::get(0)); EXPECT_TRUE(K::A::value == M::get(1)); EXPECT_TRUE(K::B::value == M::get(2));
  EXPECT_TRUE(K::Gx::value == M::get(3)); EXPECT_TRUE(K::Gy::value == M::get(4));
}
TEST(Curves, ConstantsMatchTheEngine) {
  constants_match_engine<curve_nist_p256>();
  constants_match_engine<curve_secp256k1>();
  EXPECT_TRUE(curve_nist_p256::P::value == bn_from_bytes_BE<bignum_256>("ffffffff00000001000000000000000000000000ffffffffffffffffffffffff"_hex));
  EXPECT_TRUE(curve_nist_p256::B::value == bn_from_bytes_BE<bignum_256>("5ac635d8aa3a93e7b3ebbd55769886bc651d06b0cc53b0f63bce3c3e27d2604b"_hex));
  EXPECT_TRUE(curve_nist_p256::Gx::value == bn_from_bytes_BE<bignum_256>("6b17d1f2e12c4247f8bce6e563a440f277037d812deb33a0f4a13945d898c296"_hex));
  EXPECT_TRUE(curve_nist_p256::Gy::value == bn_from_bytes_BE<bignum_256>("4fe342e2fe1a7f9b8ee7eb4a7c0f9e162bce33576b315ececbb6406837bf51f5"_hex));
  EXPECT_TRUE(curve_secp256k1::Gx::value == bn_from_bytes_BE<bignum_256>("79be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798"_hex));
  EXPECT_TRUE(curve_secp256k1::Gy::value == bn_from_bytes_BE<bignum_256>("483ada7726a3c4655da4fbfc0e1108a8fd17b448a68554199c47d08ffb10d4b8"_hex));
}

// Extensions: the windowed algorithms give the ladder's affine points (both curves), and u1*G + u2*Q composes them.
template <class K> static void windowed_paths_agree() {
  using KG = curve_group<K>;
  const size_t n = 777;
  W256 k(n, [](size_t i, size_t) { bignum_256 b; b.limbs = {0x9e3779b97f4a7c15ull * (i + 3), ~i * 0x100000001b3ull, i ^ 0x8888888888888888ull, 0xfedcba9876543210ull - (i << 33)}; return b; });
  W256 s(n, [](size_t i, size_t) { bignum_256 b; b.limbs = {i + 1, i * i, 0x0123456789abcdefull * (i + 7), 0x7000000000000000ull ^ (i << 9)}; return b; });
  const auto P = KG::scalar_mult(s, KG::WJG(n)).to_affine();                                   // lane-distinct points (ladder)
  EXPECT_TRUE(all(KG::scalar_mult_base_affine(s) == P));
  EXPECT_TRUE(all(KG::scalar_mult_base_affine_secret(s) == P));                               // the constant-time comb: same points
  const auto ladder = KG::scalar_mult_affine(k, P, false);
  EXPECT_TRUE(all(ladder == KG::scalar_mult(k, wide_jacobian_curve_point<K>::from_affine(P)).to_affine()));
  EXPECT_TRUE(all(KG::scalar_mult_affine(k, P) == ladder));
  EXPECT_TRUE(all(KG::scalar_mult_affine_secret(k, P) == ladder));                            // the constant-time window loop: same points
  const auto JP = wide_jacobian_curve_point<K>::from_affine(P);
  const auto J = KG::scalar_mult(k, JP);                                                       // k*P, Z != 1
  EXPECT_TRUE(all(KG::add_mixed_complete(J, JP).to_affine() == KG::ADD_Z2_1(J, JP).to_affine()));
  auto twoP = JP; const auto D = KG::DBLU(twoP);                                               // D = 2P
  EXPECT_TRUE(all(KG::add_mixed_complete(JP, JP).to_affine() == D.to_affine()));              // the tangent case ADD_Z2_1 cannot do
  hip::mask fin;
  const auto sum = KG::double_scalar_mult(W256(n, bignum_256::from(0)), k, P, fin);             // 0*G + k*P
  EXPECT_TRUE(all(fin)); EXPECT_TRUE(all(sum == ladder));
  const auto sum2 = KG::double_scalar_mult(s, W256(n, bignum_256::from(1)), P, fin);            // s*G + P = 2P
  EXPECT_TRUE(all(fin)); EXPECT_TRUE(all(sum2 == KG::scalar_mult_affine(W256(n, bignum_256::from(2)), P)));
}
TEST(Batch, WindowedPathsAgreeWithTheLadder) {
  windowed_paths_agree<curve_nist_p256>();
  windowed_paths_agree<curve_secp256k1>();
}

// utility.h:45-51 wide_mask_bit and the device wire formats (beyond the reference's tests)
TEST(Batch, OperandsMustAgreeOnTheLength) {
  using CG = curve_group<curve_nist_p256>;
  bool threw = false;
  try { (void)CG::scalar_mult(W256(8, bignum_256::from(3)), CG::WJG()); } catch (hip::error const&) { threw = true; }      // 8 scalars, 4 points
  EXPECT_TRUE(threw);
  threw = false;
  try { auto P = CG::WJG(4); auto O = CG::WJG(6); (void)CG::ZADDU(P, O); } catch (hip::error const&) { threw = true; }
  EXPECT_TRUE(threw);
}
TEST(Batch, DeviceGroup) {                                                                    // shards over a device group, one gather
  using CG = curve_group<curve_nist_p256>;
  using BN = bignum_256;
  EXPECT_TRUE((hip::device_group::shard_range(size_t(1) << 24, 5, 8) == std::pair<size_t, size_t>{size_t(5) << 21, size_t(1) << 21}));
  const size_t n = 37;
  std::vector<BN> k(n), s(n);
  for (size_t i = 0; i < n; ++i) { k[i] = BN{{0x9e3779b97f4a7c15ull * (i + 1), 0xbf58476d1ce4e5b9ull ^ i, i * i + 7, 0x0123456789abcdefull + i}}; s[i] = BN::from(i + 2); }
  const auto P = CG::scalar_mult(wide_bignum<BN>(s), CG::WJG(n)).to_affine();                  // P_i = (i + 2) * G
  const auto J = CG::scalar_mult(wide_bignum<BN>(k), wide_jacobian_curve_point<curve_nist_p256>::from_affine(P));
  const auto A = J.to_affine();
  const auto hx = P.x().host(), hy = P.y().host();
  for (auto const& devices : {std::vector<int>{0}, std::vector<int>{0, 0}, std::vector<int>{0, 0, 0, 0, 0}}) {
    hip::device_group g(devices);
    EXPECT_TRUE(g.size() == (int)devices.size() && !g.uses_rccl());                             // RCCL needs distinct devices: a one-GPU box has none to offer
    const auto r = CG::scalar_mult(g, k, hx, hy);
    EXPECT_TRUE(r.x == J.x().wbn().host() && r.y == J.y().wbn().host() && r.z == J.z().wbn().host());
    const auto a = CG::scalar_mult(g, k, hx, hy, true);
    EXPECT_TRUE(a.x == A.x().host() && a.y == A.y().host() && a.z.empty());
  }
}
TEST(Batch, MaskBitAndWireFormats) {
  using K = curve_nist_p256; using KG = curve_group<K>;
  const auto a = lanes<W128>("00000000000000000000000000000001"_hex, "00000000000000008000000000000000"_hex, "00000000000000010000000000000000"_hex, "80000000000000000000000000000000"_hex);
  EXPECT_TRUE(all(wide_mask_bit(a, 0, 0) == cmp_res_t<W128>{true, false, false, false}));
  EXPECT_TRUE(all(wide_mask_bit(a, 0, 63) == cmp_res_t<W128>{false, true, false, false}));
  EXPECT_TRUE(all(wide_mask_bit(a, 1, 0) == cmp_res_t<W128>{false, false, true, false}));
  EXPECT_TRUE(all(wide_mask_bit(a, 1, 63) == cmp_res_t<W128>{false, false, false, true}));
  const size_t n = 300;
  W256 k(n, [](size_t i, size_t) { bignum_256 b; b.limbs = {i * 0x9e3779b97f4a7c15ull + 1, ~i, i << 40, 0x7fffffff00000000ull ^ i}; return b; });
  const auto be = wide_to_bytes_BE(k);
  EXPECT_TRUE(be.size() == n * 32);
  EXPECT_TRUE(bn_from_bytes_BE<bignum_256>(&be[32 * 123]) == k.get(123));                       // agrees with the host-side single-value codec
  EXPECT_TRUE(all(wide_from_bytes_BE(be) == k));
  const auto pts = KG::scalar_mult(k, KG::WJG(n)).to_affine();
  for (bool compressed : {false, true}) {
    auto wire = sec1_encode(pts, compressed);
    hip::mask ok;
    const auto back = sec1_decode<K>(wire, compressed, ok);
    EXPECT_TRUE(all(ok)); EXPECT_TRUE(all(back == pts));
    wire[(compressed ? 33 : 65) * 7 + 5] ^= 0x40;                                               // corrupt one X
    const auto bad = sec1_decode<K>(wire, compressed, ok);
    (void)bad;
    size_t nbad = 0; for (auto b : ok.host()) nbad += !b;
    EXPECT_TRUE(compressed ? nbad <= 1 : nbad == 1);        // uncompressed: certainly off the curve; compressed: half of all x decompress
  }
}

int main() { return mini::run_all(); }
