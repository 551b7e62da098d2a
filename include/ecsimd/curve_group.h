// ecsimd/curve_group.h -- curve_group<Curve>: the co-Z formulas and the ladder (reference curve_group.h:20-252).
// Same static member names and parameter conventions: reference parameters that the reference
// updates in place (DBLU's P, ZADDU's P, ZDAU's Q, TRPLU's P) are updated in place here too.
#ifndef ECSIMD_CURVE_GROUP_H
#define ECSIMD_CURVE_GROUP_H
#include <ecsimd/device_group.h>
#include <ecsimd/jacobian_curve_point.h>
#include <optional>

namespace ecsimd {
template <class Curve>
struct curve_group {
  using WBN = curve_wide_bn_t<Curve>;
  using BN = typename WBN::value_type;
  using WMBN = wide_mgry_bignum<WBN, typename Curve::P>;
  using gfp = GFp<WBN, typename Curve::P>;
  using WCP = wide_curve_point<Curve>;
  using WJCP = wide_jacobian_curve_point<Curve>;
  static int curve_id() { return WJCP::curve_id(); }
  // what this curve's id can do beyond the reference's layers (ecsimd_hip_curve_capabilities): decided once, at registration
  static bool can(int capability) { static const int caps = [] { int c = 0; hip::check(ecsimd_hip_curve_capabilities(curve_id(), &c), "ecsimd_hip_curve_capabilities"); return c; }(); return (caps & capability) != 0; }

  static BN Am() { return curve_constant(8); }      // curve_group.h:32
  static BN Bm() { return curve_constant(9); }      // curve_group.h:31
  static BN curve_constant(int which) { BN r; hip::check(ecsimd_hip_get_constant(curve_id(), which, r.limbs.data()), "ecsimd_hip_get_constant"); return r; }
  static WCP WG(size_t lanes = default_lanes) { return WCP{WBN(lanes, Curve::Gx::value), WBN(lanes, Curve::Gy::value)}; }   // :35-37
  static WJCP WJG(size_t lanes = default_lanes) { return WJCP::from_affine(WG(lanes)); }                                     // :39-41

  static std::optional<WBN> compute_y(WBN const& x) {                        // :52-58 (all lanes or nothing, like the reference)
    hip::mask ok; WBN y = compute_y_lanes(x, ok); if (!all(ok)) return {}; return {y};
  }
  static WBN compute_y_lanes(WBN const& x, hip::mask& ok) {                  // per-lane validity; y^2 = x^3 + a x + b for either curve
    auto y = WBN::uninitialized(x.size()); ok = hip::mask(x.size());
    hip::check(ecsimd_hip_compute_y(hip::context(), curve_id(), x.data(), y.data(), ok.data(), x.size()), "ecsimd_hip_compute_y"); return y;
  }

  static WJCP DBLU(WJCP& P) {                                                // :64-87
    P.unshare(); WJCP r = fresh_xy(P.size());                               // co-Z: r and the rewritten P share ONE Z array, written once
    hip::check(ecsimd_hip_dblu(hip::context(), curve_id(), px(P), py(P), pz(P), px(r), py(r), pz(P), P.size()), "ecsimd_hip_dblu"); r.z() = P.z(); return r;
  }
  static WJCP ZADDU(WJCP& P, WJCP const& O) {                                // :91-116
    same_length(P.size(), O.size(), "ZADDU");
    P.unshare(); WJCP r = fresh_xy(P.size());
    hip::check(ecsimd_hip_zaddu(hip::context(), curve_id(), px(P), py(P), pz(P), px(O), py(O), pz(O), px(r), py(r), pz(P), P.size()), "ecsimd_hip_zaddu"); r.z() = P.z(); return r;
  }
  static WJCP ZDAU(WJCP const& P, WJCP& Q) {                                 // :120-153
    same_length(P.size(), Q.size(), "ZDAU");
    Q.unshare(); WJCP r = fresh_xy(P.size());
    hip::check(ecsimd_hip_zdau(hip::context(), curve_id(), px(P), py(P), pz(P), px(Q), py(Q), pz(Q), px(r), py(r), pz(Q), P.size()), "ecsimd_hip_zdau"); r.z() = Q.z(); return r;
  }
  static WJCP ADD_Z2_1(WJCP const& A, WJCP const& B) {                       // :155-179 (B.z must be mgry(1))
    same_length(A.size(), B.size(), "ADD_Z2_1");
    WJCP r = fresh(A.size());
    hip::check(ecsimd_hip_add_z2_1(hip::context(), curve_id(), px(A), py(A), pz(A), px(B), py(B), px(r), py(r), pz(r), A.size()), "ecsimd_hip_add_z2_1"); return r;
  }
  static WJCP TRPLU(WJCP& P) {                                               // :183-186
    P.unshare(); WJCP r = fresh_xy(P.size());
    hip::check(ecsimd_hip_trplu(hip::context(), curve_id(), px(P), py(P), pz(P), px(r), py(r), pz(P), P.size()), "ecsimd_hip_trplu"); r.z() = P.z(); return r;
  }
  // k[i] * P[i], P.z must be mgry(1) (:189-218).  One kernel: the whole ladder stays in registers.
  static WJCP scalar_mult(WBN const& x, WJCP P) {
    same_length(x.size(), P.size(), "scalar_mult");
    WJCP r = fresh(P.size());
    hip::check(ecsimd_hip_scalar_mult(hip::context(), curve_id(), x.data(), px(P), py(P), px(r), py(r), pz(r), P.size(), ECSIMD_HIP_BASE_MGRY | ECSIMD_HIP_OUT_JACOBIAN), "ecsimd_hip_scalar_mult"); return r;
  }
  // one scalar for every lane (:221-251)
  static WJCP scalar_mult_1s(BN const& x, WJCP P) {
    WJCP r = fresh(P.size());
    hip::check(ecsimd_hip_scalar_mult_1s(hip::context(), curve_id(), x.limbs.data(), px(P), py(P), px(r), py(r), pz(r), P.size(), ECSIMD_HIP_BASE_MGRY | ECSIMD_HIP_OUT_JACOBIAN), "ecsimd_hip_scalar_mult_1s"); return r;
  }

  // ---- extensions (not in the reference): affine-level entry points over the faster algorithms of the C ABI.
  // Same points as to_affine() of the ladder's result for every scalar where the ladder is non-degenerate.
  // k[i] * P[i], P affine classical -> affine classical.  windowed: per-element tables of 8 multiples of P + signed 4-bit
  // windows (ECSIMD_HIP_ALG_WINDOWED); otherwise the reference ladder followed by one simultaneous inversion.  A curve registered at run time has the
  // tables when it names a prime order N >= 2^255 (ECSIMD_HIP_CURVE_WINDOW_VARIABLE_BASE); without it the same points come from the ladder.
  static WCP scalar_mult_affine(WBN const& x, WCP const& P, bool windowed = true) {
    same_length(x.size(), P.size(), "scalar_mult_affine");
    WCP r{WBN::uninitialized(P.size()), WBN::uninitialized(P.size())};
    const bool tables = windowed && can(ECSIMD_HIP_CURVE_WINDOW_VARIABLE_BASE);
    hip::check(ecsimd_hip_scalar_mult(hip::context(), curve_id(), x.data(), P.x().data(), P.y().data(), r.x().data(), r.y().data(), nullptr, P.size(),
                                      ECSIMD_HIP_BASE_CLASSICAL | ECSIMD_HIP_OUT_AFFINE | (tables ? ECSIMD_HIP_ALG_WINDOWED : 0)), "ecsimd_hip_scalar_mult");
    return r;
  }
  // A + B for every input, unlike ADD_Z2_1: A = B, A = -B (Z = 0 comes back), A at infinity (Z = 0); B.z must be mgry(1).
  static WJCP add_mixed_complete(WJCP const& A, WJCP const& B) {
    same_length(A.size(), B.size(), "add_mixed_complete");
    WJCP r = fresh(A.size());
    hip::check(ecsimd_hip_add_mixed_complete(hip::context(), curve_id(), px(A), py(A), pz(A), px(B), py(B), px(r), py(r), pz(r), A.size()), "ecsimd_hip_add_mixed_complete"); return r;
  }
  // k[i] * G through the 20-bit window table of odd multiples in device memory (12 mixed additions), affine classical.  A curve registered at run time
  // (one that names its order N) has the signed 7-bit table of its generator in LDS instead (36 mixed additions).
  static WCP scalar_mult_base_affine(WBN const& x) {
    WCP r{WBN::uninitialized(x.size()), WBN::uninitialized(x.size())};
    const int alg = curve_id() >= ECSIMD_HIP_FIRST_REGISTERED_CURVE ? ECSIMD_HIP_ALG_WINDOWED_SIGNED : ECSIMD_HIP_ALG_WINDOWED_BIG;
    hip::check(ecsimd_hip_scalar_mult_base(hip::context(), curve_id(), x.data(), r.x().data(), r.y().data(), nullptr, x.size(),
                                           ECSIMD_HIP_OUT_AFFINE | alg), "ecsimd_hip_scalar_mult_base");
    return r;
  }
  // k[i] * P[i] for SECRET scalars (ECDH): the per-element window tables with ECSIMD_HIP_ALG_CONSTANT_TIME -- every entry of the lane's
  // table read in every window, kept under lane masks; 1.45 x (P-256) / 1.74 x (secp256k1) the ladder.  Affine classical in and out.
  static WCP scalar_mult_affine_secret(WBN const& x, WCP const& P) {
    same_length(x.size(), P.size(), "scalar_mult_affine_secret");
    WCP r{WBN::uninitialized(P.size()), WBN::uninitialized(P.size())};
    const int alg = can(ECSIMD_HIP_CURVE_WINDOW_VARIABLE_BASE) ? (ECSIMD_HIP_ALG_WINDOWED | ECSIMD_HIP_ALG_CONSTANT_TIME) : 0;      // (without the tables: the ladder, constant-time as it is)
    hip::check(ecsimd_hip_scalar_mult(hip::context(), curve_id(), x.data(), P.x().data(), P.y().data(), r.x().data(), r.y().data(), nullptr, P.size(),
                                      ECSIMD_HIP_BASE_CLASSICAL | ECSIMD_HIP_OUT_AFFINE | alg), "ecsimd_hip_scalar_mult");
    return r;
  }
  // k[i] * G for SECRET scalars (key generation, ECDSA nonces): an LDS comb with ECSIMD_HIP_ALG_CONSTANT_TIME -- every table entry of a
  // window read, the wanted one kept under lane masks, no address or branch formed from the scalar; 6.7 x the ladder on G.  Affine classical.
  static WCP scalar_mult_base_affine_secret(WBN const& x) {
    WCP r{WBN::uninitialized(x.size()), WBN::uninitialized(x.size())};
    hip::check(ecsimd_hip_scalar_mult_base(hip::context(), curve_id(), x.data(), r.x().data(), r.y().data(), nullptr, x.size(),
                                           ECSIMD_HIP_OUT_AFFINE | ECSIMD_HIP_ALG_WINDOWED | ECSIMD_HIP_ALG_CONSTANT_TIME), "ecsimd_hip_scalar_mult_base");
    return r;
  }
  // u1[i] * G + u2[i] * Q[i] (the ECDSA-verification shape), affine classical; finite[i] is false where the sum
  // is the point at infinity (coordinates (0, 0)).
  static WCP double_scalar_mult(WBN const& u1, WBN const& u2, WCP const& Q, hip::mask& finite) {
    same_length(u1.size(), Q.size(), "double_scalar_mult"); same_length(u2.size(), Q.size(), "double_scalar_mult");
    WCP r{WBN::uninitialized(Q.size()), WBN::uninitialized(Q.size())};
    finite = hip::mask(Q.size());
    hip::check(ecsimd_hip_double_scalar_mult(hip::context(), curve_id(), u1.data(), u2.data(), Q.x().data(), Q.y().data(), r.x().data(), r.y().data(),
                                             finite.data(), Q.size()), "ecsimd_hip_double_scalar_mult");
    return r;
  }
  // ECDSA's acceptance test for precomputed u1 = e/s, u2 = r/s (mod n): lane i is set iff u1*G + u2*Q is finite and its x mod n == r.
  static hip::mask ecdsa_verify_rx(WBN const& u1, WBN const& u2, WCP const& Q, WBN const& r) {
    same_length(u1.size(), Q.size(), "ecdsa_verify_rx"); same_length(u2.size(), Q.size(), "ecdsa_verify_rx"); same_length(r.size(), Q.size(), "ecdsa_verify_rx");
    hip::mask ok(Q.size());
    hip::check(ecsimd_hip_ecdsa_verify_rx(hip::context(), curve_id(), u1.data(), u2.data(), Q.x().data(), Q.y().data(), r.data(), ok.data(), Q.size()), "ecsimd_hip_ecdsa_verify_rx");
    return ok;
  }
  // The whole ECDSA verification (SEC 1 v2 4.1.4): e = the digest as an integer (any 256-bit value), (r, s) the signature, Q the public key.
  // Lane i is set iff 1 <= r, s < n, Q is a valid public key and x((e/s) G + (r/s) Q) mod n == r.  The arithmetic modulo the group order
  // runs on the device (the field layer on the order's field id; GFp<WBN, p256_order> / GFp<WBN, secp256k1_order> is the same arithmetic).
  static hip::mask ecdsa_verify(WBN const& e, WBN const& r, WBN const& s, WCP const& Q) {
    same_length(e.size(), Q.size(), "ecdsa_verify"); same_length(r.size(), Q.size(), "ecdsa_verify"); same_length(s.size(), Q.size(), "ecdsa_verify");
    hip::mask ok(Q.size());
    hip::check(ecsimd_hip_ecdsa_verify(hip::context(), curve_id(), e.data(), r.data(), s.data(), Q.x().data(), Q.y().data(), ok.data(), Q.size()), "ecsimd_hip_ecdsa_verify");
    return ok;
  }
  // ECDSA signing (SEC 1 v2 4.1.3): digests e, private keys d, the CALLER's nonces k (RFC 6979 or a DRBG).  Returns (r, s); ok[i] is false -- and
  // r = s = 0 -- where d or k is not in [1, n) or r or s came out zero.  k G runs on the constant-time comb; no branch or address depends on d or k.
  static std::pair<WBN, WBN> ecdsa_sign(WBN const& e, WBN const& d, WBN const& k, hip::mask& ok) {
    same_length(e.size(), d.size(), "ecdsa_sign"); same_length(k.size(), d.size(), "ecdsa_sign");
    auto r = WBN::uninitialized(d.size()), s = WBN::uninitialized(d.size());
    ok = hip::mask(d.size());
    hip::check(ecsimd_hip_ecdsa_sign(hip::context(), curve_id(), e.data(), d.data(), k.data(), r.data(), s.data(), ok.data(), d.size()), "ecsimd_hip_ecdsa_sign");
    return {r, s};
  }
  // ---- several GPUs: k[i] * P[i] for HOST arrays, sharded over a device group.  P affine classical (x, y);
  // the result is what scalar_mult(x, from_affine(P)) returns lane by lane -- Jacobian, Montgomery form -- or, with
  // affine_out, what .to_affine() of it returns.  Member m computes the slice device_group::shard_range(n, m, size());
  // one gather brings the shards to the first device and the result to the host.
  struct host_points { std::vector<BN> x, y, z; };            // z is empty for affine output
  static host_points scalar_mult(hip::device_group& g, std::vector<BN> const& k, std::vector<BN> const& px, std::vector<BN> const& py, bool affine_out = false) {
    if (k.size() != px.size() || k.size() != py.size()) throw hip::error("ecsimd: scalar_mult over host arrays of different length");
    static_assert(sizeof(BN) == 32, "a 256-bit element is 4 x u64, contiguous");
    const size_t n = k.size();
    host_points r; r.x.resize(n); r.y.resize(n); if (!affine_out) r.z.resize(n);
    auto w = [](std::vector<BN> const& v) { return reinterpret_cast<const uint64_t*>(v.data()); };
    auto m = [](std::vector<BN>& v) { return v.empty() ? nullptr : reinterpret_cast<uint64_t*>(v.data()); };
    g.check(ecsimd_hip_group_scalar_mult_host(g.handle(), curve_id(), w(k), w(px), w(py), m(r.x), m(r.y), m(r.z), n,
                                              ECSIMD_HIP_BASE_CLASSICAL | (affine_out ? ECSIMD_HIP_OUT_AFFINE : ECSIMD_HIP_OUT_JACOBIAN)), "ecsimd_hip_group_scalar_mult_host");
    return r;
  }
 private:
  // The C ABI takes one length for all operands (in the reference it is a property of the type): a shorter batch would be
  // read or written out of bounds on the device.
  static void same_length(size_t a, size_t b, const char* what) {
    if (a != b) throw hip::error(std::string("ecsimd: ") + what + " over batches of different length");
  }
  static WJCP fresh_xy(size_t n) {                  // z is attached by the caller (a shared co-Z array)
    WJCP r; r.x() = gfp{WMBN{WBN::uninitialized(n)}}; r.y() = gfp{WMBN{WBN::uninitialized(n)}}; return r;
  }
  static WJCP fresh(size_t n) {
    WJCP r; r.x() = gfp{WMBN{WBN::uninitialized(n)}}; r.y() = gfp{WMBN{WBN::uninitialized(n)}}; r.z() = gfp{WMBN{WBN::uninitialized(n)}}; return r;
  }
  static uint64_t* px(WJCP const& p) { return p.x().wbn().data(); }
  static uint64_t* py(WJCP const& p) { return p.y().wbn().data(); }
  static uint64_t* pz(WJCP const& p) { return p.z().wbn().data(); }
};
}  // namespace ecsimd
#endif
