"""Named curves beyond the two with hand-written kernels: public domain parameters (RFC 5639 3.4; GB/T 32918.5 / RFC 8998; ANSSI FRP256v1 -- JORF 241 of
16 October 2011), each y^2 = x^3 + a x + b over GF(p) with p = 3 mod 4, which is what the reference's curve_group<Curve> can be instantiated with
(curve.h:12-15, gfp.h:84).  `curve_id(name)` registers one with the engine (ecsimd_hip_register_curve) and returns its id; "p256" / "secp256k1" are the
built-in ids 0 / 1."""
from .engine import P256, SECP256K1, register_curve

NAMED = {
    "brainpoolP256r1": dict(
        p=0xa9fb57dba1eea9bc3e660a909d838d726e3bf623d52620282013481d1f6e5377, a=0x7d5a0975fc2c3057eef67530417affe7fb8055c126dc5c6ce94a4b44f330b5d9,
        b=0x26dc5c6ce94a4b44f330b5d9bbd77cbf958416295cf7e1ce6bccdc18ff8c07b6, gx=0x8bd2aeb9cb7e57cb2c4b482ffc81b7afb9de27e1e3bd23c23a4453bd9ace3262,
        gy=0x547ef835c3dac4fd97f8461a14611dc9c27745132ded8e545c1d54c72f046997, n=0xa9fb57dba1eea9bc3e660a909d838d718c397aa3b561a6f7901e0e82974856a7),
    "sm2": dict(
        p=0xfffffffeffffffffffffffffffffffffffffffff00000000ffffffffffffffff, a=0xfffffffeffffffffffffffffffffffffffffffff00000000fffffffffffffffc,
        b=0x28e9fa9e9d9f5e344d5a9e4bcf6509a7f39789f515ab8f92ddbcbd414d940e93, gx=0x32c4ae2c1f1981195f9904466a39c9948fe30bbff2660be1715a4589334c74c7,
        gy=0xbc3736a2f4f6779c59bdcee36b692153d0a9877cc62a474002df32e52139f0a0, n=0xfffffffeffffffffffffffffffffffff7203df6b21c6052b53bbf40939d54123),
    "frp256v1": dict(
        p=0xf1fd178c0b3ad58f10126de8ce42435b3961adbcabc8ca6de8fcf353d86e9c03, a=0xf1fd178c0b3ad58f10126de8ce42435b3961adbcabc8ca6de8fcf353d86e9c00,
        b=0xee353fca5428a9300d4aba754a44c00fdfec0c9ae4b1a1803075ed967b7bb73f, gx=0xb6b3d4c356c139eb31183d4749d423958c27d2dcaf98b70164c97a2dd98f5cff,
        gy=0x6142e0f7c8b204911f9271f0f3ecef8c2701c307e8e4c9e183115a1554062cfb, n=0xf1fd178c0b3ad58f10126de8ce42435b53dc67e140d2bf941ffdd459c6d655e1),
}


def curve_id(name: str) -> int:
    if name == "p256":
        return P256
    if name == "secp256k1":
        return SECP256K1
    c = NAMED[name]
    return register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"], c["n"])
