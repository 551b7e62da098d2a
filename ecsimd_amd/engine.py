"""ctypes binding of include/ecsimd_hip.h over torch device tensors.

Tensors are ``torch.int64`` (bit pattern of the u64 limbs) or ``torch.uint64`` of shape (n, 4)
[(n, 8) for 512-bit products], contiguous, on the engine's device.  Every method enqueues on
torch's CURRENT stream and returns without synchronising, like any torch op.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

P256, SECP256K1 = 0, 1
CURVES = {"p256": P256, "secp256k1": SECP256K1}
# field ids of the two group orders (include/ecsimd_hip.h enum ecsimd_hip_field): accepted wherever a method below takes a field's `curve`
P256_ORDER, SECP256K1_ORDER = 2, 3
ORDER_FIELD = {P256: P256_ORDER, SECP256K1: SECP256K1_ORDER}
MODULUS_PRIME = 1

_HERE = os.path.dirname(os.path.abspath(__file__))


class EcsimdHipError(RuntimeError):
    pass


def lib_path() -> str:
    # ECSIMD_HIP_LIBRARY: another build of the SAME library (tools/ compares kernel variants with it); never a fallback
    return os.environ.get("ECSIMD_HIP_LIBRARY") or os.path.join(_HERE, "libecsimd_hip.so")


_SYMBOLS = None


def declared_symbols():
    """Every function name declared in include/ecsimd_hip.h (parsed from the header)."""
    global _SYMBOLS
    if _SYMBOLS is None:
        import re
        hdr = os.path.join(os.path.dirname(_HERE), "include", "ecsimd_hip.h")
        text = open(hdr).read()
        _SYMBOLS = sorted(set(re.findall(r"\b(ecsimd_hip_[a-z0-9_]+)\s*\(", text)))
    return _SYMBOLS


def load_library() -> C.CDLL:
    path = lib_path()
    if not os.path.exists(path):
        raise EcsimdHipError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback for the HIP path)")
    lib = C.CDLL(path)
    lib.ecsimd_hip_last_error.restype = C.c_char_p
    lib.ecsimd_hip_version.restype = C.c_char_p
    return lib


def _u64(v):
    return C.c_uint64(int(v) & 0xFFFFFFFFFFFFFFFF)


class Engine:
    """One context (HIP stream owner) on one GPU."""

    def __init__(self, device: int = 0):
        import torch
        self.torch = torch
        self.lib = load_library()
        self.device = int(device)
        self.ctx = C.c_void_p()
        rc = self.lib.ecsimd_hip_init(C.c_int(self.device), C.byref(self.ctx))
        if rc != 0:
            raise EcsimdHipError(f"ecsimd_hip_init(device={device}) failed with {rc} "
                                 "(-2 = no gfx950 device visible; the HIP path has no CPU fallback)")
        self.tdev = torch.device("cuda", self.device)
        self._rows = set()      # batch lengths of the tensors handed to the call being assembled (see _ptr / _call)

    def close(self):
        if self.ctx:
            self.lib.ecsimd_hip_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plumbing
    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.ecsimd_hip_last_error(self.ctx)
            raise EcsimdHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def _bind_stream(self):
        s = self.torch.cuda.current_stream(self.tdev).cuda_stream
        self._check(self.lib.ecsimd_hip_set_stream(self.ctx, C.c_void_p(s)), "set_stream")

    def _ptr(self, t, words=4, dtype_ok=None):
        torch = self.torch
        if t is None:
            return C.c_void_p(0)
        assert t.is_cuda and t.device.index == self.device, "tensor on the wrong device"
        assert t.is_contiguous(), "tensor must be contiguous"
        if words:
            assert t.dtype in (torch.int64, torch.uint64) and t.dim() == 2 and t.shape[1] == words, (t.dtype, t.shape)
        else:
            assert t.dtype == torch.uint8 and t.dim() == 1
        self._rows.add(int(t.shape[0]))
        return C.c_void_p(t.data_ptr())

    def empty(self, n, words=4):
        return self.torch.empty((n, words), dtype=self.torch.int64, device=self.tdev)

    def flags(self, n):
        return self.torch.zeros((n,), dtype=self.torch.uint8, device=self.tdev)

    def to_device(self, arr):
        """numpy uint64 (n, w) -> device int64 tensor with the same bits."""
        a = np.ascontiguousarray(arr, dtype=np.uint64)
        return self.torch.from_numpy(a.view(np.int64)).to(self.tdev)

    def select_rows(self, t, rows):
        """t[rows] for a numpy index array (contiguous result on the device)."""
        return t[self.torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int64)).to(t.device)].contiguous()

    @staticmethod
    def to_numpy(t):
        a = t.detach().cpu().numpy()
        return a if a.dtype == np.uint8 else a.view(np.uint64)

    def _call(self, name, *args):
        # The C ABI takes ONE length for all operands (in the reference it is a compile-time property of the type): a
        # shorter tensor would be read or written out of bounds on the device, so it is refused here.
        rows, self._rows = self._rows, set()
        if len(rows) > 1:
            raise EcsimdHipError(f"{name}: operands disagree on the batch length: {sorted(rows)}")
        self._bind_stream()
        self._check(getattr(self.lib, "ecsimd_hip_" + name)(self.ctx, *args), name)

    def set_ref_square_compat(self, on: bool):
        """ecsimd_hip_set_ref_square_compat: square like the reference's square() as written (mul.h:160-212)."""
        self._check(self.lib.ecsimd_hip_set_ref_square_compat(self.ctx, C.c_int(int(bool(on)))), "set_ref_square_compat")

    def sync(self):
        self._bind_stream()
        self._check(self.lib.ecsimd_hip_sync(self.ctx), "sync")

    def constant(self, curve, which):
        out = (C.c_uint64 * 4)()
        rc = self.lib.ecsimd_hip_get_constant(C.c_int(curve), C.c_int(which), out)
        if rc != 0:
            raise EcsimdHipError(f"get_constant({curve},{which}) -> {rc}")
        return np.array(list(out), dtype=np.uint64)

    # ---- L2
    def add(self, a, b):
        n = a.shape[0]; out = self.empty(n); f = self.flags(n)
        self._call("add", self._ptr(a), self._ptr(b), self._ptr(out), self._ptr(f, 0), C.c_size_t(n)); return out, f

    def sub(self, a, b):
        n = a.shape[0]; out = self.empty(n); f = self.flags(n)
        self._call("sub", self._ptr(a), self._ptr(b), self._ptr(out), self._ptr(f, 0), C.c_size_t(n)); return out, f

    def sub_if_above(self, a, p):
        n = a.shape[0]; out = self.empty(n)
        self._call("sub_if_above", self._ptr(a), self._ptr(p), self._ptr(out), C.c_size_t(n)); return out

    def cmp_eq(self, a, b):
        n = a.shape[0]; f = self.flags(n)
        self._call("cmp_eq", self._ptr(a, a.shape[1]), self._ptr(b, b.shape[1]), C.c_int(a.shape[1]), self._ptr(f, 0), C.c_size_t(n)); return f

    def mask_op(self, op, a, b=None):
        n = a.shape[0]; f = self.flags(n)
        self._call("mask_op", C.c_int(op), self._ptr(a, 0), self._ptr(b, 0), self._ptr(f, 0), C.c_size_t(n)); return f

    def mask_count(self, a):
        c = C.c_size_t(0)
        self._call("mask_count", self._ptr(a, 0), C.c_size_t(a.shape[0]), C.byref(c)); return int(c.value)

    def cmp_lt(self, a, b):
        n = a.shape[0]; f = self.flags(n)
        self._call("cmp_lt", self._ptr(a), self._ptr(b), self._ptr(f, 0), C.c_size_t(n)); return f

    def shift_left_one(self, a):
        n = a.shape[0]; out = self.empty(n); f = self.flags(n)
        self._call("shift_left_one", self._ptr(a), self._ptr(out), self._ptr(f, 0), C.c_size_t(n)); return out, f

    def mul(self, a, b):
        n = a.shape[0]; out = self.empty(n, 8)
        self._call("mul", self._ptr(a), self._ptr(b), self._ptr(out, 8), C.c_size_t(n)); return out

    def square(self, a):
        n = a.shape[0]; out = self.empty(n, 8)
        self._call("square", self._ptr(a), self._ptr(out, 8), C.c_size_t(n)); return out

    def if_else(self, mask, a, b):
        n = a.shape[0]; out = self.empty(n)
        self._call("if_else", self._ptr(mask, 0), self._ptr(a), self._ptr(b), self._ptr(out), C.c_size_t(n)); return out

    def swap_if(self, mask, a, b):
        self._call("swap_if", self._ptr(mask, 0), self._ptr(a), self._ptr(b), C.c_size_t(a.shape[0]))

    # ---- wire formats
    def _bytes_ptr(self, t):
        assert t.is_cuda and t.is_contiguous() and t.dtype == self.torch.uint8
        return C.c_void_p(t.data_ptr())

    def from_bytes_be(self, b):
        n = b.numel() // 32; out = self.empty(n)
        self._call("from_bytes_be", self._bytes_ptr(b), self._ptr(out), C.c_size_t(n)); return out

    def to_bytes_be(self, a):
        n = a.shape[0]; out = self.torch.empty((n, 32), dtype=self.torch.uint8, device=self.tdev)
        self._call("to_bytes_be", self._ptr(a), self._bytes_ptr(out), C.c_size_t(n)); return out

    def wide4_to_lanes(self, wides, record_bytes=128, offset_bytes=0, n_wides=None):
        """ecsimd_hip_wide4_to_lanes: a device uint8 tensor holding records with one reference wide (u64[limb * 4 + lane]) each -> (4 * wides, 4) elements."""
        n_wides = wides.numel() // record_bytes if n_wides is None else n_wides
        out = self.empty(4 * n_wides)
        self._call("wide4_to_lanes", self._bytes_ptr(wides), C.c_size_t(record_bytes), C.c_size_t(offset_bytes), self._ptr(out), C.c_size_t(n_wides)); return out

    def lanes_to_wide4(self, a, wides, record_bytes=128, offset_bytes=0):
        """ecsimd_hip_lanes_to_wide4: elements 4w .. 4w + 3 of `a` -> the wide at offset_bytes of record w of the device uint8 tensor `wides` (in place)."""
        self._call("lanes_to_wide4", self._ptr(a), self._bytes_ptr(wides), C.c_size_t(record_bytes), C.c_size_t(offset_bytes), C.c_size_t(a.shape[0] // 4)); return wides

    def mask_bit(self, a, bit):
        n = a.shape[0]; f = self.flags(n)
        self._call("mask_bit", self._ptr(a), C.c_int(bit), self._ptr(f, 0), C.c_size_t(n)); return f

    def sec1_encode(self, curve, x, y, compressed=False):
        n = x.shape[0]; out = self.torch.empty((n, 33 if compressed else 65), dtype=self.torch.uint8, device=self.tdev)
        self._call("sec1_encode", C.c_int(curve), self._ptr(x), self._ptr(y), self._bytes_ptr(out), C.c_size_t(n), C.c_int(int(compressed))); return out

    def sec1_decode(self, curve, rec, compressed=False):
        n = rec.shape[0]; x, y, ok = self.empty(n), self.empty(n), self.flags(n)
        self._call("sec1_decode", C.c_int(curve), self._bytes_ptr(rec), self._ptr(x), self._ptr(y), self._ptr(ok, 0), C.c_size_t(n), C.c_int(int(compressed))); return x, y, ok

    # ---- L3
    def _bin(self, name, curve, a, b):
        n = a.shape[0]; out = self.empty(n)
        self._call(name, C.c_int(curve), self._ptr(a), self._ptr(b), self._ptr(out), C.c_size_t(n)); return out

    def _un(self, name, curve, a, words=4):
        n = a.shape[0]; out = self.empty(n)
        self._call(name, C.c_int(curve), self._ptr(a, words), self._ptr(out), C.c_size_t(n)); return out

    def mod_add(self, curve, a, b): return self._bin("mod_add", curve, a, b)
    def mod_sub(self, curve, a, b): return self._bin("mod_sub", curve, a, b)
    def mod_mul(self, curve, a, b): return self._bin("mod_mul", curve, a, b)
    def mgry_mul(self, curve, a, b): return self._bin("mgry_mul", curve, a, b)
    def mgry_sqr(self, curve, a): return self._un("mgry_sqr", curve, a)
    def mgry_reduce(self, curve, a8): return self._un("mgry_reduce", curve, a8, 8)
    def mgry_from_classical(self, curve, a): return self._un("mgry_from_classical", curve, a)
    def mgry_to_classical(self, curve, a): return self._un("mgry_to_classical", curve, a)
    def gfp_inverse(self, curve, a): return self._un("gfp_inverse", curve, a)
    def gfp_opposite(self, curve, a): return self._un("gfp_opposite", curve, a)

    def mod_shift_left(self, curve, a, count):
        n = a.shape[0]; out = self.empty(n)
        self._call("mod_shift_left", C.c_int(curve), self._ptr(a), C.c_int(count), self._ptr(out), C.c_size_t(n)); return out

    def mgry_pow(self, curve, a, exponent):
        n = a.shape[0]; out = self.empty(n)
        e = (C.c_uint64 * 4)(*[int(v) for v in np.asarray(exponent, dtype=np.uint64).reshape(4)])
        self._call("mgry_pow", C.c_int(curve), self._ptr(a), e, self._ptr(out), C.c_size_t(n)); return out

    def gfp_sqrt(self, curve, a):
        n = a.shape[0]; out = self.empty(n); ok = self.flags(n)
        self._call("gfp_sqrt", C.c_int(curve), self._ptr(a), self._ptr(out), self._ptr(ok, 0), C.c_size_t(n)); return out, ok

    # ---- points (in-out arguments are updated IN PLACE, like the reference's reference parameters)
    def from_affine(self, curve, x, y):
        n = x.shape[0]; j = [self.empty(n) for _ in range(3)]
        self._call("from_affine", C.c_int(curve), self._ptr(x), self._ptr(y), *[self._ptr(t) for t in j], C.c_size_t(n)); return tuple(j)

    def to_affine(self, curve, j, x_only=False):
        n = j[0].shape[0]; x = self.empty(n); y = None if x_only else self.empty(n)
        self._call("to_affine", C.c_int(curve), *[self._ptr(t) for t in j], self._ptr(x), self._ptr(y), C.c_size_t(n)); return x, y

    def on_curve(self, curve, x, y):
        n = x.shape[0]; ok = self.flags(n)
        self._call("on_curve", C.c_int(curve), self._ptr(x), self._ptr(y), self._ptr(ok, 0), C.c_size_t(n)); return ok

    def compute_y(self, curve, x):
        n = x.shape[0]; y = self.empty(n); ok = self.flags(n)
        self._call("compute_y", C.c_int(curve), self._ptr(x), self._ptr(y), self._ptr(ok, 0), C.c_size_t(n)); return y, ok

    def dblu(self, curve, p):
        n = p[0].shape[0]; r = [self.empty(n) for _ in range(3)]
        self._call("dblu", C.c_int(curve), *[self._ptr(t) for t in p], *[self._ptr(t) for t in r], C.c_size_t(n)); return tuple(r)

    def trplu(self, curve, p):
        n = p[0].shape[0]; r = [self.empty(n) for _ in range(3)]
        self._call("trplu", C.c_int(curve), *[self._ptr(t) for t in p], *[self._ptr(t) for t in r], C.c_size_t(n)); return tuple(r)

    def zaddu(self, curve, p, o):
        n = p[0].shape[0]; r = [self.empty(n) for _ in range(3)]
        self._call("zaddu", C.c_int(curve), *[self._ptr(t) for t in p], *[self._ptr(t) for t in o], *[self._ptr(t) for t in r], C.c_size_t(n)); return tuple(r)

    def zdau(self, curve, p, q):
        n = p[0].shape[0]; r = [self.empty(n) for _ in range(3)]
        self._call("zdau", C.c_int(curve), *[self._ptr(t) for t in p], *[self._ptr(t) for t in q], *[self._ptr(t) for t in r], C.c_size_t(n)); return tuple(r)

    def zdau_repeat(self, curve, p, qxy, iters, swap_bits=0, radix=29):
        """ecsimd_hip_zdau_repeat: ZDAU `iters` times in registers; returns (rx, ry, sx, sy, z) = the final P, the final Q, their Z."""
        n = p[0].shape[0]; r = [self.empty(n) for _ in range(5)]
        self._call("zdau_repeat", C.c_int(curve), *[self._ptr(t) for t in p], *[self._ptr(t) for t in qxy], *[self._ptr(t) for t in r], C.c_size_t(n),
                   C.c_int(iters), C.c_uint64(swap_bits), C.c_int(radix)); return tuple(r)

    def add_mixed_complete(self, curve, a, bxy):
        n = a[0].shape[0]; r = [self.empty(n) for _ in range(3)]
        self._call("add_mixed_complete", C.c_int(curve), *[self._ptr(t) for t in a], *[self._ptr(t) for t in bxy], *[self._ptr(t) for t in r], C.c_size_t(n)); return tuple(r)

    def add_z2_1(self, curve, a, bxy):
        n = a[0].shape[0]; r = [self.empty(n) for _ in range(3)]
        self._call("add_z2_1", C.c_int(curve), *[self._ptr(t) for t in a], *[self._ptr(t) for t in bxy], *[self._ptr(t) for t in r], C.c_size_t(n)); return tuple(r)

    def scalar_mult(self, curve, k, x, y, flags=0, out=None, x_only=False):
        """x_only (with OUT_AFFINE): no y output -- returns (x, None); ECDH's shared secret is x of k*Q."""
        n = k.shape[0]
        r = out if out is not None else self._fresh_out(n, flags, x_only)
        self._call("scalar_mult", C.c_int(curve), self._ptr(k), self._ptr(x), self._ptr(y), *[self._ptr(t) for t in r], C.c_size_t(n), C.c_int(flags))
        return tuple(r[:2]) if flags & 2 else tuple(r)

    def _fresh_out(self, n, flags, x_only):
        if x_only:
            if not flags & 2:
                raise ValueError("x_only needs OUT_AFFINE")
            return [self.empty(n), None, None]
        return [self.empty(n) for _ in range(3)]

    def scalar_mult_1s(self, curve, k1, x, y, flags=0, x_only=False):
        n = x.shape[0]; r = self._fresh_out(n, flags, x_only)
        e = (C.c_uint64 * 4)(*[int(v) for v in np.asarray(k1, dtype=np.uint64).reshape(4)])
        self._call("scalar_mult_1s", C.c_int(curve), e, self._ptr(x), self._ptr(y), *[self._ptr(t) for t in r], C.c_size_t(n), C.c_int(flags))
        return tuple(r[:2]) if flags & 2 else tuple(r)

    def scalar_mult_host(self, curve, k, x=None, y=None, flags=0, x_only=False, out=None):
        """ecsimd_hip_scalar_mult_host: numpy uint64 (n, 4) arrays in HOST memory in and out (x = y = None: the generator); chunked, copies overlapped with ladders."""
        k = np.ascontiguousarray(k, dtype=np.uint64); n = k.shape[0]
        ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)
        if x is None:
            from .flags import BASE_GENERATOR
            flags |= BASE_GENERATOR
        else:
            x = np.ascontiguousarray(x, dtype=np.uint64); y = np.ascontiguousarray(y, dtype=np.uint64)
            assert x.shape == y.shape == k.shape
        outs = (list(out) if out is not None else [np.empty_like(k) for _ in range(1 if (flags & 2 and x_only) else 2 if flags & 2 else 3)]) + [None, None]     # (out: arrays to reuse -- fresh ones are first touched inside the call)
        self._bind_stream()
        self._check(self.lib.ecsimd_hip_scalar_mult_host(self.ctx, C.c_int(curve), ptr(k), ptr(x), ptr(y), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), C.c_size_t(n), C.c_int(flags)), "scalar_mult_host")
        return tuple(o for o in outs if o is not None)

    def scalar_mult_base(self, curve, k, flags=0, out=None, x_only=False):
        n = k.shape[0]
        r = out if out is not None else self._fresh_out(n, flags, x_only)
        self._call("scalar_mult_base", C.c_int(curve), self._ptr(k), *[self._ptr(t) for t in r], C.c_size_t(n), C.c_int(flags))
        return tuple(r[:2]) if flags & 2 else tuple(r)

    def affine_add(self, curve, a, b):
        n = a[0].shape[0]; rx, ry, fin = self.empty(n), self.empty(n), self.flags(n)
        self._call("affine_add", C.c_int(curve), self._ptr(a[0]), self._ptr(a[1]), self._ptr(b[0]), self._ptr(b[1]), self._ptr(rx), self._ptr(ry), self._ptr(fin, 0), C.c_size_t(n))
        return rx, ry, fin

    def double_scalar_mult(self, curve, u1, u2, qx, qy, x_only=False):
        n = u1.shape[0]; rx, fin = self.empty(n), self.flags(n); ry = None if x_only else self.empty(n)
        self._call("double_scalar_mult", C.c_int(curve), self._ptr(u1), self._ptr(u2), self._ptr(qx), self._ptr(qy), self._ptr(rx), self._ptr(ry), self._ptr(fin, 0), C.c_size_t(n))
        return rx, ry, fin

    def ecdsa_verify_rx(self, curve, u1, u2, qx, qy, r):
        n = u1.shape[0]; ok = self.flags(n)
        self._call("ecdsa_verify_rx", C.c_int(curve), self._ptr(u1), self._ptr(u2), self._ptr(qx), self._ptr(qy), self._ptr(r), self._ptr(ok, 0), C.c_size_t(n))
        return ok

    def ecdsa_verify(self, curve, e, r, s, qx, qy):
        """ecsimd_hip_ecdsa_verify: the whole verification, u1 = e/s and u2 = r/s modulo the group order computed on the device."""
        n = e.shape[0]; ok = self.flags(n)
        self._call("ecdsa_verify", C.c_int(curve), self._ptr(e), self._ptr(r), self._ptr(s), self._ptr(qx), self._ptr(qy), self._ptr(ok, 0), C.c_size_t(n))
        return ok

    def ecdsa_sign(self, curve, e, d, k):
        """ecsimd_hip_ecdsa_sign: (r, s, ok) for digests e, private keys d and caller-supplied nonces k."""
        n = e.shape[0]; r, s, ok = self.empty(n), self.empty(n), self.flags(n)
        self._call("ecdsa_sign", C.c_int(curve), self._ptr(e), self._ptr(d), self._ptr(k), self._ptr(r), self._ptr(s), self._ptr(ok, 0), C.c_size_t(n))
        return r, s, ok

    def fe29_raw(self, curve, op, inputs, swap=0):
        """ecsimd_hip_fe29_raw: one function of the reduced-radix layer on raw int32 limbs; `inputs` is an int32 tensor (n, NIN, 9); returns (n, NOUT, 9)."""
        torch = self.torch
        nout = 6 if op in (0, 10) else 1 if op in (7, 8) else 4 if op == 9 else 3
        assert inputs.dtype == torch.int32 and inputs.dim() == 3 and inputs.shape[2] == 9 and inputs.is_contiguous() and inputs.device.index == self.device
        n = inputs.shape[0]
        out = torch.empty((n, nout, 9), dtype=torch.int32, device=self.tdev)
        self._bind_stream()
        self._check(self.lib.ecsimd_hip_fe29_raw(self.ctx, C.c_int(curve), C.c_int(op), C.c_void_p(inputs.data_ptr()), C.c_void_p(out.data_ptr()), C.c_size_t(n), C.c_int(swap)), "fe29_raw")
        return out

    def workspace_bytes(self):
        """ecsimd_hip_workspace_info as a uint8 numpy copy of the context's scratch block (diagnostic: what the last call left behind)."""
        ptr, size = C.c_void_p(), C.c_size_t()
        self._check(self.lib.ecsimd_hip_workspace_info(self.ctx, C.byref(ptr), C.byref(size)), "workspace_info")
        self.torch.cuda.synchronize(self.tdev)
        host = np.empty(size.value, dtype=np.uint8)
        if size.value:
            self._check(self.lib.ecsimd_hip_memcpy_d2h(self.ctx, host.ctypes.data_as(C.c_void_p), ptr, C.c_size_t(size.value)), "memcpy_d2h")
        return host

    def scalar_mult_p256(self, k, xm, ym, out=None):
        n = k.shape[0]
        r = out if out is not None else [self.empty(n) for _ in range(3)]
        self._call("scalar_mult_p256", self._ptr(k), self._ptr(xm), self._ptr(ym), *[self._ptr(t) for t in r], C.c_size_t(n))
        return tuple(r)

    # ---- synthetic inputs / measurement
    def fill_random(self, n, seed, stream, first_index=0, clear_top_bits=0, out=None):
        t = out if out is not None else self.empty(n)
        self._call("fill_random", self._ptr(t), C.c_size_t(n), _u64(seed), _u64(stream), _u64(first_index), C.c_int(clear_top_bits)); return t

    def peak_mad32(self, iters=2048, reps=1):
        """(mad32 executed, milliseconds) of the fastest of `reps` runs of the dependency-free
        v_mad_u64_u32 stream (several runs let the clock settle: the probe is only a few ms long)."""
        best = None
        for _ in range(max(1, reps)):
            mads, ms = C.c_double(0), C.c_double(0)
            self._call("peak_mad32", C.c_int(iters), C.byref(mads), C.byref(ms))
            if best is None or mads.value / ms.value > best[0] / best[1]:
                best = (mads.value, ms.value)
        return best


def register_modulus(p: int, prime: bool = False) -> int:
    """ecsimd_hip_register_modulus: the field id of the odd modulus p (process-wide; no GPU needed)."""
    lib = load_library()
    limbs = (C.c_uint64 * 4)(*[(p >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)])
    fid = C.c_int(-1)
    rc = lib.ecsimd_hip_register_modulus(limbs, C.c_int(MODULUS_PRIME if prime else 0), C.byref(fid))
    if rc != 0:
        raise EcsimdHipError(f"ecsimd_hip_register_modulus({p:#x}) failed with {rc} (the modulus must be odd and >= 3)")
    return fid.value


CURVE_GENERIC_KERNELS = 1
FIRST_REGISTERED_CURVE = 0x10000


def register_curve(p: int, a: int, b: int, gx: int, gy: int, n: int | None = None, generic_kernels: bool = False) -> int:
    """ecsimd_hip_register_curve: the curve id of y^2 = x^3 + a x + b over GF(p) with generator (gx, gy) -- the reference's curve_group<Curve> for any
    Curve, as a run-time registration (process-wide; no GPU needed).  generic_kernels: register P-256 / secp256k1 parameters like any other curve."""
    lib = load_library()
    lim = lambda v: (C.c_uint64 * 4)(*[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)])
    cid = C.c_int(-1)
    rc = lib.ecsimd_hip_register_curve(lim(p), lim(a % p), lim(b % p), lim(gx), lim(gy), lim(n) if n is not None else None,
                                       C.c_int(CURVE_GENERIC_KERNELS if generic_kernels else 0), C.byref(cid))
    if rc != 0:
        raise EcsimdHipError(f"ecsimd_hip_register_curve(p = {p:#x}) failed with {rc}: p must be a prime = 3 mod 4, the generator on the curve, the curve non-singular")
    return cid.value


CURVE_HAS_ORDER, CURVE_COMB, CURVE_ECDSA, CURVE_WINDOW_VARIABLE_BASE = 1, 2, 4, 8


def curve_capabilities(curve: int) -> int:
    """ecsimd_hip_curve_capabilities: what an id can do beyond the reference's layers (a mask of CURVE_*; host only)."""
    lib = load_library()
    caps = C.c_int(0)
    rc = lib.ecsimd_hip_curve_capabilities(C.c_int(curve), C.byref(caps))
    if rc != 0:
        raise EcsimdHipError(f"ecsimd_hip_curve_capabilities({curve}) -> {rc}: unknown curve id")
    return caps.value


def shard_range_c(n_total: int, member: int, members: int):
    """ecsimd_hip_shard_range: the C ABI's partition of a batch over a device group (pure host arithmetic)."""
    lib = load_library()
    first, count = C.c_size_t(0), C.c_size_t(0)
    rc = lib.ecsimd_hip_shard_range(C.c_size_t(n_total), C.c_int(member), C.c_int(members), C.byref(first), C.byref(count))
    if rc != 0:
        raise EcsimdHipError(f"ecsimd_hip_shard_range({n_total}, {member}, {members}) -> {rc}")
    return int(first.value), int(count.value)


class DeviceGroup:
    """ecsimd_hip_group_*: one batch over several GPUs behind the C ABI (one context per device, contiguous shards,
    one RCCL gather to member 0).  `devices` may list a device twice (members then exchange by device copies)."""

    def __init__(self, devices):
        import torch
        self.torch = torch
        self.lib = load_library()
        self.lib.ecsimd_hip_group_last_error.restype = C.c_char_p
        self.devices = [int(d) for d in devices]
        self.g = C.c_void_p()
        arr = (C.c_int * len(self.devices))(*self.devices)
        rc = self.lib.ecsimd_hip_group_init(arr, C.c_int(len(self.devices)), C.byref(self.g))
        if rc != 0:
            raise EcsimdHipError(f"ecsimd_hip_group_init({self.devices}) failed with {rc}")

    def close(self):
        if self.g:
            self.lib.ecsimd_hip_group_destroy(self.g)
            self.g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.ecsimd_hip_group_last_error(self.g)
            raise EcsimdHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    @property
    def size(self):
        return int(self.lib.ecsimd_hip_group_size(self.g))

    @property
    def uses_rccl(self):
        return bool(self.lib.ecsimd_hip_group_uses_rccl(self.g))

    @property
    def rccl_version(self):
        """ncclGetVersion of the RCCL the gather goes through (22606 = 2.26.6); 0 when the group needs none."""
        return int(self.lib.ecsimd_hip_group_rccl_version(self.g))

    def alloc_outputs(self, n, flags=0, x_only=False):
        torch = self.torch
        dev0 = torch.device("cuda", self.devices[0])
        return [torch.empty((n, 4), dtype=torch.int64, device=dev0) for _ in range((1 if x_only else 2) if flags & 2 else 3)]

    def enqueue(self, curve, k_shards, x_shards, y_shards, outs, n, flags=0):
        """ecsimd_hip_group_scalar_mult without the wait: *_shards[m] = member m's slice (a tensor on that member's device),
        outs = 1 (x only), 2 (affine) or 3 (Jacobian) tensors of n rows on member 0's device.  The inputs must be complete
        (they were produced on torch's streams, the group runs on its own)."""
        G = self.size
        ptrs = lambda ts: (C.c_void_p * G)(*[C.c_void_p(t.data_ptr() if t is not None and t.numel() else 0) for t in ts])
        o = [C.c_void_p(t.data_ptr()) for t in outs] + [C.c_void_p(0)] * (3 - len(outs))
        self._check(self.lib.ecsimd_hip_group_scalar_mult(self.g, C.c_int(curve), ptrs(k_shards), ptrs(x_shards), ptrs(y_shards),
                                                          o[0], o[1], o[2], C.c_size_t(n), C.c_int(flags)), "group_scalar_mult")

    def sync(self):
        """ecsimd_hip_group_sync: waits for every member and the gather; returns the last gather's duration in ms (-1: none)."""
        ms = C.c_double(0)
        self._check(self.lib.ecsimd_hip_group_sync(self.g, C.byref(ms)), "group_sync")
        return float(ms.value)

    def member_ms(self, member):
        ms = C.c_double(0)
        self._check(self.lib.ecsimd_hip_group_member_ms(self.g, C.c_int(member), C.byref(ms)), "group_member_ms")
        return float(ms.value)

    def scalar_mult(self, curve, k_shards, x_shards, y_shards, n, flags=0, x_only=False):
        """Device-resident form, synchronous: returns (ox[, oy[, oz]]) on member 0's device and the gather time in ms."""
        outs = self.alloc_outputs(n, flags, x_only)
        for m in range(self.size):
            self.torch.cuda.synchronize(self.devices[m])         # the inputs were produced on torch's streams
        self.enqueue(curve, k_shards, x_shards, y_shards, outs, n, flags)
        return tuple(outs), self.sync()

    def rccl_selftest(self, elements=1 << 16):
        self._check(self.lib.ecsimd_hip_group_rccl_selftest(self.g, C.c_size_t(elements)), "group_rccl_selftest")

    def scalar_mult_host(self, curve, k, x, y, flags=0, x_only=False):
        """Host-array form: numpy uint64 (n, 4) arrays in, numpy arrays out."""
        k, x, y = (np.ascontiguousarray(a, dtype=np.uint64) for a in (k, x, y))
        n = len(k)
        outs = [np.empty((n, 4), dtype=np.uint64) for _ in range((1 if x_only else 2) if flags & 2 else 3)]
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        o = [p(a) for a in outs] + [C.c_void_p(0)] * (3 - len(outs))
        self._check(self.lib.ecsimd_hip_group_scalar_mult_host(self.g, C.c_int(curve), p(k), p(x), p(y), o[0], o[1], o[2], C.c_size_t(n), C.c_int(flags)),
                    "group_scalar_mult_host")
        return tuple(outs)
