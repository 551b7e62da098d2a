"""GPU suite: maximum sizes.  Arrays of 2^27 + 3 elements are 4 GiB + 96 B each -- byte offsets cross 2^31 (element 2^26) and 2^32
(element 2^27) -- so a kernel that forms an offset in 32 bits, or a launcher that clips its grid, answers wrongly for the lanes picked
here: the first, the ones around both crossings, the ragged last ones.  The witness is the SAME library at a batch of 4 099 lanes (which
the rest of `-m gpu` pins to the oracle): every kernel family below must give, at those lanes of the big batch, what it gives for those
lanes alone.  One curve for the long kernels (P-256, the headline), both for the cheap ones; ~60 GB of the 288 GB of HBM."""
import numpy as np
import pytest
import torch

from helpers import P256, SECP256K1, SEED
from ecsimd_amd import OUT_AFFINE, OUT_JACOBIAN, BASE_MGRY, ALG_WINDOWED, ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG

pytestmark = pytest.mark.gpu
N = (1 << 27) + 3


def picked(n):
    parts = [np.arange(0, 1024), np.arange((1 << 26) - 512, (1 << 26) + 512), np.arange((1 << 27) - 1024, n)]
    return np.concatenate(parts).astype(np.int64)


def same(engine, big, small, rows, what):
    for j, (b, s) in enumerate(zip(big, small)):
        if b is None:
            continue
        assert torch.equal(engine.select_rows(b, rows), s), f"{what}: output {j} differs at the picked lanes of the 2^27 + 3 batch"


def test_batches_beyond_4_gib_per_array(engine):
    e = engine
    free, _ = torch.cuda.mem_get_info()
    if free < 80 << 30:
        pytest.skip("needs 80 GB of free device memory")
    rows = picked(N)
    k = e.fill_random(N, SEED, 1)
    s = e.fill_random(N, SEED, 2)
    ks, ss = e.select_rows(k, rows), e.select_rows(s, rows)
    # the synthetic generator itself: element i depends on (seed, i) alone
    for r0 in (0, (1 << 26) - 512, (1 << 27) - 1024):
        m = min(1024 + 3, N - r0)
        assert torch.equal(k[r0:r0 + m], e.fill_random(m, SEED, 1, first_index=r0))
    # element-wise field kernels (both curves), codecs
    for cv in (P256, SECP256K1):
        a = e.fill_random(N, SEED, 11, clear_top_bits=1); b = e.fill_random(N, SEED, 12, clear_top_bits=1)
        as_, bs = e.select_rows(a, rows), e.select_rows(b, rows)
        same(e, [e.mgry_mul(cv, a, b)], [e.mgry_mul(cv, as_, bs)], rows, "mgry_mul")
        same(e, [e.mod_sub(cv, a, b)], [e.mod_sub(cv, as_, bs)], rows, "mod_sub")
        same(e, [e.gfp_inverse(cv, a)], [e.gfp_inverse(cv, as_)], rows, "gfp_inverse (simultaneous inversion)")
        by = e.to_bytes_be(a)
        assert torch.equal(e.from_bytes_be(by), a)
        del a, b, as_, bs, by
    cv = P256
    # fixed base: the three comb kernels + the batched to_affine behind them -- and all three agree on EVERY lane of the big batch
    bx, by = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | ALG_WINDOWED_BIG)
    same(e, (bx, by), e.scalar_mult_base(cv, ss, flags=OUT_AFFINE | ALG_WINDOWED_BIG), rows, "scalar_mult_base 20-bit table")
    for alg, nm in ((ALG_WINDOWED, "4-bit LDS"), (ALG_WINDOWED_SIGNED, "signed 7-bit LDS")):
        big = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | alg)
        same(e, big, e.scalar_mult_base(cv, ss, flags=OUT_AFFINE | alg), rows, f"scalar_mult_base {nm}")
        assert torch.equal(big[0], bx) and torch.equal(big[1], by), f"{nm} and the 20-bit table disagree somewhere in 2^27 + 3 lanes"
        del big
    del s
    bxs, bys = e.select_rows(bx, rows), e.select_rows(by, rows)
    # the headline: variable-base ladder, Montgomery in, Jacobian out
    P = e.from_affine(cv, bx, by); Ps = e.from_affine(cv, bxs, bys)
    same(e, P, Ps, rows, "from_affine")
    xm, ym = P[0], P[1]
    del P
    out = e.scalar_mult(cv, k, xm, ym, flags=BASE_MGRY | OUT_JACOBIAN)
    outs = e.scalar_mult(cv, ks, Ps[0], Ps[1], flags=BASE_MGRY | OUT_JACOBIAN)
    same(e, out, outs, rows, "scalar_mult (ladder)")
    del xm, ym
    aff = e.to_affine(cv, list(out))
    same(e, aff, e.to_affine(cv, list(outs)), rows, "to_affine")
    del out, outs
    # x only (the ladder without Z) and the per-element window tables (chunked over the workspace): three algorithms, every lane
    # (random 256-bit scalars: none of the ladder's degenerate ones, k = 0, +-1 mod n, is among them)
    xo = e.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE, x_only=True)
    same(e, xo, e.scalar_mult(cv, ks, bxs, bys, flags=OUT_AFFINE, x_only=True), rows, "scalar_mult x only")
    assert torch.equal(xo[0], aff[0]), "the ladder without Z and the ladder disagree somewhere in 2^27 + 3 lanes"
    del xo
    w = e.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
    same(e, w, e.scalar_mult(cv, ks, bxs, bys, flags=OUT_AFFINE | ALG_WINDOWED), rows, "scalar_mult windowed")
    assert torch.equal(w[0], aff[0]) and torch.equal(w[1], aff[1]), "the window tables and the ladder disagree somewhere in 2^27 + 3 lanes"
    del w, aff, k, bx, by
    torch.cuda.empty_cache()


def test_one_context_per_host_thread(engine):
    """SURVEY.md 8(b) "Threading": a context is not thread-safe, the rule is one context per host thread -- so four host threads, each
    with its OWN context and stream on the same GPU, run different workloads at the same time (ctypes releases the GIL inside a call;
    every thread builds its own window tables and grows its own workspace) and each gets what the session's context computes for the
    same inputs afterwards, alone.  Fails if the library keeps anything mutable outside its contexts."""
    import threading
    from ecsimd_amd import Engine
    n, rounds = 1 << 16, 3
    results, errors = {}, []

    def work(t):
        try:
            cv = (P256, SECP256K1)[t & 1]
            e = Engine(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                out = []
                for r in range(rounds):
                    k = e.fill_random(n, SEED + t, 1 + r); s = e.fill_random(n, SEED + t, 20 + r)
                    alg = (ALG_WINDOWED_BIG, ALG_WINDOWED, ALG_WINDOWED_SIGNED, ALG_WINDOWED)[(t + r) & 3]
                    bx, by = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | alg)
                    J = e.scalar_mult(cv, k, bx, by)
                    w = e.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED)
                    out.append([x.clone() for x in (bx, by) + tuple(J) + tuple(w)])
                e.sync()
                torch.cuda.current_stream().synchronize()
            results[t] = (cv, out)
            e.close()
        except Exception as exc:                       # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(exc)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    e = engine
    for t, (cv, out) in results.items():
        for r in range(rounds):
            k = e.fill_random(n, SEED + t, 1 + r); s = e.fill_random(n, SEED + t, 20 + r)
            bx, by = e.scalar_mult_base(cv, s, flags=OUT_AFFINE | ALG_WINDOWED_BIG)
            exp = (bx, by) + tuple(e.scalar_mult(cv, k, bx, by)) + tuple(e.scalar_mult(cv, k, bx, by, flags=OUT_AFFINE | ALG_WINDOWED))
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(out[r], exp)), f"thread {t}, round {r}: a context running beside three others computed something else"
