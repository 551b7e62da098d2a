#!/usr/bin/env python3
"""bench.py -- P-256 scalar mults/sec (batched) on N MI355X + fraction of the integer-VALU roofline.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one batch: scalar_mult_p256(k, P) (the reference's
exported entry point, lib/scalar_mult_p256.cpp:10-12 -> curve_group.h:189-218, the co-Z Joye
ladder) over lane-distinct (scalar, point) pairs, Jacobian Montgomery output, inputs resident in HBM
before the timed region.

Default = BASELINE.json configs[3] verbatim: `--scaling strong --global-log2-batch 24`, i.e. 2^24 scalar
multiplications per step, rank r of N owning the contiguous slice shard_range(2^24, r, N) of the synthetic
streams -- 2^24 on one GPU at N = 1, 2^21 per GPU at N = 8.  `--scaling weak
--log2-batch B` keeps 2^B per GPU instead.  No collective on the data path.  For N > 1 each step's result
shard is gathered to rank 0 with ONE RCCL gather on a side stream, overlapped with the next step's
compute (BASELINE.json north_star: "RCCL over xGMI only for the final gather"); the gathers are inside
the timed region, and a second, untimed-for-`value` loop without them gives the compute-only rate
("throughput with and without the gather"; `config.gather`).

Started plainly with --gpus N > 1 (no RANK in the environment, the way the driver starts the N = 1 run) this process
becomes a LAUNCHER: before anything touches a GPU -- it imports neither torch nor the HIP library -- it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` as a
fresh child process, relays rank 0's one JSON line and exits with the child's code.
`--multi group` is the other way to use N GPUs: ONE process, the C ABI's device group (ecsimd_hip_group_*: one context
per device, ncclCommInitAll communicators, the result shards gathered to device 0 by grouped ncclSend / ncclRecv).  In
the one-process-per-GPU mode with N > 1, rank 0 also runs that mode once in a child process after the timed loops
(`multi_group` in the line; --no-group-check skips it) so that a multi-GPU node measures both.
ECSIMD_BENCH_REHEARSE_ONE_GPU=1 (a one-GPU box): N ranks share cuda:0 and gather over gloo through pinned host memory, the
group leg puts N members on the one device -- an N > 1 run's control flow on hardware, RCCL excepted (it refuses two ranks
per device); the line says `rehearsal` and its value is no scaling result.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : integer-VALU bound.  achieved = scalar-mults/s x 555 968 mad32 (SURVEY.md 8(d):
                 (2299 M + 1789 S) x 136), kernel time from HIP events on the launch stream;
                 peak = the dependency-free v_mad_u64_u32 stream measured live on the same GPU.
  cpu_baseline : the REAL reference (oracle/_ref, eve/AVX2) -- or the C port if that library did
                 not travel -- timed on the host cores on a bounded sample of the same workload,
                 and compared bit-for-bit with the GPU result on that sample; every lane that differs
                 (the reference's square() defect) is re-computed by libcrypto.
                 Its sub-object competitor_openssl is the reference's competitor benchmark
                 (benchs/p256_ref.cpp:55-91, libcrypto's EC_POINT_mul) on the same cores, with 8 192 of
                 the GPU's results checked against libcrypto at the affine level.
The process exits NON-ZERO (after printing the line) when a checker disagrees with the GPU in a way the
reference's documented defect does not explain, and on any rank's failure.
"""
import argparse
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAD32_PER_SCALAR_MULT = 555968          # 4088 field mults x 136 mad32
ALGO_BYTES_PER_SCALAR_MULT = 192        # 32 B scalar + 64 B point in, 96 B Jacobian out
# 39.32: the HALF-rate VALU class -- v_mad_u64_u32 / v_mad_i64_i32, the carry adds, 64-bit shifts -- issues one wave64 instruction per 4 cycles
# per SIMD (MI355X_MICROARCH.md: SIMD-32, 2 cycles for a full-rate instruction such as v_add_u32 / v_and_b32 / v_mov_b32); at the 2.4 GHz maximum clock
A_PRIORI_PEAK_TMAD32 = 256 * 4 * 16 * 2.4e9 / 1e12
SEED = 0x5EEDEC51D0000001
EXIT_PARITY = 3                         # a checker contradicts the GPU result
LADDER_WORKLOADS = ("ladder", "ladder-ref-compat", "ladder-radix32")


class CheckerUnavailable(Exception):
    """A CPU checker library could not be loaded or built here (not a disagreement)."""


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: --global-log2-batch units per step in total, split over the ranks (BASELINE configs[3]); "
                         "weak: --log2-batch units per step on every rank")
    ap.add_argument("--global-log2-batch", type=int, default=24, help="strong scaling: scalar mults per step over ALL GPUs = 2^this")
    ap.add_argument("--log2-batch", type=int, default=22, help="weak scaling: scalar mults per GPU per step = 2^this")
    ap.add_argument("--curve", default="p256", choices=["p256", "secp256k1", "brainpoolP256r1", "sm2", "frp256v1"],
                    help="p256 / secp256k1: the curves with special-form kernels; the others (ecsimd_amd/curves.py) are registered at run time -- the reference's "
                         "curve_group<Curve> for any Curve -- and run the ladder workloads on the generic kernels (dense 9-limb prime in SGPRs)")
    ap.add_argument("--workload", default="ladder", choices=["ladder", "ladder-ref-compat", "ladder-radix32", "ladder-x", "windowed", "windowed-ct", "fixed-base", "fixed-base-ct", "fixed-base-signed", "fixed-base-big"],
                    help="ladder: scalar_mult_p256 variable base, the reference's co-Z ladder, Jacobian out (headline, BASELINE configs[3]); "
                         "ladder-ref-compat: the same with ECSIMD_HIP_REF_SQUARE_COMPAT (the reference's square() as written, dropped carry included); "
                         "ladder-radix32: the same with ECSIMD_HIP_LADDER_RADIX32 (the 254 iterations on 8 x 32-bit canonical words instead of nine 29-bit limbs); "
                         "ladder-x: x(k*P) only, the constant-time ladder without its Z coordinate (P-256; ECDH's shared secret; affine-level parity); "
                         "windowed: variable base with per-element tables of 8 multiples of P and signed 4-bit windows, affine out (ALG_WINDOWED; affine-level parity); "
                         "windowed-ct: the same with ALG_CONSTANT_TIME (every entry of the lane's table read in every window; secp256k1 keeps the GLV split, on the complete addition law): secret scalars; "
                         "fixed-base: k*G with the 4-bit-window LDS table + simultaneous inversion, affine out (BASELINE configs[2]); "
                         "fixed-base-ct: the same kernel with ALG_CONSTANT_TIME (every table entry read, kept under lane masks: safe for secret scalars); "
                         "fixed-base-signed: the same with signed 7-bit windows (36 additions instead of 63); "
                         "fixed-base-big: 20-bit windows (odd digits) over a 436 MB table in device memory (12 additions)")
    ap.add_argument("--multi", default="procs", choices=["procs", "group"],
                    help="procs: one process per GPU under torch.distributed.run, one RCCL gather per step (what the driver launches; "
                         "started without it, --gpus N > 1 launches it); group: ONE process driving N GPUs through the C ABI's device "
                         "group (ecsimd_hip_group_*), the gather by grouped ncclSend / ncclRecv")
    ap.add_argument("--no-group-check", action="store_true", help="N > 1, --multi procs: do not run the --multi group leg in a child process")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    return ap.parse_args(argv)


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries print there too (RCCL writes a five-line version banner with
    printf when rank 0 creates its first communicator), so file descriptor 1 is pointed at stderr for the life of the process
    and the returned function writes to the real stdout."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(real, (line + "\n").encode())
    return emit


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` (N > 1) started without torch.distributed.run: start it, as a fresh child process, and
    relay.  Nothing here may touch a GPU (a process that has initialised HIP must not be replaced, and the ranks must find
    the devices untouched): no torch, no ecsimd_amd, no HIP library is imported by this process.
    ECSIMD_BENCH_LAUNCHER (tests): a replacement for the `python -m torch.distributed.run ...` prefix of the command line."""
    import shlex
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    override = os.environ.get("ECSIMD_BENCH_LAUNCHER")
    prefix = shlex.split(override) if override else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                                     "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    child = subprocess.run(prefix + [os.path.abspath(__file__)] + list(argv), stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in child.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln                                   # rank 0's one line (the last one, should a wrapper echo)
        elif ln.strip():
            print(ln, file=sys.stderr)
    if os.environ.get("ECSIMD_BENCH_LAUNCHER_REPORT"):  # tests: what this process has loaded by now
        maps = open("/proc/self/maps").read()
        print("launcher: hip_loaded=%s torch_imported=%s" % (("libamdhip64" in maps) or ("libecsimd_hip" in maps), "torch" in sys.modules), file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if child.returncode == 0 and line is None:
        print("launcher: the ranks exited 0 without printing a result line", file=sys.stderr)
        return 1
    return child.returncode


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.multi == "group":
        if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("--multi group is one process driving every GPU: start it without torch.distributed.run")
        return main_group(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args, sys.argv[1:])
    emit = claim_stdout()
    import numpy as np
    import torch
    import torch.distributed as dist
    from ecsimd_amd import Engine, BASE_MGRY, OUT_JACOBIAN, OUT_AFFINE, ALG_WINDOWED, ALG_WINDOWED_SIGNED, ALG_WINDOWED_BIG, ALG_CONSTANT_TIME, REF_SQUARE_COMPAT, LADDER_RADIX32
    from ecsimd_amd.curves import curve_id
    from ecsimd_amd.shard import ShardedRunner, plan

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world                   # under torch.distributed.run the launcher's world size is the number of GPUs
    # ECSIMD_BENCH_REHEARSE_ONE_GPU=1: N ranks SHARE cuda:0 and talk over gloo (RCCL refuses two ranks on one device) -- the
    # control flow of an N > 1 run (shard plan, per-rank streams, gather, sample check of the other ranks' shards, the device-group
    # leg, the barriers) executed on a one-GPU box.  The line says "rehearsal" and is no scaling result.
    rehearse = os.environ.get("ECSIMD_BENCH_REHEARSE_ONE_GPU") == "1" and world > 1
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    # ECSIMD_BENCH_FORCE_DIST=1 runs the process-group + gather path even with one rank (a single-GPU
    # rehearsal of the N > 1 code: RCCL init, side stream, dist.gather, barrier).
    force_dist = os.environ.get("ECSIMD_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    distributed = world > 1 or force_dist
    host_group = None
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if world > 1 and not args.no_group_check:
            # while rank 0's child process drives all N GPUs through the C ABI's device group (group_leg), the other ranks must
            # wait on the HOST: an RCCL barrier would keep a spinning kernel on every GPU the child is measuring
            host_group = dist.new_group(backend="gloo")

    import ecsimd_amd
    if not os.path.exists(ecsimd_amd.lib_path()) and local_rank == 0:
        import __graft_entry__
        __graft_entry__.build()         # built artefacts normally travel with the snapshot
    if world > 1:
        dist.barrier()
    if args.curve not in ("p256", "secp256k1") and args.workload == "ladder-x":
        raise SystemExit("ladder-x (the ladder without Z) exists for p256 / secp256k1; a curve registered at run time has every other workload")
    curve = curve_id(args.curve)        # 0 / 1, or a run-time registration (host arithmetic only)
    eng = Engine(dev_index)             # raises if the HIP library / a gfx950 device is missing: no fallback
    units = (1 << args.global_log2_batch) if args.scaling == "strong" else (1 << args.log2_batch)
    first, n, total_units, rows = plan(args.scaling, units, rank, world)       # this rank's slice of the global synthetic streams
    if n == 0:
        raise SystemExit("the batch is smaller than the number of ranks")

    # ---- inputs, resident in HBM before anything is timed
    k = eng.fill_random(n, SEED, 1, first_index=first)                       # scalars: uniform 256-bit
    s = eng.fill_random(n, SEED, 2, first_index=first)                       # point seeds: P_i = s_i * G
    # the base points come from the ladder kernel itself, at this step's own launch size: every k_scalar_mult launch of the
    # run -- this one, the warm-up, the timed steps -- is the same launch, so rocprofv3's per-kernel average is the step's time
    bx, by = eng.scalar_mult_base(curve, s, flags=OUT_AFFINE)                      # affine classical (x, y)
    P = eng.from_affine(curve, bx, by)                                        # Montgomery form, Z = mgry(1)
    xm, ym = P[0], P[1]
    del s, P
    runner = ShardedRunner((3, rows, 4), torch.int64, eng.tdev, world, rank, always_gather=force_dist, via_host=rehearse)
    view = (lambda o: [o[0][:n], o[1][:n], o[2][:n]]) if rows != n else (lambda o: [o[0], o[1], o[2]])

    if args.workload in LADDER_WORKLOADS:
        flags = BASE_MGRY | OUT_JACOBIAN | (REF_SQUARE_COMPAT if args.workload == "ladder-ref-compat" else LADDER_RADIX32 if args.workload == "ladder-radix32" else 0)

        def compute(o):
            eng.scalar_mult(curve, k, xm, ym, flags=flags, out=view(o))
    elif args.workload == "ladder-x":
        def compute(o):                                     # affine x; o[1], o[2] are unused
            eng.scalar_mult(curve, k, bx, by, flags=OUT_AFFINE, out=[view(o)[0], None, None])
        compute([eng.empty(rows) for _ in range(3)])        # sizes the context workspace
    elif args.workload in ("windowed", "windowed-ct"):
        vflags = OUT_AFFINE | ALG_WINDOWED | (ALG_CONSTANT_TIME if args.workload == "windowed-ct" else 0)

        def compute(o):                                     # affine (x, y); o[2] is unused
            eng.scalar_mult(curve, k, bx, by, flags=vflags, out=view(o))
        compute([eng.empty(rows) for _ in range(3)])        # sizes the context workspace (1 408 B per element)
    else:
        alg = {"fixed-base": ALG_WINDOWED, "fixed-base-ct": ALG_WINDOWED | ALG_CONSTANT_TIME, "fixed-base-signed": ALG_WINDOWED_SIGNED,
               "fixed-base-big": ALG_WINDOWED_BIG}[args.workload]

        def compute(o):                                     # affine (x, y); o[2] is unused
            eng.scalar_mult_base(curve, k, flags=OUT_AFFINE | alg, out=view(o))
        eng.scalar_mult_base(curve, k[:1024].contiguous(), flags=OUT_AFFINE | alg)   # builds the table and the workspace
        compute([eng.empty(rows) for _ in range(3)])

    # ECSIMD_BENCH_STEP_MARKER=1 (tools/profile_traffic.sh): a one-element fill_random launch in front of every timed step, so
    # that a per-dispatch counter listing can be cut into steps (tools/summarize_traffic.py).  Off in every measured run.
    mark_buf = eng.empty(1) if os.environ.get("ECSIMD_BENCH_STEP_MARKER") == "1" else None
    marker = (lambda: eng.fill_random(1, SEED, 99, out=mark_buf)) if mark_buf is not None else (lambda: None)

    def timed(steps, record_events):
        """`steps` steps between two fences; wall time = max over ranks."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] if record_events else None
        runner.fence()
        t0 = time.perf_counter()
        for i in range(steps):
            if evs:
                runner.step(compute, before=(lambda i=i: (marker(), evs[i][0].record())), after=evs[i][1].record)
            else:
                runner.step(compute)
        runner.fence()
        elapsed = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([elapsed], dtype=torch.float64, device=("cpu" if rehearse else eng.tdev))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, ([a.elapsed_time(b) for a, b in evs] if evs else None)      # events: same stream as the launches

    for _ in range(args.warmup):
        runner.step(compute)
    elapsed, kernel_ms = timed(args.steps, True)                     # THE measurement: gathers (N > 1) inside
    gather_ms = runner.gather_ms(args.steps) if distributed else []
    compute_only = None
    if distributed:                                                   # the same steps without the gather
        runner.gather = False
        compute_only, _ = timed(args.steps, False)
        runner.gather = True

    total = float(total_units) * args.steps
    result = base_line(args, world, total_units, n, total / elapsed, elapsed)
    result["config"]["rccl"] = ({"ranks": dist.get_world_size(), "version": ".".join(str(v) for v in torch.cuda.nccl.version()),
                                  "via": "torch.distributed backend nccl, one process per GPU"} if (distributed and not rehearse) else {"ranks": 0})
    if rehearse:
        result["rehearsal"] = (f"{world} ranks sharing ONE GPU, gathers staged through pinned host memory over gloo (ECSIMD_BENCH_REHEARSE_ONE_GPU=1): "
                               "the N > 1 control flow executed on a one-GPU box; n_gpus counts ranks here and the value is NOT a scaling result")
    if distributed:
        result["config"]["gather"] = {
            "what": ("one gloo gather of every rank's result shard, staged through pinned host memory (rehearsal)" if rehearse else
                     "one RCCL gather of every rank's result shard to rank 0 per step, on a side stream, into one pre-sized receive buffer"),
            "bytes_per_rank_per_step": 3 * rows * 32, "ms_avg_on_the_side_stream_rank0": (float(np.mean(gather_ms)) if gather_ms else None),
            "ms_per_step_with_gather": 1e3 * elapsed / args.steps, "ms_per_step_compute_only": 1e3 * compute_only / args.steps,
            "value_compute_only": total / compute_only}

    failures = []
    if rank == 0:
        result["roofline"] = roofline_object(args, eng, n, float(np.mean(kernel_ms)))
        if world == 1 and not args.no_cpu_baseline:
            try:
                attach_cpu_baseline(args, result, eng, curve, k, bx, by, [t[:n] for t in runner.last_result()], failures)
                if args.workload == "ladder":
                    # a side figure like the CPU leg: whatever goes wrong in it (a checker that cannot load, a HIP error, out of memory for its three
                    # extra outputs) is recorded in its object and costs the run neither its line nor -- unless lanes DIFFER -- its exit code
                    try:
                        result["ref_compat"] = ref_compat_leg(args, result, eng, curve, k, xm, ym, n, result["roofline"]["peak"], failures)
                    except Exception as exc:                       # noqa: BLE001 (recorded, not swallowed)
                        result["ref_compat"] = {"value": None, "unit": "scalar_mults/s", "lanes_compared": None, "lanes_differing": None, "error": repr(exc)[:300]}
            finally:                                               # json.dumps must never see the numpy arrays of the sample
                if isinstance(result.get("cpu_baseline"), dict):
                    result["cpu_baseline"].pop("_sample", None)
        if (force_dist or rehearse) and not torch.equal(runner.gathered[0].to(eng.tdev), runner.last_result()):
            failures.append("the gathered shard differs from the computed one")
        if distributed and args.workload in LADDER_WORKLOADS:
            # what arrived from the other ranks: rank 0 regenerates the first lanes of every rank's slice of the synthetic streams
            # (seed, global index), runs them through its own ladder and compares with that rank's block of the receive buffer
            m, bad_ranks = 1024, []
            for r in (range(1, world) if world > 1 else [0]):         # one rank (the rehearsal): its own block, so that this code has run
                f_r, n_r, _, _ = plan(args.scaling, units, r, world)
                mm = min(m, n_r)
                kr = eng.fill_random(mm, SEED, 1, first_index=f_r); sr = eng.fill_random(mm, SEED, 2, first_index=f_r)
                rx, ry = eng.scalar_mult_base(curve, sr, flags=OUT_AFFINE)
                Pr = eng.from_affine(curve, rx, ry)
                exp = eng.scalar_mult(curve, kr, Pr[0], Pr[1], flags=flags)
                got = runner.gathered[r]
                if not all(torch.equal(got[j][:mm].to(exp[j].device), exp[j]) for j in range(3)):
                    bad_ranks.append(r)
            result["config"]["gather"]["sample_check"] = {"lanes_per_rank": m, "ranks_checked": max(1, world - 1), "ranks_differing": bad_ranks}
            if bad_ranks:
                failures.append(f"the shards gathered from ranks {bad_ranks} differ from rank 0's own ladder on the same inputs")
        if world > 1 and not args.no_group_check:
            result["multi_group"] = group_leg(args, world)          # the other ranks wait at the barrier below
        if failures:
            result["parity_failures"] = failures
        emit(json.dumps(result))
    if distributed:
        if host_group is not None:
            dist.barrier(group=host_group)
        dist.barrier()
        dist.destroy_process_group()
    return EXIT_PARITY if failures else 0


def base_line(args, world, total_units, n, value, elapsed):
    """The contract fields of the one JSON line."""
    sizes = f"2^{args.global_log2_batch} per step over {world} GPU(s) (strong scaling, BASELINE configs[3])" if args.scaling == "strong" \
        else f"2^{args.log2_batch} per GPU per step (weak scaling)"
    names = {
        "ladder": f"scalar_mult_{args.curve} variable-base co-Z ladder (reference algorithm), batch {sizes}, Jacobian Montgomery out",
        "ladder-ref-compat": f"scalar_mult_{args.curve} variable-base co-Z ladder with ECSIMD_HIP_REF_SQUARE_COMPAT (the reference's square() as written), batch {sizes}, Jacobian Montgomery out",
        "ladder-radix32": f"scalar_mult_{args.curve} variable-base co-Z ladder with ECSIMD_HIP_LADDER_RADIX32 (the 254 iterations on 8 x 32-bit canonical words), batch {sizes}, Jacobian Montgomery out",
        "ladder-x": f"scalar_mult_{args.curve} variable-base, x coordinate only: " + ("the co-Z ladder without Z (8M + 6S per bit), x from the curve equation + simultaneous inversion"
                    if args.curve == "p256" else "the co-Z ladder + x-only simultaneous inversion") + f", batch {sizes}, affine x out",
        "windowed": f"scalar_mult_{args.curve} variable-base, per-element window tables (8 multiples of P) + signed 4-bit windows + simultaneous "
                    f"inversion{' + GLV split k = k1 + k2*lambda, the table over one Z and the loop on the isomorphic curve' if args.curve == 'secp256k1' else ''}"
                    + ("" if args.curve in ("p256", "secp256k1") else "; this curve is registered at RUN time: generic kernels (k_gvarwin.hip), the dense 9-limb prime in SGPRs, the eight odd multiples over one Z "
                       "and the loop on the isomorphic curve in modified Jacobian coordinates (a general a)") + f", batch {sizes}, affine out",
        "windowed-ct": f"scalar_mult_{args.curve} variable-base, per-element window tables (8 odd multiples of P) + signed 4-bit windows (odd digits), ALG_CONSTANT_TIME: "
                       f"all 8 entries of the lane's table read in every window, kept under lane masks"
                       + (" (secp256k1: GLV split k = k1 + k2*lambda on the complete addition law of a = 0 curves)" if args.curve == "secp256k1" else "")
                       + ("" if args.curve in ("p256", "secp256k1") else "; this curve is registered at RUN time: generic kernels (k_gvarwin.hip), the table over one Z, the loop on the isomorphic curve in modified Jacobian coordinates")
                       + f"; + simultaneous inversion, batch {sizes}, affine out",
    }
    fixed = {"fixed-base": "4-bit window table in LDS (odd digits, 32 KiB)",
             "fixed-base-ct": "ALG_CONSTANT_TIME: odd-digit comb in LDS, every entry of a window read and one kept under lane masks (52 five-bit windows x 16 entries, 53 KB, three 256-thread workgroups per CU)",
             "fixed-base-signed": "signed 7-bit window table in LDS (odd digits, 148 KiB)",
             "fixed-base-big": "20-bit window table of odd multiples (436 MB) in device memory"}
    if args.curve not in ("p256", "secp256k1"):
        fixed = {k: v + "; this curve is registered at RUN time: generic kernels (k_gcomb.hip), the dense 9-limb prime in SGPRs, the table built from the reference's ladder on first use" for k, v in fixed.items()}
    return {
        "metric": "P-256 scalar mults/sec (batched)" if args.curve == "p256" else f"{args.curve} scalar mults/sec (batched)",
        "value": value, "unit": "scalar_mults/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": names.get(args.workload) or (f"scalar_mult_{args.curve} fixed-base (G), {sizes}, random scalars, "
                                                             + fixed[args.workload] + " + simultaneous inversion, affine out"),
                   "element": ("256-bit integers, one per lane in VGPRs: canonical 8 x u32 words (= 4 x u64 limbs) at every kernel boundary; the ladder's 254 iterations run on nine "
                               "signed 29-bit digits in 32-bit words (Montgomery radix 2^261, v_mad_i64_i32 into carry-free 64-bit columns, lazy carries: fe29.cuh) -- "
                               "REF_SQUARE_COMPAT and LADDER_RADIX32 keep the 8 x u32 v_mad_u64_u32 carry chains"
                               + ("" if args.curve in ("p256", "secp256k1") else "; this curve is registered at RUN time (the reference's curve_group<Curve> for any Curve): generic kernels, "
                                  "the dense 9-limb prime in SGPRs, 81 multiply-adds per reduction where P-256's sparse form has 36")) if args.workload == "ladder" else
                              "256-bit integers: 8 x u32 words (= 4 x u64 limbs) in VGPRs, v_mad_u64_u32 carry chains",
                   "global_batch": total_units, "per_gpu_batch": n,
                   "parallelism": (f"group{world}" + ("+rccl_gather" if world > 1 else "") if args.multi == "group" else f"shard{world}" + ("+rccl_gather" if world > 1 else ""))},
    }


def roofline_object(args, eng, n, avg_ms):
    """Roofline of the dominant kernel, measured live: achieved = lanes of ONE launch x algorithmic mad32 per lane / the
    launch's device time (HIP events on the launch stream); peak = the dependency-free v_mad_u64_u32 stream on the same GPU."""
    mads, ms = eng.peak_mad32(8192, reps=5)
    peak = mads / (ms * 1e-3) / 1e12
    if args.workload in LADDER_WORKLOADS:
        mad32_unit, bytes_unit = MAD32_PER_SCALAR_MULT, ALGO_BYTES_PER_SCALAR_MULT
        kname = "k_scalar_mult" if args.curve in ("p256", "secp256k1") else "k_gc_scalar_mult"
    elif args.workload == "ladder-x":
        # P-256: TRPLU (6 + 7), 254 x (8M + 6S), the recovery (12) and the inversion walk (7 + 267 / share); secp256k1: the full ladder + 5 + 267 / share
        share = min(128, max(1, n >> 17))
        fm = ((13 + 254 * 14 + 12) if args.curve == "p256" else 4088 + 5) + 3 + (267 if args.curve == "p256" else 270) / share
        mad32_unit, bytes_unit = int(fm * 136), 128
        kname = ("k_scalar_mult_x + k_inverse_batched" if args.curve == "p256" else "k_scalar_mult + k_to_affine_batched")
    elif args.workload in ("windowed", "windowed-ct"):
        # what THIS algorithm needs per scalar.  P-256: table {1,3,..,15}P = DBLU + 7 co-Z additions
        # (6 + 7 x 7), made affine by one inversion per lane and a walk back (7 x 5 + 3 + 267/32 at 2^22), 63 windows x (3 doublings
        # + one fused double-add of 13M + 5S), the final inversion walk; a doubling is 4M + 4S (P-256) / 3M + 4S
        # (secp256k1), a mixed addition 8M + 3S; 96 B in, 64 B out.
        dbl = 8 if args.curve == "p256" else 7
        inv = 267 if args.curve == "p256" else 270                  # addition-chain inversion (point.cuh fe_inverse)
        share = min(128, max(1, min(n, 1 << 22) >> 17))            # elements per shared inversion (k_affine.inc; the windowed path works in chunks of 2^22)
        chain = 7 * 5 + 3 + inv / share                            # the table's entries made affine: ONE inversion per lane (the chain's last Z, shared by `share` lanes), 5 products per entry on the walk back
        fm = (6 + 7 * 7) + chain + 63 * (3 * dbl + 18) + (7 + inv / share)
        if args.curve == "secp256k1" and args.workload == "windowed-ct":  # GLV split on the complete addition law: 32 windows x (4 doublings of 6M + 2S, two mixed additions of 11M, beta),
            fm = (6 + 6 * 7) + chain + 32 * (4 * 8 + 2 * 11 + 1) + (2 * 11 + 1) + 3 + (7 + inv / share)   # table {1..8}P as a chain; the top window's two additions; (X Z, Y Z^2, Z)
        if args.curve == "secp256k1" and args.workload == "windowed":     # GLV split: 32 windows x (4 doublings + 2 mixed additions + beta) + the top window's two additions
            # table {1..8}P over ONE Z (k_varwin_table_iso: a doubling, P over its Z, six co-Z additions, the backward walk of 5 products per entry -- no inversion), the
            # loop on the isomorphic curve, one product by the common Z at the end
            fm = (dbl + 4 + 6 * 7 + 7 * 5) + 32 * (4 * dbl + 2 * 11 + 1) + (2 * 11 + 1) + 1 + (7 + inv / share)
        registered = args.curve not in ("p256", "secp256k1")
        if registered:
            # k_gvarwin.hip: the table over one Z (a doubling of 7, P over its Z 4, seven co-Z additions, 2 + 7 x 5 on the walk back, 4 entering products),
            # a' = a Zg^4 (4), 63 windows x (W = a' Z^4: 3; doublings of 8, 8, 7 in modified Jacobian coordinates; the fused double-add 18), Z' Zg and the three
            # leaving products (4), then the shared inversion (the field's generic division steps, priced as P-256's 267)
            fm = (7 + 4 + 7 * 7 + 2 + 7 * 5 + 4) + 4 + 63 * (3 + 8 + 8 + 7 + 18) + 4 + (7 + 267 / share)
        mad32_unit, bytes_unit = int(fm * 136), 160
        kname = ("k_gvw_mult<true>" if args.workload == "windowed-ct" else "k_gvw_mult<false>") + " + k_gvw_table + k_gc_to_affine_batched" if registered else ("k_varwin_mult_glv_ct + k_varwin_multiples_chain" if (args.workload == "windowed-ct" and args.curve == "secp256k1") else
                 "k_varwin_mult_odd<true> + k_varwin_odd_multiples" if args.workload == "windowed-ct" else "k_varwin_mult_odd<false> + k_varwin_odd_multiples" if args.curve == "p256"
                 else "k_varwin_mult_glv + k_varwin_table_iso") + ("" if (args.workload == "windowed" and args.curve == "secp256k1") else " + k_varwin_invert_last + k_varwin_chain_to_table") + " + k_to_affine_batched"
    else:
        # what THIS algorithm needs per scalar: 63 / 36 / 12 mixed additions x 11 field mults,
        # 7 mults of the simultaneous-inversion walk and 267/m (secp256k1: 270/m) of the inversion m = min(128, n / 2^17) points share; 32 B in, 64 B out.
        adds = {"fixed-base": 63, "fixed-base-ct": 51, "fixed-base-signed": 36, "fixed-base-big": 12}[args.workload]      # (the constant-time comb: 52 five-bit windows)      # odd digits everywhere: the first entry starts the sum (round 3)
        share = min(128, max(1, n >> 17))
        mad32_unit, bytes_unit = int((adds * 11 + 7 + (267 if args.curve == "p256" else 270) / share) * 136), 96
        kname = {"fixed-base": "k_base_windowed<false>", "fixed-base-ct": "k_base_windowed_s<5, true, 256>",
                 "fixed-base-signed": "k_base_windowed_s<7, false>", "fixed-base-big": "k_base_windowed_g"}[args.workload] + " + k_to_affine_batched"
        if args.curve not in ("p256", "secp256k1"):
            kname = {"fixed-base": "k_gc_base_windowed<false>", "fixed-base-ct": "k_gc_base_windowed_s<5, true, 256>", "fixed-base-signed": "k_gc_base_windowed_s<7, false, 1024>",
                     "fixed-base-big": "k_gc_base_windowed_s<20, false, 256>"}[args.workload] + " + k_gc_to_affine_batched"
    achieved = n / (avg_ms * 1e-3) * mad32_unit / 1e12
    traffic, traffic_src = committed_traffic(args, n)
    return {
        **committed_pipe(args, n),
        "bound": "valu", "kernel": kname, "achieved": achieved, "peak": peak, "unit": "Tmad32/s", "frac": achieved / peak,
        "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": avg_ms, "algorithmic_mad32_per_unit": mad32_unit,
        "hbm": {"achieved": n / (avg_ms * 1e-3) * bytes_unit / 1e9, "peak": 8000.0, "unit": "GB/s",
                "algorithmic_bytes_per_unit": bytes_unit},
        "peak_source": "ecsimd_hip_peak_mad32: dependency-free v_mad_u64_u32 stream, 8 waves/SIMD, same GPU, same run",
        # MI355X_MICROARCH.md lists no integer-multiply peak; from its chip parameters a full-rate wave64 VALU instruction
        # peaks at 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz
        "peak_a_priori": A_PRIORI_PEAK_TMAD32, "frac_of_a_priori_peak": achieved / A_PRIORI_PEAK_TMAD32,
    }


def attach_cpu_baseline(args, result, eng, curve, k, bx, by, out, failures):
    """The baseline is a reported side figure: a checker that cannot LOAD (e.g. the prebuilt reference on a host without
    AVX2) costs the run neither its line nor its exit code.  A parity DISAGREEMENT does: it is in the object and the
    process exits EXIT_PARITY after printing."""
    try:
        if args.workload in ("windowed", "windowed-ct"):
            result["cpu_baseline"] = cpu_baseline_affine(eng, curve, k, out, args.cpu_seconds, failures, base=(bx, by), name=args.curve)
        elif args.workload == "ladder-x":
            result["cpu_baseline"] = cpu_baseline_affine(eng, curve, k, out, args.cpu_seconds, failures, base=(bx, by), x_only=True)
        elif args.workload in LADDER_WORKLOADS:
            result["cpu_baseline"] = cpu_baseline(eng, curve, k, bx, by, out, args.cpu_seconds, failures, compat=(args.workload == "ladder-ref-compat"), name=args.curve)
            comp = competitor_openssl(eng, curve, k, bx, by, out, failures) if args.curve in ("p256", "secp256k1") else None      # oracle/ossl_check.c knows the two built-in curves
            if comp is not None:
                result["cpu_baseline"]["competitor_openssl"] = comp
        else:
            result["cpu_baseline"] = cpu_baseline_affine(eng, curve, k, out, args.cpu_seconds, failures, name=args.curve)
    except (CheckerUnavailable, OSError) as exc:
        result["cpu_baseline"] = {"value": None, "unit": "scalar_mults/s", "cores": usable_cores(), "kind": "unavailable",
                                  "sample": "the CPU checkers could not be loaded here", "error": repr(exc)[:300]}


def ref_compat_leg(args, result, eng, curve, k, xm, ym, n, peak, failures, steps=3):
    """the default line also TIMES the mode that is bit-identical to the reference on every lane --
    ECSIMD_HIP_REF_SQUARE_COMPAT (the reference's square() as written, mul.h:160-212, dropped carry included) -- at the same batch:
    `steps` launches between HIP events on the launch stream, and the lanes the CPU leg already pushed through the compiled
    reference compared bit for bit (X, Y, Z): not one may differ, else EXIT_PARITY.  With only the C restatement available the
    bug-for-bug oracle stands in on a smaller sample."""
    import numpy as np
    import torch
    from ecsimd_amd import BASE_MGRY, OUT_JACOBIAN, REF_SQUARE_COMPAT
    from oracle import loader
    out = [eng.empty(n) for _ in range(3)]
    fl = BASE_MGRY | OUT_JACOBIAN | REF_SQUARE_COMPAT
    eng.scalar_mult(curve, k, xm, ym, flags=fl, out=out)                      # warm-up
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record(); eng.scalar_mult(curve, k, xm, ym, flags=fl, out=out); b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    sample = (result.get("cpu_baseline") or {}).get("_sample")
    compared = differing = None
    witness = None
    if sample is not None:
        m, kn, xn, yn, ref, kind, _, name = sample
        if kind == "reference":
            witness = "the compiled reference (oracle/_ref), the lanes of cpu_baseline"
        else:                                                                 # the exact port is no witness for this mode: the bug-for-bug restatement on a bounded sample
            m = min(m, 4096)
            fa = loader.Oracle(faithful=True)
            ref = fa.scalar_mult(checker_curve(fa, name), kn[:m], xn[:m], yn[:m], threads=usable_cores())
            witness = "the bug-for-bug C restatement (oracle/ecsimd_oracle.c, faithful mode)"
        got = [eng.to_numpy(t[:m]) for t in out]
        differing = int(np.count_nonzero((got[0] != ref[0][:m]).any(axis=1) | (got[1] != ref[1][:m]).any(axis=1) | (got[2] != ref[2][:m]).any(axis=1)))
        compared = int(m)
        if differing:
            failures.append(f"ref_compat: {differing} of {m} lanes of the reference-compatible ladder differ from the reference")
    achieved = n / (ms * 1e-3) * MAD32_PER_SCALAR_MULT / 1e12
    return {"what": "the same batch through ECSIMD_HIP_REF_SQUARE_COMPAT (the reference's square() as written: bit-identical to the reference on EVERY lane; "
                    "radix-2^32 loop, the dropped carry depends on the 32-bit Montgomery digits)",
            "value": n / (ms * 1e-3), "unit": "scalar_mults/s", "kernel_ms": ms, "steps": steps, "achieved": achieved, "frac": achieved / peak,
            "frac_of_a_priori_peak": achieved / A_PRIORI_PEAK_TMAD32, "lanes_compared": compared, "lanes_differing": differing, "compared_with": witness}


def group_leg(args, world, timeout_s=120):
    """N > 1, one process per GPU: after the timed loops rank 0 runs `bench.py --multi group` over the same N devices in a
    CHILD process (the ranks idle at a barrier meanwhile) -- the C ABI's device group, whose RCCL branch (ncclCommInitAll
    communicators, grouped ncclSend / ncclRecv) only a multi-GPU node can execute.  A side figure: a failure here is
    recorded, it does not cost the line or the exit code."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--multi", "group", "--gpus", str(world), "--steps", str(min(args.steps, 5)), "--warmup", "1",
           "--scaling", args.scaling, "--global-log2-batch", str(args.global_log2_batch), "--log2-batch", str(args.log2_batch),
           "--curve", args.curve, "--workload", args.workload, "--no-cpu-baseline"]
    # the child is NOT a rank: it must not inherit torch.distributed.run's environment
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
                        "MASTER_PORT", "ECSIMD_BENCH_FORCE_DIST") and not k.startswith("TORCHELASTIC_")}
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if out.returncode != 0 or not lines:
            return {"ok": False, "returncode": out.returncode, "stderr_tail": out.stderr[-600:]}
        d = json.loads(lines[-1])
        return {"ok": True, "value": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "rccl": d["config"].get("rccl"),
                "gather": d["config"].get("gather"), "parity": d["config"].get("group_parity")}
    except (subprocess.SubprocessError, ValueError, KeyError) as exc:
        return {"ok": False, "error": repr(exc)[:300]}


def main_group(args):
    """--multi group: ONE process, ecsimd_hip_group_* (include/ecsimd_hip.h "device groups"): one context per device, member m
    owns shard_range(units, m, N) resident in ITS memory, every step is one ecsimd_hip_group_scalar_mult -- N ladders and one
    exchange into device 0's arrays (RCCL when N > 1: grouped ncclSend / ncclRecv on ncclCommInitAll communicators) -- with
    double-buffered outputs, no host synchronisation inside the timed loop, ecsimd_hip_group_sync at its end."""
    emit = claim_stdout()
    import numpy as np
    import torch
    from ecsimd_amd import Engine, DeviceGroup, CURVES, BASE_MGRY, OUT_JACOBIAN, OUT_AFFINE, GROUP_NO_GATHER, REF_SQUARE_COMPAT
    from ecsimd_amd.shard import plan
    import ecsimd_amd
    if args.workload not in ("ladder", "ladder-ref-compat", "ladder-x") or (args.workload == "ladder-x" and args.curve not in ("p256", "secp256k1")):
        raise SystemExit("--multi group runs the ladder workloads (the group entry point is ecsimd_hip_group_scalar_mult)")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    N = args.gpus
    rehearse = os.environ.get("ECSIMD_BENCH_REHEARSE_ONE_GPU") == "1" and N > 1      # N members on ONE device: the group's shared-device gather
    devices = [0] * N if rehearse else list(range(N))
    if torch.cuda.device_count() <= max(devices):
        raise SystemExit(f"--multi group --gpus {N}: only {torch.cuda.device_count()} device(s) visible")
    if not os.path.exists(ecsimd_amd.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    from ecsimd_amd.curves import curve_id
    curve = curve_id(args.curve)
    units = (1 << args.global_log2_batch) if args.scaling == "strong" else (1 << args.log2_batch)
    spans = [plan(args.scaling, units, m, N) for m in range(N)]
    total_units = spans[0][2]
    if any(sp[1] == 0 for sp in spans):
        raise SystemExit("the batch is smaller than the number of GPUs")
    grp = DeviceGroup(devices)
    engs, ks, xs, ys, bxs, bys = [], [], [], [], [], []
    for m in range(N):                                   # inputs generated where they will be used (seed, global index): nothing moves
        with torch.cuda.device(devices[m]):
            eng = Engine(devices[m])
            first, cnt = spans[m][0], spans[m][1]
            k = eng.fill_random(cnt, SEED, 1, first_index=first)
            sd = eng.fill_random(cnt, SEED, 2, first_index=first)
            bx, by = eng.scalar_mult_base(curve, sd, flags=OUT_AFFINE)
            if args.workload == "ladder-x":
                xm, ym = bx, by
            else:
                P = eng.from_affine(curve, bx, by); xm, ym = P[0], P[1]
            torch.cuda.synchronize(devices[m])
            engs.append(eng); ks.append(k); xs.append(xm); ys.append(ym); bxs.append(bx); bys.append(by)
    x_only = args.workload == "ladder-x"
    flags = (OUT_AFFINE if x_only else (BASE_MGRY | OUT_JACOBIAN)) | (REF_SQUARE_COMPAT if args.workload == "ladder-ref-compat" else 0)
    outs = [grp.alloc_outputs(total_units, flags, x_only) for _ in range(2)]
    torch.cuda.synchronize(0)

    def loop(steps, fl):
        grp.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            grp.enqueue(curve, ks, xs, ys, outs[i & 1], total_units, fl)
        g_ms = grp.sync()
        return time.perf_counter() - t0, g_ms

    loop(max(1, args.warmup), flags)                      # sizes staging and workspaces, creates RCCL's channels
    elapsed, gather_ms = loop(args.steps, flags)
    member_ms = [grp.member_ms(m) for m in range(N)]
    compute_only, _ = loop(args.steps, flags | GROUP_NO_GATHER)
    loop(1, flags)                                        # the arrays hold a gathered result again
    total = float(total_units) * args.steps
    n0 = spans[0][1]
    result = base_line(args, N, total_units, n0, total / elapsed, elapsed)
    result["config"]["rccl"] = {"ranks": N if grp.uses_rccl else 0, "version": grp.rccl_version or None,
                                "via": "ecsimd_hip_group_* (C ABI): one process, ncclCommInitAll, grouped ncclSend / ncclRecv" if grp.uses_rccl
                                       else ("ecsimd_hip_group_* (C ABI): members share one device, device-to-device copies (rehearsal)" if N > 1
                                             else "ecsimd_hip_group_* (C ABI): one member, no exchange")}
    if rehearse:
        result["rehearsal"] = f"{N} group members on ONE GPU (ECSIMD_BENCH_REHEARSE_ONE_GPU=1): not a scaling result"
    result["config"]["gather"] = {
        "what": "every other member's result shard into device 0's arrays by one grouped ncclSend / ncclRecv exchange per step, on device 0's gather stream",
        "bytes_per_member_per_step": (1 if x_only else 3) * spans[0][3] * 32, "ms_last_on_the_gather_stream": (gather_ms if gather_ms >= 0 else None),
        "ms_per_step_with_gather": 1e3 * elapsed / args.steps, "ms_per_step_compute_only": 1e3 * compute_only / args.steps,
        "value_compute_only": total / compute_only}
    failures = []
    # every member's shard of the gathered result against that member's own single-device ladder (bit for bit, on the devices)
    last = outs[0]
    ok = True
    for m in range(N):
        with torch.cuda.device(devices[m]):
            first, cnt = spans[m][0], spans[m][1]
            ref = engs[m].scalar_mult(curve, ks[m], xs[m], ys[m], flags=flags, x_only=x_only)
            torch.cuda.synchronize(devices[m])
            for a, b in zip(last, ref):
                if b is not None:
                    ok = ok and bool(torch.equal(a[first:first + cnt].to(b.device), b))
    result["config"]["group_parity"] = {"gathered_equals_each_members_own_ladder": ok, "members": N}
    if not ok:
        failures.append("multi group: the gathered result differs from a member's own ladder")
    result["roofline"] = roofline_object(args, engs[0], n0, float(max(member_ms)))
    result["roofline"]["kernel_ms_per_member"] = member_ms
    if N == 1 and not args.no_cpu_baseline:
        attach_cpu_baseline(args, result, engs[0], curve, ks[0], bxs[0], bys[0], [t[:n0] for t in last] + [None] * (3 - len(last)), failures)
        if isinstance(result.get("cpu_baseline"), dict):
            result["cpu_baseline"].pop("_sample", None)
    if failures:
        result["parity_failures"] = failures
    emit(json.dumps(result))
    grp.close()
    return EXIT_PARITY if failures else 0


def committed_pipe(args, n):
    """What the VALU pipe does during the dominant kernel, from the committed counter passes and the shipped ISA (tools/pipe_model.py ->
    profiles/pmc_pipe.json): `frac` above is an ALGORITHMIC rate over the multiply peak (136 mad32 per field multiplication); these four say
    how many instructions a scalar multiplication really issues, how many of them multiply, that a SIMD issues one every ~4 cycles all the time, and how much
    of the elapsed time the instruction mix alone accounts for at the measured per-instruction issue costs.  Not measured in this run (like `traffic`)."""
    path = os.path.join(ROOT, "profiles", "pmc_pipe.json")
    key = {"ladder": "k_scalar_mult", "ladder-ref-compat": "k_scalar_mult_refsqr", "ladder-radix32": "k_scalar_mult_radix32", "fixed-base": "fixed_base",
           "windowed": "varwin", "windowed-ct": "varwin_ct"}.get(args.workload)
    try:
        kernels = json.load(open(path))["kernels"]
        rec = (kernels.get(f"{key}_{args.curve}_2^24") or kernels.get(f"{key}_{args.curve}_2^22")) if key else None      # (the window loops run in chunks of 2^22 lanes)
    except (OSError, ValueError):
        rec = None
    if not rec or "issue_bound_frac" not in rec:
        return {}
    return {"valu_instructions_per_unit": rec["valu_instructions_per_unit"], "multiply_instructions_per_unit": rec["multiply_instructions_per_unit"],
            "cycles_per_valu_instruction_per_simd": rec["cycles_per_valu_instruction_per_simd"], "issue_bound_frac": rec["issue_bound_frac"],
            "pipe_source": f"profiles/pmc_pipe.json ({rec['source']} + the ISA of build/csrc/{rec['isa_unit']}: tools/pipe_model.py; kernel {rec['kernel']}); at 2^{rec['lanes_per_launch'].bit_length() - 1} lanes per launch, not measured in this run"}


def committed_traffic(args, n):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json), scaled to this
    run's lanes per launch: the counters cannot be read from inside the run, so the figure is NOT measured here."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath):
        return None, None
    try:
        table = json.load(open(tpath))
    except ValueError:
        return None, None
    key = {"ladder": "k_scalar_mult", "ladder-ref-compat": "k_scalar_mult_refsqr", "ladder-radix32": "k_scalar_mult_radix32", "ladder-x": "k_scalar_mult_x", "windowed": "varwin", "windowed-ct": "varwin_ct",
           "fixed-base": "fixed_base", "fixed-base-ct": "fixed_base_ct", "fixed-base-signed": "fixed_base_signed", "fixed-base-big": "fixed_base_big"}[args.workload]
    for log2 in (24, 22):                                 # a pass at this run's own launch size first
        per = table.get(f"{key}_{args.curve}_2^{log2}")
        if per is not None:
            exact = n == (1 << log2)
            return per * n / float(1 << log2), (f"profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at 2^{log2} lanes per launch, every kernel of one step "
                                                 "(FETCH_SIZE x 2: profiles/r03/hbm_counter_calibration.json)" + ("" if exact else ", scaled by lanes") + "; not measured in this run")
    return None, None


def load_checkers():
    """(timed implementation, its kind).  Raises CheckerUnavailable when neither library can be had."""
    from oracle import loader
    try:
        if loader.reference_available():
            return loader.Reference(), "reference"
        if not os.path.exists(loader.Oracle.path):
            loader.build()
        return loader.Oracle(), "port"
    except OSError as exc:
        raise CheckerUnavailable(repr(exc)) from exc


def openssl_checker():
    """oracle/ossl_check.c over libcrypto, or None where the OpenSSL headers are missing."""
    import subprocess
    from oracle import loader
    if not loader.openssl_available():
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ossl"], check=False)
        if not loader.openssl_available():
            return None
    return loader.OpenSSLCheck()


def cpu_baseline_affine(eng, curve, k, gpu_out, target_s, failures, base=None, x_only=False, name=None):
    """Affine-output workloads on the CPU: the reference has ONE way to compute k*P -- scalar_mult(k, P) followed
    by to_affine() (exactly what its benchmark times, benchs/curve_group.cpp:23-35; P = G for config 3, `base` =
    the per-element points for the windowed variable-base workload).  Compared with the GPU's windowed result at
    the affine level; lanes that differ are settled by the exact oracle and by libcrypto."""
    import numpy as np
    from oracle import loader
    cores = usable_cores()
    impl, kind = load_checkers()
    builtin = name in (None, "p256", "secp256k1")
    if not builtin:                                     # a curve registered at run time: below, `curve` is the checker's id
        curve = checker_curve(impl, name)
    if base is not None:
        c = None
    elif builtin:
        c = impl.constants(curve)
    else:                                               # the generator of a registered curve: its public parameters (ecsimd_amd/curves.py)
        from ecsimd_amd.curves import NAMED
        lim = lambda v: np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
        c = {"gx": lim(NAMED[name]["gx"]), "gy": lim(NAMED[name]["gy"])}
    m0 = 256 * cores
    kn = eng.to_numpy(k[:m0])
    points = (lambda m_: (np.tile(c["gx"], (m_, 1)), np.tile(c["gy"], (m_, 1)))) if base is None else \
             (lambda m_: (eng.to_numpy(base[0][:m_]), eng.to_numpy(base[1][:m_])))
    gx, gy = points(m0)
    t = time.perf_counter(); impl.to_affine(curve, impl.scalar_mult(curve, kn, gx, gy, threads=cores)); dt = time.perf_counter() - t
    m = int(min(k.shape[0], max(m0, (target_s / dt) * m0))); m -= m % 4
    kn = eng.to_numpy(k[:m]); gx, gy = points(m)
    t = time.perf_counter(); J = impl.scalar_mult(curve, kn, gx, gy, threads=cores); ax, ay = impl.to_affine(curve, J); dt = time.perf_counter() - t
    gx_ = eng.to_numpy(gpu_out[0][:m])
    gy_ = ay.copy() if x_only else eng.to_numpy(gpu_out[1][:m])       # x only: nothing to compare y with
    bad = np.nonzero((gx_ != ax).any(axis=1) | (gy_ != ay).any(axis=1))[0]
    explained, by_ossl = True, None
    if len(bad):                                        # reference square() defect: the exact oracle must side with the GPU
        ex = loader.Oracle(faithful=False)
        exc = curve if builtin else checker_curve(ex, name)
        ea = ex.to_affine(exc, ex.scalar_mult(exc, kn[bad], gx[bad], gy[bad], threads=min(cores, len(bad))))
        explained = bool(np.array_equal(ea[0], gx_[bad]) and (x_only or np.array_equal(ea[1], gy_[bad])))
        ossl = openssl_checker() if builtin else None    # libcrypto's harness knows the two built-in curves; a registered curve's lanes are settled by textbook arithmetic
        if not builtin:
            from ecsimd_amd.curves import NAMED
            ti = lambda v: sum(int(w) << (64 * j) for j, w in enumerate(v))
            by_ossl = sum(1 for lane in bad if textbook_scalar_mult(NAMED[name], ti(kn[lane]), ti(gx[lane]), ti(gy[lane])) == (ti(gx_[lane]), ti(gy_[lane])))
        if ossl is not None:
            vx, vy, inf = ossl.scalar_mult(curve, kn[bad], gx[bad], gy[bad], threads=1)
            by_ossl = int(np.count_nonzero(~((gx_[bad] != vx).any(axis=1) | ((gy_[bad] != vy).any(axis=1) & (not x_only)) | (inf != 0))))
    if not explained:
        failures.append("cpu_baseline: a lane differs from the reference and the exact oracle does not side with the GPU")
    if by_ossl is not None and by_ossl != len(bad):
        failures.append("cpu_baseline: libcrypto (a registered curve: textbook affine arithmetic) does not confirm the GPU on a lane where it differs from the reference")
    return {"value": m / dt, "unit": "scalar_mults/s", "cores": cores, "kind": kind,
            "per_core": (m / dt) / cores, "cpu_model": cpu_model(), "flags": build_flags(kind),
            "one_thread": one_thread_rate(lambda m1: impl.to_affine(curve, impl.scalar_mult(curve, kn[:m1], gx[:m1], gy[:m1], threads=1)), (m / dt) / cores, m),
            "sample": f"first {m} scalars of the GPU batch through scalar_mult(k, {'G' if base is None else 'P'}) + to_affine (the reference's only path to affine k*P), "
                      f"{dt:.1f} s wall, {cores} threads (to_affine single-threaded)",
            "lanes_compared": int(m), "lanes_differing_from_gpu": int(len(bad)),
            "differences_all_explained_by_reference_square_defect": bool(explained),
            ("lanes_differing_confirmed_by_openssl" if builtin else "lanes_differing_confirmed_by_textbook_arithmetic"): by_ossl}


def competitor_openssl(eng, curve, k, bx, by, gpu_out, failures, seconds=3.0, check_lanes=8192):
    """Part of the cpu_baseline leg: the reference's competitor benchmark (benchs/p256_ref.cpp:55-91, OpenSSL's
    EC_POINT_mul) on the host cores, run as a child process, plus an affine-level comparison of a sample of
    the GPU's results with libcrypto -- a check that depends on neither the reference nor the restatement.
    Returns None where the OpenSSL headers were not available to build oracle/ossl_check.c."""
    import subprocess
    import numpy as np
    ossl = openssl_checker()
    if ossl is None:
        return None
    cores = usable_cores()
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ossl_bench.py"), "--curve", str(curve), "--procs", str(cores),
                              "--seconds", str(seconds)], capture_output=True, text=True, timeout=120, check=True)
        res = json.loads(out.stdout.strip().splitlines()[-1])
    except (subprocess.SubprocessError, ValueError, IndexError) as e:      # the TIMING is a reported side figure; the comparison below is not
        res = {"error": repr(e)[:200]}
    m = min(check_lanes, k.shape[0])
    ax, ay = eng.to_affine(curve, [t[:m].contiguous() for t in gpu_out])
    vx, vy, inf = ossl.scalar_mult(curve, eng.to_numpy(k[:m]), eng.to_numpy(bx[:m]), eng.to_numpy(by[:m]), threads=cores)
    diff = (eng.to_numpy(ax) != vx).any(axis=1) | (eng.to_numpy(ay) != vy).any(axis=1) | (inf != 0)
    res["lanes_compared_with_gpu"] = int(m)
    res["lanes_differing_from_gpu"] = int(np.count_nonzero(diff))
    if res["lanes_differing_from_gpu"]:
        failures.append("competitor_openssl: libcrypto's k*P differs from the GPU's")
    return res


def usable_cores():
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def build_flags(kind):
    return "g++ -std=c++20 -O2 -mavx2 -DNDEBUG (oracle/Makefile; the reference's headers, eve 4-lane AVX2 wides)" if kind == "reference" \
        else "gcc -O2 (oracle/Makefile; scalar C restatement)"


def one_thread_rate(run, per_core_guess, limit, seconds=2.2):
    """"also run 1-thread" -- `run(m)` = the same CPU path over the first m units on ONE thread; sized for
    >= 2 s from the all-cores figure.  Returns {value, units, seconds}."""
    m = int(min(limit, max(64, per_core_guess * seconds)))
    m -= m % 4
    t = time.perf_counter(); run(m); dt = time.perf_counter() - t
    if dt < 2.0 and m < limit:                       # the guess was low (turbo on one core): once more, scaled
        m = int(min(limit, m * 2.3 / max(dt, 1e-3))); m -= m % 4
        t = time.perf_counter(); run(m); dt = time.perf_counter() - t
    return {"value": m / dt, "units": m, "seconds": dt, "threads": 1}


def config1_ops8(kind):
    """BASELINE.json configs[0] / SURVEY.md 8(d) "Config 1": benchs/ops.cpp (mgry_sqr_256 :81-90, mgry_reduce_512 :92-100,
    mul_256 :36-45, the secp256k1 prime of :22-24) on 2 wides = 8 lanes, CPU only, timed inside the checker library
    (oracle/ref_driver.cpp bench_ops: a noinline call per wide per pass, like the Google Benchmark harness the reference
    uses and this image lacks), outputs hashed for parity between the compiled reference and the C restatement."""
    import hashlib
    import numpy as np
    from oracle import loader
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import fill_random_np
    curve = loader.SECP256K1
    a = fill_random_np(8, SEED, 31, clear_top_bits=1); b = fill_random_np(8, SEED, 32, clear_top_bits=1)
    a8 = np.concatenate([fill_random_np(8, SEED, 33), fill_random_np(8, SEED, 34, clear_top_bits=2)], axis=1)     # < p * 2^256
    libs = {"port": loader.Oracle(faithful=True)}
    if kind == "reference":
        libs["reference"] = loader.Reference()
    out = {"what": "benchs/ops.cpp on 2 wides (8 lanes), secp256k1 prime, CPU only; ns per call on one 4-lane wide", "timed": kind, "ops": {}}
    for op, (name, x, y) in enumerate([("mgry_sqr_256", a, None), ("mgry_reduce_512", a8, None), ("mul_256", a, b)]):
        hashes, ns = {}, None
        for lname, lib in libs.items():
            iters = 2000
            dt, res = lib.bench_ops(curve, op, x, y, iters)
            if lname == kind:
                iters = int(max(iters, min(5e6, 0.3 / max(dt / iters, 1e-9))))        # ~0.3 s
                dt, res = lib.bench_ops(curve, op, x, y, iters)
                ns = dt / iters / 2 * 1e9
            hashes[lname] = hashlib.sha256(res.tobytes()).hexdigest()[:16]
        out["ops"][name] = {"ns_per_wide": ns, "output_sha256_16": hashes[kind], "equals_the_restatement": len(set(hashes.values())) == 1}
    return out


def checker_curve(lib, name):
    """The id of curve `name` in a CHECKER library: 0 / 1 for the built-in curves; a curve registered at run time is one of the instances the compiled
    reference has (oracle/ref_driver.cpp ops<Curve>, ids 10..12) or a run-time registration of the C restatement."""
    if name in (None, "p256", "secp256k1"):
        return {None: 0, "p256": 0, "secp256k1": 1}[name]
    from ecsimd_amd.curves import NAMED
    c = NAMED[name]
    return lib.register_curve(c["p"], c["a"], c["b"], c["gx"], c["gy"])


def cpu_baseline(eng, curve, k, bx, by, gpu_out, target_s, failures, compat=False, name=None):
    """ecsimd's own CPU path (or the C port) on the host cores, bounded sample, rank 0 only; the
    sample's CPU result is also compared bit-for-bit with what the GPU produced for those elements.  `curve` is the ENGINE's id, `name` the curve's."""
    import numpy as np
    from oracle import loader
    cores = usable_cores()
    impl, kind = load_checkers()
    if name is None:
        name = {0: "p256", 1: "secp256k1"}[curve]
    builtin = name in ("p256", "secp256k1")
    gpu_curve, curve = curve, checker_curve(impl, name)          # below, `curve` is the checker's id
    to_np = eng.to_numpy
    # calibrate on a small sample, then size the real one for ~target_s seconds
    m0 = 256 * cores
    kn, xn, yn = (to_np(t[:m0]) for t in (k, bx, by))
    t = time.perf_counter(); impl.scalar_mult(curve, kn, xn, yn, threads=cores); dt = time.perf_counter() - t
    m = int(min(k.shape[0], max(m0, (target_s / dt) * m0)))
    m -= m % 4
    kn, xn, yn = (to_np(t[:m]) for t in (k, bx, by))
    if kind == "port" and compat:
        impl = loader.Oracle(faithful=True)              # the bug-for-bug restatement stands in for the reference
        curve = checker_curve(impl, name)
    t = time.perf_counter(); ref = impl.scalar_mult(curve, kn, xn, yn, threads=cores); dt = time.perf_counter() - t
    got = [to_np(gpu_out[j][:m]) for j in range(3)]
    bad = np.nonzero((got[0] != ref[0]).any(axis=1) | (got[1] != ref[1]).any(axis=1) | (got[2] != ref[2]).any(axis=1))[0]
    # The reference's square() drops a carry with probability ~3e-6 per random scalar mult
    # (mul.h:186-190,207; oracle/ecsimd_oracle.c bn_square; DESIGN.md "Reference defect").  Every
    # lane where the reference and the GPU differ must be such a lane: there the exact oracle has
    # to agree with the GPU, the bug-for-bug oracle with the reference, and libcrypto -- which shares
    # nothing with either -- with the GPU's affine point.  With ECSIMD_HIP_REF_SQUARE_COMPAT no lane may differ.
    explained, by_ossl, by_textbook = True, None, None
    if len(bad) and compat:
        explained = False
        failures.append("cpu_baseline: the reference-compatible ladder differs from the reference")
    elif len(bad):
        if not os.path.exists(loader.Oracle.path):
            loader.build()
        ex, fa = loader.Oracle(faithful=False), loader.Oracle(faithful=True)
        sub = lambda arrs: [a[bad] for a in arrs]
        e_ = ex.scalar_mult(checker_curve(ex, name), kn[bad], xn[bad], yn[bad], threads=min(cores, len(bad)))
        f_ = fa.scalar_mult(checker_curve(fa, name), kn[bad], xn[bad], yn[bad], threads=min(cores, len(bad)))
        explained = all(np.array_equal(u, v) for u, v in zip(e_, sub(got))) and all(np.array_equal(u, v) for u, v in zip(f_, sub(ref)))
        if not explained:
            failures.append("cpu_baseline: a lane differs from the reference and the two oracles do not attribute it to the square() defect")
        ossl = openssl_checker() if builtin else None      # libcrypto adjudicates on the two curves oracle/ossl_check.c knows
        if ossl is not None:
            ax, ay = eng.to_affine(gpu_curve, [eng.select_rows(t, bad) for t in gpu_out])
            vx, vy, inf = ossl.scalar_mult(curve, kn[bad], xn[bad], yn[bad], threads=1)
            by_ossl = int(np.count_nonzero(~((to_np(ax) != vx).any(axis=1) | (to_np(ay) != vy).any(axis=1) | (inf != 0))))
            if by_ossl != len(bad):
                failures.append("cpu_baseline: libcrypto does not confirm the GPU on a lane where it differs from the reference")
        elif not builtin:
            # a curve registered at run time: libcrypto's harness has no such curve, so the handful of differing lanes is settled by textbook affine
            # double-and-add on Python integers -- it must give the GPU's affine point (and not the reference's)
            from ecsimd_amd.curves import NAMED
            cp = NAMED[name]
            ax, ay = (to_np(t) for t in eng.to_affine(gpu_curve, [eng.select_rows(t, bad) for t in gpu_out]))
            ti = lambda v: sum(int(w) << (64 * j) for j, w in enumerate(v))
            by_textbook = sum(1 for j, lane in enumerate(bad) if textbook_scalar_mult(cp, ti(kn[lane]), ti(xn[lane]), ti(yn[lane])) == (ti(ax[j]), ti(ay[j])))
            if by_textbook != len(bad):
                failures.append("cpu_baseline: textbook affine arithmetic does not confirm the GPU on a lane where it differs from the reference")
    one = one_thread_rate(lambda m1: impl.scalar_mult(curve, kn[:m1], xn[:m1], yn[:m1], threads=1), (m / dt) / cores, m)
    c1 = None
    try:
        c1 = config1_ops8(kind)
        if not all(o["equals_the_restatement"] for o in c1["ops"].values()):
            failures.append("config 1: the compiled reference and the restatement disagree on benchs/ops.cpp's operations")
    except (OSError, AttributeError) as exc:                # a prebuilt checker without bench_ops: a side figure, not a failure
        c1 = {"error": repr(exc)[:200]}
    return {"_sample": (m, kn, xn, yn, ref, kind, bool(compat), name),    # for ref_compat_leg; removed before the line is printed
            "value": m / dt, "unit": "scalar_mults/s", "cores": cores, "kind": kind,
            "per_core": (m / dt) / cores, "one_thread": one, "cpu_model": cpu_model(), "flags": build_flags(kind), "config1_ops8": c1,
            "sample": f"first {m} (scalar, point) pairs of the GPU batch, {dt:.1f} s wall, {cores} threads, "
                      + ("g++ -O2 -mavx2 build of the reference headers" if kind == "reference" else "gcc -O2 C restatement"),
            "lanes_compared": int(m), "lanes_differing_from_gpu": int(len(bad)),
            "differences_all_explained_by_reference_square_defect": bool(explained),
            "lanes_differing_confirmed_by_openssl": by_ossl, **({"lanes_differing_confirmed_by_textbook_arithmetic": by_textbook} if by_textbook is not None else {})}


def textbook_scalar_mult(c, k, x, y):
    """k (x, y) on y^2 = x^3 + a x + b over GF(p) by affine double-and-add on Python integers; (0, 0) for the point at infinity."""
    p, a = c["p"], c["a"]

    def add(P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        if P[0] == Q[0]:
            if (P[1] + Q[1]) % p == 0:
                return None
            lam = (3 * P[0] * P[0] + a) * pow(2 * P[1], -1, p) % p
        else:
            lam = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
        x3 = (lam * lam - P[0] - Q[0]) % p
        return x3, (lam * (P[0] - x3) - P[1]) % p
    R = None
    for bit in bin(k)[2:] if k else "":
        R = add(R, R)
        if bit == "1":
            R = add(R, (x, y))
    return R if R is not None else (0, 0)


if __name__ == "__main__":
    try:
        code = main()
    except SystemExit:
        raise
    except BaseException:                   # any rank's failure is the job's failure: torch.distributed.run tears the others down
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
    sys.exit(code)
