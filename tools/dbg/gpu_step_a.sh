#!/bin/bash
# GPU-box step: the -m gpu suite, then the reference-compatible ladder on both curves (used while tuning sqr8_ref).
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu_2.txt 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r03/pytest_gpu_2.txt
python bench.py --steps 5 --warmup 1 --workload ladder-ref-compat --cpu-seconds 6 > gpurun_out/r03/bench_compat_p256_1.json 2> gpurun_out/r03/bench_compat_p256_1.err; echo "rc=$?"
python bench.py --steps 5 --warmup 1 --workload ladder-ref-compat --curve secp256k1 --no-cpu-baseline > gpurun_out/r03/bench_compat_k1_1.json 2>/dev/null
python - <<PY
import json
for f in ("bench_compat_p256_1", "bench_compat_k1_1"):
    d = json.load(open("gpurun_out/r03/%s.json" % f)); c = d.get("cpu_baseline", {})
    print(f, d["value"], d["roofline"]["kernel_ms"], c.get("lanes_differing_from_gpu"), c.get("lanes_compared"))
PY
