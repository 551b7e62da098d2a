#!/bin/bash
# GPU-box step: what the driver runs at round end, on the final tree -- the -m gpu suite, smoke(), the default bench line.
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -q > gpurun_out/r03/pytest_gpu_final.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest_gpu_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 10 --warmup 2 > gpurun_out/r03/bench_final.json 2> gpurun_out/r03/bench_final.err; echo "bench rc=$? lines=$(wc -l < gpurun_out/r03/bench_final.json)"
python -c "import json; d=json.load(open('gpurun_out/r03/bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['lanes_differing_from_gpu'])"
