#!/bin/bash
# GPU-box step: the driver's own command on the final tree (clean rebuild), plus smoke.
mkdir -p gpurun_out/r03/lines
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python3 bench.py --gpus 1 --steps 10 --warmup 2 > gpurun_out/r03/lines/bench_n1_ladder_final.json 2> gpurun_out/r03/lines/bench_n1_ladder_final.err ) 2>&1 | tail -3
python3 -c "import json; d=json.load(open('gpurun_out/r03/lines/bench_n1_ladder_final.json')); print('%.3f M/s frac %.3f kernel %.2f ms; cpu %s; failures %s' % (d['value']/1e6, d['roofline']['frac'], d['roofline']['kernel_ms'], d['cpu_baseline']['value'], d.get('parity_failures')))"
