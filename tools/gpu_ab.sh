#!/bin/bash
# quick A/B of bench variants on the GPU box; each line of $@ separated by '--' is a bench arg set
mkdir -p gpurun_out
: > gpurun_out/ab.log
run() { echo "### bench.py $*" >> gpurun_out/ab.log; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>&1 | grep '"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('  it/s=%.1f ms=%.3f kernel_frac=%.3f avg_launch_ms=%.3f iter_frac=%s per-step kernels=%s' % (d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms'], r['iteration']['frac_of_hbm_peak'], {k: round(v,3) for k,v in r['kernels_ms_per_step'].items()}))
" >> gpurun_out/ab.log 2>&1; }
run
run --no-fuse
run --lean
run --lean --no-fuse
run --workload c2
run --workload c4
cat gpurun_out/ab.log
