#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/ab5.log
: > $L
fmt='
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d["roofline"]
    print("  it/s=%.1f ms=%.3f avg_launch_ms=%.3f iter_frac=%.3f kernels=%s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["iteration"]["frac_of_hbm_peak"], {k: round(v,3) for k,v in r["kernels_ms_per_step"].items()}))
'
run() { echo "### $*" >> $L; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA 2>/dev/null | grep -E '"metric"' | python -c "$fmt" >> $L 2>&1; }
V=$PWD/hpfrec_amd/variants
run HPF_HIP_SO=$V/base.so
run HPF_HIP_SO=$V/base.so HPF_SWEEP_BPC=4
run HPF_HIP_SO=$V/u4.so HPF_SWEEP_BPC=6
run HPF_HIP_SO=$V/u4.so HPF_SWEEP_BPC=12
run HPF_HIP_SO=$V/u4.so HPF_SWEEP_BPC=8
EXTRA=--no-fuse run HPF_HIP_SO=$V/base.so
run HPF_HIP_SO=$V/base.so HPF_FORCE_SHARDED=1
EXTRA="--workload c4" run HPF_HIP_SO=$V/base.so
EXTRA="--workload c2" run HPF_HIP_SO=$V/base.so
cat $L
