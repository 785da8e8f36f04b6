#!/bin/bash
# A/B of differently-tuned builds (hpfrec_amd/variants/*.so) and launch geometries on the GPU box
mkdir -p gpurun_out
L=gpurun_out/ab2.log
: > $L
fmt='
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d["roofline"]
    print("  it/s=%.1f ms=%.3f avg_launch_ms=%.3f iter_frac=%.3f kernels=%s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["iteration"]["frac_of_hbm_peak"], {k: round(v,3) for k,v in r["kernels_ms_per_step"].items()}))
'
run() { echo "### $*" >> $L; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA 2>&1 | grep '"metric"' | python -c "$fmt" >> $L 2>&1; }
for rep in 1 2; do
for v in base u8 u2 nt nt_u8 w6 w8; do
  run HPF_HIP_SO=$PWD/hpfrec_amd/variants/$v.so
done
done
for bpc in 4 6 12 16; do run HPF_SWEEP_BPC=$bpc; done
cat $L
