#!/bin/bash
# A/B runs on the GPU box (via gpurun): driver modes, kernel variants (tools/build_variants.sh) and launch
# geometries of bench.py at C3, plus the other workloads.  Output: gpurun_out/ab.log (one line per run).
mkdir -p gpurun_out
L=gpurun_out/ab.log
: > $L
fmt='
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d["roofline"]
    print("  it/s=%.1f ms=%.3f avg_launch_ms=%.3f iter_frac=%s kernels=%s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"], r["iteration"]["frac_of_hbm_peak"], {k: round(v,3) for k,v in r["kernels_ms_per_step"].items()}))
'
run() { echo "### env: ${ENVV:-} args: $*" >> $L; env $ENVV timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | grep '"metric"' | python -c "$fmt" >> $L 2>&1; }
V=$PWD/hpfrec_amd/variants
run
run --no-fuse
run --lean

ENVV="HPF_FORCE_SHARDED=1" run --no-autotune
run --workload c2
run --workload c4
if [ -d "$V" ]; then
  for v in u2 u4 u16 nt w6 w8; do [ -f $V/$v.so ] && ENVV="HPF_HIP_SO=$V/$v.so" run; done
  for bpc in 4 5 6 10 16; do ENVV="HPF_SWEEP_BPC=$bpc" run; done
fi
cat $L
