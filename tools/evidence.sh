#!/bin/bash
# Run on the GPU box (via gpurun): the raw outputs DESIGN.md quotes, one file each under gpurun_out/<tag>/
# (copied into profiles/ afterwards).  usage: tools/evidence.sh <tag> [section ...]   (default: all sections)
TAG=${1:-evidence}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
SECTIONS=${*:-"gather bench scale svi serving e2e shard"}
has() { case " $SECTIONS " in *" $1 "*) return 0;; esac; return 1; }
run() { f=$1; shift; echo "== $*" > $OUT/$f; timeout 900 "$@" >> $OUT/$f 2>&1; echo "== exit $?" >> $OUT/$f; }
has gather && run gather_probe.txt python tools/gather_probe.py
if has bench; then
  for wl in c2 c4 k30 k200 c3u; do
    run bench_$wl.json python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline
  done
fi
if has scale; then
  run scale_x10.txt python tools/scale_probe.py 10
  run scale_x25.txt python tools/scale_probe.py 25
fi
has svi && run svi_c5.txt python tools/svi_c5.py 10
has serving && run serving_latency.txt python tools/serving_latency.py
has e2e && run e2e_fit.txt python tools/e2e_fit.py c3 100
if has shard; then
  run shard_probe_c3.txt env PROBE_EVENTS=0 python tools/shard_probe.py 8
  run shard_probe_c4.txt env PROBE_WORKLOAD=c4 PROBE_EVENTS=0 python tools/shard_probe.py 8
  for sc in direct gather-early finalize-then-gather; do
    run shard_probe_c3_$sc.txt env HPF_SCHEDULE=$sc PROBE_EVENTS=0 python tools/shard_probe.py 8
  done
  run shard_probe_c3_all_n.txt env PROBE_EVENTS=0 python tools/shard_probe.py 1 2 4 8
fi
if has default; then
  run bench_c3_default.json python bench.py
fi
tail -n 3 $OUT/*
