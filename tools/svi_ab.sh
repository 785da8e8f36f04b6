#!/bin/bash
# A/B of environment switches of the SVI path on ONE box, alternating: tools/svi_ab.sh <tag> <rounds> "<ENV=..>" "<ENV=..>" ...
# every variant's tools/svi_c5.py summary lines go to gpurun_out/<tag>/svi_ab.txt
TAG=$1; ROUNDS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
: > $OUT/svi_ab.txt
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    echo "== round $r: $v" >> $OUT/svi_ab.txt
    env $v timeout 300 python tools/svi_c5.py 10 2>&1 | grep "^C5 SVI\|^epoch loop\|^roofline\|Error\|error" | cut -c1-330 >> $OUT/svi_ab.txt
  done
done
grep "== round\|steady state" $OUT/svi_ab.txt | sed 's/C5 SVI.*steady state/   steady state/; s/; llk.*//'
