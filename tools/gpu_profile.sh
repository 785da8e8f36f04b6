#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + separate PMC passes of the bench command,
# condensed into gpurun_out/profile_summary.txt (raw rocprofv3 output is deleted: too big to ship).
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
S=$OUT/profile_summary.txt
: > $S
echo "# rocprofv3 summary. kernel trace: python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $*; PMC passes: --steps 3 --warmup 1 --no-events --no-extras" >> $S
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras "$@" > $OUT/kt.log 2>&1
python tools/prof_summary.py /tmp/prof_kt hpf_ >> $S 2>&1
grep -h '"metric"' $OUT/kt.log >> $S
# (last two passes: how many of the L2's memory-side requests are DESTINED FOR DRAM (the memory controller) rather than for
#  another agent / IO -- the only HBM-side split rocprofv3 offers on gfx950; it lists no Infinity-Cache (MALL) counter at all:
#  profiles/r04_rocprofv3_memory_counters_available.txt)
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" ; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$T -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events --no-extras "$@" > $OUT/pmc_$T.log 2>&1
  python tools/prof_summary.py /tmp/prof_$T sweep_kernel row_finalize >> $S 2>&1
done
# HBM-side bytes per launch of the dominant kernel, tied to the kernel source (read by bench.py: roofline.traffic)
WL=c3; prev=""; for a in "$@"; do [ "$prev" = "--workload" ] && WL=$a; prev=$a; done
python tools/pmc_json.py /tmp/prof_FETCH_SIZE /tmp/prof_WRITE_SIZE $WL $OUT/pmc_${WL}_n1.json sweep_kernel /tmp/prof_TCC_EA0_RDREQ_sum_TCC_EA0_RDREQ_DRAM_sum /tmp/prof_TCC_EA0_WRREQ_sum_TCC_EA0_WRREQ_DRAM_sum >> $S 2>&1
cat $S
