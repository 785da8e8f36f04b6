#!/bin/bash
# Cross-compile differently-tuned builds of the same kernel source into hpfrec_amd/variants/ (run in the build
# container; the .so files travel to the GPU box).  Selected at run time with HPF_HIP_SO=<path>.
set -e
cd "$(dirname "$0")/.."
mkdir -p hpfrec_amd/variants
for v in "base:" "u2:-DHPF_U=2" "u4:-DHPF_U=4" "u16:-DHPF_U=16" "nt:-DHPF_NT=1" "w6:-DHPF_SWEEP_WAVES_PER_EU=6" "w8:-DHPF_SWEEP_WAVES_PER_EU=8"; do
  n=${v%%:*}; f=${v#*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude $f -o hpfrec_amd/variants/$n.so hpfrec_amd/csrc/hpf_hip.hip
  echo "built variants/$n.so ($f)"
done
