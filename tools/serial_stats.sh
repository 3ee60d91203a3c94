#!/bin/bash
# Kernel stats of the single-stream schedule for both configurations: tools/serial_stats.sh <tag>  ->  gpurun_out/<tag>/bench{64_f32x3,128_bf16}_serial_kernel_stats.csv
R=$GRAFT_REPO_ROOT; tag=${1:-r4s}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$tag; mkdir -p $O
for cfg in "64_f32x3:" "128_bf16:--res 128 --dtype bf16"; do
  name=${cfg%%:*}; fl=${cfg#*:}
  AGL_D_STREAMS=0 AGL_G_STREAMS=0 AGL_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$name -o run -- python3 $R/bench.py $fl --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --vary-batch 0 > $O/s_$name.log 2>&1 || exit 1
  cp $O/s_$name/run_kernel_stats.csv $O/bench${name}_serial_kernel_stats.csv
  rm -rf $O/s_$name
done
