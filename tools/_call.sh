set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2_t23.log 2>&1; rc=$?; tail -3 gpurun_out/r2_t23.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 500 python bench.py > gpurun_out/bench_r2k.log 2>&1; rc=$?; tail -1 gpurun_out/bench_r2k.log | cut -c1-200
if [ $rc -ne 0 ]; then exit $rc; fi
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "64_f32x3:--dtype f32x3" "64_f32:--dtype f32" "128_bf16:--res 128 --dtype bf16"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2k_$tag -o run -- python3 $R/bench.py $fl --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/prof_r2k_$tag.log 2>&1; rc=$?
  grep -o '"value": [0-9.]*' $R/gpurun_out/prof_r2k_$tag.log
  if [ $rc -ne 0 ]; then exit $rc; fi
  rm -f $R/gpurun_out/prof_r2k_$tag/run_kernel_trace.csv
done
for cfg in "64:--dtype f32x3" "128:--res 128 --dtype bf16"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  for pass in "f:FETCH_SIZE" "w:WRITE_SIZE"; do
    pt=${pass%%:*}; ctr=${pass#*:}
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/hbm4_${pt}$tag -o run -- python3 $R/bench.py $fl --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline > $R/gpurun_out/hbm4_${pt}$tag.log 2>&1; rc=$?
    if [ $rc -ne 0 ]; then tail -3 $R/gpurun_out/hbm4_${pt}$tag.log; exit $rc; fi
  done
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/hbm4_t$tag -o run -- python3 $R/bench.py $fl --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline > $R/gpurun_out/hbm4_t$tag.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then exit $rc; fi
done
echo all done
