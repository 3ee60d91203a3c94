set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "64:--dtype f32x3" "128:--res 128 --dtype bf16"; do
  tag=${cfg%%:*}; fl=${cfg#*:}
  for pass in "f:FETCH_SIZE" "w:WRITE_SIZE"; do
    pt=${pass%%:*}; ctr=${pass#*:}
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/hbm2_${pt}$tag -o run -- python3 $R/bench.py $fl --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline > $R/gpurun_out/hbm2_${pt}$tag.log 2>&1; rc=$?
    if [ $rc -ne 0 ]; then tail -3 $R/gpurun_out/hbm2_${pt}$tag.log; exit $rc; fi
  done
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/hbm2_t$tag -o run -- python3 $R/bench.py $fl --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline > $R/gpurun_out/hbm2_t$tag.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then exit $rc; fi
done
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clk64 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline > $R/gpurun_out/clk64.log 2>&1; rc=$?
echo done $rc
exit $rc
