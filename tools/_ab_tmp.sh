R=$PWD
cd /tmp && export TMPDIR=/tmp
for b in 1 0; do
O=$R/gpurun_out/r5/blk$b; mkdir -p $O
AGL_D_BLOCKED=$b AGL_D_STREAMS=0 AGL_G_STREAMS=0 AGL_WGRAD_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -o run -- python3 $R/bench.py --res 128 --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --vary-batch 0 > $O/s.log 2>&1
cp $O/s/run_kernel_stats.csv $O/stats.csv; rm -rf $O/s
done
echo done
