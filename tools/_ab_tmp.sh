mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -x -k "split3 or split_products or extreme or repack or fixture" > gpurun_out/r5/t11.log 2>&1; tail -2 gpurun_out/r5/t11.log
for i in 1 2 3; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-roofline 2>&1 | tail -1 | cut -c1-140; done
