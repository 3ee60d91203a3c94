set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "partials or pconv" > gpurun_out/t_st.log 2>&1; rc=$?; tail -15 gpurun_out/t_st.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu > gpurun_out/t_st2.log 2>&1; rc=$?; tail -5 gpurun_out/t_st2.log
if [ $rc -ne 0 ]; then exit $rc; fi
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['achieved'], d['roofline_hbm']['kernel_ms_per_step'], d['roofline_hbm']['launches_per_step'])" || exit 1
done
