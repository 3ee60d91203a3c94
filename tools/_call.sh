set -o pipefail
mkdir -p gpurun_out
AB=$PWD/attribute-guided-image-generation-from-layout_amd/agl/ab
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "pconv or conv" > gpurun_out/t_pf.log 2>&1; rc=$?; tail -2 gpurun_out/t_pf.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" gpurun_out/t_pf.log | head; exit $rc; fi
for m in split bf16; do
for v in old base; do
  if [ $v = base ]; then unset AGL_LIBRARY; else export AGL_LIBRARY=$AB/libagl_$v.so; fi
  if [ $m = split ]; then export AGL_SPLIT3=1; unset AGL_PREC; else unset AGL_SPLIT3; export AGL_PREC=bf16; fi
  timeout -k 10 200 python tools/conv_bench.py "" > gpurun_out/cb_${v}_${m}.txt 2>&1 || exit 1
done
done
unset AGL_SPLIT3 AGL_PREC
for v in old base old base; do
  if [ $v = base ]; then unset AGL_LIBRARY; else export AGL_LIBRARY=$AB/libagl_$v.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])" || exit 1
done
