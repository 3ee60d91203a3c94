set -o pipefail
mkdir -p gpurun_out
AB=attribute-guided-image-generation-from-layout_amd/agl/ab
export AGL_LIBRARY=$PWD/$AB/libagl_cil.so
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "pconv" > gpurun_out/t_db.log 2>&1; rc=$?; tail -3 gpurun_out/t_db.log
if [ $rc -ne 0 ]; then exit $rc; fi
for v in base cil; do
  for m in split bf16; do
    if [ $v = base ]; then unset AGL_LIBRARY; else export AGL_LIBRARY=$PWD/$AB/libagl_$v.so; fi
    if [ $m = split ]; then export AGL_SPLIT3=1; unset AGL_PREC; else unset AGL_SPLIT3; export AGL_PREC=bf16; fi
    timeout -k 10 200 python tools/conv_bench.py " k3 @" > gpurun_out/cb_${v}_${m}.txt 2>&1; rc=$?
    if [ $rc -ne 0 ]; then tail -5 gpurun_out/cb_${v}_${m}.txt; exit $rc; fi
  done
done
echo done
