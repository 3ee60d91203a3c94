set -o pipefail
mkdir -p gpurun_out
AB=$PWD/attribute-guided-image-generation-from-layout_amd/agl/ab
export AGL_SPLIT3=1
for v in base cb4; do
  if [ $v = base ]; then unset AGL_LIBRARY; else export AGL_LIBRARY=$AB/libagl_$v.so; fi
  timeout -k 10 200 python tools/conv_bench.py " 3>" > gpurun_out/cb_${v}_fa.txt 2>&1 || exit 1
  timeout -k 10 200 python tools/conv_bench.py ">3 " > gpurun_out/cb_${v}_fb.txt 2>&1 || exit 1
done
