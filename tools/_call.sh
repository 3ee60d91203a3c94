set -o pipefail
mkdir -p gpurun_out
AB=$PWD/attribute-guided-image-generation-from-layout_amd/agl/ab
for v in old base old base; do
  if [ $v = base ]; then unset AGL_LIBRARY; else export AGL_LIBRARY=$AB/libagl_$v.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline_hbm']['kernel_ms_per_step'], d['roofline_hbm']['achieved'])" || exit 1
done
