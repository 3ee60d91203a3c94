mkdir -p gpurun_out/r4
{
for rc in 22 21 12 11; do
  export AGL_PBWW_RC3=$rc AGL_PREC=bf16
  echo "== rt,ct = $rc"
  python tools/one_conv.py 210 64 64 64 3 1 1 bwd_weight
  python tools/one_conv.py 210 128 32 128 3 1 1 bwd_weight
  python tools/one_conv.py 32 64 128 64 3 1 1 bwd_weight
  python tools/one_conv.py 32 128 64 128 3 1 1 bwd_weight
  python tools/one_conv.py 32 256 32 256 3 1 1 bwd_weight
  python tools/one_conv.py 32 512 16 512 3 1 1 bwd_weight
  python tools/one_conv.py 210 256 16 256 3 1 1 bwd_weight
done
} > gpurun_out/r4/abl_rc3.txt 2>&1
