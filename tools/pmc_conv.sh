#!/bin/bash
# SQ counters of one convolution shape under rocprofv3 (one pass: 8 SQ counters): tools/pmc_conv.sh <outdir> <one_conv.py args...>
# (arithmetic through the environment: AGL_SPLIT3=1 / AGL_PREC=bf16).  The program comes directly after `--`.
out=$1; shift
cd /tmp 2>/dev/null; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace -d "$out" -- python3 tools/one_conv.py "$@" > "$out.log" 2>&1
python3 tools/pmc_summary.py "$out" pconv_k >> "$out.log" 2>&1
python3 tools/pmc_summary.py "$out" pbww_k >> "$out.log" 2>&1
