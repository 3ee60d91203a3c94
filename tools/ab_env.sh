#!/bin/bash
# Same-box A/B of environment switches on one bench configuration: tools/ab_env.sh "<bench flags>" "<envA>" "<envB>" ... (two rounds, alternating)
set -o pipefail
flags=$1; shift
for round in 1 2; do
  for e in "$@"; do
    v=$(env $e timeout -k 10 200 python bench.py $flags --no-cpu-baseline --no-secondary --no-roofline --steps 10 --warmup 3 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
    echo "round $round [$e] $flags -> $v"
  done
done
