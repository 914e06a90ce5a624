#!/bin/bash
# same-box A/B of one environment knob on the train-step bench: scripts/ab_env.sh NAME VALUE_A VALUE_B [rounds]
cd "$(dirname "$0")/.."
name=$1; a=$2; b=$3; rounds=${4:-2}
for r in $(seq $rounds); do
  for v in $a $b; do
    ms=$(env $name=$v python bench.py --no-cpu-baseline --steps 60 --warmup 8 2>/dev/null | python -c "import sys,json; print(json.load(sys.stdin)['ms_per_step'])")
    echo "$name=$v round $r: $ms ms/step"
  done
done
