#!/bin/bash
# same-box A/B of the training step: each argument is an "ENV=VAL,ENV=VAL" set (or "base"); prints clips/s per set, two rounds
for round in 1 2; do
  for set in "$@"; do
    envs=$(echo "$set" | tr ',' ' ')
    [ "$set" = "base" ] && envs=""
    v=$(env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $round  $set  ->  $v"
  done
done
