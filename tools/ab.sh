#!/bin/bash
# A/B of two builds of the library on the same GPU box: tools/ab.sh <libA.so> <libB.so> [bench args]; alternates A, B, A, B
a=$1; b=$2; shift 2
for i in 1 2 3; do
  for l in $a $b; do
    v=$(python3 bench.py --lib $PWD/$l --steps 30 --no-cpu-baseline --no-other-input --sustained 0 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    echo "$l $v"
  done
done
