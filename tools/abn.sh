#!/bin/bash
# A/B/C... of several builds of the library on one GPU box: tools/abn.sh "<bench args>" lib1.so lib2.so ... ; three alternating rounds
args=$1; shift
for i in 1 2 3; do
  for l in "$@"; do
    v=$(python3 bench.py --lib $PWD/$l --steps 30 --no-cpu-baseline --no-other-input --sustained 0 $args 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
    echo "$l $v"
  done
done
