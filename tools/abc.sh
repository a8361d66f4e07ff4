#!/bin/bash
# A/B/C... of several builds of the library on the same GPU box: tools/abc.sh "<lib1.so> <lib2.so> ..." [bench args]; three rounds, alternating
libs=$1; shift
for i in 1 2 3; do
  for l in $libs; do
    v=$(python3 bench.py --lib $PWD/$l --steps 30 --no-cpu-baseline --no-other-input --sustained 0 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    echo "$l $v"
  done
done
