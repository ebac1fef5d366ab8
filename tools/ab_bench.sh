#!/bin/bash
# same-box A/B of whole-frame variants: tools/ab_bench.sh cur lean ... -> fps and the geometry stage times
for rep in 1 2; do
  for v in "$@"; do
    cp tools/_variants/libmsi_$v.so matryodshka_amd/libmsi_hip.so
    echo -n "$v  "; python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()})"
  done
done
