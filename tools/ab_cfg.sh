#!/bin/bash
# tools/ab_cfg.sh "CFGS" [ROUNDS] [extra bench args]: every tools/_variants/libmsi_*.so installed in turn, bench.py --config for each of CFGS, interleaved rounds; prints fps + stage times
cd "$GRAFT_REPO_ROOT" || exit 1
CFGS=${1:-1}; ROUNDS=${2:-2}; shift; shift
cp matryodshka_amd/libmsi_hip.so /tmp/libmsi_saved.so
for r in $(seq $ROUNDS); do
for cfg in $CFGS; do
  for v in tools/_variants/libmsi_*.so; do
    cp "$v" matryodshka_amd/libmsi_hip.so
    python bench.py --config $cfg --no-cpu-baseline --no-alt-arithmetic --strong-frames 0 --repeats 1 "$@" 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg $cfg', '$v'.split('libmsi_')[1], j['value'], j['ms_per_step'], 'cnn', j['roofline']['ms_per_forward'], {k:v['ms'] for k,v in j['stages'].items() if isinstance(v, dict)})"
  done
done
done
cp /tmp/libmsi_saved.so matryodshka_amd/libmsi_hip.so
