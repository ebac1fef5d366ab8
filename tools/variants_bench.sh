#!/bin/bash
# tools/variants_bench.sh CONFIG [ROUNDS]: bench.py --config CONFIG with every tools/_variants/libmsi_*.so installed in turn, ROUNDS interleaved rounds
cd "$GRAFT_REPO_ROOT" || exit 1
CFG=${1:-1}; ROUNDS=${2:-2}
cp matryodshka_amd/libmsi_hip.so /tmp/libmsi_saved.so
for r in $(seq $ROUNDS); do
  for v in tools/_variants/libmsi_*.so; do
    cp "$v" matryodshka_amd/libmsi_hip.so
    python bench.py --config $CFG --no-cpu-baseline --strong-frames 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['value'], j['ms_per_step'], 'cnn', j['roofline']['ms_per_forward'])"
  done
done
cp /tmp/libmsi_saved.so matryodshka_amd/libmsi_hip.so
