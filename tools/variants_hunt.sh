#!/bin/bash
# tools/variants_hunt.sh RUNS "extra wobble_hunt args" name1 name2 ... (GPU box): tools/wobble_hunt.py with every named tools/_variants/libmsi_<name>.so installed in turn
cd "$GRAFT_REPO_ROOT" || exit 1
RUNS=$1; EXTRA=$2; shift 2
cp matryodshka_amd/libmsi_hip.so /tmp/libmsi_saved.so
for v in "$@"; do
  cp tools/_variants/libmsi_$v.so matryodshka_amd/libmsi_hip.so
  timeout 900 python tools/wobble_hunt.py --runs $RUNS --tag $v $EXTRA 2>&1 | grep -v "layer 14\|amdgpu.ids" | tail -150
done
cp /tmp/libmsi_saved.so matryodshka_amd/libmsi_hip.so
