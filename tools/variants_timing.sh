#!/bin/bash
# tools/variants_timing.sh LAYERS... : tools/conv_timing.py --bf16 --batch 16 with every tools/_variants/libmsi_*.so installed in turn
cd "$GRAFT_REPO_ROOT" || exit 1
cp matryodshka_amd/libmsi_hip.so /tmp/libmsi_saved.so
for v in tools/_variants/libmsi_*.so; do
  cp "$v" matryodshka_amd/libmsi_hip.so
  echo "== $v"
  python tools/conv_timing.py --bf16 --batch 16 "$@" 2>&1 | grep "per block\|epilogue:" | cut -c1-175
done
cp /tmp/libmsi_saved.so matryodshka_amd/libmsi_hip.so
