#!/bin/bash
# tools/variants_pmc.sh PATTERN CONFIG (on the GPU box): tools/pmc_kernel.sh with every tools/_variants/libmsi_*.so installed in turn
cd "$GRAFT_REPO_ROOT" || exit 1
cp matryodshka_amd/libmsi_hip.so /tmp/libmsi_saved.so
for v in tools/_variants/libmsi_*.so; do
  cp "$v" matryodshka_amd/libmsi_hip.so
  echo "== $v"
  bash tools/pmc_kernel.sh "$1" "${2:-1}" 2>&1 | grep -v "^err"
done
cp /tmp/libmsi_saved.so matryodshka_amd/libmsi_hip.so
