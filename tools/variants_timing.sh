#!/bin/bash
# tools/variants_timing.sh [conv_timing.py args]: tools/conv_timing.py with every tools/_variants/libmsi_*.so installed in turn
# (default arguments: --bf16 --batch 16; pass e.g. "--f32 0 1 4" for the fp32 plan's layers)
cd "$GRAFT_REPO_ROOT" || exit 1
cp matryodshka_amd/libmsi_hip.so /tmp/libmsi_saved.so
ARGS="--bf16 --batch 16 $*"
case " $* " in *" --f32 "*) ARGS=$(echo "$*" | sed 's/--f32//');; esac
for v in tools/_variants/libmsi_*.so; do
  cp "$v" matryodshka_amd/libmsi_hip.so
  echo "== $v"
  python tools/conv_timing.py $ARGS 2>&1 | grep "per block\|logue:\|CUs seen" | cut -c1-175
done
cp /tmp/libmsi_saved.so matryodshka_amd/libmsi_hip.so
