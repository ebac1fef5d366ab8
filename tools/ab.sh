#!/bin/bash
# same-box A/B of conv kernel variants: tools/ab.sh base pre ...   (libraries tools/_variants/libmsi_<name>.so)
# prints the CNN forward time of each variant twice (interleaved) and leaves the LAST variant installed.
for rep in 1 2; do
  for v in "$@"; do
    cp tools/_variants/libmsi_$v.so matryodshka_amd/libmsi_hip.so
    echo -n "$v  "; python tools/bench_cnn.py --steps 30 2>&1 | grep "cnn forward"
  done
done
