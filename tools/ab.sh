#!/bin/bash
# same-box A/B of conv kernel variants: tools/ab.sh base pre ...   (libraries tools/_variants/libmsi_<name>.so)
# a variant may carry environment settings: name:VAR=VALUE[:VAR=VALUE...]
# prints the CNN forward time of each variant twice (interleaved) and leaves the LAST variant installed.
for rep in 1 2; do
  for spec in "$@"; do
    v=${spec%%:*}
    envs=""
    if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ':' ' '); fi
    cp tools/_variants/libmsi_$v.so matryodshka_amd/libmsi_hip.so
    echo -n "$spec  "; env $envs python tools/bench_cnn.py --steps 30 2>&1 | grep "cnn forward"
  done
done
