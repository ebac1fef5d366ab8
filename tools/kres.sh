#!/bin/bash
# tools/kres.sh [defines]: VGPRs / AGPRs / scratch bytes / LDS of every kernel of the convolution path's units (hipcc -Rpass-analysis=kernel-resource-usage)
cd "$(dirname "$0")/.." || exit 1
for u in cnn_igemm cnn_halo cnn_x3 cnn_bf16 cnn_tail; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Iinclude -Imatryodshka_amd/csrc --offload-device-only "$@" -c matryodshka_amd/csrc/$u.hip -o /tmp/cnn_kres.o \
  -Rpass-analysis=kernel-resource-usage 2>&1; done | python3 -c '
import re, sys
cur = None
for l in sys.stdin:
    m = re.search(r"remark: +Function Name: (\S+)", l)
    if m: cur = m.group(1); d = {}; continue
    m = re.search(r"remark: +(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", l)
    if m and cur:
        d[m.group(1).split()[0]] = int(m.group(2))
        if m.group(1).startswith("Occupancy"):
            import subprocess
            name = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
            print("%-46s vgpr %3d agpr %3d scratch %4d occ %d" % (name, d.get("VGPRs", -1), d.get("AGPRs", -1), d.get("ScratchSize", -1), d.get("Occupancy", -1)))
            cur = None
'
