#!/bin/bash
# tools/layers.sh <variant>: installs tools/_variants/libmsi_<variant>.so, traces 5 CNN forwards, prints per-layer conv times
cp tools/_variants/libmsi_$1.so matryodshka_amd/libmsi_hip.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ks
rocprofv3 --kernel-trace -d gpurun_out/ks -o L$1 -- python tools/bench_cnn.py --steps 5 --warmup 2 > /dev/null 2>&1
python - <<PY
import sqlite3
c=sqlite3.connect("gpurun_out/ks/L$1_results.db")
rows=c.execute("select (end-start)/1e3 from kernels where name like '%conv_igemm%' order by start").fetchall()
n=len(rows)//18
import numpy as np
a=np.array([r[0] for r in rows[-18*5:]]).reshape(5,18).mean(0)
print("$1", " ".join("%.1f"%x for x in a), " sum %.1f"%a.sum())
for nm in ("ln_apply","conv_fixup"):
    r=c.execute("select sum(end-start)/1e3/7 from kernels where name like '%"+nm+"%'").fetchone()
    print("$1", nm, "%.1f us/frame"%r[0])
PY
