#!/bin/bash
# tools/kstat.sh <variant>: installs tools/_variants/libmsi_<variant>.so, profiles 5 CNN forwards and prints
# the per-kernel average durations (us)
cp tools/_variants/libmsi_$1.so matryodshka_amd/libmsi_hip.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ks
rocprofv3 --kernel-trace -d gpurun_out/ks -o $1 -- python tools/bench_cnn.py --steps 5 --warmup 2 > /dev/null 2>&1
python - <<PY
import sqlite3
c=sqlite3.connect("gpurun_out/ks/$1_results.db")
for r in c.execute("select name,count(*),avg(end-start)/1e3,sum(end-start)/1e3/7 from kernels group by name order by 4 desc").fetchall()[:8]:
    print("$1 %-70s n=%4d avg %8.2f us  per-frame %8.1f us"%(r[0][:70],r[1],r[2],r[3]))
PY
