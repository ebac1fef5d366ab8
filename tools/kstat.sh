#!/bin/bash
# tools/kstat.sh <tag> [ENV=VALUE ...]: profiles 5 CNN forwards with the installed library and prints the per-kernel
# average durations (us)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/ks
env "$@" rocprofv3 --kernel-trace -d gpurun_out/ks -o $tag -- python tools/bench_cnn.py --steps 5 --warmup 2 > /dev/null 2>&1
python - <<PY
import sqlite3
c=sqlite3.connect("gpurun_out/ks/${tag}_results.db")
for r in c.execute("select name,count(*),avg(end-start)/1e3,sum(end-start)/1e3/7 from kernels where name like '%anonymous namespace%' group by name order by 4 desc").fetchall()[:8]:
    print("$tag %-70s n=%4d avg %8.2f us  per-frame %8.1f us"%(r[0][:70],r[1],r[2],r[3]))
PY
