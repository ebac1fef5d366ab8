#!/bin/bash
# tools/ab_geometry.sh (on the GPU box, from the repo root): same-box A/B of the geometry kernels' durations by rocprofv3 --kernel-trace,
# two interleaved repeats of tools/_variants/libmsi_old.so and libmsi_new.so (build the two libraries first; the last one stays installed).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/ab
for rep in 1 2; do for v in old new; do
  cp tools/_variants/libmsi_$v.so matryodshka_amd/libmsi_hip.so
  rocprofv3 --kernel-trace -d gpurun_out/ab -o ${v}$rep -- python bench.py --steps 20 --warmup 3 --repeats 0 --no-cpu-baseline --no-sustained-probe --prewarm 0.3 --strong-frames 0 > /dev/null 2>&1
  python - <<PY
import sqlite3
c=sqlite3.connect("gpurun_out/ab/${v}${rep}_results.db")
for r in c.execute("select name,count(*),avg(end-start)/1e3,min(end-start)/1e3 from kernels where name like '%ods_sweep%' or name like '%render_kernel%' group by name").fetchall():
    print("$v$rep %-50s n=%3d avg %7.2f us min %7.2f"%(r[0].replace('(anonymous namespace)::','')[:50],r[1],r[2],r[3]))
PY
done; done
rm -rf gpurun_out/ab
