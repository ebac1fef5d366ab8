#!/bin/bash
# tools/pmc_custom.sh PATTERN CONFIG "COUNTER SET 1" ["COUNTER SET 2" ...] (on the GPU box): arbitrary counter sets (one rocprofv3 --pmc pass
# each, --kernel-trace only) for the kernels whose name contains PATTERN; BENCH_EXTRA = extra bench.py arguments.
PAT=$1; CFG=$2; shift 2
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/pmcc
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmcc/s$i -o p -- python bench.py --config $CFG ${BENCH_EXTRA:-} --steps 3 --warmup 1 --repeats 0 --no-cpu-baseline --no-sustained-probe --prewarm 0 --strong-frames 0 --no-settle > gpurun_out/pmcc/s$i.log 2>&1 || tail -3 gpurun_out/pmcc/s$i.log
  python - "s$i" "$PAT" <<'PY'
import sqlite3, sys, glob
n, pat = sys.argv[1], sys.argv[2]
for db in glob.glob("gpurun_out/pmcc/%s/*_results.db" % n):
    c=sqlite3.connect(db)
    try:
        rows=c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like ? group by kernel_name, counter_name", ('%'+pat+'%',)).fetchall()
        for r in rows: print(r[0].replace('(anonymous namespace)::','').replace('void ','')[:40], r[1], "%.4g per launch" % (r[2]/r[3]))
    except Exception as e: print("err", e)
PY
done
rm -rf gpurun_out/pmcc/*/
