#!/bin/bash
# tools/pmc_kernel.sh PATTERN CONFIG (on the GPU box): SQ counters of the kernels whose name contains PATTERN at bench --config CONFIG,
# separate --pmc passes with --kernel-trace only; prints per-launch values and the kernel's average duration.
PAT=$1; CFG=${2:-1}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/pmck
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_LDS_MEM_VIOLATIONS"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmck/$n -o p -- python bench.py --config $CFG ${BENCH_EXTRA:-} --steps 3 --warmup 1 --repeats 0 --no-cpu-baseline --no-sustained-probe --prewarm 0 --strong-frames 0 > gpurun_out/pmck/$n.log 2>&1
  python - "$n" "$PAT" <<'PY'
import sqlite3, sys, glob
n, pat = sys.argv[1], sys.argv[2]
for db in glob.glob("gpurun_out/pmck/%s/*_results.db" % n):
    c=sqlite3.connect(db)
    try:
        rows=c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like ? group by kernel_name, counter_name", ('%'+pat+'%',)).fetchall()
        for r in rows: print(r[0].replace('(anonymous namespace)::','').replace('void ','')[:36], r[1], "%.4g per launch" % (r[2]/r[3]))
        for r in c.execute("select name, avg(end-start)/1e3 from kernels where name like ? group by name", ('%'+pat+'%',)).fetchall():
            print("   duration", r[0].replace('(anonymous namespace)::','').replace('void ','')[:36], "%.1f us" % r[1])
    except Exception as e: print("err", e)
PY
done
rm -rf gpurun_out/pmck/*/
