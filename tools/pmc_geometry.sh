#!/bin/bash
# tools/pmc_geometry.sh (on the GPU box): SQ / TA counters of the sweep and render kernels at configs[1], three separate --pmc passes with
# --kernel-trace only.  r03: ods_sweep_kernel 699 VALU instructions per wave (4 samples), SQ_ACTIVE_INST_VALU x 4 = 145k of ~155k cycles per SIMD.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/pmc
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TA_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" | sort -u | tr '\n' ' ' | head -c 6000 > gpurun_out/pmc/counters.txt
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS" "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE SQ_WAVES"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc/$n -o p -- python bench.py --steps 3 --warmup 1 --repeats 0 --no-cpu-baseline --no-sustained-probe --prewarm 0 --strong-frames 0 > gpurun_out/pmc/$n.log 2>&1
  python - "$n" <<'PY'
import sqlite3, sys, glob
n=sys.argv[1]
for db in glob.glob("gpurun_out/pmc/%s/*_results.db" % n):
    c=sqlite3.connect(db)
    try:
        rows=c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like '%ods_sweep%' or kernel_name like '%render_kernel%' group by kernel_name, counter_name").fetchall()
        for r in rows: print(r[0][:40].replace('(anonymous namespace)::',''), r[1], "%.4g per launch" % (r[2]/r[3]))
    except Exception as e: print("err", e)
PY
done
rm -rf gpurun_out/pmc/*/
