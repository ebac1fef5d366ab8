#!/bin/bash
# tools/split3_timing.sh (on the GPU box; needs tools/_variants/libmsi_timing.so = a -DMSI_CONV_TIMING build): the speed half of the
# 6-product bf16 split study with the kernels that exist.  At configs[1] shapes (640x320, D = 32, ngf 64, batch 1):
#   (1) the fp32 network and the bf16 network, HIP-event time per forward (tools/bench_cnn.py);
#   (2) per layer of the bf16 plan: prologue / k-loop / epilogue shader-clock ticks per workgroup (tools/conv_timing.py).
# A 6-product split run as a K-expansion on these kernels (x' = [h, m, l, h, m, h] against w' = [h, h, h, m, m, l]) repeats every
# k-step six times and keeps the prologue / epilogue: lower bound of its launch time = launch x (pro + 6 loop + epi) / (pro + loop + epi).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
python tools/bench_cnn.py --dtype f32 --steps 30
python tools/bench_cnn.py --dtype bf16 --steps 30
rm -rf /tmp/s3; rocprofv3 --kernel-trace -d /tmp/s3 -o t -- python tools/bench_cnn.py --dtype bf16 --steps 10 > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
c = sqlite3.connect(glob.glob("/tmp/s3/*_results.db")[0])
print("# bf16 plan at batch 1, D = 32: kernels per forward")
tot = 0.0
for name, n, avg in c.execute("select name, count(*), avg(end-start)/1e3 from kernels where name like '%conv%' or name like '%ln_apply%' group by name order by sum(end-start) desc").fetchall():
    per = n / 13.0
    tot += per * avg
    print("  %-70s x%.1f  avg %8.2f us" % (name.replace("(anonymous namespace)::", "").replace("void ", "")[:70], per, avg))
print("  sum %.1f us per forward (13 forwards profiled)" % tot)
PY
cp matryodshka_amd/libmsi_hip.so /tmp/libmsi_saved.so
cp tools/_variants/libmsi_timing.so matryodshka_amd/libmsi_hip.so
python tools/conv_timing.py --bf16 --batch 1 --planes 32 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 2>&1 | grep "per block" | cut -c1-200
cp /tmp/libmsi_saved.so matryodshka_amd/libmsi_hip.so
