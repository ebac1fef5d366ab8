#!/bin/bash
# tools/variants_ktime.sh CONFIG "PAT1|PAT2" [ROUNDS] ["extra bench args"] (on the GPU box): average duration (rocprofv3 --kernel-trace) of the kernels whose name
# matches one of the |-separated substrings, with every tools/_variants/libmsi_*.so installed in turn, ROUNDS interleaved rounds.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
CFG=${1:-1}; PATS=${2:-ods_sweep}; ROUNDS=${3:-2}; EXTRA=${4:-}
cp matryodshka_amd/libmsi_hip.so /tmp/libmsi_saved.so
for r in $(seq $ROUNDS); do
  for v in tools/_variants/libmsi_*.so; do
    cp "$v" matryodshka_amd/libmsi_hip.so
    rm -rf /tmp/vkt; rocprofv3 --kernel-trace -d /tmp/vkt -o t -- python bench.py --config $CFG $EXTRA --steps 10 --warmup 3 --repeats 0 --no-cpu-baseline --no-sustained-probe --prewarm 0.5 --strong-frames 0 --no-settle > /tmp/vkt.json 2>/dev/null
    python - "$v" "$PATS" "$r" <<'PY'
import sqlite3, glob, sys, json
v, pats, r = sys.argv[1], sys.argv[2].split("|"), sys.argv[3]
c = sqlite3.connect(glob.glob("/tmp/vkt/*_results.db")[0])
try:
    j = json.loads(open("/tmp/vkt.json").read().strip().splitlines()[-1]); fps = j["value"]
except Exception:
    fps = None
for p in pats:
    for row in c.execute("select name,count(*),avg(end-start)/1e3,min(end-start)/1e3 from kernels where name like ? group by name", ("%" + p + "%",)).fetchall():
        print("r%s %-34s %-52s n=%4d avg %9.2f us min %9.2f  (%s under the profiler)" % (r, v.split("/")[-1], row[0].replace("(anonymous namespace)::", "").replace("void ", "")[:52], row[1], row[2], row[3], fps))
PY
  done
done
cp /tmp/libmsi_saved.so matryodshka_amd/libmsi_hip.so
