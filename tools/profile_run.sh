#!/bin/bash
# tools/profile_run.sh TAG CONFIG COMMIT [bench args ...]   (run on the GPU box through gpurun, from the repo root)
# One configuration's evidence set under gpurun_out/TAG/ -- copy what is to be judged into profiles/:
#   bench.json           the bench line (python bench.py --config CONFIG [args])
#   kernel_stats.txt     rocprofv3 --kernel-trace --stats of the same command, summarised by tools/rocprof_summary.py
#                        (per-kernel table, per-step totals, per-layer conv table against the MFMA peak)
#   hbm_traffic.json     HBM bytes per step and kernel: --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes with
#                        --kernel-trace only (tools/hbm_traffic.py; stamped with the kernel-source hash + COMMIT)
# Set SKIP_PMC=1 to leave the two counter passes out.
set -u
TAG=$1; CONFIG=$2; COMMIT=$3; shift 3
EXTRA="$*"
SUMARGS="--config $CONFIG"
case " $EXTRA " in *" --no-coord-net "*) SUMARGS="$SUMARGS --no-coord-net";; esac
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python bench.py --config "$CONFIG" $EXTRA > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; }
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bench:", j["value"], j["unit"], j["ms_per_step"], "ms/step  roofline", j["roofline"]["frac"], j["roofline"]["ms_per_forward"],
          "stages", {k: v.get("ms") for k, v in j["stages"].items()}, "parity", j.get("parity_max_abs_vs_oracle"))
except Exception as e:
    print("bench.json unreadable:", e)
PY
STEPS=20; [ "$CONFIG" != "1" ] && STEPS=6
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace -- python bench.py --config "$CONFIG" $EXTRA --steps $STEPS --warmup 3 --repeats 0 \
  --no-cpu-baseline --no-sustained-probe --prewarm 0 --strong-frames 0 --no-settle > "$OUT/prof.log" 2>&1
python tools/rocprof_summary.py "$OUT/prof/trace_results.db" $SUMARGS > "$OUT/kernel_stats.txt" 2> "$OUT/summary.err" || tail -3 "$OUT/summary.err"
tail -24 "$OUT/kernel_stats.txt" | cut -c1-170
if [ "${SKIP_PMC:-0}" != "1" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$C" -o pmc -- python bench.py --config "$CONFIG" $EXTRA --steps 3 --warmup 1 --repeats 0 \
      --no-cpu-baseline --no-sustained-probe --prewarm 0 --strong-frames 0 --no-settle > "$OUT/pmc_$C.log" 2>&1
  done
  python tools/hbm_traffic.py "$OUT/pmc_FETCH_SIZE/pmc_results.db" "$OUT/pmc_WRITE_SIZE/pmc_results.db" $SUMARGS --commit "$COMMIT" \
    > "$OUT/hbm_traffic.json" 2> "$OUT/traffic.err" || tail -3 "$OUT/traffic.err"
  python - "$OUT/hbm_traffic.json" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    conv = sum(v["hbm_bytes"] for k, v in j["kernels"].items() if k.startswith("conv"))
    print("hbm traffic of the conv launches per step: %.3f GB" % (conv / 1e9), {k: round(v["hbm_bytes"] / 1e6, 1) for k, v in j["kernels"].items() if v["hbm_bytes"] > 5e6})
except Exception as e:
    print("hbm_traffic.json unreadable:", e)
PY
fi
# the rocpd databases are large; keep the summaries
rm -rf "$OUT/prof" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE
