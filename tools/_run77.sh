cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02W
timeout 900 python bench.py > gpurun_out/r02W/bench.json 2> gpurun_out/r02W/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r02W/bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['repeats']['ms_per_step'], {k:(v.get('ms')) for k,v in j['stages'].items()}, j['parity_max_abs_vs_oracle'])
PY
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02W/prof -o trace -- python bench.py --steps 20 --warmup 5 --repeats 0 --no-cpu-baseline --prewarm 0 > gpurun_out/r02W/prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/r02W/prof/trace_results.db --frames 34 > gpurun_out/r02W/bench_kernel_stats.txt; tail -2 gpurun_out/r02W/bench_kernel_stats.txt | cut -c1-175
