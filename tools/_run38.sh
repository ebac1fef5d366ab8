cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02I
( time timeout 900 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -6
( time timeout 900 python bench.py > gpurun_out/r02I/bench.json 2> gpurun_out/r02I/bench.err ) 2>&1 | tail -4
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r02I/bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline'])
PY
