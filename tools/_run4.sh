cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02d/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02d/tests.log
tail -25 gpurun_out/r02d/tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02d/bench.json 2> gpurun_out/r02d/bench.err; tail -3 gpurun_out/r02d/bench.err
python -c "
import json; j=json.loads(open('gpurun_out/r02d/bench.json').read().strip().split('\n')[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['repeats']['ms_per_step'])
for k,v in j['stages'].items(): print(k, v)
print(j['parity_max_abs_vs_oracle'])"
timeout 300 python bench.py --config 2 --steps 10 --warmup 3 --repeats 1 > gpurun_out/r02d/bench_c2.json 2>/dev/null
python -c "
import json; j=json.loads(open('gpurun_out/r02d/bench_c2.json').read().strip().split('\n')[-1])
print('config2', j['value'], j['ms_per_step'], {k:v['ms'] for k,v in j['stages'].items()})"
