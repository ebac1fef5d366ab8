#!/bin/bash
# tools/evidence_run.sh TAG COMMIT (GPU box): the round's evidence set in one call -- per configuration tools/profile_run.sh (bench line, rocprofv3 kernel table,
# PMC HBM traffic stamped with the kernel-source hash) into gpurun_out/TAG/<name>/, then the bench lines once more with the fresh traffic files in place
# (traffic_stale: false) and the full GPU test log.  Copy what is to be judged into profiles/ (tools/evidence_collect.py).
TAG=$1; COMMIT=$2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
tools/profile_run.sh $TAG/config1 1 $COMMIT
tools/profile_run.sh $TAG/config1_wrapnet 1 $COMMIT --no-coord-net
tools/profile_run.sh $TAG/config2 2 $COMMIT
tools/profile_run.sh $TAG/config3 3 $COMMIT
tools/profile_run.sh $TAG/config4 4 $COMMIT
# second pass of the bench lines with this call's traffic files installed where bench.py looks for them
for n in config1 config1_wrapnet config2 config3 config4; do cp gpurun_out/$TAG/$n/hbm_traffic.json profiles/${TAG}_hbm_traffic_$n.json; done
python bench.py --config 1 > gpurun_out/$TAG/config1/bench.json 2>/dev/null
python bench.py --config 1 --no-coord-net > gpurun_out/$TAG/config1_wrapnet/bench.json 2>/dev/null
python bench.py --config 2 > gpurun_out/$TAG/config2/bench.json 2>/dev/null
python bench.py --config 3 > gpurun_out/$TAG/config3/bench.json 2>/dev/null
python bench.py --config 4 > gpurun_out/$TAG/config4/bench.json 2>/dev/null
for n in config1 config1_wrapnet config2 config3 config4; do python -c "
import json,sys
j=json.loads(open('gpurun_out/$TAG/$n/bench.json').read().strip().splitlines()[-1])
print('$n', j['value'], j['unit'], j['ms_per_step'], 'roofline', j['roofline']['frac'], j['roofline']['ms_per_forward'], 'stale', j['roofline'].get('traffic_stale'), 'ratio', j['roofline'].get('traffic_ratio'), 'alt', (j.get('alt_arithmetic') or {}).get('value'), 'cpu', (j.get('cpu_baseline') or {}).get('value'))"; done
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/$TAG/pytest_gpu.log; cat gpurun_out/$TAG/pytest_gpu.log
