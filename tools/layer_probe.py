"""tools/layer_probe.py (GPU box): run_net on uniform noise (sizes HxW[,HxW...] and repetitions from the command line), every layer's raw output against the oracle, several weight seeds -- pixels whose worst channel is off by
more than 1e-3 of the layer's scale are counted per layer.  The probe that exposed the store hazard of profiles/r05_store_hazard.txt ("the lost stores")."""
import numpy as np, sys
sys.path.insert(0,'.')
import torch
from tests.test_gpu_cnn import _run
from matryodshka_amd import MSI, nets, _native as N
from oracle import nets as onets
env=(torch, MSI, nets, N, onets)
import itertools
sizes = [(160, 320), (320, 640), (64, 128)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split('x')) for a in sys.argv[1].split(',')]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for (h, w), rep in itertools.product(sizes, range(reps)):
    pred, ref, raws, acts = _run(env, 1, h, w, 96, 32, 64, True, seed=5+rep, options={})
    res=[]; worst = 0.0
    for name, raw in raws.items():
        o=acts[name]; sc=np.abs(o).max()+1e-12
        d=np.abs(raw-o).max(axis=(0,3))/sc
        worst = max(worst, float(d.max()))
        if name in ('conv1_1','conv1_2','conv2_1') or (d>1e-3).any(): res.append('%s bad px %d'%(name,(d>1e-3).sum()))
    res.append('worst layer err %.2g' % worst)
    print('%dx%d' % (h, w), 'rep',rep,'pred err %.3g'%np.abs(pred-ref).max(),' | '.join(res), flush=True)
