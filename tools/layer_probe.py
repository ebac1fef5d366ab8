"""tools/layer_probe.py (GPU box): run_net on uniform noise at 160 x 320, every layer's raw output against the oracle, several weight seeds -- pixels whose worst channel is off by
more than 1e-3 of the layer's scale are counted per layer.  The probe that exposed the store hazard of profiles/r05_store_hazard.txt ("the lost stores")."""
import numpy as np, sys
sys.path.insert(0,'.')
import torch
from tests.test_gpu_cnn import _run
from matryodshka_amd import MSI, nets, _native as N
from oracle import nets as onets
env=(torch, MSI, nets, N, onets)
h,w=160,320
for rep in range(4):
    pred, ref, raws, acts = _run(env, 1, h, w, 96, 32, 64, True, seed=5+rep, options={})
    res=[]
    for name in ('conv1_1','conv1_2','conv2_1'):
        raw=raws[name]; o=acts[name]; sc=np.abs(o).max()+1e-12
        d=np.abs(raw-o).max(axis=(0,3))/sc
        res.append('%s bad px %d'%(name,(d>1e-3).sum()))
    print('rep',rep,'pred err %.3g'%np.abs(pred-ref).max(),' | '.join(res), flush=True)
