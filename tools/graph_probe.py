#!/usr/bin/env python
"""Probe (GPU box): one configs[1] frame (preprocess -> sweep -> network -> render -> deprocess, 25 launches) captured into a HIP graph
(torch.cuda.CUDAGraph: the library launches on torch's current stream, which is the capture stream inside the context) and replayed,
against the eager loop of bench.py.  Prints ms per frame for both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from matryodshka_amd import MSI
from matryodshka_amd.synthetic import make_inputs
from oracle import nets as onets   # (weights initialiser only)
H, W, D, NGF = 320, 640, 32, 64
dev = torch.device("cuda:0")
weights = onets.init_weights(6 * D, 2 * D, ngf=NGF, coord_net=True, seed=1)
m = MSI(weights=weights, coord_net=True, device=dev)
planes = m.inv_depths(1.0, 100.0, D)
inp = make_inputs(5, 1, H, W)
g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
src_u8, ref_u8 = g(inp["src_image"]), g(inp["ref_image"])
ref_pose, src_pose, intr = g(inp["ref_pose"]), g(inp["src_pose"]), g(inp["intrinsics"])
ref_pose_inv = g(np.linalg.inv(inp["ref_pose"].astype(np.float64)).astype(np.float32))
tgt_pose_rt, tgt_pos = g(inp["tgt_pose_rt"]), g(inp["tgt_pos"])
def frame():
    src, ref = m.preprocess_image_pair(src_u8, ref_u8)
    x = m.format_network_input(ref, src, ref_pose, src_pose, planes, intr, ref_pose_inv=ref_pose_inv)
    out = m.infer_layers(x, D, NGF)
    rgb, dep = m.msi_render_equirect_view_and_depth(out["rgba_layers"], tgt_pose_rt, tgt_pos, planes, intr)
    return m.deprocess_image_and_depth(rgb, dep)
for _ in range(20):
    ref_out = frame()
torch.cuda.synchronize()
ref_rgb = ref_out[0].clone()
def timed(fn, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager : %.4f ms per frame" % timed(frame))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): frame()
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(gr):
        out = frame()
    for _ in range(10): gr.replay()
    torch.cuda.synchronize()
    print("graph : %.4f ms per frame; output equal to eager: %s" % (timed(gr.replay), bool(torch.equal(out[0], ref_rgb))))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300])
