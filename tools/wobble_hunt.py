#!/usr/bin/env python
"""Stress detector for the LayerNorm-sum wobble (GPU box).  Runs msi_train_net (the wrapt conv-transposes) N times back to back and
compares the fixed-point LayerNorm sums of EVERY layer (workspace words, integer: must be bit-identical run to run) and the raw outputs of the
conv-transposes with the first run -- the detector reads memory only, it does not touch the kernels' code.
  python tools/wobble_hunt.py [--runs 3000] [--f16 1] [--coord 0] [--shape 4,128,256,48,16,64] [--burst 3]
Prints one line per event (layer, sample, shard, word, reference / observed value in hex, delta) and a summary line `EVENTS n of N`."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matryodshka_amd import MSI, _native as N
from oracle import nets as onets   # (weights initialiser only: test infrastructure)

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=3000)
ap.add_argument("--f16", type=int, default=1)
ap.add_argument("--coord", type=int, default=0)
ap.add_argument("--shape", default="4,128,256,48,16,64")
ap.add_argument("--burst", type=int, default=3, help="forwards queued back to back per snapshot")
ap.add_argument("--fixup", type=int, default=-1, help="FIXUP_KERNEL option (-1: plan default)")
ap.add_argument("--tag", default="")
ap.add_argument("--dump", type=int, default=-1, help="layer whose per-lane partials the dump_lanes assembly patch records (tools/asmpatch)")
ap.add_argument("--opt", action="append", default=[], help="plan option KEY=VALUE (numeric keys of include/msi_hip.h)")
a = ap.parse_args()
b, h, w, cin, nout, ngf = [int(v) for v in a.shape.split(",")]
lib = N.lib
lib.msi_debug_sums_offset.argtypes = [ctypes.c_void_p, ctypes.c_int]; lib.msi_debug_sums_offset.restype = ctypes.c_longlong
x = torch.rand((b, h, w, cin), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)) * 2 - 1
weights = onets.init_weights(cin, nout, ngf=ngf, coord_net=bool(a.coord), seed=29, randomize_affine=True)
m = MSI(weights=weights, coord_net=bool(a.coord))
if a.f16:
    m.net_options[N.NET_OPT_F32_SPLIT_F16] = 0x3ffff
if a.fixup >= 0:
    m.net_options[N.NET_OPT_FIXUP_KERNEL] = a.fixup
for kv in a.opt:
    k, v = kv.split('=')
    m.net_options[int(k)] = int(v, 0)
plan = m._plan(b, h, w, cin, nout, ngf)
key = [k for k in m._ws_cache if m._ws_cache[k][0] is plan][0]
ws = m._ws_cache[key][1]
offs = [int(lib.msi_debug_sums_offset(plan.handle, li)) for li in range(18)]
per = b * 64 * 2 * 8
kern = plan.kernels()
tlayers = [li for li in range(17) if "convt" in kern[li][0]]
print(a.tag, "conv-transpose layers:", [(li, kern[li][0].replace("(anonymous namespace)::", ""), kern[li][1], kern[li][2]) for li in tlayers], flush=True)
info = N.LayerInfo()
desc = N.NetDesc(b, h, w, cin, nout, ngf, int(a.coord), 0)
raw = {}
for li in tlayers:
    N.check(lib.msi_net_layer_info(ctypes.byref(desc), li, ctypes.byref(info)), "layer_info")
    raw[li] = (int(info.raw_offset), int(info.out_h) * int(info.out_w) * int(info.cout) * b * 4)
DUMP_OFF = 48 << 20
if a.dump >= 0:
    lib.msi_debug_partial_offset.argtypes = [ctypes.c_void_p]; lib.msi_debug_partial_offset.restype = ctypes.c_longlong
    dbase = int(lib.msi_debug_partial_offset(plan.handle)) + DUMP_OFF
    dbytes = kern[a.dump][1] * 8 * 1024
    ws[dbase:dbase + dbytes].zero_()
def snap():
    for _ in range(a.burst):
        out = m.run_net(x, nout, ngf)
    torch.cuda.synchronize()
    bits = N.c_int32(0)
    lib.msi_net_plan_status(plan.handle, ws.data_ptr(), m._stream(), N.ctypes.byref(bits))
    bits_last[0] = bits.value
    if bits.value:
        global status_seen
        status_seen[bits.value] = status_seen.get(bits.value, 0) + 1
    s = ws[offs[0]:offs[0] + 17 * per].clone().view(torch.int64).reshape(17, b, 64, 2)
    r = {li: ws[raw[li][0]:raw[li][0] + raw[li][1]].clone() for li in tlayers}
    if a.dump >= 0:
        r['dump'] = ws[dbase:dbase + dbytes].clone().view(torch.int32).reshape(-1, 2, 4, 4, 64)   # [workgroup][G][wave][s1 | s2 | cnt | -][lane]
    return out.clone(), s, r
status_seen = {}
bits_last = [0]
ro, rs, rr = snap()
ev = 0; out_only = 0
for it in range(a.runs):
    o, s, r = snap()
    # integer sums: the TOTAL over the 64 shards of a (layer, sample, word) must be bit-identical run to run; which shard a tile's share lands in depends on
    # which K-range workgroup arrives last when split tiles are summed inside the launch, so single shards are only compared when the totals differ
    if not torch.equal(s.sum(2), rs.sum(2)):
        ev += 1
        idx = torch.nonzero(s != rs).tolist() if a.fixup == 1 else [[li_, b_, -1, w_] for (li_, b_, w_) in torch.nonzero(s.sum(2) != rs.sum(2)).tolist()]
        for (li, bi, sh, wd) in idx[:6]:
            ref, now = (int(rs[li, bi, sh, wd]), int(s[li, bi, sh, wd])) if sh >= 0 else (int(rs[li, bi, :, wd].sum()), int(s[li, bi, :, wd].sum()))
            print("%s EVENT run %d layer %d (%s) sample %d shard %d (wave %d) word %d: ref %016x now %016x delta %d  raw-out equal: %s  out equal: %s" % (
                a.tag, it, li, kern[li][0].replace("(anonymous namespace)::", "")[:40], bi, sh, sh & 3, wd, ref & (2**64 - 1), now & (2**64 - 1), now - ref,
                bool(torch.equal(r[li], rr[li])) if li in r else None, bool(torch.equal(o, ro))), "status", hex(bits_last[0]), flush=True)
        if a.dump >= 0:
            d = torch.nonzero(r['dump'] != rr['dump']).tolist()
            print("%s   dump records differing: %d" % (a.tag, len(d)), flush=True)
            for (bid, G, wv, q, ln) in d[:24]:
                fr = rr['dump'][bid, G, wv, q, ln].view(torch.float32).item(); fn = r['dump'][bid, G, wv, q, ln].view(torch.float32).item()
                print("%s     workgroup %d group %d wave %d (shard %d) %s lane %d: ref %.9g now %.9g" % (a.tag, bid, G, wv, (bid * 4 + wv) & 63, ("s1", "s2", "cnt", "?")[q], ln, fr, fn), flush=True)
                if q == 1 and bid < 512:   # which term is missing?  (whole tiles of layer 13 at the default shape: 45 x 2 tiles per (sample, row parity), XCD remap with n_main = 512)
                    Hh, Ww, C = h // 2, w // 2, 128
                    y = rr[a.dump].view(torch.float32).reshape(b, Hh, Ww, C)
                    t = (bid & 7) * 64 + (bid >> 3)
                    ph = t & 1; r_ = t >> 1; tile_m = r_ % 45; r_ //= 45; tile_n = r_ % 2; bb = r_ // 2
                    tyi, txi = tile_m // 5, tile_m % 5
                    wm, wn = wv >> 1, wv & 1
                    def pix(lane, pw):
                        fr_, hf = lane & 31, lane >> 5
                        local = wm * 32 + fr_; row = local >> 4; col = (local & 15) ^ ((row & 1) * 8)
                        mh, mw = tyi * 4 + row, txi * 16 + col
                        return 2 * mh + ph - 1, 2 * mw + pw - 5, tile_n * 64 + wn * 32 + 4 * hf
                    for pw in (0, 1):
                        o0 = pix(0, pw); ol = pix(ln, pw)
                        if not (0 <= o0[0] < Hh and 0 <= o0[1] < Ww and 0 <= ol[0] < Hh and 0 <= ol[1] < Ww):
                            print("%s        pw %d: pivot or lane pixel outside the stored crop" % (a.tag, pw)); continue
                        P = y[bb, o0[0], o0[1], o0[2]].item()
                        vals = [y[bb, ol[0], ol[1], ol[2] + 8 * g + e].item() for g in range(4) for e in range(4)]
                        d2 = [(v - P) ** 2 for v in vals]
                        print("%s        pw %d: recomputed lane s2 %.7g (ref %.7g)  lost %.6g;  d^2 terms: %s" % (a.tag, pw, sum(d2), fr, fr - fn, " ".join("%.5g" % v for v in d2)), flush=True)
    elif not torch.equal(o, ro):
        out_only += 1
print("%s EVENTS %d of %d snapshots (%d forwards); output-only differences %d; status words seen %s" % (a.tag, ev, a.runs, a.runs * a.burst, out_only, {hex(k): v for k, v in status_seen.items()}), flush=True)
