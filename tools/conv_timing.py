#!/usr/bin/env python
"""Per-workgroup phase timing of one conv layer (needs a library built with -DMSI_CONV_TIMING,
e.g. tools/_variants/libmsi_timing.so installed as matryodshka_amd/libmsi_hip.so):
s_memtime stamps at kernel entry, k-loop start, k-loop end and exit of every workgroup.

    python tools/conv_timing.py [--bf16] [--split3] [--bf16x6] [--batch N] [--planes D] [layer ...]
(--bf16: the bf16 plan's conv_halo_bf16_kernel layers at batch N; stamps are 100 MHz s_memtime ticks, 10 ns each)
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from matryodshka_amd import MSI, nets, _native

lib = _native.lib
lib.msi_debug_conv_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.msi_debug_conv_timing.restype = None
names = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3",
         "conv6_1", "conv6_2", "conv6_3", "conv7_1", "conv7_2", "conv8_1", "conv8_2", "color_pred"]
argv = sys.argv[1:]
bf16 = "--bf16" in argv
batch = int(argv[argv.index("--batch") + 1]) if "--batch" in argv else 1
planes = int(argv[argv.index("--planes") + 1]) if "--planes" in argv else (64 if bf16 else 32)
split3 = "--split3" in argv
bf16x6 = "--bf16x6" in argv     # the six-product bf16 form of the split (plan option F32_SPLIT_F16 = 0)
argv = [a for i, a in enumerate(argv) if a not in ("--bf16", "--batch", "--planes", "--split3", "--bf16x6") and (i == 0 or argv[i - 1] not in ("--batch", "--planes"))]
layers = [int(a) for a in argv] or [0, 1, 4, 7, 16]
m = MSI(weights=nets.init_weights(6 * planes, 2 * planes, 64, True), coord_net=True, dtype="bf16" if bf16 else "f32")
if split3:
    m.net_options[_native.NET_OPT_F32_SPLIT3] = 0x3ffff
if bf16x6:
    m.net_options[_native.NET_OPT_F32_SPLIT_F16] = 0
x = torch.rand((batch, 320, 640, 6 * planes), device="cuda") * 2 - 1
if bf16:
    x = x.to(torch.bfloat16)
for _ in range(3):
    m.run_net(x, 2 * planes, 64)
buf = torch.zeros((16384, 24), dtype=torch.int64, device="cuda")
for li in layers:
    buf.zero_()
    lib.msi_debug_conv_timing(ctypes.c_void_p(buf.data_ptr()), li)
    m.run_net(x, 2 * planes, 64)
    torch.cuda.synchronize()
    lib.msi_debug_conv_timing(None, -1)
    t = buf.cpu().numpy().astype(np.int64)
    t = t[t[:, 0] != 0]
    if len(t) == 0:
        print("%-10s no stamps (kernel without them)" % names[li])
        continue
    pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    span = t[:, 3].max() - t[:, 0].min()
    dur = t[:, 3] - t[:, 0]
    print("%-10s blocks %5d  span %8d ticks | per block: prologue %6.0f  loop %8.0f  epilogue %6.0f  total %8.0f (min %d max %d) | sum(block time)/span = %.2f resident blocks" % (
        names[li], len(t), span, pro.mean(), loop.mean(), epi.mean(), dur.mean(), dur.min(), dur.max(), dur.sum() / span))
    if t[:, 22].any() and t[:, 23].any():   # (r06) s_memrealtime at entry / exit: the constant 100 MHz counter -> the clock s_memtime advanced at under THIS kernel
        # per CU (s_memtime is a per-XCD counter: spans are only meaningful within one CU): ticks between the CU's first entry and last exit over the same interval in real time
        hw_, xcc_ = t[:, 4], t[:, 5] & 0xf
        key_ = (xcc_ << 16) | (((hw_ >> 13) & 0x7) << 12) | (((hw_ >> 12) & 0x1) << 8) | ((hw_ >> 8) & 0xf)
        ghz, real_us = [], []
        for k_ in np.unique(key_):
            b_ = t[key_ == k_]
            rt = (b_[:, 23].max() - b_[:, 22].min()) / 100e6
            if rt > 2e-6:
                ghz.append((b_[:, 3].max() - b_[:, 0].min()) / rt / 1e9)
                real_us.append(rt * 1e6)
        if ghz:
            print("           launch span %.1f us by s_memrealtime (100 MHz) -> s_memtime advanced at %.3f GHz (median of %d CUs; min %.3f max %.3f); k-loop share of a block's ticks %.0f %%" % (
                (t[:, 23].max() - t[:, 22].min()) / 100.0, float(np.median(ghz)), len(ghz), min(ghz), max(ghz), 100.0 * loop.sum() / dur.sum()))
    if t[:, 6].any() and t[:, 10].any():   # stamps inside the prologue of conv_halo_bf16_kernel
        e = t[t[:, 6] != 0]
        ln = (e[:, 8] - e[:, 7]).mean() if e[:, 8].any() else 0.0
        tb = (e[:, 9] - e[:, 8]).mean() if e[:, 8].any() else (e[:, 9] - e[:, 7]).mean()
        print("           prologue: index setup %6.0f | patch + weight requests %6.0f | statistics %6.0f | affine table %6.0f | patch arrives %6.0f | patch -> LDS + barrier %6.0f" % (
            (e[:, 6] - e[:, 0]).mean(), (e[:, 7] - e[:, 6]).mean(), ln, tb, (e[:, 10] - e[:, 9]).mean(), (e[:, 1] - e[:, 10]).mean()))
    t = np.concatenate([t[:, :6], t[:, 16:22]], axis=1)
    if t[:, 6].any():   # stamps inside the epilogue (whole-tile path of emit_tile_impl): entry, stores issued, wave sums, atomics
        e = t[t[:, 6] != 0]
        print("           epilogue: values -> LDS strip written +%6.0f (from entry)" % ((e[:, 11] - e[:, 6]).mean()))
        print("           epilogue: loop end -> entry %6.0f | element loop (stores issued) %6.0f | wave sums %6.0f | atomics %6.0f | -> exit stamp %6.0f" % (
            (e[:, 6] - e[:, 2]).mean(), (e[:, 7] - e[:, 6]).mean(), (e[:, 8] - e[:, 7]).mean(), (e[:, 9] - e[:, 8]).mean(), (e[:, 3] - e[:, 9]).mean()))
    # per CU (XCC_ID, HW_ID se/sh/cu): time-average number of resident workgroups and of workgroups inside the k-loop
    hw, xcc = t[:, 4], t[:, 5] & 0xf
    cu_key = (xcc << 16) | (((hw >> 13) & 0x7) << 12) | (((hw >> 12) & 0x1) << 8) | ((hw >> 8) & 0xf)   # se[15:13] sh[12] cu[11:8]
    res, inl, spans, nblk = [], [], [], []
    for k in np.unique(cu_key):
        b = t[cu_key == k]
        sp = b[:, 3].max() - b[:, 0].min()
        res.append((b[:, 3] - b[:, 0]).sum() / sp)
        inl.append((b[:, 2] - b[:, 1]).sum() / sp)
        spans.append(sp)
        nblk.append(len(b))
    print("           %d CUs seen; blocks per CU %d..%d; per-CU span %.0f..%.0f ticks; resident workgroups %.2f (avg), in k-loop %.2f (avg)" % (
        len(res), min(nblk), max(nblk), min(spans), max(spans), np.mean(res), np.mean(inl)))
    # (r06) lockstep check: share of each CU's span during which exactly n of its workgroups are inside their k-loop (n = 0: the matrix pipes have nothing to issue),
    # and how spread the entry times of the workgroups that share a CU's first "round" are
    hist = np.zeros(8)
    first_spread = []
    for k in np.unique(cu_key):
        b = t[cu_key == k]
        ev = sorted([(tt_, 1) for tt_ in b[:, 1]] + [(tt_, -1) for tt_ in b[:, 2]])
        lo, hi = b[:, 0].min(), b[:, 3].max()
        cur, last = 0, lo
        for tt_, dlt in ev:
            hist[min(cur, 7)] += tt_ - last
            last = tt_; cur += dlt
        hist[min(cur, 7)] += hi - last
        st = np.sort(b[:, 0])
        nres = max(1, int(round((b[:, 3] - b[:, 0]).sum() / (hi - lo))))
        first_spread.append(st[min(nres, len(st)) - 1] - st[0])
    hist /= hist.sum()
    print("           share of CU time with n workgroups in their k-loop: " + "  ".join("n=%d %.1f%%" % (n, 100 * hist[n]) for n in range(5)) +
          " | entry spread of a CU's first round: median %.0f ticks" % np.median(first_spread))
lib.msi_debug_conv_occupancy.argtypes = [ctypes.c_int]
lib.msi_debug_conv_occupancy.restype = ctypes.c_int
for lds in (32768, 32784, 40960, 49152):
    print("occupancy API: %d B dynamic LDS -> %d workgroups per CU" % (lds, lib.msi_debug_conv_occupancy(lds)))
