#!/usr/bin/env python
"""Turns a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`, which
writes DIR/NAME_results.db on ROCm 7.2) into the text summary kept under profiles/:
per-kernel totals (the --stats table) and, for the conv kernel, one line per layer of the
last frame with its achieved TFLOP/s.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db [--frames N] > profiles/rNN_kernel_stats.txt
"""
import argparse
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--frames", type=int, default=0, help="frames in the profiled run (warmup + steps), for per-frame totals")
    args = ap.parse_args()
    c = sqlite3.connect(args.db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace --stats summary of %s" % args.db)
    print("%-100s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for r in rows[:24]:
        print("%-100s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
    if args.frames:
        print("\n# per frame (%d frames): " % args.frames + ", ".join(
            "%s %.1f us" % (r[0].split("(")[0].split("::")[-1][:40], r[2] / 1e3 / args.frames) for r in rows[:8]))
    conv = c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, accum_vgpr_count "
                     "from kernels where name like '%conv_igemm%' or name like '%conv_halo%' or name like '%convt_halo%' order by start").fetchall()
    if len(conv) >= 18:
        print("\n# conv launches of the last frame (graph order; tile<BM, BN, MODE, BF16> = conv_igemm_kernel, "
              "halo<RATE, APPLY> = conv_halo_kernel)")
        names = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2",
                 "conv4_3", "conv6_1", "conv6_2", "conv6_3", "conv7_1", "conv7_2", "conv8_1", "conv8_2", "color_pred"]
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        flops = bench.cnn_layer_flops(320, 640, 192, 64, 64, True)      # BASELINE config
        fix = c.execute("select start, end from kernels where name like '%conv_fixup%' order by start").fetchall()
        tot_us = 0.0
        # the 1x1 head is a conv_igemm launch (template MODE 2) only on the unfused path; on the fp32 blend_psv path it is
        # part of head_assemble_kernel and a frame has 17 conv launches
        fused_tail = "conv_halo" in conv[-1][0] or "convt_halo" in conv[-1][0] or ("Li2E" not in conv[-1][0] and ", 2, " not in conv[-1][0])
        nl = 17 if fused_tail else 18
        ha = c.execute("select start, end from kernels where name like '%head_assemble%' order by start").fetchall()
        for nm, r, fl in zip(names[:nl], conv[-nl:], flops[:nl]):
            tmpl = r[0].split("<")[1].split(">")[0] if "<" in r[0] else "?"
            us = (r[2] - r[1]) / 1e3
            # a fix-up launch (tail split) directly follows its conv launch
            fx = [(e - s0) / 1e3 for s0, e in fix if r[2] <= s0 < r[2] + 200000]
            nxt = [q[1] for q in conv if q[1] > r[1]]
            fx = [f for (s0, e), f in zip([x for x in fix if r[2] <= x[0] < r[2] + 200000], fx) if not nxt or s0 < nxt[0]]
            fus = sum(fx[:1])
            tot_us += us + fus
            kind = "halo" if "conv_halo" in r[0] else ("halo_convT" if "convt_halo" in r[0] else "tile")
            print("%-10s %s<%s> blocks=%d lds=%d vgpr=%d agpr=%d  %8.1f us + fixup %5.1f us  %6.1f TFLOP/s (%4.1f%%, BASELINE shapes)" % (
                nm, kind, tmpl, (r[3] // r[6]) * r[4] * r[5], r[7], r[8], r[9], us, fus, fl / (us + fus) / 1e6,
                100 * fl / (us + fus) / 1e6 / bench.PEAK_FP32_MFMA_TFLOPS))
        if fused_tail and ha:
            print("color_pred fused with the RGBA assembly: head_assemble_kernel %8.1f us (HBM-bound; not a conv_igemm launch)" % ((ha[-1][1] - ha[-1][0]) / 1e3))
        print("sum %.1f us -> %.1f TFLOP/s (conv launches listed above)" % (tot_us, sum(flops[:nl]) / tot_us / 1e6))


if __name__ == "__main__":
    sys.exit(main())
