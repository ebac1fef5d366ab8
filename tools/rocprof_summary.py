#!/usr/bin/env python
"""Turns a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`, which
writes DIR/NAME_results.db on ROCm 7.2) into the text summary kept under profiles/:
per-kernel totals (the --stats table), the per-step totals -- the number of profiled steps is COUNTED from the
render launches (one `render_kernel` / `mpi_render_kernel` launch per bench step), never passed in -- and one line
per conv layer of the last step with its achieved TFLOP/s against the MFMA peak of the configuration's compute type.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db [--config 1|2|3|4] [--no-coord-net] > profiles/rNN_kernel_stats.txt
"""
import argparse
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NAMES = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2",
         "conv4_3", "conv6_1", "conv6_2", "conv6_3", "conv7_1", "conv7_2", "conv8_1", "conv8_2", "color_pred"]


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--config", type=int, default=1, help="bench.py --config the profile was taken with (shapes / dtype of the per-layer table)")
    ap.add_argument("--no-coord-net", action="store_true")
    ap.add_argument("--batch", type=int, default=0, help="frames per step of the profiled run (default: the config's single-GPU batch)")
    args = ap.parse_args()
    import bench
    cfg = bench.CONFIGS[args.config]
    batch = args.batch or (cfg["per_rank"] or cfg["total"])
    coord = not args.no_coord_net
    c = sqlite3.connect(args.db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    steps = sum(r[1] for r in rows if "render_kernel" in r[0] and "mpi_render" not in r[0]) or \
        sum(r[1] for r in rows if "mpi_render" in r[0])
    print("# rocprofv3 --kernel-trace --stats summary of %s" % args.db)
    print("# %s%s; %d steps of %d frame(s) counted from the render launches" % (
        cfg["name"], "" if coord else ", wrap-pad net (msi_train_net)", steps, batch))
    print("%-100s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for r in rows[:24]:
        print("%-100s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
    if steps:
        print("\n# per step (%d steps, %d frame(s) each; launches per step in brackets):" % (steps, batch))
        for r in rows[:12]:
            print("#   %-60s %9.1f us  [%g]" % (short(r[0])[:60], r[2] / 1e3 / steps, round(r[1] / steps, 2)))
    # device-side gaps between consecutive kernels of the LAST step (end of one launch -> start of the next, same queue)
    ks = c.execute("select name, start, end from kernels order by start").fetchall()
    if steps and len(ks) > 4:
        starts = [i for i, k in enumerate(ks) if "preprocess" in k[0]]
        # (a step starts with its preprocess launch -- two of them on the PP path -- and ends with the trace)
        first = starts[-1] if starts else None
        if first is not None and first > 0 and "preprocess" in ks[first - 1][0]:
            first -= 1
        per = len(ks) - first if first is not None else 0
        if per > 4:
            last_step = ks[first:]
            gaps = [max(0, b[1] - a[2]) / 1e3 for a, b in zip(last_step[:-1], last_step[1:])]
            busy = sum(k[2] - k[1] for k in last_step) / 1e3
            print("\n# last step: %d launches, %.1f us inside kernels, %.1f us of gaps between consecutive launches (mean %.2f, max %.2f us) "
                  "-- under the profiler" % (per, busy, sum(gaps), sum(gaps) / len(gaps), max(gaps)))
    conv = c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, accum_vgpr_count "
                     "from kernels where name like '%conv_igemm%' or name like '%conv_halo%' or name like '%convt_halo%' "
                     "or name like '%conv_s2_%' order by start").fetchall()
    if len(conv) >= 17:
        bf16 = cfg["dtype"] == "bf16"
        peak = bench.PEAK_BF16_MFMA_TFLOPS if bf16 else bench.PEAK_FP32_MFMA_TFLOPS
        H, W, D = cfg["h"], cfg["w"], cfg["d"]
        flops = [f * batch for f in bench.cnn_layer_flops(H, W, 6 * D, 2 * D, bench.NGF, coord)]
        print("\n# conv launches of the last step (graph order; %dx%d, %d planes, batch %d, %s; tile<BM, BN, MODE, BF16> = conv_igemm_kernel, "
              "halo<...> = the halo-patch kernels); %% of the %s MFMA peak %.1f TFLOP/s" % (W, H, D, batch, cfg["dtype"], cfg["dtype"], peak))
        fix = c.execute("select start, end from kernels where name like '%conv_fixup%' order by start").fetchall()
        tot_us = 0.0
        peak_time = 0.0   # us the listed launches would take at their instructions' peaks
        # the 1x1 head is a conv_igemm launch (template MODE 2) only on the unfused path; on the blend_psv path it is
        # part of head_assemble_kernel and a step has 17 conv launches
        last = conv[-1][0]
        fused_tail = "igemm" not in last or ("Li2E" not in last and ", 2, " not in last)
        nl = 17 if fused_tail else 18
        ha = c.execute("select start, end from kernels where name like '%head_assemble%' order by start").fetchall()
        for nm, r, fl in zip(NAMES[:nl], conv[-nl:], flops[:nl]):
            tmpl = "<" + r[0].split("<")[1].split(">")[0] + ">" if "<" in r[0] else ""
            us = (r[2] - r[1]) / 1e3
            nxt = [q[1] for q in conv if q[1] > r[1]]
            fx = [(e - s0) / 1e3 for s0, e in fix if r[2] <= s0 < r[2] + 200000 and (not nxt or s0 < nxt[0])]
            fus = sum(fx[:1])
            tot_us += us + fus
            kind = short(r[0]).split("<")[0].replace("conv_igemm_kernel", "tile").replace("_kernel", "")
            # the peak of the instruction this launch runs on (fp32 plans: native fp32 MFMA 157.3, six-product bf16 split 416.7,
            # three-product fp16 split 833.3 fp32-equivalent TFLOP/s -- bench.split_peak)
            lpk = peak if bf16 else bench.split_peak(short(r[0]).split("(")[0])
            peak_time += fl / lpk / 1e6
            print("%-10s %s%s blocks=%d lds=%d vgpr=%d agpr=%d  %8.1f us + fixup %5.1f us  %7.1f TFLOP/s (%4.1f%% of %.1f)" % (
                nm, kind, tmpl, (r[3] // r[6]) * r[4] * r[5], r[7], r[8], r[9], us, fus, fl / (us + fus) / 1e6,
                100 * fl / (us + fus) / 1e6 / lpk, lpk))
        if fused_tail and ha:
            print("color_pred fused with the RGBA assembly: head_assemble_kernel %8.1f us (HBM-bound; not a conv launch)" % ((ha[-1][1] - ha[-1][0]) / 1e3))
        blended = sum(flops[:nl]) / peak_time / 1e6
        print("sum %.1f us -> %.1f TFLOP/s = %.3f of the flops-weighted peak %.1f TFLOP/s of the instructions the launches run on (conv launches listed above)" % (
            tot_us, sum(flops[:nl]) / tot_us / 1e6, sum(flops[:nl]) / tot_us / 1e6 / blended, blended))
        # the HBM-bound stages of the same step: algorithmic bytes (bench.geometry_bytes: SURVEY.md 8d) / average launch duration against the 8 TB/s HBM3E peak,
        # so that bench.py's stages.*.frac can be re-derived from this file alone.  head_assemble = the fused tail (head + K3's bytes)
        gb = bench.geometry_bytes(H, W, D, 2 if bf16 else 4)
        stage_of = (("ods_sweep_kernel", "sweep"), ("ods_sweep_lds_kernel", "sweep"), ("pp_sweep_kernel", "sweep"), ("head_assemble_kernel", "assemble"), ("assemble_kernel", "assemble"),
                    ("mpi_render_kernel", "render"), ("render_kernel", "render"))
        print("\n# HBM-bound kernels (per launch of %d frame(s); algorithmic bytes / average duration; peak %.0f GB/s):" % (batch, bench.PEAK_HBM_GBS))
        seen = set()
        for r in rows:
            nm = short(r[0])
            for key, stage in stage_of:
                if nm.startswith(key) and nm not in seen and steps:
                    seen.add(nm)
                    per_step = r[1] / steps                       # launches per step (the sweep of a big batch may run in chunks)
                    us = r[2] / 1e3 / steps                       # time per step in this kernel
                    byt = gb[stage] * batch
                    print("%-44s %5.2f launch(es) per step  %9.1f us per step  %8.1f MB algorithmic  %7.1f GB/s = %.3f of the HBM peak" % (
                        nm[:44], per_step, us, byt / 1e6, byt / us / 1e3, byt / us / 1e3 / bench.PEAK_HBM_GBS))
                    break
        ln = [r for r in rows if "ln_apply" in r[0]]
        if ln and steps:
            print("ln_apply: %.2f launches and %.1f us per step" % (sum(r[1] for r in ln) / steps, sum(r[2] for r in ln) / 1e3 / steps))


if __name__ == "__main__":
    sys.exit(main())
