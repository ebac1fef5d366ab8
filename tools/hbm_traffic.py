#!/usr/bin/env python
"""HBM bytes per bench step and per kernel from two rocprofv3 PMC passes over the same bench command
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each with --kernel-trace only, as the gfx950 section of
/opt/skills/guides/MI355X_MICROARCH.md prescribes: counters are in KB; FETCH_SIZE reports half of
the bytes of wide coalesced reads on gfx950 and is doubled here).

    python tools/hbm_traffic.py fetch_results.db write_results.db [--config N] [--no-coord-net] [--commit HASH] \
        > profiles/rNN_hbm_traffic[_configN].json

The number of profiled steps is counted from the render launches (one per bench step).  The profile is stamped with
`csrc_sha` = bench.csrc_hash() of the kernel sources it was measured with (computed on the GPU box, where there is no
.git); bench.py recomputes the hash and reports `traffic_stale` when the sources have changed since.  `--commit` is the
human-readable companion (`git rev-parse --short HEAD` from the build container).
"""
import argparse
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection "
                     "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    out = {}
    for name, n, val in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = re.split(r"[<(]", short)[0].split("::")[-1].strip()
        a = out.setdefault(short, [0, 0.0])
        a[0] += n
        a[1] += val
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_db")
    ap.add_argument("write_db")
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--no-coord-net", action="store_true")
    ap.add_argument("--commit", default="unknown")
    a = ap.parse_args()
    import bench
    f = per_kernel(a.fetch_db, "FETCH_SIZE")
    w = per_kernel(a.write_db, "WRITE_SIZE")
    rk = "render_kernel" if "render_kernel" in f else "mpi_render_kernel"
    steps = f[rk][0]
    assert steps == w[rk][0], "the two passes ran different step counts"
    cfg = bench.CONFIGS[a.config]
    kernels = {}
    for k in sorted(set(f) | set(w)):
        n = f.get(k, w.get(k))[0]
        rd = 2.0 * 1024.0 * f.get(k, [0, 0.0])[1] / steps     # KB -> B, x2 (gfx950 FETCH_SIZE correction)
        wr = 1024.0 * w.get(k, [0, 0.0])[1] / steps
        kernels[k] = {"launches_per_step": round(n / steps, 2), "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr),
                      "hbm_bytes": int(rd + wr)}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only, over "
                       "`python bench.py --config %d%s --steps 3 --warmup 1 --repeats 0 --prewarm 0 --no-cpu-baseline` (%d steps counted "
                       "from the %s launches). Counters are KB; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md "
                       "(calibration: assemble_kernel reads 209.7 MB algorithmic, ln_apply_kernel 367 MB). Bytes are PER STEP "
                       "(= per frame at config 1; a step of config 2/3/4 is a batch)." % (
                           a.config, " --no-coord-net" if a.no_coord_net else "", steps, rk),
               "config": a.config, "coord_net": not a.no_coord_net, "workload": cfg["name"],
               "frames_per_step": cfg["per_rank"] or cfg["total"],
               "git_commit": a.commit, "csrc_sha": bench.csrc_hash(), "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
