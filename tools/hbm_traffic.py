#!/usr/bin/env python
"""HBM bytes per frame and per kernel from two rocprofv3 PMC passes over the same bench command
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each with --kernel-trace only, as the gfx950 section of
/opt/skills/guides/MI355X_MICROARCH.md prescribes: counters are in KB; FETCH_SIZE reports half of
the bytes of wide coalesced reads on gfx950 and is doubled here).

    python tools/hbm_traffic.py fetch_results.db write_results.db FRAMES [GIT_COMMIT] > profiles/rNN_hbm_traffic.json

FRAMES = 0: counted from the render_kernel launches (one per frame).

GIT_COMMIT: the commit the measured library was built from (bench.py prints it next to `roofline.traffic`, so a
stale profile is visible); the GPU box has no .git, pass `git rev-parse --short HEAD` from the build container.
"""
import json
import re
import sqlite3
import sys


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
    fetch_db, write_db, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
    commit = sys.argv[4] if len(sys.argv) > 4 else "unknown"
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    if frames == 0:   # one render_kernel launch per frame (rgb + depth in one pass)
        frames = f["render_kernel"][0]
        assert frames == w["render_kernel"][0], "the two passes ran different frame counts"
    kernels = {}
    for k in sorted(set(f) | set(w)):
        n = f.get(k, w.get(k))[0]
        rd = 2.0 * 1024.0 * f.get(k, [0, 0.0])[1] / frames     # KB -> B, x2 (gfx950 FETCH_SIZE correction)
        wr = 1024.0 * w.get(k, [0, 0.0])[1] / frames
        kernels[k] = {"launches_per_frame": round(n / frames, 2), "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr),
                      "hbm_bytes": int(rd + wr)}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only, over "
                       "`python bench.py --steps 3 --warmup 1 --repeats 0 --prewarm 0 --no-cpu-baseline` (%d frames, 640x320, 32 spheres, B=1). "
                       "Counters are KB; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md (calibration: "
                       "assemble_kernel reads 209.7 MB algorithmic, ln_apply_kernel 367 MB)." % frames,
               "git_commit": commit, "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
