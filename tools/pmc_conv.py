#!/usr/bin/env python
"""Prints rocprofv3 --pmc counters (rocpd db) for the conv_igemm launches of the last frame,
with each launch's duration when the run also had --kernel-trace."""
import sqlite3, sys
names = ["conv1_1","conv1_2","conv2_1","conv2_2","conv3_1","conv3_2","conv3_3","conv4_1","conv4_2","conv4_3","conv6_1","conv6_2","conv6_3","conv7_1","conv7_2","conv8_1","conv8_2","head"]
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, counter_name, sum(value) from counters_collection where (kernel_name like '%conv_igemm%' or kernel_name like '%conv_halo%') group by dispatch_id, counter_name order by dispatch_id").fetchall()
    ids = sorted(set(r[0] for r in rows))[-18:]
    ctrs = sorted(set(r[1] for r in rows))
    try:
        dur = dict(c.execute("select dispatch_id, end-start from kernels").fetchall())
    except sqlite3.Error:
        dur = {}
    print("# " + db)
    print("%-8s %9s " % ("layer", "us") + " ".join("%14s" % k[-14:] for k in ctrs))
    for nm, i in zip(names, ids):
        vals = {r[1]: r[2] for r in rows if r[0] == i}
        print("%-8s %9.1f " % (nm, dur.get(i, 0) / 1e3) + " ".join("%14.5g" % vals.get(k, float('nan')) for k in ctrs))
