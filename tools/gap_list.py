#!/usr/bin/env python
"""tools/gap_list.py DB [N]: the last N kernel launches of a rocprofv3 --kernel-trace sqlite database in start order, each with its duration and the idle gap before it
(where a step's time outside kernels sits)."""
import sqlite3, sys
db, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()[-n:]
prev = None
for name, s, e in rows:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print("%8.2f us gap  %9.2f us  %s" % (gap, (e - s) / 1e3, name.replace("(anonymous namespace)::", "").replace("void ", "")[:90]))
    prev = e
