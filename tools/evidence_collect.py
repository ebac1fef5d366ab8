#!/usr/bin/env python
"""tools/evidence_collect.py TAG: copies what tools/evidence_run.sh TAG left under gpurun_out/TAG/ into profiles/ (tracked):
TAG_<config>_bench.json, TAG_<config>_kernel_stats.txt, TAG_hbm_traffic_<config>.json, TAG_pytest_gpu.log."""
import os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
for name in sorted(os.listdir(src)):
    d = os.path.join(src, name)
    if not os.path.isdir(d):
        continue
    for f, out in (("bench.json", "%s_%s_bench.json"), ("kernel_stats.txt", "%s_%s_kernel_stats.txt")):
        if os.path.exists(os.path.join(d, f)):
            shutil.copy(os.path.join(d, f), os.path.join(dst, out % (tag, name)))
    if os.path.exists(os.path.join(d, "hbm_traffic.json")):
        shutil.copy(os.path.join(d, "hbm_traffic.json"), os.path.join(dst, "%s_hbm_traffic_%s.json" % (tag, name)))
    print("collected", name)
if os.path.exists(os.path.join(src, "pytest_gpu.log")):
    shutil.copy(os.path.join(src, "pytest_gpu.log"), os.path.join(dst, "%s_pytest_gpu.log" % tag))
