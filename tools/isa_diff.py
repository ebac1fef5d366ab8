#!/usr/bin/env python
"""Per-kernel comparison of the device code of two builds of libmsi_hip.so (r05: refactors of the k-loops must not change the instruction stream).
  python tools/isa_diff.py snapshot NAME      -> /tmp/isa_NAME.json  {kernel: [instructions, addresses and branch targets stripped]}
  python tools/isa_diff.py diff NAME_A NAME_B -> per kernel: instruction counts, identical or not, first differing instruction"""
import json, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from matryodshka_amd import isa_lint

def snapshot(lib):
    out = {}
    for triple, blob in isa_lint.code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".hsaco") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([isa_lint.OBJDUMP, "-d", "--no-show-raw-insn", f.name], check=True, stdout=subprocess.PIPE).stdout.decode()
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1); out.setdefault(cur, []); continue
            ins = line.split("//")[0].strip()
            if cur and ins and not ins.startswith("s_nop") and not ins.startswith("s_code_end"):
                ins = re.sub(r"\b(s_cbranch_\w+|s_branch)\s+\S+", r"\1 L", ins)
                out[cur].append(ins)
    return out

if sys.argv[1] == "snapshot":
    lib = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matryodshka_amd", "libmsi_hip.so")
    json.dump(snapshot(lib), open("/tmp/isa_%s.json" % sys.argv[2], "w"))
    print("saved /tmp/isa_%s.json" % sys.argv[2])
else:
    a, b = (json.load(open("/tmp/isa_%s.json" % n)) for n in sys.argv[2:4])
    same = 0
    for k in sorted(set(a) | set(b)):
        if k not in a or k not in b:
            print("only in one build:", k[:100]); continue
        if a[k] == b[k]:
            same += 1; continue
        i = next((i for i, (x, y) in enumerate(zip(a[k], b[k])) if x != y), min(len(a[k]), len(b[k])))
        print("DIFFERS %-90s %6d vs %6d instructions; first difference at %d: %s | %s" % (k[:90], len(a[k]), len(b[k]), i, a[k][i] if i < len(a[k]) else "-", b[k][i] if i < len(b[k]) else "-"))
    print("%d kernels identical" % same)
