#!/usr/bin/env python
"""Assembly-level variants of cnn.hip (r05 hunt for the LayerNorm-sum wobble): compile ONCE with -save-temps, then for each named patch edit the DEVICE assembly of one
kernel (insert waits / nops at chosen points -- register allocation and scheduling of everything else stay byte-for-byte the compiler's), re-assemble, re-bundle, splice the
new fat binary into the host assembly and link tools/_variants/libmsi_<name>.so.   python tools/asmpatch/build.py [--flags "..."] [--regen] name1 name2 ...
Patches are functions in tools/asmpatch/patches.py: name(lines_of_the_kernel) -> new lines."""
import argparse, os, re, subprocess, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = "/tmp/asmpatch"
LLVM = "/opt/rocm/lib/llvm/bin"
ap = argparse.ArgumentParser()
ap.add_argument("--flags", default="-DMSI_DEBUG_SUMS -DMSI_DBG_WRAPT_F16=1")
ap.add_argument("--regen", action="store_true")
ap.add_argument("--unit", default="cnn_x3.hip", help="translation unit of the kernel (r05 split of cnn.hip; the hunt itself ran on the unsplit file)")
ap.add_argument("--kernel", default="_ZN12_GLOBAL__N_120convt_halo_x3_kernelILi2EEEvNS_10ConvParamsE")
ap.add_argument("names", nargs="*")
a = ap.parse_args()
os.makedirs(W, exist_ok=True)
_stem = os.path.splitext(a.unit)[0] if os.path.exists(W + "/" + os.path.splitext(a.unit)[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s") or a.regen else "cnn"
dev_s, host_s = W + "/%s-hip-amdgcn-amd-amdhsa-gfx950.s" % _stem, W + "/%s-host-x86_64-unknown-linux-gnu.s" % _stem
def run(cmd, **kw):
    subprocess.check_call(cmd, **kw)
if a.regen or not os.path.exists(dev_s):
    run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + ROOT + "/include", "-I" + ROOT + "/matryodshka_amd/csrc", "-Wno-unused-function"]
        + a.flags.split() + ["-fno-slp-vectorize", "-c", ROOT + "/matryodshka_amd/csrc/" + a.unit, "-o", "cnn.o", "-save-temps"], cwd=W, stderr=subprocess.DEVNULL)
    subprocess.check_call([sys.executable, "-m", "matryodshka_amd.build"], cwd=ROOT, stdout=subprocess.DEVNULL)
spec = importlib.util.spec_from_file_location("patches", os.path.join(os.path.dirname(os.path.abspath(__file__)), "patches.py"))
patches = importlib.util.module_from_spec(spec); spec.loader.exec_module(patches)
src = open(dev_s).read().split("\n")
k0 = next(i for i, l in enumerate(src) if l.startswith(a.kernel + ":"))
k1 = next(i for i in range(k0, len(src)) if src[i].startswith(".Lfunc_end"))
host = open(host_s).read()
m = re.search(r"(\.section\s+\.hip_fatbin[^\n]*\n\s*\.p2align[^\n]*\n(\.L__unnamed_\d+):\n)\s*\.asciz\s+\"(?:[^\"\\]|\\.)*\"\n\s*\.size\s+\2, \d+\n", host)
assert m, "fat binary not found in the host assembly"
for name in a.names:
    fn = getattr(patches, name)
    body = fn(list(src[k0:k1]))
    rest = body + list(src[k1:])
    meta = getattr(patches, name + "_meta", None)
    if meta:   # edit the kernel descriptor (register counts) of the patched kernel
        d0 = next(i for i, l in enumerate(rest) if l.strip() == ".amdhsa_kernel " + a.kernel)
        d1 = next(i for i in range(d0, len(rest)) if ".end_amdhsa_kernel" in rest[i])
        for i in range(d0, d1):
            for key, val in meta.items():
                if rest[i].strip().startswith(".amdhsa_" + key + " "):
                    rest[i] = "\t\t.amdhsa_%s %d" % (key, val)
    out = src[:k0] + rest
    ps = "%s/dev_%s.s" % (W, name)
    open(ps, "w").write("\n".join(out))
    run([LLVM + "/clang", "-cc1as", "-triple", "amdgcn-amd-amdhsa", "-filetype", "obj", "-main-file-name", "cnn.hip", "-target-cpu", "gfx950", "-mrelocation-model", "pic", "-o", "%s/dev_%s.o" % (W, name), ps])
    run([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", "%s/dev_%s.out" % (W, name), "%s/dev_%s.o" % (W, name)])
    fb = "%s/%s.hipfb" % (W, name)
    run([LLVM + "/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=%s/dev_%s.out" % (W, name), "-output=" + fb])
    hs = host[:m.start()] + m.group(1) + '\t.incbin "%s"\n\t.size\t%s, %d\n' % (fb, m.group(2), os.path.getsize(fb)) + host[m.end():]
    hp = "%s/host_%s.s" % (W, name)
    open(hp, "w").write(hs)
    run([LLVM + "/clang", "-cc1as", "-triple", "x86_64-unknown-linux-gnu", "-filetype", "obj", "-main-file-name", "cnn.hip", "-target-cpu", "x86-64", "-mrelocation-model", "pic", "-o", "%s/cnn_%s.o" % (W, name), hp])
    os.makedirs(ROOT + "/tools/_variants", exist_ok=True)
    run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "%s/tools/_variants/libmsi_%s.so" % (ROOT, name),
         ROOT + "/matryodshka_amd/csrc/_obj/common.o", ROOT + "/matryodshka_amd/csrc/_obj/geometry.o", "%s/cnn_%s.o" % (W, name)] +
        [ROOT + "/matryodshka_amd/csrc/_obj/" + u[:-4] + ".o" for u in ("cnn.hip", "cnn_igemm.hip", "cnn_halo.hip", "cnn_x3.hip", "cnn_bf16.hip", "cnn_tail.hip") if u != a.unit])
    print("built", name, "(%d -> %d lines)" % (k1 - k0, len(body)))
