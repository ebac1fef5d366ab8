"""Builds matryodshka_amd/libmsi_hip.so for gfx950 with hipcc (cross-compiles
without a GPU).  `python -m matryodshka_amd.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(HERE, "libmsi_hip.so")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC,
          "-Wall", "-Wno-unused-function"]
# (source, extra flags).  geometry.hip must not contract a*b+c into fma: see its header.
# -fno-slp-vectorize (r05): the SLP vectorizer packs scalar fp32 code into v_pk_*_f32 with cross-half operand selects; one such instruction of the
# conv epilogue was the source of a rare wrong LayerNorm statistic (matryodshka_amd/isa_lint.py, DESIGN.md section 4 "the wobble").  Packed math
# that pays is written by hand as two-float vectors; the lint below refuses a library that contains the operand routing again.
NO_SLP = ["-fno-slp-vectorize"]
SOURCES = [
    ("common.cpp", ["-x", "hip"]),
    ("probe.hip", []),                                                     # msi_probe_matrix_rate (measurement aid of bench.py)
    ("geometry.hip", ["-ffp-contract=off"] + NO_SLP + os.environ.get("MSI_GEO_DEFINES", "").split()),   # e.g. MSI_GEO_DEFINES="-DMSI_SWEEP_WAVES=5" (tuning)
]
# the K2 convolution path: one translation unit per kernel family (r05; cnn_device.h holds what they share), compiled in parallel
CNN_UNITS = ["cnn.hip", "cnn_igemm.hip", "cnn_halo.hip", "cnn_x3.hip", "cnn_bf16.hip", "cnn_tail.hip"]
SOURCES += [(u, NO_SLP + os.environ.get("MSI_CNN_DEFINES", "").split()) for u in CNN_UNITS]   # e.g. MSI_CNN_DEFINES="-DMSI_NSTAGE=2" (tuning)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(INCLUDE, "msi_hip.h"), os.path.join(CSRC, "msi_common.h"), os.path.join(CSRC, "cnn_device.h"), __file__]
    objs = []
    jobs = []
    for src, extra in SOURCES:
        spath = os.path.join(CSRC, src)
        opath = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
        objs.append(opath)
        if force or _newer(opath, [spath] + headers):
            cmd = [hipcc] + COMMON + extra + ["-c", spath, "-o", opath]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            jobs.append((src, subprocess.Popen(cmd)))
    rebuilt = bool(jobs)
    failed = [src for src, proc in jobs if proc.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(failed))
    if rebuilt or force or _newer(LIB, objs):
        # Link to a temporary name and move it onto LIB only when the lint has passed: a library that was never linted (objdump missing, lint crashed)
        # or that the lint refused must not be found by the next build() as "up to date" (the gate used to fail open).
        tmp = LIB + ".unlinted"
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        if not os.environ.get("MSI_SKIP_ISA_LINT"):
            from matryodshka_amd import isa_lint
            try:
                bad = isa_lint.lint(tmp, verbose=verbose)
            except Exception:
                os.replace(tmp, LIB + ".rejected")
                if os.path.exists(LIB):
                    os.remove(LIB)
                raise
            if bad:
                os.replace(tmp, LIB + ".rejected")
                if os.path.exists(LIB):
                    os.remove(LIB)
                raise RuntimeError("isa_lint: %d finding(s) (a packed-fp32 instruction routing a HIGH register half to the LOW lane, or a VALU write of a wide store's data "
                                   "less than two wait states behind it; first: %s in %s); see matryodshka_amd/isa_lint.py" % (len(bad), bad[0][1], bad[0][0]))
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
