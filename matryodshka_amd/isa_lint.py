"""ISA lint of the built library (run by matryodshka_amd.build after linking; `python -m matryodshka_amd.isa_lint [lib]`).

Round 5 traced a rare wrong LayerNorm statistic (one wave's sum of squares low in ~0.1 % of forwards, stored values
bit-identical) to ONE compiler-generated instruction of the generic conv epilogue:

    v_pk_mul_f32 v[46:47], v[18:19], v[50:51] op_sel:[0,1] op_sel_hi:[1,0]      ; low lane = v18 * v51 (the HIGH half of src1)

whose low product came out 0 for lanes 48-63 in the first-resident workgroups of a launch ("the
wobble"; tools/asmpatch, tools/wobble_hunt.py: replacing that one instruction by an unpacked v_mul_f32, or by the same
packed multiply without the cross-half operand select, gave 0 events in 10 000 forwards against 10-18 per 6 000-10 000).
The SLP vectorizer is what forms such operand routing from scalar code; the kernels are built with -fno-slp-vectorize and
this lint REFUSES a library in which any packed-fp32 VALU instruction takes the HIGH half of a register pair for its LOW
lane (`op_sel:[...]` with a 1) -- the hand-written two-float code only ever broadcasts a low half (`op_sel_hi`).

Second rule (also round 5).  A store of more than 64 bits reads its data registers AFTER it has issued; a VALU write of
those registers needs one wait state behind the store.  hipcc inserts that wait state for `buffer_store_dwordx3/x4` only
when the store's soffset operand is NOT a register.  With an SGPR soffset and the very next instruction writing the first
data register (the address arithmetic of the following store, allocated onto the freed register), a build of this library
lost a few 16-byte stores per launch on gfx950 -- wrong pixels in conv outputs, different ones each run.  Proven by
inserting `s_nop 0` behind the stores of one kernel in its assembly: that kernel's layer became exact, the others stayed
broken (profiles/r05_store_hazard.txt), and stand-alone by tools/ubench/store_data_hazard.hip: of 100 M stored records,
0.8 % carry the overwritten dword with an SGPR soffset and no wait state, 0 with one; with a literal soffset 24 % / 0.65 % / 0
for 0 / 1 / 2 wait states.  The epilogue no longer uses SGPR soffsets for wide stores, and this lint REFUSES a library in
which a VALU instruction (round 6: every VGPR it writes -- both operands of v_swap_b32, the whole destination range of MFMA /
v_accvgpr_read / 64-bit results) writes the data registers of a wide buffer / global / flat store less than TWO wait states behind
it.  Every store stays pending with its OWN wait-state count until it has two (`store A; store B; v_* A.data` is a finding).

Non-VALU writers.  The other instructions that write VGPRs are VMEM / LDS loads and returning atomics.  The
compiler re-uses a stored register as the destination of the very next load as a matter of course (39 such pairs in the r05 library:
`global_store_dwordx4 v[10:11], v[14:17], off` ; `global_load_dwordx4 v[14:17], ...`, and `ds_read_b128` behind the sweep's strip
stores) and that is not this hazard: the store's data registers are read by the export path a fixed one or two ISSUE cycles after
the store (which is all a "wait state" is), while a load's write-back cannot happen before its memory round trip -- >= 64 cycles for
LDS, >= 500 for VMEM (a VMEM load also queues behind the store in the same in-order address path).  The ISA's hazard table lists
VALU writes only for this case.  They are therefore COUNTED (`load_overlaps`, printed) and not refused; tests/test_native_abi.py
pins the recogniser for both kinds."""
import os
import re
import struct
import subprocess
import sys
import tempfile

def _find_objdump():
    """llvm-objdump of the toolchain that built the library: next to the resolved hipcc's LLVM, then $ROCM_PATH, then /opt/rocm, then PATH."""
    import shutil
    cands = [os.environ.get("LLVM_OBJDUMP")]
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if hipcc and os.path.exists(hipcc):
        root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(root, "llvm", "bin", "llvm-objdump")]
    cands += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "llvm-objdump"), "/opt/rocm/lib/llvm/bin/llvm-objdump",
              shutil.which("llvm-objdump")]
    for c in cands:
        if c and os.path.exists(c):
            return c
    raise RuntimeError("isa_lint: llvm-objdump not found (set LLVM_OBJDUMP)")


OBJDUMP = None   # resolved on first use (_find_objdump)
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FORBIDDEN = re.compile(r"^\s*(v_pk_(?:mul|add|fma)_f32|v_pk_mov_b32)\b.*\bop_sel:\[[01,]*1[01,]*\]")
# wide stores: (mnemonic pattern, index of the DATA operand among the instruction's operands)
WIDE_STORE = re.compile(r"^\s*((?:t?buffer_store_(?:dwordx[34]|format_xyzw?|format_d16_xyzw))|(?:global|flat|scratch)_store_dwordx[34])\s+(.*)$")
VREG = re.compile(r"^v(?:\[(\d+):(\d+)\]|(\d+))$")
NOT_A_VGPR_WRITE = ("v_cmp", "v_cmpx", "v_readfirstlane", "v_readlane", "v_nop")
WIDE_STORE_WAIT_STATES = 2   # measured (tools/ubench/store_data_hazard.hip): an SGPR-soffset store needs 1 wait state before its data is overwritten, a literal-soffset store 2


def _vrange(op):
    m = VREG.match(op.strip())
    if not m:
        return None
    return (int(m.group(1)), int(m.group(2))) if m.group(1) is not None else (int(m.group(3)), int(m.group(3)))


def _store_data(ins):
    """(first, last) data VGPR of a wide store, or None."""
    m = WIDE_STORE.match(ins)
    if not m:
        return None
    ops = [o.strip() for o in m.group(2).split(",")]
    data = ops[0] if "buffer" in m.group(1) else (ops[1] if len(ops) > 1 else "")   # buffer: vdata first; global / flat / scratch: vaddr, vdata, ...
    return _vrange(data)


def _valu_dest(ins):
    """(first, last) VGPR written by a VALU instruction (its first operand), or None."""
    t = ins.strip()
    if not t.startswith("v_") or t.startswith(NOT_A_VGPR_WRITE):
        return None
    parts = t.split(None, 1)
    if len(parts) < 2:
        return None
    return _vrange(parts[1].split(",")[0])


LOAD = re.compile(r"^(?:t?buffer_load_|global_load_|flat_load_|scratch_load_|ds_read|ds_load|ds_bpermute|ds_permute|ds_swizzle|ds_consume|ds_append|image_)")
DS_RETURN = re.compile(r"^ds_\w+_rtn_")
ATOMIC_RETURN = re.compile(r"^(?:buffer|global|flat)_atomic_")


def vgpr_writes(ins):
    """[((first, last), kind)] of every VGPR range `ins` writes; kind "valu": VALU first operand (both operands of v_swap_b32; the whole range of MFMA /
    v_accvgpr_read results); kind "load": destinations of VMEM / LDS loads and of returning atomics.  LDS-DMA loads (`... lds`) have no VGPR destination."""
    t = ins.strip()
    parts = t.split(None, 1)
    if len(parts) < 2:
        return []
    mn, ops = parts[0], [o.strip() for o in parts[1].split(",")]
    out = []
    if mn.startswith("v_"):
        d = _valu_dest(t)
        if d is not None:
            out.append((d, "valu"))
        if mn.startswith("v_swap_b32") and len(ops) > 1 and _vrange(ops[1].split()[0]) is not None:
            out.append((_vrange(ops[1].split()[0]), "valu"))
    elif LOAD.match(mn) or DS_RETURN.match(mn):
        if re.search(r"\blds\b", parts[1]):
            return []
        d = _vrange(ops[0].split()[0]) if ops and ops[0] else None
        if d is not None:
            out.append((d, "load"))
    elif ATOMIC_RETURN.match(mn) and re.search(r"\b(?:glc|sc0)\b", parts[1]):
        d = _vrange(ops[0].split()[0]) if ops and ops[0] else None     # returning atomics: buffer: vdata first (in place); global / flat: vdst first
        if d is not None:
            out.append((d, "load"))
    return out


def scan_kernel_text(lines):
    """The second rule over the instructions of ONE kernel (strings, comments already removed).  Returns (findings, number of wide stores,
    load destinations that overlap a pending store's data: counted, not refused -- see the module docstring)."""
    bad = []
    nstore = nload = 0
    pending = []                          # [data range, text, wait states elapsed since the store]: EVERY store younger than the threshold
    for ins in lines:
        if not ins.strip():
            continue
        if pending:
            writes = vgpr_writes(ins)
            keep = []
            for p in pending:
                hit = [k for d, k in writes if not (d[1] < p[0][0] or d[0] > p[0][1])]
                if hit and "valu" not in hit:
                    nload += 1
                if "valu" in hit:
                    bad.append(p[1] + "  -> (%d wait state(s)) ->  " % p[2] + ins.strip())
                    continue
                mn = re.match(r"\s*s_nop\s+(\d+)", ins)
                p[2] += int(mn.group(1)) + 1 if mn else 1
                if p[2] < WIDE_STORE_WAIT_STATES:
                    keep.append(p)
            pending = keep
        sd = _store_data(ins)
        if sd is not None:
            nstore += 1
            pending.append([sd, ins.strip(), 0])
    return bad, nstore, nload


def _fatbin_section(path):
    data = open(path, "rb").read()
    assert data[:4] == b"\x7fELF" and data[4] == 2, "64-bit ELF expected"
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    def sh(i):
        name, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return name, off, size
    _, stroff, strsize = sh(shstrndx)
    for i in range(shnum):
        name, off, size = sh(i)
        end = data.index(b"\0", stroff + name)
        if data[stroff + name:end] == b".hip_fatbin":
            return data[off:off + size]
    raise RuntimeError("no .hip_fatbin section in " + path)


def code_objects(path):
    """[(triple, bytes)] of every device code object embedded in the shared library."""
    sec = _fatbin_section(path)
    out = []
    pos = sec.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", sec, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", sec, p)
            triple = sec[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and size:
                out.append((triple, sec[pos + off:pos + off + size]))
        pos = sec.find(MAGIC, pos + len(MAGIC))
    return out


def disassemble(path):
    """[(kernel, [instruction text])] of every device code object in the library."""
    global OBJDUMP
    if OBJDUMP is None:
        OBJDUMP = _find_objdump()
    out = []
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".hsaco") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], check=True, stdout=subprocess.PIPE).stdout.decode()
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = (m.group(1), []); out.append(cur); continue
            ins = line.split("//")[0]
            if cur is not None and ins.strip():
                cur[1].append(ins)
    return out


def lint(path, verbose=False):
    """Returns the list of offending (kernel, instruction) pairs; empty = clean."""
    bad = []
    ninst = nstore = nload = 0
    for kernel, lines in disassemble(path):
        found, n, nl = scan_kernel_text(lines)
        nstore += n
        nload += nl
        bad += [(kernel, f) for f in found]
        for ins in lines:
            if "v_pk_" in ins:
                ninst += 1
                if FORBIDDEN.match(ins):
                    bad.append((kernel, ins.strip()))
    if verbose:
        print("[isa_lint] %s: %d packed VALU instructions and %d wide stores checked, %d finding(s) (high-half -> low-lane operand select / a VGPR write into a wide "
              "store's data less than two wait states behind it); %d load destinations on a just-stored register (counted, not a hazard: module docstring)" % (os.path.basename(path), ninst, nstore, len(bad), nload))
    return bad


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmsi_hip.so")
    found = lint(lib, verbose=True)
    for k, i in found[:20]:
        print("  %s: %s" % (k, i))
    sys.exit(1 if found else 0)
