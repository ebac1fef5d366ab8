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
which a VALU instruction writes the data registers of a wide buffer / global / flat store less than TWO wait states behind it."""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
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
    """(first, last) VGPR written by a VALU instruction, or None."""
    t = ins.strip()
    if not t.startswith("v_") or t.startswith(NOT_A_VGPR_WRITE):
        return None
    parts = t.split(None, 1)
    if len(parts) < 2:
        return None
    return _vrange(parts[1].split(",")[0])


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


def lint(path, verbose=False):
    """Returns the list of offending (kernel, instruction) pairs; empty = clean."""
    bad = []
    ninst = nstore = 0
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".hsaco") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], check=True, stdout=subprocess.PIPE).stdout.decode()
        kernel = "?"
        pending = None                     # (data range, text) of a wide store whose NEXT instruction has not been seen yet
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                kernel = m.group(1); pending = None; continue
            ins = line.split("//")[0]
            if not ins.strip():
                continue
            if pending is not None:                # [data range, text, wait states elapsed since the store]
                d = _valu_dest(ins)
                if d is not None and not (d[1] < pending[0][0] or d[0] > pending[0][1]):
                    bad.append((kernel, pending[1] + "  -> (%d wait state(s)) ->  " % pending[2] + ins.strip()))
                    pending = None
                else:
                    mn = re.match(r"\s*s_nop\s+(\d+)", ins)
                    pending[2] += int(mn.group(1)) + 1 if mn else 1
                    if pending[2] >= WIDE_STORE_WAIT_STATES:
                        pending = None
            sd = _store_data(ins)
            if sd is not None:
                nstore += 1
                pending = [sd, ins.strip(), 0]
            if "v_pk_" in ins:
                ninst += 1
                if FORBIDDEN.match(ins):
                    bad.append((kernel, ins.strip()))
    if verbose:
        print("[isa_lint] %s: %d packed VALU instructions and %d wide stores checked, %d finding(s) (high-half -> low-lane operand select / VALU write of a wide "
              "store's data less than two wait states behind it)" % (os.path.basename(path), ninst, nstore, len(bad)))
    return bad


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmsi_hip.so")
    found = lint(lib, verbose=True)
    for k, i in found[:20]:
        print("  %s: %s" % (k, i))
    sys.exit(1 if found else 0)
