"""Patch functions for tools/asmpatch/build.py: each takes the assembly lines of ONE kernel and returns the edited lines."""
import re

def ident(L):
    return L

def _is_inst(l):
    s = l.strip()
    return bool(s) and not s.startswith((";", ".", "#")) and not s.endswith(":")

def _epilogue_start(L):
    """index of the last line of the k-loop (the FIRST loop of the kernel in layout order)"""
    hdr = None
    for l in L:
        m = re.search(r"in Loop: Header=(BB\d+_\d+) ", l)
        if m:
            hdr = m.group(1); break
    assert hdr
    idx = [i for i, l in enumerate(L) if ("Header=" + hdr + " ") in l]
    i = idx[-1] + 1
    while not (L[i].startswith(".LBB") and "in Loop" not in L[i]):
        i += 1
    return i - 1

def nop_before_dpp(L):
    """s_nop 7 before every DPP instruction and every v_readlane in the epilogue"""
    out = []
    for l in L:
        if "_dpp" in l or "v_readlane_b32" in l or "v_readfirstlane_b32" in l:
            out.append("\ts_nop 7")
        out.append(l)
    return out

def nop_after_pk(L):
    """s_nop 3 after every packed-math instruction"""
    out = []
    for l in L:
        out.append(l)
        if re.match(r"\s+v_pk_", l):
            out.append("\ts_nop 3")
    return out

def nop_after_valu_epi(L):
    """s_nop 1 after EVERY VALU instruction behind the k-loop (the forcezero effect without the waits)"""
    e = _epilogue_start(L)
    out = []
    for i, l in enumerate(L):
        out.append(l)
        if i > e and re.match(r"\s+v_", l) and "v_mfma" not in l:
            out.append("\ts_nop 1")
    return out

def wait_after_vmem_epi(L):
    """s_waitcnt vmcnt(0) after every global / buffer memory instruction behind the k-loop"""
    e = _epilogue_start(L)
    out = []
    for i, l in enumerate(L):
        out.append(l)
        if i > e and re.match(r"\s+(global_|buffer_|flat_)", l):
            out.append("\ts_waitcnt vmcnt(0)")
    return out

def wait_all_epi(L):
    """s_waitcnt vmcnt(0) lgkmcnt(0) after every instruction behind the k-loop (forcezero, epilogue only)"""
    e = _epilogue_start(L)
    out = []
    for i, l in enumerate(L):
        out.append(l)
        if i > e and _is_inst(l) and not re.match(r"\s+(s_endpgm|s_branch|s_cbranch|s_setpc)", l):
            out.append("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
    return out

def wait_all_pre(L):
    """the same for everything UP TO the end of the k-loop (prologue + loop), epilogue untouched"""
    e = _epilogue_start(L)
    out = []
    for i, l in enumerate(L):
        out.append(l)
        if i <= e and _is_inst(l) and not re.match(r"\s+(s_endpgm|s_branch|s_cbranch|s_setpc|s_barrier|s_waitcnt)", l):
            out.append("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
    return out

def _loop_start(L):
    hdr = None
    for i, l in enumerate(L):
        m = re.search(r"in Loop: Header=(BB\d+_\d+) ", l)
        if m:
            return i
    raise AssertionError

def _after(L, pred, ins, lo=None, hi=None):
    lo = 0 if lo is None else lo
    hi = len(L) if hi is None else hi
    out = []
    for i, l in enumerate(L):
        out.append(l)
        if lo <= i <= hi and _is_inst(l) and pred(l):
            out.extend(ins)
    return out

W0 = ["\ts_waitcnt vmcnt(0) lgkmcnt(0)"]
_noctl = lambda l: not re.match(r"\s+(s_endpgm|s_branch|s_cbranch|s_setpc|s_barrier|s_waitcnt)", l)

def pre_wait_dma(L):
    """full wait after every buffer_load ... lds (weight DMA) up to the end of the k-loop"""
    return _after(L, lambda l: re.match(r"\s+buffer_load", l) and " lds" in l, W0, hi=_epilogue_start(L))

def pre_wait_vload(L):
    """full wait after every load into VGPRs (global_load / buffer_load without lds) up to the end of the k-loop"""
    return _after(L, lambda l: re.match(r"\s+(global_load|buffer_load|flat_load)", l) and " lds" not in l, W0, hi=_epilogue_start(L))

def pre_wait_ds(L):
    """full wait after every LDS instruction up to the end of the k-loop"""
    return _after(L, lambda l: re.match(r"\s+ds_", l), W0, hi=_epilogue_start(L))

def pre_wait_prologue(L):
    """full wait after every instruction BEFORE the k-loop only"""
    return _after(L, _noctl, W0, hi=_loop_start(L) - 1)

def pre_wait_loop(L):
    """full wait after every instruction INSIDE the k-loop only"""
    return _after(L, _noctl, W0, lo=_loop_start(L), hi=_epilogue_start(L))

def pre_nop_all(L):
    """s_nop 3 (no wait) after every instruction up to the end of the k-loop: the timing effect alone"""
    return _after(L, _noctl, ["\ts_nop 3"], hi=_epilogue_start(L))

def pre_wait_smem(L):
    """full wait after every scalar memory load up to the end of the k-loop"""
    return _after(L, lambda l: re.match(r"\s+(s_load|s_buffer_load)", l), W0, hi=_epilogue_start(L))


# ---- per-lane dump of the non-interior epilogue's LayerNorm partials (s1, s2, count) just before the wave reductions ----------------
# Uses registers the compiler left free (v159, s100-s105): nothing the compiler allocated moves.  Record of (workgroup, class group G, wave):
# 1 KB at partial + DUMP_OFF + (((bid * 2 + G) * 4 + wave) << 10): lane * 4 + {0: first reduced register, 256: second, 512: third}.
DUMP_OFF = 48 << 20
def dump_lanes(L):
    out = list(L)
    # reductions = v_add_f32_dpp ... quad_perm:[1,0,3,2]; groups of three consecutive ones (within 40 lines) = the non-interior statistics (s1, s2, cnt)
    red = [(i, re.match(r"\s+v_add_f32_dpp v(\d+), v(\d+), v(\d+) quad_perm:\[1,0,3,2\]", l)) for i, l in enumerate(L)]
    red = [(i, int(m.group(2))) for i, m in red if m]
    groups, cur = [], []
    for i, src in red:
        if cur and i - cur[-1][0] > 40:
            groups.append(cur); cur = []
        cur.append((i, src))
    groups.append(cur)
    tri = [g for g in groups if len(g) == 3]
    assert len(tri) >= 2, [len(g) for g in groups]
    ins = {}
    for G, g in enumerate(tri):
        code = ["\tv_readlane_b32 s100, v159, 2", "\ts_add_u32 s100, s100, %d" % ((G & 1) * 4), "\ts_lshl_b32 s100, s100, 10", "\ts_mov_b32 m0, s100",
                "\tv_readlane_b32 s100, v159, 0", "\tv_readlane_b32 s101, v159, 1",
                "\tv_mbcnt_lo_u32_b32 v159, -1, 0", "\tv_mbcnt_hi_u32_b32 v159, -1, v159", "\tv_lshlrev_b32_e32 v159, 2, v159",
                "\tv_add_u32_e32 v159, m0, v159", "\ts_nop 1"]
        for k, (_, src) in enumerate(g):
            code.append("\tglobal_store_dword v159, v%d, s[100:101] offset:%d" % (src, 256 * k))
        code += ["\ts_nop 7", "\tv_writelane_b32 v159, s100, 0", "\tv_writelane_b32 v159, s101, 1", "\ts_lshr_b32 s100, m0, 10",
                 "\ts_sub_u32 s100, s100, %d" % ((G & 1) * 4), "\tv_writelane_b32 v159, s100, 2"]
        ins[g[0][0]] = code
    res = []
    for i, l in enumerate(out):
        if i in ins:
            res.extend(ins[i])
        res.append(l)
    # entry: pointer (partial + DUMP_OFF) in lanes 0 / 1 of the free VGPR v159, record base workgroup * 8 + wave in lane 2
    e = next(i for i, l in enumerate(res) if l.strip().startswith("s_load_"))
    entry = ["\tv_lshrrev_b32_e32 v159, 6, v0", "\ts_nop 1", "\tv_readfirstlane_b32 s100, v159", "\ts_lshl_b32 s101, s2, 3", "\ts_add_u32 s100, s100, s101",
             "\tv_writelane_b32 v159, s100, 2", "\ts_load_dwordx2 s[100:101], s[0:1], 0xb0", "\ts_waitcnt lgkmcnt(0)",
             "\ts_add_u32 s100, s100, 0x%x" % DUMP_OFF, "\ts_addc_u32 s101, s101, 0", "\tv_writelane_b32 v159, s100, 0", "\tv_writelane_b32 v159, s101, 1"]
    return res[:e] + entry + res[e:]
dump_lanes_meta = {"next_free_vgpr": 160, "next_free_sgpr": 102}
dump_lanes.ngroups = None

# ---- round 3: around the instruction whose result is lost (v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]: lo = dy^2 of the first float4 group) ----
_EXECW = r"\s+s_(and|andn2|or|xor|orn2)(_saveexec)?_b64\s+(exec|s\[\d+:\d+\]), ?(exec)?.*"
def _writes_exec(l):
    return bool(re.match(r"\s+s_\w+_saveexec_b64", l) or re.match(r"\s+s_\w+_b64 exec,", l) or re.match(r"\s+s_mov_b64 exec,", l))

def x1_nop_after_execw(L):
    e = _epilogue_start(L)
    return _after(L, _writes_exec, ["\ts_nop 4"], lo=e)

def x2_nop_before_execw(L):
    e = _epilogue_start(L)
    out = []
    for i, l in enumerate(L):
        if i > e and _is_inst(l) and _writes_exec(l):
            out.append("\ts_nop 4")
        out.append(l)
    return out

def x3_unpacked_mul(L):
    out = []
    for l in L:
        m = re.match(r"\s+v_pk_mul_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1\] op_sel_hi:\[1,0\]", l)
        if m:
            d0, _, a0, _, _, b1 = [int(v) for v in m.groups()]
            out.append("\tv_mul_f32_e32 v%d, v%d, v%d" % (d0, a0, b1))
        else:
            out.append(l)
    return out

def _shift(n):
    def f(L):
        e = next(i for i, l in enumerate(L) if l.strip().startswith("s_load_"))
        return L[:e] + ["\ts_nop 0"] * n + L[e:]
    return f
x4_shift1, x4_shift2, x4_shift4, x4_shift8, x4_shift13 = _shift(1), _shift(2), _shift(4), _shift(8), _shift(13)

def x5_nop_after_mul(L):
    return _after(L, lambda l: "op_sel:[0,1] op_sel_hi:[1,0]" in l and "v_pk_mul_f32" in l, ["\ts_nop 3"])

def x6_nop_before_mul(L):
    out = []
    for l in L:
        if "op_sel:[0,1] op_sel_hi:[1,0]" in l and "v_pk_mul_f32" in l:
            out.append("\ts_nop 3")
        out.append(l)
    return out

_MUL5 = r"\s+v_pk_mul_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1\] op_sel_hi:\[1,0\]"
def y1_unpacked_add2(L):
    """the packed add three instructions before the multiply (its low half is junk written to the same register) as an unpacked add of the high half only"""
    out = list(L)
    for i, l in enumerate(L):
        if re.match(_MUL5, l):
            for j in range(i - 1, i - 6, -1):
                m = re.match(r"\s+v_pk_add_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel_hi:\[0,1\]", L[j])
                if m:
                    d0, d1, a0, a1, b0, b1 = [int(v) for v in m.groups()]
                    out[j] = "\tv_add_f32_e32 v%d, v%d, v%d" % (d1, a0, b1)
                    break
            else:
                raise AssertionError("packed add not found")
    return out

def y2_no_opsel(L):
    """the multiply as v_pk_mul_f32 d, a, a (no op_sel): lo = a.lo^2 = dy^2 as before, hi = junk as before"""
    out = []
    for l in L:
        m = re.match(_MUL5, l)
        if m:
            d0, d1, a0, a1, b0, b1 = [int(v) for v in m.groups()]
            out.append("\tv_pk_mul_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (d0, d1, a0, a1, a0, a1))
        else:
            out.append(l)
    return out

def y7_poison(L):
    """NaN into the multiply's low destination register just before it: a dropped write turns the sum into NaN (status LN_OVERFLOW), a zero product does not"""
    out = []
    for l in L:
        m = re.match(_MUL5, l)
        if m:
            out.append("\tv_mov_b32_e32 v%d, 0x7fc00000" % int(m.group(1)))
        out.append(l)
    return out


def nop_after_wide_store(L):
    """s_nop 0 after every buffer_store_dwordx3 / x4 (r05: a VALU write of the store's data registers in the very next instruction -- legal for the compiler's hazard
    recognizer when the store has an SGPR soffset -- corrupted a few stores per launch on gfx950)"""
    out = []
    for l in L:
        out.append(l)
        if re.match(r"\s*buffer_store_dwordx[34]\s", l):
            out.append("\ts_nop 0")
    return out
