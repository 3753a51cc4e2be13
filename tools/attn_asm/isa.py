"""A small gfx950 assembly builder: every instruction is recorded as an IR node (name, destination, sources, modifiers) from which
(1) the assembler text is printed, (2) the functional simulator (sim.py) executes, (3) the static checks (checks.py) read register
use.  Only the instructions the attention kernel needs are defined.

Register ids (for the checks and the simulator): architectural VGPR n -> n, AccVGPR n -> 256 + n, SGPR n -> 1000 + n,
VCC -> 1106/1107, M0 -> 1124, SCC -> 1200.
"""
import struct

VCC_LO = 106
M0_IDX = 124
SCC_ID = 1200


class R:
    """a register range: kind 'v' | 'a' | 's', first index, count"""
    __slots__ = ("kind", "idx", "n")

    def __init__(self, kind, idx, n=1):
        self.kind, self.idx, self.n = kind, idx, n

    def __getitem__(self, i):
        if isinstance(i, slice):
            start = i.start or 0
            stop = self.n if i.stop is None else i.stop
            assert 0 <= start < stop <= self.n, (self, i)
            return R(self.kind, self.idx + start, stop - start)
        assert 0 <= i < self.n, (self, i)
        return R(self.kind, self.idx + i, 1)

    def __len__(self):
        return self.n

    def ids(self):
        base = {"v": 0, "a": 256, "s": 1000}[self.kind]
        return [base + self.idx + k for k in range(self.n)]

    def t(self):
        if self.kind == "s" and self.idx == VCC_LO:
            return "vcc" if self.n == 2 else "vcc_lo"
        if self.kind == "s" and self.idx == M0_IDX:
            return "m0"
        if self.n == 1:
            return f"{self.kind}{self.idx}"
        return f"{self.kind}[{self.idx}:{self.idx + self.n - 1}]"

    def __repr__(self):
        return self.t()


VCC = R("s", VCC_LO, 2)
M0 = R("s", M0_IDX, 1)


def V(i, n=1):
    return R("v", i, n)


def A(i, n=1):
    return R("a", i, n)


def S(i, n=1):
    return R("s", i, n)


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


INLINE_F = {0.5: "0.5", 1.0: "1.0", 2.0: "2.0", 4.0: "4.0", -0.5: "-0.5", -1.0: "-1.0", -2.0: "-2.0", -4.0: "-4.0"}


class Lit:
    """a 32-bit literal / inline constant given as raw bits (is_float only affects how it is printed)"""
    __slots__ = ("bits", "txt")

    def __init__(self, bits, txt=None):
        self.bits = bits & 0xFFFFFFFF
        self.txt = txt

    def t(self):
        if self.txt is not None:
            return self.txt
        b = self.bits
        if b <= 64:
            return str(b)
        if b >= 0xFFFFFFF0:
            return str(b - (1 << 32))
        return hex(b)


def F(x):
    """float constant operand"""
    x = float(x)
    if x == 0.0:
        return Lit(0, "0")
    if x in INLINE_F:
        return Lit(f32_bits(x), INLINE_F[x])
    return Lit(f32_bits(x), hex(f32_bits(x)))


def I(x):
    return Lit(int(x))


class Label:
    def __init__(self, name):
        self.name = name

    def t(self):
        return self.name


def op_t(o):
    if isinstance(o, (R, Lit, Label)):
        return o.t()
    if isinstance(o, int):
        return Lit(o).t()
    if isinstance(o, str):
        return o
    raise TypeError(o)


def op_ids(o):
    return o.ids() if isinstance(o, R) else []


class Ins:
    __slots__ = ("name", "dst", "src", "mods", "klass", "text", "reads", "writes", "comment", "tag")

    def __init__(self, name, dst, src, mods, klass, text, reads, writes):
        self.name, self.dst, self.src, self.mods, self.klass, self.text = name, dst, src, mods, klass, text
        self.reads, self.writes = reads, writes
        self.comment = ""
        self.tag = None


class Prog:
    """instruction list + labels"""

    def __init__(self):
        self.items = []          # Ins | ("label", name) | ("comment", text)
        self._uniq = 0

    # ------------------------------------------------------------------ bookkeeping
    def label(self, name):
        self.items.append(("label", name))
        return Label(name)

    def newlabel(self, stem="L"):
        self._uniq += 1
        return f".{stem}_{self._uniq}"

    def comment(self, text):
        self.items.append(("comment", text))

    def _emit(self, name, dst, src, klass, mods=None, text=None, extra_reads=(), extra_writes=()):
        mods = mods or {}
        if text is None:
            ops = ([dst] if dst is not None else []) + list(src)
            text = name + " " + ", ".join(op_t(o) for o in ops)
        reads = []
        for o in src:
            reads += op_ids(o)
        reads += list(extra_reads)
        writes = (op_ids(dst) if dst is not None else []) + list(extra_writes)
        ins = Ins(name, dst, list(src), mods, klass, text.rstrip(), reads, writes)
        self.items.append(ins)
        return ins

    # ------------------------------------------------------------------ VALU
    def _valu(self, name, dst, *src, klass="valu"):
        return self._emit(name, dst, src, klass)

    def v_mov_b32(self, d, a): return self._valu("v_mov_b32", d, a)
    def v_add_f32(self, d, a, b): return self._valu("v_add_f32", d, a, b)
    def v_pk_add_f32(self, d, a, b):
        """packed f32 add: d[0:1] = a[0:1] + b[0:1] (64-bit aligned register pairs; full rate: two adds per lane and instruction)"""
        for r in (d, a, b):
            assert r.n == 2 and r.idx % 2 == 0, r
        return self._valu("v_pk_add_f32", d, a, b)

    def v_sub_f32(self, d, a, b): return self._valu("v_sub_f32", d, a, b)
    def v_mul_f32(self, d, a, b): return self._valu("v_mul_f32", d, a, b)
    def v_max_f32(self, d, a, b): return self._valu("v_max_f32", d, a, b)
    def v_max3_f32(self, d, a, b, c): return self._valu("v_max3_f32", d, a, b, c)
    def v_exp_f32(self, d, a): return self._valu("v_exp_f32", d, a, klass="trans")
    def v_log_f32(self, d, a): return self._valu("v_log_f32", d, a, klass="trans")
    def v_rcp_f32(self, d, a): return self._valu("v_rcp_f32", d, a, klass="trans")
    def v_cvt_pk_bf16_f32(self, d, a, b): return self._valu("v_cvt_pk_bf16_f32", d, a, b)
    def v_cvt_pk_f16_f32(self, d, a, b): return self._valu("v_cvt_pk_f16_f32", d, a, b)
    def v_add_u32(self, d, a, b): return self._valu("v_add_u32", d, a, b)
    def v_sub_u32(self, d, a, b): return self._valu("v_sub_u32", d, a, b)
    def v_lshlrev_b32(self, d, sh, a): return self._valu("v_lshlrev_b32", d, sh, a)
    def v_lshrrev_b32(self, d, sh, a): return self._valu("v_lshrrev_b32", d, sh, a)
    def v_and_b32(self, d, a, b): return self._valu("v_and_b32", d, a, b)
    def v_or_b32(self, d, a, b): return self._valu("v_or_b32", d, a, b)
    def v_xor_b32(self, d, a, b): return self._valu("v_xor_b32", d, a, b)
    def v_lshl_add_u32(self, d, a, sh, c): return self._valu("v_lshl_add_u32", d, a, sh, c)
    def v_lshl_or_b32(self, d, a, sh, c): return self._valu("v_lshl_or_b32", d, a, sh, c)
    def v_mul_lo_u32(self, d, a, b): return self._valu("v_mul_lo_u32", d, a, b)
    def v_mul_u32_u24(self, d, a, b): return self._valu("v_mul_u32_u24", d, a, b)
    def v_mad_u32_u24(self, d, a, b, c): return self._valu("v_mad_u32_u24", d, a, b, c)
    def v_bfe_u32(self, d, a, off, w): return self._valu("v_bfe_u32", d, a, off, w)

    def v_cmp(self, cond, ty, a, b):
        """v_cmp_<cond>_<ty> vcc, a, b   (a may be a constant / SGPR, b a VGPR)"""
        name = f"v_cmp_{cond}_{ty}"
        return self._emit(name, None, (a, b), "valu", text=f"{name} vcc, {op_t(a)}, {op_t(b)}", extra_writes=VCC.ids())

    def v_cndmask_b32(self, d, a, b):
        """d = vcc ? b : a"""
        return self._emit("v_cndmask_b32", d, (a, b), "valu", text=f"v_cndmask_b32 {op_t(d)}, {op_t(a)}, {op_t(b)}, vcc",
                          extra_reads=VCC.ids())

    def v_accvgpr_read_b32(self, d, a): return self._valu("v_accvgpr_read_b32", d, a)
    def v_accvgpr_write_b32(self, d, a): return self._valu("v_accvgpr_write_b32", d, a)
    def v_accvgpr_mov_b32(self, d, a): return self._valu("v_accvgpr_mov_b32", d, a)
    def v_readfirstlane_b32(self, d, a): return self._valu("v_readfirstlane_b32", d, a)
    def v_mbcnt_lo_u32_b32(self, d, a, b): return self._valu("v_mbcnt_lo_u32_b32", d, a, b)
    def v_mbcnt_hi_u32_b32(self, d, a, b): return self._valu("v_mbcnt_hi_u32_b32", d, a, b)

    def v_permlane32_swap_b32(self, a, b):
        """lanes 32..63 of a <-> lanes 0..31 of b"""
        return self._emit("v_permlane32_swap_b32", None, (a, b), "valu", text=f"v_permlane32_swap_b32 {op_t(a)}, {op_t(b)}",
                          extra_writes=a.ids() + b.ids())

    def v_permlane16_swap_b32(self, a, b):
        """16-lane rows 1 and 3 of a <-> rows 0 and 2 of b"""
        return self._emit("v_permlane16_swap_b32", None, (a, b), "valu", text=f"v_permlane16_swap_b32 {op_t(a)}, {op_t(b)}",
                          extra_writes=a.ids() + b.ids())

    # ------------------------------------------------------------------ MFMA
    def mfma(self, d, a, b, c, dtype="bf16"):
        name = f"v_mfma_f32_32x32x16_{dtype}"
        assert d.n == 16 and a.n == 4 and b.n == 4
        return self._emit(name, d, (a, b, c), "mfma")

    def mfma16(self, d, a, b, c, dtype="bf16"):
        """v_mfma_f32_16x16x32: d[m = 4 (lane >> 4) + r][n = lane & 15] = c + sum_k a[m][k] b[n][k]; a, b: lane holds row lane & 15, k = 8 (lane >> 4) .. + 7"""
        name = f"v_mfma_f32_16x16x32_{dtype}"
        assert d.n == 4 and a.n == 4 and b.n == 4
        ins = self._emit(name, d, (a, b, c), "mfma")
        ins.mods["passes"] = 4
        return ins

    # ------------------------------------------------------------------ LDS
    def ds_read_b128(self, d, addr, offset=0):
        assert d.n == 4 and 0 <= offset < 65536
        return self._emit("ds_read_b128", d, (addr,), "ds", mods={"offset": offset},
                          text=f"ds_read_b128 {op_t(d)}, {op_t(addr)}" + (f" offset:{offset}" if offset else ""))

    def ds_read_b64_tr_b16(self, d, addr, offset=0):
        assert d.n == 2 and 0 <= offset < 65536
        return self._emit("ds_read_b64_tr_b16", d, (addr,), "ds", mods={"offset": offset},
                          text=f"ds_read_b64_tr_b16 {op_t(d)}, {op_t(addr)}" + (f" offset:{offset}" if offset else ""))

    def ds_write_b128(self, addr, data, offset=0):
        assert data.n == 4
        return self._emit("ds_write_b128", None, (addr, data), "dsw", mods={"offset": offset},
                          text=f"ds_write_b128 {op_t(addr)}, {op_t(data)}" + (f" offset:{offset}" if offset else ""))

    # ------------------------------------------------------------------ VMEM (raw buffer, offen)
    def buffer_load_dwordx4(self, d, voff, rsrc, soff, offset=0):
        assert d.n == 4 and rsrc.n == 4 and 0 <= offset < 4096
        return self._emit("buffer_load_dwordx4", d, (voff, rsrc, soff), "vmem", mods={"offset": offset},
                          text=f"buffer_load_dwordx4 {op_t(d)}, {op_t(voff)}, {op_t(rsrc)}, {op_t(soff)} offen" + (f" offset:{offset}" if offset else ""))

    def buffer_load_dword(self, d, voff, rsrc, soff, offset=0):
        assert d.n == 1 and rsrc.n == 4 and 0 <= offset < 4096
        return self._emit("buffer_load_dword", d, (voff, rsrc, soff), "vmem", mods={"offset": offset},
                          text=f"buffer_load_dword {op_t(d)}, {op_t(voff)}, {op_t(rsrc)}, {op_t(soff)} offen" + (f" offset:{offset}" if offset else ""))

    def buffer_load_lds_dwordx4(self, voff, rsrc, soff, policy=""):
        """LDS-DMA: 16 bytes per lane to LDS[M0 + 16*lane]; policy: cache-policy bits as text ("nt", "sc1", "sc0 sc1", ...)"""
        return self._emit("buffer_load_lds_dwordx4", None, (voff, rsrc, soff), "vmem",
                          text=f"buffer_load_dwordx4 {op_t(voff)}, {op_t(rsrc)}, {op_t(soff)} offen {policy + ' ' if policy else ''}lds", extra_reads=M0.ids())

    def buffer_store_dwordx4(self, data, voff, rsrc, soff, offset=0, policy=""):
        assert data.n == 4 and 0 <= offset < 4096
        return self._emit("buffer_store_dwordx4", None, (data, voff, rsrc, soff), "vmem", mods={"offset": offset},
                          text=f"buffer_store_dwordx4 {op_t(data)}, {op_t(voff)}, {op_t(rsrc)}, {op_t(soff)} offen" + (f" offset:{offset}" if offset else "")
                          + (f" {policy}" if policy else ""))

    def buffer_store_dword(self, data, voff, rsrc, soff, offset=0):
        return self._emit("buffer_store_dword", None, (data, voff, rsrc, soff), "vmem", mods={"offset": offset},
                          text=f"buffer_store_dword {op_t(data)}, {op_t(voff)}, {op_t(rsrc)}, {op_t(soff)} offen" + (f" offset:{offset}" if offset else ""))

    # ------------------------------------------------------------------ SALU
    def _salu(self, name, dst, *src, scc=False, reads_scc=False):
        return self._emit(name, dst, src, "salu", extra_writes=[SCC_ID] if scc else (), extra_reads=[SCC_ID] if reads_scc else ())

    def s_mov_b32(self, d, a): return self._salu("s_mov_b32", d, a)
    def s_mov_b64(self, d, a): return self._salu("s_mov_b64", d, a)
    def s_add_u32(self, d, a, b): return self._salu("s_add_u32", d, a, b, scc=True)
    def s_addc_u32(self, d, a, b): return self._salu("s_addc_u32", d, a, b, scc=True, reads_scc=True)
    def s_sub_u32(self, d, a, b): return self._salu("s_sub_u32", d, a, b, scc=True)
    def s_subb_u32(self, d, a, b): return self._salu("s_subb_u32", d, a, b, scc=True, reads_scc=True)
    def s_mul_i32(self, d, a, b): return self._salu("s_mul_i32", d, a, b)
    def s_mul_hi_u32(self, d, a, b): return self._salu("s_mul_hi_u32", d, a, b)
    def s_lshl_b32(self, d, a, b): return self._salu("s_lshl_b32", d, a, b, scc=True)
    def s_lshr_b32(self, d, a, b): return self._salu("s_lshr_b32", d, a, b, scc=True)
    def s_and_b32(self, d, a, b): return self._salu("s_and_b32", d, a, b, scc=True)
    def s_or_b32(self, d, a, b): return self._salu("s_or_b32", d, a, b, scc=True)
    def s_min_u32(self, d, a, b): return self._salu("s_min_u32", d, a, b, scc=True)
    def s_max_i32(self, d, a, b): return self._salu("s_max_i32", d, a, b, scc=True)
    def s_cselect_b32(self, d, a, b): return self._salu("s_cselect_b32", d, a, b, reads_scc=True)

    def s_cmp(self, cond, ty, a, b):
        name = f"s_cmp_{cond}_{ty}"
        return self._emit(name, None, (a, b), "salu", extra_writes=[SCC_ID])

    def s_load(self, d, base, offset):
        name = {1: "s_load_dword", 2: "s_load_dwordx2", 4: "s_load_dwordx4", 8: "s_load_dwordx8", 16: "s_load_dwordx16"}[d.n]
        return self._emit(name, d, (base,), "smem", mods={"offset": offset}, text=f"{name} {op_t(d)}, {op_t(base)}, {hex(offset)}")

    def s_memtime(self, d): return self._emit("s_memtime", d, (), "smem", mods={"offset": 0}, text=f"s_memtime {op_t(d)}")
    def s_memrealtime(self, d): return self._emit("s_memrealtime", d, (), "smem", mods={"offset": 0}, text=f"s_memrealtime {op_t(d)}")

    def global_store_dwordx4(self, addr, data):
        assert addr.n == 2 and data.n == 4
        return self._emit("global_store_dwordx4", None, (addr, data), "vmem", text=f"global_store_dwordx4 {op_t(addr)}, {op_t(data)}, off")

    def s_waitcnt(self, vmcnt=None, lgkmcnt=None):
        parts = []
        if vmcnt is not None:
            parts.append(f"vmcnt({vmcnt})")
        if lgkmcnt is not None:
            parts.append(f"lgkmcnt({lgkmcnt})")
        assert parts
        return self._emit("s_waitcnt", None, (), "wait", mods={"vmcnt": vmcnt, "lgkmcnt": lgkmcnt}, text="s_waitcnt " + " ".join(parts))

    def s_setprio(self, n): return self._emit("s_setprio", None, (), "nop", mods={"n": 0}, text=f"s_setprio {n}")
    def s_barrier(self): return self._emit("s_barrier", None, (), "barrier", text="s_barrier")
    def s_nop(self, n): return self._emit("s_nop", None, (), "nop", mods={"n": n}, text=f"s_nop {n}")
    def s_endpgm(self): return self._emit("s_endpgm", None, (), "end", text="s_endpgm")

    def s_branch(self, lab): return self._emit("s_branch", None, (), "branch", mods={"target": lab}, text=f"s_branch {lab}")

    def s_cbranch(self, cond, lab):
        """cond in scc0 scc1 vccz vccnz"""
        rd = [SCC_ID] if cond.startswith("scc") else VCC.ids()
        return self._emit("s_cbranch_" + cond, None, (), "branch", mods={"target": lab}, text=f"s_cbranch_{cond} {lab}", extra_reads=rd)

    def s_getpc_b64(self, d): return self._emit("s_getpc_b64", d, (), "salu")

    def s_add_label_diff(self, d, a, target, anchor):
        """d = a + (target - anchor)   (32-bit, sets SCC)"""
        return self._emit("s_add_u32", d, (a,), "salu", mods={"target": target, "anchor": anchor},
                          text=f"s_add_u32 {op_t(d)}, {op_t(a)}, {target}-{anchor}", extra_writes=[SCC_ID])

    def s_swappc_b64(self, d, a): return self._emit("s_swappc_b64", d, (a,), "call")
    def s_setpc_b64(self, a): return self._emit("s_setpc_b64", None, (a,), "ret")

    # ------------------------------------------------------------------ output
    def text(self):
        out = []
        for it in self.items:
            if isinstance(it, Ins):
                out.append("\t" + it.text + (f"\t; {it.comment}" if it.comment else ""))
            elif it[0] == "label":
                out.append(f"{it[1]}:")
            else:
                out.append(f"\t; {it[1]}")
        return "\n".join(out) + "\n"

    def count(self):
        return sum(1 for it in self.items if isinstance(it, Ins))
