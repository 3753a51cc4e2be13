"""Functional simulator for the instruction subset of isa.py: one workgroup of wave64 waves, a shared LDS, a flat global memory.
Purpose: debug the generated attention kernel's LOGIC (layouts, addressing, software pipeline, counted waits, barrier protocol) on
the CPU.  It is not a timing model.  Besides executing, it checks the asynchronous-memory protocol the hardware does not enforce:

  * a register that is the destination of an LDS read / buffer load is PENDING until an s_waitcnt retires the operation (LDS
    operations retire in order against lgkmcnt, vector-memory operations in order against vmcnt); touching a pending register
    is a violation;
  * the LDS bytes an LDS-DMA writes are IN FLIGHT from its issue until the issuing wave's s_waitcnt vmcnt retires it (the data
    is written at that moment, the latest the hardware allows); reading in-flight bytes is a violation; so is another wave
    reading them before a barrier has passed after the retirement; so is issuing a DMA onto bytes another wave has read since
    the last barrier.
Violations are collected in Machine.violations (strings), execution continues.
"""
import numpy as np
from .isa import Ins, R, Lit, VCC_LO, M0_IDX

F32 = np.float32
U32 = np.uint32


def bf16_round(x):
    """f32 array -> bf16 bits (uint32 in low 16), round to nearest even, NaN kept"""
    u = x.view(U32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF
    nan = np.isnan(x)
    r = np.where(nan, 0x7FC0, r)
    return r.astype(U32)


def bf16_to_f32(h):
    return (h.astype(U32) << 16).view(F32)


def f16_round(x):
    return x.astype(np.float16).view(np.uint16).astype(U32)


def f16_to_f32(h):
    return h.astype(np.uint16).view(np.float16).astype(F32)


class Wave:
    def __init__(self, wid, nlanes=64):
        self.wid = wid
        self.v = np.zeros((512, 64), dtype=U32)
        self.s = np.zeros(128, dtype=U32)
        self.scc = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False
        self.lds_q = []          # pending LDS reads: (regs set)
        self.vm_q = []           # pending vector-memory ops: dict(kind, regs, apply)
        self.pending = {}        # reg id -> description
        self.icount = 0


class Machine:
    def __init__(self, prog, nwaves=4, lds_bytes=163840, dtype="bf16"):
        self.items = prog.items
        self.labels = {}
        for i, it in enumerate(self.items):
            if not isinstance(it, Ins) and it[0] == "label":
                self.labels[it[1]] = i
        self.mem = np.zeros(0, dtype=np.uint8)
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.lds_inflight = np.zeros(lds_bytes, dtype=np.int32)       # >0: a DMA onto this byte has been issued and not retired
        self.lds_pub_epoch = np.full(lds_bytes, -1, dtype=np.int64)   # epoch in which the last DMA onto this byte retired
        self.lds_pub_wave = np.full(lds_bytes, -1, dtype=np.int32)
        self.lds_read_epoch = np.full(lds_bytes, -1, dtype=np.int64)  # last epoch in which the byte was read
        self.lds_read_wave = np.full(lds_bytes, -1, dtype=np.int32)   # -2: several waves
        self.epoch = 0
        self.waves = [Wave(w) for w in range(nwaves)]
        self.violations = []
        self.dtype = dtype
        self.trace = None
        self.mfma_count = 0

    # ------------------------------------------------------------------ memory
    def alloc(self, nbytes, align=256):
        base = (len(self.mem) + align - 1) // align * align
        if base == 0:
            base = 4096
        new = np.zeros(base + nbytes, dtype=np.uint8)
        new[:len(self.mem)] = self.mem
        self.mem = new
        return base

    def write(self, addr, arr):
        b = np.ascontiguousarray(arr).view(np.uint8).ravel()
        self.mem[addr:addr + len(b)] = b

    def read(self, addr, nbytes):
        return self.mem[addr:addr + nbytes].copy()

    def viol(self, w, msg):
        if len(self.violations) < 200:
            self.violations.append(f"wave {w.wid} pc {w.pc} [{self._cur_text}]: {msg}")

    # ------------------------------------------------------------------ operand access
    def rd(self, w, o, lanes=True):
        """value of a 32-bit operand: per-lane uint32[64] (VGPR) or scalar broadcast"""
        if isinstance(o, R):
            assert o.n == 1, o
            if o.kind == "s":
                return np.full(64, w.s[o.idx], dtype=U32)
            idx = o.idx + (256 if o.kind == "a" else 0)
            return w.v[idx].copy()
        if isinstance(o, Lit):
            return np.full(64, o.bits, dtype=U32)
        if isinstance(o, int):
            return np.full(64, o & 0xFFFFFFFF, dtype=U32)
        raise TypeError(o)

    def rs(self, w, o):
        if isinstance(o, R):
            assert o.kind == "s" and o.n == 1, o
            return int(w.s[o.idx])
        if isinstance(o, Lit):
            return o.bits
        return int(o) & 0xFFFFFFFF

    def wv(self, w, d, val):
        idx = d.idx + (256 if d.kind == "a" else 0)
        w.v[idx] = np.asarray(val).astype(U32) if np.asarray(val).dtype != U32 else val

    def wf(self, w, d, val):
        self.wv(w, d, np.asarray(val, dtype=F32).view(U32))

    def rf(self, w, o):
        return self.rd(w, o).view(F32)

    def vblock(self, w, r):
        base = r.idx + (256 if r.kind == "a" else 0)
        return w.v[base:base + r.n]

    # ------------------------------------------------------------------ checks on pending registers
    def touch(self, w, ins):
        if not w.pending:
            return
        for rid in list(ins.reads) + list(ins.writes):
            if rid in w.pending:
                self.viol(w, f"touches register {rid} while {w.pending[rid]} is outstanding")

    def retire_lds(self, w, keep):
        while len(w.lds_q) > keep:
            regs = w.lds_q.pop(0)
            for r in regs:
                w.pending.pop(r, None)

    def retire_vm(self, w, keep):
        while len(w.vm_q) > keep:
            op = w.vm_q.pop(0)
            for r in op.get("regs", ()):
                w.pending.pop(r, None)
            if op.get("apply"):
                op["apply"]()

    # ------------------------------------------------------------------ buffer addressing
    def buf_addr(self, w, rsrc, voff, soff, inst_off, size):
        base = int(w.s[rsrc.idx]) | ((int(w.s[rsrc.idx + 1]) & 0xFFFF) << 32)
        nrec = int(w.s[rsrc.idx + 2])
        off = voff.astype(np.int64) + inst_off
        ok = (off + size) <= nrec          # raw buffer, stride 0: byte range check on voffset + inst_offset (soffset excluded)
        addr = base + soff + off
        return addr, ok

    # ------------------------------------------------------------------ one instruction
    def step(self, w):
        it = self.items[w.pc]
        if not isinstance(it, Ins):
            w.pc += 1
            return
        ins = it
        self._cur_text = ins.text
        n = ins.name
        w.icount += 1
        if ins.klass not in ("wait", "nop", "barrier"):
            self.touch(w, ins)
        nxt = w.pc + 1
        d, s = ins.dst, ins.src
        if self.trace is not None:
            self.trace(w, ins)

        if ins.klass in ("valu", "trans"):
            self.exec_valu(w, ins)
        elif ins.klass == "mfma":
            self.exec_mfma(w, ins)
        elif ins.klass == "salu":
            self.exec_salu(w, ins)
        elif ins.klass == "smem" and n in ("s_memtime", "s_memrealtime"):
            w.s[d.idx], w.s[d.idx + 1] = w.icount & 0xFFFFFFFF, 0
        elif ins.klass == "smem":
            base = int(w.s[s[0].idx]) | (int(w.s[s[0].idx + 1]) << 32)
            a = base + ins.mods["offset"]
            for k in range(d.n):
                w.s[d.idx + k] = int(self.mem[a + 4 * k:a + 4 * k + 4].view(U32)[0])
        elif ins.klass == "wait":
            if ins.mods["lgkmcnt"] is not None:
                self.retire_lds(w, ins.mods["lgkmcnt"])
            if ins.mods["vmcnt"] is not None:
                self.retire_vm(w, ins.mods["vmcnt"])
        elif ins.klass == "nop":
            pass
        elif ins.klass == "barrier":
            w.at_barrier = True
        elif ins.klass == "end":
            self.retire_vm(w, 0)
            w.done = True
        elif ins.klass == "branch":
            take = True
            if n == "s_cbranch_scc0":
                take = w.scc == 0
            elif n == "s_cbranch_scc1":
                take = w.scc == 1
            elif n == "s_cbranch_vccz":
                take = (int(w.s[VCC_LO]) | int(w.s[VCC_LO + 1])) == 0
            elif n == "s_cbranch_vccnz":
                take = (int(w.s[VCC_LO]) | int(w.s[VCC_LO + 1])) != 0
            if take:
                nxt = self.labels[ins.mods["target"]]
        elif ins.klass == "call":
            tgt = int(w.s[s[0].idx])
            w.s[d.idx] = nxt
            w.s[d.idx + 1] = 0
            nxt = tgt
        elif ins.klass == "ret":
            nxt = int(w.s[s[0].idx])
        elif ins.klass == "ds":
            self.exec_ds(w, ins)
        elif ins.klass == "vmem":
            self.exec_vmem(w, ins)
        else:
            raise NotImplementedError(n)
        w.pc = nxt

    # ------------------------------------------------------------------ VALU
    def exec_valu(self, w, ins):
        n, d, s = ins.name, ins.dst, ins.src
        rd, rf = self.rd, self.rf
        if n == "v_mov_b32":
            self.wv(w, d, rd(w, s[0]))
        elif n in ("v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_accvgpr_mov_b32"):
            self.wv(w, d, rd(w, s[0]))
        elif n == "v_add_f32":
            self.wf(w, d, rf(w, s[0]) + rf(w, s[1]))
        elif n == "v_pk_add_f32":
            for k in range(2):
                self.wf(w, d[k], rf(w, s[0][k]) + rf(w, s[1][k]))
        elif n == "v_sub_f32":
            self.wf(w, d, rf(w, s[0]) - rf(w, s[1]))
        elif n == "v_mul_f32":
            with np.errstate(all="ignore"):
                self.wf(w, d, rf(w, s[0]) * rf(w, s[1]))
        elif n == "v_max_f32":
            self.wf(w, d, np.fmax(rf(w, s[0]), rf(w, s[1])))
        elif n == "v_max3_f32":
            self.wf(w, d, np.fmax(np.fmax(rf(w, s[0]), rf(w, s[1])), rf(w, s[2])))
        elif n == "v_exp_f32":
            with np.errstate(all="ignore"):
                self.wf(w, d, np.exp2(rf(w, s[0]).astype(np.float64)).astype(F32))
        elif n == "v_log_f32":
            with np.errstate(all="ignore"):
                self.wf(w, d, np.log2(rf(w, s[0]).astype(np.float64)).astype(F32))
        elif n == "v_rcp_f32":
            with np.errstate(all="ignore"):
                self.wf(w, d, (1.0 / rf(w, s[0]).astype(np.float64)).astype(F32))
        elif n == "v_cvt_pk_bf16_f32":
            self.wv(w, d, bf16_round(rf(w, s[0])) | (bf16_round(rf(w, s[1])) << 16))
        elif n == "v_cvt_pk_f16_f32":
            with np.errstate(all="ignore"):
                self.wv(w, d, f16_round(rf(w, s[0])) | (f16_round(rf(w, s[1])) << 16))
        elif n == "v_add_u32":
            self.wv(w, d, rd(w, s[0]) + rd(w, s[1]))
        elif n == "v_sub_u32":
            self.wv(w, d, rd(w, s[0]) - rd(w, s[1]))
        elif n == "v_lshlrev_b32":
            self.wv(w, d, rd(w, s[1]) << (rd(w, s[0]) & 31))
        elif n == "v_lshrrev_b32":
            self.wv(w, d, rd(w, s[1]) >> (rd(w, s[0]) & 31))
        elif n == "v_and_b32":
            self.wv(w, d, rd(w, s[0]) & rd(w, s[1]))
        elif n == "v_or_b32":
            self.wv(w, d, rd(w, s[0]) | rd(w, s[1]))
        elif n == "v_xor_b32":
            self.wv(w, d, rd(w, s[0]) ^ rd(w, s[1]))
        elif n == "v_lshl_add_u32":
            self.wv(w, d, (rd(w, s[0]) << (rd(w, s[1]) & 31)) + rd(w, s[2]))
        elif n == "v_lshl_or_b32":
            self.wv(w, d, (rd(w, s[0]) << (rd(w, s[1]) & 31)) | rd(w, s[2]))
        elif n == "v_mul_lo_u32":
            self.wv(w, d, (rd(w, s[0]).astype(np.uint64) * rd(w, s[1]).astype(np.uint64)).astype(U32))
        elif n == "v_mul_u32_u24":
            self.wv(w, d, ((rd(w, s[0]) & 0xFFFFFF).astype(np.uint64) * (rd(w, s[1]) & 0xFFFFFF).astype(np.uint64)).astype(U32))
        elif n == "v_mad_u32_u24":
            self.wv(w, d, ((rd(w, s[0]) & 0xFFFFFF).astype(np.uint64) * (rd(w, s[1]) & 0xFFFFFF).astype(np.uint64)
                           + rd(w, s[2]).astype(np.uint64)).astype(U32))
        elif n == "v_bfe_u32":
            off, wd = self.rs_any(w, s[1]), self.rs_any(w, s[2])
            self.wv(w, d, (rd(w, s[0]) >> U32(off & 31)) & U32((1 << (wd & 31)) - 1))
        elif n.startswith("v_cmp_"):
            _, _, cond, ty = n.split("_")
            a, b = (rf(w, s[0]), rf(w, s[1])) if ty == "f32" else ((rd(w, s[0]).view(np.int32), rd(w, s[1]).view(np.int32)) if ty == "i32" else (rd(w, s[0]), rd(w, s[1])))
            with np.errstate(all="ignore"):
                res = {"lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b, "eq": a == b, "ne": a != b, "lg": a != b,
                       "nlt": ~(a < b), "nge": ~(a >= b), "ngt": ~(a > b), "nle": ~(a <= b)}[cond]
            m = 0
            for l in range(64):
                if res[l]:
                    m |= 1 << l
            w.s[VCC_LO] = m & 0xFFFFFFFF
            w.s[VCC_LO + 1] = m >> 32
        elif n == "v_cndmask_b32":
            m = int(w.s[VCC_LO]) | (int(w.s[VCC_LO + 1]) << 32)
            sel = np.array([(m >> l) & 1 for l in range(64)], dtype=bool)
            self.wv(w, d, np.where(sel, rd(w, s[1]), rd(w, s[0])))
        elif n == "v_readfirstlane_b32":
            w.s[d.idx] = rd(w, s[0])[0]
        elif n == "v_mbcnt_lo_u32_b32":
            m, add = self.rs_any(w, s[0]), rd(w, s[1])
            self.wv(w, d, np.array([bin(m & ((1 << min(l, 32)) - 1)).count("1") for l in range(64)], dtype=U32) + add)
        elif n == "v_mbcnt_hi_u32_b32":
            m, add = self.rs_any(w, s[0]), rd(w, s[1])
            self.wv(w, d, np.array([bin(m & ((1 << max(l - 32, 0)) - 1)).count("1") for l in range(64)], dtype=U32) + add)
        elif n == "v_permlane16_swap_b32":
            a, b = rd(w, s[0]), rd(w, s[1])
            na, nb = a.copy(), b.copy()
            na[16:32], nb[0:16] = b[0:16], a[16:32]
            na[48:64], nb[32:48] = b[32:48], a[48:64]
            self.wv(w, s[0], na)
            self.wv(w, s[1], nb)
        elif n == "v_permlane32_swap_b32":
            a, b = rd(w, s[0]), rd(w, s[1])
            na, nb = a.copy(), b.copy()
            na[32:] = b[:32]
            nb[:32] = a[32:]
            self.wv(w, s[0], na)
            self.wv(w, s[1], nb)
        else:
            raise NotImplementedError(n)

    def rs_any(self, w, o):
        if isinstance(o, R) and o.kind == "s":
            return int(w.s[o.idx])
        if isinstance(o, Lit):
            return o.bits
        if isinstance(o, int):
            return o & 0xFFFFFFFF
        raise TypeError(o)

    # ------------------------------------------------------------------ MFMA 32x32x16
    def frag(self, w, r):
        """4 registers x 64 lanes of packed 16-bit -> f32 [64 lanes][8 elements]"""
        blk = self.vblock(w, r)                      # [4][64]
        lo, hi = blk & 0xFFFF, blk >> 16
        el = np.stack([lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lo[3], hi[3]], axis=1)   # [64][8]
        return bf16_to_f32(el) if self.dtype == "bf16" else f16_to_f32(el)

    def exec_mfma16(self, w, ins):
        """v_mfma_f32_16x16x32: D[m][n] (lane: n = lane & 15, m = 4 (lane >> 4) + r) = C + sum_k A[m][k] B[n][k]"""
        d, (a, b, c) = ins.dst, ins.src
        fa, fb = self.frag(w, a), self.frag(w, b)                 # [64 lanes][8]
        Am = np.concatenate([fa[16 * g:16 * g + 16] for g in range(4)], axis=1)     # [16 rows][32 k]
        Bm = np.concatenate([fb[16 * g:16 * g + 16] for g in range(4)], axis=1)     # [16 cols][32 k]
        if isinstance(c, R):
            cb = self.vblock(w, c).view(F32).copy()               # [4][64]
        else:
            cb = np.full((4, 64), np.array([c.bits if isinstance(c, Lit) else c], dtype=U32).view(F32)[0], dtype=F32)
        with np.errstate(all="ignore"):
            prod = Am.astype(np.float64) @ Bm.astype(np.float64).T                  # [m][n]
            ll = np.arange(64)
            out = np.stack([(cb[r].astype(np.float64) + prod[4 * (ll >> 4) + r, ll & 15]).astype(F32) for r in range(4)])
        self.vblock(w, d)[:] = out.view(U32)

    def exec_mfma(self, w, ins):
        d, (a, b, c) = ins.dst, ins.src
        self.mfma_count += 1
        if "16x16x32" in ins.name:
            return self.exec_mfma16(w, ins)
        fa, fb = self.frag(w, a), self.frag(w, b)
        Am = np.concatenate([fa[:32], fa[32:]], axis=1)          # [32 rows][16 k]
        Bm = np.concatenate([fb[:32], fb[32:]], axis=1).T        # [16 k][32 cols]
        if isinstance(c, R):
            cb = self.vblock(w, c).view(F32).copy()
        else:
            cb = np.full((16, 64), np.array([c.bits if isinstance(c, Lit) else c], dtype=U32).view(F32)[0], dtype=F32)
        with np.errstate(all="ignore"):
            prod = (Am.astype(np.float64) @ Bm.astype(np.float64))
            out = np.zeros((16, 64), dtype=F32)
            rr = np.arange(16)
            for h in range(2):
                rows = (rr & 3) + 8 * (rr >> 2) + 4 * h
                out[:, 32 * h:32 * h + 32] = (cb[:, 32 * h:32 * h + 32].astype(np.float64) + prod[rows, :]).astype(F32)
        self.vblock(w, d)[:] = out.view(U32)

    # ------------------------------------------------------------------ SALU
    def exec_salu(self, w, ins):
        n, d, s = ins.name, ins.dst, ins.src
        rs = self.rs
        M = 0xFFFFFFFF
        if n == "s_mov_b32":
            w.s[d.idx] = rs(w, s[0])
        elif n == "s_mov_b64":
            if isinstance(s[0], R):
                w.s[d.idx], w.s[d.idx + 1] = w.s[s[0].idx], w.s[s[0].idx + 1]
            else:
                w.s[d.idx], w.s[d.idx + 1] = rs(w, s[0]), 0
        elif n == "s_add_u32":
            if "target" in ins.mods:
                b = (self.labels[ins.mods["target"]] - self.labels[ins.mods["anchor"]]) & M
            else:
                b = rs(w, s[1])
            r = rs(w, s[0]) + b
            w.s[d.idx], w.scc = r & M, int(r > M)
        elif n == "s_addc_u32":
            r = rs(w, s[0]) + rs(w, s[1]) + w.scc
            w.s[d.idx], w.scc = r & M, int(r > M)
        elif n == "s_sub_u32":
            a, b = rs(w, s[0]), rs(w, s[1])
            w.s[d.idx], w.scc = (a - b) & M, int(b > a)
        elif n == "s_subb_u32":
            a, b = rs(w, s[0]), rs(w, s[1]) + w.scc
            w.s[d.idx], w.scc = (a - b) & M, int(b > a)
        elif n == "s_mul_i32":
            w.s[d.idx] = (rs(w, s[0]) * rs(w, s[1])) & M
        elif n == "s_mul_hi_u32":
            w.s[d.idx] = (rs(w, s[0]) * rs(w, s[1])) >> 32
        elif n == "s_lshl_b32":
            r = (rs(w, s[0]) << (rs(w, s[1]) & 31)) & M
            w.s[d.idx], w.scc = r, int(r != 0)
        elif n == "s_lshr_b32":
            r = rs(w, s[0]) >> (rs(w, s[1]) & 31)
            w.s[d.idx], w.scc = r, int(r != 0)
        elif n == "s_and_b32":
            r = rs(w, s[0]) & rs(w, s[1])
            w.s[d.idx], w.scc = r, int(r != 0)
        elif n == "s_or_b32":
            r = rs(w, s[0]) | rs(w, s[1])
            w.s[d.idx], w.scc = r, int(r != 0)
        elif n == "s_min_u32":
            a, b = rs(w, s[0]), rs(w, s[1])
            w.s[d.idx], w.scc = min(a, b), int(a <= b)
        elif n == "s_max_i32":
            a, b = rs(w, s[0]), rs(w, s[1])
            sa, sb = a - (1 << 32) if a >> 31 else a, b - (1 << 32) if b >> 31 else b
            w.s[d.idx], w.scc = (max(sa, sb)) & M, int(sa >= sb)
        elif n == "s_cselect_b32":
            w.s[d.idx] = rs(w, s[0]) if w.scc else rs(w, s[1])
        elif n.startswith("s_cmp_"):
            _, _, cond, ty = n.split("_")
            a, b = rs(w, s[0]), rs(w, s[1])
            if ty == "i32":
                a, b = (a - (1 << 32) if a >> 31 else a), (b - (1 << 32) if b >> 31 else b)
            w.scc = int({"eq": a == b, "lg": a != b, "lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b}[cond])
        elif n == "s_getpc_b64":
            w.s[d.idx], w.s[d.idx + 1] = w.pc + 1, 0
        else:
            raise NotImplementedError(n)

    # ------------------------------------------------------------------ LDS reads
    def lds_note_read(self, w, addrs, size):
        for a in addrs:
            a = int(a)
            sl = slice(a, a + size)
            if (self.lds_inflight[sl] > 0).any():
                self.viol(w, f"LDS read of bytes [{a},{a + size}) while an LDS-DMA onto them is in flight")
            bad = (self.lds_pub_epoch[sl] == self.epoch) & (self.lds_pub_wave[sl] != w.wid)
            if bad.any():
                self.viol(w, f"LDS read of bytes [{a},{a + size}) that another wave's DMA retired in the same barrier interval")
            same = self.lds_read_epoch[sl] == self.epoch
            self.lds_read_wave[sl] = np.where(same & (self.lds_read_wave[sl] != w.wid), -2, w.wid)
            self.lds_read_epoch[sl] = self.epoch

    def exec_ds(self, w, ins):
        n, d, s = ins.name, ins.dst, ins.src
        addr = self.rd(w, s[0]).astype(np.int64) + ins.mods["offset"]
        if n == "ds_read_b128":
            if (addr & 15).any():
                self.viol(w, "ds_read_b128 address not 16-byte aligned")
            self.lds_note_read(w, addr, 16)
            blk = self.vblock(w, d)
            for l in range(64):
                blk[:, l] = self.lds[addr[l]:addr[l] + 16].view(U32)
        elif n == "ds_read_b64_tr_b16":
            if (addr & 7).any():
                self.viol(w, "ds_read_b64_tr_b16 address not 8-byte aligned")
            self.lds_note_read(w, addr, 8)
            blk = self.vblock(w, d)
            for g in range(4):
                block = np.zeros((4, 16), dtype=np.uint16)
                for q in range(4):
                    for p in range(4):
                        a = addr[16 * g + 4 * q + p]
                        block[q, 4 * p:4 * p + 4] = self.lds[a:a + 8].view(np.uint16)
                for i in range(16):
                    col = block[:, i].astype(U32)
                    blk[0, 16 * g + i] = col[0] | (col[1] << 16)
                    blk[1, 16 * g + i] = col[2] | (col[3] << 16)
        else:
            raise NotImplementedError(n)
        regs = d.ids()
        w.lds_q.append(regs)
        for r in regs:
            w.pending[r] = f"`{ins.text}`"

    # ------------------------------------------------------------------ vector memory
    def exec_vmem(self, w, ins):
        n, d, s = ins.name, ins.dst, ins.src
        if n == "buffer_load_dwordx4":
            voff, rsrc, soff = s
            addr, ok = self.buf_addr(w, rsrc, self.rd(w, voff), self.rs(w, soff), ins.mods["offset"], 16)
            blk = self.vblock(w, d)
            for l in range(64):
                blk[:, l] = self.mem[addr[l]:addr[l] + 16].view(U32) if ok[l] else 0
            regs = d.ids()
            w.vm_q.append({"kind": "load", "regs": regs})
            for r in regs:
                w.pending[r] = f"`{ins.text}`"
        elif n == "buffer_load_dword":
            voff, rsrc, soff = s
            addr, ok = self.buf_addr(w, rsrc, self.rd(w, voff), self.rs(w, soff), ins.mods["offset"], 4)
            blk = self.vblock(w, d)
            for l in range(64):
                blk[0, l] = self.mem[addr[l]:addr[l] + 4].view(U32)[0] if ok[l] else 0
            w.vm_q.append({"kind": "load", "regs": []})          # (the prefetch form: the destination is never read, so nothing is pending on it)
        elif n == "buffer_load_lds_dwordx4":
            voff, rsrc, soff = s
            addr, ok = self.buf_addr(w, rsrc, self.rd(w, voff), self.rs(w, soff), 0, 16)
            m0 = int(w.s[M0_IDX]) & 0x3FFFF
            data = np.zeros(1024, dtype=np.uint8)
            for l in range(64):
                if ok[l]:
                    data[16 * l:16 * l + 16] = self.mem[addr[l]:addr[l] + 16]
            sl = slice(m0, m0 + 1024)
            other = (self.lds_read_epoch[sl] == self.epoch) & (self.lds_read_wave[sl] != w.wid)
            if other.any():
                self.viol(w, f"LDS-DMA issued onto bytes [{m0},{m0 + 1024}) that another wave has read since the last barrier")
            self.lds_inflight[sl] += 1

            def apply(sl=sl, data=data, wid=w.wid):
                self.lds[sl] = data
                self.lds_inflight[sl] -= 1
                self.lds_pub_epoch[sl] = self.epoch
                self.lds_pub_wave[sl] = wid
            w.vm_q.append({"kind": "dma", "apply": apply})
        elif n == "global_store_dwordx4":
            addr, data = s
            lo, hi = self.rd(w, addr[0]).astype(np.int64), self.rd(w, addr[1]).astype(np.int64)
            blk = self.vblock(w, data)
            for l in range(64):
                a = int(lo[l] | (hi[l] << 32))
                self.mem[a:a + 16] = np.ascontiguousarray(blk[:, l]).view(np.uint8)
            w.vm_q.append({"kind": "store"})
        elif n in ("buffer_store_dwordx4", "buffer_store_dword"):
            data, voff, rsrc, soff = s
            size = 16 if n.endswith("x4") else 4
            addr, ok = self.buf_addr(w, rsrc, self.rd(w, voff), self.rs(w, soff), ins.mods["offset"], size)
            blk = self.vblock(w, data)
            for l in range(64):
                if ok[l]:
                    self.mem[addr[l]:addr[l] + size] = np.ascontiguousarray(blk[:, l]).view(np.uint8)
            w.vm_q.append({"kind": "store"})
        else:
            raise NotImplementedError(n)
        if len(w.vm_q) > 63:
            self.viol(w, "more than 63 vector-memory operations outstanding")

    # ------------------------------------------------------------------ run
    def run(self, entry, setup, max_steps=50_000_000):
        """setup(wave) initialises the SGPRs/VGPRs the launch provides"""
        for w in self.waves:
            w.pc = self.labels[entry]
            setup(w)
        steps = 0
        while True:
            alive = [w for w in self.waves if not w.done]
            if not alive:
                break
            progressed = False
            for w in alive:
                while not w.done and not w.at_barrier:
                    self.step(w)
                    steps += 1
                    progressed = True
                    if steps > max_steps:
                        raise RuntimeError("simulation step limit")
            if all(w.at_barrier or w.done for w in self.waves):
                waiting = [w for w in self.waves if w.at_barrier]
                if waiting and any(w.done for w in self.waves) and len(waiting) != len([w for w in self.waves if not w.done]):
                    raise RuntimeError("barrier mismatch")
                for w in waiting:
                    w.at_barrier = False
                self.epoch += 1
                if not waiting and not progressed:
                    break
        return steps
