#!/usr/bin/env python3
"""Generator of the second hand-placed attention-forward kernel for gfx950 (ucod_attn_fwd_pw32): the pipeline of gen_attn.py with TWO waves
per SIMD -- a workgroup = 8 waves x 32 query rows (the same 256-row work items, the same K/V traffic), at most 256 registers per wave.

Why (measured on the one-wave-per-SIMD kernel, profiles/r04_attention_asm_ablation.txt): a single in-order wave serialises everything it
issues -- 1150 vector-issue cycles per 64-key tile (64 v_exp_f32, 64 adds, 32 packs, 32 MFMAs) PLUS ~350 cycles of LDS reads, LDS-DMA, scalar
bookkeeping, waits and the barrier; every component alone hides under the matrix pipe, together they add.  With a second wave on the SIMD
the non-vector instructions of one wave issue under the vector instructions of the other, and a wave whose 32 rows lie past the last token
skips the arithmetic of the steady tiles altogether (its SIMD partner then runs alone).

Per wave: unit = (tile t, 32-key block kt), i = 2 t + kt.  Step i issues   PV(i-1) x4,  QK(i+1) x4   with SM(i) (16 v_exp_f32, 16 adds into
two per-lane partial sums, 8 packs), the LDS fragment reads and one LDS-DMA piece in the gaps; an iteration = steps 2t-1, 2t (all Q K^T
products of tile t in one iteration; the last tile's carry the key-mask MFMA).  Fragment reads of a tile all happen in its own iteration
(V after the P V products that consumed the previous contents), so the ring protocol is the 4-slot / 3-tiles-ahead one of gen_attn.py.
Q of the next item is fetched by LDS-DMA into a per-wave 4-KiB area during the first tile and read back at the end of the last one.
Reference lines: transformers modeling_dinov2.py:153-179, models/backbones/dino.py:96-120 (same contract as attn_fwd_v5_kernel).
"""
import argparse
import os
import sys

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from tools.attn_asm.isa import Prog, V, A, S, VCC, M0, F, I   # noqa: E402
    from tools.attn_asm.gen_attn import HEADER, FOOTER              # noqa: E402
else:
    from .isa import Prog, V, A, S, VCC, M0, F, I
    from .gen_attn import HEADER, FOOTER

KERNEL_NAME = "ucod_attn_fwd_pw32"
SLOT = 8192
RING = 4
LEAD = 3
V_RING = RING * SLOT
Q_AREA = 2 * RING * SLOT            # 8 waves x 4 KiB
LDS_BYTES = Q_AREA + 8 * 4096
KARG_BYTES = 72
NWAVES = 8

s_karg = S(0, 2)
s_wg = S(2)
s_qkv, s_out, s_lse = S(4, 2), S(6, 2), S(8, 2)
s_N, s_heads, s_npairs, s_nqb, s_mg_nqb, s_mg_heads, s_nt, s_stride = (S(10 + k) for k in range(8))
s_w, s_xcd, s_D2, s_ld2, s_TS, s_items, s_j, s_imgbytes = (S(18 + k) for k in range(8))
s_a1_nqb, s_a1_heads = S(26), S(27)
s_desc_kv = S(28, 4)
s_dma_next, s_jdma, s_kcol, s_vcol, s_slot_r, s_m0base, s_dead, s_loop = (S(32 + k) for k in range(8))
s_desc_q = S(40, 4)
s_qcol, s_phantom, s_ocol, s_ocol_cur = S(44), S(45), S(46), S(47)
s_desc_o, s_desc_l, s_desc_o_cur, s_desc_l_cur = S(48, 4), S(52, 4), S(56, 4), S(60, 4)
s_t = [S(64 + k) for k in range(16)]
s_slot_w = S(80)
s_qarea = S(81)          # LDS byte address of this wave's Q area
s_dead_cur = S(82)
s_stamp = [S(84 + 2 * k, 2) for k in range(4)]
s_sacc = [S(92 + k) for k in range(4)]
s_dbg = S(100, 2)

v_tid, v_lane, v_l31, v_h5 = V(0), V(1), V(2), V(3)
v_koffb, v_ka = V(4, 4), V(8, 4)
v_voffb, v_va = V(12, 2), V(14, 2)
v_dma_k, v_dma_v, v_qoff, v_ooff, v_lseoff = V(16), V(17), V(18), V(19), V(20)
v_l = V(26, 2)          # the two partial row sums: a 64-bit aligned pair (v_pk_add_f32)
v_m = V(23)
v_qdma = V(24)
v_qrd = V(25)            # LDS address of this lane's Q fragment (sd = 0)
v_E = V(28, 8)
v_P = [V(36, 8), V(44, 8)]
v_negm = V(52, 16)
v_S = [V(68, 16), V(84, 16)]
v_KF = [V(100, 16), V(116, 16)]
v_VF = [V(132, 16), V(148, 16)]
v_ep = V(164, 8)
v_wq = [V(172, 4), V(176, 4)]
v_inv, v_lg = V(180), V(181)
v_rs = V(182, 5)
v_x = V(164, 8)
ARCH_VGPRS = 188

a_O = [A(0, 16), A(16, 16)]
a_Q = A(32, 16)
a_mask = [A(48, 4), A(52, 4)]
a_onesB = A(56, 4)
ACC_VGPRS = 60


class Gen32:
    def __init__(self, dtype="bf16", margin=64, abl=(), dma_gap=2):
        self.p = Prog()
        self.dtype = dtype
        self.margin = margin
        self.abl = set(abl)
        self.dma_gap = dma_gap
        self.lds_issued = 0
        self.frag_last = {}
        self.issue_rows = []
        self.lds_bytes = LDS_BYTES
        self.stamps = "stamps" in self.abl
        self.skip_dead = "noskip" not in self.abl

    # ------------------------------------------------------------------ helpers (as gen_attn.py)
    def cvt_pk(self, d, a, b):
        (self.p.v_cvt_pk_bf16_f32 if self.dtype == "bf16" else self.p.v_cvt_pk_f16_f32)(d, a, b)

    def mfma(self, d, a, b, c):
        if "nomfma" in self.abl:
            return None
        return self.p.mfma(d, a, b, c, dtype=self.dtype)

    def one16(self):
        return 0x3F80 if self.dtype == "bf16" else 0x3C00

    def negbig16(self):
        return 0xFF7F if self.dtype == "bf16" else 0xFBFF

    def lds_read(self, kind, dst, addr, offset, buf):
        if "nolds" in self.abl:
            return
        (self.p.ds_read_b128 if kind == "b128" else self.p.ds_read_b64_tr_b16)(dst, addr, offset)
        self.frag_last[buf] = self.lds_issued
        self.lds_issued += 1

    def wait_frag(self, buf):
        if buf not in self.frag_last:
            return
        k = self.lds_issued - (self.frag_last[buf] + 1)
        self.p.s_waitcnt(lgkmcnt=k)
        last = self.frag_last[buf]
        for b in list(self.frag_last):
            if self.frag_last[b] <= last:
                del self.frag_last[b]

    def lds_all_done(self):
        self.lds_issued, self.frag_last = 0, {}

    def ptr_add_mul(self, dst, base, a, b, t0, t1):
        p = self.p
        p.s_mul_i32(t0, a, b)
        p.s_mul_hi_u32(t1, a, b)
        p.s_add_u32(dst[0], base[0], t0)
        p.s_addc_u32(dst[1], base[1], t1)

    def item_decode(self, j, pair, qblk, b, head):
        p = self.p
        t = s_t[15]
        p.s_mul_hi_u32(t, j, s_mg_nqb)
        p.s_mul_i32(qblk, j, s_a1_nqb)
        p.s_add_u32(t, t, qblk)
        p.s_mul_i32(qblk, t, s_nqb)
        p.s_sub_u32(qblk, j, qblk)
        p.s_lshl_b32(pair, t, 3)
        p.s_add_u32(pair, pair, s_xcd)
        p.s_mul_hi_u32(b, pair, s_mg_heads)
        p.s_mul_i32(head, pair, s_a1_heads)
        p.s_add_u32(b, b, head)
        p.s_mul_i32(head, b, s_heads)
        p.s_sub_u32(head, pair, head)

    def make_kv_desc(self, j):
        p = self.p
        pair, qblk, b, head = s_t[0], s_t[1], s_t[2], s_t[3]
        self.item_decode(j, pair, qblk, b, head)
        self.ptr_add_mul(s_desc_kv, s_qkv, b, s_imgbytes, s_t[4], s_t[5])
        p.s_and_b32(s_desc_kv[1], s_desc_kv[1], I(0xFFFF))
        p.s_cmp("lt", "u32", j, s_items)
        p.s_cselect_b32(s_desc_kv[2], s_imgbytes, I(0))
        p.s_lshl_b32(s_t[4], head, 7)
        p.s_add_u32(s_kcol, s_t[4], s_D2)
        p.s_add_u32(s_vcol, s_kcol, s_D2)
        p.s_mov_b32(s_dma_next, I(0))

    def rows_of(self, j, q0, rows, qblk):
        """q0 = 256 qblk + 32 w, rows = max(N - q0, 0) (0 for an item past the list)"""
        p = self.p
        p.s_lshl_b32(q0, qblk, 8)
        p.s_lshl_b32(s_t[4], s_w, 5)
        p.s_add_u32(q0, q0, s_t[4])
        p.s_sub_u32(rows, s_N, q0)
        p.s_max_i32(rows, rows, I(0))
        p.s_cmp("lt", "u32", j, s_items)
        p.s_cselect_b32(rows, rows, I(0))

    def make_q_desc(self, j):
        p = self.p
        pair, qblk, b, head = s_t[0], s_t[1], s_t[2], s_t[3]
        self.item_decode(j, pair, qblk, b, head)
        q0, rows = s_t[6], s_t[7]
        self.rows_of(j, q0, rows, qblk)
        self.ptr_add_mul([s_t[8], s_t[9]], s_qkv, b, s_imgbytes, s_t[4], s_t[5])
        self.ptr_add_mul(s_desc_q, [s_t[8], s_t[9]], q0, s_ld2, s_t[4], s_t[5])
        p.s_and_b32(s_desc_q[1], s_desc_q[1], I(0xFFFF))
        p.s_mul_i32(s_desc_q[2], rows, s_ld2)
        p.s_lshl_b32(s_qcol, head, 7)

    def make_out_desc(self, j):
        p = self.p
        pair, qblk, b, head = s_t[0], s_t[1], s_t[2], s_t[3]
        self.item_decode(j, pair, qblk, b, head)
        q0, rows = s_t[6], s_t[7]
        self.rows_of(j, q0, rows, qblk)
        p.s_cmp("eq", "u32", rows, I(0))
        p.s_cselect_b32(s_dead_cur, I(1), I(0))          # no live query row in this wave's block: the steady tiles skip the arithmetic
        p.s_mul_i32(s_t[8], b, s_N)
        p.s_add_u32(s_t[8], s_t[8], q0)
        self.ptr_add_mul(s_desc_o_cur, s_out, s_t[8], s_D2, s_t[4], s_t[5])
        p.s_and_b32(s_desc_o_cur[1], s_desc_o_cur[1], I(0xFFFF))
        p.s_mul_i32(s_desc_o_cur[2], rows, s_D2)
        p.s_lshl_b32(s_ocol_cur, head, 7)
        p.s_mul_i32(s_t[8], pair, s_N)
        p.s_add_u32(s_t[8], s_t[8], q0)
        self.ptr_add_mul(s_desc_l_cur, s_lse, s_t[8], I(4), s_t[4], s_t[5])
        p.s_and_b32(s_desc_l_cur[1], s_desc_l_cur[1], I(0xFFFF))
        p.s_lshl_b32(s_desc_l_cur[2], rows, 2)
        p.s_or_b32(s_t[4], s_lse[0], s_lse[1])
        p.s_cmp("eq", "u32", s_t[4], I(0))
        p.s_cselect_b32(s_desc_l_cur[2], I(0), s_desc_l_cur[2])

    def dma_advance(self):
        p = self.p
        lab_sw, lab_done = p.newlabel("dma_switch"), p.newlabel("dma_adv_done")
        p.s_add_u32(s_dma_next, s_dma_next, I(1))
        p.s_cmp("eq", "u32", s_dma_next, s_nt)
        p.s_cbranch("scc1", lab_sw)
        p.s_add_u32(s_desc_kv[0], s_desc_kv[0], s_TS)
        p.s_addc_u32(s_desc_kv[1], s_desc_kv[1], I(0))
        p.s_sub_u32(s_desc_kv[2], s_desc_kv[2], s_TS)
        p.s_cselect_b32(s_desc_kv[2], I(0), s_desc_kv[2])
        p.label(lab_done)
        return lab_sw, lab_done

    def dma_switch_block(self, lab_sw, lab_done):
        p = self.p
        p.label(lab_sw)
        p.s_add_u32(s_jdma, s_jdma, s_stride)
        self.make_kv_desc(s_jdma)
        p.s_branch(lab_done)

    def dma_piece(self, which):
        """this wave's 1-KiB piece (rows 8w .. 8w+7) of the stream's current tile"""
        p = self.p
        if "nodma" in self.abl:
            return lambda: None
        p.s_add_u32(M0, s_m0base, I(V_RING if which == "v" else 0))
        return lambda: p.buffer_load_lds_dwordx4(v_dma_k if which == "k" else v_dma_v, s_desc_kv, s_kcol if which == "k" else s_vcol)

    def slot_heads(self):
        """scalar part of an iteration's ring bookkeeping (the dead-wave loop runs only this)"""
        p = self.p
        p.s_add_u32(s_slot_w, s_slot_r, I(LEAD * SLOT))
        p.s_and_b32(s_slot_w, s_slot_w, I(RING * SLOT - 1))
        p.s_lshl_b32(s_m0base, s_w, 10)
        p.s_add_u32(s_m0base, s_m0base, s_slot_w)

    # ------------------------------------------------------------------ prologue
    def prologue(self):
        p = self.p
        p.label(KERNEL_NAME)
        p.s_load(S(4, 8), s_karg, 0)
        p.s_load(S(12, 4), s_karg, 32)
        p.s_load(S(16, 2), s_karg, 48)
        p.s_load(S(26, 2), s_karg, 56)
        if self.stamps:
            p.s_load(s_dbg, s_karg, 64)
            for k in range(4):
                p.s_mov_b32(s_sacc[k], I(0))
        p.s_waitcnt(lgkmcnt=0)
        p.v_and_b32(v_lane, I(63), v_tid)
        p.v_lshrrev_b32(v_x[0], I(6), v_tid)
        p.s_nop(0)
        p.v_readfirstlane_b32(s_w, v_x[0])
        p.v_and_b32(v_l31, I(31), v_lane)
        p.v_lshrrev_b32(v_h5, I(5), v_lane)
        p.s_lshl_b32(s_D2, s_heads, 7)
        p.s_mul_i32(s_ld2, s_D2, I(3))
        p.s_lshl_b32(s_TS, s_ld2, 6)
        p.s_mul_i32(s_imgbytes, s_N, s_ld2)
        p.s_and_b32(s_xcd, s_wg, I(7))
        p.s_lshr_b32(s_j, s_wg, 3)
        p.s_add_u32(s_t[0], s_npairs, I(7))
        p.s_sub_u32(s_t[0], s_t[0], s_xcd)
        p.s_lshr_b32(s_t[0], s_t[0], 3)
        p.s_mul_i32(s_items, s_t[0], s_nqb)
        p.s_cmp("ge", "u32", s_j, s_items)
        lab_go = p.newlabel("go")
        p.s_cbranch("scc0", lab_go)
        p.s_endpgm()
        p.label(lab_go)
        p.s_lshl_b32(s_qarea, s_w, 12)
        p.s_add_u32(s_qarea, s_qarea, I(Q_AREA))
        # K fragment offsets
        p.v_bfe_u32(v_x[0], v_l31, I(1), I(3))
        for sd in range(4):
            p.v_or_b32(v_x[1], I(2 * sd), v_h5)
            p.v_xor_b32(v_x[1], v_x[1], v_x[0])
            p.v_lshlrev_b32(v_x[1], I(4), v_x[1])
            p.v_lshl_add_u32(v_koffb[sd], v_l31, I(7), v_x[1])
        # V transposed-read offsets
        p.v_and_b32(v_x[0], I(15), v_lane)
        p.v_bfe_u32(v_x[1], v_lane, I(4), I(1))
        p.v_lshrrev_b32(v_x[2], I(2), v_x[0])
        p.v_lshl_add_u32(v_x[2], v_h5, I(2), v_x[2])
        p.v_and_b32(v_x[3], I(3), v_x[0])
        p.v_lshlrev_b32(v_x[3], I(2), v_x[3])
        p.v_lshl_add_u32(v_x[3], v_x[1], I(4), v_x[3])
        p.v_bfe_u32(v_x[4], v_x[2], I(1), I(1))
        p.v_lshlrev_b32(v_x[4], I(2), v_x[4])
        for dt in range(2):
            p.v_add_u32(v_x[5], I(32 * dt), v_x[3])
            p.v_lshrrev_b32(v_x[6], I(3), v_x[5])
            p.v_xor_b32(v_x[6], v_x[6], v_x[4])
            p.v_lshlrev_b32(v_x[6], I(4), v_x[6])
            p.v_and_b32(v_x[7], I(7), v_x[5])
            p.v_lshl_add_u32(v_x[6], v_x[7], I(1), v_x[6])
            p.v_lshl_add_u32(v_voffb[dt], v_x[2], I(7), v_x[6])
            p.v_add_u32(v_voffb[dt], I(V_RING), v_voffb[dt])
        # DMA source offsets of this wave's piece: row = 8 w + (lane >> 3), chunk = lane & 7
        p.v_lshrrev_b32(v_x[0], I(3), v_lane)
        p.v_mul_lo_u32(v_qdma, v_x[0], s_ld2)            # Q pieces: rows (lane >> 3) of an 8-row piece
        p.s_lshl_b32(s_t[0], s_w, 3)
        p.v_add_u32(v_x[0], s_t[0], v_x[0])
        p.v_and_b32(v_x[1], I(7), v_lane)
        p.v_lshl_add_u32(v_qdma, v_x[1], I(4), v_qdma)
        p.v_bfe_u32(v_x[2], v_x[0], I(1), I(3))
        p.v_xor_b32(v_x[2], v_x[2], v_x[1])
        p.v_lshlrev_b32(v_x[2], I(4), v_x[2])
        p.v_bfe_u32(v_x[3], v_x[0], I(1), I(1))
        p.v_lshlrev_b32(v_x[3], I(2), v_x[3])
        p.v_xor_b32(v_x[3], v_x[3], v_x[1])
        p.v_lshlrev_b32(v_x[3], I(4), v_x[3])
        p.v_mul_lo_u32(v_x[4], v_x[0], s_ld2)
        p.v_add_u32(v_dma_k, v_x[4], v_x[2])
        p.v_add_u32(v_dma_v, v_x[4], v_x[3])
        # Q / O / LSE offsets; LDS address of the lane's Q fragment: row l31, chunk h5 (+ 2 sd)
        p.v_mul_lo_u32(v_x[1], v_l31, s_ld2)
        p.v_lshl_add_u32(v_qoff, v_h5, I(4), v_x[1])
        p.v_mul_lo_u32(v_x[1], v_l31, s_D2)
        p.v_lshl_add_u32(v_ooff, v_h5, I(4), v_x[1])
        p.v_lshlrev_b32(v_lseoff, I(2), v_l31)
        p.v_lshl_or_b32(v_lseoff, v_h5, I(31), v_lseoff)
        p.v_lshlrev_b32(v_x[1], I(7), v_l31)
        p.v_lshl_add_u32(v_x[1], v_h5, I(4), v_x[1])
        p.v_add_u32(v_qrd, s_qarea, v_x[1])
        # constant operands of the key-mask MFMA
        one = self.one16()
        p.v_cmp("eq", "u32", I(0), v_h5)
        p.v_mov_b32(v_x[1], I(0))
        p.v_mov_b32(v_x[2], I(one))
        p.v_cndmask_b32(v_x[3], v_x[1], v_x[2])
        p.v_accvgpr_write_b32(a_onesB[0], v_x[3])
        for k in range(1, 4):
            p.v_accvgpr_write_b32(a_onesB[k], v_x[1])
        p.s_sub_u32(s_t[0], s_nt, I(1))
        p.s_lshl_b32(s_t[0], s_t[0], 6)
        p.s_sub_u32(s_t[0], s_N, s_t[0])
        p.v_mov_b32(v_x[2], I(self.negbig16()))
        for kt in range(2):
            p.s_sub_u32(s_t[1], s_t[0], I(32 * kt))
            p.v_cmp("le", "i32", s_t[1], v_l31)
            p.v_cndmask_b32(v_x[3], v_x[1], v_x[2])
            p.v_cmp("eq", "u32", I(0), v_h5)
            p.v_cndmask_b32(v_x[3], v_x[1], v_x[3])
            p.v_accvgpr_write_b32(a_mask[kt][0], v_x[3])
            for k in range(1, 4):
                p.v_accvgpr_write_b32(a_mask[kt][k], v_x[1])
        # phantom previous item
        for r in range(8):
            p.v_mov_b32(v_P[0][r], I(0))
            p.v_mov_b32(v_P[1][r], I(0))
        for r in range(16):
            p.v_mov_b32(v_S[1][r], I(0))
            p.v_mov_b32(v_VF[0][r], I(0))
            p.v_mov_b32(v_VF[1][r], I(0))
            p.v_mov_b32(v_negm[r], I(0))
        p.v_mov_b32(v_l[0], F(1.0))
        p.v_mov_b32(v_l[1], I(0))
        p.v_mov_b32(v_m, I(0))
        for dt in range(2):
            for r in range(16):
                p.v_accvgpr_write_b32(a_O[dt][r], v_x[1])
        for k in range(4):
            p.s_mov_b32(s_desc_o_cur[k], I(0))
            p.s_mov_b32(s_desc_l_cur[k], I(0))
        for d in (s_desc_o_cur, s_desc_l_cur, s_desc_kv, s_desc_q):
            p.s_mov_b32(d[3], I(0x00020000))
        p.s_mov_b32(s_ocol_cur, I(0))
        p.s_mov_b32(s_phantom, I(0))
        p.s_mov_b32(s_dead_cur, I(0))
        # first item: Q straight into the fragments, tiles 0 .. LEAD-1 requested
        self.make_q_desc(s_j)
        for sd in range(4):
            p.buffer_load_dwordx4(a_Q[4 * sd:4 * sd + 4], v_qoff, s_desc_q, s_qcol, offset=32 * sd)
        p.s_mov_b32(s_jdma, s_j)
        self.make_kv_desc(s_jdma)
        p.s_mov_b32(s_slot_r, I(0))
        switches = []
        for t in range(LEAD):
            p.s_lshl_b32(s_m0base, s_w, 10)
            p.s_add_u32(s_m0base, s_m0base, I(t * SLOT))
            for which in ("k", "v"):
                issue = self.dma_piece(which)
                p.s_nop(0)
                issue()
            switches.append(self.dma_advance())
        p.s_waitcnt(vmcnt=2)
        p.s_barrier()
        for sd in range(4):
            p.ds_read_b128(v_KF[0][4 * sd:4 * sd + 4], v_koffb[sd], 0)
        for sd in range(4):
            p.v_mov_b32(v_ka[sd], v_koffb[sd])
        p.s_waitcnt(lgkmcnt=0)
        self.lds_all_done()
        lab_items = p.newlabel("item_top")
        p.s_branch(lab_items)
        for sw, dn in switches:
            self.dma_switch_block(sw, dn)
        return lab_items

    # ------------------------------------------------------------------ pieces
    def sm_items(self, Sx, Px):
        p = self.p
        if "nosm" in self.abl:
            return []
        acc = [v_l[0], v_l[1]]

        def ex(k):
            return (lambda: p.v_exp_f32(v_E[k % 8], Sx[k]), 2, "exp")

        def ad(k):
            return (lambda: p.v_add_f32(acc[k & 1], acc[k & 1], v_E[k % 8]), 1, "add")

        def cv(w):
            return (lambda: self.cvt_pk(Px[w], v_E[(2 * w) % 8], v_E[(2 * w + 1) % 8]), 1, "cvt")
        def pad(w):          # (l0, l1) += (E[2w], E[2w+1]): the additions of ad(2w), ad(2w+1) in one packed instruction
            return (lambda: p.v_pk_add_f32(v_l, v_l, v_E[(2 * w) % 8:(2 * w) % 8 + 2]), 1, "add")
        if "pksum" not in self.abl:      # default: plain adds.  v_pk_add_f32 does not overlap with an MFMA in flight (~10 matrix-pipe cycles each): pw64 190 -> 230 us with it
            order = [ex(0), ex(1), ex(2), ad(0), ex(3), ad(1), cv(0)]
            for w in range(1, 7):
                order += [ex(2 * w + 2), ad(2 * w), ex(2 * w + 3), ad(2 * w + 1), cv(w)]
            order += [ad(14), ad(15), cv(7)]
            return order
        order = [ex(0), ex(1), ex(2), ex(3), pad(0), cv(0)]
        for w in range(1, 7):
            order += [ex(2 * w + 2), ex(2 * w + 3), pad(w), cv(w)]
        order += [pad(7), cv(7)]
        return order

    def item_start(self, Sx, Px):
        """the new item's first unit: m = max over its 32 keys + margin, -m block, probabilities, partial sums (straight-line)"""
        p = self.p
        t0, t1, mx = v_rs[0], v_rs[1], v_rs[2]
        p.v_max3_f32(t0, Sx[0], Sx[1], Sx[2])
        for k in range(3, 15, 2):
            p.v_max3_f32(t0, t0, Sx[k], Sx[k + 1])
        p.v_max_f32(t0, t0, Sx[15])
        p.v_mov_b32(t1, t0)
        p.s_nop(1)
        p.v_permlane32_swap_b32(t0, t1)
        p.s_nop(1)
        p.v_max_f32(mx, t0, t1)
        if self.margin:
            p.v_add_f32(mx, F(float(self.margin)), mx)
        p.v_mov_b32(v_m, mx)
        p.v_sub_f32(t0, F(0.0), mx)
        for r in range(16):
            p.v_mov_b32(v_negm[r], t0)
        p.s_nop(1)

    def item_start_part2(self, Sx, Px):
        p = self.p
        mx, acc = v_rs[2], v_rs[4]
        e = v_E
        items = []
        for half in range(2):
            for k in range(8):
                items.append(lambda k=k, half=half: p.v_sub_f32(e[k], Sx[8 * half + k], mx))
            for k in range(8):
                items.append(lambda k=k: p.v_exp_f32(e[k], e[k]))
            if half == 0:
                items.append(lambda: p.v_add_f32(acc, e[0], e[1]))
            else:
                items.append(lambda: p.v_add_f32(acc, acc, e[0]))
                items.append(lambda: p.v_add_f32(acc, acc, e[1]))
            for k in range(2, 8):
                items.append(lambda k=k: p.v_add_f32(acc, acc, e[k]))
            for w in range(4):
                items.append(lambda w=w, half=half: self.cvt_pk(Px[4 * half + w], e[2 * w], e[2 * w + 1]))
        items.append(lambda: p.v_mov_b32(v_l[0], acc))
        items.append(lambda: p.v_mov_b32(v_l[1], I(0)))
        return items

    def epilogue_head(self):
        """denominator (two partial sums, both lane halves), its reciprocal, the LSE store: before the new item's m / l overwrite the old"""
        p = self.p
        p.v_add_f32(v_l[0], v_l[0], v_l[1])
        p.v_mov_b32(v_lg, v_l[0])
        p.s_nop(1)
        p.v_permlane32_swap_b32(v_l[0], v_lg)
        p.s_nop(1)
        p.v_add_f32(v_l[0], v_l[0], v_lg)
        p.v_rcp_f32(v_inv, v_l[0])
        p.v_log_f32(v_lg, v_l[0])
        p.s_nop(0)
        p.v_add_f32(v_lg, v_lg, v_m)
        p.buffer_store_dword(v_lg, v_lseoff, s_desc_l, I(0))

    def epilogue_items(self):
        p = self.p
        it = []
        n = 0
        for dt in range(2):
            for gp in range(2):
                g = 2 * gp
                e, wq = v_ep, v_wq[n % 2]
                for k in range(8):
                    it.append(lambda k=k, dt=dt, g=g: p.v_accvgpr_read_b32(e[k], a_O[dt][4 * g + k]))
                for k in range(8):
                    it.append(lambda k=k: p.v_mul_f32(e[k], e[k], v_inv))
                for w in range(4):
                    it.append(lambda w=w, wq=wq: self.cvt_pk(wq[w], e[2 * w], e[2 * w + 1]))
                it.append(lambda: p.s_nop(1))
                it.append(lambda wq=wq: p.v_permlane32_swap_b32(wq[0], wq[2]))
                it.append(lambda wq=wq: p.v_permlane32_swap_b32(wq[1], wq[3]))
                it.append(lambda wq=wq, dt=dt, gp=gp: p.buffer_store_dwordx4(wq, v_ooff, s_desc_o, s_ocol, offset=64 * dt + 32 * gp))
                n += 1
        # a fresh accumulator for the new item (its first P V product runs in the steady body, with C = O)
        it.append(lambda: p.v_mov_b32(v_ep[0], I(0)))
        for dt in range(2):
            for r in range(16):
                it.append(lambda dt=dt, r=r: p.v_accvgpr_write_b32(a_O[dt][r], v_ep[0]))
        return it

    def q_prefetch(self):
        """Q rows of the NEXT item by LDS-DMA into this wave's area: four 8-row pieces; the descriptor steps by eight rows per piece"""
        p = self.p
        base = [s_t[8], s_t[9], s_t[10], s_t[11]]
        for k in range(4):
            p.s_mov_b32(base[k], s_desc_q[k])
        p.s_lshl_b32(s_t[12], s_ld2, 3)
        for i in range(4):
            p.s_add_u32(M0, s_qarea, I(1024 * i))
            if i:
                p.s_add_u32(base[0], base[0], s_t[12])
                p.s_addc_u32(base[1], base[1], I(0))
                p.s_sub_u32(base[2], base[2], s_t[12])
                p.s_cselect_b32(base[2], I(0), base[2])
            else:
                p.s_nop(0)
            p.buffer_load_lds_dwordx4(v_qdma, S(base[0].idx, 4), s_qcol)

    # ------------------------------------------------------------------ one step
    def step(self, kind, X):
        """X = 0 (a): unit i = U(t-1, kt 1): PV(U(t-1,0)), QK(U(t,0)), SM(S[1] -> P[1]);   X = 1 (b): unit i = U(t, 0): PV(U(t-1,1)), QK(U(t,1)), SM(S[0] -> P[0])"""
        p = self.p
        masked = kind == "last"
        first_b = kind == "first" and X == 1
        czero_qk = kind == "first" and X == 0
        mf, tags = [], []
        for ks in range(2):
            for dt in range(2):
                mf.append(lambda ks=ks, dt=dt: self.mfma(a_O[dt], v_VF[X][8 * ks + 4 * dt:8 * ks + 4 * dt + 4], v_P[X][4 * ks:4 * ks + 4], a_O[dt]))
                tags.append("pv")
        i_qk0 = len(mf)
        Sn = v_S[X]
        for sd in range(4):
            c = (I(0) if czero_qk else v_negm) if sd == 0 else Sn
            mf.append(lambda sd=sd, c=c: self.mfma(Sn, v_KF[X][4 * sd:4 * sd + 4], a_Q[4 * sd:4 * sd + 4], c))
            tags.append("qk")
        if masked:
            mf.append(lambda: self.mfma(Sn, a_mask[X], a_onesB, Sn))
            tags.append("qk")
        nm = len(mf)

        sm = [] if first_b else [x[0] for x in self.sm_items(v_S[1 - X], v_P[1 - X])]
        smc = [] if first_b else [x[1] for x in self.sm_items(v_S[1 - X], v_P[1 - X])]
        # LDS reads: K for the other buffer early, V into the buffer the P V products of this step have just consumed
        kreads, vreads = [], []
        if X == 0:
            for sd in range(4):
                kreads.append(lambda sd=sd: self.lds_read("b128", v_KF[1][4 * sd:4 * sd + 4], v_ka[sd], 4096, "KF1"))
        else:
            for sd in range(4):
                kreads.append(lambda sd=sd: self.lds_read("b128", v_KF[0][4 * sd:4 * sd + 4], v_ka[sd], 0, "KF0"))
        for ks in range(2):
            for dt in range(2):
                for hi in range(2):
                    vreads.append(lambda ks=ks, dt=dt, hi=hi: self.lds_read("tr", v_VF[X][8 * ks + 4 * dt + 2 * hi:8 * ks + 4 * dt + 2 * hi + 2], v_va[dt],
                                                                         (X * 32 + ks * 16) * 128 + hi * 1024, "VF%d" % X))
        which = "k" if X == 0 else "v"

        # head
        if X == 0:
            self.slot_heads()
            for dt in range(2):
                p.v_add_u32(v_va[dt], s_slot_r, v_voffb[dt])
        else:
            p.s_add_u32(s_slot_r, s_slot_r, I(SLOT))
            p.s_and_b32(s_slot_r, s_slot_r, I(RING * SLOT - 1))
            for sd in range(4):
                p.v_add_u32(v_ka[sd], s_slot_r, v_koffb[sd])
        if first_b:
            self.epilogue_head()

        # table: gap g = after MFMA g
        rows = [[] for _ in range(nm)]
        for k in range(4):
            rows[k].append(kreads[k])
        for k in range(8):
            rows[min(nm - 1, 4 + k // 2)].append(vreads[k])            # from gap 4: the products that read the old fragments are 8+ wait states back
        g_dma = min(nm - 2, self.dma_gap)
        dma_issue = [None]

        def do_m0():
            dma_issue[0] = self.dma_piece(which)
        rows[g_dma].append(do_m0)
        rows[g_dma + 1].append(lambda: dma_issue[0]())
        if first_b:
            part2 = self.item_start_part2(v_S[0], v_P[0])
            ep = self.epilogue_items()
            # part 2 behind the Q K^T products (they are issued right after part 1), the epilogue after it
            seq = part2 + ep
            per = (len(seq) + 3) // 4
            for k, f in enumerate(seq):
                rows[min(nm - 1, i_qk0 + k // per)].append(f)
        else:
            # SM: nothing before gap 1 (the scores' last product is the previous step's last MFMA), then level by cost
            total = sum(smc)
            level = total / (nm - 1)
            g, c = 1, 0.0
            for f, cst in zip(sm, smc):
                while g < nm - 1 and c + cst > level + 0.5:
                    g, c = g + 1, 0.0
                rows[g].append(f)
                c += cst
        if kind == "last" and X == 1:
            def qreload():
                for sd in range(4):
                    self.lds_read("b128", a_Q[4 * sd:4 * sd + 4], v_qrd, 32 * sd, "Q")
            rows[nm - 1].append(lambda: p.s_nop(7))
            rows[nm - 1].append(qreload)
        if kind == "first" and X == 1:
            rows[nm - 1].append(self.q_prefetch)

        for g in range(nm):
            if g == i_qk0:
                if X == 1:
                    self.wait_frag("KF1")
                if first_b:
                    self.item_start(v_S[0], v_P[0])
            mf[g]()
            before = p.count()
            for f in rows[g]:
                f()
            self.issue_rows.append((kind, "ab"[X], g, tags[g], p.count() - before))

    def iteration(self, kind):
        p = self.p
        p.comment(f"================ iteration: {kind}")
        st = self.stamps and kind == "steady"
        for X in range(2):
            p.comment(f"---- step {'ab'[X]} ({kind})")
            if st:
                p.s_memtime(s_stamp[X])
            self.step(kind, X)
        sw = self.dma_advance()
        if st:
            p.s_memtime(s_stamp[2])
        nvm = 2 + (5 + 4 if kind == "first" else 0)
        p.s_waitcnt(vmcnt=63 if "nodma" in self.abl else nvm, lgkmcnt=0)
        self.lds_all_done()
        if "nobar" not in self.abl:
            p.s_barrier()
        if st:
            p.s_memtime(s_stamp[3])
            p.s_waitcnt(lgkmcnt=0)
            for k in range(3):
                p.s_sub_u32(s_t[0], s_stamp[k + 1][0], s_stamp[k][0])
                p.s_add_u32(s_sacc[k], s_sacc[k], s_t[0])
            p.s_add_u32(s_sacc[3], s_sacc[3], I(1))
        return sw

    def dead_iteration(self):
        """a steady tile of a wave without live rows: its share of the K/V stream, the ring bookkeeping, the barrier"""
        p = self.p
        p.comment("================ iteration: steady, no live row")
        self.slot_heads()
        issue = self.dma_piece("k")
        p.s_add_u32(s_slot_r, s_slot_r, I(SLOT))
        issue()
        p.s_and_b32(s_slot_r, s_slot_r, I(RING * SLOT - 1))
        issue = self.dma_piece("v")
        p.s_nop(0)
        issue()
        sw = self.dma_advance()
        p.s_waitcnt(vmcnt=2, lgkmcnt=0)
        p.s_barrier()
        return sw

    def build(self):
        p = self.p
        lab_items = self.prologue()
        switches = []
        p.label(lab_items)
        for k in range(4):
            p.s_mov_b32(s_desc_o[k], s_desc_o_cur[k])
            p.s_mov_b32(s_desc_l[k], s_desc_l_cur[k])
        p.s_mov_b32(s_ocol, s_ocol_cur)
        self.make_out_desc(s_j)
        p.s_add_u32(s_t[10], s_j, s_stride)
        self.make_q_desc(s_t[10])
        switches.append(self.iteration("first"))
        lab_done = p.newlabel("done")
        p.s_cmp("lg", "u32", s_phantom, I(0))
        p.s_cbranch("scc1", lab_done)
        p.s_sub_u32(s_loop, s_nt, I(2))
        lab_dead, lab_last = p.newlabel("steady_dead"), p.newlabel("last")
        if self.skip_dead:
            p.s_cmp("lg", "u32", s_dead_cur, I(0))
            p.s_cbranch("scc1", lab_dead)
        lab_steady = p.label(p.newlabel("steady"))
        switches.append(self.iteration("steady"))
        p.s_sub_u32(s_loop, s_loop, I(1))
        p.s_cmp("lg", "u32", s_loop, I(0))
        p.s_cbranch("scc1", lab_steady.name)
        p.label(lab_last)
        switches.append(self.iteration("last"))
        p.s_add_u32(s_j, s_j, s_stride)
        p.s_cmp("ge", "u32", s_j, s_items)
        p.s_cselect_b32(s_phantom, I(1), I(0))
        p.s_branch(lab_items)
        if self.skip_dead:
            p.label(lab_dead)
            switches.append(self.dead_iteration())
            p.s_sub_u32(s_loop, s_loop, I(1))
            p.s_cmp("lg", "u32", s_loop, I(0))
            p.s_cbranch("scc1", lab_dead)
            # the last tile's body reads K(nt-1, kt 0) fragments and the address registers of its slot
            for sd in range(4):
                p.v_add_u32(v_ka[sd], s_slot_r, v_koffb[sd])
            for sd in range(4):
                p.ds_read_b128(v_KF[0][4 * sd:4 * sd + 4], v_ka[sd], 0)
            p.s_waitcnt(lgkmcnt=0)
            p.s_branch(lab_last)
        p.label(lab_done)
        p.s_waitcnt(vmcnt=0)
        if self.stamps:
            p.s_lshl_b32(s_t[0], s_wg, 3)
            p.s_add_u32(s_t[0], s_t[0], s_w)
            p.s_lshl_b32(s_t[0], s_t[0], 4)
            p.s_add_u32(s_dbg[0], s_dbg[0], s_t[0])
            p.s_addc_u32(s_dbg[1], s_dbg[1], I(0))
            p.v_mov_b32(v_x[0], s_dbg[0])
            p.v_mov_b32(v_x[1], s_dbg[1])
            for k in range(4):
                p.v_mov_b32(v_wq[0][k], s_sacc[k])
            p.s_nop(1)
            p.global_store_dwordx4(v_x[0:2], v_wq[0])
            p.s_waitcnt(vmcnt=0)
        p.s_endpgm()
        for sw, dn in switches:
            self.dma_switch_block(sw, dn)
        return p


def kernel_text(dtype="bf16", name=None, **kw):
    name = name or (KERNEL_NAME if dtype == "bf16" else KERNEL_NAME + "_f16")
    g = Gen32(dtype=dtype, **kw)
    prog = g.build()
    body = prog.text().replace(KERNEL_NAME + ":", name + ":")
    txt = HEADER.format(name=name) + body + FOOTER.format(name=name, lds=g.lds_bytes, kargs=KARG_BYTES, nvgpr=ARCH_VGPRS + ACC_VGPRS, accum=ARCH_VGPRS,
                                                        nagpr=ACC_VGPRS).replace(".max_flat_workgroup_size: 256", ".max_flat_workgroup_size: 512")
    return txt, g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--table")
    a = ap.parse_args()
    txt, g = kernel_text(a.dtype)
    with open(a.out, "w") as f:
        f.write("; GENERATED by tools/attn_asm/gen_attn32.py -- do not edit; edit the generator.\n" + txt)
    if a.table:
        with open(a.table, "w") as f:
            f.write("# iteration step gap mfma instructions_after\n")
            for row in g.issue_rows:
                f.write(" ".join(str(x) for x in row) + "\n")


if __name__ == "__main__":
    main()
