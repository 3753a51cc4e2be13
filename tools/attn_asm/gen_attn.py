#!/usr/bin/env python3
"""Generator of the hand-placed attention-forward kernel for gfx950 (ucod_attn_fwd_pw64): persistent workgroups, 4 waves x 64 query
rows, one wave per SIMD, every instruction of the tile loop assigned to an MFMA gap by the tables below.

Replaces transformers modeling_dinov2.py:153-179 (eager_attention_forward) / models/backbones/dino.py:96-120 of the reference for
head_dim 64, Q pre-scaled by head_dim^-0.5 * log2(e) (the QKV GEMM epilogue does that), same qkv / out layout as attn_fwd_v5_kernel.

Structure (DESIGN.md section 4 "Round 4: the hand-placed attention loop"):
  * work item = (image, head, 256-row query block); XCD x owns pairs x, x+8, ...; its 32 workgroups walk that list with stride 32, so the
    six blocks of a pair run at the same time on one L2;
  * a wave owns 64 query rows = two 32-row blocks (qb 0 / 1); S^T = K Q^T (keys on MFMA rows, queries on lanes), P converted in registers
    and fed back as the B operand of O^T += V^T P^T; O in AccVGPRs;
  * unit = (tile t, 32-key block kt, qb); unit index i = 4 t + 2 kt + qb.  Step i of the software pipeline issues, in this order,
        ONES(i-1)   2 MFMAs: column sums of P(i-1) (an all-ones A operand) into T      -- the softmax denominator on the matrix pipe
        QK(i+1)     4 MFMAs: S[(i+1)%4] = -m + K Q^T  (the running max rides in as the C operand)
        DETECT(i-1) T -> l += t; t >= 2^THR (or NaN) -> out-of-line rescale of unit i-1 (deferred max: m moves only then)
        PV(i-1)     4 MFMAs
     with SM(i) (16 v_exp_f32, 8 v_cvt_pk) and the LDS fragment reads / LDS-DMA pieces spread over the ten gaps;
  * an iteration = steps 4t-1 .. 4t+2, so that all Q K^T products of tile t sit in one iteration (the last tile's carry a fifth MFMA
    that adds -BIG to the keys past N);
  * K/V tiles by LDS-DMA three tiles ahead into 4-slot rings, one `s_waitcnt vmcnt(4)` + `s_barrier` per tile; the stream runs
    across work items; Q of the next item is fetched during the first tile of the current one; O is normalised and stored during
    the first tile of the next item.
"""
import argparse
import os
import sys

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from tools.attn_asm.isa import Prog, V, A, S, VCC, M0, F, I, Lit, f32_bits   # noqa: E402
else:
    from .isa import Prog, V, A, S, VCC, M0, F, I, Lit, f32_bits

KERNEL_NAME = "ucod_attn_fwd_pw64"
SLOT = 8192
KARG_BYTES = 72

# ------------------------------------------------------------------------------------------------ register map
s_karg = S(0, 2)
s_wg = S(2)
s_qkv = S(4, 2)
s_out = S(6, 2)
s_lse = S(8, 2)
s_N = S(10)
s_heads = S(11)
s_npairs = S(12)
s_nqb = S(13)
s_mg_nqb = S(14)
s_mg_heads = S(15)
s_nt = S(16)
s_stride = S(17)          # workgroups per XCD
s_a1_nqb = S(26)          # 1 when nqb == 1 (the magic multiplier cannot express a division by one)
s_a1_heads = S(27)
s_w = S(18)               # wave id in the workgroup
s_xcd = S(19)
s_D2 = S(20)              # bytes of one output row (heads * 128)
s_ld2 = S(21)             # bytes of one qkv row
s_TS = S(22)              # bytes of one 64-key tile of qkv rows
s_items = S(23)           # items of this XCD
s_j = S(24)               # current item
s_imgbytes = S(25)        # N * ld2
s_desc_kv = S(28, 4)
s_dma_next = S(32)        # tile index (in the DMA stream's item) of the next tile to request
s_jdma = S(33)
s_kcol = S(34)
s_vcol = S(35)
s_slot_r = S(36)
s_m0base = S(37)
s_thr = S(38)
s_loop = S(39)
s_desc_q = S(40, 4)
s_qcol = S(44)
s_phantom = S(45)
s_ocol = S(46)            # previous item's head column (epilogue)
s_ocol_cur = S(47)
s_desc_o = S(48, 4)       # previous item (epilogue)
s_desc_l = S(52, 4)
s_desc_o_cur = S(56, 4)
s_desc_l_cur = S(60, 4)
s_t = [S(64 + k) for k in range(16)]      # scratch
s_slot_w = S(80)
s_stamp = [S(82 + 2 * k, 2) for k in range(6)]     # diagnostic build (stamps=True) only
s_sacc = [S(94 + k) for k in range(6)]
s_dbg = S(100, 2)
s_par = S(81)

v_tid = V(0)
v_lane = V(1)
v_l31 = V(2)
v_h5 = V(3)
v_koffb = V(4, 4)
v_ka = V(8, 4)
v_voffb = V(12, 2)
v_va = V(14, 2)
v_dma_k = V(16, 2)
v_dma_v = V(18, 2)
v_qoff = V(20, 2)
v_ooff = V(22, 2)
v_lseoff = V(24)
v_lp = [V(30, 2), V(26, 2)]   # per q-block: the two partial row sums as one 64-bit aligned pair (v_pk_add_f32)
v_l = [v_lp[0][0], v_lp[1][0]]
v_tmp = [v_lp[0][1], v_lp[1][1]]
v_m = V(28, 2)
v_tt = V(25)              # DETECT's copy of T[0]
v_E = V(32, 8)
v_P = [V(40, 8), V(48, 8)]
v_negm = [V(56, 16), V(72, 16)]
v_S = [V(88 + 16 * k, 16) for k in range(4)]
v_KF = [V(152, 16), V(168, 16)]        # [buffer][sd] 4 registers each
v_VF = [V(184, 16), V(200, 16)]        # [buffer][ks][dt] 4 registers each (lo pair, hi pair)
v_ep = V(216, 16)          # epilogue: two sets of eight values
v_wq = [V(232, 4), V(236, 4)]   # epilogue: packed words of one store
v_inv = V(240)
v_lg = V(241)
v_re = V(242, 8)           # rescale: eight values
v_rs = V(250, 5)           # rescale: t0 t1 mx alpha acc
v_x = V(216, 8)            # prologue temporaries (alias of the epilogue's)
ARCH_VGPRS = 256

a_O = [[A(0, 16), A(16, 16)], [A(32, 16), A(48, 16)]]      # [qb][dt]
a_T = A(64, 16)
a_Q = [A(80, 16), A(96, 16)]                                # [qb][sd] 4 registers each
a_Qn = [A(112, 16), A(128, 16)]
a_ones = A(144, 4)
a_mask = [A(148, 4), A(152, 4)]
a_onesB = A(156, 4)
a_L = [a_T, A(160, 16)]         # without per-unit detection: the two running denominators (ONES accumulates into them)
ACC_VGPRS = 176


class Gen:
    def __init__(self, dtype="bf16", thr_exp=None, table=None, abl=(), unit_detect=None, margin=64, dma_gap=7, ring=4):
        self.p = Prog()
        # K / V rings of `ring` 8-KiB slots each; the stream runs ring - 1 (ring 4) or 4 (ring 8) tiles ahead.  ring 8: one barrier per TWO
        # tiles (a tile is visible one barrier after its wait, and its slot is rewritten two barriers after its last read)
        self.ring = ring
        self.lead = 3 if ring == 4 else 4
        self.v_ring = ring * SLOT
        self.lds_bytes = 2 * ring * SLOT
        # per-unit overflow detection + out-of-line rescale (the only form that is safe for fp16 probabilities); bf16 runs without:
        # the running max is fixed at item start with a margin (rescale_block)
        self.unit_detect = (dtype != "bf16") if unit_detect is None else unit_detect
        self.margin = margin
        self.dma_gap = dma_gap
        self.lds_policy = next((a[4:] for a in abl if a.startswith("lds_")), "early")
        # softmax denominator: "mfma" = an all-ones A operand (ONES products), "valu" = per-lane f32 adds of the unrounded probabilities
        self.sums = "mfma" if (self.unit_detect or "msum" in abl) else "valu"
        assert not (self.sums == "valu" and self.unit_detect)
        self.stamps = "stamps" in abl           # diagnostic: s_memtime at the step boundaries of the steady iteration, sums per wave to a buffer
        self.abl = set(abl)           # timing-only ablations (wrong results): see tools/attn_asm/bench_variants.py
        self.dtype = dtype
        self.thr_exp = thr_exp if thr_exp is not None else (40 if dtype == "bf16" else 13)
        self.lds_issued = 0           # LDS reads issued since the last lgkmcnt(0)
        self.frag_last = {}           # fragment buffer name -> index of its last issued read
        self.table = table or {}
        self.issue_rows = []          # (iteration kind, step, gap, [texts]) for the issue table
        self.site = 0

    # ------------------------------------------------------------------ small helpers
    def cvt_pk(self, d, a, b):
        if self.dtype == "bf16":
            self.p.v_cvt_pk_bf16_f32(d, a, b)
        else:
            self.p.v_cvt_pk_f16_f32(d, a, b)

    def mfma(self, d, a, b, c):
        if "nomfma" in self.abl:
            return None
        return self.p.mfma(d, a, b, c, dtype=self.dtype)

    def one16(self):
        return 0x3F80 if self.dtype == "bf16" else 0x3C00

    def negbig16(self):
        return 0xFF7F if self.dtype == "bf16" else 0xFBFF

    # LDS read tracking (counted lgkmcnt)
    def lds_read(self, kind, dst, addr, offset, buf):
        if "nolds" in self.abl:
            return
        if kind == "b128":
            self.p.ds_read_b128(dst, addr, offset)
        else:
            self.p.ds_read_b64_tr_b16(dst, addr, offset)
        self.frag_last[buf] = self.lds_issued
        self.lds_issued += 1

    def wait_frag(self, buf):
        """wait until every read of fragment buffer `buf` issued so far has returned"""
        if buf not in self.frag_last:
            return
        k = self.lds_issued - (self.frag_last[buf] + 1)
        self.p.s_waitcnt(lgkmcnt=k)
        # everything up to that read has retired
        for b in list(self.frag_last):
            if self.frag_last[b] <= self.frag_last[buf] and b != buf:
                del self.frag_last[b]
        del self.frag_last[buf]

    def lds_all_done(self):
        self.lds_issued = 0
        self.frag_last = {}

    # ------------------------------------------------------------------ scalar: 32x32 -> 64 multiply-add onto a pointer
    def ptr_add_mul(self, dst, base, a, b, t0, t1):
        """dst(64) = base(64) + a * b   (a, b 32-bit unsigned)"""
        p = self.p
        p.s_mul_i32(t0, a, b)
        p.s_mul_hi_u32(t1, a, b)
        p.s_add_u32(dst[0], base[0], t0)
        p.s_addc_u32(dst[1], base[1], t1)

    def item_decode(self, j, pair, qblk, b, head, valid_scc_label=None):
        """pair = xcd + 8 * (j / nqb), qblk = j % nqb, b = pair / heads, head = pair % heads   (magic-number divisions)"""
        p = self.p
        t = s_t[15]
        p.s_mul_hi_u32(t, j, s_mg_nqb)            # j / nqb
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
        """descriptor + column offsets of the K/V stream for item j (tile 0); an item past the list gets an empty descriptor"""
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

    def make_q_desc(self, j):
        """descriptor of the 64 query rows of this wave in item j"""
        p = self.p
        pair, qblk, b, head = s_t[0], s_t[1], s_t[2], s_t[3]
        self.item_decode(j, pair, qblk, b, head)
        q0, rows = s_t[6], s_t[7]
        p.s_lshl_b32(q0, qblk, 8)
        p.s_lshl_b32(s_t[4], s_w, 6)
        p.s_add_u32(q0, q0, s_t[4])
        p.s_sub_u32(rows, s_N, q0)                 # rows left (may be <= 0)
        p.s_max_i32(rows, rows, I(0))
        p.s_cmp("lt", "u32", j, s_items)
        p.s_cselect_b32(rows, rows, I(0))
        # base = qkv + b * imgbytes + q0 * ld2
        self.ptr_add_mul([s_t[8], s_t[9]], s_qkv, b, s_imgbytes, s_t[4], s_t[5])
        self.ptr_add_mul(s_desc_q, [s_t[8], s_t[9]], q0, s_ld2, s_t[4], s_t[5])
        p.s_and_b32(s_desc_q[1], s_desc_q[1], I(0xFFFF))
        p.s_mul_i32(s_desc_q[2], rows, s_ld2)
        p.s_lshl_b32(s_qcol, head, 7)

    def make_out_desc(self, j):
        """output / LSE descriptors of this wave's rows in item j -> the *_cur registers"""
        p = self.p
        pair, qblk, b, head = s_t[0], s_t[1], s_t[2], s_t[3]
        self.item_decode(j, pair, qblk, b, head)
        q0, rows = s_t[6], s_t[7]
        p.s_lshl_b32(q0, qblk, 8)
        p.s_lshl_b32(s_t[4], s_w, 6)
        p.s_add_u32(q0, q0, s_t[4])
        p.s_sub_u32(rows, s_N, q0)
        p.s_max_i32(rows, rows, I(0))
        p.s_cmp("lt", "u32", j, s_items)
        p.s_cselect_b32(rows, rows, I(0))
        # out rows: (b * N + q0) * D2
        p.s_mul_i32(s_t[8], b, s_N)
        p.s_add_u32(s_t[8], s_t[8], q0)
        self.ptr_add_mul(s_desc_o_cur, s_out, s_t[8], s_D2, s_t[4], s_t[5])
        p.s_and_b32(s_desc_o_cur[1], s_desc_o_cur[1], I(0xFFFF))
        p.s_mul_i32(s_desc_o_cur[2], rows, s_D2)
        p.s_lshl_b32(s_ocol_cur, head, 7)
        # lse rows: (pair * N + q0) * 4; no lse pointer -> empty descriptor
        p.s_mul_i32(s_t[8], pair, s_N)
        p.s_add_u32(s_t[8], s_t[8], q0)
        self.ptr_add_mul(s_desc_l_cur, s_lse, s_t[8], I(4), s_t[4], s_t[5])
        p.s_and_b32(s_desc_l_cur[1], s_desc_l_cur[1], I(0xFFFF))
        p.s_lshl_b32(s_desc_l_cur[2], rows, 2)
        p.s_or_b32(s_t[4], s_lse[0], s_lse[1])
        p.s_cmp("eq", "u32", s_t[4], I(0))
        p.s_cselect_b32(s_desc_l_cur[2], I(0), s_desc_l_cur[2])

    # ------------------------------------------------------------------ DMA stream
    def dma_advance(self):
        """after the four pieces of a tile: next tile of the stream (scalar only)"""
        p = self.p
        lab_sw, lab_done = p.newlabel("dma_switch"), p.newlabel("dma_adv_done")
        p.s_add_u32(s_dma_next, s_dma_next, I(1))
        p.s_cmp("eq", "u32", s_dma_next, s_nt)
        p.s_cbranch("scc1", lab_sw)
        p.s_add_u32(s_desc_kv[0], s_desc_kv[0], s_TS)
        p.s_addc_u32(s_desc_kv[1], s_desc_kv[1], I(0))
        p.s_sub_u32(s_desc_kv[2], s_desc_kv[2], s_TS)
        p.s_cselect_b32(s_desc_kv[2], I(0), s_desc_kv[2])  # an empty descriptor (item past the list) stays empty
        p.label(lab_done)
        return lab_sw, lab_done

    def dma_switch_block(self, lab_sw, lab_done):
        p = self.p
        p.label(lab_sw)
        p.s_add_u32(s_jdma, s_jdma, s_stride)
        self.make_kv_desc(s_jdma)
        p.s_branch(lab_done)

    def dma_piece(self, which, i):
        """one 1-KiB LDS-DMA piece of the stream's current tile: which = 'k' | 'v', i = 0 | 1 (rows 8w.. / 32+8w..)"""
        p = self.p
        off = (self.v_ring if which == "v" else 0) + i * 4096
        if "nodma" in self.abl:
            return lambda: None
        if "dma2reg" in self.abl:       # timing only: the same loads to registers (+ a 16-byte LDS store of older data)
            def issue():
                p.buffer_load_dwordx4(V(216 + 4 * ((2 * (which == "v") + i) % 4), 4), (v_dma_k if which == "k" else v_dma_v)[i], s_desc_kv, s_kcol if which == "k" else s_vcol)
                if "dswrite" in self.abl:
                    p.ds_write_b128(v_koffb[0], V(232, 4), 0)
            return issue
        p.s_add_u32(M0, s_m0base, I(off))
        return lambda: p.buffer_load_lds_dwordx4((v_dma_k if which == "k" else v_dma_v)[i], s_desc_kv, s_kcol if which == "k" else s_vcol)

    # ------------------------------------------------------------------ prologue
    def prologue(self):
        p = self.p
        p.label(KERNEL_NAME)
        p.s_load(S(4, 8), s_karg, 0)            # qkv out lse N heads
        p.s_load(S(12, 4), s_karg, 32)          # npairs nqb magic_nqb magic_heads
        p.s_load(S(16, 2), s_karg, 48)          # nt stride
        p.s_load(S(26, 2), s_karg, 56)          # add-one flags of the two divisions
        if self.stamps:
            p.s_load(s_dbg, s_karg, 64)
            for k in range(6):
                p.s_mov_b32(s_sacc[k], I(0))
        p.s_waitcnt(lgkmcnt=0)
        # ---- lane constants
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
        p.s_mov_b32(s_thr, F(2.0 ** self.thr_exp))
        # items of this XCD: pairs x, x+8, ... below npairs
        p.s_add_u32(s_t[0], s_npairs, I(7))
        p.s_sub_u32(s_t[0], s_t[0], s_xcd)
        p.s_lshr_b32(s_t[0], s_t[0], 3)           # (npairs + 7 - xcd) / 8   (npairs >= 0; xcd <= 7)
        p.s_mul_i32(s_items, s_t[0], s_nqb)
        p.s_cmp("ge", "u32", s_j, s_items)
        lab_go = p.newlabel("go")
        p.s_cbranch("scc0", lab_go)
        p.s_endpgm()
        p.label(lab_go)
        # koff[sd] = l31 * 128 + ((2 sd + h5) ^ ((l31 >> 1) & 7)) * 16
        p.v_bfe_u32(v_x[0], v_l31, I(1), I(3))
        for sd in range(4):
            p.v_or_b32(v_x[1], I(2 * sd), v_h5)
            p.v_xor_b32(v_x[1], v_x[1], v_x[0])
            p.v_lshlrev_b32(v_x[1], I(4), v_x[1])
            p.v_lshl_add_u32(v_koffb[sd], v_l31, I(7), v_x[1])
        # voff[dt]: i16 = lane & 15, g1 = (lane >> 4) & 1, key = 4 h5 + (i16 >> 2), dst = dt * 32 + g1 * 16 + 4 (i16 & 3)
        p.v_and_b32(v_x[0], I(15), v_lane)               # i16
        p.v_bfe_u32(v_x[1], v_lane, I(4), I(1))          # g1
        p.v_lshrrev_b32(v_x[2], I(2), v_x[0])
        p.v_lshl_add_u32(v_x[2], v_h5, I(2), v_x[2])     # key
        p.v_and_b32(v_x[3], I(3), v_x[0])
        p.v_lshlrev_b32(v_x[3], I(2), v_x[3])
        p.v_lshl_add_u32(v_x[3], v_x[1], I(4), v_x[3])   # dst (dt = 0)
        p.v_bfe_u32(v_x[4], v_x[2], I(1), I(1))
        p.v_lshlrev_b32(v_x[4], I(2), v_x[4])            # ((key >> 1) & 1) << 2
        for dt in range(2):
            p.v_add_u32(v_x[5], I(32 * dt), v_x[3])      # dst
            p.v_lshrrev_b32(v_x[6], I(3), v_x[5])
            p.v_xor_b32(v_x[6], v_x[6], v_x[4])          # swizzled chunk
            p.v_lshlrev_b32(v_x[6], I(4), v_x[6])
            p.v_and_b32(v_x[7], I(7), v_x[5])
            p.v_lshl_add_u32(v_x[6], v_x[7], I(1), v_x[6])
            p.v_lshl_add_u32(v_voffb[dt], v_x[2], I(7), v_x[6])
            p.v_add_u32(v_voffb[dt], I(self.v_ring), v_voffb[dt])
        # DMA source offsets: row = 8 w + (lane >> 3) (+ 32 i), chunk = lane & 7
        p.v_lshrrev_b32(v_x[0], I(3), v_lane)
        p.s_lshl_b32(s_t[0], s_w, 3)
        p.v_add_u32(v_x[0], s_t[0], v_x[0])              # row
        p.v_and_b32(v_x[1], I(7), v_lane)                # chunk
        p.v_bfe_u32(v_x[2], v_x[0], I(1), I(3))
        p.v_xor_b32(v_x[2], v_x[2], v_x[1])              # K: chunk ^ ((row >> 1) & 7)
        p.v_lshlrev_b32(v_x[2], I(4), v_x[2])
        p.v_bfe_u32(v_x[3], v_x[0], I(1), I(1))
        p.v_lshlrev_b32(v_x[3], I(2), v_x[3])
        p.v_xor_b32(v_x[3], v_x[3], v_x[1])              # V: chunk ^ (((row >> 1) & 1) << 2)
        p.v_lshlrev_b32(v_x[3], I(4), v_x[3])
        for i in range(2):
            p.v_add_u32(v_x[4], I(32 * i), v_x[0])
            p.v_mul_lo_u32(v_x[4], v_x[4], s_ld2)
            p.v_add_u32(v_dma_k[i], v_x[4], v_x[2])
            p.v_add_u32(v_dma_v[i], v_x[4], v_x[3])
        # Q / O / LSE offsets
        for qb in range(2):
            p.v_add_u32(v_x[0], I(32 * qb), v_l31)
            p.v_mul_lo_u32(v_x[1], v_x[0], s_ld2)
            p.v_lshl_add_u32(v_qoff[qb], v_h5, I(4), v_x[1])
            p.v_mul_lo_u32(v_x[1], v_x[0], s_D2)
            p.v_lshl_add_u32(v_ooff[qb], v_h5, I(4), v_x[1])
        p.v_lshlrev_b32(v_lseoff, I(2), v_l31)
        p.v_lshl_or_b32(v_lseoff, v_h5, I(31), v_lseoff)     # upper half: beyond every descriptor
        # constant operands: all-ones A fragment, one-hot B fragment, the last tile's key mask
        one = self.one16()
        p.v_mov_b32(v_x[0], I(one | (one << 16)))
        for k in range(4):
            p.v_accvgpr_write_b32(a_ones[k], v_x[0])
        p.v_cmp("eq", "u32", I(0), v_h5)
        p.v_mov_b32(v_x[1], I(0))
        p.v_mov_b32(v_x[2], I(one))
        p.v_cndmask_b32(v_x[3], v_x[1], v_x[2])              # h5 == 0 ? 1.0 (element 0) : 0
        p.v_accvgpr_write_b32(a_onesB[0], v_x[3])
        for k in range(1, 4):
            p.v_accvgpr_write_b32(a_onesB[k], v_x[1])
        # mask(kt): key 32 kt + l31 of the last tile is past N  <=>  l31 >= rem - 32 kt,  rem = N - 64 (nt - 1)
        p.s_sub_u32(s_t[0], s_nt, I(1))
        p.s_lshl_b32(s_t[0], s_t[0], 6)
        p.s_sub_u32(s_t[0], s_N, s_t[0])                      # rem in 1..64
        p.v_mov_b32(v_x[2], I(self.negbig16()))
        for kt in range(2):
            p.s_sub_u32(s_t[1], s_t[0], I(32 * kt))           # may be negative: signed compare
            p.v_cmp("le", "i32", s_t[1], v_l31)               # rem_kt <= l31
            p.v_cndmask_b32(v_x[3], v_x[1], v_x[2])           # -BIG where masked
            p.v_cmp("eq", "u32", I(0), v_h5)
            p.v_cndmask_b32(v_x[3], v_x[1], v_x[3])           # only the lower half carries element k = 0
            p.v_accvgpr_write_b32(a_mask[kt][0], v_x[3])
            for k in range(1, 4):
                p.v_accvgpr_write_b32(a_mask[kt][k], v_x[1])
        # ---- state of a phantom previous item: nothing is stored, nothing overflows
        for r in range(8):
            p.v_mov_b32(v_P[0][r], I(0))
            p.v_mov_b32(v_P[1][r], I(0))
        for r in range(16):
            p.v_mov_b32(v_S[2][r], I(0))
            p.v_mov_b32(v_S[3][r], I(0))
            p.v_mov_b32(v_VF[1][r], I(0))
        p.v_mov_b32(v_x[4], F(1.0))
        for qb in range(2):
            p.v_mov_b32(v_l[qb], F(1.0))
            p.v_mov_b32(v_tmp[qb], I(0))
            p.v_mov_b32(v_m[qb], I(0))
            for r in range(16):
                p.v_accvgpr_write_b32(a_L[qb][r], v_x[4])
            for dt in range(2):
                for r in range(16):
                    p.v_accvgpr_write_b32(a_O[qb][dt][r], v_x[1])
        for k in range(4):
            p.s_mov_b32(s_desc_o_cur[k], I(0))
            p.s_mov_b32(s_desc_l_cur[k], I(0))
        p.s_mov_b32(s_desc_o_cur[3], I(0x00020000))
        p.s_mov_b32(s_desc_l_cur[3], I(0x00020000))
        p.s_mov_b32(s_desc_kv[3], I(0x00020000))
        p.s_mov_b32(s_desc_q[3], I(0x00020000))
        p.s_mov_b32(s_ocol_cur, I(0))
        p.s_mov_b32(s_phantom, I(0))
        # ---- first item: Q straight into the live fragments, tiles 0..2 requested
        self.make_q_desc(s_j)
        for qb in range(2):
            for sd in range(4):
                p.buffer_load_dwordx4(a_Q[qb][4 * sd:4 * sd + 4], v_qoff[qb], s_desc_q, s_qcol, offset=32 * sd)
        p.s_mov_b32(s_jdma, s_j)
        self.make_kv_desc(s_jdma)
        p.s_mov_b32(s_slot_r, I(0))
        switches = []
        for t in range(self.lead):
            p.s_lshl_b32(s_m0base, s_w, 10)
            p.s_add_u32(s_m0base, s_m0base, I(t * SLOT))
            for which in ("k", "v"):
                for i in range(2):
                    issue = self.dma_piece(which, i)
                    p.s_nop(0)
                    issue()
            switches.append(self.dma_advance())
        p.s_waitcnt(vmcnt=4 if self.ring == 4 else 0)   # Q and tiles 0, 1 have landed (ring 4: tile 2 may still be in flight)
        p.s_barrier()
        p.s_mov_b32(s_par, I(0))
        for sd in range(4):
            p.ds_read_b128(v_KF[0][4 * sd:4 * sd + 4], v_koffb[sd], 0)          # K(0, kt 0): slot 0
        for sd in range(4):
            p.v_mov_b32(v_ka[sd], v_koffb[sd])
        p.s_waitcnt(lgkmcnt=0)
        self.lds_all_done()
        lab_items = p.newlabel("item_top")
        p.s_branch(lab_items)
        for sw, dn in switches:
            self.dma_switch_block(sw, dn)
        return lab_items

    # ------------------------------------------------------------------ pieces of a step
    def sm_items(self, Sx, Px, qb=None):
        """SM(i): 16 exponentials into rolling temporaries, 8 packs (+ 16 adds into two per-lane partial sums when the denominator is
        kept on the vector unit); returns a list of (emit, cost, kind)"""
        p = self.p
        order = []
        if "nosm" in self.abl:
            return order
        if self.sums == "valu":
            acc = [v_l[qb], v_tmp[qb]]

            def ex(k):
                return (lambda: p.v_exp_f32(v_E[k % 8], Sx[k]), 2, "exp")

            def ad(k):
                return (lambda: p.v_add_f32(acc[k & 1], acc[k & 1], v_E[k % 8]), 1, "add")

            def pad(w):      # both partial sums at once: (l, tmp) += (E[2w], E[2w+1]) -- the same additions as ad(2w), ad(2w+1)
                return (lambda: p.v_pk_add_f32(v_lp[qb], v_lp[qb], v_E[(2 * w) % 8:(2 * w) % 8 + 2]), 1, "add")

            def cv(w):
                return (lambda: self.cvt_pk(Px[w], v_E[(2 * w) % 8], v_E[(2 * w + 1) % 8]), 1, "cvt")
            if "pksum" not in self.abl:      # default: plain adds.  v_pk_add_f32 does not overlap with an MFMA in flight (~10 matrix-pipe cycles each): pw64 190 -> 230 us with it
                order += [ex(0), ex(1), ex(2), ad(0), ex(3), ad(1), cv(0)]
                for w in range(1, 7):
                    order += [ex(2 * w + 2), ad(2 * w), ex(2 * w + 3), ad(2 * w + 1), cv(w)]
                order += [ad(14), ad(15), cv(7)]
                return order
            order += [ex(0), ex(1), ex(2), ex(3), pad(0), cv(0)]
            for w in range(1, 7):
                order += [ex(2 * w + 2), ex(2 * w + 3), pad(w), cv(w)]
            order += [pad(7), cv(7)]
            return order

        def ex(k):
            if "exp2mov" in self.abl:
                return (lambda: p.v_mov_b32(v_E[k % 8], Sx[k]), 1, "exp")
            return (lambda: p.v_exp_f32(v_E[k % 8], Sx[k]), 2, "exp")

        def cv(w):
            return (lambda: self.cvt_pk(Px[w], v_E[(2 * w) % 8], v_E[(2 * w + 1) % 8]), 1, "cvt")
        order += [ex(0), ex(1), ex(2), ex(3), cv(0)]
        for w in range(1, 7):
            order += [ex(2 * w + 2), ex(2 * w + 3), cv(w)]
        order += [cv(7)]
        return order

    def rescale_block(self, qb, Sx, Px, Snext, first, back):
        """out-of-line (or, first=True, in-line) rescale of one unit: new running max from Sx; O, l, -m and the next unit's scores follow.
        Touches only its own temporaries (v_re, v_rs): it runs in the middle of a step whose SM / epilogue fillers are in flight.
        Without per-unit detection (bf16) only the first form exists: it sets m = max(first 32 keys) + MARGIN, so that every later
        probability 2^(s - m) stays finite up to MARGIN + 127 above that maximum and exact (a power-of-two scale) below it."""
        p = self.p
        t0, t1, mx, al, acc = v_rs[0], v_rs[1], v_rs[2], v_rs[3], v_rs[4]
        if first == "part2":
            return self.rescale_part2(qb, Sx, Px, True, None)
        if not first:
            p.s_nop(15)                 # the Q K^T products of the next unit were issued just before
            p.s_nop(15)
        p.v_max3_f32(t0, Sx[0], Sx[1], Sx[2])
        for k in range(3, 15, 2):
            p.v_max3_f32(t0, t0, Sx[k], Sx[k + 1])
        p.v_max_f32(t0, t0, Sx[15])
        p.v_mov_b32(t1, t0)
        p.s_nop(1)
        p.v_permlane32_swap_b32(t0, t1)
        p.s_nop(1)
        p.v_max_f32(mx, t0, t1)
        if first and not self.unit_detect and self.margin:
            p.v_add_f32(mx, F(float(self.margin)), mx)
        if not first:
            p.v_max_f32(mx, mx, F(0.0))
            p.v_sub_f32(t0, F(0.0), mx)
            p.v_exp_f32(al, t0)
            p.v_add_f32(v_m[qb], v_m[qb], mx)
            p.s_nop(0)
            p.v_mul_f32(v_l[qb], v_l[qb], al)
            for dt in range(2):
                for r in range(0, 16, 8):
                    for k in range(8):
                        p.v_accvgpr_read_b32(v_re[k], a_O[qb][dt][r + k])
                    for k in range(8):
                        p.v_mul_f32(v_re[k], v_re[k], al)
                    for k in range(8):
                        p.v_accvgpr_write_b32(a_O[qb][dt][r + k], v_re[k])
            for r in range(16):
                p.v_sub_f32(v_negm[qb][r], v_negm[qb][r], mx)
        else:
            p.v_mov_b32(v_m[qb], mx)
            p.v_sub_f32(t0, F(0.0), mx)
            for r in range(16):
                p.v_mov_b32(v_negm[qb][r], t0)
            p.s_nop(1)
            return                      # item start, part 1: the next unit's Q K^T is issued now, with -m as its C operand
        for r in range(16):
            p.v_sub_f32(Snext[r], Snext[r], mx)
        self.rescale_part2(qb, Sx, Px, False, back)

    def rescale_part2(self, qb, Sx, Px, first, back):
        """the unit again, against the new maximum (two halves of eight scores)"""
        p = self.p
        t0, t1, mx, al, acc = v_rs[0], v_rs[1], v_rs[2], v_rs[3], v_rs[4]
        sums = self.unit_detect or self.sums == "valu"
        for half in range(2):
            for k in range(8):
                p.v_sub_f32(v_re[k], Sx[8 * half + k], mx)
            for k in range(8):
                p.v_exp_f32(v_re[k], v_re[k])
            if sums:
                if half == 0:
                    p.v_add_f32(acc, v_re[0], v_re[1])
                else:
                    p.v_add_f32(acc, acc, v_re[0])
                    p.v_add_f32(acc, acc, v_re[1])
                for k in range(2, 8):
                    p.v_add_f32(acc, acc, v_re[k])
            else:
                p.s_nop(0)
            for w in range(4):
                self.cvt_pk(Px[4 * half + w], v_re[2 * w], v_re[2 * w + 1])
        if sums and self.sums == "valu":
            p.v_mov_b32(v_l[qb], acc)          # per-lane partial sums; the halves meet in the epilogue
            p.v_mov_b32(v_tmp[qb], I(0))
        elif sums:
            p.v_mov_b32(t1, acc)
            p.s_nop(1)
            p.v_permlane32_swap_b32(acc, t1)
            p.s_nop(1)
            p.v_add_f32(acc, acc, t1)
            if first:
                p.v_mov_b32(v_l[qb], acc)
            else:
                p.v_add_f32(v_l[qb], v_l[qb], acc)
        p.s_nop(1)
        if not first:
            p.s_branch(back)

    def epilogue_items(self, qb):
        """normalise + store O of the PREVIOUS item's 32-row block qb (+ LSE); list of (emit, cost, kind)"""
        p = self.p
        inv, lg = v_inv, v_lg
        it = []
        lsrc = v_l[qb]
        if self.sums == "valu":
            it.append((lambda: p.v_add_f32(v_l[qb], v_l[qb], v_tmp[qb]), 1, "ep"))
            it.append((lambda: p.v_mov_b32(v_lg, v_l[qb]), 1, "ep"))
            it.append((lambda: p.s_nop(1), 1, "ep"))
            it.append((lambda: p.v_permlane32_swap_b32(v_l[qb], v_lg), 1, "ep"))
            it.append((lambda: p.s_nop(0), 1, "ep"))
            it.append((lambda: p.v_add_f32(v_l[qb], v_l[qb], v_lg), 1, "ep"))
        elif not self.unit_detect:
            it.append((lambda: p.v_accvgpr_read_b32(v_l[qb], a_L[qb][0]), 1, "ep"))
            it.append((lambda: p.s_nop(0), 1, "ep"))
        it.append((lambda: p.v_rcp_f32(inv, lsrc), 2, "ep"))
        it.append((lambda: p.v_log_f32(lg, lsrc), 2, "ep"))
        it.append((lambda: p.s_nop(0), 1, "ep"))
        it.append((lambda: p.v_add_f32(lg, lg, v_m[qb]), 1, "ep"))
        it.append((lambda: p.buffer_store_dword(lg, v_lseoff, s_desc_l, I(0), offset=128 * qb), 1, "ep"))
        it.append((lambda: p.s_nop(1), 1, "ep"))          # the block's last P V product is 12 wait states back only after this
        n = 0
        for dt in range(2):
            for gp in range(2):
                g = 2 * gp
                e = v_ep[0:8] if n % 2 == 0 else v_ep[8:16]
                wq = v_wq[n % 2]
                for k in range(8):
                    it.append((lambda k=k, e=e, dt=dt, g=g: p.v_accvgpr_read_b32(e[k], a_O[qb][dt][4 * g + k]), 1, "ep"))
                for k in range(8):
                    it.append((lambda k=k, e=e: p.v_mul_f32(e[k], e[k], inv), 1, "ep"))
                it.append((lambda e=e, wq=wq: self.cvt_pk(wq[0], e[0], e[1]), 1, "ep"))
                it.append((lambda e=e, wq=wq: self.cvt_pk(wq[1], e[2], e[3]), 1, "ep"))
                it.append((lambda e=e, wq=wq: self.cvt_pk(wq[2], e[4], e[5]), 1, "ep"))
                it.append((lambda e=e, wq=wq: self.cvt_pk(wq[3], e[6], e[7]), 1, "ep"))
                it.append((lambda: p.s_nop(1), 1, "ep"))
                it.append((lambda wq=wq: p.v_permlane32_swap_b32(wq[0], wq[2]), 1, "ep"))
                it.append((lambda wq=wq: p.v_permlane32_swap_b32(wq[1], wq[3]), 1, "ep"))
                it.append((lambda: p.s_nop(0), 1, "ep"))
                it.append((lambda wq=wq, dt=dt, gp=gp: p.buffer_store_dwordx4(wq, v_ooff[qb], s_desc_o, s_ocol, offset=64 * dt + 32 * gp), 1, "ep"))
                n += 1
        return it

    # ------------------------------------------------------------------ one step
    def step(self, kind, X, slow_sites):
        """kind: 'first' | 'steady' | 'last';  X: 0..3 = a..d.
        a: unit i = U(t-1, 3)   b: U(t, 0)   c: U(t, 1)   d: U(t, 2)        (qb(i) = [1, 0, 1, 0][X])
        MFMA order: per-unit detection (f16):  ONES(i-1) x2, QK(i+1) x4, [DETECT(i-1)], PV(i-1) x4
                    none (bf16):               QK(i+1) x4, [item start: rescale], ONES(i-1) x2 (accumulating), PV(i-1) x4
        """
        p = self.p
        ud = self.unit_detect
        jS = [3, 0, 1, 2][X]                  # S buffer / position of unit i
        qb = jS & 1                           # qb of unit i
        qo = 1 - qb                           # qb of units i - 1 and i + 1
        jPrev = (jS + 3) % 4                  # unit i - 1
        jNext = (jS + 1) % 4                  # unit i + 1
        kt_next = jNext >> 1
        kbuf = kt_next                        # K fragments: buffer 0 = kt 0 (steps a, b), buffer 1 = kt 1 (steps c, d)
        vbuf = jPrev >> 1                     # V fragments of unit i - 1: kt(i - 1)
        first_new = kind == "first" and X in (2, 3)          # the mid position runs the new item's first unit (steps c, d)
        no_sm = kind == "first" and X in (1, 2)               # SM of the new item's first units is done by the in-line rescale
        no_ones = (ud and kind == "first" and X in (2, 3)) or "noones" in self.abl or self.sums == "valu"
        czero_qk = kind == "first" and X in (0, 1)            # the new item's first two units: raw scores (their maximum becomes m)
        czero_pv = kind == "first" and X in (2, 3)            # first products into a fresh O
        masked = kind == "last"

        mf, tags = [], []

        def add_ones():
            if no_ones:
                return
            if ud:
                mf.append(lambda: self.mfma(a_T, a_ones, v_P[qo][0:4], I(0)))
                mf.append(lambda: self.mfma(a_T, a_ones, v_P[qo][4:8], a_T))
            else:
                c0 = I(0) if first_new else a_L[qo]
                mf.append(lambda: self.mfma(a_L[qo], a_ones, v_P[qo][0:4], c0))
                mf.append(lambda: self.mfma(a_L[qo], a_ones, v_P[qo][4:8], a_L[qo]))
            tags.extend(["ones", "ones"])

        if ud:
            add_ones()
        i_qk0 = len(mf)
        Sn = v_S[jNext]
        for sd in range(4):
            c = (I(0) if czero_qk else v_negm[qo]) if sd == 0 else Sn
            mf.append(lambda sd=sd, c=c: self.mfma(Sn, v_KF[kbuf][4 * sd:4 * sd + 4], a_Q[qo][4 * sd:4 * sd + 4], c))
            tags.append("qk")
        if masked:
            mf.append(lambda: self.mfma(Sn, a_mask[kt_next], a_onesB, Sn))
            tags.append("qk")
        i_mid = len(mf)               # DETECT / the item's first rescale sit after the last QK MFMA
        if not ud:
            add_ones()
        i_pv0 = len(mf)
        for ks in range(2):
            for dt in range(2):
                c = I(0) if (czero_pv and ks == 0) else a_O[qo][dt]
                mf.append(lambda ks=ks, dt=dt, c=c: self.mfma(a_O[qo][dt], v_VF[vbuf][8 * ks + 4 * dt:8 * ks + 4 * dt + 4], v_P[qo][4 * ks:4 * ks + 4], c))
                tags.append("pv")
        nm = len(mf)

        # ---- fillers: queues
        sm = [] if no_sm else self.sm_items(v_S[jS], v_P[qb], qb)
        ep = []
        if kind == "first" and X in (1, 2):
            ep = self.epilogue_items(0 if X == 1 else 1)
        lds = []
        if X in (0, 1):       # K(t, kt 1) -> KF[1] (2 per step), V(t, kt 0) -> VF[0] (4 per step)
            for sd in (2 * X, 2 * X + 1):
                lds.append(lambda sd=sd: self.lds_read("b128", v_KF[1][4 * sd:4 * sd + 4], v_ka[sd], 4096, "KF1"))
            ks = X
            for dt in range(2):
                for hi in range(2):
                    lds.append(lambda ks=ks, dt=dt, hi=hi: self.lds_read("tr", v_VF[0][8 * ks + 4 * dt + 2 * hi:8 * ks + 4 * dt + 2 * hi + 2], v_va[dt],
                                                                         (0 * 32 + ks * 16) * 128 + hi * 1024, "VF0"))
        else:                 # K(t+1, kt 0) -> KF[0] (2 per step), V(t, kt 1) -> VF[1] (4 per step)
            for sd in (2 * (X - 2), 2 * (X - 2) + 1):
                lds.append(lambda sd=sd: self.lds_read("b128", v_KF[0][4 * sd:4 * sd + 4], v_ka[sd], 0, "KF0"))
            ks = X - 2
            for dt in range(2):
                for hi in range(2):
                    lds.append(lambda ks=ks, dt=dt, hi=hi: self.lds_read("tr", v_VF[1][8 * ks + 4 * dt + 2 * hi:8 * ks + 4 * dt + 2 * hi + 2], v_va[dt],
                                                                         (1 * 32 + ks * 16) * 128 + hi * 1024, "VF1"))
        which, pi = [("k", 0), ("k", 1), ("v", 0), ("v", 1)][X]

        # ---- forced items before given MFMAs
        pre = {k: [] for k in range(nm + 1)}
        if X == 2:
            pre[i_qk0].append(lambda: self.wait_frag("KF1"))
            pre[i_pv0].append(lambda: self.wait_frag("VF0"))
        # (steps a, b use KF[0] / VF[1]: complete since the iteration boundary; step d uses KF[1] / VF[0]: waited in step c)

        # ---- head of the step (before the first MFMA)
        if X == 0:
            # ring: this iteration reads tile t from slot_r; the stream writes tile t + 3
            p.s_add_u32(s_slot_w, s_slot_r, I(self.lead * SLOT))
            p.s_and_b32(s_slot_w, s_slot_w, I(self.ring * SLOT - 1))
            p.s_lshl_b32(s_m0base, s_w, 10)
            p.s_add_u32(s_m0base, s_m0base, s_slot_w)
            for dt in range(2):
                p.v_add_u32(v_va[dt], s_slot_r, v_voffb[dt])
        if X == 2:
            # K reads of steps c, d come from tile t + 1
            p.s_add_u32(s_slot_r, s_slot_r, I(SLOT))
            p.s_and_b32(s_slot_r, s_slot_r, I(self.ring * SLOT - 1))
            for sd in range(4):
                p.v_add_u32(v_ka[sd], s_slot_r, v_koffb[sd])

        # ---- interleave
        tab = self.table.get((kind, X)) or self.default_table(kind, X, nm, sm, len(lds), ep, tags)
        dma_issue = None
        qs, ql, qe = list(sm), list(lds), list(ep)
        mid_done = False
        for g in range(nm):
            for f in pre[g]:
                f()
            if g == i_qk0 and first_new:
                self.rescale_block(qo, v_S[jPrev], v_P[qo], v_S[jNext], True, None)      # part 1: m and -m of the new item's block
            if g == i_mid and not mid_done:
                self.detect(kind, X, qo, jPrev, jNext, first_new, slow_sites)
                mid_done = True
            mf[g]()
            row = tab[g]
            before = p.count()
            for what in row:
                if what == "sm" and qs:
                    qs.pop(0)[0]()
                elif what == "lds" and ql:
                    ql.pop(0)()
                elif what == "ep" and qe:
                    qe.pop(0)[0]()
                elif what == "m0":
                    dma_issue = self.dma_piece(which, pi)
                elif what == "dma":
                    dma_issue()
                elif what == "qcopy0":
                    for k in range(16):
                        p.v_accvgpr_mov_b32(a_Q[0][k], a_Qn[0][k])
                elif what == "qcopy1":
                    for k in range(16):
                        p.v_accvgpr_mov_b32(a_Q[1][k], a_Qn[1][k])
                elif what == "qload":
                    for qb_ in range(2):
                        for sd in range(4):
                            p.buffer_load_dwordx4(a_Qn[qb_][4 * sd:4 * sd + 4], v_qoff[qb_], s_desc_q, s_qcol, offset=32 * sd)
            self.issue_rows.append((kind, "abcd"[X], g, tags[g], " ".join(row)))
        assert not qs and not ql and not qe, (kind, X, len(qs), len(ql), len(qe))

    COST = {"lds": 1, "m0": 1, "dma": 4}

    def default_table(self, kind, X, nm, sm, n_lds, ep, tags):
        """per gap: the fillers that follow MFMA g, balanced by issue cost (exp 2 slots, everything else 1, a DMA piece 4).
        One LDS read per gap from the first gap on (their consumers are a step or more away), the DMA piece in the middle,
        the exponentials / packs (or the epilogue's instructions) fill every gap up to the common level."""
        rows = [[] for _ in range(nm)]
        cost = [0.0] * nm
        fixed = {}
        for k in range(n_lds):
            g_l = {"early": k, "front": k // 3, "pairs": 2 * (k // 2) // 2 + (k // 2), "late": nm - 1 - n_lds + k}[self.lds_policy]
            fixed.setdefault(max(0, min(nm - 1, g_l)), []).append("lds")
        g_dma = min(nm - 2, self.dma_gap)
        fixed.setdefault(g_dma, []).append("m0")
        fixed.setdefault(g_dma + 1, []).append("dma")
        if kind == "last" and X == 3:
            fixed.setdefault(1, []).append("qcopy0")
            fixed.setdefault(nm - 1, []).append("qcopy1")
        if kind == "first" and X == 3:
            fixed.setdefault(nm - 1, []).append("qload")
        for g, lst in fixed.items():
            for w in lst:
                rows[g].append(w)
                cost[g] += self.COST.get(w, 0)
        queue = [("sm", c) for (_, c, _) in sm] + [("ep", c) for (_, c, _) in ep]
        total = sum(cost) + sum(c for _, c in queue)
        level = total / nm
        g = 0
        for name, c in queue:
            while g < nm - 1 and cost[g] + c > level + 0.5:
                g += 1
            rows[g].append(name)
            cost[g] += c
        return rows

    def detect(self, kind, X, qo, jPrev, jNext, first_new, slow_sites):
        p = self.p
        if first_new:
            self.rescale_block(qo, v_S[jPrev], v_P[qo], v_S[jNext], "part2", None)
            return
        if not self.unit_detect or "nodetect" in self.abl:
            return
        lab_slow, lab_back = p.newlabel("slow"), p.newlabel("back")
        p.v_accvgpr_read_b32(v_tt, a_T[0])
        p.s_nop(0)
        p.v_cmp("ngt", "f32", s_thr, v_tt)          # not (thr > t): the unit needs a new maximum (or t is NaN)
        if "nobranch" not in self.abl:
            p.s_cbranch("vccnz", lab_slow)
        p.v_add_f32(v_l[qo], v_l[qo], v_tt)
        p.label(lab_back)
        slow_sites.append((lab_slow, lab_back, qo, jPrev, jNext))

    # ------------------------------------------------------------------ iteration
    def iteration(self, kind, slow_sites):
        p = self.p
        p.comment(f"================ iteration: {kind}")
        sw = None
        st = self.stamps and kind == "steady"
        for X in range(4):
            p.comment(f"---- step {'abcd'[X]} ({kind})")
            if st:
                p.s_memtime(s_stamp[X])
            self.step(kind, X, slow_sites)
        sw = self.dma_advance()
        if st:
            p.s_memtime(s_stamp[4])
        nvm = 4 + (10 + 8 if kind == "first" else 0)
        if "nodma" in self.abl:
            nvm = 63
        p.s_waitcnt(vmcnt=nvm, lgkmcnt=0)
        self.lds_all_done()
        if "nobar" not in self.abl:
            if self.ring == 4:
                p.s_barrier()
            else:
                lab = p.newlabel("nobarrier")
                p.s_add_u32(s_par, s_par, I(1))
                p.s_and_b32(s_t[0], s_par, I(1))
                p.s_cbranch("scc1", lab)          # odd count: no barrier this tile
                p.s_barrier()
                p.label(lab)
        if st:
            p.s_memtime(s_stamp[5])
            p.s_waitcnt(lgkmcnt=0)
            for k in range(5):
                p.s_sub_u32(s_t[0], s_stamp[k + 1][0], s_stamp[k][0])
                p.s_add_u32(s_sacc[k], s_sacc[k], s_t[0])
            p.s_add_u32(s_sacc[5], s_sacc[5], I(1))
        return sw

    def build(self):
        p = self.p
        lab_items = self.prologue()
        slow_sites, switches = [], []
        p.label(lab_items)
        # previous item's descriptors for the epilogue; this item's for the next one; Q of the next item
        for k in range(4):
            p.s_mov_b32(s_desc_o[k], s_desc_o_cur[k])
            p.s_mov_b32(s_desc_l[k], s_desc_l_cur[k])
        p.s_mov_b32(s_ocol, s_ocol_cur)
        self.make_out_desc(s_j)
        p.s_add_u32(s_t[10], s_j, s_stride)
        self.make_q_desc(s_t[10])
        switches.append(self.iteration("first", slow_sites))
        lab_done = p.newlabel("done")
        p.s_cmp("lg", "u32", s_phantom, I(0))
        p.s_cbranch("scc1", lab_done)
        p.s_sub_u32(s_loop, s_nt, I(2))
        lab_steady = p.label(p.newlabel("steady"))
        switches.append(self.iteration("steady", slow_sites))
        p.s_sub_u32(s_loop, s_loop, I(1))
        p.s_cmp("lg", "u32", s_loop, I(0))
        p.s_cbranch("scc1", lab_steady.name)
        switches.append(self.iteration("last", slow_sites))
        p.s_add_u32(s_j, s_j, s_stride)
        p.s_cmp("ge", "u32", s_j, s_items)
        p.s_cselect_b32(s_phantom, I(1), I(0))
        p.s_branch(lab_items)
        p.label(lab_done)
        p.s_waitcnt(vmcnt=0)
        if self.stamps:
            # record of this wave: 6 sums (steps a..d, wait, barrier .. iterations) + 2 spare, at dbg + (4 wg + w) * 32
            p.s_lshl_b32(s_t[0], s_wg, 2)
            p.s_add_u32(s_t[0], s_t[0], s_w)
            p.s_lshl_b32(s_t[0], s_t[0], 5)
            p.s_add_u32(s_dbg[0], s_dbg[0], s_t[0])
            p.s_addc_u32(s_dbg[1], s_dbg[1], I(0))
            p.v_mov_b32(v_x[0], s_dbg[0])
            p.v_mov_b32(v_x[1], s_dbg[1])
            for k in range(6):
                p.v_mov_b32(v_ep[8 + k], s_sacc[k])
            p.v_mov_b32(v_ep[14], I(0))
            p.v_mov_b32(v_ep[15], I(0))
            p.global_store_dwordx4(v_x[0:2], v_ep[8:12])
            p.v_add_u32(v_x[0], I(16), v_x[0])
            p.s_nop(1)
            p.global_store_dwordx4(v_x[0:2], v_ep[12:16])
            p.s_waitcnt(vmcnt=0)
        p.s_endpgm()
        # out-of-line blocks
        for sw, dn in switches:
            self.dma_switch_block(sw, dn)
        for lab_slow, lab_back, qo, jPrev, jNext in slow_sites:
            p.label(lab_slow)
            self.rescale_block(qo, v_S[jPrev], v_P[qo], v_S[jNext], False, lab_back)
        return p


HEADER = """\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.amdhsa_code_object_version 6
\t.text
\t.protected\t{name}
\t.globl\t{name}
\t.p2align\t8
\t.type\t{name},@function
"""

FOOTER = """.Lfunc_end_{name}:
\t.size\t{name}, .Lfunc_end_{name}-{name}
\t.section\t.rodata,"a",@progbits
\t.p2align\t6, 0x0
\t.amdhsa_kernel {name}
\t\t.amdhsa_group_segment_fixed_size {lds}
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size {kargs}
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_dispatch_ptr 0
\t\t.amdhsa_user_sgpr_queue_ptr 0
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_user_sgpr_dispatch_id 0
\t\t.amdhsa_user_sgpr_kernarg_preload_length 0
\t\t.amdhsa_user_sgpr_kernarg_preload_offset 0
\t\t.amdhsa_user_sgpr_private_segment_size 0
\t\t.amdhsa_uses_dynamic_stack 0
\t\t.amdhsa_enable_private_segment 0
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_sgpr_workgroup_id_y 0
\t\t.amdhsa_system_sgpr_workgroup_id_z 0
\t\t.amdhsa_system_sgpr_workgroup_info 0
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr {nvgpr}
\t\t.amdhsa_next_free_sgpr 102
\t\t.amdhsa_accum_offset {accum}
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_round_mode_32 0
\t\t.amdhsa_float_round_mode_16_64 0
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t\t.amdhsa_fp16_overflow 0
\t\t.amdhsa_tg_split 0
\t\t.amdhsa_exception_fp_ieee_invalid_op 0
\t\t.amdhsa_exception_fp_denorm_src 0
\t\t.amdhsa_exception_fp_ieee_div_zero 0
\t\t.amdhsa_exception_fp_ieee_overflow 0
\t\t.amdhsa_exception_fp_ieee_underflow 0
\t\t.amdhsa_exception_fp_ieee_inexact 0
\t\t.amdhsa_exception_int_div_zero 0
\t.end_amdhsa_kernel
\t.text
\t.amdgpu_metadata
---
amdhsa.kernels:
  - .agpr_count:     {nagpr}
    .args:
      - {{.address_space: global, .offset: 0, .size: 8, .value_kind: global_buffer}}
      - {{.address_space: global, .offset: 8, .size: 8, .value_kind: global_buffer}}
      - {{.address_space: global, .offset: 16, .size: 8, .value_kind: global_buffer}}
      - {{.offset: 24, .size: 4, .value_kind: by_value}}
      - {{.offset: 28, .size: 4, .value_kind: by_value}}
      - {{.offset: 32, .size: 4, .value_kind: by_value}}
      - {{.offset: 36, .size: 4, .value_kind: by_value}}
      - {{.offset: 40, .size: 4, .value_kind: by_value}}
      - {{.offset: 44, .size: 4, .value_kind: by_value}}
      - {{.offset: 48, .size: 4, .value_kind: by_value}}
      - {{.offset: 52, .size: 4, .value_kind: by_value}}
      - {{.offset: 56, .size: 4, .value_kind: by_value}}
      - {{.offset: 60, .size: 4, .value_kind: by_value}}
      - {{.address_space: global, .offset: 64, .size: 8, .value_kind: global_buffer}}
    .group_segment_fixed_size: {lds}
    .kernarg_segment_align: 8
    .kernarg_segment_size: {kargs}
    .max_flat_workgroup_size: 256
    .name:           {name}
    .private_segment_fixed_size: 0
    .sgpr_count:     102
    .sgpr_spill_count: 0
    .symbol:         {name}.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     {nvgpr}
    .vgpr_spill_count: 0
    .wavefront_size: 64
amdhsa.target:   amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
\t.end_amdgpu_metadata
"""


def kernel_text(dtype="bf16", name=None, **kw):
    name = name or (KERNEL_NAME if dtype == "bf16" else KERNEL_NAME + "_f16")
    g = Gen(dtype=dtype, **kw)
    prog = g.build()
    body = prog.text().replace(KERNEL_NAME + ":", name + ":")
    txt = HEADER.format(name=name) + body + FOOTER.format(name=name, lds=g.lds_bytes, kargs=KARG_BYTES, nvgpr=ARCH_VGPRS + ACC_VGPRS, accum=ARCH_VGPRS,
                                                        nagpr=ACC_VGPRS)
    return txt, g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--table", help="write the per-gap issue table here")
    a = ap.parse_args()
    txt, g = kernel_text(a.dtype)
    with open(a.out, "w") as f:
        f.write("; GENERATED by tools/attn_asm/gen_attn.py -- do not edit; edit the generator.\n" + txt)
    if a.table:
        with open(a.table, "w") as f:
            f.write("# iteration step gap mfma fillers_after\n")
            for row in g.issue_rows:
                f.write(" ".join(str(x) for x in row) + "\n")


if __name__ == "__main__":
    main()
