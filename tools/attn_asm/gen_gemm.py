#!/usr/bin/env python3
"""Generator of the hand-placed bf16 GEMM  out[M, N] = A[M, K] W[N, K]^T + bias  (bf16 in, f32 accumulate, bf16 out) for gfx950: the
persistent, parked-tile form of the large-tile kernel (csrc/gemm_bf16_tiles.h: gemm_bf16_big_kernel) that VERDICT r4 item 2 asks for.

Shape of the kernel (K fixed at generation time, K % 128 == 0; N % 256 == 0):
  * persistent grid of 8 x 32 workgroups, 8 waves each (2 in M x 4 in N), one workgroup per CU, two waves per SIMD (252 registers);
    a workgroup walks its XCD's chunk of the 256 x 256 output tiles with stride 32;
  * per wave 128 x 64 outputs = 8 x 4 MFMA tiles (v_mfma_f32_16x16x32) with the operands SWAPPED (weight fragment first): the accumulator
    block is the transposed tile -- lane (q = lane >> 4, c = lane & 15) holds row c, columns 4q .. 4q + 3 -- so that two column blocks
    make 16-byte row chunks with two v_permlane16_swap and no LDS staging;
  * the K-tiles of consecutive output tiles run as ONE pipeline: LDS = two 64-KiB K-tile buffers {A rows 0..255 | W rows 0..255} fed by
    16-byte LDS-DMA one K-tile ahead, the seam included (the first K-tiles of the next output tile are requested during the last ones of
    this tile);
  * a K-tile is two k-steps; per k-step and wave group an interval R (12 ds_read_b128 into 48 fragment registers) and an interval M
    (32 MFMAs), separated by raw s_barriers; the wm = 1 waves run one barrier behind the wm = 0 waves, so each SIMD alternates a wave
    in M with a wave in R (the product kernel's choreography, by hand);
  * at the seam a wave converts its 128 accumulator registers to 64 PARKED registers of packed bf16 (64 v_cvt_pk + 32 swaps) and the 16
    stores of 16 bytes per lane are issued two per K-tile during K-tiles 1..8 of the NEXT tile, behind counted vmcnt: the store-bound
    drain (8.2 k cycles per tile at the CU's 16 B/clk) overlaps the next tile's main loop;
  * bias: the whole vector sits in LDS (DMA, once); at the seam row block 0's accumulators are loaded with it and serve as the C operand
    of every row block's first MFMA.

Registers: v0-127 accumulators, v128-191 parked tile, v192-203 addresses; a0-31 activation fragments, a32-47 weight fragments.
"""
import argparse

from .isa import Prog, V, A, S, I, Lit, M0, R

KERNEL_NAME = "ucod_gemm_pk"
BUF = 65536
A_BYTES = 32768
BIAS_LDS = 2 * BUF
KARG_BYTES = 72

# ---- VGPRs
def acc(i, j):
    return V((i * 4 + j) * 4, 4)


def park(s):
    return V(128 + 4 * s, 4)


v_tid = V(0)
v_dma = [V(192 + x) for x in range(4)]
v_ldsA = [V(196), V(197)]
v_ldsB = [V(198), V(199)]
v_st = V(200)
v_bias = V(201)
v_t = [V(202), V(203)]
ARCH_VGPRS = 204
ACC_VGPRS = 48


def fragA(i):
    return A(4 * i, 4)


def fragB(j):
    return A(32 + 4 * j, 4)


# ---- SGPRs
s_karg = S(0, 2)
s_wg = S(2)
s_A, s_W, s_biasp, s_out = S(4, 2), S(6, 2), S(8, 2), S(10, 2)
s_M, s_N, s_tiles_n, s_ntiles = S(12), S(13), S(14), S(15)
s_magic, s_add1, s_chq, s_chr = S(16), S(17), S(18), S(19)
s_dbg = S(20, 2)
s_w, s_wm, s_wn = S(22), S(23), S(24)
s_idx, s_end, s_has_next = S(25), S(26), S(27)
s_dA = [S(28, 4), S(36, 4)]       # [cur, nxt]
s_dB = [S(32, 4), S(40, 4)]
s_dO = [S(44, 4), S(48, 4)]       # [cur, nxt]
s_dP = S(52, 4)                   # the parked tile's output descriptor
s_dBias = S(56, 4)
s_m0A = [S(60 + x) for x in range(4)]
s_m0B = [S(64 + x) for x in range(4)]
s_row = [S(68 + i) for i in range(8)]      # i * 16 * N * 2
s_n0x4 = [S(76), S(77)]                    # [cur, nxt]
s_koff = S(78)
s_t = [S(80 + k) for k in range(12)]
DESC3 = 0x00020000


class GemmGen:
    def __init__(self, K=768, abl=(), stores_from=1, stores_per_kt=2, stamps=False, stride=32):
        assert K % 128 == 0
        self.K, self.K2, self.NKT = K, 2 * K, K // 64
        self.abl = set(abl)
        self.p = Prog()
        self.lds_bytes = 160 * 1024
        self.stores_from, self.stores_per_kt = stores_from, stores_per_kt
        self.stamps = stamps
        self.stride = stride                                      # workgroups per XCD (the simulator runs one workgroup with stride 1)
        # which parked stores go into which K-tile of the next tile
        self.store_plan = {}
        s = 0
        t = stores_from
        while s < 16:
            assert t < self.NKT - 1, "parked stores do not fit the K loop"
            self.store_plan[t] = list(range(s, min(16, s + stores_per_kt)))
            s += stores_per_kt
            t += 1

    # ------------------------------------------------------------------ helpers
    def ptr_add(self, dst, base, off32):
        """dst[0:1] = base[0:1] + off32 (unsigned 32-bit)"""
        p = self.p
        p.s_add_u32(dst[0], base[0], off32)
        p.s_addc_u32(dst[1], base[1], I(0))

    def make_desc(self, idx, which):
        """descriptors of tile `idx` (SGPR) into set `which` (0 cur / 1 nxt); idx >= s_end -> empty descriptors.  Scalar only, ~35 instructions."""
        p = self.p
        tm, tn, m0, n0, rows, valid, x, y = s_t[0], s_t[1], s_t[2], s_t[3], s_t[4], s_t[5], s_t[6], s_t[7]
        p.s_cmp("lt", "u32", idx, s_end)
        p.s_cselect_b32(valid, I(1), I(0))
        p.s_mul_hi_u32(tm, idx, s_magic)
        p.s_mul_i32(x, idx, s_add1)
        p.s_add_u32(tm, tm, x)
        p.s_mul_i32(x, tm, s_tiles_n)
        p.s_sub_u32(tn, idx, x)
        p.s_lshl_b32(m0, tm, 8)
        p.s_lshl_b32(n0, tn, 8)
        p.s_sub_u32(rows, s_M, m0)
        p.s_max_i32(rows, rows, I(0))
        p.s_min_u32(rows, rows, I(256))
        p.s_mul_i32(rows, rows, valid)
        dA, dB, dO = s_dA[which], s_dB[which], s_dO[which]
        # A: base + m0 * K2, rows * K2 bytes
        p.s_mul_i32(x, m0, I(self.K2))
        self.ptr_add(dA, s_A, x)
        p.s_mul_i32(dA[2], rows, I(self.K2))
        # W: base + n0 * K2, 256 rows
        p.s_mul_i32(x, n0, I(self.K2))
        self.ptr_add(dB, s_W, x)
        p.s_mul_i32(dB[2], valid, I(256 * self.K2))
        # out: base + (m0 * N + n0) * 2, records = rows * N * 2 - n0 * 2 (0 when rows == 0)
        p.s_mul_i32(x, m0, s_N)
        p.s_add_u32(x, x, n0)
        p.s_lshl_b32(x, x, 1)
        self.ptr_add(dO, s_out, x)
        p.s_mul_i32(x, rows, s_N)
        p.s_sub_u32(x, x, n0)
        p.s_lshl_b32(x, x, 1)
        p.s_cmp("eq", "u32", rows, I(0))
        p.s_cselect_b32(dO[2], I(0), x)
        for d in (dA, dB, dO):
            p.s_and_b32(d[1], d[1], I(0xFFFF))
        p.s_lshl_b32(s_n0x4[which], n0, 2)

    def dma_group(self, kt, which, buf, pieces=range(8)):
        """the LDS-DMAs (this wave's 4 A pieces and 4 W pieces, or a subset) of K-tile kt of the tile whose descriptors are set `which`, into buffer buf.
        Returns a list of emitters, one per piece (each: M0 write + DMA)."""
        p = self.p
        out = []

        def piece(x):
            def emit():
                isA = x < 4
                m0s = (s_m0A if isA else s_m0B)[x & 3]
                if buf:
                    p.s_add_u32(M0, m0s, I(BUF))
                else:
                    p.s_mov_b32(M0, m0s)
                if "nodma" in self.abl:
                    return
                p.s_nop(0)
                p.buffer_load_lds_dwordx4(v_dma[x & 3], (s_dA if isA else s_dB)[which], s_koff)
            return emit
        for x in pieces:
            out.append(piece(x))
        return out

    def set_koff(self, kt):
        self.p.s_mov_b32(s_koff, I(kt * 128))

    def frag_reads(self, ks):
        """12 ds_read_b128 of k-step ks: 8 activation row blocks + 4 weight column blocks (into the single fragment set)"""
        p = self.p
        out = []
        for i in range(8):
            out.append(lambda i=i: p.ds_read_b128(fragA(i), v_ldsA[ks], i * 2048))
        for j in range(4):
            out.append(lambda j=j: p.ds_read_b128(fragB(j), v_ldsB[ks], j * 2048))
        if "nolds" in self.abl:
            return []
        return out

    def mfma_list(self, first):
        """the 32 MFMAs of a k-step as emitters; first: the tile's first k-step (C = row block 0's accumulators = the bias; row block 0 last)"""
        p = self.p
        out = []
        order = list(range(1, 8)) + [0] if first else list(range(8))
        for i in order:
            for j in range(4):
                c = acc(0, j) if first else acc(i, j)
                out.append(lambda i=i, j=j, c=c: p.mfma16(acc(i, j), fragB(j), fragA(i), c))
        if "nomfma" in self.abl:
            return []
        return out

    def toggle_lds(self):
        p = self.p
        for v in v_ldsA + v_ldsB:
            p.v_xor_b32(v, I(BUF), v)

    def convert(self):
        """accumulators -> parked registers: store s = (row block s >> 1, column-block pair s & 1)"""
        p = self.p
        pend = None
        for s in range(16):
            i, jp = s >> 1, s & 1
            pk = park(s)
            for h in range(2):
                a_ = acc(i, 2 * jp + h)
                p.v_cvt_pk_bf16_f32(pk[2 * h], a_[0], a_[1])
                p.v_cvt_pk_bf16_f32(pk[2 * h + 1], a_[2], a_[3])
            if pend is not None:
                p.v_permlane16_swap_b32(pend[0], pend[2])
                p.v_permlane16_swap_b32(pend[1], pend[3])
            pend = pk
        p.s_nop(1)
        p.v_permlane16_swap_b32(pend[0], pend[2])
        p.v_permlane16_swap_b32(pend[1], pend[3])

    def store(self, s, desc, tmp):
        p = self.p
        i, jp = s >> 1, s & 1
        p.v_add_u32(tmp, s_row[i], v_st)
        if "nostore" in self.abl:
            return
        p.buffer_store_dwordx4(park(s), tmp, desc, I(0), jp * 64)

    def seam(self):
        """top of a tile: park the finished tile (garbage before the first: its descriptor is empty), rotate the descriptor sets, bias -> row block 0"""
        p = self.p
        p.s_nop(7)
        p.s_nop(3)
        self.convert()
        for k in range(4):
            p.s_mov_b32(s_dP[k], s_dO[0][k])
        for d in (s_dA, s_dB, s_dO):
            for k in range(3):
                p.s_mov_b32(d[0][k], d[1][k])
        p.s_mov_b32(s_n0x4[0], s_n0x4[1])
        p.v_add_u32(v_t[0], s_n0x4[0], v_bias)
        for j in range(4):
            p.ds_read_b128(acc(0, j), v_t[0], j * 64)

    # ------------------------------------------------------------------ one tile, one wave group
    def tile_body(self, g):
        """12 K-tiles of one output tile for wave group g (0: leading, 1: one barrier behind)"""
        p = self.p
        NKT = self.NKT
        for t in range(NKT):
            buf = t & 1
            first = t == 0
            stores = self.store_plan.get(t, [])
            p.comment(f"---- group {g} K-tile {t}")
            # ---------------- R0
            if first:
                self.seam()
            if g == 0:
                # DMA of K-tile t+1 (the next tile's K-tile 0 at t = NKT-1) into the other buffer: all waves are past the barrier that ends its last reads
                kt, which = (t + 1, 0) if t + 1 < NKT else (0, 1)
                self.set_koff(kt)
                for e in self.dma_group(kt, which, buf ^ 1):
                    e()
            else:
                p.s_nop(7)
            for n_, s in enumerate(stores):
                self.store(s, s_dP, v_t[n_ & 1])
            for e in self.frag_reads(0):
                e()
            p.s_waitcnt(lgkmcnt=0)
            p.s_barrier()
            # ---------------- M0
            p.s_setprio(1)
            for e in self.mfma_list(first):
                e()
            p.s_setprio(0)
            p.s_barrier()
            # ---------------- R1
            p.s_nop(7)
            for e in self.frag_reads(1):
                e()
            if t == 2:
                # descriptors of the tile after this one (scalar work, hidden in a read interval)
                p.s_add_u32(s_t[8], s_idx, I(self.stride))
                self.make_desc(s_t[8], 1)
                p.s_cmp("lt", "u32", s_t[8], s_end)
                p.s_cselect_b32(s_has_next, I(1), I(0))
            self.toggle_lds()
            p.s_waitcnt(lgkmcnt=0)
            if g == 1:
                # this wave's DMAs of K-tile t+1 (issued in the previous M1) have landed; younger: the parked stores of this K-tile
                p.s_waitcnt(vmcnt=len(stores))
            p.s_barrier()
            # ---------------- M1
            p.s_setprio(1)
            mf = self.mfma_list(False)
            if g == 1:
                # DMA of K-tile t+2 into THIS buffer... no: into buffer (t+2)&1 = buf, whose last reads (this K-tile's k-step 1) ended at the barrier above
                kt2 = t + 2
                kt, which = (kt2, 0) if kt2 < NKT else (kt2 - NKT, 1)
                self.set_koff(kt)
                dm = self.dma_group(kt, which, buf)
                for k, e in enumerate(mf):
                    e()
                    if k % 4 == 1 and k // 4 < len(dm):
                        dm[k // 4]()
                if not mf:
                    for e in dm:
                        e()
            else:
                for e in mf:
                    e()
            p.s_setprio(0)
            if g == 0:
                p.s_waitcnt(vmcnt=len(stores))
            p.s_barrier()

    # ------------------------------------------------------------------ whole program
    def build(self):
        p = self.p
        p.label(KERNEL_NAME)
        p.s_load(S(4, 8), s_karg, 0)             # A W bias out
        p.s_load(S(12, 8), s_karg, 32)           # M N tiles_n ntiles magic add1 chunk_q chunk_r
        p.s_load(s_dbg, s_karg, 64)
        p.s_waitcnt(lgkmcnt=0)
        lane, l15, q, r8, c8, x0, x1 = V(1), V(2), V(3), V(4), V(5), V(6), V(7)
        p.v_and_b32(lane, I(63), v_tid)
        p.v_lshrrev_b32(x0, I(6), v_tid)
        p.s_nop(0)
        p.v_readfirstlane_b32(s_w, x0)
        p.s_lshr_b32(s_wm, s_w, 2)
        p.s_and_b32(s_wn, s_w, I(3))
        # ---- this workgroup's tiles: XCD x = wg & 7 owns [x chq + min(x, chr), + chq + (x < chr)); workgroup j = wg >> 3 takes start + j, + 32, ...
        xcd, j, start, cnt = s_t[0], s_t[1], s_t[2], s_t[3]
        p.s_and_b32(xcd, s_wg, I(7))
        p.s_lshr_b32(j, s_wg, 3)
        p.s_mul_i32(start, xcd, s_chq)
        p.s_min_u32(s_t[4], xcd, s_chr)
        p.s_add_u32(start, start, s_t[4])
        p.s_cmp("lt", "u32", xcd, s_chr)
        p.s_cselect_b32(cnt, I(1), I(0))
        p.s_add_u32(cnt, cnt, s_chq)
        p.s_add_u32(s_end, start, cnt)
        p.s_add_u32(s_idx, start, j)
        p.s_cmp("lt", "u32", s_idx, s_end)
        lab_go = p.newlabel("go")
        p.s_cbranch("scc1", lab_go)
        p.s_endpgm()
        p.label(lab_go)
        # ---- lane constants
        p.v_and_b32(l15, I(15), lane)
        p.v_lshrrev_b32(q, I(4), lane)
        p.v_lshrrev_b32(r8, I(3), lane)
        p.v_and_b32(c8, I(7), lane)
        # DMA source offsets: row = 64 x + 8 w + r8, 16-byte chunk c8 ^ r8 (the LDS image is the swizzled one)
        p.v_xor_b32(x0, c8, r8)
        p.v_lshlrev_b32(x0, I(4), x0)
        p.s_lshl_b32(s_t[4], s_w, 3)
        p.v_add_u32(x1, s_t[4], r8)
        for x in range(4):
            p.v_add_u32(v_t[0], I(64 * x), x1)
            p.v_mul_u32_u24(v_t[0], I(self.K2), v_t[0])
            p.v_add_u32(v_dma[x], v_t[0], x0)
        # fragment read addresses: row (wm 128 | wn 64) + l15, chunk (4 ks + q) ^ (l15 & 7)
        p.v_and_b32(x0, I(7), l15)
        p.s_lshl_b32(s_t[4], s_wm, 14)               # wm * 128 rows * 128 bytes
        p.s_lshl_b32(s_t[5], s_wn, 13)               # wn * 64 rows * 128 bytes
        p.s_add_u32(s_t[5], s_t[5], I(A_BYTES))
        for ks in range(2):
            p.v_or_b32(x1, I(4 * ks), q)
            p.v_xor_b32(x1, x1, x0)
            p.v_lshlrev_b32(x1, I(4), x1)
            p.v_lshl_add_u32(x1, l15, I(7), x1)
            p.v_add_u32(v_ldsA[ks], s_t[4], x1)
            p.v_add_u32(v_ldsB[ks], s_t[5], x1)
        # store offsets: row (wm 128 + l15) * N * 2 + (wn 64 + 16 (q & 1) + 8 (q >> 1)) * 2
        p.s_lshl_b32(s_t[4], s_wm, 7)
        p.v_add_u32(x0, s_t[4], l15)
        p.s_lshl_b32(s_t[6], s_N, 1)
        p.v_mul_lo_u32(x0, x0, s_t[6])
        p.v_and_b32(x1, I(1), q)
        p.v_lshlrev_b32(x1, I(4), x1)
        p.v_lshrrev_b32(v_t[0], I(1), q)
        p.v_lshl_add_u32(x1, v_t[0], I(3), x1)
        p.s_lshl_b32(s_t[5], s_wn, 6)
        p.v_add_u32(x1, s_t[5], x1)
        p.v_lshl_add_u32(v_st, x1, I(1), x0)
        # bias read offsets: BIAS_LDS + (wn 64 + 4 q) * 4
        p.v_lshlrev_b32(x1, I(2), q)
        p.v_add_u32(x1, s_t[5], x1)
        p.v_lshlrev_b32(x1, I(2), x1)
        p.v_add_u32(v_bias, I(BIAS_LDS), x1)
        # scalars: M0 bases of the DMA pieces, row-block offsets of the stores
        for x in range(4):
            p.s_lshl_b32(s_t[4], s_w, 10)
            p.s_add_u32(s_m0A[x], s_t[4], I(x * 8192))
            p.s_add_u32(s_m0B[x], s_m0A[x], I(A_BYTES))
        p.s_lshl_b32(s_t[4], s_N, 5)                 # 16 rows * N * 2 bytes
        p.s_mov_b32(s_row[0], I(0))
        for i in range(1, 8):
            p.s_add_u32(s_row[i], s_row[i - 1], s_t[4])
        # descriptors
        for d in s_dA + s_dB + s_dO + [s_dP, s_dBias]:
            p.s_mov_b32(d[3], I(DESC3))
        for d in (s_dA[0], s_dB[0], s_dO[0], s_dP):
            for k in range(3):
                p.s_mov_b32(d[k], I(0))
        p.s_mov_b32(s_dBias[0], s_biasp[0])
        p.s_and_b32(s_dBias[1], s_biasp[1], I(0xFFFF))
        p.s_lshl_b32(s_dBias[2], s_N, 2)
        self.make_desc(s_idx, 1)
        # the bias vector -> LDS: 1-KiB pieces w, w + 8, ... (beyond N * 4 bytes the descriptor returns zeros)
        p.v_lshlrev_b32(x0, I(4), lane)
        nb = p.newlabel("bias_loop")
        p.s_mov_b32(s_t[4], s_w)
        p.label(nb)
        p.s_lshl_b32(s_t[5], s_t[4], 10)
        p.s_add_u32(M0, s_t[5], I(BIAS_LDS))
        p.v_add_u32(x1, s_t[5], x0)
        p.s_nop(3)
        p.buffer_load_lds_dwordx4(x1, s_dBias, I(0))
        p.s_add_u32(s_t[4], s_t[4], I(8))
        p.s_lshl_b32(s_t[5], s_t[4], 8)              # piece * 256 floats
        p.s_cmp("lt", "u32", s_t[5], s_N)
        p.s_cbranch("scc1", nb)
        # K-tile 0 of the first tile (both groups); the trailing group also its share of K-tile 1 (the leading group issues it in its first read interval)
        lab_g1, lab_g1_loop, lab_g0_loop, lab_tail, lab_end = (p.newlabel(n) for n in ("g1", "g1_loop", "g0_loop", "tail", "end"))
        self.set_koff(0)
        for e in self.dma_group(0, 1, 0):
            e()
        p.s_cmp("eq", "u32", s_wm, I(0))
        lab_g0 = p.newlabel("g0")
        p.s_cbranch("scc1", lab_g0)
        # ================= trailing group
        self.set_koff(1)
        for e in self.dma_group(1, 1, 1):
            e()
        p.s_waitcnt(vmcnt=8)
        p.s_barrier()
        p.s_barrier()                                # one interval behind
        p.label(lab_g1_loop)
        self.tile_body(1)
        p.s_add_u32(s_idx, s_idx, I(self.stride))
        p.s_cmp("eq", "u32", s_has_next, I(1))
        p.s_cbranch("scc1", lab_g1_loop)
        p.s_branch(lab_tail)
        # ================= leading group
        p.label(lab_g0)
        p.s_waitcnt(vmcnt=0)
        p.s_barrier()
        p.label(lab_g0_loop)
        self.tile_body(0)
        p.s_add_u32(s_idx, s_idx, I(self.stride))
        p.s_cmp("eq", "u32", s_has_next, I(1))
        p.s_cbranch("scc1", lab_g0_loop)
        p.s_barrier()                                # the trailing group's extra interval
        # ================= last tile: convert and store directly
        p.label(lab_tail)
        p.s_nop(7)
        p.s_nop(3)
        self.convert()
        p.s_nop(1)
        for s in range(16):
            self.store(s, s_dO[0], v_t[s & 1])
        p.s_waitcnt(vmcnt=0)
        p.s_endpgm()
        return p


HEADER = """\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.amdhsa_code_object_version 6
\t.text
\t.globl {name}
\t.p2align 8
\t.type {name},@function
"""

FOOTER = """
\t.section .rodata,"a",@progbits
\t.p2align 6, 0x0
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
.Lfunc_end0:
\t.size {name}, .Lfunc_end0-{name}

\t.amdgpu_metadata
---
amdhsa.kernels:
  - .agpr_count:     {nagpr}
    .args:
      - {{.address_space: global, .offset: 0, .size: 8, .value_kind: global_buffer}}
      - {{.address_space: global, .offset: 8, .size: 8, .value_kind: global_buffer}}
      - {{.address_space: global, .offset: 16, .size: 8, .value_kind: global_buffer}}
      - {{.address_space: global, .offset: 24, .size: 8, .value_kind: global_buffer}}
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
    .max_flat_workgroup_size: 512
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


def kernel_text(K=768, name=None, **kw):
    name = name or f"{KERNEL_NAME}_k{K}"
    g = GemmGen(K=K, **kw)
    prog = g.build()
    body = prog.text().replace(KERNEL_NAME + ":", name + ":")
    txt = HEADER.format(name=name) + body + FOOTER.format(name=name, lds=g.lds_bytes, kargs=KARG_BYTES, nvgpr=ARCH_VGPRS + ACC_VGPRS, accum=ARCH_VGPRS,
                                                        nagpr=ACC_VGPRS)
    return txt, g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-K", type=int, default=768)
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--abl", default="", help="comma-separated timing-only ablations: nodma,nolds,nomfma,nostore")
    ap.add_argument("--name")
    a = ap.parse_args()
    txt, g = kernel_text(a.K, name=a.name, abl=[x for x in a.abl.split(",") if x])
    with open(a.out, "w") as f:
        f.write("; GENERATED by tools/attn_asm/gen_gemm.py -- do not edit; edit the generator.\n" + txt)
    print(f"{a.out}: {g.p.count()} instructions")


if __name__ == "__main__":
    main()
