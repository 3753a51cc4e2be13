#!/usr/bin/env python3
"""Generator of the hand-placed bf16 GEMM  out[M, N] = A[M, K] W[N, K]^T + bias  (bf16 in, f32 accumulate, bf16 out) for gfx950: the
persistent, parked-tile form of the large-tile kernel (csrc/gemm_bf16_tiles.h: gemm_bf16_big_kernel) that VERDICT r4 item 2 asked for.
LABORATORY (variants/gemm_asm_lab.hip): 4-10 % faster than the product kernel at whole rounds, the 105 us gate not met; DESIGN.md section 0 item 2 has
the numbers and the twenty-form ablation table this generator builds (VARIANTS below).
What it computes: transformers modeling_dinov2.py:153-179 (query / key / value projections), :348-381 (the MLP's fc1 without its activation).

Shape of the kernel (K fixed at generation time, K % 128 == 0; N % 256 == 0):
  * persistent grid of 8 x 32 workgroups, 8 waves each (2 in M x 4 in N), one workgroup per CU, two waves per SIMD (252 registers);
    a workgroup walks its XCD's chunk of the 256 x 256 output tiles with stride 32;
  * per wave 128 x 64 outputs = 8 x 4 MFMA tiles (v_mfma_f32_16x16x32) with the operands SWAPPED (weight fragment first): the accumulator
    block is the transposed tile -- lane (q = lane >> 4, c = lane & 15) holds row c, columns 4q .. 4q + 3 -- so that two column blocks
    make 16-byte row chunks with two v_permlane16_swap and no LDS staging;
  * the K-tiles of consecutive output tiles run as ONE pipeline: LDS = two 64-KiB K-tile buffers {A rows 0..255 | W rows 0..255} fed by
    16-byte LDS-DMA one K-tile ahead, the seam included (the first K-tile of the next output tile is requested during the last one of
    this tile);
  * a K-tile is two k-steps; per k-step and wave group an interval R (12 ds_read_b128 into 48 fragment registers) and an interval M
    (32 MFMAs), separated by raw s_barriers; the wm = 1 waves run one barrier behind the wm = 0 waves, so each SIMD alternates a wave
    in M with a wave in R (the product kernel's choreography, by hand);
  * every LDS-DMA and every store is issued from an R interval (a wave that issues into a full vector-memory queue stalls at issue: in an M
    interval that stall is matrix time): the 32 + 32 one-KiB pieces of a K-tile are owned 10 per leading-group wave (5 in R0, 5 in R1, waited
    for at the end of M1) and 6 per trailing-group wave (R0, waited for at the end of R1) -- tile_body() has the barrier arithmetic;
  * at the seam a wave converts its 128 accumulator registers to 64 PARKED registers of packed bf16 (64 v_cvt_pk + 32 swaps) and the 16
    non-temporal stores of 16 bytes per lane are issued two per K-tile during K-tiles 1..8 of the NEXT tile, behind counted vmcnt: the
    store-bound drain (8.2 k cycles per tile at the CU's 16 B/clk) overlaps the next tile's main loop;
  * bias: the whole vector sits in LDS (DMA, once); at the seam row block 0's accumulators are loaded with it and serve as the C operand
    of every row block's first MFMA.

Registers: v0-127 accumulators, v128-191 parked tile, v192-204 addresses; a0-31 activation fragments, a32-47 weight fragments.
Checked on the CPU by the functional simulator (sim.py: results, LDS-DMA / barrier / counted-wait protocol) and the static wait-state audit
(checks.py): tests/test_attn_asm.py; on the GPU bitwise against the product kernel: tests/test_gpu_kernels.py::test_gemm_assembly_kernel.
"""
import argparse

from .isa import Prog, V, A, S, I, M0

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
v_dma = [V(192 + x) for x in range(5)]   # row sets 8 (pp0 + 4 k) + (lane >> 3), k = 0..4 (the trailing group uses 3)
v_ldsA = [V(197), V(198)]
v_ldsB = [V(199), V(200)]
v_st = V(201)
v_bias = V(202)
v_t = [V(203), V(204)]
ARCH_VGPRS = 208
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
s_m0base = S(60)                           # LDS address of this wave's first activation piece: pp0 * 1024
s_pp0 = S(61)
s_row = [S(68 + i) for i in range(8)]      # i * 16 * N * 2
s_n0x4 = [S(76), S(77)]                    # [cur, nxt]
s_koff = S(78)
s_t = [S(80 + k) for k in range(10)]
s_now = S(90, 2)                           # diagnostic builds (stamps): s_memtime, previous stamp, 10 interval accumulators
s_prev = S(92)
NSTAMP = 8
# (the accumulators reuse the registers of the prefetch experiments' scratch and sit above every live scalar of the kernel)
DESC3 = 0x00020000


class GemmGen:
    def __init__(self, K=768, abl=(), stores_from=1, stores_per_kt="spread", stamps=False, stride=32, nk=(5, 3), g0_r0=5, g1_where="R0", pol_a="", pol_b="", pol_st="nt", sync=4):
        assert K % 128 == 0
        self.K, self.K2, self.NKT = K, 2 * K, K // 64
        self.abl = set(abl)
        self.p = Prog()
        self.lds_bytes = 160 * 1024
        self.stores_from, self.stores_per_kt = stores_from, stores_per_kt
        self.stamps = stamps
        self.stride = stride                                      # workgroups per XCD (the simulator runs one workgroup with stride 1)
        # the 32 + 32 one-KiB DMA pieces of a K-tile: a leading-group wave owns row sets k = 0..nk[0]-1 of both operands, a trailing-group wave nk[1];
        # the leading group issues its first g0_r0 pieces in R0 and the rest in R1, the trailing group all of its pieces in R0 (or in M intervals)
        assert 4 * nk[0] + 4 * nk[1] == 32 and max(nk) <= 5
        self.nk, self.g0_r0, self.g1_where = nk, g0_r0, g1_where
        self.pol_a, self.pol_b, self.pol_st = pol_a, pol_b, pol_st     # cache policies of the activation / weight DMAs and of the output stores
        self.sync = sync                                          # barriers per K-tile: 4 (R | M intervals fenced) or 1 (groups skewed by one k-step, free-running inside the K-tile)
        # stamps: 0 R0 work, 1 barrier after an R interval, 2 M0 work, 3 barrier after an M interval, 4 R1 work, 5 wait for the DMAs, 6 M1 work, 7 seam
        self.s_acc = [S(94 + k) for k in range(8)]
        # which parked stores go into which K-tile of the next tile: stores_per_kt from K-tile stores_from on, or (stores_per_kt = "spread") evenly over
        # all K-tiles of the tile (the store path is then loaded at 1.33 stores per wave and K-tile instead of 2 for two thirds of the tile)
        self.store_plan = {}
        if stores_per_kt == "spread":
            done = 0
            for t in range(self.NKT):
                upto = (16 * (t + 1) + self.NKT - 1) // self.NKT
                self.store_plan[t] = list(range(done, upto))
                done = upto
            assert done == 16
        else:
            s = 0
            t = stores_from
            while s < 16:
                assert t < self.NKT, "parked stores do not fit the K loop"
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
        tm, tn, m0, n0, rows, valid, x = s_t[0], s_t[1], s_t[2], s_t[3], s_t[4], s_t[5], s_t[6]
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

    def pieces(self, g):
        """this wave's DMA pieces of one K-tile as (operand, k): activation and weight alternate so that both streams are requested at the same rate"""
        return [(op, k) for k in range(self.nk[g]) for op in ("A", "B")]

    def dma(self, piece, which, buf):
        """emitter of one LDS-DMA piece (M0 write, one wait state, DMA) of the K-tile whose k offset is in s_koff, tile descriptors `which`, buffer buf"""
        p = self.p
        op, k = piece

        def emit():
            isA = op == "A"
            p.s_add_u32(M0, s_m0base, I(k * 4096 + (0 if isA else A_BYTES) + (BUF if buf else 0)))
            if "nodma" in self.abl or ("noA" in self.abl and isA) or ("noB" in self.abl and not isA):
                return
            p.s_nop(0)
            p.buffer_load_lds_dwordx4(v_dma[k], (s_dA if isA else s_dB)[which], s_koff, policy=self.pol_a if isA else self.pol_b)
        return emit

    def set_koff(self, kt):
        self.p.s_mov_b32(s_koff, I(kt * 128))

    def frag_reads(self, ks):
        """12 ds_read_b128 of k-step ks: 8 activation row blocks + 4 weight column blocks (into the single fragment set)"""
        p = self.p
        out = []
        ra = [lambda i=i: p.ds_read_b128(fragA(i), v_ldsA[ks], i * 2048) for i in range(8)]
        rb = [lambda j=j: p.ds_read_b128(fragB(j), v_ldsB[ks], j * 2048) for j in range(4)]
        out = ra[1:7] + rb[:3] + [ra[0], ra[7], rb[3]]              # (the last MFMAs of a k-step read A block 7 -- block 0 in a tile's first k-step -- and W block 3: rewritten last)
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

    def stamp(self, k):
        """diagnostic builds: add the cycles since the previous stamp to accumulator k (k None: only restart the clock)"""
        if not self.stamps:
            return
        p = self.p
        p.s_memtime(s_now)
        p.s_waitcnt(lgkmcnt=0)
        if k is not None:
            p.s_sub_u32(s_t[9], s_now[0], s_prev)
            p.s_add_u32(self.s_acc[k], self.s_acc[k], s_t[9])
        p.s_mov_b32(s_prev, s_now[0])

    def barrier(self):
        if "nobar" not in self.abl:
            self.p.s_barrier()

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
        p.buffer_store_dwordx4(park(s), tmp, desc, I(0), jp * 64, policy=self.pol_st)

    def seam(self):
        """top of a tile: park the finished tile (garbage before the first: its descriptor is empty), rotate the descriptor sets, bias -> row block 0"""
        p = self.p
        p.s_nop(7)
        p.s_nop(3)
        if "noconv" not in self.abl:
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
        """12 K-tiles of one output tile for wave group g (0: leading, 1: one barrier behind).  Global barrier B(4T) ends the last reads of K-tile T-1 and
        precedes the leading group's first read of K-tile T: the DMAs of K-tile T go out between B(4T-4) and B(4T), from READ intervals only (a wave that
        issues into a full vector-memory queue stalls; in an MFMA interval that stall is matrix time) --
          leading group:  R0(T-1) and R1(T-1), waited for at the end of M1(T-1);   trailing group: R0(T-1), waited for at the end of R1(T-1).
        The parked stores of a K-tile are issued behind the last DMAs of the interval that ends with the wait, so the counted wait leaves exactly them."""
        p = self.p
        NKT = self.NKT
        pc = self.pieces(g)
        for t in range(NKT):
            buf = t & 1
            first = t == 0
            stores = self.store_plan.get(t, [])
            kt, which = (t + 1, 0) if t + 1 < NKT else (0, 1)         # the K-tile being requested: t + 1 (the next tile's K-tile 0 at the end)
            p.comment(f"---- group {g} K-tile {t}")
            # ---------------- R0
            self.stamp(None)
            if first:
                self.seam()
                self.stamp(7)
            else:
                p.s_nop(7)
            for e in self.frag_reads(0):
                e()
            self.set_koff(kt)
            in_r0 = pc[:self.g0_r0] if g == 0 else (pc if self.g1_where == "R0" else [])
            for x in in_r0:
                self.dma(x, which, buf ^ 1)()
            p.s_waitcnt(lgkmcnt=0)
            self.stamp(0)
            self.barrier()
            self.stamp(1)
            # ---------------- M0
            p.s_setprio(1)
            mf = self.mfma_list(first)
            fill = [self.dma(x, which, buf ^ 1) for x in pc] if (g == 1 and self.g1_where == "M0") else []
            for k, e in enumerate(mf):
                e()
                if k % 4 == 1 and k // 4 < len(fill):
                    fill[k // 4]()
            if not mf:
                for e in fill:
                    e()
            p.s_setprio(0)
            self.stamp(2)
            self.barrier()
            self.stamp(3)
            # ---------------- R1
            p.s_nop(7)
            for e in self.frag_reads(1):
                e()
            if g == 0:
                for x in pc[self.g0_r0:]:
                    self.dma(x, which, buf ^ 1)()
            for n_, s_ in enumerate(stores):
                self.store(s_, s_dP, v_t[n_ & 1])
            if t == 2:
                # descriptors of the tile after this one (scalar work, hidden in a read interval)
                p.s_add_u32(s_t[8], s_idx, I(self.stride))
                self.make_desc(s_t[8], 1)
                p.s_cmp("lt", "u32", s_t[8], s_end)
                p.s_cselect_b32(s_has_next, I(1), I(0))
            self.toggle_lds()
            p.s_waitcnt(lgkmcnt=0)
            self.stamp(4)
            if g == 1 and "nowait" not in self.abl:
                p.s_waitcnt(vmcnt=len(stores))
                self.stamp(5)
            self.barrier()
            self.stamp(1)
            # ---------------- M1
            p.s_setprio(1)
            for e in self.mfma_list(False):
                e()
            p.s_setprio(0)
            self.stamp(6)
            if g == 0 and "nowait" not in self.abl:
                p.s_waitcnt(vmcnt=len(stores))
                self.stamp(5)
            self.barrier()
            self.stamp(3)

    def tile_body_1bar(self, g, entry_label=None):
        """The same tile with ONE barrier per K-tile.  The trailing group runs one k-step behind INSIDE the K-tile instead of one barrier behind:
            leading:   | R0(t) M0(t) R1(t) M1(t)           | barrier
            trailing:  | M1(t-1) R0(t) M0(t) R1(t)         | barrier          (its M1 of the tile's last K-tile opens the next tile's first period)
        so a SIMD's two waves still alternate MFMA and read phases, held together by the barrier once per K-tile.  The barrier is both hand-offs of the
        two-buffer ring: before it every wave has waited for its own DMAs of K-tile t+1 and has finished its reads of K-tile t; behind it K-tile t+1 is read
        and the DMAs of K-tile t+2 go out (read phases only; the trailing group's R1 is its last phase before the barrier, so it carries stores only)."""
        p = self.p
        NKT = self.NKT
        pc = self.pieces(g)
        for t in range(NKT):
            buf = t & 1
            first = t == 0
            stores = self.store_plan.get(t, [])
            kt, which = (t + 1, 0) if t + 1 < NKT else (0, 1)
            p.comment(f"---- group {g} K-tile {t} (one barrier per K-tile)")
            if g == 1:
                # M1 of the previous K-tile (for t == 0: of the previous TILE; the very first period of the kernel enters behind it)
                p.s_setprio(1)
                for e in self.mfma_list(False):
                    e()
                p.s_setprio(0)
                if first and entry_label is not None:
                    p.label(entry_label)
            if first:
                self.seam()
            for e in self.frag_reads(0):
                e()
            self.set_koff(kt)
            for x in (pc[:self.g0_r0] if g == 0 else pc):
                self.dma(x, which, buf ^ 1)()
            p.s_waitcnt(lgkmcnt=0)
            p.s_setprio(1)
            for e in self.mfma_list(first):
                e()
            p.s_setprio(0)
            for e in self.frag_reads(1):
                e()
            if g == 0:
                for x in pc[self.g0_r0:]:
                    self.dma(x, which, buf ^ 1)()
            for n_, s_ in enumerate(stores):
                self.store(s_, s_dP, v_t[n_ & 1])
            if t == 2:
                p.s_add_u32(s_t[8], s_idx, I(self.stride))
                self.make_desc(s_t[8], 1)
                p.s_cmp("lt", "u32", s_t[8], s_end)
                p.s_cselect_b32(s_has_next, I(1), I(0))
            self.toggle_lds()
            p.s_waitcnt(lgkmcnt=0)
            if g == 0:
                p.s_setprio(1)
                for e in self.mfma_list(False):
                    e()
                p.s_setprio(0)
            if "nowait" not in self.abl:
                p.s_waitcnt(vmcnt=len(stores))
            self.barrier()

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
        if self.stamps:
            for k in range(NSTAMP):
                p.s_mov_b32(self.s_acc[k], I(0))
        # ---- lane constants
        p.v_and_b32(l15, I(15), lane)
        p.v_lshrrev_b32(q, I(4), lane)
        p.v_lshrrev_b32(r8, I(3), lane)
        p.v_and_b32(c8, I(7), lane)
        # DMA pieces: piece pp (0..31) of an operand = rows 8 pp .. 8 pp + 7 of its K-tile; this wave owns pp0 + 4 k: pp0 = w (leading group, k < nk[0]) or
        # 4 nk[0] + (w & 3) (trailing group); source offset of a lane: row 8 pp + r8, 16-byte chunk c8 ^ r8 (the LDS image is the swizzled one)
        p.s_and_b32(s_pp0, s_w, I(3))
        p.s_mul_i32(s_t[4], s_wm, I(4 * self.nk[0]))
        p.s_add_u32(s_pp0, s_pp0, s_t[4])
        p.s_lshl_b32(s_m0base, s_pp0, 10)
        p.v_xor_b32(x0, c8, r8)
        p.v_lshlrev_b32(x0, I(4), x0)
        p.s_lshl_b32(s_t[4], s_pp0, 3)
        p.v_add_u32(x1, s_t[4], r8)
        for x in range(5):
            p.v_add_u32(v_t[0], I(32 * x), x1)
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
        # scalars: row-block offsets of the stores
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
        # K-tile 0 of the first tile: every wave its own pieces
        lab_g1_loop, lab_g0_loop, lab_tail = (p.newlabel(n) for n in ("g1_loop", "g0_loop", "tail"))
        lab_g0 = p.newlabel("g0")
        self.set_koff(0)
        p.s_cmp("eq", "u32", s_wm, I(0))
        p.s_cbranch("scc1", lab_g0)
        # ================= trailing group
        for x in self.pieces(1):
            self.dma(x, 1, 0)()
        p.s_waitcnt(vmcnt=0)
        p.s_barrier()
        if self.sync == 1:
            lab_entry = p.newlabel("g1_entry")
            p.s_branch(lab_entry)                    # the first period has no M1 of a previous K-tile in front of it
            p.label(lab_g1_loop)
            self.tile_body_1bar(1, entry_label=lab_entry)
        else:
            p.s_barrier()                            # one interval behind
            p.label(lab_g1_loop)
            self.tile_body(1)
        p.s_add_u32(s_idx, s_idx, I(self.stride))
        p.s_cmp("eq", "u32", s_has_next, I(1))
        p.s_cbranch("scc1", lab_g1_loop)
        if self.sync == 1:                           # M1 of the last K-tile of the last tile
            p.s_setprio(1)
            for e in self.mfma_list(False):
                e()
            p.s_setprio(0)
        p.s_branch(lab_tail)
        # ================= leading group
        p.label(lab_g0)
        for x in self.pieces(0):
            self.dma(x, 1, 0)()
        p.s_waitcnt(vmcnt=0)
        p.s_barrier()
        p.label(lab_g0_loop)
        if self.sync == 1:
            self.tile_body_1bar(0)
        else:
            self.tile_body(0)
        p.s_add_u32(s_idx, s_idx, I(self.stride))
        p.s_cmp("eq", "u32", s_has_next, I(1))
        p.s_cbranch("scc1", lab_g0_loop)
        if self.sync != 1:
            p.s_barrier()                            # the trailing group's extra interval
        # ================= last tile: convert and store directly
        p.label(lab_tail)
        p.s_nop(7)
        p.s_nop(3)
        self.convert()
        p.s_nop(1)
        for s in range(16):
            self.store(s, s_dO[0], v_t[s & 1])
        if self.stamps:
            # waves 0 and 4 of every workgroup: dbg[(wg * 2 + wm) * 8 + k] = accumulator k
            lab_nodump = p.newlabel("nodump")
            p.s_and_b32(s_t[0], s_w, I(3))
            p.s_cmp("eq", "u32", s_t[0], I(0))
            p.s_cbranch("scc0", lab_nodump)
            p.s_lshl_b32(s_t[0], s_wg, 1)
            p.s_add_u32(s_t[0], s_t[0], s_wm)
            p.s_lshl_b32(s_t[0], s_t[0], 5)
            p.s_mov_b32(s_dP[0], s_dbg[0])
            p.s_and_b32(s_dP[1], s_dbg[1], I(0xFFFF))
            p.s_mov_b32(s_dP[2], I(1 << 20))
            p.v_mov_b32(v_t[0], s_t[0])
            for k in range(NSTAMP):
                p.v_mov_b32(v_t[1], self.s_acc[k])
                p.s_nop(1)
                p.buffer_store_dword(v_t[1], v_t[0], s_dP, I(0), 4 * k)
                p.s_nop(1)
            p.label(lab_nodump)
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


# the forms built into the laboratory library (variants/gemm_asm_lab.hip: `form`); 0 is the kernel, the rest are timing-only ablations (WRONG results)
# and placement experiments
VARIANTS = [
    ("the kernel (parked stores non-temporal)", {}),
    ("ablation: no parked stores (output not written)", {"abl": ["nostore"]}),
    ("ablation: no LDS-DMA (operands not loaded)", {"abl": ["nodma"]}),
    ("ablation: no fragment reads", {"abl": ["nolds"]}),
    ("ablation: no MFMAs", {"abl": ["nomfma"]}),
    ("ablation: no barriers inside the K loop", {"abl": ["nobar"]}),
    ("ablation: no seam conversion", {"abl": ["noconv"]}),
    ("ablation: no activation DMA", {"abl": ["noA"]}),
    ("ablation: no weight DMA", {"abl": ["noB"]}),
    ("diagnostic: cycle stamps per interval kind (dbg[(wg * 2 + group) * 8 + k]; needs the dbg buffer)", {"stamps": True}),
    ("ablation: DMAs issued, never waited for (racy)", {"abl": ["nowait"]}),
    ("placement: leading group R0 7 / R1 3 pieces", {"g0_r0": 7}),
    ("placement: leading group R0 3 / R1 7 pieces", {"g0_r0": 3}),
    ("placement: trailing group's pieces between the MFMAs of M0 instead of R0", {"g1_where": "M0"}),
    ("placement: 8 + 8 pieces per wave (leading R0 4 / R1 4, trailing R0 8)", {"nk": (4, 4), "g0_r0": 4}),
    ("placement: 4 parked stores per K-tile (K-tiles 1..4)", {"stores_per_kt": 4}),
    ("policy: parked stores with the default policy (write-back, kept in L2)", {"pol_st": ""}),
    ("policy: activation DMAs nt", {"pol_a": "nt"}),
    ("policy: weight DMAs nt", {"pol_b": "nt"}),
    ("policy: parked stores sc1", {"pol_st": "sc1"}),
    ("placement: two parked stores per K-tile in K-tiles 1-8 instead of the even spread", {"stores_per_kt": 2}),
    ("synchronisation: ONE barrier per K-tile, groups skewed by a k-step", {"sync": 1}),
    ("synchronisation: one barrier per K-tile + leading group R0 6 / R1 4", {"sync": 1, "g0_r0": 6}),
    ("synchronisation: one barrier per K-tile + 8 + 8 pieces per wave", {"sync": 1, "nk": (4, 4), "g0_r0": 4}),
    ("ablation: LDS-DMA + barriers + fragment reads only (no MFMAs, no stores)", {"abl": ["nomfma", "nostore"]}),
    ("ablation: LDS-DMA + barriers only (no MFMAs, no stores, no fragment reads, no seam conversion)", {"abl": ["nomfma", "nostore", "nolds", "noconv"]}),
    ("ablation: MFMAs + fragment reads + barriers only (no DMA, no stores)", {"abl": ["nodma", "nostore"]}),
]


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
