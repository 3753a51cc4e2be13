"""Static wait-state audit of a generated program (isa.Prog).  hipcc pads none of this for hand-written code, and the hardware has no
interlock for these pairs (cdna_hip_programming.md section 5.7 item 2; LLVM GCNHazardRecognizer for gfx940/gfx950), so the generator's
output is checked in program order.  Wait states are counted per issued instruction (s_nop N = N + 1; an intervening MFMA counts as eight: the matrix pipe takes one 8-pass
instruction per 8 quad-cycles).  Rules (producer -> consumer: wait states required between them):

  R1  VALU write of a VGPR            -> MFMA reading it as A, B or C                          2
  R2  MFMA write of D                 -> any non-MFMA access of D, or an MFMA reading it as A/B 12   (C of the same range as D: 0, the accumulate chain)
  R3  transcendental write            -> VALU read                                             1
  R4  VALU write                      -> v_permlane32_swap reading it                          2
  R5  VALU write                      -> v_readfirstlane reading it                            1
  R6  SALU write of M0                -> LDS-DMA                                               1
  R7  VALU write of an SGPR / VCC     -> vector-memory instruction reading it                  5
  R8  16-byte store reading its data  -> write of those registers                              2
  R9  MFMA reading A, B or C          -> non-MFMA write of that register                       8   (the operands are read over the first passes)
The scan is linear; a label resets nothing (fall-through is the common path and out-of-line blocks open with their own s_nops).
"""
from .isa import Ins, M0_IDX, VCC_LO


def audit(prog, verbose=False):
    viol = []
    hist = []          # (wait-state time, Ins)
    t = 0
    last_write = {}    # reg id -> (time, klass, ins)
    last_mfma_read = {}    # reg id -> time
    last_mfma_write = {}   # reg id -> (time, ins)
    last_store_read = {}   # reg id -> time
    m0_write = -100
    last_mfma_issue = -100
    for it in prog.items:
        if not isinstance(it, Ins):
            continue
        ins = it
        # an s_nop N idles N + 1 wait states; an 8-pass MFMA cannot issue less than 8 quad-cycles after the previous MFMA (one matrix pipe
        # per SIMD), so everything issued BEFORE it is at least that far behind whatever follows it
        ws = (ins.mods.get("n", 0) + 1) if ins.klass == "nop" else 1
        k = ins.klass
        if k == "mfma":
            t = max(t, last_mfma_issue + (4 if ins.mods.get("passes") == 4 else 8))
            last_mfma_issue = t

        def need(reg, t_prod, gap, rule):
            if t - t_prod - 1 < gap:
                viol.append(f"{rule}: `{ins.text}` needs {gap} wait states after the producer of register {reg}, has {t - t_prod - 1}")

        if k == "mfma":
            a, b, c = ins.src
            d = ins.dst
            same_c = hasattr(c, "ids") and c.ids() == d.ids()
            for r in ins.reads:
                if r in last_write and last_write[r][1] in ("valu", "trans"):
                    need(r, last_write[r][0], 2, "R1")
            for opnd, is_c in ((a, False), (b, False), (c, True)):
                if not hasattr(opnd, "ids"):
                    continue
                for r in opnd.ids():
                    if r in last_mfma_write:
                        tw, wins = last_mfma_write[r]
                        if is_c and opnd.ids() == wins.dst.ids():
                            continue                       # accumulate chain
                        need(r, tw, 12, "R2")
            for r in ins.reads:
                last_mfma_read[r] = t
            for r in ins.writes:
                last_mfma_write[r] = (t, ins)
                last_write[r] = (t, "mfma", ins)
        else:
            touched = set(ins.reads) | set(ins.writes)
            for r in touched:
                if r in last_mfma_write:
                    need(r, last_mfma_write[r][0], 12, "R2")
            if k in ("valu", "trans"):
                for r in ins.reads:
                    if r in last_write and last_write[r][1] == "trans":
                        need(r, last_write[r][0], 1, "R3")
                if ins.name in ("v_permlane32_swap_b32", "v_permlane16_swap_b32"):
                    for r in ins.reads + ins.writes:
                        if r in last_write and last_write[r][1] in ("valu", "trans"):
                            need(r, last_write[r][0], 2, "R4")
                if ins.name == "v_readfirstlane_b32":
                    for r in ins.reads:
                        if r in last_write and last_write[r][1] in ("valu", "trans"):
                            need(r, last_write[r][0], 1, "R5")
            if ins.name == "buffer_load_lds_dwordx4":
                if t - m0_write - 1 < 1:
                    viol.append(f"R6: `{ins.text}` right behind the write of M0")
            if k == "vmem":
                for r in ins.reads:
                    if r >= 1000 and r in last_write and last_write[r][1] in ("valu", "trans"):
                        need(r, last_write[r][0], 5, "R7")
            for r in ins.writes:
                if r in last_store_read:
                    need(r, last_store_read[r], 2, "R8")
                if r in last_mfma_read:
                    need(r, last_mfma_read[r], 8, "R9")
            if ins.name == "buffer_store_dwordx4":
                for r in ins.src[0].ids():
                    last_store_read[r] = t
            for r in ins.writes:
                if k in ("valu", "trans", "salu", "ds", "vmem"):
                    last_write[r] = (t, k, ins)
                last_mfma_write.pop(r, None)
            if 1000 + M0_IDX in ins.writes:
                m0_write = t
        t += ws
    return viol


if __name__ == "__main__":
    import sys
    from .gen_attn import Gen
    from .gen_attn32 import Gen32
    dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    g = (Gen32 if len(sys.argv) > 2 and sys.argv[2] == "pw32" else Gen)(dtype=dt)
    v = audit(g.build())
    for x in v[:50]:
        print(x)
    print(len(v), "violations")
