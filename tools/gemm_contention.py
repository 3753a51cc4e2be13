#!/usr/bin/env python3
"""How much of a K = 768 tile's time is chip-level contention?  The product GEMM (QKV epilogue: bias -> bf16, N = 2304, K = 768) at row
counts that give 9 ... 2 x 252 workgroups: if one round of 252 workgroups takes much longer than one round of 9 or 36, the drain / operand
fetch of a round is bound by something the workgroups SHARE (HBM write bandwidth, fabric) rather than by the CU itself, and de-phasing the
rounds would pay.  Also the same with the output kept tiny (N = 256 columns) for the fetch side alone."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ucod_dpl_amd import native as N  # noqa: E402

fn = N.load().ucod_gemm_bf16
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
K = 768
print("# rows, cols, workgroups (256 x 256 tiles), us per launch (median of 7 x 50), us per round")
for Nn in (2304, 3072, 768):
    epi = N.EPI_BIAS_BF16
    W = (torch.randn(Nn, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(Nn, device="cuda", generator=g)
    for rows in (256, 512, 1024, 2048, 4096, 7168, 14336, 28672, 43840):
        A = torch.randn(rows, K, device="cuda", generator=g).to(torch.bfloat16)
        out = torch.empty(rows, Nn, dtype=torch.bfloat16, device="cuda")
        ts = []
        for r in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                rc = fn(epi, A.data_ptr(), W.data_ptr(), out.data_ptr(), rows, Nn, K, bias.data_ptr(), None, None, None, 1370, 0, st)
            e1.record()
            torch.cuda.synchronize()
            assert rc == 0
            ts.append(e0.elapsed_time(e1) / 50 * 1e3)
        wg = ((rows + 255) // 256) * (Nn // 256)
        rounds = (wg + 255) // 256
        med = statistics.median(ts)
        print(f"{rows:6d} x {Nn:4d}: {wg:5d} workgroups, {rounds} round(s): {med:7.1f} us  = {med / rounds:6.1f} us per round   ({2.0 * rows * Nn * K / (med * 1e-6) / 2.5e15:.3f} of 2.5 PF)")
