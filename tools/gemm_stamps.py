#!/usr/bin/env python3
"""Diagnostic (not the product library): persistent GEMM built with -DUCOD_GEMM_STAMPS; prints where a tile's cycles go."""
import ctypes as C, os, subprocess, sys, tempfile
import torch
_csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ucod_dpl_amd", "csrc")
_so = os.path.join(tempfile.gettempdir(), "libucod_dpl_stamps.so")       # built on the spot, never shipped or loaded by the package
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-DUCOD_GEMM_STAMPS",
                       "-shared", "-o", _so, os.path.join(_csrc, "gemm_bf16.hip"), os.path.join(_csrc, "prof.hip")])
lib = C.CDLL(_so)
vp, ci = C.c_void_p, C.c_int
lib.ucod_gemm_bf16.argtypes = [ci, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, ci, ci, vp]
M = 32 * 1370
for grid in (256, 128, 64, 16):
  os.environ["UCOD_PERS_GRID"] = str(grid)
  print("persistent grid", grid)
  for name, Nn, K, epi, v in (("qkv", 2304, 768, 0, 7), ("fc1", 3072, 768, 1, 7), ("proj", 768, 768, 2, 8), ("fc2", 768, 3072, 2, 8)):
      A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(Nn, K, device="cuda") * 0.05).to(torch.bfloat16)
      b = torch.randn(Nn, device="cuda"); sc = torch.ones(Nn, device="cuda"); resid = torch.randn(M, Nn, device="cuda")
      out = torch.empty(M, Nn, device="cuda", dtype=torch.float32 if epi == 2 else torch.bfloat16)
      st = torch.zeros(256 * 4, dtype=torch.int64, device="cuda")
      for _ in range(3):
          rc = lib.ucod_gemm_bf16(epi, A.data_ptr(), W.data_ptr(), out.data_ptr(), M, Nn, K, b.data_ptr(), sc.data_ptr() if epi == 2 else None,
                                  resid.data_ptr() if epi == 2 else None, st.data_ptr(), 1370, v, None)
          assert rc == 0, rc
      torch.cuda.synchronize()
      s = st.view(256, 4).double().cpu()[:grid]
      tiles = s[:, 3].sum().item()
      print(f"{name}: tiles/WG {s[:,3].mean():.2f}  cycles per tile: main loop {s[:,0].sum()/tiles:.0f}  epilogue(+next-tile DMA issue) {s[:,1].sum()/tiles:.0f}  final vmcnt(0) {s[:,2].sum()/tiles:.0f}  (s_memtime ticks = shader clocks)")
