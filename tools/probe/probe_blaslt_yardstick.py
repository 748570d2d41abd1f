"""Yardstick only (never on the product path): what the vendor library's f16 GEMM reaches on the encoder's product shapes, to size the gap of k_gemm16w.
Usage (GPU box): python tools/probe/probe_blaslt_yardstick.py"""
import torch

M = 96000
for name, N, K in [("Q/K/V/O 768x768", 768, 768), ("QKV fused 2304x768", 2304, 768), ("FC1 3072x768", 3072, 768), ("FC2 768x3072", 768, 3072)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.float16); w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.05
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print("%-22s %8.1f us  %7.1f TFLOP/s" % (name, us, 2.0 * M * N * K / us / 1e6), flush=True)
