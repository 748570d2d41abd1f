#!/usr/bin/env python3
"""What the vendor library reaches on the encoder's GEMM shapes (torch.matmul -> hipBLASLt / rocBLAS, f16 in, f32 accumulate, plain product, no epilogue): a yardstick
beside k_gemm16w's numbers in DESIGN.md section 3, not a product path.  usage: lib_gemm_reference.py"""
import torch

M = 96000      # 64 clips x 1500 frames
dev = torch.device("cuda", 0)
for name, K, N in (("Q/K/V/O", 768, 768), ("FC1", 768, 3072), ("FC2", 3072, 768)):
    a = torch.randn(M, K, device=dev, dtype=torch.float16); w = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000.0 / reps
    tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
    print("%-8s M=%d N=%d K=%d: %.1f us per product, %.0f TF/s = %.2f of 2.5 PF/s (f16 output, no bias / GELU / residual)" % (name, M, N, K, us, tf, tf / 2500.0))
