#!/usr/bin/env python3
"""Fixed vs per-k cost of the GEMM kernel: time(M, N, K) for K in {32, 320, 640, 1280} per tile config.
The K -> 0 intercept is prologue + epilogue + launch; the slope is the main loop.  (This sweep found the
epilogue at 35 % of the K = 320 GEMMs and the scratch spills of the 256-row tiles)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from pbe_amd import ops  # noqa: E402
from bench_kernels import rnd, timeit  # noqa: E402

dev = torch.device("cuda:0")
for (M, N) in ((32768, 2560), (32768, 320), (8192, 5120), (2048, 1280)):
    for K in (32, 320, 640, 1280):
        a, w = rnd(M, K), rnd(N, K)
        bias = torch.randn(N, device=dev)
        row = f"M={M:6d} N={N:5d} K={K:5d}:"
        for cfg in (8, 3, 0, 7):
            ops.tune(1, cfg)
            t = min(timeit(lambda: ops.gemm(a, w, bias)) for _ in range(2))
            row += f" cfg{cfg} {t:7.1f} us ({2.0 * M * N * K / t / 1e6:6.1f} TF) |"
        print(row, flush=True)
ops.tune(1, -1)
