#!/usr/bin/env python3
"""Fixed vs per-k cost of the 3x3-conv implicit GEMM: time vs Cin at fixed output size, per tile config."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from pbe_amd import ops  # noqa: E402
from bench_kernels import rnd, timeit  # noqa: E402

dev = torch.device("cuda:0")
B = 8
for (H, Co) in ((64, 320), (32, 640), (16, 1280)):
    for Ci in (32, 64, 160, 320, 640):
        x = rnd(B, H, H, Ci)
        w = ops.pack_conv3x3(torch.randn(Co, Ci, 3, 3, device=dev) * 0.05)
        bias = torch.randn(Co, device=dev)
        emb = rnd(B, Co)
        row = f"conv {H}^2 Cin={Ci:4d} Cout={Co:5d}:"
        for cfg in (8, 7, 0, 3):
            ops.tune(1, cfg)
            t = min(timeit(lambda: ops.conv3x3(x, w, bias, rowvec=emb)) for _ in range(2))
            row += f" cfg{cfg} {t:7.1f} us ({2.0 * B * H * H * Co * 9 * Ci / t / 1e6:6.1f} TF) |"
        print(row, flush=True)
ops.tune(1, -1)
