#!/usr/bin/env python3
"""Run only the fused attention kernel (for rocprofv3 --pmc passes): python tools/attn_only.py [N D [qw]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pbe_amd import ops  # noqa: E402

N, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 40)
if len(sys.argv) > 3:
    ops.tune(3, int(sys.argv[3]))
B, H, Cc = 8, 8, 8 * D
dev = torch.device("cuda:0")
qk = (torch.randn(B * N, 2 * Cc, device=dev) * 0.5).half()
vt = (torch.randn(B, Cc, N, device=dev) * 0.5).half()
for _ in range(4):
    ops.attention(qk, qk[:, Cc:], vt, B, H, N, N, D, D ** -0.5, q_strides=(N * 2 * Cc, 2 * Cc), k_strides=(N * 2 * Cc, 2 * Cc), vt_strides=(Cc * N, N))
torch.cuda.synchronize()
