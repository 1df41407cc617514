#!/usr/bin/env python3
"""A/B of the attention kernel's softmax forms on the U-Net's shapes, variants interleaved in ONE process (cdna guide rule 24).
    python tools/attn_ab.py            -> d = 40 padded-reference form (pbe_tune 6) on / off, queries per wave 1 / 2"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pbe_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=30, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    B, H = 8, 8
    for (N, D) in ((4096, 40), (1024, 80), (256, 160)):
        Cc = H * D
        qk = (torch.randn(B * N, 2 * Cc, device=dev) * 1.0).half()
        vt = (torch.randn(B, Cc, N, device=dev) * 1.0).half()
        fl = 4.0 * B * H * N * N * D
        call = lambda: ops.attention(qk, qk[:, Cc:], vt, B, H, N, N, D, D ** -0.5, q_strides=(N * 2 * Cc, 2 * Cc), k_strides=(N * 2 * Cc, 2 * Cc),
                                     vt_strides=(Cc * N, N))
        res = {}
        for rep in range(4):
            for mpad in ((0, 1) if D == 40 else (1,)):
                for qw in ((1, 2, 0) if D == 40 else (1, 2)):           # 0 = the dispatcher's choice (two K/V tiles per barrier at d = 40)
                    ops.tune(6, mpad); ops.tune(3, qw)
                    res.setdefault((mpad, qw), []).append(timeit(call))
        ops.tune(6, 1); ops.tune(3, 0)
        for kk, v in sorted(res.items()):
            v = sorted(v)
            print(f"attn N={N} D={D} mpad={kk[0]} qw={kk[1]}: min {v[0]:7.1f} us  median {v[len(v) // 2]:7.1f} us  {fl / v[0] / 1e6:6.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
