#!/usr/bin/env python3
"""Same-process A/B of ops.USE_CONV_GROUP_STATS (GroupNorm statistics from the producing conv's copy-out) on the U-Net forward at the
bench's batch (4 images = 8 U-Net rows, paired guidance prefix).  python tools/gstats_ab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pbe_amd import ops  # noqa: E402
from bench import build_model  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    model, _, _ = build_model(dev, 0, 1)
    unet = model.model.diffusion_model
    B = 4
    g = torch.Generator().manual_seed(1)
    x9 = ops.plms_pack_input(torch.randn(B, 4, 64, 64, generator=g).to(dev), torch.randn(B, 4, 64, 64, generator=g).to(dev), torch.ones(B, 1, 64, 64).to(dev), 1)
    ctx = torch.randn(2 * B, 1, 768, generator=g).half().to(dev)
    t = torch.full((2 * B,), 501, dtype=torch.int64, device=dev)

    def run(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            out = unet.forward_nhwc(x9, t, ctx, paired=True, step=501)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out

    outs = {}
    for flag in (True, False):
        ops.USE_CONV_GROUP_STATS = flag; run(3)
    for rep in range(4):
        for flag in (True, False):
            ops.USE_CONV_GROUP_STATS = flag
            ms, outs[flag] = run(20)
            print(f"rep {rep} conv group statistics={flag}: {ms:.3f} ms / U-Net call", flush=True)
    d = (outs[True].float() - outs[False].float()).abs().max().item()
    print(f"max |eps difference| between the two statistics paths: {d:.3e} (max |eps| {outs[False].float().abs().max().item():.3f})")


if __name__ == "__main__":
    main()
