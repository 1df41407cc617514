#!/usr/bin/env python3
"""The transformer block's GEMMs, extended epilogue against the round-2 forms, interleaved in ONE process:
GEGLU with / without the folded LayerNorm (+ the LayerNorm launch it replaces), q|k|v^T fused against q|k + V^T, out-projection with /
without row statistics.  python tools/gemm_ex_ab.py [--cfgs 8,9,3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pbe_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
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


def rnd(*shape, s=0.5):
    return (torch.randn(*shape, device=dev) * s).half()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", default="9,8,3")
    a = ap.parse_args()
    cfgs = [int(c) for c in a.cfgs.split(",")]
    flush = torch.empty(300 << 20, dtype=torch.uint8, device=dev)          # > the 256 MiB Infinity Cache: operands come from HBM like in the sampler
    for (M, C) in ((32768, 320), (8192, 640), (2048, 1280)):
        x = rnd(M, C, s=1.0)
        wg, bg = rnd(8 * C, C, s=C ** -0.5), torch.randn(8 * C, device=dev)
        gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        c1 = wg.float().sum(1).contiguous()
        st = ops.row_stats(x)
        res = {}
        for rep in range(3):
            for cfg in cfgs:
                ops._FORCE_CFG = cfg
                def plain():
                    flush.zero_()
                    return ops.gemm(x, wg, bg, act=ops.ACT_GEGLU)
                def ex_plain():                          # the EX instantiation without the fold (alpha_cols = N, alpha = 1: arithmetic unchanged)
                    flush.zero_()
                    return ops.gemm(x, wg, bg, act=ops.ACT_GEGLU, alpha_cols=8 * C)
                def folded():
                    flush.zero_()
                    return ops.gemm(x, wg, bg, act=ops.ACT_GEGLU, ln=(st, c1, 1e-5))
                def lnorm():
                    flush.zero_()
                    return ops.layernorm(x, gam, bet)
                def flush_only():
                    flush.zero_()
                base = timeit(flush_only)
                for name, fn in (("geglu plain", plain), ("geglu EX no fold", ex_plain), ("geglu LN-folded", folded), ("layernorm", lnorm)):
                    res.setdefault((name, cfg), []).append(timeit(fn) - base)
        ops._FORCE_CFG = None
        for k in sorted(res):
            print(f"M={M} C={C} {k[0]:18s} cfg{k[1]:2d}: min {min(res[k]):7.1f} us (cold operands)", flush=True)


if __name__ == "__main__":
    main()
