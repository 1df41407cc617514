#!/usr/bin/env python3
"""What the conv pays for GroupNorm statistics in its copy-out and what the norm saves (operands cold).  python tools/gstats_micro.py"""
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


def main():
    flush = torch.empty(300 << 20, dtype=torch.uint8, device=dev)
    for (B, H, Cin, Cout, resid) in ((8, 64, 320, 320, True), (8, 64, 320, 320, False), (8, 32, 640, 640, True)):
        x = (torch.randn(B, H, H, Cin, device=dev) * 0.7).half()
        wp = ops.pack_conv3x3(torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)).to(dev)
        bias = torch.randn(Cout, device=dev) * 0.1
        r = (torch.randn(B, H, H, Cout, device=dev)).half() if resid else None
        gam, bet = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        base = timeit(lambda: flush.zero_())
        res = {}
        for rep in range(3):
            for gs in (0, 32):
                def conv():
                    flush.zero_()
                    return ops.conv3x3(x, wp, bias, resid=r, group_stats=gs)
                y = conv()
                def norm():
                    flush.zero_()
                    return ops.groupnorm(y, gam, bet, 1e-5, True)
                res.setdefault(("conv", gs), []).append(timeit(conv) - base)
                res.setdefault(("norm", gs), []).append(timeit(norm) - base)
        for k in sorted(res):
            print(f"B={B} {H}x{H} {Cin}->{Cout} resid={resid} {k[0]} group_stats={k[1]:2d}: min {min(res[k]):7.1f} us", flush=True)
        y = ops.conv3x3(x, wp, bias, resid=r, group_stats=32)
        def norm_hot():
            return ops.groupnorm(y, gam, bet, 1e-5, True)
        def norm_cold():
            flush.zero_()
            return ops.groupnorm(y, gam, bet, 1e-5, True)
        for rows in (1, 2, 4, 8, 16):
            ops.tune(9, rows)
            print(f"   normalisation pass, {rows:2d} rows per thread: cold {timeit(norm_cold) - base:6.1f} us, input resident {timeit(norm_hot):6.1f} us", flush=True)
        ops.tune(9, 4)
        y0 = ops.conv3x3(x, wp, bias, resid=r)
        def norm2_cold():
            flush.zero_()
            return ops.groupnorm(y0, gam, bet, 1e-5, True)
        for rows in (0, 2, 4, 8):
            ops.tune(10, rows)
            print(f"   two-pass GroupNorm, normalisation pass with {rows:2d} rows per thread (0 = as the statistics pass): cold {timeit(norm2_cold) - base:6.1f} us, resident {timeit(lambda: ops.groupnorm(y0, gam, bet, 1e-5, True)):6.1f} us", flush=True)
        ops.tune(10, 8)


if __name__ == "__main__":
    main()
