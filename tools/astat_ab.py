#!/usr/bin/env python3
"""A-stationary persistent tiles (configs 19 / 20, igemm_astat.hip) against the streaming tiles on the K = 320 LayerNorm-folded GEGLU projection,
interleaved in ONE process, operands cold (a 300 MB fill between launches) and warm.  python tools/astat_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pbe_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
CFGS = tuple(int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "9,19,20").split(","))


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
    C = 320
    for M in (32768, 16384, 4096):
        x = (torch.randn(M, C, device=dev)).half()
        wg, bg = (torch.randn(8 * C, C, device=dev) * C ** -0.5).half(), torch.randn(8 * C, device=dev)
        c1 = wg.float().sum(1).contiguous()
        st = ops.row_stats(x)
        outs, res = {}, {}
        for rep in range(3):
            for cfg in CFGS:
                ops._FORCE_CFG = cfg
                def cold():
                    flush.zero_()
                    return ops.gemm(x, wg, bg, act=ops.ACT_GEGLU, ln=(st, c1, 1e-5))
                def warm():
                    return ops.gemm(x, wg, bg, act=ops.ACT_GEGLU, ln=(st, c1, 1e-5))
                def flush_only():
                    flush.zero_()
                outs[cfg] = warm()
                base = timeit(flush_only)
                res.setdefault(("cold", cfg), []).append(timeit(cold) - base)
                res.setdefault(("warm", cfg), []).append(timeit(warm))
        ops._FORCE_CFG = None
        for k in sorted(res):
            fl = 2.0 * M * 8 * C * C
            print(f"M={M} K=320 N=2560 {k[0]} cfg{k[1]:2d}: min {min(res[k]):7.1f} us  {fl / min(res[k]) * 1e-6:7.1f} TFLOP/s", flush=True)
        for c in CFGS[1:]:
            print(f"M={M}: tile {c} output bit-identical to tile {CFGS[0]}: {torch.equal(outs[CFGS[0]], outs[c])} (max |d| {(outs[CFGS[0]].float() - outs[c].float()).abs().max().item():.3e})", flush=True)


def qkv():
    """The fused q | k | v^T projection (K = 320, N = 960: tile 20 against the streaming tiles 9 / 8)."""
    flush = torch.empty(300 << 20, dtype=torch.uint8, device=dev)
    C = 320
    for B in (8, 4):
        N, M = 4096, B * 4096
        x = (torch.randn(M, C, device=dev)).half()
        w, b = (torch.randn(3 * C, C, device=dev) * C ** -0.5).half(), torch.randn(3 * C, device=dev) * 0.1
        c1 = w.float().sum(1).contiguous()
        st = ops.row_stats(x)
        res, outs = {}, {}
        for rep in range(3):
            for cfg in (9, 8, 20):
                ops._FORCE_CFG = cfg
                qk = torch.empty(M, 2 * C, dtype=torch.float16, device=dev)
                vt = torch.empty(B, C, N, dtype=torch.float16, device=dev)
                def call():
                    ops.gemm(x, w, b, ln=(st, c1, 1e-5), alpha=0.2281, alpha_cols=C, out=qk, vt=vt, vt_col0=2 * C, vt_tokens=N)
                def cold():
                    flush.zero_(); call()
                def flush_only():
                    flush.zero_()
                call()
                outs[cfg] = (qk.clone(), vt.clone())
                base = timeit(flush_only)
                res.setdefault(("cold", cfg), []).append(timeit(cold) - base)
                res.setdefault(("warm", cfg), []).append(timeit(call))
        ops._FORCE_CFG = None
        for k in sorted(res):
            print(f"qkv M={M} K=320 N=960 {k[0]} cfg{k[1]:2d}: min {min(res[k]):7.1f} us  {2.0 * M * 960 * 320 / min(res[k]) * 1e-6:7.1f} TFLOP/s", flush=True)
        print(f"qkv M={M}: tile 20 bit-identical to tile 9: q|k {torch.equal(outs[9][0], outs[20][0])}, v^T {torch.equal(outs[9][1], outs[20][1])}", flush=True)


if __name__ == "__main__":
    main()
    qkv()
