#!/usr/bin/env python3
"""Dense tiles on the plain token GEMMs (ff-out / out-projection with the fused residual), cold operands, one process.
python tools/dense_tile_ab.py [cfgs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pbe_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
CFGS = tuple(int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "-1,9,8,21").split(","))


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
    base = timeit(lambda: flush.zero_())
    for (M, N, K) in ((32768, 320, 1280), (8192, 640, 2560), (2048, 1280, 5120), (32768, 320, 320), (8192, 640, 640), (32768, 320, 640)):
        a, w, b = (torch.randn(M, K, device=dev) * 0.5).half(), (torch.randn(N, K, device=dev) * K ** -0.5).half(), torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev).half()
        res, outs = {}, {}
        for rep in range(3):
            for cfg in CFGS:
                ops._FORCE_CFG = None if cfg < 0 else cfg
                def call():
                    flush.zero_()
                    return ops.gemm(a, w, b, resid=r)
                outs[cfg] = ops.gemm(a, w, b, resid=r)
                res.setdefault(cfg, []).append(timeit(call) - base)
        ops._FORCE_CFG = None
        line = "  ".join(f"cfg{c:>2d} {min(res[c]):6.1f} us" for c in CFGS)
        same = all(torch.equal(outs[CFGS[0]], outs[c]) for c in CFGS[1:])
        print(f"g:{M}:{N}:{K} + resid (cfg -1 = tuned table): {line}   ({2.0 * M * N * K / min(min(v) for v in res.values()) * 1e-6:.0f} TFLOP/s best; outputs identical across tiles: {same})", flush=True)


if __name__ == "__main__":
    main()
