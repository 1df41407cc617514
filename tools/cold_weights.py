#!/usr/bin/env python3
"""Does a conv / GEMM of the path run slower when its weights come from HBM (as in the real sampler: 1.7 GB of weights per U-Net
call, each layer's last touched one call ago) than in a back-to-back microbenchmark (everything resident in the 256 MB Infinity Cache)?

Per shape: (hot) same operands every launch; (cold W) weights rotate over enough copies to exceed 600 MB; (cold W + X) both rotate.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pbe_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).half()


def timeit(calls, iters):
    n = len(calls)
    for i in range(min(n, 4)):
        calls[i]()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        calls[i % n]()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    shapes = [("c", 8, 64, 320, 0, 320), ("c", 8, 32, 640, 0, 640), ("c", 8, 16, 1280, 0, 1280), ("c", 8, 8, 1280, 0, 1280), ("c", 8, 16, 1280, 1280, 1280),
              ("g", 32768, 320, 320), ("g", 8192, 640, 640), ("g", 2048, 1280, 1280), ("g", 2048, 10240, 1280), ("g", 2048, 1280, 5120), ("g", 8192, 5120, 640)]
    for sh in shapes:
        if sh[0] == "c":
            _, B, H, c1, c2, co = sh
            wbytes = co * 9 * (c1 + c2) * 2
            xbytes = B * H * H * (c1 + c2) * 2
            fl = 2.0 * B * H * H * co * 9 * (c1 + c2)
            nW = max(1, min(400, int(700e6 / wbytes) + 1))
            nX = max(1, min(64, int(700e6 / xbytes) + 1))
            Ws = [rnd(co, 9 * (c1 + c2)) for _ in range(nW)]
            Xs = [(rnd(B, H, H, c1), rnd(B, H, H, c2) if c2 else None) for _ in range(nX)]
            bias = torch.randn(co, device=dev)
            mk = lambda w, x: (lambda: ops.conv3x3(x[0], w, bias, x2=x[1]))
        else:
            _, M, N, K = sh
            wbytes, xbytes, fl = N * K * 2, M * K * 2, 2.0 * M * N * K
            nW = max(1, min(400, int(700e6 / wbytes) + 1))
            nX = max(1, min(64, int(700e6 / xbytes) + 1))
            Ws = [rnd(N, K) for _ in range(nW)]
            Xs = [rnd(M, K) for _ in range(nX)]
            bias = torch.randn(N, device=dev)
            mk = lambda w, x: (lambda: ops.gemm(x, w, bias))
        iters = max(40, nW)
        hot = timeit([mk(Ws[0], Xs[0])], iters)
        coldw = timeit([mk(w, Xs[0]) for w in Ws], iters)
        both = timeit([mk(Ws[i % nW], Xs[i % nX]) for i in range(max(nW, nX))], iters)
        print(f"{str(sh):40s} W {wbytes / 1e6:6.1f} MB x{nW:3d}  hot {hot:7.1f} us {fl / hot / 1e6:6.0f} TF | cold W {coldw:7.1f} us {fl / coldw / 1e6:6.0f} TF | cold W+X {both:7.1f} us {fl / both / 1e6:6.0f} TF",
              flush=True)
        del Ws, Xs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
