#!/usr/bin/env python3
"""Kernel micro-benchmarks on the U-Net's real shapes at the BASELINE batch (2B = 8).
Usage (GPU box): python tools/bench_kernels.py [conv|gemm|attn|norm|all] [--variants 1,2]
Prints one line per shape: avg us, TFLOP/s (or TB/s).  Interleaves variants in ONE process."""
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
    return a.elapsed_time(b) / iters * 1e3       # us


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).half()


CONV = [  # (H, Cin1, Cin2, Cout, stride, ups, count per U-Net forward)
    (64, 320, 0, 320, 1, False, 7), (64, 320, 320, 320, 1, False, 2), (64, 640, 320, 320, 1, False, 1), (64, 320, 0, 320, 2, False, 1),
    (32, 320, 0, 640, 1, False, 1), (32, 640, 0, 640, 1, False, 6), (32, 640, 320, 640, 1, False, 1), (32, 640, 640, 640, 1, False, 1),
    (32, 1280, 640, 640, 1, False, 1), (32, 640, 0, 640, 1, True, 1), (32, 640, 0, 640, 2, False, 1),
    (16, 640, 0, 1280, 1, False, 1), (16, 1280, 0, 1280, 1, False, 7), (16, 1280, 1280, 1280, 1, False, 2), (16, 1280, 640, 1280, 1, False, 1),
    (16, 1280, 0, 1280, 1, True, 1), (16, 1280, 0, 1280, 2, False, 1),
    (8, 1280, 0, 1280, 1, False, 11), (8, 1280, 1280, 1280, 1, False, 3), (8, 1280, 0, 1280, 1, True, 1)]

GEMM = [  # (M, N, K, count)   2B=8
    (32768, 320, 320, 10), (32768, 640, 320, 5), (32768, 2560, 320, 5), (32768, 320, 1280, 5),
    (8192, 640, 640, 10), (8192, 1280, 640, 5), (8192, 5120, 640, 5), (8192, 640, 2560, 5),
    (2048, 1280, 1280, 10), (2048, 2560, 1280, 5), (2048, 10240, 1280, 5), (2048, 1280, 5120, 5),
    (512, 1280, 1280, 2), (512, 10240, 1280, 1), (512, 1280, 5120, 1), (8, 1280, 1280, 1), (8, 15680, 1280, 1)]


def bench_conv(variants, B=8):
    tot = {v: 0.0 for v in variants}
    flops_tot = 0.0
    for (H, c1, c2, co, st, ups, cnt) in CONV:
        Hin = H                                     # H is the INPUT resolution
        x = rnd(B, Hin, Hin, c1)
        x2 = rnd(B, Hin, Hin, c2) if c2 else None
        w = rnd(co, 9 * (c1 + c2))
        bias = torch.randn(co, device=dev)
        Ho = Hin * 2 if ups else (Hin // 2 if st == 2 else Hin)
        fl = 2.0 * B * Ho * Ho * co * 9 * (c1 + c2)
        flops_tot += fl * cnt
        row = f"conv {Hin:3d}^2 {c1:4d}+{c2:4d}->{co:4d} s{st} u{int(ups)} x{cnt:2d}  GF={fl / 1e9:7.1f}"
        for v in variants:
            ops.tune(1, v)
            us = timeit(lambda: ops.conv3x3(x, w, bias, x2=x2, stride=st, pad=1, upsample=ups))
            tot[v] += us * cnt
            row += f" | cfg{v}: {us:8.1f} us {fl / us / 1e6:7.1f} TF"
        print(row, flush=True)
    for v in variants:
        print(f"conv total per U-Net forward v{v}: {tot[v] / 1e3:.2f} ms  -> {flops_tot / tot[v] / 1e6:.1f} TFLOP/s")


def bench_gemm(variants):
    tot = {v: 0.0 for v in variants}
    flops_tot = 0.0
    for (M, N, K, cnt) in GEMM:
        a, w, bias = rnd(M, K), rnd(N, K), torch.randn(N, device=dev)
        fl = 2.0 * M * N * K
        flops_tot += fl * cnt
        row = f"gemm M={M:6d} N={N:6d} K={K:5d} x{cnt:2d} GF={fl / 1e9:7.1f}"
        for v in variants:
            ops.tune(1, v)
            us = timeit(lambda: ops.gemm(a, w, bias))
            tot[v] += us * cnt
            row += f" | cfg{v}: {us:8.1f} us {fl / us / 1e6:7.1f} TF"
        print(row, flush=True)
    for v in variants:
        print(f"gemm total per U-Net forward v{v}: {tot[v] / 1e3:.2f} ms  -> {flops_tot / tot[v] / 1e6:.1f} TFLOP/s")


def bench_attn(B=8):
    for (N, D, cnt) in ((4096, 40, 5), (1024, 80, 5), (256, 160, 5), (64, 160, 1)):
        H, Cc = 8, 8 * D
        qk, vt = rnd(B * N, 2 * Cc), rnd(B, Cc, N)
        fl = 4.0 * B * H * N * N * D
        row = f"attn N={N:5d} D={D:3d} x{cnt}:"
        call = lambda: ops.attention(qk, qk[:, Cc:], vt, B, H, N, N, D, D ** -0.5, q_strides=(N * 2 * Cc, 2 * Cc), k_strides=(N * 2 * Cc, 2 * Cc),
                                     vt_strides=(Cc * N, N))
        best = {}
        for rep in range(3):                        # variants interleaved, best of 3: clocks drift between launches
            for qw in (1, 2, 3):
                ops.tune(3, qw)
                best[qw] = min(best.get(qw, 1e30), timeit(call))
        for qw in (1, 2, 3):
            row += f" | qw{qw}: {best[qw]:8.1f} us {fl / best[qw] / 1e6:7.1f} TF"
        ops.tune(3, 0)
        print(row, flush=True)


def bench_norm(B=8):
    for (HW, Cc, cnt) in ((4096, 320, 17), (4096, 640, 2), (4096, 960, 1), (1024, 640, 14), (1024, 1280, 2), (1024, 1920, 1), (256, 1280, 16), (256, 2560, 3), (64, 1280, 14), (64, 2560, 3)):
        x = rnd(B, HW, Cc)
        g, b = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
        us = timeit(lambda: ops.groupnorm(x, g, b, 1e-5, True))
        print(f"groupnorm HW={HW:5d} C={Cc:5d} x{cnt:2d}: {us:7.1f} us  {4.0 * B * HW * Cc / us / 1e6:6.2f} TB/s (rd+wr once)", flush=True)
    for (rows, Cc) in ((32768, 320), (8192, 640), (2048, 1280)):
        x = rnd(rows, Cc)
        g, b = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
        us = timeit(lambda: ops.layernorm(x, g, b))
        print(f"layernorm rows={rows:6d} C={Cc:5d}: {us:7.1f} us  {4.0 * rows * Cc / us / 1e6:6.2f} TB/s", flush=True)
    h = rnd(32768, 2560)
    us = timeit(lambda: ops.geglu(h))
    print(f"geglu 32768x2560: {us:7.1f} us {32768 * 2560 * 3.0 / us / 1e6:6.2f} TB/s")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--variants", default="-1")
    a = ap.parse_args()
    vs = [int(v) for v in a.variants.split(",")]
    if a.what in ("conv", "all"):
        bench_conv(vs)
    if a.what in ("gemm", "all"):
        bench_gemm(vs)
    ops.tune(1, -1)
    if a.what in ("attn", "all"):
        bench_attn()
    if a.what in ("norm", "all"):
        bench_norm()
