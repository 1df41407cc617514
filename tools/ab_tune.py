#!/usr/bin/env python3
"""A/B of a pbe_tune knob on the U-Net's real GEMM / conv shapes (tuned tile table in force), interleaved rounds in ONE
process, median of rounds; also checks that both settings give BIT-identical outputs (the knobs change scheduling, not math).

    python tools/ab_tune.py --key 4 --values 0,1 [--what conv|gemm|all] [--batch 8]
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from pbe_amd import ops  # noqa: E402
import bench_kernels as bk  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def ab(name, fl, cnt, call, key, values, rounds, totals):
    outs = {}
    for v in values:
        ops.tune(key, v)
        outs[v] = call().clone()
    same = all(torch.equal(outs[values[0]], outs[v]) for v in values[1:])
    times = {v: [] for v in values}
    for _ in range(rounds):
        for v in values:
            ops.tune(key, v)
            call()
            times[v].append(timeit(call))
    row = f"{name:52s} x{cnt:2d}"
    for v in values:
        med = statistics.median(times[v])
        totals[v] += med * cnt
        row += f" | {v}: {med:7.1f} us {fl / med / 1e6:7.1f} TF"
    row += "  same-bits" if same else "  BITS DIFFER"
    print(row, flush=True)
    return same


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key", type=int, default=4)
    ap.add_argument("--values", default="0,1")
    ap.add_argument("--what", default="all")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    values = [int(v) for v in a.values.split(",")]
    B = a.batch
    ok = True
    if a.what in ("conv", "all"):
        tot = {v: 0.0 for v in values}
        fl_tot = 0.0
        for (H, c1, c2, co, st, ups, cnt) in bk.CONV:
            x = bk.rnd(B, H, H, c1)
            x2 = bk.rnd(B, H, H, c2) if c2 else None
            w = bk.rnd(co, 9 * (c1 + c2))
            bias = torch.randn(co, device=dev)
            Ho = H * 2 if ups else (H // 2 if st == 2 else H)
            fl = 2.0 * B * Ho * Ho * co * 9 * (c1 + c2)
            fl_tot += fl * cnt
            ok &= ab(f"conv {H:3d}^2 {c1:4d}+{c2:4d}->{co:4d} s{st} u{int(ups)}", fl, cnt,
                     lambda: ops.conv3x3(x, w, bias, x2=x2, stride=st, pad=1, upsample=ups), a.key, values, a.rounds, tot)
        for v in values:
            print(f"conv per U-Net forward, value {v}: {tot[v] / 1e3:.3f} ms -> {fl_tot / tot[v] / 1e6:.1f} TFLOP/s")
    if a.what in ("gemm", "all"):
        tot = {v: 0.0 for v in values}
        fl_tot = 0.0
        s = B // 8 if B >= 8 else 1
        for (M, N, K, cnt) in bk.GEMM:
            M = M * B // 8 if M >= 64 else M
            aa, w, bias = bk.rnd(M, K), bk.rnd(N, K), torch.randn(N, device=dev)
            geglu = N == 8 * K
            resid = bk.rnd(M, N) if (not geglu and N <= 1280 and M >= 64) else None
            fl = 2.0 * M * N * K
            fl_tot += fl * cnt
            ok &= ab(f"gemm M={M:6d} N={N:6d} K={K:5d}{' geglu' if geglu else (' +resid' if resid is not None else '')}", fl, cnt,
                     lambda: ops.gemm(aa, w, bias, act=ops.ACT_GEGLU if geglu else ops.ACT_NONE, resid=resid), a.key, values, a.rounds, tot)
        for v in values:
            print(f"gemm per U-Net forward, value {v}: {tot[v] / 1e3:.3f} ms -> {fl_tot / tot[v] / 1e6:.1f} TFLOP/s")
    print("ALL SAME BITS" if ok else "SOME OUTPUTS DIFFER")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
