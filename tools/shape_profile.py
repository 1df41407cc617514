#!/usr/bin/env python3
"""Per-shape GPU time of one U-Net forward inside the real sampler (events around every launch).

    python tools/shape_profile.py [--batch 4] [--steps 4] [--out gpurun_out/shape_profile.txt]

Unlike tools/bench_kernels.py (isolated launches, CPU-bound below ~15 us) this measures each
kernel where it runs: warm L2/MALL state of its producer, real operand bits, real clocks.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from pbe_amd import ops  # noqa: E402


def flops_of(key):
    f = key.split("|")[0].split(":")
    if f[0] == "g":
        M, N, K, b = (int(v) for v in f[1:5])
        return 2.0 * M * N * K * b
    if f[0] == "c":
        B, H, W, C1, C2, Co, st, pad, ups = (int(v) for v in f[1:10])
        Ho, Wo = ops.conv_out_hw(H, W, st, pad, bool(ups))
        return 2.0 * B * Ho * Wo * Co * 9 * (C1 + C2)
    if f[0] == "a":
        B, H, Nq, Nk, D = (int(v) for v in f[1:6])
        return 4.0 * B * H * Nq * Nk * D
    return 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "shape_profile.txt"))
    a = ap.parse_args()
    import cases
    import modelbuild
    from pbe_amd.pipeline import inpaint
    dev = torch.device("cuda:0")
    with torch.no_grad():
        model = modelbuild.full_model(dev)
        inp = {k: v.to(dev) for k, v in cases.synthetic_triples(a.batch, 512).items()}
        kw = dict(steps=a.steps, scale=5.0, x_T=inp["x_T"], post_eps=inp["post_eps"])
        inpaint(model, inp["image"], inp["mask"], inp["ref"], **kw)           # warm: packs, workspaces
        torch.cuda.synchronize()
        ops._TIMES = {}
        inpaint(model, inp["image"], inp["mask"], inp["ref"], **kw)
        torch.cuda.synchronize()
        times, ops._TIMES = ops._TIMES, None
    calls = a.steps + 1                                                        # PLMS: steps + 1 U-Net evaluations
    rows = []
    for key, evs in times.items():
        us = [e0.elapsed_time(e1) * 1e3 for e0, e1 in evs]
        rows.append((sum(us), key, len(us), sum(us) / len(us), min(us), flops_of(key)))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    lines = [f"batch {a.batch}, {a.steps} PLMS steps ({calls} U-Net calls at batch {2 * a.batch}) + CLIP + VAE; total timed {tot / 1e3:.2f} ms",
             f"{'key':52s} {'n/call':>7s} {'avg us':>9s} {'min us':>9s} {'TF/s':>7s} {'ms/call':>8s} {'cum %':>6s}"]
    cum = 0.0
    for t, key, n, avg, mn, fl in rows:
        cum += t
        lines.append(f"{key:52s} {n / calls:7.1f} {avg:9.1f} {mn:9.1f} {fl / avg / 1e6 if fl else 0:7.1f} {t / calls / 1e3:8.3f} {100 * cum / tot:6.1f}")
    kinds = {}
    for t, key, *_ in rows:
        kinds[key[0]] = kinds.get(key[0], 0.0) + t
    lines.append("per kind (ms per U-Net call): " + ", ".join(f"{k}={v / calls / 1e3:.3f}" for k, v in sorted(kinds.items())))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main()
