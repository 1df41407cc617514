#!/usr/bin/env python3
"""Measure the implicit-GEMM block-tile choice per shape on an MI355X and write the table the
product reads at import (pbe_amd/tuned_mi355x.json).  Deterministic at run time: the table is data.

    python tools/autotune.py [--batches 1,4,8] [--out pbe_amd/tuned_mi355x.json]

1. runs the full pipeline once per batch size with shape recording on (every pbe_gemm_f16 /
   pbe_conv3x3_f16 call of CLIP + VAE + 50 PLMS steps),
2. times every tile config on synthetic operands of each distinct shape (interleaved, one process),
3. stores the fastest (tile config | split-K factor << 8) per shape key: for the best three tile configs the split-K
   factor is searched as well (the built-in factor heuristic only knows the grid size).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from pbe_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
NCFG = 19
SPLITS = (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32)


def timeit(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).half()


def make_call(key):
    f = key.split(":")
    if f[0] == "g":
        M, N, K, batch = (int(v) for v in f[1:5])
        if batch == 1:
            a, w = rnd(M, K), rnd(N, K)
        else:
            a, w = rnd(batch, M, K), rnd(batch, N, K)
        bias = torch.randn(N, device=dev) if batch == 1 else None
        if batch == 1 and N == 8 * K and K in (320, 640, 1280):       # the GEGLU projections of the U-Net (N = 8 C): time their own epilogue
            return (lambda: ops.gemm(a, w, bias, act=ops.ACT_GEGLU)), 2.0 * M * N * K
        return (lambda: ops.gemm(a, w, bias)), 2.0 * M * N * K * batch
    B, H, W, C1, C2, Co, st, pad, ups = (int(v) for v in f[1:10])
    x = rnd(B, H, W, C1)
    x2 = rnd(B, H, W, C2) if C2 else None
    w = rnd(Co, 9 * (C1 + C2))
    bias = torch.randn(Co, device=dev)
    Ho, Wo = ops.conv_out_hw(H, W, st, pad, bool(ups))
    return (lambda: ops.conv3x3(x, w, bias, x2=x2, stride=st, pad=pad, upsample=bool(ups))), 2.0 * B * Ho * Wo * Co * 9 * (C1 + C2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,4,8")
    ap.add_argument("--out", default=os.path.join(ROOT, "pbe_amd", "tuned_mi355x.json"))
    ap.add_argument("--report", default=os.path.join(ROOT, "gpurun_out", "autotune_report.txt"))
    ap.add_argument("--merge", default="", help="existing table: its keys are kept as they are, only new shapes are measured")
    a = ap.parse_args()
    import cases
    import modelbuild
    from pbe_amd.pipeline import inpaint
    ops._TUNED.clear()
    t0 = time.time()
    with torch.no_grad():
        model = modelbuild.full_model(dev)
        rec = {}
        for B in [int(b) for b in a.batches.split(",")]:
            ops._RECORD = {}
            inp = {k: v.to(dev) for k, v in cases.synthetic_triples(B, 512).items()}
            inpaint(model, inp["image"], inp["mask"], inp["ref"], steps=2, scale=5.0, x_T=inp["x_T"], post_eps=inp["post_eps"])
            torch.cuda.synchronize()
            for k, v in ops._RECORD.items():
                rec[k] = max(rec.get(k, 0), v)
            print(f"[autotune] B={B}: {len(ops._RECORD)} distinct shapes ({time.time() - t0:.0f}s)", flush=True)
        ops._RECORD = None
        del model
        torch.cuda.empty_cache()
        table, lines = {}, []
        if a.merge:
            with open(a.merge) as f:
                table = json.load(f)
            rec = {k: v for k, v in rec.items() if k not in table}
            print(f"[autotune] {len(table)} shapes kept from {a.merge}, {len(rec)} new", flush=True)
        tot_h = tot_b = 0.0
        for key in sorted(rec):
            call, flops = make_call(key)
            iters = 20 if flops < 5e10 else 8
            times = []
            for cfg in [-1] + list(range(NCFG)):
                ops.tune(1, cfg)
                times.append(timeit(call, iters))
            order = sorted(range(NCFG), key=lambda c: times[c + 1])
            best, best_t = order[0], times[order[0] + 1]
            f = key.split(":")
            nk = (int(f[3]) if f[0] == "g" else 9 * (int(f[4]) + int(f[5]))) // 64
            split_note = ""
            if (f[0] == "c" or int(f[4]) == 1) and nk >= 8:                      # split-K exists for batch-1 problems with >= 8 k-tiles of 64
                for cfg in order[:5]:
                    for sp in SPLITS:
                        if sp > nk // 4:
                            break
                        ops.tune(1, cfg | (sp << 8))
                        t = timeit(call, iters)
                        if t < best_t * 0.985:                                    # a factor must win by > 1.5 % to displace the heuristic's
                            best, best_t, split_note = cfg | (sp << 8), t, f" split {sp}"
            ops.tune(1, -1)
            table[key] = best
            tot_h += times[0] * rec[key]
            tot_b += best_t * rec[key]
            lines.append(f"{key:44s} x{rec[key]:4d} heur {times[0]:8.1f} us | best cfg{best & 255}{split_note} {best_t:8.1f} us {flops / best_t / 1e6:7.1f} TF | "
                         + " ".join(f"{t:7.1f}" for t in times[1:]))
            print(lines[-1], flush=True)
        lines.append(f"weighted total (per recorded pass): heuristic {tot_h / 1e3:.2f} ms -> tuned {tot_b / 1e3:.2f} ms")
        print(lines[-1])
    os.makedirs(os.path.dirname(a.report), exist_ok=True)
    with open(a.report, "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(a.out, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print(f"[autotune] wrote {a.out} ({len(table)} shapes)")


if __name__ == "__main__":
    main()
