#!/usr/bin/env python3
"""Tile-config table measured WHERE THE KERNELS RUN: inside the real pipeline (CLIP + VAE + PLMS steps), not back to back.

tools/autotune.py times each shape in a tight loop: operands stay in L2 / the Infinity Cache and the winners it picks are up to
20 % faster there - and no faster in the sampler, where a layer's input was just written by another kernel's workgroups and its
weights come from HBM (tools/phase_stamps.py --pipeline: +31 % main-loop cycles on the 64x64 conv; profiles/r02_*).
This tuner forces ONE candidate (tile config | split-K factor) on every GEMM / conv launch of a whole pipeline pass, brackets every
launch with events, and repeats for every candidate; per shape it keeps the candidate with the lowest median time.  Candidates
that the library clamps to the same effective plan (pbe_*_plan) are merged.

    python tools/autotune_insitu.py --batches 1,2,4,8,16 --out pbe_amd/tuned_mi355x.json
"""
import argparse
import json
import re
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from pbe_amd import ops  # noqa: E402

NCFG = 22
SKIP = (19, 20)        # the A-stationary tiles apply to two K = 320 shapes only and are set by hand (tools/astat_ab.py)
SPLITS = (0, 1, 2, 3, 4, 6, 8, 12, 16, 24)        # 0 = the library's own factor for that tile


def one_pass(model, inp, steps, cand):
    from pbe_amd.pipeline import inpaint
    ops._FORCE_CFG = cand
    ops._TIMES, ops._PLANS = {}, []
    inpaint(model, inp["image"], inp["mask"], inp["ref"], steps=steps, scale=5.0, x_T=inp["x_T"], post_eps=inp["post_eps"])
    torch.cuda.synchronize()
    times, plans = ops._TIMES, ops._PLANS
    ops._TIMES, ops._PLANS, ops._FORCE_CFG = None, None, None
    eff = {}
    for key, cfg, sp, *_ in plans:                     # effective plan per shape key (same for every launch of the key)
        eff.setdefault(key, (cfg, max(1, sp)))
    out = {}
    for tkey, evs in times.items():
        if tkey[0] not in "gc":
            continue
        key = tkey.split("|")[0]
        us = sum(e0.elapsed_time(e1) for e0, e1 in evs) * 1e3
        tot, n = out.get(key, (0.0, 0))
        out[key] = (tot + us, n + len(evs))
    return {k: (v[0] / v[1], v[1], eff.get(k)) for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="4")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tuned_insitu.json"))
    ap.add_argument("--report", default=os.path.join(ROOT, "gpurun_out", "autotune_insitu_report.txt"))
    ap.add_argument("--merge", default="", help="start from this table; shapes measured here replace their entries")
    ap.add_argument("--image-size", type=int, default=512, help="768 = BASELINE configs[4] geometry")
    ap.add_argument("--precision", default="fp16", choices=("fp16", "fp8"))
    ap.add_argument("--only-regex", default="", help="keep only shape keys matching this regex (e.g. ':2$' = the phase-form upsampling convs); candidates: every tile, no split-K")
    ap.add_argument("--only-prefix", default="", help="keep only shape keys with this prefix (gx: = the extended-epilogue GEMMs); candidates are then its tiles only")
    a = ap.parse_args()
    import cases
    import modelbuild
    dev = torch.device("cuda:0")
    table = {}
    if a.merge:
        with open(a.merge) as f:
            table = json.load(f)
    lines = []
    t0 = time.time()
    with torch.no_grad():
        model = modelbuild.full_model(dev)
        if a.precision == "fp8":
            from pbe_amd.precision import set_linear_precision
            set_linear_precision(model, "fp8")
        for B in [int(b) for b in a.batches.split(",")]:
            inp = {k: v.to(dev) for k, v in cases.synthetic_triples(B, a.image_size).items()}
            one_pass(model, inp, a.steps, -1)                                   # warm: packs, workspaces
            cands = [cfg | (sp << 8) for cfg in range(NCFG) if cfg not in SKIP for sp in SPLITS]
            if a.only_regex:
                cands = [cfg | (1 << 8) for cfg in range(NCFG) if cfg not in SKIP]
            if a.only_prefix == "gx:":                                            # extended epilogue: its instantiated tiles, never split-K
                cands = [cfg | (1 << 8) for cfg in (3, 4, 5, 6, 8, 9, 15, 16, 17, 18)]
            data = {}                                                            # key -> {(cfg, splits): [us, ...]}
            counts = {}
            for rep in range(a.reps):
                order = cands if rep % 2 == 0 else cands[::-1]
                for cand in order:
                    for key, (us, n, eff) in one_pass(model, inp, a.steps, cand).items():
                        if eff is None:
                            continue
                        data.setdefault(key, {}).setdefault(eff, []).append(us)
                        counts[key] = n
                print(f"[insitu] B={B} rep {rep}: {len(data)} shapes, {time.time() - t0:.0f}s", flush=True)
            base = one_pass(model, inp, a.steps, -1)                            # the built-in heuristic, for the report
            tot_h = tot_b = 0.0
            for key in sorted(data):
                if a.only_prefix and not key.startswith(a.only_prefix):
                    continue
                if a.only_regex and not re.search(a.only_regex, key):
                    continue
                med = {eff: statistics.median(v) for eff, v in data[key].items()}
                best = min(med, key=med.get)
                table[key] = best[0] | (best[1] << 8)
                h = base.get(key, (float("nan"),))[0]
                tot_h += h * counts[key]
                tot_b += med[best] * counts[key]
                top = sorted(med.items(), key=lambda kv: kv[1])[:4]
                lines.append(f"{key:44s} x{counts[key]:4d} heur {h:8.1f} us | best cfg{best[0]} split {best[1]} {med[best]:8.1f} us | "
                             + "  ".join(f"cfg{e[0]}/s{e[1]} {t:.1f}" for e, t in top))
            lines.append(f"B={B}: weighted total per pass: heuristic {tot_h / 1e3:.2f} ms -> in-situ tuned {tot_b / 1e3:.2f} ms")
            print(lines[-1], flush=True)
    os.makedirs(os.path.dirname(a.report), exist_ok=True)
    with open(a.report, "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(a.out, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print(f"[insitu] wrote {a.out} ({len(table)} shapes)")


if __name__ == "__main__":
    main()
