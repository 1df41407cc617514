#!/usr/bin/env python3
"""Reduce two rocprofv3 PMC passes over the SAME `bench.py` command to the HBM-side traffic per launch of
the two implicit-GEMM kernel classes (3x3 conv: igemm_kernel<..., MODE = 1, S>; Linear / 1x1 GEMM: MODE = 0), as bench.py's
`roofline.traffic` / `roofline.classes.*.traffic` want it, stamped with the library source hash and the tile-table hash.  Counters collected exactly as MI355X_MICROARCH.md §HBM prescribes:

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_bench/f -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-batch16
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_bench/w -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-batch16
    python tools/pmc_bench_traffic.py gpurun_out/pmc_bench/f gpurun_out/pmc_bench/w profiles/igemm_traffic.json

(separate passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2).  Units and gfx950 corrections:
both counters are KiB; FETCH_SIZE tallies 64 B per 128-B request for wide (16 B / lane) coalesced reads —
every global read of this kernel is a 16 B / lane LDS-DMA or epilogue load — so the read side is doubled;
WRITE_SIZE is exact for its 16 B / lane stores.  Infinity-Cache hits are counted (the counters sit on the
L2's fabric side), so this is traffic leaving L2, an upper bound on HBM bytes."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    assert fs, f"no counter_collection.csv under {d}"
    tot, n = collections.Counter(), collections.Counter()
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        tot[name] += float(r["Counter_Value"])
        key = (name, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            n[name] += 1
    return tot, n


PASSES = 2          # the profiled command runs the hot path twice (1 timed step + the stage-split pass)


def main(fdir, wdir, outp):
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from pbe_amd import lib
    ft, fn = per_kernel(fdir, "FETCH_SIZE")
    wt, wn = per_kernel(wdir, "WRITE_SIZE")

    def mode_of(k):                                   # igemm_kernel<BM, BN, NWM, NWN, MODE, S, HPA, PP, F8>: MODE 1 / 2 = 3x3 conv (gather / halo), 0 = dense GEMM
        if not k.startswith("igemm_kernel<"):
            return None
        args = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
        return args[4] if len(args) >= 5 else None
    classes = {}
    # every class of bench.py's roofline.classes: the two igemm classes by MODE, the others by kernel name (all of them move 16 B per lane:
    # LDS-DMA pieces, h16x8 / f32x4 loads and stores - the access shape the guide's FETCH_SIZE x 2 correction is calibrated for)
    sel = (("conv3x3_igemm", lambda k: mode_of(k) in ("1", "2"), "igemm_kernel<*,*,*,*,1|2,*> (3x3 conv: gather and halo-resident implicit GEMM)"),
           ("gemm", lambda k: mode_of(k) == "0" or "astat_regs_kernel" in k, "igemm_kernel<*,*,*,*,0,*> + astat_regs_kernel<*,*> (Linear / 1x1 conv GEMM)"),
           ("attention", lambda k: "attn_kernel" in k, "attn_kernel<*> (fused self-attention)"),
           ("groupnorm", lambda k: any(n in k for n in ("gn_stats_kernel", "gn_apply_kernel", "gn_small_kernel")), "gn_stats + gn_apply, gn_small"),   # (non-template kernels arrive mangled)
           ("layernorm", lambda k: any(n in k for n in ("layernorm_kernel", "layernorm_f8_kernel", "row_stats_kernel")), "layernorm / row statistics kernels"),
           ("splitk_reduce", lambda k: "splitk_reduce_kernel" in k, "splitk_reduce_kernel"))
    for cname, pred, label in sel:
        ks = [k for k in ft if pred(k)]
        if not ks:
            continue
        launches = sum(fn[k] for k in ks)
        assert launches and launches == sum(wn[k] for k in ks), (cname, launches, sum(wn[k] for k in ks))
        rd = sum(ft[k] for k in ks) * 1024 * 2
        wr = sum(wt[k] for k in ks) * 1024
        classes[cname] = {"kernel": label, "launches": launches, "read_B_per_launch": rd / launches, "write_B_per_launch": wr / launches,
                          "traffic_B_per_launch": (rd + wr) / launches,
                          "traffic_B_per_pass": (rd + wr) / PASSES,          # bench.py divides by ITS launch count (a GroupNorm call = 2 dispatches)
                          "per_tile_config": {k: {"launches": fn[k], "read_B_per_launch": ft[k] * 2048 / fn[k], "write_B_per_launch": wt[k] * 1024 / max(1, wn[k])}
                                              for k in sorted(ks)}}
    with open(os.path.join(root, "pbe_amd", "tuned_mi355x.json"), "rb") as f:
        table = hashlib.sha256(f.read()).hexdigest()[:16]
    out = {"command": "bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-batch16 (2 passes of the hot path)", "passes": PASSES, "source_hash": lib.source_hash(), "tuned_table_sha": table,
           "corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950: 64 B tallied per 128-B request); WRITE_SIZE KiB x 1024", "classes": classes}
    with open(outp, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "per_tile_config"} for k, v in classes.items()}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
