#!/usr/bin/env python3
"""Reduce two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, gfx950 corrections of
MI355X_MICROARCH.md §HBM) over `tools/bench_kernels.py conv` to per-launch HBM traffic of the conv3x3
implicit-GEMM kernel, weighted by how often each shape occurs in one U-Net forward.
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r01_conv_traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced
streams, so the read side is doubled (the guide's correction); WRITE_SIZE is exact for 16-B stores."""
import collections
import csv
import glob
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tools")
from bench_kernels import CONV  # noqa: E402


def per_dispatch(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "igemm_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            out[int(r["Dispatch_Id"])] = out.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return list(out.values())


def main(fdir, wdir, outp):
    fetch, write = per_dispatch(fdir, "FETCH_SIZE"), per_dispatch(wdir, "WRITE_SIZE")
    per_shape = 23                                   # timeit(): 3 warm-up + 20 timed launches per shape, one variant
    B = 8
    assert len(fetch) == len(write) == per_shape * len(CONV), (len(fetch), len(write))
    rows, tot_alg, tot_hbm, tot_cnt = [], 0.0, 0.0, 0
    for i, (H, c1, c2, co, st, ups, cnt) in enumerate(CONV):
        f = sum(fetch[i * per_shape + 3:(i + 1) * per_shape]) / 20 * 1024 * 2      # KiB -> B, x2 gfx950 correction
        w = sum(write[i * per_shape + 3:(i + 1) * per_shape]) / 20 * 1024
        Ho = H * 2 if ups else (H // 2 if st == 2 else H)
        alg = 2.0 * (B * H * H * (c1 + c2) + co * 9 * (c1 + c2) + B * Ho * Ho * co)   # read x once, weights once, write y once (fp16)
        rows.append({"shape": f"{H}^2 {c1}+{c2}->{co} s{st} u{int(ups)}", "per_forward": cnt, "hbm_read_B": f, "hbm_write_B": w, "algorithmic_B": alg,
                     "hbm_over_algorithmic": (f + w) / alg})
        tot_alg += alg * cnt
        tot_hbm += (f + w) * cnt
        tot_cnt += cnt
    res = {"kernel": "igemm_kernel<*,*,*,*,1> (conv3x3)", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/bench_kernels.py conv, 2B=8",
           "avg_hbm_bytes_per_launch": tot_hbm / tot_cnt, "avg_algorithmic_bytes_per_launch": tot_alg / tot_cnt,
           "hbm_over_algorithmic": tot_hbm / tot_alg, "note": "back-to-back launches of one shape: operands partly served by the 256 MiB Infinity Cache", "shapes": rows}
    json.dump(res, open(outp, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "shapes"}, indent=1))
    for r in rows:
        print(f"{r['shape']:28s} x{r['per_forward']:2d} read {r['hbm_read_B'] / 1e6:8.1f} MB write {r['hbm_write_B'] / 1e6:7.1f} MB alg {r['algorithmic_B'] / 1e6:7.1f} MB ratio {r['hbm_over_algorithmic']:.2f}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
