#!/usr/bin/env python3
"""MFMA and VALU utilisation of the matrix-core kernels over one bench pass, from ONE rocprofv3 PMC pass:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES \\
        --kernel-include-regex "igemm_kernel|attn_kernel|astat_regs_kernel" -d gpurun_out/pmc_mfma/p -o p --output-format csv -- \\
        python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile
    python tools/pmc_mfma_util.py gpurun_out/pmc_mfma/p profiles/r01_mfma_utilisation.json

Units (MI355X_MICROARCH.md "rocprofv3 PMC slots"): SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (= 32 x MFMAs for
32x32x16), SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count quad-cycles, GRBM_GUI_ACTIVE counts GPU-active cycles per dispatch.
GRBM_GUI_ACTIVE arrives summed over the 8 XCDs, so active cycles = GUI_ACTIVE / 8.  Two denominators are reported:
  *_gui: SIMD-cycles = GUI_ACTIVE / 8 x 1024 SIMDs;   *_ts: SIMD-cycles = sum(End - Start) ns x 2.4 GHz x 1024 (nominal clock).
mfma_busy = MFMA_BUSY / SIMD-cycles; valu_active = 4 x ACTIVE_INST_VALU / SIMD-cycles (includes MFMA issue slots);
waves_per_simd = 4 x WAVE_CYCLES / SIMD-cycles.  (Check: conv MFMA_BUSY = FLOP / 16384 x 16 cycles to < 1 %.)"""
import collections
import csv
import glob
import json
import sys


def main(d, outp):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    ns = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if name.startswith("igemm_kernel<"):
            args = [a.strip() for a in name[name.index("<") + 1:name.rindex(">")].split(",")]
            klass = "conv3x3_igemm" if args[4] in ("1", "2") else "gemm"          # MODE 1 gather / 2 halo-resident conv, 0 dense
        elif name.startswith("astat_regs_kernel<"):
            klass = "gemm"                                                        # the A-stationary K = 320 tiles (igemm_astat.hip)
        elif name.startswith("attn_kernel<"):
            klass = "attention"
        else:
            continue
        per[klass][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in calls[klass]:
            ns[klass] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        calls[klass].add(r["Dispatch_Id"])
    out = {}
    for k, c in per.items():
        gui = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        ts = ns[k] * 2.4 * 1024.0
        out[k] = {"dispatches": len(calls[k]), "kernel_ms": ns[k] / 1e6,
                  "mfma_busy_frac_gui": c["SQ_VALU_MFMA_BUSY_CYCLES"] / gui, "mfma_busy_frac_ts": c["SQ_VALU_MFMA_BUSY_CYCLES"] / ts,
                  "valu_active_frac_gui": 4.0 * c["SQ_ACTIVE_INST_VALU"] / gui, "valu_active_frac_ts": 4.0 * c["SQ_ACTIVE_INST_VALU"] / ts,
                  "waves_per_simd_gui": 4.0 * c["SQ_WAVE_CYCLES"] / gui}
    with open(outp, "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
