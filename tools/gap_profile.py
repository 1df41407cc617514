#!/usr/bin/env python3
"""How much of the sampler's wall time is the GPU idle between kernels?  Reads a rocprofv3 --kernel-trace CSV of a bench run.

    rocprofv3 --kernel-trace -d gpurun_out/kt -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile
    python tools/gap_profile.py gpurun_out/kt profiles/r02_gpu_idle_between_kernels.txt

Takes the window from the first to the last U-Net kernel of the LAST pass (gaps above 200 us split passes / stages), sums kernel
time and the gaps between consecutive kernels, and prints the gap histogram."""
import csv
import glob
import sys


def main(d, outp):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # split into segments at gaps > 200 us (host work between passes / stages); keep the longest segment (one sampler)
    segs, cur = [], [rows[0]]
    for prev, r in zip(rows, rows[1:]):
        if r[0] - prev[1] > 200_000:
            segs.append(cur)
            cur = []
        cur.append(r)
    segs.append(cur)
    seg = max(segs, key=lambda s: s[-1][1] - s[0][0])
    span = seg[-1][1] - seg[0][0]
    busy, gaps, end = 0, [], seg[0][0]
    for s, e, _ in seg:
        if s > end:
            gaps.append(s - end)
        busy += max(0, e - max(s, end))
        end = max(end, e)
    lines = [f"longest gap-free segment (one sampler pass): {len(seg)} kernels, span {span / 1e6:.2f} ms, kernels busy {busy / 1e6:.2f} ms "
             f"({100.0 * busy / span:.1f} %), idle between kernels {sum(gaps) / 1e6:.2f} ms ({100.0 * sum(gaps) / span:.1f} %) in {len(gaps)} gaps",
             f"median gap {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, p90 {sorted(gaps)[int(len(gaps) * 0.9)] / 1e3:.2f} us, max {max(gaps) / 1e3:.1f} us"]
    for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 8), (8, 20), (20, 200)):
        g = [x for x in gaps if lo * 1000 <= x < hi * 1000]
        lines.append(f"  gaps {lo:3d}-{hi:3d} us: {len(g):6d}, {sum(g) / 1e6:7.2f} ms")
    open(outp, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
