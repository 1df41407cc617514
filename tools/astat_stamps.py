#!/usr/bin/env python3
"""Phase accounting of the A-stationary tiles (igemm_astat.hip) on the K = 320 GEGLU projection: diagnostic build with -DPBE_STAMPS
(python -m pbe_amd.build --stamps), wave 0 of every workgroup.  python tools/astat_stamps.py [cfg ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PBE_LIB_PATH"] = os.environ.get("PBE_STAMPS_LIB", os.path.join(ROOT, "tools", "_dbg", "libpbe_hip_stamps.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pbe_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def main():
    h = lib.load()
    M, C = 32768, 320
    x = torch.randn(M, C, device=dev).half()
    wg, bg = (torch.randn(8 * C, C, device=dev) * C ** -0.5).half(), torch.randn(8 * C, device=dev)
    c1 = wg.float().sum(1).contiguous()
    st = ops.row_stats(x)
    buf = torch.zeros(16 << 14, dtype=torch.int64, device=dev)
    for cfg in [int(c) for c in sys.argv[1:]] or [19]:
        ops._FORCE_CFG = cfg
        for _ in range(3):
            ops.gemm(x, wg, bg, act=ops.ACT_GEGLU, ln=(st, c1, 1e-5))
        buf.zero_()
        h.pbe_debug_set_stamps(lib.c_vp(buf.data_ptr()))
        ops.gemm(x, wg, bg, act=ops.ACT_GEGLU, ln=(st, c1, 1e-5))
        torch.cuda.synchronize()
        h.pbe_debug_set_stamps(None)
        s = buf.cpu().numpy().reshape(-1, 16)
        s = s[s[:, 0] != 0]
        tot = (s[:, 6] - s[:, 0]).astype(np.float64)
        wall = (s[:, 8] - s[:, 7]) / 100.0
        span = (s[:, 8].max() - s[:, 7].min()) / 100.0
        start = (s[:, 7] - s[:, 7].min()) / 100.0
        print(f"== cfg {cfg}: {len(s)} workgroups, span {span:.1f} us, median lifetime {np.median(tot):.0f} cycles = {np.median(wall):.1f} us "
              f"({np.median(tot / wall) / 1e3:.2f} GHz); starts: median +{np.median(start):.1f} us, p90 +{np.percentile(start, 90):.1f}, max +{start.max():.1f}")
        print(f"   prologue (A fragments issued, first weight tiles issued) {np.median(s[:, 1] - s[:, 0]):8.0f}")
        for j, n in ((9, "counted vmcnt waits"), (11, "barrier"), (12, "fragment reads + DMA issue"), (13, "MFMA block (issue)"), (10, "epilogue")):
            print(f"   wave 0: {n:28s} median {np.median(s[:, j]):9.0f} cycles ({100 * np.median(s[:, j] / tot):5.1f} %)  p10 {np.percentile(s[:, j], 10):9.0f}  p90 {np.percentile(s[:, j], 90):9.0f}")
    ops._FORCE_CFG = None


if __name__ == "__main__":
    main()
