#!/usr/bin/env python3
"""Where does a workgroup of igemm_kernel spend its cycles?  (diagnostic build, cdna_hip_programming.md section 7 "in-kernel stamps")

Builds / loads tools/_dbg/libpbe_hip_stamps.so (python -m pbe_amd.build --stamps: the shipped sources with -DPBE_STAMPS), runs one
GEMM / conv shape of the path and prints, over all workgroups, the median / p10 / p90 cycles of each phase:

    setup      kernel entry -> loader state, tap table (conv), epilogue vectors staged
    prime      issue of the first D k-tiles' LDS-DMAs
    first      until the first k-tile has been consumed (prologue latency: first DMA landing)
    mainloop   remaining k-tiles
    stage      accumulators -> epilogue math -> LDS (includes the MFMA drain)
    copyout    LDS -> global stores (residual add)

    PBE_LIB_PATH is set by this script; never use the stamps build for timing claims (stamps cost ~11 % wave cycles).
    python tools/phase_stamps.py g:32768:2560:320:geglu  g:32768:320:320:resid  c:8:64:64:320:0:320  ...
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PBE_LIB_PATH"] = os.environ.get("PBE_STAMPS_LIB", os.path.join(ROOT, "tools", "_dbg", "libpbe_hip_stamps.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pbe_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")
NAMES = ["setup", "prime", "first", "mainloop", "stage", "copyout"]


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.5).half()


def make(spec):
    f = spec.split(":")
    if f[0] == "g":
        M, N, K = int(f[1]), int(f[2]), int(f[3])
        kind = f[4] if len(f) > 4 else ""
        a, w, bias = rnd(M, K), rnd(N, K), torch.randn(N, device=dev)
        resid = rnd(M, N) if kind == "resid" else None
        return (lambda: ops.gemm(a, w, bias, act=ops.ACT_GEGLU if kind == "geglu" else ops.ACT_NONE, resid=resid)), 2.0 * M * N * K
    B, H, W, C1, C2, Co = (int(v) for v in f[1:7])
    st = int(f[7]) if len(f) > 7 else 1
    ups = bool(int(f[8])) if len(f) > 8 else False
    x = rnd(B, H, W, C1)
    x2 = rnd(B, H, W, C2) if C2 else None
    w = rnd(Co, 9 * (C1 + C2))
    bias = torch.randn(Co, device=dev)
    Ho = H * 2 if ups else (H // 2 if st == 2 else H)
    return (lambda: ops.conv3x3(x, w, bias, x2=x2, stride=st, pad=1, upsample=ups)), 2.0 * B * Ho * Ho * Co * 9 * (C1 + C2)


def report(spec, s, plan, us, fl, out=sys.stdout):
    s = s[s[:, 0] != 0]
    d = np.stack([s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 5] - s[:, 4], s[:, 6] - s[:, 5]], 1).astype(np.float64)
    if plan[2] > 1:                                    # split-K workgroups end at the slab store: no stage / copy-out stamps
        d[:, 4] = 0
        d[:, 5] = s[:, 6] - s[:, 4]
    tot = (s[:, 6] - s[:, 0]).astype(np.float64)
    wall = (s[:, 8] - s[:, 7]).astype(np.float64) / 100.0      # us per workgroup (100 MHz wall clock)
    ghz = np.median(tot / np.maximum(wall, 1e-3)) / 1e3
    start = (s[:, 7] - s[:, 7].min()) / 100.0
    span = (s[:, 8].max() - s[:, 7].min()) / 100.0
    print(f"== {spec}: {'%.1f us event-timed, ' % us if us else ''}{span:.1f} us first start -> last end ({fl / span / 1e6:.0f} TFLOP/s), tile {plan[3]}x{plan[4]} split {plan[2]}, "
          f"{len(s)} workgroups; last start +{start.max():.1f} us; median lifetime {np.median(tot):.0f} cycles = {np.median(wall):.1f} us -> in-kernel clock {ghz:.2f} GHz", file=out)
    for i, n in enumerate(NAMES):
        print(f"   {n:9s} median {np.median(d[:, i]):9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}  "
              f"({100 * np.median(d[:, i]) / np.median(tot):5.1f} % of a workgroup)", file=out)
    loop = np.maximum(s[:, 4] - s[:, 2], 1).astype(np.float64)        # prime end -> main loop issued
    for j, n in ((9, "counted vmcnt waits"), (10, "barrier after the fragment reads (ping-pong)"), (11, "barrier closing a k-tile"),
                 (12, "DMA issue + fragment reads (to lgkmcnt 0 in the ping-pong loop, to issue in the plain loop)"), (13, "MFMA block")):
        print(f"   wave 0, first + main loop: {n:46s} median {np.median(s[:, j]):9.0f} cycles ({100 * np.median(s[:, j] / loop):5.1f} %)", file=out)


def pipeline(keys):
    """The same stamps taken INSIDE the real U-Net forward (batch 8): operands come from the producing kernels, weights from HBM."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import modelbuild
    h = lib.load()
    buf = torch.zeros(16 << 17, dtype=torch.int64, device=dev)
    with torch.no_grad():
        model = modelbuild.full_model(dev)
        unet = model.model.diffusion_model
        g = torch.Generator().manual_seed(5)
        x = torch.randn(8, 9, 64, 64, generator=g).to(dev)
        ctx = torch.randn(8, 1, 768, generator=g).to(dev)
        t = torch.full((8,), 621, dtype=torch.int64, device=dev)
        x16 = ops.nchw_to_nhwc(x, unet.pk().cin_pad)
        for _ in range(3):
            unet.forward_nhwc(x16, t, ctx)
        torch.cuda.synchronize()
        got = {}
        orig = {"gemm": ops.gemm, "conv3x3": ops.conv3x3}

        def wrap(name):
            fn = orig[name]

            def w(*a, **k):
                n0 = len(ops._PLANS)
                # find the key this launch WILL have: run it stamped when it is a wanted, not yet captured key
                buf.zero_()
                h.pbe_debug_set_stamps(lib.c_vp(buf.data_ptr()))
                out = fn(*a, **k)
                h.pbe_debug_set_stamps(None)
                plan = ops._PLANS[n0]
                key = plan[0] + ("|r" if k.get("resid") is not None else "") + ("|geglu" if k.get("act") == ops.ACT_GEGLU else "")
                if key in keys and key not in got:
                    torch.cuda.synchronize()
                    got[key] = (buf.cpu().numpy().reshape(-1, 16).copy(), plan)
                return out
            return w
        ops._PLANS = []
        ops.gemm, ops.conv3x3 = wrap("gemm"), wrap("conv3x3")
        try:
            unet.forward_nhwc(x16, t, ctx)
        finally:
            ops.gemm, ops.conv3x3 = orig["gemm"], orig["conv3x3"]
            ops._PLANS = None
    for key in keys:
        if key not in got:
            print(f"== {key}: not launched by the U-Net forward")
            continue
        s, plan = got[key]
        f = key.split("|")[0].split(":")
        if f[0] == "g":
            fl = 2.0 * int(f[1]) * int(f[2]) * int(f[3])
        else:
            B, H, W, C1, C2, Co, st, pad, ups = (int(v) for v in f[1:10])
            Ho, Wo = ops.conv_out_hw(H, W, st, pad, bool(ups))
            fl = 2.0 * B * Ho * Wo * Co * 9 * (C1 + C2)
        report("[in pipeline] " + key, s, plan, 0.0, fl)


def main():
    if sys.argv[1] == "--pipeline":
        return pipeline(sys.argv[2:])
    h = lib.load()
    buf = torch.zeros(16 << 17, dtype=torch.int64, device=dev)          # room for 131072 workgroups
    if os.environ.get("PBE_STAMP_CFG"):                                  # force a tile config (| split << 8) for every spec
        ops.tune(1, int(os.environ["PBE_STAMP_CFG"]))
    for spec in sys.argv[1:]:
        call, fl = make(spec)
        h.pbe_debug_set_stamps(None)
        for _ in range(3):
            call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(10):
            call()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 10 * 1e3
        buf.zero_()
        ops._PLANS = []
        h.pbe_debug_set_stamps(lib.c_vp(buf.data_ptr()))
        call()
        torch.cuda.synchronize()
        h.pbe_debug_set_stamps(None)
        plan, ops._PLANS = ops._PLANS[0], None
        report(spec, buf.cpu().numpy().reshape(-1, 16), plan, us, fl)


if __name__ == "__main__":
    main()
