#!/usr/bin/env python3
"""In-kernel clock and cycles per (32 queries x 64 keys) unit of the attention kernel, from s_memtime / s_memrealtime stamps of wave 0
of every workgroup around its K/V sweep (diagnostic build -DPBE_ATTN_STAMPS, never the shipped library; MI355X_MICROARCH.md 'DVFS
give-back' item 6: stamp after >= 2 s of back-to-back launches on random data):
    python -m pbe_amd.build --diag=astamps:PBE_ATTN_STAMPS && PBE_LIB_PATH=tools/_dbg/libpbe_hip_astamps.so python tools/attn_stamps.py"""
import ctypes as C
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pbe_amd import lib, ops  # noqa: E402

dev = torch.device("cuda:0")


def main():
    L = lib.load()
    B, H = 8, 8
    for (N, D, qw) in ((4096, 40, 0), (4096, 40, 2), (4096, 40, 1), (1024, 80, 1)):
        Cc = H * D
        qk = torch.randn(B * N, 2 * Cc, device=dev).half()
        vt = torch.randn(B, Cc, N, device=dev).half()
        ops.tune(3, qw)
        call = lambda: ops.attention(qk, qk[:, Cc:], vt, B, H, N, N, D, D ** -0.5, q_strides=(N * 2 * Cc, 2 * Cc), k_strides=(N * 2 * Cc, 2 * Cc),
                                     vt_strides=(Cc * N, N))
        t0 = time.time()
        while time.time() - t0 < 2.0:                 # let the clock settle under this load
            for _ in range(50):
                call()
            torch.cuda.synchronize()
        QW = 1 if qw == 1 else 2
        nwg = (N // (128 * QW)) * B * H
        st = torch.zeros(nwg * 4, dtype=torch.int64, device=dev)
        L.pbe_debug_set_attn_stamps(C.c_void_p(st.data_ptr()))
        for _ in range(20):
            call()
        torch.cuda.synchronize()
        L.pbe_debug_set_attn_stamps(None)
        s = st.view(nwg, 4).cpu()
        cyc = (s[:, 2] - s[:, 0]).double()
        ns = (s[:, 3] - s[:, 1]).double() * 10.0      # s_memrealtime ticks at 100 MHz
        ghz = statistics.median((cyc / ns).tolist())
        units = (N // 64) * QW
        print(f"N={N} d={D} qw={qw}: in-kernel clock {ghz:.3f} GHz (median over {nwg} workgroups), K/V sweep {statistics.median(cyc.tolist()):.0f} cycles "
              f"= {statistics.median(cyc.tolist()) / units:.0f} cycles per unit per wave ({units} units), {statistics.median(ns.tolist()) / 1e3:.1f} us", flush=True)
    ops.tune(3, 0)


if __name__ == "__main__":
    main()
