"""Is the halo conv bound by its schedule or by the power limit?  The shipped kernel on random and on ZERO operands (same launches,
same cycle counts; zero operands draw far less matrix-pipe power), to be read beside tools/ubench_loop.hip run in the same gpurun call:
    python tools/power_limit_check.py; UBENCH_RANDOM=1 gpurun_out/ubench_loop
(profiles/r02_power_limit_same_device.txt)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbe_amd import ops
dev = torch.device("cuda:0")
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).half()
def run(B, H, C, Co, cfg, zero=False):
    x = rnd(B, H, H, C); w = rnd(Co, 9 * C); b = torch.randn(Co, device=dev)
    if zero: x.zero_(); w.zero_()
    ops.tune(1, cfg | (1 << 8))
    for _ in range(5): ops.conv3x3(x, w, b, stride=1, pad=1)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    n = 40
    for _ in range(n): ops.conv3x3(x, w, b, stride=1, pad=1)
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / n * 1e3
    fl = 2.0 * B * H * H * Co * 9 * C
    nk = 9 * C // 64
    print(f"conv {B}x{H}x{H}x{C}->{Co} cfg{cfg} {'zero' if zero else 'random'} operands: {us:7.1f} us per launch (back to back, incl. launch gap) = {fl/us/1e6:6.0f} TFLOP/s; {nk} k-tiles -> <= {us/nk*1e3:6.0f} ns per k-tile")
    ops.tune(1, -1)
for z in (False, True):
    run(8, 64, 320, 320, 10, z)
    run(8, 64, 1280, 320, 10, z)     # 180 k-tiles: prologue / epilogue share small
    run(8, 64, 1280, 256, 13, z)
