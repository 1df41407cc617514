#!/usr/bin/env python3
"""Where do two evaluations of the same U-Net sample part ways?  (VERDICT r01 "explain the 1.4e-3")

Runs the configs/v1.yaml U-Net twice on the same sample and compares EVERY kernel launch's output in launch order:

  --mode batch    sample 0 evaluated alone (U-Net batch 1) vs inside a batch of B (default 4): same launch sequence,
                  so launches are compared one to one.  Prints, per launch, the tile config / split-K factor both runs
                  used (pbe_gemm_plan / pbe_conv3x3_plan), the fraction of fp16 outputs that differ and the rel-L2;
                  marks the first differing launch and summarises the growth per U-Net block.
  --mode pinned   the same, with the batch-1 run launched under ops.pinned_batch_scale(B): every GEMM / conv takes the
                  split-K factor of the batch-B layer.  If the split-K factor (fp32 summation order) is the only
                  batch-dependent choice that changes bits, every launch is bit-identical.
  --mode paired   forward_nhwc(paired=True) vs the duplicated batch, block by block.

    python tools/layer_diff.py --mode batch --batch 4 > profiles/r02_layer_diff_batch.txt
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import modelbuild  # noqa: E402
from pbe_amd import ops  # noqa: E402

NAMES = {0: "256x256", 1: "256x128", 2: "128x256", 3: "128x128", 4: "128x64", 5: "64x128", 6: "64x64", 7: "256x320", 8: "128x320", 9: "128x160"}
WRAP = ("gemm", "conv3x3", "groupnorm", "layernorm", "attention")


class Recorder:
    """Clones the output of every wrapped ops.* call (first `rows_of` fraction only, to bound memory)."""

    def __init__(self, keep_fraction):
        self.keep, self.log, self.orig = keep_fraction, [], {}

    def __enter__(self):
        for name in WRAP:
            fn = getattr(ops, name)
            self.orig[name] = fn

            def wrapped(*a, _fn=fn, _name=name, **k):
                n0 = len(ops._PLANS)
                out = _fn(*a, **k)
                plan = ops._PLANS[n0] if len(ops._PLANS) > n0 else None
                lead = out.shape[0]
                take = max(1, int(round(lead * self.keep)))
                self.log.append((_name, tuple(out.shape), plan, out[:take].detach().clone()))
                return out
            setattr(ops, name, wrapped)
        ops._PLANS = []
        return self

    def __exit__(self, *exc):
        for name, fn in self.orig.items():
            setattr(ops, name, fn)
        ops._PLANS = None
        return False


def plan_str(plan):
    if plan is None:
        return "-"
    key, cfg, splits, bm, bn, wgs = plan
    return f"{NAMES.get(cfg, cfg)} S={splits} wg={wgs}"


def compare(la, lb, title, out):
    assert len(la) == len(lb), (len(la), len(lb))
    first = None
    print(f"# {title}: {len(la)} launches", file=out)
    print(f"{'#':>4} {'op':10s} {'shape (run A)':28s} {'plan A':24s} {'plan B':24s} {'differ':>9s} {'rel-L2':>10s}", file=out)
    n_diff_ops = 0
    for i, ((na, sa, pa, ta), (nb, sb, pb, tb)) in enumerate(zip(la, lb)):
        assert na == nb, (i, na, nb)
        n = min(ta.shape[0], tb.shape[0])
        a, b = ta[:n].float(), tb[:n].float()
        neq = (a != b).float().mean().item()
        rel = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        mark = ""
        if neq > 0:
            n_diff_ops += 1
            if first is None:
                first, mark = i, "  <-- first difference"
        print(f"{i:4d} {na:10s} {str(sa):28s} {plan_str(pa):24s} {plan_str(pb):24s} {neq:9.2e} {rel:10.2e}{mark}", file=out)
    print(f"# {title}: first differing launch = {first}, {n_diff_ops} of {len(la)} launches differ", file=out)
    return first


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=("batch", "pinned", "paired"), default="batch")
    ap.add_argument("--batch", type=int, default=4, help="images (the U-Net runs 2x this under guidance)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    out = sys.stdout
    with torch.no_grad():
        model = modelbuild.full_model(dev)
        unet = model.model.diffusion_model
        g = torch.Generator().manual_seed(5)
        NB = 2 * a.batch
        x = torch.randn(NB, 9, 64, 64, generator=g).to(dev)
        ctx = torch.randn(NB, 1, 768, generator=g).to(dev)
        t = torch.full((NB,), 621, dtype=torch.int64, device=dev)
        x16 = ops.nchw_to_nhwc(x, unet.pk().cin_pad)
        if a.mode in ("batch", "pinned"):
            with Recorder(1.0 / NB) as big:
                unet.forward_nhwc(x16, t, ctx)
            with Recorder(1.0) as one:
                if a.mode == "pinned":
                    with ops.pinned_batch_scale(NB):
                        unet.forward_nhwc(x16[:1].contiguous(), t[:1], ctx[:1].contiguous())
                else:
                    unet.forward_nhwc(x16[:1].contiguous(), t[:1], ctx[:1].contiguous())
            # the timestep / context GEMMs at the head run on [NB, C] rows: row 0 is sample 0 in both runs
            compare(one.log, big.log, f"sample 0: U-Net batch 1 ({'pinned to the batch-%d plans' % NB if a.mode == 'pinned' else 'own plans'}) "
                                      f"[A] vs batch {NB} [B]", out)
        else:
            B = a.batch
            xs = torch.randn(B, 4, 64, 64, generator=g).to(dev)
            zs = torch.randn(B, 4, 64, 64, generator=g).to(dev)
            m = (torch.rand(B, 1, 64, 64, generator=g) > 0.3).float().to(dev)
            blocks = {}

            def rec(tag):
                def hook(self, block, h, *aa, **kk):
                    r = orig(self, block, h, *aa, **kk)
                    blocks.setdefault(tag, []).append(r.detach().clone())
                    return r
                return hook
            orig = type(unet)._run_block
            for tag, paired, dup in (("dup", False, 2), ("paired", True, 1)):
                type(unet)._run_block = rec(tag)
                try:
                    y = unet.forward_nhwc(ops.plms_pack_input(xs, zs, m, dup), t, ctx, paired=paired)
                finally:
                    type(unet)._run_block = orig
                blocks[tag + "_y"] = y.detach().clone()
            # the paired run skips the first block inside _run_block (it is evaluated by run_paired): align from the end
            da, pa = blocks["dup"], blocks["paired"]
            off = len(da) - len(pa)
            print(f"# paired vs duplicated, B = {B}: block outputs (aligned from the end, {len(pa)} blocks)", file=out)
            for i, pb in enumerate(pa):
                d = da[i + off]
                neq = (d != pb).float().mean().item()
                rel = ((d.float() - pb.float()).norm() / d.float().norm()).item()
                print(f"block {i + off:3d}  differ {neq:9.2e}  rel-L2 {rel:9.2e}", file=out)
            d, pb = blocks["dup_y"], blocks["paired_y"]
            print(f"output     differ {(d != pb).float().mean().item():9.2e}  rel-L2 {((d.float() - pb.float()).norm() / d.float().norm()).item():9.2e}", file=out)


if __name__ == "__main__":
    main()
