"""COCOEE test-bench sweep on the HIP path, narrow model, synthetic 128x128 bench:
images equal the CPU oracle's pipeline on the same ids / noise, and do not depend on the rank count."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

import modelbuild as build
import cases
from oracle_loader import O
from pbe_amd import testbench as tb
from test_testbench_cpu import make_bench

pytestmark = pytest.mark.gpu
IDS = [11, 222, 3333, 44444, 5]               # 5 ids, batch 2 -> 2 full batches, id 5 dropped (drop_last)


def _png(path):
    return np.asarray(Image.open(path)).astype(np.int32)


def test_sweep_matches_oracle_and_is_rank_invariant(dev, tmp_path):
    bench = make_bench(str(tmp_path / "bench"), IDS, hw=128)
    ds = tb.COCOImageDataset(bench)
    with torch.no_grad():
        model = build.narrow_model(dev)
        one = tb.run_sweep(model, ds, str(tmp_path / "w1"), batch_size=2, steps=4, scale=5.0, plms=True, seed=7)
        two = [tb.run_sweep(model, ds, str(tmp_path / "w2"), batch_size=2, steps=4, scale=5.0, plms=True, seed=7, rank=r, world=2) for r in (0, 1)]
    assert one["ids"] == ["000000000011", "000000000222", "000000003333", "000000044444"] and one["batches"] == 2
    assert two[0]["ids"] == one["ids"][:2] and two[1]["ids"] == one["ids"][2:]
    for stem in one["ids"]:
        for sub, name in (("results", stem + ".png"), ("grid", "grid-" + stem + ".png"), ("samples", stem + "_mask.png"),
                          ("samples", stem + "_GT.png"), ("samples", stem + "_inpaint.png"), ("samples", stem + "_ref.png")):
            a, b = os.path.join(str(tmp_path), "w1", sub, name), os.path.join(str(tmp_path), "w2", sub, name)
            assert os.path.getsize(a) > 0
            assert np.array_equal(_png(a), _png(b)), f"{sub}/{name} differs between 1 and 2 ranks"
    assert not os.path.exists(os.path.join(str(tmp_path), "w1", "results", "000000000005.png"))
    # CPU oracle on batch 1 (ids 3333, 44444): same formulas, same (seed, batch) noise, fp32
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    items = [ds[i] for i in (2, 3)]
    image = torch.stack([it[0] for it in items])
    mask = torch.stack([it[1]["inpaint_mask"] for it in items])
    ref = torch.cat([it[1]["ref_imgs"] for it in items])
    nz = tb.batch_noise(7, 1, 2, 4, 16, 16)
    want = O.inpaint_pipeline(sd, image, mask, ref, nz["x_T"], nz["post_eps"], S=4, scale=5.0, unet_cfg=cases.UNET_NARROW,
                              vae_cfg=cases.VAE_NARROW, clip_cfg=cases.CLIP_NARROW, map_cfg=cases.MAPPER_NARROW)["image"]
    for k, stem in enumerate(one["ids"][2:]):
        got = _png(os.path.join(str(tmp_path), "w1", "results", stem + ".png"))
        exp = (255.0 * want[k].permute(1, 2, 0).numpy()).astype(np.uint8).astype(np.int32)
        d = np.abs(got - exp)
        assert d.mean() <= 1.5 and np.percentile(d, 99) <= 8, (d.mean(), d.max())      # fp16 path vs fp32 oracle after 5 U-Net calls, in 8-bit levels


def test_device_preprocess_matches_dataset_items(dev, tmp_path):
    """uint8 up, normalisation on the GPU (pbe_u8_to_planes_f32): equal, bit for bit, to the float tensors of the reference-shaped
    dataset item (test_bench_dataset.py:74-99); the uint8 pack (pbe_planes_to_u8_canvas) equals `(255 * x).astype(uint8)`."""
    ds = tb.COCOImageDataset(make_bench(str(tmp_path)))
    items = [ds.load_uint8(i) for i in range(3)]
    t = tb.device_preprocess(torch.from_numpy(np.stack([x[0] for x in items])).to(dev), torch.from_numpy(np.stack([x[1] for x in items])).to(dev),
                             torch.from_numpy(np.stack([x[2] for x in items])).to(dev))
    for i in range(3):
        image, kw, _ = ds[i]
        assert torch.equal(t["image"][i].cpu(), image) and torch.equal(t["mask"][i].cpu(), kw["inpaint_mask"])
        assert torch.equal(t["ref"][i].cpu(), kw["ref_imgs"][0])
    x = torch.rand(2, 3, 40, 56, generator=torch.Generator().manual_seed(1))
    x[0, :, 0, :8] = torch.tensor([0.0, 1.0, 0.5, 1.0 / 255, 254.999 / 255, 0.999999, 1e-8, 0.25])
    got = tb.device_pack_u8(x.to(dev)).cpu().numpy()
    assert np.array_equal(got, (255.0 * x.permute(0, 2, 3, 1).numpy()).astype(np.uint8))
