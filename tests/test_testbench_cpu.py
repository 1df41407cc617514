"""Host logic of the COCOEE test-bench sweep: dataset item layout and formulas of
ldm/data/test_bench_dataset.py:61-105, id partition over ranks, per-batch noise, output tree of
scripts/inference_test_bench.py:361-397.  No GPU, no oracle compute."""
import importlib.util
import os

import numpy as np
import pytest
import torch
from PIL import Image

from pbe_amd import testbench as tb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IDS = [1773, 3419, 900100376112, 42, 7]


def make_bench(root, ids=IDS, hw=64):
    rng = np.random.default_rng(0)
    for d in ("GT_3500", "Ref_3500", "Mask_bbox_3500"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    for k, i in enumerate(ids):
        s = str(i).zfill(12)
        Image.fromarray(rng.integers(0, 256, (hw, hw, 3), dtype=np.uint8)).save(os.path.join(root, "GT_3500", s + "_GT.png"))
        Image.fromarray(rng.integers(0, 256, (50 + k, 70, 3), dtype=np.uint8)).save(os.path.join(root, "Ref_3500", s + "_ref.png"))
        m = np.zeros((hw, hw), dtype=np.uint8)
        m[10 + k:40, 8:30 + k] = 255
        m[0, 0] = 100                                    # a grey pixel: the test bench does NOT threshold the mask
        Image.fromarray(m).save(os.path.join(root, "Mask_bbox_3500", s + "_mask.png"))
    np.save(os.path.join(root, "id_list.npy"), np.asarray(ids, dtype=np.int64))
    return root


def test_id_list_fixture_is_the_reference_list():
    a = np.load(os.path.join(ROOT, "test_bench", "id_list.npy"), allow_pickle=False)
    assert a.shape == (3500,) and a.dtype == np.int64
    assert a[:5].tolist() == [1773, 3419, 3658, 4594, 7544] and int(a[-1]) == 900100065455
    assert tb.id_stem(a[0]) == "000000001773" and tb.id_stem(a[-1]) == "900100065455"


def test_dataset_item_layout_and_formulas(tmp_path):
    from ldm.data.test_bench_dataset import COCOImageDataset
    assert COCOImageDataset is tb.COCOImageDataset
    root = make_bench(str(tmp_path))
    ds = COCOImageDataset(root)
    assert len(ds) == len(IDS)
    image, kw, stem = ds[2]
    assert stem == "900100376112"
    assert image.shape == (3, 64, 64) and image.dtype == torch.float32 and -1 <= image.min() and image.max() <= 1
    assert kw["inpaint_mask"].shape == (1, 64, 64) and kw["ref_imgs"].shape == (1, 3, 224, 224) and kw["inpaint_image"].shape == (3, 64, 64)
    # independent restatement of test_bench_dataset.py:74-99
    gt = np.asarray(Image.open(ds.paths(2)[0]).convert("RGB"), dtype=np.float32) / 255.0
    assert torch.allclose(image, torch.from_numpy((gt - 0.5) / 0.5).permute(2, 0, 1))
    mk = 1.0 - np.asarray(Image.open(ds.paths(2)[2]).convert("L"), dtype=np.float32) / 255.0
    assert torch.allclose(kw["inpaint_mask"][0], torch.from_numpy(mk))
    assert abs(kw["inpaint_mask"][0, 0, 0].item() - (1 - 100 / 255)) < 1e-6          # not thresholded
    assert torch.allclose(kw["inpaint_image"], image * kw["inpaint_mask"])
    ref = np.asarray(Image.open(ds.paths(2)[1]).resize((224, 224)).convert("RGB"), dtype=np.float32) / 255.0
    ref = (ref - np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)) / np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)
    assert torch.allclose(kw["ref_imgs"][0], torch.from_numpy(ref).permute(2, 0, 1), atol=1e-6)


def test_device_preprocess_has_no_cpu_fallback(tmp_path):
    """device_preprocess / device_pack_u8 are HIP kernels (pbe_u8_to_planes_f32, pbe_planes_to_u8_canvas): CPU tensors are refused.
    Their equality with the dataset items is tested on the GPU (tests/test_testbench_gpu.py)."""
    from pbe_amd.lib import PbeError
    ds = tb.COCOImageDataset(make_bench(str(tmp_path)))
    items = [ds.load_uint8(i) for i in range(2)]
    with pytest.raises(PbeError):
        tb.device_preprocess(torch.from_numpy(np.stack([x[0] for x in items])), torch.from_numpy(np.stack([x[1] for x in items])),
                             torch.from_numpy(np.stack([x[2] for x in items])))
    with pytest.raises(PbeError):
        tb.device_pack_u8(torch.rand(1, 3, 8, 8))


def test_rank_batches_drop_last_and_partition():
    for n, bs, world in ((3500, 8, 8), (3500, 8, 1), (21, 4, 2), (7, 8, 2), (64, 8, 3)):
        parts = [tb.rank_batches(n, bs, r, world) for r in range(world)]
        flat = sorted(b for p in parts for b, _ in p)
        assert flat == list(range(n // bs))                                   # every full batch exactly once; remainder dropped
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        for p in parts:
            for b, idx in p:
                assert idx == list(range(b * bs, (b + 1) * bs))
    assert len(tb.rank_batches(3500, 8, 0, 8)) == 55                          # 3500 ids, 8 x 8 in flight -> 55 rounds (437 batches)


def test_batch_noise_depends_on_seed_and_batch_only():
    a, b = tb.batch_noise(42, 3, 2, 4, 16, 16), tb.batch_noise(42, 3, 2, 4, 16, 16)
    assert torch.equal(a["x_T"], b["x_T"]) and torch.equal(a["post_eps"], b["post_eps"])
    assert not torch.equal(a["x_T"], tb.batch_noise(42, 4, 2, 4, 16, 16)["x_T"])
    assert not torch.equal(a["x_T"], tb.batch_noise(43, 3, 2, 4, 16, 16)["x_T"])
    assert not torch.equal(a["x_T"], a["post_eps"])


def test_write_item_output_tree(tmp_path):
    bench = make_bench(str(tmp_path / "bench"))
    ds = tb.COCOImageDataset(bench)
    out = str(tmp_path / "out")
    for d in ("samples", "results", "grid"):
        os.makedirs(os.path.join(out, d))
    img, ref, mask, stem = ds.load_uint8(0)
    res = np.full((64, 64, 3), 200, dtype=np.uint8)
    paths = tb.write_item(out, stem, img, mask, ref, res)
    assert sorted(os.path.relpath(p, out) for p in paths.values()) == sorted([
        "results/000000001773.png", "grid/grid-000000001773.png", "samples/000000001773_mask.png", "samples/000000001773_GT.png",
        "samples/000000001773_inpaint.png", "samples/000000001773_ref.png"])
    assert np.array_equal(np.asarray(Image.open(paths["result"])), res)
    assert np.array_equal(np.asarray(Image.open(paths["gt"])), img)
    g = np.asarray(Image.open(paths["grid"]))
    assert g.shape == (64 + 4, 4 * (64 + 2) + 2, 3)                            # make_grid: one row of 4 tiles, 2-px padding
    assert np.array_equal(g[2:66, 2:66], img) and np.array_equal(g[2:66, 2 + 3 * 66:2 + 3 * 66 + 64], res)
    m = np.asarray(Image.open(paths["mask"]))
    assert m.min() >= 127 and m.max() == 255                                   # the reference un-normalises the 0..1 mask: (m+1)/2
    inp = np.asarray(Image.open(paths["inpaint"])).astype(int)
    hole = mask == 255
    assert (inp[hole] == 127).all()                                            # image*0 -> un_norm -> 0.5 -> 127 (truncation)
    assert np.abs(inp[mask == 0] - img[mask == 0].astype(int)).max() <= 1
    assert tb.write_item(out, stem, img, mask, ref, res, skip_save=True) == {}


def test_cli_defaults_match_reference():
    spec = importlib.util.spec_from_file_location("pbe_tb_cli", os.path.join(ROOT, "scripts", "inference_test_bench.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    o = cli.parse([])
    assert (o.n_samples, o.scale, o.ddim_steps, o.seed, o.H, o.W, o.C, o.f, o.ddim_eta, o.plms, o.fixed_code, o.outdir) == \
        (5, 1, 50, 42, 512, 512, 4, 8, 0.0, False, False, "outputs/txt2img-samples")          # inference_test_bench.py:96-252
