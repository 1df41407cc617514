"""COCOEE test-bench sweep: the data format either side of the hot path (BASELINE config #4).

Mirrors, in zhanwenchen/pbe:
  * ``ldm/data/test_bench_dataset.py:61-105`` (``COCOImageDataset``): ids from ``test_bench/id_list.npy``;
    per id ``GT_3500/<id:012d>_GT.png``, ``Ref_3500/<id:012d>_ref.png`` (resized to 224x224, CLIP
    normalisation), ``Mask_bbox_3500/<id:012d>_mask.png`` (``mask = 1 - L/255``, NOT thresholded,
    unlike scripts/inference.py); ``inpaint = image * mask``.
  * ``scripts/inference_test_bench.py:295-402``: batches of ``n_samples`` with ``drop_last=True``,
    ``uc = learnable_vector.repeat(B,1,1)``, the CLIP -> proj_out -> VAE encode -> mask resize ->
    sampler -> decode -> clamp call order, and the output tree
    ``outdir/{results/<id>.png, grid/grid-<id>.png, samples/<id>_{mask,GT,inpaint,ref}.png}``.

MI355X-first differences (results identical per id):
  * one process per GPU; full batches are dealt round-robin to ranks (batch b -> rank b % world), no
    collective on the data path; the noise of batch b is seeded by (seed, b) on the CPU, so the images
    do not depend on how many ranks ran the sweep,
  * PNG decode / encode run on a small host thread pool one batch ahead of / behind the GPU,
  * normalisation, masking and the uint8 pack run on the GPU (the batch travels as uint8 both ways).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from PIL import Image

from .preprocess import CLIP_MEAN, CLIP_STD


def id_stem(i: int) -> str:
    return str(int(i)).zfill(12)


class COCOImageDataset:
    """Reader with the reference's item layout: (image [3,H,W] in [-1,1],
    {'inpaint_image' [3,H,W], 'inpaint_mask' [1,H,W], 'ref_imgs' [1,3,224,224]}, '<id:012d>')."""

    def __init__(self, test_bench_dir: str, id_list: Optional[Sequence[int]] = None):
        self.test_bench_dir = test_bench_dir
        if id_list is None:
            for cand in (os.path.join(test_bench_dir, "id_list.npy"), os.path.join("test_bench", "id_list.npy")):
                if os.path.exists(cand):
                    id_list = np.load(cand, allow_pickle=False).tolist()
                    break
            else:
                raise FileNotFoundError(f"id_list.npy not found under {test_bench_dir!r} or ./test_bench")
        self.id_list = [int(i) for i in id_list]
        self.length = len(self.id_list)

    def paths(self, index: int) -> Tuple[str, str, str]:
        s = id_stem(self.id_list[index])
        d = self.test_bench_dir
        return (os.path.join(d, "GT_3500", s + "_GT.png"), os.path.join(d, "Ref_3500", s + "_ref.png"),
                os.path.join(d, "Mask_bbox_3500", s + "_mask.png"))

    def load_uint8(self, index: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray, str]:
        """Decoded pixels only (what travels to the GPU): image HWC u8, ref 224x224x3 u8, mask HW u8."""
        gt, ref, msk = self.paths(index)
        img = np.asarray(Image.open(gt).convert("RGB"), dtype=np.uint8)
        ref_img = np.asarray(Image.open(ref).resize((224, 224)).convert("RGB"), dtype=np.uint8)     # resize THEN convert, as the reference
        mask = np.asarray(Image.open(msk).convert("L"), dtype=np.uint8)
        return img, ref_img, mask, id_stem(self.id_list[index])

    def __getitem__(self, index: int):
        img, ref_img, mask, stem = self.load_uint8(index)
        image_tensor = (torch.from_numpy(img.astype(np.float32) / 255.0).permute(2, 0, 1) - 0.5) / 0.5
        ref = torch.from_numpy(ref_img.astype(np.float32) / 255.0).permute(2, 0, 1)
        ref = (ref - torch.tensor(CLIP_MEAN)[:, None, None]) / torch.tensor(CLIP_STD)[:, None, None]
        mask_tensor = 1 - torch.from_numpy(mask.astype(np.float32) / 255.0)[None]
        return image_tensor, {"inpaint_image": image_tensor * mask_tensor, "inpaint_mask": mask_tensor, "ref_imgs": ref[None]}, stem

    def __len__(self) -> int:
        return self.length


def rank_batches(n_items: int, batch_size: int, rank: int = 0, world: int = 1) -> List[Tuple[int, List[int]]]:
    """[(global batch index, item indices)] this rank runs: full batches only (DataLoader drop_last=True,
    inference_test_bench.py:301), dealt round-robin so every rank gets the same count +-1."""
    nb = n_items // batch_size
    return [(b, list(range(b * batch_size, (b + 1) * batch_size))) for b in range(nb) if b % world == rank]


def batch_noise(seed: int, batch_index: int, B: int, C: int, h: int, w: int) -> Dict[str, torch.Tensor]:
    """Start code and posterior noise of one batch, from a CPU generator keyed by (seed, batch index)."""
    g = torch.Generator(device="cpu").manual_seed(int(seed) * 1000003 + int(batch_index))
    return {"x_T": torch.randn(B, C, h, w, generator=g), "post_eps": torch.randn(B, C, h, w, generator=g)}


def make_grid(tiles: Sequence[torch.Tensor], pad: int = 2) -> torch.Tensor:
    """torchvision.utils.make_grid(stack(tiles)) for <= 8 equal tiles: one row, 2-px zero padding."""
    H, W = tiles[0].shape[-2:]
    g = torch.zeros(3, H + 2 * pad, len(tiles) * (W + pad) + pad)
    for i, t in enumerate(tiles):
        g[:, pad:pad + H, pad + i * (W + pad):pad + i * (W + pad) + W] = t
    return g


def _png(a: np.ndarray, path: str) -> None:
    Image.fromarray(a).save(path)


def _u8(t: torch.Tensor) -> np.ndarray:
    """255 * CHW float -> HWC uint8 by truncation (numpy astype), like the reference."""
    return (255.0 * t.permute(1, 2, 0).numpy()).astype(np.uint8)


def write_item(outdir: str, stem: str, image_u8: np.ndarray, mask_u8: np.ndarray, ref_u8: np.ndarray, result_u8: np.ndarray,
               skip_save: bool = False) -> Dict[str, str]:
    """Files of inference_test_bench.py:361-397 for one id.  Inputs are uint8 pixels (image HWC, mask HW, ref 224x224x3,
    result HWC); the float round trips of the reference are reproduced where they change bytes:
    mask file = 255 * un_norm(mask) (the reference un-normalises the 0..1 mask too: values land in 127..255),
    ref file = bilinear 512-resize of the CLIP-normalised exemplar, de-normalised."""
    paths = {"result": os.path.join(outdir, "results", stem + ".png"), "grid": os.path.join(outdir, "grid", "grid-" + stem + ".png"),
             "mask": os.path.join(outdir, "samples", stem + "_mask.png"), "gt": os.path.join(outdir, "samples", stem + "_GT.png"),
             "inpaint": os.path.join(outdir, "samples", stem + "_inpaint.png"), "ref": os.path.join(outdir, "samples", stem + "_ref.png")}
    if skip_save:
        return {}
    H, W = image_u8.shape[:2]
    img = torch.from_numpy(image_u8.astype(np.float32) / 255.0).permute(2, 0, 1)                 # == un_norm(normalised image)
    mask = 1 - torch.from_numpy(mask_u8.astype(np.float32) / 255.0)[None]
    inpaint = (((img - 0.5) / 0.5) * mask + 1.0) / 2.0                                            # un_norm(image * mask)
    ref = torch.from_numpy(ref_u8.astype(np.float32) / 255.0).permute(2, 0, 1)
    ref_n = (ref - torch.tensor(CLIP_MEAN)[:, None, None]) / torch.tensor(CLIP_STD)[:, None, None]
    ref_big = torch.nn.functional.interpolate(ref_n[None], size=(H, W), mode="bilinear", align_corners=False, antialias=True)[0]
    ref_big = ref_big * torch.tensor(CLIP_STD)[:, None, None] + torch.tensor(CLIP_MEAN)[:, None, None]
    res = torch.from_numpy(result_u8.astype(np.float32) / 255.0).permute(2, 0, 1)
    _png(_u8(make_grid([img, inpaint, ref_big, res]).clamp(0, 1)), paths["grid"])
    _png(result_u8, paths["result"])
    _png(_u8(((mask + 1.0) / 2.0).expand(3, -1, -1)), paths["mask"])
    _png(image_u8, paths["gt"])
    _png(_u8(inpaint.clamp(0, 1)), paths["inpaint"])
    _png(_u8(ref_big.clamp(0, 1)), paths["ref"])
    return paths


def device_preprocess(img_u8: torch.Tensor, ref_u8: torch.Tensor, mask_u8: torch.Tensor) -> Dict[str, torch.Tensor]:
    """uint8 batches already on the GPU ([B,H,W,3], [B,224,224,3], [B,H,W]) -> the float tensors of the reference's
    dataset item, batched: image [-1,1], mask = 1 - L/255, ref CLIP-normalised (test_bench_dataset.py:74-99), by the HIP kernel
    pbe_u8_to_planes_f32 (same float operations in the same order as ToTensor + Normalize)."""
    from . import ops
    return {"image": ops.u8_to_planes(img_u8, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)), "mask": ops.u8_to_planes(mask_u8, mask_mode=2),
            "ref": ops.u8_to_planes(ref_u8, CLIP_MEAN, CLIP_STD)}


def device_pack_u8(img01: torch.Tensor) -> torch.Tensor:
    """[B,3,H,W] float in [0,1] on the GPU -> [B,H,W,3] uint8 (truncation, like ``(255 * x).astype(uint8)``): pbe_planes_to_u8_canvas."""
    from . import ops
    B, _, H, W = img01.shape
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=img01.device)
    for b in range(B):
        ops.planes_to_canvas(img01[b].float(), out[b], 0, 0)
    return out


def iter_loaded(ds: COCOImageDataset, batches, pool: cf.ThreadPoolExecutor) -> Iterator[Tuple[int, List[tuple]]]:
    """Decode batch b+1 on the pool while batch b is on the GPU."""
    def submit(item):
        return [pool.submit(ds.load_uint8, i) for i in item[1]]
    pending = submit(batches[0]) if batches else None
    for k, (b, _) in enumerate(batches):
        cur = pending
        pending = submit(batches[k + 1]) if k + 1 < len(batches) else None
        yield b, [f.result() for f in cur]


@torch.no_grad()
def run_sweep(model, ds: COCOImageDataset, outdir: str, *, batch_size: int, steps: int = 50, scale: float = 1.0, plms: bool = False,
              fixed_code: bool = False, seed: int = 42, rank: int = 0, world: int = 1, skip_save: bool = False, C: int = 4, f: int = 8,
              antialias: bool = True, max_batches: Optional[int] = None, io_threads: int = 4) -> Dict[str, object]:
    """The loop of inference_test_bench.py:316-400 for this rank's share of the id list.  Returns counters."""
    from .pipeline import inpaint
    for d in ("samples", "results", "grid"):
        os.makedirs(os.path.join(outdir, d), exist_ok=True)
    batches = rank_batches(len(ds), batch_size, rank, world)
    if max_batches is not None:
        batches = batches[:max_batches]
    dev = model.device
    done: List[str] = []
    writes: List[cf.Future] = []
    start_code = None
    with cf.ThreadPoolExecutor(max_workers=max(1, io_threads)) as pool:
        for b, items in iter_loaded(ds, batches, pool):
            img_u8 = torch.from_numpy(np.stack([it[0] for it in items])).to(dev, non_blocking=True)
            ref_u8 = torch.from_numpy(np.stack([it[1] for it in items])).to(dev, non_blocking=True)
            msk_u8 = torch.from_numpy(np.stack([it[2] for it in items])).to(dev, non_blocking=True)
            t = device_preprocess(img_u8, ref_u8, msk_u8)
            H, W = t["image"].shape[-2:]
            noise = batch_noise(seed, b, batch_size, C, H // f, W // f)
            if fixed_code:                                        # one start code for every batch (inference_test_bench.py:311-313)
                if start_code is None:
                    start_code = batch_noise(seed, -1, batch_size, C, H // f, W // f)["x_T"]
                noise["x_T"] = start_code
            out = inpaint(model, t["image"], t["mask"], t["ref"], steps=steps, scale=scale, x_T=noise["x_T"].to(dev),
                          post_eps=noise["post_eps"].to(dev), sampler="plms" if plms else "ddim", antialias=antialias)
            res_u8 = device_pack_u8(out["image"]).cpu().numpy()
            for i, it in enumerate(items):
                writes.append(pool.submit(write_item, outdir, it[3], it[0], it[2], it[1], res_u8[i], skip_save))
                done.append(it[3])
        for w in writes:
            w.result()
    return {"rank": rank, "world": world, "batches": len(batches), "ids": done}
