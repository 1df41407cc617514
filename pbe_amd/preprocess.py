"""Host-side pre/post-processing of one (image, mask, exemplar) triple — the formulas of
scripts/inference.py:305-318 and :346-399 in zhanwenchen/pbe, with PIL + numpy only
(the reference uses torchvision ToTensor/Normalize, which are these two lines of arithmetic)."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import torch
from PIL import Image

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _to_tensor(img: Image.Image) -> torch.Tensor:
    """torchvision ToTensor: HWC uint8 -> CHW float32 / 255."""
    a = np.asarray(img, dtype=np.uint8)
    return torch.from_numpy(a.astype(np.float32) / 255.0).permute(2, 0, 1).contiguous()


def load_triple(image_path: str, mask_path: str, reference_path: str) -> Dict[str, torch.Tensor]:
    """Returns image [1,3,H,W] in [-1,1], mask [1,1,H,W] in {0,1} (1 = keep), inpaint = image*mask,
    ref [1,3,224,224] CLIP-normalised (scripts/inference.py:306-319)."""
    img = _to_tensor(Image.open(image_path).convert("RGB"))
    img = ((img - 0.5) / 0.5).unsqueeze(0)
    ref = _to_tensor(Image.open(reference_path).convert("RGB").resize((224, 224)))
    ref = ((ref - torch.tensor(CLIP_MEAN)[:, None, None]) / torch.tensor(CLIP_STD)[:, None, None]).unsqueeze(0)
    m = np.array(Image.open(mask_path).convert("L"))[None, None]
    m = 1 - m.astype(np.float32) / 255.0
    m[m < 0.5] = 0
    m[m >= 0.5] = 1
    mask = torch.from_numpy(m)
    return {"image": img, "mask": mask, "inpaint": img * mask, "ref": ref}


def un_norm(x: torch.Tensor) -> torch.Tensor:
    return (x + 1.0) / 2.0


def un_norm_clip(x: torch.Tensor) -> torch.Tensor:
    x = x.clone()
    for c in range(3):
        x[c] = x[c] * CLIP_STD[c] + CLIP_MEAN[c]
    return x


def _save(t: torch.Tensor, path: str) -> None:
    a = (255.0 * t.detach().float().cpu().permute(1, 2, 0).numpy()).astype(np.uint8)     # astype truncates, like the reference
    Image.fromarray(a).save(path)


def save_outputs(outdir: str, stem: str, seed: int, triple: Dict[str, torch.Tensor], result: torch.Tensor, H: int, W: int) -> Dict[str, str]:
    """Output tree of scripts/inference.py:289-294,378-399: outdir/{source,results,grid}; file names
    <stem>_<seed>*.png.  (No watermark, no safety checker: the reference discards the checker's
    output, scripts/inference.py:350-351, and the watermark is cosmetic.)"""
    src, res, grid = (os.path.join(outdir, d) for d in ("source", "results", "grid"))
    for d in (src, res, grid):
        os.makedirs(d, exist_ok=True)
    image, inpaint, ref, mask = triple["image"][0], triple["inpaint"][0], triple["ref"][0], triple["mask"][0]
    ref_big = torch.nn.functional.interpolate(ref[None], size=(H, W), mode="bilinear", align_corners=False, antialias=True)[0]
    tiles = [un_norm(image), un_norm(inpaint), un_norm_clip(ref_big), result]
    pad = 2
    g = torch.zeros(3, H + 2 * pad, len(tiles) * (W + pad) + pad)
    for i, t in enumerate(tiles):
        g[:, pad:pad + H, pad + i * (W + pad):pad + i * (W + pad) + W] = t.clamp(0, 1)
    paths = {"grid": os.path.join(grid, f"grid-{stem}_{seed}.png"), "result": os.path.join(res, f"{stem}_{seed}.png"),
             "mask": os.path.join(src, f"{stem}_{seed}_mask.png"), "gt": os.path.join(src, f"{stem}_{seed}_GT.png"),
             "inpaint": os.path.join(src, f"{stem}_{seed}_inpaint.png"), "ref": os.path.join(src, f"{stem}_{seed}_ref.png")}
    _save(g, paths["grid"])
    _save(result, paths["result"])
    _save(un_norm(mask).expand(3, -1, -1), paths["mask"])
    _save(un_norm(image), paths["gt"])
    _save(un_norm(inpaint), paths["inpaint"])
    _save(un_norm_clip(ref_big), paths["ref"])
    return paths
