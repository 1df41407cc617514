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


def load_triple_u8(image_path: str, mask_path: str, reference_path: str) -> Dict[str, np.ndarray]:
    """Decode only (what travels to the GPU): image HWC u8, mask HW u8, exemplar PIL-resized to 224x224 HWC u8."""
    # np.array (owned, writable copies): np.asarray of a PIL image is a read-only view, which torch.from_numpy must not be handed
    return {"image": np.array(Image.open(image_path).convert("RGB"), dtype=np.uint8),
            "mask": np.array(Image.open(mask_path).convert("L"), dtype=np.uint8),
            "ref": np.array(Image.open(reference_path).convert("RGB").resize((224, 224)), dtype=np.uint8)}


def load_triple_device(image_path: str, mask_path: str, reference_path: str, device) -> Dict[str, torch.Tensor]:
    """load_triple with the arithmetic of scripts/inference.py:306-319 on the GPU (pbe_u8_to_planes_f32 / pbe_mul_planes_f32):
    the triple travels as uint8; the returned fp32 tensors live on `device` and equal load_triple()'s bit for bit."""
    from . import ops
    u8 = load_triple_u8(image_path, mask_path, reference_path)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device)[None] for k, v in u8.items()}
    img = ops.u8_to_planes(dev["image"], (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    ref = ops.u8_to_planes(dev["ref"], CLIP_MEAN, CLIP_STD)
    mask = ops.u8_to_planes(dev["mask"], mask_mode=1)
    return {"image": img, "mask": mask, "inpaint": ops.mul_planes(img, mask), "ref": ref}


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


def save_outputs_device(outdir: str, stem: str, seed: int, triple: Dict[str, torch.Tensor], result: torch.Tensor, H: int, W: int) -> Dict[str, str]:
    """save_outputs with the pixels composed on the GPU: the exemplar's bilinear resize, every
    un-normalisation, clamp, the uint8 pack and the 4-tile grid are HIP kernels (pbe_resize_bilinear_f32, pbe_planes_to_u8_canvas);
    one uint8 canvas comes back and the host only PNG-encodes.  Byte-identical files to save_outputs()."""
    from . import ops
    src, res, grid = (os.path.join(outdir, d) for d in ("source", "results", "grid"))
    for d in (src, res, grid):
        os.makedirs(d, exist_ok=True)
    dev = result.device
    image, inpaint, ref, mask = (triple[k][0].float() for k in ("image", "inpaint", "ref", "mask"))
    ref_big = ops.resize_bilinear(ref[None], (H, W), True)[0]
    pad = 2
    # one canvas: the grid on top, then the five single images side by side below it
    Wg = 4 * (W + pad) + pad
    canvas = torch.zeros((H + 2 * pad + H, max(Wg, 5 * W), 3), dtype=torch.uint8, device=dev)
    half, clip_a, clip_b = ((0.5,) * 3, (0.5,) * 3), CLIP_STD, CLIP_MEAN
    tiles = [(image, *half), (inpaint, *half), (ref_big, clip_a, clip_b), (result.float(), (1.0,) * 3, (0.0,) * 3)]
    for i, (t, a, b) in enumerate(tiles):
        ops.planes_to_canvas(t, canvas, pad, pad + i * (W + pad), a, b)
    y1 = H + 2 * pad
    singles = [("result", result.float(), (1.0,) * 3, (0.0,) * 3), ("mask", mask, *half), ("gt", image, *half), ("inpaint", inpaint, *half),
               ("ref", ref_big, clip_a, clip_b)]
    for i, (_, t, a, b) in enumerate(singles):
        ops.planes_to_canvas(t, canvas, y1, i * W, a, b)
    host = canvas.cpu().numpy()
    paths = {"grid": os.path.join(grid, f"grid-{stem}_{seed}.png"), "result": os.path.join(res, f"{stem}_{seed}.png"),
             "mask": os.path.join(src, f"{stem}_{seed}_mask.png"), "gt": os.path.join(src, f"{stem}_{seed}_GT.png"),
             "inpaint": os.path.join(src, f"{stem}_{seed}_inpaint.png"), "ref": os.path.join(src, f"{stem}_{seed}_ref.png")}
    Image.fromarray(np.ascontiguousarray(host[:y1, :Wg])).save(paths["grid"])
    for i, (name, *_rest) in enumerate(singles):
        Image.fromarray(np.ascontiguousarray(host[y1:y1 + H, i * W:(i + 1) * W])).save(paths[name])
    return paths
