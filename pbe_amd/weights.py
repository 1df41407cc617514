"""Deterministic name-seeded weights and checkpoint key handling.

No pretrained weights exist in the build container or on the GPU box, so
parity and benchmarks run on weights synthesised *by tensor name*: every side (the golden
generator that drives the reference modules, the oracle, the HIP path) calls
``synth_tensor(name, shape)`` and gets bit-identical fp32 values.

Rules (SURVEY.md §8c "Golden-vector plan"):
  * >=2-D tensors: N(0, 1/fan_in), fan_in = prod(shape[1:])
  * 1-D ``*.weight`` (norm gains): 1 + 0.1 N(0,1)
  * 1-D ``*.bias``: 0.02 N(0,1)
  * the reference's ``zero_module`` tensors (openaimodel.py:229-231, attention.py:281-285,
    openaimodel.py:827) get the same non-zero rule as everything else, otherwise a freshly
    built U-Net returns exactly 0 and parity would be vacuous.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Mapping, Tuple

import numpy as np
import torch


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(name.encode("utf-8")), seed & 0xFFFFFFFF])


def synth_tensor(name: str, shape: Iterable[int], seed: int = 0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    g = _rng(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "position_ids":
        return torch.arange(shape[-1], dtype=torch.int64).reshape(shape)
    if leaf == "learnable_vector":           # latent_diffusion.py:111  randn((1,1,768))
        a = g.standard_normal(shape, dtype=np.float32)
    elif leaf == "class_embedding":
        a = 0.02 * g.standard_normal(shape, dtype=np.float32)
    elif len(shape) <= 1:
        n = g.standard_normal(shape, dtype=np.float32)
        a = 1.0 + 0.1 * n if leaf == "weight" else 0.02 * n
    else:
        fan_in = int(np.prod(shape[1:]))
        a = g.standard_normal(shape, dtype=np.float32) * np.float32(1.0 / np.sqrt(fan_in))
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def synth_state_dict(named_shapes: Mapping[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(k, s, seed) for k, s in named_shapes.items()}


def fill_module_(module: torch.nn.Module, seed: int = 0, prefix: str = "") -> None:
    """Overwrite every parameter/buffer of ``module`` that appears in its state_dict with the
    name-seeded value (name = prefix + state_dict key)."""
    sd = module.state_dict()
    new = {k: synth_tensor(prefix + k, v.shape, seed).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)


SCHEDULE_BUFFERS = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                    "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2", "logvar")


def fill_latent_diffusion_(model: torch.nn.Module, seed: int = 0) -> None:
    """Name-seeded weights for a whole LatentDiffusion (schedule buffers are left as computed)."""
    sd = model.state_dict()
    new = {k: synth_tensor(k, v.shape, seed).to(v.dtype) for k, v in sd.items() if k not in SCHEDULE_BUFFERS}
    torch.nn.Module.load_state_dict(model, new, strict=False)
    if hasattr(model, "invalidate_packs"):
        model.invalidate_packs()
    for m in model.modules():
        if hasattr(m, "invalidate_packs"):
            m.invalidate_packs()


# --- checkpoint key handling (scripts/inference.py:58-75, ddpm.py:245-260) -------------------

_CLIP_OLD = "cond_stage_model.transformer.vision_model."
_CLIP_NEW = "cond_stage_model.transformer."


def canonical_checkpoint_keys(sd: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Return ``sd`` with keys in the layout the build's modules use (= the layout of the
    published Paint-by-Example checkpoint, transformers 4.19 naming).

    * transformers >= 5 drops the ``vision_model.`` level of ``CLIPVisionModel``; map it back.
    * EMA shadow weights (``model_ema.*``) are ignored: ``use_ema: False`` (configs/v1.yaml:19).
    """
    out: Dict[str, torch.Tensor] = {}
    has_old = any(k.startswith(_CLIP_OLD) for k in sd)
    for k, v in sd.items():
        if k.startswith("model_ema."):
            continue
        if not has_old and k.startswith(_CLIP_NEW):
            k = _CLIP_OLD + k[len(_CLIP_NEW):]
        out[k] = v
    return out
