"""Seeded inputs and reduced configs shared by oracle/gen_golden.py and the tests.

Everything here is derived from CPU ``torch.Generator`` seeds (never a device RNG: device
Philox streams are not portable), so the golden generator in the build
container and the GPU box see bit-identical inputs without committing them.
"""
from __future__ import annotations

import torch

# Narrow configs keep the v1 topology (same channel_mult / attention_resolutions / heads /
# context_dim, configs/v1.yaml:30-69) at widths that run in seconds on a CPU.
UNET_NARROW = dict(in_channels=9, out_channels=4, model_channels=64, attention_resolutions=(4, 2, 1),
                   num_res_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768)
VAE_NARROW = dict(ch=32, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4, in_channels=3, out_ch=3, embed_dim=4)
CLIP_NARROW = dict(hidden=128, heads=4, layers=2, mlp=512, patch=14, image=224, eps=1e-5)
MAPPER_NARROW = dict(width=128, layers=2)

POSTERIOR_SEED = 77
PLMS_RECORD = (0, 1, 2, 3, 25, 49)


def _box_mask(b: int, h: int, w: int, g: torch.Generator) -> torch.Tensor:
    """1 outside a random axis-aligned box covering 10-40 % of the frame, 0 inside."""
    m = torch.ones(b, 1, h, w)
    for i in range(b):
        frac = 0.10 + 0.30 * torch.rand((), generator=g).item()
        ar = 0.5 + torch.rand((), generator=g).item()
        bh = min(h, max(1, int(round((frac * h * w * ar) ** 0.5))))
        bw = min(w, max(1, int(round(frac * h * w / bh))))
        y0 = int(torch.randint(0, h - bh + 1, (), generator=g).item())
        x0 = int(torch.randint(0, w - bw + 1, (), generator=g).item())
        m[i, :, y0:y0 + bh, x0:x0 + bw] = 0.0
    return m


def synthetic_triples(b: int, hw: int, first_index: int = 0, latent_div: int = 8):
    """The synthetic workload of SURVEY.md §8d: per sample i a CPU generator seeded 1234 + i."""
    img, msk, ref, xT, eps = [], [], [], [], []
    for i in range(first_index, first_index + b):
        g = torch.Generator().manual_seed(1234 + i)
        img.append(torch.rand(1, 3, hw, hw, generator=g) * 2 - 1)
        msk.append(_box_mask(1, hw, hw, g))
        ref.append(torch.randn(1, 3, 224, 224, generator=g))
        xT.append(torch.randn(1, 4, hw // latent_div, hw // latent_div, generator=g))
        eps.append(torch.randn(1, 4, hw // latent_div, hw // latent_div, generator=g))
    cat = torch.cat
    return {"image": cat(img), "mask": cat(msk), "ref": cat(ref), "x_T": cat(xT), "post_eps": cat(eps)}


def narrow_inputs():
    """Inputs of tests/golden/narrow.npz: 2 samples, 128x128 image -> 16x16 latent."""
    d = synthetic_triples(2, 128)
    torch.manual_seed(POSTERIOR_SEED)                     # what DiagonalGaussianDistribution.sample() draws
    d["post_eps"] = torch.randn(2, 4, 16, 16)
    g = torch.Generator().manual_seed(2024)
    d["unet_x"] = torch.randn(4, 9, 16, 16, generator=g)
    d["unet_t"] = torch.tensor([981, 981, 981, 981], dtype=torch.int64)
    d["unet_ctx"] = torch.randn(4, 1, 768, generator=g)
    return d


def full_inputs():
    """Inputs of tests/golden/full.npz (configs/v1.yaml sizes)."""
    g = torch.Generator().manual_seed(4242)
    d = {"unet_x": torch.randn(2, 9, 64, 64, generator=g),
         "unet_t": torch.tensor([501, 501], dtype=torch.int64),
         "unet_ctx": torch.randn(2, 1, 768, generator=g),
         "image": torch.rand(1, 3, 512, 512, generator=g) * 2 - 1,
         "z_dec": torch.randn(1, 4, 64, 64, generator=g) * 0.18215 * 4.0,
         "ref": torch.randn(1, 3, 224, 224, generator=g)}
    return d


def full_plms_inputs():
    """Inputs of tests/golden/full_plms.npz: ONE sample at configs/v1.yaml size, 4 PLMS steps (5 U-Net calls under guidance)."""
    g = torch.Generator().manual_seed(777)
    m = torch.ones(1, 1, 64, 64)
    m[:, :, 20:44, 12:40] = 0.0
    return {"x_T": torch.randn(1, 4, 64, 64, generator=g), "z_inpaint": torch.randn(1, 4, 64, 64, generator=g) * 0.8, "mask_lat": m,
            "c": torch.randn(1, 1, 768, generator=g), "uc": torch.randn(1, 1, 768, generator=g), "steps": 4, "scale": 5.0}


FULL_PLMS50_RECORD = (0, 3, 25, 49)


def full_plms50_inputs():
    """Inputs of tests/golden/full_plms50.npz: ONE sample at configs/v1.yaml size over the HEADLINE trajectory length - 50 PLMS steps,
    guidance 5 (51 U-Net calls on a guidance pair, plms.py:118-248); x recorded after steps FULL_PLMS50_RECORD."""
    g = torch.Generator().manual_seed(778)
    m = torch.ones(1, 1, 64, 64)
    m[:, :, 14:50, 18:46] = 0.0
    return {"x_T": torch.randn(1, 4, 64, 64, generator=g), "z_inpaint": torch.randn(1, 4, 64, 64, generator=g) * 0.8, "mask_lat": m,
            "c": torch.randn(1, 1, 768, generator=g), "uc": torch.randn(1, 1, 768, generator=g), "steps": 50, "scale": 5.0}


def sampler_option_inputs():
    """Inputs of tests/golden/sampler_options.npz (narrow U-Net, 16x16 latents, 2 samples): the reference sampler options that draw
    noise inside the loop - DDIM eta > 0 (ddim.py:226-238), mask / x0 blending through q_sample (plms.py:150-153, ddim.py:178-181) - and the
    timesteps= prefix (plms.py:132-139), with the noise INJECTED (the reference draws it from the device RNG)."""
    d = narrow_inputs()
    g = torch.Generator().manual_seed(9091)
    out = {"x_T": d["x_T"], "c": torch.randn(2, 1, 768, generator=g), "uc": torch.randn(2, 1, 768, generator=g),
           "z_inpaint": torch.randn(2, 4, 16, 16, generator=g) * 0.8, "x0": torch.randn(2, 4, 16, 16, generator=g)}
    m = torch.ones(2, 1, 16, 16)
    m[:, :, 4:12, 3:11] = 0.0
    out["mask_lat"] = m
    bm = torch.zeros(2, 1, 16, 16)
    bm[:, :, :, :6] = 1.0                                   # keep the known x0 on the left third
    out["blend_mask"] = bm
    out["noises"] = [torch.randn(2, 4, 16, 16, generator=g) for _ in range(12)]
    return out
