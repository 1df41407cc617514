"""End-to-end inpainting pass = the call order of scripts/inference.py:323-348 in zhanwenchen/pbe:
CLIP exemplar -> proj_out -> VAE encode of the masked image (posterior sample x 0.18215) ->
mask resize to the latent grid -> PLMS / DDIM sampling under classifier-free guidance ->
VAE decode -> clamp((x+1)/2, 0, 1).  Inputs are already pre-processed tensors
(pbe_amd.preprocess), outputs fp32 NCHW in [0,1]."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops


def resize_mask(mask: torch.Tensor, size, antialias: bool = True) -> torch.Tensor:
    """scripts/inference.py:332 ``Resize([h, w])(mask)``: bilinear, align_corners=False; the
    antialias default differs across torchvision versions -> exposed, default True.
    Pre-processing, once per image: the HIP kernel pbe_resize_bilinear_f32 (no torch fallback)."""
    return ops.resize_bilinear(mask.float(), size, antialias)


@torch.no_grad()
def inpaint(model, image: torch.Tensor, mask: torch.Tensor, ref: torch.Tensor, *, steps: int = 50, scale: float = 5.0,
            x_T: Optional[torch.Tensor] = None, post_eps: Optional[torch.Tensor] = None, sampler: str = "plms",
            antialias: bool = True, timings: Optional[Dict[str, float]] = None) -> Dict[str, torch.Tensor]:
    """image [B,3,H,W] in [-1,1], mask [B,1,H,W] in {0,1} (1 = keep), ref [B,3,224,224] CLIP-normalised;
    everything on the model's GPU.  Returns {'image' [B,3,H,W] in [0,1], 'latent', 'c', 'z_inpaint', 'mask_lat'}."""
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.plms import PLMSSampler
    dev = model.device
    image, mask, ref = image.to(dev).float(), mask.to(dev).float(), ref.to(dev).float()
    B = image.shape[0]
    ev = None
    if timings is not None:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
    c = model.proj_out(model.get_learned_conditioning(ref))                              # inference.py:326-327
    uc = model.learnable_vector if scale != 1.0 else None                                # inference.py:323-325
    if ev:
        ev[1].record()
    post = model.encode_first_stage(image * mask)                                        # inference.py:319,330
    z_inp = model.get_first_stage_encoding(post, noise=post_eps)                         # inference.py:331
    m_lat = resize_mask(mask, z_inp.shape[-2:], antialias)                               # inference.py:332
    if ev:
        ev[2].record()
    smp = (PLMSSampler if sampler == "plms" else DDIMSampler)(model)
    z0, _ = smp.sample(S=steps, batch_size=B, shape=list(z_inp.shape[1:]), conditioning=c, verbose=False,
                       unconditional_guidance_scale=scale, unconditional_conditioning=uc, eta=0.0, x_T=x_T,
                       test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": m_lat})
    if ev:
        ev[3].record()
    img = ops.image_post(model.decode_first_stage_nhwc(z0))                              # inference.py:346-347
    if ev:
        ev[4].record()
        torch.cuda.synchronize()
        for k, i in (("clip_ms", 0), ("vae_encode_ms", 1), ("sampler_ms", 2), ("vae_decode_ms", 3)):
            timings[k] = timings.get(k, 0.0) + ev[i].elapsed_time(ev[i + 1])
    return {"image": img, "latent": z0, "c": c, "z_inpaint": z_inp, "mask_lat": m_lat}
