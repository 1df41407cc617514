"""DDIM sampler for Paint-by-Example on MI355X — drop-in for ldm/models/diffusion/ddim.py:22-283 in
zhanwenchen/pbe (selected when ``--plms`` is absent, scripts/inference.py:277-280).  One U-Net
evaluation per step; the 9-channel concat (ddim.py:197-202, incl. the ``rest=`` spelling), guidance
combine and the x_prev / pred_x0 update (:222-241, pred_x0 from x[:, :4]) are the same two fused
kernels the PLMS sampler uses.  eta > 0 (scripts/inference.py:164 --ddim_eta): the direction term uses sqrt(1 - a_prev - sigma^2) and
sigma_t * noise * temperature is added by one more element-wise kernel (ddim.py:234-238); mask / x0 blending and timesteps= as in the
PLMS sampler.  The noise comes from ``self.noise_like`` (default: the torch device generator, like the reference)."""
import numpy as np
import torch

from pbe_amd import ops
from pbe_amd.lib import PbeError
from ldm.models.diffusion.plms import PLMSSampler, inpaint_kwargs


class DDIMSampler(PLMSSampler):
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        super().make_schedule(ddim_num_steps, ddim_discretize=ddim_discretize, ddim_eta=0., verbose=verbose)
        if ddim_eta != 0:                                     # ddim.py:57-62 -> util.py:63-74
            from ldm.modules.diffusionmodules.util import make_ddim_sampling_parameters
            sig, _, _ = make_ddim_sampling_parameters(alphacums=self.model.alphas_cumprod.detach().float().cpu().numpy(), ddim_timesteps=self.ddim_timesteps,
                                                      eta=ddim_eta, verbose=verbose)
            self.register_buffer("ddim_sigmas", sig)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None, quantize_x0=False,
               eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None, verbose=True,
               x_T=None, log_every_t=100, unconditional_guidance_scale=1., unconditional_conditioning=None, disable_tqdm=True, **kwargs):
        if conditioning is None:
            raise PbeError("DDIMSampler.sample: conditioning is required")
        if quantize_x0 or score_corrector is not None or noise_dropout != 0.:
            raise PbeError("DDIMSampler: quantize_x0 / score_corrector / noise_dropout are not on the Paint-by-Example path")
        if (mask is None) != (x0 is None):
            raise PbeError("DDIMSampler: mask and x0 go together (ddim.py:178-181)")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        return self.ddim_sampling(conditioning, (batch_size, C, H, W), callback=callback, img_callback=img_callback, x_T=x_T,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, mask=mask, x0=x0, temperature=temperature, **kwargs)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, callback=None, img_callback=None, log_every_t=100, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, timesteps=None, mask=None, x0=None, temperature=1., **kwargs):
        device = self.model.betas.device
        if self.require_gpu and device.type != "cuda":
            raise PbeError("DDIMSampler: the model must live on an MI355X; there is no CPU path")
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        if "rest" in kwargs and "test_model_kwargs" not in kwargs:       # ddim.py:201-202: rest = cat(z_inpaint, mask)
            rest = kwargs["rest"]
            z_inp, msk = rest[:, :4], rest[:, 4:5]
        else:
            z_inp, msk = inpaint_kwargs(kwargs)
        z_inp = z_inp.to(device=device, dtype=torch.float32).contiguous()
        msk = msk.to(device=device, dtype=torch.float32).contiguous()
        guided = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)
        if guided:
            uc = unconditional_conditioning.to(device)
            if uc.shape[0] != b:
                uc = uc.expand(b, *uc.shape[1:])
            ctx = torch.cat((uc.to(torch.float16), cond.to(device=device, dtype=torch.float16))).contiguous()
        else:
            ctx = cond.to(device=device, dtype=torch.float16).contiguous()
        dup = 2 if guided else 1
        time_range = np.flip(self._schedule_subset(timesteps))
        total = time_range.shape[0]
        if mask is not None:
            mask, x0 = mask.to(device=device, dtype=torch.float32), x0.to(device=device, dtype=torch.float32)
        inter = {"x_inter": [img], "pred_x0": [img]}
        for i, step in enumerate(time_range):
            index = total - i - 1
            if mask is not None:
                img = self._blend_known(img, x0, mask, step)
            eps = self._eps(img, step, ctx, z_inp, msk, dup)
            coef = self._coef(index, (1.0,))
            sigma = float(self.ddim_sigmas[index])
            if sigma != 0.0:                                  # dir_xt = sqrt(1 - a_prev - sigma_t^2) e_t (ddim.py:234)
                coef[7] = float(np.sqrt(1.0 - float(self.ddim_alphas_prev[index]) - sigma * sigma))
            img, pred_x0, _ = ops.plms_update(eps, dup, float(unconditional_guidance_scale), img, [], coef, want_e_t=False)
            if sigma != 0.0:                                  # + sigma_t * noise_like(...) * temperature (ddim.py:235-238)
                ops.axpy_(img, sigma * float(temperature), self.noise_like(tuple(img.shape), img.device).float().contiguous())
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total - 1:
                inter["x_inter"].append(img)
                inter["pred_x0"].append(pred_x0)
        return img, inter
