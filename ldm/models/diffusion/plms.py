"""PLMS (pseudo linear multistep) sampler for Paint-by-Example on MI355X — drop-in for
ldm/models/diffusion/plms.py:11-248 in zhanwenchen/pbe: ``PLMSSampler(model).sample(S, batch_size,
shape, conditioning, ..., unconditional_guidance_scale, unconditional_conditioning, x_T,
test_model_kwargs)`` -> ``(samples, {'x_inter', 'pred_x0'})``.

What differs from the reference loop (same arithmetic):
  * the schedule scalars never go to the device as tensors (the reference uploads four
    ``torch.full`` scalars per update, plms.py:204-207): they are kernel arguments,
  * per step ONE kernel builds the CFG-doubled 9-channel NHWC input (plms.py:185,225) and ONE
    kernel does the guidance combine, the Adams-Bashforth blend and the x_prev / pred_x0 update
    (plms.py:188-189, 202-219, 230-244); the state x stays fp32 NCHW like the reference's,
  * the U-Net is entered through its NHWC fast path, and the [uc | c] context is built once so
    the 16 cross-attention constants are computed once per run,
  * ``test_model_kwargs`` may use either key spelling (``inpaint_image``/``inpaint_mask`` from
    scripts/inference.py:320-332 or ``images_inpaint``/``images_mask`` from plms.py:221-222).
  * ``mask`` / ``x0`` (plms.py:150-153: img = q_sample(x0, ts) * mask + (1 - mask) * img before every step) and ``timesteps=``
    (plms.py:132-139: a prefix of the schedule) run as in the reference; the q_sample noise comes from ``self.noise_like``
    (default: the torch device generator, like the reference's randn_like - tests inject it).
Unsupported reference options (score_corrector, quantize_denoised, eta != 0 - the reference's PLMS asserts that too -,
noise_dropout, ddim_use_original_steps) raise instead of being silently ignored.
"""
import math

import numpy as np
import torch

from pbe_amd import ops
from pbe_amd.graph import GraphedUNet, graphs_enabled
from pbe_amd.lib import PbeError
from ldm.modules.diffusionmodules.util import make_ddim_sampling_parameters, make_ddim_timesteps

_AB = {0: (1.0,), 1: (3 / 2, -1 / 2), 2: (23 / 12, -16 / 12, 5 / 12), 3: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}


def inpaint_kwargs(kwargs):
    tmk = kwargs.get("test_model_kwargs")
    if tmk is None:
        raise PbeError("sample(): test_model_kwargs with the inpainting latent and mask is required (plms.py:220-222)")
    z = tmk.get("images_inpaint", tmk.get("inpaint_image"))
    m = tmk.get("images_mask", tmk.get("inpaint_mask"))
    if z is None or m is None:
        raise PbeError("test_model_kwargs needs images_inpaint/images_mask (or inpaint_image/inpaint_mask)")
    return z, m


class PLMSSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.use_graph = None               # None = decide per run (pbe_amd.graph.graphs_enabled); True / False force
        self._graphed = None
        self.share_guidance_prefix = True   # evaluate the context-independent prefix of a guidance pair once (UNetModel.forward_nhwc paired=True)
        self.noise_like = lambda shape, device: torch.randn(shape, device=device)      # util.py:264-267 (q_sample / DDIM eta > 0 noise)
        self.require_gpu = True       # host-logic tests clear this and substitute the two element-wise kernels; the kernels themselves have no CPU path

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        if ddim_eta != 0:
            raise ValueError("ddim_eta must be 0 for PLMS")
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        ac = self.model.alphas_cumprod.detach().float().cpu()
        assert ac.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        self.register_buffer("betas", self.model.betas)
        self.register_buffer("alphas_cumprod", self.model.alphas_cumprod)
        self.register_buffer("alphas_cumprod_prev", self.model.alphas_cumprod_prev)
        sig, a, a_prev = make_ddim_sampling_parameters(alphacums=ac.numpy(), ddim_timesteps=self.ddim_timesteps, eta=ddim_eta, verbose=verbose)
        self.register_buffer("ddim_sigmas", sig)
        self.register_buffer("ddim_alphas", a)                      # float32 numpy, selected from the fp32 buffer like the reference
        self.register_buffer("ddim_alphas_prev", a_prev)
        self.register_buffer("ddim_sqrt_one_minus_alphas", np.sqrt(1. - a))

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None, quantize_x0=False,
               eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None, verbose=True,
               x_T=None, log_every_t=100, unconditional_guidance_scale=1., unconditional_conditioning=None, **kwargs):
        if conditioning is None:
            raise PbeError("PLMSSampler.sample: conditioning is required")
        if conditioning.shape[0] != batch_size:
            print(f"Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}")
        if quantize_x0 or score_corrector is not None or noise_dropout != 0.:
            raise PbeError("PLMSSampler: quantize_x0 / score_corrector / noise_dropout are not on the Paint-by-Example path")
        if (mask is None) != (x0 is None):
            raise PbeError("PLMSSampler: mask and x0 go together (plms.py:150-153)")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        return self.plms_sampling(conditioning, (batch_size, C, H, W), callback=callback, img_callback=img_callback, x_T=x_T,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, mask=mask, x0=x0, **kwargs)

    # ---- one U-Net evaluation with guidance (plms.py:181-195) -----------------------------------
    def _eps(self, x, step, ctx, z_inp, msk, dup):
        b = x.shape[0]
        t = torch.full((dup * b,), int(step), device=x.device, dtype=torch.int64)
        unet = self.model.model.diffusion_model
        # cat([x]*2), cat([t]*2) (plms.py:183-184): both halves share x and t; the U-Net evaluates the common prefix once
        paired = dup == 2 and self.share_guidance_prefix
        x9 = ops.plms_pack_input(x, z_inp, msk, 1 if paired else dup)
        if (graphs_enabled(dup * b) if self.use_graph is None else self.use_graph) and x.is_cuda:      # launch-bound regime: one HIP graph per call
            if self._graphed is None or self._graphed.unet is not unet:
                self._graphed = GraphedUNet(unet)
            return self._graphed(x9, t, ctx, paired)
        return unet.forward_nhwc(x9, t, ctx, paired=paired, step=int(step))      # every row of t is `step`: the embedding rows come from the per-value cache

    def _coef(self, index, weights):
        a_t, a_prev = float(self.ddim_alphas[index]), float(self.ddim_alphas_prev[index])
        w = list(weights) + [0.0] * (4 - len(weights))
        return w + [float(self.ddim_sqrt_one_minus_alphas[index]), 1.0 / math.sqrt(a_t), math.sqrt(a_prev), math.sqrt(1.0 - a_prev)]

    def _schedule_subset(self, timesteps):
        """plms.py:132-139 / ddim.py:151-155: `timesteps` = how many of the schedule's steps to run (a prefix of ddim_timesteps)."""
        if timesteps is None:
            return self.ddim_timesteps
        n = self.ddim_timesteps.shape[0]
        subset_end = int(min(timesteps / n, 1) * n) - 1
        return self.ddim_timesteps[:subset_end]

    def _blend_known(self, img, x0, mask, step):
        """img_orig = q_sample(x0, ts); img = img_orig * mask + (1 - mask) * img (plms.py:150-153), one kernel."""
        ac = float(self.model.alphas_cumprod[int(step)])
        return ops.qsample_blend(x0, self.noise_like(tuple(x0.shape), x0.device).float(), mask, img, math.sqrt(ac), math.sqrt(1.0 - ac))

    @torch.no_grad()
    def plms_sampling(self, cond, shape, x_T=None, callback=None, timesteps=None, img_callback=None, log_every_t=100,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, mask=None, x0=None, **kwargs):
        device = self.model.betas.device
        if self.require_gpu and device.type != "cuda":
            raise PbeError("PLMSSampler: the model must live on an MI355X (model.to('cuda')); there is no CPU path")
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        z_inp, msk = inpaint_kwargs(kwargs)
        z_inp = z_inp.to(device=device, dtype=torch.float32).contiguous()
        msk = msk.to(device=device, dtype=torch.float32).contiguous()
        cond = cond.to(device)
        guided = not (unconditional_conditioning is None or unconditional_guidance_scale == 1.)
        if guided:
            uc = unconditional_conditioning.to(device)
            if uc.shape[0] != b:                       # scripts/inference.py:325 hands [1,1,768]; the test bench repeats it
                uc = uc.expand(b, *uc.shape[1:])
            ctx = torch.cat((uc.to(torch.float16), cond.to(torch.float16))).contiguous()
        else:
            ctx = cond.to(torch.float16).contiguous()
        dup = 2 if guided else 1
        scale = float(unconditional_guidance_scale)

        time_range = np.flip(self._schedule_subset(timesteps))
        total = time_range.shape[0]
        if mask is not None:
            mask, x0 = mask.to(device=device, dtype=torch.float32), x0.to(device=device, dtype=torch.float32)
        inter = {"x_inter": [img], "pred_x0": [img]}
        old = []                                                      # newest last, at most 3 (plms.py:163-165)
        for i, step in enumerate(time_range):
            index = total - i - 1
            step_next = time_range[min(i + 1, total - 1)]
            if mask is not None:
                img = self._blend_known(img, x0, mask, step)
            eps = self._eps(img, step, ctx, z_inp, msk, dup)
            if len(old) == 0:
                # pseudo improved Euler (plms.py:230-235): probe x_prev with e_t, re-evaluate at t_next, average
                x_probe, _, e_t = ops.plms_update(eps, dup, scale, img, [], self._coef(index, _AB[0]), want_pred=False)
                eps2 = self._eps(x_probe, step_next, ctx, z_inp, msk, dup)
                # e' = (e_t + e_next)/2 : c0 weights the fresh eps (= e_next), history slot 1 = e_t
                img, pred_x0, _ = ops.plms_update(eps2, dup, scale, img, [e_t], self._coef(index, (0.5, 0.5)), want_e_t=False)
            else:
                hist = old[::-1]
                img, pred_x0, e_t = ops.plms_update(eps, dup, scale, img, hist, self._coef(index, _AB[len(hist)]))
            old.append(e_t)
            if len(old) >= 4:
                old.pop(0)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total - 1:
                inter["x_inter"].append(img)
                inter["pred_x0"].append(pred_x0)
        return img, inter
