"""Schedules and small primitives of the denoising path — host-side counterparts of
ldm/modules/diffusionmodules/util.py in zhanwenchen/pbe (:21-74 schedules, :151-171 timestep
embedding, :199-216 GroupNorm32, :264-267 noise_like).  The schedule tables are float64 numpy on
the host exactly like the reference; everything that touches activations is a HIP kernel."""
import math

import numpy as np
import torch
from torch import nn

from pbe_amd import ops
from pbe_amd.hipmodule import require_gpu


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule == "linear":
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2
    if schedule == "sqrt_linear":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64)
    if schedule == "sqrt":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64) ** 0.5
    if schedule == "cosine":
        ts = np.arange(n_timestep + 1, dtype=np.float64) / n_timestep + cosine_s
        al = np.cos(ts / (1 + cosine_s) * np.pi / 2) ** 2
        al = al / al[0]
        return np.clip(1 - al[1:] / al[:-1], 0, 0.999)
    raise ValueError(f"schedule '{schedule}' unknown.")


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    if ddim_discr_method == "uniform":
        step = num_ddpm_timesteps // num_ddim_timesteps
        ts = np.arange(0, num_ddpm_timesteps, step)
    elif ddim_discr_method == "quad":
        ts = (np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    out = ts + 1          # shifted by one so the final alpha is the data-scale one
    if verbose:
        print(f"Selected timesteps for ddim sampler: {out}")
    return out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    alphacums = np.asarray(alphacums)
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in the following sigma_t schedule: {sigmas}")
    return sigmas, alphas, alphas_prev


def extract_into_tensor(a, t, x_shape):
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


def checkpoint(func, inputs, params, flag):
    """Gradient checkpointing is a pass-through on the inference path (util.py:102-128 under no_grad)."""
    return func(*inputs)


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """[N] int64 timesteps -> [N, dim] fp16 sinusoid (cos first), computed on the device."""
    require_gpu(timesteps, "timestep_embedding")
    if repeat_only:
        return timesteps[:, None].to(torch.float16).expand(-1, dim).contiguous()
    return ops.timestep_embedding(timesteps.to(torch.int64), dim, float(max_period))


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def conv_nd(dims, *args, **kwargs):
    if dims != 2:
        raise ValueError(f"unsupported dimensions: {dims} (the PLMS hot path is 2-D)")
    return nn.Conv2d(*args, **kwargs)


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


class GroupNorm32(nn.GroupNorm):
    """Parameter holder with the reference's name; NCHW ``forward`` converts at the boundary, the
    network itself calls ``ops.groupnorm`` on NHWC activations directly."""

    def forward(self, x):
        require_gpu(x, "GroupNorm32")
        y = ops.groupnorm(ops.nchw_to_nhwc(x.float()), self.weight.float(), self.bias.float(), self.eps, False, groups=self.num_groups)
        return ops.nhwc_to_nchw(y).to(x.dtype)


def normalization(channels):
    return GroupNorm32(32, channels)


def noise_like(shape, device, repeat=False):
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)
