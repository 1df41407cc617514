"""Model façade of the Paint-by-Example PLMS path on MI355X.

Exports ``DDPM``, ``DiffusionWrapper`` and ``LatentDiffusion`` under the module path
``configs/v1.yaml:3`` targets (``ldm.models.diffusion.ddpm.LatentDiffusion``); in the
zhanwenchen/pbe fork the class body sits in ldm/models/diffusion/latent_diffusion.py:85-1220 and
``ddpm.py`` holds DDPM (:87-466) and DiffusionWrapper (:468-515) — both import paths work here.

Only the inference surface is implemented: schedule buffers
(ddpm.py:175-228), ``ema_scope`` (no-op, use_ema False), ``get_learned_conditioning``
(latent_diffusion.py:264-276), ``proj_out`` / ``learnable_vector`` (:111-112), ``encode_first_stage``
/ ``get_first_stage_encoding`` (:571-610, :255-262), ``decode_first_stage`` (:444-508),
``apply_model`` (:646-743), ``q_sample`` (ddpm.py:337-341).  Training, logging, EMA, fold/unfold
tiling are out of scope and raise.
"""
from contextlib import contextmanager

import numpy as np
import torch
from torch import nn

from pbe_amd import ops
from pbe_amd.hipmodule import HipModule, f32, require_gpu
from pbe_amd.lib import PbeError
from ldm.modules.diffusionmodules.util import extract_into_tensor, make_beta_schedule
from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
from ldm.util import count_params, default, instantiate_from_config


def disabled_train(self, mode=True):
    return self


class HipLinear(nn.Linear):
    """``nn.Linear`` under the reference's parameter names whose forward is the HIP GEMM (fp16 operands, fp32 accumulate, bias in
    the epilogue).  Used for ``proj_out`` (latent_diffusion.py:112) so that ``model.proj_out(c)`` called the reference's way
    (scripts/inference.py:327) runs the same kernel as the rest of the path instead of an ATen / rocBLAS dispatch."""

    def invalidate_packs(self):
        """Drop the fp16 pack: writes through ``.data`` (pbe_amd.shard.broadcast_weights_, weights.fill_*) do not bump ``_version``."""
        self.__dict__.pop("_pk", None)

    def _packed(self):
        key = (self.weight.data_ptr(), self.weight._version, self.bias.data_ptr(), self.bias._version)
        c = self.__dict__.get("_pk")
        if c is None or c[0] != key:
            c = (key, ops.pack_linear(self.weight), f32(self.bias))
            self.__dict__["_pk"] = c
        return c[1], c[2]

    def forward(self, z):
        require_gpu(z, "proj_out")
        w, b = self._packed()
        lead = z.shape[:-1]
        y = ops.gemm(z.to(torch.float16).reshape(-1, z.shape[-1]).contiguous(), w, b)
        return y.view(*lead, -1)


class DiffusionWrapper(nn.Module):
    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        if conditioning_key not in (None, "crossattn"):
            raise PbeError(f"DiffusionWrapper: conditioning_key={conditioning_key!r}; Paint-by-Example uses 'crossattn' (configs/v1.yaml:15)")
        self.conditioning_key = conditioning_key

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None):
        if self.conditioning_key is None:
            raise PbeError("DiffusionWrapper: unconditional U-Net is not on the Paint-by-Example path")
        cc = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
        return self.diffusion_model(x, t, context=cc)


class DDPM(nn.Module):
    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", loss_type="l2", ckpt_path=None, ignore_keys=(),
                 load_only_unet=False, monitor="val/loss", use_ema=True, first_stage_key="image", image_size=256, channels=3,
                 log_every_t=100, clip_denoised=True, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3, given_betas=None,
                 original_elbo_weight=0., v_posterior=0., l_simple_weight=1., conditioning_key=None, parameterization="eps",
                 scheduler_config=None, use_positional_encodings=False, learn_logvar=False, logvar_init=0., u_cond_percent=0):
        super().__init__()
        assert parameterization in {"eps", "x0"}, 'currently only supporting "eps" and "x0"'
        self.parameterization = parameterization
        self.cond_stage_model = None
        self.clip_denoised, self.log_every_t, self.first_stage_key = clip_denoised, log_every_t, first_stage_key
        self.image_size, self.channels, self.u_cond_percent = image_size, channels, u_cond_percent
        self.use_positional_encodings = use_positional_encodings
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.use_ema = use_ema
        if use_ema:
            raise PbeError("DDPM: use_ema=True needs LitEma shadow weights; configs/v1.yaml:19 sets use_ema: False (inference path)")
        self.use_scheduler = scheduler_config is not None
        if self.use_scheduler:
            self.scheduler_config = scheduler_config
        self.v_posterior, self.original_elbo_weight, self.l_simple_weight = v_posterior, original_elbo_weight, l_simple_weight
        if monitor is not None:
            self.monitor = monitor
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps, linear_start=linear_start,
                               linear_end=linear_end, cosine_s=cosine_s)
        self.loss_type = loss_type
        self.learn_logvar = learn_logvar
        self.logvar = nn.Parameter(torch.full(fill_value=logvar_init, size=(self.num_timesteps,)), requires_grad=False)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys, only_model=load_only_unet)

    @property
    def device(self):
        return self.betas.device

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if given_betas is not None else make_beta_schedule(beta_schedule, timesteps, linear_start=linear_start,
                                                                              linear_end=linear_end, cosine_s=cosine_s)
        betas = np.asarray(betas, dtype=np.float64)
        alphas = 1. - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1., ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end

        def reg(name, arr, persistent=True):
            self.register_buffer(name, torch.as_tensor(np.asarray(arr), dtype=torch.float32), persistent=persistent)

        reg("betas", betas)
        reg("alphas_cumprod", ac)
        reg("alphas_cumprod_prev", ac_prev)
        reg("sqrt_alphas_cumprod", np.sqrt(ac))
        reg("sqrt_one_minus_alphas_cumprod", np.sqrt(1. - ac))
        reg("log_one_minus_alphas_cumprod", np.log(1. - ac))
        reg("sqrt_recip_alphas_cumprod", np.sqrt(1. / ac))
        reg("sqrt_recipm1_alphas_cumprod", np.sqrt(1. / ac - 1))
        post_var = (1 - self.v_posterior) * betas * (1. - ac_prev) / (1. - ac) + self.v_posterior * betas
        reg("posterior_variance", post_var)
        reg("posterior_log_variance_clipped", np.log(np.maximum(post_var, 1e-20)))
        reg("posterior_mean_coef1", betas * np.sqrt(ac_prev) / (1. - ac))
        reg("posterior_mean_coef2", (1. - ac_prev) * np.sqrt(alphas) / (1. - ac))

    @contextmanager
    def ema_scope(self, context=None):
        yield None            # use_ema False: nothing to swap (ddpm.py:231-243)

    def init_from_ckpt(self, path, ignore_keys=(), only_model=False):
        from pbe_amd.checkpoint import read_state_dict
        from pbe_amd.weights import canonical_checkpoint_keys
        sd = canonical_checkpoint_keys(read_state_dict(path))
        sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in (ignore_keys or ()))}
        missing, unexpected = (self.model if only_model else self).load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")

    def q_sample(self, x_start, t, noise=None):
        noise = default(noise, lambda: torch.randn_like(x_start))
        return (extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start +
                extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def forward(self, *args, **kwargs):
        raise PbeError("training forward (p_losses) is out of scope: this build is the PLMS inference hot path")


class LatentDiffusion(DDPM):
    """The object scripts/inference.py drives: CLIP exemplar encoder + KL autoencoder + 9-channel U-Net."""

    def __init__(self, first_stage_config, cond_stage_config, num_timesteps_cond=None, cond_stage_key="image", cond_stage_trainable=False,
                 concat_mode=True, cond_stage_forward=None, conditioning_key=None, scale_factor=1.0, scale_by_std=False,
                 cond_embed_dim=1024, cond_context_dim=768, *args, **kwargs):
        self.num_timesteps_cond = default(num_timesteps_cond, 1)
        if scale_by_std:
            raise PbeError("LatentDiffusion: scale_by_std is a training-time feature")
        self.scale_by_std = scale_by_std
        assert self.num_timesteps_cond <= kwargs["timesteps"]
        if conditioning_key is None:
            conditioning_key = "concat" if concat_mode else "crossattn"
        ckpt_path = kwargs.pop("ckpt_path", None)
        ignore_keys = kwargs.pop("ignore_keys", [])
        super().__init__(conditioning_key=conditioning_key, *args, **kwargs)
        self.learnable_vector = nn.Parameter(torch.randn((1, 1, cond_context_dim)), requires_grad=False)   # latent_diffusion.py:111
        self.proj_out = HipLinear(cond_embed_dim, cond_context_dim)                                          # latent_diffusion.py:112
        self.concat_mode, self.cond_stage_trainable, self.cond_stage_key = concat_mode, cond_stage_trainable, cond_stage_key
        try:
            self.num_downs = len(first_stage_config["params"]["ddconfig"]["ch_mult"]) - 1
        except Exception:                                            # noqa: BLE001
            self.num_downs = 0
        self.scale_factor = scale_factor
        self.first_stage_model = instantiate_from_config(first_stage_config).eval()
        self.first_stage_model.train = disabled_train.__get__(self.first_stage_model)
        self.cond_stage_model = instantiate_from_config(cond_stage_config)
        if self.cond_stage_model is not None:
            self.cond_stage_model.eval()
        self.cond_stage_forward = cond_stage_forward
        self.clip_denoised = False
        self.restarted_from_ckpt = False
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys)
            self.restarted_from_ckpt = True
        for p in self.parameters():
            p.requires_grad_(False)

    def load_state_dict(self, state_dict, strict=True, **kw):
        from pbe_amd.weights import canonical_checkpoint_keys
        return super().load_state_dict(canonical_checkpoint_keys(state_dict), strict=strict, **kw)

    # ---- conditioning ---------------------------------------------------------------------------
    def get_learned_conditioning(self, c):
        m = self.cond_stage_model
        if self.cond_stage_forward is None:
            c = m.encode(c) if hasattr(m, "encode") and callable(m.encode) else m(c)
            if isinstance(c, DiagonalGaussianDistribution):
                c = c.mode()
            return c
        return getattr(m, self.cond_stage_forward)(c)

    def project_conditioning(self, z):
        """``model.proj_out(c)`` of scripts/inference.py:327: [B,1,1024] -> [B,1,768] fp16 (kept as an alias of the module call)."""
        return self.proj_out(z)

    # ---- first stage -----------------------------------------------------------------------------
    @torch.no_grad()
    def encode_first_stage(self, x):
        if hasattr(self, "split_input_params"):
            raise PbeError("fold/unfold tiling (split_input_params) is out of scope")
        return self.first_stage_model.encode(x)

    def get_first_stage_encoding(self, encoder_posterior, noise=None):
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            return encoder_posterior.sample(noise=noise, scale=self.scale_factor)       # sample(), not mode(): latent_diffusion.py:256-257
        if isinstance(encoder_posterior, torch.Tensor):
            return self.scale_factor * encoder_posterior
        raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")

    @torch.no_grad()
    def decode_first_stage_nhwc(self, z):
        """z fp32 NCHW [B, >=4, h, w] -> decoded image fp16 NHWC [B, 8h, 8w, 3] (no in-place mutation)."""
        require_gpu(z, "decode_first_stage")
        return self.first_stage_model.decode_nhwc(ops.scale_latent(z.float(), 1. / self.scale_factor))

    @torch.no_grad()
    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        if predict_cids or hasattr(self, "split_input_params"):
            raise PbeError("codebook / tiled decoding is out of scope")
        img = ops.nhwc_to_nchw(self.decode_first_stage_nhwc(z))
        z *= 1. / self.scale_factor          # the reference un-scales its argument in place (latent_diffusion.py:454); kept for parity
        return img

    # ---- denoiser ----------------------------------------------------------------------------------
    def apply_model(self, x_noisy, t, cond, return_ids=False):
        if not isinstance(cond, dict):
            cond = {"c_crossattn": cond if isinstance(cond, list) else [cond]}
        out = self.model(x_noisy, t, **cond)
        return out[0] if isinstance(out, tuple) and not return_ids else out

    def prepare(self):
        """Build every fp16 weight pack now (otherwise it happens lazily on first use)."""
        for m in self.modules():
            if isinstance(m, HipModule):
                m.pk()
        return self


def load_model_from_config(config, ckpt=None, device="cuda", verbose=False):
    """scripts/inference.py:58-75 counterpart: build from the config's ``model`` section and load a
    Lightning checkpoint's ``state_dict`` (strict=False, EMA keys dropped, CLIP keys remapped).  The file is read
    without executing anything from it (pbe_amd/checkpoint.py): foreign ``callbacks`` / ``hyper_parameters`` objects
    of a Lightning file are skipped instead of failing the load."""
    from pbe_amd.checkpoint import read_state_dict
    model = instantiate_from_config(config["model"])
    if ckpt:
        m, u = model.load_state_dict(read_state_dict(ckpt, verbose=verbose), strict=False)
        if verbose:
            print("missing keys:", m, "\nunexpected keys:", u)
    return model.to(device).eval()


__all__ = ["DDPM", "DiffusionWrapper", "LatentDiffusion", "load_model_from_config", "count_params"]
