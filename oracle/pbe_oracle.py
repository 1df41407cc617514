"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A plain-PyTorch fp32, *functional* restatement of the Paint-by-Example PLMS denoising hot
path.  It exists only to check the HIP path:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import it; the product (``pbe_amd`` / ``ldm``) never does and has no CPU fallback.
  * parity status: PINNED.  ``oracle/gen_golden.py`` (run in the build container, where
    ``/root/reference`` is importable) drives the reference's own modules with the same
    name-seeded weights and asserts this file reproduces their outputs to fp32 round-off
    before it writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks the
    oracle against those fixtures on every run.  The CLIP ViT-L/14 tower is third-party
    (`transformers`, pinned 4.19.2 by the reference, 5.15.0 in this image): it is pinned
    against the installed `transformers` built from a local config with name-seeded weights
    (the reference holds no fixture for it -> "parity unpinned by the reference itself").

Every function cites the reference file:line it restates (paths relative to /root/reference).
Weights arrive as a flat ``dict[str, Tensor]`` in the reference's ``state_dict`` key layout.
Tensors are NCHW fp32, exactly like the reference modules.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# configs/v1.yaml:30-46 (unet_config.params) and :48-69 (first_stage_config.params.ddconfig)
UNET_V1 = dict(in_channels=9, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1),
               num_res_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=8, context_dim=768)
VAE_V1 = dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4, in_channels=3,
              out_ch=3, embed_dim=4)
CLIP_V1 = dict(hidden=1024, heads=16, layers=24, mlp=4096, patch=14, image=224, eps=1e-5)
MAPPER_V1 = dict(width=1024, layers=5)
SCALE_FACTOR = 0.18215          # configs/v1.yaml:18


def _w(sd: SD, name: str) -> torch.Tensor:
    return sd[name].float()


def _opt(sd: SD, name: str) -> Optional[torch.Tensor]:
    return sd[name].float() if name in sd else None


# ------------------------------------------------------------------------------------------
# L1 primitives
# ------------------------------------------------------------------------------------------

def beta_schedule_linear(n: int = 1000, start: float = 0.00085, end: float = 0.012) -> np.ndarray:
    """util.py:21-25: linspace in sqrt(beta), float64, squared."""
    return np.linspace(start ** 0.5, end ** 0.5, n, dtype=np.float64) ** 2


def schedule_buffers(n: int = 1000, start: float = 0.00085, end: float = 0.012) -> Dict[str, np.ndarray]:
    """ddpm.py:175-197: betas / alphas_cumprod / alphas_cumprod_prev as float32 buffers."""
    betas = beta_schedule_linear(n, start, end)
    ac = np.cumprod(1.0 - betas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    return {"betas": betas.astype(np.float32), "alphas_cumprod": ac.astype(np.float32),
            "alphas_cumprod_prev": ac_prev.astype(np.float32)}


def ddim_timesteps_uniform(num_ddim: int, num_ddpm: int = 1000) -> np.ndarray:
    """util.py:46-60 ('uniform'): range(0, T, T//S) + 1."""
    c = num_ddpm // num_ddim
    return np.asarray(list(range(0, num_ddpm, c))) + 1


def ddim_parameters(alphacums: np.ndarray, ddim_t: np.ndarray, eta: float = 0.0):
    """util.py:63-74: (sigmas, alphas, alphas_prev) selected from the float32 alphas_cumprod."""
    alphas = alphacums[ddim_t]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_t[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """util.py:151-171: cat(cos, sin) of t * exp(-ln(P) k / half), cos FIRST."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm(x: torch.Tensor, sd: SD, name: str, eps: float, groups: int = 32) -> torch.Tensor:
    """util.py:214-216 (GroupNorm32, eps 1e-5) / attention.py:77-78, model.py:40-41 (eps 1e-6)."""
    return F.group_norm(x.float(), groups, _w(sd, name + ".weight"), _w(sd, name + ".bias"), eps)


def conv(x: torch.Tensor, sd: SD, name: str, stride: int = 1, padding: int = 0) -> torch.Tensor:
    return F.conv2d(x, _w(sd, name + ".weight"), _opt(sd, name + ".bias"), stride=stride, padding=padding)


def linear(x: torch.Tensor, sd: SD, name: str) -> torch.Tensor:
    return F.linear(x, _w(sd, name + ".weight"), _opt(sd, name + ".bias"))


def layer_norm(x: torch.Tensor, sd: SD, name: str, eps: float = 1e-5) -> torch.Tensor:
    w = _w(sd, name + ".weight")
    return F.layer_norm(x.float(), (w.shape[0],), w, _w(sd, name + ".bias"), eps)


# ------------------------------------------------------------------------------------------
# U-Net (openaimodel.py, attention.py)
# ------------------------------------------------------------------------------------------

def res_block(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """openaimodel.py:255-275 (no up/down, no scale-shift): GN-SiLU-conv, +emb, GN-SiLU-conv, +skip."""
    h = conv(F.silu(group_norm(x, sd, p + "in_layers.0", 1e-5)), sd, p + "in_layers.2", padding=1)
    e = linear(F.silu(emb), sd, p + "emb_layers.1")
    h = h + e[:, :, None, None]
    h = conv(F.silu(group_norm(h, sd, p + "out_layers.0", 1e-5)), sd, p + "out_layers.3", padding=1)
    if (p + "skip_connection.weight") in sd:
        x = conv(x, sd, p + "skip_connection")
    return x + h


def cross_attention(sd: SD, p: str, x: torch.Tensor, context: Optional[torch.Tensor], heads: int) -> torch.Tensor:
    """attention.py:207-230: q,k,v linears (no bias) -> per-head softmax(q k^T d^-1/2) v -> to_out."""
    ctx = x if context is None else context
    q, k, v = linear(x, sd, p + "to_q"), linear(ctx, sd, p + "to_k"), linear(ctx, sd, p + "to_v")
    b, n, c = q.shape
    d = c // heads

    def split(t):
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    sim = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    out = torch.matmul(sim.softmax(dim=-1), v)
    out = out.permute(0, 2, 1, 3).reshape(b, n, c)
    return linear(out, sd, p + "to_out.0")


def transformer_block(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor, heads: int) -> torch.Tensor:
    """attention.py:248-252 + GEGLU feed-forward :38-65 (erf GELU)."""
    x = cross_attention(sd, p + "attn1.", layer_norm(x, sd, p + "norm1"), None, heads) + x
    x = cross_attention(sd, p + "attn2.", layer_norm(x, sd, p + "norm2"), context, heads) + x
    h = linear(layer_norm(x, sd, p + "norm3"), sd, p + "ff.net.0.proj")
    a, gate = h.chunk(2, dim=-1)
    x = linear(a * F.gelu(gate), sd, p + "ff.net.2") + x
    return x


def spatial_transformer(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor, heads: int) -> torch.Tensor:
    """attention.py:287-298: GN(eps 1e-6) -> 1x1 -> tokens -> block -> 1x1 -> + x_in."""
    b, c, hh, ww = x.shape
    h = conv(group_norm(x, sd, p + "norm", 1e-6), sd, p + "proj_in")
    h = h.reshape(b, c, hh * ww).transpose(1, 2)
    h = transformer_block(sd, p + "transformer_blocks.0.", h, context, heads)
    h = h.transpose(1, 2).reshape(b, c, hh, ww)
    return conv(h, sd, p + "proj_out") + x


def unet_layout(cfg: dict) -> Tuple[List[List[Tuple[str, int]]], List[List[Tuple[str, int]]]]:
    """Block layout implied by openaimodel.py:658-828 for use_spatial_transformer / legacy=False.
    Returns (input_blocks, output_blocks); each block is a list of (kind, out_channels) with
    kind in {'conv_in','res','attn','down','up'}."""
    mc, mult, nrb = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"]
    inp: List[List[Tuple[str, int]]] = [[("conv_in", mc)]]
    chans, ch, ds = [mc], mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            ch = m * mc
            blk = [("res", ch)]
            if ds in cfg["attention_resolutions"]:
                blk.append(("attn", ch))
            inp.append(blk)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    out: List[List[Tuple[str, int]]] = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            chans.pop()
            ch = mc * m
            blk = [("res", ch)]
            if ds in cfg["attention_resolutions"]:
                blk.append(("attn", ch))
            if level and i == nrb:
                blk.append(("up", ch))
                ds //= 2
            out.append(blk)
    return inp, out


def unet_forward(sd: SD, x: torch.Tensor, t: torch.Tensor, context: torch.Tensor,
                 cfg: dict = UNET_V1, prefix: str = "") -> torch.Tensor:
    """openaimodel.py:852-889."""
    heads = cfg["num_heads"]
    inp, outb = unet_layout(cfg)
    emb = timestep_embedding(t, cfg["model_channels"])
    emb = linear(F.silu(linear(emb, sd, prefix + "time_embed.0")), sd, prefix + "time_embed.2")

    def run(block, base, h):
        for j, (kind, _) in enumerate(block):
            p = f"{base}.{j}."
            if kind == "conv_in":
                h = conv(h, sd, p[:-1], padding=1)
            elif kind == "res":
                h = res_block(sd, p, h, emb)
            elif kind == "attn":
                h = spatial_transformer(sd, p, h, context, heads)
            elif kind == "down":                                   # openaimodel.py:150-160
                h = conv(h, sd, p + "op", stride=2, padding=1)
            elif kind == "up":                                     # openaimodel.py:109-119
                h = conv(F.interpolate(h, scale_factor=2, mode="nearest"), sd, p + "conv", padding=1)
        return h

    h = x.float()
    hs = []
    for i, blk in enumerate(inp):
        h = run(blk, f"{prefix}input_blocks.{i}", h)
        hs.append(h)
    mp = prefix + "middle_block."
    h = res_block(sd, mp + "0.", h, emb)
    h = spatial_transformer(sd, mp + "1.", h, context, heads)
    h = res_block(sd, mp + "2.", h, emb)
    for i, blk in enumerate(outb):
        h = torch.cat([h, hs.pop()], dim=1)
        h = run(blk, f"{prefix}output_blocks.{i}", h)
    h = F.silu(group_norm(h, sd, prefix + "out.0", 1e-5))
    return conv(h, sd, prefix + "out.2", padding=1)


# ------------------------------------------------------------------------------------------
# AutoencoderKL (model.py, autoencoder.py, distributions.py)
# ------------------------------------------------------------------------------------------

def vae_resnet(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """model.py:123-143 with temb=None."""
    h = conv(F.silu(group_norm(x, sd, p + "norm1", 1e-6)), sd, p + "conv1", padding=1)
    h = conv(F.silu(group_norm(h, sd, p + "norm2", 1e-6)), sd, p + "conv2", padding=1)
    if (p + "nin_shortcut.weight") in sd:
        x = conv(x, sd, p + "nin_shortcut")
    return x + h


def vae_attn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """model.py:180-204: single head over HW tokens, scale c^-1/2, softmax over keys."""
    b, c, hh, ww = x.shape
    h = group_norm(x, sd, p + "norm", 1e-6)
    q, k, v = (conv(h, sd, p + n).reshape(b, c, hh * ww) for n in ("q", "k", "v"))
    w = torch.bmm(q.transpose(1, 2), k) * (int(c) ** -0.5)          # [b, i(query), j(key)]
    w = F.softmax(w, dim=2)
    o = torch.bmm(v, w.transpose(1, 2)).reshape(b, c, hh, ww)       # o[c, i] = sum_j v[c, j] w[i, j]
    return x + conv(o, sd, p + "proj_out")


def vae_encoder(sd: SD, x: torch.Tensor, cfg: dict = VAE_V1, prefix: str = "encoder.") -> torch.Tensor:
    """model.py:436-471."""
    nres = len(cfg["ch_mult"])
    h = conv(x.float(), sd, prefix + "conv_in", padding=1)
    for lvl in range(nres):
        for b in range(cfg["num_res_blocks"]):
            h = vae_resnet(sd, f"{prefix}down.{lvl}.block.{b}.", h)
        if lvl != nres - 1:                                         # model.py:74-78 asym pad, s2 p0
            h = conv(F.pad(h, (0, 1, 0, 1)), sd, f"{prefix}down.{lvl}.downsample.conv", stride=2)
    h = vae_resnet(sd, prefix + "mid.block_1.", h)
    h = vae_attn(sd, prefix + "mid.attn_1.", h)
    h = vae_resnet(sd, prefix + "mid.block_2.", h)
    h = F.silu(group_norm(h, sd, prefix + "norm_out", 1e-6))
    return conv(h, sd, prefix + "conv_out", padding=1)


def vae_decoder(sd: SD, z: torch.Tensor, cfg: dict = VAE_V1, prefix: str = "decoder.") -> torch.Tensor:
    """model.py:547-580."""
    nres = len(cfg["ch_mult"])
    h = conv(z.float(), sd, prefix + "conv_in", padding=1)
    h = vae_resnet(sd, prefix + "mid.block_1.", h)
    h = vae_attn(sd, prefix + "mid.attn_1.", h)
    h = vae_resnet(sd, prefix + "mid.block_2.", h)
    for lvl in reversed(range(nres)):
        for b in range(cfg["num_res_blocks"] + 1):
            h = vae_resnet(sd, f"{prefix}up.{lvl}.block.{b}.", h)
        if lvl != 0:                                                # model.py:55-59
            h = conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd, f"{prefix}up.{lvl}.upsample.conv", padding=1)
    h = F.silu(group_norm(h, sd, prefix + "norm_out", 1e-6))
    return conv(h, sd, prefix + "conv_out", padding=1)


def vae_moments(sd: SD, x: torch.Tensor, cfg: dict = VAE_V1, prefix: str = "") -> torch.Tensor:
    """autoencoder.py:57-64: encoder -> quant_conv (1x1) -> moments [B, 2*embed, h, w]."""
    return conv(vae_encoder(sd, x, cfg, prefix + "encoder."), sd, prefix + "quant_conv")


def posterior_sample(moments: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """distributions.py:25-37 with the N(0,1) draw injected: mean + exp(0.5 clamp(logvar)) * eps."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * eps


def first_stage_encode(sd: SD, x: torch.Tensor, eps: torch.Tensor, cfg: dict = VAE_V1, prefix: str = "",
                       scale: float = SCALE_FACTOR) -> torch.Tensor:
    """latent_diffusion.py:571-610 + :255-262 (sample(), then * scale_factor)."""
    return scale * posterior_sample(vae_moments(sd, x, cfg, prefix), eps)


def first_stage_decode(sd: SD, z: torch.Tensor, cfg: dict = VAE_V1, prefix: str = "",
                       scale: float = SCALE_FACTOR) -> torch.Tensor:
    """latent_diffusion.py:444-508 (first_stage_key == 'inpaint': z[:, :4]) + autoencoder.py:66-69."""
    z = (z * (1.0 / scale))[:, :4]
    return vae_decoder(sd, conv(z, sd, prefix + "post_quant_conv"), cfg, prefix + "decoder.")


# ------------------------------------------------------------------------------------------
# CLIP ViT-L/14 image tower (third-party: transformers modeling_clip.py) + mapper (xf.py)
# ------------------------------------------------------------------------------------------

def clip_vision_pooled(sd: SD, pixels: torch.Tensor, cfg: dict = CLIP_V1,
                       prefix: str = "vision_model.") -> torch.Tensor:
    """HF CLIPVisionTransformer.forward -> pooler_output (transformers 5.15 modeling_clip.py
    :138-218 embeddings, :259-335 attention, :338-384 MLP/layer, :613-657 model), called from
    modules.py:163-164.  quick_gelu = x * sigmoid(1.702 x)."""
    hid, heads, eps = cfg["hidden"], cfg["heads"], cfg["eps"]
    b = pixels.shape[0]
    pe = F.conv2d(pixels.float(), _w(sd, prefix + "embeddings.patch_embedding.weight"), None, stride=cfg["patch"])
    pe = pe.flatten(2).transpose(1, 2)                              # [B, 256, hid]
    cls = _w(sd, prefix + "embeddings.class_embedding").expand(b, 1, hid)
    x = torch.cat([cls, pe], dim=1) + _w(sd, prefix + "embeddings.position_embedding.weight")[None]
    x = layer_norm(x, sd, prefix + "pre_layrnorm", eps)
    d = hid // heads
    for i in range(cfg["layers"]):
        p = f"{prefix}encoder.layers.{i}."
        h = layer_norm(x, sd, p + "layer_norm1", eps)
        q, k, v = (linear(h, sd, p + "self_attn." + n) for n in ("q_proj", "k_proj", "v_proj"))

        def split(t):
            return t.reshape(b, -1, heads, d).transpose(1, 2)

        w = torch.matmul(split(q), split(k).transpose(-1, -2)) * (d ** -0.5)
        o = torch.matmul(w.softmax(dim=-1), split(v)).transpose(1, 2).reshape(b, -1, hid)
        x = x + linear(o, sd, p + "self_attn.out_proj")
        h = linear(layer_norm(x, sd, p + "layer_norm2", eps), sd, p + "mlp.fc1")
        h = h * torch.sigmoid(1.702 * h)
        x = x + linear(h, sd, p + "mlp.fc2")
    return layer_norm(x[:, 0], sd, prefix + "post_layernorm", eps)


def xf_mapper(sd: SD, z: torch.Tensor, cfg: dict = MAPPER_V1, prefix: str = "mapper.") -> torch.Tensor:
    """xf.py:80-130 at n_ctx = 1, heads = 1.  With one token the softmax over keys is exactly
    1, so attention(x) = c_proj(V) where V is the last third of c_qkv(ln_1 x) (xf.py:61-77:
    view(bs, n_ctx, heads, 3*attn_ch) then split -> q | k | v)."""
    w = cfg["width"]
    for i in range(cfg["layers"]):
        p = f"{prefix}resblocks.{i}."
        qkv = linear(layer_norm(z, sd, p + "ln_1"), sd, p + "attn.c_qkv")
        z = z + linear(qkv[..., 2 * w:], sd, p + "attn.c_proj")
        z = z + linear(F.gelu(linear(layer_norm(z, sd, p + "ln_2"), sd, p + "mlp.c_fc")), sd, p + "mlp.c_proj")
    return z


def learned_conditioning(sd: SD, pixels: torch.Tensor, clip_cfg: dict = CLIP_V1, map_cfg: dict = MAPPER_V1,
                         prefix: str = "cond_stage_model.", proj: str = "proj_out") -> torch.Tensor:
    """modules.py:162-168 (pooler -> unsqueeze(1) -> mapper -> final_ln) followed by
    LatentDiffusion.proj_out (latent_diffusion.py:112; scripts/inference.py:326-327)."""
    z = clip_vision_pooled(sd, pixels, clip_cfg, prefix + "transformer.vision_model.").unsqueeze(1)
    z = xf_mapper(sd, z, map_cfg, prefix + "mapper.")
    z = layer_norm(z, sd, prefix + "final_ln")
    return linear(z, sd, proj)


# ------------------------------------------------------------------------------------------
# PLMS / DDIM samplers (plms.py, ddim.py)
# ------------------------------------------------------------------------------------------

ModelFn = Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor]


def _guided_eps(model: ModelFn, x9, t, c, uc, scale):
    """plms.py:181-190: CFG doubles the batch, e = e_u + s (e_c - e_u)."""
    if uc is None or scale == 1.0:
        return model(x9, t, c)
    if uc.shape[0] != c.shape[0]:
        uc = uc.expand(c.shape[0], *uc.shape[1:])
    e_u, e_c = model(torch.cat([x9] * 2), torch.cat([t] * 2), torch.cat((uc, c))).chunk(2)
    return e_u + scale * (e_c - e_u)


def schedule_subset(ddim_t: np.ndarray, timesteps: Optional[int]) -> np.ndarray:
    """plms.py:132-139 / ddim.py:151-155: `timesteps` keeps a PREFIX of the schedule (subset_end = int(min(t / n, 1) n) - 1)."""
    if timesteps is None:
        return ddim_t
    n = ddim_t.shape[0]
    return ddim_t[:int(min(timesteps / n, 1) * n) - 1]


def q_sample(x0: torch.Tensor, step: int, noise: torch.Tensor, alphas_cumprod: np.ndarray) -> torch.Tensor:
    """ddpm.py:337-341 with the registered fp32 buffers sqrt_alphas_cumprod / sqrt_one_minus_alphas_cumprod (ddpm.py:205-206)."""
    ac = np.float64(alphas_cumprod[int(step)])
    return float(np.float32(np.sqrt(ac))) * x0 + float(np.float32(np.sqrt(1.0 - ac))) * noise


def plms_sample(model: ModelFn, S: int, x_T: torch.Tensor, cond: torch.Tensor, uc: Optional[torch.Tensor],
                scale: float, z_inpaint: torch.Tensor, mask: torch.Tensor, alphas_cumprod: np.ndarray,
                record: Sequence[int] = (), timesteps: Optional[int] = None, blend=None) -> Tuple[torch.Tensor, Dict[str, object]]:
    """plms.py:118-248 with eta = 0 (sigma = 0): returns (x_0 latent, info).
    info['calls'] counts model invocations (S + 1), info['x'] holds x after the steps in `record`.
    blend = (mask, x0, [noise per step]): plms.py:150-153, img = q_sample(x0, ts) * mask + (1 - mask) * img before every step."""
    ddim_t = ddim_timesteps_uniform(S, alphas_cumprod.shape[0])
    _, a, a_prev = ddim_parameters(alphas_cumprod, ddim_t)
    sq1m = np.sqrt(1.0 - a)
    time_range = np.flip(schedule_subset(ddim_t, timesteps))
    S = time_range.shape[0]
    b = x_T.shape[0]
    x = x_T.float()
    calls = 0
    old: List[torch.Tensor] = []
    kept: Dict[int, torch.Tensor] = {}

    def eps_at(xc, step):
        nonlocal calls
        calls += 1
        tt = torch.full((b,), int(step), dtype=torch.int64)
        return _guided_eps(model, torch.cat((xc, z_inpaint, mask), dim=1), tt, cond, uc, scale)

    def step_to_prev(e, idx):                                       # plms.py:202-219
        pred_x0 = (x - float(sq1m[idx]) * e) / math.sqrt(float(a[idx]))
        return math.sqrt(float(a_prev[idx])) * pred_x0 + math.sqrt(1.0 - float(a_prev[idx])) * e, pred_x0

    pred_x0 = x
    for i, step in enumerate(time_range):
        idx = S - i - 1
        if blend is not None:
            bm, bx0, bnoise = blend
            x = q_sample(bx0, step, bnoise[i], alphas_cumprod) * bm + (1.0 - bm) * x
        e_t = eps_at(x, step)
        if len(old) == 0:                                           # plms.py:230-235
            x_prev, _ = step_to_prev(e_t, idx)
            e_next = eps_at(x_prev, time_range[min(i + 1, S - 1)])
            e_p = (e_t + e_next) / 2
        elif len(old) == 1:                                         # plms.py:236-244
            e_p = (3 * e_t - old[-1]) / 2
        elif len(old) == 2:
            e_p = (23 * e_t - 16 * old[-1] + 5 * old[-2]) / 12
        else:
            e_p = (55 * e_t - 59 * old[-1] + 37 * old[-2] - 9 * old[-3]) / 24
        x, pred_x0 = step_to_prev(e_p, idx)
        old.append(e_t)
        if len(old) >= 4:
            old.pop(0)
        if i in record:
            kept[i] = x.clone()
    return x, {"calls": calls, "x": kept, "pred_x0": pred_x0, "timesteps": time_range.copy()}


def ddim_sample(model: ModelFn, S: int, x_T: torch.Tensor, cond: torch.Tensor, uc: Optional[torch.Tensor],
                scale: float, z_inpaint: torch.Tensor, mask: torch.Tensor, alphas_cumprod: np.ndarray,
                record: Sequence[int] = (), eta: float = 0.0, noises=None, temperature: float = 1.0, timesteps: Optional[int] = None,
                blend=None) -> Tuple[torch.Tensor, Dict[str, object]]:
    """ddim.py:193-242: one model call per step, x[:, :4] slice for pred_x0.  eta > 0 (util.py:69 sigmas, ddim.py:234-238):
    x_prev = sqrt(a_prev) pred_x0 + sqrt(1 - a_prev - sigma^2) e + sigma * noises[i] * temperature.  blend as in plms_sample."""
    ddim_t = ddim_timesteps_uniform(S, alphas_cumprod.shape[0])
    _, a, a_prev = ddim_parameters(alphas_cumprod, ddim_t)
    sig = eta * np.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
    sq1m = np.sqrt(1.0 - a)
    b = x_T.shape[0]
    x = x_T.float()
    kept: Dict[int, torch.Tensor] = {}
    calls = 0
    time_range = np.flip(schedule_subset(ddim_t, timesteps))
    S = time_range.shape[0]
    for i, step in enumerate(time_range):
        idx = S - i - 1
        if blend is not None:
            bm, bx0, bnoise = blend
            x = q_sample(bx0, step, bnoise[i], alphas_cumprod) * bm + (1.0 - bm) * x
        tt = torch.full((b,), int(step), dtype=torch.int64)
        e = _guided_eps(model, torch.cat((x, z_inpaint, mask), dim=1), tt, cond, uc, scale)
        calls += 1
        pred_x0 = (x - float(sq1m[idx]) * e) / math.sqrt(float(a[idx]))
        x = math.sqrt(float(a_prev[idx])) * pred_x0 + math.sqrt(1.0 - float(a_prev[idx]) - float(sig[idx]) ** 2) * e
        if eta != 0.0:
            x = x + float(sig[idx]) * noises[i] * temperature
        if i in record:
            kept[i] = x.clone()
    return x, {"calls": calls, "x": kept}


# ------------------------------------------------------------------------------------------
# Caller contract (scripts/inference.py:305-348)
# ------------------------------------------------------------------------------------------

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess_triple(image_u8: np.ndarray, mask_u8: np.ndarray, ref_u8_224: np.ndarray):
    """scripts/inference.py:306-318.  image_u8 [H,W,3], mask_u8 [H,W] (255 = repaint),
    ref_u8_224 [224,224,3] already PIL-resized.  Returns image[-1,1], inpaint, mask{0,1} (1 = keep), ref."""
    img = torch.from_numpy(image_u8.astype(np.float32) / 255.0).permute(2, 0, 1)[None]
    img = (img - 0.5) / 0.5
    ref = torch.from_numpy(ref_u8_224.astype(np.float32) / 255.0).permute(2, 0, 1)[None]
    ref = (ref - torch.tensor(CLIP_MEAN)[None, :, None, None]) / torch.tensor(CLIP_STD)[None, :, None, None]
    m = 1.0 - mask_u8.astype(np.float32)[None, None] / 255.0
    m = np.where(m < 0.5, 0.0, 1.0).astype(np.float32)
    m = torch.from_numpy(m)
    return img, img * m, m, ref


def resize_mask(mask: torch.Tensor, size: Tuple[int, int], antialias: bool = True) -> torch.Tensor:
    """scripts/inference.py:332: torchvision Resize on a float tensor = bilinear,
    align_corners=False; antialias default differs by torchvision version."""
    return F.interpolate(mask.float(), size=size, mode="bilinear", align_corners=False, antialias=antialias)


def inpaint_pipeline(sd: SD, image: torch.Tensor, mask: torch.Tensor, ref: torch.Tensor, x_T: torch.Tensor,
                     post_eps: torch.Tensor, S: int = 50, scale: float = 5.0, unet_cfg: dict = UNET_V1,
                     vae_cfg: dict = VAE_V1, clip_cfg: dict = CLIP_V1, map_cfg: dict = MAPPER_V1,
                     antialias: bool = True) -> Dict[str, torch.Tensor]:
    """scripts/inference.py:323-348 end to end with injected x_T and posterior eps.
    `sd` uses the LatentDiffusion key layout (model.diffusion_model.*, first_stage_model.*,
    cond_stage_model.*, proj_out.*, learnable_vector)."""
    c = learned_conditioning(sd, ref, clip_cfg, map_cfg)
    uc = sd["learnable_vector"].float() if scale != 1.0 else None
    z_inp = first_stage_encode(sd, image * mask, post_eps, vae_cfg, "first_stage_model.")
    m = resize_mask(mask, z_inp.shape[-2:], antialias)
    ac = schedule_buffers()["alphas_cumprod"]

    def model(x9, t, ctx):
        return unet_forward(sd, x9, t, ctx, unet_cfg, "model.diffusion_model.")

    z0, info = plms_sample(model, S, x_T, c, uc, scale, z_inp, m, ac)
    img = first_stage_decode(sd, z0, vae_cfg, "first_stage_model.")
    return {"c": c, "z_inpaint": z_inp, "mask64": m, "latent": z0,
            "image": torch.clamp((img + 1.0) / 2.0, 0.0, 1.0), "calls": info["calls"]}
