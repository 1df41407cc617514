#!/usr/bin/env python3
"""Golden-vector generator — runs ONLY in the build container (needs /root/reference).

It imports the reference's own pure-PyTorch modules (the ones that import cleanly: SURVEY.md
§8c), loads the same name-seeded weights into them that the tests synthesise, runs them on
seeded inputs and

  1. asserts that ``oracle/pbe_oracle.py`` reproduces every reference output to fp32
     round-off (this is what pins the oracle), and
  2. writes the inputs' seeds and the reference outputs to ``tests/golden/*.npz``.

The reference never travels to the GPU box; these fixtures do.  Usage:
    python oracle/gen_golden.py [--full]      (--full adds the 320-channel / 512x512 cases)

Glue that cannot be imported here (``ldm.models.autoencoder`` needs `lightning`,
``ldm.models.diffusion.ddpm`` needs `torchmetrics`/`torchvision`, ``ldm.modules.encoders.modules``
needs `clip`/`kornia`; ordinary ModuleNotFoundError) is ~10 lines each
(quant convs, `* scale_factor`, pooler -> mapper -> final_ln -> proj_out) and is driven here by
calling the importable reference classes (Encoder, Decoder, DiagonalGaussianDistribution,
xf.Transformer, xf.LayerNorm, PLMSSampler, DDIMSampler) directly.
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


O = _load(os.path.join(HERE, "pbe_oracle.py"), "pbe_oracle")
W = _load(os.path.join(REPO, "pbe_amd", "weights.py"), "pbe_weights")
CASES = _load(os.path.join(REPO, "tests", "cases.py"), "pbe_cases")

# the reference's `ldm` (NOT this repo's) must be the one imported below
sys.path.insert(0, REF)
# openaimodel.py:592-594 lazily imports omegaconf.listconfig.ListConfig only for a type() check
_oc = types.ModuleType("omegaconf")
_ocl = types.ModuleType("omegaconf.listconfig")
_ocl.ListConfig = type("ListConfig", (list,), {})
_oc.listconfig = _ocl
sys.modules.setdefault("omegaconf", _oc)
sys.modules.setdefault("omegaconf.listconfig", _ocl)

from ldm.modules.diffusionmodules import openaimodel as ref_unet          # noqa: E402
from ldm.modules.diffusionmodules import model as ref_vae                # noqa: E402
from ldm.modules.diffusionmodules import util as ref_util                # noqa: E402
from ldm.modules import attention as ref_attn                            # noqa: E402
from ldm.modules.distributions.distributions import DiagonalGaussianDistribution  # noqa: E402
from ldm.modules.encoders import xf as ref_xf                            # noqa: E402
from ldm.models.diffusion.plms import PLMSSampler as RefPLMS             # noqa: E402
from ldm.models.diffusion.ddim import DDIMSampler as RefDDIM             # noqa: E402

assert ref_unet.__file__.startswith(REF), ref_unet.__file__

torch.set_grad_enabled(False)
torch.set_num_threads(8)


def fill(module, prefix="", seed=0):
    W.fill_module_(module, seed=seed, prefix=prefix)
    return module.eval()


def sd_of(module, prefix=""):
    return {prefix + k: v.clone() for k, v in module.state_dict().items()}


def close(a, b, what, rtol=2e-4):
    a, b = a.double(), b.double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-12
    assert err <= rtol * ref, f"ORACLE MISMATCH {what}: max|d|={err:.3e} ref_max={ref:.3e}"
    print(f"  oracle==reference  {what:42s} max|d|/max|ref| = {err / ref:.2e}")


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------

def gen_primitives():
    t = torch.tensor([1, 21, 981], dtype=torch.int64)
    ref = ref_util.timestep_embedding(t, 320)
    close(O.timestep_embedding(t, 320), ref, "timestep_embedding")
    betas = ref_util.make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1.0 - betas, axis=0)
    sb = O.schedule_buffers()
    assert np.array_equal(sb["betas"], betas.astype(np.float32))
    assert np.array_equal(sb["alphas_cumprod"], ac.astype(np.float32))
    out = {"temb_t": t, "temb": ref, "betas_0_999": betas[[0, 999]], "acp_0_999": ac[[0, 999]]}
    for S in (50, 100):
        dt = ref_util.make_ddim_timesteps("uniform", S, 1000, verbose=False)
        sig, a, ap = ref_util.make_ddim_sampling_parameters(sb["alphas_cumprod"], dt, 0.0, verbose=False)
        assert np.array_equal(dt, O.ddim_timesteps_uniform(S))
        s2, a2, ap2 = O.ddim_parameters(sb["alphas_cumprod"], dt)
        assert np.array_equal(a, a2) and np.array_equal(ap, ap2) and np.array_equal(sig, s2)
        out[f"ddim_t_{S}"] = dt
        out[f"ddim_a_{S}"] = a
        out[f"ddim_aprev_{S}"] = ap
    print("  oracle==reference  schedule tables (bit-exact)")
    save("primitives", **out)


def gen_blocks():
    """Per-op goldens at reduced size."""
    out = {}
    g = torch.Generator().manual_seed(11)
    # ResBlock 64 -> 128 (1x1 skip) and 128 -> 128 (identity skip) @16x16, emb 256
    for tag, cin, cout in (("res_skip", 64, 128), ("res_id", 128, 128)):
        m = fill(ref_unet.ResBlock(cin, 256, 0.0, out_channels=cout, dims=2, use_checkpoint=False), tag + ".")
        x = torch.randn(2, cin, 16, 16, generator=g)
        emb = torch.randn(2, 256, generator=g)
        y = m(x, emb)
        close(O.res_block(sd_of(m, tag + "."), tag + ".", x, emb), y, f"ResBlock {cin}->{cout}")
        out[tag + "_x"], out[tag + "_emb"], out[tag + "_y"] = x, emb, y
    # SpatialTransformer C=64, 8 heads x 8, N=256, 1-token context of 768
    m = fill(ref_attn.SpatialTransformer(64, 8, 8, depth=1, context_dim=768), "st.")
    for blk in m.transformer_blocks:
        blk.checkpoint = False
    x = torch.randn(2, 64, 16, 16, generator=g)
    ctx = torch.randn(2, 1, 768, generator=g)
    y = m(x, ctx)
    close(O.spatial_transformer(sd_of(m, "st."), "st.", x, ctx, 8), y, "SpatialTransformer C=64")
    out["st_x"], out["st_ctx"], out["st_y"] = x, ctx, y
    # GEGLU feed-forward alone
    ff = fill(ref_attn.FeedForward(64, glu=True), "ff.")
    xt = torch.randn(2, 50, 64, generator=g)
    out["ff_x"], out["ff_y"] = xt, ff(xt)
    # U-Net Downsample / Upsample (conv)
    dn = fill(ref_unet.Downsample(64, True, dims=2, out_channels=64), "dn.")
    up = fill(ref_unet.Upsample(64, True, dims=2, out_channels=64), "up.")
    out["dn_y"], out["up_y"] = dn(x), up(x)
    sdd = {**sd_of(dn, "dn."), **sd_of(up, "up.")}
    close(O.conv(x, sdd, "dn.op", stride=2, padding=1), out["dn_y"], "Downsample s2 p1")
    close(O.conv(F.interpolate(x, scale_factor=2, mode="nearest"), sdd, "up.conv", padding=1), out["up_y"], "Upsample nearest+conv")
    # VAE ResnetBlock 64 -> 128, AttnBlock 64 @16x16, asym-pad Downsample
    rb = fill(ref_vae.ResnetBlock(in_channels=64, out_channels=128, dropout=0.0, temb_channels=0), "vrb.")
    ab = fill(ref_vae.AttnBlock(64), "vab.")
    vd = fill(ref_vae.Downsample(64, True), "vdn.")
    out["vrb_y"], out["vab_y"], out["vdn_y"] = rb(x, None), ab(x), vd(x)
    close(O.vae_resnet(sd_of(rb, "vrb."), "vrb.", x), out["vrb_y"], "VAE ResnetBlock")
    close(O.vae_attn(sd_of(ab, "vab."), "vab.", x), out["vab_y"], "VAE AttnBlock")
    close(O.conv(F.pad(x, (0, 1, 0, 1)), sd_of(vd, "vdn."), "vdn.conv", stride=2), out["vdn_y"], "VAE Downsample asym")
    # posterior: reference draws torch.randn from the global CPU generator (distributions.py:36)
    mom = torch.randn(2, 8, 8, 8, generator=g)
    torch.manual_seed(5)
    eps = torch.randn(2, 4, 8, 8)
    torch.manual_seed(5)
    zs = DiagonalGaussianDistribution(mom).sample()
    close(O.posterior_sample(mom, eps), zs, "DiagonalGaussian.sample")
    out["post_mom"], out["post_eps"], out["post_z"] = mom, eps, zs
    # xf mapper block at n_ctx = 1 + xf.LayerNorm
    tr = fill(ref_xf.Transformer(1, 128, 2, 1), "mapper.")
    ln = fill(ref_xf.LayerNorm(128), "final_ln.")
    z = torch.randn(3, 1, 128, generator=g)
    zo = ln(tr(z))
    sdm = {**sd_of(tr, "mapper."), **sd_of(ln, "final_ln.")}
    close(O.layer_norm(O.xf_mapper(sdm, z, dict(width=128, layers=2)), sdm, "final_ln"), zo, "xf mapper n_ctx=1 + LN")
    out["map_z"], out["map_y"] = z, zo
    save("blocks", **out)


class _RefLatentModel:
    """The attributes PLMSSampler/DDIMSampler read from LatentDiffusion (plms.py:12-55,181-195),
    with apply_model = the reference UNetModel via DiffusionWrapper's crossattn path
    (ddpm.py:477-486: cc = cat(c_crossattn, 1))."""

    def __init__(self, unet):
        sb = O.schedule_buffers()
        self.num_timesteps = 1000
        self.betas = torch.from_numpy(sb["betas"])
        self.alphas_cumprod = torch.from_numpy(sb["alphas_cumprod"])
        self.alphas_cumprod_prev = torch.from_numpy(sb["alphas_cumprod_prev"])
        self.device = torch.device("cpu")
        self.parameterization = "eps"
        self.unet = unet
        self.calls = 0

    def apply_model(self, x, t, c):
        self.calls += 1
        return self.unet(x, timesteps=t, context=c)


class _CpuPLMS(RefPLMS):
    def register_buffer(self, name, attr):       # plms.py:18-22 hard-codes cuda
        setattr(self, name, attr)


class _CpuDDIM(RefDDIM):
    def register_buffer(self, name, attr):
        setattr(self, name, attr)


def build_ref_unet(cfg, prefix):
    m = ref_unet.UNetModel(image_size=32, in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
                           model_channels=cfg["model_channels"], attention_resolutions=list(cfg["attention_resolutions"]),
                           num_res_blocks=cfg["num_res_blocks"], channel_mult=list(cfg["channel_mult"]),
                           num_heads=cfg["num_heads"], use_spatial_transformer=True, transformer_depth=1,
                           context_dim=cfg["context_dim"], use_checkpoint=False, legacy=False)
    return fill(m, prefix)


def build_ref_vae(cfg, prefix):
    dd = dict(double_z=True, z_channels=cfg["z_channels"], resolution=256, in_channels=cfg["in_channels"],
              out_ch=cfg["out_ch"], ch=cfg["ch"], ch_mult=list(cfg["ch_mult"]), num_res_blocks=cfg["num_res_blocks"],
              attn_resolutions=[], dropout=0.0)
    enc = fill(ref_vae.Encoder(**dd), prefix + "encoder.")
    dec = fill(ref_vae.Decoder(**dd), prefix + "decoder.")
    sd = {**sd_of(enc, prefix + "encoder."), **sd_of(dec, prefix + "decoder.")}
    e = cfg["embed_dim"]
    for name, shape in ((prefix + "quant_conv.weight", (2 * e, 2 * cfg["z_channels"], 1, 1)), (prefix + "quant_conv.bias", (2 * e,)),
                        (prefix + "post_quant_conv.weight", (cfg["z_channels"], e, 1, 1)), (prefix + "post_quant_conv.bias", (cfg["z_channels"],))):
        sd[name] = W.synth_tensor(name, shape)
    return enc, dec, sd


def ref_vae_encode(enc, sd, prefix, x, seed):
    """autoencoder.py:57-64 + latent_diffusion.py:255-262."""
    mom = F.conv2d(enc(x), sd[prefix + "quant_conv.weight"], sd[prefix + "quant_conv.bias"])
    torch.manual_seed(seed)
    return mom, O.SCALE_FACTOR * DiagonalGaussianDistribution(mom).sample()


def ref_vae_decode(dec, sd, prefix, z):
    """latent_diffusion.py:454,506-507 + autoencoder.py:66-69."""
    z = z.clone()
    z *= 1.0 / O.SCALE_FACTOR
    return dec(F.conv2d(z[:, :4], sd[prefix + "post_quant_conv.weight"], sd[prefix + "post_quant_conv.bias"]))


def build_ref_clip(cfg, prefix):
    """HF CLIPVisionModel from a LOCAL config (no hub access); returns (module, sd in 4.19 names)."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    conf = CLIPVisionConfig(hidden_size=cfg["hidden"], intermediate_size=cfg["mlp"], num_hidden_layers=cfg["layers"],
                            num_attention_heads=cfg["heads"], image_size=cfg["image"], patch_size=cfg["patch"],
                            hidden_act="quick_gelu", layer_norm_eps=cfg["eps"])
    m = CLIPVisionModel(conf).eval()
    keys = list(m.state_dict().keys())
    has_vm = any(k.startswith("vision_model.") for k in keys)
    sd = {}
    new = {}
    for k, v in m.state_dict().items():
        canon = k if has_vm else "vision_model." + k
        t = W.synth_tensor(prefix + canon, v.shape).to(v.dtype)
        new[k] = t
        sd[prefix + canon] = t
    m.load_state_dict(new, strict=True)
    return m, sd


def gen_narrow_and_pipeline():
    cu, cv, cc, cm = CASES.UNET_NARROW, CASES.VAE_NARROW, CASES.CLIP_NARROW, CASES.MAPPER_NARROW
    UP, VP, CP = "model.diffusion_model.", "first_stage_model.", "cond_stage_model."
    unet = build_ref_unet(cu, UP)
    enc, dec, sd_vae = build_ref_vae(cv, VP)
    clip, sd_clip = build_ref_clip(cc, CP + "transformer.")
    mapper = fill(ref_xf.Transformer(1, cm["width"], cm["layers"], 1), CP + "mapper.")
    fln = fill(ref_xf.LayerNorm(cm["width"]), CP + "final_ln.")
    sd = {**sd_of(unet, UP), **sd_vae, **sd_clip, **sd_of(mapper, CP + "mapper."), **sd_of(fln, CP + "final_ln.")}
    for name, shape in (("proj_out.weight", (768, cm["width"])), ("proj_out.bias", (768,)), ("learnable_vector", (1, 1, 768))):
        sd[name] = W.synth_tensor(name, shape)

    inp = CASES.narrow_inputs()
    out = {}
    # --- single U-Net forward (CFG pair) ---
    x9, t, ctx = inp["unet_x"], inp["unet_t"], inp["unet_ctx"]
    y = unet(x9, timesteps=t, context=ctx)
    close(O.unet_forward(sd, x9, t, ctx, cu, UP), y, "UNetModel narrow forward")
    out["unet_y"] = y
    # reference's own reduced-precision behaviour (calibrates the fp16 tolerance)
    for dt, tag in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        try:
            with torch.autocast("cpu", dtype=dt):
                ya = unet(x9, timesteps=t, context=ctx).float()
            rel = ((ya - y).norm() / y.norm()).item()
            out[f"unet_autocast_{tag}_rel_l2"] = np.float64(rel)
            print(f"  reference autocast({tag}) vs fp32 on narrow U-Net: rel L2 = {rel:.3e}")
        except Exception as e:   # noqa: BLE001
            print(f"  reference autocast({tag}) unavailable on CPU: {type(e).__name__}: {e}")
    # --- CLIP -> mapper -> final_ln -> proj_out ---
    ref_img = inp["ref"]
    pooled = clip(pixel_values=ref_img).pooler_output
    close(O.clip_vision_pooled(sd, ref_img, cc, CP + "transformer.vision_model."), pooled, "CLIPVisionModel narrow pooled")
    c = F.linear(fln(mapper(pooled.unsqueeze(1))), sd["proj_out.weight"], sd["proj_out.bias"])
    close(O.learned_conditioning(sd, ref_img, cc, cm), c, "get_learned_conditioning+proj_out")
    out["clip_pooled"], out["c"] = pooled, c
    # --- VAE encode / decode ---
    image, mask = inp["image"], inp["mask"]
    mom, z_inp = ref_vae_encode(enc, sd, VP, image * mask, seed=CASES.POSTERIOR_SEED)
    close(O.first_stage_encode(sd, image * mask, inp["post_eps"], cv, VP), z_inp, "encode_first_stage+sample")
    out["moments"], out["z_inpaint"] = mom, z_inp
    dimg = ref_vae_decode(dec, sd, VP, inp["x_T"])
    close(O.first_stage_decode(sd, inp["x_T"], cv, VP), dimg, "decode_first_stage")
    out["decoded_xT"] = dimg
    # --- samplers, 50 steps, scale 5, injected x_T ---
    m64 = O.resize_mask(mask, z_inp.shape[-2:], antialias=True)
    out["mask_lat"] = m64
    out["mask_lat_noaa"] = O.resize_mask(mask, z_inp.shape[-2:], antialias=False)
    uc = sd["learnable_vector"].repeat(image.shape[0], 1, 1)
    kw = {"images_inpaint": z_inp, "images_mask": m64}
    lm = _RefLatentModel(unet)
    steps_x = {}

    def img_cb(pred_x0, i):
        pass

    t0 = time.time()
    smp = _CpuPLMS(lm)
    z0, inter = smp.sample(S=50, batch_size=image.shape[0], shape=list(inp["x_T"].shape[1:]), conditioning=c, verbose=False,
                           unconditional_guidance_scale=5.0, unconditional_conditioning=uc, eta=0.0, x_T=inp["x_T"].clone(),
                           log_every_t=1, test_model_kwargs=kw)
    print(f"  reference PLMS 50 steps (narrow): {time.time() - t0:.1f}s, apply_model calls = {lm.calls}")
    assert lm.calls == 51
    xs = inter["x_inter"]              # [x_T, x after step 0, 1, ...]
    assert len(xs) == 51
    oz, info = O.plms_sample(lambda a, b, cc_: O.unet_forward(sd, a, b, cc_, cu, UP), 50, inp["x_T"], c, uc, 5.0, z_inp, m64,
                             O.schedule_buffers()["alphas_cumprod"], record=CASES.PLMS_RECORD)
    assert info["calls"] == 51
    for i in CASES.PLMS_RECORD:
        close(info["x"][i], xs[i + 1], f"PLMS x after step {i}", rtol=2e-3)
        out[f"plms_x_{i}"] = xs[i + 1]
    close(oz, z0, "PLMS final latent", rtol=2e-3)
    out["plms_latent"] = z0
    fin = torch.clamp((ref_vae_decode(dec, sd, VP, z0) + 1.0) / 2.0, 0.0, 1.0)
    out["plms_image"] = fin
    pipe = O.inpaint_pipeline(sd, image, mask, ref_img, inp["x_T"], inp["post_eps"], 50, 5.0, cu, cv, cc, cm)
    close(pipe["image"], fin, "inference.py pipeline final image", rtol=2e-3)
    # DDIM, 20 steps
    lm.calls = 0
    zd, _ = _CpuDDIM(lm).sample(S=20, batch_size=image.shape[0], shape=list(inp["x_T"].shape[1:]), conditioning=c, verbose=False,
                                unconditional_guidance_scale=5.0, unconditional_conditioning=uc, eta=0.0, x_T=inp["x_T"].clone(),
                                test_model_kwargs=kw, disable_tqdm=True)
    assert lm.calls == 20
    od, _ = O.ddim_sample(lambda a, b, cc_: O.unet_forward(sd, a, b, cc_, cu, UP), 20, inp["x_T"], c, uc, 5.0, z_inp, m64,
                          O.schedule_buffers()["alphas_cumprod"])
    close(od, zd, "DDIM 20-step latent", rtol=2e-3)
    out["ddim_latent"] = zd
    # S=100 call count (SURVEY.md §8c known answers)
    out["plms_calls_50"] = np.int64(51)
    save("narrow", **out)
    # state-dict key/shape manifest of the narrow model (drop-in boundary: keys must match)
    with open(os.path.join(OUT, "narrow_keys.txt"), "w") as f:
        for k in sorted(sd):
            f.write(f"{k} {'x'.join(str(int(s)) for s in sd[k].shape)}\n")


def gen_full_manifest():
    """Key/shape manifest of the full-size (configs/v1.yaml) model: built on the meta device."""
    lines = []
    with torch.device("meta"):
        unet = ref_unet.UNetModel(image_size=32, in_channels=9, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                                  num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                                  transformer_depth=1, context_dim=768, use_checkpoint=True, legacy=False)
        dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
                  num_res_blocks=2, attn_resolutions=[], dropout=0.0)
        enc, dec = ref_vae.Encoder(**dd), ref_vae.Decoder(**dd)
        mapper, fln = ref_xf.Transformer(1, 1024, 5, 1), ref_xf.LayerNorm(1024)
    for pre, mod in (("model.diffusion_model.", unet), ("first_stage_model.encoder.", enc), ("first_stage_model.decoder.", dec),
                     ("cond_stage_model.mapper.", mapper), ("cond_stage_model.final_ln.", fln)):
        for k, v in mod.state_dict().items():
            lines.append((pre + k, tuple(v.shape)))
    lines += [("first_stage_model.quant_conv.weight", (8, 8, 1, 1)), ("first_stage_model.quant_conv.bias", (8,)),
              ("first_stage_model.post_quant_conv.weight", (4, 4, 1, 1)), ("first_stage_model.post_quant_conv.bias", (4,)),
              ("proj_out.weight", (768, 1024)), ("proj_out.bias", (768,)), ("learnable_vector", (1, 1, 768))]
    n_unet = sum(int(np.prod(s)) for k, s in lines if k.startswith("model.diffusion_model."))
    assert n_unet == 859_535_364, n_unet
    with open(os.path.join(OUT, "v1_keys.txt"), "w") as f:
        for k, s in sorted(lines):
            f.write(f"{k} {'x'.join(str(int(x)) for x in s)}\n")
    print(f"wrote v1_keys.txt ({len(lines)} tensors, U-Net params {n_unet})")


def gen_full():
    """Full-size (configs/v1.yaml) single forwards: U-Net CFG pair @64x64, VAE enc/dec @512, CLIP-L."""
    UP, VP, CP = "model.diffusion_model.", "first_stage_model.", "cond_stage_model."
    inp = CASES.full_inputs()
    out = {}
    t0 = time.time()
    unet = build_ref_unet(O.UNET_V1, UP)
    y = unet(inp["unet_x"], timesteps=inp["unet_t"], context=inp["unet_ctx"])
    print(f"  reference full U-Net forward b=2: {time.time() - t0:.1f}s")
    sd = sd_of(unet, UP)
    close(O.unet_forward(sd, inp["unet_x"], inp["unet_t"], inp["unet_ctx"], O.UNET_V1, UP), y, "UNetModel v1 forward")
    out["unet_y"] = y
    del unet, sd
    enc, dec, sd = build_ref_vae(O.VAE_V1, VP)
    t0 = time.time()
    mom, z = ref_vae_encode(enc, sd, VP, inp["image"], seed=CASES.POSTERIOR_SEED)
    dimg = ref_vae_decode(dec, sd, VP, inp["z_dec"])
    print(f"  reference full VAE enc+dec b=1: {time.time() - t0:.1f}s")
    close(O.vae_moments(sd, inp["image"], O.VAE_V1, VP), mom, "VAE v1 moments")
    close(O.first_stage_decode(sd, inp["z_dec"], O.VAE_V1, VP), dimg, "VAE v1 decode")
    out["moments"], out["z"], out["decoded_sub4"] = mom, z, dimg[:, :, ::4, ::4].contiguous()
    out["decoded_mean_std"] = np.array([dimg.mean().item(), dimg.std().item()])
    del enc, dec, sd
    clip, sd = build_ref_clip(O.CLIP_V1, CP + "transformer.")
    pooled = clip(pixel_values=inp["ref"]).pooler_output
    close(O.clip_vision_pooled(sd, inp["ref"], O.CLIP_V1, CP + "transformer.vision_model."), pooled, "CLIP ViT-L/14 pooled")
    out["clip_pooled"] = pooled
    save("full", **out)


def gen_full_plms():
    """Full-size (configs/v1.yaml) PLMS trajectory with the REFERENCE sampler and U-Net: 4 steps, guidance 5, one sample."""
    UP = "model.diffusion_model."
    inp = CASES.full_plms_inputs()
    unet = build_ref_unet(O.UNET_V1, UP)
    sd = sd_of(unet, UP)
    lm = _RefLatentModel(unet)
    kw = {"images_inpaint": inp["z_inpaint"], "images_mask": inp["mask_lat"]}
    t0 = time.time()
    z0, inter = _CpuPLMS(lm).sample(S=inp["steps"], batch_size=1, shape=[4, 64, 64], conditioning=inp["c"], verbose=False,
                                    unconditional_guidance_scale=inp["scale"], unconditional_conditioning=inp["uc"], eta=0.0,
                                    x_T=inp["x_T"].clone(), log_every_t=1, test_model_kwargs=kw)
    print(f"  reference full-size PLMS {inp['steps']} steps: {time.time() - t0:.1f}s, apply_model calls = {lm.calls}")
    assert lm.calls == inp["steps"] + 1
    xs = inter["x_inter"]
    oz, info = O.plms_sample(lambda a, b, cc_: O.unet_forward(sd, a, b, cc_, O.UNET_V1, UP), inp["steps"], inp["x_T"], inp["c"], inp["uc"],
                             inp["scale"], inp["z_inpaint"], inp["mask_lat"], O.schedule_buffers()["alphas_cumprod"],
                             record=tuple(range(inp["steps"])))
    out = {}
    for i in range(inp["steps"]):
        close(info["x"][i], xs[i + 1], f"full-size PLMS x after step {i}", rtol=2e-3)
        out[f"plms_x_{i}"] = xs[i + 1]
    close(oz, z0, "full-size PLMS final latent", rtol=2e-3)
    out["plms_latent"] = z0
    save("full_plms", **out)


def gen_sampler_options():
    """Reference samplers with the options that draw noise inside the loop, noise injected (narrow U-Net): DDIM eta = 0.7 /
    temperature 0.9, DDIM with mask + x0, PLMS with mask + x0 and timesteps = 8 of 12."""
    from ldm.models.diffusion import ddim as ref_ddim_mod
    UP = "model.diffusion_model."
    cu = CASES.UNET_NARROW
    unet = build_ref_unet(cu, UP)
    sd = sd_of(unet, UP)
    inp = CASES.sampler_option_inputs()
    ac = O.schedule_buffers()["alphas_cumprod"]
    model_fn = lambda a, b, cc_: O.unet_forward(sd, a, b, cc_, cu, UP)          # noqa: E731
    kw = {"images_inpaint": inp["z_inpaint"], "images_mask": inp["mask_lat"]}
    out = {}

    class _Stub(_RefLatentModel):
        def __init__(self, unet, noises):
            super().__init__(unet)
            self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)                        # ddpm.py:205-206
            self.sqrt_one_minus_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)
            self.it = iter(noises)

        def q_sample(self, x_start, t, noise=None):                                           # ddpm.py:337-341 with the injected noise
            n = next(self.it)
            return (self.sqrt_alphas_cumprod[t].view(-1, 1, 1, 1) * x_start + self.sqrt_one_minus_alphas_cumprod[t].view(-1, 1, 1, 1) * n)

    # (a) stochastic DDIM
    it = iter(inp["noises"])
    keep = ref_ddim_mod.noise_like
    ref_ddim_mod.noise_like = lambda shape, device, repeat=False: next(it)
    try:
        lm = _RefLatentModel(unet)
        za, _ = _CpuDDIM(lm).sample(S=10, batch_size=2, shape=[4, 16, 16], conditioning=inp["c"], verbose=False, unconditional_guidance_scale=5.0,
                                    unconditional_conditioning=inp["uc"], eta=0.7, temperature=0.9, x_T=inp["x_T"].clone(), test_model_kwargs=kw,
                                    disable_tqdm=True)
    finally:
        ref_ddim_mod.noise_like = keep
    oa, _ = O.ddim_sample(model_fn, 10, inp["x_T"], inp["c"], inp["uc"], 5.0, inp["z_inpaint"], inp["mask_lat"], ac, eta=0.7,
                          noises=inp["noises"], temperature=0.9)
    close(oa, za, "DDIM eta=0.7 temperature=0.9, 10 steps", rtol=2e-3)
    out["ddim_eta_latent"] = za
    # (b) DDIM with mask / x0
    lm = _Stub(unet, inp["noises"])
    zb, _ = _CpuDDIM(lm).sample(S=10, batch_size=2, shape=[4, 16, 16], conditioning=inp["c"], verbose=False, unconditional_guidance_scale=5.0,
                                unconditional_conditioning=inp["uc"], eta=0.0, x_T=inp["x_T"].clone(), test_model_kwargs=kw, disable_tqdm=True,
                                mask=inp["blend_mask"], x0=inp["x0"])
    ob, _ = O.ddim_sample(model_fn, 10, inp["x_T"], inp["c"], inp["uc"], 5.0, inp["z_inpaint"], inp["mask_lat"], ac,
                          blend=(inp["blend_mask"], inp["x0"], inp["noises"]))
    close(ob, zb, "DDIM mask + x0, 10 steps", rtol=2e-3)
    out["ddim_blend_latent"] = zb
    # (c) PLMS with mask / x0 and a schedule prefix
    lm = _Stub(unet, inp["noises"])
    smp = _CpuPLMS(lm)
    smp.make_schedule(ddim_num_steps=12, ddim_eta=0.0, verbose=False)
    zc, _ = smp.plms_sampling(inp["c"], (2, 4, 16, 16), x_T=inp["x_T"].clone(), timesteps=8, mask=inp["blend_mask"], x0=inp["x0"],
                              unconditional_guidance_scale=5.0, unconditional_conditioning=inp["uc"], test_model_kwargs=kw)
    oc, info = O.plms_sample(model_fn, 12, inp["x_T"], inp["c"], inp["uc"], 5.0, inp["z_inpaint"], inp["mask_lat"], ac, timesteps=8,
                             blend=(inp["blend_mask"], inp["x0"], inp["noises"]))
    close(oc, zc, "PLMS mask + x0, timesteps=8 of 12", rtol=2e-3)
    out["plms_blend_subset_latent"] = zc
    out["plms_blend_subset_calls"] = np.int64(info["calls"])
    assert lm.calls == info["calls"], (lm.calls, info["calls"])
    save("sampler_options", **out)


def gen_full_plms50():
    """The HEADLINE trajectory at headline size: reference PLMSSampler + reference UNetModel at 320 channels,
    S = 50, guidance 5, one sample; x after steps 0 / 3 / 25 / 49 and the final latent.  ~51 CFG U-Net pairs on the CPU, twice
    (reference, then the oracle for the pin)."""
    UP = "model.diffusion_model."
    inp = CASES.full_plms50_inputs()
    rec = CASES.FULL_PLMS50_RECORD
    unet = build_ref_unet(O.UNET_V1, UP)
    sd = sd_of(unet, UP)
    lm = _RefLatentModel(unet)
    kw = {"images_inpaint": inp["z_inpaint"], "images_mask": inp["mask_lat"]}
    t0 = time.time()
    z0, inter = _CpuPLMS(lm).sample(S=inp["steps"], batch_size=1, shape=[4, 64, 64], conditioning=inp["c"], verbose=False,
                                    unconditional_guidance_scale=inp["scale"], unconditional_conditioning=inp["uc"], eta=0.0,
                                    x_T=inp["x_T"].clone(), log_every_t=1, test_model_kwargs=kw)
    print(f"  reference full-size PLMS {inp['steps']} steps: {time.time() - t0:.1f}s, apply_model calls = {lm.calls}", flush=True)
    assert lm.calls == inp["steps"] + 1
    xs = inter["x_inter"]
    out = {f"plms_x_{i}": xs[i + 1] for i in rec}
    out["plms_latent"] = z0
    save("full_plms50", **out)                       # written before the (equally long) oracle pass; re-written identically below
    del unet, lm
    t0 = time.time()
    oz, info = O.plms_sample(lambda a, b, cc_: O.unet_forward(sd, a, b, cc_, O.UNET_V1, UP), inp["steps"], inp["x_T"], inp["c"], inp["uc"],
                             inp["scale"], inp["z_inpaint"], inp["mask_lat"], O.schedule_buffers()["alphas_cumprod"], record=rec)
    print(f"  oracle full-size PLMS {inp['steps']} steps: {time.time() - t0:.1f}s", flush=True)
    for i in rec:
        close(info["x"][i], xs[i + 1], f"full-size PLMS-50 x after step {i}", rtol=2e-3)
    close(oz, z0, "full-size PLMS-50 final latent", rtol=2e-3)
    save("full_plms50", **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    todo = a.only.split(",") if a.only else ["primitives", "blocks", "narrow", "manifest"] + (["full"] if a.full else [])
    if "primitives" in todo:
        gen_primitives()
    if "blocks" in todo:
        gen_blocks()
    if "narrow" in todo:
        gen_narrow_and_pipeline()
    if "manifest" in todo:
        gen_full_manifest()
    if "full" in todo:
        gen_full()
    if "full_plms" in todo:
        gen_full_plms()
    if "full_plms50" in todo:
        gen_full_plms50()
    if "sampler_options" in todo:
        gen_sampler_options()
