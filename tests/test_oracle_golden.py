"""CPU suite (-m "not gpu"), part 1: the ORACLE against the golden vectors that were produced by
the REFERENCE modules (oracle/gen_golden.py, run in the build container).  This is what pins the
oracle: if pbe_oracle.py drifts from the reference arithmetic, these fail without any GPU."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle_loader import O
from pbe_amd.weights import synth_state_dict, synth_tensor


def _close(a, b, rtol=2e-4, what=""):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-12
    assert err <= rtol * ref, f"{what}: max|d|={err:.3e} vs max|ref|={ref:.3e}"


@pytest.fixture(scope="module")
def gold(golden_dir):
    return {n: np.load(os.path.join(golden_dir, n + ".npz")) for n in ("primitives", "blocks", "narrow")}


def _keys(golden_dir, name):
    out = {}
    with open(os.path.join(golden_dir, name)) as f:
        for line in f:
            k, s = line.split()
            out[k] = tuple(int(x) for x in s.split("x"))
    return out


@pytest.fixture(scope="module")
def narrow_sd(golden_dir):
    return synth_state_dict(_keys(golden_dir, "narrow_keys.txt"))


def test_schedule_known_answers(gold):
    """ddpm.py:175-197 + util.py:21-74 known answers."""
    g = gold["primitives"]
    sb = O.schedule_buffers()
    assert np.allclose(g["betas_0_999"], [0.00085, 0.012], rtol=1e-12)
    assert abs(sb["alphas_cumprod"][0] - 0.99915) < 1e-6 and abs(sb["alphas_cumprod"][999] - 0.0046601) < 1e-6
    for S in (50, 100):
        t = O.ddim_timesteps_uniform(S)
        assert np.array_equal(t, g[f"ddim_t_{S}"])
        _, a, ap = O.ddim_parameters(sb["alphas_cumprod"], t)
        assert np.array_equal(a, g[f"ddim_a_{S}"]) and np.array_equal(ap, g[f"ddim_aprev_{S}"])
    assert list(O.ddim_timesteps_uniform(50)[:3]) == [1, 21, 41] and O.ddim_timesteps_uniform(50)[-1] == 981
    assert abs(g["ddim_a_50"][0] - 0.9983) < 1e-4 and abs(g["ddim_a_50"][-1] - 0.0058) < 1e-4


def test_timestep_embedding(gold):
    g = gold["primitives"]
    assert torch.equal(O.timestep_embedding(torch.from_numpy(g["temb_t"]), 320), torch.from_numpy(g["temb"]))


def test_blocks(gold):
    g = gold["blocks"]
    T = lambda k: torch.from_numpy(g[k])      # noqa: E731

    def sd(prefix, shapes):
        return {prefix + k: synth_tensor(prefix + k, s) for k, s in shapes.items()}

    res = {"in_layers.0.weight": (64,), "in_layers.0.bias": (64,), "in_layers.2.weight": (128, 64, 3, 3), "in_layers.2.bias": (128,),
           "emb_layers.1.weight": (128, 256), "emb_layers.1.bias": (128,), "out_layers.0.weight": (128,), "out_layers.0.bias": (128,),
           "out_layers.3.weight": (128, 128, 3, 3), "out_layers.3.bias": (128,), "skip_connection.weight": (128, 64, 1, 1), "skip_connection.bias": (128,)}
    _close(O.res_block(sd("res_skip.", res), "res_skip.", T("res_skip_x"), T("res_skip_emb")), g["res_skip_y"], what="ResBlock 64->128")
    vab = {"norm.weight": (64,), "norm.bias": (64,)}
    for n in ("q", "k", "v", "proj_out"):
        vab[n + ".weight"], vab[n + ".bias"] = (64, 64, 1, 1), (64,)
    _close(O.vae_attn(sd("vab.", vab), "vab.", T("st_x")), g["vab_y"], what="VAE AttnBlock")
    _close(O.posterior_sample(T("post_mom"), T("post_eps")), g["post_z"], what="posterior sample")


def test_narrow_unet_vae_clip(gold, narrow_sd):
    g = gold["narrow"]
    inp = cases.narrow_inputs()
    with torch.no_grad():
        y = O.unet_forward(narrow_sd, inp["unet_x"], inp["unet_t"], inp["unet_ctx"], cases.UNET_NARROW, "model.diffusion_model.")
        _close(y, g["unet_y"], what="narrow UNet forward")
        c = O.learned_conditioning(narrow_sd, inp["ref"], cases.CLIP_NARROW, cases.MAPPER_NARROW)
        _close(c, g["c"], what="conditioning")
        z = O.first_stage_encode(narrow_sd, inp["image"] * inp["mask"], inp["post_eps"], cases.VAE_NARROW, "first_stage_model.")
        _close(z, g["z_inpaint"], what="z_inpaint")
        _close(O.first_stage_decode(narrow_sd, inp["x_T"], cases.VAE_NARROW, "first_stage_model."), g["decoded_xT"], what="decode")
        _close(O.resize_mask(inp["mask"], (16, 16), True), g["mask_lat"], what="mask resize (antialias)")
        _close(O.resize_mask(inp["mask"], (16, 16), False), g["mask_lat_noaa"], what="mask resize (no antialias)")


def test_narrow_plms_and_ddim(gold, narrow_sd):
    """50-step PLMS trajectory of the reference sampler (51 model calls) and 20-step DDIM."""
    g = gold["narrow"]
    inp = cases.narrow_inputs()
    c, z_inp, m = torch.from_numpy(g["c"]), torch.from_numpy(g["z_inpaint"]), torch.from_numpy(g["mask_lat"])
    uc = narrow_sd["learnable_vector"].repeat(2, 1, 1)
    ac = O.schedule_buffers()["alphas_cumprod"]
    model = lambda a, b, cc: O.unet_forward(narrow_sd, a, b, cc, cases.UNET_NARROW, "model.diffusion_model.")      # noqa: E731
    with torch.no_grad():
        z0, info = O.plms_sample(model, 50, inp["x_T"], c, uc, 5.0, z_inp, m, ac, record=cases.PLMS_RECORD)
        assert info["calls"] == 51 == int(g["plms_calls_50"])
        for i in cases.PLMS_RECORD:
            _close(info["x"][i], g[f"plms_x_{i}"], rtol=2e-3, what=f"PLMS step {i}")
        _close(z0, g["plms_latent"], rtol=2e-3, what="PLMS latent")
        zd, dinfo = O.ddim_sample(model, 20, inp["x_T"], c, uc, 5.0, z_inp, m, ac)
        assert dinfo["calls"] == 20
        _close(zd, g["ddim_latent"], rtol=2e-3, what="DDIM latent")


def test_sampler_options_against_reference(golden_dir, narrow_sd):
    """The reference samplers' noise-drawing options (DDIM eta > 0 ddim.py:226-238; mask / x0 blending plms.py:150-153, ddim.py:178-181;
    timesteps= prefix plms.py:132-139) run by the REFERENCE with injected noise (tests/golden/sampler_options.npz) against the oracle."""
    g = np.load(os.path.join(golden_dir, "sampler_options.npz"))
    inp = cases.sampler_option_inputs()
    ac = O.schedule_buffers()["alphas_cumprod"]
    model = lambda a, b, cc: O.unet_forward(narrow_sd, a, b, cc, cases.UNET_NARROW, "model.diffusion_model.")      # noqa: E731
    args = (inp["x_T"], inp["c"], inp["uc"], 5.0, inp["z_inpaint"], inp["mask_lat"], ac)
    with torch.no_grad():
        za, _ = O.ddim_sample(model, 10, *args, eta=0.7, noises=inp["noises"], temperature=0.9)
        _close(za, g["ddim_eta_latent"], rtol=2e-3, what="DDIM eta 0.7")
        zb, _ = O.ddim_sample(model, 10, *args, blend=(inp["blend_mask"], inp["x0"], inp["noises"]))
        _close(zb, g["ddim_blend_latent"], rtol=2e-3, what="DDIM mask + x0")
        zc, info = O.plms_sample(model, 12, *args, timesteps=8, blend=(inp["blend_mask"], inp["x0"], inp["noises"]))
        assert info["calls"] == int(g["plms_blend_subset_calls"]) == 8            # 7 steps of the 12-step schedule + the extra first-step call
        _close(zc, g["plms_blend_subset_latent"], rtol=2e-3, what="PLMS mask + x0, schedule prefix")


def test_plms_call_count_100_steps():
    """S = 100 -> 101 model evaluations (SURVEY.md §8c known answer), counted with a fake model."""
    n = {"c": 0}

    def fake(x, t, c):
        n["c"] += 1
        return torch.zeros(x.shape[0], 4, *x.shape[2:])

    z = torch.zeros(1, 4, 2, 2)
    O.plms_sample(fake, 100, z, torch.zeros(1, 1, 8), None, 1.0, z, torch.zeros(1, 1, 2, 2), O.schedule_buffers()["alphas_cumprod"])
    assert n["c"] == 101


def test_gelu_fit_constants():
    """The HIP kernels' exact-GELU (pbe_amd/csrc/common.h gelu_erf_f: relu(x) - |x| exp2(-|x| R(|x|) - 1), R a degree-4 fit) against
    x Phi(x) in float64 (torch.nn.functional.gelu's definition, the one the reference's GEGLU calls: ldm/modules/attention.py:46-53),
    evaluated in float32 like the kernel: |error| <= 1e-6 over the whole range, no blow-up beyond the fitted range."""
    import os
    import re
    from scipy.special import erf
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pbe_amd", "csrc", "common.h")).read()
    c = [np.float32(float(re.search(rf"#define PBE_GELU_R{k} \(?(-?[0-9.e-]+)f\)?", src).group(1))) for k in range(5)]
    assert c[4] > 0                                                  # q -> 0 far out
    x = np.concatenate([np.linspace(-14, 14, 1000001), [0.0, 1e-8, -1e-8, 100, -100, 65504, -65504, 1e30, -1e30]]).astype(np.float32)
    u = np.abs(x)
    r = np.full_like(u, c[4])
    with np.errstate(over="ignore"):
        for k in (3, 2, 1, 0):
            r = r * u + c[k]
        y = np.maximum(x, np.float32(0)) - u * np.exp2(-u * r - np.float32(1))
    x64 = x.astype(np.float64)
    exact = x64 * 0.5 * (1.0 + erf(x64 / np.sqrt(2.0)))
    assert np.isfinite(y).all()
    err = np.abs(y.astype(np.float64) - exact)
    assert (err <= np.maximum(1e-6, 2.0 ** -22 * np.abs(exact))).all(), err.max()
