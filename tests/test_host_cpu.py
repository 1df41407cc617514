"""CPU suite (-m "not gpu"), part 2: host logic and the drop-in boundary — no kernel is launched.

  * libpbe_hip.so loads and exports every symbol include/pbe_hip.h declares,
  * the module tree built from configs/v1.yaml has exactly the reference's state_dict keys/shapes
    (tests/golden/v1_keys.txt was written from the reference modules),
  * checkpoint key remap across `transformers` versions,
  * sampler host logic (schedule, call order, multistep weights) with the HIP ops replaced, IN THE
    TEST ONLY, by the oracle's arithmetic,
  * the product path fails loudly on CPU tensors (no fallback).
"""
import os
import re
import sys

import numpy as np
import pytest
import torch

import cases
from oracle_loader import O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _keys(path):
    out = {}
    with open(path) as f:
        for line in f:
            k, s = line.split()
            out[k] = tuple(int(x) for x in s.split("x"))
    return out


def test_library_exports_every_declared_symbol():
    from pbe_amd import lib
    from pbe_amd.build import build
    build()
    handle = lib.load()
    header = open(os.path.join(ROOT, "include", "pbe_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)              # the prose mentions the names too
    declared = set(re.findall(r"\b(pbe_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(lib.SYMBOLS), (declared ^ set(lib.SYMBOLS))
    for name in declared:
        assert getattr(handle, name) is not None
    assert handle.pbe_abi_version() == lib.ABI_VERSION
    m = re.search(r"#define PBE_ABI_VERSION (\d+)", header)
    assert int(m.group(1)) == lib.ABI_VERSION


def test_v1_state_dict_matches_reference_manifest(golden_dir):
    from ldm.util import instantiate_from_config, load_yaml_config
    cfg = load_yaml_config(os.path.join(ROOT, "configs", "v1.yaml"))
    assert cfg["model"]["target"] == "ldm.models.diffusion.ddpm.LatentDiffusion"
    with torch.device("meta"):
        model = instantiate_from_config(cfg["model"])
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    ref = _keys(os.path.join(golden_dir, "v1_keys.txt"))
    assert not [k for k in ref if k not in mine]
    assert not [k for k in ref if mine[k] != ref[k]]
    unet = [k for k in mine if k.startswith("model.diffusion_model.")]
    assert len(unet) == 686 and sum(int(np.prod(mine[k])) for k in unet) == 859_535_364
    clip = [k for k in mine if k.startswith("cond_stage_model.transformer.vision_model.")]
    assert len(clip) == 392 and "cond_stage_model.transformer.vision_model.pre_layrnorm.weight" in mine
    # also importable under the fork's module path
    from ldm.models.diffusion.latent_diffusion import LatentDiffusion as A
    from ldm.models.diffusion.ddpm import LatentDiffusion as B
    assert A is B


def test_checkpoint_key_remap():
    from pbe_amd.weights import canonical_checkpoint_keys
    new_style = {"cond_stage_model.transformer.embeddings.class_embedding": 1, "cond_stage_model.mapper.resblocks.0.ln_1.weight": 2,
                 "model_ema.decay": 3, "model.diffusion_model.out.2.bias": 4}
    out = canonical_checkpoint_keys(new_style)
    assert "cond_stage_model.transformer.vision_model.embeddings.class_embedding" in out and "model_ema.decay" not in out
    old_style = {"cond_stage_model.transformer.vision_model.embeddings.class_embedding": 1}
    assert canonical_checkpoint_keys(old_style) == old_style


def test_schedule_buffers_match_oracle():
    from ldm.models.diffusion.ddpm import DDPM
    from ldm.modules.diffusionmodules.util import make_ddim_sampling_parameters, make_ddim_timesteps
    sb = O.schedule_buffers()
    m = DDPM.__new__(DDPM)
    torch.nn.Module.__init__(m)
    m.v_posterior = 0.0
    m.register_schedule(linear_start=0.00085, linear_end=0.0120, timesteps=1000)
    assert np.array_equal(m.betas.numpy(), sb["betas"]) and np.array_equal(m.alphas_cumprod.numpy(), sb["alphas_cumprod"])
    t = make_ddim_timesteps("uniform", 50, 1000, verbose=False)
    assert np.array_equal(t, O.ddim_timesteps_uniform(50))
    s, a, ap = make_ddim_sampling_parameters(sb["alphas_cumprod"], t, 0.0, verbose=False)
    s2, a2, ap2 = O.ddim_parameters(sb["alphas_cumprod"], t)
    assert np.array_equal(a, a2) and np.array_equal(ap, ap2) and np.array_equal(s, s2)


class _FakeUNet:
    """Stands in for model.model.diffusion_model: eps = oracle narrow U-Net on NHWC fp16-shaped inputs."""

    def __init__(self, sd):
        self.sd, self.calls, self.batches = sd, 0, []

    def forward_nhwc(self, x9, t, ctx, paired=False, step=None):
        self.calls += 1
        assert step is not None and bool((t == int(step)).all())      # the samplers vouch that every row of t is `step` (cached embedding rows)
        if paired:                                   # shared-prefix guidance call: B inputs, 2B timesteps / contexts
            assert x9.shape[0] * 2 == t.shape[0] == ctx.shape[0]
            x9 = torch.cat([x9, x9])
        self.batches.append(int(x9.shape[0]))
        x = x9.float()[..., :9].permute(0, 3, 1, 2)
        y = O.unet_forward(self.sd, x, t, ctx.float(), cases.UNET_NARROW, "model.diffusion_model.")
        return y.permute(0, 2, 3, 1).contiguous()


def test_plms_host_logic_with_oracle_ops(monkeypatch, golden_dir):
    """PLMSSampler's control flow (51 calls, CFG doubling, both kwarg spellings, uc broadcast, multistep
    weights) reproduces the reference trajectory when ONLY the two element-wise kernels are swapped
    for their oracle arithmetic inside this test."""
    import types
    from ldm.models.diffusion import plms as P
    from pbe_amd.weights import synth_state_dict
    sd = synth_state_dict(_keys(os.path.join(golden_dir, "narrow_keys.txt")))
    g = np.load(os.path.join(golden_dir, "narrow.npz"))

    def pack(x, z, m, dup):
        x9 = torch.cat([x, z, m], 1).permute(0, 2, 3, 1)
        x9 = torch.cat([x9, torch.zeros(*x9.shape[:3], 7)], -1)
        return torch.cat([x9] * dup)

    def update(eps, dup, scale, x, hist, coef, want_e_t=True, want_pred=True):
        e = eps.float().permute(0, 3, 1, 2)[:, :4]
        if dup == 2:
            eu, ec = e.chunk(2)
            e = eu + scale * (ec - eu)
        ep = coef[0] * e
        for h, c in zip(hist, coef[1:4]):
            ep = ep + c * h
        px0 = (x - coef[4] * ep) * coef[5]
        return coef[6] * px0 + coef[7] * ep, px0, e

    monkeypatch.setattr(P.ops, "plms_pack_input", pack)
    monkeypatch.setattr(P.ops, "plms_update", update)
    sb = O.schedule_buffers()
    unet = _FakeUNet(sd)
    model = types.SimpleNamespace(num_timesteps=1000, betas=torch.from_numpy(sb["betas"]), alphas_cumprod=torch.from_numpy(sb["alphas_cumprod"]),
                                  alphas_cumprod_prev=torch.from_numpy(sb["alphas_cumprod_prev"]),
                                  model=types.SimpleNamespace(diffusion_model=unet))
    inp = cases.narrow_inputs()
    c, z_inp, m = torch.from_numpy(g["c"]), torch.from_numpy(g["z_inpaint"]), torch.from_numpy(g["mask_lat"])
    smp = P.PLMSSampler(model)
    with pytest.raises(Exception):                                            # a CPU model is refused by default
        smp.sample(S=50, batch_size=2, shape=[4, 16, 16], conditioning=c, verbose=False, x_T=inp["x_T"],
                   test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": m})
    smp.require_gpu = False                                                   # host-logic test: kernels replaced above
    with torch.no_grad():
        z0, inter = smp.sample(S=50, batch_size=2, shape=[4, 16, 16], conditioning=c, verbose=False, unconditional_guidance_scale=5.0,
                               unconditional_conditioning=sd["learnable_vector"], eta=0.0, x_T=inp["x_T"], log_every_t=1,
                               test_model_kwargs={"inpaint_image": z_inp, "inpaint_mask": m})
    assert unet.calls == 51 and set(unet.batches) == {4}                      # CFG doubles the batch; [1,1,768] uc broadcast to B
    assert np.array_equal(smp.ddim_timesteps, g_t := O.ddim_timesteps_uniform(50)) and g_t[0] == 1
    err = (z0 - torch.from_numpy(g["plms_latent"])).abs().max().item()
    assert err <= 2e-3 * np.abs(g["plms_latent"]).max(), err
    assert len(inter["x_inter"]) == 51
    with pytest.raises(ValueError):
        smp.make_schedule(50, ddim_eta=0.5, verbose=False)


def test_product_path_fails_loudly_without_gpu():
    from ldm.modules.attention import CrossAttention
    from ldm.modules.diffusionmodules.openaimodel import ResBlock
    from pbe_amd import ops
    from pbe_amd.lib import PbeError
    with pytest.raises(PbeError):
        ops.groupnorm(torch.zeros(1, 4, 32, dtype=torch.float16), torch.ones(32), torch.zeros(32), 1e-5, True)
    with pytest.raises(PbeError):
        ResBlock(64, 128, 0.0)(torch.zeros(1, 64, 8, 8), torch.zeros(1, 128))
    with pytest.raises(PbeError):
        CrossAttention(64, heads=8, dim_head=8)(torch.zeros(1, 16, 64))


def test_unsupported_reference_options_raise():
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from pbe_amd.lib import PbeError
    with pytest.raises(PbeError):
        UNetModel(image_size=32, in_channels=9, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=[1], num_heads=8)
    with pytest.raises(PbeError):
        UNetModel(image_size=32, in_channels=9, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=[1], num_heads=8,
                  use_spatial_transformer=True, context_dim=768, resblock_updown=True)


def test_instantiate_from_config_and_tuned_table():
    import json
    from ldm.util import get_obj_from_str, instantiate_from_config
    assert instantiate_from_config({"target": "torch.nn.Identity"}).__class__.__name__ == "Identity"
    assert get_obj_from_str("ldm.models.diffusion.plms.PLMSSampler").__name__ == "PLMSSampler"
    with pytest.raises(KeyError):
        instantiate_from_config({"params": {}})
    table = json.load(open(os.path.join(ROOT, "pbe_amd", "tuned_mi355x.json")))
    assert table and all(0 <= (int(v) & 255) <= 21 and 0 <= (int(v) >> 8) <= 64 for v in table.values())     # tile config | split-K factor << 8


def test_preprocessing_of_bundled_example(golden_dir):
    """scripts/inference.py:306-318 on the reference's own examples/example_1 triple (data files copied
    into tests/golden/examples/): the product's loader against the oracle's statement of the formulas."""
    from PIL import Image
    from pbe_amd import preprocess
    d = os.path.join(golden_dir, "examples")
    t = preprocess.load_triple(os.path.join(d, "image_example_1.png"), os.path.join(d, "mask_example_1.png"), os.path.join(d, "reference_example_1.jpg"))
    img = np.asarray(Image.open(os.path.join(d, "image_example_1.png")).convert("RGB"))
    msk = np.asarray(Image.open(os.path.join(d, "mask_example_1.png")).convert("L"))
    ref = np.asarray(Image.open(os.path.join(d, "reference_example_1.jpg")).convert("RGB").resize((224, 224)))
    oi, oinp, om, oref = O.preprocess_triple(img, msk, ref)
    assert t["image"].shape == (1, 3, 512, 512) and t["ref"].shape == (1, 3, 224, 224)
    assert torch.equal(t["image"], oi) and torch.equal(t["mask"], om) and torch.equal(t["inpaint"], oinp)
    assert torch.allclose(t["ref"], oref, atol=1e-6)
    assert set(torch.unique(t["mask"]).tolist()) == {0.0, 1.0} and 0.05 < 1 - t["mask"].mean().item() < 0.6
    from pbe_amd.lib import PbeError
    from pbe_amd.pipeline import resize_mask
    with pytest.raises(PbeError):                       # the product's mask resize is a HIP kernel: no CPU fallback
        resize_mask(t["mask"], (64, 64))


# ---- checkpoint files (reference scripts/inference.py:58-75, ddpm.py:245-260) -----------------------------
def _narrow_reference_state_dict(golden_dir):
    from pbe_amd.weights import synth_tensor
    sd = {}
    with open(os.path.join(golden_dir, "narrow_keys.txt")) as f:
        for line in f:
            k, s = line.split()
            sd[k] = synth_tensor(k, tuple(int(x) for x in s.split("x")) if s != "scalar" else ())
    return sd


def test_lightning_checkpoint_is_read_without_its_packages(tmp_path, golden_dir):
    """A Lightning-layout file whose pickle names pytorch_lightning / omegaconf classes: the weights-only loader refuses it,
    the restricted reader returns exactly the tensors and resolves none of the foreign globals."""
    import pickle
    import ckpt_synth
    from pbe_amd.checkpoint import read_state_dict
    from pbe_amd.weights import canonical_checkpoint_keys
    ref = {k: v for k, v in _narrow_reference_state_dict(golden_dir).items() if k.startswith(("model.diffusion_model.input_blocks.0", "cond_stage_model.transformer",
                                                                                               "first_stage_model.quant", "proj_out", "learnable_vector", "betas"))}
    path = str(tmp_path / "model.ckpt")
    ckpt_synth.write_lightning_checkpoint(path, ref)
    with pytest.raises(pickle.UnpicklingError):
        torch.load(path, map_location="cpu", weights_only=True)
    assert "pytorch_lightning" not in sys.modules and "omegaconf" not in sys.modules
    sd = read_state_dict(path)
    assert "pytorch_lightning" not in sys.modules and "omegaconf" not in sys.modules
    assert any(k.startswith("model_ema.") for k in sd) and "model_ema.decay" in sd
    can = canonical_checkpoint_keys(sd)
    assert set(can) == set(ref)
    for k, v in ref.items():
        assert can[k].dtype == v.dtype and torch.equal(can[k], v), k
    assert tuple(can["model.diffusion_model.input_blocks.0.0.weight"].shape)[1:] == (9, 3, 3)          # the 9-channel conv-in (modify_checkpoints.py:1-6)
    # transformers >= 5 spelling of the CLIP keys maps back to the reference's `vision_model.` names
    path5 = str(tmp_path / "model_hf5.ckpt")
    ckpt_synth.write_lightning_checkpoint(path5, ref, hf5_clip_names=True, foreign=False)
    sd5 = read_state_dict(path5)
    assert not any("vision_model" in k for k in sd5)
    can5 = canonical_checkpoint_keys(sd5)
    assert set(can5) == set(ref) and all(torch.equal(can5[k], ref[k]) for k in ref)
    # a bare state dict (what `torch.save(sd)` of scripts/modify_checkpoints.py writes) is accepted too
    bare = str(tmp_path / "bare.ckpt")
    torch.save({"state_dict": {k: v for k, v in ref.items()}}, bare)
    assert set(read_state_dict(bare)) == set(ref)


def test_restricted_reader_executes_nothing(tmp_path):
    """A pickle that would run code on a full unpickle: the reader maps the callable to an inert placeholder."""
    import pickle
    from pbe_amd.checkpoint import CheckpointError, read_state_dict
    marker = tmp_path / "pwned"

    class Boom:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))

    path = str(tmp_path / "evil.ckpt")
    torch.save({"state_dict": {"w": torch.arange(6.0).reshape(2, 3)}, "callbacks": Boom()}, path)
    with pytest.raises(pickle.UnpicklingError):
        torch.load(path, map_location="cpu", weights_only=True)
    sd = read_state_dict(path)
    assert not marker.exists()
    assert list(sd) == ["w"] and torch.equal(sd["w"], torch.arange(6.0).reshape(2, 3))
    with open(str(tmp_path / "junk.ckpt"), "wb") as f:
        f.write(b"not a checkpoint")
    with pytest.raises((CheckpointError, pickle.UnpicklingError, Exception)):
        read_state_dict(str(tmp_path / "junk.ckpt"))


def test_loader_refuses_a_stale_binary(monkeypatch):
    """The .so is git-ignored and travels with the tree: a kernel edit without a rebuild must not run the old binary silently."""
    from pbe_amd import lib
    handle = lib.load()
    assert handle.pbe_source_hash().decode() == lib.source_hash()
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "source_hash", lambda: "0123456789abcdef")
    with pytest.raises(lib.PbeError, match="built from other sources"):
        lib.load()
