"""Parity of the HIP path (this repo's ``ldm`` modules -> libpbe_hip.so) against golden vectors
produced by the REFERENCE modules in fp32 (tests/golden/*.npz, oracle/gen_golden.py), on
identical name-seeded weights and seeded inputs.

Stated fp16 tolerance.  The reference's own "fp16" path (torch.autocast: fp16 GEMM/conv with fp32
accumulate, fp32 norms/softmax, fp32 master weights) deviates from its fp32 result by
rel-L2 = 2.06e-3 on the narrow U-Net (measured with the reference modules on CPU and stored in
narrow.npz as ``unet_autocast_fp16_rel_l2``).  We hold the HIP path to the same yardstick:

    single network forward (U-Net / VAE / CLIP):   rel-L2 <= 4e-3   (2x the reference's own drift)
    50-step PLMS trajectory (51 chained U-Net calls, CFG scale 5 amplifies per-call error):
                                                   rel-L2 <= 3e-2 on the final latent,
                                                   mean |d| <= 1e-2 on the final [0,1] image
"""
import os

import numpy as np
import pytest
import torch

import modelbuild as build
import cases

pytestmark = pytest.mark.gpu

FWD_TOL = 4e-3            # one network forward (measured 1.4e-3 .. 2.0e-3; the reference's own autocast-fp16 drift is 2.06e-3)
BLOCK_TOL = 2e-3          # one block against its reference module (measured 3.2e-4 .. 7.2e-4)
SAMPLER_OPT_TOL = 6e-3     # 10-step narrow trajectories with injected noise (<= 3x measured)
TRAJ50_TOL = 2.7e-3        # headline-size 50-step trajectory: measured 4.0e-4 (step 0), 9.3e-4, 8.5e-4, 8.6e-4 (final) - profiles/r03_parity_report.txt
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.txt")


def rel_l2(got, ref):
    got, ref = got.detach().float().cpu().double(), torch.as_tensor(ref).double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return ((got - ref).norm() / ref.norm()).item()


def report(name, val, tol):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(f"{name:48s} rel_l2={val:.3e}  tol={tol:.1e}  {'OK' if val <= tol else 'FAIL'}\n")


def check(name, got, ref, tol=FWD_TOL):
    v = rel_l2(got, ref)
    report(name, v, tol)
    assert v <= tol, f"{name}: rel-L2 {v:.3e} > {tol:.1e}"


@pytest.fixture(scope="module")
def gold(golden_dir):
    return {n: np.load(os.path.join(golden_dir, n + ".npz")) for n in ("blocks", "narrow", "full")}


@pytest.fixture(scope="module")
def narrow(dev):
    with torch.no_grad():
        return build.narrow_model(dev)


# ---- per-block goldens (reference ResBlock / SpatialTransformer / VAE blocks) ----------------
def test_blocks_against_reference(dev, gold):
    from ldm.modules.attention import FeedForward, SpatialTransformer
    from ldm.modules.diffusionmodules import model as vae
    from ldm.modules.diffusionmodules.openaimodel import Downsample, ResBlock, Upsample
    from ldm.modules.encoders.xf import LayerNorm, Transformer
    from pbe_amd.weights import fill_module_
    g = gold["blocks"]
    T = lambda k: torch.from_numpy(g[k]).to(dev)      # noqa: E731
    with torch.no_grad():
        for tag, cin, cout in (("res_skip", 64, 128), ("res_id", 128, 128)):
            m = ResBlock(cin, 256, 0.0, out_channels=cout)
            fill_module_(m, prefix=tag + ".")
            check(f"ResBlock {cin}->{cout}", m.to(dev)(T(tag + "_x"), T(tag + "_emb")), g[tag + "_y"], BLOCK_TOL)
        st = SpatialTransformer(64, 8, 8, depth=1, context_dim=768)
        fill_module_(st, prefix="st.")
        check("SpatialTransformer C=64", st.to(dev)(T("st_x"), T("st_ctx")), g["st_y"], BLOCK_TOL)
        ff = FeedForward(64, glu=True)
        fill_module_(ff, prefix="ff.")
        check("FeedForward GEGLU", ff.to(dev)(T("ff_x")), g["ff_y"], BLOCK_TOL)
        dn, up = Downsample(64, True, out_channels=64), Upsample(64, True, out_channels=64)
        fill_module_(dn, prefix="dn."), fill_module_(up, prefix="up.")
        check("Downsample", dn.to(dev)(T("st_x")), g["dn_y"], BLOCK_TOL)
        check("Upsample", up.to(dev)(T("st_x")), g["up_y"], BLOCK_TOL)
        from pbe_amd import ops
        x_nhwc = lambda: ops.nchw_to_nhwc(T("st_x"))      # noqa: E731
        rb = vae.ResnetBlock(in_channels=64, out_channels=128, dropout=0.0, temb_channels=0)
        fill_module_(rb, prefix="vrb.")
        check("VAE ResnetBlock", ops.nhwc_to_nchw(rb.to(dev).run(x_nhwc())), g["vrb_y"], BLOCK_TOL)
        ab = vae.AttnBlock(64)
        fill_module_(ab, prefix="vab.")
        check("VAE AttnBlock", ops.nhwc_to_nchw(ab.to(dev).run(x_nhwc())), g["vab_y"], BLOCK_TOL)
        vd = vae.Downsample(64, True)
        fill_module_(vd, prefix="vdn.")
        check("VAE Downsample (asym pad)", ops.nhwc_to_nchw(vd.to(dev).run(x_nhwc())), g["vdn_y"], BLOCK_TOL)
        tr, ln = Transformer(1, 128, 2, 1), LayerNorm(128)
        fill_module_(tr, prefix="mapper."), fill_module_(ln, prefix="final_ln.")
        check("xf mapper + final_ln", ln.to(dev)(tr.to(dev)(T("map_z"))), g["map_y"], BLOCK_TOL)


# ---- narrow whole networks ------------------------------------------------------------------
def test_narrow_unet_forward(dev, gold, narrow):
    inp = cases.narrow_inputs()
    with torch.no_grad():
        y = narrow.apply_model(inp["unet_x"].to(dev), inp["unet_t"].to(dev), inp["unet_ctx"].to(dev))
    assert y.dtype == torch.float16 and y.shape == (4, 4, 16, 16)
    ref_drift = float(gold["narrow"]["unet_autocast_fp16_rel_l2"])
    report("(reference autocast-fp16 vs its fp32, narrow U-Net)", ref_drift, FWD_TOL)
    check("narrow UNetModel forward", y, gold["narrow"]["unet_y"])


def test_narrow_clip_and_conditioning(dev, gold, narrow):
    inp = cases.narrow_inputs()
    with torch.no_grad():
        pooled = narrow.cond_stage_model.transformer(pixel_values=inp["ref"].to(dev)).pooler_output
        c = narrow.project_conditioning(narrow.get_learned_conditioning(inp["ref"].to(dev)))
    check("narrow CLIP pooled", pooled, gold["narrow"]["clip_pooled"])
    check("narrow get_learned_conditioning+proj_out", c, gold["narrow"]["c"])


def test_narrow_vae(dev, gold, narrow):
    inp = cases.narrow_inputs()
    g = gold["narrow"]
    with torch.no_grad():
        post = narrow.encode_first_stage((inp["image"] * inp["mask"]).to(dev))
        mom = post.parameters.float().permute(0, 3, 1, 2)
        z = narrow.get_first_stage_encoding(post, noise=inp["post_eps"])
        torch.manual_seed(cases.POSTERIOR_SEED)                       # reference RNG contract: CPU generator (distributions.py:36)
        z_rng = narrow.get_first_stage_encoding(post)
        zc = inp["x_T"].to(dev).clone()
        dec = narrow.decode_first_stage(zc)
    check("narrow VAE moments", mom, g["moments"])
    check("narrow encode_first_stage+sample (injected eps)", z, g["z_inpaint"])
    check("narrow encode_first_stage+sample (CPU RNG)", z_rng, g["z_inpaint"])
    check("narrow decode_first_stage", dec, g["decoded_xT"])
    # decode_first_stage un-scales its argument in place like the reference (latent_diffusion.py:454)
    assert torch.allclose(zc.cpu(), inp["x_T"] * (1.0 / 0.18215), rtol=1e-6)


def _sample(narrow, dev, g, sampler_cls, S, spelling):
    inp = cases.narrow_inputs()
    c = torch.from_numpy(g["c"]).to(dev)                  # golden conditioning / latent so only the sampler+U-Net are under test
    uc = narrow.learnable_vector.repeat(2, 1, 1)
    z_inp, m = torch.from_numpy(g["z_inpaint"]).to(dev), torch.from_numpy(g["mask_lat"]).to(dev)
    kw = {"inpaint_image": z_inp, "inpaint_mask": m} if spelling == "inference.py" else {"images_inpaint": z_inp, "images_mask": m}
    calls = {"n": 0}
    unet = narrow.model.diffusion_model
    orig = unet.forward_nhwc

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)

    unet.forward_nhwc = counting
    smp = sampler_cls(narrow)
    try:
        out, inter = smp.sample(S=S, batch_size=2, shape=[4, 16, 16], conditioning=c, verbose=False, unconditional_guidance_scale=5.0,
                                unconditional_conditioning=uc, eta=0.0, x_T=inp["x_T"].to(dev), log_every_t=1, test_model_kwargs=kw)
    finally:
        unet.forward_nhwc = orig
    # U-Net evaluations = Python-level calls, or (HIP-graph regime) one eager call + replays; the capture pass is not an evaluation
    n = calls["n"] if smp._graphed is None else 1 + smp._graphed.replays
    return out, inter, n


def test_narrow_plms_trajectory(dev, gold, narrow):
    from ldm.models.diffusion.plms import PLMSSampler
    g = gold["narrow"]
    with torch.no_grad():
        z0, inter, n = _sample(narrow, dev, g, PLMSSampler, 50, "inference.py")
        img = torch.clamp((narrow.decode_first_stage(z0.clone()) + 1.0) / 2.0, 0.0, 1.0)
    assert n == 51 == int(g["plms_calls_50"])                 # S + 1 U-Net evaluations (plms.py:230-235)
    assert len(inter["x_inter"]) == 51
    # every tolerance below is <= 3x the value measured on MI355X (profiles/r03_parity_report.txt): 5.5e-4, 1.0e-3, 1.1e-3, 1.4e-3, 1.44e-3, 1.44e-3
    for i, tol in zip(cases.PLMS_RECORD, (1.7e-3, 3e-3, 3.4e-3, 4e-3, 4.3e-3, 4.3e-3)):
        check(f"PLMS x after step {i}", inter["x_inter"][i + 1], g[f"plms_x_{i}"], tol)
    check("PLMS final latent", z0, g["plms_latent"], 4.3e-3)
    mad = (img.float().cpu() - torch.from_numpy(g["plms_image"])).abs().mean().item()
    report("PLMS final image mean|d| (as rel_l2 column)", mad, 1.2e-3)                      # measured 4.1e-4
    assert mad <= 1.2e-3


def test_narrow_plms_key_spelling_and_ddim(dev, gold, narrow):
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.plms import PLMSSampler
    g = gold["narrow"]
    with torch.no_grad():
        a, _, _ = _sample(narrow, dev, g, PLMSSampler, 10, "inference.py")
        b, _, _ = _sample(narrow, dev, g, PLMSSampler, 10, "plms.py")
        zd, _, n = _sample(narrow, dev, g, DDIMSampler, 20, "plms.py")
    assert torch.equal(a, b)
    assert n == 20
    check("DDIM 20-step latent", zd, g["ddim_latent"], 5e-3)                                # measured 1.7e-3


def test_cached_timestep_embedding_rows_are_bit_identical(dev, narrow):
    """UNetModel.forward_nhwc(step=t): the time-embedding MLP + the 22 emb_layers evaluated once per timestep VALUE and broadcast to the
    samples (what the samplers pass) against the per-call evaluation on the [2B] timestep tensor - same kernels, same bits."""
    from pbe_amd import ops
    unet = narrow.model.diffusion_model
    g = torch.Generator().manual_seed(31)
    x = ops.nchw_to_nhwc(torch.randn(4, 9, 16, 16, generator=g).to(dev), unet.pk().cin_pad)
    ctx = torch.randn(4, 1, 768, generator=g).to(dev).half()
    unet.invalidate_packs()                                   # (the shared fixture's cache holds the steps of earlier sampler tests)
    with torch.no_grad():
        for step in (981, 501, 981, 1):
            t = torch.full((4,), step, dtype=torch.int64, device=dev)
            a = unet.forward_nhwc(x, t, ctx)
            b = unet.forward_nhwc(x, t, ctx, step=step)
            assert torch.equal(a, b), step
        assert len(unet.__dict__["_emb_cache"]) == 3
        unet.invalidate_packs()
        assert unet.__dict__["_emb_cache"] == {}


def test_proj_out_pack_is_dropped_by_data_writes(dev):
    """ddpm.HipLinear caches its fp16 pack keyed on (data_ptr, _version); writes through ``.data`` - what shard.broadcast_weights_ and
    weights.fill_* do - do not move ``_version``, so those paths call ``invalidate_packs()`` on every module that has one: after it the
    forward must use the NEW weight (round-2 advisor finding: a stale pack made the conditioning silently wrong)."""
    from ldm.models.diffusion.ddpm import HipLinear
    g = torch.Generator().manual_seed(21)
    lin = HipLinear(128, 768).to(dev)
    z = torch.randn(6, 1, 128, generator=g).to(dev)
    with torch.no_grad():
        y0 = lin(z).float()
        w2, b2 = torch.randn(768, 128, generator=g).to(dev) * 0.1, torch.randn(768, generator=g).to(dev)
        v0 = lin.weight._version
        lin.weight.data.copy_(w2)
        lin.bias.data.copy_(b2)
        assert lin.weight._version == v0                                   # the hazard: the cache key cannot see this write
        for m in lin.modules():                                            # what broadcast_weights_ / fill_latent_diffusion_ do after writing
            if hasattr(m, "invalidate_packs"):
                m.invalidate_packs()
        y1 = lin(z).float()
        ref = torch.nn.functional.linear(z.float(), w2, b2)
    assert rel_l2(y1, ref.cpu()) < 2e-3 and rel_l2(y0, ref.cpu()) > 0.5


def test_sampler_options_on_gpu(dev, golden_dir, narrow):
    """Stochastic DDIM (eta 0.7, temperature 0.9), mask / x0 blending in both samplers and the timesteps= prefix on the HIP path against
    the REFERENCE samplers run with the same injected noise (tests/golden/sampler_options.npz)."""
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.plms import PLMSSampler
    g = np.load(os.path.join(golden_dir, "sampler_options.npz"))
    inp = {k: ([t.to(dev) for t in v] if isinstance(v, list) else v.to(dev)) for k, v in cases.sampler_option_inputs().items()}
    kw = dict(conditioning=inp["c"], verbose=False, unconditional_guidance_scale=5.0, unconditional_conditioning=inp["uc"],
              test_model_kwargs={"images_inpaint": inp["z_inpaint"], "images_mask": inp["mask_lat"]})

    def sampler(cls):
        s = cls(narrow)
        it = iter(inp["noises"])
        s.noise_like = lambda shape, device: next(it)
        return s
    with torch.no_grad():
        za, _ = sampler(DDIMSampler).sample(S=10, batch_size=2, shape=[4, 16, 16], eta=0.7, temperature=0.9, x_T=inp["x_T"].clone(), **kw)
        check("DDIM eta 0.7, temperature 0.9, 10 steps (injected noise)", za, g["ddim_eta_latent"], SAMPLER_OPT_TOL)
        zb, _ = sampler(DDIMSampler).sample(S=10, batch_size=2, shape=[4, 16, 16], eta=0.0, x_T=inp["x_T"].clone(), mask=inp["blend_mask"], x0=inp["x0"], **kw)
        check("DDIM mask + x0 blending, 10 steps", zb, g["ddim_blend_latent"], SAMPLER_OPT_TOL)
        zc, _ = sampler(PLMSSampler).sample(S=12, batch_size=2, shape=[4, 16, 16], eta=0.0, x_T=inp["x_T"].clone(), mask=inp["blend_mask"], x0=inp["x0"],
                                             timesteps=8, **kw)
        check("PLMS mask + x0 blending, timesteps = 8 of 12", zc, g["plms_blend_subset_latent"], SAMPLER_OPT_TOL)
        # eta > 0 without injected noise draws from the device generator: runs, finite, differs between two draws
        r1, _ = DDIMSampler(narrow).sample(S=4, batch_size=2, shape=[4, 16, 16], eta=1.0, x_T=inp["x_T"].clone(), **kw)
        r2, _ = DDIMSampler(narrow).sample(S=4, batch_size=2, shape=[4, 16, 16], eta=1.0, x_T=inp["x_T"].clone(), **kw)
    assert torch.isfinite(r1).all() and not torch.equal(r1, r2)


def test_hip_graph_replay_is_bit_identical(dev, gold, narrow):
    """pbe_amd.graph: the U-Net call captured once into a HIP graph and replayed (launch-bound regime) produces the same
    bits as eager launches over a 10-step PLMS run, and really replays (10 of the 11 calls)."""
    from ldm.models.diffusion.plms import PLMSSampler
    g = gold["narrow"]
    outs = {}
    with torch.no_grad():
        for use in (False, True):
            smp = PLMSSampler(narrow)
            smp.use_graph = use
            inp = cases.narrow_inputs()
            z0, _ = smp.sample(S=10, batch_size=2, shape=[4, 16, 16], conditioning=torch.from_numpy(g["c"]).to(dev), verbose=False,
                               unconditional_guidance_scale=5.0, unconditional_conditioning=narrow.learnable_vector, eta=0.0,
                               x_T=inp["x_T"].to(dev), test_model_kwargs={"inpaint_image": torch.from_numpy(g["z_inpaint"]).to(dev),
                                                                          "inpaint_mask": torch.from_numpy(g["mask_lat"]).to(dev)})
            outs[use] = z0.clone()
            if use:
                assert smp._graphed is not None and smp._graphed.replays == 10
            else:
                assert smp._graphed is None
    assert torch.equal(outs[False], outs[True])


# ---- full-size (configs/v1.yaml) single forwards -------------------------------------------------
@pytest.fixture(scope="module")
def full(dev):
    with torch.no_grad():
        return build.full_model(dev)


def test_full_unet_forward(dev, gold, full):
    inp = cases.full_inputs()
    with torch.no_grad():
        y = full.apply_model(inp["unet_x"].to(dev), inp["unet_t"].to(dev), inp["unet_ctx"].to(dev))
    check("v1 UNetModel forward (859.5M params, 64x64)", y, gold["full"]["unet_y"])


def test_full_vae(dev, gold, full):
    inp = cases.full_inputs()
    g = gold["full"]
    with torch.no_grad():
        mom = full.encode_first_stage(inp["image"].to(dev)).parameters.float().permute(0, 3, 1, 2)
        dec = full.decode_first_stage(inp["z_dec"].to(dev).clone())
    check("v1 VAE moments 512x512", mom, g["moments"])
    check("v1 VAE decode 64x64 -> 512x512 (every 4th pixel)", dec[:, :, ::4, ::4], g["decoded_sub4"])


def test_full_clip(dev, gold, full):
    inp = cases.full_inputs()
    with torch.no_grad():
        pooled = full.cond_stage_model.transformer(pixel_values=inp["ref"].to(dev)).pooler_output
    check("v1 CLIP ViT-L/14 pooled", pooled, gold["full"]["clip_pooled"])


# ---- size-independent properties at BASELINE.json's full sizes (no CPU oracle finishes there in seconds) ----------
def test_full_pipeline_is_batch_independent(dev, full):
    """Every sample of the path is independent (GroupNorm and attention are per sample): sample i of a
    batch-4 run (BASELINE configs[1] shape; U-Net batch 8) must equal the batch-1 run of the same triple.  Different batch
    sizes pick different tile configs / split-K factors, so fp32 summation order differs, fp16 roundings flip by an ulp and
    CFG (scale 5) amplifies that: the two runs are two equally valid fp16 evaluations.  Bound = the trajectory tolerance
    the oracle comparison uses after the same number of U-Net calls (8e-3 after 3 calls), <= 1 grey level mean on the image."""
    from pbe_amd.pipeline import inpaint
    inp = {k: v.to(dev) for k, v in cases.synthetic_triples(4, 512).items()}
    with torch.no_grad():
        both = inpaint(full, inp["image"], inp["mask"], inp["ref"], steps=2, scale=5.0, x_T=inp["x_T"], post_eps=inp["post_eps"])
        for i in (0, 3):
            one = inpaint(full, inp["image"][i:i + 1], inp["mask"][i:i + 1], inp["ref"][i:i + 1], steps=2, scale=5.0, x_T=inp["x_T"][i:i + 1],
                          post_eps=inp["post_eps"][i:i + 1])
            check(f"batch independence: latent of sample {i} (B=4 vs B=1)", both["latent"][i:i + 1], one["latent"].float().cpu(), 8e-3)
            mad = (both["image"][i:i + 1] - one["image"]).abs().mean().item() * 255.0
            report(f"batch independence: image of sample {i}, mean |d| in grey levels", mad, 0.5)   # measured 0.17
            assert mad <= 0.5
        again = inpaint(full, inp["image"], inp["mask"], inp["ref"], steps=2, scale=5.0, x_T=inp["x_T"], post_eps=inp["post_eps"])
    assert torch.equal(both["latent"], again["latent"]) and torch.equal(both["image"], again["image"])       # run-to-run bit-identical


def test_full_plms_trajectory_against_reference(dev, golden_dir, full):
    """configs/v1.yaml-size PLMS under guidance against the REFERENCE sampler + UNetModel (tests/golden/full_plms.npz,
    oracle/gen_golden.py --only full_plms): 4 steps = 5 U-Net calls on one sample, every intermediate x."""
    from ldm.models.diffusion.plms import PLMSSampler
    g = np.load(os.path.join(golden_dir, "full_plms.npz"))
    inp = cases.full_plms_inputs()
    with torch.no_grad():
        z0, inter = PLMSSampler(full).sample(S=inp["steps"], batch_size=1, shape=[4, 64, 64], conditioning=inp["c"].to(dev), verbose=False,
                                             unconditional_guidance_scale=inp["scale"], unconditional_conditioning=inp["uc"].to(dev), eta=0.0,
                                             x_T=inp["x_T"].to(dev), log_every_t=1,
                                             test_model_kwargs={"images_inpaint": inp["z_inpaint"].to(dev), "images_mask": inp["mask_lat"].to(dev)})
    assert len(inter["x_inter"]) == inp["steps"] + 1
    for i, tol in zip(range(inp["steps"]), (5.5e-3, 6.5e-3, 6.5e-3, 6.5e-3)):             # measured 1.9e-3, 2.2e-3, 2.2e-3, 2.2e-3
        check(f"v1-size PLMS x after step {i}", inter["x_inter"][i + 1], g[f"plms_x_{i}"], tol)
    check("v1-size PLMS final latent (4 steps)", z0, g["plms_latent"], 6.5e-3)


def test_full_plms50_headline_trajectory_against_reference(dev, golden_dir, full):
    """The HEADLINE trajectory at headline size: 50 PLMS steps under guidance 5 (51 U-Net calls) on one configs/v1.yaml-size sample
    against the REFERENCE PLMSSampler + UNetModel (tests/golden/full_plms50.npz, oracle/gen_golden.py --only full_plms50: 9 CPU-minutes
    of the reference, oracle == reference to 1.3e-6).  x after steps 0 / 3 / 25 / 49 and the final latent."""
    from ldm.models.diffusion.plms import PLMSSampler
    g = np.load(os.path.join(golden_dir, "full_plms50.npz"))
    inp = cases.full_plms50_inputs()
    with torch.no_grad():
        z0, inter = PLMSSampler(full).sample(S=inp["steps"], batch_size=1, shape=[4, 64, 64], conditioning=inp["c"].to(dev), verbose=False,
                                             unconditional_guidance_scale=inp["scale"], unconditional_conditioning=inp["uc"].to(dev), eta=0.0,
                                             x_T=inp["x_T"].to(dev), log_every_t=1,
                                             test_model_kwargs={"images_inpaint": inp["z_inpaint"].to(dev), "images_mask": inp["mask_lat"].to(dev)})
    assert len(inter["x_inter"]) == inp["steps"] + 1
    for i in cases.FULL_PLMS50_RECORD:
        check(f"v1-size PLMS-50 x after step {i}", inter["x_inter"][i + 1], g[f"plms_x_{i}"], TRAJ50_TOL)
    check("v1-size PLMS-50 final latent (50 steps)", z0, g["plms_latent"], TRAJ50_TOL)


def test_full_unet_shared_guidance_prefix(dev, full):
    """forward_nhwc(paired=True) (the context-independent prefix of a guidance pair evaluated once at batch B) against the plain
    duplicated batch 2B.  tools/layer_diff.py shows that the ONLY batch-dependent choice that changes bits is the split-K factor
    (fp32 summation order); the prefix pins it to the batch-2B layer's choice (ops.pinned_batch_scale), so the two evaluations
    must be BIT-identical - common-subexpression elimination, not an approximation."""
    from pbe_amd import ops
    g = torch.Generator().manual_seed(5)
    for B in (4, 1):
        x = torch.randn(B, 4, 64, 64, generator=g).to(dev)
        z = torch.randn(B, 4, 64, 64, generator=g).to(dev)
        m = (torch.rand(B, 1, 64, 64, generator=g) > 0.3).float().to(dev)
        ctx = torch.randn(2 * B, 1, 768, generator=g).to(dev)
        t = torch.full((2 * B,), 621, dtype=torch.int64, device=dev)
        unet = full.model.diffusion_model
        with torch.no_grad():
            a = unet.forward_nhwc(ops.plms_pack_input(x, z, m, 2), t, ctx)
            b = unet.forward_nhwc(ops.plms_pack_input(x, z, m, 1), t, ctx, paired=True)
        assert a.shape == b.shape == (2 * B, 64, 64, 4)
        ndiff = int((a != b).sum().item())
        report(f"guidance pair B={B}: elements differing, shared prefix vs duplicated batch", float(ndiff), 0.0)
        assert torch.equal(a, b), f"B={B}: {ndiff} of {a.numel()} elements differ"
        assert not torch.equal(b[:B], b[B:])                      # the halves really differ (different contexts)


def test_full_unet_768_latents_and_batch32(dev, full):
    """BASELINE configs[4] geometry (768x768 pixels -> 96x96 latents, 9216-token self-attention) and configs[2]'s U-Net batch
    (16 images under CFG = 32): the kernels accept the sizes, outputs are finite, deterministic, and sample-independent."""
    g = torch.Generator().manual_seed(99)
    x = torch.randn(2, 9, 96, 96, generator=g).to(dev)
    t = torch.tensor([401, 401], dtype=torch.int64, device=dev)
    ctx = torch.randn(2, 1, 768, generator=g).to(dev)
    with torch.no_grad():
        y2 = full.apply_model(x, t, ctx)
        y1 = full.apply_model(x[1:], t[1:], ctx[1:])
        assert y2.shape == (2, 4, 96, 96) and torch.isfinite(y2).all()
        assert torch.equal(y2, full.apply_model(x, t, ctx))
        check("768x768: U-Net sample 1 of batch 2 vs alone", y2[1:], y1.float().cpu(), 2e-3)
        xb = torch.randn(32, 9, 64, 64, generator=g).to(dev)
        tb = torch.full((32,), 981, dtype=torch.int64, device=dev)
        cb = torch.randn(32, 1, 768, generator=g).to(dev)
        yb = full.apply_model(xb, tb, cb)
        assert yb.shape == (32, 4, 64, 64) and torch.isfinite(yb).all()
        check("U-Net batch 32: sample 17 vs alone", yb[17:18], full.apply_model(xb[17:18], tb[17:18], cb[17:18]).float().cpu(), 2e-3)


def _load_cli():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pbe_inference_cli", os.path.join(root, "scripts", "inference.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    return cli, root


def test_inference_cli_on_bundled_example(dev, golden_dir, tmp_path):
    """scripts/inference.py counterpart end to end on examples/example_1 at configs/v1.yaml size (BASELINE configs[0] inputs; seed 321,
    scale 5 as in the reference's test.sh) with name-seeded weights, 4 PLMS steps: files written, image finite, and the files the
    GPU composed (pbe_planes_to_u8_canvas) are byte-identical to the host composition of the same tensors."""
    from PIL import Image
    from pbe_amd import preprocess
    cli, root = _load_cli()
    d = os.path.join(golden_dir, "examples")
    paths = (os.path.join(d, "image_example_1.png"), os.path.join(d, "mask_example_1.png"), os.path.join(d, "reference_example_1.jpg"))
    out = cli.main(["--plms", "--outdir", str(tmp_path), "--config", os.path.join(root, "configs", "v1.yaml"), "--ddim_steps", "4",
                    "--image_path", paths[0], "--mask_path", paths[1], "--reference_path", paths[2], "--seed", "321", "--scale", "5", "--fixed_code"])
    assert out.shape == (1, 3, 512, 512) and torch.isfinite(out).all() and 0 <= out.min() and out.max() <= 1
    names = {"result": ("results", "image_example_1_321.png"), "grid": ("grid", "grid-image_example_1_321.png"), "mask": ("source", "image_example_1_321_mask.png"),
             "gt": ("source", "image_example_1_321_GT.png"), "inpaint": ("source", "image_example_1_321_inpaint.png"), "ref": ("source", "image_example_1_321_ref.png")}
    host_dir = str(tmp_path / "host")
    host = preprocess.save_outputs(host_dir, "image_example_1", 321, preprocess.load_triple(*paths), out[0], 512, 512)
    for k, (sub, name) in names.items():
        f = os.path.join(str(tmp_path), sub, name)
        assert os.path.getsize(f) > 0
        a, b = np.asarray(Image.open(f)), np.asarray(Image.open(host[k]))
        assert a.shape == b.shape and np.array_equal(a, b), f"{k}: GPU-composed file differs from the host composition in {int((a != b).sum())} bytes"
    # device pre-processing == host pre-processing, bit for bit
    td, th = preprocess.load_triple_device(*paths, dev), preprocess.load_triple(*paths)
    for k in ("image", "mask", "inpaint", "ref"):
        assert torch.equal(td[k].cpu(), th[k]), k


@pytest.mark.parametrize("example,seed,steps", [(1, 321, 4), (2, 5876, 2), (3, 5065, 2)])
def test_inference_cli_matches_oracle_pipeline(dev, golden_dir, tmp_path, example, seed, steps):
    """The CLI's RESULT IMAGE against the oracle's statement of scripts/inference.py:305-348 on the reference's three bundled triples
    (test.sh:1-29: seeds 321 / 5876 / 5065, --scale 5, --plms).  Narrow-topology weights keep the CPU oracle in seconds; the run's
    device-RNG start code and the posterior noise are handed to the oracle through --dump_tensors.  Tolerances: the per-stage bounds
    of the module docstring (conditioning / latent 4e-3 / 8e-3 after <= 5 U-Net calls), <= 1 grey level mean on the PNG."""
    import yaml
    from PIL import Image
    from oracle_loader import O
    cli, root = _load_cli()
    d = os.path.join(golden_dir, "examples")
    paths = (os.path.join(d, f"image_example_{example}.png"), os.path.join(d, f"mask_example_{example}.png"), os.path.join(d, f"reference_example_{example}.jpg"))
    cfg = str(tmp_path / "narrow.yaml")
    with open(cfg, "w") as f:
        yaml.safe_dump({"model": build.narrow_config()}, f)
    dump = str(tmp_path / "dump.npz")
    out = cli.main(["--plms", "--outdir", str(tmp_path), "--config", cfg, "--ddim_steps", str(steps), "--image_path", paths[0], "--mask_path", paths[1],
                    "--reference_path", paths[2], "--seed", str(seed), "--scale", "5", "--fixed_code", "--random_weights", "--dump_tensors", dump])
    t = np.load(dump)
    img = np.asarray(Image.open(paths[0]).convert("RGB"))
    msk = np.asarray(Image.open(paths[1]).convert("L"))
    ref = np.asarray(Image.open(paths[2]).convert("RGB").resize((224, 224)))
    oi, _, om, oref = O.preprocess_triple(img, msk, ref)
    model = build.narrow_model("cpu")
    sd = {k: v.detach().float() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want = O.inpaint_pipeline(sd, oi, om, oref, torch.from_numpy(t["x_T"]), torch.from_numpy(t["post_eps"]), S=steps, scale=5.0, unet_cfg=cases.UNET_NARROW,
                                  vae_cfg=cases.VAE_NARROW, clip_cfg=cases.CLIP_NARROW, map_cfg=cases.MAPPER_NARROW)
    assert want["calls"] == steps + 1
    check(f"CLI example_{example}: conditioning c", torch.from_numpy(t["c"]), want["c"], 4e-3)
    check(f"CLI example_{example}: z_inpaint", torch.from_numpy(t["z_inpaint"]), want["z_inpaint"], 4e-3)
    assert np.abs(t["mask64"] - want["mask64"].numpy()).max() <= 2e-6
    check(f"CLI example_{example}: final latent ({steps} PLMS steps)", torch.from_numpy(t["latent"]), want["latent"], 8e-3)
    png = np.asarray(Image.open(os.path.join(str(tmp_path), "results", f"image_example_{example}_{seed}.png"))).astype(np.float32)
    exp = (255.0 * want["image"][0].permute(1, 2, 0).numpy()).astype(np.uint8).astype(np.float32)
    mad = float(np.abs(png - exp).mean())
    report(f"CLI example_{example}: result PNG vs oracle, mean |d| in grey levels", mad, 0.55)      # measured 0.16-0.18
    assert png.shape == (512, 512, 3) and mad <= 0.55
    assert torch.equal(out, torch.from_numpy(t["image"]))


# ---- checkpoint FILES through the product loader (scripts/inference.py:58-75, ddpm.py:245-260) --------
def test_lightning_checkpoint_round_trip_narrow(dev, narrow, tmp_path):
    """A Lightning-layout file (HF-4.19 CLIP names, model_ema.* shadow weights, foreign callbacks / hyper_parameters objects)
    -> load_model_from_config -> packs -> forward: bit-identical to the model filled in memory with the same tensors, for the
    U-Net, the VAE and the CLIP conditioning; the transformers-5 key spelling loads to the same bits."""
    import ckpt_synth
    from ldm.models.diffusion.ddpm import load_model_from_config
    inp = cases.narrow_inputs()
    sd = {k: v.detach().cpu() for k, v in narrow.state_dict().items()}

    def outputs(m):
        with torch.no_grad():
            y = m.apply_model(inp["unet_x"].to(dev), inp["unet_t"].to(dev), inp["unet_ctx"].to(dev))
            c = m.project_conditioning(m.get_learned_conditioning(inp["ref"].to(dev)))
            mom = m.encode_first_stage((inp["image"] * inp["mask"]).to(dev)).parameters
            dec = m.decode_first_stage(inp["x_T"].to(dev).clone())
        return y, c, mom, dec

    want = outputs(narrow)
    for hf5 in (False, True):
        path = str(tmp_path / f"model_{int(hf5)}.ckpt")
        keys = ckpt_synth.write_lightning_checkpoint(path, sd, hf5_clip_names=hf5)
        assert any(k.startswith("model_ema.") for k in keys)
        loaded = load_model_from_config({"model": build.narrow_config()}, path, device=dev)
        lsd = loaded.state_dict()
        assert all(torch.equal(lsd[k].cpu(), sd[k]) for k in sd), "state dict differs after the file round trip"
        for name, a, b in zip(("unet", "conditioning", "vae moments", "vae decode"), outputs(loaded), want):
            assert torch.equal(a, b), f"{name} differs after loading the checkpoint file (hf5 names: {hf5})"
        del loaded


def test_partial_checkpoint_at_v1_size(dev, full, tmp_path):
    """configs/v1.yaml size: a file holding the 9-channel conv-in `[320, 9, 3, 3]` (scripts/modify_checkpoints.py:1-6), the first
    block, the output head and proj_out / learnable_vector with NEW values, loaded with init_from_ckpt (strict=False, ddpm.py:245-260)
    gives the same bits as assigning the tensors in memory, and differs from the model before the load."""
    import ckpt_synth
    from pbe_amd.weights import synth_tensor
    pre = ("model.diffusion_model.input_blocks.0.", "model.diffusion_model.input_blocks.1.", "model.diffusion_model.out.", "proj_out.", "learnable_vector")
    cur = full.state_dict()
    old = {k: v.detach().clone() for k, v in cur.items() if k.startswith(pre)}
    new = {k: synth_tensor(k, v.shape, seed=7).to(v.dtype) for k, v in old.items()}
    assert tuple(new["model.diffusion_model.input_blocks.0.0.weight"].shape) == (320, 9, 3, 3)
    inp = cases.full_inputs()
    x, t, ctx = inp["unet_x"].to(dev), inp["unet_t"].to(dev), inp["unet_ctx"].to(dev)
    z = torch.randn(2, 1, 1024, generator=torch.Generator().manual_seed(3)).to(dev)
    try:
        with torch.no_grad():
            before = full.apply_model(x, t, ctx)
            full.load_state_dict(new, strict=False)
            want = (full.apply_model(x, t, ctx), full.project_conditioning(z))
            full.load_state_dict(old, strict=False)
            assert torch.equal(full.apply_model(x, t, ctx), before)
            path = str(tmp_path / "partial.ckpt")
            ckpt_synth.write_lightning_checkpoint(path, new, ema_from_prefix="model.diffusion_model.out.")
            full.init_from_ckpt(path)
            got = (full.apply_model(x, t, ctx), full.project_conditioning(z))
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        assert not torch.equal(got[0], before)
    finally:
        full.load_state_dict(old, strict=False)


def test_pipeline_batch16_configs2_geometry(dev, full):
    """BASELINE configs[2] workload shape (1 GPU, batch 16, classifier-free guidance -> U-Net batch 32) through the WHOLE pipeline
    (CLIP + VAE encode + PLMS + VAE decode; 2 steps = 3 U-Net calls): finite, run-to-run bit-identical, and sample i equals its
    batch-1 run within the trajectory tolerance after 3 calls (same bound as the batch-4 test).  The real checkpoint of
    configs[2] does not exist offline; the file path is covered by the checkpoint tests above."""
    from pbe_amd.pipeline import inpaint
    inp = {k: v.to(dev) for k, v in cases.synthetic_triples(16, 512).items()}
    with torch.no_grad():
        out = inpaint(full, inp["image"], inp["mask"], inp["ref"], steps=2, scale=5.0, x_T=inp["x_T"], post_eps=inp["post_eps"])
        assert out["image"].shape == (16, 3, 512, 512) and torch.isfinite(out["image"]).all() and torch.isfinite(out["latent"]).all()
        for i in (5, 15):
            one = inpaint(full, inp["image"][i:i + 1], inp["mask"][i:i + 1], inp["ref"][i:i + 1], steps=2, scale=5.0, x_T=inp["x_T"][i:i + 1],
                          post_eps=inp["post_eps"][i:i + 1])
            check(f"configs[2] batch 16: latent of sample {i} vs alone", out["latent"][i:i + 1], one["latent"].float().cpu(), 8e-3)
            mad = (out["image"][i:i + 1] - one["image"]).abs().mean().item() * 255.0
            report(f"configs[2] batch 16: image of sample {i}, mean |d| in grey levels", mad, 0.5)  # measured 0.16
            assert mad <= 0.5
        again = inpaint(full, inp["image"], inp["mask"], inp["ref"], steps=2, scale=5.0, x_T=inp["x_T"], post_eps=inp["post_eps"])
    assert torch.equal(out["latent"], again["latent"])


# ---- BASELINE configs[4]: fp8 (e4m3) linear path, tolerance re-validated against the SAME fp32 goldens ---------------------------
FP8_FWD_TOL = 5e-2


def test_full_unet_fp8_linear_path(dev, gold, full):
    """configs/v1.yaml U-Net with the LayerNorm-fed projections (q|k, V^T, GEGLU: 42 % of the linear FLOPs) on e4m3 operands
    (pbe_amd.precision).  e4m3 keeps 3 mantissa bits: a rounded operand carries a relative error of up to 2^-4 (rms ~ 2^-4 / sqrt(3)
    = 3.6e-2), so ONE e4m3 x e4m3 projection output is off by ~ 3.6e-2 * sqrt(2) = 5e-2 of its own magnitude.  Stated bound for a whole
    forward against the reference's fp32 golden: rel-L2 <= 5e-2, i.e. the network must not amplify beyond the error of a single
    quantised projection (fp16 path: 4e-3).  Measured on MI355X with the name-seeded weights: 3.4e-2 (parity report); a trained
    checkpoint's activation statistics differ (outlier channels) and need their own re-validation when one is available."""
    from pbe_amd.precision import set_linear_precision
    inp = cases.full_inputs()
    x, t, ctx = inp["unet_x"].to(dev), inp["unet_t"].to(dev), inp["unet_ctx"].to(dev)
    try:
        assert set_linear_precision(full, "fp8") == 16
        with torch.no_grad():
            y8 = full.apply_model(x, t, ctx)
            again = full.apply_model(x, t, ctx)
        assert torch.equal(y8, again)                                # deterministic
    finally:
        set_linear_precision(full, "fp16")
    with torch.no_grad():
        y16 = full.apply_model(x, t, ctx)
    check("v1 UNetModel forward, fp8 linear path vs fp32 golden", y8, gold["full"]["unet_y"], FP8_FWD_TOL)
    r = rel_l2(y8, y16.float().cpu())
    report("v1 UNetModel forward, fp8 linear path vs the fp16 path", r, FP8_FWD_TOL)
    assert 1e-4 < r <= FP8_FWD_TOL                                    # it really took the other path, and stays within the bound


def test_fp8_path_768_latents_100_steps_geometry(dev, full):
    """BASELINE configs[4] geometry on the fp8 linear path: 96x96 latents (9 216-token self-attention) under guidance.  One U-Net
    evaluation of a guidance pair is compared with the fp16 path (same bound as the 64x64 forward); the full 100-step PLMS run
    (S = 100 => 101 U-Net calls, timestep spacing 10) must be finite and bit-identical run to run."""
    from ldm.models.diffusion.plms import PLMSSampler
    from pbe_amd.precision import set_linear_precision
    g = torch.Generator().manual_seed(31)
    xT = torch.randn(1, 4, 96, 96, generator=g).to(dev)
    z = (torch.randn(1, 4, 96, 96, generator=g) * 0.8).to(dev)
    m = torch.ones(1, 1, 96, 96)
    m[:, :, 30:70, 20:60] = 0
    m = m.to(dev)
    c, uc = torch.randn(1, 1, 768, generator=g).to(dev), torch.randn(1, 1, 768, generator=g).to(dev)
    x9 = torch.cat([xT, z, m], 1)
    t = torch.tensor([981, 981], dtype=torch.int64, device=dev)

    def run():
        smp = PLMSSampler(full)
        lat, _ = smp.sample(S=100, batch_size=1, shape=[4, 96, 96], conditioning=c, verbose=False, unconditional_guidance_scale=5.0,
                            unconditional_conditioning=uc, eta=0.0, x_T=xT, test_model_kwargs={"inpaint_image": z, "inpaint_mask": m})
        assert len(smp.ddim_timesteps) == 100 and int(smp.ddim_timesteps[1] - smp.ddim_timesteps[0]) == 10
        return lat
    with torch.no_grad():
        y16 = full.apply_model(torch.cat([x9, x9]), t, torch.cat([uc, c]))
        try:
            set_linear_precision(full, "fp8")
            y8 = full.apply_model(torch.cat([x9, x9]), t, torch.cat([uc, c]))
            a, b = run(), run()
        finally:
            set_linear_precision(full, "fp16")
    check("configs[4] geometry: U-Net at 96x96 (guidance pair), fp8 linear path vs fp16 path", y8, y16.float().cpu(), FP8_FWD_TOL)
    assert a.shape == (1, 4, 96, 96) and torch.isfinite(a).all() and torch.equal(a, b)
