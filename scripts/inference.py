#!/usr/bin/env python3
"""Single-triple inpainting CLI — counterpart of scripts/inference.py:127-402 in zhanwenchen/pbe with
the same flags and defaults, the same call order (CLIP -> proj_out -> VAE encode -> mask resize ->
sampler -> decode -> clamp) and the same output tree, running on the MI355X HIP path.

    python scripts/inference.py --plms --outdir results --config configs/v1.yaml --ckpt checkpoints/model.ckpt \
        --image_path examples/image/example_1.png --mask_path examples/mask/example_1.png \
        --reference_path examples/reference/example_1.jpg --seed 321 --scale 5

Differences, all deliberate: the safety checker and the invisible watermark are dropped (the
reference overwrites the checker's result, :350-351; both need hub downloads); `--ckpt ""` or
`--random_weights` runs with name-seeded random weights (no checkpoint exists offline);
`--precision` is accepted for compatibility (the HIP path is always fp16 storage / fp32 accumulate).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--outdir", type=str, nargs="?", default="outputs/txt2img-samples")
    p.add_argument("--skip_grid", action="store_true")
    p.add_argument("--skip_save", action="store_true")
    p.add_argument("--ddim_steps", type=int, default=50)
    p.add_argument("--plms", action="store_true")
    p.add_argument("--fixed_code", action="store_true")
    p.add_argument("--ddim_eta", type=float, default=0.0)
    p.add_argument("--n_iter", type=int, default=2)
    p.add_argument("--H", type=int, default=512)
    p.add_argument("--W", type=int, default=512)
    p.add_argument("--n_imgs", type=int, default=100)
    p.add_argument("--C", type=int, default=4)
    p.add_argument("--f", type=int, default=8)
    p.add_argument("--n_samples", type=int, default=1)
    p.add_argument("--n_rows", type=int, default=0)
    p.add_argument("--scale", type=float, default=1)
    p.add_argument("--config", type=str, default="")
    p.add_argument("--ckpt", type=str, default="")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--precision", type=str, choices=["full", "autocast"], default="autocast")
    p.add_argument("--image_path", type=str, default="")
    p.add_argument("--mask_path", type=str, default="")
    p.add_argument("--reference_path", type=str, default="")
    p.add_argument("--random_weights", action="store_true", help="name-seeded random weights instead of --ckpt")
    p.add_argument("--dump_tensors", type=str, default="", help="(not in the reference) save the start code, the posterior noise and the "
                   "intermediate tensors of this run to an .npz: what a CPU replay needs to reproduce the run without the device RNG")
    return p.parse_args(argv)


def seed_everything(seed):
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def main(argv=None):
    opt = parse(argv)
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.ddpm import load_model_from_config
    from ldm.models.diffusion.plms import PLMSSampler
    from ldm.util import load_yaml_config
    from pbe_amd import ops, pipeline, preprocess, weights

    seed_everything(opt.seed)
    if not torch.cuda.is_available():
        raise SystemExit("scripts/inference.py needs an MI355X: the HIP path has no CPU fallback")
    device = torch.device("cuda")
    config = load_yaml_config(opt.config or os.path.join(ROOT, "configs", "v1.yaml"))
    if opt.ckpt and not opt.random_weights:
        model = load_model_from_config(config, opt.ckpt, device=device)
    else:
        print("no --ckpt: running with name-seeded random weights")
        model = load_model_from_config(config, None, device="cpu")
        weights.fill_latent_diffusion_(model, seed=0)
        model = model.to(device).eval()
    sampler = PLMSSampler(model) if opt.plms else DDIMSampler(model)

    start_code = None
    if opt.fixed_code:
        start_code = torch.randn([opt.n_samples, opt.C, opt.H // opt.f, opt.W // opt.f], device=device)

    with torch.no_grad(), model.ema_scope():
        t = preprocess.load_triple_device(opt.image_path, opt.mask_path, opt.reference_path, device)      # uint8 up, arithmetic on the GPU
        filename = os.path.basename(opt.image_path)
        test_model_kwargs = {"inpaint_mask": t["mask"], "inpaint_image": t["inpaint"]}
        ref = t["ref"]
        uc = model.learnable_vector if opt.scale != 1.0 else None
        c = model.proj_out(model.get_learned_conditioning(ref))                      # scripts/inference.py:326-327
        post = model.encode_first_stage(test_model_kwargs["inpaint_image"])
        pb, ph, pw, _ = post.parameters.shape
        post_eps = torch.randn((pb, post.z, ph, pw)) if opt.dump_tensors else None   # the CPU draw DiagonalGaussianDistribution.sample() makes itself
        z_inpaint = model.get_first_stage_encoding(post, noise=None if post_eps is None else post_eps.to(device))
        test_model_kwargs["inpaint_image"] = z_inpaint
        test_model_kwargs["inpaint_mask"] = pipeline.resize_mask(test_model_kwargs["inpaint_mask"], z_inpaint.shape[-2:])
        shape = [opt.C, opt.H // opt.f, opt.W // opt.f]
        samples, _ = sampler.sample(S=opt.ddim_steps, conditioning=c, batch_size=opt.n_samples, shape=shape, verbose=False,
                                    unconditional_guidance_scale=opt.scale, unconditional_conditioning=uc, eta=opt.ddim_eta,
                                    x_T=start_code, test_model_kwargs=test_model_kwargs)
        xd = ops.image_post(model.decode_first_stage_nhwc(samples))                  # clamp((x+1)/2, 0, 1), still on the GPU
        if not opt.skip_save:
            for i in range(xd.shape[0]):
                paths = preprocess.save_outputs_device(opt.outdir, filename[:-4], opt.seed, t, xd[i], opt.H, opt.W)
        x = xd.cpu()
        if opt.dump_tensors:
            import numpy as np
            np.savez(opt.dump_tensors, x_T=(start_code if start_code is not None else torch.zeros(0)).float().cpu().numpy(),
                     post_eps=post_eps.numpy(), c=c.float().cpu().numpy(), z_inpaint=z_inpaint.float().cpu().numpy(),
                     mask64=test_model_kwargs["inpaint_mask"].float().cpu().numpy(), latent=samples.float().cpu().numpy(), image=x.numpy())
    print(f"Your samples are ready and waiting for you here: \n{opt.outdir} \n \nEnjoy.")
    return x


if __name__ == "__main__":
    main()
