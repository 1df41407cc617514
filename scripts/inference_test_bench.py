#!/usr/bin/env python3
"""COCOEE test-bench sweep — counterpart of scripts/inference_test_bench.py:295-402 in zhanwenchen/pbe
(BASELINE config #4), same flags / defaults / output tree, on the MI355X HIP path.

    # one GPU
    python scripts/inference_test_bench.py --plms --outdir results/test_bench --config configs/v1.yaml \
        --ckpt checkpoints/model.ckpt --scale 5 --n_samples 8 --test_bench_dir test_bench
    # eight GPUs, one process each, ids dealt by full batches (no data-path collective)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/inference_test_bench.py ...

Differences, all deliberate: no safety checker (not part of the reference's test bench either); the
text-prompt flags the reference inherited from txt2img (--prompt, --from-file, --laion400m, --n_iter,
--n_rows, --skip_grid, --precision) are accepted and ignored; `--test_bench_dir` is a flag (the
reference hard-codes ./test_bench); `--ckpt ""` / `--random_weights` runs name-seeded random weights;
`--max_batches` bounds a smoke run.  RANK / WORLD_SIZE / LOCAL_RANK come from the launcher.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--prompt", type=str, nargs="?", default="a photograph of an astronaut riding a horse")
    p.add_argument("--outdir", type=str, nargs="?", default="outputs/txt2img-samples")
    p.add_argument("--skip_grid", action="store_true")
    p.add_argument("--skip_save", action="store_true")
    p.add_argument("--ddim_steps", type=int, default=50)
    p.add_argument("--plms", action="store_true")
    p.add_argument("--laion400m", action="store_true")
    p.add_argument("--fixed_code", action="store_true")
    p.add_argument("--ddim_eta", type=float, default=0.0)
    p.add_argument("--n_iter", type=int, default=2)
    p.add_argument("--H", type=int, default=512)
    p.add_argument("--W", type=int, default=512)
    p.add_argument("--C", type=int, default=4)
    p.add_argument("--f", type=int, default=8)
    p.add_argument("--n_samples", type=int, default=5)
    p.add_argument("--n_rows", type=int, default=0)
    p.add_argument("--scale", type=float, default=1)
    p.add_argument("--from-file", type=str)
    p.add_argument("--config", type=str, default="")
    p.add_argument("--ckpt", type=str, default="")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--rank", type=int, default=0)
    p.add_argument("--precision", type=str, choices=["full", "autocast"], default="autocast")
    p.add_argument("--test_bench_dir", type=str, default="test_bench")
    p.add_argument("--random_weights", action="store_true")
    p.add_argument("--max_batches", type=int, default=None)
    return p.parse_args(argv)


def main(argv=None):
    opt = parse(argv)
    if opt.ddim_eta != 0.0:
        raise SystemExit("--ddim_eta != 0 is not supported on this path (the reference's test bench runs eta = 0)")
    rank, world = int(os.environ.get("RANK", opt.rank)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("scripts/inference_test_bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from ldm.models.diffusion.ddpm import load_model_from_config
    from ldm.util import load_yaml_config
    from pbe_amd import shard, testbench, weights

    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
    config = load_yaml_config(opt.config or os.path.join(ROOT, "configs", "v1.yaml"))
    if opt.ckpt and not opt.random_weights:
        model = load_model_from_config(config, opt.ckpt if rank == 0 else None, device="cpu")
    else:
        model = load_model_from_config(config, None, device="cpu")
        if rank == 0:
            print("no --ckpt: running with name-seeded random weights")
            weights.fill_latent_diffusion_(model, seed=0)
    model = model.to(device).eval()
    if world > 1:
        shard.broadcast_weights_(model, src=0)            # weights are read / synthesised once, on rank 0
    ds = testbench.COCOImageDataset(opt.test_bench_dir)
    if rank == 0:
        print("length of test bench", len(ds))
    stats = testbench.run_sweep(model, ds, opt.outdir, batch_size=opt.n_samples, steps=opt.ddim_steps, scale=opt.scale, plms=opt.plms,
                                fixed_code=opt.fixed_code, seed=opt.seed, rank=rank, world=world, skip_save=opt.skip_save, C=opt.C, f=opt.f,
                                max_batches=opt.max_batches)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    print(json.dumps({"rank": rank, "world": world, "batches": stats["batches"], "images": len(stats["ids"])}))
    if rank == 0:
        print(f"Your samples are ready and waiting for you here: \n{opt.outdir} \n \nEnjoy.")
    return stats


if __name__ == "__main__":
    main()
