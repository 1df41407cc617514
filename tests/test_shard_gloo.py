"""CPU suite (-m "not gpu"), part 3: the N > 1 path on 2 gloo ranks — batch-axis sharding, the one-off
bucketed weight broadcast and the result gather (pbe_amd/shard.py).  The per-rank
"compute" is the oracle's narrow first-stage decode (a test may use the oracle; the product never does)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_loader import O
        from pbe_amd import shard
        from pbe_amd.weights import synth_tensor
        # a module whose weights only rank 0 knows: several dtypes / sizes so buckets split
        torch.manual_seed(100 + rank)
        mod = torch.nn.Sequential(torch.nn.Conv2d(4, 32, 3), torch.nn.Linear(64, 300), torch.nn.LayerNorm(300))
        mod.register_buffer("steps", torch.arange(10) * (rank + 1))
        if rank == 0:
            ref = {k: v.clone() for k, v in mod.state_dict().items()}
        info = shard.broadcast_weights_(mod, src=0, bucket_bytes=40_000)
        got = {k: v.clone() for k, v in mod.state_dict().items()}
        objs = [None]
        if rank == 0:
            objs = [ref]
        dist.broadcast_object_list(objs, src=0)
        same = all(torch.equal(got[k], objs[0][k]) for k in got)
        # shard a work list of 11 items, per-rank batch 2 -> 2 full rounds, remainder dropped
        rounds = shard.shard_indices(11, rank, world, 2, drop_last=True)
        # "compute": decode latents of my samples with the oracle on name-seeded narrow VAE weights
        keys = {}
        with open(os.path.join(os.path.dirname(__file__), "golden", "narrow_keys.txt")) as f:
            for line in f:
                k, s = line.split()
                if k.startswith("first_stage_model.") and ("decoder" in k or "post_quant" in k):
                    keys[k] = tuple(int(x) for x in s.split("x"))
        sd = {k: synth_tensor(k, s) for k, s in keys.items()}
        outs = []
        with torch.no_grad():
            for rnd in rounds:
                z = torch.cat([cases.synthetic_triples(1, 64, first_index=i)["x_T"] for i in rnd])
                img = torch.clamp((O.first_stage_decode(sd, z, cases.VAE_NARROW, "first_stage_model.") + 1) / 2, 0, 1)
                u8 = (img * 255).round().to(torch.uint8)
                gathered = shard.gather_images(u8, dst=0)
                if rank == 0:
                    outs.append(shard.interleave_rank_major(gathered, world))
        if rank == 0:
            q.put({"same": same, "info": info, "rounds": rounds, "images": torch.cat(outs)})
        else:
            q.put({"same": same, "info": info, "rounds": rounds})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_broadcast_shard_gather():
    from oracle_loader import O
    from pbe_amd import shard
    from pbe_amd.weights import synth_tensor
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r["same"] for r in res), "weights differ after the broadcast"
    assert all(r["info"]["messages"] >= 2 for r in res)                   # bucketed, not one message per tensor (9 tensors)
    r0 = next(r for r in res if "images" in r)
    r1 = next(r for r in res if "images" not in r)
    assert r0["rounds"] == [[0, 2], [4, 6]] and r1["rounds"] == [[1, 3], [5, 7]]      # r::W inside each round, remainder (8..10) dropped
    # single-process result on the same 8 samples, global order
    keys = {}
    with open(os.path.join(os.path.dirname(__file__), "golden", "narrow_keys.txt")) as f:
        for line in f:
            k, s = line.split()
            if k.startswith("first_stage_model.") and ("decoder" in k or "post_quant" in k):
                keys[k] = tuple(int(x) for x in s.split("x"))
    sd = {k: synth_tensor(k, s) for k, s in keys.items()}
    with torch.no_grad():
        z = torch.cat([cases.synthetic_triples(1, 64, first_index=i)["x_T"] for i in range(8)])
        ref = (torch.clamp((O.first_stage_decode(sd, z, cases.VAE_NARROW, "first_stage_model.") + 1) / 2, 0, 1) * 255).round().to(torch.uint8)
    assert torch.equal(r0["images"], ref)
    assert shard.shard_indices(11, 0, 2, 2, drop_last=False)[-1] == [8, 10]
