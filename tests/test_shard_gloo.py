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
        same = same and shard.weights_checksum(mod)                     # after the broadcast every rank holds the same weights ...
        with torch.no_grad():
            mod[1].weight[0, 0] += float(rank)                          # ... and a single differing element on one rank is noticed
        same = same and not shard.weights_checksum(mod)
        with torch.no_grad():
            mod[1].weight[0, 0] -= float(rank)
        # the product's partition (pbe_amd.testbench.rank_batches): 11 items, batch 2 -> 5 full batches dealt round-robin, remainder dropped
        from pbe_amd.testbench import rank_batches
        mine = rank_batches(11, 2, rank, world)
        rounds = [items for _, items in mine]
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
                outs.append(u8)
            # ranks may hold different batch counts (5 batches over 2 ranks): gather batch by batch, padding the short rank
            n_rounds = (5 + world - 1) // world
            gathered = []
            for k in range(n_rounds):
                u8 = outs[k] if k < len(outs) else torch.zeros_like(outs[0])
                g = shard.gather_images(u8, dst=0)
                if rank == 0:
                    gathered.append(g)
            outs = gathered
        if rank == 0:
            q.put({"same": same, "info": info, "rounds": rounds, "images": torch.stack(outs)})        # [round, rank-major 2*b, ...]
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
    assert r0["rounds"] == [[0, 1], [4, 5], [8, 9]] and r1["rounds"] == [[2, 3], [6, 7]]      # whole batches round-robin, item 10 dropped
    # single-process result on the same 10 samples, global order
    keys = {}
    with open(os.path.join(os.path.dirname(__file__), "golden", "narrow_keys.txt")) as f:
        for line in f:
            k, s = line.split()
            if k.startswith("first_stage_model.") and ("decoder" in k or "post_quant" in k):
                keys[k] = tuple(int(x) for x in s.split("x"))
    sd = {k: synth_tensor(k, s) for k, s in keys.items()}
    with torch.no_grad():
        ref = []                                         # batch by batch, like the ranks (CPU conv kernels pick algorithms by batch size)
        for b0 in range(0, 10, 2):
            z = torch.cat([cases.synthetic_triples(1, 64, first_index=i)["x_T"] for i in (b0, b0 + 1)])
            ref.append((torch.clamp((O.first_stage_decode(sd, z, cases.VAE_NARROW, "first_stage_model.") + 1) / 2, 0, 1) * 255).round().to(torch.uint8))
        ref = torch.cat(ref)
    img = r0["images"]                                   # [3 rounds, 4 (rank 0's 2 then rank 1's 2), 3, H, W]
    for k in range(3):
        assert torch.equal(img[k, 0:2], ref[4 * k:4 * k + 2])                    # batch 2k on rank 0
        if k < 2:
            assert torch.equal(img[k, 2:4], ref[4 * k + 2:4 * k + 4])            # batch 2k+1 on rank 1


# ---- bench.py's own launcher (python bench.py --gpus N with no RANK in the environment) --------------------------------
def _run_bench(args, env_extra=None, launcher=None, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    cmd = (launcher or [sys.executable]) + [os.path.join(root, "bench.py")] + args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(300)
def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` starts its own two workers, relays rank 0's one JSON line and exits 0 (rehearsal: gloo, no GPU work)."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    # per_rank_ms: every rank's own time travels to rank 0 (min / max / all), so a scaling loss can be attributed to one slow device
    assert out == {"rehearsal": True, "n_gpus": 2, "rank_sum": 3.0, "steps": 2, "warmup": 1,
                   "per_rank_ms": {"min": 10.0, "max": 20.0, "all": [10.0, 20.0]}}


@pytest.mark.timeout(300)
def test_bench_self_launch_propagates_worker_failure():
    r = _run_bench(["--gpus", "2", "--rehearse-launch"], env_extra={"PBE_BENCH_FAIL_RANK": "1"})
    assert r.returncode == 3
    assert "rank 1 exited with code 3" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(300)
def test_bench_under_torch_distributed_run():
    """The driver's N > 1 form: torch.distributed.run sets RANK/WORLD_SIZE, so bench.py must NOT launch again."""
    import json
    import sys
    port = _free_port()
    r = _run_bench(["--gpus", "2", "--rehearse-launch"],
                   launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                             "--master-port", str(port)])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_bench_rejects_world_size_mismatch():
    r = _run_bench(["--gpus", "4", "--rehearse-launch"], env_extra={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
