"""Batch-axis sharding across the GPUs of one node.

Every image triple is independent (GroupNorm and attention are per-sample, a CFG pair stays on
one GPU), so the path shards with NO per-step collective.  One process per GPU
(``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" for the CPU tests):

  * the work partition itself is ``pbe_amd.testbench.rank_batches`` (whole batches dealt round-robin, drop_last like the
    reference's test bench, scripts/inference_test_bench.py:301); ``bench.py`` gives every rank its own B synthetic triples,
  * ``broadcast_weights_`` ONE-OFF broadcast of the weights from rank 0 in a few large contiguous
                           buffers (xGMI is point-to-point: a broadcast is per-link bound, so few big
                           messages, not 1 400 small ones),
  * ``gather_images``      finished uint8 images to rank 0.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

BUCKET_BYTES = 1 << 30          # 1 GiB buckets: 2.6-5.2 GB of weights -> 3-6 broadcasts


def _buckets(tensors: Sequence[torch.Tensor], limit: int) -> Iterable[List[torch.Tensor]]:
    cur, size = [], 0
    for t in tensors:
        nb = t.numel() * t.element_size()
        if cur and size + nb > limit:
            yield cur
            cur, size = [], 0
        cur.append(t)
        size += nb
    if cur:
        yield cur


@torch.no_grad()
def broadcast_weights_(module: torch.nn.Module, src: int = 0, bucket_bytes: int = BUCKET_BYTES) -> Dict[str, float]:
    """In-place broadcast of every parameter and buffer of `module` from `src`, bucketed per dtype
    into contiguous staging buffers.  Returns {'bytes', 'messages'}."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return {"bytes": 0.0, "messages": 0.0}
    by_dtype: Dict[torch.dtype, List[torch.Tensor]] = {}
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t.data)
    total, msgs = 0, 0
    for dtype, tensors in sorted(by_dtype.items(), key=lambda kv: str(kv[0])):
        for bucket in _buckets(tensors, bucket_bytes):
            n = sum(t.numel() for t in bucket)
            flat = torch.empty(n, dtype=dtype, device=bucket[0].device)
            if dist.get_rank() == src:
                off = 0
                for t in bucket:
                    flat[off:off + t.numel()].copy_(t.reshape(-1))
                    off += t.numel()
            dist.broadcast(flat, src=src)
            if dist.get_rank() != src:
                off = 0
                for t in bucket:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()
            total += n * flat.element_size()
            msgs += 1
            del flat
    for m in module.modules():
        if hasattr(m, "invalidate_packs"):
            m.invalidate_packs()
    return {"bytes": float(total), "messages": float(msgs)}


@torch.no_grad()
def weights_checksum(module: torch.nn.Module) -> bool:
    """True when every rank holds the same parameters / buffers: (sum, sum of |x|) over all floating tensors in fp64, MAX- and
    MIN-reduced - equal on all ranks iff both reductions agree with the local value."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return True
    dev = next(module.parameters()).device
    acc = torch.zeros(2, dtype=torch.float64, device=dev)
    for t in list(module.parameters()) + list(module.buffers()):
        if t.is_floating_point():
            d = t.data.double()
            acc[0] += d.sum()
            acc[1] += d.abs().sum()
    if dist.get_backend() != "nccl":
        acc = acc.cpu()
    hi, lo = acc.clone(), acc.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return bool(torch.equal(hi, lo))


def gather_images(images_u8: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
    """[b,3,H,W] uint8 per rank -> [W*b,3,H,W] on `dst` (rank-major), None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return images_u8
    world = dist.get_world_size()
    if dist.get_rank() == dst:
        out = [torch.empty_like(images_u8) for _ in range(world)]
        dist.gather(images_u8, out, dst=dst)
        return torch.cat(out, 0)
    dist.gather(images_u8, None, dst=dst)
    return None
