"""Multi-GPU plumbing of the path (torch.distributed: RCCL on GPUs, gloo in the CPU tests).

The path shards by independent (track, channel) units (th_shard_assign); its ONLY exchange step
is the global dB range of update_spec_imgs (core/mod.rs:169-180): two floats, one all-reduce.
The image-tile gather to a root rank is a viewer-side convenience outside the timed hot path; it
uses per-peer send/recv so that on xGMI all inbound links run concurrently (no ring).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np


def allreduce_min_max(local_min, local_max, dist=None, group=None, device=None):
    """Global (min, max) over all ranks from each rank's (min, max): ONE 2-element all-reduce of
    [min, -max] with MIN.  Works on CPU tensors (gloo) and GPU tensors (RCCL)."""
    import torch
    t = torch.tensor([float(local_min), -float(local_max)], dtype=torch.float32, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    mn, neg_mx = t.tolist()
    return mn, -neg_mx


def gather_bytes_to_root(payloads: Sequence[bytes], dist, root: int = 0, group=None, device=None) -> Optional[List[List[bytes]]]:
    """Every rank contributes a list of byte strings (encoded tiles); the root gets them all.
    Sizes first (one small all-gather), then one point-to-point transfer per peer, all posted at
    once (batch_isend_irecv) so the root's inbound links are used in parallel."""
    import torch
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = torch.tensor([len(p) for p in payloads], dtype=torch.int64, device=device)
    count = torch.tensor([len(payloads)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    max_n = int(max(int(c.item()) for c in counts))
    padded = torch.zeros(max_n, dtype=torch.int64, device=device)
    padded[: len(payloads)] = sizes
    all_sizes = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(all_sizes, padded, group=group)
    blob = torch.frombuffer(bytearray(b"".join(payloads)) or bytearray(1), dtype=torch.uint8).to(device or "cpu")
    if rank != root:
        if len(payloads):
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, blob, root, group)]):
                req.wait()
        return None
    bufs, ops = {}, []
    for r in range(world):
        n = int(all_sizes[r][: int(counts[r].item())].sum().item())
        if r == root or n == 0:
            continue
        bufs[r] = torch.empty(n, dtype=torch.uint8, device=device or "cpu")
        ops.append(dist.P2POp(dist.irecv, bufs[r], r, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    out: List[List[bytes]] = []
    for r in range(world):
        if r == root:
            out.append(list(payloads))
            continue
        data = bufs[r].cpu().numpy().tobytes() if r in bufs else b""
        parts, off = [], 0
        for i in range(int(counts[r].item())):
            n = int(all_sizes[r][i].item())
            parts.append(data[off:off + n])
            off += n
        out.append(parts)
    return out


def gather_tensor_to_root(local, dist, root: int = 0, group=None):
    """Device-resident form of the image-tile gather: every rank contributes ONE contiguous tensor that already lives
    where the collective runs (on GPUs: the RGBA tile buffer th_raster_tiles_dev wrote, no host hop, no `bytes`); the root
    gets a list of world tensors (its own entry is `local` itself), the others get None.  Sizes go first (one 1-element
    all-gather), then one point-to-point transfer per peer, all posted at once (batch_isend_irecv): on xGMI the root's
    seven inbound links run concurrently instead of a ring that would be bound by one link (core/mod.rs:169-180 is the
    path's only coupling; this gather is the viewer-side collection of the results, outside the timed step).
    `root` and the returned list are indexed by rank IN `group` (the default group: global ranks); P2POp takes global
    ranks, so peers are translated (with a sub-group whose ranks are not 0..n-1 the untranslated peer hung)."""
    import torch
    world, rank = dist.get_world_size(group), dist.get_rank(group)

    def peer(r):  # group rank -> global rank
        return r if group is None else dist.get_global_rank(group, r)
    flat = local.contiguous().view(-1)
    n_local = torch.tensor([flat.numel()], dtype=torch.int64, device=flat.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    if rank != root:
        if flat.numel():
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, flat, peer(root), group)]):
                req.wait()
        return None
    out, ops = [None] * world, []
    for r in range(world):
        n = int(counts[r].item())
        if r == root:
            out[r] = flat
            continue
        out[r] = torch.empty(n, dtype=flat.dtype, device=flat.device)
        if n:
            ops.append(dist.P2POp(dist.irecv, out[r], peer(r), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out
