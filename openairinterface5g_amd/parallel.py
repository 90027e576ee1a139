"""Sharding of independent code blocks over the GPUs of one node (one process per GPU, torch.distributed).

The reference parallelises this path by handing every code segment to a CPU thread-pool worker
(openair1/PHY/NR_TRANSPORT/nr_ulsch_decoding.c:435-468); segments never exchange data, and the only
coupling is the transport-block-wide abort flag.  The multi-GPU analogue therefore needs NO collective on
the data path: each rank decodes its own contiguous range of blocks (whole transport blocks stay on one
rank so that TB-level CRC/abort stays local).  Collectives appear only at the edges, when a batch arrives
on one rank: one scatter of the LLR shards out (root -> peers over the direct xGMI links) and one gather of
the packed bits / pass counts back.  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of `n_items` for `rank` (first n % world ranks get one more)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_transport_blocks(segments_per_tb: Sequence[int], world: int) -> List[List[int]]:
    """Assign whole transport blocks to ranks, balancing the number of code segments (LPT greedy).
    Returns per rank the list of TB indices (ascending)."""
    order = sorted(range(len(segments_per_tb)), key=lambda i: (-segments_per_tb[i], i))
    load = [0] * world
    owner: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[r].append(i)
        load[r] += segments_per_tb[i]
    return [sorted(o) for o in owner]


def scatter_blocks(llr_root, n_blocks: int, row_bytes: int, root: int = 0, group=None, device=None):
    """Distribute an [n_blocks, row_bytes] int8 batch that lives on `root` (pass None elsewhere).
    Returns this rank's [hi-lo, row_bytes] shard."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n_blocks, rank, world)
    if device is None:
        device = llr_root.device if llr_root is not None else torch.device("cpu")
    shard = torch.empty((hi - lo, row_bytes), dtype=torch.int8, device=device)
    if world == 1:
        shard.copy_(llr_root)
        return shard
    # ranks hold shards of (at most one row) different length: pad to the longest for dist.scatter
    longest = shard_range(n_blocks, 0, world)[1]
    buf = torch.zeros((longest, row_bytes), dtype=torch.int8, device=device)
    parts = None
    if rank == root:
        parts = []
        for r in range(world):
            a, b = shard_range(n_blocks, r, world)
            p = torch.zeros((longest, row_bytes), dtype=torch.int8, device=device)
            p[:b - a] = llr_root[a:b]
            parts.append(p)
    dist.scatter(buf, parts, src=root, group=group)
    shard.copy_(buf[:hi - lo])
    return shard


def gather_results(out_local, n_iter_local, n_blocks: int, root: int = 0, group=None):
    """Collect every rank's [n_local, out_bytes] uint8 bits and [n_local] int32 pass counts on `root`
    (returns (out, n_iter) there, (None, None) elsewhere)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1:
        return out_local, n_iter_local
    longest = shard_range(n_blocks, 0, world)[1]
    ob = out_local.shape[1]
    pad_out = torch.zeros((longest, ob), dtype=out_local.dtype, device=out_local.device)
    pad_it = torch.zeros((longest,), dtype=n_iter_local.dtype, device=n_iter_local.device)
    pad_out[:out_local.shape[0]] = out_local
    pad_it[:n_iter_local.shape[0]] = n_iter_local
    outs = [torch.empty_like(pad_out) for _ in range(world)] if rank == root else None
    its = [torch.empty_like(pad_it) for _ in range(world)] if rank == root else None
    dist.gather(pad_out, outs, dst=root, group=group)
    dist.gather(pad_it, its, dst=root, group=group)
    if rank != root:
        return None, None
    out = torch.cat([outs[r][:shard_range(n_blocks, r, world)[1] - shard_range(n_blocks, r, world)[0]] for r in range(world)])
    it = torch.cat([its[r][:shard_range(n_blocks, r, world)[1] - shard_range(n_blocks, r, world)[0]] for r in range(world)])
    return out, it


def decode_sharded(BG: int, Z: int, R: int, llr_root, n_blocks: int, numMaxIter: int = 8, root: int = 0, group=None,
                   decode_fn: Optional[Callable] = None, device=None):
    """Scatter -> local decode -> gather.  `decode_fn(llr_shard) -> (n_iter, out)` defaults to the HIP batch
    decoder on this rank's GPU (openairinterface5g_amd.ldpc.decode_batch_device); the CPU tests inject their own."""
    import torch
    from . import ldpc
    row = ldpc.num_llr(BG, Z, R)
    shard = scatter_blocks(llr_root, n_blocks, row, root, group, device)
    if decode_fn is None:
        out = torch.zeros((shard.shape[0], ldpc.out_bytes(BG, Z, R)), dtype=torch.uint8, device=shard.device)
        it = torch.zeros((shard.shape[0],), dtype=torch.int32, device=shard.device)
        if shard.shape[0]:
            ldpc.decode_batch_device(BG, Z, R, shard, out, it, numMaxIter=numMaxIter)
    else:
        it, out = decode_fn(shard)
    return gather_results(out, it, n_blocks, root, group)
