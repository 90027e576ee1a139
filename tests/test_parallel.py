"""Multi-rank path on CPU: world_size 2, gloo backend (the data path has no collective; scatter/gather only at
the edges).  The local decode is injected (oracle) because there is no GPU here."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from common import make_llr
from openairinterface5g_amd import parallel


def test_shard_helpers():
    for n in (0, 1, 7, 64, 1024, 1025):
        for w in (1, 2, 3, 8):
            r = [parallel.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
    segs = [18, 1, 1, 9, 4, 4, 2, 18, 3, 3, 7, 1]
    own = parallel.shard_transport_blocks(segs, 4)
    assert sorted(i for o in own for i in o) == list(range(len(segs)))
    loads = [sum(segs[i] for i in o) for o in own]
    assert max(loads) <= 19                      # LPT: no rank above the largest TB + what balance allows
    assert max(loads) - min(loads) <= 3


def _worker(rank, world, port, n_blocks, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        BG, Z, R = 2, 16, 15
        llr = None
        if rank == 0:
            rng = np.random.default_rng(77)
            llr = torch.from_numpy(np.stack([make_llr(rng, BG, Z, R, k) for k in ([1.0, -1.0, "rand"] * n_blocks)[:n_blocks]]))

        def decode_fn(shard):
            res = [O.decode(BG, Z, R, shard[i].numpy(), 8) for i in range(shard.shape[0])]
            it = torch.tensor([r[0] for r in res], dtype=torch.int32)
            out = torch.from_numpy(np.stack([r[1] for r in res])) if res else torch.zeros((0, O.out_bytes(BG, Z, R, 0)), dtype=torch.uint8)
            return it, out

        out, it = parallel.decode_sharded(BG, Z, R, llr, n_blocks, decode_fn=decode_fn)
        if rank == 0:
            ok = True
            for i in range(n_blocks):
                n_ref, o_ref = O.decode(BG, Z, R, llr[i].numpy(), 8)
                ok &= (n_ref == int(it[i])) and np.array_equal(o_ref, out[i].numpy())
            ret.put(bool(ok) and out.shape[0] == n_blocks)
        else:
            assert out is None and it is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_blocks", [7, 8])
def test_scatter_decode_gather_world2(built, n_blocks):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_blocks, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True
