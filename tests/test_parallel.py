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


    cut = parallel.partition_transport_blocks([5, 1, 1, 1, 8, 1, 1, 2], 3)
    assert cut[0] == 0 and cut[-1] == 8 and cut == sorted(cut) and len(cut) == 4
    sums = [sum([5, 1, 1, 1, 8, 1, 1, 2][a:b]) for a, b in zip(cut[:-1], cut[1:])]
    assert max(sums) <= 10                       # contiguous ranges, no part far above total / parts = 6.7


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


def _oracle_chain_fn(local_tbs, llr, harq, pay, ack, itm, max_iter):
    """Stands in for nrLDPC_hip_ulsch_decode on a rank without a GPU: the oracle's UL-SCH chain on the rank's transport
    blocks, same buffers and layout (ldpc.tb_layout) as the device call."""
    from openairinterface5g_amd import ldpc
    po, co, ho, segs = ldpc.tb_layout(local_tbs)
    for i, t in enumerate(local_tbs):
        hd = [harq[ho[i] + r * ldpc.HARQ_STRIDE: ho[i] + (r + 1) * ldpc.HARQ_STRIDE].numpy() for r in range(segs[i])]
        p, a, its, t["llrLen"] = O.ulsch_decode(t, llr[co[i]:co[i] + t["G"]].numpy(), hd, max_iter, t.get("round", 0),
                                                t.get("llrLen", 0), vec=True)
        pay[po[i]:po[i] + t["A"] // 8] = torch.from_numpy(p if a else np.zeros_like(p))  # a lost TB delivers zeros (tb_chain.hip)
        ack[i] = int(a)
        itm[i] = min(max(its), max_iter + 1)


def _tb_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openairinterface5g_amd import ldpc
        from test_gpu_tb_chain import make_tbs
        box = [None]
        if rank == 0:
            all_tbs = make_tbs()
            box[0] = [all_tbs[i] for i in (0, 2, 3, 4, 5, 10, 11, 12, 0, 2)]      # real descriptors, CPU-sized
        dist.broadcast_object_list(box, src=0)
        tbs = box[0]
        sh = parallel.ShardedUlsch(tbs, decode_fn=_oracle_chain_fn, numMaxIter=6, chunks=3)   # (the pipeline exercised whatever the model would pick)
        assert sh.cut[0] == 0 and sh.cut[-1] == len(tbs) and len(sh.cut) == world + 1 and sh.cut == sorted(sh.cut) and 0 < sh.cut[1] < len(tbs)
        # a peer's range travels and is decoded in up to three pieces (of whole transport blocks), the root's own range in one
        assert len(sh.chunk_cut[0]) == 2
        for r in range(1, world):
            c = sh.chunk_cut[r]
            assert c[0] == sh.cut[r] and c[-1] == sh.cut[r + 1] and c == sorted(c) and 2 <= len(c) <= 4
        if world == 2:
            assert len(sh.chunk_cut[1]) == 4 and sh.chunk_cut[1] == sorted(set(sh.chunk_cut[1]))
        po, co, ho, segs = ldpc.tb_layout(tbs)
        rng = np.random.default_rng(5)
        ref_tbs = [dict(t) for t in tbs]
        ref_harq = [[np.zeros(ldpc.HARQ_STRIDE, np.int16) for _ in range(c)] for c in segs]
        ok = True
        for rnd in range(2):
            llr = None
            if rank == 0:
                llr = torch.zeros(int(co[-1]), dtype=torch.int16)
                pays = []
                for i, t in enumerate(tbs):
                    p = rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) if rnd == 0 else sent[i]
                    pays.append(p)
                    c = O.dlsch_encode(t, p)
                    y = (1.0 - 2.0 * c.astype(np.float64) + (1.25 if rnd == 0 else 0.7) * rng.standard_normal(c.size)) * 8.0
                    llr[co[i]:co[i] + t["G"]] = torch.from_numpy(np.clip(np.rint(y), -127, 127).astype(np.int16))
                sent = pays
            pay, ack, itm = sh.decode(llr, rnd)
            harq_all = parallel.gather_ranges(sh.harq[:int(ho[sh.t1] - ho[sh.t0])], [(int(ho[a]), int(ho[b])) for a, b in sh.tb_ranges],
                                              int(ho[-1]))
            if rank == 0:
                n_ack = 0
                for i, t in enumerate(ref_tbs):
                    t["round"] = rnd
                    p, a, its, t["llrLen"] = O.ulsch_decode(t, llr[co[i]:co[i] + t["G"]].numpy(), ref_harq[i], 6, rnd,
                                                            t.get("llrLen", 0), vec=True)
                    ok &= bool(ack[i]) == a and int(itm[i]) == min(max(its), 7)
                    if a:
                        ok &= np.array_equal(pay[po[i]:po[i] + t["A"] // 8].numpy(), p)
                        n_ack += 1
                    for r in range(segs[i]):
                        ok &= np.array_equal(harq_all[ho[i] + r * ldpc.HARQ_STRIDE: ho[i] + (r + 1) * ldpc.HARQ_STRIDE].numpy(),
                                             ref_harq[i][r])
                ok &= (n_ack < len(tbs)) if rnd == 0 else (n_ack == len(tbs))
            else:
                assert pay is None and ack is None and itm is None
        if rank == 0:
            ret.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_transport_blocks_sharded_world2(built):
    """BASELINE configs[4] on CPU ranks: real transport-block descriptors cut into contiguous per-rank ranges, LLRs
    scattered point-to-point with exact sizes, the UL-SCH chain run per rank (oracle chain standing in for the GPU call),
    results gathered; two HARQ rounds with rank-resident soft buffers.  Everything equals the single-rank chain."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_tb_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def test_transport_blocks_sharded_world4(built):
    """The same with FOUR ranks: three peers, whose chunks the root posts chunk-major (every link busy at once) and whose
    results come back in the order they are produced -- the posting order of a real node, which two ranks cannot show."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_tb_worker, args=(r, 4, port, ret)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(400)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def _loop_worker(port, ret, backend="gloo", device="cpu"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend, rank=0, world_size=1)
    try:
        from openairinterface5g_amd import ldpc
        from test_gpu_tb_chain import make_tbs
        all_tbs = make_tbs()
        tbs = [all_tbs[i] for i in (0, 2, 3, 4, 5, 10, 11, 12, 0, 2, 3, 4)]
        sh = parallel.ShardedUlsch(tbs, decode_fn=_oracle_chain_fn, numMaxIter=6, loopback=3, chunks=3, device=torch.device(device),
                                   transport=parallel.CopyTransport() if backend == "gloo" else None)   # gloo cannot send to itself
        one = parallel.ShardedUlsch(tbs, decode_fn=_oracle_chain_fn, numMaxIter=6, device=torch.device(device))
        assert sh.world == 3 and len(sh.shares) == 3 and [len(c) for c in sh.chunk_cut] == [2, 4, 4]
        po, co, ho, segs = ldpc.tb_layout(tbs)
        rng = np.random.default_rng(6)
        ok = True
        for rnd in range(2):
            llr = torch.zeros(int(co[-1]), dtype=torch.int16)
            if rnd == 0:
                sent = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
            for i, t in enumerate(tbs):
                c = O.dlsch_encode(t, sent[i])
                y = (1.0 - 2.0 * c.astype(np.float64) + (1.25 if rnd == 0 else 0.7) * rng.standard_normal(c.size)) * 8.0
                llr[co[i]:co[i] + t["G"]] = torch.from_numpy(np.clip(np.rint(y), -127, 127).astype(np.int16))
            before = sh.p2p_bytes
            pay, ack, itm = sh.decode(llr, rnd)
            pay1, ack1, itm1 = one.decode(llr, rnd)
            ok &= torch.equal(pay[:int(po[-1])], pay1[:int(po[-1])]) and torch.equal(ack, ack1) and torch.equal(itm, itm1)
            # every LLR of the virtual peers and every result of theirs went through a send / receive pair
            expect = 2 * int(co[-1] - co[sh.cut[1]]) + int(po[-1] - po[sh.cut[1]]) + 5 * (len(tbs) - sh.cut[1])
            ok &= sh.p2p_bytes - before == expect
            harq = torch.cat([sh.shares[r].harq[:int(ho[sh.cut[r + 1]] - ho[sh.cut[r]])] for r in range(3)])
            ok &= torch.equal(harq, one.harq[:int(ho[-1])])
            ok &= (int(ack.sum()) < len(tbs)) if rnd == 0 else (int(ack.sum()) == len(tbs))
        ret.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_transport_blocks_loopback_one_process(built):
    """ShardedUlsch(loopback=3): one process plays three ranks, the virtual peers' LLR ranges and results travel through
    send / receive pairs to the process itself -- the protocol of the N > 1 run (views, chunking, posting order) with a
    world of one.  Here (gloo cannot send to its own rank) the pairs are matched in posting order and executed as copies;
    tests/test_bench.py drives the same protocol through RCCL on the GPU.  Results and soft buffers equal the unsharded
    run's over two HARQ rounds."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    p = ctx.Process(target=_loop_worker, args=(port, ret))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def test_slot_prediction_model():
    """parallel.predict_slot_ms: the N = 1 figure is the measured chain call; cutting the slot over N GPUs from ONE root is
    bounded by the root's links and the point-to-point groups, not by the decoders -- more ranks never cost more, the lower end
    of the range never exceeds the upper, and with free communication the prediction falls back to the per-rank chain time."""
    from openairinterface5g_amd import parallel as P
    res = {}
    for N in (1, 2, 4, 8):
        seg, llr, back = [1664 // N] * N, [31451136 // N] * N, [64 * 26650 // N] * N
        res[N] = P.predict_slot_ms(seg, llr, back)
    assert abs(res[1]["predicted_ms"] - P.SLOT_MODEL["chain_us_by_segments"][-1][1] / 1e3) < 1e-9 and res[1]["bound"] == "compute"
    assert res[2]["predicted_ms"] > res[4]["predicted_ms"] > res[8]["predicted_ms"] > res[1]["predicted_ms"]
    for N in (2, 4, 8):
        lo, hi = res[N]["predicted_ms_range"]
        assert lo <= hi == res[N]["predicted_ms"] and res[N]["bound"].startswith("link")
        assert res[N]["link_time_us"] == pytest.approx(31451136 / N / 50e3)
    keep = dict(P.SLOT_MODEL)
    try:
        P.SLOT_MODEL.update(group_us=0.0, group_us_low=0.0, link_GBps=1e9)
        free = P.predict_slot_ms([208] * 8, [31451136 // 8] * 8, [213200] * 8)
        assert free["predicted_ms"] == pytest.approx(max(P._chain_us(208), 3 * P._chain_us(208 / 3)) / 1e3)
    finally:
        P.SLOT_MODEL.clear()
        P.SLOT_MODEL.update(keep)
