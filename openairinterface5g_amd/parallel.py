"""Sharding of independent code blocks / transport blocks over the GPUs of one node (one process per GPU,
torch.distributed; the single-process, many-GPU form of the same split lives inside the library: NRLDPC_HIP_DEVICES).

The reference parallelises this path by handing every code segment to a CPU thread-pool worker
(openair1/PHY/NR_TRANSPORT/nr_ulsch_decoding.c:435-468, pull loop SCHED_NR/phy_procedures_nr_gNB.c:911-917); segments
never exchange data, and the only coupling is the transport-block-wide abort flag.  The multi-GPU analogue therefore
needs NO collective on the data path: each rank works on a contiguous range of blocks, whole transport blocks staying on
one rank so that TB CRC, abort flag and HARQ soft buffers stay local.  Communication appears only at the edges, when a
slot's data arrives on one rank: the LLR ranges go out root -> peers (point-to-point, exact sizes: over the direct xGMI
links, no padding, no staging copies) and the payload bytes / ACKs / pass counts come back the same way.  Backend
"nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
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
    Returns per rank the list of TB indices (ascending).  (Scattered ownership: for data that is already distributed;
    a slot that arrives on one rank is cut into contiguous ranges instead, see partition_transport_blocks.)"""
    order = sorted(range(len(segments_per_tb)), key=lambda i: (-segments_per_tb[i], i))
    load = [0] * world
    owner: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[r].append(i)
        load[r] += segments_per_tb[i]
    return [sorted(o) for o in owner]


def partition_transport_blocks(costs: Sequence[float], parts: int) -> List[int]:
    """Contiguous ranges of whole transport blocks balanced by cost; returns parts + 1 cut points.  Same rule as the
    library's in-process split (csrc/tb_api.inc.cpp tb_partition): a block goes to the part whose target its centre
    falls under."""
    total = float(sum(costs))
    cut, acc, i = [0], 0.0, 0
    for k in range(1, parts):
        target = total * k / parts
        while i < len(costs) and acc + costs[i] * 0.5 <= target:
            acc += costs[i]
            i += 1
        cut.append(i)
    cut.append(len(costs))
    return cut


def tb_cost(tb: dict) -> float:
    """Decoder work of a transport block ~ segments x edges x Zc."""
    from . import ldpc
    s = ldpc.nr_segmentation(tb["A"] + (24 if tb["A"] > 3824 else 16), tb["BG"])
    return float(s["C"] * s["Z"] * (316 if tb["BG"] == 1 else 197))


def _rank_world(group=None):
    """(rank, world); (0, 1) without an initialised process group, so that the same code runs on a single GPU."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


# ---- edges: exact-size point-to-point -------------------------------------------------------------------------------
def scatter_ranges(flat_root, ranges: Sequence[Tuple[int, int]], dtype, root: int = 0, group=None, device=None, out=None):
    """`flat_root`: 1-D tensor on `root` (None elsewhere); rank r receives elements [ranges[r][0], ranges[r][1]).
    Slices of the root tensor are sent as they are (views); a receiver gets exactly its share, into `out` if given."""
    import torch
    import torch.distributed as dist
    rank, world = _rank_world(group)
    lo, hi = ranges[rank]
    if world == 1:
        return flat_root[lo:hi]
    as_bytes = lambda t: t.view(torch.uint8) if t.dtype == torch.int16 else t   # (no 16-bit integers in the NCCL process group)
    if rank == root:
        device = flat_root.device if device is None else device
        ops = [dist.P2POp(dist.isend, as_bytes(flat_root[a:b]), dist.get_global_rank(group, r) if group is not None else r, group)
               for r, (a, b) in enumerate(ranges) if r != root and b > a]
        mine = flat_root[lo:hi]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return mine
    shard = out[:hi - lo] if out is not None else torch.empty((hi - lo,), dtype=dtype, device=device if device is not None else "cpu")
    if hi > lo:
        src = dist.get_global_rank(group, root) if group is not None else root
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, as_bytes(shard), src, group)]):
            w.wait()
    return shard


def gather_ranges(local, ranges: Sequence[Tuple[int, int]], total: int, root: int = 0, group=None):
    """Inverse of scatter_ranges: returns the assembled [total] tensor on `root` (None elsewhere)."""
    import torch
    import torch.distributed as dist
    rank, world = _rank_world(group)
    lo, hi = ranges[rank]
    if world == 1:
        return local[:hi - lo] if hi - lo == total else torch.cat([local[:hi - lo], local.new_zeros(total - (hi - lo))])
    if rank == root:
        out = torch.empty((total,), dtype=local.dtype, device=local.device)
        out[lo:hi] = local[:hi - lo]
        as_bytes = lambda t: t.view(torch.uint8) if t.dtype == torch.int16 else t
        ops = [dist.P2POp(dist.irecv, as_bytes(out[a:b]), dist.get_global_rank(group, r) if group is not None else r, group)
               for r, (a, b) in enumerate(ranges) if r != root and b > a]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return out
    if hi > lo:
        dst = dist.get_global_rank(group, root) if group is not None else root
        loc = local[:hi - lo].contiguous()
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, loc.view(torch.uint8) if loc.dtype == torch.int16 else loc, dst, group)]):
            w.wait()
    return None


# ---- raw code blocks ---------------------------------------------------------------------------------------------------
def decode_sharded(BG: int, Z: int, R: int, llr_root, n_blocks: int, numMaxIter: int = 8, root: int = 0, group=None,
                   decode_fn: Optional[Callable] = None, device=None):
    """[n_blocks, ncols*Z] int8 LLRs on `root` -> contiguous block ranges to the ranks -> local decode -> packed bits and
    pass counts back on `root`.  `decode_fn(llr_shard) -> (n_iter, out)` defaults to the HIP batch decoder on this
    rank's GPU (the library runs on the GPU that owns the tensors); the CPU tests inject their own."""
    import torch
    from . import ldpc
    rank, world = _rank_world(group)
    row, ob = ldpc.num_llr(BG, Z, R), ldpc.out_bytes(BG, Z, R)
    blocks = [shard_range(n_blocks, r, world) for r in range(world)]
    flat = llr_root.reshape(-1) if llr_root is not None else None
    shard = scatter_ranges(flat, [(a * row, b * row) for a, b in blocks], torch.int8, root, group, device).reshape(-1, row)
    if decode_fn is None:
        out = torch.zeros((shard.shape[0], ob), dtype=torch.uint8, device=shard.device)
        it = torch.zeros((shard.shape[0],), dtype=torch.int32, device=shard.device)
        if shard.shape[0]:
            ldpc.decode_batch_device(BG, Z, R, shard, out, it, numMaxIter=numMaxIter)
    else:
        it, out = decode_fn(shard)
    out_all = gather_ranges(out.reshape(-1), [(a * ob, b * ob) for a, b in blocks], n_blocks * ob, root, group)
    it_all = gather_ranges(it, blocks, n_blocks, root, group)
    if rank != root:
        return None, None
    return out_all.reshape(n_blocks, ob), it_all


# ---- what a slot cut over N GPUs should cost: written down BEFORE hardware measures it -----------------------------------
# Every number is either measured on one MI355X (file named) or an assumption (marked); bench.py puts the prediction for its N
# into the line (strong_scaling_slot.predicted_ms), so that the first real N > 1 record can be read against it.
SLOT_MODEL = {
    # UL-SCH chain, device resident, one call: microseconds by code segments in the call (BG1 Zc = 384, 64QAM, 3 passes to the
    # CRC, first transmissions on the cut graph): profiles/r06/tb_latency.txt (1 .. 64 transport blocks of 26 segments) -- the
    # slot of 64 blocks back to back: 118 us (final run of round 6, Zc = 384 instantiations; 132 before the front step's
    # straight-line scatter)
    "chain_us_by_segments": [(26, 37.7), (52, 38.0), (104, 38.8), (208, 40.8), (416, 50.4), (832, 82.6), (1664, 118.0)],
    # one batch_isend_irecv group through RCCL (Python, group launch, stream ordering): from the loopback slot of
    # profiles/r05/bench_dist1.json -- 1409 us per slot with 4 virtual ranks = 12 groups + 10 chain calls of 423 us in total +
    # 47.5 MB of device-local copies (~45 us): (1409 - 423 - 45) / 12
    # -- an UPPER bound for a real rank: in the loopback run one process posts BOTH ends of every pair
    "group_us": 78.0,
    "group_us_low": 39.0,        # ASSUMPTION for the lower end of the prediction: half of it (a rank posts one end only)
    # ASSUMPTION: one xGMI link, one direction, through RCCL send / receive: 50 GB/s (peak 76.8 GB/s per direction per link;
    # the root reaches every peer over a link of its own, /opt/skills/guides/MI355X_MICROARCH.md).  Not measurable on a 1-GPU box.
    "link_GBps": 50.0,
}


def _chain_us(segments: float) -> float:
    pts = SLOT_MODEL["chain_us_by_segments"]
    if segments <= 0:
        return 0.0
    if segments <= pts[0][0]:
        return pts[0][1]
    for (x0, y0), (x1, y1) in zip(pts, pts[1:]):
        if segments <= x1:
            return y0 + (y1 - y0) * (segments - x0) / (x1 - x0)
    (x0, y0), (x1, y1) = pts[-2], pts[-1]
    return y1 + (y1 - y0) / (x1 - x0) * (segments - x1)          # beyond a slot: the last slope (full workgroup rounds)


def predict_slot_ms(segments_per_rank: Sequence[int], llr_bytes_per_rank: Sequence[int], result_bytes_per_rank: Sequence[int],
                    chunks: int = 3, root: int = 0) -> dict:
    """ShardedUlsch.decode() of one slot on len(segments_per_rank) GPUs as a timeline of measured pieces (SLOT_MODEL):
    the root posts `chunks` send groups one after the other (group_us each on its host thread), every peer's chunk k is on
    its link from the moment its group is posted and the chunk before it has left (bytes / link_GBps), a peer decodes chunk
    k when it has arrived and the chunk before it is decoded (chain_us of the chunk's segments), returns it with one more
    group; the root decodes its own range meanwhile.  The slot is done when the root has its own range and every peer's
    last results.  N = 1: the chain call alone."""
    world = len(segments_per_rank)
    if SLOT_MODEL.get("_inner") is None and world > 1:        # the range [group_us_low, group_us]: predicted_ms is its upper end
        SLOT_MODEL["_inner"] = True
        try:
            hi = predict_slot_ms(segments_per_rank, llr_bytes_per_rank, result_bytes_per_rank, chunks, root)
            keep = SLOT_MODEL["group_us"]
            SLOT_MODEL["group_us"] = SLOT_MODEL["group_us_low"]
            lo = predict_slot_ms(segments_per_rank, llr_bytes_per_rank, result_bytes_per_rank, chunks, root)
            SLOT_MODEL["group_us"] = keep
        finally:
            SLOT_MODEL["_inner"] = None
        hi["predicted_ms_range"] = [lo["predicted_ms"], hi["predicted_ms"]]
        return hi
    g, rate = SLOT_MODEL["group_us"], SLOT_MODEL["link_GBps"] * 1e3          # bytes per microsecond
    own = _chain_us(segments_per_rank[root])
    if world == 1:
        return {"predicted_ms": own / 1e3, "root_chain_us": own, "bound": "compute"}
    posted = [g * (k + 1) for k in range(chunks)]
    root_done = posted[-1] + own                        # (its own call is enqueued behind the last send group)
    worst, detail = 0.0, []
    for r in range(world):
        if r == root or segments_per_rank[r] == 0:
            continue
        arrive = dec_end = 0.0
        seg_k, byt_k = segments_per_rank[r] / chunks, llr_bytes_per_rank[r] / chunks
        for k in range(chunks):
            arrive = max(posted[k], arrive) + byt_k / rate
            dec_end = max(arrive, dec_end) + _chain_us(seg_k)
        back = dec_end + g + result_bytes_per_rank[r] / chunks / rate
        detail.append({"rank": r, "last_chunk_arrives_us": arrive, "decoded_us": dec_end, "results_on_root_us": back})
        worst = max(worst, back)
    total = max(root_done, worst)
    link_only = max(llr_bytes_per_rank[r] for r in range(world) if r != root) / rate
    return {"predicted_ms": total / 1e3, "root_chain_us": own, "root_done_us": root_done, "slowest_peer": max(detail, key=lambda d: d["results_on_root_us"]),
            "link_time_us": link_only, "bound": "link + group overhead" if worst > root_done else "root compute",
            "assumptions": {"group_us": g, "link_GBps": SLOT_MODEL["link_GBps"]}}


# ---- a slot's transport blocks (BASELINE configs[4]) -------------------------------------------------------------------
class _DistTransport:
    """point-to-point through torch.distributed (nccl = RCCL on ROCm, gloo on CPU)"""

    def op(self, kind, tensor, peer, group):
        import torch
        import torch.distributed as dist
        if tensor.dtype == torch.int16:      # the NCCL / RCCL process group has no 16-bit integer type: the LLRs travel as bytes
            tensor = tensor.view(torch.uint8)
        return dist.P2POp(dist.isend if kind == "send" else dist.irecv, tensor, peer, group)

    def batch(self, ops):
        import torch.distributed as dist
        return dist.batch_isend_irecv(ops) if ops else []


class CopyTransport:
    """Stand-in for a backend that cannot send to its own rank (gloo): the sends and receives of a loopback run are
    matched in posting order -- what NCCL does with the operations of one group call -- and executed as copies.  For the
    CPU tests of the loopback protocol; the GPU test drives the same protocol through RCCL."""

    class _Done:
        def wait(self):
            return True

    def __init__(self):
        self.pending = []

    def op(self, kind, tensor, peer, group):
        return (kind, tensor)

    def batch(self, ops):
        for kind, t in ops:
            if kind == "send":
                self.pending.append(t)
            else:
                src = self.pending.pop(0)
                assert src.numel() == t.numel() and src.dtype == t.dtype, "send / receive mismatch"
                t.copy_(src)
        return [self._Done()]


class _Share:
    """What ONE rank holds of a slot: its contiguous range of whole transport blocks, the HARQ soft buffers of those blocks
    (they stay here from round to round), its receive buffer for the LLRs (peers) and its result buffers."""

    def __init__(self, parent, rank: int):
        import torch
        p = parent
        self.p, self.rank = p, rank
        self.t0, self.t1 = p.cut[rank], p.cut[rank + 1]
        n_h = int(p.ho[self.t1] - p.ho[self.t0])
        self.harq = torch.zeros((max(n_h, 1),), dtype=torch.int16, device=p.device)
        n_loc = self.t1 - self.t0
        n_llr = max(int(p.co[self.t1] - p.co[self.t0]), 1)
        self.llr = None if rank == p.root else torch.empty((n_llr,), dtype=torch.int16, device=p.device)
        self.pay = torch.zeros((max(int(p.po[self.t1] - p.po[self.t0]), 1),), dtype=torch.uint8, device=p.device)
        self.ack = torch.zeros((max(n_loc, 1),), dtype=torch.uint8, device=p.device)
        self.itm = torch.zeros((max(n_loc, 1),), dtype=torch.int32, device=p.device)
        self.prepared = {}

    def lo(self, off, i):
        return int(off[i] - off[self.t0])

    def decode_chunk(self, a: int, b: int, llr, rnd: int):
        """transport blocks [a, b) (global indices, inside this share's range) through the chain; llr = this share's LLRs"""
        if b <= a:
            return
        p = self.p
        views = (llr[self.lo(p.co, a):], self.harq[self.lo(p.ho, a):], self.pay[self.lo(p.po, a):], self.ack[a - self.t0:], self.itm[a - self.t0:])
        tbs = p.tbs[a:b]
        if p.decode_fn is not None:
            for t in tbs:
                t["round"] = rnd
            p.decode_fn(tbs, *views, p.numMaxIter)
            return
        # the descriptor array is marshalled once per (LLR buffer, chunk, round) and resubmitted slot after slot
        key = (llr.data_ptr(), a, b, rnd)
        if key not in self.prepared:
            for t in tbs:
                t["round"] = rnd
            self.prepared[key] = p.ldpc.PreparedTbBatch(tbs, views[2], views[0], views[1], views[3], views[4], p.numMaxIter)
        self.prepared[key].decode()


class ShardedUlsch:
    """A slot's PUSCH transport blocks decoded on the GPUs of all ranks: the UL-SCH chain of the library
    (nrLDPC_hip_ulsch_decode: de-interleave, rate de-match with HARQ combining, LDPC decode with CRC stop, reassembly,
    TB CRC) runs on every rank for its contiguous range of whole transport blocks; the HARQ soft buffers of a block live
    on the rank that owns it and stay there from round to round.

    All ranks construct it with the same descriptor list (broadcast it first if only the root has it) and call decode()
    together; the LLRs -- one flat int16 tensor in the layout of ldpc.tb_layout(tbs) -- are needed on `root` only.

    decode() is a pipeline over `chunks` sub-ranges of every peer's range (the reference keeps all segments of a slot in
    flight together, nr_ulsch_decoding.c:435-468): the root posts the LLR sends of all chunks, chunk-major, and decodes
    its own range while they drain; a peer posts all its receives, decodes chunk k as soon as it has arrived -- while
    chunks k+1.. are still on the links -- and returns that chunk's payload / ACKs / pass counts at once; the root's
    receives of the results were posted behind its sends.  With the nccl backend a work's wait() orders streams; with
    gloo (CPU tests) the same calls block.  On a peer every chunk is a call of its own into the library, whose plan cache
    (csrc/tb_api.inc.cpp TbPlanCache) holds them side by side: from the second slot on no chunk call builds or uploads
    anything.

    loopback = V > 1 (one process, world size 1): the slot is cut for V virtual ranks, all of them this process, and every
    LLR range / result range of the virtual peers travels by a real point-to-point pair -- isend and irecv to this very
    rank inside one batch_isend_irecv, the same views, chunking and posting order as with V processes.  That is how the
    send / receive path is exercised on a box with one GPU (RCCL executes ncclSend / ncclRecv to self; the peers' chain
    calls run on the same GPU, one after the other)."""

    def __init__(self, tbs: Sequence[dict], root: int = 0, group=None, device=None, numMaxIter: int = 8,
                 decode_fn: Optional[Callable] = None, chunks: Optional[int] = None, loopback: int = 0, transport=None):
        import torch
        from . import ldpc
        self.ldpc, self.root, self.group, self.numMaxIter, self.decode_fn = ldpc, root, group, numMaxIter, decode_fn
        self.rank, self.world = _rank_world(group)
        self.loopback = loopback > 1
        if self.loopback:
            assert self.world == 1 and root == 0, "loopback: one process plays every rank"
            self.world = int(loopback)               # virtual ranks from here on
        self.tbs = [dict(t) for t in tbs]
        self.po, self.co, self.ho, self.segs = ldpc.tb_layout(self.tbs)
        costs = [tb_cost(t) for t in self.tbs]
        self.cut = partition_transport_blocks(costs, self.world)
        self.device = torch.device("cpu") if device is None else device
        self.llr_ranges = [(int(self.co[a]), int(self.co[b])) for a, b in zip(self.cut[:-1], self.cut[1:])]
        self.pay_ranges = [(int(self.po[a]), int(self.po[b])) for a, b in zip(self.cut[:-1], self.cut[1:])]
        self.tb_ranges = list(zip(self.cut[:-1], self.cut[1:]))
        # chunk k of rank r = transport blocks [chunk_cut[r][k], chunk_cut[r][k+1]) (global indices), balanced by cost;
        # the root's own range is one piece (nothing travels)
        if chunks is None:
            # pipeline depth by the model of measured pieces (SLOT_MODEL): every chunk costs the root one more send group, so
            # the more peers there are the fewer chunks pay -- 3 per peer for two ranks, 2 for four, 1 for eight
            chunks = 1
            if self.world > 1:
                seg_r = [int(sum(self.segs[a:b])) for a, b in self.tb_ranges]
                llr_r = [int(self.co[b] - self.co[a]) * 2 for a, b in self.tb_ranges]
                res_r = [int(self.po[b] - self.po[a]) + 5 * int(b - a) for a, b in self.tb_ranges]
                chunks = min((1, 2, 3), key=lambda n: predict_slot_ms(seg_r, llr_r, res_r, chunks=n, root=root)["predicted_ms"])
        self.chunks = int(chunks)
        self.chunk_cut = []
        for r, (a, b) in enumerate(self.tb_ranges):
            n = 1 if (r == root or self.world == 1) else max(1, min(chunks, b - a))
            self.chunk_cut.append([a + c for c in partition_transport_blocks(costs[a:b], n)])
        self.shares = {r: _Share(self, r) for r in (range(self.world) if self.loopback else [self.rank])}
        me = self.shares[self.rank]
        self.t0, self.t1, self.harq = me.t0, me.t1, me.harq     # (this rank's own share, as before)
        self._all = None
        self.p2p_bytes = 0            # bytes this process has put on point-to-point sends (LLRs out / results back)
        self.transport = _DistTransport() if transport is None else transport

    def _sent(self, t):
        self.p2p_bytes += t.numel() * t.element_size()
        return t

    def decode(self, llr_root, rnd: int = 0):
        """Returns (payload uint8 flat in the tb_layout offsets, ack uint8[n_tb], iter_max int32[n_tb]) on root,
        (None, None, None) elsewhere."""
        import torch
        import torch.distributed as dist
        T = self.transport
        root, world, rank = self.root, self.world, self.rank
        me = self.shares[rank]
        n_loc = me.t1 - me.t0
        if world == 1:
            me.decode_chunk(me.t0, me.t1, llr_root, rnd)
            total = int(self.po[-1])
            return (me.pay[:total] if me.pay.numel() >= total else torch.cat([me.pay, me.pay.new_zeros(total - me.pay.numel())])), \
                me.ack[:len(self.tbs)], me.itm[:len(self.tbs)]
        g = self.group
        loop = self.loopback
        peer = (lambda r: 0) if loop else ((lambda r: dist.get_global_rank(g, r)) if g is not None else (lambda r: r))
        n_rounds = max(len(c) - 1 for c in self.chunk_cut)
        live = lambda r, k: r != root and k + 1 < len(self.chunk_cut[r]) and self.chunk_cut[r][k + 1] > self.chunk_cut[r][k]

        def llr_recv_op(sh, k):       # a peer's receive of its chunk k
            c = self.chunk_cut[sh.rank]
            return T.op("recv", sh.llr[sh.lo(self.co, c[k]):sh.lo(self.co, c[k + 1])], peer(root), g)

        def result_send_ops(sh, k):   # a peer's results of chunk k
            a, b = self.chunk_cut[sh.rank][k], self.chunk_cut[sh.rank][k + 1]
            return [T.op("send", self._sent(sh.pay[sh.lo(self.po, a):sh.lo(self.po, b)]), peer(root), g),
                    T.op("send", self._sent(sh.ack[a - sh.t0:b - sh.t0]), peer(root), g),
                    T.op("send", self._sent(sh.itm[a - sh.t0:b - sh.t0]), peer(root), g)]

        if rank == root:
            if self._all is None:     # assembled results of the whole slot (persistent: the per-slot path allocates nothing)
                self._all = (torch.zeros((int(self.po[-1]),), dtype=torch.uint8, device=self.device),
                             torch.zeros((len(self.tbs),), dtype=torch.uint8, device=self.device),
                             torch.zeros((len(self.tbs),), dtype=torch.int32, device=self.device))
            pay_all, ack_all, itm_all = self._all

            def result_recv_ops(r, k):
                a, b = self.chunk_cut[r][k], self.chunk_cut[r][k + 1]
                return [T.op("recv", pay_all[int(self.po[a]):int(self.po[b])], peer(r), g),
                        T.op("recv", ack_all[a:b], peer(r), g), T.op("recv", itm_all[a:b], peer(r), g)]

            works, arrived = [], {}
            for k in range(n_rounds):                                  # LLRs out, chunk-major: every link busy at once
                ops = []
                for r in range(world):
                    if live(r, k):
                        c = self.chunk_cut[r]
                        ops.append(T.op("send", self._sent(llr_root[int(self.co[c[k]]):int(self.co[c[k + 1]])]), peer(r), g))
                        if loop:                                        # ... and, playing peer r, its receive (same order)
                            ops.append(llr_recv_op(self.shares[r], k))
                w = T.batch(ops)
                works += w
                arrived[k] = w
            me.decode_chunk(me.t0, me.t1, llr_root, rnd)               # own range, while the sends drain
            if loop:
                # the virtual peers' part of the protocol, in the order real peers would run it: chunk k decoded when it has
                # arrived, its results sent at once -- with the root's matching receives in the same batch
                for k in range(n_rounds):
                    for w in arrived[k]:
                        w.wait()
                    for r in range(world):
                        if live(r, k):
                            sh, c = self.shares[r], self.chunk_cut[r]
                            sh.decode_chunk(c[k], c[k + 1], sh.llr, rnd)
                            ops = []
                            for s_op, r_op in zip(result_send_ops(sh, k), result_recv_ops(r, k)):
                                ops += [s_op, r_op]
                            works += T.batch(ops)
            else:
                for k in range(n_rounds):                              # results back, in the order the peers produce them
                    ops = []
                    for r in range(world):
                        if live(r, k):
                            ops += result_recv_ops(r, k)
                    works += T.batch(ops)
            if n_loc:
                pay_all[int(self.po[me.t0]):int(self.po[me.t1])] = me.pay[:int(self.po[me.t1] - self.po[me.t0])]
                ack_all[me.t0:me.t1] = me.ack[:n_loc]
                itm_all[me.t0:me.t1] = me.itm[:n_loc]
            for w in works:
                w.wait()
            return pay_all, ack_all, itm_all
        # ---- a peer ----
        c = self.chunk_cut[rank]
        recvs = []
        for k in range(len(c) - 1):
            recvs.append(T.batch([llr_recv_op(me, k)]) if c[k + 1] > c[k] else [])
        sends = []
        for k in range(len(c) - 1):
            a, b = c[k], c[k + 1]
            if b <= a:
                continue
            for w in recvs[k]:
                w.wait()
            me.decode_chunk(a, b, me.llr, rnd)
            sends += T.batch(result_send_ops(me, k))
        for w in sends:
            w.wait()
        return None, None, None
