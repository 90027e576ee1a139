"""-m gpu: the UL-SCH chain as the reference's caller needs it (NR_TRANSPORT/nr_ulsch_decoding.c:320 nr_ulsch_decoding(...,
short *ulsch_llr, ...): the slot's LLRs arrive in HOST memory, :168 harq_process->d[r]: the soft buffers persist per HARQ
process) -- host LLRs, page-locked or not, with the soft buffers resident on the GPU (the caller's device memory or the
library's own, keyed by an id) -- and the fused segment kernel against the four-launch path it replaces.  Every payload
byte, ACK, pass count, llrLen and soft value is compared with the oracle chain (tests/oracle_lib.py ulsch_decode)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from test_gpu_tb_chain import make_tbs, valid_tbs

pytestmark = pytest.mark.gpu


def _noisy(rng, f, sigma):
    return np.clip(np.round((1 - 2 * f.astype(np.float64)) * 8 + sigma * rng.standard_normal(f.size)), -200, 200).astype(np.int16)


@pytest.mark.parametrize("where,pinned", [("device", True), ("device", False), ("library", True), ("library", False)])
def test_host_llrs_with_device_resident_harq(hip, where, pinned):
    """Two HARQ rounds (rv 0 at a noise level where several blocks fail, then rv 2) and a fresh first transmission on the
    same buffers: LLRs, payloads and verdicts in host memory (page-locked: pulled by the segments' workgroups in place;
    pageable: staged copy), soft buffers never leaving the GPU.  Soft values are read back after every call and compared
    value for value with the oracle chain's."""
    import torch
    m = hip.ldpc
    rng = np.random.default_rng(11)
    tbs = [t for t in make_tbs() if t["rv"] == 0 and t["tbslbrm"] == 0][:7] + [dict(make_tbs()[9], rv=0)]   # + a limited-buffer block
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
    segs = [O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])["C"] for t in tbs]
    S = m.HARQ_STRIDE
    ids = [0x5000 + 7 * i for i in range(len(tbs))]
    harq_dev = torch.zeros(sum(segs) * S, dtype=torch.int16, device="cuda") if where == "device" else None
    harq_ref = [[np.zeros(S, np.int16) for _ in range(c)] for c in segs]
    state_ref = [0] * len(tbs)
    m.harq_release()

    def soft(i, r):
        if where == "device":
            row = sum(segs[:i]) + r
            return harq_dev[row * S:(row + 1) * S].cpu().numpy()
        return m.harq_read(ids[i], S, r * S)

    acks = []
    for step, (rnd, rv, sigma) in enumerate(((0, 0, 9.0), (1, 2, 5.0), (0, 0, 1.0))):
        if rnd == 0:
            state_ref = [0] * len(tbs)
        llrs = []
        for t, p in zip(tbs, pays):
            t["rv"], t["round"] = rv, rnd
            if rnd == 0:
                t["llrLen"] = 0
            llrs.append(_noisy(rng, O.dlsch_encode(t, p), sigma))
        out, ack, itm = m.ulsch_decode_host(tbs, llrs, harq_dev, numMaxIter=8, pinned=pinned,
                                            harq_ids=ids if where == "library" else None)
        for i, t in enumerate(tbs):
            p_ref, ack_ref, its, state_ref[i] = O.ulsch_decode(t, llrs[i], harq_ref[i], 8, rnd, state_ref[i])
            assert bool(ack[i]) == ack_ref and itm[i] == max(its), (step, t, its, int(itm[i]))
            assert t["llrLen"] == state_ref[i]
            if ack_ref:
                assert np.array_equal(out[i], p_ref) and np.array_equal(out[i], pays[i])
            else:
                assert not out[i].any()                              # a lost block delivers zeros
            sg = O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])
            N = (66 if t["BG"] == 1 else 50) * sg["Z"]
            ncb = N if not t["tbslbrm"] else min(N, 3 * t["tbslbrm"] // (2 * sg["C"]))    # nr_rate_matching.c:445-450
            for r in range(segs[i]):
                got, ref = soft(i, r), harq_ref[i][r]
                assert np.array_equal(got[:ncb], ref[:ncb]), (step, i, r)
        acks.append(ack.copy())
    assert not acks[0].all() and acks[1].sum() > acks[0].sum() and acks[2].all()      # combining recovers blocks
    if where == "library":
        assert m.harq_release(ids[0]) == 0 and m.harq_release(ids[0]) != 0        # a second release: unknown id
        m.harq_release()


def _random_tbs(rng, n):
    tbs = []
    for _ in range(n):
        bits = int(np.exp(rng.uniform(np.log(24), np.log(90000))))
        BG = 2 if bits <= 292 else (int(rng.integers(1, 3)) if bits <= 30000 else 1)
        Qm, Nl = int(rng.choice([2, 4, 6, 8])), int(rng.integers(1, 3))
        A = valid_tbs(bits, BG)
        rate = rng.uniform(0.2, 0.93)
        G = max(1, int(A / rate) // (Qm * Nl)) * Qm * Nl
        C = O.segmentation(None, O.len_with_crc(1, A), BG)["C"]
        G = max(G, C * Qm * Nl * 4)
        tbs.append(dict(A=A, G=G, BG=BG, Qm=Qm, Nl=Nl, rv=0, tbslbrm=int(rng.choice([0, 0, 3 * A]))))
    return tbs


def test_fused_segment_kernel_against_the_four_launch_path(hip):
    """The same heterogeneous calls -- 60 random transport blocks (1 to 11 segments, both base graphs, lifting sizes the
    fast decoder does not take among them, clean, marginal and hopeless noise levels so that blocks fail and siblings
    abort), two HARQ rounds -- through the fused segment kernel (default) and through de-matching / decoder / reassembly /
    verdict as separate launches (NRLDPC_HIP_TB_FUSED=0): payload bytes (zeros for lost blocks), ACKs, pass counts,
    llrLen and every soft value identical; the fused run is also held against the oracle chain."""
    m = hip.ldpc
    rng = np.random.default_rng(20260928)
    tbs0 = _random_tbs(rng, 60)
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs0]
    segs = [O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])["C"] for t in tbs0]
    sig = [float(rng.choice([1.0, 6.5, 40.0])) for _ in tbs0]
    rounds = []
    for rnd, rv in ((0, 0), (1, 3)):
        llrs = []
        for t, p, sg in zip(tbs0, pays, sig):
            llrs.append(_noisy(rng, O.dlsch_encode(dict(t, rv=rv), p), sg if rnd == 0 else min(sg, 7.0)))
        rounds.append((rnd, rv, llrs))
    res = {}
    prev = os.environ.get("NRLDPC_HIP_TB_FUSED")
    try:
        for mode in ("1", "0"):
            os.environ["NRLDPC_HIP_TB_FUSED"] = mode
            tbs = [dict(t) for t in tbs0]
            harq = np.zeros((sum(segs), m.HARQ_STRIDE), np.int16)
            out_all = []
            for rnd, rv, llrs in rounds:
                for t in tbs:
                    t["rv"], t["round"] = rv, rnd
                out, ack, itm = m.ulsch_decode_host(tbs, llrs, harq, numMaxIter=6)
                out_all.append((np.concatenate(out), ack.copy(), itm.copy(), np.array([t["llrLen"] for t in tbs]), harq.copy()))
            res[mode] = out_all
    finally:
        if prev is None:
            os.environ.pop("NRLDPC_HIP_TB_FUSED", None)
        else:
            os.environ["NRLDPC_HIP_TB_FUSED"] = prev
    for rnd in range(2):
        for a, b, what in zip(res["1"][rnd], res["0"][rnd], ("payload", "ack", "iter_max", "llrLen", "soft buffers")):
            assert np.array_equal(a, b), (rnd, what)
    ack0, ack1 = res["1"][0][1], res["1"][1][1]
    assert 0 < ack0.sum() < len(tbs0) and ack1.sum() > ack0.sum()
    # the fused run against the oracle chain
    harq_ref = [[np.zeros(m.HARQ_STRIDE, np.int16) for _ in range(c)] for c in segs]
    st = [0] * len(tbs0)
    for (rnd, rv, llrs), got in zip(rounds, res["1"]):
        off = 0
        for i, t in enumerate(tbs0):
            p_ref, ack_ref, its, st[i] = O.ulsch_decode(dict(t, rv=rv), llrs[i], harq_ref[i], 6, rnd, st[i], vec=True)
            nb = t["A"] // 8
            assert bool(got[1][i]) == ack_ref and got[2][i] == min(max(its), 7) and got[3][i] == st[i], (rnd, i, its)
            assert np.array_equal(got[0][off:off + nb], p_ref if ack_ref else np.zeros(nb, np.uint8)), (rnd, i)
            off += nb


def test_first_transmissions_never_upload_host_soft_buffers(hip):
    """Host-resident soft buffers (the legacy layout): a call whose blocks are all first transmissions uploads nothing of
    them -- they are cleared on the device -- and brings back max(Ncb, positions the decoder reads) values per segment: the
    Ncb soft values it produced and, with limited-buffer rate matching (Ncb < N), ZEROS in [Ncb, np) -- the reference memsets
    Ncb entries only (nr_rate_matching.c:554-555) and its decoder input reads what its calloc'ed buffer holds behind them,
    which is zero as well.  What lies behind the circular buffer's N positions is never touched."""
    m = hip.ldpc
    rng = np.random.default_rng(5)
    tbs = [dict(t, rv=0) for t in make_tbs() if t["tbslbrm"]] + make_tbs()[:3]
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
    segs = [O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])["C"] for t in tbs]
    S = m.HARQ_STRIDE
    harq = rng.integers(-30000, 30000, (sum(segs), S)).astype(np.int16)
    before = harq.copy()
    llrs = [_noisy(rng, O.dlsch_encode(t, p), 2.0) for t, p in zip(tbs, pays)]
    for t in tbs:
        t["round"] = 0
    out, ack, itm = m.ulsch_decode_host(tbs, llrs, harq, numMaxIter=8)
    assert ack.all()
    row = 0
    for i, t in enumerate(tbs):
        href = [before[row + r].copy() for r in range(segs[i])]
        O.ulsch_decode(t, llrs[i], href, 8, 0, 0)
        s = O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])
        N = (66 if t["BG"] == 1 else 50) * s["Z"]
        ncb = N if not t["tbslbrm"] else min(N, 3 * t["tbslbrm"] // (2 * s["C"]))
        for r in range(segs[i]):
            assert np.array_equal(harq[row + r, :ncb], href[r][:ncb]), (i, r)
            tail = harq[row + r, ncb:N]                                                # [Ncb, N): zeros up to the decoder's reach, else untouched
            nz = int(np.argmax(tail != 0)) if (tail != 0).any() else tail.size
            assert not tail[:nz].any() and np.array_equal(tail[nz:], before[row + r, ncb + nz:N]), (i, r)
            assert np.array_equal(harq[row + r, N:], before[row + r, N:]), (i, r)       # behind the circular buffer: never touched
        row += segs[i]


def test_device_resident_batches_sharded_over_logical_devices(hip, tmp_path):
    """NRLDPC_HIP_DEVICES=0,0,0 and DEVICE memory: a slot whose LLRs live on GPU 0 is cut into whole transport blocks per
    device (tb_partition), the peers' ranges travel GPU to GPU into the peers' staging buffers (hipMemcpyAsync between
    devices = xGMI on a real node; here three device contexts on the one GPU the box has), results come back into the
    owner's arrays, everything ordered on the caller's stream by events -- the C-level form of SURVEY 8e for a caller that
    holds its data on one GPU.  Soft buffers: the caller's (they travel both ways) and the library's (they stay where
    the block is decoded).  Every output equals the single-device run's; library-kept and caller-kept soft values agree."""
    import subprocess
    import sys
    from pathlib import Path
    script = Path(__file__).resolve().parent / "multidev_device_script.py"
    outs = []
    for devs in (None, "0,0,0"):
        env = dict(os.environ)
        env.pop("NRLDPC_HIP_DEVICES", None)
        if devs:
            env["NRLDPC_HIP_DEVICES"] = devs
            env["NRLDPC_HIP_TEST_STAGE_HARQ"] = "1"     # the aliased contexts treat the owner's soft buffers as a peer GPU's
        f = tmp_path / f"out_{devs or 'single'}.npz"
        r = subprocess.run([sys.executable, str(script), str(f)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(np.load(f))
    a, b = outs
    assert sorted(a.files) == sorted(b.files) and len(a.files) >= 21
    for k in a.files:
        if k.endswith("_harq") and k.startswith("library"):
            continue                                    # fresh library buffers are not initialised behind what a block uses
        assert np.array_equal(a[k], b[k]), k
    assert not a["caller0_ack"].all() and a["caller1_ack"].sum() > a["caller0_ack"].sum()
    for o in outs:
        for rnd in range(2):
            for what in ("pay", "ack", "itm", "llrLen"):
                assert np.array_equal(o[f"caller{rnd}_{what}"], o[f"library{rnd}_{what}"]), (rnd, what)


def test_the_64_block_slot_cut_eight_ways(hip, tmp_path):
    """SURVEY 8e at the width the node has: NRLDPC_HIP_DEVICES=0,0,0,0,0,0,0,0 (eight device contexts that alias the one GPU
    of the box) on the full slot of BASELINE configs[4] -- 64 transport blocks, 8 per device, the peers' ranges staged seven
    times -- against the single-device run of the same script: host and device-resident buffers, soft buffers kept by the
    library (migrating between devices from round 0 to round 1), by the caller on the host, and by the caller in device memory
    under host LLRs; plus the chunked host call with scrambled payload offsets (tests/multidev_slot_script.py)."""
    import subprocess
    import sys
    from pathlib import Path
    script = Path(__file__).resolve().parent / "multidev_slot_script.py"
    outs = []
    for devs in (None, "0,0,0,0,0,0,0,0"):
        env = dict(os.environ)
        env.pop("NRLDPC_HIP_DEVICES", None)
        if devs:
            env["NRLDPC_HIP_DEVICES"] = devs
            env["NRLDPC_HIP_TEST_STAGE_HARQ"] = "1"     # the aliased contexts treat the owner's soft buffers as a peer GPU's
        f = tmp_path / f"slot_{'8' if devs else '1'}.npz"
        r = subprocess.run([sys.executable, str(script), str(f)], capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs.append(np.load(f))
    a, b = outs
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    A8 = 213176 // 8
    for o in outs:
        sent = o["sent"].reshape(64, A8)
        for tag in ("host_lib", "host_host", "host_dev", "dev_lib", "dev_dev"):
            ack0, ack1 = o[f"{tag}0_ack"].astype(bool), o[f"{tag}1_ack"].astype(bool)
            assert 0 < ack0.sum() < 64 and ack1.all(), (tag, int(ack0.sum()), int(ack1.sum()))   # round 0 loses blocks, combining recovers all
            assert np.array_equal(o[f"{tag}1_pay"].reshape(64, A8), sent), tag
            assert np.array_equal(o[f"{tag}0_pay"].reshape(64, A8)[ack0], sent[ack0]), tag
            assert not o[f"{tag}0_pay"].reshape(64, A8)[~ack0].any(), tag                        # a lost block delivers zeros
            for what in ("ack", "itm", "pay", "llrLen", "harq"):                                  # every arrangement: the same numbers
                for rnd in (0, 1):
                    assert np.array_equal(o[f"{tag}{rnd}_{what}"], o[f"host_host{rnd}_{what}"]), (tag, rnd, what)
        assert o["chunked0_ack"].all() and np.array_equal(o["chunked0_pay"].reshape(64, A8), sent)


_KEEP_ALIVE = []


def test_residency_flags_and_mixed_call_errors(hip):
    """What the new entry points refuse, and the page-locking helpers a C caller without the HIP runtime uses: a soft-buffer
    pointer that is not device memory under NRLDPC_HIP_MEM_HARQ_DEVICE, both residency flags at once, unknown flag bits,
    encode with residency flags, releasing / reading an id the library does not hold, a read outside the buffers; a mixed
    code-block call in host memory or with two stop modes; nrLDPC_hip_host_register / _unregister on an ordinary array, which
    the GPU then reads in place (the call's results equal the pageable run's)."""
    import ctypes as C
    import torch
    m = hip.ldpc
    L = m._tb_lib()
    rng = np.random.default_rng(3)
    tbs = make_tbs()[:3]
    pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
    llrs = [_noisy(rng, O.dlsch_encode(t, p), 2.0) for t, p in zip(tbs, pays)]
    segs = [O.segmentation(None, O.len_with_crc(1, t["A"]), t["BG"])["C"] for t in tbs]
    po, co, ho, _ = m.tb_layout(tbs)
    pay = np.zeros(int(po[-1]) + 16, np.uint8)
    llr = np.zeros(int(co[-1]) + 16, np.int16)
    for i, x in enumerate(llrs):
        llr[co[i]:co[i] + tbs[i]["G"]] = x
    ack, itm = np.zeros(3, np.uint8), np.zeros(3, np.int32)
    harq_h = np.zeros(sum(segs) * m.HARQ_STRIDE, np.int16)
    arr = m._tb_array([dict(t, round=0) for t in tbs], po, co, ho, 8)

    def call(mem, harq_ptr, fn=L.nrLDPC_hip_ulsch_decode):
        b = m.nrLDPC_hip_tb_batch_t(n_tb=3, tb=arr, payload=pay.ctypes.data, coded=llr.ctypes.data, harq=harq_ptr, harq_stride=m.HARQ_STRIDE,
                                    ack=ack.ctypes.data, iter_max=itm.ctypes.data, mem=mem, stream=None)
        return fn(C.byref(b))

    assert call(m.MEM_HOST | m.MEM_HARQ_DEVICE, harq_h.ctypes.data) != 0 and b"harq" in L.nrLDPC_hip_last_error().lower()
    assert call(m.MEM_HOST | m.MEM_HARQ_DEVICE | m.MEM_HARQ_LIBRARY, None) != 0
    assert call(m.MEM_HOST | 8, harq_h.ctypes.data) != 0
    assert call(m.MEM_HOST, None) != 0                                          # host soft buffers need a pointer
    assert call(m.MEM_HOST | m.MEM_HARQ_LIBRARY, None, L.nrLDPC_hip_dlsch_encode) != 0
    m.harq_release()
    assert L.nrLDPC_hip_harq_release(12345) != 0 and L.nrLDPC_hip_harq_read(12345, harq_h.ctypes.data, 0, 8) != 0
    # pageable run = the reference result; then the same arrays page-locked in place
    assert call(m.MEM_HOST, harq_h.ctypes.data) == 0 and ack.all()
    ref = (pay.copy(), itm.copy(), harq_h.copy())
    assert call(m.MEM_HOST | m.MEM_HARQ_LIBRARY, None) == 0 and ack.all()       # ids = the harq offsets of the layout
    got = np.concatenate([m.harq_read(int(ho[i]), segs[i] * m.HARQ_STRIDE) for i in range(3)])
    assert L.nrLDPC_hip_harq_read(int(ho[0]), harq_h.ctypes.data, segs[0] * m.HARQ_STRIDE - 4, 8) != 0     # past the end
    for i in range(3):      # what the blocks use of their rows equals the host run's
        s = O.segmentation(None, O.len_with_crc(1, tbs[i]["A"]), tbs[i]["BG"])
        N = (66 if tbs[i]["BG"] == 1 else 50) * s["Z"]
        for r in range(segs[i]):
            a = (sum(segs[:i]) + r) * m.HARQ_STRIDE
            assert np.array_equal(got[a:a + N], ref[2][a:a + N]), (i, r)
    _KEEP_ALIVE.append(llr)   # (its pages were handed to the driver once: the allocator does not get the range back)
    assert L.nrLDPC_hip_host_register(llr.ctypes.data, llr.nbytes) == 0
    try:
        pay[:] = 0
        assert call(m.MEM_HOST | m.MEM_HARQ_LIBRARY, None) == 0 and ack.all() and np.array_equal(pay, ref[0]) and np.array_equal(itm, ref[1])
    finally:
        assert L.nrLDPC_hip_host_unregister(llr.ctypes.data) == 0
    m.harq_release()
    # an array of which only the first pages are page-locked (the library asks about both ends of the range a call reads):
    # refused with a message -- the runtime would reject the copy out of it anyway -- and usable again once unregistered
    assert L.nrLDPC_hip_host_register(llr.ctypes.data, 8192) == 0
    try:
        assert call(m.MEM_HOST | m.MEM_HARQ_LIBRARY, None) != 0 and b"in part" in L.nrLDPC_hip_last_error()
    finally:
        assert L.nrLDPC_hip_host_unregister(llr.ctypes.data) == 0
    pay[:] = 0
    assert call(m.MEM_HOST | m.MEM_HARQ_LIBRARY, None) == 0 and ack.all() and np.array_equal(pay, ref[0]) and np.array_equal(itm, ref[1])
    m.harq_release()
    # mixed code-block calls
    blk = [dict(BG=1, Z=64, R=13, llr=torch.zeros(68 * 64, dtype=torch.int8, device="cuda"),
                out=torch.zeros(m.out_bytes(1, 64, 13), dtype=torch.uint8, device="cuda")) for _ in range(2)]
    n_it = torch.zeros(2, dtype=torch.int32, device="cuda")
    jobs = m.PreparedDecJobs(blk, n_it)
    assert L.LDPCdecoder_jobs(jobs.arr, 2, n_it.data_ptr(), m.MEM_HOST, None) != 0
    jobs.arr[1].params.check_crc = m.device_crc_pointer()      # two stop modes in one call
    jobs.arr[1].params.E, jobs.arr[1].params.crc_type = 22 * 64, 1
    assert L.LDPCdecoder_jobs(jobs.arr, 2, n_it.data_ptr(), m.MEM_DEVICE, None) != 0
    assert L.LDPCdecoder_jobs(jobs.arr, 0, n_it.data_ptr(), m.MEM_DEVICE, None) == 0
