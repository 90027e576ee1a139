"""Helper of test_gpu_tb_chain.py::test_host_batches_sharded_over_logical_devices (run in a subprocess, so that
NRLDPC_HIP_DEVICES is read by a fresh library): host-buffer batches through every entry point that shards over devices;
dumps all outputs to argv[1] (.npz)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import openairinterface5g_amd as hip  # noqa: E402
import oracle_lib as O  # noqa: E402
from test_gpu_tb_chain import make_tbs  # noqa: E402

hip.LDPCinit()
rng = np.random.default_rng(42)
res = {}
# 1. code-block batch, host buffers (contiguous block ranges per device)
BG, Z, R = 1, 96, 13
K = 22 * Z
infos = rng.integers(0, 256, (300, K // 8), dtype=np.uint8)
coded = hip.encode_batch_host(BG, Z, infos)
llr = np.stack([O.awgn_llr(rng, coded[i], Z, float(rng.choice([-2.0, 0.5, 2.0]))) for i in range(300)])
it, out = hip.decode_batch_host(BG, Z, R, llr, numMaxIter=8)
res["cb_iter"], res["cb_out"] = it, out
# 2. transport-block chain, host buffers, two HARQ rounds (whole TBs per device)
tbs = make_tbs() + make_tbs()[:5]
pays = [rng.integers(0, 256, t["A"] // 8, dtype=np.uint8) for t in tbs]
f = hip.ldpc.dlsch_encode_host(tbs, pays)
res["tx"] = np.concatenate(f)
segs = [hip.ldpc.nr_segmentation(t["A"] + (24 if t["A"] > 3824 else 16), t["BG"])["C"] for t in tbs]
harq = np.zeros((sum(segs), hip.ldpc.HARQ_STRIDE), np.int16)
rx = [dict(t) for t in tbs]
# 3. the same two rounds once more with the soft buffers scattered: reverse TB order, one stride of somebody else's data
#    (a sentinel pattern) between them -- results as in 2., and not one sentinel value may change (a device's
#    share of a sharded batch must not write back soft values it does not own)
S = hip.ldpc.HARQ_STRIDE
off3, pos = [0] * len(tbs), 0
for i in reversed(range(len(tbs))):
    pos += S
    off3[i] = pos
    pos += segs[i] * S
harq3 = np.full(pos + S, 0x5A5A, np.int16)
own = np.zeros(harq3.size, bool)
for i in range(len(tbs)):
    harq3[off3[i]:off3[i] + segs[i] * S] = 0
    own[off3[i]:off3[i] + segs[i] * S] = True
rx3 = [dict(t) for t in tbs]
for rnd in range(2):
    llrs = []
    for t, c in zip(rx, f):
        t["round"] = rnd
        sigma = 1.1 if rnd == 0 else 0.6
        y = (1.0 - 2.0 * c.astype(np.float64) + sigma * rng.standard_normal(c.size)) * 8.0
        llrs.append(np.clip(np.rint(y), -127, 127).astype(np.int16))
    outp, ack, itm = hip.ldpc.ulsch_decode_host(rx, llrs, harq, numMaxIter=8)
    res[f"rx{rnd}_pay"] = np.concatenate(outp)
    res[f"rx{rnd}_ack"], res[f"rx{rnd}_itm"] = np.asarray(ack), np.asarray(itm)
    res[f"rx{rnd}_harq"] = harq.copy()
    res[f"rx{rnd}_llrLen"] = np.array([t.get("llrLen", 0) for t in rx])
    for t in rx3:
        t["round"] = rnd
    outp3, ack3, itm3 = hip.ldpc.ulsch_decode_host(rx3, llrs, harq3, numMaxIter=8, harq_off=off3)
    same = np.array_equal(np.concatenate(outp3), res[f"rx{rnd}_pay"]) and np.array_equal(ack3, ack) and np.array_equal(itm3, itm)
    same &= all(np.array_equal(harq3[off3[i]:off3[i] + segs[i] * S].reshape(segs[i], S), harq[sum(segs[:i]):sum(segs[:i + 1])])
                for i in range(len(tbs)))
    res[f"rx{rnd}_scattered_equal"] = np.array([same, bool((harq3[~own] == 0x5A5A).all())])
np.savez(sys.argv[1], **res)
print("ok")
