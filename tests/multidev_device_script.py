"""Helper of test_gpu_tb_resident.py::test_device_resident_batches_sharded_over_logical_devices (run in a subprocess, so
that NRLDPC_HIP_DEVICES is read by a fresh library): DEVICE-memory transport-block batches -- LLRs, payloads, verdicts
on GPU 0, the calls only enqueue on the caller's stream -- through nrLDPC_hip_dlsch_encode / nrLDPC_hip_ulsch_decode,
with the caller's own soft buffers and with the library's (MEM_HARQ_LIBRARY: they stay on the GPU that decodes the
block); dumps all outputs to argv[1] (.npz)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import openairinterface5g_amd as hip  # noqa: E402
from test_gpu_tb_chain import make_tbs  # noqa: E402

hip.LDPCinit()
m = hip.ldpc
rng = np.random.default_rng(43)
res = {}
tbs = make_tbs() + make_tbs()[:6]
po, co, ho, segs = m.tb_layout(tbs)
pay_h = np.zeros(int(po[-1]) + 16, np.uint8)
for i, t in enumerate(tbs):
    pay_h[po[i]:po[i] + t["A"] // 8] = rng.integers(0, 256, t["A"] // 8, dtype=np.uint8)
payload = torch.from_numpy(pay_h).cuda()
coded = torch.zeros(int(co[-1]) + 16, dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
with torch.cuda.stream(side):                      # a caller stream that is not the default one
    m.dlsch_encode_device(tbs, payload, coded)
    side.synchronize()
    res["tx"] = coded.cpu().numpy()
    harq = torch.zeros(int(ho[-1]) + 16, dtype=torch.int16, device="cuda")
    ids = [0x9000 + i for i in range(len(tbs))]
    for mode in ("caller", "library"):
        rx = [dict(t) for t in tbs]
        for rnd in range(2):
            for t in rx:
                t["round"] = rnd
            sigma = 1.1 if rnd == 0 else 0.6
            g = torch.Generator(device="cuda").manual_seed(100 + rnd)
            llr = ((1.0 - 2.0 * coded.float()) * 8 + sigma * 8 * torch.randn(coded.numel(), device="cuda", generator=g)).round().clamp(-127, 127).to(torch.int16)
            pay_out = torch.full_like(payload, 0x77)
            ack = torch.full((len(tbs),), 9, dtype=torch.uint8, device="cuda")
            itm = torch.full((len(tbs),), -1, dtype=torch.int32, device="cuda")
            if mode == "caller":
                m.ulsch_decode_device(rx, llr, harq, pay_out, ack, itm)
            else:
                m.ulsch_decode_device(rx, llr, None, pay_out, ack, itm, harq_ids=ids)
            side.synchronize()
            res[f"{mode}{rnd}_pay"] = pay_out.cpu().numpy()
            res[f"{mode}{rnd}_ack"], res[f"{mode}{rnd}_itm"] = ack.cpu().numpy(), itm.cpu().numpy()
            res[f"{mode}{rnd}_llrLen"] = np.array([t.get("llrLen", 0) for t in rx])
            if mode == "caller":
                res[f"{mode}{rnd}_harq"] = harq.cpu().numpy()
            else:
                res[f"{mode}{rnd}_harq"] = np.concatenate([m.harq_read(ids[i], segs[i] * m.HARQ_STRIDE) for i in range(len(tbs))])
np.savez(sys.argv[1], **res)
print("ok")
