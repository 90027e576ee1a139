"""Helper of test_gpu_tb_chain.py::test_abort_stops_the_siblings_of_a_failed_segment (run in a subprocess: NRLDPC_HIP_TB_ABORT
is read once per library load).  A slot of lost transport blocks next to clean ones on device buffers; prints one JSON
line with the verdicts, a digest of the soft buffers and the time of the call."""
import hashlib
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import openairinterface5g_amd as pkg  # noqa: E402

m = pkg.ldpc
pkg.LDPCinit()
torch.manual_seed(5)
MAXIT, n_lost, n_clean = 20, 60, 4
A = 213176
while m.nr_segmentation(A + 24, 1) is None:
    A += 8
G = (12 * 13 - 6) * 273 * 6                                # 26 segments of Zc = 384 per transport block
n = n_lost + n_clean
tbs = [dict(A=A, G=G, BG=1, Qm=6, Nl=1, rv=0, tbslbrm=0, round=0) for _ in range(n)]
po, co, ho, segs = m.tb_layout(tbs)
payload = torch.randint(0, 256, (int(po[-1]) + 16,), dtype=torch.uint8, device="cuda")
coded = torch.zeros(int(co[-1]) + 16, dtype=torch.uint8, device="cuda")
m.PreparedTbBatch(tbs, payload, coded).encode()
llr = ((1 - 2 * coded.to(torch.int16)) * 20).to(torch.int16)
lost = int(co[n_lost])
llr[:lost] = torch.randint(-30, 31, (lost,), dtype=torch.int16, device="cuda")   # nothing but noise: no segment converges
harq = torch.zeros(int(ho[-1]) + 16, dtype=torch.int16, device="cuda")
pay_out = torch.full_like(payload, 0xA5)
ack = torch.full((n,), 7, dtype=torch.uint8, device="cuda")
itm = torch.zeros(n, dtype=torch.int32, device="cuda")
dec = m.PreparedTbBatch(tbs, pay_out, llr, harq, ack, itm, numMaxIter=MAXIT)
best = 1e9
for rep in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dec.decode()
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
ok_clean = all(torch.equal(pay_out[int(po[i]):int(po[i]) + A // 8], payload[int(po[i]):int(po[i]) + A // 8]) for i in range(n_lost, n))
lost_zero = all(not bool(pay_out[int(po[i]):int(po[i]) + A // 8].any()) for i in range(n_lost))
print(json.dumps(dict(ms=best * 1e3, ack=ack.cpu().tolist(), itm=itm.cpu().tolist(), clean_payload_ok=bool(ok_clean),
                      lost_payload_zero=bool(lost_zero), harq=hashlib.sha256(harq.cpu().numpy().tobytes()).hexdigest())))
