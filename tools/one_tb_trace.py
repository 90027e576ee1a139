"""One transport block (26 segments) through the UL-SCH and DL-SCH calls, 30 times, every call awaited: the workload for a
rocprofv3 --kernel-trace of the single-block latency (tools/gpu_one_tb_trace.sh)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import openairinterface5g_amd as pkg
m = pkg.ldpc
pkg.LDPCinit()
A = 213176
while m.nr_segmentation(A + 24, 1) is None:
    A += 8
G = (12 * 13 - 6) * 273 * 6
tbs = [dict(A=A, G=G, BG=1, Qm=6, Nl=1, rv=0, tbslbrm=0, round=0)]
po, co, ho, segs = m.tb_layout(tbs)
payload = torch.randint(0, 256, (int(po[-1]) + 16,), dtype=torch.uint8, device="cuda")
coded = torch.zeros(int(co[-1]) + 16, dtype=torch.uint8, device="cuda")
m.PreparedTbBatch(tbs, payload, coded).encode()
llr = ((1.0 - 2.0 * coded.float()) * 10 + 1.8 * torch.randn(coded.numel(), device="cuda")).round().clamp(-127, 127).to(torch.int16)
harq = torch.zeros(int(ho[-1]) + 16, dtype=torch.int16, device="cuda")
pay_out = torch.zeros_like(payload); ack = torch.zeros(1, dtype=torch.uint8, device="cuda"); itm = torch.zeros(1, dtype=torch.int32, device="cuda")
dec = m.PreparedTbBatch(tbs, pay_out, llr, harq, ack, itm)
enc = m.PreparedTbBatch(tbs, payload, coded)
for _ in range(30):
    dec.decode(); torch.cuda.synchronize()
    enc.encode(); torch.cuda.synchronize()
