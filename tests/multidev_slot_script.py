"""Helper of test_gpu_tb_resident.py::test_the_64_block_slot_cut_eight_ways (run in a subprocess, so that NRLDPC_HIP_DEVICES
is read by a fresh library): BASELINE configs[4], the slot of 64 PUSCH transport blocks (273 PRB x 13 symbols, 64QAM, TBS
213 176 bit: 1664 code segments), through nrLDPC_hip_ulsch_decode in every buffer arrangement that is cut over the library's
GPUs, two HARQ rounds each (rv 0 at an SNR that loses blocks, then rv 2):
  host_lib    host LLRs / payloads, soft buffers kept by the library under one id per block; round 1 presents the blocks in
              REVERSE order, so every block is decoded by another device than in round 0 and its soft buffers migrate
  host_host   host LLRs and the caller's pageable host soft buffers (the reference caller's layout, nr_ulsch_decoding.c:139,320)
  host_dev    host LLRs, soft buffers in the caller's device memory (NRLDPC_HIP_MEM_HOST | NRLDPC_HIP_MEM_HARQ_DEVICE): on a
              part that is not the owner's they are a peer GPU's memory (never touched by the CPU)
  dev_lib     everything device-resident, library soft buffers, reverse order in round 1
  dev_dev     everything device-resident, the caller's soft buffers
  chunked     (one device is enough) pageable host LLRs -> the call comes in chunks; payload offsets that do NOT grow with the
              block index (a later piece's payload range must not overwrite an earlier piece's blocks)
Dumps per-block results keyed by block identity to argv[1] (.npz)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import openairinterface5g_amd as hip  # noqa: E402

hip.LDPCinit()
m = hip.ldpc
N_TB = 64
A = 213176
G = (12 * 13 - 6) * 273 * 6
S = m.HARQ_STRIDE
base = [dict(A=A, G=G, BG=1, Qm=6, Nl=1, rv=0, tbslbrm=0, round=0) for _ in range(N_TB)]
po, co, ho, segs = m.tb_layout(base)
C_ = segs[0]
g = torch.Generator(device="cuda").manual_seed(808)
payload = torch.randint(0, 256, (int(po[-1]) + 16,), dtype=torch.uint8, device="cuda", generator=g)
pays = [payload[po[i]:po[i] + A // 8].cpu().numpy() for i in range(N_TB)]


def transmit(rv, sigma, seed):
    tbs = [dict(t, rv=rv) for t in base]
    coded = torch.zeros(int(co[-1]) + 16, dtype=torch.uint8, device="cuda")
    m.dlsch_encode_device(tbs, payload, coded)
    gg = torch.Generator(device="cuda").manual_seed(seed)
    llr = ((1.0 - 2.0 * coded.float()) * 8 + sigma * 8 * torch.randn(coded.numel(), device="cuda", generator=gg)).round().clamp(-127, 127).to(torch.int16)
    torch.cuda.synchronize()
    return llr




def probe(llr):
    """ACKs of a first transmission (device-resident, scratch soft buffers)"""
    tbs = [dict(t) for t in base]
    pay_out = torch.zeros(int(po[-1]) + 16, dtype=torch.uint8, device="cuda")
    ack = torch.zeros(N_TB, dtype=torch.uint8, device="cuda")
    itm = torch.zeros(N_TB, dtype=torch.int32, device="cuda")
    scratch = torch.zeros(N_TB * C_ * S + 16, dtype=torch.int16, device="cuda")
    m.ulsch_decode_device(tbs, llr, scratch, pay_out, ack, itm)
    torch.cuda.synchronize()
    return int(ack.sum().item())


# round 0 at the noise level that loses some of the 64 blocks but not all (code rate 0.87: the cliff is narrow; the search is
# deterministic, so the single-device and the eight-way run pick the same level)
first = None
for sigma in (0.46, 0.44, 0.42, 0.40, 0.39, 0.38, 0.37, 0.36, 0.35, 0.34, 0.32):
    cand = transmit(0, sigma, 1)
    n_ack = probe(cand)
    if first is None or abs(n_ack - 32) < abs(first[1] - 32):
        first = (cand, n_ack, sigma)
    if 8 <= n_ack <= 56:
        break
print("round 0: sigma", first[2], "->", first[1], "of", N_TB, "blocks decode")
rounds = [(0, 0, first[0]), (1, 2, transmit(2, 0.30, 2))]
res = {"sent": np.concatenate(pays)}
ORDER = {0: list(range(N_TB)), 1: list(reversed(range(N_TB)))}


def keep(tag, rnd, order, outp, ack, itm, state):
    inv = np.argsort(order)                       # results back in block-identity order
    res[f"{tag}{rnd}_pay"] = np.concatenate([np.asarray(outp[j]) for j in inv])
    res[f"{tag}{rnd}_ack"] = np.asarray(ack)[inv].astype(np.uint8)
    res[f"{tag}{rnd}_itm"] = np.asarray(itm)[inv].astype(np.int32)
    res[f"{tag}{rnd}_llrLen"] = np.asarray(state)[inv]


# ---- host LLRs ------------------------------------------------------------------------------------------------------
state = {"host_lib": [0] * N_TB, "host_host": [0] * N_TB, "host_dev": [0] * N_TB}
harq_host = np.zeros((N_TB * C_, S), np.int16)
harq_dev = torch.zeros(N_TB * C_ * S + 16, dtype=torch.int16, device="cuda")
ids = [0x5100 + i for i in range(N_TB)]
for rnd, rv, llr in rounds:
    llr_h = llr.cpu().numpy()
    per_tb = [llr_h[co[i]:co[i] + G] for i in range(N_TB)]
    for tag in ("host_lib", "host_host", "host_dev"):
        order = ORDER[rnd] if tag == "host_lib" else ORDER[0]
        tbs = [dict(base[i], rv=rv, round=rnd, llrLen=state[tag][i]) for i in order]
        llrs = [per_tb[i] for i in order]
        if tag == "host_lib":
            outp, ack, itm = m.ulsch_decode_host(tbs, llrs, None, harq_ids=[ids[i] for i in order])
        elif tag == "host_host":
            outp, ack, itm = m.ulsch_decode_host(tbs, llrs, harq_host)
        else:
            outp, ack, itm = m.ulsch_decode_host(tbs, llrs, harq_dev)
        for j, i in enumerate(order):
            state[tag][i] = tbs[j]["llrLen"]
        keep(tag, rnd, order, outp, ack, itm, [t["llrLen"] for t in tbs])
    res[f"host_lib{rnd}_harq"] = np.stack([m.harq_read(ids[i], C_ * S).reshape(C_, S) for i in range(N_TB)])
    res[f"host_host{rnd}_harq"] = harq_host.reshape(N_TB, C_, S).copy()
    res[f"host_dev{rnd}_harq"] = harq_dev[:N_TB * C_ * S].cpu().numpy().reshape(N_TB, C_, S)
m.harq_release()

# ---- device-resident ------------------------------------------------------------------------------------------------
ids2 = [0x6100 + i for i in range(N_TB)]
harq_dd = torch.zeros(N_TB * C_ * S + 16, dtype=torch.int16, device="cuda")
state = {"dev_lib": [0] * N_TB, "dev_dev": [0] * N_TB}
for rnd, rv, llr in rounds:
    for tag in ("dev_lib", "dev_dev"):
        order = ORDER[rnd] if tag == "dev_lib" else ORDER[0]
        tbs = [dict(base[i], rv=rv, round=rnd, llrLen=state[tag][i]) for i in order]
        llr_o = torch.cat([llr[co[i]:co[i + 1]] for i in order] + [llr.new_zeros(16)])
        pay_out = torch.full((int(po[-1]) + 16,), 0x77, dtype=torch.uint8, device="cuda")
        ack = torch.full((N_TB,), 9, dtype=torch.uint8, device="cuda")
        itm = torch.full((N_TB,), -1, dtype=torch.int32, device="cuda")
        if tag == "dev_lib":
            m.ulsch_decode_device(tbs, llr_o, None, pay_out, ack, itm, harq_ids=[ids2[i] for i in order])
        else:
            m.ulsch_decode_device(tbs, llr_o, harq_dd, pay_out, ack, itm)
        torch.cuda.synchronize()
        pay_h = pay_out.cpu().numpy()
        outp = [pay_h[po[j]:po[j] + A // 8] for j in range(N_TB)]
        for j, i in enumerate(order):
            state[tag][i] = tbs[j]["llrLen"]
        keep(tag, rnd, order, outp, ack.cpu().numpy(), itm.cpu().numpy(), [t["llrLen"] for t in tbs])
    res[f"dev_lib{rnd}_harq"] = np.stack([m.harq_read(ids2[i], C_ * S).reshape(C_, S) for i in range(N_TB)])
    res[f"dev_dev{rnd}_harq"] = harq_dd[:N_TB * C_ * S].cpu().numpy().reshape(N_TB, C_, S)
m.harq_release()

# ---- chunked host call, payload offsets in a scrambled order ------------------------------------------------------------
rv, llr = 0, transmit(0, 0.2, 3)
llr_h = llr.cpu().numpy()
perm = np.random.default_rng(5).permutation(N_TB)
pay_off = [int(po[perm[i]]) for i in range(N_TB)]                 # block i's payload lies where block perm[i]'s would
tbs = [dict(base[i], rv=rv, round=0) for i in range(N_TB)]
outp, ack, itm = m.ulsch_decode_host(tbs, [llr_h[co[i]:co[i] + G] for i in range(N_TB)], None,
                                     harq_ids=[0x7100 + i for i in range(N_TB)], payload_off=pay_off)
keep("chunked", 0, list(range(N_TB)), outp, ack, itm, [t["llrLen"] for t in tbs])
m.harq_release()
np.savez(sys.argv[1], **res)
print("ok")
