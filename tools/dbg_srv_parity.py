"""debug: LDPCdecoder (server path) vs LDPCdecoder_batch (launch path) vs oracle on the same inputs"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import openairinterface5g_amd as hip
import oracle_lib as O
hip.LDPCinit()
rng = np.random.default_rng(3)
for (BG, Z, R, snr) in [(1, 384, 13, 1.0), (1, 384, 13, 3.0), (1, 176, 13, 1.0), (2, 208, 15, 0.0), (1, 384, 89, 6.0), (2, 6, 15, 2.0)]:
    K = (22 if BG == 1 else 10) * Z
    bad = []
    for t in range(40):
        info = rng.integers(0, 256, (K + 7) // 8, dtype=np.uint8)
        coded = O.encode(BG, Z, info)
        llr = O.awgn_llr(rng, coded, Z, snr)[: hip.ldpc.NCOLS[(BG, R)] * Z]
        p = hip.make_dec_params(BG, Z, R, 8)
        n1, o1 = hip.LDPCdecoder(p, llr)
        n2, o2 = hip.LDPCdecoder(p, llr)
        nb, ob = hip.decode_batch_host(BG, Z, R, llr[None, :], numMaxIter=8)
        no, oo = O.decode(BG, Z, R, llr, 8)
        if not (n1 == n2 == nb[0] == no and np.array_equal(o1, oo) and np.array_equal(o2, oo)):
            bad.append((t, n1, n2, int(nb[0]), no, int((o1 != oo).sum()), int((o2 != oo).sum())))
    print((BG, Z, R, snr), "mismatches (trial, srv1, srv2, batch, oracle, diff bytes 1, 2):", bad[:12], len(bad))
print(hip.ldpc.server_stats())
