#!/usr/bin/env python3
"""ldpctest-shaped harness for libldpc_hip.so (test infrastructure).

Mirrors the flow and the result lines of the reference's own LDPC bench
(openair1/PHY/CODING/TESTBENCH/ldpctest.c:99-400 test_ldpc, :401-588 main): same option letters, the same
block-length -> (BG, Kb, Zc) rule (:177-246), encoder through LDPCencoder in groups of 8 segments (:279-284),
BPSK + AWGN with OAI's own random generator and `quantize` (restated in oracle/, seeded like OAI_RNGSEED),
decoder through LDPCdecoder one segment per call with check_crc = NULL (:320-332), block error = memcmp of the first
block_length/8 bytes (:342), SNR sweep that stops at the first error-free point (:517-575).

  python tests/ldpctest_hip.py -l 8448 -s 10 -n 100         # the reference CI's acceptance run: expects BLER 0.000000
  python tests/ldpctest_hip.py -l 8448 -s 10 -n 20 --oracle   # same seeds through the CPU oracle instead (cross-check)
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as O  # noqa: E402  (OAI RNG / quantiser restatement; decoder only with --oracle)

LIFT = O.LIFT_SIZES
CODE_RATE_VEC = [15, 13, 25, 12, 23, 34, 56, 89]          # ldpctest.c:122


def run(args, out=sys.stdout):
    block_length, nom, den = args.l, args.r, args.d
    if block_length > 3840:                                 # ldpctest.c:177-194
        BG, Kb, nrows = 1, 22, 46
    else:
        BG, nrows = 2, 42
        Kb = 10 if block_length > 640 else 9 if block_length > 560 else 8 if block_length > 192 else 6
    if (nom, den) == (1, 5) and BG == 2:                    # ldpctest.c:197-229
        R_ind = 0
    elif (nom, den) == (1, 3):
        R_ind = 1
    elif (nom, den) == (2, 3):
        R_ind = 4
    elif (nom, den) == (22, 25) and BG == 1:
        R_ind = 7
    else:
        raise SystemExit(f"Not supported: nom_rate: {nom}, denom_rate: {den}")
    Zc = next(z for z in LIFT if z >= block_length / Kb)    # ldpctest.c:238-246
    kbf = 22 if BG == 1 else 10
    K = kbf * Zc
    R = CODE_RATE_VEC[R_ind]
    no_punctured_columns = int(((nrows - 2) * Zc + block_length - block_length * (den / nom)) / Zc)
    removed_bit = (nrows - no_punctured_columns - 2) * Zc + block_length - int(block_length / (nom / den))
    n_coded = (Kb + nrows - no_punctured_columns) * Zc - removed_bit - 2 * Zc   # transmitted positions 2Zc .. To
    print(f"ldpc_test: codeword_length {args.S * block_length}, n_segments {args.S}, block_length {block_length}, "
          f"BG {BG}, Zc {Zc}, Kb {Kb}", file=out)
    rate = 3 if BG == 1 else 5

    def oracle_encode(x):
        """ldpc_encoder.c with K = block_length: missing bits are zeros, the code word is cut to rate * block_length bytes
        (:82-92, :248-251): c[2Zc..block_length) || d[0 .. rate*block_length - block_length + 2Zc)"""
        if block_length == K:
            return O.encode(BG, Zc, x, Kb)
        bits = np.concatenate([np.unpackbits(np.asarray(x, np.uint8))[:block_length], np.zeros(K - block_length, np.uint8)])
        full = O.encode(BG, Zc, np.packbits(bits), Kb)
        n_info = block_length - 2 * Zc
        return np.concatenate([full[:n_info], full[K - 2 * Zc:K - 2 * Zc + rate * block_length - n_info]])

    if args.oracle:
        decode_one = lambda llr: O.decode(BG, Zc, R, llr, args.i)
        encode_many = lambda infos: [oracle_encode(x) for x in infos]
    else:
        import openairinterface5g_amd as pkg
        pkg.LDPCinit()
        p = pkg.make_dec_params(BG, Zc, R, args.i, E=block_length)
        # the timed region is the C call itself, as in ldpctest.c:329-334 (start_meas / stop_meas around LDPCdecoder)
        raw_call, _keep = pkg.ldpc.raw_decoder_call(p)
        out_buf = np.zeros(pkg.ldpc.out_bytes(BG, Zc, R, p.outMode) + 64, np.uint8)
        out_addr = out_buf.ctypes.data

        def decode_one(llr):
            llr = np.ascontiguousarray(llr, dtype=np.int8)
            addr = llr.ctypes.data
            t0 = time.perf_counter()
            n = raw_call(addr, out_addr)
            decode_one.seconds = time.perf_counter() - t0
            return n, out_buf.copy()
        def encode_many(infos):
            outs = [None] * len(infos)
            for macro in range((len(infos) + 7) // 8):
                part = pkg.LDPCencoder(infos, BG, Zc, Kb, n_segments=len(infos), macro_num=macro,
                                       block_length=None if block_length == K else block_length)
                for j in range(8 * macro, min(len(infos), 8 * macro + 8)):
                    outs[j] = part[j]
            return outs
    rng = O.OaiRng(args.seed)
    data_rng = np.random.default_rng(args.seed)
    ncols = O.NCOLS[(BG, R)]
    results = []
    snr = args.s
    while snr < args.s + 20.0:
        snr_lin = 10 ** (snr / 10.0) * nom / den              # ldpctest.c:519-522
        print(f"Linear SNR: {snr_lin:f}", file=out)
        sigma = 1.0 / np.sqrt(2 * snr_lin)
        errors = bit_errors = 0
        iters = []
        t_dec = 0.0
        for _ in range(args.n):
            infos = [data_rng.integers(0, 256, (block_length + 7) // 8, dtype=np.uint8) for _ in range(args.S)]
            coded = encode_many(infos)
            if args.n == 1 and not args.oracle:             # ldpctest.c:286-292: one trial = encoder cross-check (orig vs optim)
                for j, x in enumerate(infos):
                    ref = oracle_encode(x)
                    if not np.array_equal(ref, coded[j][:ref.size]):
                        pos = int(np.flatnonzero(ref != coded[j][:ref.size])[0])
                        print(f"differ in seg {j} pos {pos} ({ref[pos]},{coded[j][pos]})", file=out)
                        return results
            for j in range(args.S):
                # ldpctest.c:295-305: decoder position i (2Zc <= i < To) takes code-word byte i - 2Zc -- also when
                # block_length < Kb*Zc, where the reference does NOT re-insert the shortened columns
                llr = rng.ldpctest_channel(coded[j][:n_coded], Zc, sigma, args.q)
                llr = np.concatenate([llr, np.zeros(max(0, ncols * Zc - llr.size), np.int8)])[:ncols * Zc]
                t0 = time.perf_counter()
                n_iter, est = decode_one(llr)
                t_dec += getattr(decode_one, "seconds", None) or (time.perf_counter() - t0)
                iters.append(n_iter)
                if not np.array_equal(est[:block_length // 8], infos[j][:block_length // 8]):
                    errors += 1
                bit_errors += int(np.unpackbits(est[:block_length // 8] ^ infos[j][:block_length // 8]).sum())
        it = np.array(iters, dtype=np.float64)
        bler = errors / args.n
        print(f"SNR {snr:f}, BLER {bler:f} ({errors}/{args.n})", file=out)
        print(f"SNR {snr:f}, BER {bit_errors / args.n / block_length / args.S:f} ({errors}/{args.n})", file=out)
        print(f"SNR {snr:f}, Mean iterations: {it.mean():f}", file=out)
        print(f"SNR {snr:f}, Std iterations: {it.std():f}", file=out)
        print(f"SNR {snr:f}, Max iterations: {int(it.max())}", file=out)
        print(f"Decoding time mean: {t_dec / len(iters) * 1e6:15.3f} us\n", file=out)
        results.append(dict(snr=snr, bler=bler, errors=errors, iters=[int(x) for x in iters]))
        if errors == 0:                                        # ldpctest.c:575
            break
        snr += args.t
    return results


def parser():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-l", type=int, default=8448, help="block length in bits")
    ap.add_argument("-s", type=float, default=-2.0, help="first SNR (dB)")
    ap.add_argument("-t", type=float, default=0.1, help="SNR step")
    ap.add_argument("-n", type=int, default=100, help="trials")
    ap.add_argument("-i", type=int, default=5, help="max decoder iterations (ldpctest default 5)")
    ap.add_argument("-S", type=int, default=1, help="segments")
    ap.add_argument("-r", type=int, default=1, help="rate numerator")
    ap.add_argument("-d", type=int, default=3, help="rate denominator")
    ap.add_argument("-q", type=int, default=8, help="quantisation bits")
    ap.add_argument("--seed", type=int, default=1, help="OAI_RNGSEED equivalent")
    ap.add_argument("--oracle", action="store_true", help="run the CPU oracle instead of libldpc_hip.so")
    return ap


if __name__ == "__main__":
    run(parser().parse_args())
