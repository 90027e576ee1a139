"""-m gpu: the HIP encoder through the C ABI vs the oracle (bit-exact) and vs H*c = 0."""
import numpy as np
import pytest

import oracle_lib as O
from common import GOLDEN, kbits, random_info

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("BG", [1, 2])
def test_every_lifting_size(hip, BG):
    rng = np.random.default_rng(40 + BG)
    for Z in O.LIFT_SIZES:
        K = kbits(BG, Z)
        for Kb in ([22] if BG == 1 else [10, 9, 8, 6]):
            info = np.stack([random_info(rng, BG, Z) for _ in range(3)])
            if Kb < 10:                      # columns >= Kb are filler zeros in real use
                bits = np.unpackbits(info, axis=1)
                bits[:, Kb * Z:] = 0
                info = np.packbits(bits, axis=1)
            out = hip.encode_batch_host(BG, Z, info, Kb)
            for i in range(3):
                ref = O.encode(BG, Z, info[i], Kb)
                assert np.array_equal(out[i], ref), (BG, Z, Kb, i)
                x = np.concatenate([np.unpackbits(info[i])[:2 * Z], out[i]])
                assert O.syndrome_weight(BG, Z, x) == 0


def test_reference_entry_point_macro_groups(hip):
    """LDPCencoder(): n_segments / macro_num addressing of ldpc_encoder_optim8segmulti.c:64-65."""
    BG, Z = 1, 384
    rng = np.random.default_rng(3)
    segs = [random_info(rng, BG, Z) for _ in range(11)]
    out0 = hip.LDPCencoder(segs, BG, Z, n_segments=11, macro_num=0)
    out1 = hip.LDPCencoder(segs, BG, Z, n_segments=11, macro_num=1)
    for j in range(11):
        ref = O.encode(BG, Z, segs[j])
        got = out0[j] if j < 8 else out1[j]
        other = out1[j] if j < 8 else out0[j]
        assert np.array_equal(got, ref), j
        assert not other.any(), j              # a macro group never touches the other group's buffers


def test_survey_stage_vectors(hip):
    z = np.load(GOLDEN / "survey_ref_encoder.npz")
    for i, (BG, Z, Kb, n) in enumerate(z["meta"]):
        out = hip.encode_batch_host(int(BG), int(Z), z[f"info_{i}"][None, :], int(Kb))
        assert np.array_equal(np.packbits(out[0][:n]), z[f"coded_{i}"]), (BG, Z, Kb)


def test_linearity_full_batch_device(hip):
    """enc(a ^ b) == enc(a) ^ enc(b) on 1024 device-resident BG1 Zc=384 blocks."""
    import torch
    BG, Z, n = 1, 384, 1024
    K = 22 * Z
    g = torch.Generator(device="cuda").manual_seed(77)
    a = torch.randint(0, 256, (n, K // 8), dtype=torch.uint8, device="cuda", generator=g)
    b = torch.randint(0, 256, (n, K // 8), dtype=torch.uint8, device="cuda", generator=g)
    ea, eb, eab = (torch.empty((n, 66 * Z), dtype=torch.uint8, device="cuda") for _ in range(3))
    hip.encode_batch_device(BG, Z, a, ea)
    hip.encode_batch_device(BG, Z, b, eb)
    hip.encode_batch_device(BG, Z, a ^ b, eab)
    torch.cuda.synchronize()
    assert torch.equal(ea ^ eb, eab)
    assert int(eab.max()) <= 1


def test_byte_per_lane_kernel_too():
    """The default encoder kernel is the bit-packed one; re-run the every-code test with the byte-per-lane kernel."""
    import os
    import subprocess
    import sys
    if os.environ.get("NRLDPC_HIP_ENC_KERNEL") == "bytes":
        pytest.skip("already the byte-per-lane run")
    env = dict(os.environ, NRLDPC_HIP_ENC_KERNEL="bytes")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_encoder.py"), "-m", "gpu", "-q", "-x",
                        "-k", "every_lifting_size or survey_stage or reference_compiled"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_reference_meters_are_filled(hip):
    """The encoder's four optional meters (nrLDPC_defs.h:44-47) and the decoder's profiler are written the way the
    reference's start_meas()/stop_meas() write them (common/utils/time_meas.h:148-176): trials counted, ticks accumulated,
    flag cleared -- nr_dlsim's "encoder timing" lines read them."""
    m = hip.ldpc
    BG, Z = 1, 384
    rng = np.random.default_rng(3)
    infos = [rng.integers(0, 256, 22 * Z // 8, dtype=np.uint8) for _ in range(8)]
    meters = [m.time_stats_t() for _ in range(4)]
    for rep in range(3):
        out = m.LDPCencoder(infos, BG, Z, meters=meters)
    assert np.array_equal(out[0], O.encode(BG, Z, infos[0]))
    for k, ts in enumerate(meters):
        assert ts.trials == 3 and ts.meas_flag == 0 and ts.diff >= 0 and ts.max <= ts.diff, (k, ts.trials, ts.diff)
    assert meters[0].diff > 0 and meters[2].diff > 0 and meters[3].diff > 0 and meters[2].p_time > 0
    assert meters[2].diff_square > 0
    prof = m.t_nrLDPC_time_stats()
    llr = np.zeros(68 * Z, np.int8)
    llr[2 * Z:] = 20
    p = hip.make_dec_params(BG, Z, 13, 8)
    for _ in range(2):
        n, _ = hip.LDPCdecoder(p, llr, profiler=prof)
    assert n == 2 and prof.total.trials == 2 and prof.total.diff > 0 and prof.total.meas_flag == 0
    assert prof.cnProc.trials == 0                             # only `total` is written (include/nrLDPC_hip.h)


def test_reference_entry_point_without_the_resident_kernel():
    """NRLDPC_HIP_ENC_SERVER=0: LDPCencoder with one launch per call (the path a box without the resident encoder kernel
    takes) gives the same code words: the entry-point tests once more under it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NRLDPC_HIP_ENC_SERVER="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_encoder.py"), "-m", "gpu", "-q", "-x",
                        "-k", "reference_entry_point_macro or meters"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_reference_compiled_code_words(hip):
    """tests/golden/ref_encoder.npz: code words of the reference-COMPILED encoder (oracle/_ref: ldpc_encoder.c's assembly
    around encode_parity_check_part_orig of ldpc_generate_coefficient.c with the reference's generator tables) -- every
    (BG, Zc) twice, BG2 with Kb = 6, 8, 9.  Through the batch entry and the by-name LDPCencoder."""
    from common import load_ref_code_words
    words = list(load_ref_code_words())
    assert len(words) >= 204
    for v in words:                                           # (the byte-per-lane kernel: test_byte_per_lane_kernel_too)
        out = hip.encode_batch_host(v["BG"], v["Z"], v["info"][None, :], v["Kb"])
        assert np.array_equal(out[0], v["coded"]), (v["BG"], v["Z"], v["Kb"])
    by_code = {}
    for v in words:
        if v["Kb"] == (22 if v["BG"] == 1 else 10):
            by_code.setdefault((v["BG"], v["Z"]), []).append(v)
    for (BG, Z), vs in by_code.items():                       # the symbol the reference's loader binds
        outs = hip.LDPCencoder([v["info"] for v in vs], BG, Z, n_segments=len(vs), macro_num=0)
        for v, o in zip(vs, outs):
            assert np.array_equal(o, v["coded"]), (BG, Z)


def test_unaligned_buffers_on_the_word_aligned_codes(hip):
    """Zc % 32 == 0 has a path of its own (ldpc_enc_packed32.h) that wants 4-byte aligned input rows and 16-byte aligned
    output rows; anything else must fall back to the general path and give the same code words (rows at odd addresses and
    odd pitches, device buffers)."""
    import torch
    rng = np.random.default_rng(123)
    for BG, Z in ((1, 384), (2, 64), (1, 32), (2, 256)):
        K, N = kbits(BG, Z), (66 if BG == 1 else 50) * Z
        n = 6
        info = rng.integers(0, 256, (n, K // 8), dtype=np.uint8)
        ref = np.stack([O.encode(BG, Z, info[i]) for i in range(n)])
        for in_off, out_off in ((0, 0), (1, 0), (0, 4), (3, 5), (2, 16)):
            buf_in = torch.zeros((n, K // 8 + 8 + in_off), dtype=torch.uint8, device="cuda")
            buf_in[:, in_off:in_off + K // 8] = torch.from_numpy(info).cuda()
            buf_out = torch.full((n, N + 32 + out_off), 7, dtype=torch.uint8, device="cuda")
            hip.encode_batch_device(BG, Z, buf_in[:, in_off:], buf_out[:, out_off:])
            torch.cuda.synchronize()
            got = buf_out.cpu().numpy()
            assert np.array_equal(got[:, out_off:out_off + N], ref), (BG, Z, in_off, out_off)
            assert (got[:, :out_off] == 7).all() and (got[:, out_off + N:] == 7).all()       # nothing outside the rows
