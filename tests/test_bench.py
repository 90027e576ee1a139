"""bench.py's launch paths: `--gpus N` spawns its own ranks; with one GPU the process-group path runs with one rank, and
the strong-scaling slot's scatter / gather protocol is driven through real RCCL send / receive pairs to the rank itself
(parallel.ShardedUlsch loopback: four virtual ranks in one process) -- ncclSend / ncclRecv, view slicing and stream ordering
on hardware; what it does not show is a second GPU."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_gpus_n_without_a_launcher_spawns_its_own_ranks(built):
    """`python bench.py --gpus 2` re-executes itself under torch.distributed.run (one rank per GPU, rendezvous on
    127.0.0.1).  Without GPUs the ranks cannot start, but the launcher's report proves both were spawned."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=600)
    text = p.stdout + p.stderr
    assert "launch with torch.distributed.run" not in text
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0 and "local_rank: 0" in text and ("local_rank: 1" in text or "ChildFailedError" in text), text[-2000:]


@pytest.mark.gpu
def test_rccl_path_runs_on_one_gpu(hip):
    """BENCH_FORCE_DIST=1: process group of one rank on the nccl (= RCCL) backend; the slot of configs[4] is cut for four
    virtual ranks, 48 of its 64 transport blocks travel as isend / irecv pairs to rank 0 itself in two chunks per virtual
    peer and their payloads / ACKs / pass counts come back the same way; every payload byte must equal what was sent."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["rccl_ranks"] == 1 and line["rank_devices"] == [0] and line["n_gpus"] == 1
    assert line["value"] > 10.0 and line["roofline"]["frac"] > 0 and line["config"]["mean_passes"] == 9.0
    s = line["strong_scaling_slot"]
    assert "error" not in s and s["all_ack_and_payload_equal"] is True and s["transport_blocks_per_rank"] == [16, 16, 16, 16]
    assert s["loopback_virtual_ranks"] == 4 and s["pipeline_chunks_per_rank"] == [1, 2, 2, 2]   # (depth by the model: parallel.ShardedUlsch)
    # 48 transport blocks' LLRs out (int16) + their payload bytes, ACKs and pass counts back, per slot
    assert s["rccl_p2p_bytes_per_slot"] >= 48 * (245700 * 2 + 213176 // 8 + 5)
    c = line["chain_roofline"]
    assert "error" not in c and c["all_ack"] and 0.2 < c["frac"] < 1.0 and c["fused_segment_kernel_us"] > 0
    assert line["roofline"]["binding_resource"]["stale"] in (True, False) and line["build"]["version"].startswith("libldpc_hip")
    assert line["roofline"]["traffic_source"].startswith("profiles/hbm_traffic.json") and 0 < line["roofline"]["binding_resource"]["overhead_frac"] < 1
    assert len(line["ms_per_step_per_rank"]) == 1 and len(s["ms_per_slot_per_rank"]) == 1
    assert s["ms_per_slot_events_rank0"] <= s["ms_per_slot"] * 1.05 and s["ms_per_slot_synchronised_each"] > 0
    assert s["ms_per_slot_is"] == "back_to_back" and s["ms_per_slot_back_to_back"] == s["ms_per_slot"]
    # the prediction written down for the N = 4 ranks this loopback stands in for (parallel.predict_slot_ms)
    assert s["prediction"]["for_ranks"] == 4 and 0.2 < s["predicted_ms_range"][0] <= s["predicted_ms"] < 1.0


@pytest.mark.gpu
def test_rccl_path_with_eight_virtual_ranks(hip):
    """The node's width before the node does it: the slot cut for EIGHT virtual ranks (8 transport blocks each, seven peers'
    LLRs and results through RCCL send / receive pairs, one chunk per peer -- the depth the model picks for eight ranks), every payload byte compared."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", BENCH_LOOPBACK_RANKS="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-chain",
                        "--no-operating-point"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    s = line["strong_scaling_slot"]
    assert "error" not in s and s["all_ack_and_payload_equal"] is True and s["loopback_virtual_ranks"] == 8
    assert s["transport_blocks_per_rank"] == [8] * 8 and s["segments_per_rank"] == [208] * 8
    assert s["pipeline_chunks_per_rank"] == [1] + [1] * 7       # (eight ranks: one send group per peer, by the model)
    assert s["rccl_p2p_bytes_per_slot"] >= 56 * (245700 * 2 + 213176 // 8 + 5)


@pytest.mark.gpu
def test_plain_single_process_line(hip):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "BENCH_FORCE_DIST")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "2", "--no-cpu-baseline",
                        "--no-chain"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["rccl_ranks"] == 0 and line["n_gpus"] == 1 and line["metric"] == "ldpc_decoder_coded_throughput"
    # N = 1: the slot's prediction is the measured chain call itself (parallel.SLOT_MODEL) -- the line must agree with it
    s = line["strong_scaling_slot"]
    assert s["prediction"]["for_ranks"] == 1 and s["predicted_ms_range"] is None
    assert abs(s["predicted_ms"] / s["ms_per_slot_events_rank0"] - 1.0) < 0.10, (s["predicted_ms"], s["ms_per_slot_events_rank0"])


@pytest.mark.gpu
def test_a_secondary_leg_that_never_returns_does_not_cost_the_headline_line(hip):
    """On a node nobody has run the N > 1 legs on, a point-to-point group that never completes would end in the process
    group's watchdog aborting every rank -- and the measured headline with them.  bench.py's own deadline fires first:
    the line is printed with the leg marked, every rank exits with 0."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", BENCH_TEST_STALL="1", BENCH_DEADLINE_S="5", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29549")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-chain",
                        "--no-operating-point"], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["value"] > 10 and line["roofline"]["frac"] > 0
    assert "leg abandoned" in line["strong_scaling_slot"]["error"] and line["chain_roofline"] is None
