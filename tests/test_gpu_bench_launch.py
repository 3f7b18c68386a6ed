"""GPU: bench.py for N > 1 started as ONE bare command: the process spawns its two ranks itself, both share
the test box's single GPU (DOTRING_BENCH_SHARE_GPU=1: device 0 for every rank, TCP all-gather since RCCL refuses two ranks per
device), the line carries n_gpus = 2, a CPU baseline and per-rank rooflines, the config5 leg (ONE batch over ring 3839 / domain 4096 sharded by
parallel.prove_batch_sharded, gathered, verified, proofs of both shards against the oracle) and the base-sharded MSM legs;
and a collective that cannot come up still prints the line but exits non-zero."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *flags, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", *flags],
                          capture_output=True, text=True, timeout=timeout, env=env)


def test_bare_command_two_ranks_sharing_the_gpu():
    proc = _bench({"DOTRING_BENCH_SHARE_GPU": "1"}, "--batch", "256", "--cpu-proofs", "2", "--msm-log2n", "16")
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [t for t in proc.stdout.splitlines() if t.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["parity_ok"] and line["metric"] == "ringvrf_proofs_per_sec"
    assert line["value"] > 0 and line["config"]["ring_size"] == 1024 and line["config"]["batch_per_gpu"] == 256
    # an N > 1 line is complete: the CPU port timed in the same run (rank 0), a roofline for every rank
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] == 1
    assert len(line["roofline"]["by_rank"]) == 2 and all(r["frac"] > 0 for r in line["roofline"]["by_rank"])
    # BASELINE configs[4] through the library call: ONE batch of 512 proofs sharded over the two ranks, gathered and verified
    c5 = line["config5"]
    assert (c5["ring_size"], c5["domain_size"], c5["ranks"], c5["batch_total"], c5["batch_per_rank"]) == (3839, 4096, 2, 512, 256)
    assert c5["parity_ok"] and c5["parity_proof_indices"] == [0, 256, 511] and c5["proofs_per_s"] > 0
    assert c5["gather"]["bytes"] == 784 * 512 and c5["gather"]["inside_timed_region"] and c5["gather"]["to_every_rank_ms"] > 0
    legs = line["g1_msm_sharded"]
    assert [leg["scaling"] for leg in legs] == ["strong", "weak"]
    assert all(leg["parity_closed_form_all_ranks"] and leg["ranks"] == 2 and leg["collective"] == "SocketComm" for leg in legs)


def test_failing_collective_prints_the_line_and_exits_nonzero():
    """two ranks on device 0 with the RCCL communicator kept (DOTRING_BENCH_SHARE_GPU=rccl) and a collective library that cannot be
    loaded: rank 0's failure travels to its peer as a marker instead of leaving it in recv, the headline survives in the line,
    g1_msm_sharded names the failing call and ranks, and the exit code is non-zero"""
    proc = _bench({"DOTRING_BENCH_SHARE_GPU": "rccl", "DOTRING_BENCH_SHARDED_TIMEOUT": "60", "DOTRING_RCCL_LIB": "/nonexistent/librccl.so"},
                  "--batch", "64", "--cpu-proofs", "0", "--msm-log2n", "12", "--extras", "0")
    assert proc.returncode not in (0, 1, 2), proc.stderr[-3000:]
    lines = [t for t in proc.stdout.splitlines() if t.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0
    sh = line["g1_msm_sharded"]
    assert "error" in sh and sorted(sh["failed_ranks"]) == [0, 1] and "ncclCommInitRank" in sh["error"]
