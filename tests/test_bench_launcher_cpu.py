"""CPU: `python bench.py --gpus N` as a bare command starts its own rank processes: fresh children with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, rank 0's stdout relayed, worst exit code returned — and the parent
never loads the GPU library.  The children here are a probe script (no GPU in this suite)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_launcher(tmp_path, body, n, extra_env=None):
    probe = tmp_path / "probe.py"
    probe.write_text(textwrap.dedent(body))
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        rc = bench.launch_ranks({n}, ["--x", "1"], script={str(probe)!r})
        assert "dot_ring_amd" not in sys.modules and "dot_ring_amd._native" not in sys.modules   # the parent never touches the GPU
        sys.exit(rc)
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, str(driver)], capture_output=True, text=True, timeout=120, env=env)


def test_launcher_exports_rank_environment_and_relays_rank0(tmp_path):
    proc = _run_launcher(tmp_path, """
        import json, os, sys
        rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        rec["argv"] = sys.argv[1:]
        print(json.dumps(rec), flush=True)            # only rank 0's stdout is the launcher's stdout
    """, 3)
    assert proc.returncode == 0, proc.stderr
    lines = [json.loads(t) for t in proc.stdout.splitlines() if t.startswith("{")]
    assert len(lines) == 1
    assert lines[0]["RANK"] == "0" and lines[0]["WORLD_SIZE"] == "3" and lines[0]["MASTER_ADDR"] == "127.0.0.1"
    assert lines[0]["argv"] == ["--x", "1"] and int(lines[0]["MASTER_PORT"]) > 0
    others = [json.loads(t) for t in proc.stderr.splitlines() if t.startswith("{")]
    assert sorted(o["RANK"] for o in others) == ["1", "2"]
    assert {o["MASTER_PORT"] for o in others} == {lines[0]["MASTER_PORT"]}


def test_launcher_returns_worst_exit_code(tmp_path):
    proc = _run_launcher(tmp_path, """
        import os, sys
        print("{}", flush=True)
        sys.exit(3 * int(os.environ["RANK"]))
    """, 3)
    assert proc.returncode == 6
    assert "rank exit codes [0, 3, 6]" in proc.stderr


def test_bare_bench_command_fails_loudly_without_gpu():
    """no GPU here: both ranks fail in dr_ctx_create, the launcher reports it and exits non-zero within seconds (no CPU fallback)"""
    import pytest

    from dot_ring_amd import _native
    try:
        _native.Context(0).close()
        pytest.skip("a GPU is present: the bare command is exercised by tests/test_gpu_bench_launch.py")
    except _native.DotRingHipError:
        pass
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                          capture_output=True, text=True, timeout=300, env=env)
    assert proc.returncode != 0
    assert "rank exit codes" in proc.stderr and "no HIP device" in proc.stderr
