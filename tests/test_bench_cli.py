"""bench.py's launcher half (no GPU): `python bench.py --gpus N` issued like the N = 1 command becomes the launcher of its own N ranks
(it used to raise SystemExit, and a scaling run issued that way would have recorded rc != 0 at every N > 1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _run_main(monkeypatch, argv, env=None):
    import subprocess
    import bench
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MGF_RCCL_LIB", "MGF_BENCH_ALLOW_RCCL_OVERRIDE", "MGF_BENCH_DEVICE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    try:
        bench.main()
    except SystemExit as e:
        return calls, e.code
    return calls, None


def test_gpus_n_without_a_launcher_spawns_the_ranks(monkeypatch):
    calls, code = _run_main(monkeypatch, ["--gpus", "4", "--steps", "7", "--warmup", "3"])
    assert code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "3"]   # the ranks get the command line as it was given
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_the_stand_in_transport_puts_the_ranks_on_one_device(monkeypatch, tmp_path):
    lib = tmp_path / "libfake.so"
    lib.write_bytes(b"")
    calls, code = _run_main(monkeypatch, ["--gpus", "2", "--scene", "config5", "--rccl-lib", str(lib)])
    assert code == 0
    cmd, env = calls[0]
    assert env["MGF_BENCH_DEVICE"] == "0" and cmd[-2:] == ["--backend", "gloo"]   # (this container has no device at all: fewer than 2)


def test_a_mismatched_launcher_is_refused(monkeypatch):
    calls, code = _run_main(monkeypatch, ["--gpus", "4"], env={"WORLD_SIZE": "2", "RANK": "0"})
    assert not calls and "WORLD_SIZE=2" in str(code)
