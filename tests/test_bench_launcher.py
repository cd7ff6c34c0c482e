"""`python bench.py --gpus N` from a bare shell starts N fresh rank processes itself.

The reference leaves process launch to Lightning (`trainer: devices: N`, /root/reference/configs/sdxl/sdxl.example.yaml:3-15); here the
bench is its own launcher.  What must hold: the launcher runs before this process touched the GPU, the children are ordinary
subprocesses on a loopback rendezvous, the parent passes their exit code on, and a process that already is a rank never launches.
"""
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


class _Done:
    def __init__(self, rc):
        self.returncode = rc


def test_launcher_starts_fresh_ranks_before_any_gpu_call(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    seen = {}

    def runner(cmd, env):
        seen["cmd"], seen["env"] = cmd, env
        seen["cuda_initialised"] = torch.cuda.is_initialized()
        return _Done(7)

    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1", "--backend", "gloo"]
    rc = bench.launch_ranks(bench.parse_args(argv), argv, runner=runner)
    assert rc == 7                                         # the children's exit code is the parent's
    assert seen["cuda_initialised"] is False               # nothing touched the GPU before the ranks were started
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(str(ROOT / "bench.py"))
    assert cmd[script + 1:] == argv                        # the rank processes get the very same arguments
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_rank_never_launches(monkeypatch):
    def runner(cmd, env):  # pragma: no cover - must not be reached
        raise AssertionError("a process that already is a rank started a launcher")

    monkeypatch.setenv("WORLD_SIZE", "4")
    assert bench.launch_ranks(bench.parse_args(["--gpus", "4"]), ["--gpus", "4"], runner=runner) is None
    monkeypatch.delenv("WORLD_SIZE")
    assert bench.launch_ranks(bench.parse_args([]), [], runner=runner) is None      # N = 1 runs in this process


def test_main_takes_the_launcher_path_first(monkeypatch):
    """main() must reach the launcher before set_device / lib.load: the first GPU call of the parent would make it unsafe to spawn."""
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    calls = []
    monkeypatch.setattr(bench, "launch_ranks", lambda args, argv: calls.append((args.gpus, torch.cuda.is_initialized())) or 0)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *_: pytest.fail("GPU call before the launcher"))
    with pytest.raises(SystemExit) as ex:
        bench.main(["--gpus", "2"])
    assert ex.value.code == 0 and calls == [(2, False)]


def test_bare_shell_command_relays_the_childrens_exit_code(tmp_path):
    """The real thing with real processes, CPU only: `python bench.py --gpus 2` in a container without a GPU starts two ranks through
    torch.distributed.run; they fail at their first GPU call (this container has none) and the parent exits with THEIR non-zero code
    instead of the SystemExit message of rounds 1-4."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.is_available():
        pytest.skip("CPU rehearsal of the launcher; the GPU box runs tests/test_dp_gpu.py::test_bench_two_ranks_from_a_bare_shell")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert "launching 2 ranks" in p.stderr
    assert "launch multi-GPU runs with torch.distributed.run" not in p.stderr + p.stdout
    assert p.returncode != 0                 # no GPU here: the ranks fail, and the parent says so
