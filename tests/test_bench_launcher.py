"""bench.py's N > 1 launcher: `--gpus N` without WORLD_SIZE starts N torchrun ranks as a CHILD process
before anything touches the GPU, and refuses loudly when fewer than N devices are visible.  CPU-only checks: the command /
environment it would start, the refusals, and that the launcher side imports nothing GPU-side."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_launcher_plan_builds_the_drivers_command():
    b = _bench()
    cmd, env = b.launcher_plan(4, ["--gpus", "4", "--steps", "20", "--warmup", "5"], {"PATH": "/x"}, n_visible=8)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    # no port named: torchrun's c10d rendezvous picks a free one itself (no bind-and-close race), still on 127.0.0.1
    assert "--rdzv-endpoint=127.0.0.1:0" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"
    assert "--master-port" not in cmd
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]    # the same arguments, unchanged
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/x"
    assert "WORLD_SIZE" not in env                                              # torchrun sets it for the ranks
    cmd2, _ = b.launcher_plan(2, [], {"MASTER_PORT": "29777"}, n_visible=2)
    assert cmd2[cmd2.index("--master-port") + 1] == "29777" and cmd2[cmd2.index("--master-addr") + 1] == "127.0.0.1"


def test_launcher_refuses_fewer_devices_than_ranks():
    b = _bench()
    with pytest.raises(SystemExit) as e:
        b.launcher_plan(8, [], {}, n_visible=1)
    assert "needs 8 visible GPUs" in str(e.value) and "shows 1" in str(e.value)


def test_share_gpu_rehearsal_passes_the_device_check_only_when_asked():
    """TH_BENCH_SHARE_GPU=1: N ranks on GPU 0 over gloo — the launcher lets N > visible through (a GPU must still be there)."""
    b = _bench()
    cmd, env = b.launcher_plan(4, ["--gpus", "4"], {"TH_BENCH_SHARE_GPU": "1"}, n_visible=1)
    assert "--nproc-per-node=4" in cmd and env["TH_BENCH_SHARE_GPU"] == "1"
    with pytest.raises(SystemExit):
        b.launcher_plan(4, [], {"TH_BENCH_SHARE_GPU": "1"}, n_visible=0)
    with pytest.raises(SystemExit):
        b.launcher_plan(4, [], {}, n_visible=1)


def test_visible_gpu_count_honours_visibility_masks():
    b = _bench()
    assert b.visible_gpu_count({"TH_BENCH_ASSUME_GPUS": "8"}) == 8
    assert b.visible_gpu_count({"TH_BENCH_ASSUME_GPUS": "0"}) == 0


def _run(args, env_extra, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=ROOT)


def test_gpus_2_on_a_box_without_two_gpus_fails_loudly():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"TH_BENCH_ASSUME_GPUS": "1"})
    assert r.returncode != 0
    assert "--gpus 2 needs 2 visible GPUs" in r.stderr and r.stdout.strip() == ""     # no record of a smaller job


def test_world_size_must_equal_gpus():
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_launcher_side_imports_nothing_gpu_side_and_relays_the_child(tmp_path):
    """The parent of `--gpus N`: no torch (or thesia_amd) import before / while the child runs; the child's stdout is passed
    through and its exit code returned.  A stand-in child (TH_BENCH_ASSUME_GPUS lets the plan through; `python -m
    torch.distributed.run` is replaced by a stub module on PYTHONPATH) keeps this a CPU test."""
    stub = tmp_path / "torch" / "distributed"
    stub.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (stub / "__init__.py").write_text("")
    (stub / "run.py").write_text(
        "import json, sys\n"
        "print('banner')\n"
        "print(json.dumps({'argv': sys.argv[1:], 'n_gpus': 3}))\n"
        "sys.exit(7)\n")
    probe = tmp_path / "probe.py"
    probe.write_text(
        "import sys, os\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "sys.argv = ['bench.py', '--gpus', '3', '--steps', '2']\n"
        "try:\n"
        "    bench.main()\n"
        "except SystemExit as e:\n"
        "    code = e.code\n"
        "bad = [m for m in sys.modules if m == 'torch' or m.startswith('torch.') or m.startswith('thesia_amd')]\n"
        "print(json_line := __import__('json').dumps({'code': code, 'gpu_side_modules': bad}), file=sys.stderr)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"TH_BENCH_ASSUME_GPUS": "3", "PYTHONPATH": str(tmp_path)})
    r = subprocess.run([sys.executable, str(probe)], capture_output=True, text=True, env=env, timeout=120)
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "banner"
    rec = json.loads(lines[-1])                                     # the child's record is the parent's last stdout line
    assert rec["n_gpus"] == 3 and "--nproc-per-node=3" in rec["argv"] and rec["argv"][-4:] == ["--gpus", "3", "--steps", "2"]
    verdict = json.loads(r.stderr.strip().splitlines()[-1])
    assert verdict == {"code": 7, "gpu_side_modules": []}           # exit code relayed; nothing GPU-side imported
