"""The multi-rank route of bench.py on real hardware (`core/mod.rs:153-180`): launcher ->
`python -m torch.distributed.run` -> rank(s) -> process group -> sharding (`th_shard_assign`) -> the step with the
2-float dB-range exchange between its two kernels -> max-over-ranks timing -> image-tile gather -> one JSON record.

Every case starts bench.py as a CHILD process (`subprocess`, never an exec of the test process) and gives it a port of its
own.  A one-GPU box can show three things, and they are the three tests:
  (i)   `TH_BENCH_FORCE_LAUNCHER=1 --gpus 1`: the SCALE command's whole route at one rank, over RCCL (`nccl` backend);
  (ii)  `TH_BENCH_SHARE_GPU=1 --gpus 2`: two real ranks on the one card (gloo — RCCL refuses two ranks on one device);
        every rank must have quantised against the same global dB range;
  (iii) `--gpus 2` on a box with one GPU: refused, exit code != 0, nothing on stdout (never a smaller job under that name).
The ranks of (ii) are 2 processes on the card: inside the box's limit of 6.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEAN = ["--steps", "3", "--warmup", "1", "--spin-up-steps", "4", "--no-single-track", "--no-cpu-baseline", "--no-skeleton"]


def _free_port() -> str:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _bench(args, env_extra, timeout=600):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "TH_BENCH_ASSUME_GPUS", "TH_BENCH_FORCE_DIST", "TH_BENCH_FORCE_LAUNCHER", "TH_BENCH_SHARE_GPU")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=ROOT)


def _record(r):
    assert r.returncode == 0, f"rc {r.returncode}\nstderr tail:\n{r.stderr[-3000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert lines, "no record on stdout"
    rec = json.loads(lines[-1])           # the record is the LAST stdout line (the driver reads it that way)
    assert "metric" in rec and "bench_extras" not in rec
    return rec


def _n_gpus() -> int:
    import torch
    return torch.cuda.device_count()   # counting devices creates no HIP context on this image


def test_launcher_torchrun_rccl_route_at_one_rank():
    """launcher -> torchrun -> rank 0 -> RCCL process group -> all_reduce(MIN) inside the timed step -> gather -> record."""
    r = _bench(["--gpus", "1"] + LEAN, {"TH_BENCH_FORCE_LAUNCHER": "1", "MASTER_PORT": _free_port()})
    rec = _record(r)
    assert "bench.py launcher:" in r.stderr and "torch.distributed.run" in r.stderr   # the child was torchrun
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert rec["range_allreduce"]["backend"] == "nccl (RCCL)" and rec["range_allreduce"]["ranks"] == 1
    assert rec["range_allreduce"]["in_step_ms"] > 0
    assert rec["tile_gather"]["ranks"] == 1
    assert rec["global_dB_range"]["identical_on_all_ranks"] is True
    lo, hi = rec["global_dB_range"]["min_max_dB"]
    assert hi <= 0.0 and hi - lo <= 100.0 + 1e-3          # core/mod.rs:179-180: max = min(max, 0), min = max(min, max - 100)
    assert rec["value"] > 0 and abs(rec["value"] - rec["config"]["frames_per_gpu"] * 3 / (rec["ms_per_step"] * 3e-3)) <= 1e-6 * rec["value"]
    assert rec["roofline"]["range_allreduce_ranks"] == 1 and rec["roofline"]["tile_gather_ranks"] == 1


def test_two_ranks_share_the_card_and_agree_on_the_global_range():
    """N = 2 with real processes: sharding, the exchange between the two kernels of the step, max-over-ranks time, gather."""
    # (no MASTER_PORT: the launcher lets torchrun's c10d rendezvous pick a free port on 127.0.0.1 itself)
    r = _bench(["--gpus", "2", "--tracks-per-gpu", "32"] + LEAN, {"TH_BENCH_SHARE_GPU": "1"})
    rec = _record(r)
    assert "--rdzv-endpoint=127.0.0.1:0" in r.stderr
    assert rec["n_gpus"] == 2 and rec["config"]["ranks_share_one_gpu"] is True and rec["config"]["backend"] == "gloo"
    assert rec["global_dB_range"] == {"min_max_dB": rec["global_dB_range"]["min_max_dB"], "identical_on_all_ranks": True, "ranks": 2}
    assert rec["tile_gather"]["ranks"] == 2 and rec["tile_gather"]["inbound_GBs"] > 0
    # value = frames of BOTH ranks / the slower rank's time
    frames = rec["config"]["frames_per_gpu"] * 2 * rec["steps"]
    assert abs(rec["value"] - frames / (rec["ms_per_step"] * rec["steps"] * 1e-3)) <= 1e-6 * rec["value"]
    # one rank alone over the same 64 tracks must find the same global range: the exchange is a MIN over ranks of
    # [min, -max], bit-exact (no arithmetic), so the two-rank range equals the one-rank range of the union
    one = _record(_bench(["--gpus", "1", "--tracks-per-gpu", "64"] + LEAN, {"TH_BENCH_FORCE_DIST": "1", "MASTER_PORT": _free_port()}))
    assert one["global_dB_range"]["min_max_dB"] == rec["global_dB_range"]["min_max_dB"]


def test_more_ranks_than_gpus_is_refused_with_nothing_on_stdout():
    n = _n_gpus()
    r = _bench(["--gpus", str(n + 1)] + LEAN, {}, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert f"--gpus {n + 1} needs {n + 1} visible GPUs" in r.stderr
