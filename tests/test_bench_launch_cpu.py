"""The N > 1 launch path of bench.py, on CPU: `python bench.py --gpus N` with NO launcher must start its own ranks
(torch.distributed.run on 127.0.0.1), rendezvous, see every rank in the process group and complete an all-gather --
`--plumbing-check` runs exactly that with gloo and no model, so the round-end multi-GPU run cannot die in plumbing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_plain_invocation_spawns_its_own_ranks(world):
    res = _run(["--gpus", str(world), "--plumbing-check"])
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["plumbing_check"] and d["n_gpus"] == world and d["all_gather_ok"]
    seen = d["ranks_seen"]
    assert seen["world_size"] == world and sorted(r["rank"] for r in seen["ranks"]) == list(range(world))
    assert len({r["pid"] for r in seen["ranks"]}) == world               # really one process per rank
    assert sum(d["pages_per_rank"]) == 64 and max(d["pages_per_rank"]) - min(d["pages_per_rank"]) <= 64 // world


def test_launched_by_torchrun_is_accepted_as_is():
    """The driver's own form: python -m torch.distributed.run ... bench.py --gpus N (ranks already exist: no re-launch)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-check"],
                         capture_output=True, text=True, env=env, timeout=240, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["all_gather_ok"]


def test_world_size_mismatch_is_refused():
    res = _run(["--gpus", "3", "--plumbing-check"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    # (a WORLD_SIZE in the environment means "already launched": the check itself runs with that world and would hang
    # waiting for rank 1, so bench.py must refuse the mismatch before any rendezvous)
    assert res.returncode != 0 and "WORLD_SIZE=2 but --gpus 3" in (res.stderr + res.stdout)


def test_collective_flag_plumbing():
    """SBBSEG_BENCH_COLLECTIVE=capi (the all-gather inside the C ABI instead of torch.distributed): the flag reaches every rank, rank 0's
    128-byte communicator id arrives unchanged on every rank, the default stays torch, and anything else is refused before a
    rendezvous.  `--workload auto` is the page workload at every N (one workload along the driver's 1/2/4/8 sweep)."""
    res = _run(["--gpus", "2", "--plumbing-check"], env_extra={"SBBSEG_BENCH_COLLECTIVE": "capi"})
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["collective"] == "capi" and d["unique_id_broadcast_ok"] is True and d["all_gather_ok"]
    assert d["default_workload"] == "page"
    res = _run(["--gpus", "2", "--plumbing-check"])
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert res.returncode == 0 and d["collective"] == "torch" and d["unique_id_broadcast_ok"] is None and d["default_workload"] == "page"
    res = _run(["--gpus", "2", "--plumbing-check"], env_extra={"SBBSEG_BENCH_COLLECTIVE": "mpi"})
    assert res.returncode != 0 and "SBBSEG_BENCH_COLLECTIVE=mpi" in (res.stderr + res.stdout)
