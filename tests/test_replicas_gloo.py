"""N>1 path of bench.py (replicas, no data-path collective) under world_size=2 gloo on the CPU."""
import os
import socket
import sys

import pytest

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from llama_cu_awq_amd import replicas
    dist = replicas.init("gloo", rank, world)
    dist.barrier()
    elapsed, tokens = replicas.aggregate(1.0 + rank * 0.5, 255 * (rank + 1), dist)
    dist.barrier()
    q.put((rank, elapsed, tokens))
    dist.destroy_process_group()


def test_two_replicas_aggregate():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, elapsed, tokens in out:
        assert elapsed == pytest.approx(1.5)          # max over ranks
        assert tokens == 255 * 3                      # sum over ranks


def test_single_process_passthrough():
    from llama_cu_awq_amd import replicas
    assert replicas.aggregate(2.5, 100) == (2.5, 100)


_REPLICA = r"""
import json, os, sys
sys.path.insert(0, %r)
from llama_cu_awq_amd import replicas
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
if len(sys.argv) > 1 and rank == int(sys.argv[1]):
    sys.exit(7)                       # this replica "has no GPU"
dist = replicas.init("gloo", rank, world)
dist.barrier()
elapsed, tokens = replicas.aggregate(1.0 + rank, 255, dist)
dist.barrier()
if rank == 0:
    print(json.dumps({"n_gpus": world, "elapsed": elapsed, "tokens": tokens}))
dist.destroy_process_group()
"""


def test_spawned_replicas_report_one_line_from_rank_0(tmp_path):
    """bench.py --gpus N without a launcher: replicas.spawn starts N replicas with the launcher's environment; rank 0's stdout is
    the job's stdout, the aggregate covers every rank."""
    import json
    import subprocess
    script = tmp_path / "replica.py"
    script.write_text(_REPLICA % ROOT)
    drv = ("import sys; sys.path.insert(0, %r); from llama_cu_awq_amd import replicas; "
           "sys.exit(replicas.spawn(2, [sys.executable, %r], timeout=120))" % (ROOT, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"n_gpus": 2, "elapsed": 2.0, "tokens": 510}


def test_a_failing_replica_fails_the_job(tmp_path):
    """A replica that cannot get its GPU (q4_set_device fails) must fail the whole job, not leave the others in the barrier."""
    import subprocess
    script = tmp_path / "replica.py"
    script.write_text(_REPLICA % ROOT)
    drv = ("import sys; sys.path.insert(0, %r); from llama_cu_awq_amd import replicas; "
           "sys.exit(replicas.spawn(2, [sys.executable, %r, '1'], timeout=120))" % (ROOT, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 7
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


_STUCK = """
import os, signal, sys, time
rank = int(os.environ["RANK"])
if rank == 0:
    time.sleep(0.5)
    sys.exit(5)
signal.signal(signal.SIGTERM, signal.SIG_IGN)     # a peer inside a HIP / RCCL call that does not return
open(sys.argv[1], "w").write(str(os.getpid()))
time.sleep(600)
"""


def test_a_peer_that_ignores_sigterm_is_killed_and_reaped(tmp_path):
    """After a replica has failed, the others get SIGTERM, a grace period, SIGKILL, and are waited for: the launcher returns the failure's
    exit code within seconds and leaves no process behind."""
    import subprocess
    import time
    script = tmp_path / "stuck.py"
    script.write_text(_STUCK)
    pidfile = tmp_path / "pid"
    drv = ("import sys; sys.path.insert(0, %r); from llama_cu_awq_amd import replicas; "
           "sys.exit(replicas.spawn(2, [sys.executable, %r, %r], timeout=120, grace=1.0))" % (ROOT, str(script), str(pidfile)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 5 and time.time() - t0 < 30
    pid = int(pidfile.read_text())
    gone = False
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        gone = True
    assert gone, "the stuck replica is still there"


def test_bench_gpus_flag_spawns_replicas():
    """bench.py reads --gpus: without WORLD_SIZE and N > 1 it goes through replicas.spawn before anything touches a GPU."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[src.index("args = ap.parse_args()"):src.index('rank = int(os.environ.get("RANK"')]
    assert "args.gpus > 1" in head and "replicas.spawn(args.gpus" in head
