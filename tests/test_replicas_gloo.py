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
