"""Multi-GPU = independent replicas (the token loop is sequential, the reference has no collective).

One process per GPU, each decoding its own sequence on its own copy of the weights; `torch.distributed` (backend
"nccl" = RCCL on ROCm, "gloo" in the CPU tests) is used only for the barrier around the timed region and for the
max-over-ranks / sum-over-ranks of the two scalars bench.py reports. No data-path collective exists or is needed.
"""


def init(backend, rank, world_size, device=None):
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world_size)
    return dist


def aggregate(elapsed_s, tokens, dist=None, device="cpu"):
    """(max over ranks of elapsed, sum over ranks of tokens): whole-job tokens/s = tokens_sum / elapsed_max."""
    if dist is None:
        return float(elapsed_s), int(tokens)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    k = torch.tensor([float(tokens)], dtype=torch.float64, device=device)
    dist.all_reduce(k, op=dist.ReduceOp.SUM)
    return float(t.item()), int(round(k.item()))


def free_port():
    """A port nobody listens on right now. (The socket is closed before the children bind -- torch's rendezvous needs to bind it itself --, so another
    process can take the port in between; spawn() then fails loudly in the rendezvous of rank 0 and can simply be run again.)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _stop(procs, live, grace):
    """terminate() the live replicas, give them `grace` seconds, kill() what is left, and reap every one of them (no zombies)."""
    import time
    for o in live:
        if procs[o].poll() is None:
            procs[o].terminate()
    t_end = time.time() + grace
    while time.time() < t_end and any(procs[o].poll() is None for o in live):
        time.sleep(0.05)
    for o in live:
        if procs[o].poll() is None:      # stuck in a HIP / RCCL call that ignores SIGTERM
            procs[o].kill()
    for o in live:
        try:
            procs[o].wait(timeout=10)
        except Exception:
            pass


def spawn(n, argv, env=None, timeout=1800, grace=10.0):
    """Launch `n` replica processes of the command `argv` (one per GPU of this node) with the torch.distributed environment a
    launcher would give them -- RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT = a free port -- and wait for all
    of them. HIP_VISIBLE_DEVICES is left alone: every replica selects its own GPU by LOCAL_RANK (q4_set_device), so a box with fewer
    than `n` GPUs fails loudly there. Rank 0's stdout is this process's stdout (the one JSON line of bench.py); the other ranks'
    stdout goes to stderr. Returns the largest exit code. A replica that fails ends the others (they would wait in the barrier for
    ever): SIGTERM, `grace` seconds, SIGKILL, and every process is reaped. After `timeout` seconds (None: never) the same happens
    and the result is 124."""
    import os
    import subprocess
    import sys
    import time
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base["MASTER_PORT"] = str(free_port())
    base["WORLD_SIZE"] = str(n)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(list(argv), env=e, stdout=None if r == 0 else sys.stderr))
    t0 = time.time()
    rc = 0
    live = set(range(n))
    while live:
        failed = False
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0:
                rc = max(rc, c if c > 0 else 1)
                failed = True
        if failed and live:
            _stop(procs, live, grace)
            return rc
        if timeout is not None and time.time() - t0 > timeout:
            _stop(procs, live, grace)
            return max(rc, 124)
        time.sleep(0.05)
    return rc
