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
