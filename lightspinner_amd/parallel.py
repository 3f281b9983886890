"""Multi-GPU: columns are independent 1-D problems, so the path shards
trivially -- a block partition of columns over ranks, one process per GPU, no data-path
collective.  The ONLY exchange is the global stop criterion of the MALI loop
(`while dJ > 2e-3 or dPops > 1e-3`, test.py:23): one all-reduce(MAX) of two float64 per
iteration, over RCCL (torch.distributed backend "nccl") on GPUs, gloo in the CPU tests."""
import os
from typing import Tuple

import numpy as np


def shard_columns(ncol_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: -> (first column, number of columns) of `rank`.
    The first (ncol_total % world) ranks get one extra column."""
    if not (0 <= rank < world) or ncol_total < 0:
        raise ValueError('bad shard request')
    base, rem = divmod(ncol_total, world)
    n = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, n


def env_rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


class MaxReducer:
    """all-reduce(MAX) of (dJ, dPops).  NaN must win like in the single-process max
    (numpy max semantics, rh_method.py:706): MAX collectives do not define NaN ordering, so a
    NaN flag travels as a third element."""

    def __init__(self, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.buf = torch.zeros(3, dtype=torch.float64, device=device if device is not None else 'cpu')

    def __call__(self, dJ: float, dPops: float):
        if not self.active:
            return dJ, dPops
        nan = float(np.isnan(dJ) or np.isnan(dPops))
        vals = [0.0 if np.isnan(dJ) else dJ, 0.0 if np.isnan(dPops) else dPops, nan]
        self.buf.copy_(self.torch.tensor(vals, dtype=self.torch.float64))
        self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.MAX, group=self.group)
        out = self.buf.tolist()
        if out[2] > 0:
            return float('nan'), float('nan')
        return out[0], out[1]
