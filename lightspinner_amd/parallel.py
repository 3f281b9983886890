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


class OptionsMismatch(RuntimeError):
    """ranks of one job hold engines that would associate their sums differently (different options, rule, sweep mapping or plan)"""


def check_same_options(eng, group=None):
    """Start-up check of a sharded job ("per-column results must be bitwise equal to the 1-GPU run"): every rank's engine
    must have been made with the same options, rule, sweep mapping and plan -- `Engine.options_signature()` (include/lsx.h,
    lsx_options_signature) is exchanged once and a mismatch is refused ON EVERY RANK, naming the ranks that differ from rank 0 and this
    rank's own effective options.  (An environment variable such as LSX_NO_LINKED set on one rank only used to change that rank's bits
    silently.)  `eng`: a problem.Engine, or anything with options_signature() / effective_options().  No-op in a single process."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sig = int(eng.options_signature()) & 0xffffffffffffffff
    dev = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
    mine = torch.tensor([sig >> 32, sig & 0xffffffff], dtype=torch.int64, device=dev)
    every = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    sigs = [(int(t[0]) << 32) | int(t[1]) for t in every]
    bad = [r for r in range(world) if sigs[r] != sigs[0]]
    if bad:
        raise OptionsMismatch('rank %d: the engines of this job were not made alike -- ranks %s differ from rank 0 in their options '
                              'signature (%s); this rank: %s' % (rank, bad, ', '.join('%016x' % x for x in sigs), eng.effective_options()))


class MaxReducer:
    """all-reduce(MAX) of the convergence monitors (dJ, dPops) over the ranks -- the one exchange of the sharded MALI
    loop.  NaN must win like in the single-process max (numpy max semantics, rh_method.py:706): MAX collectives do not
    define NaN ordering, so a NaN flag travels as a third element (and a singular-system flag as a fourth).

    Two entry points:
      reducer(dJ, dPops)        host values in, host values out (what the CPU tests and single-rank runs use)
      reducer.engine(engine)    the monitors never visit the host before the collective: lsx_monitors leaves them in a
                                device buffer on the engine's stream, RCCL reduces that buffer in place on the same
                                stream, and ONE device-to-host copy brings the result back (per MALI iteration)."""

    def __init__(self, device=None, group=None, stream=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.device = torch.device(device) if device is not None else torch.device('cpu')
        self.stream = stream            # torch.cuda.Stream the engine launches on (its handle went to lsx_create)
        self.buf = torch.zeros(4, dtype=torch.float64, device=self.device)
        self._dev_scratch = None
        self._pending = None            # engine_begin / engine_end: the result of an exchange that had nothing to overlap with
        self._host4 = None
        self._checked = set()           # engines whose options signature has been compared across the ranks (once each)

    def attach(self, eng):
        """Compare `eng`'s options signature across the ranks NOW -- a collective: call it at a point every rank reaches with its
        engine of the job (after the engines are made, before the loop; bench.py, response.py).  The reducer's entry points fall back
        to doing it at an engine's first exchange, which is only symmetric if every rank presents its engines in the same order."""
        if self.active and hasattr(eng, 'options_signature'):
            key = getattr(eng, 'serial', None)
            if key is None or key not in self._checked:      # (a per-engine serial, not id(): the id of a collected engine can come back)
                check_same_options(eng, self.group)
                if key is not None:
                    self._checked.add(key)
        return self

    def _check(self, eng):
        self.attach(eng)

    def __call__(self, dJ: float, dPops: float):
        if not self.active:
            return dJ, dPops
        nan = float(np.isnan(dJ) or np.isnan(dPops))
        vals = [0.0 if np.isnan(dJ) else dJ, 0.0 if np.isnan(dPops) else dPops, nan, 0.0]
        self.buf.copy_(self.torch.tensor(vals, dtype=self.torch.float64))
        self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.MAX, group=self.group)
        out = self.buf.tolist()
        if out[2] > 0:
            return float('nan'), float('nan')
        return out[0], out[1]

    def engine(self, eng):
        """-> (dJ, dPops) over all columns of all ranks for the calls enqueued on `eng` (problem.Engine).
        Raises LsxSingularError on every rank if any rank met a singular system."""
        self._check(eng)
        torch = self.torch
        on_device = eng.lib.backend.startswith('hip')
        if on_device and self.buf.is_cuda:
            ctx = torch.cuda.stream(self.stream) if self.stream is not None else _null_context()
            with ctx:
                eng.monitors_to(self.buf.data_ptr())
                if self.stream is None:
                    eng.sync()          # the engine owns its stream: order the collective behind it from the host
                if self.active:
                    self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.MAX, group=self.group)
                out = self.buf.tolist()             # the one device-to-host copy of the iteration
        else:
            if on_device:               # HIP engine, CPU collective (gloo rehearsal): stage through a device scratch
                if self._dev_scratch is None:
                    self._dev_scratch = torch.zeros(4, dtype=torch.float64, device='cuda')
                    torch.cuda.synchronize()        # the fill runs on torch's stream, lsx_monitors on the engine's: order them once
                eng.monitors_to(self._dev_scratch.data_ptr())
                eng.sync()
                self.buf.copy_(self._dev_scratch)
            else:                       # the oracle writes host memory
                host = np.zeros(4)
                eng.monitors_to(host.ctypes.data)
                self.buf.copy_(torch.from_numpy(host))
            if self.active:
                self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.MAX, group=self.group)
            out = self.buf.tolist()
        if out[3] > 0:
            from ._capi import LsxSingularError
            raise LsxSingularError(3, 'stat_equil: singular matrix on some rank (cf. LinAlgError at rh_method.py:739)')
        dJ = float('nan') if out[2] > 0 else out[0]
        return dJ, out[1]


    # ---- the same exchange split in two, for the loop without a host round trip per iteration (drivers.mali_steps /
    # iterate_mali_engine): begin enqueues reduction, all-reduce and read-back behind the iteration's kernels, the caller then
    # enqueues the next formal solution, end waits for the read-back only
    def engine_begin(self, eng):
        self._check(eng)
        torch = self.torch
        on_device = eng.lib.backend.startswith('hip')
        if on_device and self.buf.is_cuda and self.stream is not None:
            if self._host4 is None:
                self._host4 = torch.empty(4, dtype=torch.float64).pin_memory()
                self._ev = torch.cuda.Event()
            with torch.cuda.stream(self.stream):
                eng.monitors_to(self.buf.data_ptr())
                if self.active:
                    self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.MAX, group=self.group)
                self._host4.copy_(self.buf, non_blocking=True)
                self._ev.record(self.stream)
            self._pending = None
        else:                           # nothing to overlap with (CPU collective, oracle, engine-owned stream): the plain exchange now
            try:
                self._pending = ('ok', self.engine(eng))
            except Exception as e:      # raised where the plain loop would see it: at the end of the iteration
                self._pending = ('err', e)

    def engine_end(self, eng):
        if self._pending is not None:
            kind, val = self._pending
            self._pending = None
            if kind == 'err':
                raise val
            return val
        self._ev.synchronize()
        out = self._host4.tolist()
        if out[3] > 0:
            from ._capi import LsxSingularError
            raise LsxSingularError(3, 'stat_equil: singular matrix on some rank (cf. LinAlgError at rh_method.py:739)')
        dJ = float('nan') if out[2] > 0 else out[0]
        return dJ, out[1]


class _null_context:
    def __enter__(self): return self
    def __exit__(self, *a): return False


class AllDone:
    """logical AND over ranks of "every column of my shard has converged" (the per-column driver's stop test,
    drivers.iterate_mali_columns(all_done=...))"""

    def __init__(self, group=None, engine=None):
        """engine: this rank's problem.Engine -- its options signature is compared across the ranks before the first exchange
        (check_same_options)"""
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        dev = 'cuda' if self.active and dist.get_backend(group) == 'nccl' else 'cpu'
        self.buf = torch.zeros(1, dtype=torch.int32, device=dev)
        if self.active and engine is not None:
            check_same_options(engine, group)

    def __call__(self, done: bool) -> bool:
        if not self.active:
            return done
        self.buf.fill_(1 if done else 0)
        self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.MIN, group=self.group)
        return bool(self.buf.item())
