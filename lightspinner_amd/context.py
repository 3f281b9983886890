"""BatchContext: the MALI engine over many independent atmosphere columns.

Same two verbs as rh_method.Context (formal_sol_gamma_matrices / stat_equil,
rh_method.py:565,710), results with a leading column index.  The reference solves
one column per Context; columns are independent 1-D problems, so a batch is just
many Contexts side by side on one GPU."""
import numpy as np

from . import _capi
from .problem import Problem, ColumnBlock, Engine


class BatchContext:
    def __init__(self, problem: Problem, block: ColumnBlock, device: int = 0, stream=None, lib=None,
                 upload_chunk: int = 256):
        self.problem = problem
        self.ncol = block.ncol
        self.engine = Engine(problem, self.ncol, device=device, stream=stream, lib=lib)
        for c0 in range(0, self.ncol, upload_chunk):
            c1 = min(self.ncol, c0 + upload_chunk)
            self.engine.set_columns(c0, block.slice(c0, c1))
        self.dJ = None
        self.dPops = None

    # -- the two verbs ---------------------------------------------------------
    def formal_sol_gamma_matrices(self) -> float:
        """max over columns of the reference's dJ (rh_method.py:705-708)"""
        self.dJ = self.engine.formal_sol_gamma()
        return self.dJ

    def stat_equil(self) -> float:
        """max over columns of the reference's maxRelChange (rh_method.py:741-745)"""
        self.dPops = self.engine.stat_equil()
        return self.dPops

    # -- results (fetched on demand, reference layouts with leading [ncol]) -----
    @property
    def I(self): return self.engine.get(_capi.LSX_I)
    @property
    def J(self): return self.engine.get(_capi.LSX_J)
    @property
    def n(self): return self.engine.get(_capi.LSX_N)
    @property
    def Gamma(self): return self.engine.get(_capi.LSX_GAMMA)
    @property
    def dJ_columns(self): return self.engine.get(_capi.LSX_DJ_COL)
    @property
    def dPops_columns(self): return self.engine.get(_capi.LSX_DPOPS_COL)

    def set_pops(self, n, col0=0):
        """warm start (response_fn.py:33)"""
        self.engine.set(_capi.LSX_N, n, col0)

    def close(self):
        self.engine.close()
