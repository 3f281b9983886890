"""lightspinner_amd -- MI355X-native MALI formal-solution engine behind
Lightspinner's Context.formal_sol_gamma_matrices()/stat_equil() API."""
__version__ = '0.1.0'

from .problem import Problem, Transition, ColumnBlock, Engine  # noqa: F401
