"""lightspinner_amd -- MI355X-native MALI formal-solution engine behind
Lightspinner's Context.formal_sol_gamma_matrices()/stat_equil() API."""
__version__ = '0.1.0'

import os as _os

# A context with more than three tile classes launches them on more streams than the HIP runtime has hardware queues by default
# (4): the streams that share a queue wait for each other, and the small classes of FALC Ca + H then start only when a large one
# has drained (1250 columns: 5.35 -> 5.20 ms per iteration with 8 queues, interleaved on one box; profiles/r03/ab_hardware_queues.txt).
# The runtime reads this when it initialises, so it only takes effect if the package is imported before the first HIP call of
# the process; a value the caller has set is left alone.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

from .problem import Problem, Transition, ColumnBlock, Engine  # noqa: F401
