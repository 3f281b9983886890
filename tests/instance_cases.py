"""The ledger of compiled sweep instances: every template instance of the sweep kernels that the
product library compiles and can dispatch to -- lsx_sweep_kernel<NPT, NL, 5, SCAL, LK, TOPO> (one ray per lane),
lsx_sweep_rs_kernel<NPT, NL, LK, TOPO> (ray-serial) and the parabolic rule's sweep_tile_par<...> -- must be reached by a GPU
parity test against the oracle.  This module is the shared data of the two sides:

  tests/test_instance_ledger.py (CPU)   walks the instance lists of lsx_plan.h (lsx_plan_instances) and the plans of the
                                        problems below (lsx_plan_probe) and fails if an instance is planned by none of them;
  tests/test_instances_gpu.py   (GPU)   runs every problem below on the one-ray-per-lane kernels, the ray-serial kernels and the
                                        parabolic rule against the oracle and asserts that the expected instances did run.

The topologies follow the reference's own multi-stage model atoms (rh_atoms.py:194 C_atom, :355 Fe_simple_atom: lines of the
upper ionisation stage start on the level a lower stage's continua end on, rh_method.py:606-627, 654-681); nothing else here
comes from the reference.  An atom: (Nlevel, [(kind, lower, upper, blue end, red end of its range as fractions of the spectrum)])."""
from toy import spec_problem


def decode(code):
    """lsx_class_code -> (per-ray slots, lines, linked, two-line relation); generic instances: (-1, 0, linked, 0)"""
    if code < 0:
        return (-1, 0, 1 if code == -3 else 0, 0)
    topo, r = divmod(code, 128)
    lk, r = divmod(r, 64)
    npt, nl = divmod(r, 8)
    return (npt, nl, lk, topo)


# ---- topologies ---------------------------------------------------------------------------------------------------------------------------------
# two ionisation stages: levels 0, 1 (stage I), 2, 3, 4 (stage II), 5 (stage III).  The stage-II resonance lines 2 -> 3, 2 -> 4 start
# on the level the stage-I continua 0 -> 2, 1 -> 2 end on: those continua go through the sweep as per-ray slots (the line's
# Gamma integrand needs atom.U[2] ray by ray), and with them every continuum of the atom in the tile (2 -> 5)
STAGES_1 = [(6, [('l', 2, 3, 0.06, 0.50),
                 ('c', 0, 2, 0.00, 0.62), ('c', 1, 2, 0.22, 0.40), ('c', 2, 5, 0.31, 0.40)])]        # (2,1) (3,1) (4,1)
STAGES_2 = [(6, [('l', 2, 3, 0.06, 0.56), ('l', 2, 4, 0.18, 0.52), ('l', 0, 1, 0.70, 0.80),
                 ('c', 0, 2, 0.00, 0.62), ('c', 1, 2, 0.30, 0.44)])]                                   # (2,1) (3,2) (4,2)
STAGES_3 = [(6, [('l', 2, 3, 0.06, 0.56), ('l', 2, 4, 0.12, 0.50), ('l', 3, 5, 0.20, 0.44),
                 ('c', 0, 2, 0.00, 0.62)])]                                                            # (3,2) (4,3)
# lines only: a multiplet to a common upper level (relation 0), three and four overlapping lines
MULTI_UP = [(5, [('l', 0, 4, 0.08, 0.52), ('l', 1, 4, 0.20, 0.64), ('l', 2, 4, 0.32, 0.74), ('l', 3, 4, 0.40, 0.48)])]   # (1,1) (2,2,0) (3,3) (4,4)
# two atoms whose lines overlap (relation 2: unrelated lines), one of them under a bound-free continuum of its own atom that no line
# touches (a linked continuum)
TWO_ATOMS = [(3, [('l', 0, 1, 0.10, 0.52), ('c', 0, 2, 0.00, 0.42)]),
             (3, [('l', 0, 1, 0.28, 0.74)])]                                                           # (1,1,LK) (2,2,LK,2) (2,2,2) (1,1)
# two lines to a common upper level under a linked continuum of the same atom (relation 0 with linked continua); and the same
# with a common LOWER level (relation 1, what H Lyman / Ca II H & K are)
LINKED_UP = [(4, [('l', 0, 2, 0.10, 0.52), ('l', 1, 2, 0.28, 0.74), ('c', 0, 3, 0.00, 0.62)])]        # (1,1,LK) (2,2,LK,0)
LINKED_LOW = [(4, [('l', 0, 1, 0.10, 0.52), ('l', 0, 2, 0.28, 0.74), ('c', 1, 3, 0.00, 0.62)])]       # (1,1,LK) (2,2,LK,1)
# atom X: a line starting on the level its continuum ends on (per-ray continuum); atom Y: a line under a linked continuum
MIXED = [(3, [('l', 1, 2, 0.16, 0.60), ('c', 0, 1, 0.00, 0.50)]),
         (3, [('l', 0, 1, 0.24, 0.70), ('c', 0, 2, 0.00, 0.46)])]                                     # (2,1) (3,2,LK)
# three lines from a common lower level under linked continua
TRIPLET = [(6, [('l', 0, 1, 0.30, 0.60), ('l', 0, 2, 0.34, 0.64), ('l', 0, 3, 0.38, 0.68), ('c', 0, 5, 0.00, 0.95), ('c', 1, 5, 0.00, 0.55)])]   # (3,3,LK)
# more overlapping per-ray slots than any instance has, with and without linked continua
CROWD = [(7, [('l', 0, u, 0.20 + 0.02 * u, 0.60 + 0.02 * u) for u in range(1, 6)] + [('c', 0, 6, 0.00, 0.50)])]          # generic, generic linked

TOPOLOGIES = dict(stages1=STAGES_1, stages2=STAGES_2, stages3=STAGES_3, multi_up=MULTI_UP, two_atoms=TWO_ATOMS, linked_up=LINKED_UP,
                  linked_low=LINKED_LOW, mixed=MIXED, triplet=TRIPLET, crowd=CROWD)

# (topology, ncol, Nspace, phi_compact): >= 32 columns so that the per-class launches run (one kernel instance per tile class);
# ragged column groups for the five-column wavefronts of the ray-serial kernel; odd and even depth counts
CASES = [
    ('stages1', 33, 31, False), ('stages2', 34, 40, False), ('stages3', 36, 33, False), ('multi_up', 37, 38, False),
    ('two_atoms', 33, 35, False), ('linked_up', 34, 36, False), ('linked_low', 35, 29, True), ('mixed', 33, 34, False),
    ('triplet', 32, 30, False), ('crowd', 33, 27, False),
]
NSPECT = 260


def build(name, ncol, Nspace, phi_compact, seed=None):
    return spec_problem(TOPOLOGIES[name], seed=100 + list(TOPOLOGIES).index(name) if seed is None else seed, Nspace=Nspace, Nrays=5,
                        Nspect=NSPECT, ncol=ncol, phi_compact=phi_compact)


# instances that only the reference's own problems (FALC CaII / Ca+H: tests/test_production_classes.py) or the LSX_NO_LINKED switch
# reach are listed there; what each case here is EXPECTED to plan (checked on the CPU against the plan, on the GPU against what ran)
EXPECT = {
    'stages1': [(2, 1, 0, 0), (3, 1, 0, 0), (4, 1, 0, 0)],
    'stages2': [(2, 1, 0, 0), (3, 2, 0, 0), (4, 2, 0, 0)],
    'stages3': [(3, 2, 0, 0), (4, 3, 0, 0)],
    'multi_up': [(1, 1, 0, 0), (2, 2, 0, 0), (3, 3, 0, 0), (4, 4, 0, 0)],
    'two_atoms': [(1, 1, 1, 0), (2, 2, 1, 2), (2, 2, 0, 2), (1, 1, 0, 0)],
    'linked_up': [(1, 1, 1, 0), (2, 2, 1, 0)],
    'linked_low': [(1, 1, 1, 0), (2, 2, 1, 1)],
    'mixed': [(2, 1, 0, 0), (3, 2, 1, 0)],
    'triplet': [(3, 3, 1, 0)],
    'crowd': [(-1, 0, 0, 0), (-1, 0, 1, 0)],
}
# the classes that have a ray-serial instance of the parabolic rule (lsx_plan.h, LSX_RSP_INSTANCES; the CPU ledger checks the two lists
# against each other): mode 'parabolic-serial' of tests/test_instances_gpu.py expects exactly these on the ray-serial kernel
PARABOLIC_SERIAL = [(0, 0, 0, 0), (1, 1, 0, 0), (1, 1, 1, 0), (2, 2, 0, 1), (2, 2, 0, 2)]      # (round 5: the two-line classes with a known relation)
