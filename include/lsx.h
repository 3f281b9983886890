/* lsx.h -- C ABI of the MALI formal-solution engine ("lsx").
 *
 * This is the drop-in boundary for Lightspinner's hot path.  The reference has no
 * FFI layer: its boundary is the Python class API of rh_method.Context
 *   Context.__init__                     /root/reference/rh_method.py:531-563
 *   Context.formal_sol_gamma_matrices    /root/reference/rh_method.py:565-708
 *   Context.stat_equil                   /root/reference/rh_method.py:710-745
 * and, one level down, formal_solver.piecewise_linear_1d
 *                                        /root/reference/formal_solver.py:144-212.
 * The entry points below are what a ctypes binding for exactly those four call
 * sites needs (see INTEGRATION.md for the reference-side stub).  Plain pointers and
 * sizes only; all floating point is IEEE float64; all arrays are C-contiguous with
 * the index order written in brackets (last index fastest).
 *
 * Two shared libraries export this identical ABI:
 *   lightspinner_amd/csrc/liblsx_hip.so   the product: HIP kernels for gfx950
 *   oracle/liblsx_oracle.so               TEST INFRASTRUCTURE ONLY: scalar C
 *                                         restatement of the reference (checker and
 *                                         timed CPU baseline, never the product path)
 *
 * Return value of every int function: 0 = ok, otherwise an LSX_E* code;
 * lsx_last_error() then returns a human-readable message (thread-local).
 * A context owns one device + one stream; contexts are independent; a single
 * context is not thread-safe (same as the reference, rh_method.py:432-436).
 */
#ifndef LSX_H
#define LSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSX_ABI_VERSION 1

enum {
    LSX_OK = 0,
    LSX_EINVAL = 1,      /* bad argument / inconsistent descriptor                  */
    LSX_EDEVICE = 2,     /* HIP runtime error (message carries hipGetErrorString)    */
    LSX_ESINGULAR = 3,   /* singular statistical-equilibrium system (cf. LinAlgError,
                            rh_method.py:739); populations of that (col,k) untouched */
    /* 4 is unused: non-finite values are not an error, they propagate through dJ exactly as in
       the reference (numpy max, rh_method.py:705-706)                                            */
    LSX_EUNSUPPORTED = 5 /* valid request this build cannot run (e.g. Nrays > 8)     */
};

/* One radiative transition as rh_method.ComputationalTransition sees it
 * (rh_method.py:93-131).  Order of the table = order of
 * [atom.trans for atom in ctx.activeAtoms]: atoms by ascending atomic weight
 * (atomic_set.py:269-273), within an atom lines first then continua
 * (rh_method.py:399-405). */
typedef struct lsx_transition {
    int32_t atom;      /* index into the active-atom list                             */
    int32_t is_line;   /* 1 = bound-bound (VoigtLine), 0 = bound-free continuum       */
    int32_t i, j;      /* lower / upper level index inside the atom                   */
    int32_t Nblue;     /* index of the transition's bluest point in the global grid   */
    int32_t Nlambda;   /* number of points of its local grid (a slice of the global)  */
    double Aji, Bji, Bij, lambda0; /* lines only (lambda0 in nm), 0 for continua      */
} lsx_transition;

/* Column-independent problem description (what Context.__init__ derives from
 * spect / the atomic models). */
typedef struct lsx_problem {
    int32_t abi_version;        /* = LSX_ABI_VERSION                                   */
    int32_t Nspace, Nrays, Nspect, Natoms, Ntrans;
    const int32_t* Nlevel;      /* [Natoms]                                            */
    const double* wavelength;   /* [Nspect] nm, ascending (spect.wavelength)           */
    const double* muz;          /* [Nrays]  (atmos.muz)                                */
    const double* wmu;          /* [Nrays]  (atmos.wmu)                                */
    const lsx_transition* trans;/* [Ntrans]                                            */
    const uint8_t* active;      /* [Ntrans][Nspect] t.active (rh_method.py:124-127)    */
    const double* alpha;        /* continuum cross-sections, concatenated over the
                                   continua in table order, Nlambda entries each       */
    int32_t sca_per_lambda;     /* 0: bg_sca is [Nspace] per column (Thomson only, as
                                   background.py:45-47 produces); 1: [Nspect][Nspace]  */
    int32_t phi_compact;        /* 0: phi is [Nl][Nrays][2][Nspace] per line as
                                   rh_method.py:224; 1: phi is [Nl][Nspace] (valid when
                                   vlos == 0 so the profile is ray independent)        */
} lsx_problem;

/* Per-column hot-path inputs for `ncol` consecutive columns; every pointer has a
 * leading [ncol] index.  NLtot = sum Nlevel, NL2tot = sum Nlevel^2,
 * Nlines = number of line transitions, SNl = sum of Nlambda over lines
 * (both in table order). */
typedef struct lsx_columns {
    const double* height;       /* [ncol][Nspace] m            atmos.height            */
    const double* temperature;  /* [ncol][Nspace] K            atmos.temperature       */
    const double* nStar;        /* [ncol][NLtot][Nspace]       atom.nStar, atoms concatenated */
    const double* nTotal;       /* [ncol][Natoms][Nspace]      atom.nTotal             */
    const double* n;            /* [ncol][NLtot][Nspace]       initial atom.n (warm start,
                                                               rh_method.py:412-416)   */
    const double* C;            /* [ncol][NL2tot][Nspace]      atom.C[to][from][k] after
                                                               compute_collisions incl. the
                                                               clamp (rh_method.py:474-487) */
    const double* bg_chi;       /* [ncol][Nspect][Nspace]      background.chi          */
    const double* bg_eta;       /* [ncol][Nspect][Nspace]      background.eta          */
    const double* bg_sca;       /* [ncol][Nspace] or [ncol][Nspect][Nspace]            */
    const double* phi;          /* [ncol][SNl][Nrays][2][Nspace] or [ncol][SNl][Nspace];
                                   phi and wphi may both be NULL when the profiles of these
                                   columns are set with lsx_set_line_profiles afterwards      */
    const double* wphi;         /* [ncol][Nlines][Nspace]      t.wphi                  */
} lsx_columns;

typedef struct lsx_ctx lsx_ctx;

/* What lsx_get / lsx_set address.  Shapes per column (leading [ncol] implied): */
enum {
    LSX_I = 0,         /* [Nspect][Nrays]   ctx.I  (emergent, rh_method.py:638)       */
    LSX_J = 1,         /* [Nspect][Nspace]  ctx.J                                     */
    LSX_N = 2,         /* [NLtot][Nspace]   atom.n                                    */
    LSX_GAMMA = 3,     /* [NL2tot][Nspace]  atom.Gamma[i][j][k], atoms concatenated   */
    LSX_DJ_COL = 4,    /* [1]  per-column max|1-Jdag/J| of the last FS call           */
    LSX_DPOPS_COL = 5, /* [1]  per-column max rel. population change of the last SE   */
                       /*      (valid from that lsx_stat_equil until the next lsx_formal_sol_gamma, whose epilogue clears it) */
    LSX_NSTAR = 6,     /* [NLtot][Nspace]                                             */
    LSX_C = 7,         /* [NL2tot][Nspace]                                            */
    /* 8, 9: not assigned.  t.Rij / t.Rji (rh_method.py:691-692) are write-only state of the reference -- accumulated
       over calls without ever being zeroed or read -- and are not part of this interface */
    LSX_PHI = 10,      /* [SNl][Nrays][2][Nspace] (or [SNl][Nspace] when phi_compact)  t.phi  */
    LSX_WPHI = 11,     /* [Nlines][Nspace]  t.wphi                                     */
    LSX_VBROAD = 12,   /* [Natoms][Nspace]  atom.vBroad            (after lsx_set_atmosphere)  */
    LSX_ADAMP = 13     /* [Nlines][Nspace]  line.damping(...)[0]   (after lsx_set_atmosphere)  */
};

/* Create a context for `ncol` columns on HIP device `device` (ignored by the
 * oracle).  `stream` is a hipStream_t to launch on (e.g. torch's current stream),
 * or NULL for a stream owned by the context.  The HIP backend takes at most 65535 columns per
 * context (LSX_EUNSUPPORTED beyond; use several contexts). */
int lsx_create(const lsx_problem* desc, int32_t ncol, int32_t device, void* stream,
               lsx_ctx** out);
void lsx_destroy(lsx_ctx* ctx);

/* Upload (copy) inputs for columns [col0, col0+ncol).  J is reset to 0 for those
 * columns (rh_method.py:562).  Host pointers. */
int lsx_set_columns(lsx_ctx* ctx, int32_t col0, int32_t ncol, const lsx_columns* cols);

/* ComputationalTransition.compute_phi (rh_method.py:198-243) for every line of columns
 * [col0, col0+ncol), evaluated by the library instead of being handed over:
 *   v = (lambda - lambda0) c / (vBroad lambda0);  vk = v -+ muz vlos / vBroad  (- down, + up; :231-238)
 *   phi = H(aDamp, vk) / (sqrt(pi) vBroad)        (:239, H = Re w(vk + i aDamp), utils.py:13-15)
 *   wphi = 1 / sum_{lambda, mu, dir} phi wlambda (wmu/2)                      (:236-242)
 * aDamp:  [ncol][Nlines][Nspace]  line.damping(atmos, vBroad, hGround)[0]     (:223)
 * vBroad: [ncol][Natoms][Nspace]  atom.vBroad                                 (:409)
 * vlos:   [ncol][Nspace] m/s, or NULL for 0.  A context created with phi_compact = 1 accepts
 *         only vlos == NULL (the profile is then ray independent).
 * The Voigt function is the library's own (exponentially convergent trapezoid sum with pole
 * correction, relative error < 1e-13 against scipy.special.wofz for 1e-4 <= aDamp). */
int lsx_set_line_profiles(lsx_ctx* ctx, int32_t col0, int32_t ncol, const double* aDamp,
                          const double* vBroad, const double* vlos);

/* ---- set-up chain on the device: what Context.__init__ derives from the atmosphere ---------------------
 * Atomic data as the reference's model classes hold it (atomic_model.py): given once per context, column independent. */
typedef struct lsx_level {           /* atomic_model.AtomicLevel (:111-137) */
    double E_SI;                     /* level energy [J] (E_SI, :132-134)             */
    double g;                        /* statistical weight                            */
    int32_t stage;                   /* ionisation stage                              */
    int32_t reserved;
} lsx_level;

typedef struct lsx_line_model {      /* atomic_model.VoigtLine (:250-502), broadening part */
    int32_t i, j;                    /* levels; must equal the transition table's line of the same rank */
    double gRad;                     /* radiative damping [1/s]                       */
    double stark;                    /* > 0: quadratic Stark coefficient, < 0: -stark * ne, 0: none (:318-345) */
    int32_t vdw_kind;                /* 0 none, 1 Unsold (:166-198)                   */
    int32_t reserved;
    double vdw[2];                   /* VdwUnsold.vals: H and He enhancement factors  */
} lsx_line_model;

enum { LSX_COLL_OMEGA = 0, LSX_COLL_CI = 1, LSX_COLL_CE = 2 };   /* collisional_rates.py:21-96 */
typedef struct lsx_collision {
    int32_t kind;                    /* LSX_COLL_*                                    */
    int32_t i, j;                    /* levels, i < j                                 */
    int32_t nT;                      /* points of the temperature grid (2, or >= 4 for the not-a-knot cubic scipy's
                                        interp1d(kind=3) builds, collisional_rates.py:15-19)             */
    const double* temperature;       /* [nT] K, ascending                             */
    const double* rates;             /* [nT]                                          */
} lsx_collision;

typedef struct lsx_atom_model {
    double weight;                   /* atomic weight [amu] (atomicTable[name].weight)                */
    int32_t is_hydrogen;             /* linear Stark broadening applies (atomic_model.py:343-344)     */
    int32_t Nlevel;
    const lsx_level* levels;         /* [Nlevel]                                      */
    int32_t Nline;                   /* lines of this atom in the transition table    */
    int32_t Ncollision;
    const lsx_line_model* lines;     /* [Nline], in the table's order                 */
    const lsx_collision* collisions; /* [Ncollision], in the model's order            */
} lsx_atom_model;

typedef struct lsx_atomic_data {
    int32_t Natoms;                  /* = lsx_problem.Natoms, same order              */
    int32_t reserved;
    const lsx_atom_model* atoms;
    double weight_H, weight_He, abundance_He;   /* atomic table entries the Unsold cross-section needs (:183-190) */
} lsx_atomic_data;

int lsx_set_atomic_data(lsx_ctx* ctx, const lsx_atomic_data* data);

/* Atmosphere of columns [col0, col0 + ncol): every array [ncol][Nspace] unless noted, SI units as after
 * Atmosphere.nondimensionalise (atmosphere.py:424-436).  From it the library derives, per depth,
 *   vBroad = sqrt(2 k T / (amu A) + vturb^2)                                  atomic_model.py:66-69
 *   aDamp  = (gRad + Q_vdW + Q_Stark) lambda0 / (4 pi vBroad)                  :491-502, 166-198, 300-345
 *   phi, wphi (as lsx_set_line_profiles)                                       rh_method.py:198-243
 *   nStar  = Saha-Boltzmann with Debye lowering (if lte_pops != 0; n := nStar)  atomic_set.py:105-145
 *   C      = collisional rates, negatives clamped to 0                          collisional_rates.py:21-96, rh_method.py:474-487
 * and stores temperature and nTotal.  Needs lsx_set_atomic_data before, and lsx_set_columns for the same columns
 * (geometry, background, and -- when lte_pops == 0 -- nStar and n). */
typedef struct lsx_atmosphere {
    const double* temperature;       /* K                                             */
    const double* ne;                /* m^-3                                          */
    const double* vturb;             /* m/s                                           */
    const double* vlos;              /* m/s, or NULL for 0                            */
    const double* nHGround;          /* m^-3: eqPops['H'].n[0] (rh_method.py:223)     */
    const double* nTotal;            /* [ncol][Natoms][Nspace]                        */
    int32_t lte_pops;
    int32_t reserved;
} lsx_atmosphere;

int lsx_set_atmosphere(lsx_ctx* ctx, int32_t col0, int32_t ncol, const lsx_atmosphere* atm);

/* One Context.formal_sol_gamma_matrices() over all columns.  *dJ_max receives the
 * maximum over columns of the reference's return value (rh_method.py:705-708). */
int lsx_formal_sol_gamma(lsx_ctx* ctx, double* dJ_max);

/* One Context.stat_equil() over all columns; *dPops_max = max over columns
 * (rh_method.py:741-745).  Updates the populations held by the context. */
int lsx_stat_equil(lsx_ctx* ctx, double* dPops_max);

/* Asynchronous forms: enqueue only; lsx_sync waits and returns both maxima of the
 * most recent calls (either pointer may be NULL). */
int lsx_formal_sol_gamma_async(lsx_ctx* ctx);
int lsx_stat_equil_async(lsx_ctx* ctx);
/* lsx_sync returns when the monitors of the calls enqueued so far have been read back -- which implies that those calls have
 * finished.  The HIP library may leave work of its own running behind that read-back (it rebuilds the ray-serial sweeps' operand
 * table there after a stat_equil, while the host decides on the next iteration): every later call, lsx_get included, is ordered behind
 * it on the context's stream. */
int lsx_sync(lsx_ctx* ctx, double* dJ_max, double* dPops_max);

/* ---- the MALI loop without a host round trip per iteration (test.py:20-29: `while dJ > 2e-3 or dPops > 1e-3`) ----------
 * The reference decides after every iteration, on the host, whether to run another one.  Waiting for that decision leaves the
 * GPU idle between iterations (and, over several GPUs, for the all-reduce of the monitors).  These entries let the caller
 * enqueue the NEXT iteration's formal solution before the decision is known and take it back if the loop has ended:
 *
 *   lsx_formal_sol_gamma_async; [lsx_stat_equil_async]; lsx_sync_begin            -- iteration 1
 *   repeat:  lsx_formal_sol_gamma_speculative                                      -- iteration i + 1, ahead of the decision
 *            lsx_sync_end(&dJ, &dPops)                                             -- monitors of iteration i
 *            converged?  lsx_discard_formal_sol; stop                              -- I, J, Gamma, monitors are iteration i's again
 *            else        [lsx_stat_equil_async]; lsx_sync_begin                    -- iteration i + 1 goes on
 *
 * The results are those of the plain loop, bit for bit (tests/test_pipelined_loop.py).
 * lsx_sync_begin: enqueue the read-back of the monitors of the calls enqueued so far (what lsx_sync would return); one at a time.
 * lsx_sync_end: wait for THAT read-back only -- work enqueued behind it keeps running -- and return the maxima; LSX_ESINGULAR as
 *   lsx_sync.  Without a pending lsx_sync_begin it is lsx_sync.
 * lsx_formal_sol_gamma_speculative: lsx_formal_sol_gamma_async whose I, Gamma and monitors go to a second set of buffers (J is a
 *   pair already), so that the previous call's remain available.  LSX_EUNSUPPORTED while columns are frozen
 *   (lsx_set_active_columns).
 * lsx_discard_formal_sol: undo the last call if it was speculative and nothing has been built on it (no lsx_stat_equil, no new
 *   inputs since): lsx_get, lsx_sync and the next calls see the previous formal solution (an earlier call whose monitors nobody
 *   has read is pending again).  LSX_EINVAL otherwise -- also while a read-back begun AFTER the speculative call is in flight
 *   (lsx_sync_begin; collect it with lsx_sync_end first).
 * Monitors are not lost between calls: a formal solution enqueued while a statistical equilibrium's maxima and singular flag
 *   have not been read back (`FS; SE; FS; lsx_sync`, `SE_async; lsx_formal_sol_gamma(&dJ)`) leaves them in place. */
int lsx_sync_begin(lsx_ctx* ctx);
int lsx_sync_end(lsx_ctx* ctx, double* dJ_max, double* dPops_max);
/* lsx_sync_begin that also brings the POPULATIONS back (round 6): behind the monitors the read-back copies `n` of every column into a
 * host buffer of the context; once lsx_sync_end has returned, lsx_fetch_populations copies it out ([ncol][NLtot][Nspace], the layout of
 * lsx_get(LSX_N); nbytes must match) -- host memory to host memory, whatever has been enqueued behind the read-back.
 * What it is for: Context.stat_equil() must leave the new populations in the CALLER's array (atom.n is eqPops[name].pops,
 * rh_method.py:412-416, 736-741; read by response_fn.py:62) before it returns.  With lsx_get that copy would wait for everything
 * enqueued so far -- including the next iteration's speculative formal solution; with this pair the drop-in Context enqueues that
 * formal solution behind the read-back and the GPU goes from stat_equil into it while the host returns to the driver's loop
 * (lightspinner_amd/rh_method.py).  lsx_fetch_populations without such a read-back collected: LSX_EINVAL. */
int lsx_sync_begin_populations(lsx_ctx* ctx);
int lsx_fetch_populations(lsx_ctx* ctx, double* dst, size_t nbytes);
int lsx_formal_sol_gamma_speculative(lsx_ctx* ctx);
int lsx_discard_formal_sol(lsx_ctx* ctx);
/* 1 if enqueueing ahead pays for this context, else 0.  HIP library: 1 for contexts whose formal solution is one launch chain
 * on the context's own stream (fewer than 32 columns: the latency-bound case, a FALC column: 4.37 -> 3.91 ms for its 46
 * iterations).  Larger contexts fork their tile classes onto several streams; enqueued ahead, the fork is a wait on a PENDING
 * event across hardware queues, which costs what the host round trip it replaces costs (29 against 28 us from the read-back to
 * the first sweep, 1000 CaII columns: 1.10 ms per iteration either way; with the default system-scope events it was 1.44), so 0.
 * The oracle computes synchronously: 0. */
int lsx_prefers_lookahead(lsx_ctx* ctx);

/* The convergence monitors of the most recent (enqueued) calls, reduced over this context's columns and left where the
 * caller's collective can take them without a host round trip: dst[0] = max dJ, dst[1] = max dPops, dst[2] = 1 if dJ is
 * NaN anywhere else 0 (MAX collectives do not order NaN; with the flag set dst[0] holds the maximum of the other
 * columns), dst[3] = 1 if a statistical-equilibrium system was singular else 0.  For the HIP library dst is a DEVICE
 * pointer to 4 doubles and the reduction is enqueued on the context's stream (hand dst to the all-reduce on that stream,
 * read it once afterwards); for the oracle it is host memory.  This is the per-iteration exchange of a multi-GPU MALI
 * loop (test.py:23 over all ranks: one all-reduce(MAX) of these four numbers). */
int lsx_monitors(lsx_ctx* ctx, double* dst);

/* Copy results for columns [col0, col0+ncol) to host memory (reference layouts). */
int lsx_get(lsx_ctx* ctx, int32_t what, int32_t col0, int32_t ncol, double* dst,
            size_t nbytes);
/* Overwrite LSX_N (warm start, response_fn.py:33) or LSX_J for a column range. */
int lsx_set(lsx_ctx* ctx, int32_t what, int32_t col0, int32_t ncol, const double* src,
            size_t nbytes);

/* Per-column convergence: columns whose byte is 0 are frozen -- the following
 * formal_sol_gamma / stat_equil calls neither read nor write them, and their DJ_COL / DPOPS_COL
 * read 0.  This is how a batch reproduces what the reference does when it runs one Context per
 * column until `dJ <= 2e-3 and dPops <= 1e-3` (test.py:23, response_fn.py:15): every column
 * performs exactly the iterations it would perform alone.  active == NULL: all columns active. */
int lsx_set_active_columns(lsx_ctx* ctx, const uint8_t* active);

/* formal_solver.piecewise_linear_1d for `nray` independent rays sharing one depth
 * grid (formal_solver.py:144-212).  chi, S: [nray][Nspace]; mu, wav: [nray];
 * to_obs: [nray] (1 = up-going/toFrom True); temperature: [Nspace] (only the last
 * two entries are used, formal_solver.py:206).  Out: I, PsiStar [nray][Nspace]. */
int lsx_piecewise_linear_1d(int32_t device, int32_t nray, int32_t Nspace,
                            const double* height, const double* temperature,
                            const double* mu, const int32_t* to_obs, const double* wav,
                            const double* chi, const double* S, double* I, double* PsiStar);

/* formal_solver.piecewise_1d_impl (formal_solver.py:46-142) for `nray` independent rays sharing one
 * depth grid: the short-characteristics recurrence itself, with the incident intensity handed over
 * (Istart: [nray]) instead of derived from the boundary condition.  Same arrays as above. */
int lsx_piecewise_1d_impl(int32_t device, int32_t nray, int32_t Nspace, const double* height,
                          const double* mu, const int32_t* to_obs, const double* Istart,
                          const double* chi, const double* S, double* I, double* PsiStar);

/* formal_solver.w2 (formal_solver.py:14-44) for n optical-depth steps, evaluated by the same
 * device function the sweep kernel uses.  w0w1: [n][2]. */
int lsx_w2(int32_t device, int32_t n, const double* dtau, double* w0w1);

/* ---- wavelength grid and active set: RadiativeSet.compute_wavelength_grid (atomic_set.py:377-455) -------------------
 * Host-side construction of what lsx_problem takes (wavelength, Nblue, Nlambda, active) from the transitions' own grids,
 * so a caller can hand over an arbitrary set of atoms instead of a pre-built transition table.  All wavelengths in nm. */
typedef struct lsx_trans_grid {
    int32_t is_line;
    int32_t n;                       /* points of the transition's own grid, >= 1                                  */
    const double* wavelength;        /* [n] ascending: line.wavelength (atomic_model.py:347-380) or cont.wavelength
                                        (:585-597, 645-660) BEFORE the merge                                       */
    double lambdaEdge;               /* continua: the edge (atomic_model.py:575-577); lines: ignored               */
} lsx_trans_grid;

/* grid = unique(sort(extra U {lambdaReference} U line grids U continuum edges U continuum points <= edge));
 * blueIdx[kr] = searchsorted(grid, wavelength_kr[0]); redIdx[kr] = searchsorted(grid, wavelength_kr[-1]) + 1, walked
 * down for a continuum while grid[redIdx - 1] > lambdaEdge (atomic_set.py:401-416).  The transition's grid after the
 * merge is wavelength[blueIdx : redIdx] (Nlambda = redIdx - blueIdx).  extra may be NULL (Nextra = 0).
 * wavelength: [capacity]; LSX_EINVAL if the merged grid needs more (then *Nspect holds the size needed). */
int lsx_wavelength_grid(int32_t Ntrans, const lsx_trans_grid* trans, int32_t Nextra, const double* extra,
                        double lambdaReference, int32_t capacity, double* wavelength, int32_t* Nspect,
                        int32_t* blueIdx, int32_t* redIdx);

/* active[kr][la] = blueIdx[kr] <= la < redIdx[kr]: the membership table of spect.activeSet (atomic_set.py:418-453) in the
 * layout of lsx_problem.active */
int lsx_active_set(int32_t Ntrans, int32_t Nspect, const int32_t* blueIdx, const int32_t* redIdx, uint8_t* active);

/* VoigtLine.setup_wavelength (atomic_model.py:347-380): the line's own grid, 2 (NlambdaGen / 2 rounded as the reference
 * does) + 1 points, symmetric about lambda0, linear in the core and logarithmic in the wing.  *n receives the count. */
int lsx_line_wavelength(double lambda0, double qCore, double qWing, int32_t NlambdaGen, int32_t capacity,
                        double* wavelength, int32_t* n);

/* Continuum cross-section on a new grid: compute_alpha (atomic_model.py:606-612 explicit: not-a-knot cubic through the
 * tabulated points, zero outside [minLambda, lambdaEdge], linear interpolation instead if the cubic goes negative anywhere;
 * :662-671 hydrogenic: alpha0 gbf / gbf0 (lambda / lambda0)^3 with Seaton's Gaunt factor, utils.py:24-32). */
typedef struct lsx_continuum_model {
    int32_t hydrogenic;
    int32_t n;                       /* explicit: tabulated points (>= 4 for the cubic)                            */
    const double* wavelength;        /* explicit: [n] ascending                                                    */
    const double* alpha;             /* explicit: [n]                                                              */
    double lambdaEdge, minLambda;
    double alpha0;                   /* hydrogenic: cross-section at the edge                                      */
    double E_i, E_j;                 /* hydrogenic: level energies [J]                                             */
    int32_t stage_j;                 /* hydrogenic: ionisation stage of the upper level (charge Z)                 */
    int32_t reserved;
} lsx_continuum_model;
int lsx_continuum_alpha(const lsx_continuum_model* cont, int32_t n, const double* wavelength, double* alpha);

/* ---- higher-order formal solver: monotonic piecewise-parabolic short characteristics ----------------
 * The extension README.md:19 of the reference names (Auer & Paletou 1994, A&A 285, 675); the reference itself has no such
 * routine, so there is nothing of the reference to pin this on (parity unpinned: property tests only).  Along a ray, at
 * point k with upwind neighbour u = k - dk and downwind neighbour d = k + dk, t = optical distance from k towards u:
 *   dtau_u = (chi_u + chi_k)/2 |z_u - z_k| / mu,   dtau_d likewise,   p = (S_u - S_k)/dtau_u,   q = (S_k - S_d)/dtau_d
 *   S(t) = S_k + a t + b t^2 on the upwind interval, through (0, S_k) and (dtau_u, S_u) with slope a at k: b = (p - a)/dtau_u
 *   I_k  = I_u e^{-dtau_u} + w0 S_k + w1 a + w2 b,        w_n = int_0^{dtau_u} t^n e^{-t} dt  (lsx_w3)
 * The slope is the weighted harmonic mean of the two difference quotients (Fritsch & Butland 1984; Auer 2003),
 *   a = p q / (alpha q + (1 - alpha) p),  alpha = (1 + dtau_d/(dtau_u + dtau_d))/3   if p q > 0,   a = 0 otherwise,
 * limited to |a| <= 2 |p|: then S'(t) keeps the sign of p over the whole interval -- the parabola never leaves the range of
 * the data (monotonic) -- and a depends CONTINUOUSLY on the data.  (A hard switch "parabola through the three points,
 * linear where it overshoots" was measured first: the accelerated lambda iteration then flips between the two branches from
 * one iteration to the next and stalls, FALC CaII at dJ = 0.1.)  Third order in the grid spacing where S is monotonic.
 *   Psi*_k = dI_k/dS_k at fixed I_u, / chi_k = [w0 + (w1 - w2/dtau_u) da/dS_k - w2/dtau_u^2] / chi_k,
 *   da/dS_k = (beta p^2/dtau_d - alpha q^2/dtau_u)/(alpha q + beta p)^2 (beta = 1 - alpha), -2/dtau_u where the limit is
 *   active, 0 where a = 0.
 * Last point of the ray (no downwind neighbour): a = p (the linear rule with its own interval's weights).  First point:
 * I = Istart, Psi* = 0, as formal_solver.py:116-118.  Arrays as lsx_piecewise_1d_impl. */
int lsx_piecewise_parabolic_1d_impl(int32_t device, int32_t nray, int32_t Nspace, const double* height,
                                    const double* mu, const int32_t* to_obs, const double* Istart,
                                    const double* chi, const double* S, double* I, double* PsiStar);
/* w0, w1, w2 for n optical-depth steps: below 0.25 the series w_n = sum_m (-1)^m dtau^(m+n+1) / (m! (m+n+1)), m <= 11
 * (the closed forms cancel there: at the linear rule's switch, 5e-4, w2 would keep 11 digits), (1, 1, 2) above 50, else
 * w0 = 1 - e, w1 = w0 - dtau e, w2 = 2 w1 - dtau^2 e.  Relative accuracy 1e-13 throughout.  w: [n][3]. */
int lsx_w3(int32_t device, int32_t n, const double* dtau, double* w);

/* Which rule the context's formal solution uses from the next lsx_formal_sol_gamma on: LSX_SOLVER_LINEAR (default, the
 * reference's piecewise_linear_1d) or LSX_SOLVER_PARABOLIC (the rule above, boundary conditions of
 * formal_solver.py:203-209 unchanged). */
enum { LSX_SOLVER_LINEAR = 0, LSX_SOLVER_PARABOLIC = 1 };
int lsx_set_formal_solver(lsx_ctx* ctx, int32_t solver);

/* ---- which sweep kernel runs: part of the PROBLEM, not of the shard ("per-column results must be bitwise equal
 * to the 1-GPU run"; response_fn.py:61-65 is one loop over all columns) ---------------------------------------------------
 * The HIP library has two mappings of the formal solution onto the wavefront: one ray per lane (lsx_sweep.hip; the fused
 * launch of small contexts and the per-class launches give the same bits) and the ray-serial mapping (lsx_sweep_rs.hip: a
 * lane walks the five rays of one wavelength, five columns per wavefront).  They compute the same terms and associate the
 * angle / wavelength sums differently: results agree to rounding (1e-13), not bit for bit.  Which one pays depends on how many
 * columns there are, so LSX_SWEEP_AUTO decides by a column count: ray-serial from 160 columns on (if the context's shape
 * admits it: five rays, wavelength-independent scattering), else one ray per lane; the parabolic rule (above) likewise takes its
 * compile-time tile classes from 32 columns on and its generic instance below, and from 160 columns on the ray-serial instances it
 * has for tiles of at most one line (the other classes stay on one ray per lane).
 * `decide_for_columns` is that count: 0 = the context's own columns (the default), > 0 = decide as for a context of that many
 * columns.  A driver that shards N columns over several contexts (ranks, GPUs, sub-batches) passes N to every one of them: every
 * column then gets the bits it gets when all N sit in one context, whatever the shard sizes (tests/test_sharding_invariance.py).
 * LSX_SWEEP_RAY_PER_LANE / LSX_SWEEP_RAY_SERIAL pin the mapping outright (LSX_EUNSUPPORTED if the context's shape does not
 * admit the ray-serial kernel).  Takes effect from the next formal solution on.  The oracle accepts and ignores the call. */
enum { LSX_SWEEP_AUTO = 0, LSX_SWEEP_RAY_PER_LANE = 1, LSX_SWEEP_RAY_SERIAL = 2 };
int lsx_set_sweep_policy(lsx_ctx* ctx, int32_t policy, int32_t decide_for_columns);
/* the mapping the policy selects for the next formal solution: LSX_SWEEP_RAY_PER_LANE or LSX_SWEEP_RAY_SERIAL (under either rule: the
 * classes that have an instance of the rule on that mapping use it); the oracle: 0 */
int32_t lsx_sweep_policy(const lsx_ctx* ctx);

/* ---- explicit options (round 5) --------------------------------------------------------------------------------------------
 * The HIP library's plan (tile schedule, linked continua, which classes have a ray-serial instance, ...) and a few runtime
 * choices can be steered.  Until round 4 that was possible through LSX_* environment variables only, read inside lsx_create:
 * a sharded job whose ranks differ in their environment would silently associate its sums differently per rank.  Now
 *   - lsx_create_with_options takes an explicit list "key=value,key=value" (NULL or "" = lsx_create).  Keys that change the
 *     association of a sum: linked=0|1 (continua of a line's atom outside the sweep), tiler=dp|natural, topo=0|1, fast_rows=0|1,
 *     rs=0|1 (ray-serial instances at all), rs_min_columns=N (LSX_SWEEP_AUTO's threshold), rs_max_npt=0..2,
 *     fold=0|1 and epi=0|1 (the fast continua's opacities / rate integrands formed inside the ray-serial sweep); launch shape and
 *     measurement only (same bits, tested): order=plan|cost, occ_wg=N, phi_group=0|1, se_lds, serial, finish_big, finish_lds (the Gamma
 *     epilogue with a thread's matrix in LDS instead of a thread per column), fused_epilogue, graph, fused_fast (each 0|1), trace_classes,
 *     class_chunk=N (cut a class into launch groups of N tiles: a measured alternative, same bits).  Unknown keys / malformed values: LSX_EINVAL.  The LSX_* environment variables
 *     of rounds 1-4 remain as diagnostic DEFAULTS that an explicit entry overrides;
 *   - lsx_effective_options writes what the context ended up with, plus the rule, the sweep mapping the policy selects and the
 *     plan's class list, the ABI version and the library's build id (lsx_build_id), as one "key=value;..." string.  If `n` is too
 *     small: LSX_EINVAL, and the message of lsx_last_error() names the bytes needed; 4096 bytes hold every context of the reference's
 *     atoms (the class list grows with the plan: class_chunk can make it longer);
 *   - lsx_options_signature is a 64-bit hash of that string.  Contexts with equal signatures on equal problems give every column
 *     the same bits -- the string names the BUILD of the library too (two ranks that load different builds of liblsx_hip.so, e.g. through
 *     LSX_HIP_LIBRARY, differ in it and are refused like ranks with different options); a multi-process driver exchanges the signatures once and refuses to start on a mismatch
 *     (lightspinner_amd/parallel.py, check_same_options).
 * The oracle accepts any well-formed list, ignores it and reports "backend=oracle-c". */
int lsx_create_with_options(const lsx_problem* desc, int32_t ncol, int32_t device, void* stream, const char* options, lsx_ctx** out);
int lsx_effective_options(const lsx_ctx* ctx, char* buf, size_t n);
uint64_t lsx_options_signature(const lsx_ctx* ctx);

/* Measurement hooks (bench.py): time `reps` back-to-back FS calls with device events
 * on the context's stream.  ms_total = whole FS call (all kernels), ms_sweep = the
 * dominant sweep kernel(s) alone, both averaged per call. */
int lsx_time_formal_sol(lsx_ctx* ctx, int32_t warmup, int32_t reps, double* ms_total,
                        double* ms_sweep);

/* Introspection */
const char* lsx_last_error(void);
const char* lsx_backend_name(void);    /* "hip-gfx950" or "oracle-c"                   */
/* 16 hex digits: hash of the sources this library was built from (csrc/Makefile; the oracle: "oracle-c").  Part of
 * lsx_effective_options / lsx_options_signature. */
const char* lsx_build_id(void);
int32_t lsx_abi_version(void);
/* Algorithmic bytes one FS call moves per column (SURVEY 8d formula) */
double lsx_algorithmic_bytes_per_column(const lsx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* LSX_H */
