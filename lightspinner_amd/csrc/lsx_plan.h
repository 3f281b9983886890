// lsx_plan.h -- the host-side PLAN of a context: everything lsx_create derives from the problem descriptor before it
// touches the device -- transition tables, the tile schedule (where the wavelength axis is cut, which transition of a
// tile is a per-ray slot / fast / linked continuum), the slot table, the sweep classes, HBM strides and 32-bit offset
// checks, LDS sizes and launch shapes of every kernel of a formal-solution call.  Plain C++ (no HIP types): the same
// translation unit (lsx_plan.cpp) is built into the product library and, with -fsanitize=address,undefined, into a
// CPU test library (tests/test_plan_sanitized.py).
//
// The device-side half of SURVEY 8f N3 ("lambda-signature grouping as a device-side schedule") is this plan:
// atomic_set.py:377-455 gives every transition its [Nblue, Nblue + Nlambda) range and the `active` table; the plan groups
// the wavelengths by the set of transitions active on them.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/lsx.h"
#include "lsx_dev.h"

#define LSX_MAX_TILE_LINES 4  // lines of one tile that linked continua may couple to (k_fast_prepass / k_fast_gamma)
#define LSX_FAST_NQ 6         // k_fast_gamma_cols: fast continua per atom
#define LSX_FGC_ROWS 32       // k_fast_gamma_cols: (column, depth) rows per wave
#define LSX_FGC_MAXF_BIG 48   // ... of the instances for atoms with MORE than LSX_FAST_NQ continua at a wavelength (k_fast_gamma_cols_big: carbon, iron, MgII)
#define LSX_FGC_LISTS 7       // tile lists of the column-mapped epilogue: [0..2] 0 / 1 / 2 lines fed by linked continua, [3] unused, [4..6] the same, big sets
#define LSX_FGC_TAB(L) (2 * LSX_FGC_MAXF_BIG * (L) + 4 * (L))   // doubles of a tile's ready-made table (lsx_create): [q][j]{alpha, w} | [j] u | [2][j] line weights | [j] -hc / (k lambda) (round 6: the Boltzmann factor is formed from it)
#define LSX_FGC_MAXF 12       // k_fast_gamma_cols: fast continua per tile (sizes its operand table)

// ---- the compiled instances of lsx_sweep_kernel<NPT, NL, NR, SCAL, LK, TOPO> (per-ray slots, lines among them, linked
// continua, two-line relation): ONE list.  lsx_sweep.hip expands it into the launch switch, the plan asks it before it
// files a tile under a class; a tile whose shape has no instance runs the generic one (runtime slot loops).
// Shapes the plan can never make are not on the list (round 4, tests/test_instance_ledger.py): a per-ray continuum needs a
// line of its own atom in the tile, so there is no tile with per-ray slots and no line (1,0) (2,0); and if one continuum of an
// atom is per-ray all of that atom's continua in the tile are, so a tile whose only line belongs to that atom has no linked
// continuum left: (2,1,linked) (3,1,linked).
#define LSX_SWEEP_INSTANCES(X)                                                                                   \
    X(0, 0, false, 0) X(1, 1, false, 0) X(2, 1, false, 0) X(2, 2, false, 0)                                      \
    X(3, 1, false, 0) X(3, 2, false, 0) X(3, 3, false, 0)                                                        \
    X(4, 1, false, 0) X(4, 2, false, 0) X(4, 3, false, 0) X(4, 4, false, 0)                                      \
    X(1, 1, true, 0) X(2, 2, true, 0) X(3, 2, true, 0) X(3, 3, true, 0)                                          \
    X(2, 2, false, 1) X(2, 2, false, 2) X(2, 2, true, 1) X(2, 2, true, 2)
// launch code of a class: >= 0 a compiled instance; -1 generic, -3 generic with linked continua, -2 fused small-batch
// kernel, -4 parabolic rule (N4)
constexpr int lsx_class_code(int npt, int nl, bool lk, int topo) { return npt * 8 + nl + (lk ? 64 : 0) + 128 * topo; }
inline bool lsx_sweep_instance_exists(int npt, int nl, bool lk, int topo)
{
    switch (lsx_class_code(npt, nl, lk, topo)) {
#define LSX_X(NPT, NL, LK, TOPO) case lsx_class_code(NPT, NL, LK, TOPO):
        LSX_SWEEP_INSTANCES(LSX_X)
#undef LSX_X
        return npt >= 0;
    default: return false;
    }
}

// ---- LDS layout of one sweep workgroup (doubles from the start of dynamic LDS), shared by the kernel (offsets) and the plan
// (bytes to request).  -DLSX_RED_LDS (measured variant, not the product): lane sums of the Gamma integrands through a
// transposition buffer -- the lanes park their values for lsx_red_steps(npt) depth steps ([step][value][lane], rows padded to
// 66 doubles: conflict-free 16-byte reads), then every lane adds up one chunk of one row and a short DPP tail finishes: 12 %
// fewer vector instructions per step, 3-5 % MORE time on C3 (the added LDS round trip lengthens each wave's serial chain;
// profiles/r03_bound_evidence.md).
#ifndef LSX_RED_T1
#define LSX_RED_T1 4
#endif
#ifndef LSX_RED_T2
#define LSX_RED_T2 2
#endif
constexpr int lsx_red_steps(int npt) { return npt == 1 ? LSX_RED_T1 : (npt == 2 ? LSX_RED_T2 : 1); }   // depth steps per batch
constexpr int lsx_red_vectors(int npt) { return npt > 0 ? 2 * npt * lsx_red_steps(npt) : 0; }        // rows per batch: 8, 8, 6, 8
#ifndef LSX_RED_LDS      // the product's reduction (DPP / permlane trees): [2 waves][2 npt] parked rows of 64 totals instead
#define LSX_RED_ROW 64
#undef LSX_RED_T1
#undef LSX_RED_T2
#define LSX_RED_T1 1
#define LSX_RED_T2 1
#else
#define LSX_RED_ROW 66
#endif
struct SweepLds {
    int rows;        // 64-double rows per wave: level cells, atom cells, one angle-sum row
    int xwg;         // [2][64] cross-wave exchange
    int utab;        // [(Ns + 1)][3 npt + 2] per-depth wave-uniform operands (compile-time slot counts only)
    int tb;          // [2 waves][vectors][LSX_RED_ROW] transposition buffer
    int gpk;         // -DLSX_RED_PARK: [2 waves][2 npt][64] totals parked until 64 depths can leave in one store
    int ctab;        // [npt][npt - 1][5] slot-pair factors (three and four slots)
    int xrow2;       // [2 waves][npt][64] linked tiles: exchange rows of the Psi* phi sums
    int total;
};
constexpr SweepLds lsx_sweep_lds(int npt, bool linked, int Ns, int ncell_lev, int ncell_atom)
{
    SweepLds l{};
    l.rows = 2 * ncell_lev + ncell_atom + 1;
    l.xwg = LSX_EXP_TAB + 2 * l.rows * LSX_WAVE;
    l.utab = l.xwg + 2 * LSX_WAVE;
    l.tb = l.utab + (npt >= 0 ? (Ns + 1) * (3 * npt + 2) : 0);
    l.tb += l.tb & 1;                                           // 16-byte aligned rows
    l.gpk = l.tb + 2 * lsx_red_vectors(npt) * LSX_RED_ROW;
#if defined(LSX_RED_PARK) && defined(LSX_RED_LDS)
    l.ctab = l.gpk + (npt > 0 ? 2 * 2 * npt * LSX_WAVE : 0);
#else
    l.ctab = l.gpk;
#endif
    l.xrow2 = l.ctab + (npt >= 3 ? npt * (npt - 1) * 5 : 0);
    l.total = l.xrow2 + (linked && npt > 0 ? 2 * npt * LSX_WAVE : 0);
    return l;
}

// ---- the ray-serial sweep (lsx_sweep_rs.hip): lane = one wavelength of one column, the lane's LSX_RS_RAYS rays one after the
// other in registers, LSX_RS_COLS columns per wavefront.  Instances exist for the classes below (at most two per-ray slots);
// a context uses them when it has LSX_RS_RAYS rays, a wavelength-independent scattering coefficient, enough columns to fill
// the machine with five-column wavefronts, and column blocks small enough for 32-bit offsets across a column group.
#ifndef LSX_CLASS_CHUNK
#define LSX_CLASS_CHUNK 0
#endif
#define LSX_RS_RAYS 5
#define LSX_RS_COLS 5
#define LSX_RS_MIN_COLUMNS 160         // default of PlanOptions::rs_min_columns
#define LSX_RS_INSTANCES(X)                                                                                      \
    X(0, 0, false, 0) X(1, 1, false, 0) X(2, 1, false, 0) X(2, 2, false, 0)                                      \
    X(1, 1, true, 0) X(2, 2, true, 0)                                                                            \
    X(2, 2, false, 1) X(2, 2, false, 2) X(2, 2, true, 1) X(2, 2, true, 2)
// ... and for the parabolic rule (N4): line-only classes whose Gamma integrands factor (one line; two lines that share one level)
#define LSX_RSP_INSTANCES(X) X(0, 0, false, 0) X(1, 1, false, 0) X(1, 1, true, 0) X(2, 2, false, 1) X(2, 2, false, 2)
#ifndef LSX_RS_WPE1
#define LSX_RS_WPE1 2
#endif
#ifndef LSX_RS_WPE2
#define LSX_RS_WPE2 2
#endif
#define LSX_RS_WPE(NPT, LK) ((NPT) <= 1 ? LSX_RS_WPE1 : LSX_RS_WPE2)       // register budget: waves per SIMD the compiler aims for
inline bool lsx_rs_instance_exists(int npt, int nl, bool lk, int topo)
{
    switch (lsx_class_code(npt, nl, lk, topo)) {
#define LSX_X(NPT, NL, LK, TOPO) case lsx_class_code(NPT, NL, LK, TOPO):
        LSX_RS_INSTANCES(LSX_X)
#undef LSX_X
        return npt >= 0;
    default: return false;
    }
}
inline bool lsx_rsp_instance_exists(int npt, int nl, bool lk, int topo)
{
    switch (lsx_class_code(npt, nl, lk, topo)) {
#define LSX_X(NPT, NL, LK, TOPO) case lsx_class_code(NPT, NL, LK, TOPO):
        LSX_RSP_INSTANCES(LSX_X)
#undef LSX_X
        return npt >= 0;
    default: return false;
    }
}
// ---- the ray-serial sweep's per-depth operands (round 5): a RING in LDS, fed from a table in HBM ---------------------------------
// What a depth step needs of the lane's column beyond its streams -- per per-ray slot three numbers (lines: cB (n_i - g n_j), n_j Uc,
// wphi; continua: n_i, n_j, nStar_i / nStar_j), the half length of the interval above the depth and the scattering coefficient -- is
// wave-uniform per column.  Rounds 3-4 staged the WHOLE column of the five columns of a wavefront in LDS when the workgroup started
// ([5][Nspace + 1][3 npt + 2]: 17-27 kB at 82 depths, growing with Nspace, and the reason contexts of more than ~160 depths fell back
// to the one-ray-per-lane kernel).  Now `k_build_optab` (lsx_hip.hip) writes those numbers once per formal solution into a table
//     optab[group of 5 columns][transition t < Ntrans | geometry (down, up) | continuum (folded instances)][row][LSX_RS_SEG = 16: [c < 5][3], pad]
// (53 kB per column for FALC Ca+H), and
// each WAVE keeps a ring of LSX_RS_RING rows -- one 16-double segment per per-ray slot, the geometry (half interval, sigma, 1 / T), every
// folded continuum -- of the depths around its own: every depth step lane e fetches the PAIR of elements (2 e, 2 e + 1) of the row
// LSX_RS_RING - 1 steps ahead (one 16-byte load per lane, landing a step later) and writes it over the row that was consumed two steps
// ago.  LDS per workgroup no longer depends on Nspace.  (Round 5: rows of 15 npt + 10 (+ 15 nF) doubles, one 8-byte load per element.)
// FOLDED fast continua (round 5): a class whose tiles have fast continua can run instances that form the continua's opacity and
// emissivity themselves (what k_fast_prepass added to the background, rh_method.py:284-286, 453-455, 613-614) -- per continuum q and
// depth the table holds a third kind of block, [c][3] = n_i, n_j nStar_i / nStar_j, nStar_i / nStar_j, the row of a tile with nF fast
// continua is 16 (npt + 1 + nF) doubles (two elements per lane: at most 128), and per (wavelength, depth) the lane needs two fused
// multiply-adds per continuum: chi += sum_q alpha_q n_i,q - E sum_q alpha_q (n_j nsr)_q, eta += (2hc/lambda^3) E sum_q alpha_q (n_j nsr)_q,
// E = exp(-hc / k lambda T): round 5 the tile's Boltzmann stream, round 6 formed in the lane from the row's 1 / T (LSX_ELANE below).
// No pre-pass launch, no effective-background streams for those classes.
// ... and their GAMMA INTEGRANDS (EPI instances; what k_fast_gamma_cols did, rh_method.py:652, 677-681 for a ray-independent
// transition): the wave that visits a depth SECOND has the total mean intensity there; it also fetches the first visitor's half of
// Psibar (and of sum_mu w Psi* phi per linked line), forms every fast continuum's two rate integrands and the linked lines'
// corrections for its wavelength, reduces them over the tile's wavelengths (twelve values per round through LDS rows, like the
// per-ray integrands) and stores the slabs -- one entry per rate, both directions in it, as the epilogue kernel did: k_gamma_finish does
// not change.  No k_fast_gamma_cols launch for those classes, and the second visitor stores neither Psibar nor the Psi* phi sums.
#define LSX_RS_RING 8                   // rows per wave (a power of two)
#define LSX_RS_RING_EPI 4               // ... of the EPI instances (their reduction rows need the LDS)
#define LSX_RS_EPI_ROUND 12             // values per reduction round: 64 lanes / 5 columns
#define LSX_RS_FOLD_ROW_MAX 128         // doubles of a folded row: two elements per lane
// doubles per column of a geometry row: the half length of the interval behind the ray, the scattering coefficient and (round 6)
// 1 / T: the Boltzmann factor exp(-hc / k lambda T) of the continua (rh_method.py:453) is formed in the lane from it instead of being
// read as a stream by both directions (LSX_ELANE=0: the stream, round 5's form)
#ifndef LSX_ELANE
#define LSX_ELANE 1
#endif
#define LSX_RS_GEO (LSX_ELANE ? 3 : 2)
#ifndef LSX_BG_PAIRS
#define LSX_BG_PAIRS 1       // the ray-serial instances read background chi / eta as one 16-byte pair per lane and depth (lsx_dev.h, bgce_T / bgxce_T)
#endif
#ifndef LSX_EPI_ELANE
#define LSX_EPI_ELANE 0      // the same in the column-mapped fast-continuum epilogue: a measured alternative (lsx_hip.hip, enqueue_fs)
#endif
// A row is made of SEGMENTS -- one per per-ray slot, the geometry, one per folded continuum --, each [column][3] = 15 doubles padded to
// LSX_RS_SEG = 16 (round 6): a lane then fetches TWO consecutive elements of the row with one 16-byte load (64 lanes: the 128 doubles a
// folded row may have), where it used to fetch elements e and 64 + e with two loads -- one request per step less in every folded
// instance (what a wave waits for is its requests: profiles/r06_bound_evidence.md 6).  The table's block rows have the same pitch.
#define LSX_RS_SEG 16
static_assert(3 * LSX_RS_COLS <= LSX_RS_SEG && LSX_RS_GEO * LSX_RS_COLS <= LSX_RS_SEG, "a segment holds [column][3]");
constexpr int lsx_rs_row_doubles(int npt, int nF = 0) { return LSX_RS_SEG * (npt + 1 + nF); }
// the folded instances take the fast continua four at a time in straight-line code (the LDS reads of a chunk in flight together; a
// continuum the tile does not have: zero cross-section against a zeroed pad of the row), so the rings' rows and the cross-section
// table are laid out for the class's largest tile rounded up to a multiple of four
// (nor of the unfactored two-line instance with linked continua: it reads the pre-pass's correction streams)
// (round 6: with the Boltzmann factor formed in the lane -- one operand stream less -- the two-line instances with a known relation fit
// the fold too: 255 registers, no scratch; LSX_FOLD_TWOLINE=0 restores round 5's list, in which they kept the pre-pass)
#ifndef LSX_FOLD_TWOLINE
#define LSX_FOLD_TWOLINE LSX_ELANE
#endif
constexpr bool lsx_rs_fold_instance_exists(int npt, bool lk, int topo) { return (LSX_FOLD_TWOLINE || !(npt == 2 && !lk && topo != 0)) && !(npt == 2 && lk && topo == 0); }
constexpr int lsx_rs_fold_pad(int nF) { return (nF + 3) & ~3; }
constexpr int lsx_rs_row_pitch(int npt, int nF = 0) { return (lsx_rs_row_doubles(npt, lsx_rs_fold_pad(nF)) + 1) & ~1; }
// doubles per column group of the table: Ntrans blocks of (Nspace + 1) rows of 15, the geometry block of (Nspace + 1) rows of 10,
// Ncont blocks of (Nspace + 1) rows of 10 (the folded fast continua's operands)
// Every block has LSX_RS_RING zero rows in front of depth 0 and behind depth Nspace - 1 (depth r is row r + LSX_RS_RING): the ring runs
// ahead of the sweep and past its end without clamping its row index -- a lane's table offset just moves by one row per step.
// Two geometry blocks: [half length of the interval ABOVE the depth, sigma] for the down-going sweep, [... BELOW the depth, sigma]
// for the up-going one -- "the interval behind this ray" sits in the depth's own row for both.
constexpr int lsx_optab_rows(int Ns) { return Ns + 2 * LSX_RS_RING; }
constexpr size_t lsx_optab_group_doubles(int Ntrans, int Ns, int Ncont)
{
    return ((size_t)Ntrans + 2 + (size_t)Ncont) * LSX_RS_SEG * (size_t)lsx_optab_rows(Ns);
}
// (the parabolic instances park 16 depths in every class: they need the LDS for the lane-private cells below)
constexpr int lsx_rs_park(int npt, bool par = false) { return (npt >= 2 || par) ? 16 : 64; }      // depths a row of parked Gamma totals holds (two slots: 16, for two workgroups more per CU)
// parabolic instances: rows of 64 lane-private cells per wave -- 1 / opacity, opacity and the line profiles of the rays at the point
// that waits for its downwind neighbour
constexpr int lsx_rs_par_rows(int npt) { return LSX_RS_RAYS * (2 + npt); }
// LDS doubles of a ray-serial workgroup: exp table, [2 waves][2 npt + 1] rows of 64 (parked Gamma integrands, dJ), [2][64] J
// exchange, the two waves' operand rings, the parked Gamma totals, the angle quadrature, the parabolic instances' cells
// (fold_nF >= 0: a folded instance whose tiles have at most that many fast continua -- the rings' row pitch and the [q][64] table of
// the continua's cross-sections per lane)
// (epi: the EPI instance of the class -- shorter rings and parked rows, [2 waves][12] reduction rows for the fast values, a second [q][64]
// table (the continua's wavelength weights), the midpoint's exchange of Psibar and the Psi* phi sums)
constexpr int lsx_rs_lds_doubles(int npt, int Ns, bool par = false, int fold_nF = -1, bool epi = false, int nlk = 0)
{
    (void)Ns;
    return LSX_EXP_TAB + 2 * (2 * (npt > 0 ? npt : 1) + 1) * 64 + 2 * LSX_WAVE +
           2 * (epi ? LSX_RS_RING_EPI : LSX_RS_RING) * (fold_nF >= 0 ? lsx_rs_row_pitch(npt, fold_nF) : lsx_rs_row_doubles(npt)) +
           (fold_nF > 0 ? (epi ? 2 : 1) * lsx_rs_fold_pad(fold_nF) * LSX_WAVE : 0) +
           (epi ? 2 * LSX_RS_EPI_ROUND * LSX_WAVE + 2 * (1 + nlk) * LSX_WAVE + 2 * LSX_WAVE : 0) +
           2 * LSX_RS_COLS * 2 * (npt > 0 ? npt : 1) * lsx_rs_park(npt, par || epi) +  // + [2 waves][columns x values][entries] parked Gamma totals
           2 * LSX_RS_RAYS + 2 +                                            // + the angle quadrature (two-slot instances read it from here)
           (par ? 2 * lsx_rs_par_rows(npt) * LSX_WAVE : 0);
}

namespace lsxd {

struct PlanClass {             // tiles that run the same kernel instantiation
    int npt = -1;              // compile-time per-ray slot count, -1 = generic
    int nl = 0;                // lines among them (compile-time too)
    bool linked = false;       // the class's tiles have linked continua (compile-time too)
    int topo = 0;              // two-line classes: known relation of the two lines (lsx_sweep.hip, TOPO)
    bool has_fast = false;     // some tile of the class has fast continua: the class reads the pre-pass output
    std::vector<int> tiles;
    std::vector<int> fast_tiles;                  // the class's tiles that have fast continua ...
    std::vector<int> fast_cols[LSX_FGC_LISTS], fast_rest;     // ... split by the kernel that builds their Gamma slabs
    int ncell_lev = 1, ncell_atom = 1;
    size_t lds_bytes = 0;
    double work = 0.0;         // estimated share of the call (launch order; stream priority tiers under LSX_PRIO)
    bool rs = false;           // the class has a ray-serial instance (lsx_sweep_rs.hip); lsx_create decides by the column count
    bool rsp = false;          // ... and a ray-serial instance of the parabolic rule (N4; LSX_RSP_INSTANCES)
    bool lk_epi = false;       // ... whose linked corrections the fast-continuum epilogue applies (lsx_fast.h), not the sweep
    bool fold = false;         // the ray-serial instance forms the fast continua's opacity / emissivity itself: no pre-pass for this class
    int fold_nF = 0;           // ... the most fast continua a tile of the class has
    bool epi = false;          // ... and the Gamma integrands of its fast continua (the EPI instance): no epilogue launch for this class either
    int code() const { return npt >= 0 ? lsx_class_code(npt, nl, linked, topo) : (linked ? -3 : -1); }
};

struct PlanOptions {           // diagnostic switches (lsx_create reads them from the environment, once)
    bool no_linked = false;    // LSX_NO_LINKED: round-1 classification (continua of an atom with a line go through the sweep)
    bool natural_tiles = false; // LSX_TILER=natural: cut every L wavelengths
    bool no_topo = false;      // LSX_NO_TOPO
    bool fast_rows = false;    // LSX_FAST_ROWS: the row-mapped epilogue for every tile
    bool finish_lds = false;   // finish_lds=1 / LSX_FINISH_LDS: the Gamma epilogue with a thread's whole matrix in LDS (k_gamma_finish) where that fits, instead of a thread per column (k_gamma_finish_levels) -- the same bits; measurement
    bool order_by_cost = false; // LSX_ORDER=cost
    int occ_wg = 0;            // LSX_OCC_WG: at most this many workgroups per CU (through the LDS request)
    int class_chunk = LSX_CLASS_CHUNK;   // class_chunk / LSX_CLASS_CHUNK: a tile class is cut into launch groups of at most this many tiles, each
                               // with its own stream and pre-pass -> sweep -> epilogue chain (0: one group per class); same kernels, same bits
    bool no_rs = false;        // LSX_NO_RS: every class through lsx_sweep.hip (one ray per lane)
    int rs_min_columns = LSX_RS_MIN_COLUMNS;   // LSX_RS_MIN_COLUMNS: contexts with fewer columns keep one ray per lane (too few wavefronts otherwise)
    bool no_phi_group = false; // LSX_PHI_GROUP=1: the plain per-column profile store also where the ray-serial sweep can run (measurements)
    bool no_epi = true;        // epi=1 / LSX_EPI=1 switches the EPI instances on (the second visitor of a depth forms the fast continua's Gamma
                               // integrands itself).  OFF by default: built, parity-green on every GPU test, and measured 2 % SLOWER than the
                               // epilogue kernel it replaces (profiles/r05/ab_epilogue_in_sweep.txt; DESIGN.md 4.2 says what it would take)
    bool no_fold = false;      // fold=0 / LSX_NO_FOLD: the ray-serial classes keep the pre-pass and the effective-background streams (round 4)
    int rs_max_npt = 2;        // LSX_RS_MAX_NPT: classes with more per-ray slots keep one ray per lane (diagnostic: 1 leaves the two-slot tiles to lsx_sweep.hip)
};

struct RunOptions {            // switches of the runtime around the plan (lsx_hip.hip); same life cycle as PlanOptions
    bool se_lds = false;       // LSX_SE_LDS / se_lds: stat_equil with its system in LDS also for small atoms
    bool serial = false;       // LSX_SERIAL / serial: every class on the context's stream, one after the other
    bool finish_big = false;   // LSX_FINISH_BIG / finish_big: the many-column Gamma epilogue also for small batches
    bool fused_epilogue = false; // LSX_FUSED_EPILOGUE / fused_epilogue
    bool graph = false;        // LSX_GRAPH / graph: a formal solution's launches as a captured HIP graph
    bool no_fused_fast = false; // LSX_NO_FUSED_FAST / fused_fast=0
    int abl_fast = 0;          // LSX_ABL_FUSED_FAST (timing ablation)
    bool trace_classes = false; // LSX_TRACE_CLASSES (prints; changes nothing)
};
struct CtxOptions { PlanOptions plan; RunOptions run; };

// Every switch that decides how a context's sums are associated or its kernels launched, in one place (round 5):
//   options_from_env    the LSX_* diagnostic variables as DEFAULTS (read once per lsx_create, here only);
//   options_apply       an explicit "key=value,key=value" list (lsx_create_with_options) on top of them; unknown keys and
//                       malformed values are errors;
//   options_string      the canonical "key=value;..." form of the result -- what lsx_effective_options reports and what
//                       lsx_options_signature hashes together with the plan's class list, the sweep mapping and the rule.
void options_from_env(CtxOptions* o);
int options_apply(const char* list, CtxOptions* o, std::string* err);
std::string options_string(const CtxOptions& o);
std::string plan_class_string(const struct LsxPlan& P);
uint64_t fnv1a64(const std::string& s);

// launch shapes of the kernels around the sweep, fixed when the plan is made so that no enqueue path can fail on them
struct LaunchShapes {
    int prepass_seg = 0; size_t prepass_lds = 0;                  // k_fast_prepass: depths staged at a time, LDS bytes
    int rows_lp = 16, rows_nt = 256, rows_seg = 0; size_t rows_lds = 0;   // k_fast_gamma
    size_t cols_lds[LSX_FGC_LISTS] = {0, 0, 0, 0, 0, 0, 0};      // k_fast_gamma_cols<0, 1, 2>, [4..6]: k_fast_gamma_cols_big<0, 1, 2>
    int finish_nt = 128; size_t finish_lds = 0;                   // k_gamma_finish
    int finish_w = 0;                                             // > 0: k_gamma_finish_levels, a thread per column of its atom's Gamma (atoms of many levels)
    size_t fused_lds = 0; int fused_ncell_lev = 1, fused_ncell_atom = 1;   // fused small-batch / parabolic launch
    bool fused_fast = false; size_t fused_fast_lds = 0;          // the fused launch runs pre-pass and Gamma epilogue of its fast tiles itself
};

struct LsxPlan {
    int Nspace = 0, Nrays = 0, Nspect = 0, Natoms = 0, Ntrans = 0;
    int NLtot = 0, NL2tot = 0, Nlines = 0, SNl = 0, SNc = 0;
    int sca_per_lambda = 0, phi_compact = 0;
    int L = 0;                                   // wavelengths per tile = 64 / Nrays
    std::vector<int> Nlevel, lev_off, lev2_off;
    std::vector<lsx_transition> trans;
    std::vector<DevTrans> htrans;
    std::vector<double> wave, wl, alpha, zmu, wmuh, u_la;   // column-independent tables as uploaded
    std::vector<uint8_t> active;
    std::vector<DevTile> tiles;
    std::vector<int> tile_slots;
    std::vector<uint8_t> tile_slot_fast;         // per slot: fast continuum (its slabs use the first direction entry only)
    std::vector<DevSlot> slots;
    std::vector<PlanClass> plan_classes;         // in launch order
    std::vector<int> fast_tiles, fast_cols[LSX_FGC_LISTS], fast_rest;
    std::vector<int> cont_li, cont_lj;
    std::vector<int> trans_row;                  // per transition: its row of wphi (lines: the line index) / of nsr (continua: the continuum index)
    int nF_max = 0, Ncont = 0, static_max = -1, nL_linked_max = 0;
    bool fast_generic = false, any_cont = false;
    size_t lds_bytes = 0;                        // largest sweep class
    // per-column strides in doubles
    size_t phi_col = 0, phi_in_col = 0, corr_col = 0, pp_col = 0, sca_col = 0, til_col = 0;
    LaunchShapes shapes;
    int phi_group = 1;         // columns per group of the line-profile store (lsx_dev.h, phi_elem): LSX_RS_COLS where rs_ok
    bool rs_ok = false;        // the context's shape admits the ray-serial sweep (rays, scattering, 32-bit offsets over a column group)
    int rs_min_columns = LSX_RS_MIN_COLUMNS;
};

// -> LSX_OK or an error code with the message in *err.  Checks the descriptor like lsx_create did.
int plan_build(const lsx_problem* d, const PlanOptions& opt, LsxPlan* out, std::string* err);

// k_fast_gamma_cols instance list a tile belongs to: kLkLines[v - 1] < lines fed by linked continua <= kLkLines[v]
static const int kLkLines[4] = {0, 1, 2, LSX_MAX_TILE_LINES};
inline int lkclass(const DevTile& tl)
{
    const int n = tl.nK > 0 ? (tl.nL < LSX_MAX_TILE_LINES ? tl.nL : LSX_MAX_TILE_LINES) : 0;
    return n == 0 ? 0 : (n == 1 ? 1 : (n == 2 ? 2 : 3));
}
// ... and the list of the column-mapped epilogue it is filed in (DevTile.fast_simple: 2 the plain instances, 3 the big-set ones)
inline int fgc_list(const DevTile& tl) { return lkclass(tl) + (tl.fast_simple == 3 ? 4 : 0); }
inline int fgc_lines(int v) { return kLkLines[v & 3]; }
// LDS of a workgroup of NW waves of the column-mapped epilogue: tables for MAXF continua, NW x (3 + lines) streams of LSX_FGC_ROWS rows
// (LSX_EPI_ELANE = 1, a measured alternative: + the Boltzmann constants of the tile's wavelengths and the exponential's table -- the kernel
// then forms exp(-hc / k lambda T) itself; LSX_FGC_EXTRA: those doubles)
#define LSX_FGC_EXTRA(L) (LSX_EPI_ELANE ? (L) + LSX_EXP_TAB : 0)
inline size_t fgc_lds_bytes(int L, int maxf, int nw, int lines) { return ((size_t)2 * maxf * L + (size_t)3 * L + LSX_FGC_EXTRA(L) + (size_t)nw * (3 + lines) * LSX_FGC_ROWS * L) * sizeof(double); }

} // namespace lsxd
