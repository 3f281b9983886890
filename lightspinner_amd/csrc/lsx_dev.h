// lsx_dev.h -- device-side data model shared by the host runtime (lsx_hip.hip) and the
// sweep kernel (lsx_sweep.hip).  gfx950 / wave64 only.
//
// HBM layout (all float64, one block per context, index order left = slowest):
//   per column, TILE-major then depth-major: everything one wavefront (tile, direction) reads is
//   one contiguous stream in the order it walks the depths, private to that tile (no 128-B line is
//   shared between workgroups), j = wavelength index inside the tile, L = tile width:
//     bgchi_T, bgeta_T   [col][tile][k][j<L]
//     J_T[2]             [col][tile][k][j<L]      ping-pong: Jdag <- previous call
//     sca                [col][k]  (or [col][tile][k][j] when sca_per_lambda)
//     phi_T              [col] { per (tile, line slot): [dir][k][mu][l<len] }   (compact: [k][l])
//     E_T                [col][tile][k][j<L]      exp(-hc / (k lambda T)): the Boltzmann factor of every continuum's g_ij
//                                                 (rh_method.py:453-454), built at upload (constant while T is fixed)
//     corr_T             [col] { per tile with linked continua: [line slot][EC, XCi, XCj][k][j<L] }  (k_fast_prepass)
//     Psi3_T             [dir][col] { per tile with linked continua: [line slot][k][j<L] }  sum_mu w Psi* phi  (sweep)
//   where the slot's block covers the len wavelengths of the tile inside the transition's range
//   per column, reference layout (level-major; the sweep reads them with wave-uniform
//   addresses through the scalar cache):
//     n, nStar           [col][NLtot][k]
//     C, Gamma           [col][NL2tot][k]
//     wphi               [col][Nlines][k]
//   outputs of the sweep:
//     Iout               [col][la][mu]
//     Gpart              [col][slot][e=ij,ji][dir][k]     one value per (tile-slot, depth, dir): no atomics
//     dJpart             [col][tile][dir]
#pragma once
#include <stdint.h>

#define LSX_WAVE 64
#define LSX_MAX_ATOMS 8
#define LSX_MAX_FAST 64   // fast continua per tile (all five of the reference's model atoms active: 44 bound-free continua overlap at 91 nm)
#define LSX_MAX_PER_RAY 32

struct DevTrans {           // one radiative transition, column independent (host + Gamma epilogue)
    int32_t atom, is_line;
    int32_t li, lj;         // global level index (lev_off[atom] + i / j)
    int32_t Nblue, Nlam;
    int32_t phi_off;        // lines: offset in lambda points into the column's SNl pool
    int32_t line_idx;       // lines: index among lines (wphi row)
    int32_t cont_off;       // continua: offset in lambda points into the SNc pool
    int32_t wl_off;         // offset into the wl / alpha per-(transition, lt) tables
    int32_t gam_ij, gam_ji; // element offsets of Gamma[i][j], Gamma[j][i] inside NL2tot
    double cB;              // lines: (hc/4pi) * Bij                     rh_method.py:268,279
    double gij;             // lines: Bji / Bij                          rh_method.py:450
    double AB;              // lines: Aji / Bji                          rh_method.py:281
    double lambda0;         // lines: rest wavelength [nm] (profile set-up, rh_method.py:234)
};

// level-bookkeeping flags of a slot inside its tile (atom.chi / atom.U / atom.eta of
// rh_method.py:616-627 are only materialised, in lane-private LDS cells, where two transitions
// of the tile actually share a level or an atom)
enum {
    SLOT_LINE = 1,
    SLOT_LI_CELL = 2,      // lower level shared with another slot  -> chi_lev[i] via cell
    SLOT_LJ_CELL = 4,      // upper level shared                    -> chi_lev[j], U_lev[j] via cells
    SLOT_UI_READ = 8,      // some slot's upper level is this slot's lower level (U_lev[i] != 0)
    SLOT_ETA_CELL = 16,    // another slot of the same atom in the tile -> eta_atom via cell
    SLOT_CHI_I_FIRST = 32, // first writer of the cell in execution order (store, else add)
    SLOT_CHI_J_FIRST = 64,
    SLOT_U_J_FIRST = 128,
    SLOT_ETA_FIRST = 256,
    SLOT_FAST = 512,       // fast continuum: handled by k_fast_prepass / k_fast_gamma, not by the sweep
    SLOT_LINKED = 1024     // a fast continuum whose atom HAS a line in the tile ("linked"): the line enters its Gamma
                           // integrand through sum_mu w Psi* phi, and it enters the line's through three ray-independent sums
};

// One transition as one tile sees it ("slot"); wave-uniform, read through the scalar cache.
// A tile's slots are ordered: per-ray slots first (lines, then continua of atoms that have
// a line in the tile), then "fast" continua (atoms without a line in the tile: their
// opacity, emissivity and level bookkeeping do not depend on the ray).
#define LSX_EXP_TAB 128     // doubles at the start of the sweep's LDS: the exp table

struct DevSlot {
    int32_t flags;
    int32_t li, lj;        // global level ids (rows of n)
    int32_t ci, cj, ca;    // tile-local LDS cell ids: level cells of i and j, atom cell
    int32_t atom;          // active-atom index
    int32_t Nblue, Nlam;
    int32_t base;          // lines: element offset of this (tile, line) block inside the column's phi_T; fast continua: row of nsr
    int32_t first, len;    // global index of the block's first wavelength, number of wavelengths in the block
    int32_t wl_off;        // into wl / alpha
    int32_t trans;         // row of the `active` table
    int32_t wphi_off;      // lines: line_idx * Nspace
    double cB;             // lines: (hc/4pi) Bij                    Vij = cB phi        rh_method.py:279
    double g;              // lines: Bji/Bij                         Vji = g Vij         :280, :450
    double Vc;             // lines: g cB
    double Uc;             // lines: (Aji/Bji) g cB                  Uji = Uc phi        :281
    // tiles with a compile-time slot count keep the level bookkeeping of rh_method.py:616-627 in registers: what
    // every OTHER per-ray slot v of the tile (ascending, o = 0, 1, 2) adds to this slot's atom.chi[i], atom.chi[j],
    // atom.U[j], atom.U[i], atom.eta, as factors in {-1, 0, 1}:
    //   [li_v==li] - [lj_v==li],  [li_v==lj] - [lj_v==lj],  [lj_v==lj],  [lj_v==li],  [atom_v==atom]
    double rel[3][5];
    uint32_t relmask;      // bit o: rel[o] has a non-zero factor
    uint32_t lkbits;       // linked continuum: for line slot u of the tile (u < 4) bits 8u + {0, 1, 2} = the line belongs to the
                           // continuum's atom, the continuum's LOWER level is the line's lower level, ... the line's upper level
};
enum { REL_CI = 0, REL_CJ = 1, REL_UJ = 2, REL_UI = 3, REL_EA = 4 };

struct DevTile {            // L consecutive wavelengths (L = 64 / Nrays) of one column
    int32_t la0, nla;       // first global wavelength index, count (<= L)
    int32_t nP;             // per-ray slots
    int32_t nF;             // fast continua
    int32_t slot0;          // first entry in the slot table / first Gpart slab
    int32_t fast_simple;    // the fast continua need no level cells (k_fast_gamma)
    int32_t nL;             // lines among the per-ray slots (they come first)
    int32_t nK;             // linked continua among the nF fast ones
    int32_t corr_off;       // nK > 0: element offset of the tile's [nL][3][Nspace][L] block inside a column of corr_T
    int32_t pp_off;         // nK > 0: element offset of the tile's [nL][Nspace][L] block inside a column of Psi3_T
    int32_t nX;             // nK > 0: nL correction slots behind the tile's nP + nF slots -- Gpart slabs (one entry per rate, like a fast
                            // continuum's) in which the fast-continuum epilogue leaves what the linked continua add to each LINE's
                            // rates where the sweep does not apply it itself (ray-serial instances, lsx_fast.h); zeros otherwise
    int32_t pad_;
};

struct SweepParams {
    // dimensions
    int32_t Nspace, Nrays, Nspect, Natoms, Ntrans, ncol;
    int32_t NLtot, NL2tot, Nlines;
    int32_t sca_per_lambda;
    int32_t phi_compact;
    int32_t nslot_total, ntile_total;
    int32_t L;                  // wavelengths per tile
    int32_t ncell_lev, ncell_atom, nstash; // LDS layout: [2*ncell_lev level cells][ncell_atom][nstash][1 exchange row]
    int32_t n_class_tiles;      // tiles of the launched class
    // column-independent tables
    const double* wavelength;   // [Nspect]
    const double* zmu;          // [Nrays] 1/muz
    const double* wmuh;         // [Nrays] 0.5*wmu                              rh_method.py:661
    const double* wl;           // per (transition, lt): wlambda(lt)/hc (lines) | wlambda(lt)/lambda/h (continua)
    const double* alpha;        // per (transition, lt) (continua; 0 for lines)
    const double* u_la;         // [Nspect] 2hc/lambda^3                        rh_method.py:286
    const double* fgtab;        // the column-mapped fast-continuum epilogue's per-tile tables (lsx_fast.h)
    const uint8_t* active;      // [Ntrans][Nspect]
    const DevTile* tiles;
    const DevSlot* slots;       // per (tile, slot) parameters
    const int32_t* class_tiles; // tile ids of the launched class
    // per-column strides (in doubles)
    int64_t phi_col_stride, corr_col_stride, pp_col_stride;
    // per-column arrays
    const double* height;       // [col][k]
    const double* temperature;  // [col][k]
    const double* n;            // [col][NLtot][k]
    const double* wphi;         // [col][Nlines][k]
    const double* bgchi_T;
    const double* bgeta_T;
    const double* bgce_T;       // [col][tile][k][j<L][2]: background opacity and emissivity as PAIRS (round 6: one 16-byte load per lane and depth
                                // instead of two 8-byte ones in the ray-serial instances; LSX_BG_PAIRS), or nullptr
    const double* bgxce_T;      // ... and the effective background (pre-pass output) of the ray-serial classes that keep the pre-pass
    const double* bgxchi_T;     // effective background of tiles with fast continua (k_fast_prepass)
    const double* bgxeta_T;
    double* Psi2_T;             // [dir][col][tile][k][j]  sum_mu w Psi* per direction (tiles with fast continua)
    const double* sca;
    const double* phi_T;
    const double* nsr;          // [col][Ncont][k] nStar_i / nStar_j of every continuum
    int32_t Ncont, pad_nc;
    const double* E_T;          // [col][tile][k][j]
    const double* corr_T;       // corrections of the line slots for linked continua (k_fast_prepass)
    double* Psi3_T;             // [dir] x pp_col_stride * ncol: sum_mu w Psi* phi per line slot of tiles with linked continua
    const double* Jdag_T;
    double* Jnew_T;
    double* Iout;
    double* Gpart;
    double* dJpart;
    const uint8_t* colmask;     // per-column activity (frozen columns exit at once), or null
    double* debug;              // diagnostic builds only
    int32_t static_max;         // fused launch: tiles with more per-ray slots than this take the generic path
    const double* exp2_tab;     // [64][2]: 2^(j/64) as a (head, tail) pair, for the sweep's exp(-dtau)
    int32_t phi_G;              // columns per group of the line-profile store (phi_elem below): 5 where the ray-serial sweep can run, else 1
    int32_t fused_fast;         // fused small-batch launch: the workgroup of a tile with fast continua runs the tile's pre-pass before and
    int32_t nF_max;             // its Gamma epilogue after the sweep itself (lsx_fast.h); nF_max: the pre-pass's LDS layout
    // ray-serial sweep: the per-depth operands of every transition and the geometry, per group of five columns (lsx_plan.h,
    // "a RING in LDS"): optab[group]{[t < Ntrans][r <= Nspace][c < 5][3], [r <= Nspace][c < 5][2]}, rebuilt per formal solution
    const double* optab;
    int64_t optab_group_stride; // doubles per group
    const int32_t* trans_row;   // per transition: its row of wphi (lines) / its continuum index (continua)
    int32_t fold, fold_nF;      // the launched class runs its FOLDED instance (lsx_plan.h); the most fast continua a tile of it has
    int32_t epi, pad_epi;       // ... its EPI instance: the second visitor of a depth forms the fast continua's Gamma integrands itself
};

// ---- the line-profile store phi_T --------------------------------------------------------------------------------------------
// Per (tile, line) block and column the profile is [dir][k][mu][l < len] (x = (dir Nspace + k) Nrays + mu is the block's row; compact
// profiles: x = k).  The ray-serial sweep reads row x of FIVE consecutive columns with one load per wavefront, so the store
// interleaves the columns of a group of G = 5 inside every block row: [group]{block: [x][c < G][l < len]} -- the five 96-byte pieces
// of a load are one contiguous run of G len doubles, and a wavefront's whole depth step is one run of Nrays G len doubles (2400 B)
// instead of five runs a column apart.  G = 1 (contexts whose shape excludes the ray-serial sweep) is the plain per-column store.
// base: the block's offset in a column's own numbering (DevSlot.base); col_stride: doubles per column (LsxPlan.phi_col; its last
// two doubles per column stay zero: where lanes outside a line's range point their loads -- element G col_stride - 1 of a group).
#ifdef __HIPCC__
#define LSX_HD __host__ __device__
#else
#define LSX_HD
#endif
static LSX_HD inline size_t phi_elem(size_t col, int G, size_t col_stride, size_t base, size_t x, int len, int l)
{
    return (col / G) * G * col_stride + (size_t)G * (base + x * len) + (col % G) * len + l;
}

// ---- device functions shared by the sweep kernel and the stand-alone formal solver -------------------------
#ifdef __HIPCC__
// 1/x: v_rcp_f64 seed (about 2^-26 relative) + one Newton step: relative error <= 4e-15, three instructions instead of
// the ~12 of an IEEE division.  The reference divides; measured effect on the parity figures against two steps (exact to
// an ulp): J, I unchanged at 1e-13 / 2e-12, off-diagonal Gamma 2e-13 -> 6e-13 (CaII), 5e-11 -> 9e-11 (Ca+H) -- inside
// the stated tolerances; -DLSX_NEWTON2 restores the second step.
static __device__ __forceinline__ double rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
#ifdef LSX_NEWTON2
    r = fma(fma(-x, r, 1.0), r, r);
#endif
    return r;
}
// d = a * b + c as the three-address VOP3 form.  The compiler prefers v_fmac (d += a * b), which costs an extra
// v_mov_b64 whenever c must survive (polynomial coefficients, running sums that are read again).
static __device__ __forceinline__ double fma3(double a, double b, double c)
{
#ifndef LSX_NO_FMA3
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return fma(a, b, c);
#endif
}
// the same with a wave-uniform second factor (a literal constant: one scalar register pair, no vector registers)
static __device__ __forceinline__ double fma3s(double a, double b_uniform, double c)
{
#ifndef LSX_NO_FMA3
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_uniform), "v"(c));
    return d;
#else
    return fma(a, b_uniform, c);
#endif
}

// ... and with a wave-uniform addend (Horner steps: the coefficient sits in a scalar register pair)
static __device__ __forceinline__ double fma3c(double a, double b, double c_uniform)
{
#ifndef LSX_NO_FMA3
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
    return d;
#else
    return fma(a, b, c_uniform);
#endif
}

typedef __attribute__((address_space(3))) double lds_f64;   // LDS pointers carry their address space: ds_ instructions, no flat-pointer checks

// exp(x) for the sweep (x = -dtau, -50 <= x <= -5e-4 where the value is used): x = (64 q + j) ln2/64 + r with
// |r| <= ln2/128, exp(x) = 2^q * 2^(j/64) * exp(r).  2^(j/64) comes from a 64-entry (head, tail) table in LDS, exp(r)
// from a degree-6 polynomial (truncation < 2e-19); about half the instructions of the library routine, and the
// coefficients fit the scalar operand slot.  Error below 1 ulp, like the library's.  NaN propagates.
static __device__ __forceinline__ double exp_tab64(double x, const lds_f64* tab)
{
    const double kf = __builtin_rint(x * 0x1.71547652b82fep+6);        // 64 / ln2
    double r = fma(kf, -0x1.62e42fef80000p-7, x);                       // ln2/64, head (18 trailing zero bits)
    r = fma(kf, -0x1.1cf79abc9e3b4p-42, r);                             //         tail
    const int ki = (int)kf;
    const lds_f64* e = tab + 2 * (ki & 63);
    const double th = e[0], tl = e[1];
    double t = fma3s(r, 1.0 / 720.0, 1.0 / 120.0);
    t = fma3(r, t, 1.0 / 24.0);
    t = fma3(r, t, 1.0 / 6.0);
    t = fma(r, t, 0.5);
    const double m = fma(r * r, t, r);                                  // exp(r) - 1
    return ldexp(fma(th, m, tl) + th, ki >> 6);
}

// Boltzmann factor of a continuum's g_ij, exp(-hc / (k lambda T)) (rh_method.py:453): a per-wavelength constant times 1 / T through
// the table exponential above.  The ray-serial sweep forms it in the lane (1 / T arrives with the depth's operand row, lsx_plan.h
// LSX_RS_GEO) and k_build_E writes the SAME bits into the stream the other kernels read.  Against the reference's two divisions the
// argument differs by at most 1.5 ulp, i.e. the factor by |x| 1.7e-16 relative -- and the factor only matters where |x| is small
// (x e^-x <= 0.37): below 1e-16 of the opacity it enters.
static __device__ __forceinline__ double boltzmann_lane_constant(double wavelength_nm)
{
    return -(6.6260755E-34 * 2.99792458E+08 / (1.380658E-23 * 1.0E-09)) / wavelength_nm;
}
static __device__ __forceinline__ double boltzmann_factor(double lane_constant, double rT, const lds_f64* tab)
{
    return exp_tab64(fmax(lane_constant * rT, -740.0), tab);
}

// min of two doubles as ONE instruction: fmin() is preceded by a canonicalising v_max_f64(x, x) (signalling NaNs); the
// instruction itself already returns the other operand for any NaN, which is all the callers rely on
static __device__ __forceinline__ double min_noquiet(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// formal_solver.py:14-44.  The three regimes are selected per lane; the exponential is skipped
// for the whole wavefront when no lane is in the middle regime (top / bottom of the atmosphere).
static __device__ __forceinline__ void w2(double dtau, double& w0, double& w1, const lds_f64* exp2_tab)
{
#ifdef LSX_W2_SELECT    // diagnostic variant: round 2's form (the series always, two selects)
    const bool small = dtau < 5e-4;
    const bool large = dtau > 50.0;
    double a0 = 1.0, a1 = 1.0;
    if (__builtin_amdgcn_ballot_w64(!(small || large)) != 0) {
        const double dc = min_noquiet(dtau, 700.0);
        const double e = exp_tab64(-dc, exp2_tab);
        a0 = 1.0 - e;
        a1 = a0 - dc * e;
    }
    const double t0 = dtau * (1.0 - 0.5 * dtau);
    const double t1 = (dtau * dtau) * (0.5 - dtau * (1.0 / 3.0));
    w0 = small ? t0 : a0;
    w1 = small ? t1 : a1;
}
#else
    const bool small = dtau < 5e-4;
    const bool large = dtau > 50.0;
    double a0, a1;
    if (__builtin_amdgcn_ballot_w64(!(small || large)) != 0) {
        // The saturated regime needs no select of its own: for dtau > 50 the middle formulae give exactly (1, 1) --
        // e = exp(-dtau) < 2e-22 is below half an ulp of 1, and so is dtau e (< 1e-17 up to dtau ~ 4e4; beyond that e
        // the product only shrinks).  The argument is clamped so that the table index stays in range for any dtau.
        const double dc = min_noquiet(dtau, 700.0);
        const double e = exp_tab64(-dc, exp2_tab);
        a0 = 1.0 - e;
        a1 = a0 - dc * e;
    } else {
        a0 = 1.0;             // every lane small or saturated
        a1 = 1.0;
    }
    // the series and its selects only where some lane of the wavefront needs them (below the top of the atmosphere none does)
    if (__builtin_amdgcn_fcmp(dtau, 5e-4, 4 /* ordered < */) != 0) {       // the lane mask of `small`, as the compare writes it
        const double t0 = dtau * (1.0 - 0.5 * dtau);
        const double t1 = (dtau * dtau) * (0.5 - dtau * (1.0 / 3.0));
        a0 = small ? t0 : a0;
        a1 = small ? t1 : a1;
        asm volatile("" : "+v"(a0), "+v"(a1));      // keeps this a branch (the compiler would speculate the block and select twice)
    }
    w0 = a0;
    w1 = a1;
}
#endif

// Weights of the parabolic rule (include/lsx.h, N4): w_n = int_0^dtau t^n e^-t dt, n = 0, 1, 2; same regimes and the same
// exponential as w2.
// SCOEF: the series' coefficients as scalar operands of three-address fmas (no vector moves, 24 scalar registers held across the
// depth loop -- for the instances that have them to spare); otherwise the compiler's choice (vector constants, v_fmac + moves)
template <bool SCOEF = false>
static __device__ __forceinline__ void w3(double dtau, double& w0, double& w1, double& w2q, const lds_f64* exp2_tab)
{
    // Below dtau = 0.25 the closed forms cancel (oracle/lsx_oracle.c, w3: three twelve-term series there).  Here ONE series,
    // for the second moment, and the recurrence of the moments run DOWNWARDS, which only adds positive terms:
    //   w_{n-1} = (w_n + dtau^n e^-dtau) / n     =>     w1 = (w2 + dtau^2 e) / 2,   w0 = w1 + dtau e
    // (relative error of every weight ~1e-16, like the series; one third of the coefficients, i.e. 48 registers fewer held
    // across the depth loop, and 20 instructions fewer where a wavefront has lanes in the series regime)
    const bool small = dtau < 0.25;
    const bool large = dtau > 50.0;
    double a0 = 1.0, a1 = 1.0, a2 = 2.0, e = 0.0, dc = dtau;
    if (__builtin_amdgcn_ballot_w64(!large) != 0) {
        dc = min_noquiet(dtau, 700.0);
        e = exp_tab64(-dc, exp2_tab);
        a0 = 1.0 - e;
        a1 = a0 - dc * e;
        a2 = 2.0 * a1 - (dc * dc) * e;
    }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (__builtin_amdgcn_ballot_w64(small) != 0) {          // no lane in the series regime (deep layers): skip it
        const double x = dtau;
        // sum_n (-1)^n x^(n+3) / (n! (n + 3)), twelve terms (x < 0.25: the next one is below 1e-19 of the first)
        double t;
        if constexpr (SCOEF) {
            t = fma3c(x, -1.0 / 558835200.0, 1.0 / 47174400.0);
            t = fma3c(x, t, -1.0 / 4354560.0);
            t = fma3c(x, t, 1.0 / 443520.0);
            t = fma3c(x, t, -1.0 / 50400.0);
            t = fma3c(x, t, 1.0 / 6480.0);
            t = fma3c(x, t, -1.0 / 960.0);
            t = fma3c(x, t, 1.0 / 168.0);
            t = fma3c(x, t, -1.0 / 36.0);
            t = fma3c(x, t, 1.0 / 10.0);
            t = fma3c(x, t, -1.0 / 4.0);
            t = fma3c(x, t, 1.0 / 3.0);
        } else {
            t = 1.0 / 3.0 + x * (-1.0 / 4.0 + x * (1.0 / 10.0 + x * (-1.0 / 36.0 + x * (1.0 / 168.0 + x * (-1.0 / 960.0 + x * (1.0 / 6480.0 + x * (-1.0 / 50400.0 + x * (1.0 / 443520.0 + x * (-1.0 / 4354560.0 + x * (1.0 / 47174400.0 + x * (-1.0 / 558835200.0)))))))))));
        }
        s2 = (x * x) * (x * t);
        s1 = 0.5 * fma(x * x, e, s2);
        s0 = fma(x, e, s1);
    }
    w0 = small ? s0 : (large ? 1.0 : a0);
    w1 = small ? s1 : (large ? 1.0 : a1);
    w2q = small ? s2 : (large ? 2.0 : a2);
}

// one point of the monotonic piecewise-parabolic recurrence (include/lsx.h, N4); IEEE divisions: this rule is not the hot one
struct Para { double I, Lam; };
static __device__ __forceinline__ Para parabolic_point(double Iu, double S_u, double S_k, double S_d, double dtau_u, double dtau_d, bool has_d,
                                                       const lds_f64* etab)
{
    double w0, w1, w2q;
    w3(dtau_u, w0, w1, w2q, etab);
    const double p = (S_u - S_k) / dtau_u;
    double a = p, dadS = -1.0 / dtau_u;                               // end point: the linear rule
    if (has_d) {
        const double q = (S_k - S_d) / dtau_d;
        if (p * q > 0.0) {
            const double alpha = (1.0 + dtau_d / (dtau_u + dtau_d)) / 3.0, beta = 1.0 - alpha;
            const double den = alpha * q + beta * p;
            a = p * q / den;
            dadS = (beta * p * p / dtau_d - alpha * q * q / dtau_u) / (den * den);
            if (fabs(a) > 2.0 * fabs(p)) { a = 2.0 * p; dadS = -2.0 / dtau_u; }
        } else {
            a = 0.0;
            dadS = 0.0;
        }
    }
    const double b = (p - a) / dtau_u;
    Para r;
    r.I = Iu * (1.0 - w0) + w0 * S_k + w1 * a + w2q * b;
    r.Lam = w0 + (w1 - w2q / dtau_u) * dadS - w2q / (dtau_u * dtau_u);
    return r;
}

// the same point with the divisions as v_rcp_f64 + one Newton step: the compile-time tile classes of the parabolic rule
// (lsx_sweep.hip, sweep_tile_par); relative differences of a few 1e-15.  Two reciprocals per point instead of nine divisions:
// 1 / dtau_u is the previous point's 1 / dtau_d (handed over in `ru`, returned in `rd`), and the harmonic mean is taken over one
// common denominator: with alpha = (u + 2 d) / (3 (u + d)), beta = (2 u + d) / (3 (u + d)),
//   a = p q / (alpha q + beta p) = 3 (u + d) p q / D,   da/dS = (beta p^2 / d - alpha q^2 / u) / (alpha q + beta p)^2
//                                                             = ((2 u + d) p^2 / d - (u + 2 d) q^2 / u) 3 (u + d) / D^2,
//   D = (u + 2 d) q + (2 u + d) p      (p q > 0: both terms have one sign, D != 0)
template <bool SCOEF>
static __device__ __forceinline__ Para parabolic_point_fast(double Iu, double S_u, double S_k, double S_d, double dtau_u, double ru, double dtau_d,
                                                            bool has_d, const lds_f64* etab, double& rd)
{
    double w0, w1, w2q;
    w3<SCOEF>(dtau_u, w0, w1, w2q, etab);
    const double p = (S_u - S_k) * ru;
    double a = p, dadS = -ru;                                        // end point: the linear rule
    rd = 1.0;
    if (has_d) {
        rd = rcp(dtau_d);
        const double q = (S_k - S_d) * rd;
        if (p * q > 0.0) {
            const double sum = dtau_u + dtau_d, cu = sum + dtau_d, cd = sum + dtau_u;
            const double rden = rcp(cu * q + cd * p);
            const double t3 = (3.0 * sum) * rden;
            a = (p * q) * t3;
            dadS = ((cd * p) * p * rd - (cu * q) * q * ru) * (t3 * rden);
            if (fabs(a) > 2.0 * fabs(p)) { a = 2.0 * p; dadS = -2.0 * ru; }
        } else {
            a = 0.0;
            dadS = 0.0;
        }
    }
    const double b = (p - a) * ru;
    Para r;
    r.I = Iu * (1.0 - w0) + w0 * S_k + w1 * a + w2q * b;
    r.Lam = w0 + (w1 - w2q * ru) * dadS - (w2q * ru) * ru;
    return r;
}

// DPP (VALU cross-lane moves, no LDS crossbar traffic).  The total of the 64 lanes ends up in
// lane 63 (the lane that stores it).
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    // `old` = the source itself: no zero has to be materialised; lanes a row mask leaves out keep their own
    // value (their sums are never consumed, see the callers)
    if constexpr (ROW_MASK == 0xf) {                // every lane is written: no `old` operand to set up
        lo = __builtin_amdgcn_mov_dpp(lo, CTRL, ROW_MASK, 0xf, true);
        hi = __builtin_amdgcn_mov_dpp(hi, CTRL, ROW_MASK, 0xf, true);
    } else {
        lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    }
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_f64<0xB1, 0xf>(v);  // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xf>(v);  // quad_perm [2,3,0,1]: every lane of a quad holds the quad sum
    v += dpp_f64<0x141, 0xf>(v); // row_half_mirror: 8-lane sums
    v += dpp_f64<0x140, 0xf>(v); // row_mirror: 16-lane (row) sums in every lane of the row
    v += dpp_f64<0x142, 0xa>(v); // row_bcast15 into rows 1 and 3: lane 31 = rows 0+1, lane 63 = rows 2+3
    v += dpp_f64<0x143, 0xc>(v); // row_bcast31 into rows 2 and 3: lane 63 = all four rows
    return v;
}
static __device__ __forceinline__ double row_sums(double v)   // every lane of a 16-lane row gets the row's sum
{
    v += dpp_f64<0xB1, 0xf>(v);
    v += dpp_f64<0x4E, 0xf>(v);
    v += dpp_f64<0x141, 0xf>(v);
    v += dpp_f64<0x140, 0xf>(v);
    return v;
}
#endif // __HIPCC__
