// lsx_fast.h -- the fast-continuum work around the sweep as DEVICE FUNCTIONS, shared by the stand-alone kernels of lsx_hip.hip
// (k_fast_prepass, k_fast_gamma_cols: one launch per tile class before / after the class's sweep) and the fused small-batch sweep
// launch of lsx_sweep.hip, which runs them inside the workgroup of a tile that has fast continua (one launch instead of three
// for a single column: the pre-pass and the epilogue of the few continuum tiles hide behind the longer line tiles).  One
// source, one summation order: both routes give the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "lsx_dev.h"
#include "lsx_plan.h"

namespace {

// ---- fast continua: handled outside the sweep -----------------------------------------------------------------------
// A continuum whose Gamma integrand is affine in the ray quantities with ray-independent coefficients never enters the
// sweep (lsx_create, roles_of): "fast" = its atom has no line in the tile; "linked" = it has, but no line touches the
// continuum's upper level.  Per (wavelength, depth) and summed over the rays of one direction the sweep hands over
//   J (-> sI = 4 pi J over both directions),  Psibar = sum_mu w Psi*,  and per line of the tile  PsiPhi = sum_mu w Psi* phi.
struct FastParams {
    int Nspace, Nspect, Nrays, ncol, ntile, L, NLtot, Natoms, nslot_total, n_fast_tiles;
    const DevTile* tiles;
    const DevSlot* slots;
    const int* fast_tiles;      // ids of the tiles with nF > 0
    const uint8_t* active;
    const double* alpha;
    const double* wl;
    const double* u_la;
    const double* wmuh;
    const double* n;            // [col][NLtot][k]
    const double* nsr;          // [col][Ncont][k]  nStar_i / nStar_j of every continuum (rh_method.py:453)
    const double* E_T;          // [col][tile][k][j] exp(-hc / (k lambda T))
    int Ncont, nF_max, generic;
    int seg_depths;             // depths staged in LDS at a time: a multiple of the rows per block, >= Nspace where that fits
    const double* bgchi_T;
    const double* bgeta_T;
    double* bgxchi_T;
    double* bgxeta_T;
    double* bgxce_T;            // the same as (chi, eta) pairs: what the pre-pass of a RAY-SERIAL class writes instead (pairs_out; lsx_dev.h)
    int pairs_out;
    double* corr_T;             // [col]{tile: [line][EC, XCi, XCj][k][j]}
    int64_t corr_col_stride, pp_col_stride;
    const double* J_T;          // the NEW J (after the sweep), tile-major
    const double* Psi2_T;       // [dir][col][tile][k][j]
    const double* Psi3_T;       // [dir][col]{tile: [line][k][j]}
    double* Gpart;
    const uint8_t* colmask;
    // what the linked continua add to the rates of the tile's LINES (rh_method.py:616-627 through :652, :677-681): the sweep applies
    // it ray by ray from the pre-pass's correction streams, or -- epi_corr, the ray-serial instances -- the column-mapped epilogue
    // applies it from the sums it has anyway (fast_gamma_cols_rows) and the pre-pass writes no correction streams
    int epi_corr, Nlines;
    const double* fgtab;        // [tile][LSX_FGC_TAB(L)] the column-mapped epilogue's per-tile tables, ready made (lsx_create)
    const double* wphi;         // [col][Nlines][k]
    // round 6: the column-mapped epilogue forms the Boltzmann factor exp(-hc / k lambda T) from the temperature and the tile's table
    // (lsx_dev.h, boltzmann_factor: the bits E_T holds) instead of reading the stream: one stream of four less.  nullptr: read E_T
    // (the fused small-batch launch: a single column is latency bound, not byte bound)
    const double* temperature;  // [col][k]
    const double* exp2_tab;     // the exponential's table (lsx_dev.h, exp_tab64)
};

// effective background of the tiles that have fast continua: bgx = bg + sum over the tile's fast continua; for the
// lines of a tile with linked continua also the three sums the line's own Gamma integrand needs from them
// (rh_method.py:616-627: atom.eta, atom.chi[i_line], atom.chi[j_line], continuum part)
// One block per (tile, column), 256 threads = (depth in chunk, wavelength), rows exactly as wide as the tile;
// the block walks the column's depth chunks with the next chunk's loads in flight (the kernel is latency bound otherwise:
// a staging phase, a barrier and one dependent load per thread for a few dozen instructions of arithmetic).
template <bool SEG, int NT>       // SEG: the column is too deep for its operands to be staged at once; NT: threads of the block
static __device__ __forceinline__ void fast_prepass_tile(const FastParams& f, const int t, const size_t col, double* sm)
{
    if (f.colmask && !f.colmask[col]) return;
    const DevTile tl = f.tiles[t];
    const int tid = threadIdx.x;
    // a (tile, column) plane [depth][wavelength] is one contiguous array: thread t takes element t of a chunk of KR whole rows,
    // every lane works whatever the tile width (the pre-pass has no row reductions that would want power-of-two rows)
    const int LP = f.L, KR = NT / LP;                          // depths per chunk
    const int kc = tid / LP, j = tid - kc * LP;
    const int Ns = f.Nspace;
    const DevSlot* fs = f.slots + tl.slot0 + tl.nP;
    const bool lane_on = kc < KR;                               // the last NT - KR L threads of a block idle
    const size_t tb = (col * f.ntile + t) * (size_t)Ns * f.L;
    auto load3 = [&](int k, double& a, double& b, double& c) {
        const bool on = lane_on && k < Ns;
        const size_t o = tb + (size_t)(on ? k : 0) * f.L + (lane_on ? j : 0);
        a = f.bgchi_T[o]; b = f.bgeta_T[o]; c = f.E_T[o];
    };
    double n_chi, n_eta, n_E;
    load3(kc, n_chi, n_eta, n_E);                               // first chunk's streams: in flight during the staging
    // operands of the (slot, depth) and (slot, wavelength) pairs, staged per depth segment of KS depths (the whole column
    // where it fits): sN[q][k - ks0] = {n_i, n_j}, sR[q][k - ks0] = nStar_i/nStar_j, sA[q][j] = alpha where the continuum is
    // active, else 0
    const int KS = SEG ? f.seg_depths : Ns;
    double* sN = sm;                                            // [q][k]{n_i, n_j}
    double* sR = sN + (size_t)2 * f.nF_max * KS;                // [q][k] nStar_i / nStar_j
    double* sA = sR + (size_t)f.nF_max * KS;
    auto stage = [&](int ks0) {
        for (int x = tid; x < tl.nF * KS; x += NT) {
            const int q = x / KS, kk = min(ks0 + (x - q * KS), Ns - 1);
            sN[x * 2 + 0] = f.n[(col * f.NLtot + fs[q].li) * Ns + kk];
            sN[x * 2 + 1] = f.n[(col * f.NLtot + fs[q].lj) * Ns + kk];
            sR[x] = f.nsr[col * f.Ncont * Ns + fs[q].base + kk];
        }
    };
    stage(0);
    for (int x = tid; x < tl.nF * LP; x += NT) {
        const int q = x / LP, jj = x % LP, lq = tl.la0 + min(jj, tl.nla - 1), lt = lq - fs[q].Nblue;
        const bool a = jj < tl.nla && lt >= 0 && lt < fs[q].Nlam && f.active[(size_t)fs[q].trans * f.Nspect + lq] != 0;
        sA[x] = a ? f.alpha[fs[q].wl_off + lt] : 0.0;
    }
    __syncthreads();
    const int nLc = tl.nK > 0 ? min(tl.nL, LSX_MAX_TILE_LINES) : 0;
    const double ula = f.u_la[tl.la0 + min(j, tl.nla - 1)];
    const size_t plane = (size_t)Ns * f.L;
    const int kpad = Ns + KR - 1 - (Ns + KR - 1) % KR;          // whole passes only
    for (int ks0 = 0; ks0 < (SEG ? Ns : 1); ks0 += KS) {        // depth segments: one, unless the column is deep
    if (SEG && ks0 > 0) {
        __syncthreads();
        stage(ks0);
        __syncthreads();
    }
    const int kend = SEG ? min(ks0 + KS, kpad) : kpad;
    for (int k = ks0 + kc; k < kend; k += KR) {                 // (every thread runs every pass; k >= Ns computes nothing)
        double chi = n_chi, eta = n_eta;
        const double E = n_E;
        load3(k + KR, n_chi, n_eta, n_E);
        if (k >= Ns || !lane_on) continue;
        double EC[LSX_MAX_TILE_LINES] = {0.0, 0.0, 0.0, 0.0}, XCi[LSX_MAX_TILE_LINES] = {0.0, 0.0, 0.0, 0.0},
               XCj[LSX_MAX_TILE_LINES] = {0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < tl.nF; ++q) {                        // rh_method.py:284-286, 453-455, 613-614
            const double alf = sA[q * LP + j];
            const double2 n01 = *reinterpret_cast<const double2*>(sN + (size_t)(q * KS + (k - ks0)) * 2);
            const double nsr = sR[q * KS + (k - ks0)];
            const double Vji = (nsr * E) * alf;
            const double Uji = ula * Vji;
            const double chq = n01.x * alf - n01.y * Vji, etq = n01.y * Uji;
            chi += chq;
            eta += etq;
            const unsigned lk = fs[q].lkbits;                    // wave-uniform: which line slots this continuum feeds
            if (lk) {
#pragma unroll
                for (int u = 0; u < LSX_MAX_TILE_LINES; ++u) {
                    if (lk & (1u << (8 * u))) EC[u] += etq;      // atom.eta, :614
                    if (lk & (2u << (8 * u))) XCi[u] += chq;     // atom.chi[i_line], :616 (no line touches lj of a linked
                    if (lk & (4u << (8 * u))) XCj[u] += chq;     // continuum, so only its lower level counts)
                }
            }
        }
        const size_t o = tb + (size_t)k * f.L + j;
        if (f.pairs_out) *reinterpret_cast<double2*>(f.bgxce_T + 2 * o) = make_double2(chi, eta);
        else { f.bgxchi_T[o] = chi; f.bgxeta_T[o] = eta; }
        if (nLc > 0 && !f.epi_corr) {
            double* cr = f.corr_T + col * f.corr_col_stride + tl.corr_off + (size_t)k * f.L + j;
#pragma unroll
            for (int u = 0; u < LSX_MAX_TILE_LINES; ++u)
                if (u < nLc) {
                    cr[(size_t)(3 * u + 0) * plane] = EC[u];
                    cr[(size_t)(3 * u + 1) * plane] = XCi[u];
                    // a tile with a single per-ray slot never reads atom.chi[j_line] (it multiplies atom.U[i_line], which only
                    // another slot feeds): its third stream stays as lsx_create zeroed it
                    if (tl.nP > 1) cr[(size_t)(3 * u + 2) * plane] = XCj[u];
                }
        }
    }
    }
}

// The same Gamma slabs for tiles whose fast continua form "simple" sets of at most LSX_FAST_NQ per atom (every bound-free
// set of an ordinary model atom), with the work laid out the other way round: TWO lanes per (column, depth), the tile's
// wavelength pairs dealt alternately to them.  The wavelength quadrature becomes a running sum in registers (one two-lane
// DPP add at the end instead of a row reduction per wavelength and slot: in k_fast_gamma the reductions cost as much as
// the arithmetic), the per-depth operands n_i, n_j, nStar_i/nStar_j stay in registers (no staging), every lane works (a
// row of LP = 16 lanes holds 12 wavelengths at 5 rays) and the slabs leave as coalesced stores.
// The streams J, Psibar, E, PsiPhi are [depth][wavelength] in memory: read per thread they would be 16-byte pieces at a
// stride of L doubles (48 cache lines per wave load, measured: the kernel then sits on the vector-memory path at a
// fifth of its arithmetic rate).  So each wave first copies the contiguous block of its 32 (column, depth) rows into a
// wave-private LDS area with coalesced 16-byte loads (both directions summed on the way, pad wavelengths zeroed) and the
// arithmetic reads it from there.  Same terms, same order per wavelength as k_fast_gamma; the sum over the wavelengths
// runs over the lane's pairs in ascending order, then lane 0 + lane 1.
// (LSX_FAST_NQ, LSX_FGC_ROWS, LSX_FGC_MAXF: lsx_plan.h -- the plan sizes this kernel's LDS and decides which tiles it takes)
// NPC: wavelength pairs per row known at compile time (6: twelve wavelengths, the five-ray tiling) -- the staging loop then has
// three rounds, unrolled, with unconditional loads: all of a wave's loads are in flight together instead of round after round
// As a device function: the NW waves of a block take the (column, depth) rows [row_begin + wave R, ... + R) below `nrows`
// (k_fast_gamma_cols: consecutive chunks of all columns' rows; the fused small-batch sweep: the rows of the workgroup's own column).
// BIG (round 5): the instances for tiles in which an atom has MORE than LSX_FAST_NQ fast continua, or the tile more than LSX_FGC_MAXF
// (carbon's and iron's fourteen and MgII's ten bound-free continua onto one level: with all five model atoms of the reference active
// 192 of 328 tiles with fast continua).  An atom's sums over its continua at a wavelength -- atom.U[j], atom.eta, atom.chi[j] and the
// linked lines' shares -- are formed FIRST, for the lane's six wavelengths, in a pass over the atom's continua in chunks of LSX_FAST_NQ;
// a second pass takes the continua chunk by chunk through the arithmetic of the plain instance with the sums read instead of formed.
// Same terms, same order of the sums over the continua and over the wavelengths.  Such tiles went through the row-mapped k_fast_gamma.
template <int NLC, int NPC, int NW, bool BIG = false>   // NLC: lines of the tile that linked continua feed (0: the tile has no linked continuum)
static __device__ __forceinline__ void fast_gamma_cols_rows(const FastParams& f, const int t, const long row_begin, const long nrows, double* sm)
{
    constexpr bool LINKS = NLC > 0;
    constexpr int NL1 = NLC > 0 ? NLC : 1, NST = 3 + NLC, R = LSX_FGC_ROWS, NT = NW * 64;
    constexpr int MAXF = BIG ? LSX_FGC_MAXF_BIG : LSX_FGC_MAXF;      // continua the LDS table area holds
    static_assert(!BIG || NPC == 6, "big-set instances: twelve wavelengths per tile");
    const DevTile tl = f.tiles[t];
    __syncthreads();                                          // (a block that comes back for more rows: the previous call's readers are done)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, Ns = f.Nspace, L = NPC > 0 ? 2 * NPC : f.L, NP = NPC > 0 ? NPC : L / 2;
    const DevSlot* fs = f.slots + tl.slot0 + tl.nP;
    const DevSlot* ls = f.slots + tl.slot0;
    double* sA = sm;                                          // [q][j]{alpha, wlambda}, 0 where the continuum is inactive
    double* sU = sA + (size_t)2 * MAXF * L;                    // [j] 2hc/lambda^3
    double* sLW = sU + L;                                      // [u < 2][j] the linked lines' wavelength weights (0 outside the line)
    [[maybe_unused]] double* sAE = sLW + 2 * L;                // LSX_EPI_ELANE: [j] -hc / (k lambda), the Boltzmann factor's per-wavelength constant,
    [[maybe_unused]] double* sET = sAE + L;                    // and [LSX_EXP_TAB] the exponential's table (lsx_plan.h, LSX_FGC_EXTRA)
    double* sS = sLW + 2 * L + LSX_FGC_EXTRA(L) + (size_t)wv * NST * R * L;     // this wave's streams: [J | Psibar | E | PsiPhi_u][row][j]
    // The tile's tables -- cross-section and wavelength weight of every fast continuum (0 where it is not active), 2hc/lambda^3, the
    // linked lines' wavelength weights -- come ready made from lsx_create (f.fgtab, an image of this LDS area per tile): ONE coalesced
    // copy.  (Round 5.  Gathered here from the slot table, the activity table and the atoms' arrays they were a chain of four
    // dependent small loads at the head of every workgroup: profiles/r05/ablation_epilogue_kernel.txt.)
    {
        const double* tab = f.fgtab + (size_t)t * LSX_FGC_TAB(L);
        const int nA = tl.nF * L, nB = (LSX_EPI_ELANE ? 4 : 3) * L / 2;      // double2 pieces: [q][j]{alpha, wlambda} | u, the two lines' weights (, the Boltzmann constants)
#if LSX_EPI_ELANE
        if (f.temperature) for (int e = tid; e < LSX_EXP_TAB; e += NT) sET[e] = f.exp2_tab[e];
#endif
        for (int e = tid; e < nA + nB; e += NT) {       // (the image in memory is laid out for LSX_FGC_MAXF_BIG continua, the LDS area for MAXF)
            const int os = e < nA ? 2 * e : 2 * LSX_FGC_MAXF_BIG * L + 2 * (e - nA), od = e < nA ? 2 * e : 2 * MAXF * L + 2 * (e - nA);
            *reinterpret_cast<double2*>(sm + od) = *reinterpret_cast<const double2*>(tab + os);
        }
    }
    __syncthreads();
    const long row0 = row_begin + (long)wv * R;               // first (column, depth) row of this wave
    if (row0 >= nrows) return;
    const int nLc = LINKS ? min(tl.nL, NLC) : 0;
    const size_t dstride = (size_t)f.ncol * f.ntile * Ns * L, pstride = (size_t)f.ncol * f.pp_col_stride, plane = (size_t)Ns * L;
    // ---- the wave's rows, global -> LDS: piece c = 16 bytes = wavelength pair c % NP of row c / NP
    const double2 zero2 = make_double2(0.0, 0.0);
    auto stage = [&](const int c, auto masked) __attribute__((always_inline)) {
        const int r = c / NP, p = c - r * NP;
        const long g0 = row0 + r;
        // a row past the batch or of a frozen column reads the batch's last row (any valid address) and stores zeros; so do the pad
        // wavelengths, which hold nothing defined
        const unsigned g = (unsigned)(g0 < nrows ? g0 : nrows - 1);       // columns x depths < 2^31 (checked with the launch shapes)
        const int col = (int)(g / (unsigned)Ns), k = (int)(g - (unsigned)col * (unsigned)Ns);
        bool live = g0 < nrows;
        if constexpr (decltype(masked)::value) {       // (an unconditional load, no short circuit: no branch between the rounds)
            const bool on = f.colmask[col] != 0;
            live = live & on;
        }
        const bool kx = live && 2 * p < tl.nla, ky = live && 2 * p + 1 < tl.nla;
        const size_t o = ((size_t)((size_t)col * f.ntile + t) * Ns + k) * L + 2 * p;
#ifdef LSX_NT_EPI     // (measured alternative, profiles/r06_bound_evidence.md 1: the epilogue's read-once streams non-temporally)
        typedef double nt_d2 __attribute__((ext_vector_type(2)));
        auto ldnt = [](const double* q) __attribute__((always_inline)) { const nt_d2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_d2*>(q)); return make_double2(v.x, v.y); };
        const double2 vJ = ldnt(f.J_T + o);
        const double2 a = ldnt(f.Psi2_T + o), b = ldnt(f.Psi2_T + dstride + o);
#else
        const double2 vJ = *reinterpret_cast<const double2*>(f.J_T + o);
        const double2 a = *reinterpret_cast<const double2*>(f.Psi2_T + o), b = *reinterpret_cast<const double2*>(f.Psi2_T + dstride + o);
#endif
        double2 vE;
#if LSX_EPI_ELANE
        if (f.temperature) {       // (wave-uniform) exp(-hc / k lambda T) of the pair's two wavelengths: the bits k_build_E writes into E_T
            const double rT = 1.0 / f.temperature[(size_t)col * Ns + k];
            vE = make_double2(boltzmann_factor(sAE[2 * p], rT, (const lds_f64*)sET), boltzmann_factor(sAE[2 * p + 1], rT, (const lds_f64*)sET));
        } else
#endif
        vE = *reinterpret_cast<const double2*>(f.E_T + o);
        double2 vL[NL1];
#pragma unroll
        for (int u = 0; u < NL1; ++u) {
            vL[u] = zero2;
            if (LINKS && u < nLc) {
                const double* pp = f.Psi3_T + (size_t)col * f.pp_col_stride + tl.pp_off + (size_t)u * plane + (size_t)k * L + 2 * p;
#ifdef LSX_NT_EPI
                const double2 x = ldnt(pp), y = ldnt(pp + pstride);
#else
                const double2 x = *reinterpret_cast<const double2*>(pp), y = *reinterpret_cast<const double2*>(pp + pstride);
#endif
                vL[u] = make_double2(x.x + y.x, x.y + y.y);
            }
        }
        *reinterpret_cast<double2*>(sS + (size_t)c * 2) = make_double2(kx ? vJ.x : 0.0, ky ? vJ.y : 0.0);
        *reinterpret_cast<double2*>(sS + (size_t)R * L + (size_t)c * 2) = make_double2(kx ? a.x + b.x : 0.0, ky ? a.y + b.y : 0.0);
        *reinterpret_cast<double2*>(sS + (size_t)2 * R * L + (size_t)c * 2) = make_double2(kx ? vE.x : 0.0, ky ? vE.y : 0.0);
#pragma unroll
        for (int u = 0; u < NL1; ++u)
            if (LINKS) *reinterpret_cast<double2*>(sS + (size_t)(3 + u) * R * L + (size_t)c * 2) = make_double2(kx ? vL[u].x : 0.0, ky ? vL[u].y : 0.0);
    };
    if constexpr (NPC > 0 && (R * NPC) % 64 == 0) {
        if (f.colmask) {
#pragma unroll
            for (int it = 0; it < R * NPC / 64; ++it) stage(lane + 64 * it, std::true_type{});
        } else {
#pragma unroll
            for (int it = 0; it < R * NPC / 64; ++it) stage(lane + 64 * it, std::false_type{});
        }
    } else {
        for (int c = lane; c < R * NP; c += 64) { if (f.colmask) stage(c, std::true_type{}); else stage(c, std::false_type{}); }
    }
    __builtin_amdgcn_wave_barrier();                          // a wave's LDS operations complete in order
    // ---- arithmetic: lane = (row, h), h = which of the row's wavelength pairs
    const int r = lane >> 1, h = lane & 1;
    const long g = row0 + r;
    if (g >= nrows) return;
    const int col = (int)((unsigned)g / (unsigned)Ns), k = (int)((unsigned)g - (unsigned)col * (unsigned)Ns);
    if (f.colmask && !f.colmask[col]) return;
    double sW = 0.0;
    for (int m = 0; m < f.Nrays; ++m) sW += 2.0 * f.wmuh[m] * (4.0 * M_PI);   // both directions
    const double* nc = f.n + (size_t)col * f.NLtot * Ns + k;
    const double* nsrc = f.nsr + (size_t)col * f.Ncont * Ns + k;
    // the lines' depth coefficients: chi_line = Lx phi, eta_line = Ly phi (rh_method.py:279-281, 613-614), Uji_line = LU phi
    double Lx[NL1], Ly[NL1], LU[NL1];
#pragma unroll
    for (int u = 0; u < NL1; ++u) {
        Lx[u] = Ly[u] = LU[u] = 0.0;
        if (LINKS && u < nLc) {
            const double ni = nc[(size_t)ls[u].li * Ns], nj = nc[(size_t)ls[u].lj * Ns];
            Lx[u] = ls[u].cB * (ni - ls[u].g * nj);
            Ly[u] = nj * ls[u].Uc;
            LU[u] = ls[u].Uc;
        }
    }
    const double* srow = sS + (size_t)r * L;
    // what the linked continua add to line u's rates, summed over this lane's wavelengths: dA = sum w_line EC sPP, dB = sum w_line XC_i sPP
    // (EC, XC_i: the continua's share of atom.eta and atom.chi[i_line]; the sweep's own terms are phi [Uc (1 - Psi* chi_i) + Vc Ieff],
    // cB phi Ieff with Ieff = I - Psi* eta: the shares enter as -EC sum_mu w Psi* phi and -XC_i sum_mu w Psi* phi)
    double dA[NL1], dB[NL1];
#pragma unroll
    for (int u = 0; u < NL1; ++u) dA[u] = dB[u] = 0.0;
    for (int q0 = 0; q0 < tl.nF;) {                           // one atom at a time
        const int atom = fs[q0].atom;
        int q1 = q0;
        while (q1 < tl.nF && fs[q1].atom == atom) ++q1;
        const int nq = q1 - q0;                               // <= LSX_FAST_NQ (lsx_create) -- unless BIG
        if constexpr (BIG) {
            if (nq > LSX_FAST_NQ) {
                constexpr int NQ = NLC == 0 ? LSX_FAST_NQ : (NLC == 1 ? 3 : 4), NPL = NPC / 2, NWL = NPC;      // continua per chunk (fewer where the lines' sums take registers too: what fits 256 registers without scratch); pairs and wavelengths of a lane: pairs h, h + 2, h + 4
                const unsigned lk0 = fs[q0].lkbits;
                auto operands = [&](int c0, int nqc, double (&ni)[NQ], double (&nj)[NQ], double (&nr)[NQ]) __attribute__((always_inline)) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        ni[q] = nj[q] = nr[q] = 0.0;
                        if (q < nqc) {
                            ni[q] = nc[(size_t)fs[c0 + q].li * Ns];
                            nj[q] = nc[(size_t)fs[c0 + q].lj * Ns];
                            nr[q] = nsrc[(size_t)fs[c0 + q].base];
                        }
                    }
                };
                // ---- the atom's sums at the lane's wavelengths: U_a[j] / u, eta_a / u, -chi_a[j], chi of the continua on line u's lower level
                // (the continua's share of chi_a[i_line] enters the line's correction dB = sum_w t_w sum_q h_q alpha_q only: it is added up
                // term by term here, not kept per wavelength -- six registers per line less; the plain instance sums over q first)
                double Us[NWL], Es[NWL], Cs[NWL];
#pragma unroll
                for (int wi = 0; wi < NWL; ++wi) Us[wi] = Es[wi] = Cs[wi] = 0.0;
                for (int c0 = q0; c0 < q1; c0 += NQ) {
                    const int nqc = min(NQ, q1 - c0);
                    double ni[NQ], nj[NQ], nr[NQ];
                    operands(c0, nqc, ni, nj, nr);
#pragma unroll
                    for (int i = 0; i < NPL; ++i) {
                        const int p = h + 2 * i;
                        const double2 E2 = *reinterpret_cast<const double2*>(srow + (size_t)2 * R * L + 2 * p);
                        double2 L2[NL1];
#pragma unroll
                        for (int u = 0; u < NL1; ++u) L2[u] = LINKS ? *reinterpret_cast<const double2*>(srow + (size_t)(3 + u) * R * L + 2 * p) : zero2;
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            const int jw = 2 * p + w, wi = 2 * i + w;
                            const double E = w ? E2.y : E2.x;
                            double tt[NL1];
#pragma unroll
                            for (int u = 0; u < NL1; ++u) tt[u] = (LINKS && f.epi_corr) ? sLW[u * L + jw] * (w ? L2[u].y : L2[u].x) : 0.0;
#pragma unroll
                            for (int q = 0; q < NQ; ++q) {
                                if (q < nqc) {
                                    const double alf = sA[(size_t)((c0 + q) * L + jw) * 2];
                                    const double g = nr[q] * E, ng = nj[q] * g;
                                    Us[wi] = fma(g, alf, Us[wi]);
                                    Es[wi] = fma(ng, alf, Es[wi]);
                                    if constexpr (LINKS) {
                                        const double hq = ni[q] - ng;
                                        Cs[wi] = fma(hq, alf, Cs[wi]);
                                        const unsigned lkq = fs[c0 + q].lkbits;             // (wave-uniform)
#pragma unroll
                                        for (int u = 0; u < NL1; ++u)
                                            if ((lkq & (2u << (8 * u))) && (lk0 & (1u << (8 * u)))) dB[u] = fma(tt[u], hq * alf, dB[u]);
                                    }
                                }
                            }
                        }
                    }
                }
                // ---- the continua, chunk by chunk; the linked lines' corrections with the first chunk
                for (int c0 = q0; c0 < q1; c0 += NQ) {
                    const int nqc = min(NQ, q1 - c0);
                    double ni[NQ], nj[NQ], nr[NQ], a1[NQ], a2[NQ];
                    operands(c0, nqc, ni, nj, nr);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) a1[q] = a2[q] = 0.0;
#pragma unroll
                    for (int i = 0; i < NPL; ++i) {
                        const int p = h + 2 * i;
                        const double2 J2 = *reinterpret_cast<const double2*>(srow + 2 * p), P2 = *reinterpret_cast<const double2*>(srow + (size_t)R * L + 2 * p),
                                      E2 = *reinterpret_cast<const double2*>(srow + (size_t)2 * R * L + 2 * p), U2 = *reinterpret_cast<const double2*>(sU + 2 * p);
                        double2 L2[NL1];
#pragma unroll
                        for (int u = 0; u < NL1; ++u) L2[u] = LINKS ? *reinterpret_cast<const double2*>(srow + (size_t)(3 + u) * R * L + 2 * p) : zero2;
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            const int jw = 2 * p + w, wi = 2 * i + w;
                            const double sI = (w ? J2.y : J2.x) * (4.0 * M_PI), sPsi = w ? P2.y : P2.x, E = w ? E2.y : E2.x, ula = w ? U2.y : U2.x;
                            double tchi[NL1], teta[NL1], tU[NL1];
#pragma unroll
                            for (int u = 0; u < NL1; ++u) {
                                const double sPP = w ? L2[u].y : L2[u].x;
                                tchi[u] = Lx[u] * sPP;
                                teta[u] = Ly[u] * sPP;
                                tU[u] = LU[u] * sPP;
                            }
                            double le = 0.0;
                            if constexpr (LINKS) {
#pragma unroll
                                for (int u = 0; u < NL1; ++u)
                                    if (lk0 & (1u << (8 * u))) le += teta[u];
                            }
                            const double U_j = ula * Us[wi], etaA = ula * Es[wi];
                            if constexpr (LINKS) {
                                if (f.epi_corr && c0 == q0) {
#pragma unroll
                                    for (int u = 0; u < NL1; ++u)
                                        if (lk0 & (1u << (8 * u))) dA[u] = fma(sLW[u * L + jw] * (w ? L2[u].y : L2[u].x), etaA, dA[u]);
                                }
                            }
                            const double sIe = (sI - etaA * sPsi) - le;
                            const double T = fma(ula, sW, sIe), UP = U_j * sPsi;
#pragma unroll
                            for (int q = 0; q < NQ; ++q) {
                                if (q < nqc) {
                                    const double2 A = *reinterpret_cast<const double2*>(sA + (size_t)((c0 + q) * L + jw) * 2);
                                    const double g = nr[q] * E, hq = ni[q] - nj[q] * g;
                                    const double wa = A.x * A.y;
                                    const double tt = fma(-hq, UP, g * T);
                                    a1[q] = fma(wa, tt, a1[q]);
                                    a2[q] = fma(wa, sIe, a2[q]);
                                    if constexpr (LINKS) {
                                        const unsigned lk = fs[c0 + q].lkbits;
                                        double lc = 0.0, lU = 0.0;
#pragma unroll
                                        for (int u = 0; u < NL1; ++u) {
                                            if (lk & (2u << (8 * u))) lc += tchi[u];
                                            if (lk & (4u << (8 * u))) { lc -= tchi[u]; lU += tU[u]; }
                                        }
                                        a1[q] = fma(-A.y, lc * U_j, a1[q]);
                                        a2[q] = fma(A.y, Cs[wi] * lU, a2[q]);
                                    }
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        if (q < nqc) {
                            const double s1 = a1[q] + dpp_f64<0xB1, 0xf>(a1[q]), s2 = a2[q] + dpp_f64<0xB1, 0xf>(a2[q]);
                            if (h == 0) {
                                double* gp = f.Gpart + (((size_t)col * f.nslot_total + tl.slot0 + tl.nP + c0 + q) * 4) * (size_t)Ns + k;
                                gp[0] = s1;
                                gp[2 * (size_t)Ns] = s2;
                            }
                        }
                    }
                }
                q0 = q1;
                continue;
            }
        }
        double ni[LSX_FAST_NQ], nj[LSX_FAST_NQ], nr[LSX_FAST_NQ], a1[LSX_FAST_NQ], a2[LSX_FAST_NQ];
#pragma unroll
        for (int q = 0; q < LSX_FAST_NQ; ++q) {
            ni[q] = nj[q] = nr[q] = a1[q] = a2[q] = 0.0;
            if (q < nq) {
                ni[q] = nc[(size_t)fs[q0 + q].li * Ns];
                nj[q] = nc[(size_t)fs[q0 + q].lj * Ns];
                nr[q] = nsrc[(size_t)fs[q0 + q].base];
            }
        }
        const unsigned lk0 = fs[q0].lkbits;
        for (int p = h; p < NP; p += 2) {
            const double2 J2 = *reinterpret_cast<const double2*>(srow + 2 * p), P2 = *reinterpret_cast<const double2*>(srow + (size_t)R * L + 2 * p),
                          E2 = *reinterpret_cast<const double2*>(srow + (size_t)2 * R * L + 2 * p), U2 = *reinterpret_cast<const double2*>(sU + 2 * p);
            double2 L2[NL1];
#pragma unroll
            for (int u = 0; u < NL1; ++u) L2[u] = LINKS ? *reinterpret_cast<const double2*>(srow + (size_t)(3 + u) * R * L + 2 * p) : zero2;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int jw = 2 * p + w;
                const double sI = (w ? J2.y : J2.x) * (4.0 * M_PI), sPsi = w ? P2.y : P2.x, E = w ? E2.y : E2.x, ula = w ? U2.y : U2.x;
                double tchi[NL1], teta[NL1], tU[NL1];
#pragma unroll
                for (int u = 0; u < NL1; ++u) {
                    const double sPP = w ? L2[u].y : L2[u].x;
                    tchi[u] = Lx[u] * sPP;
                    teta[u] = Ly[u] * sPP;
                    tU[u] = LU[u] * sPP;
                }
                auto line_chi = [&](unsigned lk) {
                    double x = 0.0;
                    if constexpr (LINKS) {
#pragma unroll
                        for (int u = 0; u < NL1; ++u) {
                            if (lk & (2u << (8 * u))) x += tchi[u];
                            if (lk & (4u << (8 * u))) x -= tchi[u];
                        }
                    }
                    return x;
                };
                auto line_U = [&](unsigned lk) {
                    double x = 0.0;
                    if constexpr (LINKS) {
#pragma unroll
                        for (int u = 0; u < NL1; ++u)
                            if (lk & (4u << (8 * u))) x += tU[u];
                    }
                    return x;
                };
                double le = 0.0;
                if constexpr (LINKS) {
#pragma unroll
                    for (int u = 0; u < NL1; ++u)
                        if (lk0 & (1u << (8 * u))) le += teta[u];
                }
                // rh_method.py:284-286, 453-455, 613-614 for the atom's continua; atom.chi[j], atom.U[j], atom.eta of :616-627
                // The same terms with the common factors taken out (round 4: 10 instead of 21 fp64 instructions per (depth, wavelength,
                // continuum) in a kernel that runs on the vector pipe).  With g = (nStar_i / nStar_j) E:
                //   Vji = g alpha,  Uji = u Vji,  chi = alpha (n_i - n_j g) = alpha h,   U_a[j] = u sum g alpha,  eta_a = u sum n_j g alpha
                //   w [(Uji sW + Vji sIe) - chi U_a[j] sPsi] = (w alpha) [g (u sW + sIe) - h (U_a[j] sPsi)],     w [alpha sIe] = (w alpha) sIe
                // -- no difference is formed that the reference's own expression does not form.
                double g[LSX_FAST_NQ], hq[LSX_FAST_NQ], Usum = 0.0, Esum = 0.0, Csum = 0.0, XCi[NL1];
#pragma unroll
                for (int u = 0; u < NL1; ++u) XCi[u] = 0.0;
#pragma unroll
                for (int q = 0; q < LSX_FAST_NQ; ++q) {
                    g[q] = hq[q] = 0.0;
                    if (q < nq) {
                        const double alf = sA[(size_t)((q0 + q) * L + jw) * 2];
                        g[q] = nr[q] * E;
                        const double ng = nj[q] * g[q];
                        hq[q] = ni[q] - ng;
                        Usum = fma(g[q], alf, Usum);
                        Esum = fma(ng, alf, Esum);
                        if constexpr (LINKS) {
                            Csum = fma(hq[q], alf, Csum);                       // = -atom.chi[j]
                            const unsigned lkq = fs[q0 + q].lkbits;             // (wave-uniform)
#pragma unroll
                            for (int u = 0; u < NL1; ++u)
                                if (lkq & (2u << (8 * u))) XCi[u] = fma(hq[q], alf, XCi[u]);      // chi of the continua on line u's lower level
                        }
                    }
                }
                const double U_j = ula * Usum, etaA = ula * Esum;
                if constexpr (LINKS) {
                    if (f.epi_corr) {
#pragma unroll
                        for (int u = 0; u < NL1; ++u)
                            if (lk0 & (1u << (8 * u))) {                        // line u belongs to this atom: its EC is this atom's eta
                                const double t = sLW[u * L + jw] * (w ? L2[u].y : L2[u].x);
                                dA[u] = fma(t, etaA, dA[u]);
                                dB[u] = fma(t, XCi[u], dB[u]);
                            }
                    }
                }
                const double sIe = (sI - etaA * sPsi) - le;
                const double T = fma(ula, sW, sIe), UP = U_j * sPsi;
#pragma unroll
                for (int q = 0; q < LSX_FAST_NQ; ++q) {
                    if (q < nq) {
                        const double2 A = *reinterpret_cast<const double2*>(sA + (size_t)((q0 + q) * L + jw) * 2);
                        const double wa = A.x * A.y;
                        const double t = fma(-hq[q], UP, g[q] * T);
                        a1[q] = fma(wa, t, a1[q]);
                        a2[q] = fma(wa, sIe, a2[q]);
                        if constexpr (LINKS) {
                            const unsigned lk = fs[q0 + q].lkbits;
                            a1[q] = fma(-A.y, line_chi(lk) * U_j, a1[q]);       // the lines on the continuum's lower level: chi_a[i] Psi U_a[j]
                            a2[q] = fma(A.y, Csum * line_U(lk), a2[q]);         // -chi_a[j] (Psi U_a[i]), U_a[i] from the lines that end there
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < LSX_FAST_NQ; ++q) {
            if (q < nq) {                                     // wave-uniform
                const double s1 = a1[q] + dpp_f64<0xB1, 0xf>(a1[q]), s2 = a2[q] + dpp_f64<0xB1, 0xf>(a2[q]);   // lane 0 + lane 1 of the row
                if (h == 0) {
                    double* gp = f.Gpart + (((size_t)col * f.nslot_total + tl.slot0 + tl.nP + q0 + q) * 4) * (size_t)Ns + k;
                    gp[0] = s1;
                    gp[2 * (size_t)Ns] = s2;
                }
            }
        }
        q0 = q1;
    }
    if constexpr (LINKS) {
        // the correction slots of the tile's lines (DevTile.nX): what the linked continua add to Gamma[i][j] and Gamma[j][i] of line u,
        // -wphi [Uc dB + Vc dA] and -wphi cB dA (the sweep's wavelength weight is 4 pi w_line wphi; the 4 pi sits in sPP); zeros where
        // the sweep has applied the corrections itself
#pragma unroll
        for (int u = 0; u < NL1; ++u) {
            if (u < nLc) {
                double x1 = 0.0, x2 = 0.0;
                if (f.epi_corr) {
                    const double wp = f.wphi[((size_t)col * f.Nlines) * Ns + ls[u].wphi_off + k];
                    x1 = -wp * fma(ls[u].Uc, dB[u], ls[u].Vc * dA[u]);
                    x2 = -wp * (ls[u].cB * dA[u]);
                }
                const double s1 = x1 + dpp_f64<0xB1, 0xf>(x1), s2 = x2 + dpp_f64<0xB1, 0xf>(x2);
                if (h == 0) {
                    double* gp = f.Gpart + (((size_t)col * f.nslot_total + tl.slot0 + tl.nP + tl.nF + u) * 4) * (size_t)Ns + k;
                    gp[0] = s1;
                    gp[2 * (size_t)Ns] = s2;
                }
            }
        }
    }
}


} // namespace
