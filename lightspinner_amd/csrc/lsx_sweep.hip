// lsx_sweep.hip -- the hot kernel: eta/chi/U-V build + piecewise-linear short-characteristics
// sweep + Psi*/Gamma/J accumulation, fused, for gfx950 (CDNA4, wave64).
//
// Restates, per (column, wavelength, ray):
//   rh_method.py:595-692   (loop body of Context.formal_sol_gamma_matrices)
//   rh_method.py:245-288   (uv), 425-455 (setup_wavelength)
//   formal_solver.py:14-212 (w2, piecewise_1d_impl, piecewise_linear_1d)
//   utils.py:17-22         (planck, lower boundary condition)
//
// Work decomposition (MI355X-first, not the reference's loop nest):
//   * one LANE = one ray (wavelength, mu); one WAVEFRONT = one direction of a TILE of
//     L = 64/Nrays consecutive wavelengths x all Nrays angles of ONE column (lane = mu*L + j);
//     one WORKGROUP = the two directions of that tile (wave 0 sweeps down, wave 1 sweeps up).
//     The depth recurrence is serial inside a lane; the depth index k is WAVE-UNIFORM, so
//     every per-(transition, depth) quantity (level populations, line normalisation, geometry)
//     is fetched through the scalar cache and costs no vector registers.
//   * per-lane state is one ray (I_upwind, chi, S, dtau of the previous depth): ~8 VGPRs;
//     occupancy, not unrolling, hides HBM latency.
//   * all per-(lambda, depth) inputs are depth-major in HBM: at a fixed depth the L lanes of one
//     angle read L consecutive doubles; every input byte is read once per call.
//   * angle quadrature (J, and the ray sums the fast continua need): Nrays-lane strided sum
//     through one LDS row.  Wavelength+angle quadrature of Gamma: one wave reduction (DPP) per
//     (transition, depth); lane 63 stores it into a per-(tile, transition, direction, depth)
//     slab.  Every slab element is written exactly once -> no atomics, bitwise reproducible,
//     independent of how columns are distributed over GPUs.
//   * the per-level "effective opacity"/U bookkeeping of overlapping transitions (atom.chi /
//     atom.U / atom.eta, rh_method.py:616-627) lives in lane-private LDS cells, and only where two
//     transitions of the tile really share a level or an atom (host-computed flags).
//   * transitions of a tile come in two kinds.  PER-RAY slots (lines, and continua of an atom
//     that has a line in the tile) go through both passes here.  FAST continua (atoms with no line
//     in the tile) are ray independent and never enter this kernel: a pre-pass (k_fast_prepass)
//     folds their opacity/emissivity into an effective background, and because their Gamma
//     integrand is affine in I and Psi* with ray-independent coefficients, a post-pass
//     (k_fast_gamma) forms it from J = sum w I and Psibar = sum w Psi*, which this kernel stores
//     per direction for the tiles that have such continua.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "lsx_dev.h"
#include "lsx_plan.h"
#include "lsx_fast.h"

namespace {

// constants.py:1-27
constexpr double kCLight = 2.99792458E+08;
constexpr double kHPlanck = 6.6260755E-34;
constexpr double kKBoltzmann = 1.380658E-23;
constexpr double kNM_TO_M = 1.0E-09;
constexpr double kHC = kHPlanck * kCLight;
constexpr double kPi = 3.14159265358979323846;

// wave-uniform data is read through the constant address space: the compiler then uses scalar
// loads (s_load) and scalar address arithmetic even though the kernel also stores to global
// memory (a plain global pointer would be treated as possibly clobbered -> vector loads).
// Legal because nothing read this way is written while the kernel runs.
#define LSX_CONST(T, ptr) ((const __attribute__((address_space(4))) T*)(ptr))

// utils.py:17-22
__device__ __forceinline__ double planck(double temp, double wav)
{
    const double hc_Tkla = kHC / (kKBoltzmann * kNM_TO_M * wav) / temp;
    const double x = kNM_TO_M * wav;
    const double twohnu3_c2 = (2.0 * kHC) / (x * x * x);
    return twohnu3_c2 / (exp(hc_Tkla) - 1.0);
}

// Two values reduced for the price of one: v_permlane32_swap exchanges the upper half of `a` with
// the lower half of `b`, so ONE add folds both vectors to 32 partial sums each (a's in lanes 0-31,
// b's in lanes 32-63).  Result: lane 31 = sum(a), lane 63 = sum(b).
__device__ __forceinline__ double fold32(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double reduce_pair(double a, double b)   // lane 31: sum(a), lane 63: sum(b)
{
    double t = row_sums(fold32(a, b));
    t += dpp_f64<0x142, 0xf>(t);      // row_bcast15 into every row (row 0 receives 0): lanes 31 / 63 hold rows 0+1 / 2+3; no `old` copy
    return t;
}
// Four values: a second fold with v_permlane16_swap (odd rows of the first operand <-> even rows of
// the second) leaves one value per 16-lane row.  Result: lane 15 = sum(a), lane 31 = sum(c),
// lane 47 = sum(b), lane 63 = sum(d).
__device__ __forceinline__ double reduce_quad(double a, double b, double c, double d)
{
    const double ab = fold32(a, b), cd = fold32(c, d);
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(ab), __double2loint(cd), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(ab), __double2hiint(cd), false, false);
    return row_sums(__hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]));
}

// element at a 32-bit BYTE offset from a wave-uniform base: one scalar base + one 32-bit vector offset
// (global_load ... v_off, s[base]) instead of 64-bit vector address arithmetic per access
__device__ __forceinline__ const double& at(const double* base, unsigned byte_off)
{
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ double& at(double* base, unsigned byte_off)
{
    return *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + byte_off);
}

__device__ __forceinline__ double nanmax(double a, double b)
{
    // max that propagates NaN like numpy's ndarray.max (rh_method.py:706)
    return (a != a || b != b) ? __builtin_nan("") : fmax(a, b);
}

__device__ __forceinline__ double wave_max_nan(double v)
{
#pragma unroll
    for (int m = 1; m < LSX_WAVE; m <<= 1) v = nanmax(v, __shfl_xor(v, m, LSX_WAVE));
    return v;
}

// lane-private LDS cell: first writer of a pass stores, later writers add (DS add, no return)
__device__ __forceinline__ void cell_acc(lds_f64* p, double v, bool first)
{
    if (first) *p = v;
    else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

} // namespace

#ifndef LSX_WAVES_PER_EU
#define LSX_WAVES_PER_EU 4
#endif
// wave-uniform copy of a slot's parameters (one batch of scalar loads, then registers)
struct SlotS {
    int flags, noff_i, noff_j, ci, cj, ca, Nblue, Nlam, base, first, len, wl_off, wphi_off, trans;
    double cB, g, Vc, Uc;
};
__device__ __forceinline__ SlotS load_slot(const __attribute__((address_space(4))) DevSlot* q, int Ns)
{
    SlotS r;
    r.flags = q->flags; r.noff_i = q->li * Ns; r.noff_j = q->lj * Ns;
    r.ci = q->ci; r.cj = q->cj; r.ca = q->ca; r.Nblue = q->Nblue; r.Nlam = q->Nlam; r.base = q->base; r.first = q->first; r.len = q->len;
    r.wl_off = q->wl_off; r.wphi_off = q->wphi_off; r.trans = q->trans;
    r.cB = q->cB; r.g = q->g; r.Vc = q->Vc; r.Uc = q->Uc;
    return r;
}

// NPT >= 0: the tile's per-ray slot count as a compile-time constant: slot state lives in registers,
// every load of a depth step is issued in one batch at the top of the step (one wait), the two
// passes are pure VALU + LDS.  NPT < 0: generic tile (runtime slot loops, loads in place).
// NR > 0: number of rays per wavelength as a compile-time constant (angle sums unroll into independent
// LDS reads); SCAL: the scattering coefficient may be wavelength dependent (vector load) -- otherwise
// it is one scalar per depth.
// NL: how many of the NPT per-ray slots are lines (they come first, lsx_create) -- the slot kind is then a
// compile-time property of the unrolled slot index and only one of the two formula sets is emitted.
// LK: the tile has linked continua (lsx_dev.h, SLOT_LINKED): per line three more streams come in (the continua's share of
// atom.eta, atom.chi[i], atom.chi[j], written by k_fast_prepass) and one more angle sum goes out (sum_mu w Psi* phi).
// TOPO (two-slot tiles): how the two transitions are related, when it is one of the two common cases -- 1: same atom, common
// LOWER level, nothing else shared (Ca II H & K); 2: nothing shared (lines of different atoms); 0: anything else (factors
// from the slot table).  The known cases drop the terms that vanish (atom.U[i] = 0, atom.chi[j] = -chi, atom.U[j] = Uji).
template <int NPT, int NL, int NR, bool SCAL, bool LK, int TOPO = 0>
__device__ __forceinline__ void sweep_tile(const SweepParams& p, const int vb, const int tile_id)
{
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
#if defined(LSX_STAMPS) || defined(LSX_CLOCK)
    unsigned long long tk_entry;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk_entry)::"memory");
#endif
    lds_f64* const etab = (lds_f64*)lds_raw;            // [64][2] exp table (16-byte aligned pairs)
    lds_f64* const lds = etab + LSX_EXP_TAB;
    constexpr bool STATIC = NPT >= 0;
    constexpr int NS = NPT > 0 ? NPT : 1;
    const int lane = threadIdx.x & (LSX_WAVE - 1);
    const int dir = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // 0: down (toFrom False), 1: up (True)
    const int ntile = p.ntile_total;
    const int col = vb / p.n_class_tiles;
    const auto* tilep = LSX_CONST(DevTile, p.tiles) + tile_id;
    const int la0 = tilep->la0, nla = tilep->nla, slot0 = tilep->slot0;
    const int nP = STATIC ? NPT : tilep->nP;
    const int nF = tilep->nF;
    const auto* slots = LSX_CONST(DevSlot, p.slots) + slot0;   // [0, nP): per-ray, [nP, nP+nF): fast
    const int Ns = p.Nspace;
    const int Nspect = p.Nspect;
    const int Nrays = NR > 0 ? NR : p.Nrays;
    constexpr int NRD = NR > 0 ? NR : 1;
    const int L = NR > 0 ? LSX_WAVE / NRD : p.L;

    // lane -> ray.  Lanes without a ray shadow a real one (finite arithmetic) and are masked out of
    // every store and reduction.
    const int mu_raw = lane / L;
    const int j_raw = lane - mu_raw * L;
    const bool valid = mu_raw < Nrays && j_raw < nla;
    const int mu = mu_raw < Nrays ? mu_raw : Nrays - 1;
    const int j = j_raw < nla ? j_raw : nla - 1;
    const int la = la0 + j;
    const bool lead = valid && mu_raw == 0; // one lane per wavelength: owns J[la, k]

    // LDS layout (lsx_plan.h, lsx_sweep_lds): rows of 64 doubles private to this wave, then the two cross-wave exchange rows
    constexpr int TR = STATIC ? 3 * NPT + 2 : 1;
    const SweepLds lay = lsx_sweep_lds(STATIC ? NPT : -1, LK, p.Nspace, p.ncell_lev, p.ncell_atom);
    const int rows = lay.rows;
    lds_f64* const wrow = lds + (size_t)dir * rows * LSX_WAVE + lane;
#define CCHI(c) wrow[(2 * (c)) * LSX_WAVE]
#define CU(c) wrow[(2 * (c) + 1) * LSX_WAVE]
#define CETA(a) wrow[(2 * p.ncell_lev + (a)) * LSX_WAVE]
    lds_f64* const xrow = lds + (size_t)dir * rows * LSX_WAVE + (size_t)(rows - 1) * LSX_WAVE; // angle sums
    lds_f64* const xwg = etab + lay.xwg;                                                       // [2][64] cross-wave
    // static path: per-depth wave-uniform operands of the tile (n_i, n_j, wphi per slot, z, sigma) are
    // staged once per workgroup as a depth-major table utab[k][TR] -> one address register, immediate
    // offsets, counted LDS waits (scalar-cache loads return out of order and serialise on lgkmcnt(0))
    lds_f64* const utab = etab + lay.utab;

    // column bases; wave-uniform reads go through the scalar cache
    const auto* n_col = LSX_CONST(double, p.n + (size_t)col * p.NLtot * Ns);
    const auto* wphi_col = LSX_CONST(double, p.wphi + (size_t)col * p.Nlines * Ns);
    const auto* z = LSX_CONST(double, p.height + (size_t)col * Ns);
    const auto* tcol = LSX_CONST(double, p.temperature + (size_t)col * Ns);
    // tile-major streams of this (column, tile): [k][j]
    const size_t tbase = ((size_t)col * ntile + tile_id) * Ns * L;
    // tiles with fast continua read the effective background written by k_fast_prepass
    const double* __restrict__ bgchi = (nF > 0 ? p.bgxchi_T : p.bgchi_T) + tbase;
    const double* __restrict__ bgeta = (nF > 0 ? p.bgxeta_T : p.bgeta_T) + tbase;
    double* __restrict__ psibar = p.Psi2_T + ((size_t)dir * p.ncol * ntile) * Ns * L + tbase;   // [dir][col][tile][k][j]
    const double* __restrict__ Jdag = p.Jdag_T + tbase;
    double* __restrict__ Jnew = p.Jnew_T + tbase;
    const bool sca_l = SCAL && p.sca_per_lambda;
    const double* __restrict__ sca = sca_l ? p.sca + tbase : p.sca + (size_t)col * Ns;
    const int PG = p.phi_G, cphi = col % PG;                       // the grouped line-profile store (lsx_dev.h, phi_elem)
    const double* __restrict__ phi_col = p.phi_T + (size_t)(col - cphi) * p.phi_col_stride;      // the group's first element
    const double* __restrict__ Eb = p.E_T + tbase;                    // Boltzmann factor of the continuum g_ij, [k][j]
    const auto* nsr_col = LSX_CONST(double, p.nsr + (size_t)col * p.Ncont * Ns);
    // linked continua: the tile's blocks of the correction streams [line][3][k][j] and of the Psi* phi sums [line][k][j]
    const int nLt = STATIC ? NL : tilep->nL;
    const size_t plane = (size_t)Ns * L;
    const double* __restrict__ corr = LK ? p.corr_T + (size_t)col * p.corr_col_stride + tilep->corr_off : nullptr;
    double* __restrict__ ppsum = LK ? p.Psi3_T + ((size_t)dir * p.ncol + col) * p.pp_col_stride + tilep->pp_off : nullptr;
    double* __restrict__ gpart = p.Gpart + ((size_t)col * p.nslot_total + slot0) * 4 * Ns;

    //   per slot: lines (cB (n_i - g n_j), n_j, wphi)   continua (n_i, n_j, nStar_i / nStar_j);   then the half length of the
    //   interval ABOVE depth k (0.5 |z[k-1] - z[k]|, formal_solver.py:123/129) and the scattering coefficient
    if constexpr (STATIC) {
        const double* ncolp = p.n + (size_t)col * p.NLtot * Ns;
        const double* zc = p.height + (size_t)col * Ns;
        if (threadIdx.x < TR) utab[Ns * TR + threadIdx.x] = 0.0;     // row Ns: the up sweep reads row k + 1
        for (int e = threadIdx.x; e < Ns; e += 2 * LSX_WAVE) {
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const double ni = ncolp[(size_t)slots[u].li * Ns + e], nj = ncolp[(size_t)slots[u].lj * Ns + e];
                const bool line = u < NL;
                utab[e * TR + 3 * u + 0] = line ? slots[u].cB * (ni - slots[u].g * nj) : ni;   // :279-280, :613
                utab[e * TR + 3 * u + 1] = nj;
                utab[e * TR + 3 * u + 2] = line ? p.wphi[(size_t)col * p.Nlines * Ns + slots[u].wphi_off + e]
                                                : p.nsr[(size_t)col * p.Ncont * Ns + slots[u].base + e];     // g_ij = this * E, :453
            }
            utab[e * TR + 3 * NPT + 0] = e > 0 ? 0.5 * fabs(zc[e - 1] - zc[e]) : 0.0;
            utab[e * TR + 3 * NPT + 1] = sca_l ? 1.0 : p.sca[(size_t)col * Ns + e];
        }
    }
    etab[threadIdx.x] = p.exp2_tab[threadIdx.x];          // 2 x 64 threads, 64 x 2 doubles
    // three- and four-slot tiles: the slot-pair factors of the level bookkeeping, [u][other o][5]
    lds_f64* const ctab = etab + lay.ctab;
    // linked tiles: one more exchange row per line and wave (behind the slot-pair factors)
    lds_f64* const xrow2 = etab + lay.xrow2 + (size_t)dir * NS * LSX_WAVE;
    if constexpr (NPT >= 3) {
        if (threadIdx.x < NPT * (NPT - 1) * 5) {
            const int u = threadIdx.x / ((NPT - 1) * 5), r = threadIdx.x % ((NPT - 1) * 5);
            ctab[threadIdx.x] = p.slots[slot0 + u].rel[r / 5][r % 5];
        }
    }
    __syncthreads();

    const double wav = p.wavelength[la];
    const double u_la = p.u_la[la];
    const double zmu_l = p.zmu[mu];
    const double wmuh_l = valid ? p.wmuh[mu] : 0.0;
    const double wq_l = wmuh_l * (4.0 * kPi);
    const bool compact = p.phi_compact != 0;
    const int kS = dir ? Ns - 1 : 0;
    const int dk = dir ? -1 : 1;
    // line-profile index inside a (tile, line) block: ((dir*Ns + k)*Nrays + mu) * len + l, i.e. one
    // contiguous stream per direction (compact profile: k*len + l)
    const int raysel = compact ? 0 : dir * Ns * Nrays + mu;
    const int kmul = compact ? 1 : Nrays;

    // activity bits of this lane's wavelength
    unsigned pact = 0;
    for (int u = 0; u < nP; ++u) {
        const int l = la - slots[u].Nblue;
        if (l >= 0 && l < slots[u].Nlam && p.active[slots[u].trans * Nspect + la] != 0) pact |= 1u << u;
    }

    // static path: per-slot lane state in registers
    //   idx0: element index of (depth 0, this ray, this wavelength) in phi_T (lines; continua read the tile's shared E stream)
    //   wl: wavelength quadrature weight, al: alpha (continua)
    int idx0[NS], kstr[NS];
    double wlv[NS], alv[NS];    // wlv: (w_mu/2 4pi) x wavelength weight of this ray, 0 where the transition is inactive
    if constexpr (STATIC) {
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const bool a = (pact >> u) & 1u;
            const int l = a ? la - slots[u].Nblue : 0;
            const int Nlam = slots[u].Nlam;
            const bool line = u < NL;
            const int len = slots[u].len;
            const int lb = a ? la - slots[u].first : 0;      // position inside the (tile, transition) block
            (void)Nlam;
            // lines: the lane's element of depth 0; a lane outside the line's range reads the column's zero pad at every depth
            idx0[u] = line ? (a ? PG * (slots[u].base + raysel * len) + cphi * len + lb : PG * (int)p.phi_col_stride - 1) : 0;      // continua read the tile's shared E stream
            kstr[u] = (line && a) ? PG * kmul * len : 0;
            wlv[u] = (a && valid) ? wq_l * p.wl[slots[u].wl_off + l] : 0.0;                // :451/:455, :665
            alv[u] = (a && !line) ? p.alpha[slots[u].wl_off + l] : 0.0;
        }
    }

    // total opacity at depth kk (boundary-condition look-ahead, formal_solver.py:204-207)
    auto chi_at = [&](int kk) -> double {
        double c = bgchi[kk * L + j];
        for (int u = 0; u < nP; ++u) {
            const SlotS sl = load_slot(slots + u, Ns);
            const double ni = n_col[sl.noff_i + kk];
            const double nj = n_col[sl.noff_j + kk];
            const bool a = (pact >> u) & 1u;
            const int l = a ? la - sl.Nblue : 0;
            const int lb = a ? la - sl.first : 0;
            if (sl.flags & SLOT_LINE) {
                const double pv = a ? phi_col[PG * (sl.base + (kk * kmul + raysel) * sl.len) + cphi * sl.len + lb] : 0.0;
                c += (sl.cB * (ni - sl.g * nj)) * pv;
            } else {
                const double g = a ? nsr_col[sl.base + kk] * Eb[kk * L + j] : 0.0;
                const double alf = a ? p.alpha[sl.wl_off + l] : 0.0;
                c += ni * alf - nj * (g * alf);
            }
        }
        return c;
    };

    // ---- boundary conditions: formal_solver.py:203-209 -------------------------------
    double Iu = 0.0;
    if (dir) {
        const double c0 = chi_at(kS), c1 = chi_at(kS + dk);
        const double dtau_uw = zmu_l * (c0 + c1) * 0.5 * fabs(z[kS] - z[kS + dk]);
        const double B0 = planck(tcol[Ns - 2], wav);
        const double B1 = planck(tcol[Ns - 1], wav);
        Iu = B1 - (B0 - B1) / dtau_uw;
    }

    double chi_prev = 1.0, S_prev = 0.0, dtau_prev = 1.0;
    double dJ = 0.0;
    double zprev = z[kS];
    // diagnostic build only (-DLSX_STAMPS): per-segment shader-clock totals of a few sample waves go
    // to p.debug, a buffer nothing else reads (cdna_hip_programming.md, In-kernel stamps)
    // -DLSX_CLOCK: only the two clock reads around the loop (the production instruction stream between them): the clock the
    // chip holds under this load = shader ticks / 100 MHz ticks
#if defined(LSX_STAMPS) || defined(LSX_CLOCK)
    unsigned long long T[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0, t1, tk0, tr0, tk1, tr1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk0), "=s"(tr0)::"memory");   // shader clock / 100 MHz clock
    t0 = tk0;
    (void)t0; (void)t1;
#endif
#ifdef LSX_STAMPS
#define STAMP(i)                                                                                          \
    do {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        T[i] += t1 - t0;                                                                                  \
        t0 = t1;                                                                                          \
    } while (0)
#else
#define STAMP(i)
#endif

    // static path: the HBM-streaming operands of a step (background, Jdag, the slots' profile / g_ij
    // values) are requested one depth ahead.  Memory returns in order, and the static path issues no
    // other vector load inside a step (slot constants sit in registers; the half-J read comes last),
    // so the requests of step s+1 are in flight during the arithmetic of step s.
    constexpr bool HASC = STATIC && NPT > NL;       // compile-time per-ray continua: they share the tile's E stream
    constexpr int NLK = (LK && NL > 0) ? NL : 1;
    // a tile with a single per-ray slot never reads atom.chi[j_line]: it multiplies U[i_line], which only another slot feeds
    constexpr int NCR = NPT == 1 ? 2 : 3;
    double n_bc = 0.0, n_be = 0.0, n_jd = 0.0, n_sv[NS], n_E = 0.0, n_cr[NLK][3];

    auto stream_loads = [&](int kk, double& bc, double& be, double& jdv, double (&v)[NS], double& Ev, double (&cr)[NLK][3]) {
        const unsigned kko = (unsigned)(kk * L + j) * 8u;
#ifdef LSX_ABL_NOLOAD     // diagnostic build (profiles/ablate.sh): what the step costs without its HBM streams; results are meaningless
        {
            const double f = 1.0 + 1e-3 * (double)(kko & 1023u);
            jdv = 1e-9 * f; bc = 1e-6 * f; be = 1e-15 * f;
            if constexpr (STATIC) {
#pragma unroll
                for (int u = 0; u < NL; ++u) v[u] = 1e-12 * f;
                if constexpr (HASC) Ev = 0.5 * f;
                if constexpr (LK) {
#pragma unroll
                    for (int u = 0; u < NL; ++u)
#pragma unroll
                        for (int q = 0; q < NCR; ++q) cr[u][q] = 1e-20 * f;
                }
            }
            return;
        }
#endif
        jdv = at(Jdag, kko);
        bc = at(bgchi, kko);
        be = at(bgeta, kko);
        if constexpr (STATIC) {
#pragma unroll
            for (int u = 0; u < NL; ++u)    // inactive lanes read the column's zero pad
                v[u] = at(phi_col, (unsigned)(idx0[u] + kk * kstr[u]) * 8u);
            if constexpr (HASC) Ev = at(Eb, kko);
            if constexpr (LK) {
#pragma unroll
                for (int u = 0; u < NL; ++u)
#pragma unroll
                    for (int q = 0; q < NCR; ++q) cr[u][q] = at(corr, (unsigned)((3 * u + q) * plane) * 8u + kko);
            }
        }
    };
    if constexpr (STATIC) stream_loads(kS, n_bc, n_be, n_jd, n_sv, n_E, n_cr);

#ifndef LSX_RED_LDS     // lane reduction by DPP / permlane trees, totals parked in 64-entry LDS rows (-DLSX_RED_LDS: the measured alternative)
    // Gamma totals wait in LDS, one 64-entry row per (slot, entry) and wave, until 64 depths can leave in one store.
    // The lanes that hold totals after a reduction (31 / 63, or 15 / 47 / 31 / 63) each own one row.
    lds_f64* const gpk = etab + lay.tb + (size_t)dir * (2 * NS) * LSX_WAVE;
    const bool own_pair = lane == 31 || lane == 63, own_quad = (lane & 15) == 15;
    const int row_pair = (lane >> 5) * LSX_WAVE;                                   // 31 -> row 0, 63 -> row 1
    const int row_quad = ((((lane >> 4) & 1) << 1) | (lane >> 5)) * LSX_WAVE;      // 15 -> 0, 47 -> 1, 31 -> 2, 63 -> 3

#else
    // Lane sums of the Gamma integrands: the lanes park their 2 NPT values of RT consecutive depth steps in this wave's
    // transposition buffer tb[step in batch][value][lane] (rows of LSX_RED_ROW doubles); at the end of a batch every lane adds
    // up one chunk of one row and a DPP tail over the chunk's lanes finishes (lsx_plan.h).  The first lane of a row's group
    // stores the total: each batch leaves as ONE store instruction with one lane per (depth, value).
    constexpr int RT = lsx_red_steps(NPT > 0 ? NPT : 1), NV = 2 * NS, RV = NV * RT;     // RV rows per batch (<= 8)
    constexpr int RVP = RV <= 4 ? 4 : 8, LPV = LSX_WAVE / RVP;                           // lanes per row; doubles per lane = RVP
    lds_f64* const tb = etab + lay.tb + (size_t)dir * RV * LSX_RED_ROW;
#ifdef LSX_RED_PARK
    lds_f64* const gpk = etab + lay.gpk + (size_t)dir * (2 * NS) * LSX_WAVE;
#endif
    const int rv_raw = lane / LPV, rc = lane - rv_raw * LPV;
    const int rv = rv_raw < RV ? rv_raw : RV - 1;                                         // (RV = 6: the last two groups idle)
    const int rt = rv / NV, rq = rv - rt * NV;                                            // step in batch, value index (2 slot + entry)
    const bool r_own = rc == 0 && rv_raw < RV;

#endif
    // A sweep runs in three phases with a fixed set of memory operations each, so the compiler's wait counts are
    // exact and neither a store acknowledgement nor the half-J read-back is waited for inside a step:
    //   phase 0: this wave is the first visitor of its depths (stores its half of J)
    //   phase 1: odd Nspace only, the one depth both waves visit in the same step (exchange through LDS)
    //   phase 2: second visitor (the partner's half is requested at the top of the step, used at its end)
    auto step = [&](const int s, auto phase_c) {
        constexpr int PHX = decltype(phase_c)::value;        // 0 first visitor, 1 midpoint, 2 second visitor, 3 its last step (the end point),
        constexpr bool FIRST = PHX == 4;                     // 4 the ray's first point (a first-visitor step without a formal solution)
        constexpr int PH = FIRST ? 0 : PHX;
        constexpr bool SECOND = PH >= 2, LAST = PH == 3;
        const int k = kS + dk * s;
        const unsigned kl = (unsigned)(k * L + j) * 8u;     // byte position in the tile-major [k][j] streams
        double jd, chiTot, be_l, Ev = 0.0;
        double sv[NS], sni[NS], snj[NS], cr[NLK][3];
        const lds_f64* tk = utab + k * TR;
        double jhalf = 0.0;
        if constexpr (SECOND) {
            if (2 * s == Ns || 2 * s == Ns + 1) __syncthreads();   // the partner wave's first-half stores
        }
        if constexpr (STATIC) {
            jd = n_jd; chiTot = n_bc; be_l = n_be; Ev = n_E;
#pragma unroll
            for (int u = 0; u < NL; ++u) sv[u] = n_sv[u];                 // (inactive lanes loaded the zero pad)
#pragma unroll
            for (int u = NL; u < NPT; ++u) sv[u] = ((pact >> u) & 1u) ? tk[3 * u + 2] * Ev : 0.0;       // g_ij = (nStar_i / nStar_j) E, :453-454
            if constexpr (LK) {
#pragma unroll
                for (int u = 0; u < NL; ++u)
#pragma unroll
                    for (int q = 0; q < NCR; ++q) cr[u][q] = n_cr[u][q];
            }
            if constexpr (!LAST) stream_loads(k + dk, n_bc, n_be, n_jd, n_sv, n_E, n_cr);
            if constexpr (SECOND) jhalf = at(Jnew, kl);
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                sni[u] = tk[3 * u + 0];
                snj[u] = tk[3 * u + 1];
            }
        } else {
            stream_loads(k, chiTot, be_l, jd, sv, Ev, cr);
        }
        double hdzm, scv;
        if constexpr (STATIC) {
            hdzm = (utab + TR * dir)[k * TR + 3 * NPT] * zmu_l;   // the interval behind this ray: row k (down) / k + 1 (up)
            scv = sca_l ? at(sca, kl) : utab[k * TR + 3 * NPT + 1];
        } else {
            const double zk = z[k];
            scv = sca_l ? at(sca, kl) : LSX_CONST(double, sca)[k];
            hdzm = (0.5 * fabs(zprev - zk)) * zmu_l;
            zprev = zk;
        }
        double etaTot = be_l + scv * jd;
        STAMP(0);

        STAMP(1);

        // ---- pass 1: opacity / emissivity of the per-ray transitions (rh_method.py:601-627) ----
        //   kept for pass 2 (static path): pv = phi | Vji, chi, Uji
        double spv[NS], schi[NS], sUji[NS], seta[NS];
        auto pass1 = [&](const bool line, const SlotS& sl, double v, double ni, double nj, double alf, double& pv, double& chi,
                         double& Uji, double& eta) {
            // a tile with a single per-ray slot shares no level and no atom with anything (lsx_create): no cells
            const int fl = STATIC ? 0 : sl.flags;   // compile-time slot counts: bookkeeping in registers (pass 2)
            if (line) {
                pv = v;
                chi = (STATIC ? ni : sl.cB * (ni - sl.g * nj)) * pv;   // n_i Vij - n_j Vji, :279-280, :613 (static: ni holds the product)
                Uji = sl.Uc * pv;                        // :281
            } else {
                pv = v * alf;                            // Vji = g_ij alpha, :284-285
                chi = ni * alf - nj * pv;
                Uji = u_la * pv;                         // :286
            }
            eta = nj * Uji;                              // :614
            if (fl & SLOT_LI_CELL) cell_acc(&CCHI(sl.ci), chi, fl & SLOT_CHI_I_FIRST);
            if (fl & SLOT_LJ_CELL) {
                cell_acc(&CCHI(sl.cj), -chi, fl & SLOT_CHI_J_FIRST);
                cell_acc(&CU(sl.cj), Uji, fl & SLOT_U_J_FIRST);
            }
            if (fl & SLOT_ETA_CELL) cell_acc(&CETA(sl.ca), eta, fl & SLOT_ETA_FIRST);
            chiTot += chi;
            etaTot += eta;
        };
        if constexpr (STATIC) {
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const SlotS sl = load_slot(slots + u, Ns);
                pass1(u < NL, sl, sv[u], sni[u], snj[u], alv[u], spv[u], schi[u], sUji[u], seta[u]);
            }
        } else {
            for (int u = 0; u < nP; ++u) {
                const SlotS sl = load_slot(slots + u, Ns);
                const double ni = n_col[sl.noff_i + k];
                const double nj = n_col[sl.noff_j + k];
                const bool a = (pact >> u) & 1u;
                const int l = a ? la - sl.Nblue : 0;
                double v, alf = 0.0;
                if (sl.flags & SLOT_LINE) {
                    v = a ? phi_col[PG * (sl.base + (k * kmul + raysel) * sl.len) + cphi * sl.len + (la - sl.first)] : 0.0;
                } else {
                    v = a ? nsr_col[sl.base + k] * at(Eb, kl) : 0.0;
                    alf = a ? p.alpha[sl.wl_off + l] : 0.0;
                }
                double pv, chi, Uji, eta;
                pass1((sl.flags & SLOT_LINE) != 0, sl, v, ni, nj, alf, pv, chi, Uji, eta);
            }
        }
        STAMP(2);
        // ---- formal solution at this depth (formal_solver.py:107-139) ----
        // the two divisions of a step (by chi, :632, and by dtau, :113/:121) share ONE reciprocal, 1 / (chi dtau)
        double I, Lam, rchi, S;
        if constexpr (FIRST) {
            rchi = rcp(chiTot);
            S = etaTot * rchi;                          // :632
            I = Iu;
            Lam = 0.0;
        } else {
            const double dtau = (chi_prev + chiTot) * hdzm;
#ifdef LSX_RCP2        // diagnostic variant: one reciprocal per division
            rchi = rcp(chiTot);
            const double rdt = rcp(dtau);
#else
            const double rcd = rcp(chiTot * dtau);
            rchi = rcd * dtau;
            const double rdt = rcd * chiTot;
#endif
            S = etaTot * rchi;                          // :632
            const double dS = (S_prev - S) * rdt;
            // formal_solver.py:138-139: the end point re-uses the PREVIOUS interval's w and
            // S[kEnd - dk] with the fresh dS, dtau (reference behaviour, reproduced deliberately)
            constexpr bool last = LAST;                     // the end point lies in the second-visitor phase (Nspace >= 3): its own instance
            double w0, w1;
            w2(last ? dtau_prev : dtau, w0, w1, etab);
            const double Sx = last ? S_prev : S;
            I = Iu * (1.0 - w0) + w0 * Sx + w1 * dS;
            Lam = w0 - w1 * rdt;
            dtau_prev = dtau;
        }
        const double Psi = Lam * rchi;
        Iu = I;
        chi_prev = chiTot;
        S_prev = S;
        if constexpr (LAST) {
            if (dir == 1 && valid)                       // emergent intensity, :638
                p.Iout[((size_t)col * Nspect + la) * Nrays + mu] = I;
        }
        STAMP(3);

        // ---- angle quadrature of this wavelength: J (:640) and the two ray sums of the fast path ----
        xrow[lane] = wmuh_l * I;
        __builtin_amdgcn_wave_barrier();
        double Jsum = xrow[j];
#pragma unroll
        for (int m = 1; m < Nrays; ++m) Jsum += xrow[m * L + j];
        if constexpr (LK && STATIC) {                   // Psibar and sum_mu w Psi* phi of every line: one exchange through LDS rows
            const double wP = wq_l * Psi;
            __builtin_amdgcn_wave_barrier();
            xrow[lane] = wP;
#pragma unroll
            for (int u = 0; u < NL; ++u) xrow2[u * LSX_WAVE + lane] = wP * spv[u];
            __builtin_amdgcn_wave_barrier();
            double sPsi = xrow[j];
#pragma unroll
            for (int m = 1; m < Nrays; ++m) sPsi += xrow[m * L + j];
            at(psibar, kl) = sPsi;
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                double sPP = xrow2[u * LSX_WAVE + j];
#pragma unroll
                for (int m = 1; m < Nrays; ++m) sPP += xrow2[u * LSX_WAVE + m * L + j];
                at(ppsum, (unsigned)(u * plane) * 8u + kl) = sPP;
            }
        } else if (nF > 0) {                            // Psibar of this direction, for k_fast_gamma
            __builtin_amdgcn_wave_barrier();
            xrow[lane] = wq_l * Psi;
            __builtin_amdgcn_wave_barrier();
            double sPsi = xrow[j];
#pragma unroll
            for (int m = 1; m < Nrays; ++m) sPsi += xrow[m * L + j];
            at(psibar, kl) = sPsi;                          // every lane of a wavelength holds the same sum: no branch
        }
        __builtin_amdgcn_wave_barrier();
        STAMP(4);

        // ---- pass 2: Gamma integrands of the per-ray transitions (rh_method.py:643-681) ----
        auto pass2x = [&](bool line, double Vc, double pv, double Uji, double Vij, double wt, double etaA,
                          double chi_i, double chi_j, double U_j, double U_i, double& wg1, double& wg2) {
            const double Vji = line ? Vc * pv : pv;
            const double Ieff = I - Psi * etaA;                            // :652
            const double g1 = (Uji + Vji * Ieff) - (chi_i * Psi) * U_j;    // :677
            const double g2 = (Vij * Ieff) - (chi_j * Psi) * U_i;          // :680
            wg1 = wt * g1;                                                 // wt: :451/:455, :665
            wg2 = wt * g2;
        };
        auto pass2 = [&](const bool line, const SlotS& sl, double pv, double chi, double Uji, double Vij, double eta, double wt,
                         double cEC, double cXi, double cXj, double& wg1, double& wg2) {
            const int fl = NPT == 1 ? 0 : sl.flags;
            const double Vji = line ? sl.Vc * pv : pv;
            const double etaA = ((fl & SLOT_ETA_CELL) ? CETA(sl.ca) : eta) + cEC;      // + the linked continua's share
            const double chi_i = ((fl & SLOT_LI_CELL) ? CCHI(sl.ci) : chi) + cXi;
            const double chi_j = ((fl & SLOT_LJ_CELL) ? CCHI(sl.cj) : -chi) + cXj;
            const double U_j = (fl & SLOT_LJ_CELL) ? CU(sl.cj) : Uji;
            const double U_i = (fl & SLOT_UI_READ) ? CU(sl.ci) : 0.0;
            const double Ieff = I - Psi * etaA;                            // :652
            const double g1 = (Uji + Vji * Ieff) - (chi_i * Psi) * U_j;    // :677
            const double g2 = (Vij * Ieff) - (chi_j * Psi) * U_i;          // :680
            wg1 = wt * g1;
            wg2 = wt * g2;
        };
        // the reductions of slot u go to Gpart[slot][e][dir][k], e = 0: Gamma[i][j], e = 1: Gamma[j][i]
        auto gslot = [&](int u, int e) -> double* { return gpart + ((u * 2 + e) * 2 + dir) * Ns + k; };
        if constexpr (STATIC) {
            double w1[NS], w2v[NS];
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const SlotS sl = load_slot(slots + u, Ns);
                const bool line = u < NL;
                const double Vij = line ? sl.cB * spv[u] : alv[u];
#ifndef LSX_RED_LDS
                const double wt = line ? wlv[u] * tk[3 * u + 2] : wlv[u];         // :451 (lines: x wphi), :455, :665
#else
                const double wt = wlv[u];         // :455, :665; the lines' wphi (:451) is wave-uniform: applied to the lane sum
#endif
                // linked continua add their ray-independent share to the line's atom.eta, atom.chi[i], atom.chi[j]
                // (x + 0.0 is not x to the compiler: the corrections are added only where the instance has them)
                const bool lkl = LK && line;
                auto plus = [](double x, double c, bool on) { return on ? x + c : x; };
                const double cEC = (LK && line) ? cr[u < NLK ? u : 0][0] : 0.0, cXi = (LK && line) ? cr[u < NLK ? u : 0][1] : 0.0,
                             cXj = (LK && line && NCR > 2) ? cr[u < NLK ? u : 0][2] : 0.0;
                if constexpr (NPT >= 3) {
                    // atom.chi / atom.U / atom.eta of this slot's levels, accumulated in transition order; another slot
                    // enters through five factors in {-1, 0, 1} (one fma each), and only if any of them is non-zero
                    // (the slot's own terms first, then the related slots in transition order: for three and more terms the
                    // sums are associated differently from the reference's running sums -- last-bit differences)
                    const unsigned mask = slots[u].relmask;
                    double etaA = seta[u], chi_i = schi[u], chi_j = -schi[u], U_j = sUji[u], U_i = 0.0;
#pragma unroll
                    for (int v = 0; v < NPT; ++v) {
                        if (v != u) {
                            const int o = v < u ? v : v - 1;
                            if ((mask >> o) & 1u) {
                                const lds_f64* r = ctab + (u * (NPT - 1) + o) * 5;
                                chi_i = fma(r[REL_CI], schi[v], chi_i);
                                chi_j = fma(r[REL_CJ], schi[v], chi_j);
                                U_j = fma(r[REL_UJ], sUji[v], U_j);
                                U_i = fma(r[REL_UI], sUji[v], U_i);
                                etaA = fma(r[REL_EA], seta[v], etaA);
                            }
                        }
                    }
                    pass2x(line, sl.Vc, spv[u], sUji[u], Vij, wt, plus(etaA, cEC, lkl), plus(chi_i, cXi, lkl), plus(chi_j, cXj, lkl), U_j, U_i, w1[u], w2v[u]);
                } else if constexpr (NPT == 2 && TOPO != 0) {
                    // rh_method.py:652, 677-681 with atom.U[i] = 0, atom.U[j] = Uji, atom.chi[j] = -chi (TOPO 1 and 2) and
                    // atom.chi[i] = chi + chi_other, atom.eta = eta + eta_other (TOPO 1: common lower level, same atom)
                    const int v = 1 - u;
                    const double etaA = plus(TOPO == 1 ? seta[u] + seta[v] : seta[u], cEC, lkl);
                    const double chi_i = plus(TOPO == 1 ? schi[u] + schi[v] : schi[u], cXi, lkl);
                    const double Vji = line ? sl.Vc * spv[u] : spv[u];
                    const double Ieff = I - Psi * etaA;
                    w1[u] = wt * ((sUji[u] + Vji * Ieff) - (chi_i * Psi) * sUji[u]);
                    w2v[u] = wt * (Vij * Ieff);
                } else if constexpr (NPT == 2) {
                    // atom.chi / atom.U / atom.eta of this slot's levels from the two slots' values, in transition
                    // order (a factor 0 drops the other slot, +-1 adds it with one rounding)
                    const int v = 1 - u;
                    const auto* rel = slots[u].rel[0];
                    const double etaA = fma(rel[REL_EA], seta[v], seta[u]);
                    const double chi_i = fma(rel[REL_CI], schi[v], schi[u]);
                    const double chi_j = fma(rel[REL_CJ], schi[v], -schi[u]);
                    const double U_j = fma(rel[REL_UJ], sUji[v], sUji[u]);
                    const double U_i = rel[REL_UI] * sUji[v];
                    pass2x(line, sl.Vc, spv[u], sUji[u], Vij, wt, plus(etaA, cEC, lkl), plus(chi_i, cXi, lkl), plus(chi_j, cXj, lkl), U_j, U_i, w1[u], w2v[u]);
                } else {
                    // a single per-ray slot: atom.U[i] = 0, atom.U[j] = Uji, atom.chi[j] = -chi (its product with U[i] vanishes)
                    const double Vji = line ? sl.Vc * spv[u] : spv[u];
                    const double Ieff = I - Psi * plus(seta[u], cEC, lkl);                              // :652
                    w1[u] = wt * ((sUji[u] + Vji * Ieff) - (plus(schi[u], cXi, lkl) * Psi) * sUji[u]);   // :677
                    w2v[u] = wt * (Vij * Ieff);                                                          // :680
                }
            }
#ifndef LSX_RED_LDS
            // the totals of step s are parked in entry (s mod 64) of per-(slot, entry) LDS rows and leave as one
            // 64-wide store every 64 steps: no store (and no store acknowledgement to wait for) inside a step
            const int sl64 = s & 63;
            if constexpr (NPT == 1) {
                const double t = reduce_pair(w1[0], w2v[0]);
                if (own_pair) gpk[row_pair + sl64] = t;
            } else if constexpr (NPT >= 2) {
                const double t = reduce_quad(w1[0], w2v[0], w1[1], w2v[1]);
                if (own_quad) gpk[row_quad + sl64] = t;
                if constexpr (NPT == 3) {
                    const double t2 = reduce_pair(w1[2], w2v[2]);
                    if (own_pair) gpk[4 * LSX_WAVE + row_pair + sl64] = t2;
                }
                if constexpr (NPT == 4) {
                    const double t2 = reduce_quad(w1[2], w2v[2], w1[3], w2v[3]);
                    if (own_quad) gpk[4 * LSX_WAVE + row_quad + sl64] = t2;
                }
            }
            if (sl64 == 63 || s == Ns - 1) {
                const int ks = kS + dk * (s - sl64 + lane);         // the depth parked in entry `lane` of every row
                if (lane <= sl64) {
#pragma unroll
                    for (int q = 0; q < 2 * NPT; ++q) gpart[(q * 2 + dir) * Ns + ks] = gpk[q * LSX_WAVE + lane];
                }
            }
#else
            if constexpr (NPT >= 1) {
                const int pos = s & (RT - 1);
                lds_f64* const tbw = tb + pos * NV * LSX_RED_ROW + lane;
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    tbw[(2 * u) * LSX_RED_ROW] = w1[u];
                    tbw[(2 * u + 1) * LSX_RED_ROW] = w2v[u];
                }
                if (pos == RT - 1 || s == Ns - 1) {
                    __builtin_amdgcn_wave_barrier();
                    // 16-byte reads (ds_read_b128: 64 banks, conflict free on rows of LSX_RED_ROW doubles; the 8-byte forms
                    // would put a 16-lane group on two banks)
                    typedef double lds_pair __attribute__((ext_vector_type(2)));
                    const auto* src = (const __attribute__((address_space(3))) lds_pair*)(tb + rv * LSX_RED_ROW + rc * RVP);   // 16-byte aligned (lsx_sweep_lds)
                    lds_pair v2 = src[0];
                    double acc = v2.x + v2.y;
#pragma unroll
                    for (int e = 1; e < RVP / 2; ++e) { v2 = src[e]; acc += v2.x + v2.y; }
                    acc += dpp_f64<0xB1, 0xf>(acc);                          // quad_perm [1,0,3,2]
                    acc += dpp_f64<0x4E, 0xf>(acc);                          // quad_perm [2,3,0,1]
                    acc += dpp_f64<0x141, 0xf>(acc);                         // row_half_mirror: 8-lane sums
                    if constexpr (LPV == 16) acc += dpp_f64<0x140, 0xf>(acc); // row_mirror: 16-lane sums
                    const int kt = kS + dk * (s - pos + rt);                 // the depth this lane's row belongs to
                    if (r_own && rt <= pos) {
                        // lines: x wphi of that depth (rh_method.py:451); continua (:455) carry their whole weight in wlv
                        const double wn = (rq >> 1) < NL ? utab[kt * TR + 3 * (rq >> 1) + 2] : 1.0;
#ifdef LSX_RED_PARK
                        gpk[rq * LSX_WAVE + ((s - pos + rt) & 63)] = acc * wn;
#else
                        gpart[(rq * 2 + dir) * Ns + kt] = acc * wn;
#endif
                    }
                    __builtin_amdgcn_wave_barrier();
#ifdef LSX_RED_PARK
                    const int sl64 = s & 63;
                    if (sl64 == 63 || s == Ns - 1) {
                        const int ks = kS + dk * (s - sl64 + lane);         // the depth parked in entry `lane` of every row
                        if (lane <= sl64) {
#pragma unroll
                            for (int q = 0; q < 2 * NPT; ++q) gpart[(q * 2 + dir) * Ns + ks] = gpk[q * LSX_WAVE + lane];
                        }
                    }
#endif
                }
            }
#endif
        } else {
            for (int u = 0; u < nP; ++u) {
                const SlotS sl = load_slot(slots + u, Ns);
                const double ni = n_col[sl.noff_i + k];
                const double nj = n_col[sl.noff_j + k];
                const bool a = (pact >> u) & 1u;
                const int l = a ? la - sl.Nblue : 0;
                double wla = a ? p.wl[sl.wl_off + l] : 0.0;
                double pv, Vij, Uji, chi;
                if (sl.flags & SLOT_LINE) {
                    pv = a ? phi_col[PG * (sl.base + (k * kmul + raysel) * sl.len) + cphi * sl.len + (la - sl.first)] : 0.0; // L1/L2 hit
                    Vij = sl.cB * pv;
                    Uji = sl.Uc * pv;
                    chi = (sl.cB * (ni - sl.g * nj)) * pv;
                    wla *= wphi_col[sl.wphi_off + k];
                } else {
                    Vij = a ? p.alpha[sl.wl_off + l] : 0.0;
                    pv = (a ? nsr_col[sl.base + k] * at(Eb, kl) : 0.0) * Vij;                          // Vji
                    Uji = u_la * pv;
                    chi = ni * Vij - nj * pv;
                }
                double wg1, wg2, cEC = 0.0, cXi = 0.0, cXj = 0.0;
                if constexpr (LK) {
                    if (u < nLt) {          // line slot of a tile with linked continua: corrections in, sum_mu w Psi* phi out
                        cEC = at(corr, (unsigned)((3 * u + 0) * plane) * 8u + kl);
                        cXi = at(corr, (unsigned)((3 * u + 1) * plane) * 8u + kl);
                        cXj = at(corr, (unsigned)((3 * u + 2) * plane) * 8u + kl);
                        __builtin_amdgcn_wave_barrier();
                        xrow[lane] = (wq_l * Psi) * pv;
                        __builtin_amdgcn_wave_barrier();
                        double sPP = xrow[j];
                        for (int m = 1; m < Nrays; ++m) sPP += xrow[m * L + j];
                        at(ppsum, (unsigned)(u * plane) * 8u + kl) = sPP;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                pass2((sl.flags & SLOT_LINE) != 0, sl, pv, chi, Uji, Vij, nj * Uji, (a && valid) ? wq_l * wla : 0.0, cEC, cXi, cXj, wg1, wg2);
                const double t = reduce_pair(wg1, wg2);
                if (lane == 31) *gslot(u, 0) = t;
                if (lane == 63) *gslot(u, 1) = t;
            }
        }
        STAMP(5);

        // ---- J: the two directions meet at depth k at different steps ----
        // (the Nrays lanes of a wavelength hold the same Jsum and write the same word: stores need no branch)
        if constexpr (PH == 0) {
            at(Jnew, kl) = Jsum;                              // first visitor stores its half
        } else if constexpr (PH == 1) {                       // odd Nspace: both waves are at the same k
            if (lead) xwg[dir * LSX_WAVE + j] = Jsum;
            __syncthreads();
            if (lead && dir == 0) {
                const double Jv = Jsum + xwg[LSX_WAVE + j];
                at(Jnew, kl) = Jv;
                dJ = nanmax(dJ, fabs(1.0 - jd * rcp(Jv)));         // :705
            }
        } else {
            if constexpr (!STATIC) jhalf = at(Jnew, kl);
            const double Jv = jhalf + Jsum;
            at(Jnew, kl) = Jv;
            dJ = nanmax(dJ, fabs(1.0 - jd * rcp(Jv)));         // :705
        }
        STAMP(6);
    };
    {
        const int nA = Ns / 2;                                // depths this wave reaches first: 2 s < Nspace - 1
        step(0, std::integral_constant<int, 4>{});              // the ray's first point (Nspace >= 3: nA >= 1)
        for (int s = 1; s < nA; ++s) step(s, std::integral_constant<int, 0>{});
        if (Ns & 1) step(nA, std::integral_constant<int, 1>{});
        for (int s = nA + (Ns & 1); s < Ns - 1; ++s) step(s, std::integral_constant<int, 2>{});
        step(Ns - 1, std::integral_constant<int, 3>{});       // the end point (Nspace >= 3: always a second-visitor step)
    }

#if defined(LSX_STAMPS) || defined(LSX_CLOCK)
    STAMP(7);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk1), "=s"(tr1)::"memory");
    T[7] = tk1 - tk0;                  // the loop in shader-clock ticks ...
#ifdef LSX_CLOCK
    T[0] = tk0 - tk_entry;             // ... and the wave's prologue (operand table, boundary condition, first loads)
    T[1] = tr0;                        // when the loop started (100 MHz ticks, absolute): the order the grid's waves ran in
#endif
    if (lane == 0 && dir == 0 && p.debug && col % 100 == 5) {    // every tile of columns 5, 105, ... (1024 records), down-going wave
        unsigned long long* D = (unsigned long long*)p.debug + (size_t)(((col / 100) * ntile + tile_id) & 1023) * 16;
        for (int i = 0; i < 8; ++i) D[i] = T[i];
        D[8] = tile_id; D[9] = nP; D[10] = nF; D[11] = col; D[12] = STATIC ? NL : nLt; D[13] = LK; D[14] = TOPO;
        D[15] = tr1 - tr0;             // ... and in 100 MHz ticks: clock = D[7] / D[15] x 100 MHz (MI355X_MICROARCH.md, DVFS give-back 6)
    }
#endif
#undef STAMP
    const double dJw = wave_max_nan(lead ? dJ : 0.0);
    if (lane == 0) p.dJpart[((size_t)col * ntile + tile_id) * 2 + dir] = dJw;
#undef CCHI
#undef CU
#undef CETA
}


// ---- N4: the same tile with the monotonic piecewise-parabolic rule (include/lsx.h) -----------------------------------
// The parabolic rule at depth k needs chi and S of the DOWNWIND neighbour, so the formal solution (and everything that
// follows it: angle sums, Gamma integrands, J) of depth m runs one step behind the opacities: step t evaluates chi, S at
// depth t, then finishes depth t - 1.  One generic instance for every tile (runtime slot loops, lane-private LDS cells for
// the level bookkeeping, linked continua at run time); the cells of depth m are refilled before its Gamma pass.  Not tuned:
// this is the "what comes next" row, the linear rule is the product path.
template <int NR, bool SCAL>
__device__ __forceinline__ void sweep_tile_parabolic(const SweepParams& p, const int vb, const int tile_id)
{
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    lds_f64* const etab = (lds_f64*)lds_raw;
    lds_f64* const lds = etab + LSX_EXP_TAB;
    const int lane = threadIdx.x & (LSX_WAVE - 1);
    const int dir = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntile = p.ntile_total;
    const int col = vb / p.n_class_tiles;
    const auto* tilep = LSX_CONST(DevTile, p.tiles) + tile_id;
    const int la0 = tilep->la0, nla = tilep->nla, slot0 = tilep->slot0;
    const int nP = tilep->nP, nF = tilep->nF;
    const bool LK = tilep->nK > 0;
    const int nLt = tilep->nL;
    const auto* slots = LSX_CONST(DevSlot, p.slots) + slot0;
    const int Ns = p.Nspace, Nspect = p.Nspect;
    const int Nrays = NR > 0 ? NR : p.Nrays;
    constexpr int NRD = NR > 0 ? NR : 1;
    const int L = NR > 0 ? LSX_WAVE / NRD : p.L;
    const int mu_raw = lane / L;
    const int j_raw = lane - mu_raw * L;
    const bool valid = mu_raw < Nrays && j_raw < nla;
    const int mu = mu_raw < Nrays ? mu_raw : Nrays - 1;
    const int j = j_raw < nla ? j_raw : nla - 1;
    const int la = la0 + j;
    const bool lead = valid && mu_raw == 0;
    const int rows = 2 * p.ncell_lev + p.ncell_atom + 1;
    lds_f64* const wrow = lds + (size_t)dir * rows * LSX_WAVE + lane;
#define CCHI(c) wrow[(2 * (c)) * LSX_WAVE]
#define CU(c) wrow[(2 * (c) + 1) * LSX_WAVE]
#define CETA(a) wrow[(2 * p.ncell_lev + (a)) * LSX_WAVE]
    lds_f64* const xrow = lds + (size_t)dir * rows * LSX_WAVE + (size_t)(rows - 1) * LSX_WAVE;
    lds_f64* const xwg = lds + (size_t)2 * rows * LSX_WAVE;
    const auto* n_col = LSX_CONST(double, p.n + (size_t)col * p.NLtot * Ns);
    const auto* wphi_col = LSX_CONST(double, p.wphi + (size_t)col * p.Nlines * Ns);
    const auto* z = LSX_CONST(double, p.height + (size_t)col * Ns);
    const auto* tcol = LSX_CONST(double, p.temperature + (size_t)col * Ns);
    const size_t tbase = ((size_t)col * ntile + tile_id) * Ns * L;
    const double* __restrict__ bgchi = (nF > 0 ? p.bgxchi_T : p.bgchi_T) + tbase;
    const double* __restrict__ bgeta = (nF > 0 ? p.bgxeta_T : p.bgeta_T) + tbase;
    double* __restrict__ psibar = p.Psi2_T + ((size_t)dir * p.ncol * ntile) * Ns * L + tbase;
    const double* __restrict__ Jdag = p.Jdag_T + tbase;
    double* __restrict__ Jnew = p.Jnew_T + tbase;
    const bool sca_l = SCAL && p.sca_per_lambda;
    const double* __restrict__ sca = sca_l ? p.sca + tbase : p.sca + (size_t)col * Ns;
    const int PG = p.phi_G, cphi = col % PG;                       // the grouped line-profile store (lsx_dev.h, phi_elem)
    const double* __restrict__ phi_col = p.phi_T + (size_t)(col - cphi) * p.phi_col_stride;      // the group's first element
    const double* __restrict__ Eb = p.E_T + tbase;
    const auto* nsr_col = LSX_CONST(double, p.nsr + (size_t)col * p.Ncont * Ns);
    const size_t plane = (size_t)Ns * L;
    const double* __restrict__ corr = LK ? p.corr_T + (size_t)col * p.corr_col_stride + tilep->corr_off : nullptr;
    double* __restrict__ ppsum = LK ? p.Psi3_T + ((size_t)dir * p.ncol + col) * p.pp_col_stride + tilep->pp_off : nullptr;
    double* __restrict__ gpart = p.Gpart + ((size_t)col * p.nslot_total + slot0) * 4 * Ns;
    etab[threadIdx.x] = p.exp2_tab[threadIdx.x];
    __syncthreads();

    const double wav = p.wavelength[la];
    const double u_la = p.u_la[la];
    const double zmu_l = p.zmu[mu];
    const double wmuh_l = valid ? p.wmuh[mu] : 0.0;
    const double wq_l = wmuh_l * (4.0 * kPi);
    const bool compact = p.phi_compact != 0;
    const int kS = dir ? Ns - 1 : 0;
    const int dk = dir ? -1 : 1;
    const int raysel = compact ? 0 : dir * Ns * Nrays + mu;
    const int kmul = compact ? 1 : Nrays;
    unsigned pact = 0;
    for (int u = 0; u < nP; ++u) {
        const int l = la - slots[u].Nblue;
        if (l >= 0 && l < slots[u].Nlam && p.active[slots[u].trans * Nspect + la] != 0) pact |= 1u << u;
    }
    // one per-ray slot at depth k: rh_method.py:279-286, 453-455
    struct SlotV { double pv, chi, Uji, Vij, eta, wla; };
    auto slot_at = [&](int u, const SlotS& sl, int k) {
        SlotV v;
        const double ni = n_col[sl.noff_i + k], nj = n_col[sl.noff_j + k];
        const bool a = (pact >> u) & 1u;
        const int l = a ? la - sl.Nblue : 0;
        v.wla = a ? p.wl[sl.wl_off + l] : 0.0;
        if (sl.flags & SLOT_LINE) {
            v.pv = a ? phi_col[PG * (sl.base + (k * kmul + raysel) * sl.len) + cphi * sl.len + (la - sl.first)] : 0.0;
            v.Vij = sl.cB * v.pv;
            v.Uji = sl.Uc * v.pv;
            v.chi = (sl.cB * (ni - sl.g * nj)) * v.pv;
            v.wla *= wphi_col[sl.wphi_off + k];
        } else {
            v.Vij = a ? p.alpha[sl.wl_off + l] : 0.0;
            v.pv = (a ? nsr_col[sl.base + k] * Eb[k * L + j] : 0.0) * v.Vij;      // Vji
            v.Uji = u_la * v.pv;
            v.chi = ni * v.Vij - nj * v.pv;
        }
        v.eta = nj * v.Uji;
        return v;
    };
    // opacity and source function at depth k (rh_method.py:601-632); cells: also fill the level / atom cells of that depth
    auto opac = [&](int k, bool cells, double& chiTot, double& S) {
        const unsigned kl = (unsigned)(k * L + j) * 8u;
        chiTot = at(bgchi, kl);
        const double scv = sca_l ? at(sca, kl) : LSX_CONST(double, sca)[k];
        double etaTot = at(bgeta, kl) + scv * at(Jdag, kl);
        for (int u = 0; u < nP; ++u) {
            const SlotS sl = load_slot(slots + u, Ns);
            const SlotV v = slot_at(u, sl, k);
            if (cells) {
                const int fl = sl.flags;
                if (fl & SLOT_LI_CELL) cell_acc(&CCHI(sl.ci), v.chi, fl & SLOT_CHI_I_FIRST);
                if (fl & SLOT_LJ_CELL) {
                    cell_acc(&CCHI(sl.cj), -v.chi, fl & SLOT_CHI_J_FIRST);
                    cell_acc(&CU(sl.cj), v.Uji, fl & SLOT_U_J_FIRST);
                }
                if (fl & SLOT_ETA_CELL) cell_acc(&CETA(sl.ca), v.eta, fl & SLOT_ETA_FIRST);
            }
            chiTot += v.chi;
            etaTot += v.eta;
        }
        S = etaTot / chiTot;
    };
    // does any slot of this tile keep level / atom sums in cells (overlapping transitions)?  Only then does the Gamma pass of a
    // depth need the cells of that depth refilled
    bool any_cells = false;
    for (int u = 0; u < nP; ++u) any_cells = any_cells || (slots[u].flags & (SLOT_LI_CELL | SLOT_LJ_CELL | SLOT_ETA_CELL | SLOT_UI_READ)) != 0;
    // ---- boundary conditions: formal_solver.py:203-209 (unchanged) ----
    double Iu = 0.0;
    double c_u = 1.0, S_u = 0.0, c_k = 1.0, S_k = 0.0, c_d = 1.0, S_d = 0.0;     // upwind / local / downwind of the depth being finished
    opac(kS, false, c_d, S_d);
    if (dir) {
        double c1, S1;
        opac(kS + dk, false, c1, S1);
        const double dtau_uw = zmu_l * (c_d + c1) * 0.5 * fabs(z[kS] - z[kS + dk]);
        const double B0 = planck(tcol[Ns - 2], wav), B1 = planck(tcol[Ns - 1], wav);
        Iu = B1 - (B0 - B1) / dtau_uw;
    }
    double dJ = 0.0;
    auto gslot = [&](int k, int u, int e) -> double* { return gpart + ((u * 2 + e) * 2 + dir) * Ns + k; };

    // finish depth m (visited by this wave at its step m): formal solution, angle sums, Gamma integrands, J
    auto finish = [&](const int m, auto phase_c) {
        constexpr int PH = decltype(phase_c)::value;
        const int k = kS + dk * m;
        const unsigned kl = (unsigned)(k * L + j) * 8u;
        if constexpr (PH == 2) {
            if (2 * m == Ns || 2 * m == Ns + 1) __syncthreads();
        }
        double I, Lam;
        if (m == 0) {
            I = Iu;
            Lam = 0.0;
        } else {
            const bool has_d = m < Ns - 1;
            const double dtau_u = (c_u + c_k) * (0.5 * fabs(z[k - dk] - z[k])) * zmu_l;
            const double dtau_d = has_d ? (c_k + c_d) * (0.5 * fabs(z[k] - z[k + dk])) * zmu_l : 1.0;
            const Para r = parabolic_point(Iu, S_u, S_k, has_d ? S_d : 0.0, dtau_u, dtau_d, has_d, etab);
            I = r.I;
            Lam = r.Lam;
        }
        const double Psi = Lam / c_k;
        Iu = I;
        if (m == Ns - 1 && dir == 1 && valid) p.Iout[((size_t)col * Nspect + la) * Nrays + mu] = I;
        // angle quadrature: J (:640) and Psibar for the fast continua
        xrow[lane] = wmuh_l * I;
        __builtin_amdgcn_wave_barrier();
        double Jsum = xrow[j];
        for (int q = 1; q < Nrays; ++q) Jsum += xrow[q * L + j];
        if (nF > 0) {
            __builtin_amdgcn_wave_barrier();
            xrow[lane] = wq_l * Psi;
            __builtin_amdgcn_wave_barrier();
            double sPsi = xrow[j];
            for (int q = 1; q < Nrays; ++q) sPsi += xrow[q * L + j];
            at(psibar, kl) = sPsi;
        }
        __builtin_amdgcn_wave_barrier();
        // Gamma integrands of the per-ray transitions at depth k (rh_method.py:643-681): cells of THIS depth first
        if (any_cells) {
            double c0, s0;
            opac(k, true, c0, s0);
            __builtin_amdgcn_wave_barrier();
        }
        for (int u = 0; u < nP; ++u) {
            const SlotS sl = load_slot(slots + u, Ns);
            const SlotV v = slot_at(u, sl, k);
            const bool a = (pact >> u) & 1u;
            const bool line = (sl.flags & SLOT_LINE) != 0;
            double cEC = 0.0, cXi = 0.0, cXj = 0.0;
            if (LK && u < nLt) {          // line slot of a tile with linked continua: corrections in, sum_mu w Psi* phi out
                cEC = at(corr, (unsigned)((3 * u + 0) * plane) * 8u + kl);
                cXi = at(corr, (unsigned)((3 * u + 1) * plane) * 8u + kl);
                cXj = at(corr, (unsigned)((3 * u + 2) * plane) * 8u + kl);
                __builtin_amdgcn_wave_barrier();
                xrow[lane] = (wq_l * Psi) * v.pv;
                __builtin_amdgcn_wave_barrier();
                double sPP = xrow[j];
                for (int q = 1; q < Nrays; ++q) sPP += xrow[q * L + j];
                at(ppsum, (unsigned)(u * plane) * 8u + kl) = sPP;
                __builtin_amdgcn_wave_barrier();
            }
            const int fl = sl.flags;
            const double Vji = line ? sl.Vc * v.pv : v.pv;
            const double etaA = ((fl & SLOT_ETA_CELL) ? CETA(sl.ca) : v.eta) + cEC;
            const double chi_i = ((fl & SLOT_LI_CELL) ? CCHI(sl.ci) : v.chi) + cXi;
            const double chi_j = ((fl & SLOT_LJ_CELL) ? CCHI(sl.cj) : -v.chi) + cXj;
            const double U_j = (fl & SLOT_LJ_CELL) ? CU(sl.cj) : v.Uji;
            const double U_i = (fl & SLOT_UI_READ) ? CU(sl.ci) : 0.0;
            const double Ieff = I - Psi * etaA;                               // :652
            const double g1 = (v.Uji + Vji * Ieff) - (chi_i * Psi) * U_j;     // :677
            const double g2 = (v.Vij * Ieff) - (chi_j * Psi) * U_i;           // :680
            const double wt = (a && valid) ? wq_l * v.wla : 0.0;
            const double t = reduce_pair(wt * g1, wt * g2);
            if (lane == 31) *gslot(k, u, 0) = t;
            if (lane == 63) *gslot(k, u, 1) = t;
        }
        // J: the two directions meet at depth k at different steps
        const double jd = at(Jdag, kl);
        if constexpr (PH == 0) {
            at(Jnew, kl) = Jsum;
        } else if constexpr (PH == 1) {
            if (lead) xwg[dir * LSX_WAVE + j] = Jsum;
            __syncthreads();
            if (lead && dir == 0) {
                const double Jv = Jsum + xwg[LSX_WAVE + j];
                at(Jnew, kl) = Jv;
                dJ = nanmax(dJ, fabs(1.0 - jd / Jv));
            }
        } else {
            const double Jv = at(Jnew, kl) + Jsum;
            at(Jnew, kl) = Jv;
            dJ = nanmax(dJ, fabs(1.0 - jd / Jv));
        }
    };
    // step t: opacities of depth t + 1 (the downwind neighbour), then depth t is finished
    auto advance = [&](int m) {
        c_u = c_k; S_u = S_k;
        c_k = c_d; S_k = S_d;
        if (m + 1 < Ns) opac(kS + dk * (m + 1), false, c_d, S_d);
    };
    {
        const int nA = Ns / 2;
        for (int m = 0; m < nA; ++m) { advance(m); finish(m, std::integral_constant<int, 0>{}); }
        if (Ns & 1) { advance(nA); finish(nA, std::integral_constant<int, 1>{}); }
        for (int m = nA + (Ns & 1); m < Ns; ++m) { advance(m); finish(m, std::integral_constant<int, 2>{}); }
    }
    const double dJw = wave_max_nan(lead ? dJ : 0.0);
    if (lane == 0) p.dJpart[((size_t)col * ntile + tile_id) * 2 + dir] = dJw;
#undef CCHI
#undef CU
#undef CETA
}

// ---- N4 with compile-time tile classes: the monotonic piecewise-parabolic rule (include/lsx.h) for tiles with at most two
// per-ray slots, on the one-ray-per-lane mapping of sweep_tile's static path (slot state in registers, per-depth operands in
// the LDS table, streams requested one depth ahead, DPP / permlane lane sums).  The rule at depth m needs the source function
// and the opacity of the DOWNWIND neighbour, so step s evaluates opacity and source function at depth s and then FINISHES depth
// s - 1 (formal solution, angle sums, Gamma integrands, J) from the values step s - 1 left behind; the last depth is finished
// after the loop with the linear end rule.  Same terms as sweep_tile_parabolic (the generic instance, which stays for the
// other tile shapes and small batches); the reciprocals are v_rcp_f64 + one Newton step there IEEE divisions.
template <int NPT, int NL, int NR, bool LK, int TOPO>
__device__ __forceinline__ void sweep_tile_par(const SweepParams& p, const int vb, const int tile_id)
{
    static_assert(NPT >= 0 && NPT <= 2 && NR > 0, "compile-time classes with at most two per-ray slots");
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    lds_f64* const etab = (lds_f64*)lds_raw;
    lds_f64* const lds = etab + LSX_EXP_TAB;
    constexpr int NS = NPT > 0 ? NPT : 1;
    constexpr int TR = 3 * NPT + 2;
    constexpr bool HASC = NPT > NL;
    constexpr int NLK = (LK && NL > 0) ? NL : 1;
    constexpr int NCR = NPT == 1 ? 2 : 3;
    constexpr int L = LSX_WAVE / NR;
    const int lane = threadIdx.x & (LSX_WAVE - 1);
    const int dir = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntile = p.ntile_total;
    const int col = vb / p.n_class_tiles;
    const auto* tilep = LSX_CONST(DevTile, p.tiles) + tile_id;
    const int la0 = tilep->la0, nla = tilep->nla, slot0 = tilep->slot0, nF = tilep->nF;
    const auto* slots = LSX_CONST(DevSlot, p.slots) + slot0;
    const int Ns = p.Nspace, Nspect = p.Nspect;
    const int mu_raw = lane / L, j_raw = lane - mu_raw * L;
    const bool valid = mu_raw < NR && j_raw < nla;
    const int mu = mu_raw < NR ? mu_raw : NR - 1;
    const int j = j_raw < nla ? j_raw : nla - 1;
    const int la = la0 + j;
    const bool lead = valid && mu_raw == 0;
    const SweepLds lay = lsx_sweep_lds(NPT, LK, p.Nspace, p.ncell_lev, p.ncell_atom);
    const int rows = lay.rows;
    lds_f64* const xrow = lds + (size_t)dir * rows * LSX_WAVE + (size_t)(rows - 1) * LSX_WAVE;
    lds_f64* const xwg = etab + lay.xwg;
    lds_f64* const utab = etab + lay.utab;
    lds_f64* const xrow2 = etab + lay.xrow2 + (size_t)dir * NS * LSX_WAVE;
    const auto* z = LSX_CONST(double, p.height + (size_t)col * Ns);
    const auto* tcol = LSX_CONST(double, p.temperature + (size_t)col * Ns);
    const size_t tbase = ((size_t)col * ntile + tile_id) * Ns * L;
    const double* __restrict__ bgchi = (nF > 0 ? p.bgxchi_T : p.bgchi_T) + tbase;
    const double* __restrict__ bgeta = (nF > 0 ? p.bgxeta_T : p.bgeta_T) + tbase;
    double* __restrict__ psibar = p.Psi2_T + ((size_t)dir * p.ncol * ntile) * Ns * L + tbase;
    const double* __restrict__ Jdag = p.Jdag_T + tbase;
    double* __restrict__ Jnew = p.Jnew_T + tbase;
    const int PG = p.phi_G, cphi = col % PG;                       // the grouped line-profile store (lsx_dev.h, phi_elem)
    const double* __restrict__ phi_col = p.phi_T + (size_t)(col - cphi) * p.phi_col_stride;      // the group's first element
    const double* __restrict__ Eb = p.E_T + tbase;
    const size_t plane = (size_t)Ns * L;
    const double* __restrict__ corr = LK ? p.corr_T + (size_t)col * p.corr_col_stride + tilep->corr_off : nullptr;
    double* __restrict__ ppsum = LK ? p.Psi3_T + ((size_t)dir * p.ncol + col) * p.pp_col_stride + tilep->pp_off : nullptr;
    double* __restrict__ gpart = p.Gpart + ((size_t)col * p.nslot_total + slot0) * 4 * Ns;
    // the per-depth operand table (as sweep_tile): per slot lines (cB (n_i - g n_j), n_j, wphi) / continua (n_i, n_j, nStar ratio),
    // then the half length of the interval above depth k and the scattering coefficient
    {
        const double* ncolp = p.n + (size_t)col * p.NLtot * Ns;
        const double* zc = p.height + (size_t)col * Ns;
        if (threadIdx.x < TR) utab[Ns * TR + threadIdx.x] = 0.0;
        for (int e = threadIdx.x; e < Ns; e += 2 * LSX_WAVE) {
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const double ni = ncolp[(size_t)slots[u].li * Ns + e], nj = ncolp[(size_t)slots[u].lj * Ns + e];
                const bool line = u < NL;
                utab[e * TR + 3 * u + 0] = line ? slots[u].cB * (ni - slots[u].g * nj) : ni;
                utab[e * TR + 3 * u + 1] = nj;
                utab[e * TR + 3 * u + 2] = line ? p.wphi[(size_t)col * p.Nlines * Ns + slots[u].wphi_off + e]
                                                : p.nsr[(size_t)col * p.Ncont * Ns + slots[u].base + e];
            }
            utab[e * TR + 3 * NPT + 0] = e > 0 ? 0.5 * fabs(zc[e - 1] - zc[e]) : 0.0;
            utab[e * TR + 3 * NPT + 1] = p.sca[(size_t)col * Ns + e];
        }
    }
    etab[threadIdx.x] = p.exp2_tab[threadIdx.x];
    __syncthreads();

    const double wav = p.wavelength[la];
    const double u_la = p.u_la[la];
    const double zmu_l = p.zmu[mu];
    const double wmuh_l = valid ? p.wmuh[mu] : 0.0;
    const double wq_l = wmuh_l * (4.0 * kPi);
    const bool compact = p.phi_compact != 0;
    const int kS = dir ? Ns - 1 : 0;
    const int dk = dir ? -1 : 1;
    const int raysel = compact ? 0 : dir * Ns * NR + mu;
    const int kmul = compact ? 1 : NR;
    unsigned pact = 0;
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const int l = la - slots[u].Nblue;
        if (l >= 0 && l < slots[u].Nlam && p.active[slots[u].trans * Nspect + la] != 0) pact |= 1u << u;
    }
    int idx0[NS], kstr[NS];
    double wlv[NS], alv[NS];
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const bool a = (pact >> u) & 1u;
        const int l = a ? la - slots[u].Nblue : 0;
        const bool line = u < NL;
        const int len = slots[u].len;
        const int lb = a ? la - slots[u].first : 0;
        idx0[u] = line ? (a ? PG * (slots[u].base + raysel * len) + cphi * len + lb : PG * (int)p.phi_col_stride - 1) : 0;
        kstr[u] = (line && a) ? PG * kmul * len : 0;
        wlv[u] = (a && valid) ? wq_l * p.wl[slots[u].wl_off + l] : 0.0;
        alv[u] = (a && !line) ? p.alpha[slots[u].wl_off + l] : 0.0;
    }

    // ---- one depth: the streams (requested one depth ahead) and what the front half of a step makes of them
    struct Str { double bc, be, jd, E, sv[NS], cr[NLK][3]; };
    auto stream_loads = [&](int kk, Str& o) __attribute__((always_inline)) {
        const unsigned kko = (unsigned)(kk * L + j) * 8u;
        o.jd = at(Jdag, kko);
        o.bc = at(bgchi, kko);
        o.be = at(bgeta, kko);
        o.E = 0.0;
#pragma unroll
        for (int u = 0; u < NL; ++u) o.sv[u] = at(phi_col, (unsigned)(idx0[u] + kk * kstr[u]) * 8u);
        if constexpr (HASC) o.E = at(Eb, kko);
        if constexpr (LK) {
#pragma unroll
            for (int u = 0; u < NL; ++u)
#pragma unroll
                for (int q = 0; q < NCR; ++q) o.cr[u][q] = at(corr, (unsigned)((3 * u + q) * plane) * 8u + kko);
        }
    };
    struct Dep { double chi, rchi, S, jd, pv[NS], ch[NS], Uji[NS], eta[NS], cr[NLK][3]; };
    // opacity, emissivity, source function of depth k from its streams (rh_method.py:601-632)
    auto front = [&](int k, const Str& o, Dep& d) __attribute__((always_inline)) {
        const lds_f64* tk = utab + k * TR;
        double chiTot = o.bc, etaTot = o.be + tk[3 * NPT + 1] * o.jd;
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const double ni = tk[3 * u + 0], nj = tk[3 * u + 1];
            if (u < NL) {
                d.pv[u] = o.sv[u];
                d.ch[u] = ni * d.pv[u];                                    // (the table holds cB (n_i - g n_j))
                d.Uji[u] = slots[u].Uc * d.pv[u];
            } else {
                const double g = ((pact >> u) & 1u) ? tk[3 * u + 2] * o.E : 0.0;
                d.pv[u] = g * alv[u];                                      // Vji
                d.ch[u] = ni * alv[u] - nj * d.pv[u];
                d.Uji[u] = u_la * d.pv[u];
            }
            d.eta[u] = nj * d.Uji[u];
            chiTot += d.ch[u];
            etaTot += d.eta[u];
        }
        d.chi = chiTot;
        d.rchi = rcp(chiTot);
        d.S = etaTot * d.rchi;
        d.jd = o.jd;
        if constexpr (LK) {
#pragma unroll
            for (int u = 0; u < NL; ++u)
#pragma unroll
                for (int q = 0; q < NCR; ++q) d.cr[u][q] = o.cr[u][q];
        }
    };

    // ---- boundary conditions: formal_solver.py:203-209 (unchanged by the rule)
    Str sA, sB;
    Dep dprev, dcur;
    stream_loads(kS, sA);
    stream_loads(kS + dk, sB);
    front(kS, sA, dprev);
    double Iu = 0.0;
    if (dir) {
        Dep d1;
        front(kS + dk, sB, d1);
        const double dtau_uw = zmu_l * (dprev.chi + d1.chi) * 0.5 * fabs(z[kS] - z[kS + dk]);
        const double B0 = planck(tcol[Ns - 2], wav), B1 = planck(tcol[Ns - 1], wav);
        Iu = B1 - (B0 - B1) / dtau_uw;
    }
    double S_u = 0.0, dtau_u = 1.0, rdt_u = 1.0;   // source function of the upwind depth, optical depth of the interval behind the local one and its reciprocal
    double dJ = 0.0;

    // finish depth m (this wave's m-th): formal solution, angle sums, Gamma integrands, J.  `d`: the depth's own values,
    // (Sd, dtau_d, has_d): its downwind neighbour
    auto finish = [&](const int m, auto phase_c, const Dep& d, double Sd, double dtau_d, bool has_d) __attribute__((always_inline)) {
        constexpr int PH = decltype(phase_c)::value;      // 0 first visitor, 1 midpoint, 2 second visitor
        const int k = kS + dk * m;
        const unsigned kl = (unsigned)(k * L + j) * 8u;
        const lds_f64* tk = utab + k * TR;
        if constexpr (PH == 2) {
            if (2 * m == Ns || 2 * m == Ns + 1) __syncthreads();      // the partner wave's first-half stores
        }
        double jhalf = 0.0;
        if constexpr (PH == 2) jhalf = at(Jnew, kl);
        double I, Lam, rdt_d;
        if (m == 0) {
            I = Iu;
            Lam = 0.0;
            rdt_d = rcp(dtau_d);
        } else {
            const Para r = parabolic_point_fast<!LK>(Iu, S_u, d.S, has_d ? Sd : 0.0, dtau_u, rdt_u, has_d ? dtau_d : 1.0, has_d, etab, rdt_d);
            I = r.I;
            Lam = r.Lam;
        }
        const double Psi = Lam * d.rchi;
        Iu = I;
        S_u = d.S;
        dtau_u = dtau_d;
        rdt_u = rdt_d;
        if (m == Ns - 1 && dir == 1 && valid) p.Iout[((size_t)col * Nspect + la) * NR + mu] = I;
        // angle quadrature: J (:640), Psibar and sum_mu w Psi* phi for the fast / linked continua
        xrow[lane] = wmuh_l * I;
        __builtin_amdgcn_wave_barrier();
        double Jsum = xrow[j];
#pragma unroll
        for (int q = 1; q < NR; ++q) Jsum += xrow[q * L + j];
        if (nF > 0) {
            const double wP = wq_l * Psi;
            __builtin_amdgcn_wave_barrier();
            xrow[lane] = wP;
            if constexpr (LK) {
#pragma unroll
                for (int u = 0; u < NL; ++u) xrow2[u * LSX_WAVE + lane] = wP * d.pv[u];
            }
            __builtin_amdgcn_wave_barrier();
            double sPsi = xrow[j];
#pragma unroll
            for (int q = 1; q < NR; ++q) sPsi += xrow[q * L + j];
            at(psibar, kl) = sPsi;
            if constexpr (LK) {
#pragma unroll
                for (int u = 0; u < NL; ++u) {
                    double sPP = xrow2[u * LSX_WAVE + j];
#pragma unroll
                    for (int q = 1; q < NR; ++q) sPP += xrow2[u * LSX_WAVE + q * L + j];
                    at(ppsum, (unsigned)(u * plane) * 8u + kl) = sPP;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // Gamma integrands (rh_method.py:643-681), the level bookkeeping of :616-627 from the tile's (at most two) slots
        double g1v[NS], g2v[NS];
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const bool line = u < NL;
            const double Vij = line ? slots[u].cB * d.pv[u] : alv[u];
            const double Vji = line ? slots[u].Vc * d.pv[u] : d.pv[u];
            double etaA = d.eta[u], chi_i = d.ch[u], chi_j = -d.ch[u], U_j = d.Uji[u], U_i = 0.0;
            if constexpr (NPT == 2) {
                const int v = 1 - u;
                if constexpr (TOPO == 1) { etaA += d.eta[v]; chi_i += d.ch[v]; }
                else if constexpr (TOPO == 0) {
                    const auto* rel = slots[u].rel[0];
                    etaA = fma(rel[REL_EA], d.eta[v], etaA);
                    chi_i = fma(rel[REL_CI], d.ch[v], chi_i);
                    chi_j = fma(rel[REL_CJ], d.ch[v], chi_j);
                    U_j = fma(rel[REL_UJ], d.Uji[v], U_j);
                    U_i = rel[REL_UI] * d.Uji[v];
                }
            }
            if constexpr (LK) {
                if (line) {
                    etaA += d.cr[u < NLK ? u : 0][0];
                    chi_i += d.cr[u < NLK ? u : 0][1];
                    if constexpr (NCR > 2) chi_j += d.cr[u < NLK ? u : 0][2];
                }
            }
            const double wt = line ? wlv[u] * tk[3 * u + 2] : wlv[u];          // :451 (lines: x wphi), :455, :665
            const double Ieff = I - Psi * etaA;                                // :652
            g1v[u] = wt * ((d.Uji[u] + Vji * Ieff) - (chi_i * Psi) * U_j);      // :677
            g2v[u] = wt * ((Vij * Ieff) - (chi_j * Psi) * U_i);                // :680
        }
        if constexpr (NPT == 1) {
            const double t = reduce_pair(g1v[0], g2v[0]);
            if (lane == 31) gpart[(0 * 2 + dir) * Ns + k] = t;
            if (lane == 63) gpart[(1 * 2 + dir) * Ns + k] = t;
        } else if constexpr (NPT == 2) {
            const double t = reduce_quad(g1v[0], g2v[0], g1v[1], g2v[1]);      // lane 15: g1[0], 47: g2[0], 31: g1[1], 63: g2[1]
            if ((lane & 15) == 15) {
                const int q = ((lane >> 4) & 1) * 2 + (lane >> 5);            // 15 -> 0, 47 -> 1, 31 -> 2, 63 -> 3
                gpart[(q * 2 + dir) * Ns + k] = t;
            }
        }
        // J: the two directions meet at depth k at different steps
        if constexpr (PH == 0) {
            at(Jnew, kl) = Jsum;
        } else if constexpr (PH == 1) {
            if (lead) xwg[dir * LSX_WAVE + j] = Jsum;
            __syncthreads();
            if (lead && dir == 0) {
                const double Jv = Jsum + xwg[LSX_WAVE + j];
                at(Jnew, kl) = Jv;
                dJ = nanmax(dJ, fabs(1.0 - d.jd * rcp(Jv)));
            }
        } else {
            const double Jv = jhalf + Jsum;
            at(Jnew, kl) = Jv;
            dJ = nanmax(dJ, fabs(1.0 - d.jd * rcp(Jv)));
        }
    };
    // step s (s >= 1): depth s's streams are in `cs`, the next depth's are requested into `ns`; depth s - 1 is finished
    auto step = [&](const int s, auto phase_c, Str& cs, Str& ns) __attribute__((always_inline)) {
        const int k = kS + dk * s;
        if (s + 1 < Ns) stream_loads(k + dk, ns);
        front(k, cs, dcur);
        // optical depth of the interval between depths s - 1 and s along this ray: the table row "behind" depth s
        const double dtau_d = (dprev.chi + dcur.chi) * ((utab + TR * dir)[k * TR + 3 * NPT] * zmu_l);
        finish(s - 1, phase_c, dprev, dcur.S, dtau_d, true);
        dprev = dcur;
    };
    {
        // depth m is finished in step m + 1; m < nA: first visitor, m = nA (odd Nspace): the midpoint, then second visitor
        const int nA = Ns / 2;
        auto one = [&](int s, auto ph) __attribute__((always_inline)) { if (s & 1) step(s, ph, sB, sA); else step(s, ph, sA, sB); };
        int s = 1;
        for (; s <= nA; ++s) one(s, std::integral_constant<int, 0>{});                     // finishes m = 0 .. nA - 1
        if (Ns & 1) { one(s, std::integral_constant<int, 1>{}); ++s; }                     // m = nA
        for (; s < Ns; ++s) one(s, std::integral_constant<int, 2>{});                      // m up to Ns - 2
        finish(Ns - 1, std::integral_constant<int, 2>{}, dprev, 0.0, 1.0, false);          // the end point: the linear rule with its own weights
    }
    const double dJw = wave_max_nan(lead ? dJ : 0.0);
    if (lane == 0) p.dJpart[((size_t)col * ntile + tile_id) * 2 + dir] = dJw;
}

// One kernel per (NPT, NR, SCAL) class; the host launches the classes of a call on separate streams
// so they share the machine (a class alone would leave a tail).
// register budget per class: tiles without per-ray slots fit 5 waves/SIMD, the others 4 (more would spill)
#ifndef LSX_WPE0
#define LSX_WPE0 (LSX_WAVES_PER_EU + 1)
#endif
#ifndef LSX_WPE1
#define LSX_WPE1 (LSX_WAVES_PER_EU + 1)     // one per-ray slot: 96 VGPRs, no spills
#endif
#ifndef LSX_WPE2
#define LSX_WPE2 LSX_WAVES_PER_EU
#endif
#ifndef LSX_WPE3
#define LSX_WPE3 (LSX_WAVES_PER_EU - 1)     // three per-ray slots: 4 waves/SIMD would spill ~60 VGPRs
#endif
#ifndef LSX_WPE4
#define LSX_WPE4 (LSX_WAVES_PER_EU - 1)
#endif
#define LSX_WPE(NPT) ((NPT) == 0 ? LSX_WPE0 : ((NPT) == 1 ? LSX_WPE1 : ((NPT) == 2 ? LSX_WPE2 : ((NPT) == 3 ? LSX_WPE3 : ((NPT) == 4 ? LSX_WPE4 : LSX_WAVES_PER_EU)))))
// ... and one wave per SIMD less where the class's budget spilled (round 5: no instance the planner can dispatch uses scratch --
// tests/test_no_scratch.py parses the compiler's resource report): two and three slots with linked continua (their correction
// streams), one slot with linked continua at a run-time ray count, four slots
#define LSX_WPE_OF(NPT, NR, LK) (LSX_WPE(NPT) - ((((NPT) >= 2 && (LK)) || ((NPT) == 1 && (LK) && (NR) == 0) || (NPT) == 4) ? 1 : 0))
template <int NPT, int NL, int NR, bool SCAL, bool LK, int TOPO = 0>
__global__ void __launch_bounds__(2 * LSX_WAVE) __attribute__((amdgpu_waves_per_eu(LSX_WPE_OF(NPT, NR, LK))))
lsx_sweep_kernel(const SweepParams p)
{
    // XCD-aware block -> (column, tile): workgroups are dealt round-robin over the 8 XCDs (b and
    // b+8 share one), so every XCD gets a contiguous range of (column, tile) pairs.  A different
    // placement would change speed only, never results.
    int vb;
    {
        const int nb = gridDim.x, x = blockIdx.x & 7, q = blockIdx.x >> 3;
        const int nb8 = nb >> 3, rem = nb & 7;
        vb = x * nb8 + (x < rem ? x : rem) + q;
    }
    const int col = vb / p.n_class_tiles;
    const int tile_id = LSX_CONST(int32_t, p.class_tiles)[vb - col * p.n_class_tiles];
    if (p.colmask && LSX_CONST(uint8_t, p.colmask)[col] == 0) {
        // frozen column: nothing is computed; J only moves to the other half of the ping-pong pair
        const size_t tb = ((size_t)col * p.ntile_total + tile_id) * p.Nspace * p.L;
        for (int e = threadIdx.x; e < p.Nspace * p.L; e += 2 * LSX_WAVE) p.Jnew_T[tb + e] = p.Jdag_T[tb + e];
        return;
    }
    sweep_tile<NPT, NL, NR, SCAL, LK, TOPO>(p, vb, tile_id);
}

// Small batches (a few columns) are latency bound: one launch that dispatches on the tile's class inside
// beats five tiny launches on five streams.
// The instances are INLINED into the one kernel, side by side behind the dispatch: no call frame (the called form kept 400 bytes of
// stack per lane for the callee-saved registers), and the scalar registers the instances compete for spill into vector lanes
// (383 `v_writelane`s, no memory) -- at the one or two waves per SIMD this kernel runs with there are vector registers to spare.
// Measured on the single FALC column: formal solution 83.6 -> 78.8 us, MALI iteration 79.0 -> 74.4 us
// (profiles/r04/c2_instances_inlined.txt).  LSX_TILE_CALL='__attribute__((noinline))' builds the called form.
#ifndef LSX_TILE_CALL
#define LSX_TILE_CALL __attribute__((always_inline))
#endif
template <int NPT, int NL, int NR, bool SCAL, bool LK, int TOPO = 0>
__device__ LSX_TILE_CALL void sweep_tile_call(const SweepParams& p, const int vb, const int tile_id)
{
    sweep_tile<NPT, NL, NR, SCAL, LK, TOPO>(p, vb, tile_id);
}
// the fast-continuum kernels' view of a sweep launch's parameters (lsx_fast.h; the fused launch only)
static __device__ __forceinline__ FastParams fast_params_of(const SweepParams& p)
{
    FastParams f{};
    f.Nspace = p.Nspace; f.Nspect = p.Nspect; f.Nrays = p.Nrays; f.ncol = p.ncol; f.ntile = p.ntile_total; f.L = p.L; f.NLtot = p.NLtot;
    f.Natoms = p.Natoms; f.nslot_total = p.nslot_total; f.tiles = p.tiles; f.slots = p.slots; f.active = p.active; f.alpha = p.alpha;
    f.wl = p.wl; f.u_la = p.u_la; f.wmuh = p.wmuh; f.n = p.n; f.nsr = p.nsr; f.E_T = p.E_T; f.Ncont = p.Ncont; f.nF_max = p.nF_max;
    f.seg_depths = p.Nspace; f.bgchi_T = p.bgchi_T; f.bgeta_T = p.bgeta_T;
    f.bgxchi_T = const_cast<double*>(p.bgxchi_T); f.bgxeta_T = const_cast<double*>(p.bgxeta_T); f.corr_T = const_cast<double*>(p.corr_T);
    f.corr_col_stride = p.corr_col_stride; f.pp_col_stride = p.pp_col_stride; f.J_T = p.Jnew_T; f.Psi2_T = p.Psi2_T; f.Psi3_T = p.Psi3_T;
    f.Gpart = p.Gpart; f.colmask = p.colmask;
    f.fgtab = p.fgtab;
    f.epi_corr = 0; f.Nlines = p.Nlines; f.wphi = p.wphi;       // (one ray per lane: the sweep applies the linked corrections itself)
    return f;
}

template <int NR, bool SCAL>
__global__ void __launch_bounds__(2 * LSX_WAVE) __attribute__((amdgpu_waves_per_eu(1, 2)))
lsx_sweep_kernel_all(const SweepParams p)
{
    const int vb = blockIdx.x;
    const int col = vb / p.ntile_total;
    const int tile_id = vb - col * p.ntile_total;
    if (p.colmask && LSX_CONST(uint8_t, p.colmask)[col] == 0) {
        const size_t tb = (size_t)vb * p.Nspace * p.L;
        for (int e = threadIdx.x; e < p.Nspace * p.L; e += 2 * LSX_WAVE) p.Jnew_T[tb + e] = p.Jdag_T[tb + e];
        return;
    }
    // A single column (any batch below 32) is latency bound: three launches in a row -- pre-pass of the tiles with fast continua,
    // sweep, their Gamma epilogue -- cost 8 + 54 + 9 us for FALC CaII, although only three short continuum tiles need the first and
    // the last.  Here the workgroup of such a tile does both itself, around its own sweep: the line tiles next to it take longer
    // anyway.  The same device functions as the stand-alone kernels (lsx_fast.h): the same bits.
    extern __shared__ __attribute__((aligned(16))) double lds_fast[];
    const bool fast_here = NR == 5 && !SCAL && p.fused_fast != 0 && (LSX_CONST(DevTile, p.tiles) + tile_id)->nF > 0;
    if constexpr (NR == 5 && !SCAL) {
        if (fast_here) {
            const FastParams f = fast_params_of(p);
            if (!(p.fused_fast & 2)) fast_prepass_tile<false, 2 * LSX_WAVE>(f, tile_id, (size_t)col, lds_fast);
            __syncthreads();                 // (the workgroup's own global writes are visible to it behind the barrier)
        }
    }
    const int nP = (LSX_CONST(DevTile, p.tiles) + tile_id)->nP, nL = (LSX_CONST(DevTile, p.tiles) + tile_id)->nL;
    const bool lk = (LSX_CONST(DevTile, p.tiles) + tile_id)->nK > 0;
    if (nP > p.static_max) { if (lk) sweep_tile_call<-1, 0, NR, SCAL, true>(p, vb, tile_id); else sweep_tile_call<-1, 0, NR, SCAL, false>(p, vb, tile_id); }
    else if (nP == 0) sweep_tile_call<0, 0, NR, SCAL, false>(p, vb, tile_id);
    else if (nP == 1 && nL == 1) { if (lk) sweep_tile_call<1, 1, NR, SCAL, true>(p, vb, tile_id); else sweep_tile_call<1, 1, NR, SCAL, false>(p, vb, tile_id); }
    else if (nP == 2 && nL == 2) { if (lk) sweep_tile_call<2, 2, NR, SCAL, true>(p, vb, tile_id); else sweep_tile_call<2, 2, NR, SCAL, false>(p, vb, tile_id); }
    else if (nP == 2 && nL == 1 && !lk) sweep_tile_call<2, 1, NR, SCAL, false>(p, vb, tile_id);
    else if (lk) sweep_tile_call<-1, 0, NR, SCAL, true>(p, vb, tile_id);     // the remaining shapes take the generic path here (code size)
    else sweep_tile_call<-1, 0, NR, SCAL, false>(p, vb, tile_id);
    if constexpr (NR == 5 && !SCAL) {
        if (fast_here && !(p.fused_fast & 4)) {       // J, Psibar and Psi* phi of this (tile, column) are complete: both directions are this workgroup's waves
            __syncthreads();
            const FastParams f = fast_params_of(p);
            const auto* tp = LSX_CONST(DevTile, p.tiles) + tile_id;
            const int nlc = tp->nK > 0 ? tp->nL : 0;            // lines the linked continua feed: 0, 1 or 2 (the host checked)
            const long r0 = (long)col * p.Nspace, r1 = r0 + p.Nspace;
            for (long rb = r0; rb < r1; rb += 2 * LSX_FGC_ROWS) {
                if (nlc == 0) fast_gamma_cols_rows<0, 6, 2>(f, tile_id, rb, r1, lds_fast);
                else if (nlc == 1) fast_gamma_cols_rows<1, 6, 2>(f, tile_id, rb, r1, lds_fast);
                else fast_gamma_cols_rows<2, 6, 2>(f, tile_id, rb, r1, lds_fast);
            }
        }
    }
}


// N4: every tile of every column through the parabolic instance (one launch, like the fused small-batch kernel)
template <int NR, bool SCAL>
__global__ void __launch_bounds__(2 * LSX_WAVE) __attribute__((amdgpu_waves_per_eu(LSX_WAVES_PER_EU)))
lsx_sweep_kernel_parabolic(const SweepParams p)
{
    // every tile of every column (fused launch: no tile list), or the tiles of one class that has no compile-time instance
    const int vb = blockIdx.x;
    const int col = vb / p.n_class_tiles;
    const int tile_id = p.class_tiles ? LSX_CONST(int32_t, p.class_tiles)[vb - col * p.n_class_tiles] : vb - col * p.n_class_tiles;
    if (p.colmask && LSX_CONST(uint8_t, p.colmask)[col] == 0) {
        const size_t tb = ((size_t)col * p.ntile_total + tile_id) * p.Nspace * p.L;
        for (int e = threadIdx.x; e < p.Nspace * p.L; e += 2 * LSX_WAVE) p.Jnew_T[tb + e] = p.Jdag_T[tb + e];
        return;
    }
    sweep_tile_parabolic<NR, SCAL>(p, vb, tile_id);
}

#ifndef LSX_PAR_WPE0
#define LSX_PAR_WPE0 4
#endif
#ifndef LSX_PAR_WPE1
#define LSX_PAR_WPE1 3
#endif
#ifndef LSX_PAR_WPE2
#define LSX_PAR_WPE2 2
#endif
// N4, compile-time classes: one kernel per (slots, lines, linked, relation), five rays
template <int NPT, int NL, bool LK, int TOPO>
__global__ void __launch_bounds__(2 * LSX_WAVE) __attribute__((amdgpu_waves_per_eu(NPT == 0 ? LSX_PAR_WPE0 : NPT == 1 ? (LK ? LSX_PAR_WPE1 : LSX_PAR_WPE1 + 1) : (LK ? LSX_PAR_WPE2 : LSX_PAR_WPE2 + 1))))
lsx_sweep_kernel_par(const SweepParams p)
{
    int vb;
    {
        const int nb = gridDim.x, x = blockIdx.x & 7, q = blockIdx.x >> 3;
        const int nb8 = nb >> 3, rem = nb & 7;
        vb = x * nb8 + (x < rem ? x : rem) + q;
    }
    const int col = vb / p.n_class_tiles;
    const int tile_id = LSX_CONST(int32_t, p.class_tiles)[vb - col * p.n_class_tiles];
    if (p.colmask && LSX_CONST(uint8_t, p.colmask)[col] == 0) {
        const size_t tb = ((size_t)col * p.ntile_total + tile_id) * p.Nspace * p.L;
        for (int e = threadIdx.x; e < p.Nspace * p.L; e += 2 * LSX_WAVE) p.Jnew_T[tb + e] = p.Jdag_T[tb + e];
        return;
    }
    sweep_tile_par<NPT, NL, LSX_RS_RAYS, LK, TOPO>(p, vb, tile_id);
}

// the parabolic rule for one class: its compile-time instance where one exists (the shapes of LSX_RS_INSTANCES, five rays, one
// scattering coefficient per depth), the generic instance on the class's tile list otherwise
extern "C" hipError_t lsx_launch_sweep_par(const SweepParams* p, int code, int nblocks, size_t lds_bytes, hipStream_t st)
{
    const dim3 g(nblocks), b(2 * LSX_WAVE);
    if (p->Nrays == LSX_RS_RAYS && !p->sca_per_lambda) {
        switch (code) {
#define LSX_X(NPT, NL, LK, TOPO) \
        case lsx_class_code(NPT, NL, LK, TOPO): hipLaunchKernelGGL((lsx_sweep_kernel_par<NPT, NL, LK, TOPO>), g, b, lds_bytes, st, *p); return hipGetLastError();
        LSX_RS_INSTANCES(LSX_X)
#undef LSX_X
        default: break;
        }
    }
    if (!p->sca_per_lambda && p->Nrays == 5) hipLaunchKernelGGL((lsx_sweep_kernel_parabolic<5, false>), g, b, lds_bytes, st, *p);
#ifndef LSX_ONLY_NR5
    else if (!p->sca_per_lambda && p->Nrays == 3) hipLaunchKernelGGL((lsx_sweep_kernel_parabolic<3, false>), g, b, lds_bytes, st, *p);
    else hipLaunchKernelGGL((lsx_sweep_kernel_parabolic<0, true>), g, b, lds_bytes, st, *p);
#else
    else return hipErrorNotSupported;
#endif
    return hipGetLastError();
}

template <int NR, bool SCAL>
static hipError_t launch_class(const SweepParams& p, int code, dim3 g, dim3 b, size_t lds_bytes, hipStream_t st)
{
    // code: -2 fused, -4 every tile with the parabolic rule (N4), -1 generic, -3 generic with linked continua, else
    // lsx_class_code(per-ray slots, lines among them, linked continua, TOPO) of a compiled instance (lsx_plan.h: the list the
    // plan consults before it files a tile under a class).  Anything else is refused, never mapped to another instance.
    switch (code) {
    case -2: hipLaunchKernelGGL((lsx_sweep_kernel_all<NR, SCAL>), g, b, lds_bytes, st, p); break;
    case -4: hipLaunchKernelGGL((lsx_sweep_kernel_parabolic<NR, SCAL>), g, b, lds_bytes, st, p); break;
    case -3: hipLaunchKernelGGL((lsx_sweep_kernel<-1, 0, NR, SCAL, true>), g, b, lds_bytes, st, p); break;
    case -1: hipLaunchKernelGGL((lsx_sweep_kernel<-1, 0, NR, SCAL, false>), g, b, lds_bytes, st, p); break;
#define LSX_X(NPT, NL, LK, TOPO) \
    case lsx_class_code(NPT, NL, LK, TOPO): hipLaunchKernelGGL((lsx_sweep_kernel<NPT, NL, NR, SCAL, LK, TOPO>), g, b, lds_bytes, st, p); break;
    LSX_SWEEP_INSTANCES(LSX_X)
#undef LSX_X
    default: return hipErrorNotSupported;
    }
    return hipGetLastError();
}

extern "C" hipError_t lsx_launch_sweep(const SweepParams* p, int npt, int nblocks, size_t lds_bytes, hipStream_t st)
{
    const dim3 g(nblocks), b(2 * LSX_WAVE);
    if (!p->sca_per_lambda && p->Nrays == 5) return launch_class<5, false>(*p, npt, g, b, lds_bytes, st);
#ifndef LSX_ONLY_NR5    // diagnostic builds (profiles/ab.sh variants) compile the 5-ray instances only
    if (!p->sca_per_lambda && p->Nrays == 3) return launch_class<3, false>(*p, npt, g, b, lds_bytes, st);
    return launch_class<0, true>(*p, npt, g, b, lds_bytes, st);
#else
    return hipErrorNotSupported;
#endif
}
