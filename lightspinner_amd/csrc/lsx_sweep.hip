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
//     that has a line in the tile) go through both passes.  FAST continua (atoms with no line in
//     the tile) are ray independent: their Gamma integrand is affine in I and Psi* with
//     ray-independent coefficients, so it follows from sum_mu w I and sum_mu w Psi*.
#include <hip/hip_runtime.h>
#include "lsx_dev.h"

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

// 1/x: v_rcp_f64 seed + two Newton steps (~1 ulp; 5 instructions instead of the ~12 of an
// IEEE division).  The reference divides; the difference is at the last-bit level.
__device__ __forceinline__ double rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// formal_solver.py:14-44.  The three regimes are selected per lane; the exponential is skipped
// for the whole wavefront when no lane is in the middle regime (top / bottom of the atmosphere).
__device__ __forceinline__ void w2(double dtau, double& w0, double& w1)
{
    const bool small = dtau < 5e-4;
    const bool large = dtau > 50.0;
    double a0 = 1.0, a1 = 1.0;
    if (__builtin_amdgcn_ballot_w64(!(small || large)) != 0) {
        const double e = exp(-dtau);
        a0 = 1.0 - e;
        a1 = a0 - dtau * e;
    }
    const double t0 = dtau * (1.0 - 0.5 * dtau);
    const double t1 = (dtau * dtau) * (0.5 - dtau * (1.0 / 3.0));
    w0 = small ? t0 : (large ? 1.0 : a0);
    w1 = small ? t1 : (large ? 1.0 : a1);
}

// utils.py:17-22
__device__ __forceinline__ double planck(double temp, double wav)
{
    const double hc_Tkla = kHC / (kKBoltzmann * kNM_TO_M * wav) / temp;
    const double x = kNM_TO_M * wav;
    const double twohnu3_c2 = (2.0 * kHC) / (x * x * x);
    return twohnu3_c2 / (exp(hc_Tkla) - 1.0);
}

#ifdef LSX_REDUCE_SHFL
__device__ __forceinline__ double wave_sum(double v) // total in every lane
{
#pragma unroll
    for (int m = 1; m < LSX_WAVE; m <<= 1) v += __shfl_xor(v, m, LSX_WAVE);
    return v;
}
#else
// DPP (VALU cross-lane moves, no LDS crossbar traffic).  The total of the 64 lanes ends up in
// lane 63 (the lane that stores it).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_f64<0xB1, 0xf>(v);  // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xf>(v);  // quad_perm [2,3,0,1]: every lane of a quad holds the quad sum
    v += dpp_f64<0x141, 0xf>(v); // row_half_mirror: 8-lane sums
    v += dpp_f64<0x140, 0xf>(v); // row_mirror: 16-lane (row) sums in every lane of the row
    v += dpp_f64<0x142, 0xa>(v); // row_bcast15 into rows 1 and 3: lane 31 = rows 0+1, lane 63 = rows 2+3
    v += dpp_f64<0x143, 0xc>(v); // row_bcast31 into rows 2 and 3: lane 63 = all four rows
    return v;
}
#endif

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
__device__ __forceinline__ void cell_acc(double* p, double v, bool first)
{
    if (first) *p = v;
    else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

} // namespace

#ifndef LSX_WAVES_PER_EU
#define LSX_WAVES_PER_EU 5
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
template <int NPT>
__device__ __forceinline__ void sweep_tile(const SweepParams& p, const int vb)
{
    extern __shared__ double lds[];
    constexpr bool STATIC = NPT >= 0;
    constexpr int NS = NPT > 0 ? NPT : 1;
    const int lane = threadIdx.x & (LSX_WAVE - 1);
    const int dir = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // 0: down (toFrom False), 1: up (True)
    const int ntile = p.ntile_total;
    const int col = vb / ntile;
    const int tile_id = vb - col * ntile;
    const auto* tilep = LSX_CONST(DevTile, p.tiles) + tile_id;
    const int la0 = tilep->la0, nla = tilep->nla, slot0 = tilep->slot0;
    const int nP = STATIC ? NPT : tilep->nP;
    const int nF = tilep->nF;
    const auto* slots = LSX_CONST(DevSlot, p.slots) + slot0;   // [0, nP): per-ray, [nP, nP+nF): fast
    const auto* fslots = slots + nP;
    const int Ns = p.Nspace;
    const int Nspect = p.Nspect;
    const int Nrays = p.Nrays;
    const int L = p.L;

    // lane -> ray.  Lanes without a ray shadow a real one (finite arithmetic) and are masked out of
    // every store and reduction.
    const int mu_raw = lane / L;
    const int j_raw = lane - mu_raw * L;
    const bool valid = mu_raw < Nrays && j_raw < nla;
    const int mu = mu_raw < Nrays ? mu_raw : Nrays - 1;
    const int j = j_raw < nla ? j_raw : nla - 1;
    const int la = la0 + j;
    const bool lead = valid && mu_raw == 0; // one lane per wavelength: owns J[la, k]

    // LDS rows (64 doubles each), private to this wave except the two exchange rows at the end
    const int rows = 2 * p.ncell_lev + p.ncell_atom + 1;
    double* const wrow = lds + (size_t)dir * rows * LSX_WAVE + lane;
#define CCHI(c) wrow[(2 * (c)) * LSX_WAVE]
#define CU(c) wrow[(2 * (c) + 1) * LSX_WAVE]
#define CETA(a) wrow[(2 * p.ncell_lev + (a)) * LSX_WAVE]
    double* const xrow = lds + (size_t)dir * rows * LSX_WAVE + (size_t)(rows - 1) * LSX_WAVE; // angle sums
    double* const xwg = lds + (size_t)2 * rows * LSX_WAVE;                                    // [2][64] cross-wave

    // column bases; wave-uniform reads go through the scalar cache
    const auto* n_col = LSX_CONST(double, p.n + (size_t)col * p.NLtot * Ns);
    const auto* wphi_col = LSX_CONST(double, p.wphi + (size_t)col * p.Nlines * Ns);
    const auto* z = LSX_CONST(double, p.height + (size_t)col * Ns);
    const auto* tcol = LSX_CONST(double, p.temperature + (size_t)col * Ns);
    // tile-major streams of this (column, tile): [k][j]
    const size_t tbase = ((size_t)col * ntile + tile_id) * Ns * L;
    const double* __restrict__ bgchi = p.bgchi_T + tbase;
    const double* __restrict__ bgeta = p.bgeta_T + tbase;
    const double* __restrict__ Jdag = p.Jdag_T + tbase;
    double* __restrict__ Jnew = p.Jnew_T + tbase;
    const double* __restrict__ sca = p.sca_per_lambda ? p.sca + tbase : p.sca + (size_t)col * Ns;
    const double* __restrict__ phi_col = p.phi_T + (size_t)col * p.phi_col_stride;
    const double* __restrict__ gijc_col = p.gijc_T + (size_t)col * p.gijc_col_stride;
    double* __restrict__ gpart = p.Gpart + ((size_t)col * p.nslot_total + slot0) * 4 * Ns;

    const double wav = p.wavelength[la];
    const double u_la = p.u_la[la];
    const double zmu_l = p.zmu[mu];
    const double wmuh_l = valid ? p.wmuh[mu] : 0.0;
    const double wq_l = wmuh_l * (4.0 * kPi);
    const bool compact = p.phi_compact != 0;
    const int kS = dir ? Ns - 1 : 0;
    const int dk = dir ? -1 : 1;
    // ray part of a line-profile index: ((k*2 + dir)*Nrays + mu) * Nlam + lt
    const int raysel = compact ? 0 : dir * Nrays + mu;
    const int kmul = compact ? 1 : 2 * Nrays;

    // activity bits of this lane's wavelength
    unsigned pact = 0, fact = 0;
    for (int u = 0; u < nP; ++u) {
        const int l = la - slots[u].Nblue;
        if (l >= 0 && l < slots[u].Nlam && p.active[slots[u].trans * Nspect + la] != 0) pact |= 1u << u;
    }
    for (int f = 0; f < nF; ++f) {
        const int l = la - fslots[f].Nblue;
        if (l >= 0 && l < fslots[f].Nlam && p.active[fslots[f].trans * Nspect + la] != 0) fact |= 1u << f;
    }

    // static path: per-slot lane state in registers
    //   idx0: element index of (depth 0, this ray, this wavelength) in phi_T (line) / gijc_T (continuum)
    //   wl: wavelength quadrature weight, al: alpha (continua)
    int idx0[NS], kstr[NS];
    double wlv[NS], alv[NS];
    if constexpr (STATIC) {
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            const bool a = (pact >> u) & 1u;
            const int l = a ? la - slots[u].Nblue : 0;
            const int Nlam = slots[u].Nlam;
            const bool line = (slots[u].flags & SLOT_LINE) != 0;
            const int len = slots[u].len;
            const int lb = a ? la - slots[u].first : 0;      // position inside the (tile, transition) block
            (void)Nlam;
            idx0[u] = slots[u].base + (line ? raysel * len : 0) + lb;
            kstr[u] = line ? kmul * len : len;
            wlv[u] = a ? p.wl[slots[u].wl_off + l] : 0.0;
            alv[u] = (a && !line) ? p.alpha[slots[u].wl_off + l] : 0.0;
        }
    }

    // total opacity at depth kk (boundary-condition look-ahead, formal_solver.py:204-207)
    auto chi_at = [&](int kk) -> double {
        double c = bgchi[kk * L + j];
        for (int u = 0; u < nP + nF; ++u) {
            const SlotS sl = load_slot(slots + u, Ns);
            const double ni = n_col[sl.noff_i + kk];
            const double nj = n_col[sl.noff_j + kk];
            const bool a = u < nP ? (pact >> u) & 1u : (fact >> (u - nP)) & 1u;
            const int l = a ? la - sl.Nblue : 0;
            const int lb = a ? la - sl.first : 0;
            if (sl.flags & SLOT_LINE) {
                const double pv = a ? phi_col[sl.base + (kk * kmul + raysel) * sl.len + lb] : 0.0;
                c += (sl.cB * (ni - sl.g * nj)) * pv;
            } else {
                const double g = a ? gijc_col[sl.base + kk * sl.len + lb] : 0.0;
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
    double sW = 0.0;
    for (int m = 0; m < Nrays; ++m) sW += LSX_CONST(double, p.wmuh)[m] * (4.0 * kPi);
    // diagnostic build only (-DLSX_STAMPS): per-segment shader-clock totals of a few sample waves go
    // to p.debug, a buffer nothing else reads (cdna_hip_programming.md, In-kernel stamps)
#ifdef LSX_STAMPS
    unsigned long long T[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#define STAMP(i)                                                                                          \
    do {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        T[i] += t1 - t0;                                                                                  \
        t0 = t1;                                                                                          \
    } while (0)
#else
#define STAMP(i)
#endif

    for (int s = 0; s < Ns; ++s) {
        const int k = kS + dk * s;
        const int kl = k * L + j;                       // position in the tile-major [k][j] streams
        // ---- every HBM / table read of the per-ray slots for this depth, in one batch ----
        const double jd = Jdag[kl];
        double chiTot = bgchi[kl];
        const double be_l = bgeta[kl];
        double sv[NS], sni[NS], snj[NS], swp[NS];
        if constexpr (STATIC) {
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const bool a = (pact >> u) & 1u;
                const bool line = (slots[u].flags & SLOT_LINE) != 0;
                const double* tab = line ? phi_col : gijc_col;
                sv[u] = a ? tab[idx0[u] + k * kstr[u]] : 0.0;
                sni[u] = n_col[slots[u].li * Ns + k];
                snj[u] = n_col[slots[u].lj * Ns + k];
                swp[u] = line ? wphi_col[slots[u].wphi_off + k] : 1.0;
            }
        }
        const double zk = z[k];
        const double hdzm = (0.5 * fabs(zprev - zk)) * zmu_l;
        zprev = zk;
        double etaTot = be_l + (p.sca_per_lambda ? sca[kl] : LSX_CONST(double, sca)[k]) * jd;
        STAMP(0);

        // ---- fast continua: opacity, emissivity, level cells (ray independent) ----------
        for (int f = 0; f < nF; ++f) {
            const SlotS sl = load_slot(fslots + f, Ns);
            const int fl = sl.flags;
            const double ni = n_col[sl.noff_i + k];
            const double nj = n_col[sl.noff_j + k];
            const bool a = (fact >> f) & 1u;
            const int l = a ? la - sl.Nblue : 0;
            const double g = a ? gijc_col[sl.base + k * sl.len + (la - sl.first)] : 0.0;
            const double alf = a ? p.alpha[sl.wl_off + l] : 0.0;
            const double Vji = g * alf;                 // rh_method.py:284-285
            const double chi = ni * alf - nj * Vji;     // :613
            const double Uji = u_la * Vji;              // :286
            const double eta = nj * Uji;                // :614
            if (fl & SLOT_LI_CELL) cell_acc(&CCHI(sl.ci), chi, fl & SLOT_CHI_I_FIRST);   // :619
            if (fl & SLOT_LJ_CELL) {
                cell_acc(&CCHI(sl.cj), -chi, fl & SLOT_CHI_J_FIRST);                     // :620
                cell_acc(&CU(sl.cj), Uji, fl & SLOT_U_J_FIRST);                          // :622
            }
            if (fl & SLOT_ETA_CELL) cell_acc(&CETA(sl.ca), eta, fl & SLOT_ETA_FIRST);    // :627
            chiTot += chi;
            etaTot += eta;
        }
        STAMP(1);

        // ---- pass 1: opacity / emissivity of the per-ray transitions (rh_method.py:601-627) ----
        //   kept for pass 2 (static path): pv = phi | Vji, chi, Uji
        double spv[NS], schi[NS], sUji[NS];
        auto pass1 = [&](const SlotS& sl, double v, double ni, double nj, double alf, double& pv, double& chi,
                         double& Uji) {
            const int fl = sl.flags;
            if (fl & SLOT_LINE) {
                pv = v;
                chi = (sl.cB * (ni - sl.g * nj)) * pv;   // n_i Vij - n_j Vji, :279-280, :613
                Uji = sl.Uc * pv;                        // :281
            } else {
                pv = v * alf;                            // Vji = g_ij alpha, :284-285
                chi = ni * alf - nj * pv;
                Uji = u_la * pv;                         // :286
            }
            const double eta = nj * Uji;                 // :614
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
                pass1(sl, sv[u], sni[u], snj[u], alv[u], spv[u], schi[u], sUji[u]);
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
                    v = a ? phi_col[sl.base + (k * kmul + raysel) * sl.len + (la - sl.first)] : 0.0;
                } else {
                    v = a ? gijc_col[sl.base + k * sl.len + (la - sl.first)] : 0.0;
                    alf = a ? p.alpha[sl.wl_off + l] : 0.0;
                }
                double pv, chi, Uji;
                pass1(sl, v, ni, nj, alf, pv, chi, Uji);
            }
        }
        STAMP(2);
        const double rchi = rcp(chiTot);
        const double S = etaTot * rchi;                 // :632

        // ---- formal solution at this depth (formal_solver.py:107-139) ----
        double I, Lam;
        if (s == 0) {
            I = Iu;
            Lam = 0.0;
        } else {
            const double dtau = (chi_prev + chiTot) * hdzm;
            const double rdt = rcp(dtau);
            const double dS = (S_prev - S) * rdt;
            // formal_solver.py:138-139: the end point re-uses the PREVIOUS interval's w and
            // S[kEnd - dk] with the fresh dS, dtau (reference behaviour, reproduced deliberately)
            const bool last = (s == Ns - 1);
            double w0, w1;
            w2(last ? dtau_prev : dtau, w0, w1);
            const double Sx = last ? S_prev : S;
            I = Iu * (1.0 - w0) + w0 * Sx + w1 * dS;
            Lam = w0 - w1 * rdt;
            dtau_prev = dtau;
        }
        const double Psi = Lam * rchi;
        Iu = I;
        chi_prev = chiTot;
        S_prev = S;
        if (s == Ns - 1 && dir == 1 && valid)            // emergent intensity, :638
            p.Iout[((size_t)col * Nspect + la) * Nrays + mu] = I;
        STAMP(3);

        // ---- angle quadrature of this wavelength: J (:640) and the two ray sums of the fast path ----
        xrow[lane] = wmuh_l * I;
        __builtin_amdgcn_wave_barrier();
        double Jsum = 0.0;
        for (int m = 0; m < Nrays; ++m) Jsum += xrow[m * L + j];
        double sPsi = 0.0;
        if (nF > 0) {
            __builtin_amdgcn_wave_barrier();
            xrow[lane] = wq_l * Psi;
            __builtin_amdgcn_wave_barrier();
            for (int m = 0; m < Nrays; ++m) sPsi += xrow[m * L + j];
        }
        __builtin_amdgcn_wave_barrier();
        STAMP(4);

        // ---- pass 2: Gamma integrands of the per-ray transitions (rh_method.py:643-681) ----
        auto pass2 = [&](const SlotS& sl, int u, bool a, double pv, double chi, double Uji, double Vij, double nj,
                         double wla) {
            const int fl = sl.flags;
            const double Vji = (fl & SLOT_LINE) ? sl.Vc * pv : pv;
            const double eta = nj * Uji;
            const double etaA = (fl & SLOT_ETA_CELL) ? CETA(sl.ca) : eta;
            const double chi_i = (fl & SLOT_LI_CELL) ? CCHI(sl.ci) : chi;
            const double chi_j = (fl & SLOT_LJ_CELL) ? CCHI(sl.cj) : -chi;
            const double U_j = (fl & SLOT_LJ_CELL) ? CU(sl.cj) : Uji;
            const double U_i = (fl & SLOT_UI_READ) ? CU(sl.ci) : 0.0;
            const double Ieff = I - Psi * etaA;                            // :652
            const double g1 = (Uji + Vji * Ieff) - (chi_i * Psi) * U_j;    // :677
            const double g2 = (Vij * Ieff) - (chi_j * Psi) * U_i;          // :680
            const double wt = (a && valid) ? wq_l * wla : 0.0;             // :665
            const double r1 = wave_sum(wt * g1);
            const double r2 = wave_sum(wt * g2);
            if (lane == LSX_WAVE - 1) {
                double* g = gpart + (u * 4 + dir) * Ns + k; // [slot][e][dir][k]
                g[0] = r1;          // e = 0: Gamma[i][j]
                g[2 * Ns] = r2;     // e = 1: Gamma[j][i]
            }
        };
        if constexpr (STATIC) {
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const SlotS sl = load_slot(slots + u, Ns);
                const bool line = (sl.flags & SLOT_LINE) != 0;
                const double Vij = line ? sl.cB * spv[u] : alv[u];
                pass2(sl, u, (pact >> u) & 1u, spv[u], schi[u], sUji[u], Vij, snj[u], wlv[u] * swp[u]); // :451, :455
            }
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
                    pv = a ? phi_col[sl.base + (k * kmul + raysel) * sl.len + (la - sl.first)] : 0.0; // L1/L2 hit
                    Vij = sl.cB * pv;
                    Uji = sl.Uc * pv;
                    chi = (sl.cB * (ni - sl.g * nj)) * pv;
                    wla *= wphi_col[sl.wphi_off + k];
                } else {
                    Vij = a ? p.alpha[sl.wl_off + l] : 0.0;
                    pv = (a ? gijc_col[sl.base + k * sl.len + (la - sl.first)] : 0.0) * Vij;          // Vji
                    Uji = u_la * pv;
                    chi = ni * Vij - nj * pv;
                }
                pass2(sl, u, a, pv, chi, Uji, Vij, nj, wla);
            }
        }
        STAMP(5);

        // ---- fast continua: Gamma integrand from the ray sums (one lane per wavelength) ----
        if (nF > 0) {
            const double sI = Jsum * (4.0 * kPi);       // sum_mu (w_mu/2 4pi) I
            for (int f = 0; f < nF; ++f) {
                const SlotS sl = load_slot(fslots + f, Ns);
                const int fl = sl.flags;
                const double ni = n_col[sl.noff_i + k];
                const double nj = n_col[sl.noff_j + k];
                const bool a = (fact >> f) & 1u;
                const int l = a ? la - sl.Nblue : 0;
                const double g = a ? gijc_col[sl.base + k * sl.len + (la - sl.first)] : 0.0;
                const double alf = a ? p.alpha[sl.wl_off + l] : 0.0;
                const double wla = a ? p.wl[sl.wl_off + l] : 0.0;
                const double Vji = g * alf;
                const double Uji = u_la * Vji;
                const double chi = ni * alf - nj * Vji;
                const double eta = nj * Uji;
                const double etaA = (fl & SLOT_ETA_CELL) ? CETA(sl.ca) : eta;
                const double chi_i = (fl & SLOT_LI_CELL) ? CCHI(sl.ci) : chi;
                const double chi_j = (fl & SLOT_LJ_CELL) ? CCHI(sl.cj) : -chi;
                const double U_j = (fl & SLOT_LJ_CELL) ? CU(sl.cj) : Uji;
                const double U_i = (fl & SLOT_UI_READ) ? CU(sl.ci) : 0.0;
                const double sIe = sI - etaA * sPsi;                       // sum_mu w (I - Psi eta)
                const double g1 = (Uji * sW + Vji * sIe) - (chi_i * U_j) * sPsi;
                const double g2 = (alf * sIe) - (chi_j * U_i) * sPsi;
                const double wt = (a && lead) ? wla : 0.0;
                const double r1 = wave_sum(wt * g1);
                const double r2 = wave_sum(wt * g2);
                if (lane == LSX_WAVE - 1) {
                    double* gp = gpart + ((nP + f) * 4 + dir) * Ns + k;
                    gp[0] = r1;
                    gp[2 * Ns] = r2;
                }
            }
        }
        STAMP(6);

        // ---- J: the two directions meet at depth k at different steps ----
        const int s2 = 2 * s, nm1 = Ns - 1;
        if (s2 < nm1) {
            if (lead) Jnew[kl] = Jsum;                        // first visitor stores its half
        } else if (s2 == nm1) {                               // odd Nspace: both waves are at the same k
            if (lead) xwg[dir * LSX_WAVE + j] = Jsum;
            __syncthreads();
            if (lead && dir == 0) {
                const double Jv = Jsum + xwg[LSX_WAVE + j];
                Jnew[kl] = Jv;
                dJ = nanmax(dJ, fabs(1.0 - jd / Jv));         // :705
            }
        } else {
            if (s2 == nm1 + 1 || s2 == nm1 + 2) __syncthreads(); // the partner wave's first-half stores
            if (lead) {
                const double Jv = Jnew[kl] + Jsum;
                Jnew[kl] = Jv;
                dJ = nanmax(dJ, fabs(1.0 - jd / Jv));         // :705
            }
        }
    }

#ifdef LSX_STAMPS
    STAMP(7);
    if (lane == 0 && dir == 0 && p.debug && (vb % 997) == 5 && vb / 997 < 64) {
        unsigned long long* D = (unsigned long long*)p.debug + (size_t)(vb / 997) * 16;
        for (int i = 0; i < 8; ++i) D[i] = T[i];
        D[8] = tile_id; D[9] = nP; D[10] = nF; D[11] = col;
    }
#endif
#undef STAMP
    const double dJw = wave_max_nan(lead ? dJ : 0.0);
    if (lane == 0) p.dJpart[((size_t)col * ntile + tile_id) * 2 + dir] = dJw;
#undef CCHI
#undef CU
#undef CETA
}

__global__ void __launch_bounds__(2 * LSX_WAVE) __attribute__((amdgpu_waves_per_eu(LSX_WAVES_PER_EU)))
lsx_sweep_kernel(const SweepParams p)
{
    int vb;
    {
        const int nb = gridDim.x, x = blockIdx.x & 7, q = blockIdx.x >> 3;
        const int nb8 = nb >> 3, rem = nb & 7;
        vb = x * nb8 + (x < rem ? x : rem) + q;
    }
    const int tile_id = vb % p.ntile_total;
    if (p.colmask && LSX_CONST(uint8_t, p.colmask)[vb / p.ntile_total] == 0) {
        // frozen column: nothing is computed; J only moves to the other half of the ping-pong pair
        const size_t tb = (size_t)vb * p.Nspace * p.L;    // vb = col * ntile + tile
        for (int e = threadIdx.x; e < p.Nspace * p.L; e += 2 * LSX_WAVE) p.Jnew_T[tb + e] = p.Jdag_T[tb + e];
        return;
    }
    const int nP = (LSX_CONST(DevTile, p.tiles) + tile_id)->nP;
#ifndef LSX_NO_SPECIALIZE
    if (nP == 0) sweep_tile<0>(p, vb);
    else if (nP == 1) sweep_tile<1>(p, vb);
    else if (nP == 2) sweep_tile<2>(p, vb);
    else if (nP == 3) sweep_tile<3>(p, vb);
    else
#endif
        sweep_tile<-1>(p, vb);
}

extern "C" hipError_t lsx_launch_sweep(const SweepParams* p, int nblocks, size_t lds_bytes, hipStream_t st)
{
    hipLaunchKernelGGL(lsx_sweep_kernel, dim3(nblocks), dim3(2 * LSX_WAVE), lds_bytes, st, *p);
    return hipGetLastError();
}
