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
//   * one wavefront = one TILE = 32 consecutive wavelengths x 2 directions of ONE column.
//     lanes 0-31 walk down (k = 0..N-1), lanes 32-63 walk up (k = N-1..0): the depth
//     recurrence stays serial inside a lane, both halves advance in lock step.
//   * the Nrays mu-angles of a wavelength are advanced INSIDE the lane (unrolled, M
//     independent recurrences = ILP; the angle quadrature for J and Gamma is an in-register
//     sum, no cross-lane traffic).
//   * the wavelength quadrature for Gamma is one half-wave reduction per (transition, depth);
//     lane 31 / 63 store the result into a per-(tile, transition, direction, depth) slab.
//     Every slab element is written exactly once -> no atomics, bitwise reproducible, and the
//     result does not depend on how columns are distributed over GPUs.
//   * all per-(lambda, depth) inputs are depth-major in HBM, so at a fixed depth the 32
//     lanes of a half-wave read 256 consecutive bytes; every input byte is read once.
//   * the per-level "effective opacity"/U bookkeeping of overlapping transitions
//     (atom.chi / atom.U / atom.eta, rh_method.py:616-627) lives in lane-private LDS columns
//     (address = level*64 + lane: conflict free, no barriers -- a workgroup is one wave).
//
// Template: UMAX = max transitions overlapping in the tile (register arrays are statically
// indexed by unrolling to UMAX with a wave-uniform guard), M = rays per lane (Nrays padded
// with zero-weight rays).
#include <hip/hip_runtime.h>
#include "lsx_dev.h"

#ifndef LSX_WAVES_PER_EU
#define LSX_WAVES_PER_EU
#endif

namespace {

// constants.py:1-27
constexpr double kCLight = 2.99792458E+08;
constexpr double kHPlanck = 6.6260755E-34;
constexpr double kKBoltzmann = 1.380658E-23;
constexpr double kNM_TO_M = 1.0E-09;
constexpr double kHC = kHPlanck * kCLight;
constexpr double kPi = 3.14159265358979323846;

// formal_solver.py:14-44, branch free: all three forms are evaluated and selected, so a
// wavefront never diverges on the optical-depth regime.
__device__ __forceinline__ void w2(double dtau, double& w0, double& w1)
{
    const double e = exp(-dtau);
    const double a0 = 1.0 - e;
    const double a1 = a0 - dtau * e;
    const double t0 = dtau * (1.0 - 0.5 * dtau);
    const double t1 = (dtau * dtau) * (0.5 - dtau / 3.0);
    const bool small = dtau < 5e-4;
    const bool large = dtau > 50.0;
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

// sum over the 32 lanes of each half-wave (xor 1..16 never crosses bit 5); all lanes get it
__device__ __forceinline__ double half_sum(double v)
{
#pragma unroll
    for (int m = 1; m < LSX_HALF; m <<= 1) v += __shfl_xor(v, m, LSX_WAVE);
    return v;
}

__device__ __forceinline__ double wave_max_nan(double v)
{
    // max that propagates NaN like numpy's ndarray.max (rh_method.py:706)
#pragma unroll
    for (int m = 1; m < LSX_WAVE; m <<= 1) {
        const double o = __shfl_xor(v, m, LSX_WAVE);
        v = (v != v || o != o) ? __builtin_nan("") : fmax(v, o);
    }
    return v;
}

__device__ __forceinline__ void lds_add(double* p, double v)
{
    // lane-private location: a plain DS add (no return) is enough, nobody else touches it
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

} // namespace

template <int UMAX, int M>
__global__ void __launch_bounds__(LSX_WAVE) LSX_WAVES_PER_EU lsx_sweep_kernel(const SweepParams p)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int hl = lane & (LSX_HALF - 1);
    const int dir = lane >> 5; // 0: down (toFrom False), 1: up (toFrom True)
    const int col = blockIdx.x / p.n_class_tiles;
    const int tile_id = p.class_tiles[blockIdx.x - col * p.n_class_tiles];
    const DevTile tile = p.tiles[tile_id];
    const int U = tile.nslot;
    const int Ns = p.Nspace;
    const int Nspect = p.Nspect;
    const int Nrays = p.Nrays;
    const bool valid = hl < tile.nla;
    const int la = tile.la0 + (valid ? hl : 0);

    // lane-private LDS columns: lchi[level], lU[level], leta[atom]
    double* const lchi = lds + lane;
    double* const lU = lds + p.NLtot * LSX_WAVE + lane;
    double* const leta = lds + 2 * p.NLtot * LSX_WAVE + lane;

    // column bases (wave-uniform 64-bit), everything below is indexed with 32-bit offsets
    const double* __restrict__ n_col = p.n + (size_t)col * p.NLtot * Ns;
    const double* __restrict__ wphi_col = p.wphi + (size_t)col * p.Nlines * Ns;
    const double* __restrict__ z = p.height + (size_t)col * Ns;
    const double* __restrict__ bgchi = p.bgchi_T + (size_t)col * Ns * Nspect;
    const double* __restrict__ bgeta = p.bgeta_T + (size_t)col * Ns * Nspect;
    const double* __restrict__ Jdag = p.Jdag_T + (size_t)col * Ns * Nspect;
    double* __restrict__ Jnew = p.Jnew_T + (size_t)col * Ns * Nspect;
    const double* __restrict__ sca = p.sca + (size_t)col * (p.sca_per_lambda ? (size_t)Ns * Nspect : (size_t)Ns);
    const double* __restrict__ phi_col = p.phi_T + (size_t)col * p.phi_col_stride;
    const double* __restrict__ gijc_col = p.gijc_T + (size_t)col * p.gijc_col_stride;
    double* __restrict__ gpart = p.Gpart + ((size_t)col * p.nslot_total + tile.slot0) * 4 * Ns;
    const int32_t* __restrict__ slots = p.tile_slots + tile.slot0;
    const int32_t* __restrict__ tlev = p.tile_levels + tile.lev0;

    const double wav = p.wavelength[la];
    const double u_la = p.u_la[la];
    const bool compact = p.phi_mu_stride_is_zero != 0;

    // per-slot lane state: local wavelength index, -1 when the transition is not active here
    int lt[UMAX];
#pragma unroll
    for (int u = 0; u < UMAX; ++u) {
        lt[u] = -1;
        if (u < U) {
            const int t = slots[u];
            const DevTrans& tr = p.trans[t];
            const int l = la - tr.Nblue;
            const bool a = valid && l >= 0 && l < tr.Nlam && p.active[t * Nspect + la] != 0;
            lt[u] = a ? l : -1;
        }
    }

    // element offset of phi(line, depth kk, this lane's direction, ray m, this lane's wavelength)
    auto phi_at = [&](const DevTrans& tr, int kk, int m, int l) -> double {
        const int idx = compact ? tr.phi_off * Ns + kk * tr.Nlam + l
                                : tr.phi_off * 2 * Nrays * Ns + ((kk * 2 + dir) * Nrays + m) * tr.Nlam + l;
        return phi_col[idx];
    };

    // total opacity at depth kk for every ray (needed one depth ahead for the lower boundary
    // condition only, formal_solver.py:204-207)
    auto chi_only = [&](int kk, double (&chi)[M]) {
        const double bc = valid ? bgchi[kk * Nspect + la] : 1.0;
#pragma unroll
        for (int m = 0; m < M; ++m) chi[m] = 0.0;
#pragma unroll
        for (int u = 0; u < UMAX; ++u) {
            if (u < U) {
                const DevTrans& tr = p.trans[slots[u]];
                const double ni = n_col[tr.li * Ns + kk];
                const double nj = n_col[tr.lj * Ns + kk];
                if (tr.is_line) {
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        double phv = 0.0;
                        if (lt[u] >= 0 && m < Nrays) phv = phi_at(tr, kk, m, lt[u]);
                        const double Vij = tr.cB * phv;
                        const double Vji = tr.gij * Vij;
                        chi[m] += ni * Vij - nj * Vji;
                    }
                } else {
                    double g = 0.0, al = 0.0;
                    if (lt[u] >= 0) {
                        g = gijc_col[tr.cont_off * Ns + kk * tr.Nlam + lt[u]];
                        al = p.alpha[tr.wl_off + lt[u]];
                    }
                    const double c = ni * al - nj * (g * al);
#pragma unroll
                    for (int m = 0; m < M; ++m) chi[m] += c;
                }
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) chi[m] += bc;
    };

    // ---- boundary conditions: formal_solver.py:203-209 -------------------------------
    double Iu[M];        // upwind intensity
    {
        const int kS = dir ? Ns - 1 : 0;
        const int dk = dir ? -1 : 1;
        double c0[M], c1[M];
        chi_only(kS, c0);
        chi_only(kS + dk, c1);
        const double dz = fabs(z[kS] - z[kS + dk]);
        const double B0 = planck(p.temperature[(size_t)col * Ns + Ns - 2], wav);
        const double B1 = planck(p.temperature[(size_t)col * Ns + Ns - 1], wav);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const double dtau_uw = p.zmu[m] * (c0[m] + c1[m]) * 0.5 * dz;
            Iu[m] = dir ? (B1 - (B0 - B1) / dtau_uw) : 0.0;
        }
    }

    double chi_prev[M], S_prev[M], dtau_prev[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { chi_prev[m] = 1.0; S_prev[m] = 0.0; dtau_prev[m] = 1.0; }

    double dJ = 0.0;

    for (int s = 0; s < Ns; ++s) {
        const int k = dir ? Ns - 1 - s : s;
        const int kl = k * Nspect + la;
        const double bc = valid ? bgchi[kl] : 1.0;
        const double be = valid ? bgeta[kl] : 0.0;
        const double jd = valid ? Jdag[kl] : 0.0;
        const double sc = sca[p.sca_per_lambda ? kl : k];
        const double scaJ = valid ? sc * jd : 0.0;
        const double dz = (s > 0) ? fabs(z[dir ? k + 1 : k - 1] - z[k]) : 0.0;

        // per-slot depth values shared by all rays of the lane
        //   line:      a0 = n_i, a1 = n_j
        //   continuum: a0 = Vij (= alpha), a1 = Vji, a2 = chi, a3 = eta   (ray independent)
        double a0[UMAX], a1[UMAX], a2[UMAX], a3[UMAX];
#pragma unroll
        for (int u = 0; u < UMAX; ++u) {
            a0[u] = a1[u] = a2[u] = a3[u] = 0.0;
            if (u < U) {
                const DevTrans& tr = p.trans[slots[u]];
                const double ni = n_col[tr.li * Ns + k];
                const double nj = n_col[tr.lj * Ns + k];
                if (tr.is_line) {
                    a0[u] = ni;
                    a1[u] = nj;
                } else if (lt[u] >= 0) {
                    const double g = gijc_col[tr.cont_off * Ns + k * tr.Nlam + lt[u]];
                    const double al = p.alpha[tr.wl_off + lt[u]];
                    const double Vji = g * al;              // rh_method.py:284-285
                    a0[u] = al;
                    a1[u] = Vji;
                    a2[u] = ni * al - nj * Vji;             // :613
                    a3[u] = nj * (u_la * Vji);              // :614, :286
                }
            }
        }

        double acc1[UMAX], acc2[UMAX];
#pragma unroll
        for (int u = 0; u < UMAX; ++u) acc1[u] = acc2[u] = 0.0;
        double Jsum = 0.0;

#pragma unroll
        for (int m = 0; m < M; ++m) {
            // ---- pass 1: uv + opacity/emissivity (rh_method.py:601-627) ----
            for (int q = 0; q < tile.nlev; ++q) {
                const int l = tlev[q];
                lchi[l * LSX_WAVE] = 0.0;
                lU[l * LSX_WAVE] = 0.0;
            }
            for (int a = 0; a < p.Natoms; ++a) leta[a * LSX_WAVE] = 0.0;

            double Vij[UMAX], Vji[UMAX];
            double chiTot = 0.0, etaTot = 0.0;
#pragma unroll
            for (int u = 0; u < UMAX; ++u) {
                Vij[u] = Vji[u] = 0.0;
                if (u < U) {
                    const DevTrans& tr = p.trans[slots[u]];
                    double chi, eta, Uji;
                    if (tr.is_line) {
                        double phv = 0.0;
                        if (lt[u] >= 0 && m < Nrays) phv = phi_at(tr, k, m, lt[u]);
                        Vij[u] = tr.cB * phv;               // :279
                        Vji[u] = tr.gij * Vij[u];           // :280
                        Uji = tr.AB * Vji[u];               // :281
                        chi = a0[u] * Vij[u] - a1[u] * Vji[u];
                        eta = a1[u] * Uji;
                    } else {
                        Vij[u] = a0[u];
                        Vji[u] = a1[u];
                        Uji = u_la * Vji[u];
                        chi = a2[u];
                        eta = a3[u];
                    }
                    lds_add(&lchi[tr.li * LSX_WAVE], chi);   // :619
                    lds_add(&lchi[tr.lj * LSX_WAVE], -chi);  // :620
                    lds_add(&lU[tr.lj * LSX_WAVE], Uji);     // :622
                    lds_add(&leta[tr.atom * LSX_WAVE], eta); // :627
                    chiTot += chi;
                    etaTot += eta;
                }
            }
            chiTot += bc;                                   // :630
            const double S = (etaTot + be + scaJ) / chiTot; // :632

            // ---- formal solution at this depth (formal_solver.py:107-139) ----
            double I, Lam;
            if (s == 0) {
                I = Iu[m];
                Lam = 0.0;
            } else {
                const double dtau = 0.5 * (chi_prev[m] + chiTot) * p.zmu[m] * dz;
                const double dS = (S_prev[m] - S) / dtau;
                // formal_solver.py:138-139: the end point re-uses the PREVIOUS interval's w and
                // S[kEnd - dk] with the fresh dS, dtau (reference behaviour, reproduced deliberately)
                const bool last = (s == Ns - 1);
                double w0, w1;
                w2(last ? dtau_prev[m] : dtau, w0, w1);
                I = Iu[m] * (1.0 - w0) + w0 * (last ? S_prev[m] : S) + w1 * dS;
                Lam = w0 - w1 / dtau;
                dtau_prev[m] = dtau;
            }
            const double Psi = Lam / chiTot;
            Iu[m] = I;
            chi_prev[m] = chiTot;
            S_prev[m] = S;
            Jsum += p.wmuh[m] * I;                          // :640
            if (s == Ns - 1 && dir == 1 && valid && m < Nrays)  // emergent intensity, :638
                p.Iout[((size_t)col * Nspect + la) * Nrays + m] = I;

            // ---- pass 2: Gamma integrands (rh_method.py:643-681) ----
            const double wq = p.wmuh[m] * 4.0 * kPi;
#pragma unroll
            for (int u = 0; u < UMAX; ++u) {
                if (u < U) {
                    const DevTrans& tr = p.trans[slots[u]];
                    const double Ieff = I - Psi * leta[tr.atom * LSX_WAVE];
                    const double chi_i = lchi[tr.li * LSX_WAVE];
                    const double chi_j = lchi[tr.lj * LSX_WAVE];
                    const double U_i = lU[tr.li * LSX_WAVE];
                    const double U_j = lU[tr.lj * LSX_WAVE];
                    const double Uji = (tr.is_line ? tr.AB : u_la) * Vji[u];
                    const double g1 = (Uji + Vji[u] * Ieff) - (chi_i * Psi * U_j);
                    const double g2 = (Vij[u] * Ieff) - (chi_j * Psi * U_i);
                    acc1[u] += wq * g1;
                    acc2[u] += wq * g2;
                }
            }
        }

        // ---- wavelength quadrature: one half-wave reduction per (slot, entry) ----
#pragma unroll
        for (int u = 0; u < UMAX; ++u) {
            if (u < U) {
                const DevTrans& tr = p.trans[slots[u]];
                double wla = 0.0;
                if (lt[u] >= 0) {
                    const double wl = p.wl[tr.wl_off + lt[u]];
                    wla = tr.is_line ? wl * wphi_col[tr.line_idx * Ns + k] / kHC : wl; // :451,455
                }
                const double v1 = (lt[u] >= 0) ? acc1[u] * wla : 0.0;
                const double v2 = (lt[u] >= 0) ? acc2[u] * wla : 0.0;
                const double r1 = half_sum(v1);
                const double r2 = half_sum(v2);
                if (hl == LSX_HALF - 1) {
                    double* g = gpart + (u * 4 + dir) * Ns + k; // [slot][e][dir][k]
                    g[0] = r1;          // e = 0: Gamma[i][j]
                    g[2 * Ns] = r2;     // e = 1: Gamma[j][i]
                }
            }
        }

        // ---- J: the two directions meet at depth k at different steps ----
        const int s2 = 2 * s, nm1 = Ns - 1;
        if (s2 < nm1) {
            if (valid) Jnew[kl] = Jsum;                       // first visitor stores its half
        } else {
            double Jv;
            if (s2 == nm1) {                                  // odd Nspace: both halves are at the same k
                Jv = Jsum + __shfl_xor(Jsum, LSX_HALF, LSX_WAVE);
            } else {
                if (s2 == nm1 + 1 || s2 == nm1 + 2)
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); // partner half's stores
                Jv = (valid ? Jnew[kl] : 1.0) + Jsum;
            }
            if (valid && (s2 > nm1 || dir == 0)) {
                Jnew[kl] = Jv;
                const double d = fabs(1.0 - jd / Jv);         // :705
                dJ = (d != d || dJ != dJ) ? __builtin_nan("") : fmax(dJ, d);
            }
        }
    }

    const double dJw = wave_max_nan(dJ);
    if (lane == 0) p.dJpart[(size_t)col * p.ntile_total + tile_id] = dJw;
}

// ---------------------------------------------------------------------------------------
// explicit instantiations + C launchers (one translation unit per M keeps builds parallel)
#ifndef LSX_M
#error "compile with -DLSX_M=<rays per lane>"
#endif

template <int UMAX>
static hipError_t launch_u(const SweepParams& p, int nblocks, size_t lds_bytes, hipStream_t st)
{
    hipLaunchKernelGGL((lsx_sweep_kernel<UMAX, LSX_M>), dim3(nblocks), dim3(LSX_WAVE), lds_bytes, st, p);
    return hipGetLastError();
}

#define LSX_CAT2(a, b) a##b
#define LSX_CAT(a, b) LSX_CAT2(a, b)

extern "C" hipError_t LSX_CAT(lsx_launch_sweep_m, LSX_M)(const SweepParams* p, int umax, int nblocks,
                                                          size_t lds_bytes, hipStream_t st)
{
    switch (umax) {
    case 2: return launch_u<2>(*p, nblocks, lds_bytes, st);
    case 4: return launch_u<4>(*p, nblocks, lds_bytes, st);
    case 8: return launch_u<8>(*p, nblocks, lds_bytes, st);
    case 12: return launch_u<12>(*p, nblocks, lds_bytes, st);
    default: return hipErrorInvalidValue;
    }
}
