// lsx_setup.hip -- the set-up chain that turns an atmosphere into hot-path inputs, on the device:
//   v_broad            atomic_model.py:66-69
//   damping            atomic_model.py:491-502  = radiative + Unsold van der Waals (:166-198) + Stark (:300-345)
//   lte_pops           atomic_set.py:105-145    (Saha-Boltzmann with Debye lowering)
//   compute_collisions rh_method.py:474-487 over collisional_rates.py:21-96 (Omega, CI, CE) with the not-a-knot cubic
//                      in temperature that scipy.interpolate.interp1d(kind=3) builds (collisional_rates.py:15-19)
// Everything is pointwise in (column, depth): one thread each.  The temperature-independent constants of every line and
// collision are worked out once on the host (lsx_set_atomic_data).  gfx950 only.
#include <algorithm>
#include <cmath>

#include "lsx_ctx.h"

using namespace lsxd;

namespace {

// constants.py:1-27
constexpr double kCLight = 2.99792458E+08, kHPlanck = 6.6260755E-34, kKBoltzmann = 1.380658E-23, kAmu = 1.6605402E-27,
                 kMElectron = 9.1093897E-31, kQElectron = 1.60217733E-19, kEpsilon0 = 8.854187817E-12, kRBohr = 5.29177349E-11,
                 kERydberg = 2.1798741E-18, kABarH = 7.42E-41, kCM_TO_M = 1.0E-02;
constexpr double kHC = kHPlanck * kCLight;

struct LineDev {            // temperature-independent constants of one line's damping (atomic_model.py:491-502)
    double gRad, cDop;      // cDop = lambda0_m / (4 pi)
    double vdw_cross;       // Unsold: Q_vdW = cross T^0.3 nH_ground (:197-198), 0 = none
    double stark_c23, stark_C, stark_Cm;    // quadratic Stark: c23 ((C T)^(1/6) Cm) ne (:318-338)
    double stark_lin;       // stark < 0: |stark| ne (:339-340)
    double hlin;            // hydrogen: linear Stark a1 0.6 (nu^2 - nl^2) cm^2 ne^(2/3) (:307-316, 343-344)
    int32_t atom, pad;
};

struct CollDev {            // one collisional process (collisional_rates.py)
    int32_t kind, i, j, nT; // levels inside the atom
    int32_t off, atom;      // offset of its (x, y, M) tables
    double a;               // Omega: C0 / g_j (:43); CI: dE / k (:68); CE: g_i / g_j (:94)
};

struct AtomDev {
    int32_t Nl, lev_off, lev2_off, coll0, ncoll, pad;
    double vtherm;          // 2 k / (amu A), :67
};

struct SetupParams {
    int Ns, Natoms, Nlines, NLtot, NL2tot, ncol;
    const AtomDev* atoms;
    const LineDev* lines;
    const CollDev* colls;
    const double* spl;          // per collision: x[nT], y[nT], M[nT] (second derivatives of the not-a-knot cubic)
    const double* lev_E;        // [NLtot] E_SI
    const double* lev_g;        // [NLtot]
    const int32_t* lev_dZ;      // [NLtot] stage - stage of the atom's level 0
    const double* lev_nDebye;   // [NLtot]
    const double* T;            // [ncol][Ns] (chunk-local pointers below)
    const double* ne;
    const double* vturb;
    const double* nHG;
    const double* nTotal;       // [ncol][Natoms][Ns]
    double* vBroad;             // [ncol][Natoms][Ns]
    double* aDamp;              // [ncol][Nlines][Ns]
    double* nStar;              // [ncol][NLtot][Ns]
    double* n;
    double* C;                  // [ncol][NL2tot][Ns]
    int lte_pops;
};

// atomic_model.py:66-69 and :491-502
__global__ void k_setup_broadening(const SetupParams q)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)q.ncol * q.Ns) return;
    const size_t col = gid / q.Ns;
    const int k = gid % q.Ns;
    const double T = q.T[gid], ne = q.ne[gid], vt = q.vturb[gid], nH = q.nHG[gid];
    for (int a = 0; a < q.Natoms; ++a)
        q.vBroad[(col * q.Natoms + a) * q.Ns + k] = sqrt(q.atoms[a].vtherm * T + vt * vt);
    const double T03 = pow(T, 0.3), ne23 = pow(ne, 2.0 / 3.0);
    for (int l = 0; l < q.Nlines; ++l) {
        const LineDev L = q.lines[l];
        double Qelast = 0.0;
        if (L.vdw_cross != 0.0) Qelast += L.vdw_cross * T03 * nH;                            // :197-198
        double stark = 0.0;
        if (L.stark_c23 != 0.0) stark = L.stark_c23 * (pow(L.stark_C * T, 1.0 / 6.0) * L.stark_Cm) * ne;   // :337-338
        else if (L.stark_lin != 0.0) stark = L.stark_lin * ne;                                // :340
        if (L.hlin != 0.0) stark += L.hlin * ne23;                                            // :314-315, :344
        Qelast += stark;
        q.aDamp[(col * q.Nlines + l) * q.Ns + k] = (L.gRad + Qelast) * L.cDop / q.vBroad[(col * q.Natoms + L.atom) * q.Ns + k];
    }
}

// atomic_set.py:105-145 (debye=True)
__global__ void k_setup_lte_pops(const SetupParams q)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)q.ncol * q.Ns) return;
    const size_t col = gid / q.Ns;
    const int k = gid % q.Ns;
    const double T = q.T[gid], ne = q.ne[gid];
    const double c1 = (kHPlanck / (2.0 * M_PI * kMElectron)) * (kHPlanck / kKBoltzmann);
    const double c2 = sqrt(8.0 * M_PI / kKBoltzmann) * pow(kQElectron * kQElectron / (4.0 * M_PI * kEpsilon0), 1.5);
    const double dEion = c2 * sqrt(ne / T);
    const double cNe_T = 0.5 * ne * pow(c1 / T, 1.5);
    for (int a = 0; a < q.Natoms; ++a) {
        const AtomDev A = q.atoms[a];
        double* ns = q.nStar + (col * q.NLtot + A.lev_off) * q.Ns + k;
        double total = 1.0;
        for (int i = 1; i < A.Nl; ++i) {
            const int gi = A.lev_off + i;
            const double dE = q.lev_E[gi] - q.lev_E[A.lev_off];
            const double gi0 = q.lev_g[gi] / q.lev_g[A.lev_off];
            const int dZ = q.lev_dZ[gi];
            const double dE_kT = (dE - q.lev_nDebye[gi] * dEion) / (kKBoltzmann * T);
            double nst = gi0 * exp(-dE_kT);
            // cNe_T ** dZ with an integer exponent (numpy: 0 -> 1, 1 -> x, 2 -> x x, else pow)
            const double den = dZ == 0 ? 1.0 : (dZ == 1 ? cNe_T : (dZ == 2 ? cNe_T * cNe_T : pow(cNe_T, (double)dZ)));
            nst /= den;
            ns[(size_t)i * q.Ns] = nst;
            total += nst;
        }
        const double n0 = q.nTotal[(col * q.Natoms + a) * q.Ns + k] / total;
        ns[0] = n0;
        for (int i = 1; i < A.Nl; ++i) ns[(size_t)i * q.Ns] *= n0;
        double* nn = q.n + (col * q.NLtot + A.lev_off) * q.Ns + k;          // n starts as a copy of nStar (rh_method.py:414-416)
        for (int i = 0; i < A.Nl; ++i) nn[(size_t)i * q.Ns] = ns[(size_t)i * q.Ns];
    }
}

// the interpolant scipy.interpolate.interp1d(x, y, kind=3, fill_value=(y[0], y[-1]), bounds_error=False) evaluates
// (not-a-knot cubic; linear for two points): tab = x[n], y[n], M[n]
__device__ double spline_eval(const double* tab, int n, double t)
{
    const double *x = tab, *y = tab + n, *M = tab + 2 * n;
    if (t < x[0]) return y[0];
    if (t > x[n - 1]) return y[n - 1];
    int i = 0;
    while (i < n - 2 && t > x[i + 1]) ++i;
    const double h = x[i + 1] - x[i], a = x[i + 1] - t, b = t - x[i];
    return (M[i] * a * a * a + M[i + 1] * b * b * b) / (6.0 * h) + (y[i] / h - M[i] * h / 6.0) * a + (y[i + 1] / h - M[i + 1] * h / 6.0) * b;
}

// rh_method.py:474-487 over collisional_rates.py:21-96
__global__ void k_setup_collisions(const SetupParams q)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)q.ncol * q.Ns) return;
    const size_t col = gid / q.Ns;
    const int k = gid % q.Ns;
    const double T = q.T[gid], ne = q.ne[gid], sqrtT = sqrt(T);
    for (int a = 0; a < q.Natoms; ++a) {
        const AtomDev A = q.atoms[a];
        double* Cm = q.C + (col * q.NL2tot + A.lev2_off) * q.Ns + k;           // C[to][from][k]
        const double* ns = q.nStar + (col * q.NLtot + A.lev_off) * q.Ns + k;
        for (int e = 0; e < A.Nl * A.Nl; ++e) Cm[(size_t)e * q.Ns] = 0.0;
        for (int c = A.coll0; c < A.coll0 + A.ncoll; ++c) {
            const CollDev K = q.colls[c];
            const double Cv = spline_eval(q.spl + K.off, K.nT, T);
            const double nsi = ns[(size_t)K.i * q.Ns], nsj = ns[(size_t)K.j * q.Ns];
            double* Cij = Cm + (size_t)(K.i * A.Nl + K.j) * q.Ns;
            double* Cji = Cm + (size_t)(K.j * A.Nl + K.i) * q.Ns;
            if (K.kind == LSX_COLL_OMEGA) {
                const double Cdown = K.a * ne * Cv / sqrtT;                      // C0 ne Omega / (g_j sqrt T), :43
                *Cij += Cdown;
                *Cji += Cdown * nsj / nsi;                                       // :45
            } else if (K.kind == LSX_COLL_CI) {
                const double Cup = Cv * ne * exp(-K.a / T) * sqrtT;              // :68
                *Cji += Cup;
                *Cij += Cup * nsi / nsj;                                         // :70
            } else {
                const double Cdown = Cv * ne * K.a * sqrtT;                      // :94
                *Cij += Cdown;
                *Cji += Cdown * nsj / nsi;                                       // :96
            }
        }
        for (int e = 0; e < A.Nl * A.Nl; ++e)
            if (Cm[(size_t)e * q.Ns] < 0.0) Cm[(size_t)e * q.Ns] = 0.0;        // rh_method.py:487
    }
}

// second derivatives of the not-a-knot cubic through (x, y): the spline make_interp_spline(x, y, k=3) represents
bool notaknot_moments(const std::vector<double>& x, const std::vector<double>& y, std::vector<double>& M)
{
    const int n = (int)x.size();
    M.assign(n, 0.0);
    if (n == 2) return true;                    // linear
    if (n < 4) return false;
    std::vector<double> A((size_t)n * n, 0.0), b(n, 0.0);
    auto h = [&](int i) { return x[i + 1] - x[i]; };
    // third derivative continuous at x[1] and x[n-2]
    A[0 * n + 0] = h(1); A[0 * n + 1] = -(h(0) + h(1)); A[0 * n + 2] = h(0);
    A[(size_t)(n - 1) * n + n - 3] = h(n - 2); A[(size_t)(n - 1) * n + n - 2] = -(h(n - 3) + h(n - 2)); A[(size_t)(n - 1) * n + n - 1] = h(n - 3);
    for (int i = 1; i < n - 1; ++i) {
        A[(size_t)i * n + i - 1] = h(i - 1);
        A[(size_t)i * n + i] = 2.0 * (h(i - 1) + h(i));
        A[(size_t)i * n + i + 1] = h(i);
        b[i] = 6.0 * ((y[i + 1] - y[i]) / h(i) - (y[i] - y[i - 1]) / h(i - 1));
    }
    for (int c = 0; c < n; ++c) {               // Gaussian elimination with partial pivoting
        int p = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(A[(size_t)r * n + c]) > std::fabs(A[(size_t)p * n + c])) p = r;
        if (A[(size_t)p * n + c] == 0.0) return false;
        if (p != c) {
            for (int q = 0; q < n; ++q) std::swap(A[(size_t)c * n + q], A[(size_t)p * n + q]);
            std::swap(b[c], b[p]);
        }
        for (int r = c + 1; r < n; ++r) {
            const double f = A[(size_t)r * n + c] / A[(size_t)c * n + c];
            if (f == 0.0) continue;
            for (int q = c; q < n; ++q) A[(size_t)r * n + q] -= f * A[(size_t)c * n + q];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; --r) {
        double s = b[r];
        for (int q = r + 1; q < n; ++q) s -= A[(size_t)r * n + q] * M[q];
        M[r] = s / A[(size_t)r * n + r];
    }
    return true;
}

} // namespace

extern "C" {

int lsx_set_atomic_data(lsx_ctx* c, const lsx_atomic_data* d)
{
    if (!c || !d || !d->atoms) return fail(LSX_EINVAL, "lsx_set_atomic_data: null argument");
    if (d->Natoms != c->Natoms) return fail(LSX_EINVAL, "lsx_set_atomic_data: %d atoms, the context has %d", d->Natoms, c->Natoms);
    HIPCHK(hipSetDevice(c->device));
    std::vector<AtomDev> atoms;
    std::vector<LineDev> lines;
    std::vector<CollDev> colls;
    std::vector<double> spl, lev_E, lev_g, lev_nD;
    std::vector<int32_t> lev_dZ;
    // the context's lines per atom, in table order
    std::vector<std::vector<int>> ctx_lines(c->Natoms);
    for (int t = 0; t < c->Ntrans; ++t)
        if (c->htrans[t].is_line) ctx_lines[c->htrans[t].atom].push_back(t);
    for (int a = 0; a < c->Natoms; ++a) {
        const lsx_atom_model& m = d->atoms[a];
        if (m.Nlevel != c->Nlevel[a] || !m.levels) return fail(LSX_EINVAL, "lsx_set_atomic_data: atom %d has %d levels, the context %d", a, m.Nlevel, c->Nlevel[a]);
        if (m.Nline != (int)ctx_lines[a].size()) return fail(LSX_EINVAL, "lsx_set_atomic_data: atom %d: %d lines given, %zu in the transition table", a, m.Nline, ctx_lines[a].size());
        if (!(m.weight > 0.0)) return fail(LSX_EINVAL, "lsx_set_atomic_data: atom %d: weight must be positive", a);
        AtomDev A{};
        A.Nl = m.Nlevel; A.lev_off = c->lev_off[a]; A.lev2_off = c->lev2_off[a];
        A.vtherm = 2.0 * kKBoltzmann / (kAmu * m.weight);                                      // atomic_model.py:67
        A.coll0 = (int)colls.size(); A.ncoll = m.Ncollision;
        for (int i = 0; i < m.Nlevel; ++i) {
            lev_E.push_back(m.levels[i].E_SI);
            lev_g.push_back(m.levels[i].g);
            lev_dZ.push_back(m.levels[i].stage - m.levels[0].stage);
            double nD = 0.0;                                                                  // atomic_set.py:113-119
            int Z = m.levels[i].stage;
            for (int s = 1; i >= 1 && s < m.levels[i].stage - m.levels[0].stage + 1; ++s) { nD += Z; Z += 1; }
            lev_nD.push_back(nD);
        }
        for (int q = 0; q < m.Nline; ++q) {
            const lsx_line_model& L = m.lines[q];
            const DevTrans& h = c->htrans[ctx_lines[a][q]];
            if (L.i < 0 || L.j >= m.Nlevel || L.i >= L.j || c->lev_off[a] + L.i != h.li || c->lev_off[a] + L.j != h.lj)
                return fail(LSX_EINVAL, "lsx_set_atomic_data: atom %d line %d (%d -> %d) is not line %d of the transition table", a, q, L.j, L.i, q);
            const lsx_level &lo = m.levels[L.i], &up = m.levels[L.j];
            LineDev D{};
            D.atom = a;
            D.gRad = L.gRad;
            D.cDop = (kHC / (up.E_SI - lo.E_SI)) / (4.0 * M_PI);                                // :499, :464-467
            if (L.vdw_kind == 1) {                                                             // VdwUnsold.setup, :166-194
                const int Z = up.stage + 1;
                int ic = L.j + 1;
                while (ic < m.Nlevel && m.levels[ic].stage < Z) ++ic;
                if (ic >= m.Nlevel) return fail(LSX_EINVAL, "lsx_set_atomic_data: atom %d line %d: no overlying continuum level for the van der Waals broadening", a, q);
                const double Ec = m.levels[ic].E_SI;
                const double r1 = kERydberg / (Ec - up.E_SI), r2 = kERydberg / (Ec - lo.E_SI);
                const double deltaR = r1 * r1 - r2 * r2;
                const double fourPiEps0 = 4.0 * M_PI * kEpsilon0;
                const double C625 = std::pow(2.5 * kQElectron * kQElectron / fourPiEps0 * kABarH / fourPiEps0 * 2.0 * M_PI *
                                                 (Z * kRBohr) * (Z * kRBohr) / kHPlanck * deltaR, 0.4);
                const double vRel35He = std::pow(8.0 * kKBoltzmann / (M_PI * kAmu * m.weight) * (1.0 + m.weight / d->weight_He), 0.3);
                const double vRel35H = std::pow(8.0 * kKBoltzmann / (M_PI * kAmu * m.weight) * (1.0 + m.weight / d->weight_H), 0.3);
                D.vdw_cross = 8.08 * (L.vdw[0] * vRel35H + L.vdw[1] * d->abundance_He * vRel35He) * C625;
            } else if (L.vdw_kind != 0) {
                return fail(LSX_EUNSUPPORTED, "lsx_set_atomic_data: van der Waals recipe %d is not implemented", L.vdw_kind);
            }
            if (L.stark > 0.0) {                                                               // :318-338
                D.stark_C = 8.0 * kKBoltzmann / (M_PI * kAmu * m.weight);
                D.stark_Cm = std::pow(1.0 + m.weight / (kMElectron / kAmu), 1.0 / 6.0) + std::pow(1.0 + m.weight / 28.0, 1.0 / 6.0);
                const int Z = lo.stage + 1;
                int ic = L.i + 1;
                while (ic < m.Nlevel && m.levels[ic].stage < Z) ++ic;
                if (ic >= m.Nlevel || m.levels[ic].stage == lo.stage)
                    return fail(LSX_EINVAL, "lsx_set_atomic_data: atom %d line %d: no overlying continuum level for the Stark broadening", a, q);
                const double E_Ryd = kERydberg / (1.0 + kMElectron / (m.weight * kAmu));
                const double neff_l = Z * std::sqrt(E_Ryd / (m.levels[ic].E_SI - lo.E_SI));
                const double neff_u = Z * std::sqrt(E_Ryd / (m.levels[ic].E_SI - up.E_SI));
                const double tu = neff_u * (5.0 * neff_u * neff_u + 1.0), tl = neff_l * (5.0 * neff_l * neff_l + 1.0);
                const double C4 = kQElectron * kQElectron / (4.0 * M_PI * kEpsilon0) * kRBohr * (2.0 * M_PI * kRBohr * kRBohr / kHPlanck) /
                                  (18.0 * (double)Z * Z * Z * Z) * (tu * tu - tl * tl);
                D.stark_c23 = 11.37 * std::pow(L.stark * C4, 2.0 / 3.0);
            } else if (L.stark < 0.0) {
                D.stark_lin = std::fabs(L.stark);
            }
            if (m.is_hydrogen) {                                                               // :307-316
                const int nUpper = (int)std::lround(std::sqrt(0.5 * up.g)), nLower = (int)std::lround(std::sqrt(0.5 * lo.g));
                const double a1 = (nUpper - nLower == 1) ? 0.642 : 1.0;
                D.hlin = a1 * 0.6 * (nUpper * nUpper - nLower * nLower) * kCM_TO_M * kCM_TO_M;
            }
            lines.push_back(D);
        }
        for (int q = 0; q < m.Ncollision; ++q) {
            const lsx_collision& K = m.collisions[q];
            if (K.i < 0 || K.j >= m.Nlevel || K.i >= K.j || K.nT < 2 || !K.temperature || !K.rates)
                return fail(LSX_EINVAL, "lsx_set_atomic_data: atom %d collision %d is inconsistent", a, q);
            CollDev D{};
            D.kind = K.kind; D.i = K.i; D.j = K.j; D.nT = K.nT; D.atom = a; D.off = (int)spl.size();
            const lsx_level &lo = m.levels[K.i], &up = m.levels[K.j];
            if (K.kind == LSX_COLL_OMEGA)
                D.a = (kERydberg / std::sqrt(kMElectron) * M_PI * kRBohr * kRBohr * std::sqrt(8.0 / (M_PI * kKBoltzmann))) / up.g;   // :38, :43
            else if (K.kind == LSX_COLL_CI) D.a = (up.E_SI - lo.E_SI) / kKBoltzmann;           // :62, :68
            else if (K.kind == LSX_COLL_CE) D.a = lo.g / up.g;                                  // :88
            else return fail(LSX_EUNSUPPORTED, "lsx_set_atomic_data: collision kind %d is not implemented", K.kind);
            std::vector<double> x(K.temperature, K.temperature + K.nT), y(K.rates, K.rates + K.nT), M;
            for (int e = 1; e < K.nT; ++e)
                if (!(x[e] > x[e - 1])) return fail(LSX_EINVAL, "lsx_set_atomic_data: atom %d collision %d: temperatures must ascend", a, q);
            if (!notaknot_moments(x, y, M))
                return fail(LSX_EINVAL, "lsx_set_atomic_data: atom %d collision %d: a cubic needs at least 4 temperatures (or 2 for a linear one)", a, q);
            spl.insert(spl.end(), x.begin(), x.end());
            spl.insert(spl.end(), y.begin(), y.end());
            spl.insert(spl.end(), M.begin(), M.end());
            colls.push_back(D);
        }
        atoms.push_back(A);
    }
    for (void* p : {(void*)c->d_sa_atoms, (void*)c->d_sa_lines, (void*)c->d_sa_colls, (void*)c->d_sa_spl, (void*)c->d_sa_levE,
                    (void*)c->d_sa_levg, (void*)c->d_sa_levdZ, (void*)c->d_sa_levnD})
        if (p) (void)hipFree(p);
    c->d_sa_atoms = c->d_sa_lines = c->d_sa_colls = nullptr;
    c->d_sa_spl = c->d_sa_levE = c->d_sa_levg = c->d_sa_levnD = nullptr;
    c->d_sa_levdZ = nullptr;
    int rc;
    std::vector<char> b_atoms((char*)atoms.data(), (char*)atoms.data() + atoms.size() * sizeof(AtomDev)),
        b_lines((char*)lines.data(), (char*)lines.data() + lines.size() * sizeof(LineDev)),
        b_colls((char*)colls.data(), (char*)colls.data() + colls.size() * sizeof(CollDev));
    if ((rc = upload(&c->d_sa_atoms, b_atoms, c->stream)) || (rc = upload(&c->d_sa_lines, b_lines, c->stream)) ||
        (rc = upload(&c->d_sa_colls, b_colls, c->stream)) || (rc = upload(&c->d_sa_spl, spl, c->stream)) ||
        (rc = upload(&c->d_sa_levE, lev_E, c->stream)) || (rc = upload(&c->d_sa_levg, lev_g, c->stream)) ||
        (rc = upload(&c->d_sa_levdZ, lev_dZ, c->stream)) || (rc = upload(&c->d_sa_levnD, lev_nD, c->stream)))
        return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->have_atomic_data = true;
    return LSX_OK;
}

int lsx_set_atmosphere(lsx_ctx* c, int32_t col0, int32_t ncol, const lsx_atmosphere* s)
{
    if (c) c->optab_fresh = false;
    if (!c || !s || col0 < 0 || ncol < 1 || col0 + ncol > c->ncol) return fail(LSX_EINVAL, "lsx_set_atmosphere: bad range");
    if (!c->have_atomic_data) return fail(LSX_EINVAL, "lsx_set_atmosphere: lsx_set_atomic_data has not been called");
    if (!s->temperature || !s->ne || !s->vturb || !s->nHGround || !s->nTotal) return fail(LSX_EINVAL, "lsx_set_atmosphere: null array pointer");
    if (c->phi_compact && s->vlos) return fail(LSX_EINVAL, "lsx_set_atmosphere: a phi_compact context takes vlos == NULL");
    HIPCHK(hipSetDevice(c->device));
    const int Ns = c->Nspace;
    const size_t nc = (size_t)c->ncol;
    int rc;
    // (either both exist or neither: a failed second allocation must not leave a half-made pair behind for the next call)
    if (!c->d_vBroad || !c->d_aDamp) {
        if (c->d_vBroad) { (void)hipFree(c->d_vBroad); c->d_vBroad = nullptr; }
        if (c->d_aDamp) { (void)hipFree(c->d_aDamp); c->d_aDamp = nullptr; }
        if ((rc = dmalloc(&c->d_vBroad, nc * c->Natoms * Ns))) return rc;
        if ((rc = dmalloc(&c->d_aDamp, nc * std::max(1, c->Nlines) * Ns))) { (void)hipFree(c->d_vBroad); c->d_vBroad = nullptr; return rc; }
    }
    // staging: ne, vturb, nHGround, vlos
    const size_t per = (size_t)4 * Ns;
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(ncol, ((size_t)32 << 20) / per));
    if ((rc = ensure_stage(c, chunk * per))) return rc;
    for (size_t b0 = 0; b0 < (size_t)ncol; b0 += chunk) {
        const size_t nb = std::min(chunk, (size_t)ncol - b0), cc = (size_t)col0 + b0;
        double *dNe = c->d_stage, *dVt = dNe + nb * Ns, *dNH = dVt + nb * Ns, *dVl = dNH + nb * Ns;
        HIPCHK(hipMemcpyAsync(c->d_temperature + cc * Ns, s->temperature + b0 * Ns, nb * Ns * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_nTotal + cc * c->Natoms * Ns, s->nTotal + b0 * c->Natoms * Ns, nb * c->Natoms * Ns * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(dNe, s->ne + b0 * Ns, nb * Ns * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(dVt, s->vturb + b0 * Ns, nb * Ns * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(dNH, s->nHGround + b0 * Ns, nb * Ns * 8, hipMemcpyHostToDevice, c->stream));
        if (s->vlos) HIPCHK(hipMemcpyAsync(dVl, s->vlos + b0 * Ns, nb * Ns * 8, hipMemcpyHostToDevice, c->stream));
        SetupParams q{};
        q.Ns = Ns; q.Natoms = c->Natoms; q.Nlines = c->Nlines; q.NLtot = c->NLtot; q.NL2tot = c->NL2tot; q.ncol = (int)nb;
        q.atoms = reinterpret_cast<const AtomDev*>(c->d_sa_atoms); q.lines = reinterpret_cast<const LineDev*>(c->d_sa_lines);
        q.colls = reinterpret_cast<const CollDev*>(c->d_sa_colls); q.spl = c->d_sa_spl; q.lev_E = c->d_sa_levE; q.lev_g = c->d_sa_levg;
        q.lev_dZ = c->d_sa_levdZ; q.lev_nDebye = c->d_sa_levnD;
        q.T = c->d_temperature + cc * Ns; q.ne = dNe; q.vturb = dVt; q.nHG = dNH; q.nTotal = c->d_nTotal + cc * c->Natoms * Ns;
        q.vBroad = c->d_vBroad + cc * c->Natoms * Ns; q.aDamp = c->d_aDamp + cc * std::max(1, c->Nlines) * Ns;
        q.nStar = c->d_nStar + cc * c->NLtot * Ns; q.n = c->d_n + cc * c->NLtot * Ns; q.C = c->d_C + cc * c->NL2tot * Ns;
        q.lte_pops = s->lte_pops;
        const long nth = (long)nb * Ns;
        const dim3 grid((unsigned)((nth + 127) / 128)), blk(128);
        hipLaunchKernelGGL(k_setup_broadening, grid, blk, 0, c->stream, q);
        if (s->lte_pops) hipLaunchKernelGGL(k_setup_lte_pops, grid, blk, 0, c->stream, q);
        hipLaunchKernelGGL(k_setup_collisions, grid, blk, 0, c->stream, q);
        HIPCHK(hipGetLastError());
        if (c->Nlines && (rc = profiles_from_device(c, cc, nb, q.aDamp, q.vBroad, s->vlos ? dVl : nullptr))) return rc;
        if ((rc = rebuild_derived(c, cc, nb))) return rc;        // g_ij factors follow (nStar, T)
        HIPCHK(hipStreamSynchronize(c->stream));                 // the staging buffer is re-used by the next sub-chunk
    }
    if (c->Nlines) mark_profiles_set(c, (size_t)col0, (size_t)ncol);
    return LSX_OK;
}

} // extern "C"
