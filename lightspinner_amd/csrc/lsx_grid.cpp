// lsx_grid.cpp -- wavelength grid, active set and the transitions' own grids: the host-side construction
// of what lsx_problem takes, from atomic_set.py:377-455 and atomic_model.py:347-380, 606-612, 662-671.  Pure host code
// (a few thousand points, once per problem); the device-side part of N3 -- grouping wavelengths by their set of active
// transitions into tile classes -- lives in lsx_hip.hip (build_schedule).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/lsx.h"

namespace lsxd {
int fail(int code, const char* fmt, ...);
}
using lsxd::fail;

namespace {
constexpr double kCLight = 2.99792458E+08, kHPlanck = 6.6260755E-34, kERydberg = 2.1798741E-18, kNM = 1.0E-09,
                 kVMicroChar = 3.0E3;

// position of the first grid point >= x (numpy.searchsorted, side='left')
inline int32_t lower(const std::vector<double>& g, double x) { return (int32_t)(std::lower_bound(g.begin(), g.end(), x) - g.begin()); }

// second derivatives of the cubic through (x, y) with continuous third derivative at x[1] and x[n-2] (what scipy's
// interp1d(kind=3) builds); dense elimination with partial pivoting -- n is a few dozen
std::vector<double> cubic_moments(int n, const double* x, const double* y)
{
    std::vector<double> A((size_t)n * n, 0.0), b(n, 0.0), M(n, 0.0);
    auto h = [&](int i) { return x[i + 1] - x[i]; };
    A[0] = h(1); A[1] = -(h(0) + h(1)); A[2] = h(0);
    double* last = &A[(size_t)(n - 1) * n];
    last[n - 3] = h(n - 2); last[n - 2] = -(h(n - 3) + h(n - 2)); last[n - 1] = h(n - 3);
    for (int i = 1; i < n - 1; ++i) {
        double* r = &A[(size_t)i * n];
        r[i - 1] = h(i - 1); r[i] = 2.0 * (h(i - 1) + h(i)); r[i + 1] = h(i);
        b[i] = 6.0 * ((y[i + 1] - y[i]) / h(i) - (y[i] - y[i - 1]) / h(i - 1));
    }
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(A[(size_t)r * n + c]) > std::fabs(A[(size_t)p * n + c])) p = r;
        if (p != c) { std::swap_ranges(&A[(size_t)c * n], &A[(size_t)c * n] + n, &A[(size_t)p * n]); std::swap(b[c], b[p]); }
        for (int r = c + 1; r < n; ++r) {
            const double f = A[(size_t)r * n + c] / A[(size_t)c * n + c];
            if (f == 0.0) continue;
            for (int q = c; q < n; ++q) A[(size_t)r * n + q] -= f * A[(size_t)c * n + q];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; --r) {
        double acc = b[r];
        for (int q = r + 1; q < n; ++q) acc -= A[(size_t)r * n + q] * M[q];
        M[r] = acc / A[(size_t)r * n + r];
    }
    return M;
}

// Seaton (1960) bound-free Gaunt factor, utils.py:24-32
inline double gaunt_bf(double wvl, double nEff, double charge)
{
    const double x = kHPlanck * kCLight / (wvl * kNM) / (kERydberg * charge * charge);
    const double x3 = std::pow(x, 1.0 / 3.0);
    const double nsqx = 1.0 / (nEff * nEff * x);
    return 1.0 + 0.1728 * x3 * (1.0 - 2.0 * nsqx) - 0.0496 * x3 * x3 * (1.0 - (1.0 - nsqx) * (2.0 / 3.0) * nsqx);
}
} // namespace

extern "C" int lsx_wavelength_grid(int32_t Ntrans, const lsx_trans_grid* trans, int32_t Nextra, const double* extra,
                                   double lambdaReference, int32_t capacity, double* wavelength, int32_t* Nspect,
                                   int32_t* blueIdx, int32_t* redIdx)
{
    if (Ntrans < 0 || (Ntrans && !trans) || Nextra < 0 || (Nextra && !extra) || !Nspect || capacity < 0 || (capacity && !wavelength))
        return fail(LSX_EINVAL, "lsx_wavelength_grid: bad arguments");
    std::vector<double> g;
    g.insert(g.end(), extra, extra + Nextra);
    g.push_back(lambdaReference);
    for (int kr = 0; kr < Ntrans; ++kr) {
        const lsx_trans_grid& t = trans[kr];
        if (t.n < 1 || !t.wavelength) return fail(LSX_EINVAL, "lsx_wavelength_grid: transition %d has no grid", kr);
        if (!std::is_sorted(t.wavelength, t.wavelength + t.n)) return fail(LSX_EINVAL, "lsx_wavelength_grid: grid of transition %d is not ascending", kr);
        if (t.is_line) {
            g.insert(g.end(), t.wavelength, t.wavelength + t.n);
        } else {                                                              // the edge itself and the points up to it, :397-399
            g.push_back(t.lambdaEdge);
            for (int q = 0; q < t.n; ++q) if (t.wavelength[q] <= t.lambdaEdge) g.push_back(t.wavelength[q]);
        }
    }
    std::sort(g.begin(), g.end());
    g.erase(std::unique(g.begin(), g.end()), g.end());
    *Nspect = (int32_t)g.size();
    if ((size_t)capacity < g.size()) return fail(LSX_EINVAL, "lsx_wavelength_grid: the merged grid has %zu points, capacity %d", g.size(), capacity);
    std::copy(g.begin(), g.end(), wavelength);
    for (int kr = 0; kr < Ntrans; ++kr) {
        const lsx_trans_grid& t = trans[kr];
        int32_t blue = lower(g, t.wavelength[0]), red = lower(g, t.wavelength[t.n - 1]) + 1;
        if (!t.is_line) {
            if (red > (int32_t)g.size()) red = (int32_t)g.size();          // a grid running past the edge: its end is not in g
            while (red > blue && g[red - 1] > t.lambdaEdge) --red;           // :411-414
        }
        if (blueIdx) blueIdx[kr] = blue;
        if (redIdx) redIdx[kr] = red;
    }
    return LSX_OK;
}

extern "C" int lsx_active_set(int32_t Ntrans, int32_t Nspect, const int32_t* blueIdx, const int32_t* redIdx, uint8_t* active)
{
    if (Ntrans < 0 || Nspect < 0 || (Ntrans && (!blueIdx || !redIdx || !active))) return fail(LSX_EINVAL, "lsx_active_set: bad arguments");
    for (int kr = 0; kr < Ntrans; ++kr) {
        if (blueIdx[kr] < 0 || redIdx[kr] > Nspect || blueIdx[kr] > redIdx[kr])
            return fail(LSX_EINVAL, "lsx_active_set: transition %d covers [%d, %d) of %d", kr, blueIdx[kr], redIdx[kr], Nspect);
        uint8_t* row = active + (size_t)kr * Nspect;
        std::fill(row, row + Nspect, (uint8_t)0);
        std::fill(row + blueIdx[kr], row + redIdx[kr], (uint8_t)1);
    }
    return LSX_OK;
}

extern "C" int lsx_line_wavelength(double lambda0, double qCore, double qWing, int32_t NlambdaGen, int32_t capacity,
                                   double* wavelength, int32_t* n)
{
    if (!n || NlambdaGen < 3 || !(lambda0 > 0.0) || !(qCore > 0.0) || !(qWing > 0.0)) return fail(LSX_EINVAL, "lsx_line_wavelength: bad arguments");
    const int half = (NlambdaGen % 2 == 1 ? NlambdaGen / 2 : (NlambdaGen - 1) / 2) + 1;      // points of one wing incl. the core, :353-354
    *n = 2 * half - 1;
    if (capacity < *n || !wavelength) return fail(LSX_EINVAL, "lsx_line_wavelength: %d points, capacity %d", *n, capacity);
    const double beta = qWing <= 2.0 * qCore ? 1.0 : qWing / (2.0 * qCore);                    // :356-361
    const double y = beta + std::sqrt(beta * beta + (beta - 1.0) * half + 2.0 - 3.0 * beta);
    const double b = 2.0 * std::log(y) / (half - 1), a = qWing / (half - 2.0 + y * y);
    const double qToLambda = lambda0 * (kVMicroChar / kCLight);
    const int mid = half - 1;
    wavelength[mid] = lambda0;
    for (int q = 1; q < half; ++q) {
        const double d = qToLambda * (a * (q + (std::exp(b * q) - 1.0)));
        wavelength[mid - q] = lambda0 - d;
        wavelength[mid + q] = lambda0 + d;
    }
    return LSX_OK;
}

extern "C" int lsx_continuum_alpha(const lsx_continuum_model* c, int32_t n, const double* wavelength, double* alpha)
{
    if (!c || n < 0 || (n && (!wavelength || !alpha))) return fail(LSX_EINVAL, "lsx_continuum_alpha: bad arguments");
    if (c->hydrogenic) {
        if (!(c->E_j > c->E_i) || c->stage_j < 1) return fail(LSX_EINVAL, "lsx_continuum_alpha: hydrogenic continuum needs E_j > E_i and stage_j >= 1");
        const double Z = c->stage_j, nEff = Z * std::sqrt(kERydberg / (c->E_j - c->E_i));
        const double gbf0 = gaunt_bf(c->lambdaEdge, nEff, Z);
        for (int q = 0; q < n; ++q) {
            const double w = wavelength[q], r = w / c->lambdaEdge;
            alpha[q] = (w < c->minLambda || w > c->lambdaEdge) ? 0.0 : c->alpha0 * gaunt_bf(w, nEff, Z) / gbf0 * (r * r * r);
        }
        return LSX_OK;
    }
    const int m = c->n;
    const double *x = c->wavelength, *y = c->alpha;
    if (m < 4 || !x || !y) return fail(LSX_EINVAL, "lsx_continuum_alpha: an explicit continuum needs >= 4 tabulated points");
    for (int i = 0; i + 1 < m; ++i) if (!(x[i + 1] > x[i])) return fail(LSX_EINVAL, "lsx_continuum_alpha: tabulated wavelengths must ascend strictly");
    const std::vector<double> M = cubic_moments(m, x, y);
    auto outside = [&](double w) { return w < x[0] || w > x[m - 1]; };                        // interp1d fill_value = 0
    auto cut = [&](double w) { return w < c->minLambda || w > c->lambdaEdge; };
    bool negative = false;
    for (int q = 0; q < n; ++q) {
        const double w = wavelength[q];
        double v = 0.0;
        if (!outside(w)) {
            int i = (int)(std::upper_bound(x, x + m, w) - x) - 1;
            i = std::min(std::max(i, 0), m - 2);
            const double h = x[i + 1] - x[i], p = x[i + 1] - w, r = w - x[i];
            v = (M[i] * p * p * p + M[i + 1] * r * r * r) / (6.0 * h) + (y[i] / h - M[i] * h / 6.0) * p + (y[i + 1] / h - M[i + 1] * h / 6.0) * r;
        }
        alpha[q] = cut(w) ? 0.0 : v;
        negative = negative || alpha[q] < 0.0;
    }
    if (negative) {                                   // :610-611: the whole array again, linear, WITHOUT the edge cuts
        for (int q = 0; q < n; ++q) {
            const double w = wavelength[q];
            if (outside(w)) { alpha[q] = 0.0; continue; }
            int hi = (int)(std::lower_bound(x, x + m, w) - x);
            hi = std::min(std::max(hi, 1), m - 1);
            const int lo = hi - 1;
            alpha[q] = (y[hi] - y[lo]) / (x[hi] - x[lo]) * (w - x[lo]) + y[lo];
        }
    }
    return LSX_OK;
}
