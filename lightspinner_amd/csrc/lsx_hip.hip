// lsx_hip.hip -- host runtime of the lsx C ABI (include/lsx.h) on HIP + the small kernels
// around the sweep: upload-time re-layout, Gamma epilogue, statistical equilibrium,
// convergence reductions, stand-alone batched formal solver.  gfx950 only.
//
// Reference lines restated here:
//   rh_method.py:587-590, 698-708  Gamma = C prologue, diagonal fix-up, dJ      (k_gamma_finish)
//   rh_method.py:710-745           stat_equil                                   (k_stat_equil)
//   rh_method.py:453-454           Boltzmann factor of the continuum g_ij        (k_build_E)
//   formal_solver.py:46-212        stand-alone piecewise_linear_1d               (k_piecewise)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "lsx_ctx.h"
#include "lsx_fast.h"

// A context launches up to six tile classes on streams of their own; with the HIP runtime's default of four hardware queues two
// of them wait for the others.  The runtime reads GPU_MAX_HW_QUEUES when it initialises, i.e. at the process's first
// HIP call.  The library does NOT touch the process environment on its own (a dlopen'ed library that calls setenv races with the
// host's other threads and changes every HIP user of the process): the Python package and bench.py set the default at import;
// a host that binds the C ABI directly sets the variable itself or calls lsx_hip_request_hw_queues() before its first HIP call
// (INTEGRATION.md 2).
extern "C" int lsx_hip_request_hw_queues(int32_t n)
{
    if (n < 1 || n > 64) return LSX_EINVAL;
    char buf[16];
    snprintf(buf, sizeof buf, "%d", (int)n);
    return setenv("GPU_MAX_HW_QUEUES", buf, 0) == 0 ? LSX_OK : LSX_EDEVICE;      // (a value the caller has set is left alone)
}

namespace lsxd {
thread_local std::string g_err;
int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
int ensure_stage(lsx_ctx* c, size_t doubles)
{
    if (doubles <= c->stage_doubles) return LSX_OK;
    if (c->d_stage) HIPCHK(hipFree(c->d_stage));
    c->d_stage = nullptr;
    c->stage_doubles = 0;
    int rc = dmalloc(&c->d_stage, doubles);
    if (rc) return rc;
    c->stage_doubles = doubles;
    return LSX_OK;
}
} // namespace lsxd
using namespace lsxd;

// launcher defined in lsx_sweep.hip
extern "C" hipError_t lsx_launch_sweep(const SweepParams*, int, int, size_t, hipStream_t);
// launcher defined in lsx_sweep_rs.hip (ray-serial instances)
extern "C" hipError_t lsx_launch_sweep_rs(const SweepParams*, int, hipStream_t);
extern "C" hipError_t lsx_launch_sweep_rs_par(const SweepParams*, int, hipStream_t);     // N4 (lsx_sweep_rs.hip built with LSX_RS_PARABOLIC_TU)
// the parabolic rule (N4) for one class: compile-time instance or the generic one on the class's tile list (lsx_sweep.hip)
extern "C" hipError_t lsx_launch_sweep_par(const SweepParams*, int, int, size_t, hipStream_t);

namespace {

constexpr double kCLight = 2.99792458E+08;
constexpr double kHPlanck = 6.6260755E-34;
constexpr double kKBoltzmann = 1.380658E-23;
constexpr double kNM_TO_M = 1.0E-09;
constexpr double kHC = kHPlanck * kCLight;


// ------------------------------------------------------------------------------- kernels
// where a (tile, line) block of the line-profile store lives (lsx_dev.h, phi_elem): columns col0 + blockIdx.y of the context
struct PhiBlock {
    double* store;          // phi_T
    size_t col_stride;      // doubles per column
    size_t col0;            // first column of this launch
    size_t base;            // the block's offset in a column's own numbering
    int G;                  // columns per group
};

// one (tile, line) block of the profile: in [col][lt][mu][dir][k] (rows lt0 .. lt0+len of the line)
//   ->  the block's rows [dir][k][mu] x [l<len] in the store     (compact: pass Nrays = 1, ndir = 1)
__global__ void k_pack_phi(const double* __restrict__ in, const PhiBlock out, int lt0, int len, int Nrays, int ndir,
                           int Ns, size_t in_col_stride)
{
    const size_t col = blockIdx.y;
    const size_t total = (size_t)len * Nrays * ndir * Ns;
    for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        size_t r = o;
        const int l = r % len; r /= len;
        const size_t x = r;                      // (d Ns + k) Nrays + mu
        const int mu = r % Nrays; r /= Nrays;
        const int k = r % Ns; r /= Ns;
        const int d = (int)r;
        out.store[phi_elem(out.col0 + col, out.G, out.col_stride, out.base, x, len, l)] =
            in[col * in_col_stride + (((size_t)(lt0 + l) * Nrays + mu) * ndir + d) * Ns + k];
    }
}


// inverse of k_pack_phi (lsx_get of LSX_PHI)
__global__ void k_unpack_phi(const PhiBlock in, double* __restrict__ out, int lt0, int len, int Nrays, int ndir,
                             int Ns, size_t out_col_stride)
{
    const size_t col = blockIdx.y;
    const size_t total = (size_t)len * Nrays * ndir * Ns;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        size_t r = o;
        const int l = r % len; r /= len;
        const size_t x = r;
        const int mu = r % Nrays; r /= Nrays;
        const int k = r % Ns; r /= Ns;
        const int d = (int)r;
        out[col * out_col_stride + (((size_t)(lt0 + l) * Nrays + mu) * ndir + d) * Ns + k] =
            in.store[phi_elem(in.col0 + col, in.G, in.col_stride, in.base, x, len, l)];
    }
}

// ---- Voigt function H(a, v) = Re w(v + i a), a > 0 (utils.py:13-15 calls scipy's wofz) ----
// Trapezoid rule with step h = 1/2 on w(z) = (i/pi) int exp(-t^2)/(z - t) dt plus the residue of the pole the
// contour crosses (Chiarella & Reichel 1968; Matta & Reichel 1971):
//   H = (h a/pi) sum_n exp(-g_n^2) / ((v - g_n)^2 + a^2) + Re[ 2 exp(-z^2) / (1 -+ exp(-2 pi i z/h)) ]
// on the grid g_n = n h (sign -) or (n + 1/2) h (sign +), whichever keeps v at least h/4 away from a node; error
// ~ exp(-pi^2/h^2) = 7e-18.  Every term of the sum is positive (no cancellation in the far wings).
// W: [2][28] = exp(-g_n^2) for n = -14 .. 13 on the two grids (host-computed).
__device__ __forceinline__ double dev_voigt(double a, double v, const double* __restrict__ W)
{
    const double h = 0.5;
    const double x = fabs(v);
    const double t = x * 2.0, fr = t - floor(t);
    const bool half = !(fr >= 0.25 && fr < 0.75);
    const double shift = half ? 0.5 : 0.0;
    const double* w = W + (half ? 28 : 0);
    const double a2 = a * a;
    double s = 0.0;
#pragma unroll 4
    for (int n = -14; n <= 13; ++n) {
        const double d = x - ((double)n + shift) * h;
        s += w[n + 14] / (d * d + a2);
    }
    double H = (h / M_PI) * a * s;
    if (x < 27.0 && a < 2.0 * M_PI) {
        // exp(-z^2) = exp(a^2 - x^2) (cos 2xa - i sin 2xa);  exp(-2 pi i z/h) = exp(4 pi a) (cos - i sin)(4 pi x)
        double s1, c1, st, ct;
        sincos(2.0 * x * a, &s1, &c1);
        sincospi(4.0 * x, &st, &ct);
        const double er = exp(a2 - x * x), E = exp(4.0 * M_PI * a), sg = half ? 1.0 : -1.0;
        const double dr = 1.0 + sg * E * ct, di = -sg * E * st;
        H += 2.0 * er * (c1 * dr - s1 * di) / (dr * dr + di * di);
    }
    return H;
}

struct VoigtParams {
    int Ns, Nrays, ndir, Nlines, Natoms;
    const double* wavelength;   // [Nspect]
    const double* muz;          // [Nrays]
    const double* wmu;          // [Nrays]
    const double* W;            // [2][28]
    const double* aDamp;        // [nb][Nlines][Ns]
    const double* vBroad;       // [nb][Natoms][Ns]
    const double* vlos;         // [nb][Ns] or null
};

// compute_phi (rh_method.py:228-239) straight into one (tile, line) block of phi_T: rows [dir][k][mu] x [l<len]
__global__ void k_voigt_block(const VoigtParams q, const PhiBlock out, int first, int len,
                              int line, int atom, double lambda0)
{
    const size_t col = blockIdx.y;
    const size_t total = (size_t)len * q.Nrays * q.ndir * q.Ns;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        size_t r = o;
        const int l = r % len; r /= len;
        const int mu = r % q.Nrays; r /= q.Nrays;
        const int k = r % q.Ns; r /= q.Ns;
        const int d = (int)r;
        const double vb = q.vBroad[(col * q.Natoms + atom) * q.Ns + k];
        const double ad = q.aDamp[(col * q.Nlines + line) * q.Ns + k];
        const double vl = (q.vlos && q.ndir == 2) ? q.vlos[col * q.Ns + k] : 0.0;
        const double v = (q.wavelength[first + l] - lambda0) * kCLight / (vb * lambda0);          // :234
        const double vk = v + (d ? 1.0 : -1.0) * (q.muz[mu] * vl / vb);                            // :231, :237-238
        out.store[phi_elem(out.col0 + col, out.G, out.col_stride, out.base, o / len, len, l)] = dev_voigt(ad, vk, q.W) / (sqrt(M_PI) * vb);     // :239
    }
}

// wphi = 1 / sum_{lambda, mu, dir} phi wlambda (wmu / 2)  (rh_method.py:236-242): one thread per (column, depth)
__global__ void k_voigt_wphi(const VoigtParams q, double* __restrict__ wphi, int ncol, int Nblue, int Nlam, int line, int atom,
                             double lambda0)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)ncol * q.Ns) return;
    const size_t col = gid / q.Ns;
    const int k = gid % q.Ns;
    const double vb = q.vBroad[(col * q.Natoms + atom) * q.Ns + k];
    const double ad = q.aDamp[(col * q.Nlines + line) * q.Ns + k];
    const double vl = q.vlos ? q.vlos[col * q.Ns + k] : 0.0;
    const double* wl = q.wavelength + Nblue;
    double acc = 0.0;
    for (int la = 0; la < Nlam; ++la) {
        double wla = la == 0 ? 0.5 * (wl[1] - wl[0]) : (la == Nlam - 1 ? 0.5 * (wl[la] - wl[la - 1]) : 0.5 * (wl[la + 1] - wl[la - 1]));
        wla *= kCLight / lambda0;                                                                   // :157-196
        const double v = (wl[la] - lambda0) * kCLight / (vb * lambda0);
        for (int mu = 0; mu < q.Nrays; ++mu) {
            const double sh = q.muz[mu] * vl / vb;
            const double wt = wla * 0.5 * q.wmu[mu];
            for (int d = 0; d < 2; ++d) {
                const double p = dev_voigt(ad, v + (d ? sh : -sh), q.W) / (sqrt(M_PI) * vb);
                acc += p * wt;
                if (vl == 0.0) { acc += p * wt; break; }        // ray independent profile: both directions are equal
            }
        }
    }
    wphi[col * q.Nlines * q.Ns + (size_t)line * q.Ns + k] = 1.0 / acc;
}

// Boltzmann factor of the continuum g_ij (rh_method.py:453-454: g_ij = nStar_i / nStar_j * exp(-hc / (k lambda T))),
// tile-major like the background streams: E_T[col][tile][k][j].  Depends on the temperature only: built at upload.
__global__ void k_build_E(const double* __restrict__ temperature, const double* __restrict__ wavelength,
                          double* __restrict__ out, const DevTile* __restrict__ tiles, int ntile, int L, int Ns,
                          const double* __restrict__ exp2_tab)
{
    // ONE Boltzmann factor in the library (round 6): the stream holds the bits the ray-serial sweep forms in the lane
    // (lsx_dev.h, boltzmann_factor: (-hc / k lambda) * (1 / T) through the sweep's table exponential)
    __shared__ double etab_s[LSX_EXP_TAB];
    for (int e = threadIdx.x; e < LSX_EXP_TAB; e += blockDim.x) etab_s[e] = exp2_tab[e];
    __syncthreads();
    const size_t col = blockIdx.y;
    const int total = ntile * Ns * L;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
        const int j = o % L, k = (o / L) % Ns, t = o / (L * Ns);
        const int la = tiles[t].la0 + (j < tiles[t].nla ? j : tiles[t].nla - 1);
        out[col * (size_t)total + o] = boltzmann_factor(boltzmann_lane_constant(wavelength[la]), 1.0 / temperature[col * Ns + k], (const lds_f64*)etab_s);
    }
}

// out[i] = (a[i], b[i]) as pairs (the background's opacity and emissivity for the folded ray-serial instances: lsx_dev.h, bgce_T)
__global__ void k_interleave2(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        out[2 * i] = a[i];
        out[2 * i + 1] = b[i];
    }
}

// nStar_i / nStar_j of every continuum: out[col][cont][k]
__global__ void k_build_nsr(const double* __restrict__ nStar, double* __restrict__ out, const int* __restrict__ li,
                            const int* __restrict__ lj, int Ncont, int Ns, int NLtot)
{
    const size_t col = blockIdx.y;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < Ncont * Ns; o += gridDim.x * blockDim.x) {
        const int k = o % Ns, q = o / Ns;
        out[col * Ncont * Ns + o] = nStar[(col * NLtot + li[q]) * Ns + k] / nStar[(col * NLtot + lj[q]) * Ns + k];
    }
}

// The ray-serial sweep's operand table (lsx_plan.h, "a RING in LDS"): per group of five columns and transition the three per-depth
// numbers a slot of that transition needs -- lines: cB (n_i - g n_j) (rh_method.py:279-280, 613), n_j Uc (:281, 614), wphi (:451);
// continua: n_i, n_j, nStar_i / nStar_j (:453) -- and per group the geometry: half length of the interval above the depth, scattering
// coefficient.  Row Nspace of every block is zero.  A group's missing columns (the last group of a batch) repeat its last one.
// One block per (group, transition | geometry); rebuilt at the start of every formal solution that runs ray-serial classes (the
// populations change with every statistical equilibrium): 50 kB per column for FALC Ca+H, written once and read by the sweeps.
struct OptabParams {
    int Ns, Ntrans, ncol, NLtot, Nlines, Ncont;
    const DevTrans* trans;
    const int* trans_row;
    const int *cont_li, *cont_lj;
    const double *n, *wphi, *nsr, *height, *sca, *temperature;
    double* optab;
    size_t gstride;
};
__global__ void k_build_optab(const OptabParams p)
{
    const int g = blockIdx.x, t = blockIdx.y, Ns = p.Ns;
    constexpr int NC = LSX_RS_COLS, PAD = LSX_RS_RING, SEG = LSX_RS_SEG;      // block rows of SEG doubles: [column][3], the rest zero (never written: the table starts zeroed)
    const int NR = lsx_optab_rows(Ns);                           // rows per block: depth r is row r + PAD, zero rows around
    const int ncg = min(NC, p.ncol - g * NC);
    double* const grp = p.optab + (size_t)g * p.gstride;
    if (t < p.Ntrans) {
        const DevTrans tr = p.trans[t];
        const double Uc = tr.AB * (tr.gij * tr.cB);               // DevSlot.Uc (lsx_plan.cpp)
        double* const blk = grp + (size_t)t * NR * SEG;
        for (int e = threadIdx.x; e < NR * NC; e += blockDim.x) {
            const int row = e / NC, r = row - PAD, c = e - row * NC;
            double v0 = 0.0, v1 = 0.0, v2 = 0.0;
            if (r >= 0 && r < Ns) {
                const size_t col = (size_t)g * NC + (c < ncg ? c : ncg - 1);
                const double ni = p.n[(col * p.NLtot + tr.li) * Ns + r], nj = p.n[(col * p.NLtot + tr.lj) * Ns + r];
                if (tr.is_line) {
                    v0 = tr.cB * (ni - tr.gij * nj);
                    v1 = nj * Uc;
                    v2 = p.wphi[(col * p.Nlines + p.trans_row[t]) * Ns + r];
                } else {
                    v0 = ni;
                    v1 = nj;
                    v2 = p.nsr[(col * p.Ncont + p.trans_row[t]) * Ns + r];
                }
            }
            double* const o = blk + (size_t)row * SEG + c * 3;
            o[0] = v0; o[1] = v1; o[2] = v2;
        }
    } else if (t < p.Ntrans + 2) {
        const int up = t - p.Ntrans;                              // 0: the down-going sweep's geometry, 1: the up-going one's
        constexpr int GEO = LSX_RS_GEO;
        double* const blk = grp + ((size_t)p.Ntrans + (size_t)up) * SEG * NR;
        for (int e = threadIdx.x; e < NR * NC; e += blockDim.x) {
            const int row = e / NC, r = row - PAD, c = e - row * NC;
            const size_t col = (size_t)g * NC + (c < ncg ? c : ncg - 1);
            const double* z = p.height + col * Ns;
            double hz = 0.0;
            if (r >= 0 && r < Ns) hz = up ? (r + 1 < Ns ? 0.5 * fabs(z[r] - z[r + 1]) : 0.0) : (r > 0 ? 0.5 * fabs(z[r - 1] - z[r]) : 0.0);
            double* const o = blk + (size_t)row * SEG + c * GEO;
            o[0] = hz;
            o[1] = (r >= 0 && r < Ns) ? p.sca[col * Ns + r] : 0.0;
            if constexpr (GEO > 2) o[2] = (r >= 0 && r < Ns) ? 1.0 / p.temperature[col * Ns + r] : 0.0;     // (lsx_dev.h, boltzmann_factor)
        }
    } else {
        const int q = t - p.Ntrans - 2;                           // continuum q: n_i, n_j nStar_i / nStar_j, nStar_i / nStar_j (the folded instances' operands)
        double* const blk = grp + ((size_t)p.Ntrans + 2 + (size_t)q) * SEG * NR;
        const int li = p.cont_li[q], lj = p.cont_lj[q];
        for (int e = threadIdx.x; e < NR * NC; e += blockDim.x) {
            const int row = e / NC, r = row - PAD, c = e - row * NC;
            double v0 = 0.0, v1 = 0.0, v2 = 0.0;
            if (r >= 0 && r < Ns) {
                const size_t col = (size_t)g * NC + (c < ncg ? c : ncg - 1);
                v2 = p.nsr[(col * p.Ncont + q) * Ns + r];
                v0 = p.n[(col * p.NLtot + li) * Ns + r];
                v1 = p.n[(col * p.NLtot + lj) * Ns + r] * v2;
            }
            double* const o = blk + (size_t)row * SEG + c * 3;
            o[0] = v0; o[1] = v1; o[2] = v2;
        }
    }
}

// [col][la][k] (reference layout)  <->  [col][tile][k][j<L] (tile-major streams); unused j are zero
__global__ void k_tiles_pack(const double* __restrict__ in, double* __restrict__ out, const DevTile* __restrict__ tiles,
                             int ntile, int L, int Ns, int Nspect, bool unpack)
{
    const size_t col = blockIdx.y;
    const int total = ntile * Ns * L;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
        const int j = o % L, k = (o / L) % Ns, t = o / (L * Ns);
        const bool in_tile = j < tiles[t].nla;
        const size_t ref = (col * Nspect + (size_t)(tiles[t].la0 + j)) * Ns + k;
        const size_t til = col * (size_t)total + o;
        if (unpack) { if (in_tile) out[ref] = in[til]; }
        else out[til] = in_tile ? in[ref] : 0.0;
    }
}

struct FinishParams {
    int Nspace, Natoms, NL2tot, ncol, ntile, nslot_total;
    const int* Nlevel;      // [Natoms]
    const int* lev2_off;    // [Natoms]
    const DevTile* tiles;
    const int* tile_slots;
    const DevTrans* trans;
    const double* C;
    const double* Gpart;
    const double* dJpart;
    double* Gamma;
    double* dJcol;
    const uint8_t* colmask;
    double* dPcol;                      // zeroed here for the stat_equil that follows (its per-column maxima are atomic) ...
    unsigned long long* singular;       // likewise ...
    int clear_dp;                       // ... unless a stat_equil's monitors are still waiting to be read back (0)
    const int* fin_ptr;                 // [NL2tot + 1]  k_gamma_finish_small: the slabs that add up to one entry,
    const int* fin_idx;                 //               as (slot * 4 + slab), in slot order
    const int* atom_ptr;                // [Natoms + 1]  k_gamma_finish: the slots of one atom's transitions, in slot order
    const int* atom_slots;
};

// Gamma = C + sum of the sweep's slabs in (tile, slot, entry, direction) order; then the
// diagonal (rh_method.py:587-590, 698-703).  One thread per (column, depth, ATOM) (blockIdx.y = atom, round 4): a thread keeps only
// its atom's Nlevel^2 entries in LDS and walks only its atom's slots (lists made with the context, in slot order: every entry is
// summed in the order it always was -- the same bits), so a two-atom problem has twice the resident waves (Ca+H: 37 kB of LDS per 128
// threads instead of per 64) and half the chain per thread: C4 0.33 -> see DESIGN.md 4.3.
__global__ void k_gamma_finish(const FinishParams f)
{
    extern __shared__ double sm[];
    const int Ns = f.Nspace;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int atom = blockIdx.y;
    const long gid = (long)blockIdx.x * nt + tid;
    if (gid >= (long)f.ncol * Ns) return;
    const int col = gid / Ns, k = gid % Ns;
    if (f.clear_dp && atom == 0) {
        if (k == 0) f.dPcol[col] = 0.0;
        if (gid == 0) *f.singular = 0ull;
    }
    if (f.colmask && !f.colmask[col]) {
        if (k == 0 && atom == 0) f.dJcol[col] = 0.0;
        return;
    }
    const int Nl = f.Nlevel[atom], off2 = f.lev2_off[atom], Nl2 = Nl * Nl;
    double* G = sm + tid;                                   // thread-private Gamma of this atom: G[e * nt]
    const double* Cm = f.C + ((size_t)col * f.NL2tot + off2) * Ns + k;
    for (int e = 0; e < Nl2; ++e) G[e * nt] = 0.0 + Cm[(size_t)e * Ns];     // Gamma = C, :587-590
    const double* P = f.Gpart + (size_t)col * f.nslot_total * 4 * Ns + k;
    // slabs in (tile, slot, entry, direction) order = slot-table order; the loads of several slots are in flight at once
    const int s0 = f.atom_ptr[atom], s1 = f.atom_ptr[atom + 1];
#pragma unroll 8
    for (int v = s0; v < s1; ++v) {
        const int u = f.atom_slots[v];
        const int ts = f.tile_slots[u];
        const bool fast = (ts >> 30) & 1;                    // fast continuum: one entry carries both directions
        const DevTrans& tr = f.trans[ts & 0x3fffffff];
        const double* q = P + (size_t)u * 4 * Ns;
        // (unconditional loads, so that the loads of several slots stay in flight together: a fast slot re-reads its first entries)
        const double q0 = q[0], r1 = q[fast ? 0 : (size_t)Ns], q2 = q[(size_t)2 * Ns], r3 = q[(size_t)(fast ? 2 : 3) * Ns];
        const double q1 = fast ? 0.0 : r1, q3 = fast ? 0.0 : r3;
        const int eij = tr.gam_ij - off2, eji = tr.gam_ji - off2;
        double gij = G[eij * nt], gji = G[eji * nt];
        gij += q0;
        gij += q1;
        gji += q2;
        gji += q3;
        G[eij * nt] = gij;
        G[eji * nt] = gji;
    }
    double* Gout = f.Gamma + ((size_t)col * f.NL2tot + off2) * Ns + k;
    for (int i = 0; i < Nl; ++i) G[(i * Nl + i) * nt] = 0.0;
    for (int i = 0; i < Nl; ++i) {                           // Gamma_ii = -sum_{l != i} Gamma_li, :698-703
        double s = 0.0;
        for (int l = 0; l < Nl; ++l) s += G[(l * Nl + i) * nt];
        G[(i * Nl + i) * nt] = -s;
    }
    for (int e = 0; e < Nl2; ++e) Gout[(size_t)e * Ns] = G[e * nt];
    if (k == 0 && atom == 0) { // per-column dJ: max over the column's tiles, NaN propagating (rh_method.py:705-706)
        double m = 0.0;
        for (int t = 0; t < 2 * f.ntile; ++t) {
            const double v = f.dJpart[(size_t)col * 2 * f.ntile + t];
            m = (v != v || m != m) ? __builtin_nan("") : fmax(m, v);
        }
        f.dJcol[col] = m;
    }
}

// The same epilogue for atoms of many levels (LaunchShapes.finish_w > 0, lsx_plan.cpp: a thread's whole Gamma no longer fits the LDS
// of 128 threads -- carbon, iron: 15 levels, MgII: 11): a thread takes ONE COLUMN i of its atom's Gamma, the entries (l, i) of every
// row l, one after the other in registers -- each summed from the entry's slab list (fin_ptr / fin_idx, what k_gamma_finish_small
// walks: slot order, i.e. the order k_gamma_finish adds them in), stored, and added to the column sum that becomes the diagonal
// (Gamma_ii = -sum_l Gamma_li with the diagonal's own place counted as zero: a column sum, complete inside one thread).  No LDS; one
// workgroup row per level of the problem.  The same bits as k_gamma_finish and k_gamma_finish_small.
__global__ void __launch_bounds__(128) k_gamma_finish_levels(const FinishParams f, const int* __restrict__ level_atom, const int* __restrict__ level_first)
{
    const int Ns = f.Nspace;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)f.ncol * Ns) return;
    const int col = gid / Ns, k = gid % Ns;
    const int lev = blockIdx.y;                              // global level index: column i of atom `atom`
    const int atom = level_atom[lev], i = lev - level_first[atom];
    const bool first = lev == 0;
    if (f.clear_dp && first) {
        if (k == 0) f.dPcol[col] = 0.0;
        if (gid == 0) *f.singular = 0ull;
    }
    if (f.colmask && !f.colmask[col]) {
        if (k == 0 && first) f.dJcol[col] = 0.0;
        return;
    }
    const int Nl = f.Nlevel[atom], off2 = f.lev2_off[atom];
    const double* P = f.Gpart + (size_t)col * f.nslot_total * 4 * Ns + k;
    const double* Cm = f.C + ((size_t)col * f.NL2tot + off2) * Ns + k;
    double* Gout = f.Gamma + ((size_t)col * f.NL2tot + off2) * Ns + k;
    double s = 0.0;
    for (int l = 0; l < Nl; ++l) {
        if (l == i) { s += 0.0; continue; }
        const int e = l * Nl + i;
        double g = 0.0 + Cm[(size_t)e * Ns];                 // Gamma = C, :587-590
        const int b = f.fin_ptr[off2 + e], n = f.fin_ptr[off2 + e + 1];
#pragma unroll 4
        for (int x = b; x < n; ++x) g += P[(size_t)f.fin_idx[x] * Ns];
        Gout[(size_t)e * Ns] = g;
        s += g;
    }
    Gout[(size_t)(i * Nl + i) * Ns] = -s;                    // Gamma_ii = -sum_{l != i} Gamma_li, :698-703
    if (k == 0 && first) { // per-column dJ: max over the column's tiles, NaN propagating (rh_method.py:705-706)
        double m = 0.0;
        for (int t = 0; t < 2 * f.ntile; ++t) {
            const double v = f.dJpart[(size_t)col * 2 * f.ntile + t];
            m = (v != v || m != m) ? __builtin_nan("") : fmax(m, v);
        }
        f.dJcol[col] = m;
    }
}

// The same epilogue for small batches (fewer than 32 columns: the fused sweep launch), where k_gamma_finish is one long chain
// of dependent adds per thread and nothing else runs: one wavefront per (column, depth), lane = Gamma entry.  The slabs of an
// entry are listed in slot order (fin_ptr / fin_idx, made with the context), i.e. each entry is summed in exactly the order
// k_gamma_finish sums it: the two kernels give the same bits.
__global__ void __launch_bounds__(64) k_gamma_finish_small(const FinishParams f)
{
    extern __shared__ double sm[];                          // [NL2tot]
    const int Ns = f.Nspace;
    const int col = blockIdx.x / Ns, k = blockIdx.x - col * Ns;
    const int lane = threadIdx.x;
    if (f.clear_dp && k == 0 && lane == 0) {
        f.dPcol[col] = 0.0;
        if (col == 0) *f.singular = 0ull;
    }
    if (f.colmask && !f.colmask[col]) {
        if (k == 0 && lane == 0) f.dJcol[col] = 0.0;
        return;
    }
    const double* P = f.Gpart + (size_t)col * f.nslot_total * 4 * Ns + k;
    const double* Cm = f.C + (size_t)col * f.NL2tot * Ns + k;
    for (int e = lane; e < f.NL2tot; e += 64) {
        double g = 0.0 + Cm[(size_t)e * Ns];                // Gamma = C, :587-590
        const int b = f.fin_ptr[e], n = f.fin_ptr[e + 1];
#pragma unroll 4
        for (int i = b; i < n; ++i) g += P[(size_t)f.fin_idx[i] * Ns];
        sm[e] = g;
    }
    __syncthreads();
    double diag = 0.0;                                      // Gamma_ii = -sum_{l != i} Gamma_li, :698-703 (one level per lane)
    int dpos = -1;
    for (int a = 0, base = 0; a < f.Natoms; ++a) {
        const int Nl = f.Nlevel[a];
        const int i = lane - base;
        if (i >= 0 && i < Nl) {
            const double* Ga = sm + f.lev2_off[a];
            double s = 0.0;
            for (int l = 0; l < Nl; ++l) s += (l == i) ? 0.0 : Ga[l * Nl + i];
            diag = -s;
            dpos = f.lev2_off[a] + i * Nl + i;
        }
        base += Nl;
    }
    __syncthreads();
    if (dpos >= 0) sm[dpos] = diag;
    __syncthreads();
    double* Gout = f.Gamma + (size_t)col * f.NL2tot * Ns + k;
    for (int e = lane; e < f.NL2tot; e += 64) Gout[(size_t)e * Ns] = sm[e];
    if (k == 0) {     // per-column dJ: max over the column's tiles, NaN propagating (rh_method.py:705-706)
        double m = 0.0;
        for (int t = lane; t < 2 * f.ntile; t += 64) {
            const double v = f.dJpart[(size_t)col * 2 * f.ntile + t];
            m = (v != v || m != m) ? __builtin_nan("") : fmax(m, v);
        }
#pragma unroll
        for (int w = 1; w < 64; w <<= 1) {
            const double o = __shfl_xor(m, w, 64);
            m = (o != o || m != m) ? __builtin_nan("") : fmax(m, o);
        }
        if (lane == 0) f.dJcol[col] = m;
    }
}

// ---- fast continua: FastParams, the pre-pass and the column-mapped epilogue live in lsx_fast.h (shared with the fused sweep launch)
// one fast continuum at (lambda, depth): rh_method.py:284-286, 453-455, 613-614, with g_ij = (nStar_i / nStar_j) E
struct FastVal { double alf, Vji, Uji, chi, eta; bool a; };
template <bool SEG>               // SEG: the column is too deep for its operands to be staged at once
__global__ void __launch_bounds__(256) k_fast_prepass(const FastParams f)
{
    extern __shared__ double sm[];
    fast_prepass_tile<SEG, 256>(f, f.fast_tiles[blockIdx.x], blockIdx.y, sm);       // lsx_fast.h
}

// Gamma slabs of the fast continua from J, Psibar and PsiPhi (both directions summed).  For continuum c of atom a
// (rh_method.py:652, 677-681; w = (w_mu/2) 4 pi per ray and direction, :661-665):
//   sum w (Uji + Vji Ieff - chi_a[i] Psi U_a[j]),  Ieff = I - Psi eta_a
//   eta_a    = etaC + sum_lines n_j Uc phi            chi_a[i] = XC[i] + sum_lines s(line, i) cB (n_i - g n_j) phi
//   U_a[j]   = UC[j]   (no line touches a linked continuum's upper level)
//   chi_a[j] = XC[j]                                  U_a[i]   = UC[i] + sum_lines [j_line == i] Uc phi
// with etaC, XC, UC the sums over the atom's continua of the tile, s(line, lev) = [i_line == lev] - [j_line == lev], so
//   = Uji sW + Vji sIe - (XC[i] sPsi + sum_lines s cBn sPP) UC[j],    sIe = sI - etaC sPsi - sum_lines n_j Uc sPP
// and for the reverse rate  Vij sIe - XC[j] (UC[i] sPsi + sum_lines [j_line == i] Uc sPP);
//   sI = 4 pi J, sPsi = sum w Psi*, sPP = sum w Psi* phi_line, sW = 4 pi sum_mu w_mu (all over both directions).
// One block per (depth chunk, fast tile, column); thread = (depth in chunk, wavelength of the tile).
// Tiles whose fast continua are "simple" (per atom: one common upper level, distinct lower levels, none of them
// that upper level -- every bound-free set of an ordinary model atom) need three running sums per atom and no
// level cells; anything else takes the generic path with thread-private LDS cells.  Same arithmetic, same order.
// One block per (tile, column), NT threads = (depth in chunk, wavelength) with LP = 16 / 32 / 64 lanes per depth row; the block
// walks the column's depth chunks with the next chunk's loads in flight.
template <int LP>
static __device__ __forceinline__ double row_total(double v)     // sum over the LP lanes of a depth row, valid in its last lane
{
    v = row_sums(v);
    if constexpr (LP >= 32) v += dpp_f64<0x142, 0xa>(v);         // row_bcast15: lanes 31 / 63 = two rows
    if constexpr (LP >= 64) v += dpp_f64<0x143, 0xc>(v);         // row_bcast31: lane 63 = the wave
    return v;
}
template <int LP, int NT, bool SEG>
__global__ void __launch_bounds__(NT) k_fast_gamma(const FastParams f)
{
    constexpr int KR = NT / LP;                              // depths per chunk
    extern __shared__ double sm[];
    const size_t col = blockIdx.y;
    if (f.colmask && !f.colmask[col]) return;
    const int t = f.fast_tiles[blockIdx.x];
    const DevTile tl = f.tiles[t];
    const int tid = threadIdx.x;
    const int kc = tid / LP, j = tid % LP;
    const int Ns = f.Nspace;
    const DevSlot* fs = f.slots + tl.slot0 + tl.nP;
    const DevSlot* ls = f.slots + tl.slot0;
    const int nLc = tl.nK > 0 ? min(tl.nL, LSX_MAX_TILE_LINES) : 0;
    const bool lane_on = j < tl.nla;
    const int la = tl.la0 + (lane_on ? j : 0);
    const size_t tb = (col * f.ntile + t) * (size_t)Ns * f.L;
    const size_t dstride = (size_t)f.ncol * f.ntile * Ns * f.L, pstride = (size_t)f.ncol * f.pp_col_stride, plane = (size_t)Ns * f.L;
    struct In { double J, P0, P1, E, pp[LSX_MAX_TILE_LINES][2]; };
    auto load_in = [&](int k, In& x) {
        const size_t kk = (size_t)((lane_on && k < Ns) ? k : 0) * f.L + (lane_on ? j : 0);
        x.J = f.J_T[tb + kk];
        x.P0 = f.Psi2_T[tb + kk];
        x.P1 = f.Psi2_T[dstride + tb + kk];
        x.E = f.E_T[tb + kk];
#pragma unroll
        for (int u = 0; u < LSX_MAX_TILE_LINES; ++u) {
            x.pp[u][0] = x.pp[u][1] = 0.0;
            if (u < nLc) {
                const double* pp = f.Psi3_T + col * f.pp_col_stride + tl.pp_off + (size_t)u * plane + kk;
                x.pp[u][0] = pp[0];
                x.pp[u][1] = pp[pstride];
            }
        }
    };
    In nxt;
    load_in(kc, nxt);                                        // first chunk's streams: in flight during the staging
    // LDS: cells (generic tiles only) | staged per depth segment (the whole column where it fits): sN, sR, sL | sA[q][j] =
    // {alpha, wlambda} (both 0 where the continuum is inactive: every quantity below is then 0)
    const int KS = SEG ? f.seg_depths : Ns;                   // depths staged at a time (the whole column where it fits)
    double* cellbase = sm;
    double* sN = cellbase + (size_t)(f.generic ? 2 * f.NLtot + f.Natoms : 0) * NT;    // [q][k]{n_i, n_j}
    double* sR = sN + (size_t)2 * f.nF_max * KS;                                      // [q][k] nStar_i / nStar_j
    double* sA = sR + (size_t)f.nF_max * KS;                                          // [q][j]{alpha, wlambda}
    double* sL = sA + (size_t)2 * f.nF_max * LP;                                      // [u][k]{cB (n_i - g n_j), n_j Uc}
    auto stage = [&](int ks0) {
        for (int e = tid; e < tl.nF * KS; e += NT) {
            const int q = e / KS, kk = min(ks0 + (e - q * KS), Ns - 1);
            sN[e * 2 + 0] = f.n[(col * f.NLtot + fs[q].li) * Ns + kk];
            sN[e * 2 + 1] = f.n[(col * f.NLtot + fs[q].lj) * Ns + kk];
            sR[e] = f.nsr[col * f.Ncont * Ns + fs[q].base + kk];
        }
        for (int e = tid; e < nLc * KS; e += NT) {
            const int u = e / KS, kk = min(ks0 + (e - u * KS), Ns - 1);
            const double ni = f.n[(col * f.NLtot + ls[u].li) * Ns + kk], nj = f.n[(col * f.NLtot + ls[u].lj) * Ns + kk];
            sL[e * 2 + 0] = ls[u].cB * (ni - ls[u].g * nj);      // chi_line = this * phi, rh_method.py:279-280, :613
            sL[e * 2 + 1] = nj * ls[u].Uc;                       // eta_line = this * phi, :281, :614 (Uji_line = Uc phi)
        }
    };
    stage(0);
    for (int e = tid; e < tl.nF * LP; e += NT) {
        const int q = e / LP, jj = e % LP, lq = tl.la0 + min(jj, tl.nla - 1), lt = lq - fs[q].Nblue;
        const bool a = jj < tl.nla && lt >= 0 && lt < fs[q].Nlam && f.active[(size_t)fs[q].trans * f.Nspect + lq] != 0;
        sA[e * 2 + 0] = a ? f.alpha[fs[q].wl_off + lt] : 0.0;
        sA[e * 2 + 1] = a ? f.wl[fs[q].wl_off + lt] : 0.0;
    }
    double sW = 0.0;
    for (int m = 0; m < f.Nrays; ++m) sW += 2.0 * f.wmuh[m] * (4.0 * M_PI);   // both directions
    const double ula = f.u_la[la];
    __syncthreads();
    const bool last_lane = j == LP - 1;
    const int kpad = Ns + KR - 1 - (Ns + KR - 1) % KR;       // whole passes only
    for (int ks0 = 0; ks0 < (SEG ? Ns : 1); ks0 += KS) {     // depth segments: one, unless the column is deep
    if (SEG && ks0 > 0) {
        __syncthreads();
        stage(ks0);
        __syncthreads();
    }
    const int kend = SEG ? min(ks0 + KS, kpad) : kpad;
    for (int k = ks0 + kc; k < kend; k += KR) {              // (every thread runs every pass: the DPP reductions need whole rows)
        const In x = nxt;
        load_in(k + KR, nxt);
        const bool on = lane_on && k < Ns;
        const int ks = min(k - ks0, KS - 1);                 // staged row of this depth (results dropped if k >= Ns)
        const double sI = x.J * (4.0 * M_PI), sPsi = x.P0 + x.P1, E = x.E;
        // the lines' ray sums times their depth coefficients: sum_rays w Psi* {chi, eta, Uji}_line
        double tchi[LSX_MAX_TILE_LINES], teta[LSX_MAX_TILE_LINES], tU[LSX_MAX_TILE_LINES];
#pragma unroll
        for (int u = 0; u < LSX_MAX_TILE_LINES; ++u) {
            tchi[u] = teta[u] = tU[u] = 0.0;
            if (u < nLc) {
                const double2 Lq = *reinterpret_cast<const double2*>(sL + (size_t)(u * KS + ks) * 2);
                const double sPP = x.pp[u][0] + x.pp[u][1];
                tchi[u] = Lq.x * sPP;
                teta[u] = Lq.y * sPP;
                tU[u] = ls[u].Uc * sPP;
            }
        }
        // one fast continuum at (lambda, depth) from the staged operands: rh_method.py:284-286, 453-455, 613-614
        auto value = [&](int q) {
            FastVal v;
            const double2 A = *reinterpret_cast<const double2*>(sA + (size_t)(q * LP + j) * 2);
            const double2 N = *reinterpret_cast<const double2*>(sN + (size_t)(q * KS + ks) * 2);
            const double nsr = sR[q * KS + ks];
            v.alf = A.x;
            v.a = A.x != 0.0;
            v.Vji = (nsr * E) * v.alf;
            v.Uji = ula * v.Vji;
            v.chi = N.x * v.alf - N.y * v.Vji;
            v.eta = N.y * v.Uji;
            return v;
        };
        // the lines' share of (chi_a[lev] Psi) and (U_a[lev] Psi), lev = the lower level of continuum q (its lkbits say which
        // lines touch it: bit 1 as their lower level, bit 2 as their upper level), and of (eta_a Psi) (bit 0: same atom)
        auto line_chi = [&](unsigned lk) {
            double r = 0.0;
#pragma unroll
            for (int u = 0; u < LSX_MAX_TILE_LINES; ++u) {
                if (lk & (2u << (8 * u))) r += tchi[u];
                if (lk & (4u << (8 * u))) r -= tchi[u];
            }
            return r;
        };
        auto line_U = [&](unsigned lk) {
            double r = 0.0;
#pragma unroll
            for (int u = 0; u < LSX_MAX_TILE_LINES; ++u)
                if (lk & (4u << (8 * u))) r += tU[u];
            return r;
        };
        auto line_eta = [&](unsigned lk) {
            double r = 0.0;
#pragma unroll
            for (int u = 0; u < LSX_MAX_TILE_LINES; ++u)
                if (lk & (1u << (8 * u))) r += teta[u];
            return r;
        };
        // wavelength quadrature of slot q: a DPP reduction over the row (fixed order); its last lane stores the total to
        // Gpart[slot][e][dir 0][k], direction 1 carries nothing.  (Summing through LDS with two barriers per group of slots was
        // measured 50 % slower.)
        auto emit = [&](int q, double g1, double g2) {
            const double s1 = row_total<LP>(on ? g1 : 0.0), s2 = row_total<LP>(on ? g2 : 0.0);
            if (last_lane && k < Ns) {
                double* g = f.Gpart + ((col * f.nslot_total + tl.slot0 + tl.nP + q) * 4) * (size_t)Ns + k;
                g[0] = s1;
                g[2 * (size_t)Ns] = s2;
            }
        };
        // the correction slots of the tile's lines (DevTile.nX) belong to the column-mapped epilogue (lsx_fast.h); the tiles this
        // kernel takes get their linked corrections in the sweep: zeros here
        if (last_lane && k < Ns)
            for (int x = 0; x < tl.nX; ++x) {
                double* g = f.Gpart + ((col * f.nslot_total + tl.slot0 + tl.nP + tl.nF + x) * 4) * (size_t)Ns + k;
                g[0] = 0.0;
                g[2 * (size_t)Ns] = 0.0;
            }
        if (tl.fast_simple) {
            for (int q0 = 0; q0 < tl.nF;) {                 // one atom at a time
                const int atom = fs[q0].atom;
                int q1 = q0;
                double chi_j = 0.0, U_j = 0.0, etaA = 0.0;  // atom.chi[j], atom.U[j], atom.eta of rh_method.py:616-627
                for (; q1 < tl.nF && fs[q1].atom == atom; ++q1) {
                    const FastVal v = value(q1);
                    chi_j -= v.chi;
                    U_j += v.Uji;
                    etaA += v.eta;
                }
                const double sIe = (sI - etaA * sPsi) - line_eta(fs[q0].lkbits);
                for (int q = q0; q < q1; ++q) {
                    const FastVal v = value(q);
                    const unsigned lk = fs[q].lkbits;
                    const double wla = sA[(size_t)(q * LP + j) * 2 + 1];
                    // (chi_a[i] Psi) U_a[j] with chi_a[i] = this continuum + the lines on its lower level; chi_a[j] (Psi U_a[i])
                    // with U_a[i] = the lines that end on its lower level (no continuum does in a simple set)
                    const double cU = (v.chi * U_j) * sPsi + line_chi(lk) * U_j;
                    const double cU2 = chi_j * line_U(lk);
                    emit(q, wla * ((v.Uji * sW + v.Vji * sIe) - cU), wla * ((v.alf * sIe) - cU2));
                }
                q0 = q1;
            }
        } else {
            double* cell = cellbase + tid;                   // thread-private cells: cell[c * NT]
            const int ncell = 2 * f.NLtot + f.Natoms;
            for (int c = 0; c < ncell; ++c) cell[c * NT] = 0.0;
            for (int q = 0; q < tl.nF; ++q) {
                const FastVal v = value(q);
                cell[fs[q].li * NT] += v.chi;
                cell[fs[q].lj * NT] -= v.chi;
                cell[(f.NLtot + fs[q].lj) * NT] += v.Uji;
                cell[(2 * f.NLtot + fs[q].atom) * NT] += v.eta;
            }
            for (int q = 0; q < tl.nF; ++q) {
                const FastVal v = value(q);
                const unsigned lk = fs[q].lkbits;
                const double etaA = cell[(2 * f.NLtot + fs[q].atom) * NT];
                const double chi_i = cell[fs[q].li * NT], chi_j = cell[fs[q].lj * NT];
                const double U_i = cell[(f.NLtot + fs[q].li) * NT], U_j = cell[(f.NLtot + fs[q].lj) * NT];
                const double sIe = (sI - etaA * sPsi) - line_eta(lk);
                const double wla = sA[(size_t)(q * LP + j) * 2 + 1];
                const double cU = (chi_i * U_j) * sPsi + line_chi(lk) * U_j;
                const double cU2 = (chi_j * U_i) * sPsi + chi_j * line_U(lk);
                emit(q, wla * ((v.Uji * sW + v.Vji * sIe) - cU), wla * ((v.alf * sIe) - cU2));
            }
        }
    }
    }
}

template <int NLC, int NPC>   // NLC: lines of the tile that linked continua feed (0: the tile has no linked continuum)
// four waves per SIMD asked for where no linked continuum is carried: <0,6> then fits 128 VGPRs without scratch (140 before);
// the time did not change (profiles/r04/ab_epilogue_waves_per_simd.txt): the kernel runs beside the sweeps, whose waves hold the registers
// (the instances for tile widths other than twelve wavelengths, NPC = 0: one wave less -- they spilled 12 - 100 bytes per lane)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((NLC == 0 ? 4 : 3) - (NPC == 0 ? 1 : 0)))) k_fast_gamma_cols(const FastParams f)
{
    extern __shared__ double sm[];
    fast_gamma_cols_rows<NLC, NPC, 4>(f, f.fast_tiles[blockIdx.y], (long)blockIdx.x * 4 * LSX_FGC_ROWS, (long)f.ncol * f.Nspace, sm);   // lsx_fast.h
}

// the big-set instances (lsx_fast.h): workgroups of two waves
template <int NLC>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(NLC == 2 ? 1 : 2))) k_fast_gamma_cols_big(const FastParams f)
{
    extern __shared__ double sm[];
    fast_gamma_cols_rows<NLC, 6, 2, true>(f, f.fast_tiles[blockIdx.y], (long)blockIdx.x * 2 * LSX_FGC_ROWS, (long)f.ncol * f.Nspace, sm);
}

// singular system at (column, depth) = gid, atom: the flag keeps the FIRST one in (column, depth, atom) order --
// one order-independent 64-bit atomicMax of (2^48 - key), 0 = none -- for lsx_last_error (cf. LinAlgError, rh_method.py:739)
#define LSX_SING_BASE (1ull << 48)
__device__ __forceinline__ void flag_singular(unsigned long long* flag, long gid, int atom)
{
    atomicMax(flag, LSX_SING_BASE - (((unsigned long long)gid << 8) | (unsigned)atom));
}

__device__ __forceinline__ void atomic_max_nonneg(double* addr, double v)
{
    // for non-negative doubles the IEEE bit pattern orders like the value
    atomicMax(reinterpret_cast<unsigned long long*>(addr),
              static_cast<unsigned long long>(__double_as_longlong(fabs(v))));
}

// rh_method.py:710-745.  One thread per (column, depth) of one atom; the Nl x Nl system of
// each thread lives in a thread-private LDS column (A[e][tid]) -- dense LU with partial
// pivoting in the operation order of LAPACK dgetf2/dgetrs.
__global__ void k_stat_equil(const double* __restrict__ Gamma, const double* __restrict__ nTotal, double* __restrict__ n,
                             double* __restrict__ dPcol, unsigned long long* __restrict__ singular, int Nl, int lev_off, int lev2_off,
                             int atom, int Natoms, int NLtot, int NL2tot, int Ns, int ncol, const uint8_t* __restrict__ colmask)
{
    extern __shared__ double sm[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const long gid = (long)blockIdx.x * nt + tid;
    if (gid >= (long)ncol * Ns) return;
    const int col = gid / Ns, k = gid % Ns;
    if (colmask && !colmask[col]) return;
    double* A = sm + tid;                       // A[(i + j*Nl) * nt]
    double* b = sm + (size_t)Nl * Nl * nt + tid; // b[i * nt]
    double* nOld = b + (size_t)Nl * nt;
    double* nk = n + ((size_t)col * NLtot + lev_off) * Ns + k;
    const double* G = Gamma + ((size_t)col * NL2tot + lev2_off) * Ns + k;

    int iE = 0;
    double nmax = nk[0];
    for (int l = 0; l < Nl; ++l) {
        const double v = nk[(size_t)l * Ns];
        nOld[l * nt] = v;
        if (v > nmax) { nmax = v; iE = l; }      // np.argmax: first maximum
    }
    for (int i = 0; i < Nl; ++i)
        for (int j = 0; j < Nl; ++j) A[(i + j * Nl) * nt] = (i == iE) ? 1.0 : G[(size_t)(i * Nl + j) * Ns];
    for (int i = 0; i < Nl; ++i) b[i * nt] = 0.0;
    b[iE * nt] = nTotal[((size_t)col * Natoms + atom) * Ns + k];

    bool sing = false;
    for (int j = 0; j < Nl && !sing; ++j) {
        int pv = j;
        double amax = fabs(A[(j + j * Nl) * nt]);
        for (int i = j + 1; i < Nl; ++i) {
            const double v = fabs(A[(i + j * Nl) * nt]);
            if (v > amax) { amax = v; pv = i; }
        }
        if (A[(pv + j * Nl) * nt] == 0.0 || amax != amax) { sing = true; break; }
        if (pv != j) {
            for (int q = 0; q < Nl; ++q) {
                const double t = A[(j + q * Nl) * nt];
                A[(j + q * Nl) * nt] = A[(pv + q * Nl) * nt];
                A[(pv + q * Nl) * nt] = t;
            }
            const double t = b[j * nt]; b[j * nt] = b[pv * nt]; b[pv * nt] = t;
        }
        const double r = 1.0 / A[(j + j * Nl) * nt];
        for (int i = j + 1; i < Nl; ++i) A[(i + j * Nl) * nt] *= r;
        for (int q = j + 1; q < Nl; ++q) {
            const double ajq = A[(j + q * Nl) * nt];
            for (int i = j + 1; i < Nl; ++i) A[(i + q * Nl) * nt] -= A[(i + j * Nl) * nt] * ajq;
        }
    }
    if (sing) { flag_singular(singular, gid, atom); return; }
    for (int j = 0; j < Nl; ++j)
        for (int i = j + 1; i < Nl; ++i) b[i * nt] -= A[(i + j * Nl) * nt] * b[j * nt];
    for (int j = Nl - 1; j >= 0; --j) {
        b[j * nt] /= A[(j + j * Nl) * nt];
        for (int i = 0; i < j; ++i) b[i * nt] -= A[(i + j * Nl) * nt] * b[j * nt];
    }
    double mx = 0.0;
    for (int i = 0; i < Nl; ++i) {
        const double nn = b[i * nt];
        const double ch = fabs(1.0 - nOld[i * nt] / nn);
        mx = (ch != ch || mx != mx) ? __builtin_nan("") : fmax(mx, ch);   // change.max(): NaN wins inside one depth
        nk[(size_t)i * Ns] = nn;
    }
    // maxRelChange = max(maxRelChange, change.max()) with Python's builtin max (rh_method.py:741): a NaN never
    // replaces the running maximum, so a depth whose change is NaN drops out
    if (mx == mx) atomic_max_nonneg(&dPcol[col], mx);
}

// max over the context's columns of the per-column monitors (lsx_monitors): one block, fixed order
__global__ void __launch_bounds__(256)
k_monitors(const double* __restrict__ res, int ncol, double* __restrict__ dst)
{
    __shared__ double sj[256], sp[256];
    __shared__ int sn[256];
    double mj = 0.0, mp = 0.0;
    int nan = 0;
    for (int c = threadIdx.x; c < ncol; c += 256) {
        const double a = res[c], b = res[ncol + c];
        if (a != a) nan = 1; else mj = fmax(mj, a);
        mp = fmax(mp, b);                       // per-column dPops is never NaN (k_stat_equil)
    }
    sj[threadIdx.x] = mj; sp[threadIdx.x] = mp; sn[threadIdx.x] = nan;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) {
            sj[threadIdx.x] = fmax(sj[threadIdx.x], sj[threadIdx.x + h]);
            sp[threadIdx.x] = fmax(sp[threadIdx.x], sp[threadIdx.x + h]);
            sn[threadIdx.x] |= sn[threadIdx.x + h];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        dst[0] = sj[0];
        dst[1] = sp[0];
        dst[2] = sn[0] ? 1.0 : 0.0;
        dst[3] = reinterpret_cast<const unsigned long long*>(res)[2 * (size_t)ncol] ? 1.0 : 0.0;
    }
}

// The same elimination with the system in registers: NL is a compile-time constant, every loop is unrolled and the
// data-dependent row choices (the eliminated row, the pivot) become predicated selects.  Same operations in the same
// order as k_stat_equil => identical results; used for the small atoms (NL <= 8) that stellar problems have.
template <int NL>
__global__ void __launch_bounds__(64)
k_stat_equil_reg(const double* __restrict__ Gamma, const double* __restrict__ nTotal, double* __restrict__ n,
                 double* __restrict__ dPcol, unsigned long long* __restrict__ singular, int lev_off, int lev2_off, int atom, int Natoms,
                 int NLtot, int NL2tot, int Ns, int ncol, const uint8_t* __restrict__ colmask)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)ncol * Ns) return;
    const int col = gid / Ns, k = gid % Ns;
    if (colmask && !colmask[col]) return;
    double* nk = n + ((size_t)col * NLtot + lev_off) * Ns + k;
    const double* G = Gamma + ((size_t)col * NL2tot + lev2_off) * Ns + k;
    double a[NL][NL], b[NL], nOld[NL];           // a[i][j]: row i, column j

    int iE = 0;
    double nmax = nk[0];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        nOld[l] = nk[(size_t)l * Ns];
        if (nOld[l] > nmax) { nmax = nOld[l]; iE = l; }      // np.argmax: first maximum
    }
    const double ntot = nTotal[((size_t)col * Natoms + atom) * Ns + k];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const double g = G[(size_t)(i * NL + j) * Ns];
            a[i][j] = (i == iE) ? 1.0 : g;
        }
        b[i] = (i == iE) ? ntot : 0.0;
    }
    bool sing = false;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        int pv = j;
        double amax = fabs(a[j][j]), apv = a[j][j];
#pragma unroll
        for (int i = j + 1; i < NL; ++i) {
            const double v = fabs(a[i][j]);
            if (v > amax) { amax = v; pv = i; apv = a[i][j]; }
        }
        if (!sing && (apv == 0.0 || amax != amax)) sing = true;
#pragma unroll
        for (int r = j + 1; r < NL; ++r) {
            const bool sw = r == pv;
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                const double t = a[j][q];
                a[j][q] = sw ? a[r][q] : t;
                a[r][q] = sw ? t : a[r][q];
            }
            const double t = b[j];
            b[j] = sw ? b[r] : t;
            b[r] = sw ? t : b[r];
        }
        const double rr = 1.0 / a[j][j];
#pragma unroll
        for (int i = j + 1; i < NL; ++i) a[i][j] *= rr;
#pragma unroll
        for (int q = j + 1; q < NL; ++q) {
            const double ajq = a[j][q];
#pragma unroll
            for (int i = j + 1; i < NL; ++i) a[i][q] -= a[i][j] * ajq;
        }
    }
    if (sing) { flag_singular(singular, gid, atom); return; }
#pragma unroll
    for (int j = 0; j < NL; ++j)
#pragma unroll
        for (int i = j + 1; i < NL; ++i) b[i] -= a[i][j] * b[j];
#pragma unroll
    for (int j = NL - 1; j >= 0; --j) {
        b[j] /= a[j][j];
#pragma unroll
        for (int i = 0; i < j; ++i) b[i] -= a[i][j] * b[j];
    }
    double mx = 0.0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const double ch = fabs(1.0 - nOld[i] / b[i]);
        mx = (ch != ch || mx != mx) ? __builtin_nan("") : fmax(mx, ch);   // change.max(): NaN wins inside one depth
        nk[(size_t)i * Ns] = b[i];
    }
    if (mx == mx) atomic_max_nonneg(&dPcol[col], mx);          // builtin max over depths drops NaN (rh_method.py:741)
}

// FETCH_SIZE calibration (profiles/calibrate.py): read `rows` x `seg` doubles exactly once in the sweep
// kernel's access shape -- one wave-instruction = 64/seg segments of `seg` consecutive doubles (8 B per
// lane), the segments `stride` doubles apart -- and fold them into one value per wave.
__global__ void k_calib_read(const double* __restrict__ buf, double* __restrict__ out, int seg, long stride, long nwave_rows)
{
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int per = 64 / seg;
    const int r = lane / seg, j = lane - r * seg;
    double acc = 0.0;
    if (r < per)
        for (long q = 0; q < nwave_rows; ++q) acc += buf[((wave * nwave_rows + q) * per + r) * stride + j];
    if (acc == 1.2345e300) out[wave] = acc; // keep the loads alive
}

// formal_solver.py:14-212 for independent rays (one ray per thread, [ray][k] layout).  w2 is the sweep kernel's own
// device function (lsx_dev.h: table-driven exp); the divisions are IEEE here (this entry point is not hot).
__device__ __forceinline__ double dev_planck(double temp, double wav)
{
    const double hc_Tkla = kHC / (kKBoltzmann * kNM_TO_M * wav) / temp;
    const double x = kNM_TO_M * wav;
    return (2.0 * kHC) / (x * x * x) / (exp(hc_Tkla) - 1.0);
}
// Istart == nullptr: boundary condition of piecewise_linear_1d (formal_solver.py:203-209, needs T and wav);
// otherwise piecewise_1d_impl with the incident intensity handed over (formal_solver.py:46-142)
__global__ void __launch_bounds__(64)
k_piecewise(int nray, int Ns, const double* __restrict__ z, const double* __restrict__ T, const double* __restrict__ mu,
            const int* __restrict__ to_obs, const double* __restrict__ wav, const double* __restrict__ Istart,
            const double* __restrict__ chi, const double* __restrict__ S, double* __restrict__ I, double* __restrict__ Psi,
            const double* __restrict__ exp2_tab)
{
    __shared__ double etab_s[LSX_EXP_TAB];
    for (int e = threadIdx.x; e < LSX_EXP_TAB; e += blockDim.x) etab_s[e] = exp2_tab[e];
    __syncthreads();
    const lds_f64* etab = (const lds_f64*)etab_s;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nray) return;
    const double* c = chi + (size_t)r * Ns;
    const double* s = S + (size_t)r * Ns;
    double* Io = I + (size_t)r * Ns;
    double* Po = Psi + (size_t)r * Ns;
    const double zmu = 1.0 / mu[r];
    const int up = to_obs[r];
    const int dk = up ? -1 : 1, kS = up ? Ns - 1 : 0, kE = up ? 0 : Ns - 1;
    double dtau = 0.5 * (c[kS] + c[kS + dk]) * zmu * fabs(z[kS] - z[kS + dk]);
    double Iu = 0.0;
    if (Istart) {
        Iu = Istart[r];
    } else if (up) {
        const double dt0 = zmu * (c[kS] + c[kS + dk]) * 0.5 * fabs(z[kS] - z[kS + dk]);
        const double B0 = dev_planck(T[Ns - 2], wav[r]), B1 = dev_planck(T[Ns - 1], wav[r]);
        Iu = B1 - (B0 - B1) / dt0;
    }
    double dS = (s[kS] - s[kS + dk]) / dtau;
    Io[kS] = Iu;
    Po[kS] = 0.0;
    double w0 = 0.0, w1 = 0.0;
    for (int k = kS + dk; k != kE; k += dk) {
        w2(dtau, w0, w1, etab);
        const double Ik = Iu * (1.0 - w0) + w0 * s[k] + w1 * dS;
        Io[k] = Ik;
        Po[k] = (w0 - w1 / dtau) / c[k];
        const double dt2 = 0.5 * (c[k] + c[k + dk]) * zmu * fabs(z[k] - z[k + dk]);
        dS = (s[k] - s[k + dk]) / dt2;
        dtau = dt2;
        Iu = Ik;
    }
    Io[kE] = (1.0 - w0) * Iu + w0 * s[kE - dk] + w1 * dS; // stale w, S[kE-dk]: formal_solver.py:138
    Po[kE] = (w0 - w1 / dtau) / c[kE];
}

// N4 (include/lsx.h): monotonic piecewise-parabolic short characteristics for independent rays, one ray per thread;
// one point of the recurrence as a device function the sweep shares
__global__ void __launch_bounds__(64)
k_piecewise_parabolic(int nray, int Ns, const double* __restrict__ z, const double* __restrict__ mu, const int* __restrict__ to_obs,
                      const double* __restrict__ Istart, const double* __restrict__ chi, const double* __restrict__ S,
                      double* __restrict__ I, double* __restrict__ Psi, const double* __restrict__ exp2_tab)
{
    __shared__ double etab_s[LSX_EXP_TAB];
    for (int e = threadIdx.x; e < LSX_EXP_TAB; e += blockDim.x) etab_s[e] = exp2_tab[e];
    __syncthreads();
    const lds_f64* etab = (const lds_f64*)etab_s;
    const int r0 = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = r0 < nray ? r0 : nray - 1;                   // every lane walks a ray (w3 is wave-wide); the spare ones store nothing
    const double* c = chi + (size_t)r * Ns;
    const double* s = S + (size_t)r * Ns;
    double* Io = I + (size_t)r * Ns;
    double* Po = Psi + (size_t)r * Ns;
    const double zmu = 1.0 / mu[r];
    const int up = to_obs[r];
    const int dk = up ? -1 : 1, kS = up ? Ns - 1 : 0, kE = up ? 0 : Ns - 1;
    double Iu = Istart[r];
    if (r0 < nray) { Io[kS] = Iu; Po[kS] = 0.0; }
    for (int k = kS + dk;; k += dk) {
        const bool has_d = k != kE;
        const double dtau_u = 0.5 * (c[k - dk] + c[k]) * zmu * fabs(z[k - dk] - z[k]);
        const double dtau_d = has_d ? 0.5 * (c[k] + c[k + dk]) * zmu * fabs(z[k] - z[k + dk]) : 1.0;
        const Para p = parabolic_point(Iu, s[k - dk], s[k], has_d ? s[k + dk] : 0.0, dtau_u, dtau_d, has_d, etab);
        if (r0 < nray) { Io[k] = p.I; Po[k] = p.Lam / c[k]; }
        Iu = p.I;
        if (!has_d) break;
    }
}
__global__ void __launch_bounds__(64)
k_w3(int n, const double* __restrict__ dtau, double* __restrict__ out, const double* __restrict__ exp2_tab)
{
    __shared__ double etab_s[LSX_EXP_TAB];
    for (int e = threadIdx.x; e < LSX_EXP_TAB; e += blockDim.x) etab_s[e] = exp2_tab[e];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double x = dtau[i < n ? i : n - 1];
    double w0, w1, w2q;
    w3(x, w0, w1, w2q, (const lds_f64*)etab_s);
    if (i < n) { out[3 * i] = w0; out[3 * i + 1] = w1; out[3 * i + 2] = w2q; }
}

// formal_solver.py:14-44 on an array: the sweep's w2, one value per lane
__global__ void __launch_bounds__(64)
k_w2(int n, const double* __restrict__ dtau, double* __restrict__ out, const double* __restrict__ exp2_tab)
{
    __shared__ double etab_s[LSX_EXP_TAB];
    for (int e = threadIdx.x; e < LSX_EXP_TAB; e += blockDim.x) etab_s[e] = exp2_tab[e];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double x = dtau[i < n ? i : n - 1];
    double w0, w1;
    w2(x, w0, w1, (const lds_f64*)etab_s);
    if (i < n) { out[2 * i] = w0; out[2 * i + 1] = w1; }
}

// 2^(j/64) as head + tail (extended precision on the host): the table exp_tab64 (lsx_dev.h) reads
std::vector<double> make_exp2_table()
{
    std::vector<double> et(LSX_EXP_TAB);
    for (int j = 0; j < 64; ++j) {
        const long double v = exp2l((long double)j / 64.0L);
        et[2 * j] = (double)v;
        et[2 * j + 1] = (double)(v - (long double)et[2 * j]);
    }
    return et;
}

} // namespace

// ------------------------------------------------------------------------------- context
namespace {

int launch_tiles_pack(lsx_ctx* c, const double* in, double* out, int B, bool unpack)
{
    const int total = (int)c->til_col;
    dim3 grid((total + 255) / 256, B);
    hipLaunchKernelGGL(k_tiles_pack, grid, dim3(256), 0, c->stream, in, out, c->d_tiles, (int)c->tiles.size(), c->L, c->Nspace,
                       c->Nspect, unpack);
    HIPCHK(hipGetLastError());
    return LSX_OK;
}


} // namespace

namespace lsxd {

// what depends on (nStar, temperature) of columns [cc, cc + nb): the two factors of the continuum g_ij
// (rh_method.py:453-454) -- the Boltzmann factor per (wavelength, depth) and the nStar ratio per (continuum, depth).
// Enqueued on the context's stream.
int rebuild_derived(lsx_ctx* c, size_t cc, size_t nb)
{
    c->optab_fresh = false;
    const int Ns = c->Nspace;
    if (c->d_E) {
        dim3 grid((unsigned)((c->til_col + 255) / 256), (unsigned)nb);
        hipLaunchKernelGGL(k_build_E, grid, dim3(256), 0, c->stream, c->d_temperature + cc * Ns, c->d_wavelength,
                           c->d_E + cc * c->til_col, c->d_tiles, (int)c->tiles.size(), c->L, Ns, c->d_exp2_tab);
        HIPCHK(hipGetLastError());
    }
    if (c->d_nsr) {
        dim3 grid((c->Ncont * Ns + 255) / 256, (unsigned)nb);
        hipLaunchKernelGGL(k_build_nsr, grid, dim3(256), 0, c->stream, c->d_nStar + cc * c->NLtot * Ns,
                           c->d_nsr + cc * c->Ncont * Ns, c->d_cont_li, c->d_cont_lj, c->Ncont, (int)Ns, c->NLtot);
        HIPCHK(hipGetLastError());
    }
    return LSX_OK;
}

// compute_phi (rh_method.py:198-243) for columns [cc, cc + nb) from DEVICE arrays dA [nb][Nlines][Ns], dV [nb][Natoms][Ns],
// dL [nb][Ns] or null: the sweep's (tile, line) profile blocks and the normalisation wphi.  Enqueued on the context's stream.
int profiles_from_device(lsx_ctx* c, size_t cc, size_t nb, const double* dA, const double* dV, const double* dL)
{
    c->optab_fresh = false;
    const int Ns = c->Nspace;
    if (!c->d_voigt_w) {
        std::vector<double> W(56);
        for (int g = 0; g < 2; ++g)
            for (int n = -14; n <= 13; ++n) { const double x = (n + 0.5 * g) * 0.5; W[g * 28 + n + 14] = std::exp(-x * x); }
        int rc = upload(&c->d_voigt_w, W, c->stream);
        if (rc) return rc;
    }
    VoigtParams q{};
    q.Ns = Ns; q.Nlines = c->Nlines; q.Natoms = c->Natoms; q.wavelength = c->d_wavelength; q.muz = c->d_muz; q.wmu = c->d_wmu;
    q.W = c->d_voigt_w; q.aDamp = dA; q.vBroad = dV; q.vlos = dL;
    // the profile blocks of the sweep, one launch per (tile, line)
    const int R = c->phi_compact ? 1 : c->Nrays, D = c->phi_compact ? 1 : 2;
    q.Nrays = R; q.ndir = D;
    for (const DevSlot& sl : c->slots) {
        if (!(sl.flags & SLOT_LINE) || sl.len <= 0) continue;
        const DevTrans& h = c->htrans[sl.trans];
        const size_t total = (size_t)sl.len * R * D * Ns;
        dim3 grid((unsigned)std::min<size_t>((total + 255) / 256, 256), (unsigned)nb);
        hipLaunchKernelGGL(k_voigt_block, grid, dim3(256), 0, c->stream, q, PhiBlock{c->d_phi, c->phi_col, cc, (size_t)sl.base, c->phi_group},
                           sl.first, sl.len, h.line_idx, h.atom, h.lambda0);
        HIPCHK(hipGetLastError());
    }
    // the normalisation of every line
    q.Nrays = c->Nrays; q.ndir = 2;
    for (int t = 0; t < c->Ntrans; ++t) {
        const DevTrans& h = c->htrans[t];
        if (!h.is_line) continue;
        const long nth = (long)nb * Ns;
        hipLaunchKernelGGL(k_voigt_wphi, dim3((unsigned)((nth + 63) / 64)), dim3(64), 0, c->stream, q,
                           c->d_wphi + cc * c->Nlines * Ns, (int)nb, h.Nblue, h.Nlam, h.line_idx, h.atom, h.lambda0);
        HIPCHK(hipGetLastError());
    }
    return LSX_OK;
}

void mark_profiles_set(lsx_ctx* c, size_t col0, size_t ncol)
{
    c->optab_fresh = false;
    for (size_t q = 0; q < ncol; ++q) {
        c->n_phi_set += (size_t)1 - c->phi_set[col0 + q];
        c->phi_set[col0 + q] = 1;
    }
}

} // namespace lsxd

extern "C" {

const char* lsx_last_error(void) { return lsxd::g_err.c_str(); }
const char* lsx_backend_name(void) { return "hip-gfx950"; }
int32_t lsx_abi_version(void) { return LSX_ABI_VERSION; }
// lsx_build_id(): build/lsx_build_id.cpp, written by the Makefile from a hash of the sources

void lsx_destroy(lsx_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void* ptrs[] = {c->d_wavelength, c->d_zmu, c->d_wmuh, c->d_wl, c->d_alpha, c->d_u_la, c->d_active, c->d_trans,
                    c->d_tiles, c->d_slots, c->d_tile_slots, c->d_fin_ptr, c->d_fin_idx, c->d_atom_ptr, c->d_atom_slots, c->d_Nlevel, c->d_lev2_off, c->d_height,
                    c->d_temperature, c->d_nStar, c->d_nTotal, c->d_n, c->d_C, c->d_Gamma, c->d_wphi, c->d_bgchi, c->d_bgce, c->d_bgxce,
                    c->d_bgeta, c->d_sca, c->d_phi, c->d_E, c->d_corr, c->d_Psi3, c->d_J[0], c->d_J[1], c->d_I, c->d_Gpart, c->d_dJpart,
                    c->d_res, c->d_stage, c->d_debug, c->d_colmask, c->d_bgxchi, c->d_bgxeta, c->d_Psi2, c->d_fast_tiles, c->d_fast_rest, c->d_nsr, c->d_cont_li, c->d_cont_lj, c->d_exp2_tab, c->d_voigt_w, c->d_muz, c->d_wmu, c->d_optab, c->d_trans_row, c->d_fgtab, c->d_level_atom,
                    c->d_sa_atoms, c->d_sa_lines, c->d_sa_colls, c->d_sa_spl, c->d_sa_levE, c->d_sa_levg, c->d_sa_levnD, c->d_sa_levdZ,
                    c->d_vBroad, c->d_aDamp};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (int v = 0; v < LSX_FGC_LISTS; ++v) if (c->d_fast_cols[v]) (void)hipFree(c->d_fast_cols[v]);
    for (auto& k : c->classes) {
        if (k.d_tiles) (void)hipFree(k.d_tiles);
        if (k.done) (void)hipEventDestroy(k.done);
        if (k.tdone) (void)hipEventDestroy(k.tdone);
        if (k.d_fast_tiles) (void)hipFree(k.d_fast_tiles);
        for (int v = 0; v < LSX_FGC_LISTS; ++v) if (k.d_fast_cols[v]) (void)hipFree(k.d_fast_cols[v]);
        if (k.d_fast_rest) (void)hipFree(k.d_fast_rest);
        if (k.stream) { (void)hipStreamSynchronize(k.stream); (void)hipStreamDestroy(k.stream); }
    }
    for (auto& g : c->fs_graphs) if (g.second) (void)hipGraphExecDestroy(g.second);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (hipEvent_t e : {c->ev0, c->ev1, c->ev2})
        if (e) (void)hipEventDestroy(e);
    if (c->ev_mon) (void)hipEventDestroy(c->ev_mon);
    for (void* q : {(void*)c->d_I_alt, (void*)c->d_Gamma_alt, (void*)c->d_res_alt})
        if (q) (void)hipFree(q);
    if (c->evA) (void)hipEventDestroy(c->evA);
    if (c->evB) (void)hipEventDestroy(c->evB);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->h_n) (void)hipHostFree(c->h_n);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int lsx_create(const lsx_problem* d, int32_t ncol, int32_t device, void* stream, lsx_ctx** out)
{
    return lsx_create_with_options(d, ncol, device, stream, nullptr, out);
}

int lsx_create_with_options(const lsx_problem* d, int32_t ncol, int32_t device, void* stream, const char* options, lsx_ctx** out)
{
    if (!d || !out || ncol < 1) return fail(LSX_EINVAL, "lsx_create: null argument or ncol < 1");
    if (ncol > 65535) return fail(LSX_EUNSUPPORTED, "lsx_create: at most 65535 columns per context (the column is a grid dimension); use several contexts");
    // ---- the plan (lsx_plan.cpp, host only): descriptor checks, transition tables, tile schedule, slot table, sweep classes,
    // strides, LDS sizes and launch shapes.  Nothing below can fail on the problem's shape any more.
    // Switches: the LSX_* environment variables are diagnostic DEFAULTS, read here only (lsx_plan.cpp, options_from_env); an explicit
    // `options` list overrides them entry by entry; what the context ended up with is reported by lsx_effective_options.
    CtxOptions copt;
    options_from_env(&copt);
    {
        std::string oerr;
        const int orc = options_apply(options, &copt, &oerr);
        if (orc) return fail(orc, "lsx_create_with_options: %s", oerr.c_str());
    }
    const PlanOptions& opt = copt.plan;
    LsxPlan plan;
    {
        std::string perr;
        const int prc = plan_build(d, opt, &plan, &perr);
        if (prc) return fail(prc, "%s", perr.c_str());
    }
    if ((long)plan.tiles.size() * ncol > 0x7fffffffL) return fail(LSX_EUNSUPPORTED, "lsx_create: %zu tiles x %d columns exceed the grid", plan.tiles.size(), ncol);
    if ((long)d->Nspace * ncol > 0x7fffffffL) return fail(LSX_EUNSUPPORTED, "lsx_create: %d depths x %d columns exceed 32-bit row indices", d->Nspace, ncol);
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(LSX_EDEVICE, "lsx_create: no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(LSX_EINVAL, "lsx_create: device %d out of range (%d visible)", device, ndev);
    HIPCHK(hipSetDevice(device));

    lsx_ctx* c = new lsx_ctx();
    static_cast<LsxPlan&>(*c) = std::move(plan);
    for (const PlanClass& pc : c->plan_classes) {
        c->classes.push_back(SweepClass());
        static_cast<PlanClass&>(c->classes.back()) = pc;
    }
    c->device = device;
    c->ncol = ncol;
    if (stream) {
        c->stream = reinterpret_cast<hipStream_t>(stream);
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(LSX_EDEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); }
        c->own_stream = true;
    }
    c->options = copt;
    c->opt_se_lds = copt.run.se_lds;
    c->opt_trace_classes = copt.run.trace_classes;
    if (c->opt_trace_classes) {
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        fprintf(stderr, "lsx_create: GPU_MAX_HW_QUEUES=%s at this point (%zu tile classes, each on a stream of its own; the runtime read the "
                        "variable at the process's first HIP call)\n", q ? q : "(unset: the runtime's default of 4)", c->plan_classes.size());
    }
    c->opt_serial = copt.run.serial;                    // every class on the context's stream, one after the other
    c->opt_finish_big = copt.run.finish_big;            // the many-column Gamma epilogue also for small batches
    c->opt_abl_fast = copt.run.abl_fast;
    c->opt_fused_epilogue = copt.run.fused_epilogue;
    c->opt_graph = copt.run.graph;
    c->opt_no_fused_fast = copt.run.no_fused_fast;      // small batches: the fast-continuum kernels as launches of their own (tests)
    const int Ns = c->Nspace, Nspect = c->Nspect;
    double work_total = 0.0, work_seen = 0.0;
    for (auto& k : c->classes) work_total += k.work;

    // ---- uploads of the column independent tables
    int rc = LSX_OK;
#define TRY(x) do { rc = (x); if (rc) { lsx_destroy(c); return rc; } } while (0)
    TRY(upload(&c->d_wavelength, c->wave, c->stream));
    TRY(upload(&c->d_zmu, c->zmu, c->stream));
    {
        std::vector<double> muzv(d->muz, d->muz + c->Nrays), wmuv(d->wmu, d->wmu + c->Nrays);
        TRY(upload(&c->d_muz, muzv, c->stream));
        TRY(upload(&c->d_wmu, wmuv, c->stream));
    }
    TRY(upload(&c->d_wmuh, c->wmuh, c->stream));
    TRY(upload(&c->d_wl, c->wl, c->stream));
    TRY(upload(&c->d_alpha, c->alpha, c->stream));
    TRY(upload(&c->d_u_la, c->u_la, c->stream));
    TRY(upload(&c->d_exp2_tab, make_exp2_table(), c->stream));
    TRY(upload(&c->d_active, c->active, c->stream));
    TRY(upload(&c->d_trans, c->htrans, c->stream));
    TRY(upload(&c->d_tiles, c->tiles, c->stream));
    {   // the epilogue's copy of the slot table carries "fast continuum" in bit 30: such a slot's slabs hold both directions in
        // their first entry, the second is neither written nor read
        std::vector<int> enc(c->tile_slots);
        for (size_t u = 0; u < enc.size(); ++u) enc[u] |= c->tile_slot_fast[u] ? (1 << 30) : 0;
        TRY(upload(&c->d_tile_slots, enc, c->stream));
    }
    {   // the slabs behind each Gamma entry, in slot order (k_gamma_finish_small): ij <- slabs 0, 1; ji <- slabs 2, 3; a fast
        // continuum has one slab per entry
        std::vector<std::vector<int>> lists((size_t)c->NL2tot);
        for (size_t u = 0; u < c->tile_slots.size(); ++u) {
            const DevTrans& tr = c->htrans[(size_t)c->tile_slots[u]];
            const bool fast = c->tile_slot_fast[u];
            lists[(size_t)tr.gam_ij].push_back((int)u * 4 + 0);
            if (!fast) lists[(size_t)tr.gam_ij].push_back((int)u * 4 + 1);
            lists[(size_t)tr.gam_ji].push_back((int)u * 4 + 2);
            if (!fast) lists[(size_t)tr.gam_ji].push_back((int)u * 4 + 3);
        }
        std::vector<int> ptr((size_t)c->NL2tot + 1, 0), idx;
        for (int e = 0; e < c->NL2tot; ++e) {
            idx.insert(idx.end(), lists[(size_t)e].begin(), lists[(size_t)e].end());
            ptr[(size_t)e + 1] = (int)idx.size();
        }
        if (idx.empty()) idx.push_back(0);
        TRY(upload(&c->d_fin_ptr, ptr, c->stream));
        {   // k_gamma_finish_levels: [NLtot] the atom of every level, then [Natoms] each atom's first level
            std::vector<int> lt;
            for (int a = 0; a < c->Natoms; ++a) for (int l = 0; l < c->Nlevel[a]; ++l) lt.push_back(a);
            for (int a = 0; a < c->Natoms; ++a) lt.push_back(c->lev_off[a]);
            TRY(upload(&c->d_level_atom, lt, c->stream));
        }
        TRY(upload(&c->d_fin_idx, idx, c->stream));
        // k_gamma_finish: the slots of each atom's transitions, in slot order
        std::vector<int> aptr((size_t)c->Natoms + 1, 0), aslots;
        for (int a = 0; a < c->Natoms; ++a) {
            for (size_t u = 0; u < c->tile_slots.size(); ++u)
                if (c->htrans[(size_t)c->tile_slots[u]].atom == a) aslots.push_back((int)u);
            aptr[(size_t)a + 1] = (int)aslots.size();
        }
        if (aslots.empty()) aslots.push_back(0);
        TRY(upload(&c->d_atom_ptr, aptr, c->stream));
        TRY(upload(&c->d_atom_slots, aslots, c->stream));
    }
    TRY(upload(&c->d_slots, c->slots, c->stream));
    {   // the column-mapped fast-continuum epilogue's tables, one image of its LDS area per tile it takes (lsx_fast.h, fast_gamma_cols_rows):
        // plain copies of alpha / wl / u_la entries, zero where a continuum or line is not active at the wavelength
        const int L = c->L;
        std::vector<double> tab(c->tiles.size() * (size_t)LSX_FGC_TAB(L), 0.0);
        for (size_t t = 0; t < c->tiles.size(); ++t) {
            const DevTile& tl = c->tiles[t];
            if (tl.nF == 0 || tl.fast_simple < 2) continue;
            double* T = tab.data() + t * (size_t)LSX_FGC_TAB(L);
            const DevSlot* fs = c->slots.data() + tl.slot0 + tl.nP;
            const DevSlot* ls = c->slots.data() + tl.slot0;
            auto on = [&](const DevSlot& sl, int jj, int* lt) {
                const int lq = tl.la0 + std::min(jj, tl.nla - 1);
                *lt = lq - sl.Nblue;
                return jj < tl.nla && *lt >= 0 && *lt < sl.Nlam && c->active[(size_t)sl.trans * c->Nspect + lq] != 0;
            };
            for (int q = 0; q < tl.nF; ++q)
                for (int jj = 0; jj < L; ++jj) {
                    int lt;
                    const bool a = on(fs[q], jj, &lt);
                    T[(q * L + jj) * 2 + 0] = a ? c->alpha[fs[q].wl_off + lt] : 0.0;
                    T[(q * L + jj) * 2 + 1] = a ? c->wl[fs[q].wl_off + lt] : 0.0;
                }
            double* U = T + 2 * LSX_FGC_MAXF_BIG * L;
            for (int jj = 0; jj < L; ++jj) U[jj] = c->u_la[tl.la0 + std::min(jj, tl.nla - 1)];
            for (int jj = 0; jj < L; ++jj)      // (lsx_dev.h, boltzmann_lane_constant: the same expression, the same bits as the sweep's lanes)
                U[3 * L + jj] = -(6.6260755E-34 * 2.99792458E+08 / (1.380658E-23 * 1.0E-09)) / c->wave[tl.la0 + std::min(jj, tl.nla - 1)];
            const int nlc = tl.nK > 0 ? std::min(tl.nL, 2) : 0;
            for (int u = 0; u < nlc; ++u)
                for (int jj = 0; jj < L; ++jj) {
                    int lt;
                    if (on(ls[u], jj, &lt)) U[L + u * L + jj] = c->wl[ls[u].wl_off + lt];
                }
        }
        TRY(upload(&c->d_fgtab, tab, c->stream));
    }
    TRY(upload(&c->d_Nlevel, c->Nlevel, c->stream));
    TRY(upload(&c->d_lev2_off, c->lev2_off, c->stream));
    // the fork / join events order kernels of THIS device only: a device-scope release when they are recorded, not the default
    // system-scope fence (cache write-back and invalidation before every class starts and before the epilogue);
    // LSX_SYSTEM_EVENTS=1: the default events (a measured alternative)
    const unsigned kDevEvent = hipEventDisableTiming | (getenv("LSX_SYSTEM_EVENTS") ? 0u : (unsigned)hipEventReleaseToDevice);
    for (auto& k : c->classes) {
        TRY(upload(&k.d_tiles, k.tiles, c->stream));
        if (!k.fast_tiles.empty()) TRY(upload(&k.d_fast_tiles, k.fast_tiles, c->stream));
        for (int v = 0; v < LSX_FGC_LISTS; ++v) if (!k.fast_cols[v].empty()) TRY(upload(&k.d_fast_cols[v], k.fast_cols[v], c->stream));
        if (!k.fast_rest.empty()) TRY(upload(&k.d_fast_rest, k.fast_rest, c->stream));
        int prio_lo = 0, prio_hi = 0;       // numerically lower = higher priority
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        // workgroups that need many registers (three or more slots, generic) find room only while the machine is not
        // yet full of small ones: they go first; the light classes fill in behind them
        // ... and so does a class that carries a pre-pass -> sweep -> epilogue chain: a serial chain must not start last
        // the classes that make up the first half of the work (in launch order) get the highest priority, the next third the
        // middle one
        const int want = work_seen < 0.5 * work_total ? 0 : (work_seen < 0.85 * work_total ? 1 : 2);
        work_seen += k.work;
        // (measured on MI355X with the round-2 kernels: equal priorities are as fast or faster on both workloads -- every class
        // is bound by vector issue, so there is no idle resource for a favoured class to pick up; LSX_PRIO=1 restores the tiers)
        const int prio = getenv("LSX_PRIO") ? std::min(prio_lo, prio_hi + want) : prio_lo;
        if (c->opt_trace_classes) fprintf(stderr, "class npt=%d nl=%d: stream priority %d (range %d .. %d)\n", k.npt, k.nl, prio, prio_hi, prio_lo);
        if (hipStreamCreateWithPriority(&k.stream, hipStreamNonBlocking, prio) != hipSuccess || hipEventCreateWithFlags(&k.done, kDevEvent) != hipSuccess) { lsx_destroy(c); return fail(LSX_EDEVICE, "class stream"); }
    }
    if (hipEventCreateWithFlags(&c->ev_fork, kDevEvent) != hipSuccess) { lsx_destroy(c); return fail(LSX_EDEVICE, "hipEventCreate"); }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->ev2) != hipSuccess) { lsx_destroy(c); return fail(LSX_EDEVICE, "hipEventCreate"); }
    if (hipEventCreate(&c->evA) != hipSuccess || hipEventCreate(&c->evB) != hipSuccess) { lsx_destroy(c); return fail(LSX_EDEVICE, "hipEventCreate"); }

    // ---- per-column storage
    const size_t nc = ncol;
    TRY(dmalloc(&c->d_height, nc * Ns));
    TRY(dmalloc(&c->d_temperature, nc * Ns));
    TRY(dmalloc(&c->d_nStar, nc * c->NLtot * Ns));
    TRY(dmalloc(&c->d_nTotal, nc * c->Natoms * Ns));
    TRY(dmalloc(&c->d_n, nc * c->NLtot * Ns));
    TRY(dmalloc(&c->d_C, nc * c->NL2tot * Ns));
    TRY(dmalloc(&c->d_Gamma, nc * c->NL2tot * Ns));
    TRY(dmalloc(&c->d_wphi, nc * c->Nlines * Ns));
    TRY(dmalloc(&c->d_bgchi, nc * c->til_col));
    TRY(dmalloc(&c->d_bgeta, nc * c->til_col));
    if (LSX_BG_PAIRS && c->rs_ok) TRY(dmalloc(&c->d_bgce, 2 * nc * c->til_col));      // the same two arrays as pairs, for the folded ray-serial instances
    TRY(dmalloc(&c->d_sca, nc * c->sca_col));
    // (whole column groups: the store interleaves the columns of a group, lsx_dev.h phi_elem)
    const size_t nc_phi = (nc + c->phi_group - 1) / c->phi_group * c->phi_group;
    TRY(dmalloc(&c->d_phi, nc_phi * c->phi_col));
    if (c->phi_col) (void)hipMemsetAsync(c->d_phi, 0, nc_phi * c->phi_col * sizeof(double), c->stream);
    if (c->any_cont) TRY(dmalloc(&c->d_E, nc * c->til_col));
    if (c->corr_col) {
        TRY(dmalloc(&c->d_corr, nc * c->corr_col));
        (void)hipMemsetAsync(c->d_corr, 0, nc * c->corr_col * 8, c->stream);
        TRY(dmalloc(&c->d_Psi3, 2 * nc * c->pp_col));
        (void)hipMemsetAsync(c->d_Psi3, 0, 2 * nc * c->pp_col * 8, c->stream);
    }
    TRY(dmalloc(&c->d_J[0], nc * c->til_col));
    TRY(dmalloc(&c->d_J[1], nc * c->til_col));
    TRY(dmalloc(&c->d_I, nc * Nspect * c->Nrays));
    TRY(dmalloc(&c->d_Gpart, nc * c->tile_slots.size() * 4 * Ns));
    TRY(dmalloc(&c->d_dJpart, nc * 2 * c->tiles.size()));
    // per-column convergence monitors and the singular flag in one block [dJcol | dPcol | flag]: lsx_sync brings
    // it back with one copy and takes the maxima on the host (no reduction kernels on the stream)
    TRY(dmalloc(&c->d_res, 2 * nc + 1));
    c->d_dJcol = c->d_res;
    c->d_dPcol = c->d_res + nc;
    c->d_singular = reinterpret_cast<unsigned long long*>(c->d_res + 2 * nc);
    c->phi_set.assign(nc, 0);
    TRY(dmalloc(&c->d_debug, 1024 * 16));
    if (c->any_cont) {          // the nStar ratio of every continuum (g_ij = ratio x E_T)
        TRY(upload(&c->d_cont_li, c->cont_li, c->stream));
        TRY(upload(&c->d_cont_lj, c->cont_lj, c->stream));
        TRY(dmalloc(&c->d_nsr, nc * c->Ncont * Ns));
    }
    if (!c->fast_tiles.empty()) {
        TRY(upload(&c->d_fast_tiles, c->fast_tiles, c->stream));
        for (int v = 0; v < LSX_FGC_LISTS; ++v) if (!c->fast_cols[v].empty()) TRY(upload(&c->d_fast_cols[v], c->fast_cols[v], c->stream));
        if (!c->fast_rest.empty()) TRY(upload(&c->d_fast_rest, c->fast_rest, c->stream));
        TRY(dmalloc(&c->d_bgxchi, nc * c->til_col));
        TRY(dmalloc(&c->d_bgxeta, nc * c->til_col));
        if (c->d_bgce) TRY(dmalloc(&c->d_bgxce, 2 * nc * c->til_col));
        TRY(dmalloc(&c->d_Psi2, 2 * nc * c->til_col));
        (void)hipMemsetAsync(c->d_Psi2, 0, 2 * nc * c->til_col * 8, c->stream);
    }
    (void)hipMemsetAsync(c->d_debug, 0, 1024 * 16 * 8, c->stream);
#undef TRY
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_pinned), (2 * nc + 1) * sizeof(double), hipHostMallocDefault) != hipSuccess) {
        lsx_destroy(c);
        return fail(LSX_EDEVICE, "hipHostMalloc failed");
    }
    (void)hipMemsetAsync(c->d_J[0], 0, nc * c->til_col * 8, c->stream);
    (void)hipMemsetAsync(c->d_J[1], 0, nc * c->til_col * 8, c->stream);
    (void)hipMemsetAsync(c->d_I, 0, nc * Nspect * c->Nrays * 8, c->stream);
    (void)hipMemsetAsync(c->d_Gamma, 0, nc * c->NL2tot * Ns * 8, c->stream);
    (void)hipMemsetAsync(c->d_res, 0, (2 * nc + 1) * 8, c->stream);
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = c;
    return LSX_OK;
}

int lsx_set_columns(lsx_ctx* c, int32_t col0, int32_t ncol, const lsx_columns* s)
{
    if (c) c->optab_fresh = false;
    if (c) c->spec_valid = false;         // new inputs: a speculative formal solution can no longer be discarded
    if (!c || !s || col0 < 0 || ncol < 1 || col0 + ncol > c->ncol) return fail(LSX_EINVAL, "lsx_set_columns: bad range");
    if (!s->height || !s->temperature || !s->nStar || !s->nTotal || !s->n || !s->C || !s->bg_chi || !s->bg_eta || !s->bg_sca ||
        (c->Nlines && ((s->phi == nullptr) != (s->wphi == nullptr))))
        return fail(LSX_EINVAL, "lsx_set_columns: null array pointer");
    const bool have_phi = c->Nlines && s->phi;     // else: lsx_set_line_profiles follows
    HIPCHK(hipSetDevice(c->device));
    const int Ns = c->Nspace, Nspect = c->Nspect;
    const size_t o = col0;
    auto h2d = [&](double* dst, const double* src, size_t per, size_t cnt) -> int {
        HIPCHK(hipMemcpyAsync(dst, src, per * cnt * 8, hipMemcpyHostToDevice, c->stream));
        return LSX_OK;
    };
    int rc;
#define TRY(x) do { rc = (x); if (rc) return rc; } while (0)
    TRY(h2d(c->d_height + o * Ns, s->height, Ns, ncol));
    TRY(h2d(c->d_temperature + o * Ns, s->temperature, Ns, ncol));
    TRY(h2d(c->d_nStar + o * c->NLtot * Ns, s->nStar, (size_t)c->NLtot * Ns, ncol));
    TRY(h2d(c->d_nTotal + o * c->Natoms * Ns, s->nTotal, (size_t)c->Natoms * Ns, ncol));
    TRY(h2d(c->d_n + o * c->NLtot * Ns, s->n, (size_t)c->NLtot * Ns, ncol));
    TRY(h2d(c->d_C + o * c->NL2tot * Ns, s->C, (size_t)c->NL2tot * Ns, ncol));
    if (have_phi) TRY(h2d(c->d_wphi + o * c->Nlines * Ns, s->wphi, (size_t)c->Nlines * Ns, ncol));

    // arrays that change layout go through the staging buffer in sub-chunks (<= 256 MiB of staging)
    const size_t per_col_max = std::max<size_t>({(size_t)Nspect * Ns, c->phi_in_col, (size_t)1});
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(ncol, ((size_t)32 << 20) / per_col_max));
    TRY(ensure_stage(c, chunk * per_col_max));
    for (size_t b0 = 0; b0 < (size_t)ncol; b0 += chunk) {
        const size_t nb = std::min(chunk, (size_t)ncol - b0);
        const size_t cc = o + b0;
        // background opacity / emissivity: [la][k] -> tile-major [tile][k][j]
        TRY(h2d(c->d_stage, s->bg_chi + b0 * Nspect * Ns, (size_t)Nspect * Ns, nb));
        TRY(launch_tiles_pack(c, c->d_stage, c->d_bgchi + cc * c->til_col, (int)nb, false));
        TRY(h2d(c->d_stage, s->bg_eta + b0 * Nspect * Ns, (size_t)Nspect * Ns, nb));
        TRY(launch_tiles_pack(c, c->d_stage, c->d_bgeta + cc * c->til_col, (int)nb, false));
        if (c->d_bgce) {
            const size_t n = nb * c->til_col;
            hipLaunchKernelGGL(k_interleave2, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)), dim3(256), 0, c->stream,
                               c->d_bgchi + cc * c->til_col, c->d_bgeta + cc * c->til_col, c->d_bgce + 2 * cc * c->til_col, n);
            HIPCHK(hipGetLastError());
        }
        if (c->sca_per_lambda) {
            TRY(h2d(c->d_stage, s->bg_sca + b0 * Nspect * Ns, (size_t)Nspect * Ns, nb));
            TRY(launch_tiles_pack(c, c->d_stage, c->d_sca + cc * c->sca_col, (int)nb, false));
        } else {
            TRY(h2d(c->d_sca + cc * Ns, s->bg_sca + b0 * Ns, Ns, nb));
        }
        // line profiles: rows [lt][mu][dir][k] of a line -> one [k][dir][mu][l] block per (tile, line)
        if (have_phi) {
            TRY(h2d(c->d_stage, s->phi + b0 * c->phi_in_col, c->phi_in_col, nb));
            const int R = c->phi_compact ? 1 : c->Nrays, D = c->phi_compact ? 1 : 2;
            for (const DevSlot& sl : c->slots) {
                if (!(sl.flags & SLOT_LINE) || sl.len <= 0) continue;
                const DevTrans& h = c->htrans[sl.trans];
                const size_t in_off = (size_t)h.phi_off * R * D * Ns;
                const size_t total = (size_t)sl.len * R * D * Ns;
                dim3 grid((unsigned)std::min<size_t>((total + 255) / 256, 256), (unsigned)nb);
                hipLaunchKernelGGL(k_pack_phi, grid, dim3(256), 0, c->stream, c->d_stage + in_off, PhiBlock{c->d_phi, c->phi_col, cc, (size_t)sl.base, c->phi_group},
                                   sl.first - h.Nblue, sl.len, R, D, Ns, c->phi_in_col);
                HIPCHK(hipGetLastError());
            }
        }
        TRY(rebuild_derived(c, cc, nb));            // continuum g_ij tables and nStar ratios of these columns
        HIPCHK(hipStreamSynchronize(c->stream)); // the staging buffer is re-used by the next sub-chunk
    }
#undef TRY
    for (int q = 0; q < ncol; ++q) {       // profiles of these columns: handed over, or still to come (lsx_set_line_profiles)
        const uint8_t v = (have_phi || !c->Nlines) ? 1 : 0;
        c->n_phi_set += (size_t)v - c->phi_set[o + q];
        c->phi_set[o + q] = v;
    }
    // J starts at 0 (rh_method.py:562); so do I and the convergence monitors of these columns
    HIPCHK(hipMemsetAsync(c->d_J[c->jcur] + o * c->til_col, 0, (size_t)ncol * c->til_col * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_I + o * Nspect * c->Nrays, 0, (size_t)ncol * Nspect * c->Nrays * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_Gamma + o * c->NL2tot * Ns, 0, (size_t)ncol * c->NL2tot * Ns * 8, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return LSX_OK;
}

static void swap_result_buffers(lsx_ctx* c)
{
    std::swap(c->d_I, c->d_I_alt);
    std::swap(c->d_Gamma, c->d_Gamma_alt);
    std::swap(c->d_res, c->d_res_alt);
    c->d_dJcol = c->d_res;
    c->d_dPcol = c->d_res + c->ncol;
    c->d_singular = reinterpret_cast<unsigned long long*>(c->d_res + 2 * (size_t)c->ncol);
}

// the ray-serial sweeps' operand table (k_build_optab) from the context's current populations, profiles' norms and geometry, on the
// context's stream.  Built where its inputs change inside the MALI loop -- behind every statistical equilibrium, where it runs
// while the host waits for the monitors -- and in front of a formal solution whenever something else has touched them since
// (c->optab_fresh: cleared by every call that writes n, wphi, the nStar ratios, the heights or the scattering coefficient).
static void launch_build_optab(lsx_ctx* c)
{
    OptabParams op{};
    op.Ns = c->Nspace; op.Ntrans = c->Ntrans; op.ncol = c->ncol; op.NLtot = c->NLtot; op.Nlines = c->Nlines; op.Ncont = c->Ncont;
    op.trans = c->d_trans; op.trans_row = c->d_trans_row; op.cont_li = c->d_cont_li; op.cont_lj = c->d_cont_lj;
    op.n = c->d_n; op.wphi = c->d_wphi; op.nsr = c->d_nsr; op.height = c->d_height;
    op.sca = c->d_sca; op.temperature = c->d_temperature; op.optab = c->d_optab; op.gstride = lsx_optab_group_doubles(c->Ntrans, c->Nspace, c->Ncont);
    bool fold_any = false;
    for (auto& k : c->classes) fold_any = fold_any || k.fold;
    hipLaunchKernelGGL(k_build_optab, dim3((unsigned)((c->ncol + LSX_RS_COLS - 1) / LSX_RS_COLS), (unsigned)(c->Ntrans + 2 + (fold_any ? c->Ncont : 0))),
                       dim3(128), 0, c->stream, op);
}

static int enqueue_fs(lsx_ctx* c, bool timed, bool speculative = false)
{
    if (c->n_phi_set != (size_t)c->ncol) {
        size_t q = 0;
        while (q < (size_t)c->ncol && c->phi_set[q]) ++q;
        return fail(LSX_EINVAL, "formal_sol_gamma: column %zu has no line profiles (lsx_set_columns with phi == NULL "
                                "must be followed by lsx_set_line_profiles)", q);
    }
    HIPCHK(hipSetDevice(c->device));
    if (speculative) {
        // I, Gamma and the monitors of this call go to the second set of buffers (J is a pair anyway): the previous call's stay
        // intact until lsx_discard_formal_sol or the next call
        if (c->d_colmask) return fail(LSX_EUNSUPPORTED, "lsx_formal_sol_gamma_speculative: not with frozen columns (lsx_set_active_columns)");
        if (!c->d_I_alt) {
            const size_t nc = (size_t)c->ncol;
            int rc = dmalloc(&c->d_I_alt, nc * c->Nspect * c->Nrays);
            if (!rc) rc = dmalloc(&c->d_Gamma_alt, nc * c->NL2tot * c->Nspace);
            if (!rc) rc = dmalloc(&c->d_res_alt, 2 * nc + 1);
            if (rc) {
                for (void* q : {(void*)c->d_I_alt, (void*)c->d_Gamma_alt, (void*)c->d_res_alt}) if (q) (void)hipFree(q);
                c->d_I_alt = c->d_Gamma_alt = c->d_res_alt = nullptr;
                return rc;
            }
        }
        swap_result_buffers(c);
        c->spec_dp_zeroed = c->dp_zeroed;
        c->spec_fs_pending = c->fs_pending;
        c->spec_last_dJ = c->last_dJ;
    }
    c->spec_valid = speculative;
    // the ray-serial classes read their per-depth operands from a table that is rebuilt per call (k_build_optab): made on first use
    bool rs_any = false;
    if (use_ray_serial(c) && per_class_launches(c))
        for (auto& k : c->classes) rs_any = rs_any || (c->solver == LSX_SOLVER_PARABOLIC ? k.rsp : k.rs);
    if (rs_any && !c->d_optab) {
        int rc = dmalloc(&c->d_optab, (size_t)((c->ncol + LSX_RS_COLS - 1) / LSX_RS_COLS) * lsx_optab_group_doubles(c->Ntrans, c->Nspace, c->Ncont));
        if (!rc) rc = upload(&c->d_trans_row, c->trans_row, c->stream);
        if (rc) return rc;
        // (the pad element of every block row -- [column][3] in LSX_RS_SEG doubles -- is fetched with its neighbour and never written)
        HIPCHK(hipMemsetAsync(c->d_optab, 0, (size_t)((c->ncol + LSX_RS_COLS - 1) / LSX_RS_COLS) * lsx_optab_group_doubles(c->Ntrans, c->Nspace, c->Ncont) * sizeof(double), c->stream));
        c->optab_fresh = false;
    }
    SweepParams p{};
    p.optab = c->d_optab; p.optab_group_stride = (int64_t)lsx_optab_group_doubles(c->Ntrans, c->Nspace, c->Ncont); p.trans_row = c->d_trans_row;
    p.Nspace = c->Nspace; p.Nrays = c->Nrays; p.Nspect = c->Nspect; p.Natoms = c->Natoms; p.Ntrans = c->Ntrans;
    p.ncol = c->ncol; p.NLtot = c->NLtot; p.NL2tot = c->NL2tot; p.Nlines = c->Nlines;
    p.sca_per_lambda = c->sca_per_lambda; p.phi_compact = c->phi_compact;
    p.nslot_total = (int)c->tile_slots.size(); p.ntile_total = (int)c->tiles.size();
    p.L = c->L;
    p.wavelength = c->d_wavelength; p.zmu = c->d_zmu; p.wmuh = c->d_wmuh; p.wl = c->d_wl; p.alpha = c->d_alpha;
    p.u_la = c->d_u_la; p.fgtab = c->d_fgtab; p.active = c->d_active; p.tiles = c->d_tiles; p.slots = c->d_slots;
    p.phi_G = c->phi_group;
    p.phi_col_stride = (int64_t)c->phi_col; p.corr_col_stride = (int64_t)c->corr_col; p.pp_col_stride = (int64_t)c->pp_col;
    p.height = c->d_height; p.temperature = c->d_temperature; p.n = c->d_n; p.wphi = c->d_wphi;
    p.bgchi_T = c->d_bgchi; p.bgeta_T = c->d_bgeta; p.bgce_T = c->d_bgce; p.bgxce_T = c->d_bgxce; p.bgxchi_T = c->d_bgxchi; p.bgxeta_T = c->d_bgxeta; p.Psi2_T = c->d_Psi2; p.sca = c->d_sca; p.phi_T = c->d_phi; p.nsr = c->d_nsr; p.Ncont = c->Ncont; p.E_T = c->d_E; p.corr_T = c->d_corr; p.Psi3_T = c->d_Psi3;
    p.Jdag_T = c->d_J[c->jcur]; p.Jnew_T = c->d_J[c->jcur ^ 1];
    p.Iout = c->d_I; p.Gpart = c->d_Gpart; p.dJpart = c->d_dJpart; p.debug = c->d_debug; p.colmask = c->d_colmask; p.exp2_tab = c->d_exp2_tab; p.static_max = c->static_max;

    FastParams ff{};
    const bool has_fast = !c->fast_tiles.empty();
    if (has_fast) {
        ff.Nspace = c->Nspace; ff.Nspect = c->Nspect; ff.Nrays = c->Nrays; ff.ncol = c->ncol; ff.ntile = (int)c->tiles.size();
        ff.L = c->L; ff.NLtot = c->NLtot; ff.Natoms = c->Natoms; ff.nslot_total = (int)c->tile_slots.size();
        ff.n_fast_tiles = (int)c->fast_tiles.size(); ff.tiles = c->d_tiles; ff.slots = c->d_slots; ff.fast_tiles = c->d_fast_tiles;
        ff.active = c->d_active; ff.alpha = c->d_alpha; ff.wl = c->d_wl; ff.u_la = c->d_u_la; ff.wmuh = c->d_wmuh; ff.n = c->d_n;
        ff.nsr = c->d_nsr; ff.E_T = c->d_E; ff.corr_T = c->d_corr; ff.Psi3_T = c->d_Psi3; ff.corr_col_stride = (int64_t)c->corr_col; ff.pp_col_stride = (int64_t)c->pp_col; ff.Ncont = c->Ncont; ff.nF_max = c->nF_max; ff.generic = c->fast_generic ? 1 : 0;
        ff.bgchi_T = c->d_bgchi; ff.bgeta_T = c->d_bgeta;
        ff.bgxchi_T = c->d_bgxchi; ff.bgxeta_T = c->d_bgxeta; ff.bgxce_T = c->d_bgxce; ff.pairs_out = 0; ff.J_T = c->d_J[c->jcur ^ 1]; ff.Psi2_T = c->d_Psi2;
        ff.Gpart = c->d_Gpart; ff.colmask = c->d_colmask;
        ff.epi_corr = 0; ff.Nlines = c->Nlines; ff.wphi = c->d_wphi; ff.fgtab = c->d_fgtab;
        // LSX_EPI_ELANE=1: the column-mapped epilogue forms the Boltzmann factor itself instead of reading the stream -- built and measured
        // (profiles/r06_bound_evidence.md 4): 0.5 MB per column of C4 traffic less, the call 0.5 % SLOWER (the kernel is a chain of dependent
        // phases, not a byte mover); off by default
        ff.temperature = (LSX_ELANE && LSX_EPI_ELANE) ? c->d_temperature : nullptr; ff.exp2_tab = c->d_exp2_tab;
    }
    // launch shapes (rows per pass, staged depths, LDS bytes): fixed and checked when the plan was made (lsx_plan.cpp)
    const LaunchShapes& S = c->shapes;
    const int LP = S.rows_lp;
    auto launch_prepass = [&](hipStream_t st, const int* d_list, size_t n, bool epi = false, bool pairs = false) {
        FastParams fq = ff;
        fq.epi_corr = epi ? 1 : 0;
        fq.pairs_out = (pairs && c->d_bgxce) ? 1 : 0;          // a ray-serial class reads its effective background as (chi, eta) pairs
        fq.fast_tiles = d_list; fq.n_fast_tiles = (int)n;
        fq.seg_depths = S.prepass_seg;
        dim3 grid((unsigned)n, (unsigned)c->ncol);
        if (S.prepass_seg < c->Nspace) hipLaunchKernelGGL((k_fast_prepass<true>), grid, dim3(256), S.prepass_lds, st, fq);
        else hipLaunchKernelGGL((k_fast_prepass<false>), grid, dim3(256), S.prepass_lds, st, fq);
    };
    auto launch_fast_rows = [&](hipStream_t st, const int* d_list, size_t n) {
        FastParams fq = ff;
        fq.fast_tiles = d_list; fq.n_fast_tiles = (int)n;
        fq.seg_depths = S.rows_seg;
        const int nt = S.rows_nt;
        const size_t sm = S.rows_lds;
        dim3 grid((unsigned)n, (unsigned)c->ncol);
        const bool seg = fq.seg_depths < c->Nspace;
#define LSX_FG(LPV, NTV) if (LP == LPV && nt == NTV) { if (seg) hipLaunchKernelGGL((k_fast_gamma<LPV, NTV, true>), grid, dim3(NTV), sm, st, fq); \
                                                       else hipLaunchKernelGGL((k_fast_gamma<LPV, NTV, false>), grid, dim3(NTV), sm, st, fq); }
        LSX_FG(16, 256); LSX_FG(16, 128); LSX_FG(16, 64);
        LSX_FG(32, 256); LSX_FG(32, 128); LSX_FG(32, 64);
        LSX_FG(64, 256); LSX_FG(64, 128); LSX_FG(64, 64);
#undef LSX_FG
    };
    // the column-mapped kernel: two lanes per (column, depth), blocks of 256 over the columns' depths x the tiles
    auto launch_fast_cols = [&](hipStream_t st, const int* d_list, size_t n, int v, bool epi) {
        FastParams fq = ff;
        fq.epi_corr = epi ? 1 : 0;
        fq.fast_tiles = d_list; fq.n_fast_tiles = (int)n;
        if (v >= 4) {       // big sets: two waves per workgroup
            dim3 gridb((unsigned)(((size_t)c->ncol * c->Nspace + 2 * LSX_FGC_ROWS - 1) / (2 * LSX_FGC_ROWS)), (unsigned)n);
#define LSX_FC(NLCV) if (fgc_lines(v) == NLCV) hipLaunchKernelGGL((k_fast_gamma_cols_big<NLCV>), gridb, dim3(128), S.cols_lds[v], st, fq);
            LSX_FC(0) LSX_FC(1) LSX_FC(2)
#undef LSX_FC
            return;
        }
        dim3 grid((unsigned)(((size_t)c->ncol * c->Nspace + 4 * LSX_FGC_ROWS - 1) / (4 * LSX_FGC_ROWS)), (unsigned)n);
#define LSX_FC(NLCV) if (kLkLines[v] == NLCV) { if (c->L == 12) hipLaunchKernelGGL((k_fast_gamma_cols<NLCV, 6>), grid, dim3(256), S.cols_lds[v], st, fq); \
                                                else hipLaunchKernelGGL((k_fast_gamma_cols<NLCV, 0>), grid, dim3(256), S.cols_lds[v], st, fq); }
        LSX_FC(0) LSX_FC(1) LSX_FC(2)
#undef LSX_FC
    };
    auto launch_fast_gamma = [&](hipStream_t st, const std::vector<int>* cols, int* const* d_cols, const int* d_rest, size_t nrest, bool epi = false) {
        for (int v = 0; v < LSX_FGC_LISTS; ++v)
            if (v != 3 && !cols[v].empty()) launch_fast_cols(st, d_cols[v], cols[v].size(), v, epi);
        if (nrest) launch_fast_rows(st, d_rest, nrest);
    };
    // From here on nothing returns before the class streams have been joined back and jcur / fs_pending advanced: a launch
    // error is remembered and reported at the end (shapes, LDS sizes and grids were checked when the context was made).
    hipError_t lerr = hipSuccess;
    auto note = [&](hipError_t e) { if (e != hipSuccess && lerr == hipSuccess) lerr = e; };
    if (rs_any && !c->optab_fresh) {
        launch_build_optab(c);
        c->optab_fresh = true;
    }
    if (timed) note(hipEventRecord(c->ev0, c->stream));       // the sweep span starts behind the operand-table build (the whole-call figure, evA, in front of it)
    // small batches: one fused launch (n_class_tiles == ntile, identity tile list is not needed); the parabolic rule (N4) has one
    // generic instance for every tile and takes the same route at any size
    const bool parabolic = c->solver == LSX_SOLVER_PARABOLIC;
    const bool ray_serial = !parabolic && use_ray_serial(c);
    const bool ray_serial_par = parabolic && use_ray_serial(c);
    const int clear_dp = (c->se_pending && !speculative) ? 0 : 1;
    // LSX_GRAPH=1 (measurement): the launches of a call -- fork, the classes' chains on their streams, join, Gamma epilogue -- captured
    // once per (J buffer parity, result buffers, epilogue flavour, rule, mapping, column mask) as a HIP graph and replayed with ONE host
    // call: the host's enqueue time of a dozen launches on five streams is then not what staggers the classes' starts
    hipGraphExec_t gexec = nullptr;
    bool capturing = false;
    const FsGraphKey gkey{c->jcur, (const void*)c->d_I, clear_dp, (int)c->solver, (int)c->sweep_policy, c->policy_columns, (const void*)c->d_colmask};
    if (c->opt_graph && !timed) {
        for (auto& g : c->fs_graphs) if (g.first == gkey) gexec = g.second;
        if (!gexec) {
            note(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
            capturing = lerr == hipSuccess;
        }
    }
    if (gexec) {
        if (!per_class_launches(c)) c->fused_launches++;
        else for (auto& k : c->classes) k.launches++;
    } else {
    if (!per_class_launches(c)) {
        p.ncell_lev = S.fused_ncell_lev; p.ncell_atom = S.fused_ncell_atom;
        c->fused_launches++;
        const int code = parabolic ? -4 : -2;
        // The pre-pass of the tiles with fast continua runs INSIDE the fused launch, in those tiles' own workgroups (lsx_sweep.hip,
        // lsx_fast.h), where every such tile takes the column-mapped epilogue and the column's operands fit the pre-pass's LDS at
        // once (the plan decides: shapes.fused_fast); else as a launch of its own in front.  One FALC column, per MALI iteration:
        // 85.0 -> 79.0 us (the pre-pass hides behind the longer line tiles).  The Gamma epilogue stays a launch of its own behind the
        // sweep: inside (LSX_FUSED_EPILOGUE=1) it extends the workgroups that already finish last, 84.3 us.
        // (measured before that: starting the tiles without fast continua on a second stream, ahead of the pre-pass, costs more in the
        // fork and join than the 8 us it takes off the critical path: 4.62 against 4.40 ms for the 46 iterations of a FALC column)
        const bool inside = has_fast && !parabolic && S.fused_fast && !c->opt_no_fused_fast;
        p.fused_fast = inside ? 1 : 0;
        if (inside && c->opt_abl_fast) p.fused_fast |= c->opt_abl_fast;      // timing ablations (LSX_ABL_FUSED_FAST: 2 no pre-pass, 4 no epilogue; wrong results)
        if (inside && !c->opt_fused_epilogue) p.fused_fast |= 4;
        p.nF_max = c->nF_max;
        if (has_fast && !inside) launch_prepass(c->stream, c->d_fast_tiles, c->fast_tiles.size());
        p.class_tiles = nullptr;
        p.n_class_tiles = (int)c->tiles.size();
        note(lsx_launch_sweep(&p, code, (int)((long)c->tiles.size() * c->ncol), inside ? S.fused_fast_lds : S.fused_lds, c->stream));
        const bool epilogue_inside = inside && c->opt_fused_epilogue && !(p.fused_fast & 4);
        if (has_fast && !epilogue_inside) launch_fast_gamma(c->stream, c->fast_cols, c->d_fast_cols, c->d_fast_rest, c->fast_rest.size());
    } else {
        // The classes of one call run side by side on their own streams, forked from the context's stream and joined
        // back into it.  The fast continua of a class's tiles are handled on the class's own stream, pre-pass before
        // and epilogue after the sweep, so no class waits for another and those two light kernels fill gaps.
        const bool fork = c->classes.size() > 1 && !c->opt_serial;
        if (fork) note(hipEventRecord(c->ev_fork, c->stream));
        for (auto& k : c->classes) {
            hipStream_t st = fork ? k.stream : c->stream;
            if (fork) note(hipStreamWaitEvent(st, c->ev_fork, 0));
            // a class of line tiles with linked continua on the ray-serial kernel: the sweep reads no correction streams, the pre-pass
            // writes none, the column-mapped epilogue applies the corrections to the lines' rates (lsx_fast.h; the plan has checked
            // that every tile of the class takes that epilogue)
            const bool rs_here = (k.rs && ray_serial) || (k.rsp && ray_serial_par);
            const bool epi = k.lk_epi && rs_here;
            // folded instances (lsx_plan.h): the sweep forms the fast continua's opacity and emissivity itself -- no pre-pass
            const bool fold = k.fold && k.rs && ray_serial;
            const bool epi_in_sweep = fold && k.epi;          // ... and their Gamma integrands: no epilogue launch
            p.fold = fold ? 1 : 0; p.fold_nF = k.fold_nF; p.epi = epi_in_sweep ? 1 : 0;
            if (!k.fast_tiles.empty() && !fold) launch_prepass(st, k.d_fast_tiles, k.fast_tiles.size(), epi, rs_here);
            const long nblocks = (long)k.tiles.size() * c->ncol;
            p.class_tiles = k.d_tiles;
            p.n_class_tiles = (int)k.tiles.size();
            p.ncell_lev = k.npt >= 0 ? 0 : k.ncell_lev; p.ncell_atom = k.npt >= 0 ? 0 : k.ncell_atom; p.nstash = 0;
            k.launches++;
            if (k.rsp && ray_serial_par) note(lsx_launch_sweep_rs_par(&p, k.code(), st));     // N4, five columns per wavefront
            else if (parabolic) {
                // a class with a compile-time instance of the rule runs it on its own LDS layout; every other class runs the
                // generic instance (level / atom cells, the fused launch's layout) on the class's tile list
                const bool inst = c->Nrays == LSX_RS_RAYS && !c->sca_per_lambda && k.npt >= 0 && lsx_rs_instance_exists(k.npt, k.nl, k.linked, k.topo);
                if (!inst) { p.ncell_lev = S.fused_ncell_lev; p.ncell_atom = S.fused_ncell_atom; }
                note(lsx_launch_sweep_par(&p, inst ? k.code() : -1, (int)nblocks, inst ? k.lds_bytes : S.fused_lds, st));
            } else if (k.rs && ray_serial) note(lsx_launch_sweep_rs(&p, k.code(), st));       // five columns per wavefront
            else note(lsx_launch_sweep(&p, k.code(), (int)nblocks, k.lds_bytes, st));
            if (timed) {
                if (!k.tdone) note(hipEventCreate(&k.tdone));
                if (k.tdone) note(hipEventRecord(k.tdone, st));
            }
#ifndef LSX_ABL_NO_FAST_GAMMA       // (ablation build, wrong results: what a call costs without the fast-continuum epilogue -- profiles/r05)
            if (!k.fast_tiles.empty() && !epi_in_sweep) launch_fast_gamma(st, k.fast_cols, k.d_fast_cols, k.d_fast_rest, k.fast_rest.size(), epi);
#endif
            if (fork) {
                note(hipEventRecord(k.done, st));
                note(hipStreamWaitEvent(c->stream, k.done, 0));
            }
        }
    }
    note(hipGetLastError());
    if (timed) note(hipEventRecord(c->ev1, c->stream));

    FinishParams f{};
    f.Nspace = c->Nspace; f.Natoms = c->Natoms; f.NL2tot = c->NL2tot; f.ncol = c->ncol; f.ntile = (int)c->tiles.size();
    f.nslot_total = (int)c->tile_slots.size(); f.Nlevel = c->d_Nlevel; f.lev2_off = c->d_lev2_off; f.tiles = c->d_tiles;
    f.tile_slots = c->d_tile_slots; f.trans = c->d_trans; f.C = c->d_C; f.Gpart = c->d_Gpart; f.dJpart = c->d_dJpart;
    f.Gamma = c->d_Gamma; f.dJcol = c->d_dJcol; f.colmask = c->d_colmask;
    f.dPcol = c->d_dPcol; f.singular = c->d_singular; f.fin_ptr = c->d_fin_ptr; f.fin_idx = c->d_fin_idx; f.atom_ptr = c->d_atom_ptr; f.atom_slots = c->d_atom_slots;
    // `FS; SE; FS; sync` (and `SE; formal_sol_gamma(&dJ)`): the statistical equilibrium's per-column maxima and its singular flag
    // have not been read back yet -- they live in the block this epilogue would clear.  Then the epilogue leaves them alone and
    // the next stat_equil clears them itself.  (A speculative call writes the second block: nothing pending there.)
    f.clear_dp = clear_dp;
    const long nthreads = (long)c->ncol * c->Nspace;
    if (c->ncol < 32 && !c->opt_finish_big)
        hipLaunchKernelGGL(k_gamma_finish_small, dim3((unsigned)nthreads), dim3(64), (size_t)c->NL2tot * sizeof(double), c->stream, f);
    else if (S.finish_w > 0) {
        hipLaunchKernelGGL(k_gamma_finish_levels, dim3((unsigned)((nthreads + 127) / 128), (unsigned)c->NLtot), dim3(128), 0, c->stream, f,
                           (const int*)c->d_level_atom, (const int*)c->d_level_atom + c->NLtot);
    } else
        hipLaunchKernelGGL(k_gamma_finish, dim3((unsigned)((nthreads + S.finish_nt - 1) / S.finish_nt), (unsigned)c->Natoms), dim3(S.finish_nt), S.finish_lds, c->stream, f);
    }   // (!gexec)
    if (capturing) {
        hipGraph_t g = nullptr;
        note(hipStreamEndCapture(c->stream, &g));
        if (lerr == hipSuccess) note(hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0));
        if (g) (void)hipGraphDestroy(g);
        if (lerr == hipSuccess) c->fs_graphs.emplace_back(gkey, gexec);
    }
    if (gexec) note(hipGraphLaunch(gexec, c->stream));
    c->dp_zeroed = clear_dp != 0;
    note(hipGetLastError());
    if (timed) note(hipEventRecord(c->ev2, c->stream));
    c->jcur ^= 1;
    c->fs_pending = true;
    // (LSX_EDEVICE whatever the HIP error: shapes were checked when the context was made, and LSX_EUNSUPPORTED from the speculative
    // entry means "refused, nothing enqueued" to the drivers)
    if (lerr != hipSuccess) return fail(LSX_EDEVICE, "formal_sol_gamma: a launch failed: %s", hipGetErrorString(lerr));
    return LSX_OK;
}

int lsx_set_line_profiles(lsx_ctx* c, int32_t col0, int32_t ncol, const double* aDamp, const double* vBroad, const double* vlos)
{
    if (!c || !aDamp || !vBroad || col0 < 0 || ncol < 1 || col0 + ncol > c->ncol)
        return fail(LSX_EINVAL, "lsx_set_line_profiles: bad argument");
    if (c->phi_compact && vlos) return fail(LSX_EINVAL, "lsx_set_line_profiles: a phi_compact context takes vlos == NULL");
    if (!c->Nlines) return LSX_OK;
    HIPCHK(hipSetDevice(c->device));
    const int Ns = c->Nspace;
    const size_t per = (size_t)(c->Nlines + c->Natoms + 1) * Ns;
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(ncol, ((size_t)32 << 20) / per));
    int rc = ensure_stage(c, chunk * per);
    if (rc) return rc;
    for (size_t b0 = 0; b0 < (size_t)ncol; b0 += chunk) {
        const size_t nb = std::min(chunk, (size_t)ncol - b0);
        const size_t cc = (size_t)col0 + b0;
        double* dA = c->d_stage;
        double* dV = dA + nb * c->Nlines * Ns;
        double* dL = dV + nb * c->Natoms * Ns;
        HIPCHK(hipMemcpyAsync(dA, aDamp + b0 * c->Nlines * Ns, nb * c->Nlines * Ns * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(dV, vBroad + b0 * c->Natoms * Ns, nb * c->Natoms * Ns * 8, hipMemcpyHostToDevice, c->stream));
        if (vlos) HIPCHK(hipMemcpyAsync(dL, vlos + b0 * Ns, nb * Ns * 8, hipMemcpyHostToDevice, c->stream));
        if ((rc = profiles_from_device(c, cc, nb, dA, dV, vlos ? dL : nullptr))) return rc;
        HIPCHK(hipStreamSynchronize(c->stream));   // the staging buffer is re-used by the next sub-chunk
    }
    mark_profiles_set(c, (size_t)col0, (size_t)ncol);
    return LSX_OK;
}

int lsx_formal_sol_gamma_async(lsx_ctx* c)
{
    if (!c) return fail(LSX_EINVAL, "null ctx");
    return enqueue_fs(c, false);
}

int lsx_formal_sol_gamma_speculative(lsx_ctx* c)
{
    if (!c) return fail(LSX_EINVAL, "null ctx");
    return enqueue_fs(c, false, true);
}

int lsx_prefers_lookahead(lsx_ctx* c)
{
    return c && !per_class_launches(c) && !c->d_colmask ? 1 : 0;      // the fused launch on the context's stream (enqueue_fs)
}

int lsx_discard_formal_sol(lsx_ctx* c)
{
    if (!c) return fail(LSX_EINVAL, "null ctx");
    if (!c->spec_valid) return fail(LSX_EINVAL, "lsx_discard_formal_sol: the last call was not a speculative formal solution");
    if (c->mon_outstanding && c->mon_spec)
        return fail(LSX_EINVAL, "lsx_discard_formal_sol: a read-back begun after the speculative call is in flight (lsx_sync_end first: "
                                "it would report the discarded call's monitors)");
    // (its kernels may still be running: everything that follows is ordered behind them on the context's stream)
    swap_result_buffers(c);
    c->jcur ^= 1;
    c->dp_zeroed = c->spec_dp_zeroed;
    c->fs_pending = c->spec_fs_pending;      // an earlier formal solution whose monitors nobody has read yet is pending again
    c->last_dJ = c->spec_last_dJ;            // (and lsx_sync reports the accepted call's dJ, whether or not the discarded one's was read)
    c->spec_valid = false;
    return LSX_OK;
}

int lsx_stat_equil_async(lsx_ctx* c)
{
    if (!c) return fail(LSX_EINVAL, "null ctx");
    c->spec_valid = false;            // the populations now build on the last formal solution
    c->optab_fresh = false;           // the populations change (cleared FIRST: an early return below must not leave a stale table marked fresh)
    for (int a = 0; a < c->Natoms; ++a) {       // nothing is launched unless every atom's system can be: the call does not fail partway
        const size_t sm = (size_t)(c->Nlevel[a] * c->Nlevel[a] + 2 * c->Nlevel[a]) * 64 * sizeof(double);
        if ((c->opt_se_lds || c->Nlevel[a] > 8) && sm > 160 * 1024)
            return fail(LSX_EUNSUPPORTED, "stat_equil: Nlevel = %d needs %zu B of LDS", c->Nlevel[a], sm);
    }
    HIPCHK(hipSetDevice(c->device));
    // dPcol and the singular flag behind it start at zero: the Gamma epilogue of the formal solution has done that, unless
    // this is a second stat_equil on the same Gamma
    if (!c->dp_zeroed) HIPCHK(hipMemsetAsync(c->d_dPcol, 0, ((size_t)c->ncol + 1) * 8, c->stream));
    c->dp_zeroed = false;
    const long nthreads = (long)c->ncol * c->Nspace;
    for (int a = 0; a < c->Natoms; ++a) {
        const int Nl = c->Nlevel[a];
        const int nt = 64;
        const dim3 grid((unsigned)((nthreads + nt - 1) / nt));
#define SE_REG(NLC) case NLC: hipLaunchKernelGGL((k_stat_equil_reg<NLC>), grid, dim3(nt), 0, c->stream, c->d_Gamma, c->d_nTotal, c->d_n, \
                              c->d_dPcol, c->d_singular, c->lev_off[a], c->lev2_off[a], a, c->Natoms, c->NLtot, c->NL2tot,      \
                              c->Nspace, c->ncol, c->d_colmask); break;
        switch (c->opt_se_lds ? 0 : Nl) {
        SE_REG(2) SE_REG(3) SE_REG(4) SE_REG(5) SE_REG(6) SE_REG(7) SE_REG(8)
        default: {
            const size_t sm = (size_t)(Nl * Nl + 2 * Nl) * nt * sizeof(double);
            if (sm > 160 * 1024) return fail(LSX_EUNSUPPORTED, "stat_equil: Nlevel = %d needs %zu B of LDS", Nl, sm);
            hipLaunchKernelGGL(k_stat_equil, grid, dim3(nt), sm, c->stream, c->d_Gamma, c->d_nTotal, c->d_n, c->d_dPcol,
                               c->d_singular, Nl, c->lev_off[a], c->lev2_off[a], a, c->Natoms, c->NLtot, c->NL2tot, c->Nspace,
                               c->ncol, c->d_colmask);
        }
        }
#undef SE_REG
        HIPCHK(hipGetLastError());
    }
    c->se_pending = true;
    c->optab_fresh = false;           // the populations have changed (the table is rebuilt behind the read-back of this call's monitors: lsx_sync)
    return LSX_OK;
}

int lsx_set_active_columns(lsx_ctx* c, const uint8_t* active)
{
    if (c) c->spec_valid = false;         // new inputs: a speculative formal solution can no longer be discarded
    if (!c) return fail(LSX_EINVAL, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (!active) {
        if (c->d_colmask) HIPCHK(hipFree(c->d_colmask));
        c->d_colmask = nullptr;
        return LSX_OK;
    }
    if (!c->d_colmask) {
        int rc = dmalloc(&c->d_colmask, (size_t)c->ncol);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpy(c->d_colmask, active, (size_t)c->ncol, hipMemcpyHostToDevice));
    return LSX_OK;
}

// the maxima of a read-back block (h_pinned): -> the singular flag
static unsigned long long digest_monitors(lsx_ctx* c, bool fs, bool se)
{
    const size_t nc = (size_t)c->ncol;
    unsigned long long sing = 0;
    auto nanmax = [](const double* v, size_t n) {       // numpy max semantics: NaN wins (rh_method.py:706)
        double m = 0.0;
        for (size_t i = 0; i < n; ++i)
            if (v[i] != v[i]) return v[i];
            else if (v[i] > m) m = v[i];
        return m;
    };
    if (fs) c->last_dJ = nanmax(c->h_pinned, nc);
    if (se) {
        c->last_dP = nanmax(c->h_pinned + nc, nc);      // the per-column values are never NaN (k_stat_equil)
        memcpy(&sing, c->h_pinned + 2 * nc, sizeof sing);
    }
    return sing;
}

static int sync_begin(lsx_ctx* c, bool with_n)
{
    if (!c) return fail(LSX_EINVAL, "null ctx");
    if (c->mon_outstanding) return fail(LSX_EINVAL, "lsx_sync_begin: the previous read-back has not been collected (lsx_sync_end)");
    HIPCHK(hipSetDevice(c->device));
    if (!c->ev_mon) HIPCHK(hipEventCreateWithFlags(&c->ev_mon, hipEventDisableTiming));
    c->h_n_valid = false;
    c->h_n_pending = false;
    const size_t n_bytes = (size_t)c->ncol * c->NLtot * c->Nspace * sizeof(double);
    if (with_n && !c->h_n && hipHostMalloc(reinterpret_cast<void**>(&c->h_n), n_bytes, hipHostMallocDefault) != hipSuccess) {
        c->h_n = nullptr;
        return fail(LSX_EDEVICE, "lsx_sync_begin_populations: no pinned host memory for %zu bytes of populations", n_bytes);
    }
    HIPCHK(hipMemcpyAsync(c->h_pinned, c->d_res, (2 * (size_t)c->ncol + 1) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (with_n) {
        HIPCHK(hipMemcpyAsync(c->h_n, c->d_n, n_bytes, hipMemcpyDeviceToHost, c->stream));
        c->h_n_pending = true;
    }
    HIPCHK(hipEventRecord(c->ev_mon, c->stream));
    c->mon_spec = c->spec_valid;      // this read-back carries a speculative call's monitors: collect it before discarding that call
    c->mon_fs = c->fs_pending; c->mon_se = c->se_pending;
    c->fs_pending = c->se_pending = false;
    c->mon_outstanding = true;
    return LSX_OK;
}

int lsx_sync_begin(lsx_ctx* c) { return sync_begin(c, false); }
int lsx_sync_begin_populations(lsx_ctx* c) { return sync_begin(c, true); }

int lsx_fetch_populations(lsx_ctx* c, double* dst, size_t nbytes)
{
    if (!c || !dst) return fail(LSX_EINVAL, "lsx_fetch_populations: null argument");
    if (!c->h_n_valid) return fail(LSX_EINVAL, "lsx_fetch_populations: no collected read-back of the populations (lsx_sync_begin_populations, lsx_sync_end)");
    if (nbytes != (size_t)c->ncol * c->NLtot * c->Nspace * sizeof(double)) return fail(LSX_EINVAL, "lsx_fetch_populations: nbytes does not match [ncol][NLtot][Nspace]");
    memcpy(dst, c->h_n, nbytes);
    return LSX_OK;
}

static int singular_error(lsx_ctx* c, unsigned long long sing);

int lsx_sync_end(lsx_ctx* c, double* dJ, double* dP)
{
    if (!c) return fail(LSX_EINVAL, "null ctx");
    if (!c->mon_outstanding) return lsx_sync(c, dJ, dP);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventSynchronize(c->ev_mon));             // the read-back only: what was enqueued behind it keeps running
    c->mon_outstanding = false;
    c->h_n_valid = c->h_n_pending; c->h_n_pending = false;
    const unsigned long long sing = digest_monitors(c, c->mon_fs, c->mon_se);
    if (dJ) *dJ = c->last_dJ;
    if (dP) *dP = c->last_dP;
    return sing ? singular_error(c, sing) : LSX_OK;
}

int lsx_sync(lsx_ctx* c, double* dJ, double* dP)
{
    if (!c) return fail(LSX_EINVAL, "null ctx");
    HIPCHK(hipSetDevice(c->device));
    unsigned long long sing = 0;
    if (c->mon_outstanding) {                           // a read-back in flight: collect it first (its maxima are superseded below
        HIPCHK(hipEventSynchronize(c->ev_mon));         // if later calls are pending)
        c->mon_outstanding = false;
        c->h_n_valid = c->h_n_pending; c->h_n_pending = false;
        sing = digest_monitors(c, c->mon_fs, c->mon_se);
    }
    if (c->fs_pending || c->se_pending) {
        const size_t nc = (size_t)c->ncol;
        HIPCHK(hipMemcpyAsync(c->h_pinned, c->d_res, (2 * nc + 1) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        // The ray-serial sweeps' operand table follows the populations a statistical equilibrium has just changed.  Built HERE,
        // behind the read-back the host is about to wait for, it runs while the host digests the monitors and decides on the next
        // iteration -- not in front of the next formal solution (C3: 18 us of a 1 ms iteration; C4: 43 us).  The wait below is for
        // the read-back only: later calls are ordered behind the build by the stream.
        if (!c->optab_fresh && c->d_optab && c->se_pending && use_ray_serial(c) && per_class_launches(c)) {
            if (!c->ev_mon) HIPCHK(hipEventCreateWithFlags(&c->ev_mon, hipEventDisableTiming));
            HIPCHK(hipEventRecord(c->ev_mon, c->stream));
            launch_build_optab(c);
            HIPCHK(hipGetLastError());
            c->optab_fresh = true;
            HIPCHK(hipEventSynchronize(c->ev_mon));
        } else
        HIPCHK(hipStreamSynchronize(c->stream));
        const unsigned long long s2 = digest_monitors(c, c->fs_pending, c->se_pending);
        if (!sing) sing = s2;
        c->fs_pending = c->se_pending = false;
    } else {
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (dJ) *dJ = c->last_dJ;
    if (dP) *dP = c->last_dP;
    return sing ? singular_error(c, sing) : LSX_OK;
}

static int singular_error(lsx_ctx* c, unsigned long long sing)
{
    const unsigned long long key = LSX_SING_BASE - sing;
    const long gid = (long)(key >> 8);
    return fail(LSX_ESINGULAR, "stat_equil: singular matrix at column %ld, depth %ld, atom %d (the first such system; cf. "
                               "LinAlgError at rh_method.py:739); its populations are left untouched",
                gid / c->Nspace, gid % c->Nspace, (int)(key & 0xff));
}

int lsx_monitors(lsx_ctx* c, double* dst)
{
    if (!c || !dst) return fail(LSX_EINVAL, "lsx_monitors: null argument");
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_monitors, dim3(1), dim3(256), 0, c->stream, c->d_res, c->ncol, dst);
    HIPCHK(hipGetLastError());
    return LSX_OK;
}

int lsx_formal_sol_gamma(lsx_ctx* c, double* dJ)
{
    int rc = lsx_formal_sol_gamma_async(c);
    if (rc) return rc;
    return lsx_sync(c, dJ, nullptr);
}

int lsx_stat_equil(lsx_ctx* c, double* dP)
{
    int rc = lsx_stat_equil_async(c);
    if (rc) return rc;
    return lsx_sync(c, nullptr, dP);
}

int lsx_get(lsx_ctx* c, int32_t what, int32_t col0, int32_t ncol, double* dst, size_t nbytes)
{
    if (!c || !dst || col0 < 0 || ncol < 1 || col0 + ncol > c->ncol) return fail(LSX_EINVAL, "lsx_get: bad range");
    HIPCHK(hipSetDevice(c->device));
    const size_t Ns = c->Nspace;
    const double* base = nullptr;
    size_t per = 0;
    switch (what) {
    case LSX_I: base = c->d_I; per = (size_t)c->Nspect * c->Nrays; break;
    case LSX_J: per = (size_t)c->Nspect * Ns; break;
    case LSX_N: base = c->d_n; per = (size_t)c->NLtot * Ns; break;
    case LSX_GAMMA: base = c->d_Gamma; per = (size_t)c->NL2tot * Ns; break;
    case LSX_DJ_COL: base = c->d_dJcol; per = 1; break;
    case LSX_DPOPS_COL: base = c->d_dPcol; per = 1; break;
    case LSX_NSTAR: base = c->d_nStar; per = (size_t)c->NLtot * Ns; break;
    case LSX_C: base = c->d_C; per = (size_t)c->NL2tot * Ns; break;
    case LSX_WPHI: base = c->d_wphi; per = (size_t)c->Nlines * Ns; break;
    case LSX_VBROAD:
    case LSX_ADAMP:
        if (!c->d_vBroad) return fail(LSX_EINVAL, "lsx_get: vBroad / aDamp exist after lsx_set_atmosphere only");
        base = what == LSX_VBROAD ? c->d_vBroad : c->d_aDamp;
        per = (size_t)(what == LSX_VBROAD ? c->Natoms : std::max(1, c->Nlines)) * Ns;
        break;
    case LSX_PHI: per = c->phi_in_col; break;
    default: return fail(LSX_EINVAL, "lsx_get: unknown item %d", what);
    }
    if (nbytes != per * ncol * 8) return fail(LSX_EINVAL, "lsx_get: nbytes does not match the item's shape");
    if (what == LSX_PHI) { // device holds one block per (tile, line): back to [lt][mu][dir][k]
        if (per == 0) return LSX_OK;
        const size_t chunk = std::max<size_t>(1, std::min<size_t>(ncol, ((size_t)32 << 20) / per));
        int rc = ensure_stage(c, chunk * per);
        if (rc) return rc;
        const int R = c->phi_compact ? 1 : c->Nrays, D = c->phi_compact ? 1 : 2;
        for (size_t b0 = 0; b0 < (size_t)ncol; b0 += chunk) {
            const size_t nb = std::min(chunk, (size_t)ncol - b0);
            for (const DevSlot& sl : c->slots) {
                if (!(sl.flags & SLOT_LINE) || sl.len <= 0) continue;
                const DevTrans& h = c->htrans[sl.trans];
                const size_t total = (size_t)sl.len * R * D * Ns;
                dim3 grid((unsigned)std::min<size_t>((total + 255) / 256, 256), (unsigned)nb);
                hipLaunchKernelGGL(k_unpack_phi, grid, dim3(256), 0, c->stream, PhiBlock{c->d_phi, c->phi_col, (size_t)col0 + b0, (size_t)sl.base, c->phi_group},
                                   c->d_stage + (size_t)h.phi_off * R * D * Ns, sl.first - h.Nblue, sl.len, R, D, (int)Ns, c->phi_in_col);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipMemcpyAsync(dst + b0 * per, c->d_stage, nb * per * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        return LSX_OK;
    }
    if (what == LSX_J) { // device holds J depth-major: [k][la] -> [la][k]
        const size_t chunk = std::max<size_t>(1, std::min<size_t>(ncol, ((size_t)32 << 20) / per));
        int rc = ensure_stage(c, chunk * per);
        if (rc) return rc;
        for (size_t b0 = 0; b0 < (size_t)ncol; b0 += chunk) {
            const size_t nb = std::min(chunk, (size_t)ncol - b0);
            rc = launch_tiles_pack(c, c->d_J[c->jcur] + (col0 + b0) * c->til_col, c->d_stage, (int)nb, true);
            if (rc) return rc;
            HIPCHK(hipMemcpyAsync(dst + b0 * per, c->d_stage, nb * per * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        return LSX_OK;
    }
    HIPCHK(hipMemcpyAsync(dst, base + per * col0, nbytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return LSX_OK;
}

int lsx_set(lsx_ctx* c, int32_t what, int32_t col0, int32_t ncol, const double* src, size_t nbytes)
{
    if (c) c->optab_fresh = false;
    if (c) c->spec_valid = false;         // new inputs: a speculative formal solution can no longer be discarded
    if (!c || !src || col0 < 0 || ncol < 1 || col0 + ncol > c->ncol) return fail(LSX_EINVAL, "lsx_set: bad range");
    if (what != LSX_N && what != LSX_J) return fail(LSX_EINVAL, "lsx_set: only LSX_N and LSX_J are writable");
    HIPCHK(hipSetDevice(c->device));
    const size_t Ns = c->Nspace;
    const size_t per = (what == LSX_N ? (size_t)c->NLtot : (size_t)c->Nspect) * Ns;
    if (nbytes != per * ncol * 8) return fail(LSX_EINVAL, "lsx_set: nbytes does not match the item's shape");
    if (what == LSX_N) {
        HIPCHK(hipMemcpyAsync(c->d_n + per * col0, src, nbytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return LSX_OK;
    }
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(ncol, ((size_t)32 << 20) / per));
    int rc = ensure_stage(c, chunk * per);
    if (rc) return rc;
    for (size_t b0 = 0; b0 < (size_t)ncol; b0 += chunk) {
        const size_t nb = std::min(chunk, (size_t)ncol - b0);
        HIPCHK(hipMemcpyAsync(c->d_stage, src + b0 * per, nb * per * 8, hipMemcpyHostToDevice, c->stream));
        rc = launch_tiles_pack(c, c->d_stage, c->d_J[c->jcur] + (col0 + b0) * c->til_col, (int)nb, false);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return LSX_OK;
}

// Stand-alone formal solver entry points: per-device scratch (a stream, one growing device buffer, the exp table) that
// lives for the life of the library, so a call costs copies + one launch, not nine hipMalloc/hipFree pairs.
namespace {
struct PwScratch {
    hipStream_t stream = nullptr;
    char* buf = nullptr;
    size_t cap = 0;
    double* exp_tab = nullptr;
};
std::mutex g_pw_mutex;
PwScratch g_pw[64];

int pw_scratch(int device, size_t bytes, PwScratch** out)
{
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(LSX_EDEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev || device >= 64) return fail(LSX_EINVAL, "bad device %d (%d visible)", device, ndev);
    HIPCHK(hipSetDevice(device));
    PwScratch& q = g_pw[device];
    if (!q.stream) HIPCHK(hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking));
    if (!q.exp_tab) {
        const std::vector<double> et = make_exp2_table();
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&q.exp_tab), et.size() * sizeof(double)));
        HIPCHK(hipMemcpy(q.exp_tab, et.data(), et.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (bytes > q.cap) {
        if (q.buf) HIPCHK(hipFree(q.buf));
        q.buf = nullptr;
        q.cap = 0;
        const size_t want = std::max<size_t>(bytes + bytes / 4, (size_t)1 << 20);
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&q.buf), want));
        q.cap = want;
    }
    *out = &q;
    return LSX_OK;
}

int run_piecewise(int32_t device, int32_t nray, int32_t Nspace, const double* height, const double* temperature,
                  const double* mu, const int32_t* to_obs, const double* wav, const double* Istart, const double* chi,
                  const double* S, double* I, double* PsiStar, const char* who, bool parabolic = false)
{
    if (nray < 0 || Nspace < 3) return fail(LSX_EINVAL, "%s: need Nspace >= 3 (formal_solver.py:120-139)", who);
    if (nray == 0) return LSX_OK;
    if (!height || !mu || !to_obs || !chi || !S || !I || !PsiStar || (!Istart && (!temperature || !wav)))
        return fail(LSX_EINVAL, "%s: null array pointer", who);
    std::lock_guard<std::mutex> lock(g_pw_mutex);
    const size_t Ns = Nspace, nr = nray;
    const size_t rnd = 32;   // keep every array 256-B aligned
    auto al = [&](size_t n) { return (n + rnd - 1) / rnd * rnd; };
    const size_t doubles = 2 * al(Ns) + 3 * al(nr) + 4 * al(nr * Ns) + al((nr + 1) / 2);
    PwScratch* q = nullptr;
    int rc = pw_scratch(device, doubles * sizeof(double), &q);
    if (rc) return rc;
    double* p = reinterpret_cast<double*>(q->buf);
    double* dz = p; p += al(Ns);
    double* dT = p; p += al(Ns);
    double* dmu = p; p += al(nr);
    double* dwav = p; p += al(nr);
    double* dI0 = p; p += al(nr);
    double* dchi = p; p += al(nr * Ns);
    double* dS = p; p += al(nr * Ns);
    double* dI = p; p += al(nr * Ns);
    double* dP = p; p += al(nr * Ns);
    int* dto = reinterpret_cast<int*>(p);
    hipStream_t st = q->stream;
    HIPCHK(hipMemcpyAsync(dz, height, Ns * 8, hipMemcpyHostToDevice, st));
    if (temperature) HIPCHK(hipMemcpyAsync(dT, temperature, Ns * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dmu, mu, nr * 8, hipMemcpyHostToDevice, st));
    if (wav) HIPCHK(hipMemcpyAsync(dwav, wav, nr * 8, hipMemcpyHostToDevice, st));
    if (Istart) HIPCHK(hipMemcpyAsync(dI0, Istart, nr * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dto, to_obs, nr * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dchi, chi, nr * Ns * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dS, S, nr * Ns * 8, hipMemcpyHostToDevice, st));
    if (parabolic)
        hipLaunchKernelGGL(k_piecewise_parabolic, dim3((nray + 63) / 64), dim3(64), 0, st, nray, Nspace, dz, dmu, dto, dI0, dchi, dS, dI, dP,
                           q->exp_tab);
    else
        hipLaunchKernelGGL(k_piecewise, dim3((nray + 63) / 64), dim3(64), 0, st, nray, Nspace, dz, dT, dmu, dto, dwav,
                           Istart ? dI0 : nullptr, dchi, dS, dI, dP, q->exp_tab);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(I, dI, nr * Ns * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(PsiStar, dP, nr * Ns * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return LSX_OK;
}
} // namespace

int lsx_piecewise_linear_1d(int32_t device, int32_t nray, int32_t Nspace, const double* height, const double* temperature,
                            const double* mu, const int32_t* to_obs, const double* wav, const double* chi, const double* S,
                            double* I, double* PsiStar)
{
    if (nray > 0 && (!temperature || !wav)) return fail(LSX_EINVAL, "lsx_piecewise_linear_1d: null array pointer");
    return run_piecewise(device, nray, Nspace, height, temperature, mu, to_obs, wav, nullptr, chi, S, I, PsiStar,
                         "lsx_piecewise_linear_1d");
}

int lsx_piecewise_1d_impl(int32_t device, int32_t nray, int32_t Nspace, const double* height, const double* mu,
                          const int32_t* to_obs, const double* Istart, const double* chi, const double* S, double* I,
                          double* PsiStar)
{
    if (nray > 0 && !Istart) return fail(LSX_EINVAL, "lsx_piecewise_1d_impl: null array pointer");
    return run_piecewise(device, nray, Nspace, height, nullptr, mu, to_obs, nullptr, Istart, chi, S, I, PsiStar,
                         "lsx_piecewise_1d_impl");
}

int lsx_piecewise_parabolic_1d_impl(int32_t device, int32_t nray, int32_t Nspace, const double* height, const double* mu,
                                    const int32_t* to_obs, const double* Istart, const double* chi, const double* S, double* I,
                                    double* PsiStar)
{
    if (nray > 0 && !Istart) return fail(LSX_EINVAL, "lsx_piecewise_parabolic_1d_impl: null array pointer");
    return run_piecewise(device, nray, Nspace, height, nullptr, mu, to_obs, nullptr, Istart, chi, S, I, PsiStar,
                         "lsx_piecewise_parabolic_1d_impl", true);
}

int lsx_w3(int32_t device, int32_t n, const double* dtau, double* w)
{
    if (n < 0 || (n > 0 && (!dtau || !w))) return fail(LSX_EINVAL, "lsx_w3: bad argument");
    if (n == 0) return LSX_OK;
    std::lock_guard<std::mutex> lock(g_pw_mutex);
    PwScratch* q = nullptr;
    int rc = pw_scratch(device, (size_t)n * 4 * sizeof(double), &q);
    if (rc) return rc;
    double* din = reinterpret_cast<double*>(q->buf);
    double* dout = din + n;
    HIPCHK(hipMemcpyAsync(din, dtau, (size_t)n * 8, hipMemcpyHostToDevice, q->stream));
    hipLaunchKernelGGL(k_w3, dim3((n + 63) / 64), dim3(64), 0, q->stream, n, din, dout, q->exp_tab);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(w, dout, (size_t)n * 24, hipMemcpyDeviceToHost, q->stream));
    HIPCHK(hipStreamSynchronize(q->stream));
    return LSX_OK;
}

int lsx_set_sweep_policy(lsx_ctx* c, int32_t policy, int32_t decide_for_columns)
{
    if (!c || policy < LSX_SWEEP_AUTO || policy > LSX_SWEEP_RAY_SERIAL || decide_for_columns < 0)
        return fail(LSX_EINVAL, "lsx_set_sweep_policy: bad argument");
    if (policy == LSX_SWEEP_RAY_SERIAL && !c->rs_ok)
        return fail(LSX_EUNSUPPORTED, "lsx_set_sweep_policy: this context's shape has no ray-serial kernel (it needs %d rays, a "
                                      "wavelength-independent scattering coefficient and column blocks within 32-bit offsets)", LSX_RS_RAYS);
    c->spec_valid = false;            // a speculative formal solution was made under the old policy: it stands
    c->sweep_policy = policy;
    c->policy_columns = decide_for_columns;
    return LSX_OK;
}

int32_t lsx_sweep_policy(const lsx_ctx* c)
{
    return c && use_ray_serial(c) ? LSX_SWEEP_RAY_SERIAL : LSX_SWEEP_RAY_PER_LANE;
}

// the canonical description of everything that decides how this context associates its sums and launches its kernels
static std::string effective_options(const lsx_ctx* c)
{
    std::string s = "backend=hip-gfx950;";
    s += options_string(c->options);
    s += c->solver == LSX_SOLVER_PARABOLIC ? ";solver=parabolic" : ";solver=linear";
    s += use_ray_serial(c) ? ";mapping=ray-serial" : ";mapping=ray-per-lane";
    if (c->solver == LSX_SOLVER_PARABOLIC) s += per_class_launches(c) ? ";parabolic=classes" : ";parabolic=generic";
    s += ";classes=" + plan_class_string(*c);
    // which library this is: two ranks with different builds must not pass for alike (parallel.check_same_options)
    s += ";abi=" + std::to_string(LSX_ABI_VERSION) + ";build=" + lsx_build_id();
    return s;
}

int lsx_effective_options(const lsx_ctx* c, char* buf, size_t n)
{
    if (!c || !buf || n == 0) return fail(LSX_EINVAL, "lsx_effective_options: bad argument");
    const std::string s = effective_options(c);
    if (s.size() + 1 > n) return fail(LSX_EINVAL, "lsx_effective_options: %zu bytes needed", s.size() + 1);
    memcpy(buf, s.c_str(), s.size() + 1);
    return LSX_OK;
}

uint64_t lsx_options_signature(const lsx_ctx* c)
{
    return c ? fnv1a64(effective_options(c)) : 0;
}

int lsx_set_formal_solver(lsx_ctx* c, int32_t solver)
{
    if (!c || (solver != LSX_SOLVER_LINEAR && solver != LSX_SOLVER_PARABOLIC)) return fail(LSX_EINVAL, "lsx_set_formal_solver: bad argument");
    c->solver = solver;
    return LSX_OK;
}

int lsx_w2(int32_t device, int32_t n, const double* dtau, double* w0w1)
{
    if (n < 0 || (n > 0 && (!dtau || !w0w1))) return fail(LSX_EINVAL, "lsx_w2: bad argument");
    if (n == 0) return LSX_OK;
    std::lock_guard<std::mutex> lock(g_pw_mutex);
    PwScratch* q = nullptr;
    int rc = pw_scratch(device, (size_t)n * 3 * sizeof(double), &q);
    if (rc) return rc;
    double* din = reinterpret_cast<double*>(q->buf);
    double* dout = din + n;
    HIPCHK(hipMemcpyAsync(din, dtau, (size_t)n * 8, hipMemcpyHostToDevice, q->stream));
    hipLaunchKernelGGL(k_w2, dim3((n + 63) / 64), dim3(64), 0, q->stream, n, din, dout, q->exp_tab);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(w0w1, dout, (size_t)n * 16, hipMemcpyDeviceToHost, q->stream));
    HIPCHK(hipStreamSynchronize(q->stream));
    return LSX_OK;
}

int lsx_time_formal_sol(lsx_ctx* c, int32_t warmup, int32_t reps, double* ms_total, double* ms_sweep)
{
    if (!c || reps < 1 || warmup < 0) return fail(LSX_EINVAL, "lsx_time_formal_sol: bad argument");
    HIPCHK(hipSetDevice(c->device));
    int rc;
    for (int i = 0; i < warmup; ++i)
        if ((rc = enqueue_fs(c, false))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    double tot = 0.0, sw = 0.0, fin = 0.0, tail = 0.0;
    for (int i = 0; i < reps; ++i) {
        // a MALI iteration rebuilds the ray-serial sweeps' operand table once (behind stat_equil's read-back, lsx_sync): the timed
        // call pays for it too, so that the per-call figure covers every kernel an iteration runs for its formal solution
        c->optab_fresh = false;
        HIPCHK(hipEventRecord(c->evA, c->stream));
        if ((rc = enqueue_fs(c, true))) return rc;
        HIPCHK(hipEventRecord(c->evB, c->stream));
        HIPCHK(hipEventSynchronize(c->evB));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->evA, c->evB));
        tot += ms;
        // the sweep: from the fork to the end of the last class (the classes run side by side; the fast-continuum
        // pre-pass and epilogue run next to them and are not part of this figure unless they delay a class)
        HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        if (per_class_launches(c)) {
            float mx = 0.f;
            for (auto& k : c->classes)
                if (k.tdone) {
                    float t = 0.f;
                    HIPCHK(hipEventElapsedTime(&t, c->ev0, k.tdone));
                    if (c->opt_trace_classes && i == reps - 1) fprintf(stderr, "class npt=%d nl=%d linked=%d topo=%d fast=%d tiles=%zu (first tile %d): sweep done at %.3f ms\n", k.npt, k.nl, (int)k.linked, k.topo, (int)k.has_fast, k.tiles.size(), k.tiles[0], t);
                    mx = std::max(mx, t);
                }
            if (mx > 0.f) ms = mx;
        }
        sw += ms;
        {   // fork -> join minus the sweep span: what the fast-continuum epilogue of the class that finishes last adds behind the sweeps
            float fj = 0.f;
            HIPCHK(hipEventElapsedTime(&fj, c->ev0, c->ev1));
            tail += std::max(0.f, fj - ms);
        }
        HIPCHK(hipEventElapsedTime(&ms, c->ev1, c->ev2));
        fin += ms;
    }
    c->ms_sweep = sw / reps;
    c->ms_finish = fin / reps;
    c->ms_epi_tail = tail / reps;
    if (ms_total) *ms_total = tot / reps;
    if (ms_sweep) *ms_sweep = c->ms_sweep;
    return lsx_sync(c, nullptr, nullptr);
}

// HIP-only introspection used by bench.py.  what: 0 = algorithmic bytes per column the sweep
// kernel itself must move (B_alg without the C read / Gamma write of the epilogue), 1 = tiles per
// column, 2 = LDS bytes per workgroup (the largest class of the mapping in use), 3 = wavelengths per tile, 4 = ms of the Gamma epilogue in the
// last lsx_time_formal_sol, 5 = slab bytes per column (extra, non-algorithmic traffic of the design), 6 = ms between the end of the last
// class's sweep and the join (the exposed part of the fast-continuum epilogue)
double lsx_hip_info(const lsx_ctx* c, int32_t what)
{
    if (!c) return 0.0;
    switch (what) {
    case 0: return lsx_algorithmic_bytes_per_column(c) - 8.0 * c->Nspace * 2.0 * c->NL2tot;
    case 1: return (double)c->tiles.size();
    case 2: {   // the largest workgroup of the sweep the next formal solution launches (the ray-serial instances: operand rings of eight
                // rows per wave, lsx_plan.h -- independent of Nspace since round 5; folded classes: wider rows and the cross-section table)
        if (c->solver == LSX_SOLVER_PARABOLIC || !use_ray_serial(c)) return (double)c->lds_bytes;
        size_t b = 0;
        for (const auto& k : c->classes)
            b = std::max(b, k.rs ? (size_t)lsx_rs_lds_doubles(k.npt, c->Nspace, false, k.fold ? k.fold_nF : -1, k.epi, k.linked ? k.nl : 0) * sizeof(double) : k.lds_bytes);
        return (double)b;
    }
    case 3: return (double)c->L;
    case 4: return c->ms_finish;      // the Gamma epilogue (k_gamma_finish*) of the last lsx_time_formal_sol
    case 6: return c->ms_epi_tail;    // ... and what lay between the end of the last class's sweep and the join: fast-continuum epilogue kernels the sweeps did not hide
    case 5: return 8.0 * c->Nspace * 4.0 * (double)c->tile_slots.size();
    default: return 0.0;
    }
}

// HIP-only introspection for the tests: which sweep instantiations a context launches.  Returns the number of tile
// classes; for 0 <= idx < that number out[0..6] = per-ray slots (compile time, -1 generic), lines among them, tiles per
// column, launches so far, 1 if the class's tiles have linked continua, the two-line relation (TOPO), 1 if the class runs the
// ray-serial kernel (lsx_sweep_rs.hip) in this context.  idx == -1: out[0] =
// launches of the fused small-batch kernel.
int32_t lsx_hip_class_info(const lsx_ctx* c, int32_t idx, int64_t* out)
{
    if (!c) return 0;
    if (out && idx == -1) out[0] = c->fused_launches;
    if (out && idx >= 0 && idx < (int)c->classes.size()) {
        const SweepClass& k = c->classes[idx];
        out[0] = k.npt; out[1] = k.nl; out[2] = (int64_t)k.tiles.size(); out[3] = k.launches; out[4] = k.linked ? 1 : 0; out[5] = k.topo;
        out[6] = ((c->solver == LSX_SOLVER_PARABOLIC ? k.rsp : k.rs) && use_ray_serial(c)) ? 1 : 0;
    }
    return (int32_t)c->classes.size();
}

// diagnostic (tests/test_lds_hygiene.py): fill the LDS of every CU with signalling garbage.  LDS keeps its contents from one kernel
// to the next, and on an idle machine they are mostly zeros -- a kernel that reads LDS it has not written then looks correct.  After
// this call such a read returns NaN (round 5: a folded ray-serial tile without fast continua read four rows of cross-sections nobody
// had written; the failure showed up once in a dozen runs).  `rounds` launches of 4 x (CUs) workgroups of 64 kB.
__global__ void k_poison_lds(double* sink)
{
    extern __shared__ double sm[];
    const double bad = __builtin_nan("");
    for (int e = threadIdx.x; e < 8192; e += blockDim.x) sm[e] = bad;
    __syncthreads();
    if (sink && sm[(threadIdx.x * 37) & 8191] == 1.0) *sink = 1.0;      // (keeps the stores)
}
extern "C" int lsx_hip_poison_lds(int32_t device, int32_t rounds)
{
    if (hipSetDevice(device) != hipSuccess) return LSX_EDEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return LSX_EDEVICE;
    for (int r = 0; r < std::max(1, (int)rounds); ++r)
        hipLaunchKernelGGL(k_poison_lds, dim3((unsigned)(4 * prop.multiProcessorCount)), dim3(256), 65536, 0, (double*)nullptr);
    return hipDeviceSynchronize() == hipSuccess ? LSX_OK : LSX_EDEVICE;
}

// diagnostic: stream `gib` GiB once in the sweep's access shape (see k_calib_read); returns the bytes read
double lsx_hip_calibrate_read(int32_t device, double gib, int32_t seg)
{
    if (hipSetDevice(device) != hipSuccess || seg < 1 || seg > 64) return 0.0;
    const int per = 64 / seg;
    const long stride = 96;                 // doubles between segments (a 768-B row pitch, like a 96-point line)
    const long nwave_rows = 64;
    const long waves = (long)(gib * 1073741824.0 / 8.0 / (double)(nwave_rows * per * stride));
    const size_t n = (size_t)waves * nwave_rows * per * stride;
    double *buf = nullptr, *out = nullptr;
    if (hipMalloc((void**)&buf, n * 8) != hipSuccess) return 0.0;
    if (hipMalloc((void**)&out, (size_t)waves * 8) != hipSuccess) { (void)hipFree(buf); return 0.0; }
    (void)hipMemset(buf, 0, n * 8);
    (void)hipMemset(out, 0, (size_t)waves * 8);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_calib_read, dim3((unsigned)(waves / 4)), dim3(256), 0, 0, buf, out, seg, stride, nwave_rows);
    (void)hipDeviceSynchronize();
    (void)hipFree(buf);
    (void)hipFree(out);
    return (double)(waves / 4 * 4) * nwave_rows * per * seg * 8.0;
}

// diagnostic: copy the sweep kernel's stamp buffer (1024 records x 16 x u64) to host
int lsx_hip_debug_read(lsx_ctx* c, unsigned long long* out)
{
    if (!c || !out) return fail(LSX_EINVAL, "null");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->d_debug, 1024 * 16 * 8, hipMemcpyDeviceToHost));
    return LSX_OK;
}

// B_alg = 8 Ns [P SNl + 2 Nspect (bg) + 2 Nspect (Jdag, J) + 1 (sigma) + NLtot (n)
//                          + 2 NL2tot (C, Gamma) + 6] + 8 Nspect Nrays
double lsx_algorithmic_bytes_per_column(const lsx_ctx* c)
{
    const double P = c->phi_compact ? 1.0 : 2.0 * c->Nrays;
    const double per_depth = P * c->SNl + 4.0 * c->Nspect + (c->sca_per_lambda ? c->Nspect : 1.0) + c->NLtot + 2.0 * c->NL2tot + 6.0;
    return 8.0 * c->Nspace * per_depth + 8.0 * c->Nspect * c->Nrays;
}

} // extern "C"
