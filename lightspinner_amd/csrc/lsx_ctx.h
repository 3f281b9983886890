// lsx_ctx.h -- the context object behind the C ABI and the helpers the host-side translation units share
// (lsx_hip.hip: runtime + small kernels, lsx_setup.hip: set-up chain).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/lsx.h"
#include "lsx_dev.h"

namespace lsxd {
extern thread_local std::string g_err;
int fail(int code, const char* fmt, ...);    // records the message lsx_last_error returns; -> code
} // namespace lsxd

#define HIPCHK(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return lsxd::fail(LSX_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

struct SweepClass {           // tiles that run the same kernel instantiation, launched on their own stream
    int npt = -1;              // compile-time per-ray slot count, -1 = generic
    long launches = 0;         // how often this class's kernel has been launched (introspection for the tests)
    int nl = 0;                // lines among them (compile-time too)
    bool linked = false;       // the class's tiles have linked continua (compile-time too)
    int topo = 0;              // two-line classes: known relation of the two lines (lsx_sweep.hip, TOPO)
    bool has_fast = false;     // some tile of the class has fast continua: the class reads the pre-pass output
    hipEvent_t tdone = nullptr; // timed runs: end of this class's launch
    std::vector<int> fast_tiles; // the class's tiles that have fast continua
    int* d_fast_tiles = nullptr;
    std::vector<int> fast_cols[4], fast_rest;   // ... split by the kernel that builds their Gamma slabs (k_fast_gamma_cols
    int *d_fast_cols[4] = {nullptr, nullptr, nullptr, nullptr}, *d_fast_rest = nullptr;   // instance / k_fast_gamma)
    std::vector<int> tiles;
    int* d_tiles = nullptr;
    int ncell_lev = 1, ncell_atom = 1;
    size_t lds_bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
};

struct lsx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int Nspace = 0, Nrays = 0, Nspect = 0, Natoms = 0, Ntrans = 0, ncol = 0;
    int NLtot = 0, NL2tot = 0, Nlines = 0, SNl = 0, SNc = 0;
    int sca_per_lambda = 0, phi_compact = 0;
    std::vector<int> Nlevel, lev_off, lev2_off;
    std::vector<lsx_transition> trans;
    std::vector<DevTrans> htrans;
    std::vector<DevTile> tiles;
    std::vector<int> tile_slots;
    std::vector<DevSlot> slots;
    DevSlot* d_slots = nullptr;
    int L = 0;
    std::vector<SweepClass> classes;
    hipEvent_t ev_fork = nullptr;
    double ms_sweep = 0.0, ms_finish = 0.0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    size_t lds_bytes = 0;
    // device: column independent
    double *d_wavelength = nullptr, *d_zmu = nullptr, *d_wmuh = nullptr, *d_wl = nullptr, *d_alpha = nullptr,
           *d_u_la = nullptr;
    uint8_t* d_active = nullptr;
    DevTrans* d_trans = nullptr;
    DevTile* d_tiles = nullptr;
    std::vector<uint8_t> tile_slot_fast;     // per slot: fast continuum (its slabs use the first direction entry only)
    int *d_tile_slots = nullptr, *d_Nlevel = nullptr, *d_lev2_off = nullptr;
    // device: per column
    double *d_height = nullptr, *d_temperature = nullptr, *d_nStar = nullptr, *d_nTotal = nullptr, *d_n = nullptr,
           *d_C = nullptr, *d_Gamma = nullptr, *d_wphi = nullptr, *d_bgchi = nullptr, *d_bgeta = nullptr,
           *d_sca = nullptr, *d_phi = nullptr, *d_E = nullptr, *d_corr = nullptr, *d_Psi3 = nullptr, *d_J[2] = {nullptr, nullptr}, *d_I = nullptr,
           *d_Gpart = nullptr, *d_dJpart = nullptr, *d_dJcol = nullptr, *d_dPcol = nullptr, *d_res = nullptr;
    unsigned long long* d_singular = nullptr;
    std::vector<uint8_t> phi_set;    // per column: line profiles have been handed over or built
    size_t n_phi_set = 0;
    int solver = 0;               // LSX_SOLVER_* (lsx_set_formal_solver)
    bool opt_se_lds = false, opt_trace_classes = false, opt_serial = false;   // LSX_SE_LDS / LSX_TRACE_CLASSES, read once in lsx_create
    long fused_launches = 0;
    uint8_t* d_colmask = nullptr; // per-column activity, nullptr = all active
    double *d_bgxchi = nullptr, *d_bgxeta = nullptr, *d_Psi2 = nullptr; // fast-continuum side arrays
    std::vector<int> fast_tiles;
    int* d_fast_tiles = nullptr;
    std::vector<int> fast_cols[4], fast_rest;
    int *d_fast_cols[4] = {nullptr, nullptr, nullptr, nullptr}, *d_fast_rest = nullptr;
    int *d_cont_li = nullptr, *d_cont_lj = nullptr;
    double* d_exp2_tab = nullptr;
    double* d_voigt_w = nullptr;
    double *d_muz = nullptr, *d_wmu = nullptr;
    int nF_max = 0, Ncont = 0, static_max = -1, nL_linked_max = 0;
    bool fast_generic = false;
    double* d_nsr = nullptr;     // [col][Ncont][k] nStar_i / nStar_j of the continua
    std::vector<int> cont_li, cont_lj;
    double* d_debug = nullptr;   // 64 x 16 x 8 B, diagnostic builds of the sweep kernel write stamps here
    int jcur = 0; // d_J[jcur] holds the current J (Jdag of the next call)
    size_t phi_col = 0, phi_in_col = 0, corr_col = 0, pp_col = 0, sca_col = 0, til_col = 0;
    bool any_cont = false;      // some tile has a continuum: E_T is kept
    // set-up chain (lsx_setup.hip): atomic data tables and what lsx_set_atmosphere derives per column
    char *d_sa_atoms = nullptr, *d_sa_lines = nullptr, *d_sa_colls = nullptr;
    double *d_sa_spl = nullptr, *d_sa_levE = nullptr, *d_sa_levg = nullptr, *d_sa_levnD = nullptr;
    int32_t* d_sa_levdZ = nullptr;
    bool have_atomic_data = false;
    double *d_vBroad = nullptr, *d_aDamp = nullptr;     // [col][Natoms][k], [col][Nlines][k]
    // staging
    double* d_stage = nullptr;
    size_t stage_doubles = 0;
    double* h_pinned = nullptr; // host mirror of d_res
    double last_dJ = 0.0, last_dP = 0.0;
    bool fs_pending = false, se_pending = false;
    hipEvent_t evA = nullptr, evB = nullptr;
};

namespace lsxd {

template <typename T>
int dmalloc(T** p, size_t count)
{
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
    if (e != hipSuccess) return fail(LSX_EDEVICE, "hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e));
    return LSX_OK;
}

template <typename T>
int upload(T** dptr, const std::vector<T>& v, hipStream_t st)
{
    int rc = dmalloc(dptr, v.size());
    if (rc) return rc;
    if (!v.empty()) HIPCHK(hipMemcpyAsync(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
    return LSX_OK;
}

int ensure_stage(lsx_ctx* c, size_t doubles);    // grow the context's staging buffer
int rebuild_derived(lsx_ctx* c, size_t col0, size_t ncol);     // continuum g_ij tables + nStar ratios from (nStar, T)
int profiles_from_device(lsx_ctx* c, size_t col0, size_t ncol, const double* dA, const double* dV, const double* dL);
void mark_profiles_set(lsx_ctx* c, size_t col0, size_t ncol);

} // namespace lsxd
