// lsx_ctx.h -- the context object behind the C ABI and the helpers the host-side translation units share
// (lsx_hip.hip: runtime + small kernels, lsx_setup.hip: set-up chain).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "../../include/lsx.h"
#include "lsx_dev.h"
#include "lsx_plan.h"

namespace lsxd {
extern thread_local std::string g_err;
int fail(int code, const char* fmt, ...);    // records the message lsx_last_error returns; -> code
} // namespace lsxd

#define HIPCHK(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return lsxd::fail(LSX_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

struct SweepClass : lsxd::PlanClass {   // a plan class + what the runtime keeps for it: its own stream, events, device tile lists
    long launches = 0;         // how often this class's kernel has been launched (introspection for the tests)
    hipEvent_t tdone = nullptr; // timed runs: end of this class's launch
    int* d_fast_tiles = nullptr;
    int *d_fast_cols[LSX_FGC_LISTS] = {}, *d_fast_rest = nullptr;
    int* d_tiles = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
};

struct FsGraphKey {            // what a captured formal solution's kernel arguments depend on beyond the context's fixed state
    int jcur; const void* results; int clear_dp, solver, policy, policy_columns; const void* colmask;
    bool operator==(const FsGraphKey& o) const
    {
        return jcur == o.jcur && results == o.results && clear_dp == o.clear_dp && solver == o.solver && policy == o.policy &&
               policy_columns == o.policy_columns && colmask == o.colmask;
    }
};

struct lsx_ctx : lsxd::LsxPlan {        // the plan (lsx_plan.h: dimensions, tables, tile schedule, strides, launch shapes) + device state
    int device = 0;
    lsxd::CtxOptions options;       // what lsx_create_with_options ended up with (environment defaults + explicit list)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int ncol = 0;
    DevSlot* d_slots = nullptr;
    std::vector<SweepClass> classes;     // plan_classes, in launch order
    hipEvent_t ev_fork = nullptr;
    double ms_sweep = 0.0, ms_finish = 0.0;
    double ms_epi_tail = 0.0;   // lsx_time_formal_sol: join - end of the last class's sweep (exposed fast-continuum epilogue)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    // device: column independent
    double *d_wavelength = nullptr, *d_zmu = nullptr, *d_wmuh = nullptr, *d_wl = nullptr, *d_alpha = nullptr,
           *d_u_la = nullptr, *d_fgtab = nullptr;
    int* d_level_atom = nullptr;
    uint8_t* d_active = nullptr;
    DevTrans* d_trans = nullptr;
    DevTile* d_tiles = nullptr;
    int *d_tile_slots = nullptr, *d_Nlevel = nullptr, *d_lev2_off = nullptr, *d_fin_ptr = nullptr, *d_fin_idx = nullptr, *d_atom_ptr = nullptr, *d_atom_slots = nullptr;
    bool dp_zeroed = false;          // the Gamma epilogue has zeroed dPcol / the singular flag for the next stat_equil
    bool opt_finish_big = false;     // LSX_FINISH_BIG=1: the many-column Gamma epilogue also for small batches (tests)
    bool opt_graph = false;          // LSX_GRAPH=1: a formal solution's launches as a captured HIP graph (enqueue_fs; measurement)
    std::vector<std::pair<FsGraphKey, hipGraphExec_t>> fs_graphs;
    bool opt_fused_epilogue = false; // LSX_FUSED_EPILOGUE=1: the fused launch also runs its fast tiles' Gamma epilogue (measured: slower, see enqueue_fs)
    int opt_abl_fast = 0;            // LSX_ABL_FUSED_FAST: timing ablations of the fused launch's fast-continuum work
    bool opt_no_fused_fast = false;  // LSX_NO_FUSED_FAST=1: small batches launch the fast-continuum kernels around the fused sweep (tests)
    // device: per column
    double *d_height = nullptr, *d_temperature = nullptr, *d_nStar = nullptr, *d_nTotal = nullptr, *d_n = nullptr,
           *d_C = nullptr, *d_Gamma = nullptr, *d_wphi = nullptr, *d_bgchi = nullptr, *d_bgeta = nullptr, *d_bgce = nullptr, *d_bgxce = nullptr,
           *d_sca = nullptr, *d_phi = nullptr, *d_E = nullptr, *d_corr = nullptr, *d_Psi3 = nullptr, *d_J[2] = {nullptr, nullptr}, *d_I = nullptr,
           *d_Gpart = nullptr, *d_dJpart = nullptr, *d_dJcol = nullptr, *d_dPcol = nullptr, *d_res = nullptr;
    unsigned long long* d_singular = nullptr;
    std::vector<uint8_t> phi_set;    // per column: line profiles have been handed over or built
    size_t n_phi_set = 0;
    int solver = 0;               // LSX_SOLVER_* (lsx_set_formal_solver)
    int sweep_policy = LSX_SWEEP_AUTO;   // lsx_set_sweep_policy: which wavefront mapping the formal solution runs ...
    int policy_columns = 0;              // ... and, under AUTO, the column count that decides (0: this context's own)
    bool opt_se_lds = false, opt_trace_classes = false, opt_serial = false;   // LSX_SE_LDS / LSX_TRACE_CLASSES / LSX_SERIAL, read once in lsx_create
    long fused_launches = 0;
    uint8_t* d_colmask = nullptr; // per-column activity, nullptr = all active
    double *d_bgxchi = nullptr, *d_bgxeta = nullptr, *d_Psi2 = nullptr; // fast-continuum side arrays
    int* d_fast_tiles = nullptr;
    int *d_fast_cols[LSX_FGC_LISTS] = {}, *d_fast_rest = nullptr;
    int *d_cont_li = nullptr, *d_cont_lj = nullptr;
    double* d_exp2_tab = nullptr;
    double* d_voigt_w = nullptr;
    double *d_muz = nullptr, *d_wmu = nullptr;
    double* d_nsr = nullptr;     // [col][Ncont][k] nStar_i / nStar_j of the continua
    double* d_optab = nullptr;   // ray-serial sweep: per-depth operands per (column group, transition) + geometry (lsx_plan.h), made on first use
    int* d_trans_row = nullptr;  // per transition: row of wphi / nsr
    bool optab_fresh = false;    // d_optab was built from the current n, wphi, nStar ratios, heights and sigma
    double* d_debug = nullptr;   // 64 x 16 x 8 B, diagnostic builds of the sweep kernel write stamps here
    int jcur = 0; // d_J[jcur] holds the current J (Jdag of the next call)
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
    double* h_n = nullptr;      // pinned host mirror of the populations (lsx_sync_begin_populations), made on first use
    bool h_n_pending = false, h_n_valid = false;      // a read-back of n is in flight / has been collected by lsx_sync_end
    double last_dJ = 0.0, last_dP = 0.0;
    bool fs_pending = false, se_pending = false;
    hipEvent_t evA = nullptr, evB = nullptr;
    // pipelined MALI loop (include/lsx.h: lsx_sync_begin / lsx_formal_sol_gamma_speculative / lsx_discard_formal_sol): the second set
    // of the buffers a formal solution overwrites (J already is a pair), and the read-back in flight
    double *d_I_alt = nullptr, *d_Gamma_alt = nullptr, *d_res_alt = nullptr;
    bool spec_valid = false;         // the last formal solution was speculative and nothing has built on it: it can be discarded
    bool spec_dp_zeroed = false;     // dp_zeroed as it was before that call
    bool spec_fs_pending = false;    // fs_pending likewise
    double spec_last_dJ = 0.0;       // last_dJ likewise
    bool mon_spec = false;           // the read-back in flight was begun while a speculative call could still be discarded
    bool mon_outstanding = false, mon_fs = false, mon_se = false;
    hipEvent_t ev_mon = nullptr;
};

namespace lsxd {

// the column count the bit-relevant kernel choices are made for (include/lsx.h, lsx_set_sweep_policy)
inline int policy_ncol(const lsx_ctx* c) { return c->policy_columns > 0 ? c->policy_columns : c->ncol; }
// the ray-serial mapping (classes that have an instance of it for the context's rule) or one ray per lane
inline bool use_ray_serial(const lsx_ctx* c)
{
    if (!c->rs_ok || c->sweep_policy == LSX_SWEEP_RAY_PER_LANE) return false;
    return c->sweep_policy == LSX_SWEEP_RAY_SERIAL || policy_ncol(c) >= c->rs_min_columns;
}
// one launch per tile class on forked streams (else: one fused launch on the context's stream).  For the linear rule the two
// give the same bits per mapping, so the context's OWN size decides unless the ray-serial kernels run (they exist per class
// only); the parabolic rule's compile-time classes and its generic instance differ in the last bits: the policy count decides.
inline bool per_class_launches(const lsx_ctx* c)
{
    if (c->solver == LSX_SOLVER_PARABOLIC) return policy_ncol(c) >= 32 || use_ray_serial(c);
    return c->ncol >= 32 || use_ray_serial(c);
}

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
