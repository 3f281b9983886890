// lsx_plan_capi.cpp -- TEST ENTRY of the host-side plan: builds the plan of a problem descriptor and verifies its
// invariants (every index the kernels will form from it stays inside its array).  Compiled only into the sanitizer
// library `liblsx_host_asan.so` (make asan) that tests/test_plan_sanitized.py drives under AddressSanitizer +
// UndefinedBehaviorSanitizer; the product library does not contain it.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "lsx_plan.h"

using namespace lsxd;

namespace {

struct Checker {
    std::string msg;
    bool ok = true;
    void fail(const char* fmt, long a = 0, long b = 0, long c = 0)
    {
        if (!ok) return;
        char buf[256];
        snprintf(buf, sizeof buf, fmt, a, b, c);
        msg = buf;
        ok = false;
    }
};

// intervals [first, first + len) must not overlap and must lie in [0, limit)
void disjoint(Checker& ck, std::vector<std::pair<size_t, size_t>> iv, size_t limit, const char* what)
{
    std::sort(iv.begin(), iv.end());
    size_t end = 0;
    for (auto& x : iv) {
        if (x.first < end) ck.fail((std::string(what) + ": blocks overlap at %ld").c_str(), (long)x.first);
        end = x.first + x.second;
        if (end > limit) ck.fail((std::string(what) + ": block [%ld, %ld) beyond %ld").c_str(), (long)x.first, (long)end, (long)limit);
    }
}

void verify(const LsxPlan& P, Checker& ck)
{
    const int Ns = P.Nspace, L = P.L;
    const size_t P_line = P.phi_compact ? 1 : 2 * (size_t)P.Nrays;
    if (L != LSX_WAVE / P.Nrays || L < 1) ck.fail("L = %ld", L);
    // tiles: a partition of the wavelength axis
    int la = 0, slot = 0;
    std::vector<std::pair<size_t, size_t>> phi_blocks, corr_blocks, pp_blocks;
    std::vector<int> tile_class(P.tiles.size(), -1);
    for (size_t c = 0; c < P.plan_classes.size(); ++c)
        for (int t : P.plan_classes[c].tiles) {
            if (t < 0 || t >= (int)P.tiles.size()) { ck.fail("class %ld lists tile %ld", (long)c, t); return; }
            if (tile_class[t] != -1) ck.fail("tile %ld is in two classes", t);
            tile_class[t] = (int)c;
        }
    for (size_t ti = 0; ti < P.tiles.size(); ++ti) {
        const DevTile& tl = P.tiles[ti];
        if (tl.la0 != la || tl.nla < 1 || tl.nla > L) ck.fail("tile %ld: la0 %ld nla %ld", (long)ti, tl.la0, tl.nla);
        la += tl.nla;
        if (tl.slot0 != slot) ck.fail("tile %ld: slot0 %ld, expected %ld", (long)ti, tl.slot0, slot);
        slot += tl.nP + tl.nF + tl.nX;
        if (tl.nX != (tl.nK > 0 ? tl.nL : 0)) ck.fail("tile %ld: correction slots", (long)ti);
        for (int x = 0; x < tl.nX && slot <= (int)P.slots.size(); ++x) {
            const int q = tl.slot0 + tl.nP + tl.nF + x;
            if (P.tile_slots[q] != P.tile_slots[tl.slot0 + x] || !P.tile_slot_fast[q] || P.slots[q].len != 0 || (P.slots[q].flags & SLOT_LINE))
                ck.fail("tile %ld: correction slot %ld", (long)ti, x);
        }
        if (slot > (int)P.slots.size()) { ck.fail("tile %ld: slots run past the table", (long)ti); return; }
        if (tl.nL > tl.nP || tl.nK > tl.nF || tl.nP > LSX_MAX_PER_RAY || tl.nF > LSX_MAX_FAST) ck.fail("tile %ld: counts", (long)ti);
        if (tile_class[ti] < 0) { ck.fail("tile %ld is in no class", (long)ti); return; }
        const PlanClass& k = P.plan_classes[tile_class[ti]];
        if (k.npt >= 0) {
            if (k.npt != tl.nP || k.nl != tl.nL || k.linked != (tl.nK > 0)) ck.fail("tile %ld does not have its class's shape", (long)ti);
            if (!lsx_sweep_instance_exists(k.npt, k.nl, k.linked, k.topo)) ck.fail("class of tile %ld has no compiled instance (code %ld)", (long)ti, k.code());
        } else if (k.linked != (tl.nK > 0)) ck.fail("generic tile %ld: linked flag", (long)ti);
        // every transition active in the tile appears exactly once among its slots
        std::vector<int> seen(P.Ntrans, 0);
        int nlinked = 0;
        for (int u = 0; u < tl.nP + tl.nF; ++u) {
            const DevSlot& s = P.slots[tl.slot0 + u];
            if (s.trans < 0 || s.trans >= P.Ntrans) { ck.fail("tile %ld slot %ld: transition id", (long)ti, u); return; }
            seen[s.trans]++;
            const DevTrans& h = P.htrans[s.trans];
            if (P.tile_slots[tl.slot0 + u] != s.trans) ck.fail("tile %ld slot %ld: tile_slots disagrees", (long)ti, u);
            const bool line = (s.flags & SLOT_LINE) != 0, fast = (s.flags & SLOT_FAST) != 0;
            if (line != (h.is_line != 0) || line != (u < tl.nL) || fast != (u >= tl.nP) || (P.tile_slot_fast[tl.slot0 + u] != 0) != fast)
                ck.fail("tile %ld slot %ld: kind / order", (long)ti, u);
            nlinked += (s.flags & SLOT_LINKED) ? 1 : 0;
            if (s.first < tl.la0 || s.first < h.Nblue || s.len < 1 || s.first + s.len > tl.la0 + tl.nla || s.first + s.len > h.Nblue + h.Nlam)
                ck.fail("tile %ld slot %ld: block [first, first + len)", (long)ti, u);
            if (s.wl_off < 0 || (size_t)s.wl_off + h.Nlam > P.wl.size() || P.wl.size() != P.alpha.size()) ck.fail("tile %ld slot %ld: wl table", (long)ti, u);
            if (s.li < 0 || s.lj >= P.NLtot || s.li >= s.lj) ck.fail("tile %ld slot %ld: levels", (long)ti, u);
            if (line) {
                phi_blocks.push_back({(size_t)s.base, (size_t)s.len * P_line * Ns});
                if (s.wphi_off < 0 || s.wphi_off + Ns > P.Nlines * Ns) ck.fail("tile %ld slot %ld: wphi row", (long)ti, u);
            } else if (s.base < 0 || s.base + Ns > P.Ncont * Ns) ck.fail("tile %ld slot %ld: nsr row", (long)ti, u);
            if (k.npt < 0 && !fast) {
                if ((s.flags & (SLOT_LI_CELL | SLOT_UI_READ)) && s.ci >= k.ncell_lev) ck.fail("tile %ld slot %ld: level cell", (long)ti, u);
                if ((s.flags & SLOT_LJ_CELL) && s.cj >= k.ncell_lev) ck.fail("tile %ld slot %ld: level cell", (long)ti, u);
                if ((s.flags & SLOT_ETA_CELL) && s.ca >= k.ncell_atom) ck.fail("tile %ld slot %ld: atom cell", (long)ti, u);
            }
        }
        if (nlinked != tl.nK) ck.fail("tile %ld: linked continua %ld vs nK %ld", (long)ti, nlinked, tl.nK);
        for (int t = 0; t < P.Ntrans; ++t) {
            bool any = false;
            for (int q = tl.la0; q < tl.la0 + tl.nla; ++q) any = any || P.active[(size_t)t * P.Nspect + q];
            if ((int)any != seen[t]) ck.fail("tile %ld: transition %ld appears %ld times", (long)ti, t, seen[t]);
        }
        if (tl.nK > 0) {
            corr_blocks.push_back({(size_t)tl.corr_off, (size_t)tl.nL * 3 * Ns * L});
            pp_blocks.push_back({(size_t)tl.pp_off, (size_t)tl.nL * Ns * L});
            if (tl.nL < 1) ck.fail("tile %ld: linked continua without a line", (long)ti);
        }
        // fast lists
        const bool in_fast = std::find(P.fast_tiles.begin(), P.fast_tiles.end(), (int)ti) != P.fast_tiles.end();
        if (in_fast != (tl.nF > 0)) ck.fail("tile %ld: fast_tiles membership", (long)ti);
        int homes = 0, khomes = 0;
        for (int v = 0; v < LSX_FGC_LISTS; ++v) {
            homes += (int)std::count(P.fast_cols[v].begin(), P.fast_cols[v].end(), (int)ti);
            khomes += (int)std::count(k.fast_cols[v].begin(), k.fast_cols[v].end(), (int)ti);
            if (std::count(P.fast_cols[v].begin(), P.fast_cols[v].end(), (int)ti) && (tl.fast_simple < 2 || v != fgc_list(tl) || (v & 3) > 2)) ck.fail("tile %ld: wrong column-mapped list %ld", (long)ti, v);
        }
        homes += (int)std::count(P.fast_rest.begin(), P.fast_rest.end(), (int)ti);
        khomes += (int)std::count(k.fast_rest.begin(), k.fast_rest.end(), (int)ti);
        if (homes != (tl.nF > 0) || khomes != (tl.nF > 0)) ck.fail("tile %ld: in %ld epilogue lists (class: %ld)", (long)ti, homes, khomes);
    }
    if (la != P.Nspect) ck.fail("tiles cover %ld of %ld wavelengths", la, P.Nspect);
    if (slot != (int)P.slots.size() || P.slots.size() != P.tile_slots.size() || P.slots.size() != P.tile_slot_fast.size()) ck.fail("slot tables: sizes");
    disjoint(ck, phi_blocks, P.phi_col >= 2 ? P.phi_col - 2 : 0, "phi_T");
    disjoint(ck, corr_blocks, P.corr_col, "corr_T");
    disjoint(ck, pp_blocks, P.pp_col, "Psi3_T");
    if (P.til_col != P.tiles.size() * (size_t)L * Ns) ck.fail("til_col");
    if (P.phi_col * 8 > 0x7fffffffu || P.corr_col * 8 > 0x7fffffffu || P.til_col * 8 > 0x7fffffffu) ck.fail("a column block exceeds 32-bit byte offsets");
    for (auto& k : P.plan_classes) {
        const int cl = k.npt >= 0 ? 0 : k.ncell_lev, ca = k.npt >= 0 ? 0 : k.ncell_atom;
        const size_t need = (size_t)lsx_sweep_lds(k.npt, k.linked, Ns, cl, ca).total * 8;
        if (k.lds_bytes < need || k.lds_bytes > 64 * 1024) ck.fail("class code %ld: LDS %ld B, needs %ld", k.code(), (long)k.lds_bytes, (long)need);
        if (k.tiles.empty()) ck.fail("empty class");
    }
    if (P.wave.size() != (size_t)P.Nspect || P.u_la.size() != (size_t)P.Nspect || P.zmu.size() != (size_t)P.Nrays) ck.fail("table sizes");
    if ((int)P.cont_li.size() != P.Ncont) ck.fail("continuum list");
}

} // namespace

extern "C" {

// summary[16]: tiles, slots, classes, phi_col, corr_col, pp_col, til_col, lds_bytes, static_max, nF_max, Ncont, generic tiles,
// fast tiles, linked tiles, per-ray slots in all, lanes carrying a wavelength per 1000
// tiles_out[max_tiles][8]: la0, nla, nP, nF, nL, nK, class code, epilogue kind
int lsx_plan_probe(const lsx_problem* d, uint32_t option_bits, int64_t* summary, int32_t* tiles_out, int32_t max_tiles, char* err, int32_t errlen)
{
    PlanOptions opt;
    opt.no_linked = option_bits & 1; opt.natural_tiles = option_bits & 2; opt.no_topo = option_bits & 4; opt.fast_rows = option_bits & 8;
    opt.order_by_cost = option_bits & 16; opt.occ_wg = (int)(option_bits >> 8);
    LsxPlan P;
    std::string e;
    int rc = plan_build(d, opt, &P, &e);
    if (rc == LSX_OK) {
        Checker ck;
        verify(P, ck);
        if (!ck.ok) { rc = -1000; e = "plan invariant violated: " + ck.msg; }
    }
    if (err && errlen > 0) { strncpy(err, e.c_str(), (size_t)errlen - 1); err[errlen - 1] = 0; }
    if (rc) return rc;
    long generic = 0, linked = 0, perray = 0, lanes = 0;
    std::vector<int> code(P.tiles.size(), 0);
    for (auto& k : P.plan_classes)
        for (int t : k.tiles) { code[t] = k.code(); generic += k.npt < 0; }
    for (auto& tl : P.tiles) { linked += tl.nK > 0; perray += tl.nP; lanes += tl.nla; }
    if (summary) {
        const int64_t s[16] = {(int64_t)P.tiles.size(), (int64_t)P.slots.size(), (int64_t)P.plan_classes.size(), (int64_t)P.phi_col, (int64_t)P.corr_col,
                               (int64_t)P.pp_col, (int64_t)P.til_col, (int64_t)P.lds_bytes, P.static_max, P.nF_max, P.Ncont, generic,
                               (int64_t)P.fast_tiles.size(), linked, perray, P.tiles.empty() ? 0 : 1000 * lanes / ((long)P.tiles.size() * P.L)};
        memcpy(summary, s, sizeof s);
    }
    for (int t = 0; tiles_out && t < (int)P.tiles.size() && t < max_tiles; ++t) {
        const DevTile& tl = P.tiles[t];
        const int32_t r[8] = {tl.la0, tl.nla, tl.nP, tl.nF, tl.nL, tl.nK, code[t], tl.fast_simple};
        memcpy(tiles_out + 8 * t, r, sizeof r);
    }
    return LSX_OK;
}

// What lsx_effective_options / lsx_options_signature of the product report for a context of this problem, as far as the host side
// decides it: the options (environment defaults + explicit list, lsx_plan.cpp) and the plan's class list.  -> the FNV-1a hash of
// the string (0 and a message in `buf` if the list or the problem is refused).  tests/test_distributed_gloo.py: a rank whose
// environment differs is refused by parallel.check_same_options.
extern "C" uint64_t lsx_plan_signature(const lsx_problem* d, const char* options, char* buf, int32_t buflen)
{
    CtxOptions o;
    options_from_env(&o);
    std::string e;
    LsxPlan P;
    int rc = options_apply(options, &o, &e);
    if (rc == LSX_OK) rc = plan_build(d, o.plan, &P, &e);
    const std::string s = rc ? e : "backend=host-plan;" + options_string(o) + ";classes=" + plan_class_string(P);
    if (buf && buflen > 0) { strncpy(buf, s.c_str(), (size_t)buflen - 1); buf[buflen - 1] = 0; }
    return rc ? 0 : fnv1a64(s);
}

// The compiled template instances, from the lists the kernels are instantiated from (lsx_plan.h): which = 0 the one-ray-per-lane
// sweep (LSX_SWEEP_INSTANCES), 1 the ray-serial sweep = the parabolic rule's compile-time classes (LSX_RS_INSTANCES).  -> count;
// codes[i] = lsx_class_code(per-ray slots, lines, linked, two-line relation).  tests/test_instance_ledger.py: every instance that
// is compiled must be planned -- and run against the oracle -- by some GPU test.
int32_t lsx_plan_instances(int32_t which, int32_t* codes, int32_t max)
{
    std::vector<int32_t> v;
#define LSX_X(NPT, NL, LK, TOPO) v.push_back(lsx_class_code(NPT, NL, LK, TOPO));
    if (which == 0) { LSX_SWEEP_INSTANCES(LSX_X) }
    else if (which == 1) { LSX_RS_INSTANCES(LSX_X) }
    else if (which == 2) { LSX_RSP_INSTANCES(LSX_X) }        // the ray-serial instances of the parabolic rule
#undef LSX_X
    for (int32_t i = 0; codes && i < (int32_t)v.size() && i < max; ++i) codes[i] = v[i];
    return (int32_t)v.size();
}

// the ray-serial eligibility of a problem's classes: -> number of classes; out[c][2] = class code, 1 if the class has a
// ray-serial instance AND the context's shape admits the kernel (five rays, wavelength-independent scattering, 32-bit offsets),
// + 2 if it has one for the parabolic rule as well
int32_t lsx_plan_rs_classes(const lsx_problem* d, int32_t* out, int32_t max)
{
    PlanOptions opt;
    LsxPlan P;
    std::string e;
    if (plan_build(d, opt, &P, &e) != LSX_OK) return -1;
    int32_t n = 0;
    for (auto& k : P.plan_classes) {
        if (out && n < max) { out[2 * n] = k.code(); out[2 * n + 1] = (k.rs ? 1 : 0) | (k.rsp ? 2 : 0); }
        ++n;
    }
    return n;
}

} // extern "C"

// what lsx_grid.cpp expects from the runtime (the product defines these in lsx_hip.hip)
#include <cstdarg>
namespace lsxd {
static thread_local std::string g_err_host;
int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err_host = buf;
    return code;
}
} // namespace lsxd
extern "C" const char* lsx_last_error(void) { return lsxd::g_err_host.c_str(); }
