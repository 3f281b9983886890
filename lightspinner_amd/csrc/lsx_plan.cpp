// lsx_plan.cpp -- builds the host-side plan of a context (lsx_plan.h).  Plain C++: no HIP call, no device memory.
//
// Reference lines restated here:
//   rh_method.py:157-196   wlambda: trapezoid weights on a transition's own wavelength range
//   rh_method.py:268-286   the per-transition constants of uv()
//   rh_method.py:451, 455  wla = wlambda wphi / hc (lines), wlambda / lambda / h (continua)
//   atomic_set.py:412-424  [Nblue, Nblue + Nlambda) and the `active` table the plan groups wavelengths by
#include "lsx_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace lsxd {
namespace {

constexpr double kCLight = 2.99792458E+08;
constexpr double kHPlanck = 6.6260755E-34;
constexpr double kNM_TO_M = 1.0E-09;
constexpr double kHC = kHPlanck * kCLight;

int perr(std::string* err, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (err) *err = buf;
    return code;
}

double wlambda(const std::vector<double>& wave, const lsx_transition& t, int lt)
{
    // rh_method.py:157-196 (single index form)
    const double* wl = wave.data() + t.Nblue;
    const double dopplerWidth = t.is_line ? kCLight / t.lambda0 : 1.0;
    const int N = t.Nlambda;
    if (lt == 0) return 0.5 * (wl[1] - wl[0]) * dopplerWidth;
    if (lt == N - 1) return 0.5 * (wl[N - 1] - wl[N - 2]) * dopplerWidth;
    return 0.5 * (wl[lt + 1] - wl[lt - 1]) * dopplerWidth;
}

} // namespace

int plan_build(const lsx_problem* d, const PlanOptions& opt, LsxPlan* out, std::string* err)
{
    if (!d || !out) return perr(err, LSX_EINVAL, "lsx_create: null argument");
    if (d->abi_version != LSX_ABI_VERSION) return perr(err, LSX_EINVAL, "lsx_create: ABI version mismatch");
    if (d->Nspace < 3) return perr(err, LSX_EINVAL, "lsx_create: Nspace must be >= 3 (formal_solver.py:120-139)");
    if (d->Nrays < 1 || d->Nspect < 1 || d->Natoms < 1 || d->Ntrans < 0) return perr(err, LSX_EINVAL, "lsx_create: bad dimensions");
    if (d->Nrays > LSX_WAVE) return perr(err, LSX_EUNSUPPORTED, "lsx_create: Nrays > 64 is not supported by this build");
    if (d->Natoms > LSX_MAX_ATOMS) return perr(err, LSX_EUNSUPPORTED, "lsx_create: more than %d active atoms", LSX_MAX_ATOMS);
    if (!d->wavelength || !d->muz || !d->wmu || !d->Nlevel || (d->Ntrans > 0 && (!d->trans || !d->active)))
        return perr(err, LSX_EINVAL, "lsx_create: null table in the problem descriptor");
    LsxPlan& P = *out;
    P = LsxPlan();
    P.Nspace = d->Nspace; P.Nrays = d->Nrays; P.Nspect = d->Nspect; P.Natoms = d->Natoms; P.Ntrans = d->Ntrans;
    P.sca_per_lambda = d->sca_per_lambda ? 1 : 0;
    P.phi_compact = d->phi_compact ? 1 : 0;
    P.L = LSX_WAVE / d->Nrays;
    const int Ns = P.Nspace, Nspect = P.Nspect;
    for (int a = 0; a < P.Natoms; ++a) {
        if (d->Nlevel[a] < 2) { return perr(err, LSX_EINVAL, "lsx_create: Nlevel < 2"); }
        P.Nlevel.push_back(d->Nlevel[a]);
        P.lev_off.push_back(P.NLtot);
        P.lev2_off.push_back(P.NL2tot);
        P.NLtot += d->Nlevel[a];
        P.NL2tot += d->Nlevel[a] * d->Nlevel[a];
    }

    P.wave.assign(d->wavelength, d->wavelength + Nspect);
    std::vector<double>& wave = P.wave;
    std::vector<double>&wl = P.wl, &alpha = P.alpha;
    P.active.assign(d->active, d->active + (size_t)P.Ntrans * Nspect);
    std::vector<uint8_t>& active = P.active;
    int alpha_in = 0;
    for (int t = 0; t < P.Ntrans; ++t) {
        const lsx_transition& tr = d->trans[t];
        if (tr.atom < 0 || tr.atom >= P.Natoms || tr.i < 0 || tr.j >= P.Nlevel[tr.atom] || tr.i >= tr.j || tr.Nblue < 0 ||
            tr.Nlambda < 2 || tr.Nblue + tr.Nlambda > Nspect) {
            return perr(err, LSX_EINVAL, "lsx_create: inconsistent transition table entry %d", t);
        }
        P.trans.push_back(tr);
        DevTrans h{};
        h.atom = tr.atom; h.is_line = tr.is_line ? 1 : 0;
        h.li = P.lev_off[tr.atom] + tr.i; h.lj = P.lev_off[tr.atom] + tr.j;
        h.Nblue = tr.Nblue; h.Nlam = tr.Nlambda;
        h.wl_off = (int)wl.size();
        const int Nl = P.Nlevel[tr.atom];
        h.gam_ij = P.lev2_off[tr.atom] + tr.i * Nl + tr.j;
        h.gam_ji = P.lev2_off[tr.atom] + tr.j * Nl + tr.i;
        if (tr.is_line) {
            h.phi_off = P.SNl; h.line_idx = P.Nlines;
            P.SNl += tr.Nlambda; P.Nlines++;
            h.cB = (0.25 * kHC / M_PI) * tr.Bij; // rh_method.py:268,279
            h.gij = tr.Bji / tr.Bij;             // :450
            h.AB = tr.Aji / tr.Bji;              // :281
            h.lambda0 = tr.lambda0;
            for (int lt = 0; lt < tr.Nlambda; ++lt) { wl.push_back(wlambda(wave, tr, lt) / kHC); alpha.push_back(0.0); } // :451
        } else {
            h.cont_off = P.SNc;
            P.SNc += tr.Nlambda;
            for (int lt = 0; lt < tr.Nlambda; ++lt) {
                wl.push_back(wlambda(wave, tr, lt) / wave[tr.Nblue + lt] / kHPlanck); // :455
                alpha.push_back(d->alpha[alpha_in + lt]);
            }
            alpha_in += tr.Nlambda;
        }
        P.htrans.push_back(h);
    }
    // ---- tile schedule: L = 64/Nrays consecutive wavelengths per wavefront pair
    // ---- tile schedule: L = 64/Nrays consecutive wavelengths per wavefront pair
    const int P_line = P.phi_compact ? 1 : 2 * P.Nrays;
    size_t phi_run = 0, corr_run = 0, pp_run = 0;   // running block offsets inside a column's phi_T / corr_T / Psi3_T
    // The transitions of the wavelengths [a, b) by role.  Lines are per-ray slots of the sweep.  A continuum is
    //   fast     its atom has no line in the tile: ray independent, handled by k_fast_prepass / k_fast_gamma;
    //   linked   its atom has lines in the tile but none of them touches the continuum's UPPER level: its Gamma integrand
    //            is then affine in I, Psi* and Psi* phi_line with ray-independent coefficients, so it stays out of the
    //            sweep too -- the sweep stores sum_mu w Psi* phi per line, the pre-pass hands the line three
    //            ray-independent sums (atom.eta, atom.chi[i], atom.chi[j] of the continua, rh_method.py:616-627);
    //   per-ray  otherwise (a line ends on the continuum's upper level): goes through the sweep like a line.
    // If one continuum of an atom has to be per-ray, all continua of that atom in the tile are (their sums couple).
    struct Roles { std::vector<int> lines, per_ray_conts, fast; int nlinked = 0; };
    const bool no_linked = opt.no_linked;      // diagnostic: the round-1 classification
    auto roles_of = [&](int a, int b) {
        Roles r;
        std::vector<int> conts;
        unsigned atoms_with_line = 0;
        for (int t = 0; t < P.Ntrans; ++t) {
            bool any = false;
            for (int la = a; la < b && !any; ++la) any = active[(size_t)t * Nspect + la];
            if (!any) continue;
            if (P.htrans[t].is_line) { r.lines.push_back(t); atoms_with_line |= 1u << P.htrans[t].atom; }
            else conts.push_back(t);
        }
        unsigned atoms_per_ray = 0;
        for (int t : conts) {
            const DevTrans& h = P.htrans[t];
            if (!((atoms_with_line >> h.atom) & 1u)) continue;
            bool touches = no_linked || (int)r.lines.size() > LSX_MAX_TILE_LINES;   // (the fast kernels couple at most that many lines)
            for (int l : r.lines) touches = touches || P.htrans[l].li == h.lj || P.htrans[l].lj == h.lj;
            if (touches) atoms_per_ray |= 1u << h.atom;
        }
        for (int t : conts) {
            const DevTrans& h = P.htrans[t];
            if ((atoms_per_ray >> h.atom) & 1u) r.per_ray_conts.push_back(t);
            else { r.fast.push_back(t); r.nlinked += (atoms_with_line >> h.atom) & 1u; }
        }
        return r;
    };
    // ---- where to cut: a wavefront costs the same for 1 or L wavelengths, and roughly C(nP) per depth
    // step with nP = per-ray slots of the tile (measured shader cycles, profiles/).  Dynamic programme over
    // the cut positions; ties favour fewer tiles.  Any tiling gives the same results.
    std::vector<int> cuts;
    {
        auto cost = [&](int a, int b) {
            // SIMD time of one wavefront (wave cycles / resident waves per SIMD), PMC-measured on MI355X for the
            // classes with 0 .. 4 compile-time slots (profiles/); more slots run the generic instance.  Linked continua
            // cost the sweep three stream loads and one more angle sum per line.
            static const double C[] = {2500.0, 2950.0, 3550.0, 6600.0, 9600.0};
            const Roles r = roles_of(a, b);
            const int np = (int)(r.lines.size() + r.per_ray_conts.size());
            return (np <= 4 ? C[np] : 14600.0 + 2500.0 * (np - 5)) + (r.nlinked ? 250.0 * r.lines.size() : 0.0);
        };
        const bool natural = opt.natural_tiles;
        std::vector<double> best(Nspect + 1, 1e300);
        std::vector<int> from(Nspect + 1, 0);
        best[0] = 0.0;
        for (int i = 1; i <= Nspect; ++i)
            for (int w = 1; w <= P.L && w <= i; ++w) {
                if (natural && w != P.L && i != Nspect) continue;
                if (natural && ((i - w) % P.L) != 0) continue;
                const double v = best[i - w] + cost(i - w, i) + 1.0;
                if (v < best[i]) { best[i] = v; from[i] = i - w; }
            }
        for (int i = Nspect; i > 0; i = from[i]) cuts.push_back(from[i]);
        std::reverse(cuts.begin(), cuts.end());
        cuts.push_back(Nspect);
    }
    std::vector<int> cont_index(P.Ntrans, -1);
    P.Ncont = 0;
    for (int t = 0; t < P.Ntrans; ++t)
        if (!P.htrans[t].is_line) cont_index[t] = P.Ncont++;
    P.trans_row.assign(P.Ntrans, 0);
    for (int t = 0; t < P.Ntrans; ++t) P.trans_row[t] = P.htrans[t].is_line ? P.htrans[t].line_idx : cont_index[t];
    for (size_t ic = 0; ic + 1 < cuts.size(); ++ic) {
        const int la0 = cuts[ic];
        DevTile tl{};
        tl.la0 = la0;
        tl.nla = cuts[ic + 1] - la0;
        tl.slot0 = (int)P.tile_slots.size();
        const Roles roles = roles_of(la0, la0 + tl.nla);
        const std::vector<int>& lines = roles.lines;
        // per-ray slots: lines, then the continua that must go through the sweep; fast: the other continua (linked ones flagged)
        std::vector<int> per_ray = lines, fast = roles.fast;
        per_ray.insert(per_ray.end(), roles.per_ray_conts.begin(), roles.per_ray_conts.end());
        unsigned atoms_with_line = 0;
        for (int t : lines) atoms_with_line |= 1u << P.htrans[t].atom;
        if ((int)per_ray.size() > LSX_MAX_PER_RAY || (int)fast.size() > LSX_MAX_FAST) {
            return perr(err, LSX_EUNSUPPORTED, "lsx_create: more than %d overlapping transitions in wavelengths [%d, %d)",
                        LSX_MAX_PER_RAY, la0, la0 + tl.nla);
        }
        tl.nP = (int)per_ray.size();
        tl.nF = (int)fast.size();
        tl.nK = roles.nlinked;
        tl.nL = (int)lines.size();
        if (tl.nK > 0) P.nL_linked_max = std::max(P.nL_linked_max, tl.nL);
        if (tl.nK > 0) {            // the line slots' correction streams and the sweep's sum_mu w Psi* phi streams
            tl.corr_off = (int)corr_run;
            corr_run += (size_t)tl.nL * 3 * Ns * P.L;
            tl.pp_off = (int)pp_run;
            pp_run += (size_t)tl.nL * Ns * P.L;
        }
        P.any_cont = P.any_cont || !fast.empty() || per_ray.size() > lines.size();
        std::vector<int> order = per_ray;
        order.insert(order.end(), fast.begin(), fast.end());
        // tile-local cell ids
        std::vector<int> lev_ids, atom_ids;
        auto local = [](std::vector<int>& v, int x) {
            auto it = std::find(v.begin(), v.end(), x);
            if (it != v.end()) return (int)(it - v.begin());
            v.push_back(x);
            return (int)v.size() - 1;
        };
        // level / atom cells of the generic instance: shared among the PER-RAY slots only (fast and linked continua never
        // enter the sweep's bookkeeping); first-writer flags follow the execution order of pass 1
        std::vector<int> exec = per_ray;
        std::vector<int> chi_written, u_written, eta_written;
        auto seen = [](std::vector<int>& v, int x) { bool sn = std::find(v.begin(), v.end(), x) != v.end(); if (!sn) v.push_back(x); return sn; };
        std::vector<int> first_flags(P.Ntrans, 0);
        auto share_flags = [&](int t) {
            const DevTrans& h = P.htrans[t];
            int nli = 0, nlj = 0, natom = 0, uiread = 0;
            for (int v : per_ray) {
                const DevTrans& o = P.htrans[v];
                if (o.li == h.li || o.lj == h.li) nli++;
                if (o.li == h.lj || o.lj == h.lj) nlj++;
                if (o.atom == h.atom) natom++;
                if (o.lj == h.li) uiread = 1;
            }
            int f = h.is_line ? SLOT_LINE : 0;
            if (nli > 1) f |= SLOT_LI_CELL;
            if (nlj > 1) f |= SLOT_LJ_CELL;
            if (uiread) f |= SLOT_UI_READ;
            if (natom > 1) f |= SLOT_ETA_CELL;
            return f;
        };
        for (int t : exec) {
            const DevTrans& h = P.htrans[t];
            int f = share_flags(t);
            if ((f & SLOT_LI_CELL) && !seen(chi_written, h.li)) f |= SLOT_CHI_I_FIRST;
            if (f & SLOT_LJ_CELL) {
                if (!seen(chi_written, h.lj)) f |= SLOT_CHI_J_FIRST;
                if (!seen(u_written, h.lj)) f |= SLOT_U_J_FIRST;
            }
            if ((f & SLOT_ETA_CELL) && !seen(eta_written, h.atom)) f |= SLOT_ETA_FIRST;
            first_flags[t] = f;
        }
        for (int t : order) {
            const DevTrans& h = P.htrans[t];
            DevSlot sl{};
            sl.flags = first_flags[t];
            sl.li = h.li; sl.lj = h.lj; sl.atom = h.atom;
            // cells exist only for levels / atoms that two transitions of the tile share
            sl.ci = (sl.flags & (SLOT_LI_CELL | SLOT_UI_READ)) ? local(lev_ids, h.li) : 0;
            sl.cj = (sl.flags & SLOT_LJ_CELL) ? local(lev_ids, h.lj) : 0;
            sl.ca = (sl.flags & SLOT_ETA_CELL) ? local(atom_ids, h.atom) : 0;
            sl.Nblue = h.Nblue; sl.Nlam = h.Nlam; sl.wl_off = h.wl_off; sl.trans = t;
            // the block of this (tile, transition): the tile's wavelengths inside the transition's range
            sl.first = std::max(tl.la0, h.Nblue);
            sl.len = std::min(tl.la0 + tl.nla, h.Nblue + h.Nlam) - sl.first;
            if (h.is_line) {
                sl.base = (int)phi_run;
                phi_run += (size_t)sl.len * P_line * Ns;
                sl.wphi_off = h.line_idx * Ns;
                sl.cB = h.cB; sl.g = h.gij; sl.Vc = h.gij * h.cB; sl.Uc = h.AB * (h.gij * h.cB);
            } else {
                sl.base = cont_index[t] * Ns;                // row of the nStar-ratio table: g_ij = nsr * E_T
                if (std::find(fast.begin(), fast.end(), t) != fast.end()) {
                    sl.flags |= SLOT_FAST;
                    if ((atoms_with_line >> h.atom) & 1u) {
                        sl.flags |= SLOT_LINKED;
                        for (size_t u = 0; u < lines.size() && u < 4; ++u) {
                            const DevTrans& x = P.htrans[lines[u]];
                            if (x.atom == h.atom) sl.lkbits |= 1u << (8 * u);
                            if (x.li == h.li) sl.lkbits |= 2u << (8 * u);
                            if (x.lj == h.li) sl.lkbits |= 4u << (8 * u);
                        }
                    }
                }
            }
            if (tl.nP >= 2 && tl.nP <= 4 && !(sl.flags & SLOT_FAST)) {
                int o = 0;
                for (int tv : per_ray) {
                    if (tv == t) continue;
                    const DevTrans& x = P.htrans[tv];
                    sl.rel[o][REL_CI] = (double)(x.li == h.li) - (double)(x.lj == h.li);
                    sl.rel[o][REL_CJ] = (double)(x.li == h.lj) - (double)(x.lj == h.lj);
                    sl.rel[o][REL_UJ] = (double)(x.lj == h.lj);
                    sl.rel[o][REL_UI] = (double)(x.lj == h.li);
                    sl.rel[o][REL_EA] = (double)(x.atom == h.atom);
                    for (int q = 0; q < 5; ++q)
                        if (sl.rel[o][q] != 0.0) sl.relmask |= 1u << o;
                    ++o;
                }
            }
            if (tl.nP == 1 && !(sl.flags & SLOT_FAST) && (sl.flags & (SLOT_LI_CELL | SLOT_LJ_CELL | SLOT_UI_READ | SLOT_ETA_CELL))) {
                return perr(err, LSX_EUNSUPPORTED, "lsx_create: internal: single per-ray slot with shared levels");
            }
            P.slots.push_back(sl);
            P.tile_slots.push_back(t);
            P.tile_slot_fast.push_back((sl.flags & SLOT_FAST) ? 1 : 0);
        }
        // correction slots of a tile with linked continua: one per line, carrying the line's transition so that the Gamma
        // epilogue adds their slabs to the line's rates; not a transition of their own (no wavelengths, no profile block)
        tl.nX = tl.nK > 0 ? tl.nL : 0;
        for (int x = 0; x < tl.nX; ++x) {
            const DevTrans& h = P.htrans[lines[x]];
            DevSlot sl{};
            sl.flags = SLOT_FAST;
            sl.li = h.li; sl.lj = h.lj; sl.atom = h.atom; sl.trans = lines[x];
            P.slots.push_back(sl);
            P.tile_slots.push_back(lines[x]);
            P.tile_slot_fast.push_back(1);
        }
        if (tl.nF > 0) {
            P.fast_tiles.push_back((int)P.tiles.size());
            P.nF_max = std::max(P.nF_max, tl.nF);
            // simple: per atom one common upper level, distinct lower levels, no lower level equal to that upper level
            bool simple = true;
            for (size_t a = 0; a < fast.size() && simple; ++a)
                for (size_t b = 0; b < fast.size() && simple; ++b) {
                    const DevTrans &x = P.htrans[fast[a]], &y = P.htrans[fast[b]];
                    if (x.atom != y.atom) continue;
                    if (x.lj != y.lj || x.li == y.lj || (a != b && x.li == y.li)) simple = false;
                }
            // 2: additionally at most LSX_FAST_NQ continua per atom -> k_fast_gamma_cols (LSX_FAST_ROWS: diagnostic, the
            // row-mapped kernel for every tile)
            int group = 0, group_max = 0;
            for (size_t a = 0; a < fast.size(); ++a) {
                group = (a > 0 && P.htrans[fast[a]].atom == P.htrans[fast[a - 1]].atom) ? group + 1 : 1;
                group_max = std::max(group_max, group);
            }
            const int lkn = tl.nK > 0 ? (int)lines.size() : 0;       // lines the linked continua feed
            const size_t lds_cols = fgc_lds_bytes(P.L, LSX_FGC_MAXF, 4, lkn);
            const bool cols = simple && group_max <= LSX_FAST_NQ && (int)fast.size() <= LSX_FGC_MAXF && P.L % 2 == 0 && lkn <= 2 && lds_cols <= 64 * 1024 && !opt.fast_rows;
            // 3: a simple set with MORE continua per atom (or per tile) than that -- carbon's and iron's fourteen, MgII's ten bound-free
            // continua onto one level -- takes the big-set instances of the same kernel (lsx_fast.h: the atom's sums first, then its
            // continua in chunks of LSX_FAST_NQ; twelve wavelengths per tile only)
            const bool big = simple && !cols && P.L == 12 && (int)fast.size() <= LSX_FGC_MAXF_BIG && lkn <= 2 && !opt.fast_rows &&
                             fgc_lds_bytes(P.L, LSX_FGC_MAXF_BIG, 2, lkn) <= 64 * 1024;
            tl.fast_simple = simple ? (cols ? 2 : (big ? 3 : 1)) : 0;
            if (!simple) P.fast_generic = true;
            (tl.fast_simple >= 2 ? P.fast_cols[fgc_list(tl)] : P.fast_rest).push_back((int)P.tiles.size());
        }
        // a compile-time slot count needs the per-depth operand table in LDS; very deep columns fall back to the
        // generic instance (runtime slot loops, operands through the scalar cache)
        const bool table_fits = (size_t)(Ns + 1) * (3 * tl.nP + 2) * sizeof(double) <= 32 * 1024;
        int npt = (tl.nP <= 4 && table_fits) ? tl.nP : -1;
        int nl = npt >= 0 ? (int)lines.size() : 0;
        const bool lk = tl.nK > 0;
        // two lines: is their relation one of the two common cases the sweep has a leaner instance for?
        int topo = 0;
        if (npt == 2 && nl == 2 && !opt.no_topo) {
            const DevTrans &x = P.htrans[per_ray[0]], &y = P.htrans[per_ray[1]];
            const bool share_any = x.li == y.li || x.li == y.lj || x.lj == y.li || x.lj == y.lj;
            if (x.atom == y.atom && x.li == y.li && x.lj != y.lj && x.lj != y.li && x.li != y.lj) topo = 1;
            else if (x.atom != y.atom && !share_any) topo = 2;
        }
        // a shape without a compiled instance (e.g. four per-ray slots with linked continua) runs the generic one, which
        // reads slot counts, cells and linked streams at run time -- never an instance of another shape
        if (npt >= 0 && !lsx_sweep_instance_exists(npt, nl, lk, topo)) {
            if (topo != 0 && lsx_sweep_instance_exists(npt, nl, lk, 0)) topo = 0;
            else { npt = -1; nl = 0; topo = 0; }
        }
        if (npt >= 0) P.static_max = std::max(P.static_max, npt);
        PlanClass* k = nullptr;
        for (auto& q : P.plan_classes)
            if (q.npt == npt && q.nl == nl && q.linked == lk && q.topo == topo && (opt.class_chunk <= 0 || (int)q.tiles.size() < opt.class_chunk)) k = &q;
        if (!k) { P.plan_classes.push_back(PlanClass()); k = &P.plan_classes.back(); k->npt = npt; k->nl = nl; k->linked = lk; k->topo = topo; }
        k->tiles.push_back((int)P.tiles.size());
        if (tl.nF > 0) {
            k->has_fast = true;
            k->fast_tiles.push_back((int)P.tiles.size());
            (tl.fast_simple >= 2 ? k->fast_cols[fgc_list(tl)] : k->fast_rest).push_back((int)P.tiles.size());
        }
        k->ncell_lev = std::max(k->ncell_lev, (int)lev_ids.size());
        k->ncell_atom = std::max(k->ncell_atom, (int)atom_ids.size());
        P.tiles.push_back(tl);
    }
    // Launch order and stream priority: by the class's estimated share of the call (tiles x measured cost of a tile, a class with a
    // pre-pass -> sweep -> epilogue chain counting half again), largest first -- the chain of the largest class is the call's
    // critical path, the small classes fill in behind it.  (LSX_ORDER=cost: round 1's order, by cost of one workgroup.)
    auto class_work = [](const PlanClass& k) {
        static const double C[] = {2500.0, 2950.0, 3550.0, 6600.0, 9600.0};
        const double per_tile = k.npt < 0 ? 14600.0 : C[std::min(k.npt, 4)] * (k.linked ? 1.25 : 1.0);
        return per_tile * (double)k.tiles.size() * (k.fast_tiles.empty() ? 1.0 : 1.5);
    };
    {
        if (opt.order_by_cost) {
            auto wg_cost = [](const PlanClass& k) { return k.npt < 0 ? 100 : k.npt; };
            size_t chain_tiles = 0;
            for (auto& k : P.plan_classes)
                if (!k.fast_tiles.empty()) chain_tiles += k.tiles.size();
            const bool chains_first = 2 * chain_tiles > P.tiles.size();
            std::stable_sort(P.plan_classes.begin(), P.plan_classes.end(), [&](const PlanClass& a, const PlanClass& b) {
                if (chains_first && a.fast_tiles.empty() != b.fast_tiles.empty()) return !a.fast_tiles.empty();
                return wg_cost(a) > wg_cost(b);
            });
        } else {
            std::stable_sort(P.plan_classes.begin(), P.plan_classes.end(),
                             [&](const PlanClass& a, const PlanClass& b) { return class_work(a) > class_work(b); });
        }
    }
    for (auto& k : P.plan_classes) {
        k.work = class_work(k);
        // level / atom cells exist only in the generic instance; compile-time classes keep that bookkeeping in registers
        const int cl = k.npt >= 0 ? 0 : k.ncell_lev, ca = k.npt >= 0 ? 0 : k.ncell_atom;
        k.lds_bytes = (size_t)lsx_sweep_lds(k.npt, k.linked, Ns, cl, ca).total * sizeof(double);
        if (k.lds_bytes > 64 * 1024) { return perr(err, LSX_EUNSUPPORTED, "lsx_create: a tile needs %zu B of LDS", k.lds_bytes); }
        // diagnostic (profiles/occupancy.sh): at most LSX_OCC_WG workgroups per CU, enforced through the LDS request -- how a
        // class's time depends on the waves resident per SIMD (workgroups / 2)
        if (opt.occ_wg >= 3) k.lds_bytes = std::max(k.lds_bytes, (size_t)(160 * 1024 / opt.occ_wg) & ~(size_t)15);
        P.lds_bytes = std::max(P.lds_bytes, k.lds_bytes);
    }

    // ---- column-independent tables as the kernels read them
    P.zmu.assign(P.Nrays, 1.0); P.wmuh.assign(P.Nrays, 0.0); P.u_la.resize(Nspect);
    for (int m = 0; m < P.Nrays; ++m) { P.zmu[m] = 1.0 / d->muz[m]; P.wmuh[m] = 0.5 * d->wmu[m]; }
    for (int la = 0; la < Nspect; ++la) P.u_la[la] = 2.0 * kHC / std::pow(kNM_TO_M * wave[la], 3.0); // :286
    for (int t = 0; t < P.Ntrans; ++t)
        if (!P.htrans[t].is_line) { P.cont_li.push_back(P.htrans[t].li); P.cont_lj.push_back(P.htrans[t].lj); }

    // ---- per-column strides (doubles); all in-kernel addressing is a scalar base + a 32-bit byte offset per lane
    P.phi_in_col = (size_t)P.SNl * (P.phi_compact ? 1 : 2 * (size_t)P.Nrays) * Ns; // as handed over (rh_method.py:224)
    // as stored: the (tile, line) blocks, then two doubles that stay zero -- where the lanes of a tile whose wavelength lies
    // outside a line's range point their profile loads (no select on the loaded value)
    P.phi_col = phi_run + (phi_run ? 2 : 0);
    P.corr_col = corr_run;
    P.pp_col = pp_run;
    P.til_col = P.tiles.size() * (size_t)P.L * Ns;        // one tile-major [tile][k][j] array
    P.sca_col = P.sca_per_lambda ? P.til_col : (size_t)Ns;
    if (P.phi_col > 0x0fffffff || P.corr_col > 0x0fffffff || P.til_col > 0x0fffffff) return perr(err, LSX_EUNSUPPORTED, "column too large for 32-bit byte offsets");

    // ---- ray-serial sweep: five columns per wavefront, addressed as one base + 32-bit byte offsets
    {
        // (LSX_BG_PAIRS: the background pairs are two tile-major arrays in one: 16 bytes per element)
        const size_t big = std::max(std::max((LSX_BG_PAIRS ? 2 : 1) * P.til_col, P.phi_col), std::max(std::max(P.corr_col, P.pp_col), (size_t)std::max(P.NLtot, std::max(P.Nlines, P.Ncont)) * Ns));
        // (round 5: the per-depth operands come through a ring in LDS, lsx_plan.h -- no limit on the depth count any more; the table
        // that feeds it is addressed with 32-bit byte offsets inside a column group)
        P.rs_ok = !opt.no_rs && P.Nrays == LSX_RS_RAYS && !P.sca_per_lambda && (LSX_RS_COLS + 1) * big * 8 < 0xffffffffull &&
                  lsx_optab_group_doubles(P.Ntrans, Ns, P.Ncont) * 8 < 0xffffffffull;
        P.rs_min_columns = opt.rs_min_columns;
        P.phi_group = (P.rs_ok && !opt.no_phi_group) ? LSX_RS_COLS : 1;
        for (auto& k : P.plan_classes) {
            k.rs = P.rs_ok && k.npt >= 0 && k.npt <= opt.rs_max_npt && lsx_rs_instance_exists(k.npt, k.nl, k.linked, k.topo);
            // The ray-serial instances for line-only tiles with linked continua (one line; two lines with a known relation) do not
            // read the continua's corrections: the column-mapped fast-continuum epilogue applies them to the lines' rates
            // (lsx_fast.h).  A class with a tile that epilogue cannot take keeps one ray per lane.
            k.lk_epi = k.linked && k.npt >= 1 && k.nl == k.npt && (k.npt == 1 || k.topo != 0);
            if (k.rs && k.lk_epi)
                for (int t : k.tiles)
                    if (P.tiles[t].fast_simple < 2) k.rs = false;
            // folded fast continua (lsx_plan.h): every tile's row fits two elements per lane, and the class needs no correction
            // streams from the pre-pass (the unfactored linked instance reads them)
            // (round 5 left the two-line instances with a known relation and no linked continua out: at the register limit, folded they
            // spilled.  With the Boltzmann factor formed in the lane they fit: lsx_plan.h, LSX_FOLD_TWOLINE)
            k.fold = k.rs && !opt.no_fold && k.has_fast && (!k.linked || k.lk_epi) && lsx_rs_fold_instance_exists(k.npt, k.linked, k.topo);
            k.fold_nF = 0;
            for (int t : k.tiles) {
                k.fold_nF = std::max(k.fold_nF, (int)P.tiles[t].nF);
                if (lsx_rs_row_doubles(k.npt, P.tiles[t].nF) > LSX_RS_FOLD_ROW_MAX || P.tiles[t].nF > 12) k.fold = false;
            }
            if (k.fold && (size_t)lsx_rs_lds_doubles(k.npt, Ns, false, k.fold_nF) * sizeof(double) > 64 * 1024) k.fold = false;
            // ... and the fast continua's Gamma integrands (EPI, lsx_plan.h): every tile of the class is one the column-mapped epilogue
            // takes (simple bound-free sets, at most two linked lines), its class needs no correction streams (checked above), and the
            // workgroup's LDS leaves room for four per CU
            k.epi = k.fold && !opt.no_epi;
            for (int t : k.tiles)
                if (P.tiles[t].nF > 0 && P.tiles[t].fast_simple != 2) k.epi = false;
            if (k.epi && (size_t)lsx_rs_lds_doubles(k.npt, Ns, false, k.fold_nF, true, k.linked ? k.nl : 0) * sizeof(double) > 40 * 1024) k.epi = false;
            k.rsp = k.rs && lsx_rsp_instance_exists(k.npt, k.nl, k.linked, k.topo) &&
                    (size_t)lsx_rs_lds_doubles(k.npt, Ns, true) * sizeof(double) <= 64 * 1024;
        }
    }

    // ---- launch shapes of the kernels around the sweep: decided (and refused) here, not inside a half-enqueued call
    LaunchShapes& S = P.shapes;
    // depths whose operands are staged in LDS at a time: the whole column if `budget` bytes allow, else a multiple of `rows`
    auto seg_for = [&](int rows, size_t doubles_per_depth, size_t fixed_doubles, size_t budget) {
        const size_t room = budget / 8 > fixed_doubles ? budget / 8 - fixed_doubles : 0;
        const long fit = (long)(room / std::max<size_t>(1, doubles_per_depth));
        if (fit >= Ns) return Ns;
        return (int)std::max<long>(rows, fit / rows * rows);
    };
    if (!P.fast_tiles.empty()) {
        // k_fast_prepass: one block per (tile, column), rows exactly as wide as the tile
        S.prepass_seg = seg_for(256 / P.L, (size_t)3 * P.nF_max, (size_t)P.nF_max * P.L, 24 * 1024);
        S.prepass_lds = ((size_t)3 * P.nF_max * S.prepass_seg + (size_t)P.nF_max * P.L) * sizeof(double);
        if (S.prepass_lds > 64 * 1024) return perr(err, LSX_EUNSUPPORTED, "lsx_create: the fast-continuum pre-pass needs %zu B of LDS", S.prepass_lds);
        // k_fast_gamma (row mapped): LP = 16 / 32 / 64 lanes per depth row
        S.rows_lp = P.L <= 16 ? 16 : (P.L <= 32 ? 32 : 64);
        const size_t per_depth = (size_t)3 * P.nF_max + (size_t)2 * std::min(P.nL_linked_max, LSX_MAX_TILE_LINES);
        auto fixed_for = [&](int ntv) { return (size_t)(P.fast_generic ? 2 * P.NLtot + P.Natoms : 0) * ntv + (size_t)2 * P.nF_max * S.rows_lp; };
        S.rows_nt = 256;
        while (S.rows_nt > 64 && fixed_for(S.rows_nt) * 8 > 24 * 1024) S.rows_nt >>= 1;
        S.rows_seg = seg_for(S.rows_nt / S.rows_lp, per_depth, fixed_for(S.rows_nt), 40 * 1024);
        S.rows_lds = (per_depth * S.rows_seg + fixed_for(S.rows_nt)) * sizeof(double);
        if (!P.fast_rest.empty() && S.rows_lds > 64 * 1024) return perr(err, LSX_EUNSUPPORTED, "lsx_create: the fast-continuum epilogue needs %zu B of LDS", S.rows_lds);
        for (int v = 0; v < LSX_FGC_LISTS; ++v) {
            if (v == 3) continue;
            S.cols_lds[v] = v < 3 ? fgc_lds_bytes(P.L, LSX_FGC_MAXF, 4, fgc_lines(v)) : fgc_lds_bytes(P.L, LSX_FGC_MAXF_BIG, 2, fgc_lines(v));
            if (!P.fast_cols[v].empty() && S.cols_lds[v] > 64 * 1024) return perr(err, LSX_EUNSUPPORTED, "lsx_create: the fast-continuum epilogue (column mapped) needs %zu B of LDS", S.cols_lds[v]);
        }
    }
    // k_gamma_finish: one thread per (column, depth, atom) with the atom's Nlevel^2 entries in LDS
    size_t nl2max = 1;
    for (int a = 0; a < P.Natoms; ++a) nl2max = std::max(nl2max, (size_t)P.Nlevel[a] * P.Nlevel[a]);
    S.finish_nt = 128;
    int nlmax = 1;
    for (int a = 0; a < P.Natoms; ++a) nlmax = std::max(nlmax, (int)P.Nlevel[a]);
    if (!opt.finish_lds || nl2max * 128 * sizeof(double) > 48 * 1024) {
        // k_gamma_finish_levels (lsx_hip.hip): a thread per (column, depth, level) takes one COLUMN of its atom's Gamma entry by entry in
        // registers, no LDS -- the default since the end of round 5.  k_gamma_finish (a thread's whole matrix in LDS, finish_lds=1) is 1 %
        // slower on C4 (profiles/r05/ab_gamma_epilogue_thread_per_column.txt) and collapses for atoms of more than six levels: 15 levels
        // are 1.8 kB per thread -- 32 threads per workgroup, two half-filled waves per CU (all five atoms active: 1.76 against 0.66 ms)
        S.finish_w = 1;
        S.finish_lds = 0;
    } else {
    while (S.finish_nt > 32 && nl2max * S.finish_nt * sizeof(double) > 48 * 1024) S.finish_nt >>= 1;
    S.finish_lds = nl2max * S.finish_nt * sizeof(double);
    }
    if (S.finish_lds > 64 * 1024) return perr(err, LSX_EUNSUPPORTED, "lsx_create: the Gamma epilogue needs %zu B of LDS", S.finish_lds);
    for (int a = 0; a < P.Natoms; ++a) {       // k_stat_equil: atoms with more than 8 levels keep their system in LDS
        const int Nl = P.Nlevel[a];
        if (Nl > 8 && (size_t)(Nl * Nl + 2 * Nl) * 64 * sizeof(double) > 160 * 1024)
            return perr(err, LSX_EUNSUPPORTED, "lsx_create: stat_equil with Nlevel = %d needs %zu B of LDS", Nl, (size_t)(Nl * Nl + 2 * Nl) * 64 * sizeof(double));
    }
    // the fused small-batch launch (and the parabolic rule): one kernel for every tile
    for (auto& k : P.plan_classes) { S.fused_ncell_lev = std::max(S.fused_ncell_lev, k.ncell_lev); S.fused_ncell_atom = std::max(S.fused_ncell_atom, k.ncell_atom); }
    {
        int npt_max = -1;
        for (auto& k : P.plan_classes) npt_max = std::max(npt_max, k.npt);
        // every instance the fused kernel may dispatch to lays its own LDS out from its own slot count, with the cell rows of
        // the largest generic tile in front
        for (int q = -1; q <= npt_max; ++q)
            S.fused_lds = std::max(S.fused_lds, (size_t)lsx_sweep_lds(q, P.corr_col > 0, Ns, S.fused_ncell_lev, S.fused_ncell_atom).total * sizeof(double));
        if (S.fused_lds > 64 * 1024) return perr(err, LSX_EUNSUPPORTED, "lsx_create: the fused sweep launch needs %zu B of LDS", S.fused_lds);
        // the fused launch takes over the fast-continuum work of its tiles (lsx_sweep.hip: lsx_sweep_kernel_all<5, false>) where every
        // fast tile has the column-mapped epilogue with at most two linked lines, twelve wavelengths per tile, and a column's operands
        // fit the pre-pass's LDS in one piece; its two waves then need the larger of the three LDS layouts
        if (!P.fast_tiles.empty()) {
            bool ok = P.Nrays == 5 && !P.sca_per_lambda && P.L == 12 && P.fast_rest.empty() && S.prepass_seg >= Ns;
            for (int v = 3; v < LSX_FGC_LISTS; ++v) ok = ok && P.fast_cols[v].empty();       // (big sets: launches of their own at any size)
            size_t need = std::max(S.fused_lds, S.prepass_lds);
            for (int v = 0; v < 3; ++v)
                if (!P.fast_cols[v].empty())
                    need = std::max(need, fgc_lds_bytes(P.L, LSX_FGC_MAXF, 2, kLkLines[v]));
            S.fused_fast = ok && need <= 64 * 1024;
            S.fused_fast_lds = need;
        }
    }
    return LSX_OK;
}

} // namespace lsxd

// ---- options (lsx_plan.h): environment defaults, explicit list, canonical string -----------------------------------
namespace lsxd {

void options_from_env(CtxOptions* o)
{
    const char* e;
    PlanOptions& p = o->plan;
    RunOptions& r = o->run;
    p.no_linked = getenv("LSX_NO_LINKED") != nullptr;
    p.natural_tiles = (e = getenv("LSX_TILER")) && std::string(e) == "natural";
    p.no_topo = getenv("LSX_NO_TOPO") != nullptr;
    p.fast_rows = getenv("LSX_FAST_ROWS") != nullptr;
    p.finish_lds = getenv("LSX_FINISH_LDS") != nullptr;
    p.order_by_cost = (e = getenv("LSX_ORDER")) && std::string(e) == "cost";
    p.occ_wg = (e = getenv("LSX_OCC_WG")) ? atoi(e) : 0;
    p.no_rs = getenv("LSX_NO_RS") != nullptr;
    p.no_fold = getenv("LSX_NO_FOLD") != nullptr;
    p.no_epi = getenv("LSX_EPI") == nullptr;
    p.no_phi_group = (e = getenv("LSX_PHI_GROUP")) && atoi(e) == 1;
    if ((e = getenv("LSX_RS_MIN_COLUMNS"))) p.rs_min_columns = atoi(e);
    if ((e = getenv("LSX_RS_MAX_NPT"))) p.rs_max_npt = atoi(e);
    if ((e = getenv("LSX_CLASS_CHUNK"))) p.class_chunk = atoi(e);
    r.se_lds = getenv("LSX_SE_LDS") != nullptr;
    r.trace_classes = getenv("LSX_TRACE_CLASSES") != nullptr;
    r.serial = getenv("LSX_SERIAL") != nullptr;
    r.finish_big = getenv("LSX_FINISH_BIG") != nullptr;
    r.abl_fast = (e = getenv("LSX_ABL_FUSED_FAST")) ? atoi(e) & 6 : 0;
    r.fused_epilogue = getenv("LSX_FUSED_EPILOGUE") != nullptr;
    r.graph = getenv("LSX_GRAPH") != nullptr;
    r.no_fused_fast = getenv("LSX_NO_FUSED_FAST") != nullptr;
}

int options_apply(const char* list, CtxOptions* o, std::string* err)
{
    if (!list) return LSX_OK;
    PlanOptions& p = o->plan;
    RunOptions& r = o->run;
    std::string s(list);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t end = s.find_first_of(",;", pos);
        if (end == std::string::npos) end = s.size();
        std::string kv = s.substr(pos, end - pos);
        pos = end + 1;
        while (!kv.empty() && kv.front() == ' ') kv.erase(kv.begin());
        while (!kv.empty() && kv.back() == ' ') kv.pop_back();
        if (kv.empty()) continue;
        const size_t eq = kv.find('=');
        if (eq == std::string::npos || eq == 0 || eq + 1 >= kv.size()) { *err = "options: expected key=value, got '" + kv + "'"; return LSX_EINVAL; }
        const std::string key = kv.substr(0, eq), val = kv.substr(eq + 1);
        char* endp = nullptr;
        const long iv = strtol(val.c_str(), &endp, 10);
        const bool is_int = endp && *endp == 0;
        auto flag = [&](bool* dst, bool invert) {
            if (!is_int || (iv != 0 && iv != 1)) { *err = "options: " + key + " takes 0 or 1, got '" + val + "'"; return false; }
            *dst = invert ? iv == 0 : iv == 1;
            return true;
        };
        auto count = [&](int* dst, long lo, long hi) {
            if (!is_int || iv < lo || iv > hi) { *err = "options: " + key + " out of range: '" + val + "'"; return false; }
            *dst = (int)iv;
            return true;
        };
        bool ok = true;
        if (key == "linked") ok = flag(&p.no_linked, true);
        else if (key == "topo") ok = flag(&p.no_topo, true);
        else if (key == "fast_rows") ok = flag(&p.fast_rows, false);
        else if (key == "finish_lds") ok = flag(&p.finish_lds, false);
        else if (key == "rs") ok = flag(&p.no_rs, true);
        else if (key == "fold") ok = flag(&p.no_fold, true);
        else if (key == "epi") ok = flag(&p.no_epi, true);
        else if (key == "phi_group") ok = flag(&p.no_phi_group, true);      // 1: grouped where the ray-serial sweep can run (default), 0: per column
        else if (key == "rs_min_columns") ok = count(&p.rs_min_columns, 1, 1 << 30);
        else if (key == "rs_max_npt") ok = count(&p.rs_max_npt, 0, 2);
        else if (key == "occ_wg") ok = count(&p.occ_wg, 0, 64);
        else if (key == "class_chunk") ok = count(&p.class_chunk, 0, 4096);
        else if (key == "tiler") {
            if (val == "natural") p.natural_tiles = true; else if (val == "dp") p.natural_tiles = false;
            else { *err = "options: tiler takes dp or natural"; ok = false; }
        } else if (key == "order") {
            if (val == "cost") p.order_by_cost = true; else if (val == "plan") p.order_by_cost = false;
            else { *err = "options: order takes plan or cost"; ok = false; }
        }
        else if (key == "se_lds") ok = flag(&r.se_lds, false);
        else if (key == "serial") ok = flag(&r.serial, false);
        else if (key == "finish_big") ok = flag(&r.finish_big, false);
        else if (key == "fused_epilogue") ok = flag(&r.fused_epilogue, false);
        else if (key == "graph") ok = flag(&r.graph, false);
        else if (key == "fused_fast") ok = flag(&r.no_fused_fast, true);
        else if (key == "trace_classes") ok = flag(&r.trace_classes, false);
        else { *err = "options: unknown key '" + key + "'"; ok = false; }
        if (!ok) return LSX_EINVAL;
    }
    return LSX_OK;
}

std::string options_string(const CtxOptions& o)
{
    const PlanOptions& p = o.plan;
    const RunOptions& r = o.run;
    char b[512];
    snprintf(b, sizeof b,
             "linked=%d;tiler=%s;topo=%d;fast_rows=%d;finish_lds=%d;order=%s;occ_wg=%d;class_chunk=%d;rs=%d;fold=%d;epi=%d;phi_group=%d;rs_min_columns=%d;rs_max_npt=%d;"
             "se_lds=%d;serial=%d;finish_big=%d;fused_epilogue=%d;graph=%d;fused_fast=%d;abl_fused_fast=%d",
             !p.no_linked, p.natural_tiles ? "natural" : "dp", !p.no_topo, (int)p.fast_rows, (int)p.finish_lds, p.order_by_cost ? "cost" : "plan", p.occ_wg, p.class_chunk,
             !p.no_rs, !p.no_fold, !p.no_epi, !p.no_phi_group, p.rs_min_columns, p.rs_max_npt, (int)r.se_lds, (int)r.serial, (int)r.finish_big,
             (int)r.fused_epilogue, (int)r.graph, !r.no_fused_fast, r.abl_fast);
    return b;
}

std::string plan_class_string(const LsxPlan& P)
{
    std::string s;
    char b[64];
    for (const PlanClass& k : P.plan_classes) {
        snprintf(b, sizeof b, "%s%d.%d.%d.%d:%zu%s", s.empty() ? "" : ",", k.npt, k.nl, (int)k.linked, k.topo, k.tiles.size(), k.rs ? (k.fold ? (k.epi ? "sfe" : "sf") : "s") : "");
        s += b;
    }
    return s;
}

uint64_t fnv1a64(const std::string& s)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ull; }
    return h;
}

} // namespace lsxd
