// lsx_sweep_rs.hip -- the "ray-serial" sweep: the same fused eta/chi/U-V build + piecewise-linear short-characteristics
// sweep + Psi*/Gamma/J accumulation as lsx_sweep.hip (rh_method.py:595-692, formal_solver.py:14-212), for the tile classes
// with at most two per-ray slots and five rays, mapped onto the wavefront the other way round:
//
//   lsx_sweep.hip     lane = one RAY (wavelength, mu);       wavefront = one direction of (tile, column)
//   this file         lane = one WAVELENGTH of one column;   wavefront = one direction of (tile, FIVE columns);
//                     the lane walks its five rays one after the other inside every depth step (registers).
//
// Why (measured, profiles/r03_bound_evidence.md): with one ray per lane the sweep runs at the vector-issue rate of the clock
// the chip holds under it AND on its own serial chain -- removing vector instructions did not make it faster while each
// wave still had one dependency chain and three LDS round trips per depth step.  Here
//   * everything that does not depend on the angle -- background, sigma J-dagger, the per-depth level populations, stream
//     addressing, the J / Psi-bar / Psi* phi sums, the J store and dJ -- is done once per wavelength instead of once per ray;
//   * the angle quadrature is a sum in registers: the LDS exchanges of a depth step are gone, and so is the operand table
//     (per-depth operands of the lane's own column arrive as ordinary, 12-lane-uniform loads one depth ahead);
//   * the wavelength quadrature of Gamma is reduced across lanes once per five rays;
//   * a lane carries five independent recurrences: the latency of one ray's reciprocal / exponential / fma chain is filled
//     by the other four.
// Results: the same terms as lsx_sweep.hip; the sums over mu and over wavelength are associated differently (last-bit
// differences, inside the stated tolerances; a column's result still does not depend on its position in the batch: the
// five columns of a wavefront share nothing but the instruction stream).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "lsx_dev.h"
#include "lsx_plan.h"

namespace {

constexpr double kCLight = 2.99792458E+08;
constexpr double kHPlanck = 6.6260755E-34;
constexpr double kKBoltzmann = 1.380658E-23;
constexpr double kNM_TO_M = 1.0E-09;
constexpr double kHC = kHPlanck * kCLight;
constexpr double kPi = 3.14159265358979323846;

#define LSX_CONST(T, ptr) ((const __attribute__((address_space(4))) T*)(ptr))

__device__ __forceinline__ double planck(double temp, double wav)     // utils.py:17-22
{
    const double hc_Tkla = kHC / (kKBoltzmann * kNM_TO_M * wav) / temp;
    const double x = kNM_TO_M * wav;
    const double twohnu3_c2 = (2.0 * kHC) / (x * x * x);
    return twohnu3_c2 / (exp(hc_Tkla) - 1.0);
}
__device__ __forceinline__ const double& at(const double* base, unsigned byte_off)
{
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ double& at(double* base, unsigned byte_off)
{
    return *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + byte_off);
}
// cache-policy hints (round 6; profiles/r06_bound_evidence.md 1).  The line profiles -- read exactly once per call, the largest stream --
// are requested non-temporally (global_load ... nt): FETCH_SIZE does not move (the partner wave's re-reads are 100 us away, far beyond
// any cache), but the profile bytes no longer wash through the CU's vector cache, where the 96-byte rows of the other streams share
// their 128-byte lines from one depth to the next: C3 -4 % (three boxes: -3, -6, -4), C4 -0.7 %.  LSX_NT_PHI=0: plain loads.
// Measured and NOT kept: the ray-independent streams non-temporally too (LSX_NT_BG: C3 +8 %, C4 +3.5 %: their lines ARE shared between
// consecutive depths), non-temporal stores of the sums and slabs (LSX_NT_ST: C4 +8 %: partial lines past the L2's write combining)
#ifndef LSX_NT_PHI
#define LSX_NT_PHI 1
#endif
__device__ __forceinline__ double ld_once(const double* base, unsigned byte_off)
{
#if LSX_NT_PHI
    return __builtin_nontemporal_load(reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off));
#else
    return at(base, byte_off);
#endif
}
__device__ __forceinline__ void st_once(double* base, unsigned byte_off, double v)
{
#ifdef LSX_NT_ST
    __builtin_nontemporal_store(v, reinterpret_cast<double*>(reinterpret_cast<char*>(base) + byte_off));
#else
    at(base, byte_off) = v;
#endif
}
__device__ __forceinline__ double nanmax(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : fmax(a, b); }

constexpr int NR = LSX_RS_RAYS;        // rays per wavelength (compile time: the ray loop is unrolled)
constexpr int NC = LSX_RS_COLS;        // columns per wavefront
constexpr int LW = LSX_WAVE / NR;      // wavelengths per tile (the context's tile width for NR rays)
constexpr int RROW = 64;               // doubles per row of the reduction buffer

} // namespace

// NPT per-ray slots (0 .. 2), NL of them lines (they come first), LK: the tile has linked continua, TOPO: relation of the two
// slots of a two-line tile (lsx_sweep.hip).  PAR: the monotonic piecewise-parabolic rule (N4, include/lsx.h) instead of the
// reference's piecewise-linear one -- the same mapping, the formal solution one depth behind the opacities (see `pstep`).
// The parabolic instances run ONE wave per SIMD: a lane carries five more doubles of recurrence state per wavelength than under the
// linear rule and the compiler's schedule of the five interleaved rays needs 310-320 registers (256 + 55..59 accumulation registers
// as spill space, no scratch).  At two waves per SIMD (256 registers) the same source spills 50-60 registers to scratch; the reloads
// sit between the stream loads and the waits, so every wait for a reload also waits for the next depth's operands: measured
// slower than one wave per SIMD whenever the spills land inside the step (C3 / C4 formal solution 3.30 / 14.4 ms against
// 2.12 / 7.92), equal at best (profiles/r04/n4_ray_serial_waves_per_simd.txt).
#ifndef LSX_RSP_WPE
#define LSX_RSP_WPE 1
#endif
template <int NPT, int NL, bool LK, int TOPO, bool PAR, bool FOLD, bool EPI>
__global__ void __launch_bounds__(2 * LSX_WAVE) __attribute__((amdgpu_waves_per_eu(PAR ? LSX_RSP_WPE : (NPT == 0 ? 3 : LSX_RS_WPE(NPT, LK)))))
lsx_sweep_rs_kernel(const SweepParams p)
{
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
#ifdef LSX_CLOCK
    unsigned long long tk_in;                                        // (diagnostic build: the shader clock at the wave's first instruction)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk_in)::"memory");
#endif
    lds_f64* const etab = (lds_f64*)lds_raw;                         // [64][2] exp table
    constexpr int NS = NPT > 0 ? NPT : 1;
    constexpr int NV = 2 * NS;                                       // Gamma integrands per lane and depth
    constexpr bool HASC = NPT > NL;                                  // per-ray continua: they share the tile's Boltzmann factor (E_of)
    constexpr int NLK = (LK && NL > 0) ? NL : 1;
    constexpr int NCR = NPT == 1 ? 2 : 3;                            // a single slot never reads atom.chi[j_line]
    constexpr bool FACT = NPT >= 1 && NL == NPT && (NPT == 1 || TOPO != 0);    // factored Gamma integrands (step, pass C)
    // Linked continua add ray-independent shares EC, XC_i to atom.eta and atom.chi[i_line] (rh_method.py:616-627).  In the factored
    // integrands they enter as -EC sum_mu w Psi* phi and -XC_i sum_mu w Psi* phi -- products of a (depth, wavelength) quantity with the
    // sum this kernel stores for the fast-continuum epilogue anyway -- so the epilogue adds them to the line's rates (lsx_fast.h,
    // correction slots) and these instances read NO correction streams: two loads per line, wavelength and depth less, and the
    // pre-pass writes none.  (The unfactored instance <2,2,true,0> applies them ray by ray, like the one-ray-per-lane kernels.)
    constexpr bool CORR = LK && !FACT;
    // a second depth of stream prefetch (a third operand set) where the register file has room for it: at most one per-ray slot
#if defined(LSX_RS_PF2)
    constexpr bool PF2 = !PAR && NPT <= LSX_RS_PF2;
#else
    constexpr bool PF2 = false;
#endif
    const int lane = threadIdx.x & (LSX_WAVE - 1);
    const int dir = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0: down (toFrom False), 1: up (True)
    lds_f64* const red = etab + LSX_EXP_TAB + (size_t)dir * (NV + 1) * RROW;   // this wave's reduction rows (+ one for dJ)
    // per-depth operands of the group's columns: a ring of RING rows per wave (lsx_plan.h, "a RING in LDS"), row of step v =
    // [slot u][column c][3] (lines: cB (n_i - g n_j), n_j Uc, wphi; continua: n_i, n_j, nStar_i / nStar_j) then [c][2] (the half
    // length of the interval above the depth, the scattering coefficient), fed one row per step from the table k_build_optab made
    // FOLD (lsx_plan.h, "FOLDED fast continua"): the row also carries [fast continuum q][c][2] = n_i, n_j nStar_i / nStar_j, and the
    // step forms the fast continua's opacity and emissivity itself -- two elements per lane and row, the pitch from the class's
    // largest tile (p.fold_nF)
    // EPI (lsx_plan.h): the wave that visits a depth second also forms the Gamma integrands of the tile's fast continua and the linked
    // lines' corrections (what k_fast_gamma_cols did) -- see epi_fast below
    constexpr int RING = EPI ? LSX_RS_RING_EPI : LSX_RS_RING, RL0 = lsx_rs_row_doubles(NPT);
    static_assert((RING & (RING - 1)) == 0 && RING >= 4 && RING <= LSX_RS_RING && RL0 <= LSX_WAVE, "operand ring");
    static_assert(!(FOLD && PAR) && (FOLD || !EPI), "the parabolic instances keep the pre-pass; EPI instances are folded ones");
    const int RLP = FOLD ? lsx_rs_row_pitch(NPT, p.fold_nF) : RL0;                        // doubles between two rows of a ring
    const int NFP = FOLD ? lsx_rs_fold_pad(p.fold_nF) : 0;
    lds_f64* const utab = etab + LSX_EXP_TAB + 2 * (NV + 1) * RROW + 2 * LSX_WAVE;      // [2 waves][RING][RLP]
    lds_f64* const atab = utab + 2 * RING * RLP;                                        // FOLD: [q < NFP][64] the continua's cross-sections per lane
    lds_f64* const wtab = atab + NFP * LSX_WAVE;                                        // EPI: [q < NFP][64] their wavelength weights (0 in lanes without a wavelength)
    lds_f64* const fred = wtab + (EPI ? NFP * LSX_WAVE : 0);                            // EPI: [2 waves][LSX_RS_EPI_ROUND][64] reduction rows of the fast values
    lds_f64* const xw2 = fred + (EPI ? 2 * LSX_RS_EPI_ROUND * LSX_WAVE : 0);            // EPI: [2 waves][1 + lines][64] the midpoint's exchange of Psibar, Psi* phi
    lds_f64* const qinf = xw2 + (EPI ? 2 * (1 + (LK ? NL : 0)) * LSX_WAVE : 0);         // EPI: [64] per fast continuum: its linking bits, first-of-its-atom flag (as integers), then 64 spare
    lds_f64* const ring_end = qinf + (EPI ? 2 * LSX_WAVE : 0);

    // XCD-aware block -> (column group, tile): every XCD gets a contiguous range (speed only)
    int vb;
    {
        const int nb = gridDim.x, x = blockIdx.x & 7, q = blockIdx.x >> 3;
        const int nb8 = nb >> 3, rem = nb & 7;
        vb = x * nb8 + (x < rem ? x : rem) + q;
    }
    const int grp = vb / p.n_class_tiles;
    const int tile_id = LSX_CONST(int32_t, p.class_tiles)[vb - grp * p.n_class_tiles];
    const int col0 = grp * NC;
    const int ncg = min(NC, p.ncol - col0);                          // columns of this group
    const auto* tilep = LSX_CONST(DevTile, p.tiles) + tile_id;
    const int la0 = tilep->la0, nla = tilep->nla, slot0 = tilep->slot0, nF = tilep->nF;
    const auto* slots = LSX_CONST(DevSlot, p.slots) + slot0;
    const int Ns = p.Nspace, Nspect = p.Nspect, ntile = p.ntile_total;

    // lane -> (column of the group, wavelength of the tile); lanes without one shadow a real one and store nothing
    const int c_raw = lane / LW, j_raw = lane - c_raw * LW;
    const int cc = c_raw < ncg ? c_raw : ncg - 1;
    const int j = j_raw < nla ? j_raw : nla - 1;
    const int col = col0 + cc;
    const int la = la0 + j;
    const bool live_col = p.colmask ? p.colmask[col] != 0 : true;    // a frozen column keeps everything; J only changes buffers
    const bool valid = c_raw < ncg && j_raw < nla;
    const bool act = valid && live_col;

    etab[threadIdx.x] = p.exp2_tab[threadIdx.x];
    const int kS = dir ? Ns - 1 : 0;
    const int dk = dir ? -1 : 1;
    // ---- the operand ring.  Step v of this wave (depth kS + dk v) is row RING + kS + dk v of the table's blocks (RING zero rows
    // in front of depth 0 and behind depth Nspace - 1: the ring runs ahead of the sweep and past its end without clamping).
    // A row is a run of SEGMENTS of LSX_RS_SEG = 16 doubles ([column][3] + one pad): per-ray slots, geometry, folded continua -- the
    // pitch of the table's block rows too.  Lane e with 2 e < RL owns the PAIR of elements (2 e, 2 e + 1) of every row: it fetches the
    // pair of the row RING - 1 steps ahead with ONE 16-byte load and, a step later, writes it over the row that was consumed two steps
    // ago (the steps read rows v - 1 and v only); its table offset moves by one row per step.  (Round 5 fetched elements e and 64 + e
    // with two loads in the folded instances: one request per step more.)
    constexpr int SEG = LSX_RS_SEG;
    typedef double ring_pair __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) ring_pair lds_pair2;
    lds_f64* const ring = utab + (size_t)dir * RING * RLP;
    const int RL = FOLD ? lsx_rs_row_doubles(NPT, nF) : RL0;        // this tile's row
    static_assert(lsx_rs_row_doubles(2, 0) <= 2 * LSX_WAVE && LSX_RS_FOLD_ROW_MAX <= 2 * LSX_WAVE, "a lane fetches two elements of a row");
    if constexpr (FOLD) {       // the rows' pad behind the tile's own continua is read (against zero cross-sections): keep it finite
        for (int e = lane; e < RING * RLP; e += LSX_WAVE) ring[e] = 0.0;
    }
    const double* __restrict__ otab = p.optab + (size_t)grp * p.optab_group_stride;
    const int NRT = lsx_optab_rows(Ns);
    // byte offset of the lane's pair in the row of step 0 inside the group's table; bytes per step (signed: the up sweep walks backwards)
    const bool ring_lane = 2 * lane < RL;
    unsigned o_e0;
    {
        const int e = ring_lane ? 2 * lane : RL - 2;       // (a lane beyond the row repeats its last pair and writes nothing)
        const int sgm = e / SEG, w = e - sgm * SEG;
        size_t blkrow;                                     // the segment's block, in block rows
        if (sgm < NPT) blkrow = (size_t)slots[NPT > 0 ? sgm : 0].trans;
        else if (sgm == NPT) blkrow = (size_t)p.Ntrans + (size_t)dir;
        else blkrow = (size_t)p.Ntrans + 2 + (size_t)LSX_CONST(int32_t, p.trans_row)[slots[NPT + (FOLD ? sgm - NPT - 1 : 0)].trans];
        o_e0 = (unsigned)(((blkrow * NRT + (size_t)(LSX_RS_RING + kS)) * SEG + w) * 8u);      // (the table's pad: LSX_RS_RING rows, whatever this instance's ring)
    }
    const int o_s0 = dk * SEG * 8;
    auto ring_row = [&](int v) __attribute__((always_inline)) { return ring + ((v + 1) & (RING - 1)) * RLP; };
    auto ring_load = [&](unsigned off) __attribute__((always_inline)) {
        return *reinterpret_cast<const ring_pair*>(reinterpret_cast<const char*>(otab) + off);
    };
    auto ring_put = [&](int v, ring_pair x) __attribute__((always_inline)) {
        if (ring_lane) *reinterpret_cast<lds_pair2*>(ring_row(v) + 2 * lane) = x;
    };
    o_e0 -= (unsigned)o_s0;                               // row -1 first
    {   // rows -1 .. RING - 3: all requests first, then the writes (one memory round trip, not RING - 1 of them one after the other)
        ring_pair x0[RING - 1];
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) {
            x0[i] = ring_load(o_e0);
            o_e0 += (unsigned)o_s0;
        }
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) ring_put(i - 1, x0[i]);
    }
    ring_pair ring_pend0 = ring_load(o_e0);                                       // row RING - 2: written at step 0
    // one row per step: the pair fetched a step ago goes over row s - 2, the pair of row s + RING - 1 is requested
    auto ring_step = [&](const int s) __attribute__((always_inline)) {
        ring_put(s - 2, ring_pend0);
        o_e0 += (unsigned)o_s0;
        ring_pend0 = ring_load(o_e0);
    };
    if constexpr (FOLD) {
        // the fast continua's cross-sections for this lane's wavelength (0 where the continuum is not active there, and for the
        // continua the tile does not have up to the next multiple of four): [q][lane]
        // (a tile without fast continua in a folded class still runs the first chunk: four zero rows)
        for (int e = nF * LSX_WAVE + threadIdx.x; e < max(lsx_rs_fold_pad(nF), 4) * LSX_WAVE; e += 2 * LSX_WAVE) atab[e] = 0.0;
        for (int e = threadIdx.x; e < nF * LSX_WAVE; e += 2 * LSX_WAVE) {
            const int q = e >> 6, l = e & (LSX_WAVE - 1);
            // (a lane without a wavelength of its own shadows the tile's last one and stores the same bits to the same address: it needs
            // that wavelength's cross-sections, not zeros)
            const int jl = l - (l / LW) * LW;
            const int laq = la0 + (jl < nla ? jl : nla - 1), lt = laq - slots[NPT + q].Nblue;
            const bool a = lt >= 0 && lt < slots[NPT + q].Nlam && p.active[(size_t)slots[NPT + q].trans * Nspect + laq] != 0;
            atab[e] = a ? p.alpha[slots[NPT + q].wl_off + lt] : 0.0;
            if constexpr (EPI) wtab[e] = (a && jl < nla) ? p.wl[slots[NPT + q].wl_off + lt] : 0.0;      // (the wavelength sums take real wavelengths only)
        }
        if constexpr (EPI) {
            for (int e = nF * LSX_WAVE + threadIdx.x; e < max(lsx_rs_fold_pad(nF), 4) * LSX_WAVE; e += 2 * LSX_WAVE) wtab[e] = 0.0;
            // per fast continuum: its linking bits (DevSlot.lkbits) and whether it is the first of its atom's run
            for (int q = threadIdx.x; q < LSX_WAVE; q += 2 * LSX_WAVE) {
                unsigned bits = 0;
                if (q < nF) bits = (slots[NPT + q].lkbits & 0x00ffffffu) | ((q == 0 || slots[NPT + q].atom != slots[NPT + q - 1].atom) ? 0x80000000u : 0u);
                ((__attribute__((address_space(3))) unsigned*)qinf)[q] = bits;
            }
        }
    }
    __syncthreads();
#ifdef LSX_CLOCK
    unsigned long long tk_s1; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk_s1)::"memory");
#endif
    const int lc3 = cc * 3, lcg = LSX_RS_SEG * NPT + cc * LSX_RS_GEO;   // the lane's column inside a row: slot values at lc3 + 16 u + t, geometry at lcg + {0, 1 (, 2: 1 / T)}
    const int lcf = RL0 + cc * 3;                          // ... the fast continua's triples (n_i, n_j nsr, nsr) at lcf + 16 q + {0, 1, 2}
    constexpr int TU = LSX_RS_SEG;                         // doubles between two segments of a row
    // what the tile's fast continua add to opacity and emissivity at the depth of step v (rh_method.py:284-286, 453-455, 613-614):
    // chi += sum_q alpha_q n_i,q - E sum_q alpha_q (n_j nsr)_q,  eta += u_la E sum_q alpha_q (n_j nsr)_q
    auto fast_fold = [&](const int v, const double E, const double ula, double& chi, double& eta) __attribute__((always_inline)) {
        if constexpr (FOLD) {
            // four continua at a time, straight-line: the chunk's eight LDS reads are in flight together (a loop over nF waited
            // for each continuum's reads in turn: measured, the fold then cost what the pre-pass had cost)
            const lds_f64* fr = ring_row(v) + lcf;
            const lds_f64* al = atab + lane;
            double SA = 0.0, SB = 0.0;
            // (two at a time where registers are short: the two-slot instances, and the continuum tiles' instance at three waves per SIMD)
            constexpr int CH = NPT <= 1 ? 4 : 2;
            auto chunk = [&](const int q0) __attribute__((always_inline)) {
                double a[CH], x[CH], y[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) { a[i] = al[(q0 + i) * LSX_WAVE]; x[i] = fr[TU * (q0 + i) + 0]; y[i] = fr[TU * (q0 + i) + 1]; }
#pragma unroll
                for (int i = 0; i < CH; ++i) { SA = fma(a[i], x[i], SA); SB = fma(a[i], y[i], SB); }
            };
#ifdef LSX_ABL_FOLD_NOCHUNK
            SA = al[0]; SB = fr[0];                    // ablation build (wrong results): no sums over the continua
#else
            chunk(0);
            if constexpr (CH == 2) { if (nF > 2) chunk(2); }
            if (nF > 4) { chunk(4); if constexpr (CH == 2) { if (nF > 6) chunk(6); } }
            if (nF > 8) { chunk(8); if constexpr (CH == 2) { if (nF > 10) chunk(10); } }
#endif
            const double EB = E * SB;
            chi += SA - EB;
            eta = fma(ula, EB, eta);
        }
    };

    // ---- per-lane bases: every stream of a column is addressed as (wave-uniform base of column col0) + 32-bit byte offset
    const size_t til_col = (size_t)ntile * Ns * LW;
    const size_t tb0 = ((size_t)col0 * ntile + tile_id) * Ns * LW;
    const unsigned o_til = (unsigned)((size_t)cc * til_col * 8u) + (unsigned)j * 8u;           // + k * LW * 8
    const double* __restrict__ bgchi = ((nF > 0 && !FOLD) ? p.bgxchi_T : p.bgchi_T) + tb0;      // (FOLD: the plain background; see fast_fold)
    const double* __restrict__ bgeta = ((nF > 0 && !FOLD) ? p.bgxeta_T : p.bgeta_T) + tb0;
    // ... read as (chi, eta) pairs, ONE 16-byte load per lane and depth (lsx_dev.h, bgce_T / bgxce_T; round 6: the same bytes in one
    // request less per step: C4 -2 %, profiles/r06_bound_evidence.md 6 -- what a wave waits for is its requests, not their bytes)
    constexpr bool BGP = LSX_BG_PAIRS != 0;
    [[maybe_unused]] const double* __restrict__ bgce = BGP ? ((nF > 0 && !FOLD) ? p.bgxce_T : p.bgce_T) + 2 * tb0 : nullptr;
    const double* __restrict__ Jdag = p.Jdag_T + tb0;
    double* __restrict__ Jnew = p.Jnew_T + tb0;
    double* __restrict__ psibar = p.Psi2_T + ((size_t)dir * p.ncol * ntile) * Ns * LW + tb0;
    [[maybe_unused]] const double* __restrict__ Eb = p.E_T + tb0;      // (LSX_ELANE=0 and the ablation builds: the Boltzmann factor as a stream)
    // the line-profile store interleaves the columns of a group inside every block row (lsx_dev.h, phi_elem): with G = NC the five
    // columns of this wavefront are ONE group, and row x of a block is one contiguous run [c < NC][l < len] for the whole wavefront
    const int PG = p.phi_G;                                          // NC, or 1: the plain per-column store (LSX_PHI_GROUP=1)
    const double* __restrict__ phi0 = p.phi_T + (size_t)col0 * p.phi_col_stride;
#ifdef LSX_ABL_PHI_ONECOL
    const int cphi = 0;              // ablation build (wrong results): the five columns of a wavefront read the FIRST column's profiles
#else
    const int cphi = cc;
#endif
    const double* __restrict__ corr = CORR ? p.corr_T + (size_t)col0 * p.corr_col_stride + tilep->corr_off : nullptr;
    const unsigned o_corr = CORR ? (unsigned)((size_t)cc * p.corr_col_stride * 8u) + (unsigned)j * 8u : 0u;
    double* __restrict__ ppsum = LK ? p.Psi3_T + ((size_t)dir * p.ncol + col0) * p.pp_col_stride + tilep->pp_off : nullptr;
    // EPI: the partner wave's halves of the same sums (what it stored as the first visitor of the depths this wave visits second)
    const double* __restrict__ psibar_o = p.Psi2_T + ((size_t)(1 - dir) * p.ncol * ntile) * Ns * LW + tb0;
    const double* __restrict__ ppsum_o = LK ? p.Psi3_T + ((size_t)(1 - dir) * p.ncol + col0) * p.pp_col_stride + tilep->pp_off : nullptr;
    const unsigned o_pp = LK ? (unsigned)((size_t)cc * p.pp_col_stride * 8u) + (unsigned)j * 8u : 0u;
    const size_t plane = (size_t)Ns * LW;
    const bool compact = p.phi_compact != 0;
    const double wav = p.wavelength[la];
    const double u_la = p.u_la[la];
    // the Boltzmann factor of the tile's continua at depth row v (rh_method.py:453), formed from the row's 1 / T (lsx_dev.h,
    // boltzmann_factor; the stream E_T holds the same bits for the kernels that read it)
    const double aE = boltzmann_lane_constant(wav);
    auto E_of = [&](const int v, const double E_stream) __attribute__((always_inline)) {
#if LSX_ELANE
        if constexpr (HASC || FOLD) return boltzmann_factor(aE, ring_row(v)[lcg + 2], etab);
        else return 0.0;
#else
        return E_stream;
#endif
    };
    // angle quadrature: wave-uniform
    // two-slot instances: the ten quadrature constants live in LDS behind the parked totals and are read where they are used
    // (broadcast reads with immediate offsets): twenty vector registers less in the instances that sit at the 256-register limit
    #ifdef LSX_RS_PF2_QLDS
    constexpr bool QLDS = NPT >= 2 || (PAR && LSX_RSP_WPE >= 2) || (PF2 && NPT >= 1);
#else
    constexpr bool QLDS = NPT >= 2 || (PAR && LSX_RSP_WPE >= 2) || (FOLD && NPT == 0);      // (the folded continuum instance: three waves per SIMD)
#endif
    lds_f64* const qtab = ring_end + (size_t)2 * NC * NV * lsx_rs_park(NPT, PAR || EPI);
    if (QLDS && threadIdx.x < 2 * NR) qtab[threadIdx.x] = threadIdx.x < NR ? LSX_CONST(double, p.zmu)[threadIdx.x] : LSX_CONST(double, p.wmuh)[threadIdx.x - NR];
    double zmu_r[NR], wmuh_r[NR];
#pragma unroll
    for (int m = 0; m < NR; ++m) {
        zmu_r[m] = LSX_CONST(double, p.zmu)[m];
        wmuh_r[m] = LSX_CONST(double, p.wmuh)[m];                     // w_mu / 2; the 4 pi of rh_method.py:661-665 goes into wlam
        // kept in vector registers (the kernel has them to spare at two waves per SIMD; its scalar registers are what runs out)
        if constexpr (!QLDS) asm volatile("" : "+v"(zmu_r[m]), "+v"(wmuh_r[m]));
    }
    auto zmu_of = [&](int m) __attribute__((always_inline)) { return QLDS ? (double)qtab[m] : zmu_r[m]; };
    auto wmuh_of = [&](int m) __attribute__((always_inline)) { return QLDS ? (double)qtab[NR + m] : wmuh_r[m]; };
#define zmu(m) zmu_of(m)
#define wmuh(m) wmuh_of(m)

    if constexpr (QLDS) __syncthreads();
#ifdef LSX_CLOCK
    unsigned long long tk_s2; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk_s2)::"memory");
#endif
    // ---- per-slot lane constants
    unsigned pact = 0;
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const int l = la - slots[u].Nblue;
        if (l >= 0 && l < slots[u].Nlam && p.active[slots[u].trans * Nspect + la] != 0) pact |= 1u << u;
    }
    unsigned phi_o[NS], phi_k[NS], phi_m[NS];       // lines: byte offset of (depth 0, ray 0), per depth, per ray
#if defined(LSX_ABL_PHI_WIDE) || defined(LSX_ABL_PHI_PAIRS)
    unsigned phi_w[NS];
#endif
    double wlam[NS], alv[NS], cB[NS], Vc[NS], Uc[NS];
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        const bool a = (pact >> u) & 1u;
        const bool line = u < NL;
        const int l = a ? la - slots[u].Nblue : 0;
        const int len = slots[u].len;
        const int lb = a ? la - slots[u].first : 0;
        // element ((dir Ns + k) Nrays + mu) len + lb of the (tile, line) block (compact: k len + lb); a lane outside the line's
        // range reads the column's zero pad at every depth and ray
        // grouped store: element PG (base + x len) + c len + l of the group (x = (dir Ns + k) NR + mu); plain store (PG = 1): the
        // column's own base + x len + l.  A lane outside the line's range reads a zero pad at every depth and ray: the last
        // element of the group (of its column)
        const long xl0 = (long)slots[u].base + (compact ? 0L : (long)dir * Ns * NR * len);
        const long e0 = !line ? 0L : PG > 1 ? (a ? (long)PG * xl0 + (long)cphi * len + lb : (long)PG * p.phi_col_stride - 1)
                                            : (long)cphi * p.phi_col_stride + (a ? xl0 + lb : (long)p.phi_col_stride - 1);
        phi_o[u] = (unsigned)(e0 * 8);
#if defined(LSX_ABL_PHI_WIDE) || defined(LSX_ABL_PHI_PAIRS)
        // ablation build (wrong results; profiles/r06_bound_evidence.md 6): the five rays' profile values of a lane as THREE loads (16 + 16 + 8
        // bytes) at lane-contiguous 40-byte pieces of the same 2400-byte region the five 480-byte rows of (direction, depth) occupy --
        // what a [column][wavelength][ray] order of the block rows would cost, before anybody changes the layout
        phi_w[u] = (line && a && !compact && PG > 1) ? (unsigned)(((long)cphi * len + lb) * 32) : 0u;
#endif
        phi_k[u] = (line && a) ? (unsigned)(PG * (compact ? 1 : NR) * len * 8) : 0u;
        phi_m[u] = (line && a && !compact) ? (unsigned)(PG * len * 8) : 0u;
        wlam[u] = (a && act) ? (4.0 * kPi) * p.wl[slots[u].wl_off + l] : 0.0;   // :451/:455, :665 without the angle weight
        alv[u] = (a && !line) ? p.alpha[slots[u].wl_off + l] : 0.0;
        cB[u] = slots[u].cB; Vc[u] = slots[u].Vc; Uc[u] = slots[u].Uc;
        if constexpr (NPT == 1 && !PAR) asm volatile("" : "+v"(cB[u]), "+v"(Vc[u]), "+v"(Uc[u]));
    }

#ifdef LSX_CLOCK
    unsigned long long tk_s3; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk_s3)::"memory");
#endif
    // ---- one depth's operands ------------------------------------------------------------------------------------------
    struct Ops {
        double bc, be, jd, E;
        double ph[NS][NR];                   // line profile per ray
        double cr[CORR ? NLK : 1][CORR ? 3 : 1];   // <2,2,true,0>: the continua's share of atom.eta, atom.chi[i], atom.chi[j]
    };
    auto load_ops = [&](int kk, Ops& o) __attribute__((always_inline)) {
#ifdef LSX_ABL_NOLOAD
        // ablation build (profiles/r03_bound_evidence.md): every depth gets the operands of the middle one, fetched once; the
        // values pass through an opaque move so that nothing computed from them leaves the loop
        kk = Ns / 2;
#endif
#ifdef LSX_ABL_SHARED_READS
        // ablation build (wrong results; profiles/r06_bound_evidence.md 5): the up-going wave reads the ray-independent streams at the depth
        // its partner is reading at the same moment -- what a call would cost if both directions of a column could share those reads
        const unsigned kt = o_til + (unsigned)((dir ? Ns - 1 - kk : kk) * LW) * 8u;
#else
        const unsigned kt = o_til + (unsigned)(kk * LW) * 8u;
#endif
#ifdef LSX_NT_BG      // (measured alternative: the ray-independent streams non-temporally too)
        o.jd = ld_once(Jdag, kt);
        o.bc = ld_once(bgchi, kt);
        o.be = ld_once(bgeta, kt);
#else
        o.jd = at(Jdag, kt);
        if constexpr (BGP) {
            typedef double bg_pair __attribute__((ext_vector_type(2)));
            const bg_pair v = *reinterpret_cast<const bg_pair*>(reinterpret_cast<const char*>(bgce) + 2u * kt);
            o.bc = v.x; o.be = v.y;
        } else {
            o.bc = at(bgchi, kt);
            o.be = at(bgeta, kt);
        }
#endif
        o.E = 0.0;
#ifdef LSX_ABL_FOLD_NOE
        if constexpr (HASC) o.E = at(Eb, kt);          // ablation build (wrong results): the folded instances do not read the Boltzmann stream
        else if constexpr (FOLD) o.E = 0.5;
#elif !LSX_ELANE
        if constexpr (HASC || FOLD) o.E = at(Eb, kt);
#endif
#if defined(LSX_ABL_PHI_PAIRS)
        // ablation build (wrong results; profiles/r06_bound_evidence.md 6): the rays' profile values as (ray 0, ray 1), (ray 2, ray 3) PAIRS and
        // ray 4 -- two 16-byte loads and one 8-byte load whose lanes stay contiguous (960 / 960 / 480 bytes per wave) over the bytes rows
        // 0-1, 2-3 and 4 occupy today: what a [ray pair][column][wavelength][2] order of the block rows would cost
        static_assert(NR == 5, "ablation: five rays");
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            typedef double ph2 __attribute__((ext_vector_type(2)));
            const unsigned o0 = phi_o[u] + (unsigned)kk * phi_k[u];
            // + (c len + l) 8: the lane's piece doubles; a lane outside the line reads the group's last element: its pair ends there
            const char* q = reinterpret_cast<const char*>(phi0) + o0 + phi_w[u] / 4u - (phi_m[u] ? 0u : 8u);
            const ph2 v0 = __builtin_nontemporal_load(reinterpret_cast<const ph2*>(q)), v1 = __builtin_nontemporal_load(reinterpret_cast<const ph2*>(q + 2u * phi_m[u]));
            o.ph[u][0] = v0.x; o.ph[u][1] = v0.y; o.ph[u][2] = v1.x; o.ph[u][3] = v1.y;
            o.ph[u][4] = ld_once(phi0, o0 + 4u * phi_m[u]);
        }
#elif defined(LSX_ABL_PHI_WIDE)
        static_assert(NR == 5, "ablation: five rays");
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            typedef double ph2 __attribute__((ext_vector_type(2)));
            const char* q = reinterpret_cast<const char*>(phi0) + (phi_o[u] + (unsigned)kk * phi_k[u] + phi_w[u]);
            const ph2 v0 = __builtin_nontemporal_load(reinterpret_cast<const ph2*>(q)), v1 = __builtin_nontemporal_load(reinterpret_cast<const ph2*>(q + 16));
            o.ph[u][0] = v0.x; o.ph[u][1] = v0.y; o.ph[u][2] = v1.x; o.ph[u][3] = v1.y;
            o.ph[u][4] = __builtin_nontemporal_load(reinterpret_cast<const double*>(q + 32));
        }
#else
#pragma unroll
        for (int u = 0; u < NL; ++u)
#pragma unroll
            for (int m = 0; m < NR; ++m) o.ph[u][m] = ld_once(phi0, phi_o[u] + (unsigned)kk * phi_k[u] + (unsigned)m * phi_m[u]);
#endif
        if constexpr (CORR) {
            const unsigned kq = o_corr + (unsigned)(kk * LW) * 8u;
#pragma unroll
            for (int u = 0; u < NL; ++u)
#pragma unroll
                for (int q = 0; q < NCR; ++q) o.cr[u][q] = at(corr, (unsigned)((3 * u + q) * plane) * 8u + kq);
        }
#ifdef LSX_ABL_NOLOAD
        asm volatile("" : "+v"(o.jd), "+v"(o.bc), "+v"(o.be), "+v"(o.E));
#pragma unroll
        for (int u = 0; u < NL; ++u)
#pragma unroll
            for (int m = 0; m < NR; ++m) asm volatile("" : "+v"(o.ph[u][m]));
#endif
    };
    // total opacity of ray m from one depth's operands (rh_method.py:613, 279-285)
    auto chi_of = [&](const Ops& o, int v, int m) __attribute__((always_inline)) {          // v: step index of the depth
        const lds_f64* tk = ring_row(v) + lc3;
        double c = o.bc;
        const double Ev = E_of(v, o.E);
        if constexpr (FOLD) { double e_ = 0.0; fast_fold(v, Ev, u_la, c, e_); }
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            if (u < NL) c = fma(tk[TU * u + 0], o.ph[u][m], c);
            else {
                const bool a = (pact >> u) & 1u;
                const double Vji = a ? (tk[TU * u + 2] * Ev) * alv[u] : 0.0;
                c += tk[TU * u + 0] * alv[u] - tk[TU * u + 1] * Vji;
            }
        }
        return c;
    };

    // ---- boundary conditions: formal_solver.py:203-209 ------------------------------------------------------------------
    double Iu[NR], chi_prev[NR], S_prev[NR], dtau_prev[NR];
#pragma unroll
    for (int m = 0; m < NR; ++m) { Iu[m] = 0.0; chi_prev[m] = 1.0; S_prev[m] = 0.0; dtau_prev[m] = 1.0; }
    Ops opA, opB, opC;
    load_ops(kS, opA);
    load_ops(kS + dk, opB);
    if (dir) {
        const Ops &cur = opA, &nxt = opB;
        const auto* tcol = p.temperature + (size_t)col * Ns;
        const double B0 = planck(tcol[Ns - 2], wav), B1 = planck(tcol[Ns - 1], wav);
        const double hz = ring_row(1)[lcg];                          // 0.5 |z[Ns - 2] - z[Ns - 1]|: the interval behind depth kS + dk (the up sweep's geometry: below the depth)
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            const double dtau_uw = zmu(m) * (chi_of(cur, 0, m) + chi_of(nxt, 1, m)) * hz;
            Iu[m] = B1 - (B0 - B1) / dtau_uw;
        }
    }
#ifdef LSX_CLOCK
    unsigned long long tk_s4; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk_s4)::"memory");
#endif
    double dJ = 0.0;

    // Gamma integrands of the previous depth wait in this wave's reduction rows [value][lane]; lane (c, q) of the first
    // NC x NV lanes adds up the 12 wavelengths of column c for value q and parks the total in entry (step mod PE) of its row
    // park[c NV + q][PE]; every PE steps (and at the end) each row leaves as ONE coalesced store of PE consecutive depths.
    const int o_c = lane / NV, o_q = lane - o_c * NV;
    const bool own = lane < NC * NV;
    constexpr int PE = lsx_rs_park(NPT, PAR || EPI);
    lds_f64* const park = ring_end + (size_t)dir * NC * NV * PE;
    double* __restrict__ gbase = p.Gpart + (((size_t)col0 * p.nslot_total + slot0) * 4 + (size_t)dir) * Ns;      // + (c nslot 4 + q 2) Ns + k
    auto flush = [&](int sprev) __attribute__((always_inline)) {         // the totals of step sprev (depth kS + dk sprev)
        if constexpr (NPT >= 1) {
            typedef double lds_pair __attribute__((ext_vector_type(2)));
            const auto* src = (const __attribute__((address_space(3))) lds_pair*)(red + o_q * RROW + (o_c < NC ? o_c : 0) * LW);
            lds_pair v2 = src[0];
            double acc = v2.x + v2.y;
#pragma unroll
            for (int e = 1; e < LW / 2; ++e) { v2 = src[e]; acc += v2.x + v2.y; }
            const int e64 = sprev & (PE - 1);
            // every lane writes: the lanes that own no (column, value) pair put their sum into the wave's dJ row, which is only
            // read after the last step has overwritten it -- no branch around the reads and the adds, so they can be scheduled
            // into the ray-independent part of the step
            *(own ? &park[lane * PE + e64] : &red[NV * RROW + lane]) = acc;
            if (e64 == PE - 1 || sprev == Ns - 1) {
                __builtin_amdgcn_wave_barrier();
                const int kk = kS + dk * (sprev - e64 + lane);             // the depth parked in entry `lane` of every row
                if (lane <= e64) {
                    for (int c = 0; c < ncg; ++c) {
                        if (p.colmask && LSX_CONST(uint8_t, p.colmask)[col0 + c] == 0) continue;      // a frozen column keeps its slabs
#pragma unroll
                        for (int q = 0; q < NV; ++q) {
#ifdef LSX_NT_ST
                            __builtin_nontemporal_store((double)park[(c * NV + q) * PE + lane], &gbase[((size_t)c * p.nslot_total * 4 + (size_t)q * 2) * Ns + kk]);
#else
                            gbase[((size_t)c * p.nslot_total * 4 + (size_t)q * 2) * Ns + kk] = park[(c * NV + q) * PE + lane];
#endif
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    };

    // ---- EPI: the Gamma integrands of the tile's fast continua at the depth of step s, by the wave that visits it second ---------
    // (rh_method.py:652, 677-681 for transitions whose coefficients do not depend on the ray; the arithmetic of lsx_fast.h,
    // fast_gamma_cols_rows, per wavelength -- here one wavelength per lane).  Jv: the depth's mean intensity (both directions), Ptot:
    // Psibar (both directions, x 4 pi), PPt[u]: sum_mu w Psi* phi of line u (likewise); X, njUc, w3k: this depth's line operands
    // (chi_line = X phi, eta_line = njUc phi, wphi).  Per continuum two values (rates j -> i and i -> j), per linked line two more (what
    // the linked continua add to the line's own rates: the correction slots), reduced over the tile's wavelengths LSX_RS_EPI_ROUND
    // values at a time through this wave's rows `fred` -- lane (c, v) of the first 60 sums the twelve wavelengths of column c for value
    // v -- and stored straight into the slabs (one entry per rate and depth, both directions in it, as k_fast_gamma_cols wrote them).
    constexpr int ER = LSX_RS_EPI_ROUND;
    double sW = 0.0;
    if constexpr (EPI) {
#pragma unroll
        for (int m = 0; m < NR; ++m) sW += 2.0 * wmuh(m) * (4.0 * kPi);          // both directions
    }
    auto epi_fast = [&](const int s, const int k, const double Jv, const double Ptot, const double (&PPt)[NLK], const double E,
                        const double (&Xl)[NS], const double (&nUl)[NS], const double (&wpl)[NS]) __attribute__((always_inline)) {
        if constexpr (EPI) {
            if (nF > 0) {
                constexpr int NLL = LK ? NL : 0;
                const double sI = Jv * (4.0 * kPi), sPsi = Ptot;
                const lds_f64* fr = ring_row(s) + lcf;
                const lds_f64* al = atab + lane;
                const lds_f64* wq = wtab + lane;
                const auto* qi = (const __attribute__((address_space(3))) unsigned*)qinf;
                lds_f64* const rows = fred + (size_t)dir * ER * LSX_WAVE;
                double tchi[NLK], teta[NLK], tU[NLK], xa[NLK], xb[NLK];
#pragma unroll
                for (int u = 0; u < NLK; ++u) {
                    tchi[u] = teta[u] = tU[u] = xa[u] = xb[u] = 0.0;
                    if (u < NLL) { tchi[u] = Xl[u] * PPt[u]; teta[u] = nUl[u] * PPt[u]; tU[u] = Uc[u] * PPt[u]; }
                }
                // one round of the wavelength reduction: rows [0, vr) hold values v0 .. v0 + vr - 1 (value 2 q + e: rate e of fast
                // continuum q; then 2 u + e: correction e of linked line u)
                const int oc = lane / ER, ov = lane - oc * ER;
                const bool ocol = oc < ncg && (p.colmask ? LSX_CONST(uint8_t, p.colmask)[col0 + (oc < ncg ? oc : 0)] != 0 : true);
                auto flush_round = [&](const int v0, const int vr) __attribute__((always_inline)) {
#ifdef LSX_ABL_EPI_NOFLUSH
                    return;                 // ablation build (wrong results): the fast values are formed and written to the rows, not reduced or stored
#endif
                    typedef double lds_pair __attribute__((ext_vector_type(2)));
                    __builtin_amdgcn_wave_barrier();
                    const auto* src = (const __attribute__((address_space(3))) lds_pair*)(rows + ov * LSX_WAVE + (oc < NC ? oc : 0) * LW);
                    lds_pair v2 = src[0];
                    double acc = v2.x + v2.y;
#pragma unroll
                    for (int e = 1; e < LW / 2; ++e) { v2 = src[e]; acc += v2.x + v2.y; }
                    if (ocol && ov < vr) {
                        const int gv = v0 + ov;
                        const int slot = slot0 + NPT + (gv >> 1);          // fast continua first, then the correction slots: consecutive in the slot table
                        p.Gpart[(((size_t)(col0 + oc) * p.nslot_total + slot) * 4 + (size_t)(gv & 1) * 2) * Ns + k] = acc;
                    }
                    __builtin_amdgcn_wave_barrier();
                };
                int v0 = 0, vr = 0;
#ifdef LSX_ABL_EPI_NOMATH
                // ablation build (wrong results): the rounds of the reduction and their stores without the arithmetic in front of them
                for (int q = 0; q < nF; ++q) {
                    rows[vr * LSX_WAVE + lane] = sI; rows[(vr + 1) * LSX_WAVE + lane] = sPsi;
                    vr += 2;
                    if (vr == ER) { flush_round(v0, ER); v0 += ER; vr = 0; }
                }
                if (vr > 0) flush_round(v0, vr);
                return;
#endif
                for (int q0 = 0; q0 < nF;) {                               // one atom's run of continua at a time
                    int q1 = q0 + 1;
                    while (q1 < nF && !(qi[q1] & 0x80000000u)) ++q1;
                    const unsigned lk0 = qi[q0];
                    double Usum = 0.0, Esum = 0.0, Csum = 0.0, XCi[NLK];
#pragma unroll
                    for (int u = 0; u < NLK; ++u) XCi[u] = 0.0;
                    for (int q = q0; q < q1; ++q) {                        // rh_method.py:284-286, 453-455, 613-614; atom.chi / atom.U / atom.eta of :616-627
                        const double alf = al[q * LSX_WAVE], ni = fr[TU * q + 0], br = fr[TU * q + 1], nr = fr[TU * q + 2];
                        const double g = nr * E, ng = br * E, hq = ni - ng;
                        Usum = fma(g, alf, Usum);
                        Esum = fma(ng, alf, Esum);
                        if constexpr (LK) {
                            Csum = fma(hq, alf, Csum);                     // = -atom.chi[j]
                            const unsigned lkq = qi[q];
#pragma unroll
                            for (int u = 0; u < NLL; ++u)
                                if (lkq & (2u << (8 * u))) XCi[u] = fma(hq, alf, XCi[u]);      // chi of the continua on line u's lower level
                        }
                    }
                    const double U_j = u_la * Usum, etaA = u_la * Esum;
                    double le = 0.0;
                    if constexpr (LK) {
#pragma unroll
                        for (int u = 0; u < NLL; ++u)
                            if (lk0 & (1u << (8 * u))) {                    // line u belongs to this atom
                                le += teta[u];
                                const double tt = (wlam[u] * (1.0 / (4.0 * kPi))) * PPt[u];
                                xa[u] = fma(tt, etaA, xa[u]);
                                xb[u] = fma(tt, XCi[u], xb[u]);
                            }
                    }
                    const double sIe = (sI - etaA * sPsi) - le;
                    const double T = fma(u_la, sW, sIe), UP = U_j * sPsi;
                    for (int q = q0; q < q1; ++q) {
                        const double alf = al[q * LSX_WAVE], wl_ = wq[q * LSX_WAVE];
                        const double ni = fr[TU * q + 0], br = fr[TU * q + 1], nr = fr[TU * q + 2];
                        const double g = nr * E, hq = ni - br * E;
                        const double wa = alf * wl_;
                        double a1 = wa * fma(-hq, UP, g * T), a2 = wa * sIe;
                        if constexpr (LK) {
                            const unsigned lkq = qi[q];
                            double lchi = 0.0, lU = 0.0;
#pragma unroll
                            for (int u = 0; u < NLL; ++u) {
                                if (lkq & (2u << (8 * u))) lchi += tchi[u];
                                if (lkq & (4u << (8 * u))) { lchi -= tchi[u]; lU += tU[u]; }
                            }
                            a1 = fma(-wl_, lchi * U_j, a1);                 // the lines on the continuum's lower level: chi_a[i] Psi U_a[j]
                            a2 = fma(wl_, Csum * lU, a2);                   // -chi_a[j] (Psi U_a[i]), U_a[i] from the lines that end there
                        }
                        rows[vr * LSX_WAVE + lane] = a1;
                        rows[(vr + 1) * LSX_WAVE + lane] = a2;
                        vr += 2;
                        if (vr == ER) { flush_round(v0, ER); v0 += ER; vr = 0; }
                    }
                    q0 = q1;
                }
                if constexpr (LK) {
                    // the correction slots of the tile's lines: -wphi [Uc dB + Vc dA] and -wphi cB dA (lsx_fast.h)
#pragma unroll
                    for (int u = 0; u < NLL; ++u) {
                        rows[vr * LSX_WAVE + lane] = -wpl[u] * fma(Uc[u], xb[u], Vc[u] * xa[u]);
                        rows[(vr + 1) * LSX_WAVE + lane] = -wpl[u] * (cB[u] * xa[u]);
                        vr += 2;
                        if (vr == ER) { flush_round(v0, ER); v0 += ER; vr = 0; }
                    }
                }
                if (vr > 0) flush_round(v0, vr);
            }
        }
    };

    if constexpr (PAR) {
    // ---- N4: the monotonic piecewise-parabolic rule (include/lsx.h; lsx_dev.h, parabolic_point_fast; the one-ray-per-lane form is
    // sweep_tile_par of lsx_sweep.hip).  Point m needs the source function and the optical depth of its DOWNWIND interval, so step s
    // (depth k = kS + dk s) builds opacity, emissivity and source function of depth k from the operands requested one step ago and
    // then finishes point m = s - 1: formal solution, angle sums, Gamma integrands, J.  What a lane carries from step to step per ray:
    // I and the upwind difference quotient of point m - 1, and of point m its source function, opacity, 1 / opacity, the optical
    // depth of the upwind interval and its reciprocal; per wavelength: point m's line profiles (a copy of the operand registers:
    // the buffer they arrived in takes the request for depth k + 1) and J-dagger.
    static_assert(NPT == NL && (NPT == 0 || FACT) && !CORR, "ray-serial parabolic instances: line-only tiles with factored Gamma integrands");
    // 1 / opacity, opacity and the line profiles of point m.  At one wave per SIMD (the default, see LSX_RSP_WPE) they stay in registers:
    // a lone wave waits out every LDS round trip.  A build for two waves per SIMD (LSX_RSP_WPE=2) parks them in lane-private LDS cells
    // (one row of 64 per value and wave, behind the angle quadrature; volatile: a parked value must not travel from its store to the
    // next step's load in a register) -- twenty to thirty registers less, which was not enough (profiles/r04_bound_evidence.md 4).
    constexpr bool CELLS = LSX_RSP_WPE >= 2;
    double pu[NR], S_c[NR], dtau_u[NR], ru[NR];       // (+ Iu: the intensity of point m - 1)
    double jd_c;
    double rchi_r[NR], chi_r[NR], ph_r[NS][NR];
    volatile lds_f64* const cell = qtab + 2 * NR + 2 + (size_t)dir * lsx_rs_par_rows(NPT) * LSX_WAVE + lane;      // row r: cell[r * 64]
    auto rchi_c = [&](int m) __attribute__((always_inline)) { if constexpr (CELLS) return (double)cell[m * LSX_WAVE]; else return rchi_r[m]; };
    auto chi_c = [&](int m) __attribute__((always_inline)) { if constexpr (CELLS) return (double)cell[(NR + m) * LSX_WAVE]; else return chi_r[m]; };
    auto ph_c = [&](int u, int m) __attribute__((always_inline)) { if constexpr (CELLS) return (double)cell[(2 * NR + u * NR + m) * LSX_WAVE]; else return ph_r[u][m]; };
    auto set_rchi = [&](int m, double v) __attribute__((always_inline)) { if constexpr (CELLS) cell[m * LSX_WAVE] = v; else rchi_r[m] = v; };
    auto set_chi = [&](int m, double v) __attribute__((always_inline)) { if constexpr (CELLS) cell[(NR + m) * LSX_WAVE] = v; else chi_r[m] = v; };
    auto set_ph = [&](int u, int m, double v) __attribute__((always_inline)) { if constexpr (CELLS) cell[(2 * NR + u * NR + m) * LSX_WAVE] = v; else ph_r[u][m] = v; };
    {   // point 0's own values (the boundary condition above has read the same operands)
        const lds_f64* tk = ring_row(0) + lc3;
        const double etaB = opA.be + ring_row(0)[lcg + 1] * opA.jd;
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            double chiTot = opA.bc, etaTot = etaB;
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                chiTot = fma(tk[TU * u + 0], opA.ph[u][m], chiTot);
                etaTot = fma(tk[TU * u + 1], opA.ph[u][m], etaTot);
                set_ph(u, m, opA.ph[u][m]);
            }
            set_chi(m, chiTot);
            const double rc = rcp(chiTot);
            set_rchi(m, rc);
            S_c[m] = etaTot * rc;                                         // :632
            pu[m] = 0.0; dtau_u[m] = 1.0; ru[m] = 1.0;
        }
        jd_c = opA.jd;
    }
    // One point, ray by ray: `ray(m, I, Psi)` hands over the formal solution of ray m at point mpt; the angle sums, the Gamma integrands and
    // the stores follow (rh_method.py:638-681, as pass C of the linear step).  The per-ray values die inside the loop: nothing of a ray
    // but its recurrence state outlives its turn.
    auto point = [&](const int mpt, auto phase_c, auto last_c, const double jhalf, auto&& ray) __attribute__((always_inline)) {
        constexpr int PH = decltype(phase_c)::value;              // 0 first visitor, 1 midpoint, 2 second visitor
        constexpr bool LASTPT = decltype(last_c)::value;
        const int km = kS + dk * mpt;
        const unsigned kt = o_til + (unsigned)(km * LW) * 8u;
        const lds_f64* tm = ring_row(mpt) + lc3;
        double X[NS], njUc[NS], w3k[NS];
#pragma unroll
        for (int u = 0; u < NPT; ++u) { X[u] = tm[TU * u + 0]; njUc[u] = tm[TU * u + 1]; w3k[u] = tm[TU * u + 2]; }
        double Jacc = 0.0, Pacc = 0.0, PP[NLK], G1[NS], G2[NS];
#pragma unroll
        for (int u = 0; u < NLK; ++u) PP[u] = 0.0;
#pragma unroll
        for (int u = 0; u < NS; ++u) G1[u] = G2[u] = 0.0;
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            double phm[NS];
#pragma unroll
            for (int u = 0; u < NL; ++u) phm[u] = ph_c(u, m);
            double I, Psi;
            ray(m, I, Psi, phm);
            if constexpr (LASTPT) {
                if (dir == 1 && act) p.Iout[((size_t)col * Nspect + la) * NR + m] = I;      // emergent intensity, :638
            }
            Jacc = fma(wmuh(m), I, Jacc);                                  // :640
            const double wP = wmuh(m) * Psi;
            Pacc += wP;
            if constexpr (LK) {
#pragma unroll
                for (int u = 0; u < NL; ++u) PP[u] = fma(wP, phm[u], PP[u]);
            }
            if constexpr (NPT >= 1) {       // the factored integrands (see the linear step)
                double qv[NS], Ie[NS], tv[NS];
#pragma unroll
                for (int u = 0; u < NPT; ++u) qv[u] = Psi * phm[u];
                if constexpr (NPT == 2 && TOPO == 1) {
                    const double Ic = fma(-njUc[1], qv[1], fma(-njUc[0], qv[0], I));
                    const double tc = fma(-X[1], qv[1], fma(-X[0], qv[0], 1.0));
                    Ie[0] = Ie[1] = Ic;
                    tv[0] = tv[1] = tc;
                } else {
#pragma unroll
                    for (int u = 0; u < NPT; ++u) {
                        Ie[u] = fma(-njUc[u], qv[u], I);            // Ieff = I - Psi* eta, :652
                        tv[u] = fma(-X[u], qv[u], 1.0);
                    }
                }
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    const double wph = wmuh(m) * phm[u];
                    G2[u] = fma(wph, Ie[u], G2[u]);
                    G1[u] = fma(wph, tv[u], G1[u]);
                }
            }
        }
        if constexpr (NPT >= 1) {
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const double A = G2[u], B = G1[u];
                G1[u] = fma(Uc[u], B, Vc[u] * A);                   // :677
                G2[u] = cB[u] * A;                                  // :680
            }
        }
        Pacc *= 4.0 * kPi;
        if constexpr (LK) {
#pragma unroll
            for (int u = 0; u < NL; ++u) PP[u] *= 4.0 * kPi;
        }
        // (stores without lane masks: see the linear step)
        if (LK || nF > 0) at(psibar, kt) = Pacc;
        if constexpr (LK) {
#pragma unroll
            for (int u = 0; u < NL; ++u) at(ppsum, (unsigned)(u * plane) * 8u + o_pp + (unsigned)(km * LW) * 8u) = PP[u];
        }
        if constexpr (NPT >= 1) {
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const double wt = wlam[u] * w3k[u];                             // lines: x wphi (rh_method.py:451)
                red[(2 * u) * RROW + lane] = wt * G1[u];
                red[(2 * u + 1) * RROW + lane] = wt * G2[u];
            }
        }
        if constexpr (PH == 0) {
            at(Jnew, kt) = live_col ? Jacc : jd_c;
        } else if constexpr (PH == 1) {
            lds_f64* const xwg = etab + LSX_EXP_TAB + 2 * (NV + 1) * RROW;
            xwg[dir * LSX_WAVE + lane] = Jacc;
            __syncthreads();
            if (dir == 0 && valid) {
                const double Jv = Jacc + xwg[LSX_WAVE + lane];
                at(Jnew, kt) = live_col ? Jv : jd_c;
                if (live_col) dJ = nanmax(dJ, fabs(1.0 - jd_c * rcp(Jv)));      // :705
            }
        } else {
            const double Jv = jhalf + Jacc;
            at(Jnew, kt) = live_col ? Jv : jd_c;
            if (act) dJ = nanmax(dJ, fabs(1.0 - jd_c * rcp(Jv)));               // :705
        }
    };
    // the three moments w_n = int_0^dtau t^n e^-t dt of the upwind intervals (lsx_dev.h, w3: the same regimes, the same exponential, ONE
    // series and the downward recurrence).  Ahead of the ray loop, for the five rays together: e^-dtau and -- where some lane of the
    // wavefront is below dtau = 0.25 -- the series of the second moment; `moments` forms the weights of one ray from the two when its
    // turn comes (both forms and three selects: a ray's weights never wait in registers for the other rays' turns).
    auto exps = [&](double (&ev)[NR], double (&s2)[NR]) __attribute__((always_inline)) {
        unsigned long long m_nl = 0, m_small = 0;
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            m_small |= __builtin_amdgcn_fcmp(dtau_u[m], 0.25, 4 /* ordered < */);
            m_nl |= __builtin_amdgcn_fcmp(dtau_u[m], 50.0, 13 /* unordered or <= */);
        }
        if (m_nl != 0) {
            // (saturated lanes need no select: for dtau > 50 the closed forms give exactly (1, 1, 2))
            double r[NR], th[NR], tl[NR];
            int ki[NR];
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                const double dc = min_noquiet(dtau_u[m], 700.0);
                const double kf = __builtin_rint(-dc * 0x1.71547652b82fep+6);       // 64 / ln2
                r[m] = fma(kf, -0x1.62e42fef80000p-7, -dc);
                r[m] = fma(kf, -0x1.1cf79abc9e3b4p-42, r[m]);
                ki[m] = (int)kf;
            }
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                const lds_f64* e = etab + 2 * (ki[m] & 63);
                th[m] = e[0];
                tl[m] = e[1];
            }
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                double t = fma3s(r[m], 1.0 / 720.0, 1.0 / 120.0);
                t = fma3(r[m], t, 1.0 / 24.0);
                t = fma3(r[m], t, 1.0 / 6.0);
                t = fma(r[m], t, 0.5);
                r[m] = fma(r[m] * r[m], t, r[m]);                                       // exp(r) - 1
            }
#pragma unroll
            for (int m = 0; m < NR; ++m) ev[m] = ldexp(fma(th[m], r[m], tl[m]) + th[m], ki[m] >> 6);
        } else {
#pragma unroll
            for (int m = 0; m < NR; ++m) ev[m] = 0.0;
        }
        if (m_small != 0) {
            // sum_n (-1)^n x^(n+3) / (n! (n + 3)), twelve terms (x < 0.25: the next one is below 1e-19 of the first); the coefficients
            // as scalar operands: each is used by the five rays in turn
            double t[NR];
#pragma unroll
            for (int m = 0; m < NR; ++m) t[m] = fma3c(dtau_u[m], -1.0 / 558835200.0, 1.0 / 47174400.0);
#define LSX_HORNER(C) _Pragma("unroll") for (int m = 0; m < NR; ++m) t[m] = fma3c(dtau_u[m], t[m], C);
            LSX_HORNER(-1.0 / 4354560.0) LSX_HORNER(1.0 / 443520.0) LSX_HORNER(-1.0 / 50400.0) LSX_HORNER(1.0 / 6480.0) LSX_HORNER(-1.0 / 960.0)
            LSX_HORNER(1.0 / 168.0) LSX_HORNER(-1.0 / 36.0) LSX_HORNER(1.0 / 10.0) LSX_HORNER(-1.0 / 4.0) LSX_HORNER(1.0 / 3.0)
#undef LSX_HORNER
#pragma unroll
            for (int m = 0; m < NR; ++m) s2[m] = (dtau_u[m] * dtau_u[m]) * (dtau_u[m] * t[m]);
        } else {
#pragma unroll
            for (int m = 0; m < NR; ++m) s2[m] = 0.0;
        }
    };
    auto moments = [&](const double x, const double e, const double s2, double& w0, double& w1, double& w2q) __attribute__((always_inline)) {
        // (the closed forms exactly as lsx_dev.h, w3, and the oracle form them: `I_u (1 - w0)` is the attenuated upwind intensity, and at
        // large dtau it inherits the rounding of w0 = 1 - e -- measured: w0 formed as w1 + dtau e instead, equal to 1e-16, moves J by
        // 1.5e-8 (CaII) / 1.5e-3 (Ca+H) of its value at far-UV wavelengths where that term is all there is)
        const double dc = min_noquiet(x, 700.0);
        const double a0 = 1.0 - e, a1 = a0 - dc * e, a2 = 2.0 * a1 - (dc * dc) * e;
        const double s1 = 0.5 * fma(x * x, e, s2), s0 = fma(x, e, s1);
        const bool small = x < 0.25;
        w0 = small ? s0 : a0;
        w1 = small ? s1 : a1;
        w2q = small ? s2 : a2;
    };
    // step s >= 1: depth k = kS + dk s from `cur`, the request for depth k + 1 into `nxt`, point s - 1 finished
    auto pstep = [&](const int s, auto phase_c, auto flags_c, Ops& cur, Ops& nxt) __attribute__((always_inline)) {
        constexpr int PH = decltype(phase_c)::value;             // of the FINISHED point
        constexpr bool FIRSTPT = (decltype(flags_c)::value & 1) != 0, NOREQ = (decltype(flags_c)::value & 2) != 0;
        const int k = kS + dk * s, mpt = s - 1;
        if constexpr (PH == 2) {
            if (2 * mpt == Ns || 2 * mpt == Ns + 1) __syncthreads();   // the partner wave's first-half stores
        }
        double jhalf = 0.0;
        if constexpr (PH == 2) jhalf = at(Jnew, o_til + (unsigned)((k - dk) * LW) * 8u);
        if constexpr (!FIRSTPT && NPT >= 1) flush(mpt - 1);
        const lds_f64* tk = ring_row(s) + lc3;
        const double hdz = ring_row(s)[lcg];                         // the interval between points m and m + 1: row k (down) / k + 1 (up)
        const double etaB = cur.be + ring_row(s)[lcg + 1] * cur.jd, chiB = cur.bc, jd_k = cur.jd;
        double Xk[NS], njk[NS];
#pragma unroll
        for (int u = 0; u < NPT; ++u) { Xk[u] = tk[TU * u + 0]; njk[u] = tk[TU * u + 1]; }
        if constexpr (!NOREQ) load_ops(k + dk, nxt);
        ring_step(s);
        double ev[NR], s2[NR];
        if constexpr (!FIRSTPT) exps(ev, s2);
        point(mpt, phase_c, std::false_type{}, jhalf, [&](const int m, double& I, double& Psi, const double (&)[NS]) __attribute__((always_inline)) {
            // depth k: opacity, source function, the optical depth of the interval (m, m + 1) and its reciprocal (one v_rcp for both divisions)
            double chiTot = chiB, etaTot = etaB;
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                chiTot = fma(Xk[u], cur.ph[u][m], chiTot);
                etaTot = fma(njk[u], cur.ph[u][m], etaTot);
            }
            const double dtd = (chi_c(m) + chiTot) * (hdz * zmu(m));
            const double rcd = rcp(chiTot * dtd);
            const double rchi_k = rcd * dtd, rd = rcd * chiTot;
            const double S_k = etaTot * rchi_k;
            const double q = (S_c[m] - S_k) * rd;
            if constexpr (FIRSTPT) {
                I = Iu[m];
                Psi = 0.0;
            } else {
                // (lsx_dev.h, parabolic_point_fast: the same expressions; the upwind quotient is the previous point's downwind one)
                const double pp = pu[m];
                const double sum = dtau_u[m] + dtd, cu = sum + dtd, cd = sum + dtau_u[m];
                const double rden = rcp(cu * q + cd * pp);
                const double t3 = (3.0 * sum) * rden;
                double a = (pp * q) * t3;
                double dadS = ((cd * pp) * pp * rd - (cu * q) * q * ru[m]) * (t3 * rden);
                const bool big = fabs(a) > 2.0 * fabs(pp);
                a = big ? 2.0 * pp : a;
                dadS = big ? -2.0 * ru[m] : dadS;
                const bool mono = pp * q > 0.0;
                a = mono ? a : 0.0;
                dadS = mono ? dadS : 0.0;
                const double b = (pp - a) * ru[m];
                double w0, w1, w2q;
                moments(dtau_u[m], ev[m], s2[m], w0, w1, w2q);
                I = Iu[m] * (1.0 - w0) + w0 * S_c[m] + w1 * a + w2q * b;
                const double Lam = w0 + (w1 - w2q * ru[m]) * dadS - (w2q * ru[m]) * ru[m];
                Psi = Lam * rchi_c(m);
            }
            // depth k becomes point m + 1
            Iu[m] = I;
            pu[m] = q;
            S_c[m] = S_k;
            set_chi(m, chiTot);
            set_rchi(m, rchi_k);
            dtau_u[m] = dtd;
            ru[m] = rd;
#pragma unroll
            for (int u = 0; u < NL; ++u) set_ph(u, m, cur.ph[u][m]);
        });
        jd_c = jd_k;
    };
    {
        typedef std::integral_constant<int, 0> F0;
        typedef std::integral_constant<int, 1> F1;
        typedef std::integral_constant<int, 2> F2;
        auto one = [&](int s, auto ph, auto fl) __attribute__((always_inline)) { if (s & 1) pstep(s, ph, fl, opB, opA); else pstep(s, ph, fl, opA, opB); };
        auto run = [&](int s0, int s1, auto ph) __attribute__((always_inline)) {           // steps [s0, s1) of one phase
            int s = s0;
            if (s < s1 && (s & 1)) { pstep(s, ph, F0{}, opB, opA); ++s; }
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): no load is pending on any path into the loop
            for (; s + 1 < s1; s += 2) { pstep(s, ph, F0{}, opA, opB); pstep(s + 1, ph, F0{}, opB, opA); }
            if (s < s1) pstep(s, ph, F0{}, opA, opB);
        };
        // point m is finished in step m + 1: m < nA first visitor, m = nA (odd Nspace) the midpoint, then second visitor
        const int nA = Ns / 2;
        ring_step(0);                                                                // (there is no step 0 here: the ring moves on by its row)
        one(1, std::integral_constant<int, 0>{}, F1{});                              // point 0: the boundary value (depth 1 is already requested... and 2 is now)
        if (Ns == 3) {
            one(2, std::integral_constant<int, 1>{}, F2{});                          // three depths: the midpoint's step is the last one, nothing left to request
        } else {
            run(2, nA + 1, std::integral_constant<int, 0>{});
            if (Ns & 1) one(nA + 1, std::integral_constant<int, 1>{}, F0{});
            run(nA + 1 + (Ns & 1), Ns - 1, std::integral_constant<int, 2>{});
            one(Ns - 1, std::integral_constant<int, 2>{}, F2{});                     // finishes point Ns - 2; nothing left to request
        }
        // the end point: the linear rule with its own weights (no downwind neighbour)
        {
            const int mpt = Ns - 1;
            if (2 * mpt == Ns || 2 * mpt == Ns + 1) __syncthreads();
            const double jhalf = at(Jnew, o_til + (unsigned)((kS + dk * mpt) * LW) * 8u);
            if constexpr (NPT >= 1) flush(mpt - 1);
            double ev[NR], s2[NR];
            exps(ev, s2);
            point(mpt, std::integral_constant<int, 2>{}, std::true_type{}, jhalf, [&](const int m, double& I, double& Psi, const double (&)[NS]) __attribute__((always_inline)) {
                double w0, w1, w2q;
                moments(dtau_u[m], ev[m], s2[m], w0, w1, w2q);
                I = Iu[m] * (1.0 - w0) + w0 * S_c[m] + w1 * pu[m];
                Psi = (w0 - w1 * ru[m]) * rchi_c(m);
            });
        }
        if constexpr (NPT >= 1) {
            __builtin_amdgcn_wave_barrier();
            flush(Ns - 1);
        }
    }
    } else {
    // `cur` holds the operands of this step's depth (requested one step ago), the next depth's are requested into `nxt`: the
    // two buffers swap roles from step to step (even steps: cur = opA), so nothing is copied
    auto step = [&](const int s, auto phase_c, Ops& cur, Ops& nxt) __attribute__((always_inline)) {
        constexpr int PHX = decltype(phase_c)::value;        // 0 first visitor, 1 midpoint, 2 second visitor, 3 the end point, 4 the first point
        constexpr bool FIRST = PHX == 4;
        constexpr int PH = FIRST ? 0 : PHX;
        constexpr bool SECOND = PH >= 2, LAST = PH == 3;
        const int k = kS + dk * s;
        const unsigned kt = o_til + (unsigned)(k * LW) * 8u;
        if constexpr (SECOND) {
            if (2 * s == Ns || 2 * s == Ns + 1) __syncthreads();   // the partner wave's first-half stores
        }
        double jhalf = 0.0;
        if constexpr (SECOND) jhalf = at(Jnew, kt);
        // EPI: the partner's halves of Psibar and the Psi* phi sums at this depth (requested with J's half, used at the end of the step)
        double phalf = 0.0, pph[NLK];
#pragma unroll
        for (int u = 0; u < NLK; ++u) pph[u] = 0.0;
        if constexpr (EPI && SECOND) {
            phalf = at(psibar_o, kt);
            if constexpr (LK) {
#pragma unroll
                for (int u = 0; u < NL; ++u) pph[u] = at(ppsum_o, (unsigned)(u * plane) * 8u + o_pp + (unsigned)(k * LW) * 8u);
            }
        }
        // the Gamma totals of the previous depth (their values were parked at the end of the previous step)
        if constexpr (!FIRST && NPT >= 1) {
            flush(s - 1);
        }

        // ---- ray-independent part (rh_method.py:601-632): continuum slots, emissivity without the lines
        const lds_f64* tk = ring_row(s) + lc3;                             // this depth's row, the lane's column
        const double hdz = ring_row(s)[lcg];                         // the interval behind this ray: row k (down) / k + 1 (up)
        double etaB = cur.be + ring_row(s)[lcg + 1] * cur.jd;
        double chiB = cur.bc;
        const double Ek = E_of(s, cur.E);
        fast_fold(s, Ek, u_la, chiB, etaB);
        double X[NS], Vjc[NS], Ujc[NS], chic[NS], njUc[NS], njc[NS], w3k[NS];   // lines: X = cB (n_i - g n_j), n_j Uc, wphi; continua: Vji, Uji, chi, n_j
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            w3k[u] = tk[TU * u + 2];
            if (u < NL) {
                X[u] = tk[TU * u + 0];
                njUc[u] = tk[TU * u + 1];
                Vjc[u] = Ujc[u] = chic[u] = njc[u] = 0.0;
            } else {
                const bool a = (pact >> u) & 1u;
                X[u] = njUc[u] = 0.0;
                njc[u] = tk[TU * u + 1];
                Vjc[u] = a ? (w3k[u] * Ek) * alv[u] : 0.0;                // g_ij alpha, :284-285, :453
                Ujc[u] = u_la * Vjc[u];                                   // :286
                chic[u] = tk[TU * u + 0] * alv[u] - njc[u] * Vjc[u];
                chiB += chic[u];
                etaB = fma(njc[u], Ujc[u], etaB);
            }
        }

        // the next depth's operands are requested HERE, after the ray-independent part has consumed this depth's background,
        // populations and geometry: their registers are free again, so the two operand sets overlap only in the profiles
        // (the first point's neighbour was loaded for the boundary condition)
        if constexpr (PF2) {
            // two depths of prefetch: the request goes two steps ahead into the buffer the PREVIOUS step consumed (three operand sets
            // rotate, nothing is copied); the last-but-one step repeats the end point's request (a valid address, never used), so
            // every step of a phase issues the same loads and the compiler's vmcnt counts stay exact
            if constexpr (!LAST) load_ops(s + 2 < Ns ? k + 2 * dk : kS + dk * (Ns - 1), nxt);
        } else {
            if constexpr (!LAST && !FIRST) load_ops(k + dk, nxt);
        }
        ring_step(s);       // the operand ring moves on by one row (one more load per step, the same in every step)

        // ---- the five rays of this wavelength, in three straight-line passes so that the five independent chains interleave
        // (a branch per ray -- the skipped exponential, the skipped series of w2 -- would cut the instruction stream into
        // blocks the scheduler cannot mix; the regime tests of w2 are taken for the five rays together instead)
        // pass A: opacity and the optical depth of the interval behind each ray (formal_solver.py:107-129).  What the later
        // passes need of it IS the recurrence's state (chi_prev, dtau_prev now hold this depth's values): nothing else is kept
        double dt[NR], w0[NR], w1[NR];
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            double chiTot = chiB;
#pragma unroll
            for (int u = 0; u < NL; ++u) chiTot = fma(X[u], cur.ph[u][m], chiTot);       // n_i Vij - n_j Vji, :613
            if constexpr (FIRST) {
                dt[m] = 1.0;
            } else {
                const double dtau = (chi_prev[m] + chiTot) * (hdz * zmu(m));
                // formal_solver.py:138-139: the end point re-uses the PREVIOUS interval's w (and S[kEnd - dk]) with the fresh dS, dtau
                dt[m] = LAST ? dtau_prev[m] : dtau;
                dtau_prev[m] = dtau;
            }
            chi_prev[m] = chiTot;
        }
        // pass B: w2 (formal_solver.py:14-44) of the five rays; the exponential and the series each behind ONE wave-uniform test
        if constexpr (!FIRST) {
            unsigned long long m_mid = 0, m_small = 0;
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                m_small |= __builtin_amdgcn_fcmp(dt[m], 5e-4, 4 /* ordered < */);
                // two compares straight into lane masks (the ballot of a boolean expression materialises it first: five vector
                // instructions per ray); unordered-or predicates: a NaN takes the exponential's path, as !(dt < 5e-4 || dt > 50) did
                m_mid |= __builtin_amdgcn_fcmp(dt[m], 5e-4, 11 /* unordered or >= */) & __builtin_amdgcn_fcmp(dt[m], 50.0, 13 /* unordered or <= */);
            }
            if (m_mid != 0) {
                // (saturated lanes need no select: for dtau > 50 the formulae give exactly (1, 1); lsx_dev.h, w2)
                // exp(-dtau) of the five rays, staged by hand (lsx_dev.h, exp_tab64): all five table reads are in flight
                // before the first polynomial starts
                double dc[NR], r[NR], th[NR], tl[NR];
                int ki[NR];
#pragma unroll
                for (int m = 0; m < NR; ++m) {
                    dc[m] = min_noquiet(dt[m], 700.0);
                    const double kf = __builtin_rint(-dc[m] * 0x1.71547652b82fep+6);       // 64 / ln2
                    r[m] = fma(kf, -0x1.62e42fef80000p-7, -dc[m]);
                    r[m] = fma(kf, -0x1.1cf79abc9e3b4p-42, r[m]);
                    ki[m] = (int)kf;
                }
#pragma unroll
                for (int m = 0; m < NR; ++m) {
                    const lds_f64* e = etab + 2 * (ki[m] & 63);
                    th[m] = e[0];
                    tl[m] = e[1];
                }
#pragma unroll
                for (int m = 0; m < NR; ++m) {
                    // (the r^6 / 720 term is only a third of an ulp of exp(-dtau), but just above the Taylor switch of w2, dtau = 5e-4,
                    // w1 = (1 - e) - dtau e cancels to dtau^2 / 2: leaving it out was measured as 2.5e-10 on single rays' emergent
                    // intensity against the oracle, with it 1.5e-10 -- it stays)
                    double t = fma3s(r[m], 1.0 / 720.0, 1.0 / 120.0);
                    t = fma3(r[m], t, 1.0 / 24.0);
                    t = fma3(r[m], t, 1.0 / 6.0);
                    t = fma(r[m], t, 0.5);
                    r[m] = fma(r[m] * r[m], t, r[m]);                                       // exp(r) - 1
                }
#pragma unroll
                for (int m = 0; m < NR; ++m) {
                    const double e = ldexp(fma(th[m], r[m], tl[m]) + th[m], ki[m] >> 6);
                    w0[m] = 1.0 - e;
                    w1[m] = w0[m] - dc[m] * e;
                }
            } else {
#pragma unroll
                for (int m = 0; m < NR; ++m) w0[m] = w1[m] = 1.0;
            }
            if (m_small != 0) {
#pragma unroll
                for (int m = 0; m < NR; ++m) {
                    const bool small = dt[m] < 5e-4;
                    const double t0 = dt[m] * (1.0 - 0.5 * dt[m]);
                    const double t1 = (dt[m] * dt[m]) * (0.5 - dt[m] * (1.0 / 3.0));
                    w0[m] = small ? t0 : w0[m];
                    w1[m] = small ? t1 : w1[m];
                    asm volatile("" : "+v"(w0[m]), "+v"(w1[m]));          // keeps the block a branch
                }
            }
        }
        // pass C: intensity, Psi*, the angle sums and the Gamma integrands (rh_method.py:638-681)
        double Jacc = 0.0, Pacc = 0.0, PP[NLK], G1[NS], G2[NS];
#pragma unroll
        for (int u = 0; u < NLK; ++u) PP[u] = 0.0;
#pragma unroll
        for (int u = 0; u < NS; ++u) G1[u] = G2[u] = 0.0;
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            const double chiTot = chi_prev[m];                              // (this depth's, pass A)
            double etaTot = etaB;
#pragma unroll
            for (int u = 0; u < NL; ++u) etaTot = fma(njUc[u], cur.ph[u][m], etaTot);     // n_j Uji, :281, :614
            double I, Lam, rchi, Sv;
            if constexpr (FIRST) {
                rchi = rcp(chiTot);
                Sv = etaTot * rchi;                                        // :632
                I = Iu[m];
                Lam = 0.0;
            } else {
                // the two divisions of a step (by chi, :632, and by dtau, :113/:121) share one reciprocal, 1 / (chi dtau)
                const double dtau = dtau_prev[m];
                const double rcd = rcp(chiTot * dtau);
                rchi = rcd * dtau;
                const double rdt = rcd * chiTot;
                Sv = etaTot * rchi;
                const double dS = (S_prev[m] - Sv) * rdt;
                const double Sx = LAST ? S_prev[m] : Sv;
                I = Iu[m] * (1.0 - w0[m]) + w0[m] * Sx + w1[m] * dS;
                Lam = w0[m] - w1[m] * rdt;
            }
            const double Psi = Lam * rchi;
            Iu[m] = I;
            S_prev[m] = Sv;
            if constexpr (LAST) {
                if (dir == 1 && act) p.Iout[((size_t)col * Nspect + la) * NR + m] = I;      // emergent intensity, :638
            }
            Jacc = fma(wmuh(m), I, Jacc);                                  // :640
            const double wP = wmuh(m) * Psi;                               // (x 4 pi where the sums leave)
            Pacc += wP;
            if constexpr (LK) {
#pragma unroll
                for (int u = 0; u < NL; ++u) PP[u] = fma(wP, cur.ph[u][m], PP[u]);
            }
            if constexpr (FACT) {
                // Line-only tiles whose lines share at most their lower level (one line; two lines with TOPO 1 / 2): every term
                // of rh_method.py:652, 677-681 carries the line's profile, atom.U[i] = 0 and atom.U[j] = Uji, so
                //   Gamma_ij integrand = phi [Uc (1 - Psi* chi_i) + Vc Ieff],   Gamma_ji integrand = cB phi Ieff
                // and only a = phi Ieff, b = phi (1 - Psi* chi_i) are summed over the rays (Uc, Vc, cB applied to the sums)
                double qv[NS];
#pragma unroll
                for (int u = 0; u < NPT; ++u) qv[u] = Psi * cur.ph[u][m];
                double Ie[NS], tv[NS];
                if constexpr (NPT == 2 && TOPO == 1) {              // same atom, common lower level: atom.eta, atom.chi[i] are the sums
                    const double Ic = fma(-njUc[1], qv[1], fma(-njUc[0], qv[0], I));
                    const double tc = fma(-X[1], qv[1], fma(-X[0], qv[0], 1.0));
                    Ie[0] = Ie[1] = Ic;
                    tv[0] = tv[1] = tc;
                } else {
#pragma unroll
                    for (int u = 0; u < NPT; ++u) {
                        Ie[u] = fma(-njUc[u], qv[u], I);            // Ieff = I - Psi* eta, :652
                        tv[u] = fma(-X[u], qv[u], 1.0);
                    }
                }
#pragma unroll
                for (int u = 0; u < NPT; ++u) {
                    // (+ the linked continua's share of atom.eta, atom.chi[i]: applied by the epilogue, see CORR above)
                    const double wph = wmuh(m) * cur.ph[u][m];
                    G2[u] = fma(wph, Ie[u], G2[u]);                 // sum w phi Ieff
                    G1[u] = fma(wph, tv[u], G1[u]);                 // sum w phi (1 - Psi* chi_i)
                }
            } else {
            // the level bookkeeping of :616-627 from the tile's (at most two) slots
            double chi[NS], Uji[NS], eta[NS];
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                if (u < NL) {
                    chi[u] = X[u] * cur.ph[u][m];
                    Uji[u] = Uc[u] * cur.ph[u][m];                         // :281
                    eta[u] = njUc[u] * cur.ph[u][m];                       // :614
                } else { chi[u] = chic[u]; Uji[u] = Ujc[u]; eta[u] = njc[u] * Ujc[u]; }
            }
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const bool line = u < NL;
                const double pv = line ? cur.ph[u][m] : 0.0;
                const double Vij = line ? cB[u] * pv : alv[u];
                const double Vji = line ? Vc[u] * pv : Vjc[u];
                double etaA = eta[u], chi_i = chi[u], chi_j = -chi[u], U_j = Uji[u], U_i = 0.0;
                if constexpr (NPT == 2) {
                    const int v = 1 - u;
                    if constexpr (TOPO == 1) { etaA += eta[v]; chi_i += chi[v]; }       // same atom, common lower level
                    else if constexpr (TOPO == 0) {
                        const auto* rel = slots[u].rel[0];
                        etaA = fma(rel[REL_EA], eta[v], etaA);
                        chi_i = fma(rel[REL_CI], chi[v], chi_i);
                        chi_j = fma(rel[REL_CJ], chi[v], chi_j);
                        U_j = fma(rel[REL_UJ], Uji[v], U_j);
                        U_i = rel[REL_UI] * Uji[v];
                    }
                }
                if constexpr (CORR) {
                    if (line) {
                        etaA += cur.cr[u < NLK ? u : 0][0];
                        chi_i += cur.cr[u < NLK ? u : 0][1];
                        if constexpr (NCR > 2) chi_j += cur.cr[u < NLK ? u : 0][2];
                    }
                }
                const double Ieff = I - Psi * etaA;                            // :652
                double g1 = (Uji[u] + Vji * Ieff) - (chi_i * Psi) * U_j;        // :677
                double g2 = Vij * Ieff;                                        // :680
                if constexpr (NPT == 2 && TOPO == 0) g2 -= (chi_j * Psi) * U_i;
                G1[u] = fma(wmuh(m), g1, G1[u]);                               // w_mu / 2 (:661); 4 pi and the wavelength weight below
                G2[u] = fma(wmuh(m), g2, G2[u]);
            }
            }
        }
        if constexpr (FACT) {
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const double A = G2[u], B = G1[u];
                G1[u] = fma(Uc[u], B, Vc[u] * A);                   // :677
                G2[u] = cB[u] * A;                                  // :680
            }
        }
        Pacc *= 4.0 * kPi;
        if constexpr (LK) {
#pragma unroll
            for (int u = 0; u < NL; ++u) PP[u] *= 4.0 * kPi;
        }

        // ---- the wavelength's sums leave: Psibar, Psi* phi, the Gamma integrands (parked for the lane reduction), J
#ifndef LSX_ABL_NOSTORE
        // lanes without a wavelength of their own shadow a real lane and hold its values: they store the same bits to the same
        // address, so the store needs no lane mask (no branch around it: the compiler counts it when it places its waits -- loads and
        // stores retire in issue order, a wait for the next depth's operands waits for every store before them); the
        // Psibar of a frozen column is read by nobody (the fast-continuum kernels skip frozen columns)
        // (EPI: only the FIRST visitor of a depth stores them -- for the second visitor, which finishes the fast continua's rates itself)
        if constexpr (!EPI || PH == 0) { if (LK || nF > 0) st_once(psibar, kt, Pacc); }      // (a tile with linked continua has fast continua: no test at all in those instances)
#endif
        if constexpr (LK && (!EPI || PH == 0)) {       // (no lane mask either: shadow lanes repeat their lane's store, a frozen column's sums are read by nobody)
#pragma unroll
            for (int u = 0; u < NL; ++u) st_once(ppsum, (unsigned)(u * plane) * 8u + o_pp + (unsigned)(k * LW) * 8u, PP[u]);
        }
        if constexpr (NPT >= 1) {
            // lanes (c, j) -> element c * 12 + j of the value's row (= the lane number); idle lanes park zeros
#pragma unroll
            for (int u = 0; u < NPT; ++u) {
                const double wt = u < NL ? wlam[u] * w3k[u] : wlam[u];         // lines: x wphi (rh_method.py:451); continua :455
                red[(2 * u) * RROW + lane] = wt * G1[u];
                red[(2 * u + 1) * RROW + lane] = wt * G2[u];
            }
        }
        if constexpr (PH == 0) {
#ifdef LSX_ABL_NOSTORE
            if (valid && Jacc == 1.2345) at(Jnew, kt) = Jacc;
#else
            at(Jnew, kt) = live_col ? Jacc : cur.jd;                      // first visitor stores its half (frozen: J moves over; no lane mask, see Psibar)
#endif
        } else if constexpr (PH == 1) {                                   // odd Nspace: both waves are at the same depth
            lds_f64* const xwg = etab + LSX_EXP_TAB + 2 * (NV + 1) * RROW;
            xwg[dir * LSX_WAVE + lane] = Jacc;
            if constexpr (EPI) {                                          // ... and the down-going wave finishes the fast continua's rates there
                xw2[(dir * (1 + (LK ? NL : 0))) * LSX_WAVE + lane] = Pacc;
                if constexpr (LK) {
#pragma unroll
                    for (int u = 0; u < NL; ++u) xw2[(dir * (1 + NL) + 1 + u) * LSX_WAVE + lane] = PP[u];
                }
            }
            __syncthreads();
            if (dir == 0 && valid) {
                const double Jv = Jacc + xwg[LSX_WAVE + lane];
                at(Jnew, kt) = live_col ? Jv : cur.jd;
                if (live_col) dJ = nanmax(dJ, fabs(1.0 - cur.jd * rcp(Jv)));    // :705
            }
            if constexpr (EPI) {
                if (dir == 0) {
                    double PPt[NLK];
#pragma unroll
                    for (int u = 0; u < NLK; ++u) PPt[u] = (LK && u < NL) ? PP[u] + xw2[((1 + NL) + 1 + u) * LSX_WAVE + lane] : 0.0;
                    epi_fast(s, k, Jacc + xwg[LSX_WAVE + lane], Pacc + xw2[(1 + (LK ? NL : 0)) * LSX_WAVE + lane], PPt, Ek, X, njUc, w3k);
                }
            }
        } else {
            const double Jv = jhalf + Jacc;
            if constexpr (EPI) {
                double PPt[NLK];
#pragma unroll
                for (int u = 0; u < NLK; ++u) PPt[u] = pph[u] + ((LK && u < NL) ? PP[u] : 0.0);
                epi_fast(s, k, Jv, phalf + Pacc, PPt, Ek, X, njUc, w3k);
            }
#ifdef LSX_ABL_NOSTORE
            if (valid && Jv == 1.2345) at(Jnew, kt) = Jv;
#else
            st_once(Jnew, kt, live_col ? Jv : cur.jd);       // (the total: read again by the NEXT call only)
#endif
            if (act) dJ = nanmax(dJ, fabs(1.0 - cur.jd * rcp(Jv)));             // :705
        }
    };
    {
        // the operand buffers swap roles, two steps per loop trip -- where the two-step body fits the register file; the other
        // two-slot tiles take one step per trip and one copy of the operands per step
#if defined(LSX_RS_SWAP2)
        constexpr bool SWAP = true;
#elif defined(LSX_RS_NOSWAP)
        constexpr bool SWAP = false;
#else
        constexpr bool SWAP = NPT <= 1 || (!LK && TOPO != 0);      // (two lines with a known relation, no linked continua: 2 spilled registers)
#endif
        auto one = [&](int s, auto ph) __attribute__((always_inline)) {
            if constexpr (PF2) {
                const int r = s % 3;                 // step s reads set s mod 3 and requests depth s + 2 into set (s + 2) mod 3
                if (r == 0) step(s, ph, opA, opC); else if (r == 1) step(s, ph, opB, opA); else step(s, ph, opC, opB);
            }
            else if constexpr (SWAP) { if (s & 1) step(s, ph, opB, opA); else step(s, ph, opA, opB); }
            else { step(s, ph, opA, opB); opA = opB; }
        };
        auto run = [&](int s0, int s1, auto ph) __attribute__((always_inline)) {           // steps [s0, s1) of one phase
            int s = s0;
            if constexpr (PF2) {
                while (s < s1 && s % 3 != 0) { one(s, ph); ++s; }
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): no load is pending on any path into the loop
                for (; s + 2 < s1; s += 3) { step(s, ph, opA, opC); step(s + 1, ph, opB, opA); step(s + 2, ph, opC, opB); }
                for (; s < s1; ++s) one(s, ph);
            } else if constexpr (SWAP) {
                if (s < s1 && (s & 1)) { step(s, ph, opB, opA); ++s; }
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): no load is pending on any path into the loop
                for (; s + 1 < s1; s += 2) { step(s, ph, opA, opB); step(s + 1, ph, opB, opA); }
                if (s < s1) step(s, ph, opA, opB);
            } else {
                for (; s < s1; ++s) { step(s, ph, opA, opB); opA = opB; }
            }
        };
        const int nA = Ns / 2;
#ifdef LSX_CLOCK
        // diagnostic build: the shader clock against the 100 MHz clock around the depth loop (the production instruction stream in
        // between) -> p.debug, records as lsx_sweep.hip writes them (profiles/stamps.py): the clock the chip holds under this kernel
        unsigned long long tk0, tr0, tk1, tr1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk0), "=s"(tr0)::"memory");
#endif
        if constexpr (PF2) step(0, std::integral_constant<int, 4>{}, opA, opC);     // ... and requests depth 2 into the third set
        else step(0, std::integral_constant<int, 4>{}, opA, opB);     // the ray's first point (depth 1 is already requested: its load is skipped there)
        if constexpr (!SWAP && !PF2) opA = opB;
        run(1, nA, std::integral_constant<int, 0>{});
        if (Ns & 1) one(nA, std::integral_constant<int, 1>{});
        run(nA + (Ns & 1), Ns - 1, std::integral_constant<int, 2>{});
        one(Ns - 1, std::integral_constant<int, 3>{});
        if constexpr (NPT >= 1) {
            __builtin_amdgcn_wave_barrier();
            flush(Ns - 1);
        }
#ifdef LSX_CLOCK
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk1), "=s"(tr1)::"memory");
        if (lane == 0 && dir == 0 && p.debug && col0 % 100 == 5) {      // the column groups that start at column 5, 105, ...
            unsigned long long* D = (unsigned long long*)p.debug + (size_t)(((col0 / 100) * ntile + tile_id) & 1023) * 16;
            for (int i = 0; i < 8; ++i) D[i] = 0;
            D[7] = tk1 - tk0; D[1] = tr0; D[0] = tk0 - tk_in; D[2] = tk_s1 - tk_in; D[3] = tk_s2 - tk_in; D[4] = tk_s3 - tk_in; D[5] = tk_s4 - tk_in;        // (D[0]: entry -> first depth step, shader ticks: profiles/stamps.py)
            D[8] = tile_id; D[9] = NPT; D[10] = nF; D[11] = col0; D[12] = NL; D[13] = LK; D[14] = TOPO;
            D[15] = tr1 - tr0;
        }
#endif
    }
    }
    // dJ of every (column, tile, direction): the maximum over the column's wavelengths (NaN propagates, rh_method.py:706)
    __builtin_amdgcn_wave_barrier();
    red[NV * RROW + lane] = act ? dJ : 0.0;
    __builtin_amdgcn_wave_barrier();
    if (lane < ncg && (p.colmask ? p.colmask[col0 + lane] != 0 : true)) {
        double m = 0.0;
        for (int e = 0; e < nla; ++e) m = nanmax(m, red[NV * RROW + lane * LW + e]);
        p.dJpart[((size_t)(col0 + lane) * ntile + tile_id) * 2 + dir] = m;
    }
}

#undef zmu
#undef wmuh

template <int NPT, int NL, bool LK, int TOPO, bool PAR, bool FOLD, bool EPI>
static hipError_t launch_rs(const SweepParams& p, int ngroups, hipStream_t st)
{
    const dim3 g((unsigned)(ngroups * p.n_class_tiles)), b(2 * LSX_WAVE);
    hipLaunchKernelGGL((lsx_sweep_rs_kernel<NPT, NL, LK, TOPO, PAR, FOLD, EPI>), g, b,
                       lsx_rs_lds_doubles(NPT, p.Nspace, PAR, FOLD ? p.fold_nF : -1, EPI, LK ? NL : 0) * sizeof(double), st, p);
    return hipGetLastError();
}

// the folded or the plain instance of a class.  The two-line instances with a known relation and no linked continua sit at the
// register limit (folded they would spill): the plan never folds their classes (lsx_plan.cpp), and no folded instance of them exists.
#ifndef LSX_RS_PARABOLIC_TU
template <int NPT, int NL, bool LK, int TOPO>
static hipError_t launch_rs_pick(const SweepParams& p, int ngroups, hipStream_t st)
{
    constexpr bool kFoldable = lsx_rs_fold_instance_exists(NPT, LK, TOPO);
    if (p.fold) {
        if constexpr (kFoldable) {
            if (p.epi) return launch_rs<NPT, NL, LK, TOPO, false, true, true>(p, ngroups, st);
            return launch_rs<NPT, NL, LK, TOPO, false, true, false>(p, ngroups, st);
        } else return hipErrorNotSupported;
    }
    return launch_rs<NPT, NL, LK, TOPO, false, false, false>(p, ngroups, st);
}
#endif

// the ray-serial instance of a class (code = lsx_class_code of the class), NC columns per wavefront.  This file is compiled twice
// (Makefile): as it is for the reference's piecewise-linear rule, and with LSX_RS_PARABOLIC_TU for the parabolic instances (N4).
#ifndef LSX_RS_PARABOLIC_TU
extern "C" hipError_t lsx_launch_sweep_rs(const SweepParams* p, int code, hipStream_t st)
{
    if (p->Nrays != LSX_RS_RAYS || p->sca_per_lambda || p->L != LW) return hipErrorNotSupported;
    const int ngroups = (p->ncol + NC - 1) / NC;
    switch (code) {
#define LSX_X(NPT, NL, LK, TOPO) case lsx_class_code(NPT, NL, LK, TOPO): return launch_rs_pick<NPT, NL, LK, TOPO>(*p, ngroups, st);
        LSX_RS_INSTANCES(LSX_X)
#undef LSX_X
    default: return hipErrorNotSupported;
    }
}
#else
extern "C" hipError_t lsx_launch_sweep_rs_par(const SweepParams* p, int code, hipStream_t st)
{
    if (p->Nrays != LSX_RS_RAYS || p->sca_per_lambda || p->L != LW) return hipErrorNotSupported;
    const int ngroups = (p->ncol + NC - 1) / NC;
    switch (code) {
#define LSX_X(NPT, NL, LK, TOPO) case lsx_class_code(NPT, NL, LK, TOPO): return launch_rs<NPT, NL, LK, TOPO, true, false, false>(*p, ngroups, st);
        LSX_RSP_INSTANCES(LSX_X)
#undef LSX_X
    default: return hipErrorNotSupported;
    }
}
#endif
