// lsx_dev.h -- device-side data model shared by the host runtime (lsx_hip.cpp) and the
// kernels (lsx_kernels.hip, lsx_sweep.hip).  gfx950 / wave64 only.
//
// HBM layout (all float64, one block per context, index order left = slowest):
//   per column, "depth-major" (k-major) so that the 32 wavelength-lanes of a half-wave read
//   consecutive addresses at a fixed depth (the sweep is serial in k inside a lane):
//     bgchi_T, bgeta_T   [col][k][la]
//     J_T[2]             [col][k][la]            ping-pong: Jdag <- previous call
//     sca                [col][k]  (or [col][k][la] when sca_per_lambda)
//     phi_T              [col] { per line: [k][dir][mu][lt] }   (compact: [k][lt], mu/dir strides 0)
//     gijc_T             [col] { per continuum: [k][lt] }       g_ij of (26) in [U01], built at upload
//   per column, reference layout (level-major, read with half-wave-uniform addresses):
//     n, nStar           [col][NLtot][k]
//     C, Gamma           [col][NL2tot][k]
//     wphi               [col][Nlines][k]
//   outputs of the sweep:
//     Iout               [col][la][mu]
//     Gpart              [col][slot][e=ij,ji][dir][k]     one value per (tile-slot, depth, dir): no atomics
//     dJpart             [col][tile]
#pragma once
#include <stdint.h>

#define LSX_WAVE 64
#define LSX_HALF 32       // wavelengths per tile = lanes per direction
#define LSX_MAX_ATOMS 8
#define LSX_MAX_LEVELS 64 // NLtot cap for the lane-private LDS level arrays

struct DevTrans {           // one radiative transition, column independent
    int32_t atom, is_line;
    int32_t li, lj;         // global level index (lev_off[atom] + i / j)
    int32_t Nblue, Nlam;
    int32_t phi_off;        // lines: offset in lambda points into the column's SNl pool
    int32_t line_idx;       // lines: index among lines (wphi row)
    int32_t cont_off;       // continua: offset in lambda points into the SNc pool
    int32_t wl_off;         // offset into the wl / alpha per-(transition, lt) tables
    int32_t gam_ij, gam_ji; // element offsets of Gamma[i][j], Gamma[j][i] inside NL2tot
    double cB;              // lines: (hc/4pi) * Bij                     rh_method.py:268,279
    double gij;             // lines: Bji / Bij                          rh_method.py:450
    double AB;              // lines: Aji / Bji                          rh_method.py:281
};

struct DevTile {            // 32 consecutive wavelengths x both directions = one wavefront
    int32_t la0, nla;       // first global wavelength index, count (<= 32)
    int32_t nslot;          // transitions that are active somewhere in the tile
    int32_t slot0;          // first entry in the slot table / first Gpart slab
    int32_t nlev, lev0;     // levels those transitions touch: count, first entry in tile_levels
    int32_t natom_mask;     // bit a set: atom a has a slot in this tile
    int32_t pad;
};

struct SweepParams {
    // dimensions
    int32_t Nspace, Nrays, Nspect, Natoms, Ntrans, ncol;
    int32_t NLtot, NL2tot, Nlines;
    int32_t sca_per_lambda;
    int32_t phi_mu_stride_is_zero; // compact profile
    int32_t nslot_total, ntile_total;
    // column-independent tables
    const double* wavelength;   // [Nspect]
    const double* zmu;          // [M] 1/muz                (1 for padded rays)
    const double* wmuh;         // [M] 0.5*wmu              (0 for padded rays)   rh_method.py:661
    const double* wl;           // per (transition, lt): wlambda(lt) (lines) | wlambda(lt)/lambda/h (continua)
    const double* alpha;        // per (transition, lt) (continua; 0 for lines)
    const double* u_la;         // [Nspect] 2hc/lambda^3                  rh_method.py:286
    const uint8_t* active;      // [Ntrans][Nspect]
    const DevTrans* trans;
    const DevTile* tiles;
    const int32_t* tile_slots;  // transition id per slot
    const int32_t* tile_levels; // global level ids per tile
    const int32_t* class_tiles; // tile ids of the launched UMAX class
    int32_t n_class_tiles;
    int32_t pad0;
    // per-column strides (in doubles)
    int64_t phi_col_stride, gijc_col_stride;
    // per-column arrays
    const double* height;       // [col][k]
    const double* temperature;  // [col][k]
    const double* n;            // [col][NLtot][k]
    const double* wphi;         // [col][Nlines][k]
    const double* bgchi_T;
    const double* bgeta_T;
    const double* sca;
    const double* phi_T;
    const double* gijc_T;
    const double* Jdag_T;
    double* Jnew_T;
    double* Iout;
    double* Gpart;
    double* dJpart;
};
