// k_halo.h -- lateral boundary fills and periodic ghost copies on the GPU tile.
//
// Replaces, for closed/periodic edges (the LBC kinds of the BASELINE configs):
//   exchange_{p,r,u,v}2d_tile  ROMS/Nonlinear/exchange_2d.F:63-807
//   exchange_{r,u,v,w}3d_tile  ROMS/Nonlinear/exchange_3d.F:280-1126
//   bc_{r,u,v}2d_tile          ROMS/Nonlinear/bc_2d.F:41-516     bc_w3d_tile bc_3d.F:588
//   zetabc_tile zetabc.F:60 (closed :577-590), u2dbc_tile u2dbc_im.F:51, v2dbc_tile v2dbc_im.F:52,
//   t3dbc_tile t3dbc_im.F:50, u3dbc_tile u3dbc_im.F:50, v3dbc_tile v3dbc_im.F:50
//
// One thread block per horizontal plane; three barrier-separated phases
// (edge fills -> corner averages -> periodic copies), as the reference orders them.
#pragma once
#include "roms_ctx.h"


struct HaloItem {
  double *A;           // first plane
  int nk;              // number of consecutive planes
  int bc;              // BC_*
  int gtype;           // 'r','u','v','p' (exchange transverse ranges), 0 = no exchange
};
#define HALO_MAXITEMS 8
struct HaloArgs {
  DGrid G;
  int nitems;
  HaloItem it[HALO_MAXITEMS];
};

// the plane a block works on: item and plane of block bz, selected without indexing the
// kernel-argument array dynamically (a dynamic index makes the compiler copy the whole argument
// struct to scratch memory)
template <class ArgT>
KDEV double *halo_plane(const ArgT &a, int bz, int &bc, int &gtype) {
  double *A = nullptr;
  bc = BC_NONE; gtype = 0;
  int first = 0;
#pragma unroll
  for (int k = 0; k < HALO_MAXITEMS; k++) {
    const int nk = k < a.nitems ? a.it[k].nk : 0;
    if (bz >= first && bz < first + nk) {
      A = a.it[k].A + (size_t)(bz - first) * (size_t)a.G.nij;
      bc = a.it[k].bc;
      gtype = a.it[k].gtype;
    }
    first += nk;
  }
  return A;
}

// WET_DRY, barotropic state: the conditions at the end of zetabc.F:783-874 ("water level on boundary cells above bed
// elevation"), u2dbc_im.F:1190-1318 and v2dbc_im.F:1239-1367 AS WRITTEN, whatever the kind of boundary condition: the factor
// of the barotropic step from the wet mask and the value at one point, applied at that point -- except v2dbc's western edge,
// which takes mask and sign at Istr-1 and scales the value at Istr (v2dbc_im.F:1250-1255), and u2dbc's northern edge, whose
// loop starts at Istr (:1243).  By one thread block, behind the edge fills and corner values (halo_fill below, k_obc.h).
KDEV void wet_tail2(const DGrid &G, double *A, int bc) {
  const TB &B = G.T;
  const int Istr = B.Istr, Iend = B.Iend, Jstr = B.Jstr, Jend = B.Jend;
  if (bc == BC_R) {
    const double cff = G.Dcrit - 1.0E-20;
#define WDZ_(i_, j_) do { const size_t q_ = X2(i_, j_); if (A[q_] <= (G.Dcrit - G.hbath[q_])) A[q_] = cff - G.hbath[q_]; } while (0)
    if (!G.ewp) {
      if (B.west) KLOOP1(j, Jstr, Jend) WDZ_(Istr - 1, j);
      if (B.east) KLOOP1(j, Jstr, Jend) WDZ_(Iend + 1, j);
    }
    if (!G.nsp) {
      if (B.south) KLOOP1(i, Istr, Iend) WDZ_(i, Jstr - 1);
      if (B.north) KLOOP1(i, Istr, Iend) WDZ_(i, Jend + 1);
    }
    if (!(G.ewp || G.nsp) && KTID == 0) {
      if (B.sw) WDZ_(Istr - 1, Jstr - 1);
      if (B.se) WDZ_(Iend + 1, Jstr - 1);
      if (B.nw) WDZ_(Istr - 1, Jend + 1);
      if (B.ne) WDZ_(Iend + 1, Jend + 1);
    }
#undef WDZ_
  } else if (bc == BC_U || bc == BC_V) {
    const double *mw = bc == BC_U ? G.umask_wet : G.vmask_wet;
#define WDP_(mi, mj, ti, tj) do { const size_t m_ = X2(mi, mj), t_ = X2(ti, tj); A[t_] = A[t_] * wd_fac(mw[m_], A[m_]); } while (0)
    if (bc == BC_U) {
      if (!G.ewp) {
        if (B.west) KLOOP1(j, Jstr, Jend) WDP_(Istr, j, Istr, j);
        if (B.east) KLOOP1(j, Jstr, Jend) WDP_(Iend + 1, j, Iend + 1, j);
      }
      KSYNC();
      if (!G.nsp) {
        if (B.south) KLOOP1(i, B.IstrU, Iend) WDP_(i, Jstr - 1, i, Jstr - 1);
        if (B.north) KLOOP1(i, Istr, Iend) WDP_(i, Jend + 1, i, Jend + 1);
      }
      KSYNC();
      if (!(G.ewp || G.nsp) && KTID == 0) {
        if (B.sw) WDP_(Istr, Jstr - 1, Istr, Jstr - 1);
        if (B.se) WDP_(Iend + 1, Jstr - 1, Iend + 1, Jstr - 1);
        if (B.nw) WDP_(Istr, Jend + 1, Istr, Jend + 1);
        if (B.ne) WDP_(Iend + 1, Jend + 1, Iend + 1, Jend + 1);
      }
    } else {
      if (!G.ewp) {
        if (B.west) KLOOP1(j, B.JstrV, Jend) WDP_(Istr - 1, j, Istr, j);
        if (B.east) KLOOP1(j, B.JstrV, Jend) WDP_(Iend + 1, j, Iend + 1, j);
      }
      KSYNC();
      if (!G.nsp) {
        if (B.south) KLOOP1(i, Istr, Iend) WDP_(i, Jstr, i, Jstr);
        if (B.north) KLOOP1(i, Istr, Iend) WDP_(i, Jend + 1, i, Jend + 1);
      }
      KSYNC();
      if (!(G.ewp || G.nsp) && KTID == 0) {
        if (B.sw) WDP_(Istr - 1, Jstr, Istr - 1, Jstr);
        if (B.se) WDP_(Iend + 1, Jstr, Iend + 1, Jstr);
        if (B.nw) WDP_(Istr - 1, Jend + 1, Istr - 1, Jend + 1);
        if (B.ne) WDP_(Iend + 1, Jend + 1, Iend + 1, Jend + 1);
      }
    }
#undef WDP_
  }
  KSYNC();
}

// boundary fills and local periodic copies of one plane, by one thread block
KDEV void halo_fill(const DGrid &G, double *A, int bcf, int gtype) {
  const TB &B = G.T;
  const int Istr = B.Istr, Iend = B.Iend, Jstr = B.Jstr, Jend = B.Jend;
  const int Lm = G.Lm, Mm = G.Mm;
  const double gamma2 = G.gamma2;
  const int bc = bcf & BC_KIND;
  // MASKING: the value stored at a boundary point times the mask of THAT point (roms_ctx.h: BC_MASKF, BC_MASKALL)
  const bool mskr = G.masking && (bcf & BC_MASKF), msku = G.masking && bc == BC_U, mskv = G.masking && bc == BC_V;
  // ---- phase 1: edges of closed boundaries
  if (bc == BC_R && mskr) {
    if (!G.ewp) {
      if (B.west) KLOOP1(j, Jstr, Jend) A[X2(Istr - 1, j)] = A[X2(Istr, j)] * G.rmask[X2(Istr - 1, j)];
      if (B.east) KLOOP1(j, Jstr, Jend) A[X2(Iend + 1, j)] = A[X2(Iend, j)] * G.rmask[X2(Iend + 1, j)];
    }
    if (!G.nsp) {
      if (B.south) KLOOP1(i, Istr, Iend) A[X2(i, Jstr - 1)] = A[X2(i, Jstr)] * G.rmask[X2(i, Jstr - 1)];
      if (B.north) KLOOP1(i, Istr, Iend) A[X2(i, Jend + 1)] = A[X2(i, Jend)] * G.rmask[X2(i, Jend + 1)];
    }
  } else if (bc == BC_R) {
    if (!G.ewp) {
      if (B.west) KLOOP1(j, Jstr, Jend) A[X2(Istr - 1, j)] = A[X2(Istr, j)];
      if (B.east) KLOOP1(j, Jstr, Jend) A[X2(Iend + 1, j)] = A[X2(Iend, j)];
    }
    if (!G.nsp) {
      if (B.south) KLOOP1(i, Istr, Iend) A[X2(i, Jstr - 1)] = A[X2(i, Jstr)];
      if (B.north) KLOOP1(i, Istr, Iend) A[X2(i, Jend + 1)] = A[X2(i, Jend)];
    }
  } else if (bc == BC_U && (bcf & BC_LBC2D)) {
    // bc_u2d_tile with open edges (bc_2d.F:197-290): closed where LBC(edge,isUbar) is, else zero gradient over the
    // open condition's own range
    const int cl = (int)(G.lbc_closed >> (4 * ROMS_ISUBAR));
    if (!G.ewp) {
      if (B.east) { if (cl & (1 << ROMS_IEAST)) KLOOP1(j, Jstr, Jend) A[X2(Iend + 1, j)] = 0.0; else KLOOP1(j, Jstr, Jend) A[X2(Iend + 1, j)] = A[X2(Iend, j)]; }
      if (B.west) { if (cl & (1 << ROMS_IWEST)) KLOOP1(j, Jstr, Jend) A[X2(Istr, j)] = 0.0; else KLOOP1(j, Jstr, Jend) A[X2(Istr, j)] = A[X2(Istr + 1, j)]; }
    }
    KSYNC();
    if (!G.nsp) {
      const int Imin = G.ewp ? B.IstrU : B.Istr, Imax = G.ewp ? B.Iend : B.IendR;
      if (B.north) {
        if (cl & (1 << ROMS_INORTH)) KLOOP1(i, Imin, Imax) A[X2(i, Jend + 1)] = msku ? gamma2 * A[X2(i, Jend)] * G.umask[X2(i, Jend + 1)] : gamma2 * A[X2(i, Jend)];
        else KLOOP1(i, B.IstrU, Iend) A[X2(i, Jend + 1)] = A[X2(i, Jend)];
      }
      if (B.south) {
        if (cl & (1 << ROMS_ISOUTH)) KLOOP1(i, Imin, Imax) A[X2(i, Jstr - 1)] = msku ? gamma2 * A[X2(i, Jstr)] * G.umask[X2(i, Jstr - 1)] : gamma2 * A[X2(i, Jstr)];
        else KLOOP1(i, B.IstrU, Iend) A[X2(i, Jstr - 1)] = A[X2(i, Jstr)];
      }
    }
  } else if (bc == BC_V && (bcf & BC_LBC2D)) {
    const int cl = (int)(G.lbc_closed >> (4 * ROMS_ISVBAR));
    if (!G.ewp) {
      const int Jmin = G.nsp ? B.JstrV : B.Jstr, Jmax = G.nsp ? B.Jend : B.JendR;
      if (B.east) {
        if (cl & (1 << ROMS_IEAST)) KLOOP1(j, Jmin, Jmax) A[X2(Iend + 1, j)] = mskv ? gamma2 * A[X2(Iend, j)] * G.vmask[X2(Iend + 1, j)] : gamma2 * A[X2(Iend, j)];
        else KLOOP1(j, B.JstrV, Jend) A[X2(Iend + 1, j)] = A[X2(Iend, j)];
      }
      if (B.west) {
        if (cl & (1 << ROMS_IWEST)) KLOOP1(j, Jmin, Jmax) A[X2(Istr - 1, j)] = mskv ? gamma2 * A[X2(Istr, j)] * G.vmask[X2(Istr - 1, j)] : gamma2 * A[X2(Istr, j)];
        else KLOOP1(j, B.JstrV, Jend) A[X2(Istr - 1, j)] = A[X2(Istr, j)];
      }
    }
    KSYNC();
    if (!G.nsp) {
      if (B.north) { if (cl & (1 << ROMS_INORTH)) KLOOP1(i, Istr, Iend) A[X2(i, Jend + 1)] = 0.0; else KLOOP1(i, Istr, Iend) A[X2(i, Jend + 1)] = A[X2(i, Jend)]; }
      if (B.south) { if (cl & (1 << ROMS_ISOUTH)) KLOOP1(i, Istr, Iend) A[X2(i, Jstr)] = 0.0; else KLOOP1(i, Istr, Iend) A[X2(i, Jstr)] = A[X2(i, Jstr + 1)]; }
    }
  } else if (bc == BC_U) {
    if (!G.ewp) {
      if (B.west) KLOOP1(j, Jstr, Jend) A[X2(Istr, j)] = 0.0;
      if (B.east) KLOOP1(j, Jstr, Jend) A[X2(Iend + 1, j)] = 0.0;
    }
    KSYNC();
    if (!G.nsp) {
      const int Imin = G.ewp ? B.IstrU : B.Istr, Imax = G.ewp ? B.Iend : B.IendR;
      if (msku) {
        if (B.south) KLOOP1(i, Imin, Imax) A[X2(i, Jstr - 1)] = gamma2 * A[X2(i, Jstr)] * G.umask[X2(i, Jstr - 1)];
        if (B.north) KLOOP1(i, Imin, Imax) A[X2(i, Jend + 1)] = gamma2 * A[X2(i, Jend)] * G.umask[X2(i, Jend + 1)];
      } else {
        if (B.south) KLOOP1(i, Imin, Imax) A[X2(i, Jstr - 1)] = gamma2 * A[X2(i, Jstr)];
        if (B.north) KLOOP1(i, Imin, Imax) A[X2(i, Jend + 1)] = gamma2 * A[X2(i, Jend)];
      }
    }
  } else if (bc == BC_V) {
    if (!G.ewp) {
      const int Jmin = G.nsp ? B.JstrV : B.Jstr, Jmax = G.nsp ? B.Jend : B.JendR;
      if (mskv) {
        if (B.west) KLOOP1(j, Jmin, Jmax) A[X2(Istr - 1, j)] = gamma2 * A[X2(Istr, j)] * G.vmask[X2(Istr - 1, j)];
        if (B.east) KLOOP1(j, Jmin, Jmax) A[X2(Iend + 1, j)] = gamma2 * A[X2(Iend, j)] * G.vmask[X2(Iend + 1, j)];
      } else {
        if (B.west) KLOOP1(j, Jmin, Jmax) A[X2(Istr - 1, j)] = gamma2 * A[X2(Istr, j)];
        if (B.east) KLOOP1(j, Jmin, Jmax) A[X2(Iend + 1, j)] = gamma2 * A[X2(Iend, j)];
      }
    }
    KSYNC();
    if (!G.nsp) {
      if (B.south) KLOOP1(i, Istr, Iend) A[X2(i, Jstr)] = 0.0;
      if (B.north) KLOOP1(i, Istr, Iend) A[X2(i, Jend + 1)] = 0.0;
    }
  }
  KSYNC();
  // ---- WET_DRY, 3-D momentum: the slip value times the wet mask of the boundary point (u3dbc_im.F:523,681, v3dbc_im.F)
  if (G.wet_dry && (bcf & BC_WET3)) {
    if (bc == BC_U && !G.nsp) {
      const int Imin = G.ewp ? B.IstrU : B.Istr, Imax = G.ewp ? B.Iend : B.IendR;
      if (B.south) KLOOP1(i, Imin, Imax) A[X2(i, Jstr - 1)] = A[X2(i, Jstr - 1)] * G.umask_wet[X2(i, Jstr - 1)];
      if (B.north) KLOOP1(i, Imin, Imax) A[X2(i, Jend + 1)] = A[X2(i, Jend + 1)] * G.umask_wet[X2(i, Jend + 1)];
    } else if (bc == BC_V && !G.ewp) {
      const int Jmin = G.nsp ? B.JstrV : B.Jstr, Jmax = G.nsp ? B.Jend : B.JendR;
      if (B.west) KLOOP1(j, Jmin, Jmax) A[X2(Istr - 1, j)] = A[X2(Istr - 1, j)] * G.vmask_wet[X2(Istr - 1, j)];
      if (B.east) KLOOP1(j, Jmin, Jmax) A[X2(Iend + 1, j)] = A[X2(Iend + 1, j)] * G.vmask_wet[X2(Iend + 1, j)];
    }
    KSYNC();
  }
  // ---- phase 2: corners (only when neither direction is periodic)
  if (bc != BC_NONE && !(G.ewp || G.nsp) && KTID == 0) {
    if (bc == BC_R) {
      if (B.sw) A[X2(Istr - 1, Jstr - 1)] = 0.5 * (A[X2(Istr, Jstr - 1)] + A[X2(Istr - 1, Jstr)]);
      if (B.se) A[X2(Iend + 1, Jstr - 1)] = 0.5 * (A[X2(Iend, Jstr - 1)] + A[X2(Iend + 1, Jstr)]);
      if (B.nw) A[X2(Istr - 1, Jend + 1)] = 0.5 * (A[X2(Istr - 1, Jend)] + A[X2(Istr, Jend + 1)]);
      if (B.ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
    } else if (bc == BC_U) {
      if (B.sw) A[X2(Istr, Jstr - 1)] = 0.5 * (A[X2(Istr + 1, Jstr - 1)] + A[X2(Istr, Jstr)]);
      if (B.se) A[X2(Iend + 1, Jstr - 1)] = 0.5 * (A[X2(Iend, Jstr - 1)] + A[X2(Iend + 1, Jstr)]);
      if (B.nw) A[X2(Istr, Jend + 1)] = 0.5 * (A[X2(Istr, Jend)] + A[X2(Istr + 1, Jend + 1)]);
      if (B.ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
    } else {
      if (B.sw) A[X2(Istr - 1, Jstr)] = 0.5 * (A[X2(Istr, Jstr)] + A[X2(Istr - 1, Jstr + 1)]);
      if (B.se) A[X2(Iend + 1, Jstr)] = 0.5 * (A[X2(Iend, Jstr)] + A[X2(Iend + 1, Jstr + 1)]);
      if (B.nw) A[X2(Istr - 1, Jend + 1)] = 0.5 * (A[X2(Istr - 1, Jend)] + A[X2(Istr, Jend + 1)]);
      if (B.ne) A[X2(Iend + 1, Jend + 1)] = 0.5 * (A[X2(Iend + 1, Jend)] + A[X2(Iend, Jend + 1)]);
    }
  }
  KSYNC();
  if (G.wet_dry && (bcf & BC_WET2)) wet_tail2(G, A, bc);          // WET_DRY: the wetting/drying conditions of zetabc / u2dbc / v2dbc
  // ---- MASKING: the whole plane times rmask, boundary points included (step3d_t.F:1880-1890)
  if (G.masking && (bcf & BC_MASKALL)) {
    KLOOP2(i, j, B.IstrR, B.IendR, B.JstrR, B.JendR) A[X2(i, j)] = A[X2(i, j)] * G.rmask[X2(i, j)];
    KSYNC();
  }
  // ---- phase 3: periodic ghost copies (single tile in the periodic direction)
  if (gtype != 0 && (G.ewp || G.nsp)) {
    const int gt = gtype;
    int Jmin, Jmax, Imin, Imax;
    if (G.nsp) { Jmin = B.Jstr; Jmax = B.Jend; }
    else { Jmin = (gt == 'r' || gt == 'u') ? B.JstrR : B.Jstr; Jmax = B.JendR; }
    if (G.ewp) { Imin = B.Istr; Imax = B.Iend; }
    else { Imin = (gt == 'r' || gt == 'v') ? B.IstrR : B.Istr; Imax = B.IendR; }
    const int ng3 = G.Nghost == 3;
    if (G.ewp && B.west && B.east && G.xloc) {
      KLOOP1(j, Jmin, Jmax) {
        A[X2(Lm + 1, j)] = A[X2(1, j)];
        A[X2(Lm + 2, j)] = A[X2(2, j)];
        if (ng3) A[X2(Lm + 3, j)] = A[X2(3, j)];
        A[X2(-2, j)] = A[X2(Lm - 2, j)];
        A[X2(-1, j)] = A[X2(Lm - 1, j)];
        A[X2(0, j)] = A[X2(Lm, j)];
      }
    }
    if (G.nsp && B.south && B.north && G.yloc) {
      KLOOP1(i, Imin, Imax) {
        A[X2(i, Mm + 1)] = A[X2(i, 1)];
        A[X2(i, Mm + 2)] = A[X2(i, 2)];
        if (ng3) A[X2(i, Mm + 3)] = A[X2(i, 3)];
        A[X2(i, -2)] = A[X2(i, Mm - 2)];
        A[X2(i, -1)] = A[X2(i, Mm - 1)];
        A[X2(i, 0)] = A[X2(i, Mm)];
      }
    }
    if (G.ewp && G.nsp && B.sw && B.ne && G.xloc && G.yloc && KTID == 0) {
      const int ne = ng3 ? 3 : 2;
      for (int dj = 1; dj <= ne; dj++)
        for (int di = 1; di <= ne; di++) A[X2(Lm + di, Mm + dj)] = A[X2(di, dj)];
      for (int dj = 1; dj <= ne; dj++)
        for (int di = -2; di <= 0; di++) A[X2(di, Mm + dj)] = A[X2(Lm + di, dj)];
      for (int dj = -2; dj <= 0; dj++)
        for (int di = 1; di <= ne; di++) A[X2(Lm + di, dj)] = A[X2(di, Mm + dj)];
      for (int dj = -2; dj <= 0; dj++)
        for (int di = -2; di <= 0; di++) A[X2(di, dj)] = A[X2(Lm + di, Mm + dj)];
    }
  }
}

COOP_KERNEL(halo_kernel, HaloArgs) {
  (void)bx; (void)by; (void)lds;
  int bc, gtype;
  double *A = halo_plane(a, bz, bc, gtype);
  halo_fill(a.G, A, bc, gtype);
}
COOP_GLOBAL(halo_kernel, HaloArgs)

// Caller layout <-> library layout (roms_hip.cpp:relayout): plane gz of the caller's window, point (gx,gy) of it
struct RelayoutArgs {
  double *lib, *win;
  int to_lib, ni, cni, di, dj;
  long nij, cnij;
};
THREAD_KERNEL(k_relayout, RelayoutArgs) {
  const size_t w = (size_t)gx + (size_t)gy * (size_t)a.cni + (size_t)gz * (size_t)a.cnij;
  const size_t l = (size_t)(gx + a.di) + (size_t)(gy + a.dj) * (size_t)a.ni + (size_t)gz * (size_t)a.nij;
  if (a.to_lib) a.lib[l] = a.win[w];
  else a.win[w] = a.lib[l];
}
THREAD_GLOBAL(k_relayout, RelayoutArgs)

// ------------------------------------------------------------------------------------------
// Inter-tile halo strips (multi-GPU): the data movement of mp_exchange2d/3d/4d
// (ROMS/Utility/mp_exchange.F:28-2300) with the strip geometry of the periodic copies above --
// a tile's three west ghost columns {Istr-3..Istr-1} come from its west neighbour's last three
// interior columns {Iend-2..Iend}; its Nghost east ghost columns {Iend+1..Iend+Nghost} from the
// east neighbour's first Nghost interior columns; the same along eta.
//
// The reference moves full-height xi strips, then full-width eta strips which carry the corners
// (two dependent message phases).  Here ONE message phase gives the same ghost zone: full-height
// xi strips, full-width eta strips and the corner blocks from the four diagonal neighbours go out
// together; the receiver unpacks xi, then eta, then the corners, so the corner cells end up with
// the diagonal tile's interior values exactly as after the reference's second phase.  The
// boundary fills of the exchange point run in the pack launch (same block, same plane), so an
// exchange point costs two launches and one send/recv group.
//
// Directions d: 0 W, 1 E, 2 S, 3 N, 4 SW, 5 SE, 6 NW, 7 NE.
// Message layout: [plane][row][column] of the rectangle, columns fastest.
// ------------------------------------------------------------------------------------------
struct XchgArgs {
  DGrid G;
  int nitems;
  HaloItem it[HALO_MAXITEMS];
  int unpack;          // 0: (boundary fills and) pack own lines into send buffers, 1: unpack into ghosts
  int fill;            // pack launch: run halo_fill first
  double *buf[8];      // pack: message for neighbour d ; unpack: message received from neighbour d (null: none)
};

// rectangle of direction d: source lines of the message sent to neighbour d (unpack = 0) or ghost
// lines filled by the message from neighbour d (unpack = 1)
KDEV void xchg_rect(const DGrid &G, int d, int unpack, int &i0, int &i1, int &j0, int &j1) {
  const TB &B = G.T;
  const int gl = G.xgl, gh = G.xgh;       // ghost lines filled on the low | high side: 3 | Nghost, or B2D_GL | B2D_GH
  const int dx = (d == 0 || d == 4 || d == 6) ? -1 : ((d == 1 || d == 5 || d == 7) ? 1 : 0);
  const int dy = (d == 2 || d == 4 || d == 5) ? -1 : ((d == 3 || d == 6 || d == 7) ? 1 : 0);
  if (dx == 0) { i0 = G.LBi; i1 = G.LBi + G.ni - 1; }
  else if (dx < 0) { i0 = unpack ? B.Istr - gl : B.Istr; i1 = unpack ? B.Istr - 1 : B.Istr + gh - 1; }
  else { i0 = unpack ? B.Iend + 1 : B.Iend - gl + 1; i1 = unpack ? B.Iend + gh : B.Iend; }
  if (dy == 0) { j0 = G.LBj; j1 = G.LBj + G.nj - 1; }
  else if (dy < 0) { j0 = unpack ? B.Jstr - gl : B.Jstr; j1 = unpack ? B.Jstr - 1 : B.Jstr + gh - 1; }
  else { j0 = unpack ? B.Jend + 1 : B.Jend - gl + 1; j1 = unpack ? B.Jend + gh : B.Jend; }
}

KDEV void xchg_move(const DGrid &G, double *A, double *msg, int bz, int d, int unpack) {
  if (!msg) return;
  int i0, i1, j0, j1;
  xchg_rect(G, d, unpack, i0, i1, j0, j1);
  const int w = i1 - i0 + 1;
  double *m = msg + (size_t)bz * (size_t)w * (size_t)(j1 - j0 + 1);
  if (unpack) KLOOP2(i, j, i0, i1, j0, j1) A[X2(i, j)] = m[(j - j0) * w + (i - i0)];
  else KLOOP2(i, j, i0, i1, j0, j1) m[(j - j0) * w + (i - i0)] = A[X2(i, j)];
}

COOP_KERNEL(xchg_kernel, XchgArgs) {
  (void)bx; (void)by; (void)lds;
  const DGrid &G = a.G;
  int bc, gtype;
  double *A = halo_plane(a, bz, bc, gtype);
  if (!a.unpack) {
    if (a.fill) { halo_fill(G, A, bc, gtype); KSYNC(); }
#pragma unroll
    for (int d = 0; d < 8; d++) xchg_move(G, A, a.buf[d], bz, d, 0);
  } else {
    xchg_move(G, A, a.buf[0], bz, 0, 1);
    xchg_move(G, A, a.buf[1], bz, 1, 1);
    KSYNC();
    xchg_move(G, A, a.buf[2], bz, 2, 1);
    xchg_move(G, A, a.buf[3], bz, 3, 1);
    KSYNC();
#pragma unroll
    for (int d = 4; d < 8; d++) xchg_move(G, A, a.buf[d], bz, d, 1);
  }
}
COOP_GLOBAL(xchg_kernel, XchgArgs)

#ifndef ROMS_CPU_EMU
// ------------------------------------------------------------------------------------------
// Mailbox transport (roms_hip_comm_peer): the same pack / unpack bodies, with the neighbours'
// receive slots (mapped over xGMI) as the pack targets.  Ordering:
//   pack    one block per plane: strips -> the neighbours' slots, waits until its stores are acknowledged,
//           then writes this exchange's number into the plane's arrival word at each neighbour.
//   unpack  the first eight threads of a block poll the plane's arrival words until they have reached this
//           exchange's number, then the block copies its plane.
// Slots alternate with the parity of the exchange number: a neighbour can be at most one exchange
// ahead (its pack s+1 follows its unpack s, which needs my pack s), so the slot of exchange s is
// rewritten by exchange s+2 only, which the neighbour issues after it has seen my pack s+1 -- and that
// follows my unpack s in stream order.  The slab is uncached memory (what RCCL uses for its own
// peer buffers): stores land in the owner's HBM, its loads never hit a stale line.
// ------------------------------------------------------------------------------------------
struct PeerSync {
  unsigned long long seq;          // number of this exchange on its channel (1, 2, ...)
  unsigned long long *word[8];     // arrival words, one per plane -- pack: neighbour d's for my message; unpack: mine for neighbour d's
  unsigned long long *err;         // unpack: host word, set when a message did not arrive in time
  long long timeout;               // in wall_clock64 ticks (100 MHz)
};
struct XchgPeerArgs { XchgArgs x; PeerSync s; };

// Strips to / from a mailbox slot.  Slot accesses go past the caches in both directions -- stores with
// system scope (sc0 sc1: written through), loads from the uncached slab -- so that neither side needs a cache
// writeback or invalidate around the exchange (a __threadfence_system() per block made an exchange point 35 us:
// it writes the whole dirty L2 back, and the acquire empties it under the barotropic kernel).  A load from
// uncached memory is a full HBM round trip and gfx9 returns loads and stores through one in-order counter, so
// a thread first issues ALL its loads -- the strips of the eight directions of a plane taken as one list,
// PEER_U elements per thread -- and only then stores (dependent load/store pairs, 14 per thread: 15 us per launch).
#define PEER_U 8
struct PeerList { int n[8], i0[8], j0[8], w[8], tot; double *m[8]; };
template <int UNPACK>
KDEV void peer_list(const XchgArgs &xa, PeerList &L) {
  const DGrid &G = xa.G;
  L.tot = 0;
#pragma unroll
  for (int d = 0; d < 8; d++) {
    int i0, i1, j0, j1;
    xchg_rect(G, d, UNPACK, i0, i1, j0, j1);
    L.i0[d] = i0; L.j0[d] = j0; L.w[d] = i1 - i0 + 1;
    L.m[d] = xa.buf[d];
    L.n[d] = L.m[d] ? L.w[d] * (j1 - j0 + 1) : 0;
    L.tot += L.n[d];
  }
}
// element q of the list: direction, address in that direction's slot, index in the plane.  Static indices only
// (a run-time index into the tables sends them to scratch memory: 42 us per launch).
KDEV void peer_elem(const DGrid &G, const PeerList &L, int bz, int q, int &d, double *&m, int &x, int &ii, int &jj) {
  int r = q, w = 1, i0 = 0, j0 = 0, n = 0;
  d = 8; m = nullptr;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const bool here = d == 8 && r < L.n[e];
    if (here) { d = e; w = L.w[e]; i0 = L.i0[e]; j0 = L.j0[e]; n = L.n[e]; m = L.m[e]; }
    if (d == 8) r -= L.n[e];
  }
  const int j = r / w;
  ii = i0 + r - j * w; jj = j0 + j;
  x = (int)X2(ii, jj);
  m += (size_t)bz * (size_t)n + r;
}
// Unpacking without ordering.  The reference's xi phase followed by its eta phase -- here: xi strips, then eta
// strips, then corner blocks -- leaves in a cell the value of the LAST message that covers it; the rectangles
// overlap in the corner blocks only (a diagonal neighbour exists exactly where both adjacent ones do).  Each
// element is therefore written unless a later message covers its cell, and no barrier separates the three groups.
KDEV bool peer_covered_later(const DGrid &G, const PeerList &L, int d, int i, int j) {
  const TB &B = G.T;
  const int gl = G.xgl, gh = G.xgh;
  if (d < 2) return (L.m[2] && j >= B.Jstr - gl && j <= B.Jstr - 1) || (L.m[3] && j >= B.Jend + 1 && j <= B.Jend + gh);
  if (d < 4) {
    const bool low = i >= B.Istr - gl && i <= B.Istr - 1, high = i >= B.Iend + 1 && i <= B.Iend + gh;
    return d == 2 ? ((L.m[4] && low) || (L.m[5] && high)) : ((L.m[6] && low) || (L.m[7] && high));
  }
  return false;
}
// the general form (a plane's strips exceed PEER_U elements per thread): direction by direction
template <int UNPACK>
KDEV void peer_move(const DGrid &G, const PeerList &L, double *A, double *msg, int bz, int d) {
  if (!msg) return;
  int i0, i1, j0, j1;
  xchg_rect(G, d, UNPACK, i0, i1, j0, j1);
  const int w = i1 - i0 + 1, n = w * (j1 - j0 + 1), nt = (int)blockDim.x;
  double *m = msg + (size_t)bz * (size_t)n;
  for (int q0 = (int)threadIdx.x; q0 < n; q0 += 4 * nt) {
    double v[4];
    size_t x[4];
    bool keep[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int q = q0 + u * nt;
      if (q < n) {
        const int j = q / w;
        x[u] = X2(i0 + q - j * w, j0 + j);
        keep[u] = !(UNPACK && peer_covered_later(G, L, d, i0 + q - j * w, j0 + j));
        v[u] = UNPACK ? __builtin_nontemporal_load(&m[q]) : A[x[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int q = q0 + u * nt;
      if (q < n) {
        if (UNPACK) { if (keep[u]) A[x[u]] = v[u]; }
        else __hip_atomic_store(&m[q], v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

static __global__ void __launch_bounds__(1024) xchg_peer_pack(const XchgPeerArgs a) {
  const int tid = (int)threadIdx.x, bz = (int)blockIdx.z, nt = (int)blockDim.x;
  const DGrid &G = a.x.G;
  int bc, gtype;
  double *A = halo_plane(a.x, bz, bc, gtype);
  if (a.x.fill && (G.T.west || G.T.east || G.T.south || G.T.north)) { halo_fill(G, A, bc, gtype); __syncthreads(); }
  PeerList L;
  peer_list<0>(a.x, L);
  if (L.tot <= PEER_U * nt) {
    double v[PEER_U];
    double *mm[PEER_U];
#pragma unroll
    for (int u = 0; u < PEER_U; u++) {
      const int q = tid + u * nt;
      if (q < L.tot) { int d, x, i, j; peer_elem(G, L, bz, q, d, mm[u], x, i, j); v[u] = A[x]; }
    }
#pragma unroll
    for (int u = 0; u < PEER_U; u++) {
      const int q = tid + u * nt;
      if (q < L.tot) __hip_atomic_store(mm[u], v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  } else {
#pragma unroll
    for (int d = 0; d < 8; d++) peer_move<0>(G, L, A, a.x.buf[d], bz, d);
  }
  // every store of this block has been acknowledged before the plane's arrival words are written
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 8; d++)      // (static indices: a run-time index makes the compiler copy the arguments to scratch)
    if (tid == d && a.s.word[d]) __hip_atomic_store(a.s.word[d] + bz, a.s.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Pack and unpack of one exchange point in ONE launch (round 3): a block stores its plane's strips into the neighbours'
// slots and releases their arrival words, then waits for its own words and copies the received strips into the ghost
// zone.  Every rank's kernel packs before it waits, so the kernels of neighbouring ranks -- resident at the same time
// on their GPUs, one block per plane -- cannot wait for each other in a cycle.  Saves one launch (about 5 us of the
// 11 us of an exchange point on one MI355X).
struct XchgPeerBothArgs {
  XchgArgs x;                      // items; buf[d] = neighbour d's slot for my message (pack)
  double *ubuf[8];                 // my slot for neighbour d's message (unpack)
  PeerSync sp, su;                 // arrival words: the neighbours' (pack), mine (unpack)
};
static __global__ void __launch_bounds__(1024) xchg_peer_both(const XchgPeerBothArgs a) {
  const int tid = (int)threadIdx.x, bz = (int)blockIdx.z, nt = (int)blockDim.x;
  const DGrid &G = a.x.G;
  int bc, gtype;
  double *A = halo_plane(a.x, bz, bc, gtype);
  if (a.x.fill && (G.T.west || G.T.east || G.T.south || G.T.north)) { halo_fill(G, A, bc, gtype); __syncthreads(); }
  {
    PeerList L;
    peer_list<0>(a.x, L);
    if (L.tot <= PEER_U * nt) {
      double v[PEER_U];
      double *mm[PEER_U];
#pragma unroll
      for (int u = 0; u < PEER_U; u++) {
        const int q = tid + u * nt;
        if (q < L.tot) { int d, x, i, j; peer_elem(G, L, bz, q, d, mm[u], x, i, j); v[u] = A[x]; }
      }
#pragma unroll
      for (int u = 0; u < PEER_U; u++) {
        const int q = tid + u * nt;
        if (q < L.tot) __hip_atomic_store(mm[u], v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    } else {
#pragma unroll
      for (int d = 0; d < 8; d++) peer_move<0>(G, L, A, a.x.buf[d], bz, d);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned long long *word = nullptr;
#pragma unroll
  for (int d = 0; d < 8; d++) {
    if (tid == d && a.sp.word[d]) __hip_atomic_store(a.sp.word[d] + bz, a.sp.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid == d) word = a.su.word[d];
  }
  if (word) {
    const long long t0 = (long long)wall_clock64();
    while (__hip_atomic_load(word + bz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.su.seq) {
      __builtin_amdgcn_s_sleep(1);
      if ((long long)wall_clock64() - t0 > a.su.timeout) {
        __hip_atomic_store(a.su.err, a.su.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
  XchgArgs xu;
  xu.G = a.x.G; xu.nitems = a.x.nitems; xu.unpack = 1; xu.fill = 0;
#pragma unroll
  for (int k = 0; k < HALO_MAXITEMS; k++) xu.it[k] = a.x.it[k];
#pragma unroll
  for (int d = 0; d < 8; d++) xu.buf[d] = a.ubuf[d];
  PeerList L;
  peer_list<1>(xu, L);
  if (L.tot <= PEER_U * nt) {
    double v[PEER_U];
    int xx[PEER_U];
#pragma unroll
    for (int u = 0; u < PEER_U; u++) {
      const int q = tid + u * nt;
      xx[u] = -1;
      if (q < L.tot) {
        double *m;
        int d, x, i, j;
        peer_elem(G, L, bz, q, d, m, x, i, j);
        v[u] = __builtin_nontemporal_load(m);
        if (!peer_covered_later(G, L, d, i, j)) xx[u] = x;
      }
    }
#pragma unroll
    for (int u = 0; u < PEER_U; u++)
      if (xx[u] >= 0) A[xx[u]] = v[u];
    return;
  }
#pragma unroll
  for (int d = 0; d < 8; d++) peer_move<1>(G, L, A, xu.buf[d], bz, d);
}

static __global__ void __launch_bounds__(1024) xchg_peer_unpack(const XchgPeerArgs a) {
  const int tid = (int)threadIdx.x, bz = (int)blockIdx.z, nt = (int)blockDim.x;
  const DGrid &G = a.x.G;
  int bc, gtype;
  double *A = halo_plane(a.x, bz, bc, gtype);
  unsigned long long *word = nullptr;
#pragma unroll
  for (int d = 0; d < 8; d++)
    if (tid == d) word = a.s.word[d];
  if (word) {
    const long long t0 = (long long)wall_clock64();
    while (__hip_atomic_load(word + bz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.s.seq) {
      __builtin_amdgcn_s_sleep(1);
      if ((long long)wall_clock64() - t0 > a.s.timeout) {
        __hip_atomic_store(a.s.err, a.s.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
  PeerList L;
  peer_list<1>(a.x, L);
  if (L.tot <= PEER_U * nt) {
    double v[PEER_U];
    int xx[PEER_U];
#pragma unroll
    for (int u = 0; u < PEER_U; u++) {
      const int q = tid + u * nt;
      xx[u] = -1;
      if (q < L.tot) {
        double *m;
        int d, x, i, j;
        peer_elem(G, L, bz, q, d, m, x, i, j);
        v[u] = __builtin_nontemporal_load(m);
        if (!peer_covered_later(G, L, d, i, j)) xx[u] = x;
      }
    }
#pragma unroll
    for (int u = 0; u < PEER_U; u++)
      if (xx[u] >= 0) A[xx[u]] = v[u];
    return;
  }
#pragma unroll
  for (int d = 0; d < 8; d++) peer_move<1>(G, L, A, a.x.buf[d], bz, d);
}
#endif


// ---------------------------------------------------------------------------------------------- exchange soak (self-check)
// roms_hip_exchange_soak: a production-like stream of exchanges with NO host synchronisation in between -- fill the own
// points of `planes` work planes with a code of (global i, j, plane, repetition), exchange, count on the DEVICE the ghost
// points that do not hold the code of the point they image, fill for the next repetition ...  A slot read before the
// neighbour's strips of THIS repetition have landed, or a stale line, holds the previous repetition's code and is counted.
struct SoakArgs {
  DGrid G;
  double *A;
  unsigned long long *bad;        // bad[0]: mismatching ghost points so far; bad[1..3]: rep, packed (i,j), plane of the first
  int planes, rep, gl, gh;
  int nbr[8];
};
KDEV double soak_code(const DGrid &G, int i, int j, int k, int rep) {
  if (i < 1) i += G.Lm; else if (i > G.Lm) i -= G.Lm;
  if (j < 1) j += G.Mm; else if (j > G.Mm) j -= G.Mm;
  return ((double)(rep & 0xFFFF) * 64.0 + (double)k) * 67108864.0 + (double)j * 8192.0 + (double)i;
}
THREAD_KERNEL(k_soak_fill, SoakArgs) {           // index space: the whole array (ni x nj x planes)
  const DGrid &G = a.G;
  const int i = G.LBi + gx, j = G.LBj + gy;
  const TB &B = G.T;
  const bool own = i >= B.Istr && i <= B.Iend && j >= B.Jstr && j <= B.Jend;
  a.A[(size_t)gz * G.nij + X2(i, j)] = own ? soak_code(G, i, j, gz, a.rep) : -1.0;
}
THREAD_GLOBAL(k_soak_fill, SoakArgs)
THREAD_KERNEL(k_soak_check, SoakArgs) {          // index space: the whole array; ghost points that image a neighbour's own points
  const DGrid &G = a.G;
  const int i = G.LBi + gx, j = G.LBj + gy;
  const TB &B = G.T;
  const int dx = i < B.Istr ? -1 : (i > B.Iend ? 1 : 0), dy = j < B.Jstr ? -1 : (j > B.Jend ? 1 : 0);
  if (dx == 0 && dy == 0) return;
  if ((dx < 0 && i < B.Istr - a.gl) || (dx > 0 && i > B.Iend + a.gh) || (dy < 0 && j < B.Jstr - a.gl) || (dy > 0 && j > B.Jend + a.gh)) return;
  // direction index as in roms_hip.cpp: 0 W, 1 E, 2 S, 3 N, 4 SW, 5 SE, 6 NW, 7 NE
  const int d = dy == 0 ? (dx < 0 ? 0 : 1) : (dx == 0 ? (dy < 0 ? 2 : 3) : (dy < 0 ? (dx < 0 ? 4 : 5) : (dx < 0 ? 6 : 7)));
  if (a.nbr[d] < 0) return;
  const double v = a.A[(size_t)gz * G.nij + X2(i, j)];
  if (v != soak_code(G, i, j, gz, a.rep)) {
#ifdef ROMS_CPU_EMU
    if (a.bad[0]++ == 0) {
#else
    if (atomicAdd(a.bad, 1ull) == 0) {
#endif
      a.bad[1] = (unsigned long long)a.rep;
      a.bad[2] = ((unsigned long long)(unsigned)(i + 4096) << 32) | (unsigned)(j + 4096);
      a.bad[3] = (unsigned long long)gz;
    }
  }
}
THREAD_GLOBAL(k_soak_check, SoakArgs)
