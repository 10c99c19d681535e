// fv3_del2x.hip -- round 5: the three del2_cubed iterations of the damping heat and apply_diffusive_heating as ONE pass over the field.
//
// The staged form (fv3_nh.hip: fv3_del2_cubed + fv3_apply_diffusive_heating) is one launch per iteration, each a full read + write of the
// field (3 x 5.5 GB at 2.9 TB/s), and a fourth pass that reads the result back beside delp / delz / cappa / pt: 30 GB, 8.2 ms per call at
// C768 L79.  Here a workgroup owns a 64 x 8 tile of one plane and walks 16 levels: the tile with a three-cell frame goes to LDS once
// (70 x 14 values), iteration 1 is evaluated on the tile + 2 cells, iteration 2 on the tile + 1, iteration 3 on the tile -- each from the
// LDS copy of the one before -- and the heating of pt follows from the register that holds the result.  The smoothed field itself is not
// stored at all inside the sequencer (the next call zeroes it; `out` != null keeps it: the FV3_ALT that lets the heat accumulate over calls).
// The five metric terms of a cell sit in the registers of the thread that owns the cell through the level loop.  16 GB instead of 30.
//
// Values are the staged form's bit for bit: the same expressions in the same order (the library is built with -ffp-contract=off), the same
// update boxes (iteration n updates [1 - nt, n + nt]^2, nt = 3 - n; other cells keep their value), the same corner treatment: the three cells
// at a cube corner are averaged before every iteration, and the reads of an iteration with nt > 0 go through the two copy_corners remaps
// (cc_index).  Tiles with a cube corner in them run the GEN instantiation (six LDS offsets per cell instead of +-1 / +-pitch); the tile
// boundaries are placed so that everything an iteration reads or averages at a corner lies in ONE tile (d2_tile_range).
// CPU twin: oracle/fv3_oracle/nh.py (del2_cubed, apply_diffusive_heating).  [SURVEY A.12]
#include "fv3_ops.h"
#include "fv3_math.h"

namespace {

constexpr int D2_TI = 64, D2_TJ = 8;  // tile
constexpr int D2_P = 72, D2_R = 14;   // LDS frame: pitch, rows (tile + 3 either side)
constexpr int D2_KC = 16;             // levels a workgroup walks
constexpr int D2_NT = 256;            // threads
#ifdef FV3_HOST_EMU
constexpr int D2_S0 = (D2_TI + 6) * (D2_TJ + 6), D2_S1 = (D2_TI + 4) * (D2_TJ + 4), D2_S2 = (D2_TI + 2) * (D2_TJ + 2), D2_S3 = D2_TI * D2_TJ;  // one "thread" owns every cell
#else
constexpr int D2_S0 = ((D2_TI + 6) * (D2_TJ + 6) + D2_NT - 1) / D2_NT, D2_S1 = ((D2_TI + 4) * (D2_TJ + 4) + D2_NT - 1) / D2_NT,
              D2_S2 = ((D2_TI + 2) * (D2_TJ + 2) + D2_NT - 1) / D2_NT, D2_S3 = (D2_TI * D2_TJ + D2_NT - 1) / D2_NT;
#endif

// Tile m of width T along an axis with n cells (halo 3: cells -2 .. n + 3), first / last cell.  Natural boundaries every T cells from -2; a
// boundary that would fall within two cells of the far tile edge (last cell of the lower tile in [n - 2, n + 2]) moves to n - 5 | n - 4: the
// cells an iteration reads through a corner remap, and the three cells averaged at a corner, then all belong to the last tile and lie at least
// three cells inside it (or at the array edge), so its LDS copies of iterations 1 and 2 hold them.  (The near edge needs nothing: the second
// tile starts at cell 6 or later.)  Needs n >= 10.
FV3_HD inline void d2_tile_range(int m, int T, int n, int &F, int &L) {
  F = -2 + T * m;
  L = F + T - 1;
  if (L >= n - 2 && L <= n + 2) L = n - 5;
  if (F - 1 >= n - 2 && F - 1 <= n + 2) F = n - 4;
  if (L > n + 3) L = n + 3;
}
// cube corners inside tile (bx, by): bit 0 SW, 1 SE, 2 NE, 3 NW
FV3_HD inline int d2_corner_mask(int fl, int bx, int by, int nti, int ntj) {
  const bool W = fl & FV3_W, E = fl & FV3_E, S = fl & FV3_S, N = fl & FV3_N;
  int m = 0;
  if (W && S && bx == 0 && by == 0) m |= 1;
  if (E && S && bx == nti - 1 && by == 0) m |= 2;
  if (E && N && bx == nti - 1 && by == ntj - 1) m |= 4;
  if (W && N && bx == 0 && by == ntj - 1) m |= 8;
  return m;
}

struct D2Heat {
  const Real *delp, *delz, *cappa;
  Real *pt;
  Real rdg, cv_air, lim0;
  bool on;
};

template <bool GEN, bool HEAT>
void d2_launch(fv3_ctx *c, fv3_stream_t s, const Real *q, Real *out, Real cd, const D2Heat h) {
  const Geo g = c->g;
  const int nx = g.nx, ny = g.ny, nz1 = g.nz - 1;
  const int nti = (nx + 6 + D2_TI - 1) / D2_TI, ntj = (ny + 6 + D2_TJ - 1) / D2_TJ;
  const int nkc = (nz1 + D2_KC) / D2_KC;
  const int isd = 1 - g.nh, ied = nx + g.nh, jsd = 1 - g.nh, jed = ny + g.nh;
  const int sj32 = g.sj32, go = g.o;
  const size_t smem = sizeof(Real) * 2 * D2_R * D2_P;
  launch_blocks(c, s, GEN ? 4 : nti, GEN ? 1 : ntj, g.nsub * nkc, D2_NT, smem, [=] FV3_HD(const Blk &blk, char *smem_) {
    const int t = blk.bz / nkc, kch = blk.bz - t * nkc;
    const int k0 = kch * D2_KC, k1 = k0 + D2_KC - 1 < nz1 ? k0 + D2_KC - 1 : nz1;
    const int fl = g.flags[t];
    int bx, by, cmask;
    if (GEN) {  // block = corner number: its tile, once per tile, only where the sub-domain has that cube corner
      const int cn = blk.bx;
      bx = (cn == 1 || cn == 2) ? nti - 1 : 0;
      by = (cn == 2 || cn == 3) ? ntj - 1 : 0;
      cmask = d2_corner_mask(fl, bx, by, nti, ntj);
      if (!(cmask & (1 << cn))) return;
      for (int lo = 0; lo < cn; ++lo) {
        const int bx2 = (lo == 1 || lo == 2) ? nti - 1 : 0, by2 = (lo == 2 || lo == 3) ? ntj - 1 : 0;
        if (bx2 == bx && by2 == by && (cmask & (1 << lo))) return;  // (a lower-numbered corner of the same tile runs it)
      }
    } else {
      bx = blk.bx;
      by = blk.by;
      cmask = 0;
      if (d2_corner_mask(fl, bx, by, nti, ntj)) return;  // (the GEN launch's)
    }
    int Fi, Li, Fj, Lj;
    d2_tile_range(bx, D2_TI, nx, Fi, Li);
    d2_tile_range(by, D2_TJ, ny, Fj, Lj);
    const int tw = Li - Fi + 1, th = Lj - Fj + 1;
    if (tw <= 0 || th <= 0) return;
    const int oi = Fi - 3, oj = Fj - 3;  // cell at LDS frame position (0, 0)
    Real *const A = (Real *)smem_, *const B = A + D2_R * D2_P;
    const long m2 = t * g.st2;
    auto lds_of = [&](int i, int j) -> int { return (j - oj) * D2_P + (i - oi); };
    auto lds_of_ix = [&](unsigned off) -> int {  // in-plane offset (IX) -> LDS frame offset
      const int jj = (int)(off / (unsigned)sj32) - go, ii = (int)(off - (unsigned)(jj + go) * (unsigned)sj32) - go;
      return lds_of(ii, jj);
    };
    auto in_arr = [&](int i, int j) -> bool { return i >= isd && i <= ied && j >= jsd && j <= jed; };

    // ---- what a thread owns, fixed through the level loop
    struct Cell {                   // a cell of iteration n's region
      int lc;                       // LDS offset of the cell (-1: outside the array)
      int xw, xc, xe, ys, yc, yn;   // GEN: LDS offsets of the six points (plain tiles: lc -+ 1, lc -+ pitch)
      bool upd;
      Real mvx0, mvx1, muy0, muy1, cra;
    };
    unsigned g0[D2_S0];  // iteration 0 (the loaded frame): in-plane offset (0xffffffff: outside the array) ...
    int l0[D2_S0];       // ... and LDS offset
    Cell C1[D2_S1], C2[D2_S2], C3[D2_S3];
    unsigned p3[D2_S3];  // in-plane offset of the thread's tile cells
    bool dom3[D2_S3];    // ... inside the compute domain (the heating acts there)
    {
      const int w0 = tw + 6, n0 = w0 * (th + 6);
#pragma unroll
      for (int sl = 0; sl < D2_S0; ++sl) {
        const int e = blk.tid + sl * blk.nthr;
        g0[sl] = 0xffffffffu;
        l0[sl] = 0;
        if (e >= n0) continue;
        const int ej = e / w0, ei = e - ej * w0, i = oi + ei, j = oj + ej;
        if (!in_arr(i, j)) continue;
        g0[sl] = IX(i, j);
        l0[sl] = ej * D2_P + ei;
      }
    }
    auto setup = [&](auto &C, int rad, int nt) {
      constexpr int ns = (int)(sizeof(C) / sizeof(C[0]));
      const int w = tw + 2 * rad, n = w * (th + 2 * rad);
#pragma unroll
      for (int sl = 0; sl < ns; ++sl) {
        const int e = blk.tid + sl * blk.nthr;
        Cell &x = C[sl];
        x.lc = -1;
        x.upd = false;
        x.xw = x.xc = x.xe = x.ys = x.yc = x.yn = 0;
        x.mvx0 = x.mvx1 = x.muy0 = x.muy1 = x.cra = (Real)0;
        if (e >= n) continue;
        const int ej = e / w, ei = e - ej * w, i = Fi - rad + ei, j = Fj - rad + ej;
        if (!in_arr(i, j)) continue;
        x.lc = lds_of(i, j);
        x.upd = i >= 1 - nt && i <= nx + nt && j >= 1 - nt && j <= ny + nt;
        if (!x.upd) continue;
        const unsigned p0 = IX(i, j);
        x.mvx0 = (g.del6_v + m2)[p0];
        x.mvx1 = (g.del6_v + m2)[IX(i + 1, j)];
        x.muy0 = (g.del6_u + m2)[p0];
        x.muy1 = (g.del6_u + m2)[IX(i, j + 1)];
        x.cra = cd * (g.rarea + m2)[p0];
        if (!GEN) continue;  // (plain tiles form the four neighbours from lc)
        if (nt > 0) {
          x.xw = lds_of_ix(cc_index<1>(g, fl, i - 1, j));
          x.xc = lds_of_ix(cc_index<1>(g, fl, i, j));
          x.xe = lds_of_ix(cc_index<1>(g, fl, i + 1, j));
          x.ys = lds_of_ix(cc_index<2>(g, fl, i, j - 1));
          x.yc = lds_of_ix(cc_index<2>(g, fl, i, j));
          x.yn = lds_of_ix(cc_index<2>(g, fl, i, j + 1));
        } else {
          x.xw = x.lc - 1;
          x.xc = x.yc = x.lc;
          x.xe = x.lc + 1;
          x.ys = x.lc - D2_P;
          x.yn = x.lc + D2_P;
        }
      }
    };
    setup(C1, 2, 2);
    setup(C2, 1, 1);
    setup(C3, 0, 0);
    {
      const int n3 = tw * th;
#pragma unroll
      for (int sl = 0; sl < D2_S3; ++sl) {
        const int e = blk.tid + sl * blk.nthr;
        p3[sl] = 0;
        dom3[sl] = false;
        if (e >= n3) continue;
        const int ej = e / tw, ei = e - ej * tw, i = Fi + ei, j = Fj + ej;
        if (!in_arr(i, j)) continue;
        p3[sl] = IX(i, j);
        dom3[sl] = i >= 1 && i <= nx && j >= 1 && j <= ny;
      }
    }
    // one iteration on the cells of a region: reads P (the iteration before), writes the value to W (LDS) or hands it to `sink`
    auto iterate = [&](const auto &C, const Real *P, auto &&sink) {
      constexpr int ns = (int)(sizeof(C) / sizeof(C[0]));
#pragma unroll
      for (int sl = 0; sl < ns; ++sl) {
        const Cell &x = C[sl];
        if (x.lc < 0) continue;
        Real v = P[x.lc];
        if (x.upd) {
          const Real xc = GEN ? P[x.xc] : v, yc = GEN ? P[x.yc] : v;
          const Real xw = P[GEN ? x.xw : x.lc - 1], xe = P[GEN ? x.xe : x.lc + 1], ys = P[GEN ? x.ys : x.lc - D2_P], yn = P[GEN ? x.yn : x.lc + D2_P];
          v = v + x.cra * (x.mvx0 * (xw - xc) - x.mvx1 * (xc - xe) + x.muy0 * (ys - yc) - x.muy1 * (yc - yn));
        }
        sink(sl, x, v);
      }
    };
    // the three cells at each cube corner of the tile become their mean (before iterations 2 and 3 here; before iteration 1 in the field itself)
    auto fill = [&](Real *P) {
      if (!GEN) return;
      const Real r3 = (Real)(1.0 / 3.0);
      const int npx = g.npx, npy = g.npy, ie = nx, je = ny;
#ifdef FV3_HOST_EMU
      for (int cn = 0; cn < 4; ++cn) {
#else
      {
        const int cn = blk.tid;
#endif
        if (cn < 4 && (cmask & (1 << cn))) {
          int a, b_, c_;
          if (cn == 0) a = lds_of(1, 1), b_ = lds_of(0, 1), c_ = lds_of(1, 0);
          else if (cn == 1) a = lds_of(ie, 1), b_ = lds_of(npx, 1), c_ = lds_of(ie, 0);
          else if (cn == 2) a = lds_of(ie, je), b_ = lds_of(npx, je), c_ = lds_of(ie, npy);
          else a = lds_of(1, je), b_ = lds_of(0, je), c_ = lds_of(1, npy);
          const Real m = (P[a] + P[b_] + P[c_]) * r3;
          P[a] = m;
          P[b_] = m;
          P[c_] = m;
        }
      }
      blk.group_sync();
    };

    Real nxt[D2_S0];
    {
      const Real *qq = q + t * g.st + (long)k0 * g.sk;
#pragma unroll
      for (int sl = 0; sl < D2_S0; ++sl) nxt[sl] = g0[sl] != 0xffffffffu ? qq[g0[sl]] : (Real)0;
    }
    for (int k = k0; k <= k1; ++k) {
      const long b = t * g.st + (long)k * g.sk;
#pragma unroll
      for (int sl = 0; sl < D2_S0; ++sl)
        if (g0[sl] != 0xffffffffu) A[l0[sl]] = nxt[sl];
      // the heating's inputs of this level and the frame of the next one: in flight while the iterations run
      Real hdp[D2_S3], hdz[D2_S3], hcp[D2_S3], hpt[D2_S3];
      if (HEAT) {
#pragma unroll
        for (int sl = 0; sl < D2_S3; ++sl)
          if (dom3[sl]) {
            hdp[sl] = (h.delp + b)[p3[sl]];
            hdz[sl] = (h.delz + b)[p3[sl]];
            hcp[sl] = (h.cappa + b)[p3[sl]];
            hpt[sl] = (h.pt + b)[p3[sl]];
          }
      }
      if (k < k1) {
        const Real *qq = q + b + g.sk;
#pragma unroll
        for (int sl = 0; sl < D2_S0; ++sl)
          if (g0[sl] != 0xffffffffu) nxt[sl] = qq[g0[sl]];
      }
      blk.group_sync();
      iterate(C1, A, [&](int, const Cell &x, Real v) { B[x.lc] = v; });
      blk.group_sync();
      fill(B);
      iterate(C2, B, [&](int, const Cell &x, Real v) { A[x.lc] = v; });
      blk.group_sync();
      fill(A);
      iterate(C3, A, [&](int sl, const Cell &, Real v) {
        if (out) (out + b)[p3[sl]] = v;
        if (HEAT && dom3[sl]) {
          const Real cp = hcp[sl];
          const Real pkz = fv3_exp(cp / ((Real)1.0 - cp) * fv3_log(h.rdg * hdp[sl] / hdz[sl] * hpt[sl]));
          const Real dtmp = v / (h.cv_air * hdp[sl]);
          Real lim = h.lim0;
          if (k == 0) lim = lim * (Real)0.1;
          if (k == 1) lim = lim * (Real)0.5;
          const Real mag = fv3_min(lim, fabs(dtmp));
          const Real sg = dtmp > (Real)0 ? (Real)1 : (dtmp < (Real)0 ? (Real)-1 : (Real)0);
          (h.pt + b)[p3[sl]] = hpt[sl] + sg * mag / pkz;
        }
      });
      blk.group_sync();  // (the next level's frame overwrites A)
    }
  });
}

}  // namespace

// del2_cubed (three iterations) + apply_diffusive_heating in one pass.  keep_q: also leave the smoothed field in q (through a scratch copy: the
// tiles of one plane read each other's frames).  Returns 1 when this configuration has no fused form (the caller runs the two staged operators).
int fv3_del2_heat_fused(fv3_ctx *c, const fv3_field *q_, double cdd, int nmax, const fv3_field *delp_, const fv3_field *delz_, const fv3_field *cappa_,
                        const fv3_field *pt_, double delt, bool keep_q, void *stream) {
  if (!c) return FV3_ERR_ARG;
  const char *e = getenv("FV3_DEL2_FUSED");  // (read per call: the parity test flips it in one process)
  if (e && e[0] == '0') return 1;
  const Geo g = c->g;
  if (nmax < 3 || g.nx < 10 || g.ny < 10 || g.nh != 3) return 1;
  FV3_FIELD(q, q_) FV3_FIELD(delp, delp_) FV3_FIELD(delz, delz_) FV3_FIELD(cappa, cappa_) FV3_FIELD(pt, pt_)
  fv3_stream_t s = (fv3_stream_t)stream;
  if (getenv("FV3_DEBUG_DEL2")) fprintf(stderr, "[del2_heat_fused] %d x %d, keep_q %d\n", g.nx, g.ny, (int)keep_q);
  del2_fill_corners(c, s, q);
  // FV3_DEL2_HEAT=fused: the heating as the epilogue of the third iteration (nothing of the smoothed field stored); default: the smoothed field goes to
  // scratch and the heating is its own launch (experiment R5-24: the fused epilogue's exp / log / divisions make the tile kernel issue-bound at
  // two waves per SIMD -- 6.6 ms -- where the two launches are each bound by their bytes)
  const char *hm = getenv("FV3_DEL2_HEAT");
  const bool heat_in = hm && !strcmp(hm, "fused");
  Real *out = keep_q || !heat_in ? c->scratch[SC_A] : nullptr;
  const D2Heat h{delp, delz, cappa, pt, (Real)(-c->cst.rdgas / c->cst.grav), (Real)(c->cst.cp_air - c->cst.rdgas), (Real)delt, true};
  if (heat_in)
    d2_launch<false, true>(c, s, q, out, (Real)cdd, h);
  else
    d2_launch<false, false>(c, s, q, out, (Real)cdd, h);
  int any = 0;
  for (int t = 0; t < g.nsub; ++t) {
    const int fl = g.flags[t];
    any |= ((fl & (FV3_W | FV3_S)) == (FV3_W | FV3_S)) | ((fl & (FV3_E | FV3_S)) == (FV3_E | FV3_S)) | ((fl & (FV3_E | FV3_N)) == (FV3_E | FV3_N)) |
           ((fl & (FV3_W | FV3_N)) == (FV3_W | FV3_N));
  }
  if (any) {
    if (heat_in)
      d2_launch<true, true>(c, s, q, out, (Real)cdd, h);
    else
      d2_launch<true, false>(c, s, q, out, (Real)cdd, h);
  }
  if (!heat_in) {  // (fv3_apply_diffusive_heating's expressions, the smoothed field read from scratch)
    const Real *hs = out;
    launch3(c, s, Box{1, g.nx, 1, g.ny, 0, g.nz - 1}, [=] FV3_HD(int t, int k, int i, int j) {
      const long p = t * g.st + k * g.sk + IX(i, j);
      const Real cp = cappa[p];
      const Real pkz = fv3_exp(cp / ((Real)1.0 - cp) * fv3_log(h.rdg * delp[p] / delz[p] * pt[p]));
      const Real dtmp = hs[p] / (h.cv_air * delp[p]);
      Real lim = h.lim0;
      if (k == 0) lim = lim * (Real)0.1;
      if (k == 1) lim = lim * (Real)0.5;
      const Real mag = fv3_min(lim, fabs(dtmp));
      const Real sg = dtmp > (Real)0 ? (Real)1 : (dtmp < (Real)0 ? (Real)-1 : (Real)0);
      pt[p] = pt[p] + sg * mag / pkz;
    });
  }
  if (out && keep_q) {
    const int isd = 1 - g.nh, ied = g.nx + g.nh, jsd = 1 - g.nh, jed = g.ny + g.nh;
    launch3<4>(c, s, Box{isd, ied, jsd, jed, 0, g.nz - 1}, [=] FV3_HD(int t, int k, int i, int j) {
      const long p = t * g.st + k * g.sk + IX(i, j);
      q[p] = out[p];
    });
  }
  return fv3_post(c, s, "del2_heat_fused");
}
