// fv3_a2b.hip -- plain-store instantiation of the corner interpolation + the C-ABI entry.  [kernels: fv3_a2b.h]
#include "fv3_a2b.h"

void a2b_ord4(fv3_ctx *c, fv3_stream_t s, Real *qin, Real *qout, int kin0, int kout0, int nk, bool replace, Real scale) {
  const Geo g = c->g;
  Real *out = replace ? c->scratch[SC_K] : qout;
  a2b_ord4_t<8>(c, s, qin, kin0, replace ? kin0 : kout0, nk, scale, A2bStore{out, g.st, g.sk});
  if (replace) {
    launch3<4>(c, s, Box{1, g.nx + 1, 1, g.ny + 1, kin0, kin0 + nk - 1}, [=] FV3_HD(int t, int k, int i, int j) {
      const long p = t * g.st + k * g.sk + IX(i, j);
      qin[p] = out[p];
    });
  }
}

extern "C" int fv3_a2b_ord4(fv3_ctx *c, const fv3_field *qin_, const fv3_field *qout_, int kstart, int nk, int replace, void *stream) {
  if (!c) return FV3_ERR_ARG;
  FV3_FIELD(qin, qin_) FV3_FIELD(qout, qout_)
  if (kstart < 0 || nk < 1 || kstart + nk > c->g.nkA) return fv3_fail(c, FV3_ERR_ARG, "a2b_ord4: vertical range outside the allocation");
  if (qin == qout) {
    if (!replace) return fv3_fail(c, FV3_ERR_ARG, "a2b_ord4: qin and qout alias but replace is off");
    a2b_ord4(c, (fv3_stream_t)stream, qin, qout, kstart, kstart, nk, true);
  } else {
    a2b_ord4(c, (fv3_stream_t)stream, qin, qout, kstart, kstart, nk, false);
    if (replace) {
      const Geo g = c->g;
      launch3(c, (fv3_stream_t)stream, Box{1, g.nx + 1, 1, g.ny + 1, kstart, kstart + nk - 1}, [=] FV3_HD(int t, int k, int i, int j) {
        const long p = t * g.st + k * g.sk + IX(i, j);
        qin[p] = qout[p];
      });
    }
  }
  return fv3_post(c, (fv3_stream_t)stream, "a2b_ord4");
}
