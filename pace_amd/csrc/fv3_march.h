// fv3_march.h -- building blocks of the round-5 marching kernels (fv3_tp4x.hip: d_sw's two-tracer marches; fv3_tp2x.hip: the single-tracer
// marches of the vorticity / interface-height transports).  What they have in common (the reasons are in the header of fv3_tp4x.hip): a row step
// without rare paths, unrolled by three with static rotation of the register sets in flight, scalar-base addressing, shared reconstructions
// between neighbouring lanes, one refined reciprocal per denominator.
#pragma once
#include "fv3_common.h"
#include "fv3_math.h"

#ifndef PX_NO_FENCE
#define PX_FENCE() FV3_SCHED_FENCE()  // the scheduler may not move code across the phases of a step (unfenced, the unrolled march overlaps them until the registers spill)
#else
#define PX_FENCE() ((void)0)
#endif
// Ablation builds (diagnostic, never the product library; profiles/r05_ablation.md): -DPX_ABL=1 keeps every load and store of a step and
// replaces the arithmetic by one sum of what was loaded (the floor the memory system sets for this access pattern); -DPX_ABL=2 keeps the
// arithmetic and the stores and replaces the loads inside the march by the registers of the first rows, passed through an empty asm so
// that nothing becomes loop-invariant (the floor instruction issue sets).
#ifndef PX_ABL
#define PX_ABL 0
#endif
#if PX_ABL == 2 && !defined(FV3_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
#define PX_KEEP(x) asm volatile("" : "+v"(x))
#else
#define PX_KEEP(x) ((void)0)
#endif
#if !defined(FV3_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
#define PX_OPAQUE_S(x) asm volatile("" : "+s"(x))
#else
#define PX_OPAQUE_S(x) ((void)0)
#endif
#ifdef PX_TRACER_FENCE
#define PX_FENCE_T() FV3_SCHED_FENCE()  // ... nor across the tracers inside a phase
#else
#define PX_FENCE_T() ((void)0)
#endif
// the limiter flag of the previous lane: the lane mask shifted by one (a scalar instruction); false for lane 0
#if !defined(FV3_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
#define FV3_LANE_SHR1_FLAG(arr, l, lane) __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_ballot_w64((arr)[l]) << 1)
#else
#define FV3_LANE_SHR1_FLAG(arr, l, lane) ((lane) >= 1 ? (arr)[(l) - 1] : false)
#endif
// A value that outlives the step which consumes its load (q -> the PPM windows, the M area flux, the old air mass) is MOVED out of the load's
// destination register at the point of consumption.  Left to the compiler, the register the load was issued into stays the value's home for
// up to four more steps, i.e. across the back edge of the unrolled march, where it is copied while a younger load into the same name is still in
// flight -- and that copy waits for it with vmcnt(0): the whole prefetch, once per three steps (seen in the ISA of the first build).
#if !defined(FV3_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
FV3_DEV inline double px_move(double x) {
  double y;
  asm volatile("v_mov_b64 %0, %1" : "=v"(y) : "v"(x));
  return y;
}
FV3_DEV inline float px_move(float x) {
  float y;
  asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(x));
  return y;
}
#else
inline Real px_move(Real x) { return x; }
#endif
// x / y with the denominator's refined reciprocal formed once: r = px_rcp(y); q = px_quot(x, y, r) -- the instruction sequence of
// fv3_div split at the point where the numerator enters (bit for bit `/` for normal operands; the host emulation and fp32 divide)
FV3_HD inline Real px_rcp(Real y) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FV3_DIV_PLAIN)
  if constexpr (sizeof(Real) == 8) {
    const double r0 = __builtin_amdgcn_rcp((double)y);
    const double e0 = __builtin_fma(-(double)y, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-(double)y, r1, 1.0);
    return (Real)__builtin_fma(r1, e1, r1);
  }
#endif
  return y;
}
FV3_HD inline Real px_quot(Real x, Real y, Real r) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(FV3_DIV_PLAIN)
  if constexpr (sizeof(Real) == 8) {
    const double q = (double)x * (double)r;
    const double e2 = __builtin_fma(-(double)y, q, (double)x);
    return (Real)__builtin_fma(e2, (double)r, q);
  }
#endif
  (void)r;
  return x / y;
}

// element at (uniform base) + (32-bit byte offset)
// (-DPX_METRIC_HOT, timing experiment R5-20 only: every 2-D metric read lands in the first 2 KB of its array -- the instruction stream is the
//  product's, the reads always hit; wrong values.  Never in a product library.)
#ifdef PX_METRIC_HOT
FV3_HD inline Real px_ld(const Real *base, unsigned boff) { return *fv3_at(base, boff & 0x7f8u); }
#else
FV3_HD inline Real px_ld(const Real *base, unsigned boff) { return *fv3_at(base, boff); }
#endif
// ... of a 3-D field row a wave reads ONCE (-DPX_NT_LOADS: with the streaming hint, so that the rows of the 2-D metric terms, which the
// sixteen levels of a tile share through the XCD's L2, are not evicted by them -- experiment R5-15)
FV3_HD inline Real px_ld3(const Real *base, unsigned boff) {
#if defined(PX_NT_LOADS) && !defined(FV3_HOST_EMU)
  return __builtin_nontemporal_load(fv3_at(base, boff));
#else
  return *fv3_at(base, boff);
#endif
}

// The edge value at the low face of cell s next to a W / E tile edge (fv3_ppm.h: ppm_al), for the lane that holds cell s: kind 1 = the face one
// before the edge (s == 0 / np-1), 2 = the edge face itself (the two-sided, width-weighted mean; s == 1 / np), 3 = the face one behind it
// (s == 2 / np+1); 0 = an interior face (al_int).  a, b, c_, d = q(s-2 .. s+1); ma .. md = the cell widths of those cells.  Every form is evaluated in
// every lane of the strip (the lanes execute together anyway) and the lane keeps its own: the expressions are ppm_al's, so are the bits.
#include "fv3_ppm.h"
FV3_HD inline Real px_al_edge(Real al_int, int kind, Real a, Real b, Real c_, Real d, Real ma, Real mb, Real mc, Real md) {
  const Real f1 = PPM_C1 * a + PPM_C2 * b + PPM_C3 * c_;
  const Real f2 = ppm_edge_mean(a, b, c_, d, ma, mb, mc, md);
  const Real f3 = PPM_C3 * b + PPM_C2 * c_ + PPM_C1 * d;
  return kind == 1 ? f1 : kind == 2 ? f2 : kind == 3 ? f3 : al_int;
}
