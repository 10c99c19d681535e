// fv3_agpr.h -- a column's worth of one fp64 temporary in the ACCUMULATION registers of a lane (gfx950, one wave per SIMD).
//
// Why: at one wave per SIMD (what the 40 KB LDS line of the wave Riemann solver allows) a wave owns 512 registers per lane -- 256
// architectural + 256 accumulation -- and the solver uses ~170 architectural ones and no accumulation register at all: 64 KB per wave of
// on-chip storage nobody uses, while the tridiagonal `gam` goes through HBM (written by the two forward sweeps, read back by the two back
// substitutions: four field passes per call in kernels that run at the HBM rate).  The compiler cannot index registers with a run-time
// level (a register-tuple array indexed through `s_set_gpr_idx` ends up copied in and out of the accumulation file around every
// access: measured 40 % slower), so the access is written by hand: a computed branch (`s_getpc` + level x 20 bytes,
// `s_setpc`) into a table of 80 cases, each `v_accvgpr_write a[2k], lo; v_accvgpr_write a[2k+1], hi; s_branch end` (8 + 8 + 4 bytes).
// The level is wave-uniform.  The compiler does not know that the values live there: it must not use accumulation registers itself in
// these kernels -- it does so only when it runs out of architectural registers.  Guards (tests/test_kernel_budgets.py): the kernels must stay
// well below the architectural limit (<= 240 of 256, no spill) with exactly the 160 accumulation registers claimed here, and the disassembly
// of the built library must not name an accumulation register anywhere outside the jump tables of this file; every access statement, reads
// included, lists a0 .. a159 as clobbered.
#pragma once
#include "fv3_common.h"

#define FV3_AGPR_LEVELS 80      // fp64: two registers per level
#define FV3_AGPR_LEVELS_F32 128  // fp32: one
#if defined(__HIP_DEVICE_COMPILE__)
// clang-format off
#define FV3_AG_CASES(X) \
  X(0, 1) X(2, 3) X(4, 5) X(6, 7) X(8, 9) X(10, 11) X(12, 13) X(14, 15) \
  X(16, 17) X(18, 19) X(20, 21) X(22, 23) X(24, 25) X(26, 27) X(28, 29) X(30, 31) \
  X(32, 33) X(34, 35) X(36, 37) X(38, 39) X(40, 41) X(42, 43) X(44, 45) X(46, 47) \
  X(48, 49) X(50, 51) X(52, 53) X(54, 55) X(56, 57) X(58, 59) X(60, 61) X(62, 63) \
  X(64, 65) X(66, 67) X(68, 69) X(70, 71) X(72, 73) X(74, 75) X(76, 77) X(78, 79) \
  X(80, 81) X(82, 83) X(84, 85) X(86, 87) X(88, 89) X(90, 91) X(92, 93) X(94, 95) \
  X(96, 97) X(98, 99) X(100, 101) X(102, 103) X(104, 105) X(106, 107) X(108, 109) X(110, 111) \
  X(112, 113) X(114, 115) X(116, 117) X(118, 119) X(120, 121) X(122, 123) X(124, 125) X(126, 127) \
  X(128, 129) X(130, 131) X(132, 133) X(134, 135) X(136, 137) X(138, 139) X(140, 141) X(142, 143) \
  X(144, 145) X(146, 147) X(148, 149) X(150, 151) X(152, 153) X(154, 155) X(156, 157) X(158, 159)
#define FV3_AG_CLOBBERS \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
  "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
  "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
  "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
  "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
  "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
  "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
  "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", \
  "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
  "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159"
// clang-format on
#define FV3_AG_W(n0, n1) "v_accvgpr_write_b32 a" #n0 ", %[lo]\n v_accvgpr_write_b32 a" #n1 ", %[hi]\n s_branch .LFV3AG%=\n"
#define FV3_AG_R(n0, n1) "v_accvgpr_read_b32 %[lo], a" #n0 "\n v_accvgpr_read_b32 %[hi], a" #n1 "\n s_branch .LFV3AG%=\n"
// (s_getpc returns the address of the instruction behind it; the three instructions up to the table are 12 bytes)
#define FV3_AG_JUMP "s_getpc_b64 vcc\n s_add_u32 vcc_lo, vcc_lo, %[t]\n s_addc_u32 vcc_hi, vcc_hi, 0\n s_setpc_b64 vcc\n"

__device__ __attribute__((always_inline)) inline void fv3_agpr_set(int k, double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int t = k * 20 + 12;
  asm volatile(FV3_AG_JUMP FV3_AG_CASES(FV3_AG_W) ".LFV3AG%=:\n" : : [t] "s"(t), [lo] "v"(lo), [hi] "v"(hi) : "vcc", "scc", FV3_AG_CLOBBERS);
}
__device__ __attribute__((always_inline)) inline double fv3_agpr_get(int k) {
  int lo, hi;
  const int t = k * 20 + 12;
  // (the clobber list on a READ: it tells the compiler that no value of its own survives in a0 .. a159 across this statement either, so that the
  //  only live ranges it could ever place there are those between two consecutive accesses -- and tests/test_kernel_budgets.py checks in the
  //  disassembly that it places none: no instruction outside these tables names an accumulation register)
  asm volatile(FV3_AG_JUMP FV3_AG_CASES(FV3_AG_R) ".LFV3AG%=:\n" : [lo] "=&v"(lo), [hi] "=&v"(hi) : [t] "s"(t) : "vcc", "scc", FV3_AG_CLOBBERS);
  return __hiloint2double(hi, lo);
}

// four consecutive levels at once (one access site for the U = 4 levels a forward sweep produces per loop iteration): 20 cases of
// 8 writes + branch = 68 bytes; slot s of the column is a[2s], a[2s+1] as above, group g = slots 4g .. 4g+3
// clang-format off
#define FV3_AG_GROUPS(X) \
  X(0, 1, 2, 3, 4, 5, 6, 7) X(8, 9, 10, 11, 12, 13, 14, 15) \
  X(16, 17, 18, 19, 20, 21, 22, 23) X(24, 25, 26, 27, 28, 29, 30, 31) \
  X(32, 33, 34, 35, 36, 37, 38, 39) X(40, 41, 42, 43, 44, 45, 46, 47) \
  X(48, 49, 50, 51, 52, 53, 54, 55) X(56, 57, 58, 59, 60, 61, 62, 63) \
  X(64, 65, 66, 67, 68, 69, 70, 71) X(72, 73, 74, 75, 76, 77, 78, 79) \
  X(80, 81, 82, 83, 84, 85, 86, 87) X(88, 89, 90, 91, 92, 93, 94, 95) \
  X(96, 97, 98, 99, 100, 101, 102, 103) X(104, 105, 106, 107, 108, 109, 110, 111) \
  X(112, 113, 114, 115, 116, 117, 118, 119) X(120, 121, 122, 123, 124, 125, 126, 127) \
  X(128, 129, 130, 131, 132, 133, 134, 135) X(136, 137, 138, 139, 140, 141, 142, 143) \
  X(144, 145, 146, 147, 148, 149, 150, 151) X(152, 153, 154, 155, 156, 157, 158, 159)
// clang-format on
#define FV3_AG_W4(r0, r1, r2, r3, r4, r5, r6, r7)                                                                                       \
  "v_accvgpr_write_b32 a" #r0 ", %[l0]\n v_accvgpr_write_b32 a" #r1 ", %[h0]\n v_accvgpr_write_b32 a" #r2 ", %[l1]\n v_accvgpr_write_b32 a" #r3 ", %[h1]\n" \
  "v_accvgpr_write_b32 a" #r4 ", %[l2]\n v_accvgpr_write_b32 a" #r5 ", %[h2]\n v_accvgpr_write_b32 a" #r6 ", %[l3]\n v_accvgpr_write_b32 a" #r7 ", %[h3]\n s_branch .LFV3AG%=\n"
__device__ __attribute__((always_inline)) inline void fv3_agpr_set4(int g, double v0, double v1, double v2, double v3) {
  const int l0 = __double2loint(v0), h0 = __double2hiint(v0), l1 = __double2loint(v1), h1 = __double2hiint(v1);
  const int l2 = __double2loint(v2), h2 = __double2hiint(v2), l3 = __double2loint(v3), h3 = __double2hiint(v3);
  const int t = g * 68 + 12;
  asm volatile(FV3_AG_JUMP FV3_AG_GROUPS(FV3_AG_W4) ".LFV3AG%=:\n"
               :
               : [t] "s"(t), [l0] "v"(l0), [h0] "v"(h0), [l1] "v"(l1), [h1] "v"(h1), [l2] "v"(l2), [h2] "v"(h2), [l3] "v"(l3), [h3] "v"(h3)
               : "vcc", "scc", FV3_AG_CLOBBERS);
}

// fp32 build: one register per level, 128 levels (the L127 configurations), 12-byte cases
// clang-format off
#define FV3_AG_CASES1(X) \
  X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
  X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) \
  X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) X(41) X(42) X(43) X(44) X(45) X(46) X(47) \
  X(48) X(49) X(50) X(51) X(52) X(53) X(54) X(55) X(56) X(57) X(58) X(59) X(60) X(61) X(62) X(63) \
  X(64) X(65) X(66) X(67) X(68) X(69) X(70) X(71) X(72) X(73) X(74) X(75) X(76) X(77) X(78) X(79) \
  X(80) X(81) X(82) X(83) X(84) X(85) X(86) X(87) X(88) X(89) X(90) X(91) X(92) X(93) X(94) X(95) \
  X(96) X(97) X(98) X(99) X(100) X(101) X(102) X(103) X(104) X(105) X(106) X(107) X(108) X(109) X(110) X(111) \
  X(112) X(113) X(114) X(115) X(116) X(117) X(118) X(119) X(120) X(121) X(122) X(123) X(124) X(125) X(126) X(127)
#define FV3_AG_CLOBBERS1 \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", \
  "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", \
  "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
  "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", \
  "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", \
  "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
  "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", \
  "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
// clang-format on
#define FV3_AG_W1(n0) "v_accvgpr_write_b32 a" #n0 ", %[lo]\n s_branch .LFV3AG%=\n"
#define FV3_AG_R1(n0) "v_accvgpr_read_b32 %[lo], a" #n0 "\n s_branch .LFV3AG%=\n"
__device__ __attribute__((always_inline)) inline void fv3_agpr_set(int k, float v) {
  const int lo = __float_as_int(v);
  const int t = k * 12 + 12;
  asm volatile(FV3_AG_JUMP FV3_AG_CASES1(FV3_AG_W1) ".LFV3AG%=:\n" : : [t] "s"(t), [lo] "v"(lo) : "vcc", "scc", FV3_AG_CLOBBERS1);
}
__device__ __attribute__((always_inline)) inline float fv3_agpr_get_f32(int k) {
  int lo;
  const int t = k * 12 + 12;
  asm volatile(FV3_AG_JUMP FV3_AG_CASES1(FV3_AG_R1) ".LFV3AG%=:\n" : [lo] "=&v"(lo) : [t] "s"(t) : "vcc", "scc", FV3_AG_CLOBBERS1);
  return __int_as_float(lo);
}

// clang-format off
#define FV3_AG_GROUPS1(X) \
  X(0, 1, 2, 3) X(4, 5, 6, 7) X(8, 9, 10, 11) X(12, 13, 14, 15) \
  X(16, 17, 18, 19) X(20, 21, 22, 23) X(24, 25, 26, 27) X(28, 29, 30, 31) \
  X(32, 33, 34, 35) X(36, 37, 38, 39) X(40, 41, 42, 43) X(44, 45, 46, 47) \
  X(48, 49, 50, 51) X(52, 53, 54, 55) X(56, 57, 58, 59) X(60, 61, 62, 63) \
  X(64, 65, 66, 67) X(68, 69, 70, 71) X(72, 73, 74, 75) X(76, 77, 78, 79) \
  X(80, 81, 82, 83) X(84, 85, 86, 87) X(88, 89, 90, 91) X(92, 93, 94, 95) \
  X(96, 97, 98, 99) X(100, 101, 102, 103) X(104, 105, 106, 107) X(108, 109, 110, 111) \
  X(112, 113, 114, 115) X(116, 117, 118, 119) X(120, 121, 122, 123) X(124, 125, 126, 127)
// clang-format on
#define FV3_AG_W41(r0, r1, r2, r3) \
  "v_accvgpr_write_b32 a" #r0 ", %[l0]\n v_accvgpr_write_b32 a" #r1 ", %[l1]\n v_accvgpr_write_b32 a" #r2 ", %[l2]\n v_accvgpr_write_b32 a" #r3 ", %[l3]\n s_branch .LFV3AG%=\n"
__device__ __attribute__((always_inline)) inline void fv3_agpr_set4(int g, float v0, float v1, float v2, float v3) {
  const int l0 = __float_as_int(v0), l1 = __float_as_int(v1), l2 = __float_as_int(v2), l3 = __float_as_int(v3);
  const int t = g * 36 + 12;
  asm volatile(FV3_AG_JUMP FV3_AG_GROUPS1(FV3_AG_W41) ".LFV3AG%=:\n" : : [t] "s"(t), [l0] "v"(l0), [l1] "v"(l1), [l2] "v"(l2), [l3] "v"(l3) : "vcc", "scc", FV3_AG_CLOBBERS1);
}
#endif
