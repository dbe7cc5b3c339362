// Cooperative interior-point solve, general form: ONE PROBLEM PER HALF DPP ROW (8 lanes), eight problems per
// wavefront, second-order cones of dimension 2 AND 3 -- the planar push (src/models/planar_push/model.jl:121-187:
// one contact, four 3-d friction cones at the corners of the block, one 2-d cone at the pusher; BASELINE config 3 runs
// 12 850 independent nz = 35 solves of it, src/gradient_bundle.jl:87-104).
//
// Same idea as od_coop.h (one problem per 16-lane row, cones of dimension 2): every contact and every friction cone owns
// a lane ("D" values), the configuration q, the dynamics rows and their nq x nq Schur complement are replicated in the
// lanes of the problem ("R" values), the static elimination of gen/<model>.h becomes block algebra inside the lanes.
// What is different here:
//   * a cone lane holds (psi, b1, b2 | s_psi, s_b1, s_b2) with TWO tangential-velocity rows and two tail rows; its 3 x 3
//     arrow block in (d s_psi, d b1, d b2) is eliminated in the order of the serial program (role swap |psi| > |s_psi|
//     as a row / column permutation chosen by selects, then three scalar pivots); a 2-d cone is the same code with
//     b2 = s_b2 = 0 and no second velocity row;
//   * 8 lanes per problem (6 roles for the planar push), so no mirror half: a lane computes the primal and the dual step
//     length of its cone one after the other, the two violation norms are two butterflies;
//   * cross-lane reads of 64-bit values within a HALF row: v_mov_b64_dpp / v_fmac_f64_dpp row_newbcast:L with
//     bank_mask 0x3 (lanes 0..7 <- lane L) followed by row_newbcast:L+8 with bank_mask 0xc (lanes 8..15 <- lane L+8).
//     Measured (tools/ubench/dpp_bank_mask*.hip): bank masks work on the DP forms; the one hazard is a bank-0xc
//     fmac IMMEDIATELY after a bank-0x3 fmac on the same accumulator (stale forwarding; one wait state cures it), so a
//     generated block issues all its 0x3 instructions, one s_nop, then all its 0xc instructions.
//
// Type discipline as in od_coop.h: R values are `double`; D values are `RO::V` -- `double` on the device (Row8Dev), an
// 8-lane vector in the host test build (Row8Emu, tests/host_emu), where mixing the two up does not compile.
#pragma once
#include "od_units.h"

namespace od {

// ---------------------------------------------------------------------------------------------------------------
// host emulation of an 8-lane group (TEST HARNESS: only the host build instantiates it)
// ---------------------------------------------------------------------------------------------------------------
struct Mask8 { bool m[8]; };
struct Vec8 {
  double v[8];
  Vec8() = default;
  Vec8(double x) { for (int i = 0; i < 8; ++i) v[i] = x; }
};
#define OD_V8_BIN(op)                                                                                                \
  inline Vec8 operator op(const Vec8& a, const Vec8& b) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = a.v[i] op b.v[i]; return r; } \
  inline Vec8 operator op(const Vec8& a, double b) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = a.v[i] op b; return r; }              \
  inline Vec8 operator op(double a, const Vec8& b) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = a op b.v[i]; return r; }
OD_V8_BIN(+) OD_V8_BIN(-) OD_V8_BIN(*) OD_V8_BIN(/)
#undef OD_V8_BIN
inline Vec8 operator-(const Vec8& a) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = -a.v[i]; return r; }
#define OD_V8_CMP(op)                                                                                                \
  inline Mask8 operator op(const Vec8& a, const Vec8& b) { Mask8 r; for (int i = 0; i < 8; ++i) r.m[i] = a.v[i] op b.v[i]; return r; } \
  inline Mask8 operator op(const Vec8& a, double b) { Mask8 r; for (int i = 0; i < 8; ++i) r.m[i] = a.v[i] op b; return r; }
OD_V8_CMP(>) OD_V8_CMP(<) OD_V8_CMP(!=) OD_V8_CMP(<=) OD_V8_CMP(==)
#undef OD_V8_CMP
inline Mask8 operator&&(const Mask8& a, const Mask8& b) { Mask8 r; for (int i = 0; i < 8; ++i) r.m[i] = a.m[i] && b.m[i]; return r; }
inline Mask8 operator!(const Mask8& a) { Mask8 r; for (int i = 0; i < 8; ++i) r.m[i] = !a.m[i]; return r; }
inline Mask8 operator||(const Mask8& a, const Mask8& b) { Mask8 r; for (int i = 0; i < 8; ++i) r.m[i] = a.m[i] || b.m[i]; return r; }
inline Vec8 od_rcp(const Vec8& a) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = od_rcp(a.v[i]); return r; }
inline Vec8 od_rsqrt(const Vec8& a) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = od_rsqrt(a.v[i]); return r; }
inline Vec8 od_sqrt(const Vec8& a) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = od_sqrt(a.v[i]); return r; }
inline Vec8 od_sin(const Vec8& a) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = od_sin(a.v[i]); return r; }
inline Vec8 od_cos(const Vec8& a) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = od_cos(a.v[i]); return r; }
template <int Q> inline Vec8 od_rootinv(const Vec8& a) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = od_rootinv<Q>(a.v[i]); return r; }
inline Vec8 od_abs(const Vec8& a) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = od_abs(a.v[i]); return r; }
inline Vec8 od_max(const Vec8& a, const Vec8& b) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = od_max(a.v[i], b.v[i]); return r; }
inline Vec8 od_fmax(const Vec8& a, const Vec8& b) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = od_fmax(a.v[i], b.v[i]); return r; }
inline Vec8 od_fmin(const Vec8& a, const Vec8& b) { Vec8 r; for (int i = 0; i < 8; ++i) r.v[i] = od_fmin(a.v[i], b.v[i]); return r; }

struct Row8Emu {
  using V = Vec8;
  using B = Mask8;
  static constexpr bool DEVICE = false;
  static V lane_table(const double (&t)[8]) { V r; for (int i = 0; i < 8; ++i) r.v[i] = t[i]; return r; }
  static B lane_flag(unsigned bits) { B r; for (int i = 0; i < 8; ++i) r.m[i] = (bits >> i) & 1u; return r; }
  static V sel(const B& m, const V& a, const V& b) { V r; for (int i = 0; i < 8; ++i) r.v[i] = m.m[i] ? a.v[i] : b.v[i]; return r; }
  static V sel(const B& m, double a, const V& b) { return sel(m, V(a), b); }
  static V sel(const B& m, const V& a, double b) { return sel(m, a, V(b)); }
  static V sel(const B& m, double a, double b) { return sel(m, V(a), V(b)); }
  template <int L> static double bc(const V& x) { return x.v[L]; }
  template <int L> static void fmac(double& acc, const V& x, double m) { acc = od_fma(x.v[L], m, acc); }
  template <int L> static void fnmac(double& acc, const V& x, double m) { acc = od_fma(x.v[L], -m, acc); }
  template <int N> static V shr(const V& x) { V r = x; for (int i = N; i < 8; ++i) r.v[i] = x.v[i - N]; return r; }
  static V xor1(const V& x) { V r; for (int i = 0; i < 8; ++i) r.v[i] = x.v[i ^ 1]; return r; }
  static V xor2(const V& x) { V r; for (int i = 0; i < 8; ++i) r.v[i] = x.v[i ^ 2]; return r; }
  static V half_mirror(const V& x) { V r; for (int i = 0; i < 8; ++i) r.v[i] = x.v[7 - i]; return r; }
  static double first(const V& x) { return x.v[0]; }              // a value every lane of the group holds (after a reduction)
  // destination-row tables (-1 = lane holds nothing to store), three per packed int
  struct I { int v[8]; };
  static I lane_pack3(const int (&a)[8], const int (&b)[8], const int (&c)[8]) {
    I r;
    for (int i = 0; i < 8; ++i) r.v[i] = (a[i] & 0xFF) | ((b[i] & 0xFF) << 8) | ((c[i] & 0xFF) << 16);
    return r;
  }
  template <int W, class View_> static void store(const View_& v, const I& idx, long k, const V& x) {
    for (int i = 0; i < 8; ++i) { const int j = (idx.v[i] >> (8 * W)) & 0xFF; if (j != 0xFF) v.at(j, k) = x.v[i]; }
  }
  static V vmax(const V& a, const V& b) { return od_fmax(a, b); }
  static V vmin(const V& a, const V& b) { return od_fmin(a, b); }
  static bool first_lane() { return true; }
  static void arrived(const double&) {}
  // lane-parallel line-search trials: lane g of the group takes step alpha 2^-g
  static V lane_ldexp(double a) { V r; for (int i = 0; i < 8; ++i) r.v[i] = od_ldexp(a, -i); return r; }
  static B lane_below(int n) { B r; for (int i = 0; i < 8; ++i) r.m[i] = i < n; return r; }
  static unsigned grp_ballot(const B& b) { unsigned m = 0; for (int i = 0; i < 8; ++i) m |= (b.m[i] ? 1u : 0u) << i; return m; }
  static double opaque(double x) { return x; }
};

#if defined(__HIP_DEVICE_COMPILE__)
// the device group: lanes 8g .. 8g+7 of a wavefront (two groups per DPP row)
struct Row8Dev {
  using V = double;
  using B = bool;
  static constexpr bool DEVICE = true;
  __device__ __forceinline__ static int lane() { return (int)(threadIdx.x & 7); }
  __device__ __forceinline__ static V lane_table(const double (&t)[8]) { return t[lane()]; }
  __device__ __forceinline__ static B lane_flag(unsigned bits) { return (bits >> lane()) & 1u; }
  __device__ __forceinline__ static V sel(B m, V a, V b) { return m ? a : b; }
  template <int L> __device__ __forceinline__ static double bc(double x) {
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0x3\n\tv_mov_b64_dpp %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xc"
        : "=&v"(r) : "v"(x), "n"(L), "n"(L + 8));
    return r;
  }
  template <int L> __device__ __forceinline__ static void fmac(double& acc, double x, double m) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0x3\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xc"
        : "+v"(acc) : "v"(x), "v"(m), "n"(L), "n"(L + 8));
  }
  template <int L> __device__ __forceinline__ static void fnmac(double& acc, double x, double m) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0x3\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%4 row_mask:0xf bank_mask:0xc"
        : "+v"(acc) : "v"(x), "v"(m), "n"(L), "n"(L + 8));
  }
  template <int CTRL> __device__ __forceinline__ static double dpp32(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  }
  // lane l <- lane l-N of the ROW: lanes 8..8+N-1 read the other group's lanes (callers select those results away)
  template <int N> __device__ __forceinline__ static double shr(double x) { return dpp32<0x110 + N>(x); }
  __device__ __forceinline__ static double xor1(double x) { return dpp32<0xB1>(x); }          // quad_perm [1,0,3,2]
  __device__ __forceinline__ static double xor2(double x) { return dpp32<0x4E>(x); }          // quad_perm [2,3,0,1]
  __device__ __forceinline__ static double half_mirror(double x) { return dpp32<0x141>(x); }  // lane l <-> 7-l within 8
  __device__ __forceinline__ static double first(double x) { return x; }
  using I = int;
  __device__ __forceinline__ static I lane_pack3(const int (&a)[8], const int (&b)[8], const int (&c)[8]) {
    const int l = lane();
    return (a[l] & 0xFF) | ((b[l] & 0xFF) << 8) | ((c[l] & 0xFF) << 16);
  }
  template <int W, class View_> __device__ __forceinline__ static void store(const View_& v, I idx, long k, double x) {
    const int i = (idx >> (8 * W)) & 0xFF;
    if (i != 0xFF) v.at(i, k) = x;
  }
  __device__ __forceinline__ static double vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
  __device__ __forceinline__ static double vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
  __device__ __forceinline__ static bool first_lane() { return lane() == 0; }
  __device__ __forceinline__ static void arrived(const double& x) { asm volatile("" ::"v"(x)); }
  __device__ __forceinline__ static V lane_ldexp(double a) { return od_ldexp(a, -lane()); }
  __device__ __forceinline__ static B lane_below(int n) { return lane() < n; }
  // the 8 lanes' predicate bits of this group (under divergence: of the groups that execute)
  __device__ __forceinline__ static unsigned grp_ballot(bool b) { return (unsigned)(__builtin_amdgcn_ballot_w64(b) >> (threadIdx.x & 56)) & 0xFFu; }
  __device__ __forceinline__ static double opaque(double x) { asm volatile("" : "+v"(x)); return x; }
};
#endif

// sum / max / min over the 8 lanes of the group; every lane ends with the same bits
template <class RO> OD_HD typename RO::V grp_sum(typename RO::V v) {
  v = v + RO::xor1(v);
  v = v + RO::xor2(v);
  v = v + RO::half_mirror(v);
  return v;
}
template <class RO> OD_HD typename RO::V grp_max(typename RO::V v) {
  v = RO::vmax(v, RO::xor1(v));
  v = RO::vmax(v, RO::xor2(v));
  v = RO::vmax(v, RO::half_mirror(v));
  return v;
}
template <class RO> OD_HD typename RO::V grp_min(typename RO::V v) {
  v = RO::vmin(v, RO::xor1(v));
  v = RO::vmin(v, RO::xor2(v));
  v = RO::vmin(v, RO::half_mirror(v));
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// data of one problem, spread over its lanes
// ---------------------------------------------------------------------------------------------------------------
// z:  q replicated; per lane (P0, P1, P2 | D0, D1, D2) = primal | dual members:
//       contact i : P0 = gamma_i, D0 = s_i, the rest 0       cone c : (psi, b1, b2 | s_psi, s_b1, s_b2), b2 = s_b2 = 0 for a 2-d cone
//       lanes without a role hold (1, 0, 0 | 1, 0, 0) and are masked out of every reduction
// r:  dynamics rows replicated; per lane r1a (contact: slack row | cone: first tangential-velocity row), r1b (second velocity
//     row), r2 (psi row), rA (bilinear row | cone head row), rB1, rB2 (cone tail rows)
template <int NQ, class V> struct C3Vec { double q[NQ]; V P0, P1, P2, D0, D1, D2; };
template <int NQ, class V> struct C3Res { double rd[NQ]; V r1a, r1b, r2, rA, rB1, rB2; };

template <class CM, class RO> struct C3Lanes {
  using V = typename RO::V;
  using B = typename RO::B;
  static constexpr int NR = CM::NC + CM::NK;
  B is_contact, is_cone, is_role, has_partner;
  B role[NR > 0 ? NR : 1];
  V jfa[CM::NQ], jfb[CM::NQ];   // constant entries of the lane's aux-row Jacobians w.r.t. q (slack | velocity row 1 ; velocity row 2)
  V c_s, c_va, c_vb, c_psi;     // r1a = e1a + c_s D0 + c_va D1 ;  r1b = e1b + c_vb D2  (c_v = d(velocity row)/d s_b = +-1, 0 where there is none)
  V gcoef, gconst;              // psi row: c_psi P0 + gcoef gamma_partner + gconst (set per knot)
  V reg_floor;                  // 0 on contact lanes, -inf elsewhere
  typename RO::I zg_p, zg_d;    // rows of z the lane's (P0, P1, P2) / (D0, D1, D2) go to in the gradient hand-over
  OD_HD void init() {
    zg_p = RO::lane_pack3(CM::IDX_P0, CM::IDX_P1, CM::IDX_P2);
    zg_d = RO::lane_pack3(CM::IDX_D0, CM::IDX_D1, CM::IDX_D2);
    constexpr unsigned CB = ((1u << CM::NC) - 1u), KB = ((1u << CM::NK) - 1u) << CM::NC;
    is_contact = RO::lane_flag(CB);
    is_cone = RO::lane_flag(KB);
    is_role = RO::lane_flag(CB | KB);
    has_partner = RO::lane_flag(CM::PARTNER_BITS);
#pragma unroll
    for (int r = 0; r < NR; ++r) role[r] = RO::lane_flag(1u << r);
#pragma unroll
    for (int j = 0; j < CM::NQ; ++j) { jfa[j] = RO::lane_table(CM::JFA[j]); jfb[j] = RO::lane_table(CM::JFB[j]); }
    c_s = RO::sel(is_contact, 1.0, 0.0);
    c_va = RO::lane_table(CM::CVA);
    c_vb = RO::lane_table(CM::CVB);
    c_psi = RO::sel(is_cone, 1.0, 0.0);
    reg_floor = RO::sel(is_contact, 0.0, -__builtin_inf());
    gcoef = V(0.0);
    gconst = V(0.0);
  }
  OD_HD void set_theta(const double* th) {
    if constexpr (CM::NK > 0) {
      double g[CM::NK], gc[CM::NK];
      CM::eval_gcoef(th, g, gc);
      gcoef = V(0.0);
      gconst = V(0.0);
#pragma unroll
      for (int c = 0; c < CM::NK; ++c) {
        gcoef = RO::sel(role[CM::NC + c], g[c], gcoef);
        gconst = RO::sel(role[CM::NC + c], gc[c], gconst);
      }
    }
  }
};

// factors of one KKT matrix
template <class CM, class RO> struct C3Fact {
  using V = typename RO::V;
  using B = typename RO::B;
  static constexpr int NQ = CM::NQ;
  // contacts: bilinear pivot 1/s (floored), clamped gamma, t[j] = gamma/s * JFa[j]
  V ipc, gc, t[NQ], JFa[NQ], JFb[NQ];
  // cones: psi-row / contact elimination, role swap, the 3 x 3 block's multipliers and pivots, d x_m = a_m - w_m . dq
  V gA, gB1, gB2, ip1, ip2, ip3, l21, l31, l32, m12, m13, m23, w1[NQ], w2[NQ], w3[NQ];
  B sw;
  // replicated: couplings of the dynamics rows to gamma_i / b_{c,1|2}, LU of the nq x nq Schur complement
  double nv[CM::NC > 0 ? CM::NC : 1][NQ], nbv[CM::NK > 0 ? CM::NK : 1][2][NQ];
  double lu[NQ * NQ];
  int piv[NQ];
};

// ---------------------------------------------------------------------------------------------------------------
// residual r(z; theta, 0)
// ---------------------------------------------------------------------------------------------------------------
template <class CM, class RO>
OD_HD void c3_eval_r(const C3Lanes<CM, RO>& L, const C3Vec<CM::NQ, typename RO::V>& z, const double* th, const double* pre,
                     double* tr, C3Res<CM::NQ, typename RO::V>& r) {
  using M = typename CM::M;
  using V = typename RO::V;
  // the lane-per-problem residual on a replicated z that holds q and the contact forces the dynamics rows read; s_i and
  // s_b are zero there, so its slack / velocity rows return the lanes' aux expressions.  Rows nobody reads are dead code.
  double zr[M::NZ], rr[M::NZ];
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) zr[i] = 0.0;
#pragma unroll
  for (int k = 0; k < CM::NQ; ++k) zr[CM::ZQ[k]] = z.q[k];
  CM::template gather_r<RO>(z.P0, z.P1, z.P2, zr);
  M::eval_r(zr, th, pre, tr, rr);
#pragma unroll
  for (int k = 0; k < CM::NQ; ++k) r.rd[k] = rr[CM::RDYN[k]];
  r.r1a = CM::template pick_e1a<RO>(L, rr) + L.c_s * z.D0 + L.c_va * z.D1;
  if constexpr (CM::DIM3) r.r1b = CM::template pick_e1b<RO>(L, rr) + L.c_vb * z.D2;
  else r.r1b = V(0.0);
  if constexpr (CM::NK > 0) {
    V gp = V(0.0);
    if constexpr (CM::SH > 0) gp = RO::sel(L.has_partner, L.gcoef * RO::template shr<CM::SH>(z.P0), 0.0);
    r.r2 = L.c_psi * z.P0 + gp + L.gconst;
  } else {
    r.r2 = V(0.0);
  }
  r.rA = z.P0 * z.D0 + z.P1 * z.D1;
  r.rB1 = z.P0 * z.D1 + z.P1 * z.D0;
  if constexpr (CM::DIM3) {
    r.rA = r.rA + z.P2 * z.D2;
    r.rB2 = z.P0 * z.D2 + z.P2 * z.D0;
  } else {
    r.rB2 = V(0.0);
  }
}

// max |r| over the equality rows and over the complementarity rows, NaN-sticky like od_solver.h::viol_eq / viol_bil
template <class CM, class RO>
OD_HD void c3_viol(const C3Lanes<CM, RO>& L, const C3Res<CM::NQ, typename RO::V>& r, double& r_vio, double& k_vio) {
  using V = typename RO::V;
  const double inf = __builtin_inf(), nan = __builtin_nan("");
  double ve = 0.0, se = 0.0;
#pragma unroll
  for (int k = 0; k < CM::NQ; ++k) { const double a = od_abs(r.rd[k]); ve = od_fmax(ve, a); se += a; }
  const V a1 = od_abs(r.r1a), a1b = od_abs(r.r1b), a2 = od_abs(r.r2), aA = od_abs(r.rA), aB1 = od_abs(r.rB1), aB2 = od_abs(r.rB2);
  V e = od_fmax(od_fmax(a1, a1b), a2);
  const V es = a1 + a1b + a2;
  e = RO::sel(es != es, inf, e);              // hardware max drops NaNs: carry them as +inf through the reduction
  e = RO::sel(L.is_role, e, 0.0);
  V c = od_fmax(od_fmax(aA, aB1), aB2);
  const V cs = aA + aB1 + aB2;
  c = RO::sel(cs != cs, inf, c);
  c = RO::sel(L.is_role, c, 0.0);
  double de = RO::first(grp_max<RO>(e));
  const double dk = RO::first(grp_max<RO>(c));
  de = od_fmax(de, ve);
  r_vio = (se != se || de == inf) ? nan : de;
  k_vio = (dk == inf) ? nan : dk;
}

// ---------------------------------------------------------------------------------------------------------------
// Jacobian + factorisation, in the order of the generated static elimination (gen/<model>.h, state program):
//   slack rows -> s, psi rows -> psi, velocity rows -> s_b, bilinear rows -> gamma (pivot s, floored),
//   cone: role swap, pivots (B1 -> b1, B2 -> b2, A -> s_psi) resp. (A -> s_psi, B2 -> b2, B1 -> b1),
//   then the nq x nq Schur complement (LU, replicated).
// ---------------------------------------------------------------------------------------------------------------
template <class CM, bool PIV, class RO>
OD_HD bool c3_eval_factor(const C3Lanes<CM, RO>& L, const C3Vec<CM::NQ, typename RO::V>& z, const double* th, const double* pre,
                          const double* tr, double reg, C3Fact<CM, RO>& f) {
  using M = typename CM::M;
  using V = typename RO::V;
  constexpr int NQ = CM::NQ;
  const V regl = L.reg_floor + reg;            // reg on contact lanes, -inf elsewhere
  const V P0c = RO::vmax(z.P0, regl), D0c = RO::vmax(z.D0, regl);
  double zr[M::NZ], a[M::NNZ];
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) zr[i] = 0.0;
#pragma unroll
  for (int k = 0; k < NQ; ++k) zr[CM::ZQ[k]] = z.q[k];
  CM::template gather_rz<RO>(P0c, z.P1, z.P2, zr);
  M::eval_rz(zr, th, pre, tr, a);
  double dqq[NQ * NQ];
  CM::dqq_from(a, dqq);
  CM::couplings(a, f.nv, f.nbv);
  CM::template build_jf<RO>(L, a, f.JFa, f.JFb);
  // ---- contacts:  d gamma = ty + t . dq
  f.gc = P0c;
  if constexpr (CM::NC > 0) {
    f.ipc = od_rcp(od_max(D0c, V(OD_PIVOT_FLOOR)));
    const V w = f.ipc * P0c;
#pragma unroll
    for (int j = 0; j < NQ; ++j) f.t[j] = CM::UPJ[j] ? w * f.JFa[j] : V(0.0);
  }
  // ---- cones
  V Wb1[NQ], Wb2[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) { Wb1[j] = V(0.0); Wb2[j] = V(0.0); }
  if constexpr (CM::NK > 0) {
    f.gA = -(z.D0 * L.gcoef);
    f.gB1 = -(z.D1 * L.gcoef);
    f.gB2 = -(z.D2 * L.gcoef);
    // rows after d psi and d s_b are substituted, unknowns (d s_psi, d b1, d b2):
    //   A : psi  d s_psi + s_b1 d b1 + s_b2 d b2 + qA  . dq = yA        qA  = b1 nJa + b2 nJb + gA  t'
    //   B1: b1   d s_psi + s_psi d b1             + qB1 . dq = yB1       qB1 = psi nJa          + gB1 t'
    //   B2: b2   d s_psi             + s_psi d b2 + qB2 . dq = yB2       qB2 = psi nJb          + gB2 t'
    // (nJ = -c_v JF: d s_b = c_v (r_v - JF . dq); t' = the partner contact's t).  Role swap of the serial program
    // (sw = |psi| > |s_psi|): rows (A, B2, B1) on columns (s_psi, b2, b1) if sw, rows (B1, B2, A) on (b1, b2, s_psi)
    // otherwise -- a permutation picked by selects, then one elimination for both.
    f.sw = od_abs(z.P0) > od_abs(z.D0);
    const V c1a = RO::sel(f.sw, z.P1, z.P0), c1b = RO::sel(f.sw, z.P2, 0.0), g1 = RO::sel(f.sw, f.gA, f.gB1);
    const V c3a = RO::sel(f.sw, z.P0, z.P1), c3b = RO::sel(f.sw, 0.0, z.P2), g3 = RO::sel(f.sw, f.gB1, f.gA);
    V Q1[NQ], Q2[NQ], Q3[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const V nJa = -(L.c_va * f.JFa[j]);
      Q1[j] = CM::UPV[j] ? c1a * nJa : V(0.0);
      Q3[j] = CM::UPV[j] ? c3a * nJa : V(0.0);
      Q2[j] = V(0.0);
      if constexpr (CM::DIM3) {
        if (CM::UPV[j]) {
          const V nJb = -(L.c_vb * f.JFb[j]);
          Q1[j] = Q1[j] + c1b * nJb;
          Q3[j] = Q3[j] + c3b * nJb;
          Q2[j] = z.P0 * nJb;
        }
      }
      if constexpr (CM::SH > 0) {
        if (CM::UPJ[j]) {
          // (the lanes of a cone without a partner contact read another lane's t: selected away, not multiplied by 0)
          const V tp = RO::sel(L.has_partner, RO::template shr<CM::SH>(f.t[j]), 0.0);
          Q1[j] = Q1[j] + g1 * tp;
          Q3[j] = Q3[j] + g3 * tp;
          if constexpr (CM::DIM3) Q2[j] = Q2[j] + f.gB2 * tp;
        }
      }
    }
    const V m11 = RO::sel(f.sw, z.P0, z.D0), m33 = RO::sel(f.sw, z.D0, z.P0);
    f.m13 = RO::sel(f.sw, z.D1, z.P1);
    const V m31 = RO::sel(f.sw, z.P1, z.D1);
    f.ip1 = od_rcp(m11);
    f.l31 = m31 * f.ip1;
    V m33e = m33 - f.l31 * f.m13;
    if constexpr (CM::DIM3) {
      f.m12 = RO::sel(f.sw, z.D2, 0.0);
      const V m21 = RO::sel(f.sw, z.P2, 0.0), m23 = RO::sel(f.sw, 0.0, z.P2), m32 = RO::sel(f.sw, 0.0, z.D2);
      f.l21 = m21 * f.ip1;
      const V m22e = z.D0 - f.l21 * f.m12;
      f.m23 = m23 - f.l21 * f.m13;
      const V m32e = m32 - f.l31 * f.m12;
      f.ip2 = od_rcp(m22e);
      f.l32 = m32e * f.ip2;
      m33e = m33e - f.l32 * f.m23;
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        Q2[j] = Q2[j] - f.l21 * Q1[j];
        Q3[j] = Q3[j] - f.l31 * Q1[j];
        Q3[j] = Q3[j] - f.l32 * Q2[j];
      }
    } else {
      f.m12 = V(0.0); f.m23 = V(0.0); f.l21 = V(0.0); f.l32 = V(0.0); f.ip2 = V(1.0);
#pragma unroll
      for (int j = 0; j < NQ; ++j) Q3[j] = Q3[j] - f.l31 * Q1[j];
    }
    f.ip3 = od_rcp(m33e);
    // back-substituted coupling vectors: x_m = a_m - w_m . dq
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      f.w3[j] = f.ip3 * Q3[j];
      if constexpr (CM::DIM3) {
        f.w2[j] = f.ip2 * (Q2[j] - f.m23 * f.w3[j]);
        f.w1[j] = f.ip1 * (Q1[j] - f.m12 * f.w2[j] - f.m13 * f.w3[j]);
      } else {
        f.w2[j] = V(0.0);
        f.w1[j] = f.ip1 * (Q1[j] - f.m13 * f.w3[j]);
      }
      Wb1[j] = RO::sel(f.sw, f.w3[j], f.w1[j]);     // d b1 = Wy1 - Wb1 . dq
      Wb2[j] = f.w2[j];                              // d b2 = Wy2 - Wb2 . dq
    }
  }
  // ---- Schur complement on the dynamics rows (replicated), then its LU
  CM::template schur<RO>(f, Wb1, Wb2, dqq);
#pragma unroll
  for (int i = 0; i < NQ * NQ; ++i) f.lu[i] = dqq[i];
  if constexpr (PIV) return od_lu_factor<double, NQ>(f.lu, f.piv);
  else return od_lu_factor_static<double, NQ>(f.lu);
}

// x = rz^{-1} r with the stored factors
template <class CM, bool PIV, class RO>
OD_HD void c3_solve(const C3Lanes<CM, RO>& L, const C3Fact<CM, RO>& f, const C3Vec<CM::NQ, typename RO::V>& z,
                    const C3Res<CM::NQ, typename RO::V>& r, C3Vec<CM::NQ, typename RO::V>& x) {
  using V = typename RO::V;
  constexpr int NQ = CM::NQ;
  double rd[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) rd[k] = r.rd[k];
  // forward: contacts
  const V y = r.rA - f.gc * r.r1a;
  V ty = V(0.0), Wy1 = V(0.0), Wy2 = V(0.0), a1 = V(0.0), a2 = V(0.0), a3 = V(0.0);
  if constexpr (CM::NC > 0) ty = f.ipc * y;
  // forward: cones
  if constexpr (CM::NK > 0) {
    const V nra = L.c_va * r.r1a, nrb = L.c_vb * r.r1b;
    V yA = r.rA - z.D0 * r.r2 - z.P1 * nra;
    V yB1 = r.rB1 - z.D1 * r.r2 - z.P0 * nra;
    V yB2 = V(0.0);
    if constexpr (CM::DIM3) {
      yA = yA - z.P2 * nrb;
      yB2 = r.rB2 - z.D2 * r.r2 - z.P0 * nrb;
    }
    if constexpr (CM::SH > 0) {
      const V typ = RO::sel(L.has_partner, RO::template shr<CM::SH>(ty), 0.0);
      yA = yA - f.gA * typ;
      yB1 = yB1 - f.gB1 * typ;
      if constexpr (CM::DIM3) yB2 = yB2 - f.gB2 * typ;
    }
    const V Y1 = RO::sel(f.sw, yA, yB1);
    V Y3 = RO::sel(f.sw, yB1, yA) - f.l31 * Y1;
    if constexpr (CM::DIM3) {
      const V Y2 = yB2 - f.l21 * Y1;
      Y3 = Y3 - f.l32 * Y2;
      a3 = f.ip3 * Y3;
      a2 = f.ip2 * (Y2 - f.m23 * a3);
      a1 = f.ip1 * (Y1 - f.m12 * a2 - f.m13 * a3);
    } else {
      a3 = f.ip3 * Y3;
      a1 = f.ip1 * (Y1 - f.m13 * a3);
    }
    Wy1 = RO::sel(f.sw, a3, a1);
    Wy2 = a2;
  }
  CM::template rhs_update<RO>(f, ty, Wy1, Wy2, rd);
  if constexpr (PIV) od_lu_solve<double, NQ>(f.lu, f.piv, rd);
  else od_lu_solve_static<double, NQ>(f.lu, rd);
#pragma unroll
  for (int k = 0; k < NQ; ++k) x.q[k] = rd[k];
  // back substitution inside the lanes
  V dg = ty, ea = r.r1a, eb = r.r1b, x1 = a1, x2 = a2, x3 = a3;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    if (CM::UPJ[j]) dg = dg + f.t[j] * rd[j];
    if (CM::UPJ[j] || CM::UPV[j]) ea = ea - f.JFa[j] * rd[j];
    if constexpr (CM::NK > 0) {
      x1 = x1 - f.w1[j] * rd[j];
      x3 = x3 - f.w3[j] * rd[j];
      if constexpr (CM::DIM3) {
        if (CM::UPV[j]) eb = eb - f.JFb[j] * rd[j];
        x2 = x2 - f.w2[j] * rd[j];
      }
    }
  }
  if constexpr (CM::NK > 0) {
    const V db1 = RO::sel(f.sw, x3, x1), dsp = RO::sel(f.sw, x1, x3);
    V dgp = V(0.0);
    if constexpr (CM::SH > 0) dgp = RO::sel(L.has_partner, L.gcoef * RO::template shr<CM::SH>(dg), 0.0);
    const V dpsi = r.r2 - dgp;
    x.P0 = RO::sel(L.is_cone, dpsi, dg);
    // b, s_b exist on cone lanes only; contact lanes keep exact zeros (a select, not a multiplication by 0: the cone
    // arithmetic of a contact lane may overflow when gamma or s underflow, and 0 * inf would poison its rows)
    x.P1 = RO::sel(L.is_cone, db1, 0.0);
    x.D0 = RO::sel(L.is_cone, dsp, ea);
    x.D1 = L.c_va * ea;                          // c_v is 0 off the cone lanes, ea their finite slack direction
    if constexpr (CM::DIM3) {
      x.P2 = RO::sel(L.is_cone, x2, 0.0);
      x.D2 = L.c_vb * eb;
    } else {
      x.P2 = V(0.0); x.D2 = V(0.0);
    }
  } else {
    x.P0 = dg; x.P1 = V(0.0); x.P2 = V(0.0); x.D0 = ea; x.D1 = V(0.0); x.D2 = V(0.0);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// step length, centering
// ---------------------------------------------------------------------------------------------------------------
// CVXOPT sec. 8.2 step for one member (l0, l1, l2) of a cone along -(d0, d1, d2) (od_solver.h::soc_step_one), lane-parallel;
// what depends on the member alone is computed once per iterate (StepPre) and serves predictor and corrector
template <class RO> struct C3StepPre { typename RO::V isq, ill, rc1; };

template <class CM, class RO>
OD_HD C3StepPre<RO> c3_step_pre(typename RO::V l0, typename RO::V l1, typename RO::V l2) {
  using V = typename RO::V;
  C3StepPre<RO> p;
  V ll = l0 * l0;
  ll = ll - l1 * l1;
  if constexpr (CM::DIM3) ll = ll - l2 * l2;
  ll = od_fmax(ll, V(1e-25)) + 1e-14;
  p.isq = od_rsqrt(ll);
  p.ill = p.isq * p.isq;
  p.rc1 = od_rcp(l0 * p.isq + 1.0);
  return p;
}

// -> tau / den where den > 0 (the caller caps at 1); direction of the member is -(d0, d1, d2)
template <class CM, class RO>
OD_HD typename RO::V c3_soc_step(const C3StepPre<RO>& p, typename RO::V l0, typename RO::V l1, typename RO::V l2,
                                 typename RO::V d0, typename RO::V d1, typename RO::V d2, double tau, typename RO::V& den) {
  using V = typename RO::V;
  V ld = l0 * d0;
  ld = ld - l1 * d1;
  if constexpr (CM::DIM3) ld = ld - l2 * d2;
  ld = ld + 1e-14;
  const V rs = ld * p.ill;
  const V c = (ld * p.isq + d0) * p.rc1;
  const V cl = c * p.ill;
  const V v1 = d1 * p.isq - cl * l1;
  V nv;
  if constexpr (CM::DIM3) {
    const V v2 = d2 * p.isq - cl * l2;
    V n2 = v1 * v1;
    n2 = n2 + v2 * v2;
    nv = od_sqrt(n2);
  } else {
    nv = od_abs(v1);
  }
  den = nv - rs;
  return tau * od_rcp(den);
}

template <class CM, class RO> struct C3Pre { C3StepPre<RO> p, d; };

template <class CM, class RO>
OD_HD double c3_step_length(const C3Lanes<CM, RO>& L, const C3Pre<CM, RO>& sp, const C3Vec<CM::NQ, typename RO::V>& z,
                            const C3Vec<CM::NQ, typename RO::V>& d, double tau_ort, double tau_soc) {
  using V = typename RO::V;
  V a = V(1.0);
  if constexpr (CM::NC > 0) {
    // alpha <= tau * z / d  where d > 0, for gamma and for s
    a = RO::sel(L.is_contact && (d.P0 > 0.0), (tau_ort * z.P0) * od_rcp(d.P0), a);
    a = od_fmin(a, RO::sel(L.is_contact && (d.D0 > 0.0), (tau_ort * z.D0) * od_rcp(d.D0), 1.0));
  }
  if constexpr (CM::NK > 0) {
    V den;
    const V ap = c3_soc_step<CM, RO>(sp.p, z.P0, z.P1, z.P2, -d.P0, -d.P1, -d.P2, tau_soc, den);
    a = RO::sel(L.is_cone && (den > 0.0), ap, a);
    const V ad = c3_soc_step<CM, RO>(sp.d, z.D0, z.D1, z.D2, -d.D0, -d.D1, -d.D2, tau_soc, den);
    a = od_fmin(a, RO::sel(L.is_cone && (den > 0.0), ad, 1.0));
  }
  a = od_fmin(a, V(1.0));
  return RO::first(grp_min<RO>(a));
}

// CVXOPT sec. 5.1.3: mu = <primal, dual>/ncones ; sigma = clamp(mu_aff/mu, 0, 1)^3  (od_solver.h::centering_kappa)
template <class CM, class RO>
OD_HD double c3_centering(const C3Lanes<CM, RO>& L, const C3Vec<CM::NQ, typename RO::V>& z, const C3Vec<CM::NQ, typename RO::V>& d, double aaff) {
  using V = typename RO::V;
  constexpr int n = CM::NC + CM::NK;
  V p = z.P0 * z.D0 + z.P1 * z.D1;
  V pa = (z.P0 - aaff * d.P0) * (z.D0 - aaff * d.D0) + (z.P1 - aaff * d.P1) * (z.D1 - aaff * d.D1);
  if constexpr (CM::DIM3) {
    p = p + z.P2 * z.D2;
    pa = pa + (z.P2 - aaff * d.P2) * (z.D2 - aaff * d.D2);
  }
  const double s = RO::first(grp_sum<RO>(RO::sel(L.is_role, p, 0.0)));
  const double sa = RO::first(grp_sum<RO>(RO::sel(L.is_role, pa, 0.0)));
  const double mu = s * (1.0 / n);
  double q = sa * od_rcp(s);
  q = od_fmax(q, 0.0);
  q = od_fmin(q, 1.0);
  return q * q * q * mu;
}

// ---------------------------------------------------------------------------------------------------------------
// Eight line-search trials at once (as od_coop.h::coop_trials_lanes does sixteen): lane g of the group evaluates
// r(z - a_g D; theta, 0) for ITS step a_g = alpha 2^-g -- the whole residual in one lane, every contact and cone in turn,
// with the arithmetic of c3_eval_r / c3_viol, so that a trial is accepted here exactly when the cooperative evaluation of
// the same step accepts it.
// ---------------------------------------------------------------------------------------------------------------
template <class CM, class RO, int R = 0>
OD_HD void c3_fetch_roles(const C3Vec<CM::NQ, typename RO::V>& z, typename RO::V (*P)[CM::NC + CM::NK > 0 ? CM::NC + CM::NK : 1]) {
  if constexpr (R < CM::NC + CM::NK) {
    using V = typename RO::V;
    P[0][R] = V(RO::template bc<R>(z.P0)); P[1][R] = V(RO::template bc<R>(z.P1)); P[2][R] = V(RO::template bc<R>(z.P2));
    P[3][R] = V(RO::template bc<R>(z.D0)); P[4][R] = V(RO::template bc<R>(z.D1)); P[5][R] = V(RO::template bc<R>(z.D2));
    c3_fetch_roles<CM, RO, R + 1>(z, P);
  }
}

template <class CM, class RO>
OD_HD typename RO::B c3_trials_lanes(const C3Lanes<CM, RO>& L, const double* th, const double* pre, const C3Vec<CM::NQ, typename RO::V>& z,
                                     const C3Vec<CM::NQ, typename RO::V>& D, typename RO::V aj, double r_vio, double k_vio) {
  using M = typename CM::M;
  using V = typename RO::V;
  constexpr int NQ = CM::NQ, NR = (CM::NC + CM::NK) > 0 ? (CM::NC + CM::NK) : 1;
  const double inf = __builtin_inf();
  V Z[6][NR], DD[6][NR];
  c3_fetch_roles<CM, RO>(z, Z);
  c3_fetch_roles<CM, RO>(D, DD);
  V zr[M::NZ], rr[M::NZ], thv[M::NTH], prev[M::NPRE], trv[M::NTR];
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) zr[i] = V(0.0);
#pragma unroll
  for (int k = 0; k < NQ; ++k) zr[CM::ZQ[k]] = V(z.q[k]) - aj * V(D.q[k]);
#pragma unroll
  for (int f = 0; f < 6; ++f) {
#pragma unroll
    for (int i = 0; i < NR; ++i) Z[f][i] = Z[f][i] - aj * DD[f][i];
  }
  CM::scatter_r(Z[0], Z[1], Z[2], zr);
#pragma unroll
  for (int i = 0; i < M::NTH; ++i) thv[i] = V(th[i]);
#pragma unroll
  for (int i = 0; i < M::NPRE; ++i) prev[i] = V(pre[i]);
  M::eval_r(zr, thv, prev, trv, rr);
  V ve = V(0.0), se = V(0.0);
#pragma unroll
  for (int k = 0; k < NQ; ++k) { const V a = od_abs(rr[CM::RDYN[k]]); ve = od_fmax(ve, a); se = se + a; }
  double g[CM::NK > 0 ? CM::NK : 1], gc[CM::NK > 0 ? CM::NK : 1];
  if constexpr (CM::NK > 0) CM::eval_gcoef(th, g, gc);
  V de = V(0.0), dk = V(0.0);
#pragma unroll
  for (int i = 0; i < CM::NC + CM::NK; ++i) {
    const bool cone = i >= CM::NC;
    const V &P0 = Z[0][i], &P1 = Z[1][i], &P2 = Z[2][i], &D0 = Z[3][i], &D1 = Z[4][i], &D2 = Z[5][i];
    const double c_s = RO::opaque(cone ? 0.0 : 1.0), c_va = RO::opaque(CM::CVA[i]), c_vb = RO::opaque(CM::CVB[i]);
    const V r1a = rr[CM::E1ROWA[i]] + c_s * D0 + c_va * D1;
    V r1b = V(0.0);
    if constexpr (CM::DIM3) {
      if (CM::E1ROWB[i] >= 0) r1b = rr[CM::E1ROWB[i] >= 0 ? CM::E1ROWB[i] : 0] + c_vb * D2;
      else r1b = V(0.0) + c_vb * D2;
    }
    V r2 = V(0.0);
    if constexpr (CM::NK > 0) {
      if (cone) {
        const int c = i - CM::NC;
        const double c_psi = RO::opaque(1.0);
        V gp = V(0.0);
        if (CM::SH > 0 && ((CM::PARTNER_BITS >> i) & 1u)) gp = V(g[c]) * Z[0][i - CM::SH >= 0 ? i - CM::SH : 0];
        r2 = c_psi * P0 + gp + V(gc[c]);
      }
    }
    V rA = P0 * D0 + P1 * D1;
    const V rB1 = P0 * D1 + P1 * D0;
    V rB2 = V(0.0);
    if constexpr (CM::DIM3) {
      rA = rA + P2 * D2;
      rB2 = P0 * D2 + P2 * D0;
    }
    const V a1 = od_abs(r1a), a1b = od_abs(r1b), a2 = od_abs(r2), aA = od_abs(rA), aB1 = od_abs(rB1), aB2 = od_abs(rB2);
    V e = od_fmax(od_fmax(a1, a1b), a2);
    const V es = a1 + a1b + a2;
    e = RO::sel(es != es, inf, e);
    V k = od_fmax(od_fmax(aA, aB1), aB2);
    const V ks = aA + aB1 + aB2;
    k = RO::sel(ks != ks, inf, k);
    de = RO::vmax(de, e);
    dk = RO::vmax(dk, k);
  }
  de = od_fmax(de, ve);
  const typename RO::B r_nan = (se != se) || (de == inf), k_nan = (dk == inf);
  const typename RO::B r_ok = (de <= r_vio) && !r_nan, k_ok = (dk <= k_vio) && !k_nan;
  return r_ok || k_ok;
}

// ---------------------------------------------------------------------------------------------------------------
// one predictor-corrector iteration (od_solver.h::ip_iteration), line search included
// ---------------------------------------------------------------------------------------------------------------
template <class CM, class RO>
OD_HD bool c3_iteration(const C3Lanes<CM, RO>& L, const Opts<double>& o, const double* th, const double* pre, double* tr,
                        C3Vec<CM::NQ, typename RO::V>& z, C3Res<CM::NQ, typename RO::V>& r, double& r_vio, double& k_vio,
                        double& reg_prev, int& status, C3Fact<CM, RO>& f, int& ls_hint) {
  using V = typename RO::V;
  using Vec = C3Vec<CM::NQ, V>;
  using Res = C3Res<CM::NQ, V>;
  constexpr int NQ = CM::NQ;
  constexpr bool CONES = (CM::NC + CM::NK) > 0;
  constexpr bool PIV = CM::M::STATE_TAIL_PIVOT;
  const double reg = (k_vio < o.kappa_reg) ? k_vio * o.gamma_reg : 0.0;
  reg_prev = reg;
  if (!c3_eval_factor<CM, PIV, RO>(L, z, th, pre, tr, reg, f)) status &= ~OD_ST_FACTOR_OK;
  Vec D;
  c3_solve<CM, PIV, RO>(L, f, z, r, D);
  C3Pre<CM, RO> sp;
  if constexpr (CM::NK > 0) {
    sp.p = c3_step_pre<CM, RO>(z.P0, z.P1, z.P2);
    sp.d = c3_step_pre<CM, RO>(z.D0, z.D1, z.D2);
  }
  if constexpr (CONES) {
    const double aaff = c3_step_length<CM, RO>(L, sp, z, D, 1.0, 1.0);
    double kap = c3_centering<CM, RO>(L, z, D, aaff);
    kap = od_fmax(kap, o.kappa_eval * o.undercut_inv);
    // r(z; kappa) from r(z; 0) on the head rows, then the second-order correction of the predictor
    r.rA = r.rA - kap + (D.P0 * D.D0 + D.P1 * D.D1);
    r.rB1 = r.rB1 + (D.P0 * D.D1 + D.P1 * D.D0);
    if constexpr (CM::DIM3) {
      r.rA = r.rA + D.P2 * D.D2;
      r.rB2 = r.rB2 + (D.P0 * D.D2 + D.P2 * D.D0);
    }
    c3_solve<CM, PIV, RO>(L, f, z, r, D);
  }
  const double vio = od_fmax(r_vio, k_vio);
  const double tau = 1.0 - od_fmin(o.eps_min, vio * vio);
  double alpha = c3_step_length<CM, RO>(L, sp, z, D, tau, od_fmin(tau, 0.99));
  // backtracking until either violation does not increase (od_solver.h::line_search), every trial in turn.  The lane-parallel
  // rounds of od_coop.h::coop_iteration exist as a measurement variant (tools/variants/c3_parallel_ls.patch: the 8 lanes of the group each
  // try a step size, c3_trials_lanes): bit-identical results (profiles/r3_hash_c3_parallel_ls.json), planar push unchanged,
  // hopper rollouts 8 % (8192) to 48 % (>= 16 384, spills at two wavefronts per SIMD) slower -- profiles/r3_c3_parallel_ls_ab.json
  Vec zc;                                    // (set by the first trial: max_ls >= 1 is enforced at the API)
  Res rc;
  double r_c, k_c;
  auto trial = [&](double a) {
#pragma unroll
    for (int k = 0; k < NQ; ++k) zc.q[k] = z.q[k] - a * D.q[k];
    zc.P0 = z.P0 - a * D.P0; zc.P1 = z.P1 - a * D.P1; zc.P2 = z.P2 - a * D.P2;
    zc.D0 = z.D0 - a * D.D0; zc.D1 = z.D1 - a * D.D1; zc.D2 = z.D2 - a * D.D2;
    c3_eval_r<CM, RO>(L, zc, th, pre, tr, rc);
    c3_viol<CM, RO>(L, rc, r_c, k_c);
    return r_c <= r_vio || k_c <= k_vio;
  };
  const int nseq = o.max_ls;                 // shipped: every trial in turn (the lane-parallel rounds measured slower here)
  bool done = false;
  int ls = 0;
  for (; ls < nseq; ++ls) {
    if (trial(alpha)) { done = true; break; }
    if (ls + 1 < o.max_ls) alpha *= 0.5;
  }
  // (a zero step whose trial reproduced both violations bit for bit: every further iteration would repeat this one)
  const bool fixed_point = (alpha == 0.0) && done && ls == 0 && r_c == r_vio && k_c == k_vio;
  ls_hint = ls;
  z = zc;
  r = rc;
  r_vio = r_c;
  k_vio = k_c;
  return fixed_point;
}

// ---------------------------------------------------------------------------------------------------------------
// the solve of one knot (od_solver.h::ip_step_grad with a deferred gradient): on return z.q is the configuration
// at (r_tol, kappa_eval); `defer(z, reg)` is called at the first iterate satisfying (r_tol, kappa_grad)
// ---------------------------------------------------------------------------------------------------------------
template <class CM, class RO, class Defer>
OD_HD int c3_ip_step(const C3Lanes<CM, RO>& L, const Opts<double>& o, const double* th, C3Vec<CM::NQ, typename RO::V>& z, bool want_grad,
                     Defer& defer, int* iters) {
  using M = typename CM::M;
  using V = typename RO::V;
  constexpr int NQ = CM::NQ;
  double pre[M::NPRE], tr[M::NTR], qs[NQ];
  C3Res<NQ, V> r;
  C3Fact<CM, RO> f;
  M::eval_pre(th, pre);
  c3_eval_r<CM, RO>(L, z, th, pre, tr, r);
  double r_vio, k_vio;
  c3_viol<CM, RO>(L, r, r_vio, k_vio);
  bool eval_done = false, grad_done = !want_grad;
  int status = OD_ST_FACTOR_OK;
  double reg_prev = 0.0;
  int ls_hint = 0;
  iters[0] = iters[1] = 0;
  for (int it = 0;; ++it) {
    const bool req = r_vio < o.r_tol;
    const bool last = it >= o.max_iter;
    if (!grad_done && ((req && k_vio < o.kappa_grad) || last)) {
      defer(z, od_max(reg_prev, o.kappa_grad * o.gamma_reg));
      grad_done = true;
      iters[1] = it;
      if (!last) status |= OD_ST_GRAD_OK;
    }
    if (!eval_done && ((req && k_vio < o.kappa_eval) || last)) {
#pragma unroll
      for (int k = 0; k < NQ; ++k) qs[k] = z.q[k];
      eval_done = true;
      iters[0] = it;
      if (!last) status |= OD_ST_EVAL_OK;
    }
    if (eval_done && grad_done) break;
    if (c3_iteration<CM, RO>(L, o, th, pre, tr, z, r, r_vio, k_vio, reg_prev, status, f, ls_hint) && it + 1 < o.max_iter) it = o.max_iter - 1;
  }
#pragma unroll
  for (int k = 0; k < NQ; ++k) z.q[k] = qs[k];
  return status;
}

// ---------------------------------------------------------------------------------------------------------------
// units of work: one knot / one rollout / one bundle sample per group.  Same outputs as od_units.h.
// ---------------------------------------------------------------------------------------------------------------
template <class CM, class RO> struct C3Defer {
  using V = typename RO::V;
  const View<double>& zg;
  long k;
  typename RO::I ip, id;
  OD_HD void operator()(const C3Vec<CM::NQ, V>& z, double reg) const {
    if (!zg.ok()) return;
    RO::template store<0>(zg, ip, k, z.P0);
    RO::template store<1>(zg, ip, k, z.P1);
    RO::template store<0>(zg, id, k, z.D0);
    RO::template store<1>(zg, id, k, z.D1);
    if constexpr (CM::DIM3) {
      RO::template store<2>(zg, ip, k, z.P2);
      RO::template store<2>(zg, id, k, z.D2);
    }
    if (RO::first_lane()) {
#pragma unroll
      for (int i = 0; i < CM::NQ; ++i) zg.at(CM::ZQ[i], k) = z.q[i];
      zg.at(CM::M::NZ, k) = reg;
    }
  }
};
template <class CM, class RO> struct C3NoDefer {
  OD_HD void operator()(const C3Vec<CM::NQ, typename RO::V>&, double) const {}
};

template <class CM, class RO> OD_HD void c3_init_z(const double* z0, C3Vec<CM::NQ, typename RO::V>& z) {
#pragma unroll
  for (int i = 0; i < CM::NQ; ++i) z.q[i] = z0[CM::ZQ[i]];
  z.P0 = RO::lane_table(CM::ZI_P0); z.P1 = RO::lane_table(CM::ZI_P1); z.P2 = RO::lane_table(CM::ZI_P2);
  z.D0 = RO::lane_table(CM::ZI_D0); z.D1 = RO::lane_table(CM::ZI_D1); z.D2 = RO::lane_table(CM::ZI_D2);
}

struct C3NoHook { OD_HD void operator()() const {} };
template <class CM, class RO, class Hook = C3NoHook>
OD_HD int c3_knot_state(const C3Lanes<CM, RO>& L0, const StepArgs<double>& a, long k, const double* xin, const double* uin, double* q3out,
                        const Hook& before_stores = Hook()) {
  using M = typename CM::M;
  using V = typename RO::V;
  constexpr int nq = M::NQ;
  double th[M::NTH], z0[M::NZ];
  mech_setup<M>(xin, xin + nq, uin, a.fric, a.h, th, z0);
  C3Lanes<CM, RO> L = L0;
  L.set_theta(th);
  C3Vec<nq, V> z;
  c3_init_z<CM, RO>(z0, z);
  C3Defer<CM, RO> defer{a.zg, k, L0.zg_p, L0.zg_d};
  int it[2];
  const int st = c3_ip_step<CM, RO>(L, a.opts, th, z, a.want_grad != 0, defer, it);
#pragma unroll
  for (int i = 0; i < nq; ++i) q3out[i] = z.q[i];
  before_stores();
  if (RO::first_lane()) {
    if (a.d.ok()) {
      auto c = a.d.cursor(k);
#pragma unroll
      for (int i = 0; i < nq; ++i) c.put(xin[nq + i]);
#pragma unroll
      for (int i = 0; i < nq; ++i) c.put(q3out[i]);
    }
    if (a.q3.ok()) {
      auto c = a.q3.cursor(k);
#pragma unroll
      for (int i = 0; i < nq; ++i) c.put(q3out[i]);
    }
    if (a.merge_grad_status) {   // the separate grad solve of a non-fusable step (od_units.h::knot_state)
      if (a.status.ok()) { const int e = a.status.at(0, k); a.status.at(0, k) = (e & ~OD_ST_FACTOR_OK) | (st & OD_ST_GRAD_OK) | (e & st & OD_ST_FACTOR_OK); }
      if (a.iters.ok()) { auto c = a.iters.cursor(k); c.skip(1); c.put(it[1]); }
    } else {
      if (a.status.ok()) a.status.at(0, k) = st;
      if (a.iters.ok()) { auto c = a.iters.cursor(k); c.put(it[0]); c.put(it[1]); }
    }
  }
  return st;
}

template <class CM, class RO> OD_HD void c3_unit_step_state(const StepArgs<double>& a, long b) {
  using M = typename CM::M;
  constexpr int nq = M::NQ, n = 2 * M::NQ;
  double x[n], u[M::NU > 0 ? M::NU : 1], q3[nq];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = a.x.at(i, b);
#pragma unroll
  for (int i = 0; i < M::NU; ++i) u[i] = a.u.at(i, b);
  C3Lanes<CM, RO> L;
  L.init();
  c3_knot_state<CM, RO>(L, a, b, x, u, q3);
}

template <class CM, class RO> OD_HD void c3_unit_rollout_state(const RolloutArgs<double>& ra, long b) {
  using M = typename CM::M;
  constexpr int nq = M::NQ, n = 2 * M::NQ;
  const StepArgs<double>& a = ra.s;
  double x[n], u[M::NU > 0 ? M::NU : 1], un[M::NU > 0 ? M::NU : 1], q3[nq];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = a.x.at(i, b);
  if (ra.x0.ok() && RO::first_lane()) {
#pragma unroll
    for (int i = 0; i < n; ++i) ra.x0.at(i, b) = x[i];
  }
  C3Lanes<CM, RO> L;
  L.init();
#pragma unroll
  for (int i = 0; i < M::NU; ++i) un[i] = a.u.at(i, b);
  for (int t = 0; t < ra.Tn; ++t) {
    const long k = (long)t * a.B + b;
#pragma unroll
    for (int i = 0; i < M::NU; ++i) u[i] = un[i];
    if (t + 1 < ra.Tn) {
#pragma unroll
      for (int i = 0; i < M::NU; ++i) un[i] = a.u.at(i, k + a.B);
    }
    c3_knot_state<CM, RO>(L, a, k, x, u, q3, [&]() {
#pragma unroll
      for (int i = 0; i < M::NU; ++i) RO::arrived(un[i]);
    });
#pragma unroll
    for (int i = 0; i < nq; ++i) { x[i] = x[nq + i]; x[nq + i] = q3[i]; }
  }
}

// closed-loop rollout = forward pass of iLQR (od_units.h::unit_rollout_policy), one candidate per group
template <class CM, class RO> OD_HD void c3_unit_rollout_policy(const PolicyArgs<double>& pa, long p) {
  using M = typename CM::M;
  constexpr int nq = M::NQ, n = 2 * M::NQ, nu = M::NU > 0 ? M::NU : 1;
  const StepArgs<double>& a = pa.r.s;
  const long b = p % pa.Bnom;
  const double alpha = pa.alphas[p / pa.Bnom];
  double x[n], u[nu], q3[nq];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = a.x.at(i, b);
  if (pa.r.x0.ok() && RO::first_lane()) {
#pragma unroll
    for (int i = 0; i < n; ++i) pa.r.x0.at(i, p) = x[i];
  }
  C3Lanes<CM, RO> L;
  L.init();
  for (int t = 0; t < pa.r.Tn; ++t) {
    const long kn = (long)t * pa.Bnom + b, kc = (long)t * a.B + p;
    double dx[n];
#pragma unroll
    for (int i = 0; i < n; ++i) dx[i] = x[i] - pa.xbar.at(i, kn);
#pragma unroll
    for (int j = 0; j < M::NU; ++j) u[j] = pa.ubar.at(j, kn) + alpha * pa.kff.at(j, kn);
#pragma unroll
    for (int i = 0; i < n; ++i) {            // K is nu x n col-major: column i multiplies dx[i]
#pragma unroll
      for (int j = 0; j < M::NU; ++j) u[j] += pa.K.at(j + M::NU * i, kn) * dx[i];
    }
    if (RO::first_lane()) {
#pragma unroll
      for (int j = 0; j < M::NU; ++j) pa.U.at(j, kc) = u[j];
    }
    const int st = c3_knot_state<CM, RO>(L, a, kc, x, u, q3);
    if (pa.stop_failed && !(st & OD_ST_EVAL_OK)) {          // (PolicyArgs::stop_failed; the whole group leaves)
      if (RO::first_lane()) policy_mark_rest_failed(pa, t, p);
      break;
    }
#pragma unroll
    for (int i = 0; i < nq; ++i) { x[i] = x[nq + i]; x[nq + i] = q3[i]; }
  }
}

// gradient-bundle sample (od_units.h::unit_bundle_sample): problem p = b (N+1) + i, i = 0 nominal, i >= 1 perturbed by
// eta[:, i-1]; EVAL simulator, no gradient
template <class CM, class RO> OD_HD void c3_unit_bundle_sample(const BundleArgs<double>& ba, long p) {
  using M = typename CM::M;
  using V = typename RO::V;
  constexpr int nq = M::NQ, n = 2 * M::NQ, nzb = 2 * M::NQ + M::NU;
  const StepArgs<double>& a = ba.s;
  const long b = p / (ba.N + 1);
  const int i = (int)(p - b * (ba.N + 1));
  double x[n], u[M::NU > 0 ? M::NU : 1];
#pragma unroll
  for (int k = 0; k < n; ++k) x[k] = a.x.at(k, b) + (i > 0 ? ba.eta[k + nzb * (i - 1)] : 0.0);
#pragma unroll
  for (int k = 0; k < M::NU; ++k) u[k] = a.u.at(k, b) + (i > 0 ? ba.eta[n + k + nzb * (i - 1)] : 0.0);
  double th[M::NTH], z0[M::NZ];
  mech_setup<M>(x, x + nq, u, a.fric, a.h, th, z0);
  C3Lanes<CM, RO> L;
  L.init();
  L.set_theta(th);
  C3Vec<nq, V> z;
  c3_init_z<CM, RO>(z0, z);
  C3NoDefer<CM, RO> nd;
  int it[2];
  const int st = c3_ip_step<CM, RO>(L, a.opts, th, z, false, nd, it);
  if (RO::first_lane()) {
#pragma unroll
    for (int k = 0; k < nq; ++k) ba.feta.at(k, p) = z.q[k];
    if (ba.status.ok()) ba.status.at(0, p) = st;
  }
}

}  // namespace od
