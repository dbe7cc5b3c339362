// Cooperative interior-point solve: ONE PROBLEM PER DPP ROW (16 lanes), four problems per wavefront.
//
// The lane-per-problem solver (od_solver.h) runs the whole iteration of a problem in one lane; a rollout batch of
// 4096 trajectories then fills 4 of 64 lanes and its wall time is the latency of one trajectory.  Here the 16 lanes
// of a row SPLIT the iteration of one problem (reference: the per-knot solve behind src/dynamics.jl:81-94,
// RoboDojo.step! -> interior_point_solve!):
//
//   * every contact (orthant pair gamma_i, s_i with its slack and bilinear rows) and every friction cone
//     (psi, b, s_psi, s_b with its psi-, tangential-velocity- and two cone-product rows) lives in its own lane:
//     "D" (distributed) values, one register per lane whose meaning depends on the lane's role;
//   * the configuration q, the dynamics rows and their nq x nq Schur complement are "R" (replicated) values: every
//     lane of the row computes the same number;
//   * the static elimination of od_solver.h / gen/<model>.h becomes block algebra inside the lanes (all contacts at
//     once, all cones at once: the reciprocals, the cone role swap, the multipliers), the Schur complement is
//     assembled with v_fmac_f64_dpp row_newbcast (one instruction: R += D[lane L] * R), cone step lengths, ratio
//     tests, residual rows and the z update are one instruction stream for all contacts / cones;
//   * lanes 8..15 mirror lanes 0..7 (same roles, same numbers) so that the primal and dual halves of a cone's step
//     length, and the two violation norms, are computed side by side;
//   * reductions over the row: xor-butterflies with 32-bit DPP (every lane ends with bit-identical results, which
//     the row-uniform control flow relies on).
//
// Type discipline: R values are `double`; D values are `RO::V` -- `double` on the device (RowDev), a 16-lane vector
// in the host test build (RowEmu, tests/host_emu), where mixing the two up does not compile.
#pragma once
#include "od_units.h"

namespace od {

// ---------------------------------------------------------------------------------------------------------------
// host emulation of a 16-lane row (TEST HARNESS: only the host build instantiates it)
// ---------------------------------------------------------------------------------------------------------------
struct Mask16 { bool m[16]; };
struct Vec16 {
  double v[16];
  Vec16() = default;
  Vec16(double x) { for (int i = 0; i < 16; ++i) v[i] = x; }
};
#define OD_V16_BIN(op)                                                                                               \
  inline Vec16 operator op(const Vec16& a, const Vec16& b) { Vec16 r; for (int i = 0; i < 16; ++i) r.v[i] = a.v[i] op b.v[i]; return r; } \
  inline Vec16 operator op(const Vec16& a, double b) { Vec16 r; for (int i = 0; i < 16; ++i) r.v[i] = a.v[i] op b; return r; }              \
  inline Vec16 operator op(double a, const Vec16& b) { Vec16 r; for (int i = 0; i < 16; ++i) r.v[i] = a op b.v[i]; return r; }
OD_V16_BIN(+) OD_V16_BIN(-) OD_V16_BIN(*)
#undef OD_V16_BIN
inline Vec16 operator-(const Vec16& a) { Vec16 r; for (int i = 0; i < 16; ++i) r.v[i] = -a.v[i]; return r; }
inline Vec16& operator+=(Vec16& a, const Vec16& b) { a = a + b; return a; }
inline Vec16& operator-=(Vec16& a, const Vec16& b) { a = a - b; return a; }
inline Vec16& operator*=(Vec16& a, const Vec16& b) { a = a * b; return a; }
#define OD_V16_CMP(op)                                                                                               \
  inline Mask16 operator op(const Vec16& a, const Vec16& b) { Mask16 r; for (int i = 0; i < 16; ++i) r.m[i] = a.v[i] op b.v[i]; return r; } \
  inline Mask16 operator op(const Vec16& a, double b) { Mask16 r; for (int i = 0; i < 16; ++i) r.m[i] = a.v[i] op b; return r; }
OD_V16_CMP(>) OD_V16_CMP(<) OD_V16_CMP(!=) OD_V16_CMP(<=) OD_V16_CMP(==)
#undef OD_V16_CMP
inline Mask16 operator&&(const Mask16& a, const Mask16& b) { Mask16 r; for (int i = 0; i < 16; ++i) r.m[i] = a.m[i] && b.m[i]; return r; }
inline Mask16 operator!(const Mask16& a) { Mask16 r; for (int i = 0; i < 16; ++i) r.m[i] = !a.m[i]; return r; }
inline Mask16 operator||(const Mask16& a, const Mask16& b) { Mask16 r; for (int i = 0; i < 16; ++i) r.m[i] = a.m[i] || b.m[i]; return r; }
inline Vec16 od_rcp(const Vec16& a) { Vec16 r; for (int i = 0; i < 16; ++i) r.v[i] = od_rcp(a.v[i]); return r; }
inline Vec16 od_rsqrt(const Vec16& a) { Vec16 r; for (int i = 0; i < 16; ++i) r.v[i] = od_rsqrt(a.v[i]); return r; }
inline Vec16 od_abs(const Vec16& a) { Vec16 r; for (int i = 0; i < 16; ++i) r.v[i] = od_abs(a.v[i]); return r; }
inline Vec16 od_max(const Vec16& a, const Vec16& b) { Vec16 r; for (int i = 0; i < 16; ++i) r.v[i] = od_max(a.v[i], b.v[i]); return r; }
inline Vec16 od_min(const Vec16& a, const Vec16& b) { Vec16 r; for (int i = 0; i < 16; ++i) r.v[i] = od_min(a.v[i], b.v[i]); return r; }
inline Vec16 od_fmax(const Vec16& a, const Vec16& b) { Vec16 r; for (int i = 0; i < 16; ++i) r.v[i] = od_fmax(a.v[i], b.v[i]); return r; }
inline Vec16 od_fmin(const Vec16& a, const Vec16& b) { Vec16 r; for (int i = 0; i < 16; ++i) r.v[i] = od_fmin(a.v[i], b.v[i]); return r; }

struct RowEmu {
  using V = Vec16;
  using B = Mask16;
  static constexpr bool DEVICE = false;
  static V lane_table(const double (&t)[16]) { V r; for (int i = 0; i < 16; ++i) r.v[i] = t[i]; return r; }
  static B lane_flag(unsigned bits) { B r; for (int i = 0; i < 16; ++i) r.m[i] = (bits >> i) & 1u; return r; }
  static V sel(const B& m, const V& a, const V& b) { V r; for (int i = 0; i < 16; ++i) r.v[i] = m.m[i] ? a.v[i] : b.v[i]; return r; }
  static V sel(const B& m, double a, const V& b) { return sel(m, V(a), b); }
  static V sel(const B& m, const V& a, double b) { return sel(m, a, V(b)); }
  static V sel(const B& m, double a, double b) { return sel(m, V(a), V(b)); }
  template <int L> static double bc(const V& x) { return x.v[L]; }
  template <int L0, int L1> static void bc2(const V& x, double& a, double& b) { a = x.v[L0]; b = x.v[L1]; }
  template <int L> static void fmac(double& acc, const V& x, double m) { acc = od_fma(x.v[L], m, acc); }
  template <int L> static void fnmac(double& acc, const V& x, double m) { acc = od_fma(x.v[L], -m, acc); }
  template <int N> static V shr(const V& x) { V r = x; for (int i = N; i < 16; ++i) r.v[i] = x.v[i - N]; return r; }
  static V xor1(const V& x) { V r; for (int i = 0; i < 16; ++i) r.v[i] = x.v[i ^ 1]; return r; }
  static V xor2(const V& x) { V r; for (int i = 0; i < 16; ++i) r.v[i] = x.v[i ^ 2]; return r; }
  static V half_mirror(const V& x) { V r; for (int i = 0; i < 16; ++i) r.v[i] = x.v[(i & 8) | (7 - (i & 7))]; return r; }
  // four destination-row tables (one per lane vector; -1 = lane holds nothing to store) packed into one int per lane
  struct I { int v[16]; };
  static I lane_pack4(const int (&a)[16], const int (&b)[16], const int (&c)[16], const int (&d)[16]) {
    I r;
    for (int i = 0; i < 16; ++i) r.v[i] = (a[i] & 0xFF) | ((b[i] & 0xFF) << 8) | ((c[i] & 0xFF) << 16) | ((d[i] & 0xFF) << 24);
    return r;
  }
  template <int W, class View_> static void store(const View_& v, const I& idx, long k, const V& x) {
    for (int i = 0; i < 8; ++i) { const int j = (idx.v[i] >> (8 * W)) & 0xFF; if (j != 0xFF) v.at(j, k) = x.v[i]; }
  }
  static V vmax(const V& a, const V& b) { return od_fmax(a, b); }
  static V vmin(const V& a, const V& b) { return od_fmin(a, b); }
  static double vmax(double a, double b) { return od_fmax(a, b); }
  static double vmin(double a, double b) { return od_fmin(a, b); }
  static bool first_lane() { return true; }
  static void arrived(const double&) {}
  // parallel line-search trials: lane g of the row takes step alpha 2^-g
  static V lane_ldexp(double a) { V r; for (int i = 0; i < 16; ++i) r.v[i] = od_ldexp(a, -i); return r; }
  static B lane_below(int n) { B r; for (int i = 0; i < 16; ++i) r.m[i] = i < n; return r; }
  static unsigned row_ballot(const B& b) { unsigned m = 0; for (int i = 0; i < 16; ++i) m |= (b.m[i] ? 1u : 0u) << i; return m; }
  static double opaque(double x) { return x; }
};

#if defined(__HIP_DEVICE_COMPILE__)
// the device row: lanes 16p .. 16p+15 of a wavefront.  Cross-lane reads of a 64-bit value:
//   bc<L>     v_mov_b64_dpp  row_newbcast:L          (DP-ALU DPP supports row_newbcast only)
//   fmac<L>   v_fmac_f64_dpp row_newbcast:L          (VOP2; v_fma_f64 is VOP3 and has no DPP form on gfx9)
//   shr / xor / mirror: two v_mov_b32_dpp through the compiler builtin.
// The DPP read-after-VALU-write hazard is NOT interlocked on gfx950 (tools/ubench/dpp_hazard.hip: wrong values
// without the two wait states); the builtin path is covered by the compiler, the hand-written forms carry their own
// `s_nop 1`.
struct RowDev {
  using V = double;
  using B = bool;
  static constexpr bool DEVICE = true;
  __device__ __forceinline__ static int lane() { return (int)(threadIdx.x & 15); }
  __device__ __forceinline__ static V lane_table(const double (&t)[16]) { return t[lane()]; }
  __device__ __forceinline__ static B lane_flag(unsigned bits) { return (bits >> lane()) & 1u; }
  __device__ __forceinline__ static V sel(B m, V a, V b) { return m ? a : b; }
  template <int L> __device__ __forceinline__ static double bc(double x) {
    double r;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(L));
    return r;
  }
  template <int L0, int L1> __device__ __forceinline__ static void bc2(double x, double& a, double& b) {   // one nop for both
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
        : "=&v"(a), "=&v"(b) : "v"(x), "n"(L0), "n"(L1));
  }
  template <int L> __device__ __forceinline__ static void fmac(double& acc, double x, double m) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(L));
  }
  template <int L> __device__ __forceinline__ static void fnmac(double& acc, double x, double m) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(L));
  }
  template <int CTRL> __device__ __forceinline__ static double dpp32(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    // bound_ctrl: lanes whose source is outside the row read 0 (row_shr; nobody uses those lanes' results) -- without
    // it the destination is tied to an `old` value and the compiler copies the source first (two more moves)
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  }
  template <int N> __device__ __forceinline__ static double shr(double x) { return dpp32<0x110 + N>(x); }   // lane l <- lane l-N
  __device__ __forceinline__ static double xor1(double x) { return dpp32<0xB1>(x); }          // quad_perm [1,0,3,2]
  __device__ __forceinline__ static double xor2(double x) { return dpp32<0x4E>(x); }          // quad_perm [2,3,0,1]
  __device__ __forceinline__ static double half_mirror(double x) { return dpp32<0x141>(x); }  // lane l <-> 7-l within 8
  // the tables are read once per kernel (CoopLanes::init): a lookup per store is a dependent memory round trip per knot
  using I = int;
  __device__ __forceinline__ static I lane_pack4(const int (&a)[16], const int (&b)[16], const int (&c)[16], const int (&d)[16]) {
    const int l = lane();
    return (a[l] & 0xFF) | ((b[l] & 0xFF) << 8) | ((c[l] & 0xFF) << 16) | ((d[l] & 0xFF) << 24);
  }
  template <int W, class View_> __device__ __forceinline__ static void store(const View_& v, I idx, long k, double x) {
    const int i = (idx >> (8 * W)) & 0xFF;
    if (lane() < 8 && i != 0xFF) v.at(i, k) = x;
  }
  // max / min of values that came out of lane moves: the bare instruction (od_fmax would re-quiet both operands first)
  __device__ __forceinline__ static double vmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
  __device__ __forceinline__ static double vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
  __device__ __forceinline__ static bool first_lane() { return lane() == 0; }
  // the value of an earlier load is needed from here on (places the s_waitcnt)
  __device__ __forceinline__ static void arrived(const double& x) { asm volatile("" ::"v"(x)); }
  __device__ __forceinline__ static V lane_ldexp(double a) { return od_ldexp(a, -lane()); }
  __device__ __forceinline__ static B lane_below(int n) { return lane() < n; }
  // the 16 lanes' predicate bits of this row (under divergence: of the rows that execute)
  __device__ __forceinline__ static unsigned row_ballot(bool b) { return (unsigned)(__builtin_amdgcn_ballot_w64(b) >> (threadIdx.x & 48)) & 0xFFFFu; }
  // a constant the compiler must treat as a run-time value (keeps an expression's shape, hence its FMA contraction)
  __device__ __forceinline__ static double opaque(double x) { asm volatile("" : "+v"(x)); return x; }
};
#endif

// ---------------------------------------------------------------------------------------------------------------
// data of one problem, spread over its row
// ---------------------------------------------------------------------------------------------------------------
// z:  q replicated; per lane (P0, P1 | D0, D1) = primal | dual members:
//       contact i : P0 = gamma_i, D0 = s_i, P1 = D1 = 0          cone c : (P0, P1) = (psi, b), (D0, D1) = (s_psi, s_b)
//       lanes without a role hold (1, 0 | 1, 0) and are masked out of every reduction
// r:  dynamics rows replicated; per lane r1 (contact: slack row | cone: tangential-velocity row), r2 (cone: psi row),
//     rA (bilinear row | cone head row), rB (cone tail row)
template <int NQ, class V> struct CoopVec { double q[NQ]; V P0, P1, D0, D1; };
template <int NQ, class V> struct CoopRes { double rd[NQ]; V r1, r2, rA, rB; };

template <class CM, class RO> struct CoopLanes {
  using V = typename RO::V;
  using B = typename RO::B;
  B is_contact, is_cone, is_role, half;
  B role[CM::NC + CM::NK > 0 ? CM::NC + CM::NK : 1];
  V jfc[CM::NQ];          // constant entries of the lane's aux-row Jacobian (slack row | velocity row) w.r.t. q
  V c_s, c_v, c_psi;      // r1 = e1 + c_s*D0 + c_v*D1 (c_v = d(velocity row)/d s_b = +-1 on cone lanes) ;  r2 = c_psi*P0 + gcoef*gamma_partner + gconst
  V gcoef, gconst;        // psi row: d/d gamma_partner, theta-only constant (set per knot)
  V reg_floor;            // 0 on contact lanes, -inf elsewhere (the regularisation floor applies to orthant members)
  typename RO::I zg_idx;  // rows of z the lane's (P0, P1, D0, D1) go to in the gradient hand-over
  OD_HD void init() {
    zg_idx = RO::lane_pack4(CM::IDX_P0, CM::IDX_P1, CM::IDX_D0, CM::IDX_D1);
    constexpr unsigned CB = ((1u << CM::NC) - 1u), KB = ((1u << CM::NK) - 1u) << CM::NC;
    is_contact = RO::lane_flag(CB | (CB << 8));
    is_cone = RO::lane_flag(KB | (KB << 8));
    is_role = RO::lane_flag((CB | KB) | ((CB | KB) << 8));
    half = RO::lane_flag(0xFF00u);
#pragma unroll
    for (int r = 0; r < CM::NC + CM::NK; ++r) role[r] = RO::lane_flag((1u << r) | (1u << (r + 8)));
#pragma unroll
    for (int j = 0; j < CM::NQ; ++j) jfc[j] = RO::lane_table(CM::JFC[j]);
    c_s = RO::sel(is_contact, 1.0, 0.0);
    c_v = RO::lane_table(CM::CV);
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
template <class CM, class RO> struct CoopFact {
  using V = typename RO::V;
  using B = typename RO::B;
  static constexpr int NQ = CM::NQ;
  // contacts: bilinear pivot 1/s (floored), clamped gamma, t[j] = gamma/s * JF[j]
  V ipc, gc, t[NQ], JF[NQ];
  // cones: psi-row and contact elimination, role swap, the two scalar pivots
  V gA, gB, ip1, o1, l2, ip2, u1, u3, q1[NQ], q2[NQ];
  B sw;
  // replicated: couplings of the dynamics rows to gamma_i / b_c, LU of the nq x nq Schur complement
  double nv[CM::NC > 0 ? CM::NC : 1][NQ], nbv[CM::NK > 0 ? CM::NK : 1][NQ];
  double lu[NQ * NQ];
  int piv[NQ];
};

// Sines and cosines of the replicated residual (gen/<model>.h::eval_r<TP>): angles 2p and 2p+1 are evaluated side by
// side, lanes 0..7 take the even one, their mirrors the odd one, and both results are broadcast back (a sin/cos pair
// is ~45 instructions, the exchange 5).
inline Vec16 od_sin(const Vec16& a) { Vec16 r; for (int i = 0; i < 16; ++i) r.v[i] = od_sin(a.v[i]); return r; }
inline Vec16 od_cos(const Vec16& a) { Vec16 r; for (int i = 0; i < 16; ++i) r.v[i] = od_cos(a.v[i]); return r; }
template <class RO> struct TrigHalves {
  template <int N> OD_HD static void sincos_n(const double* a, double* s, double* c) {
    using V = typename RO::V;
    const typename RO::B half = RO::lane_flag(0xFF00u);
#pragma unroll
    for (int p = 0; p + 1 < N; p += 2) {
      const V x = RO::sel(half, a[p + 1], a[p]);
      const V sx = od_sin(x), cx = od_cos(x);
      RO::template bc2<0, 8>(sx, s[p], s[p + 1]);
      RO::template bc2<0, 8>(cx, c[p], c[p + 1]);
    }
    if constexpr (N & 1) { s[N - 1] = od_sin(a[N - 1]); c[N - 1] = od_cos(a[N - 1]); }
  }
};

// sum / max over the 8 lanes of each half row; every lane of the half ends with the same bits
template <class RO> OD_HD typename RO::V half_sum(typename RO::V v) {
  v = v + RO::xor1(v);
  v = v + RO::xor2(v);
  v = v + RO::half_mirror(v);
  return v;
}
template <class RO> OD_HD typename RO::V half_max(typename RO::V v) {
  v = RO::vmax(v, RO::xor1(v));
  v = RO::vmax(v, RO::xor2(v));
  v = RO::vmax(v, RO::half_mirror(v));
  return v;
}
template <class RO> OD_HD typename RO::V half_min(typename RO::V v) {
  v = RO::vmin(v, RO::xor1(v));
  v = RO::vmin(v, RO::xor2(v));
  v = RO::vmin(v, RO::half_mirror(v));
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// residual r(z; theta, 0)
// ---------------------------------------------------------------------------------------------------------------
template <class CM, class RO>
OD_HD void coop_eval_r(const CoopLanes<CM, RO>& L, const CoopVec<CM::NQ, typename RO::V>& z, const double* th, const double* pre,
                       double* tr, CoopRes<CM::NQ, typename RO::V>& r) {
  using M = typename CM::M;
  using V = typename RO::V;
  // the lane-per-problem residual on a replicated z that holds q and the contact forces the dynamics rows read;
  // s_i and s_b are zero there, so its slack / velocity rows return the lanes' aux-1 expressions.  Rows nobody
  // reads (bilinear, cone, psi) are dead code.
  double zr[M::NZ], rr[M::NZ];
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) zr[i] = 0.0;
#pragma unroll
  for (int k = 0; k < CM::NQ; ++k) zr[CM::ZQ[k]] = z.q[k];
  CM::template gather_r<RO>(z.P0, z.P1, z.D0, z.D1, zr);
  M::template eval_r<TrigHalves<RO>>(zr, th, pre, tr, rr);
#pragma unroll
  for (int k = 0; k < CM::NQ; ++k) r.rd[k] = rr[CM::RDYN[k]];
  const V e1 = CM::template pick_e1<RO>(L, rr);
  r.r1 = e1 + L.c_s * z.D0 + L.c_v * z.D1;
  if constexpr (CM::NK > 0) {
    V gp = V(0.0);
    if constexpr (CM::SH > 0) gp = RO::template shr<CM::SH>(z.P0);
    r.r2 = L.c_psi * z.P0 + L.gcoef * gp + L.gconst;
  } else {
    r.r2 = V(0.0);
  }
  r.rA = z.P0 * z.D0 + z.P1 * z.D1;
  r.rB = z.P0 * z.D1 + z.P1 * z.D0;
}

// max |r| over the equality rows and over the complementarity rows, NaN-sticky like od_solver.h::viol_eq / viol_bil
// (a NaN anywhere in a group makes that violation NaN): lanes 0..7 reduce the equality rows, their mirrors 8..15
// the complementarity rows, one butterfly for both.
template <class CM, class RO>
OD_HD void coop_viol(const CoopLanes<CM, RO>& L, const CoopRes<CM::NQ, typename RO::V>& r, double& r_vio, double& k_vio) {
  using V = typename RO::V;
  const double inf = __builtin_inf();
  double ve = 0.0, se = 0.0;
#pragma unroll
  for (int k = 0; k < CM::NQ; ++k) { const double a = od_abs(r.rd[k]); ve = od_fmax(ve, a); se += a; }
  const V a1 = od_abs(r.r1), a2 = od_abs(r.r2), aA = od_abs(r.rA), aB = od_abs(r.rB);
  V v = RO::sel(L.half, od_fmax(aA, aB), od_fmax(a1, a2));
  const V s = RO::sel(L.half, aA + aB, a1 + a2);
  v = RO::sel(s != s, inf, v);              // hardware max drops NaNs: carry them as +inf through the reduction
  v = RO::sel(L.is_role, v, 0.0);
  v = half_max<RO>(v);
  double de, dk;
  RO::template bc2<0, 8>(v, de, dk);
  const double nan = __builtin_nan("");
  de = RO::vmax(de, ve);
  r_vio = (se != se || de == inf) ? nan : de;
  k_vio = (dk == inf) ? nan : dk;
}

// ---------------------------------------------------------------------------------------------------------------
// Jacobian + factorisation.  Block elimination in the order of the generated static elimination (gen/<model>.h):
//   slack rows -> s, psi rows -> psi, velocity rows -> s_b, bilinear rows -> gamma (pivot s, floored),
//   cone: runtime role swap, first pivot, second pivot; then the nq x nq Schur complement (LU, replicated).
// ---------------------------------------------------------------------------------------------------------------
template <class CM, bool PIV, class RO>
OD_HD bool coop_eval_factor(const CoopLanes<CM, RO>& L, const CoopVec<CM::NQ, typename RO::V>& z, const double* th, const double* pre,
                            const double* tr, double reg, CoopFact<CM, RO>& f) {
  using M = typename CM::M;
  using V = typename RO::V;
  constexpr int NQ = CM::NQ;
  // orthant members clamped from below at reg (rz!(...; reg)); cone members are not
  const V regl = L.reg_floor + reg;            // reg on contact lanes, -inf elsewhere
  const V P0c = RO::vmax(z.P0, regl), D0c = RO::vmax(z.D0, regl);
  double zr[M::NZ], a[M::NNZ];
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) zr[i] = 0.0;
#pragma unroll
  for (int k = 0; k < NQ; ++k) zr[CM::ZQ[k]] = z.q[k];
  CM::template gather_rz<RO>(P0c, z.P1, D0c, z.D1, zr);
  M::eval_rz(zr, th, pre, tr, a);
  double dqq[NQ * NQ];
  CM::dqq_from(a, dqq);
  CM::couplings(a, f.nv, f.nbv);
  CM::template build_jf<RO>(L, a, f.JF);
  // ---- contacts
  f.gc = P0c;
  if constexpr (CM::NC > 0) {
    f.ipc = od_rcp(od_max(D0c, V(OD_PIVOT_FLOOR)));
    const V w = f.ipc * P0c;
#pragma unroll
    for (int j = 0; j < NQ; ++j) f.t[j] = CM::UPJ[j] ? w * f.JF[j] : V(0.0);
  }
  // ---- cones
  if constexpr (CM::NK > 0) {
    f.gA = -(z.D0 * L.gcoef);
    f.gB = -(z.D1 * L.gcoef);
    // role swap (gen/<model>.h: sw = |psi| > |s_psi|): the first pivot row is the head row A (on s_psi) if sw, the tail
    // row B (on b) otherwise.  The rows are exchanged physically -- through the four coefficients they are built from,
    // row A = P1 nJ + gA t', row B = P0 nJ + gB t' -- and the second row is reduced entry by entry, in the order of the
    // lane-per-problem elimination.  (Folding the exchange into scalar coefficients of the finished rows,
    // W = cA qA + cB qB, subtracts two large products when the second pivot is small; a build with that form and with
    // multiplicative role masks stopped converging on the device -- not bisected further, this is the stable form.)
    f.sw = od_abs(z.P0) > od_abs(z.D0);
    const V p1 = RO::sel(f.sw, z.P0, z.D0), o2 = RO::sel(f.sw, z.P1, z.D1), p2 = RO::sel(f.sw, z.D0, z.P0);
    f.o1 = RO::sel(f.sw, z.D1, z.P1);
    const V m1 = RO::sel(f.sw, z.P1, z.P0), m2 = RO::sel(f.sw, z.P0, z.P1);
    const V g1 = RO::sel(f.sw, f.gA, f.gB), g2 = RO::sel(f.sw, f.gB, f.gA);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      // s_b eliminated through its velocity row (c_v d s_b + JF . dq = r_v, c_v = +-1): d s_b = c_v (r_v - JF . dq)
      const V nJ = -(L.c_v * f.JF[j]);
      f.q1[j] = CM::UPV[j] ? m1 * nJ : V(0.0);
      f.q2[j] = CM::UPV[j] ? m2 * nJ : V(0.0);
      if constexpr (CM::SH > 0) {
        if (CM::UPJ[j]) {
          const V tp = RO::template shr<CM::SH>(f.t[j]);
          f.q1[j] = f.q1[j] + g1 * tp;
          f.q2[j] = f.q2[j] + g2 * tp;
        }
      }
    }
    f.ip1 = od_rcp(p1);
    f.l2 = o2 * f.ip1;
    const V p2e = p2 - f.l2 * f.o1;
#pragma unroll
    for (int j = 0; j < NQ; ++j) f.q2[j] = f.q2[j] - f.l2 * f.q1[j];
    f.ip2 = od_rcp(p2e);
    f.u1 = RO::sel(f.sw, 0.0, f.ip1);
    const V u2 = RO::sel(f.sw, 1.0, 0.0) - f.u1 * f.o1;
    f.u3 = u2 * f.ip2;
  }
  // ---- Schur complement on the dynamics rows (replicated), then its LU.  d b_c = Wy - W . dq  (see coop_solve)
  V W[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    if constexpr (CM::NK > 0) W[j] = f.u1 * f.q1[j] + f.u3 * f.q2[j];
    else W[j] = V(0.0);
  }
  CM::template schur<RO>(f, W, dqq);
#pragma unroll
  for (int i = 0; i < NQ * NQ; ++i) f.lu[i] = dqq[i];
  if constexpr (PIV) return od_lu_factor<double, NQ>(f.lu, f.piv);
  else return od_lu_factor_static<double, NQ>(f.lu);
}

// x = rz^{-1} r with the stored factors
template <class CM, bool PIV, class RO>
OD_HD void coop_solve(const CoopLanes<CM, RO>& L, const CoopFact<CM, RO>& f, const CoopVec<CM::NQ, typename RO::V>& z,
                      const CoopRes<CM::NQ, typename RO::V>& r, CoopVec<CM::NQ, typename RO::V>& x) {
  using V = typename RO::V;
  constexpr int NQ = CM::NQ;
  double rd[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) rd[k] = r.rd[k];
  // forward: contacts
  const V y = r.rA - f.gc * r.r1;
  V ty = V(0.0), Wy = V(0.0), y1 = V(0.0), y2 = V(0.0);
  if constexpr (CM::NC > 0) ty = f.ipc * y;
  // forward: cones
  if constexpr (CM::NK > 0) {
    const V nr1 = L.c_v * r.r1;
    V yA = r.rA - z.D0 * r.r2 - z.P1 * nr1;
    V yB = r.rB - z.D1 * r.r2 - z.P0 * nr1;
    if constexpr (CM::SH > 0) {
      const V typ = RO::template shr<CM::SH>(ty);
      yA = yA - f.gA * typ;
      yB = yB - f.gB * typ;
    }
    y1 = RO::sel(f.sw, yA, yB);
    y2 = RO::sel(f.sw, yB, yA) - f.l2 * y1;
    Wy = f.u1 * y1 + f.u3 * y2;
  }
  CM::template rhs_update<RO>(f, ty, Wy, rd);
  if constexpr (PIV) od_lu_solve<double, NQ>(f.lu, f.piv, rd);
  else od_lu_solve_static<double, NQ>(f.lu, rd);
#pragma unroll
  for (int k = 0; k < NQ; ++k) x.q[k] = rd[k];
  // back substitution inside the lanes
  V dg = ty, a1 = r.r1, s1 = V(0.0), s2 = V(0.0);
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    if (CM::UPJ[j]) dg = dg + f.t[j] * rd[j];
    if (CM::UPJ[j] || CM::UPV[j]) a1 = a1 - f.JF[j] * rd[j];
    if constexpr (CM::NK > 0) { s1 = s1 + f.q1[j] * rd[j]; s2 = s2 + f.q2[j] * rd[j]; }
  }
  if constexpr (CM::NK > 0) {
    const V x2 = f.ip2 * (y2 - s2);
    const V x1 = f.ip1 * (y1 - s1 - f.o1 * x2);
    const V db = RO::sel(f.sw, x2, x1), dsp = RO::sel(f.sw, x1, x2);
    V dgp = V(0.0);
    if constexpr (CM::SH > 0) dgp = RO::template shr<CM::SH>(dg);
    const V dpsi = r.r2 - L.gcoef * dgp;
    x.P0 = RO::sel(L.is_cone, dpsi, dg);
    // b, s_b exist on cone lanes only; contact lanes keep exact zeros (a select, not a multiplication by 0: the cone
    // arithmetic of a contact lane may overflow when gamma or s underflow, and 0 * inf would poison its rows)
    x.P1 = RO::sel(L.is_cone, db, 0.0);
    x.D0 = RO::sel(L.is_cone, dsp, a1);
    x.D1 = L.c_v * a1;                          // c_v is 0 off the cone lanes, a1 their finite slack direction
  } else {
    x.P0 = dg; x.P1 = V(0.0); x.D0 = a1; x.D1 = V(0.0);
  }
  // (lanes without a role compute along with finite garbage; every reduction masks them and no lane reads them)
}

// ---------------------------------------------------------------------------------------------------------------
// step length, centering, correction
// ---------------------------------------------------------------------------------------------------------------
// CVXOPT sec. 8.2 step for a two-dimensional cone (od_solver.h::soc_step_one<2>), lane-parallel.  Everything that
// depends on the cone variable alone -- not on the direction -- is computed once per iterate (StepPre) and serves the
// predictor's and the corrector's step length: 1/sqrt(l0^2 - l1^2) with its guards, its square, 1/(l0/sqrt(.) + 1).
template <class RO> struct StepPre {
  typename RO::V zz;             // contacts: gamma (lanes 0..7) | s (mirrors)
  typename RO::V l0, l1;         // cones: primal (lanes 0..7) | dual (mirrors) member
  typename RO::V isq, ill, rc1;
};

template <class CM, class RO>
OD_HD StepPre<RO> coop_step_pre(const CoopLanes<CM, RO>& L, const CoopVec<CM::NQ, typename RO::V>& z) {
  using V = typename RO::V;
  StepPre<RO> p;
  p.zz = RO::sel(L.half, z.D0, z.P0);
  p.l0 = p.zz;
  p.l1 = RO::sel(L.half, z.D1, z.P1);
  if constexpr (CM::NK > 0) {
    V ll = p.l0 * p.l0;
    ll = ll - p.l1 * p.l1;
    ll = od_fmax(ll, V(1e-25)) + 1e-14;
    p.isq = od_rsqrt(ll);
    p.ill = p.isq * p.isq;
    p.rc1 = od_rcp(p.l0 * p.isq + 1.0);
  } else {
    p.isq = p.ill = p.rc1 = V(0.0);
  }
  return p;
}

template <class RO> OD_HD typename RO::V coop_soc_step2(const StepPre<RO>& p, typename RO::V d0, typename RO::V d1, double tau, typename RO::V& den) {
  using V = typename RO::V;
  V ld = p.l0 * d0;
  ld = ld - p.l1 * d1;
  ld = ld + 1e-14;
  const V rs = ld * p.ill;
  const V c = (ld * p.isq + d0) * p.rc1;
  const V nv = od_abs(d1 * p.isq - c * p.l1 * p.ill);
  den = nv - rs;
  return tau * od_rcp(den);                                // where den > 0; the caller caps at 1
}

template <class CM, class RO>
OD_HD double coop_step_length(const CoopLanes<CM, RO>& L, const StepPre<RO>& p, const CoopVec<CM::NQ, typename RO::V>& d,
                              double tau_ort, double tau_soc) {
  using V = typename RO::V;
  V a = V(1.0);
  const V d0h = RO::sel(L.half, d.D0, d.P0);
  if constexpr (CM::NC > 0) {
    // lanes 0..7 test gamma, their mirrors test s:  alpha <= tau * z / d  where d > 0
    a = RO::sel(L.is_contact && (d0h > 0.0), (tau_ort * p.zz) * od_rcp(d0h), a);
  }
  if constexpr (CM::NK > 0) {
    const V d1h = RO::sel(L.half, d.D1, d.P1);
    V den;
    const V as = coop_soc_step2<RO>(p, -d0h, -d1h, tau_soc, den);
    a = RO::sel(L.is_cone && (den > 0.0), as, a);
  }
  a = od_fmin(a, V(1.0));
  a = half_min<RO>(a);
  double a0, a1;
  RO::template bc2<0, 8>(a, a0, a1);
  return RO::vmin(a0, a1);
}

// CVXOPT sec. 5.1.3: mu = <primal, dual>/ncones ; sigma = clamp(mu_aff/mu, 0, 1)^3  (od_solver.h::centering_kappa)
template <class CM, class RO>
OD_HD double coop_centering(const CoopLanes<CM, RO>& L, const CoopVec<CM::NQ, typename RO::V>& z, const CoopVec<CM::NQ, typename RO::V>& d, double aaff) {
  using V = typename RO::V;
  constexpr int n = CM::NC + CM::NK;
  const V p = z.P0 * z.D0 + z.P1 * z.D1;
  const V pa = (z.P0 - aaff * d.P0) * (z.D0 - aaff * d.D0) + (z.P1 - aaff * d.P1) * (z.D1 - aaff * d.D1);
  V v = RO::sel(L.half, pa, p);            // lanes 0..7 sum <z1, z2>, the mirrors the affine products
  v = RO::sel(L.is_role, v, 0.0);
  v = half_sum<RO>(v);
  double s, sa;
  RO::template bc2<0, 8>(v, s, sa);
  const double mu = s * (1.0 / n);
  double q = sa * od_rcp(s);
  q = od_fmax(q, 0.0);
  q = od_fmin(q, 1.0);
  return q * q * q * mu;
}

// ---------------------------------------------------------------------------------------------------------------
// Sixteen line-search trials at once: lane g of the row evaluates r(z - a_g D; theta, 0) for ITS step a_g = alpha 2^-g --
// the whole residual in one lane, every contact and cone in turn, with the arithmetic of coop_eval_r / coop_viol (the
// same generated eval_r on the same replicated z, the same row formulas; the maxima are order independent), so that a
// trial is accepted here exactly when the cooperative evaluation of the same step accepts it.
// ---------------------------------------------------------------------------------------------------------------
template <class CM, class RO, int R = 0>
OD_HD void coop_fetch_roles(const CoopVec<CM::NQ, typename RO::V>& z, typename RO::V* P0, typename RO::V* P1, typename RO::V* D0, typename RO::V* D1) {
  if constexpr (R < CM::NC + CM::NK) {
    using V = typename RO::V;
    P0[R] = V(RO::template bc<R>(z.P0)); P1[R] = V(RO::template bc<R>(z.P1));
    D0[R] = V(RO::template bc<R>(z.D0)); D1[R] = V(RO::template bc<R>(z.D1));
    coop_fetch_roles<CM, RO, R + 1>(z, P0, P1, D0, D1);
  }
}

template <class CM, class RO>
OD_HD typename RO::B coop_trials_lanes(const CoopLanes<CM, RO>& L, const double* th, const double* pre, const CoopVec<CM::NQ, typename RO::V>& z,
                                       const CoopVec<CM::NQ, typename RO::V>& D, typename RO::V aj, double r_vio, double k_vio) {
  using M = typename CM::M;
  using V = typename RO::V;
  constexpr int NQ = CM::NQ, NR = (CM::NC + CM::NK) > 0 ? (CM::NC + CM::NK) : 1;
  const double inf = __builtin_inf();
  V zP0[NR], zP1[NR], zD0[NR], zD1[NR], dP0[NR], dP1[NR], dD0[NR], dD1[NR];
  coop_fetch_roles<CM, RO>(z, zP0, zP1, zD0, zD1);
  coop_fetch_roles<CM, RO>(D, dP0, dP1, dD0, dD1);
  V zr[M::NZ], rr[M::NZ], thv[M::NTH], prev[M::NPRE], trv[M::NTR];
#pragma unroll
  for (int i = 0; i < M::NZ; ++i) zr[i] = V(0.0);
#pragma unroll
  for (int k = 0; k < NQ; ++k) zr[CM::ZQ[k]] = V(z.q[k]) - aj * V(D.q[k]);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    zP0[i] = zP0[i] - aj * dP0[i]; zP1[i] = zP1[i] - aj * dP1[i];
    zD0[i] = zD0[i] - aj * dD0[i]; zD1[i] = zD1[i] - aj * dD1[i];
  }
  CM::scatter_r(zP0, zP1, zD0, zD1, zr);
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
    const double c_s = RO::opaque(cone ? 0.0 : 1.0), c_v = RO::opaque(CM::CV[i]);
    const V r1 = rr[CM::E1ROW[i]] + c_s * zD0[i] + c_v * zD1[i];
    V r2 = V(0.0);
    if constexpr (CM::NK > 0) {
      if (cone) {
        const int c = i - CM::NC;
        const double c_psi = RO::opaque(1.0);
        const V gp = (CM::SH > 0 && i - CM::SH >= 0) ? zP0[i - CM::SH >= 0 ? i - CM::SH : 0] : V(0.0);
        r2 = c_psi * zP0[i] + V(g[c]) * gp + V(gc[c]);
      }
    }
    const V rA = zP0[i] * zD0[i] + zP1[i] * zD1[i];
    const V rB = zP0[i] * zD1[i] + zP1[i] * zD0[i];
    const V a1 = od_abs(r1), a2 = od_abs(r2), aA = od_abs(rA), aB = od_abs(rB);
    V e = od_fmax(a1, a2);
    const V es = a1 + a2;
    e = RO::sel(es != es, inf, e);
    V k = od_fmax(aA, aB);
    const V ks = aA + aB;
    k = RO::sel(ks != ks, inf, k);
    de = RO::vmax(de, e);
    dk = RO::vmax(dk, k);
  }
  de = RO::vmax(de, ve);
  // NaN-sticky violations (coop_viol): a NaN fails both comparisons
  const typename RO::B r_nan = (se != se) || (de == inf), k_nan = (dk == inf);
  const typename RO::B r_ok = (de <= r_vio) && !r_nan, k_ok = (dk <= k_vio) && !k_nan;
  return r_ok || k_ok;
}

// ---------------------------------------------------------------------------------------------------------------
// one predictor-corrector iteration (od_solver.h::ip_iteration), line search included
// ---------------------------------------------------------------------------------------------------------------
// Returns true if the iteration left EVERYTHING it reads unchanged -- a zero step length (a cone variable exactly on its
// boundary) whose first trial reproduced both violations bit for bit: every further iteration would then repeat this one, so
// the caller may go straight to max_iter (same iterate, status and iteration counts; 9 of the 29 knots of the headline
// workload that jam do this, from their 34th iteration on average: profiles/r3_jam_knots.txt).
template <class CM, class RO>
OD_HD bool coop_iteration(const CoopLanes<CM, RO>& L, const Opts<double>& o, const double* th, const double* pre, double* tr,
                          CoopVec<CM::NQ, typename RO::V>& z, CoopRes<CM::NQ, typename RO::V>& r, double& r_vio, double& k_vio,
                          double& reg_prev, int& status, CoopFact<CM, RO>& f, int& ls_hint) {
  using V = typename RO::V;
  using Vec = CoopVec<CM::NQ, V>;
  using Res = CoopRes<CM::NQ, V>;
  constexpr int NQ = CM::NQ;
  constexpr bool CONES = (CM::NC + CM::NK) > 0;
  constexpr bool PIV = !CM::M::STATIC_TAIL;
  const double reg = (k_vio < o.kappa_reg) ? k_vio * o.gamma_reg : 0.0;
  reg_prev = reg;
  if (!coop_eval_factor<CM, PIV, RO>(L, z, th, pre, tr, reg, f)) status &= ~OD_ST_FACTOR_OK;
  Vec D;
  coop_solve<CM, PIV, RO>(L, f, z, r, D);
  const StepPre<RO> sp = coop_step_pre<CM, RO>(L, z);
  if constexpr (CONES) {
    const double aaff = coop_step_length<CM, RO>(L, sp, D, 1.0, 1.0);
    double kap = coop_centering<CM, RO>(L, z, D, aaff);
    kap = od_fmax(kap, o.kappa_eval * o.undercut_inv);
    // r(z; kappa) from r(z; 0) on the head rows, then the second-order correction of the predictor
    r.rA = r.rA - kap + (D.P0 * D.D0 + D.P1 * D.D1);
    r.rB = r.rB + (D.P0 * D.D1 + D.P1 * D.D0);
    coop_solve<CM, PIV, RO>(L, f, z, r, D);
  }
  const double vio = RO::vmax(r_vio, k_vio);
  const double tau = 1.0 - od_fmin(o.eps_min, vio * vio);
  double alpha = coop_step_length<CM, RO>(L, sp, D, tau, od_fmin(tau, 0.99));
  // backtracking until either violation does not increase (od_solver.h::line_search): two trials one after the other,
  // then -- a solve that jams spends most of its time here, ~8 trials in each of its 100 iterations -- the 16 lanes of the
  // row each try a step size (coop_trials_lanes), agree on the first accepted one (the one the sequential loop would
  // find) and the row re-evaluates that one in its cooperative form: same iterates, bit for bit
  // (at least one trial is evaluated -- max_ls >= 1 is enforced at the API --, so the candidates start out unset: initialising
  // them from z / r costs 16 register moves per iteration that nothing reads)
  Vec zc;
  Res rc;
  double r_c, k_c;
  auto trial = [&](double a) {
#pragma unroll
    for (int k = 0; k < NQ; ++k) zc.q[k] = z.q[k] - a * D.q[k];
    zc.P0 = z.P0 - a * D.P0; zc.P1 = z.P1 - a * D.P1;
    zc.D0 = z.D0 - a * D.D0; zc.D1 = z.D1 - a * D.D1;
    coop_eval_r<CM, RO>(L, zc, th, pre, tr, rc);
    coop_viol<CM, RO>(L, rc, r_c, k_c);
    return r_c <= r_vio || k_c <= k_vio;
  };
  // (a knot whose previous iteration needed more than two trials -- a jam -- goes to the lane-parallel round at once: the
  // accepted trial is the same either way)
  const int nseq = ls_hint >= 2 ? 0 : (o.max_ls < 2 ? o.max_ls : 2);
  bool done = false;
  int ls = 0;
  for (; ls < nseq; ++ls) {
    if (trial(alpha)) { done = true; break; }
    if (ls + 1 < o.max_ls) alpha *= 0.5;
  }
  if (!done && ls < o.max_ls) {
    // alpha is the step of trial `ls`; the lanes try trials j0 .. j0 + 15
    bool found = false;
    for (int j0 = ls; j0 < o.max_ls && !found; j0 += 16) {
      const typename RO::B acc = coop_trials_lanes<CM, RO>(L, th, pre, z, D, RO::lane_ldexp(alpha), r_vio, k_vio) && RO::lane_below(o.max_ls - j0);
      const unsigned m = RO::row_ballot(acc);
      if (m != 0) {
        alpha = od_ldexp(alpha, -__builtin_ctz(m));                      // first accepted trial of the round
        found = true;
        ls = j0 + __builtin_ctz(m);
      } else {
        ls = o.max_ls - 1;
        const int left = o.max_ls - 1 - j0;                              // trials after j0: move on by 16, or to the last
        alpha = od_ldexp(alpha, -(left < 16 ? left : 16));
        if (left < 16) break;                                            // alpha is now the last trial's step
      }
    }
    trial(alpha);                                                        // the row lands on the chosen trial
  }
  const bool fixed_point = (alpha == 0.0) && done && ls == 0 && r_c == r_vio && k_c == k_vio;
  ls_hint = ls;                                                          // index of the accepted trial
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
OD_HD int coop_ip_step(const CoopLanes<CM, RO>& L, const Opts<double>& o, const double* th, CoopVec<CM::NQ, typename RO::V>& z, bool want_grad,
                       Defer& defer, int* iters) {
  using M = typename CM::M;
  using V = typename RO::V;
  constexpr int NQ = CM::NQ;
  double pre[M::NPRE], tr[M::NTR], qs[NQ];
  CoopRes<NQ, V> r;
  CoopFact<CM, RO> f;
  M::eval_pre(th, pre);
  coop_eval_r<CM, RO>(L, z, th, pre, tr, r);
  double r_vio, k_vio;
  coop_viol<CM, RO>(L, r, r_vio, k_vio);
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
    if (coop_iteration<CM, RO>(L, o, th, pre, tr, z, r, r_vio, k_vio, reg_prev, status, f, ls_hint) && it + 1 < o.max_iter) it = o.max_iter - 1;
  }
#pragma unroll
  for (int k = 0; k < NQ; ++k) z.q[k] = qs[k];
  return status;
}

// ---------------------------------------------------------------------------------------------------------------
// units of work: one knot / one rollout per row.  Same outputs as od_units.h::knot_state.
// ---------------------------------------------------------------------------------------------------------------
template <class CM, class RO> struct CoopDefer {
  using V = typename RO::V;
  const View<double>& zg;
  long k;
  typename RO::I idx;
  OD_HD void operator()(const CoopVec<CM::NQ, V>& z, double reg) const {
    if (!zg.ok()) return;
    RO::template store<0>(zg, idx, k, z.P0);
    RO::template store<1>(zg, idx, k, z.P1);
    RO::template store<2>(zg, idx, k, z.D0);
    RO::template store<3>(zg, idx, k, z.D1);
    if (RO::first_lane()) {
#pragma unroll
      for (int i = 0; i < CM::NQ; ++i) zg.at(CM::ZQ[i], k) = z.q[i];
      zg.at(CM::M::NZ, k) = reg;
    }
  }
};

struct CoopNoHook { OD_HD void operator()() const {} };
// `before_stores` runs between the solve and the knot's stores: a rollout waits there for its prefetched next control,
// while the only memory operations in flight are a knot old (the wait counter is in order: after the stores it would
// wait for them as well, one store round trip per knot)
template <class CM, class RO, class Hook = CoopNoHook>
OD_HD int coop_knot_state(const CoopLanes<CM, RO>& L0, const StepArgs<double>& a, long k, const double* xin, const double* uin, double* q3out,
                          const Hook& before_stores = Hook()) {
  using M = typename CM::M;
  using V = typename RO::V;
  constexpr int nq = M::NQ;
  double th[M::NTH], z0[M::NZ];
  mech_setup<M>(xin, xin + nq, uin, a.fric, a.h, th, z0);
  CoopLanes<CM, RO> L = L0;
  L.set_theta(th);
  CoopVec<nq, V> z;
#pragma unroll
  for (int i = 0; i < nq; ++i) z.q[i] = z0[CM::ZQ[i]];
  z.P0 = RO::lane_table(CM::ZI_P0); z.P1 = RO::lane_table(CM::ZI_P1);
  z.D0 = RO::lane_table(CM::ZI_D0); z.D1 = RO::lane_table(CM::ZI_D1);
  CoopDefer<CM, RO> defer{a.zg, k, L0.zg_idx};
  int it[2];
  const int st = coop_ip_step<CM, RO>(L, a.opts, th, z, a.want_grad != 0, defer, it);
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
  return st;                    // (the same in every lane of the row: the solve's exits are row decisions)
}

template <class CM, class RO> OD_HD void coop_unit_step_state(const StepArgs<double>& a, long b) {
  using M = typename CM::M;
  constexpr int nq = M::NQ, n = 2 * M::NQ;
  double x[n], u[M::NU > 0 ? M::NU : 1], q3[nq];
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = a.x.at(i, b);
#pragma unroll
  for (int i = 0; i < M::NU; ++i) u[i] = a.u.at(i, b);
  CoopLanes<CM, RO> L;
  L.init();
  coop_knot_state<CM, RO>(L, a, b, x, u, q3);
}

template <class CM, class RO> OD_HD void coop_unit_rollout_state(const RolloutArgs<double>& ra, long b) {
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
  CoopLanes<CM, RO> L;
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
    coop_knot_state<CM, RO>(L, a, k, x, u, q3, [&]() {
#pragma unroll
      for (int i = 0; i < M::NU; ++i) RO::arrived(un[i]);
    });
#pragma unroll
    for (int i = 0; i < nq; ++i) { x[i] = x[nq + i]; x[nq + i] = q3[i]; }
  }
}


// closed-loop rollout = forward pass of iLQR (od_units.h::unit_rollout_policy): candidate p = a*Bnom + b follows the
// nominal trajectory b with step size alphas[a], u_t = ubar_t + alpha k_t + K_t (x_t - xbar_t); one candidate per row
template <class CM, class RO> OD_HD void coop_unit_rollout_policy(const PolicyArgs<double>& pa, long p) {
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
  CoopLanes<CM, RO> L;
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
    const int st = coop_knot_state<CM, RO>(L, a, kc, x, u, q3);
    if (pa.stop_failed && !(st & OD_ST_EVAL_OK)) {          // (PolicyArgs::stop_failed; the whole row leaves, like a row past the batch)
      if (RO::first_lane()) policy_mark_rest_failed(pa, t, p);
      break;
    }
#pragma unroll
    for (int i = 0; i < nq; ++i) { x[i] = x[nq + i]; x[nq + i] = q3[i]; }
  }
}

}  // namespace od
