// ORACLE -- TEST INFRASTRUCTURE ONLY (see reak_math.hpp header).
//
// Exact operation count of one x' = f(x, u) evaluation of the restated reference
// (kte_nl_system::get_state_derivative, ctrl/ctrl_sys/kte_nl_system.hpp:239-290, with the KTE passes, the dense
// mass_matrix_calc product and the Cholesky solve behind it): the restatement's headers are compiled a second time with
// `double` replaced by a counting scalar, so the count is what that code executes, operation by operation.
// Two figures per operation class:
//   * all      every fp64 operation the reference's code performs, including its dense products over structural zeros
//              (Mcm is block-diagonal and Tcm is block-lower-triangular, but mass_matrix_calculator.cpp:262-295 multiplies
//              them as dense matrices);
//   * useful   operations none of whose operands is an exact zero coming from that structure (x * 0, s + 0 * y): the
//              count a structure-aware implementation (the HIP kernels) has to perform for the same result bits.
// SURVEY.md 8(d) "flops_per_edge = steps * 4 F_eval (+ F_eval)": bench.py reports its roofline figures with `useful`.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <vector>

namespace oracle_count {
struct Counters {
  uint64_t add = 0, mul = 0, div = 0, sqrt = 0, trig = 0, cmp = 0;  // all
  uint64_t add_useful = 0, mul_useful = 0;                          // without structural zeros
};
inline Counters& C() {
  static Counters c;
  return c;
}
struct CD {  // counting double: same size and layout as a double
  double v;
  CD() : v(0.0) {}
  CD(double x) : v(x) {}
  CD(int x) : v(x) {}
  CD(unsigned x) : v(x) {}
  CD(long x) : v(double(x)) {}
  CD(unsigned long x) : v(double(x)) {}
  explicit operator double() const { return v; }
  explicit operator int() const { return int(v); }
  explicit operator bool() const { return v != 0.0; }
  CD& operator+=(CD b);
  CD& operator-=(CD b);
  CD& operator*=(CD b);
  CD& operator/=(CD b);
};
static_assert(sizeof(CD) == sizeof(double), "layout");
inline CD operator+(CD a, CD b) {
  ++C().add;
  if (a.v != 0.0 && b.v != 0.0) ++C().add_useful;
  return CD(a.v + b.v);
}
inline CD operator-(CD a, CD b) {
  ++C().add;
  if (a.v != 0.0 && b.v != 0.0) ++C().add_useful;
  return CD(a.v - b.v);
}
inline CD operator*(CD a, CD b) {
  ++C().mul;
  if (a.v != 0.0 && b.v != 0.0) ++C().mul_useful;
  return CD(a.v * b.v);
}
inline CD operator/(CD a, CD b) {
  ++C().div;
  return CD(a.v / b.v);
}
inline CD operator-(CD a) { return CD(-a.v); }  // sign flips are operand modifiers, not operations
inline CD operator+(CD a) { return a; }
inline CD& CD::operator+=(CD b) { return *this = *this + b; }
inline CD& CD::operator-=(CD b) { return *this = *this - b; }
inline CD& CD::operator*=(CD b) { return *this = *this * b; }
inline CD& CD::operator/=(CD b) { return *this = *this / b; }
#define ORACLE_CMP(op) \
  inline bool operator op(CD a, CD b) { ++C().cmp; return a.v op b.v; }
ORACLE_CMP(<) ORACLE_CMP(>) ORACLE_CMP(<=) ORACLE_CMP(>=) ORACLE_CMP(==) ORACLE_CMP(!=)
#undef ORACLE_CMP
}  // namespace oracle_count

namespace std {  // the restatement calls std::sqrt etc. on its scalar type
inline oracle_count::CD sqrt(oracle_count::CD a) { ++oracle_count::C().sqrt; return oracle_count::CD(std::sqrt(a.v)); }
inline oracle_count::CD sin(oracle_count::CD a) { ++oracle_count::C().trig; return oracle_count::CD(std::sin(a.v)); }
inline oracle_count::CD cos(oracle_count::CD a) { ++oracle_count::C().trig; return oracle_count::CD(std::cos(a.v)); }
inline oracle_count::CD acos(oracle_count::CD a) { ++oracle_count::C().trig; return oracle_count::CD(std::acos(a.v)); }
inline oracle_count::CD atan2(oracle_count::CD a, oracle_count::CD b) { ++oracle_count::C().trig; return oracle_count::CD(std::atan2(a.v, b.v)); }
inline oracle_count::CD fabs(oracle_count::CD a) { return oracle_count::CD(std::fabs(a.v)); }
inline oracle_count::CD abs(oracle_count::CD a) { return oracle_count::CD(std::fabs(a.v)); }
inline oracle_count::CD pow(oracle_count::CD a, oracle_count::CD b) { ++oracle_count::C().trig; return oracle_count::CD(std::pow(a.v, b.v)); }
inline oracle_count::CD exp(oracle_count::CD a) { ++oracle_count::C().trig; return oracle_count::CD(std::exp(a.v)); }
inline oracle_count::CD floor(oracle_count::CD a) { return oracle_count::CD(std::floor(a.v)); }
inline oracle_count::CD ceil(oracle_count::CD a) { return oracle_count::CD(std::ceil(a.v)); }
inline bool isinf(oracle_count::CD a) { return std::isinf(a.v); }
inline bool isnan(oracle_count::CD a) { return std::isnan(a.v); }
inline bool isfinite(oracle_count::CD a) { return std::isfinite(a.v); }
template <>
struct numeric_limits<oracle_count::CD> {
  static oracle_count::CD infinity() { return oracle_count::CD(numeric_limits<double>::infinity()); }
  static oracle_count::CD max() { return oracle_count::CD(numeric_limits<double>::max()); }
  static oracle_count::CD min() { return oracle_count::CD(numeric_limits<double>::min()); }
  static oracle_count::CD epsilon() { return oracle_count::CD(numeric_limits<double>::epsilon()); }
  static oracle_count::CD quiet_NaN() { return oracle_count::CD(numeric_limits<double>::quiet_NaN()); }
};
}  // namespace std

// the restatement, with its scalar type replaced (rkh_types.h included under the same replacement: the PODs keep
// their layout because CD is one double)
#define double oracle_count::CD
#include "../include/rkh_types.h"
#include "reak_kte.hpp"
#undef double

extern "C" {
// counts[8] = {add, mul, div, sqrt, trig, cmp, add_useful, mul_useful} of ONE get_state_derivative call at (x, u).
// Returns 0, or -1 if the reference would throw (singular mass matrix).
int oracle_feval_op_count(const void* prog, int n_ops, const void* base, const double* x, const double* u,
                          uint64_t* counts) {
  using oracle_count::CD;
  oracle::KteChain chain(static_cast<const rkh_kte_op*>(prog), n_ops, *static_cast<const rkh_chain_base*>(base));
  const int n = chain.n_coords;
  std::vector<CD> xs(2 * n), us(n), pd(2 * n);
  for (int i = 0; i < 2 * n; ++i) xs[i] = CD(x[i]);
  for (int i = 0; i < n; ++i) us[i] = CD(u[i]);
  oracle_count::C() = oracle_count::Counters();
  try {
    chain.get_state_derivative(xs.data(), us.data(), pd.data());
  } catch (...) {
    return -1;
  }
  const oracle_count::Counters& c = oracle_count::C();
  counts[0] = c.add; counts[1] = c.mul; counts[2] = c.div; counts[3] = c.sqrt; counts[4] = c.trig; counts[5] = c.cmp;
  counts[6] = c.add_useful; counts[7] = c.mul_useful;
  return 0;
}
}
