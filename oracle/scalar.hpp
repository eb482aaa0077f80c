// TEST INFRASTRUCTURE — CPU oracle for the FV3 TL/AD dycore hot path.  Not product code:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Scalar types the oracle restatement is instantiated on:
//   double : the nonlinear routine itself (the lower-case/non-_TLM copy each TLM file carries,
//            e.g. FV_TP_2D at model_tlmadm/tp_core_tlm.F90:83).
//   Dual   : forward (tangent) mode — one (value, tangent) pair per variable, exactly the
//            "x, x_tl" pairing of Tapenade's *_TLM routines.  Branches test the
//            value only, like the reference (`IF (c(i,j) .GT. 0.)`, tp_core_tlm.F90:2447).
//   Rev    : reverse (adjoint) mode with a tape — the role adStack.c / *_FWD + *_BWD play in the
//            reference (utils/tapenade/adStack.c:43, tp_core_adm.F90:7552/7750).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>
#include <cstdio>
#include <cstdlib>

namespace orc {

// ---------------------------------------------------------------- forward mode
struct Dual {
  double v, d;
  Dual() : v(0.0), d(0.0) {}
  Dual(double v_) : v(v_), d(0.0) {}
  Dual(double v_, double d_) : v(v_), d(d_) {}
};
inline Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
inline Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
inline Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
inline Dual operator/(Dual a, Dual b) { double q = a.v / b.v; return Dual(q, (a.d - q * b.d) / b.v); }
inline Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
inline Dual operator+(Dual a, double b) { return Dual(a.v + b, a.d); }
inline Dual operator+(double a, Dual b) { return Dual(a + b.v, b.d); }
inline Dual operator-(Dual a, double b) { return Dual(a.v - b, a.d); }
inline Dual operator-(double a, Dual b) { return Dual(a - b.v, -b.d); }
inline Dual operator*(Dual a, double b) { return Dual(a.v * b, a.d * b); }
inline Dual operator*(double a, Dual b) { return Dual(a * b.v, a * b.d); }
inline Dual operator/(Dual a, double b) { return Dual(a.v / b, a.d / b); }
inline Dual operator/(double a, Dual b) { double q = a / b.v; return Dual(q, -q * b.d / b.v); }
inline Dual& operator+=(Dual& a, Dual b) { a = a + b; return a; }
inline Dual& operator-=(Dual& a, Dual b) { a = a - b; return a; }
inline Dual log(Dual a) { return Dual(std::log(a.v), a.d / a.v); }
inline Dual exp(Dual a) { double e = std::exp(a.v); return Dual(e, a.d * e); }
// Tapenade: result_tl = 0 where the argument is exactly 0 (sw_core_tlm.F90:8543-8547)
inline Dual sqrt(Dual a) { double s = std::sqrt(a.v); return Dual(s, a.v == 0.0 ? 0.0 : a.d / (2.0 * s)); }

// ---------------------------------------------------------------- reverse mode
struct Tape {
  std::vector<int32_t> ia, ib;
  std::vector<double> wa, wb;
  int32_t fresh() { ia.push_back(-1); ib.push_back(-1); wa.push_back(0.0); wb.push_back(0.0); return (int32_t)ia.size() - 1; }
  int32_t push(int32_t a, double w_a, int32_t b, double w_b) {
    ia.push_back(a); ib.push_back(b); wa.push_back(w_a); wb.push_back(w_b); return (int32_t)ia.size() - 1;
  }
  size_t size() const { return ia.size(); }
  void clear() { ia.clear(); ib.clear(); wa.clear(); wb.clear(); }
  // adj has size()==tape size, seeded on outputs; after the sweep adj[input id] holds the adjoint.
  void reverse(std::vector<double>& adj) const {
    for (int64_t n = (int64_t)ia.size() - 1; n >= 0; --n) {
      double a = adj[n];
      if (a == 0.0) continue;
      if (ia[n] >= 0) adj[ia[n]] += wa[n] * a;
      if (ib[n] >= 0) adj[ib[n]] += wb[n] * a;
    }
  }
};
inline Tape*& active_tape() { static Tape* t = nullptr; return t; }

struct Rev {
  double v; int32_t id;
  Rev() : v(0.0), id(-1) {}
  Rev(double v_) : v(v_), id(-1) {}
  Rev(double v_, int32_t id_) : v(v_), id(id_) {}
};
inline Rev mk(double v, int32_t a, double wa, int32_t b, double wb) {
  if (a < 0 && b < 0) return Rev(v);
  return Rev(v, active_tape()->push(a, wa, b, wb));
}
inline Rev operator+(Rev a, Rev b) { return mk(a.v + b.v, a.id, 1.0, b.id, 1.0); }
inline Rev operator-(Rev a, Rev b) { return mk(a.v - b.v, a.id, 1.0, b.id, -1.0); }
inline Rev operator*(Rev a, Rev b) { return mk(a.v * b.v, a.id, b.v, b.id, a.v); }
inline Rev operator/(Rev a, Rev b) { double q = a.v / b.v; return mk(q, a.id, 1.0 / b.v, b.id, -q / b.v); }
inline Rev operator-(Rev a) { return mk(-a.v, a.id, -1.0, -1, 0.0); }
inline Rev operator+(Rev a, double b) { return mk(a.v + b, a.id, 1.0, -1, 0.0); }
inline Rev operator+(double a, Rev b) { return mk(a + b.v, b.id, 1.0, -1, 0.0); }
inline Rev operator-(Rev a, double b) { return mk(a.v - b, a.id, 1.0, -1, 0.0); }
inline Rev operator-(double a, Rev b) { return mk(a - b.v, b.id, -1.0, -1, 0.0); }
inline Rev operator*(Rev a, double b) { return mk(a.v * b, a.id, b, -1, 0.0); }
inline Rev operator*(double a, Rev b) { return mk(a * b.v, b.id, a, -1, 0.0); }
inline Rev operator/(Rev a, double b) { return mk(a.v / b, a.id, 1.0 / b, -1, 0.0); }
inline Rev operator/(double a, Rev b) { double q = a / b.v; return mk(q, b.id, -q / b.v, -1, 0.0); }
inline Rev& operator+=(Rev& a, Rev b) { a = a + b; return a; }
inline Rev& operator-=(Rev& a, Rev b) { a = a - b; return a; }
inline Rev log(Rev a) { return mk(std::log(a.v), a.id, 1.0 / a.v, -1, 0.0); }
inline Rev exp(Rev a) { double e = std::exp(a.v); return mk(e, a.id, e, -1, 0.0); }
inline Rev sqrt(Rev a) { double s = std::sqrt(a.v); return mk(s, a.id, a.v == 0.0 ? 0.0 : 0.5 / s, -1, 0.0); }

// value of a, sensitivity of b: for the spots where the reference evaluates the trajectory and the
// perturbation with different coefficients (sw_core_tlm.F90:2436-2452).
inline double combine(double a, double) { return a; }
inline Dual combine(const Dual& a, const Dual& b) { return Dual(a.v, b.d); }
inline Rev combine(const Rev& a, const Rev& b) { return mk(a.v, b.id, 1.0, -1, 0.0); }

// the value replaced, the sensitivities kept (split schemes: the nonlinear routine's result over the _TLM routine's, tp_core.hpp fv_tp_2d_split)
inline void set_val(double& a, double x) { a = x; }
inline void set_val(Dual& a, double x) { a.v = x; }
inline void set_val(Rev& a, double x) { a.v = x; }

// value extraction used for every trajectory-dependent branch
inline double val(double a) { return a; }
inline double val(const Dual& a) { return a.v; }
inline double val(const Rev& a) { return a.v; }
using std::log; using std::exp; using std::sqrt;

}  // namespace orc
