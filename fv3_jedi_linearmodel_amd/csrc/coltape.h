// fv3lm-hip: reverse mode for column (k-sequential) operators without hand-written adjoints.
//
// The stencil stages get their adjoint from the gather launcher (exec.h).  Column operators — the non-hydrostatic
// solvers with their Thomas sweeps — are sequential in k and have hundreds of dependent statements per column, so they
// are written once on a generic scalar and run three ways by the same kernel:
//   nonlinear  T = double
//   tangent    T = Dual
//   adjoint    T = TV, a taped scalar: every arithmetic operation appends (operand ids, partial derivatives) to the
//              thread's tape, laid out [entry][column] in global memory so that the lanes of a wave write and read
//              contiguous rows; after the forward replay the same thread walks its tape backwards.  Inputs are leaves
//              whose adjoint is added to the field's adjoint buffer; an output store moves the field's incoming adjoint onto
//              the stored variable.  This is the role utils/tapenade/adStack.c plays for the reference, confined to the
//              column operators and resident in HBM.
#pragma once
#include "remap.h"

namespace fv3 {

struct TapePart { double a, b; };
struct TapeIdx { int a, b; };
struct TapeMem {
  TapePart* part = nullptr; // [cap][stride] the two partial derivatives of an entry (one 16-byte store per lane)
  TapeIdx* idx = nullptr;   // [cap][stride] operand ids (>= 0 variable, -1 none, <= -2 leaf: field slot, level)
  double* adj = nullptr;    // [cap][stride]
  int* overflow = nullptr;  // device flag
  size_t stride = 0; int cap = 0;
};
struct Tape {
  TapeMem m; size_t col; int n;
  HD int push(int ia, int ib, double pa, double pb) {
    if (n >= m.cap) { *m.overflow = 1; return -1; }
    const int id = n++;
    const size_t e = (size_t)id * m.stride + col;
    m.part[e] = TapePart{pa, pb}; m.idx[e] = TapeIdx{ia, ib};
    m.adj[e] = 0.;
    return id;
  }
  HD double& ad(int id) const { return m.adj[(size_t)id * m.stride + col]; }
};
// A taped value is (value, scale, id): its derivative with respect to tape variable `id` is `scale`.  Sums with and multiples
// of untaped numbers only change value and scale, so they cost no tape entry.
struct TV {
  double v, s; int id; Tape* t;
  HD TV() : v(0.), s(1.), id(-1), t(nullptr) {}
  HD TV(double v_) : v(v_), s(1.), id(-1), t(nullptr) {}
  HD TV(double v_, double s_, int id_, Tape* t_) : v(v_), s(s_), id(id_), t(t_) {}
};
HD TV tv2(const TV& a, const TV& b, double v, double pa, double pb) {     // pa, pb: partials with respect to the VALUES of a, b
  if (a.id >= 0 && b.id >= 0) return TV(v, 1., a.t->push(a.id, b.id, pa * a.s, pb * b.s), a.t);
  if (a.id >= 0) return TV(v, pa * a.s, a.id, a.t);
  if (b.id >= 0) return TV(v, pb * b.s, b.id, b.t);
  return TV(v);
}
HD TV tv1n(const TV& a, double v, double pa) {                            // nonlinear function of one taped value: always an entry
  if (a.id < 0) return TV(v);
  return TV(v, 1., a.t->push(a.id, -1, pa * a.s, 0.), a.t);
}
HD TV tv1l(const TV& a, double v, double pa) { return a.id < 0 ? TV(v) : TV(v, pa * a.s, a.id, a.t); }   // affine: folded into the scale
HD TV operator+(const TV& a, const TV& b) { return tv2(a, b, a.v + b.v, 1., 1.); }
HD TV operator-(const TV& a, const TV& b) { return tv2(a, b, a.v - b.v, 1., -1.); }
HD TV operator*(const TV& a, const TV& b) { return tv2(a, b, a.v * b.v, b.v, a.v); }
HD TV operator/(const TV& a, const TV& b) {
  const double r = 1. / b.v, q = a.v * r;
  if (b.id < 0) return tv1l(a, q, r);
  if (a.id < 0) return tv1n(b, q, -q * r);
  return tv2(a, b, q, r, -q * r);
}
HD TV operator-(const TV& a) { return tv1l(a, -a.v, -1.); }
HD TV operator+(const TV& a, double b) { return tv1l(a, a.v + b, 1.); }
HD TV operator+(double a, const TV& b) { return tv1l(b, a + b.v, 1.); }
HD TV operator-(const TV& a, double b) { return tv1l(a, a.v - b, 1.); }
HD TV operator-(double a, const TV& b) { return tv1l(b, a - b.v, -1.); }
HD TV operator*(const TV& a, double b) { return tv1l(a, a.v * b, b); }
HD TV operator*(double a, const TV& b) { return tv1l(b, a * b.v, a); }
HD TV operator/(const TV& a, double b) { return tv1l(a, a.v / b, 1. / b); }
HD TV operator/(double a, const TV& b) { const double q = a / b.v; return tv1n(b, q, -q / b.v); }
HD TV dlog(const TV& a) { return tv1n(a, log(a.v), 1. / a.v); }
HD TV dexp(const TV& a) { const double e = exp(a.v); return tv1n(a, e, e); }
HD double val(const TV& a) { return a.v; }
HD void set_val(TV& a, double x) { a.v = x; }      // the derivative stays (split_kord: values by the limited profile)
// scale 1 form (what the workspace stores)
HD TV tv_unit(const TV& x) { return (x.id < 0 || x.s == 1.) ? x : TV(x.v, 1., x.t->push(x.id, -1, x.s, 0.), x.t); }

// workspace storage of a taped scalar: value + id (as a double)
template <> struct WsIO<TV> {
  static constexpr int W = 2;
  HD static TV get(const ColWs& w, int s, int k, Tape* t) { return TV(w.at(2 * s, k), 1., (int)w.at(2 * s + 1, k), t); }
  HD static void set(const ColWs& w, int s, int k, const TV& x0) { const TV x = tv_unit(x0); w.at(2 * s, k) = x.v; w.at(2 * s + 1, k) = (double)x.id; }
};

// ---- the three ways a column operator touches its fields: f[slot] element (i, j, k) of the thread's column
struct ColNL {
  typedef double T;
  const Geom& g; const Fld* f; int tile, i, j; Tape* tape;
  HD double ld(int slot, int k) const { return f[slot].t[fidx(g, f[slot], tile, i, j, k)]; }
  HD void st(int slot, int k, double x) const { f[slot].t[fidx(g, f[slot], tile, i, j, k)] = x; }
  HD double wget(const ColWs& w, int s, int k) const { return w.at(s, k); }
  HD void wset(const ColWs& w, int s, int k, double x) const { w.at(s, k) = x; }
  HD double cst(double x) const { return x; }
};
struct ColTL {
  typedef Dual T;
  const Geom& g; const Fld* f; int tile, i, j; Tape* tape;
  HD Dual ld(int slot, int k) const { const size_t n = fidx(g, f[slot], tile, i, j, k); return Dual(f[slot].t[n], f[slot].p[n]); }
  HD void st(int slot, int k, const Dual& x) const { const size_t n = fidx(g, f[slot], tile, i, j, k); f[slot].t[n] = x.v; f[slot].p[n] = x.d; }
  HD Dual wget(const ColWs& w, int s, int k) const { return WsIO<Dual>::get(w, s, k); }
  HD void wset(const ColWs& w, int s, int k, const Dual& x) const { WsIO<Dual>::set(w, s, k, x); }
  HD Dual cst(double x) const { return Dual(x); }
};
struct ColAD {
  typedef TV T;
  const Geom& g; const Fld* f; int tile, i, j; Tape* tape;
  HD TV ld(int slot, int k) const { return TV(f[slot].t[fidx(g, f[slot], tile, i, j, k)], 1., tape->push(-2 - slot, k, 0., 0.), tape); }
  // trajectory is NOT rewritten (the backward sweep must not disturb it); the field's incoming adjoint moves onto the variable
  HD void st(int slot, int k, const TV& x) const {
    const size_t n = fidx(g, f[slot], tile, i, j, k);
    if (x.id >= 0) tape->ad(x.id) += x.s * f[slot].p[n];
    f[slot].p[n] = 0.;
  }
  HD TV wget(const ColWs& w, int s, int k) const { return WsIO<TV>::get(w, s, k, tape); }
  HD void wset(const ColWs& w, int s, int k, const TV& x) const { WsIO<TV>::set(w, s, k, x); }
  HD TV cst(double x) const { return TV(x); }
  // walk the tape backwards; leaves hand their adjoint to the fields
  HD void reverse() const {
    for (int id = tape->n - 1; id >= 0; --id) {
      const size_t e = (size_t)id * tape->m.stride + tape->col;
      const double a = tape->ad(id);
      const TapeIdx ix = tape->m.idx[e];
      const int ia = ix.a, ib = ix.b;
      if (ia <= -2) { const int slot = -2 - ia; f[slot].p[fidx(g, f[slot], tile, i, j, ib)] += a; continue; }
      if (a == 0.) continue;
      const TapePart pt = tape->m.part[e];
      if (ia >= 0) tape->ad(ia) += pt.a * a;
      if (ib >= 0) tape->ad(ib) += pt.b * a;
    }
  }
};

// array of T living in the column workspace, 1-based level index
template <class IO>
struct WArr {
  const IO& io; const ColWs& ws; int slot;
  HD typename IO::T operator()(int k) const { return io.wget(ws, slot, k); }
  HD void set(int k, const typename IO::T& x) const { io.wset(ws, slot, k, x); }
};

}  // namespace fv3
