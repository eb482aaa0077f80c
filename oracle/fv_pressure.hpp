// TEST INFRASTRUCTURE — CPU oracle (see scalar.hpp).
// Restatement of model_tlmadm/fv_pressure.F90: compute_fv3_pressures (:23-72), _tlm (:74-135),
// _bwd (:137-203).  This is the one routine of the path whose reference Fortran builds from its own
// sources (it only needs utils/MAPL_Constants.F90), so the restatement is pinned against
// oracle/_ref/libfvpressure_ref.so (see oracle/ref/Makefile and tests/test_oracle_pin.py).
#pragma once
#include "arrays.hpp"

namespace orc {

// pe/peln are (i,k,j) in the reference; stored (i,j,k) here.  Only is..ie, js..je is defined (:36-44).
template <class T>
void compute_fv3_pressures(const Bounds& bd, int npz, double kappa, double ptop, const Arr3<T>& delp, Arr3<T>& pe,
                           Arr3<T>& pk, Arr3<T>& pkz, Arr3<T>& peln) {
  for (int j = bd.js; j <= bd.je; ++j)
    for (int i = bd.is; i <= bd.ie; ++i) {
      pe(i, j, 1) = T(ptop);
      for (int k = 2; k <= npz + 1; ++k) pe(i, j, k) = pe(i, j, k - 1) + delp(i, j, k - 1);
      for (int k = 1; k <= npz + 1; ++k) peln(i, j, k) = log(pe(i, j, k));
      for (int k = 1; k <= npz + 1; ++k) pk(i, j, k) = exp(kappa * peln(i, j, k));
      for (int k = 1; k <= npz; ++k)
        pkz(i, j, k) = (pk(i, j, k + 1) - pk(i, j, k)) / (kappa * (peln(i, j, k + 1) - peln(i, j, k)));
    }
}

}  // namespace orc
