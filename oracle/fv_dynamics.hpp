// TEST INFRASTRUCTURE — CPU oracle (see scalar.hpp).  fv_dynamics / tracer_2d / fv_mapz restatement.
#pragma once
#include "dyn_core.hpp"

namespace orc {
struct RemapOpts { int kord_tm = -17, kord_mt = 17, kord_wz = 17, kord_tr = 17; };
}  // namespace orc
