// translation unit of the split HIP build (core.h FV3LM_LINK): fv_tp_2d nonlinear / tangent, wavefront layout
#define FV3LM_IMPL_TP2
#include "tp2.h"
