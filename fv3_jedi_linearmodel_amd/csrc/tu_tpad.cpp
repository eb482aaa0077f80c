// translation unit of the split HIP build (core.h FV3LM_LINK): the fused adjoint of fv_tp_2d
#define FV3LM_IMPL_TPAD
#include "tpad.h"
