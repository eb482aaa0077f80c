// translation unit of the split HIP build (core.h FV3LM_LINK): the fused fv_tp_2d kernels and their launch functions
#define FV3LM_IMPL_TPFUSED
#include "tpfused.h"
