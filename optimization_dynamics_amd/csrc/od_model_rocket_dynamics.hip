#define OD_MODEL rocket_dynamics
#define OD_MODEL_MECH 0
#define OD_MODEL_FP32 1
#include "od_model_tu.inc"
