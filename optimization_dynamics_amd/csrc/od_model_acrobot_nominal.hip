#define OD_MODEL acrobot_nominal
#define OD_MODEL_MECH 1
#define OD_MODEL_FP32 0
#include "od_model_tu.inc"
