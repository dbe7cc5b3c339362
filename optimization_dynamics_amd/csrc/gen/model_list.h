// GENERATED -- registry of device models: X(name, id)
#pragma once
#define OD_MODEL_COUNT 8
#define OD_FOR_EACH_MODEL(X) \
  X(acrobot_impact, 0) \
  X(acrobot_nominal, 1) \
  X(cartpole_friction, 2) \
  X(cartpole_frictionless, 3) \
  X(planar_push, 4) \
  X(rocket_dynamics, 5) \
  X(rocket_projection, 6) \
  X(hopper, 7)
