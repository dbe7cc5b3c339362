// GENERATED -- registry of device models
#pragma once
#include "acrobot_impact.h"
#include "acrobot_nominal.h"
#include "cartpole_friction.h"
#include "cartpole_frictionless.h"
#include "planar_push.h"
#include "rocket_dynamics.h"
#include "rocket_projection.h"
#include "hopper.h"
#define OD_FOR_EACH_MODEL(X) \
  X(acrobot_impact) \
  X(acrobot_nominal) \
  X(cartpole_friction) \
  X(cartpole_frictionless) \
  X(planar_push) \
  X(rocket_dynamics) \
  X(rocket_projection) \
  X(hopper)
