/* GENERATED -- registry of oracle models (index = model id) */
#include "acrobot_impact.h"
#include "acrobot_nominal.h"
#include "cartpole_friction.h"
#include "cartpole_frictionless.h"
#include "planar_push.h"
#include "rocket_dynamics.h"
#include "rocket_projection.h"
#include "hopper.h"
static const od_oracle_model* const od_oracle_models[] = {
  &acrobot_impact_model,
  &acrobot_nominal_model,
  &cartpole_friction_model,
  &cartpole_frictionless_model,
  &planar_push_model,
  &rocket_dynamics_model,
  &rocket_projection_model,
  &hopper_model,
};
static const int od_oracle_num_models = 8;
