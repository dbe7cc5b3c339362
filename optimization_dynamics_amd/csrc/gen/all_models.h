// GENERATED -- every device model header
#pragma once
#include "acrobot_impact.h"
#include "acrobot_nominal.h"
#include "cartpole_friction.h"
#include "cartpole_frictionless.h"
#include "planar_push.h"
#include "rocket_dynamics.h"
#include "rocket_projection.h"
#include "hopper.h"
