// GENERATED -- registry of constraint functions: X(name, id)
#pragma once
#include "con_hopper_foot.h"
#define OD_CONSTRAINT_COUNT 1
#define OD_FOR_EACH_CONSTRAINT(X) \
  X(hopper_foot, 0)
