# GENERATED -- model translation units of the library (one od_model_<name>.hip each)
MODELS = acrobot_impact acrobot_nominal cartpole_friction cartpole_frictionless planar_push rocket_dynamics rocket_projection hopper
