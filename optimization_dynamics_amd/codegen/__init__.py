"""Offline model generator (sympy -> C / HIP device functions).

Counterpart of the reference's build-time code generation (`deps/build.jl:27-48`,
`src/models/*/codegen.jl`).  Run `python -m optimization_dynamics_amd.codegen` to regenerate
`optimization_dynamics_amd/csrc/gen/*.h` (product) and `oracle/gen/models_gen.h` (test oracle).
The generated files are committed; sympy is needed only to regenerate them.
"""
