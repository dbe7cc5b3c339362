# Measurement-only variant builds (never shipped): bash tools/build_variants.sh
# fused-gradient cost probe: the hopper TU with -DOD_EXPERIMENT_FUSED_GRAD_COST linked against the other objects
# The measurement switches (OD_EXPERIMENT_*, OD_LU_*) are NOT in the shipped translation units: tools/variants/experiment_switches.patch
# adds them to a copy of optimization_dynamics_amd/csrc under variants/src (git-ignored), and the variant objects are compiled there.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
make -C "$ROOT/optimization_dynamics_amd/csrc" -j8 > /dev/null
rm -rf "$ROOT/variants/src" && mkdir -p "$ROOT/variants/src" "$ROOT/variants/include"
cp -r "$ROOT/optimization_dynamics_amd/csrc/"*.h "$ROOT/optimization_dynamics_amd/csrc/"*.inc "$ROOT/optimization_dynamics_amd/csrc/"*.hip "$ROOT/optimization_dynamics_amd/csrc/gen" "$ROOT/variants/src/"
mkdir -p "$ROOT/variants/src/build" && cp "$ROOT/optimization_dynamics_amd/csrc/build/"*.o "$ROOT/variants/src/build/"
(cd "$ROOT/variants/src" && patch -p1 -s < "$ROOT/tools/variants/experiment_switches.patch")
sed -i 's#"../../include/od_mi355x.h"#"'"$ROOT"'/include/od_mi355x.h"#' "$ROOT/variants/src/od_capi.hip"
cd "$ROOT/variants/src"
mkdir -p ../build_fg
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -DOD_EXPERIMENT_FUSED_GRAD_COST -c od_model_hopper.hip -o ../build_fg/od_model_hopper.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libod_fusedgradcost.so $(ls build/*.o | grep -v od_model_hopper) ../build_fg/od_model_hopper.o
echo "variants/libod_fusedgradcost.so: python tools/time_rollout.py variants/libod_fusedgradcost.so 0 4096 100"
# row-decoupling probe: per-row progress in the 16-lane cooperative rollout
mkdir -p ../build_rd
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -DOD_EXPERIMENT_ROW_DECOUPLING -c od_model_hopper.hip -o ../build_rd/od_model_hopper.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libod_rowdecoupled.so $(ls build/*.o | grep -v od_model_hopper) ../build_rd/od_model_hopper.o
echo "variants/libod_rowdecoupled.so: python tools/time_rollout.py variants/libod_rowdecoupled.so 0 4096 100"
# lane-parallel line search in the 8-lane form (the shipped build tries every step in turn)
mkdir -p ../build_c3pls
for m in hopper planar_push; do
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -DOD_EXPERIMENT_C3_PARALLEL_LS -c od_model_$m.hip -o ../build_c3pls/od_model_$m.o &
done; wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libod_c3pls.so $(ls build/*.o | grep -v "od_model_hopper\|od_model_planar_push") ../build_c3pls/od_model_hopper.o ../build_c3pls/od_model_planar_push.o
echo "variants/libod_c3pls.so: OD_LIB=variants/libod_c3pls.so python tools/sweep_pp.py"
# iteration counts of the rocket's two solves in the status word of the rollout kernels (tools/diag_config5_rollout.py)
mkdir -p ../build_its
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -fno-slp-vectorize -DOD_EXPERIMENT_ITERS_IN_STATUS -c od_rocket.hip -o ../build_its/od_rocket.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libod_itstat.so $(ls build/*.o | grep -v od_rocket) ../build_its/od_rocket.o
echo "variants/libod_itstat.so: OD_LIB=variants/libod_itstat.so python tools/diag_config5_rollout.py 3"
# the rocket rollout kernels held to 2 / 3 wavefronts per SIMD (256 / 168 registers; tools/sweep_rocket_ppw.py): slower than the
# 358-register build at every mapping
for occ in 2 3; do
mkdir -p ../build_occ$occ
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -fno-slp-vectorize -DOD_EXPERIMENT_ROLLOUT_OCC=$occ -c od_rocket.hip -o ../build_occ$occ/od_rocket.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libod_rollout_occ$occ.so $(ls build/*.o | grep -v od_rocket) ../build_occ$occ/od_rocket.o
done
echo "variants/libod_rollout_occ2.so: OD_LIB=variants/libod_rollout_occ2.so python tools/sweep_rocket_ppw.py"
# the matrix-core Riccati kernel (csrc/od_ilqr_mfma.inc) without its 3 x 3 phase (1) / without the loads of the knot loop (2),
# and with the trajectories per workgroup taken from OD_ILM_W (w): OD_LIB=variants/libod_ilmw.so OD_ILM_W=8 python tools/time_backward.py
for v in 1 2 w; do
mkdir -p ../build_ilm$v
if [ $v = w ]; then D="-DOD_EXPERIMENT_ILM_W"; else D="-DOD_EXPERIMENT_ILM=$v -DOD_EXPERIMENT_ILM_W"; fi
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value $D -c od_capi.hip -o ../build_ilm$v/od_capi.o &
done; wait
for v in 1 2 w; do
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libod_ilm$v.so $(ls build/*.o | grep -v od_capi) ../build_ilm$v/od_capi.o
done
echo "variants/libod_ilm{1,2,w}.so: OD_LIB=... OD_ILM_W=4|8|16 OD_BS=64,4096 python tools/time_backward.py"
