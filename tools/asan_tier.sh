# The CPU test tier against an AddressSanitizer + UndefinedBehaviorSanitizer build of the product's sources (host emulation):
#   bash tools/asan_tier.sh [pytest args]      -> gpurun_out/asan/tier.log; prints the sanitizer findings (none expected)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/asan; mkdir -p $O
SRC=$R/optimization_dynamics_amd/csrc; E=$R/tests/host_emu
MODELS=$(grep -o "MODELS *=.*" $SRC/gen/models.mk | sed 's/MODELS *= *//')
FL="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -std=c++17 -fPIC -fopenmp -I$E -I$SRC -x c++ -ffp-contract=off -Wno-unknown-pragmas -Wno-attributes -DOD_TRACE -DOD_EMU_RCCL"
for m in $MODELS; do g++ $FL -c $SRC/od_model_$m.hip -o $O/od_model_$m.o & done
g++ $FL -c $SRC/od_rocket.hip -o $O/od_rocket.o &
g++ $FL -c $SRC/od_capi.hip -o $O/od_capi.o &
g++ $FL -c $E/emu_globals.cpp -o $O/emu_globals.o &
g++ -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -std=c++17 -fPIC -c $E/emu_rccl.cpp -o $O/emu_rccl.o &
wait
g++ -shared -fPIC -fopenmp -pthread -fsanitize=address,undefined -o $O/libod_emu_asan.so $O/*.o
# the checker too: the CPU oracle (oracle/ip_oracle.c + arbiter.c) under the same sanitizers
gcc -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fPIC -std=gnu99 -ffp-contract=off -fopenmp -shared -o $O/libod_oracle_asan.so $R/oracle/ip_oracle.c -lquadmath -lm
export OD_ORACLE_LIB=$O/libod_oracle_asan.so
cd $R
ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:detect_odr_violation=0 UBSAN_OPTIONS=print_stacktrace=1 \
  LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" OD_EMU_LIB=$O/libod_emu_asan.so \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect "tests/test_emu_parity.py::test_lane_cooperation_in_lockstep_rows" --deselect "tests/test_model_generator.py::test_add_a_ninth_model_emulated" "$@" > $O/tier.log 2>&1 || true
# (test_add_a_ninth_model_emulated builds its own oracle with a ninth model in a scratch copy; OD_ORACLE_LIB above would hand it this eight-model one)
# (deselected: with g++ 11 -O1 and -fsanitize=shift or =integer-divide-by-zero the harness's lockstep-row mode -- 16 host threads per DPP
# row -- hands lanes another copy's share of the step-length tests (coop_group read through a thread_local threadIdx): no sanitizer report, -O0
# with the same flags, the AddressSanitizer-only build, clang with -fsanitize=undefined -fsanitize-trap=undefined and clang with
# -ftrivial-auto-var-init=pattern all give the sequential build's numbers bit for bit; docs/DESIGN_LOG_r4-r5.md)
tail -3 $O/tier.log
echo "sanitizer findings:"; grep -c -E "ERROR: AddressSanitizer|runtime error:" $O/tier.log || true
grep -E "ERROR: AddressSanitizer|runtime error:" $O/tier.log | sort | uniq -c | sort -rn | head -20
