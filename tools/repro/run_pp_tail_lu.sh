#!/bin/bash
# host sanitizer runs of the planar-push tail LU variants (see pp_tail_lu_host.cpp)
cd "$(dirname "$0")/../.."
INC="-Ioptimization_dynamics_amd/csrc -Itests/host_emu"
for v in "shipped:" "ballot+guard:-DOD_LU_BRANCHY_MAX=16" "execmask+noguard:-DOD_LU_BRANCHY_MAX=16 -DOD_LU_EXEC_MASKED_EXCHANGE -DOD_LU_NO_PIVOT_GUARD" \
         "execmask+guard:-DOD_LU_BRANCHY_MAX=16 -DOD_LU_EXEC_MASKED_EXCHANGE" "ballot+noguard:-DOD_LU_BRANCHY_MAX=16 -DOD_LU_NO_PIVOT_GUARD"; do
  n=${v%%:*}; f=${v#*:}
  g++ -O1 -g -std=c++17 -ffp-contract=off -fsanitize=undefined,address -fno-sanitize-recover=undefined \
      -Wall -Wuninitialized -Wmaybe-uninitialized -Wno-unknown-pragmas -Wno-unused-variable $INC $f tools/repro/pp_tail_lu_host.cpp -o /tmp/pp_tail_$$ 2> /tmp/pp_tail_warn_$$
  w=$(grep -c "uninitialized" /tmp/pp_tail_warn_$$)
  printf "%-18s g++ -fsanitize=undefined,address: %s uninitialized-warnings; " "$n" "$w"
  /tmp/pp_tail_$$ 2>&1 | tail -3 | tr '\n' ' '; echo
  # the same with locals pattern-initialised (clang): an uninitialised read would change the numbers
  /opt/rocm/lib/llvm/bin/clang++ -O2 -std=c++17 -ffp-contract=off -ftrivial-auto-var-init=pattern -Wno-unknown-pragmas $INC $f tools/repro/pp_tail_lu_host.cpp -o /tmp/pp_tail_$$ 2>/dev/null \
    && { printf "%-18s clang++ -ftrivial-auto-var-init=pattern:                  " ""; /tmp/pp_tail_$$ 2>&1 | tail -1; }
done
rm -f /tmp/pp_tail_$$ /tmp/pp_tail_warn_$$
