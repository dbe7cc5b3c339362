# Dry run of the GPU tier's test bodies on the HOST BUILD (no GPU): a copy of tests/ with the device strings turned to "cpu" and the gpu_lib
# fixture handing out tests/host_emu/libod_emu.so, then `pytest -m gpu`.  Finds Python-level mistakes and bars that do not hold on another
# realisation of the arithmetic before a GPU box does; tests that need the hardware itself (HIP graphs, streams, RCCL, /proc/self/maps of the
# HIP library, hipcc-built models) fail or skip here by construction -- read the list, not the count.
#   bash tools/gpu_tier_on_host.sh [pytest args]        -> gpurun_out/gpu_tier_on_host.txt
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d)
cp -r $R/tests $D/tests
make -C $R/tests/host_emu -j8 > /dev/null
rm -rf $D/tests/host_emu && ln -s $R/tests/host_emu $D/tests/host_emu
for d in oracle optimization_dynamics_amd examples bench.py bench_configs.py include julia tools __graft_entry__.py; do ln -s $R/$d $D/$d; done
sed -i 's/"cuda:0"/"cpu"/g; s/"cuda"/"cpu"/g; s/torch\.cuda\.synchronize([^)]*)/None/g' $D/tests/*.py
python - $D <<'PY'
import sys, re
p = sys.argv[1] + "/tests/conftest.py"
s = open(p).read()
s = s.replace('    import torch\n    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"\n    from optimization_dynamics_amd import _lib\n    return _lib.default_library()',
              '    from optimization_dynamics_amd import _lib\n    import os\n    return _lib.Library(os.environ.get("OD_EMU_LIB") or os.path.join(ROOT, "tests", "host_emu", "libod_emu.so"))')
open(p, "w").write(s)
PY
mkdir -p $R/gpurun_out
cd $D && python -m pytest tests -q -m gpu -p no:cacheprovider "$@" > $R/gpurun_out/gpu_tier_on_host.txt 2>&1
tail -40 $R/gpurun_out/gpu_tier_on_host.txt
rm -rf $D; rm -f /dev/shm/odemu_*
