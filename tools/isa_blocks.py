"""Static basic-block profile of one kernel of a model TU (development aid).
usage: python tools/isa_blocks.py hopper k_rollout_state [min_block]"""
import collections, re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
model, kern = sys.argv[1], sys.argv[2]
minb = int(sys.argv[3]) if len(sys.argv) > 3 else 30
src = os.path.join(ROOT, "optimization_dynamics_amd", "csrc", "od_model_%s.hip" % model)
out = "/tmp/isa_%s.s" % model
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src] + os.environ.get("EXTRA", "").split(),
                      stderr=subprocess.DEVNULL)
txt = open(out).read()
m = re.search(r"^(_ZN2od\d+%s\w+):" % kern, txt, re.M)
i = m.start(); j = txt.index(".Lfunc_end", i)
blocks = []; cur = ("entry", [])
for l in txt[i:j].split("\n"):
    mm = re.match(r"^(\.LBB\w+):", l)
    if mm:
        blocks.append(cur); cur = (mm.group(1), []); continue
    if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
        cur[1].append(l.strip())
blocks.append(cur)
print("total", sum(len(b[1]) for b in blocks))
keys = ("rcp", "rsq", "sqrt", "div_scale", "rndne", "cndmask", "accvgpr", "_f64", "readlane", "writelane", "global_", "flat_", "scratch_", "cmp")
for name, ins in blocks:
    if len(ins) >= minb:
        c = collections.Counter()
        for s in ins:
            op = s.split()[0]
            for k in keys:
                if k in op: c[k] += 1
        print(name, len(ins), dict(c))
k = re.search(r"\.vgpr_count:\s+(\d+)", txt[j:]); a = re.search(r"\.agpr_count:\s+(\d+)", txt[j:])
for key in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy"):
    mm = re.search(r"; %s: (\d+)" % key, txt[j:])
    if mm: print(key, mm.group(1))
