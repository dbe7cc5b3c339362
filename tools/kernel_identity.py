"""Which device kernels changed since a given commit?  Compiles every translation unit of optimization_dynamics_amd/csrc for gfx950 at that
commit and in the working tree (hipcc -S, device only, the Makefile's flags) and compares each kernel's instruction stream (comments
stripped).  A kernel whose stream is identical runs the same machine code: what was measured / verified on the MI355X for it at that commit
still holds.  No GPU needed.     usage: python tools/kernel_identity.py <rev> [out.json]"""
import hashlib, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rev = sys.argv[1]


def streams(csrc, src):
    out = tempfile.mktemp(suffix=".s")
    extra = ["-fno-slp-vectorize"] if ("rocket" in src) else []          # (csrc/Makefile)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value"] + extra + ["-S", "--cuda-device-only", "-o", out, os.path.join(csrc, src)],
                          cwd=csrc, stderr=subprocess.DEVNULL)
    txt = open(out).read(); os.remove(out)
    res = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", txt, re.M | re.S):
        ins = [re.sub(r"\s*;.*$", "", l.strip()) for l in m.group(2).split("\n") if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")]
        res[m.group(1)] = (len(ins), hashlib.sha1("\n".join(ins).encode()).hexdigest()[:12])
    return res


old = tempfile.mkdtemp()
subprocess.check_call("git -C %s archive %s optimization_dynamics_amd/csrc include | tar -x -C %s" % (ROOT, rev, old), shell=True)
new_csrc, old_csrc = os.path.join(ROOT, "optimization_dynamics_amd", "csrc"), os.path.join(old, "optimization_dynamics_amd", "csrc")
rec = dict(against=subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", rev], text=True).strip(), units={}, changed=[])
tot = same = 0
for src in sorted(f for f in os.listdir(new_csrc) if f.endswith(".hip")):
    if not os.path.exists(os.path.join(old_csrc, src)):
        rec["units"][src] = "new translation unit"; continue
    a, b = streams(old_csrc, src), streams(new_csrc, src)
    ident = [k for k in b if a.get(k) == b[k]]
    tot += len(b); same += len(ident)
    rec["units"][src] = dict(kernels=len(b), identical=len(ident))
    for k in b:
        if a.get(k) != b[k]:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            rec["changed"].append(dict(kernel=name[:120], instructions_before=(a[k][0] if k in a else None), instructions_now=b[k][0]))
    print(src, rec["units"][src], flush=True)
rec["kernels"], rec["identical"] = tot, same
print(json.dumps(rec["changed"], indent=1))
if len(sys.argv) > 2:
    json.dump(rec, open(sys.argv[2], "w"), indent=1)
