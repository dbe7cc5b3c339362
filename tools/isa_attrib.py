"""Attribute the instructions of a kernel's basic blocks to source functions (development aid; needs line tables).
usage: python tools/isa_attrib.py hopper 'k_rollout_state_coop\w+Li1E' [block ...]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
model, kern = sys.argv[1], sys.argv[2]
want = set(sys.argv[3:])
csrc = os.path.join(ROOT, "optimization_dynamics_amd", "csrc")
out = "/tmp/isa_g_%s.s" % model
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-gline-tables-only", "-S", "--cuda-device-only", "-o", out,
                       os.path.join(csrc, "od_rocket.hip" if model == "rocket" else "od_model_%s.hip" % model)], stderr=subprocess.DEVNULL, cwd=csrc)
txt = open(out).read()
files = {int(m.group(1)): (m.group(3) or m.group(2)) for m in re.finditer(r'\.file\t(\d+) "([^"]*)"(?: "([^"]*)")?', txt)}
# function line ranges per file: a crude scan for definitions at brace depth <= 2
def func_ranges(path):
    res = []
    try: lines = open(path).read().split("\n")
    except OSError: return res
    for i, l in enumerate(lines):
        m = re.match(r"\s*(?:template <[^>]*>\s*)?(?:OD_HD|__device__|static|inline|__host__|__forceinline__|constexpr|\s)*[\w:<>,\s\*&]*?\b(\w+)\s*\([^;]*$", l)
        if m and not l.strip().startswith(("if", "for", "while", "return", "//", "#", "else", "switch")) and ("{" in l or (i + 1 < len(lines) and "{" in lines[i + 1]) or l.rstrip().endswith(",")):
            res.append((i + 1, m.group(1)))
    return res
ranges = {}
for fid, name in files.items():
    for d in (csrc, os.path.join(csrc, "gen")):
        p = os.path.join(d, name)
        if os.path.exists(p): ranges[fid] = func_ranges(p); break
def func_of(fid, line):
    best = files.get(fid, "?")
    for ln, fn in ranges.get(fid, []):
        if ln <= line: best = "%s:%s" % (files[fid], fn)
        else: break
    return best
m = re.search(r"^(_ZN2od\d+%s\w*):" % kern, txt, re.M)
i = m.start(); j = txt.index(".Lfunc_end", i)
cur = "entry"; loc = (0, 0)
acc = collections.defaultdict(collections.Counter)
for l in txt[i:j].split("\n"):
    mm = re.match(r"^(\.LBB\w+):", l)
    if mm: cur = mm.group(1); continue
    mm = re.match(r"\s*\.loc\s+(\d+) (\d+)", l)
    if mm: loc = (int(mm.group(1)), int(mm.group(2))); continue
    if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
        acc[cur][func_of(*loc)] += 1
for b, c in acc.items():
    n = sum(c.values())
    if (want and b in want) or (not want and n >= 100):
        print("==", b, n)
        for k, v in c.most_common(40): print("   %4d  %s" % (v, k))
