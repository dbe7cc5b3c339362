"""Critical path of one basic block of a kernel (development aid): longest chain of register dependences with the latencies a
LONE wavefront sees on gfx950 (profiles/r3_ubench_op_latency.txt: dependent fp64 VALU 8.3 cycles, issue 4; v_rcp / v_rsq_f64 20.4;
v_cndmask_b32 10.1 after a compare; DPP after s_nop 16.4), against the issue-bound time of the block.
usage: python tools/isa_critpath.py hopper 'k_rollout_state_coop\\w+' .LBB5_41 [-v]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
model, kern, block = sys.argv[1], sys.argv[2], sys.argv[3]
verbose = "-v" in sys.argv
csrc = os.path.join(ROOT, "optimization_dynamics_amd", "csrc")
out = "/tmp/isa_g_%s.s" % model
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-gline-tables-only", "-S", "--cuda-device-only", "-o", out,
                       os.path.join(csrc, "od_rocket.hip" if model == "rocket" else "od_model_%s.hip" % model)] + os.environ.get("EXTRA", "").split(), stderr=subprocess.DEVNULL, cwd=csrc)
txt = open(out).read()
files = {int(m.group(1)): (m.group(3) or m.group(2)) for m in re.finditer(r'\.file\t(\d+) "([^"]*)"(?: "([^"]*)")?', txt)}
m = re.search(r"^(_ZN2od\d+%s\w*):" % kern, txt, re.M)
i = m.start(); j = txt.index(".Lfunc_end", i)
cur, loc, ins = "entry", (0, 0), []
for l in txt[i:j].split("\n"):
    mm = re.match(r"^(\.LBB\w+):", l)
    if mm: cur = mm.group(1); continue
    mm = re.match(r"\s*\.loc\s+(\d+) (\d+)", l)
    if mm: loc = (int(mm.group(1)), int(mm.group(2))); continue
    if cur == block and l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;") and l.split():
        ins.append((l.strip().split(";")[0].strip(), loc))

def regs(tok):
    tok = tok.strip()
    tok = re.sub(r"^-|^\||\|$|^neg\(|^abs\(|\)$", "", tok)
    mm = re.match(r"^([vsa])\[(\d+):(\d+)\]$", tok)
    if mm: return ["%s%d" % (mm.group(1), k) for k in range(int(mm.group(2)), int(mm.group(3)) + 1)]
    mm = re.match(r"^([vsa])(\d+)$", tok)
    if mm: return [tok]
    if tok in ("vcc", "exec", "scc", "vcc_lo", "vcc_hi", "exec_lo", "exec_hi"): return [tok[:3] if tok.startswith(("vcc", "exe")) else tok]
    return []

def lat_issue(op, text):
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")): return 24.0, 16.0
    if "dpp" in text or "row_" in text: return 12.0, 8.0 if "_f64" in op else 4.0
    if op.startswith("v_cndmask"): return 6.0, 4.0
    if op.startswith("v_") and "_f64" in op: return 8.3, 4.0
    if op.startswith(("v_readlane", "v_readfirstlane")): return 8.0, 4.0
    if op.startswith("v_writelane"): return 8.0, 4.0
    if op.startswith(("v_accvgpr", "v_mov", "v_cmp", "v_and", "v_or", "v_xor", "v_add_u", "v_lshl", "v_cvt", "v_bfe", "v_sub_u", "v_add_co", "v_mul_lo", "v_mul_hi")): return 5.0, 4.0
    if op.startswith("s_nop"): return 0.0, 2.0
    if op.startswith("s_"): return 4.0, 1.0
    if op.startswith(("global_load", "flat_load", "ds_read", "scratch_load", "s_load")): return 500.0, 4.0
    return 5.0, 4.0

ready = collections.defaultdict(float)       # register -> time its value is available
who = {}                                     # register -> index of the producing instruction
t_done, parent = [], []
issue_sum = 0.0
for k, (text, loc) in enumerate(ins):
    op = text.split()[0]
    ops = [o for o in re.split(r",\s*", text[len(op):].strip()) if o]
    ops = [o.split(" ")[0] for o in ops]
    lat, iss = lat_issue(op, text)
    issue_sum += iss
    is_store = op.startswith(("global_store", "flat_store", "ds_write", "scratch_store", "s_waitcnt", "s_cbranch", "s_branch", "s_nop", "s_barrier"))
    dst = [] if is_store or not ops else regs(ops[0])
    if op.startswith("v_cmp") and not regs(ops[0]): dst = ["vcc"]
    src_ops = ops if is_store else ops[1:]
    src = [r for o in src_ops for r in regs(o)]
    if op.startswith(("v_fmac", "v_mac", "v_writelane")) or "dpp" in text: src += dst          # (old value of the destination is read)
    if op.startswith("v_cndmask") and len(ops) == 3: src.append("vcc")
    if op.startswith(("s_cbranch_vcc", "s_and_saveexec")): src.append("vcc")
    if op.startswith("v_div_fmas"): src.append("vcc")
    start, par = 0.0, None
    for r in src:
        if ready[r] > start: start, par = ready[r], who.get(r)
    t_done.append(start + lat); parent.append(par)
    for r in dst: ready[r] = start + lat; who[r] = k
# in-order issue of the compiler's order, and of a greedy list schedule of the same instructions under RAW + WAR + WAW constraints
def parse(k):
    text = ins[k][0]; op = text.split()[0]
    ops = [o.split(" ")[0] for o in re.split(r",\s*", text[len(op):].strip()) if o]
    is_store = op.startswith(("global_store", "flat_store", "ds_write", "scratch_store", "s_waitcnt", "s_cbranch", "s_branch", "s_nop", "s_barrier"))
    dst = [] if is_store or not ops else regs(ops[0])
    if op.startswith("v_cmp") and not regs(ops[0]): dst = ["vcc"]
    src = [r for o in (ops if is_store else ops[1:]) for r in regs(o)]
    if op.startswith(("v_fmac", "v_mac", "v_writelane")) or "dpp" in text: src += dst
    if op.startswith("v_cndmask") and len(ops) == 3: src.append("vcc")
    if op.startswith(("s_cbranch_vcc", "s_and_saveexec")): src.append("vcc")
    if op.startswith("s_cbranch_scc"): src.append("scc")
    if op.startswith(("s_cmp", "s_and_b", "s_or_b", "s_add", "s_sub", "s_lshl")): dst = dst + ["scc"]
    fixed = op.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_barrier", "s_nop", "s_mov_b64 exec", "s_and_saveexec", "s_or_saveexec", "global_", "flat_", "ds_", "scratch_", "v_readlane", "v_writelane", "v_readfirstlane")) or "exec" in text
    return op, src, dst, fixed
P = [parse(k) for k in range(len(ins))]
def inorder(order):
    rd = collections.defaultdict(float); t = 0.0
    for k in order:
        op, src, dst, fixed = P[k]
        lat, iss = lat_issue(op, ins[k][0])
        st = max([t] + [rd[r] for r in src])
        t = st + iss
        for r in dst: rd[r] = st + lat
    return max([t] + list(rd.values()))
n = len(ins)
print("in-order issue of the compiler's schedule: %.0f cycles" % inorder(range(n)))
preds = [set() for _ in range(n)]
lastw, lastr, lastfixed = {}, collections.defaultdict(list), None
for k in range(n):
    op, src, dst, fixed = P[k]
    for r in src:
        if r in lastw: preds[k].add(lastw[r])
    for r in dst:
        if r in lastw: preds[k].add(lastw[r])
        for q in lastr[r]: preds[k].add(q)
    if fixed:
        for q in range(k): preds[k].add(q)          # (memory, exec, lane-exchange and control instructions keep their place)
        lastfixed = k
    elif lastfixed is not None: preds[k].add(lastfixed)
    for r in dst: lastw[r] = k; lastr[r] = []
    for r in src: lastr[r].append(k)
    preds[k].discard(k)
# priority = longest latency path to the end
succs = [[] for _ in range(n)]
for k in range(n):
    for q in preds[k]: succs[q].append(k)
prio = [0.0] * n
for k in range(n - 1, -1, -1):
    prio[k] = lat_issue(P[k][0], ins[k][0])[0] + max([prio[q] for q in succs[k]] + [0.0])
done, order, rd, t = set(), [], collections.defaultdict(float), 0.0
npred = [len(p) for p in preds]
avail = [k for k in range(n) if npred[k] == 0]
while avail:
    def start(k): return max([t] + [rd[r] for r in P[k][1]])
    k = min(avail, key=lambda k: (start(k), -prio[k]))
    avail.remove(k)
    st = start(k); lat, iss = lat_issue(P[k][0], ins[k][0]); t = st + iss
    for r in P[k][2]: rd[r] = st + lat
    order.append(k)
    for q in succs[k]:
        npred[q] -= 1
        if npred[q] == 0: avail.append(q)
assert len(order) == n
print("greedy list schedule of the same instructions (registers as allocated): %.0f cycles" % inorder(order))
end = max(range(len(ins)), key=lambda k: t_done[k])
print("%s: %d instructions, issue-bound %.0f cycles, critical path %.0f cycles" % (block, len(ins), issue_sum, t_done[end]))
# walk the critical path and attribute it to source functions
path = []
k = end
while k is not None:
    path.append(k); k = parent[k]
path.reverse()
def func_ranges(path_):
    res = []
    try: lines = open(path_).read().split("\n")
    except OSError: return res
    for i_, l in enumerate(lines):
        mm = re.match(r"\s*(?:template <[^>]*>\s*)?(?:OD_HD|__device__|static|inline|__host__|__forceinline__|constexpr|\s)*[\w:<>,\s\*&]*?\b(\w+)\s*\([^;]*$", l)
        if mm and not l.strip().startswith(("if", "for", "while", "return", "//", "#", "else", "switch")) and ("{" in l or (i_ + 1 < len(lines) and "{" in lines[i_ + 1]) or l.rstrip().endswith(",")):
            res.append((i_ + 1, mm.group(1)))
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
acc = collections.Counter(); cnt = collections.Counter()
prev = 0.0
for k in path:
    f = func_of(*ins[k][1])
    acc[f] += t_done[k] - prev; cnt[f] += 1; prev = t_done[k]
print("critical path: %d instructions" % len(path))
for f, v in acc.most_common(20): print("   %7.0f cycles  %3d instr  %s" % (v, cnt[f], f))
if verbose:
    prev = 0.0
    for k in path:
        print("%7.0f  +%5.1f  %-60s %s:%d" % (t_done[k], t_done[k] - prev, ins[k][0][:60], files.get(ins[k][1][0], "?"), ins[k][1][1])); prev = t_done[k]
