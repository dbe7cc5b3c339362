"""Static check of the DPP read-after-VALU-write hazard in compiled kernels (gfx950: NOT interlocked, tools/ubench/dpp_hazard.hip):
a DPP instruction that reads VGPR v through its DPP operand needs >= 2 wait states after the VALU instruction that wrote v
(an s_nop N counts N + 1, any other instruction 1).  Scans every *_dpp instruction of the given model's kernels, within basic
blocks (conservative at block entries: the two instructions before a label are not looked through).
usage: python tools/isa_dpp_hazard_check.py planar_push [hopper ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    """v3 -> {3}; v[4:5] -> {4, 5}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check(model):
    src = model if model.endswith(".hip") else os.path.join(ROOT, "optimization_dynamics_amd", "csrc", "od_model_%s.hip" % model)
    model = os.path.basename(model).replace(".hip", "")
    out = "/tmp/isa_hz_%s.s" % model
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    kernel, ndpp, bad, hist = None, 0, [], []
    per_kernel = {}
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, hist = m.group(1), []
            continue
        if re.match(r"^\.LBB", line):
            hist = []                      # block entry: unknown predecessors -> nothing to compare with (the compiler's own
            continue                       # hazard recognizer handles cross-block cases; inline asm carries its own s_nop)
        if not line.startswith("\t") or line.startswith("\t.") or line.startswith("\t;"):
            continue
        ins = line.strip()
        op = ins.split()[0]
        args = [a.strip() for a in ins[len(op):].split(",")]
        if "_dpp" in op and kernel:
            ndpp += 1
            per_kernel[kernel] = per_kernel.get(kernel, 0) + 1
            # the DPP operand is src0: the first source (v_mov: args[1]; v_fmac / v_max ...: args[1])
            src0 = regs(args[1].split()[0]) if len(args) > 1 else set()
            wait = 0
            for pop, pdst in reversed(hist[-4:]):
                if pop == "s_nop":
                    wait += pdst + 1
                    continue
                if pop.startswith("v_") and (pdst & src0) and wait < 2:
                    bad.append((kernel, ins, pop))
                wait += 1
                if wait >= 2:
                    break
        if op == "s_nop":
            hist.append(("s_nop", int(args[0])))
        else:
            hist.append((op, regs(args[0].split()[0]) if args and args[0] else set()))
    return ndpp, bad, per_kernel


if __name__ == "__main__":
    for model in sys.argv[1:] or ["planar_push", "hopper"]:
        n, bad, pk = check(model)
        print("%s: %d DPP instructions in %d kernels, %d with fewer than 2 wait states after a VALU write of their DPP operand" % (model, n, len(pk), len(bad)))
        for k, ins, pop in bad[:20]:
            print("   ", k[:60], "|", ins, "| after", pop)
