"""agreement rate (status and both iteration counts equal) between the cooperative and the lane-per-problem kernels under
each of the edge options of tests/test_coop.py / test_coop3.py, over several seeds -> the thresholds stored in the tests"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import workloads as W, parity_checks as P
import test_coop as TC
import optimization_dynamics_amd as od
lib = od.default_library(); dev = "cuda:0"
out = {}
for name, modes in (("hopper", (1, 2)), ("cartpole_friction", (1, 2)), ("acrobot_impact", (1, 2)), ("hopper", (1, 3)), ("planar_push", (1, 3))):
    for kw in TC.EDGE_OPTIONS:
        rates = []
        for seed in range(7, 27):
            X, U = W.knots(name, 96, seed=seed)
            Xd, Ud = torch.tensor(X, device=dev), torch.tensor(U, device=dev)
            o = []
            for mode in modes:
                im = P.make_im(name, lib, dev); im.set_options(**kw); im.set_cooperative(mode)
                o.append([t.cpu().numpy() for t in im.step_grad(Xd, Ud)])
            rates.append(float(((o[0][3] == o[1][3]) & (o[0][4] == o[1][4]).all(0)).mean()))
        key = "%s/%s/%s" % (name, "8lane" if modes[1] == 3 else "16lane", ",".join("%s=%g" % kv for kv in kw.items()))
        out[key] = dict(min=min(rates), mean=float(np.mean(rates)))
        print(key, out[key], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "edge_rates.json"), "w"), indent=1)
