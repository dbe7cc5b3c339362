"""Config 5's forward pass on the HOST BUILD (no GPU needed): interior-point trip counts of the thrust-cone projection on every candidate knot
of an iLQR iteration (inputs of examples/rocket.jl or the hover problem), and what lockstep -- 64 candidates per wavefront: sum over the 60 knots
of the max over the lanes -- makes of them under different assignments of candidates to wavefronts, among them the one the round-5 review
proposed (candidates sorted by their trip counts in the PREVIOUS iteration).  A launch of <= 1024 wavefronts ends with its slowest wavefront,
so the `max` column is what the kernel's time follows.   usage: python tools/lockstep_model_host.py [out.json]   (B, WHICH=example|hover, N0)"""
import json
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import numpy as np, torch
import ilqr_checks as C
import optimization_dynamics_amd as od
from optimization_dynamics_amd import interior_point as IP, _lib
lib=_lib.Library(ROOT+"/tests/host_emu/libod_emu.so"); dev="cpu"
B,T=int(os.environ.get("B","1024")),60
dt=torch.float64
which=os.environ.get("WHICH","example")
if which=="example": dyn,obj,x1,U0=C.config5_problem(lib,dev,B,dtype=dt)
else: dyn,obj,x1,U0=C.rocket_problem(lib,dev,B,T,dtype=dt,seed=1)
x1t,Ut=torch.tensor(x1),torch.tensor(U0)
sol=od.ILQR(dyn,obj,T)
na=sol.alphas.numel(); P=na*B
ipp=IP.InteriorPoint("rocket_projection",device=dev,lib=lib)
def trips(n0):
    d=sol.device_solver(B,max_iter=50,obj_tol=0.0)
    d.init(x1t,Ut); d.iterate(n0)
    X,U,J=d.get()
    X,A,Bm,st=sol.linearize(x1t,U)
    quad=obj.expansion(X,U,None,0.0)
    K,k,dV,bst=sol.backward(A,Bm,quad,1e-6)
    Xc,Uc,cst=sol.forward(x1t,X,U,K,k)
    Uk=Uc.double().reshape(3,T*P).contiguous(); nk=Uk.shape[1]
    z0=torch.tensor([0.1,0.1,1.1,0.1,0.1,0.1,0.0,0.1,0.1,1.1])[:,None].repeat(1,nk)
    zp,_,stp,itp=ipp.solve(z0,torch.cat([Uk,torch.full((1,nk),12.5,dtype=torch.float64)]),diff_sol=False)
    it=itp[0].reshape(T,na,B).double()
    # the solver's stall exit: a stalled solve is abandoned after ~ (iterations to reach alpha < 1e-9) + 4: modelled as min(it, 20)
    return it.clamp(max=float(os.environ.get("CLAMP","20")))
def cost(it_flat_TP, order):
    x=it_flat_TP[:,order]; N=x.shape[1]; nw=(N+63)//64
    y=torch.zeros(T,nw*64); y[:,:N]=x
    w=y.reshape(T,nw,64).max(2).values.sum(0)
    return w.mean().item(), w.max().item()
rec={}
for n0 in [int(t) for t in os.environ.get("N0","3,8").split(",")]:
    t=time.time(); a=trips(n0); b=trips(n0+1); print("trips computed %.0fs"%(time.time()-t))
    fa=a.reshape(T,P); fb=b.reshape(T,P)
    print("mean trips per knot %.2f; ideal (no lockstep) sum_t mean = %.1f"%(fb.mean().item(), fb.mean(1).sum().item()))
    shipped=torch.arange(P)
    byprob=torch.arange(P).reshape(na,B).t().reshape(-1)
    key_prev=fa.sum(0); key_self=fb.sum(0)
    for name,order in (("shipped: 64 consecutive problems of one step size",shipped),("all step sizes of a problem adjacent",byprob),
                       ("sorted by the previous iteration's total trips",torch.argsort(key_prev)),("sorted by this iteration's own total (bound)",torch.argsort(key_self)),
                       ("sorted by previous max trips",torch.argsort(fa.max(0).values*1000+key_prev))):
        m,M=cost(fb,order); print("  %-55s mean %7.1f  max %7.1f"%(name,m,M)); rec.setdefault("after_%d_iterations"%n0,{"mean_trips_per_knot":fb.mean().item(),"no_lockstep_sum_t_mean":fb.mean(1).sum().item()})[name]={"mean_over_wavefronts":m,"max_over_wavefronts":M}

if len(sys.argv) > 1:
    json.dump(dict(workload=which, problems=B, step_sizes=na, knots=T, stall_exit_modelled_as_trip_cap=float(os.environ.get("CLAMP","20")), assignments=rec), open(sys.argv[1], "w"), indent=1)
