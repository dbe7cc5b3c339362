import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from optimization_dynamics_amd import dynamics as dyn, models
from oracle import oracle as O
print(torch.cuda.get_device_name(0))
rng=np.random.default_rng(3)
q=np.array([0,0.55,0,0.5])[:,None]
for B in [256, 4096, 65536, 262144]:
    q1=q+rng.normal(0,0.02,(4,B)); q2=q1+0.5*rng.normal(0,0.02,(4,B))
    U=np.array([0,9.81*3*0.5*0.05])[:,None]+rng.normal(size=(2,B))
    X=np.vstack([q1,q2])
    im=dyn.ImplicitDynamics(models.hopper,0.05,kappa_eval_tol=1e-4,kappa_grad_tol=1e-3)
    Xd=torch.tensor(X,device='cuda'); Ud=torch.tensor(U,device='cuda')
    D,DX,DU,st,it=im.step_grad(Xd,Ud); torch.cuda.synchronize()
    t0=time.time()
    for _ in range(5): D,DX,DU,st,it=im.step_grad(Xd,Ud)
    torch.cuda.synchronize(); dt=(time.time()-t0)/5
    print('B',B,'time %.3f ms'%(dt*1e3),'units/s %.3e'%(B/dt), 'status', torch.bincount(st,minlength=8).tolist(), 'iters', it.double().mean(1).tolist(), it.max(1).values.tolist())
    if B<=4096:
        sim=O.make_sim('hopper',0.05,kappa_tol=1e-4,kappa_grad_tol=1e-3)
        Do,DXo,DUo,bad=O.step_grad_batch(sim,X,U)
        G=np.concatenate([DX.cpu().numpy(),DU.cpu().numpy()],1).reshape(-1,B); Go=np.concatenate([DXo,DUo],1).reshape(-1,B)
        rel=np.abs(G-Go).max(0)/np.abs(Go).max(0)
        print('  state err',np.abs(D.cpu().numpy()-Do).max(),'grad rel max %.2e p99 %.2e med %.2e'%(rel.max(),np.quantile(rel,.99),np.median(rel)))
# rollout
B=4096; T=100
q1=q+rng.normal(0,0.02,(4,B)); x1=np.vstack([q1,q1])
U=np.array([0,9.81*3*0.5*0.05])[:,None,None]+rng.normal(size=(2,T,B))
im=dyn.ImplicitDynamics(models.hopper,0.05,kappa_eval_tol=1e-4,kappa_grad_tol=1e-3)
x1d=torch.tensor(x1,device='cuda'); Ud=torch.tensor(U,device='cuda')
Xr,A,Bm,st,it,out=im.rollout(x1d,Ud); torch.cuda.synchronize()
t0=time.time()
for _ in range(3): Xr,A,Bm,st,it,out=im.rollout(x1d,Ud,out=out)
torch.cuda.synchronize(); dt=(time.time()-t0)/3
print('rollout B=4096 T=100: %.2f ms, units/s %.3e'%(dt*1e3, B*T/dt),'status',torch.bincount(st.flatten(),minlength=8).tolist(),'iters mean',it.double().mean().item(),'max',it.max().item())
Xo,Ao,Bo,bad=O.rollout(O.make_sim('hopper',0.05,kappa_tol=1e-4,kappa_grad_tol=1e-3), x1[:,:64], U[:,:,:64])
print('rollout parity (64 traj): X err', np.abs(Xr.cpu().numpy()[:,:,:64]-Xo).max(), 'oracle bad',bad)
