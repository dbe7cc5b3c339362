import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import workloads as W, parity_checks as P
from optimization_dynamics_amd import _lib
lib=_lib.default_library()
x1,U=W.hopper_rollout_inputs(256,20,seed=21,u_sigma=0.3)
for split in (0,1):
    for ppw in (0,4,16):
        im=P.make_im('hopper',lib,'cuda:0'); im.set_launch_config(ppw,split)
        X,A,Bm,st,it,_=im.rollout(torch.tensor(x1),torch.tensor(U)); torch.cuda.synchronize()
        t0=time.time(); X,A,Bm,st,it,_=im.rollout(torch.tensor(x1),torch.tensor(U)); torch.cuda.synchronize(); dt=time.time()-t0
        print('split',split,'ppw',ppw,'ms %.2f'%(dt*1e3),'status',torch.bincount(st.flatten(),minlength=8).tolist(),'iters',it.double().mean().item(),it.max().item(),'X',X[:,-1,:3].abs().sum().item(),'A nan',torch.isnan(A).any().item())
