"""The jammed knots of the headline workload (hopper, 4096 x 100, seed 0: knots that run into max_iter) on the host build: does the iterate
sequence end in an exact cycle (period p: z_k == z_{k-p} bit for bit -- the rest of the loop could then be skipped exactly), and how far is the
iterate after 10 / 20 / 30 / 50 / 80 iterations from the one after 100?  (`python tools/jam_cycles.py > profiles/r4_jam_cycles.txt`)"""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, torch
import parity_checks as P
import bench
from optimization_dynamics_amd import _lib
lib=_lib.Library(os.path.join(ROOT,'tests','host_emu','libod_emu.so'))
x1,U=bench.make_inputs(4096,100,seed=0)
im=P.make_im('hopper',lib,'cpu')
X,G,st,it,_=im.rollout_compact(torch.tensor(x1),torch.tensor(U))
it=it.numpy(); 
jam=np.argwhere(it[0]>=100)
print('jams',len(jam))
Xn=X.numpy()
for (t,b) in jam:
    x=Xn[:,t,b:b+1]; u=U[:,t,b:b+1]
    hs=[]
    zs=[]
    for k in range(1,101):
        im2=P.make_im('hopper',lib,'cpu'); im2.set_options(max_iter=k)
        Z,DZ,s,i=im2.step_full(torch.tensor(x),torch.tensor(u))
        z=Z.numpy()[:,0]; zs.append(z.copy()); hs.append(hash(z.tobytes()))
    # find smallest period at the end
    per=None
    for p in range(1,9):
        if all(hs[-1-j]==hs[-1-j-p] for j in range(p)):
            per=p; break
    first=None
    if per:
        for k in range(100-per-1,-1,-1):
            if hs[k]!=hs[k+per]: first=k+1; break
    d=[np.abs(zs[k]-zs[-1]).max() for k in (9,19,29,49,79)]
    print('knot',t,b,'period',per,'cycle from iteration',first,'|z_k - z_100| at k=10,20,30,50,80:',['%.1e'%v for v in d])
