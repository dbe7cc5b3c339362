"""numpy restatement of the Riccati backward pass used by optimization_dynamics_amd.ilqr -- TEST
INFRASTRUCTURE ONLY (checker for od_ilqr_backward).  Gauss-Newton iLQR as in IterativeLQR.jl's backward
pass as recalled (un-vendored, unpinned)."""
import numpy as np


def backward(A, Bm, lxx, luu, lux, lx, lu, Vxx, Vx, reg):
    """one trajectory.  A: (T,n,n) B: (T,n,m) lxx: (T,n,n) luu: (T,m,m) lux: (T,m,n) lx: (T,n) lu: (T,m)"""
    T, n, m = Bm.shape
    K = np.zeros((T, m, n)); k = np.zeros((T, m)); dV = np.zeros(2)
    Vxx = Vxx.copy(); Vx = Vx.copy()
    for t in range(T - 1, -1, -1):
        Qx = lx[t] + A[t].T @ Vx
        Qu = lu[t] + Bm[t].T @ Vx
        Qxx = lxx[t] + A[t].T @ Vxx @ A[t]
        Quu = luu[t] + Bm[t].T @ Vxx @ Bm[t]
        Qux = lux[t] + Bm[t].T @ Vxx @ A[t]
        Qr = Quu + reg * np.eye(m)
        K[t] = -np.linalg.solve(Qr, Qux)
        k[t] = -np.linalg.solve(Qr, Qu)
        dV += [k[t] @ Qu, 0.5 * k[t] @ Quu @ k[t]]
        Vx = Qx + K[t].T @ Quu @ k[t] + K[t].T @ Qu + Qux.T @ k[t]
        Vxx = Qxx + K[t].T @ Quu @ K[t] + K[t].T @ Qux + Qux.T @ K[t]
        Vxx = 0.5 * (Vxx + Vxx.T)
    return K, k, dV


def backward_batch(A, Bm, lxx, luu, lux, lx, lu, Vxx, Vx, reg):
    """the same recursion for P trajectories at once (vectorised over the leading axis; numpy's batched solve).
    A: (P,T,n,n) B: (P,T,n,m) lxx: (P,T,n,n) luu: (P,T,m,m) lux: (P,T,m,n) lx: (P,T,n) lu: (P,T,m) Vxx: (P,n,n) Vx: (P,n);
    reg: scalar or (P,).  -> K (P,T,m,n), k (P,T,m), dV (P,2)"""
    P, T, n, m = Bm.shape
    K = np.zeros((P, T, m, n)); k = np.zeros((P, T, m)); dV = np.zeros((P, 2))
    Vxx = Vxx.copy(); Vx = Vx.copy()
    reg = np.broadcast_to(np.asarray(reg, dtype=np.float64), (P,))
    I = np.eye(m)[None] * reg[:, None, None]
    tr = lambda M: np.swapaxes(M, -1, -2)
    for t in range(T - 1, -1, -1):
        At, Bt = A[:, t], Bm[:, t]
        Qx = lx[:, t] + np.einsum("pji,pj->pi", At, Vx)
        Qu = lu[:, t] + np.einsum("pji,pj->pi", Bt, Vx)
        VA, VB = Vxx @ At, Vxx @ Bt
        Qxx = lxx[:, t] + tr(At) @ VA
        Quu = luu[:, t] + tr(Bt) @ VB
        Qux = lux[:, t] + tr(Bt) @ VA
        Qr = Quu + I
        K[:, t] = -np.linalg.solve(Qr, Qux)
        k[:, t] = -np.linalg.solve(Qr, Qu[:, :, None])[:, :, 0]
        dV[:, 0] += np.einsum("pi,pi->p", k[:, t], Qu)
        dV[:, 1] += 0.5 * np.einsum("pi,pij,pj->p", k[:, t], Quu, k[:, t])
        Kt = K[:, t]
        Vx = Qx + np.einsum("pji,pjk,pk->pi", Kt, Quu, k[:, t]) + np.einsum("pji,pj->pi", Kt, Qu) + np.einsum("pji,pj->pi", Qux, k[:, t])
        Vxx = Qxx + tr(Kt) @ Quu @ Kt + tr(Kt) @ Qux + tr(Qux) @ Kt
        Vxx = 0.5 * (Vxx + tr(Vxx))
    return K, k, dV
