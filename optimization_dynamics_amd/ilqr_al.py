"""Augmented-Lagrangian iLQR with general costs and stage constraints -- the interface the reference's
examples drive (IterativeLQR.jl, un-vendored; call sites examples/acrobot.jl:33-113, examples/hopper.jl:165-292):

    iLQR.Dynamics(f, fx, fu, ny, nx, nu)       -> ImplicitStage / any object with step / step_grad
    iLQR.Cost(fn, nx, nu)                      -> Cost(fn)
    iLQR.Constraint(fn, nx, nu, idx_ineq=[..]) -> Constraint(fn, idx_ineq)
    iLQR.solver(model, obj, cons, opts)        -> Solver(stages, costs, cons, n, m, **opts)
    initialize_controls! / solve! / get_trajectory

P independent problems run in lockstep (batch last).  Costs and constraints are written per sample with
torch operations, like the Julia closures, and differentiated with torch.func (the reference differentiates
them with Symbolics).  The heavy steps are the library's: `od_step(_grad)` for the dynamics of every
candidate knot and `od_ilqr_backward` for the Riccati recursion.  State and control dimensions are uniform
over the horizon; problems whose first stage differs (examples/hopper.jl: nx 8 -> 16, nu 10 -> 2) pad
(examples/hopper_gait.py).  Internals of IterativeLQR are recalled, not pinned:
constraints c <= 0 (idx_ineq) / c = 0, multiplier update lam <- P(lam + rho c), rho <- rho * scale,
active set (c >= 0 or lam > 0), Gauss-Newton Hessians, Armijo backtracking on the AL merit.
"""
import torch
from torch.func import grad, hessian, jacrev, vmap

from .dynamics import ImplicitDynamics, _ptr


class Cost:
    """fn(x, u) -> scalar for one sample (x: (n,), u: (m,)); `terminal` costs ignore u."""

    def __init__(self, fn):
        self.fn = fn

    def value(self, x, u):                      # x (n, P), u (m, P) -> (P,)
        return vmap(self.fn, in_dims=(1, 1))(x, u)

    def expansion(self, x, u):
        g = vmap(grad(self.fn, argnums=(0, 1)), in_dims=(1, 1))(x, u)
        H = vmap(hessian(self.fn, argnums=(0, 1)), in_dims=(1, 1))(x, u)
        return g[0], g[1], H[0][0], H[1][1], H[1][0]      # lx (P,n) lu (P,m) lxx (P,n,n) luu (P,m,m) lux (P,m,n)


class Constraint:
    """fn(x, u) -> (nc,) for one sample; rows in idx_ineq mean c <= 0, the others c = 0."""

    def __init__(self, fn=None, idx_ineq=()):
        self.fn = fn
        self.idx_ineq = list(idx_ineq)

    def value(self, x, u):                      # -> (P, nc)
        return vmap(self.fn, in_dims=(1, 1))(x, u)

    def jacobian(self, x, u):                   # -> cx (P, nc, n), cu (P, nc, m)
        return vmap(jacrev(self.fn, argnums=(0, 1)), in_dims=(1, 1))(x, u)


class ImplicitStage:
    """one stage of `iLQR.Dynamics(f, fx, fu, ...)` over an ImplicitDynamics: x+ = [q2; q3(x, u)]"""

    def __init__(self, im: ImplicitDynamics):
        self.im = im

    def step(self, x, u):
        return self.im.step(x.contiguous(), u.contiguous())[0]

    def step_grad(self, x, u):
        D, DX, DU, st, it = self.im.step_grad(x.contiguous(), u.contiguous())
        return D, DX, DU


class Solver:
    def __init__(self, stages, costs, cons, n, m, *, im, alpha_min=1.0e-5, obj_tol=1.0e-3, grad_tol=1.0e-3, max_iter=10,
                 max_al_iter=15, con_tol=1.0e-3, rho_init=1.0, rho_scale=10.0, rho_max=1.0e8, reg=1.0e-6, c1=1.0e-4, verbose=False):
        """stages: T-1 dynamics objects (step / step_grad on (n, P), (m, P)); costs: T Cost (last terminal);
        cons: T Constraint (fn None = unconstrained); im: any ImplicitDynamics handle (device, stream and
        the Riccati kernel are reached through it)."""
        assert len(costs) == len(stages) + 1 and len(cons) == len(costs)
        self.stages, self.costs, self.cons = stages, costs, cons
        self.T, self.n, self.m, self.im = len(stages), n, m, im
        self.dev = im.device
        self.o = dict(alpha_min=alpha_min, obj_tol=obj_tol, grad_tol=grad_tol, max_iter=max_iter, max_al_iter=max_al_iter,
                      con_tol=con_tol, rho_init=rho_init, rho_scale=rho_scale, rho_max=rho_max, reg=reg, c1=c1, verbose=verbose)
        na = 1
        while 2.0 ** -(na - 1) > alpha_min and na < 18:
            na += 1
        self.alphas = torch.tensor([2.0 ** -i for i in range(na)], dtype=torch.float64, device=self.dev)
        self.iters = 0

    @staticmethod
    def _groups(objs):
        """knots sharing one object (the examples reuse one stage cost / constraint / dynamics for most of
        the horizon) are evaluated in one batched call"""
        g = {}
        for t, ob in enumerate(objs):
            g.setdefault(id(ob), (ob, []))[1].append(t)
        return list(g.values())

    def _xu(self, X, U, ts):
        """knots ts flattened into the batch: x (n, len(ts)*P), u (m, len(ts)*P); the terminal knot has u = 0"""
        P = X.shape[-1]
        Up = torch.cat([U, self._uT(P)[:, None, :]], 1) if max(ts) >= self.T else U
        return X[:, ts].reshape(self.n, -1), Up[:, ts].reshape(self.m, -1)

    # ---- problem data ------------------------------------------------------------------------------
    def initialize_controls(self, U):           # (m, T, P)
        self.U = U.to(self.dev, torch.float64).clone()

    def rollout(self, x1, U):
        X = [x1]
        for t in range(self.T):
            X.append(self.stages[t].step(X[-1], U[:, t]))
        return torch.stack(X, 1)

    def _uT(self, P):
        return torch.zeros(self.m, P, dtype=torch.float64, device=self.dev)

    def _knots(self, X, U):
        """(x_t, u_t) for t = 0..T with a zero control at the terminal knot"""
        P = X.shape[-1]
        return [(X[:, t], U[:, t] if t < self.T else self._uT(P)) for t in range(self.T + 1)]

    # ---- merit -------------------------------------------------------------------------------------
    def objective(self, X, U):
        P = X.shape[-1]
        J = 0.0
        for cost, ts in self._groups(self.costs):
            x, u = self._xu(X, U, ts)
            J = J + cost.value(x, u).view(len(ts), P).sum(0)
        return J

    def constraints(self, X, U):
        P = X.shape[-1]
        out = [None] * (self.T + 1)
        for con, ts in self._groups(self.cons):
            if con.fn is None:
                continue
            x, u = self._xu(X, U, ts)
            c = con.value(x, u).view(len(ts), P, -1)
            for i, t in enumerate(ts):
                out[t] = c[i]
        return out

    def _active(self, t, c, lam):
        a = torch.ones_like(c)
        idx = self.cons[t].idx_ineq
        if idx:
            a[:, idx] = ((c[:, idx] >= 0) | (lam[:, idx] > 0)).to(c.dtype)
        return a

    def merit(self, X, U, lam, rho):
        J = self.objective(X, U)
        for t, c in enumerate(self.constraints(X, U)):
            if c is not None:
                a = self._active(t, c, lam[t])
                J = J + (lam[t] * c).sum(1) + 0.5 * rho * (a * c * c).sum(1)
        return J

    def violation(self, X, U):
        v = torch.zeros(X.shape[-1], dtype=torch.float64, device=self.dev)
        for t, c in enumerate(self.constraints(X, U)):
            if c is not None:
                cc = c.clone()
                idx = self.cons[t].idx_ineq
                if idx:
                    cc[:, idx] = cc[:, idx].clamp_min(0.0)
                v = torch.maximum(v, cc.abs().max(1).values)
        return v

    # ---- one iLQR iteration --------------------------------------------------------------------------
    def _linearize(self, X, U):
        n, m, T, P = self.n, self.m, self.T, X.shape[-1]
        A = torch.empty(n, n, T, P, dtype=torch.float64, device=self.dev)
        B = torch.empty(n, m, T, P, dtype=torch.float64, device=self.dev)
        for stage, ts in self._groups(self.stages):
            x, u = self._xu(X, U, ts)
            _, dx, du = stage.step_grad(x, u)                # all knots of the group in one launch
            A[:, :, ts] = dx.reshape(n, n, len(ts), P)
            B[:, :, ts] = du.reshape(n, m, len(ts), P)
        return A, B

    def _expansion(self, X, U, lam, rho):
        n, m, T = self.n, self.m, self.T
        P = X.shape[-1]
        lx = torch.zeros(n, T, P, dtype=torch.float64, device=self.dev); lu = torch.zeros(m, T, P, dtype=torch.float64, device=self.dev)
        lxx = torch.zeros(n * n, T, P, dtype=torch.float64, device=self.dev); luu = torch.zeros(m * m, T, P, dtype=torch.float64, device=self.dev)
        lux = torch.zeros(m * n, T, P, dtype=torch.float64, device=self.dev)
        Vx = Vxx = None
        exp = [None] * (T + 1)
        for cost, ts in self._groups(self.costs):
            x, u = self._xu(X, U, ts)
            parts = [q.reshape(len(ts), P, *q.shape[1:]) for q in cost.expansion(x, u)]
            for i, t in enumerate(ts):
                exp[t] = [q[i] for q in parts]
        for con, ts in self._groups(self.cons):
            if con.fn is None:
                continue
            x, u = self._xu(X, U, ts)
            c_all = con.value(x, u).view(len(ts), P, -1)
            cx_all, cu_all = [q.reshape(len(ts), P, *q.shape[1:]) for q in con.jacobian(x, u)]
            for i, t in enumerate(ts):
                c, cx, cu = c_all[i], cx_all[i], cu_all[i]
                gx, gu, hxx, huu, hux = exp[t]
                a = self._active(t, c, lam[t])
                w = lam[t] + rho * a * c                          # (P, nc)
                exp[t] = [gx + torch.einsum("pcn,pc->pn", cx, w), gu + torch.einsum("pcm,pc->pm", cu, w),
                          hxx + rho * torch.einsum("pcn,pc,pck->pnk", cx, a, cx),
                          huu + rho * torch.einsum("pcm,pc,pck->pmk", cu, a, cu),
                          hux + rho * torch.einsum("pcm,pc,pcn->pmn", cu, a, cx)]
        for t in range(T + 1):
            gx, gu, hxx, huu, hux = exp[t]
            if t < T:
                lx[:, t] = gx.T; lu[:, t] = gu.T
                lxx[:, t] = hxx.permute(2, 1, 0).reshape(n * n, P)     # column-major (i + n j)
                luu[:, t] = huu.permute(2, 1, 0).reshape(m * m, P)
                lux[:, t] = hux.permute(2, 1, 0).reshape(m * n, P)
            else:
                Vx, Vxx = gx.T.contiguous(), hxx.permute(2, 1, 0).reshape(n * n, P).contiguous()
        return lxx, luu, lux, lx, lu, Vxx, Vx

    def _backward(self, A, Bm, quad, reg):
        lxx, luu, lux, lx, lu, VxxT, VxT = quad
        n, m, T = self.n, self.m, self.T
        P = A.shape[-1]
        im = self.im
        im._use_current_stream()
        Af = A.transpose(0, 1).reshape(n * n, T, P).contiguous()
        Bf = Bm.transpose(0, 1).reshape(n * m, T, P).contiguous()
        K = torch.empty(m * n, T, P, dtype=torch.float64, device=self.dev)
        k = torch.empty(m, T, P, dtype=torch.float64, device=self.dev)
        dV = torch.empty(2, P, dtype=torch.float64, device=self.dev)
        st = torch.empty(P, dtype=torch.int32, device=self.dev)
        im.lib.check(im.lib.cdll.od_ilqr_backward(im._h, P, T, n, m, _ptr(Af), _ptr(Bf), _ptr(lxx.contiguous()), _ptr(luu.contiguous()),
                                                  _ptr(lux.contiguous()), _ptr(lx.contiguous()), _ptr(lu.contiguous()), _ptr(VxxT), _ptr(VxT),
                                                  float(reg), _ptr(K), _ptr(k), _ptr(dV), _ptr(st)))
        return K.view(n, m, T, P).transpose(0, 1), k, dV, st      # K: (m, n, T, P)

    def _forward(self, x1, X, U, K, k):
        """closed-loop rollouts for all step sizes at once: candidate a*P + p"""
        na, P = self.alphas.numel(), x1.shape[-1]
        al = self.alphas.repeat_interleave(P)[None, :]
        rep = lambda v: v.repeat(*([1] * (v.dim() - 1)), na)
        x = rep(x1)
        Xc, Uc = [x], []
        for t in range(self.T):
            dx = x - rep(X[:, t])
            u = rep(U[:, t]) + al * rep(k[:, t]) + torch.einsum("mnp,np->mp", rep(K[:, :, t]), dx)
            x = self.stages[t].step(x, u)
            Xc.append(x); Uc.append(u)
        return torch.stack(Xc, 1), torch.stack(Uc, 1)

    # ---- solve! ------------------------------------------------------------------------------------
    def solve(self, x1, U0=None):
        o = self.o
        if U0 is not None:
            self.initialize_controls(U0)
        x1 = x1.to(self.dev, torch.float64)
        U = self.U
        P, na = x1.shape[-1], self.alphas.numel()
        X = self.rollout(x1, U)
        cs = self.constraints(X, U)
        lam = [None if c is None else torch.zeros_like(c) for c in cs]
        rho = o["rho_init"]
        self.iters = 0
        for al_it in range(o["max_al_iter"]):
            J = self.merit(X, U, lam, rho)
            reg = o["reg"]
            for it in range(o["max_iter"]):
                A, Bm = self._linearize(X, U)
                K, k, dV, bst = self._backward(A, Bm, self._expansion(X, U, lam, rho), reg)
                if (bst != 1).any():
                    reg = min(reg * 10.0, 1e8)
                    if reg >= 1e8:
                        break
                    continue
                Xc, Uc = self._forward(x1, X, U, K, k)
                lam_c = [None if l is None else l.repeat(na, 1) for l in lam]
                Jc = self.merit(Xc, Uc, lam_c, rho).view(na, P)
                expected = self.alphas[:, None] * dV[0][None, :] + self.alphas[:, None] ** 2 * dV[1][None, :]
                accept = torch.isfinite(Jc) & (Jc <= J[None, :] + o["c1"] * expected)
                took = accept.any(0)
                first = torch.where(took, accept.float().argmax(0), torch.zeros(P, dtype=torch.long, device=self.dev))
                sel = first * P + torch.arange(P, device=self.dev)
                U = torch.where(took[None, None, :], Uc[:, :, sel], U)
                X = torch.where(took[None, None, :], Xc[:, :, sel], X)
                Jn = torch.where(took, Jc.reshape(-1)[sel], J)
                dJ = J - Jn
                J = Jn
                self.iters += 1
                gnorm = k.abs().amax((0, 1))
                if o["verbose"]:
                    print("al %2d it %3d  merit %.6g  accepted %d/%d  dJ %.3g  |k| %.3g  viol %.3g" %
                          (al_it, it, J.mean().item(), int(took.sum()), P, dJ.max().item(), gnorm.max().item(), self.violation(X, U).max().item()))
                if not took.any():
                    reg = min(reg * 10.0, 1e8)
                    if reg >= 1e8:
                        break
                    continue
                reg = max(reg / 5.0, o["reg"])
                if dJ.max().item() < o["obj_tol"] or gnorm.max().item() < o["grad_tol"]:
                    break
            viol = self.violation(X, U)
            if viol.max().item() <= o["con_tol"]:
                break
            for t, c in enumerate(self.constraints(X, U)):
                if c is not None:
                    lam[t] = lam[t] + rho * c
                    idx = self.cons[t].idx_ineq
                    if idx:
                        lam[t][:, idx] = lam[t][:, idx].clamp_min(0.0)
            rho = min(rho * o["rho_scale"], o["rho_max"])
        self.X, self.U, self.lam, self.rho = X, U, lam, rho
        return X, U

    def get_trajectory(self):
        return self.X, self.U
