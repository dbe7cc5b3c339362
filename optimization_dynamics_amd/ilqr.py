"""Batched iLQR around the implicit dynamics -- the consumer of the path.

The reference hands its `f / fx / fu` callbacks to IterativeLQR.jl (un-vendored; interface visible at
examples/acrobot.jl:33-36,97-113): an augmented-Lagrangian outer loop around iLQR iterations made of
(1) a derivative sweep, (2) a Riccati backward pass, (3) a forward pass with Armijo backtracking.
Here B independent trajectory optimisations run at once:

    (1) `od_rollout`          nominal rollout + linearisation (x+, A, B) of every knot
    (2) `od_ilqr_backward`    Riccati recursion, one lane per trajectory
    (3) `od_rollout_policy`   ALL Armijo step sizes of all trajectories rolled out speculatively in one launch

Costs are quadratic (what the reference's examples use for tracking / effort terms), terminal equality
constraints x_T[idx] = goal are handled by the augmented Lagrangian like the reference's `terminal_con`.
Internals of IterativeLQR are recalled, not pinned (same status as the solver).
"""
import numpy as np
import torch

from .dynamics import ImplicitDynamics, _ptr


class QuadraticObjective:
    """sum_t 1/2 (x_t - x_ref)'Q(x_t - x_ref) + 1/2 u_t'R u_t  +  1/2 (x_T - x_ref)'QT(x_T - x_ref)
    (cf. objt / objT of examples/hopper.jl:207-220), plus an AL term for  x_T[idx] = goal."""

    def __init__(self, Q, R, QT, x_ref, goal_idx=None, goal=None, device="cuda"):
        dev = torch.device(device)
        as_t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)
        self.Q, self.R, self.QT, self.x_ref = as_t(Q), as_t(R), as_t(QT), as_t(x_ref)
        self.n, self.m = self.Q.shape[0], self.R.shape[0]
        self.goal_idx = None if goal_idx is None else torch.as_tensor(np.asarray(goal_idx), device=dev, dtype=torch.long)
        self.goal = None if goal is None else as_t(goal)
        self.device = dev
        self._host = {id(M): np.asarray(h, dtype=np.float64) for M, h in ((self.Q, Q), (self.R, R), (self.QT, QT))}   # (no device read-back in `expansion`)
        self.stage = self.terminal = None

    def set_constraints(self, stage=None, terminal=None):
        """affine constraints by augmented Lagrangian (iLQR.Constraint with idx_ineq, examples/rocket.jl:82-110):
        stage = (C (ns, n), D (ns, m), d (ns,), n_ineq):  C x_t + D u_t - d, t < T;  terminal = (C (nt, n), d (nt,), n_ineq):  C x_T - d;
        the first n_ineq rows of each are inequalities (<= 0), the rest equalities (od_ilqr_set_constraints)"""
        f = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
        self.stage = None if stage is None else (f(stage[0]).reshape(-1, self.n), f(stage[1]).reshape(-1, self.m), f(stage[2]).reshape(-1), int(stage[3]))
        self.terminal = None if terminal is None else (f(terminal[0]).reshape(-1, self.n), f(terminal[1]).reshape(-1), int(terminal[2]))
        return self

    def set_parameter_stage(self, w_theta, constraint=None, p=None, terminal=None, cost_const=0.0):
        """od_ilqr_set_parameter_stage (examples/hopper.jl: the initial configurations theta = [q1; q2] optimised with the controls):
        slot 0 of every trajectory is theta, costing 1/2 theta' diag(w_theta) theta + cost_const; `constraint`: name of a generated
        constraint function c(theta; p) = 0 (codegen --add-constraint); terminal = (Ct_x (nt, n), Ct_theta (nt, n), dt (nt,), n_ineq):
        rows Ct_x x_T + Ct_theta theta - dt.  Device-resident solver only (ILQR.solve)."""
        f = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
        self.parameter_stage = dict(w_theta=f(w_theta), constraint=constraint, p=None if p is None else f(p), cost_const=float(cost_const),
                                    terminal=None if terminal is None else (f(terminal[0]).reshape(-1, self.n), f(terminal[1]).reshape(-1, self.n), f(terminal[2]).reshape(-1), int(terminal[3])))
        return self

    @property
    def constrained(self):
        return self.goal_idx is not None or self.stage is not None or self.terminal is not None or getattr(self, "parameter_stage", None) is not None

    # the constraint rows, evaluated in the order of the device kernels (od_ilqr_solver.inc: sum over x then over u, then - d)
    def stage_c(self, X, U):
        """(ns, T, P)"""
        C, D, d, _ = self.stage
        out = []
        for r in range(C.shape[0]):
            c = torch.zeros_like(X[0, :-1])
            for j in range(self.n):
                c = c + float(C[r, j]) * X[j, :-1]
            for j in range(self.m):
                c = c + float(D[r, j]) * U[j]
            out.append(c - float(d[r]))
        return torch.stack(out)

    def terminal_c(self, X):
        """(nt, P)"""
        C, d, _ = self.terminal
        out = []
        for r in range(C.shape[0]):
            c = torch.zeros_like(X[0, -1])
            for j in range(self.n):
                c = c + float(C[r, j]) * X[j, -1]
            out.append(c - float(d[r]))
        return torch.stack(out)

    @staticmethod
    def _active(c, lam, n_in):
        a = torch.ones_like(c, dtype=torch.bool)
        a[:n_in] = (c[:n_in] >= 0) | (lam[:n_in] > 0)
        return a

    def constraint_terms(self, X, U, mult, rho):
        """the merit's constraint terms per trajectory; mult = dict(lam=goal multipliers, lam_s=(ns, T, P), lam_t=(nt, P))"""
        X, U = X.double(), U.double()
        J = torch.zeros(U.shape[2], dtype=torch.float64, device=X.device)
        if self.stage is not None:
            c, l = self.stage_c(X, U), mult["lam_s"]
            a = self._active(c, l, self.stage[3])
            for t in range(c.shape[1]):                       # (the kernel's order: knot after knot, row after row)
                for r in range(c.shape[0]):
                    J = J + (l[r, t] * c[r, t] + torch.where(a[r, t], 0.5 * rho * c[r, t] * c[r, t], torch.zeros_like(J)))
        if self.terminal is not None:
            c, l = self.terminal_c(X), mult["lam_t"]
            a = self._active(c, l, self.terminal[2])
            for r in range(c.shape[0]):
                J = J + (l[r] * c[r] + torch.where(a[r], 0.5 * rho * c[r] * c[r], torch.zeros_like(J)))
        if self.goal_idx is not None:
            c = self.constraint(X)
            J = J + ((mult["lam"] * c).sum(0) + 0.5 * rho * (c * c).sum(0))
        return J

    def violation(self, X, U):
        """max over all constraint rows and knots of |c_eq|, max(c_ineq, 0): (P,)"""
        v = torch.zeros(U.shape[2], dtype=torch.float64, device=X.device)
        if self.goal_idx is not None:
            v = torch.maximum(v, self.constraint(X).abs().max(0).values)
        if self.terminal is not None:
            c = self.terminal_c(X); k = self.terminal[2]
            c = torch.cat([c[:k].clamp_min(0.0), c[k:].abs()])
            v = torch.maximum(v, c.max(0).values)
        if self.stage is not None:
            c = self.stage_c(X, U); k = self.stage[3]
            c = torch.cat([c[:k].clamp_min(0.0), c[k:].abs()])
            v = torch.maximum(v, c.reshape(-1, c.shape[-1]).max(0).values)
        return v

    def constraint(self, X):
        """c = x_T[idx] - goal, shape (nc, P)"""
        return X[self.goal_idx, -1, :] - self.goal[:, None]

    def _column_major(self):
        if getattr(self, "_cm", None) is None:
            self._cm = tuple(M.T.contiguous() for M in (self.Q, self.R, self.QT))       # column-major flattenings
        return self._cm

    def _value_kernel(self, im, X, U):
        """the library's one-pass cost kernel (od_quad_cost) through the dynamics object `im`'s handle"""
        from ._lib import OD_F32, OD_F64
        X, U = X.contiguous(), U.contiguous()
        T, P = U.shape[1], U.shape[2]
        J = torch.empty(P, dtype=torch.float64, device=X.device)
        cm = self._column_major()
        im._use_current_stream()
        im.lib.check(im.lib.cdll.od_quad_cost(im._h, P, T, self.n, self.m, OD_F32 if X.dtype == torch.float32 else OD_F64, _ptr(X), _ptr(U),
                                              _ptr(cm[0]), _ptr(cm[1]), _ptr(cm[2]), _ptr(self.x_ref), _ptr(J)))
        return J

    def value(self, X, U, lam=None, rho=0.0, im=None):
        """X: (n, T+1, P), U: (m, T, P) -> cost per trajectory (P,).  im: a dynamics object whose handle runs the cost kernel
        (batch-minor layout, live handle); without it, or where the kernel does not apply, the same formula in torch"""
        T, P = U.shape[1], U.shape[2]
        if im is not None and getattr(im, "_h", None) and getattr(im, "layout", 0) == 0 and X.dtype == U.dtype \
                and X.dtype in (torch.float32, torch.float64) and X.device == self.x_ref.device and self.n <= 16 and self.m <= 12:
            J = self._value_kernel(im, X, U)
        else:
            dx = X.double() - self.x_ref[:, None, None]

            def quad(M, v):                     # sum over knots of 1/2 v'Mv per trajectory: one skinny GEMM, not an einsum
                vf = v.reshape(v.shape[0], -1)
                return 0.5 * (vf * (M @ vf)).sum(0).view(-1, P).sum(0)

            J = quad(self.Q, dx[:, :-1]) + quad(self.R, U.double()) + quad(self.QT, dx[:, -1])
        if isinstance(lam, dict):                               # all constraint kinds (goal rows, stage and terminal rows)
            return J + self.constraint_terms(X, U, lam, rho)
        if self.goal_idx is not None and lam is not None:
            c = self.constraint(X)
            J = J + (lam * c).sum(0) + 0.5 * rho * (c * c).sum(0)
        return J

    def expansion(self, X, U, lam=None, rho=0.0):
        """quadratic model per knot in the library's layout (column-major flattening, batch last)"""
        n, m = self.n, self.m
        T, P = U.shape[1], U.shape[2]
        dx = X - self.x_ref[:, None, None]

        def matvec(M, v):       # sum_j M[i, j] v[j] accumulated in the order j = 0, 1, ... (the order of k_il_expand, od_ilqr_solver.inc)
            out = torch.zeros((M.shape[0],) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
            Mh = self._host[id(M)]
            for i in range(M.shape[0]):
                for j in range(M.shape[1]):
                    out[i] = out[i] + float(Mh[i, j]) * v[j]
            return out

        lx = matvec(self.Q, dx[:, :-1]).contiguous()
        lu = matvec(self.R, U).contiguous()
        lxx = self.Q.T.reshape(n * n, 1, 1).expand(n * n, T, P).contiguous()
        luu = self.R.T.reshape(m * m, 1, 1).expand(m * m, T, P).contiguous()
        lux = torch.zeros(m * n, T, P, dtype=torch.float64, device=X.device)
        Vxx = self.QT.clone()[:, :, None].repeat(1, 1, P)
        Vx = matvec(self.QT, dx[:, -1])
        mult = lam if isinstance(lam, dict) else dict(lam=lam)
        if self.stage is not None:
            # (the order of il_expand_slot: w = lam + rho A c; gradients += C'w, D'w row after row; Hessians objective + rho sum (row)(row)')
            C, D, d, k_in = self.stage
            c, l = self.stage_c(X, U), mult["lam_s"]
            ra = torch.where(self._active(c, l, k_in), rho * torch.ones_like(c), torch.zeros_like(c))
            w = l + ra * c
            ns = C.shape[0]
            for j in range(n):
                for r in range(ns):
                    lx[j] = lx[j] + float(C[r, j]) * w[r]
            for j in range(m):
                for r in range(ns):
                    lu[j] = lu[j] + float(D[r, j]) * w[r]
            Qh, Rh = self._host[id(self.Q)], self._host[id(self.R)]
            lxx = torch.empty(n * n, T, P, dtype=torch.float64, device=X.device)
            luu = torch.empty(m * m, T, P, dtype=torch.float64, device=X.device)
            for j in range(n):
                for i in range(n):
                    t_ = torch.full_like(c[0], float(Qh[i, j]))
                    for r in range(ns):
                        t_ = t_ + ra[r] * float(C[r, i]) * float(C[r, j])
                    lxx[i + n * j] = t_
            for j in range(m):
                for i in range(m):
                    t_ = torch.full_like(c[0], float(Rh[i, j]))
                    for r in range(ns):
                        t_ = t_ + ra[r] * float(D[r, i]) * float(D[r, j])
                    luu[i + m * j] = t_
            for j in range(n):
                for i in range(m):
                    t_ = torch.zeros_like(c[0])
                    for r in range(ns):
                        t_ = t_ + ra[r] * float(D[r, i]) * float(C[r, j])
                    lux[i + m * j] = t_
        if self.goal_idx is not None and mult.get("lam") is not None:
            c = self.constraint(X)
            Vx[self.goal_idx] += mult["lam"] + rho * c
            Vxx[self.goal_idx, self.goal_idx] += rho
        if self.terminal is not None:
            C, d, k_in = self.terminal
            c, l = self.terminal_c(X), mult["lam_t"]
            ra = torch.where(self._active(c, l, k_in), rho * torch.ones_like(c), torch.zeros_like(c))
            w = l + ra * c
            for r in range(C.shape[0]):
                for j in range(n):
                    Vx[j] = Vx[j] + float(C[r, j]) * w[r]
                    for i in range(n):
                        Vxx[i, j] = Vxx[i, j] + ra[r] * float(C[r, i]) * float(C[r, j])
        VxxT = Vxx.transpose(0, 1).reshape(n * n, P).contiguous()     # column-major flattening
        return lxx, luu, lux, lx.contiguous(), lu.contiguous(), VxxT, Vx.contiguous()


class ILQR:
    """B trajectory optimisations in lockstep."""

    def __init__(self, im, objective: QuadraticObjective, T,
                 alphas=tuple(2.0 ** -i for i in range(11)), reg=1e-6, c1=1e-4, bundle=None):
        """im: an ImplicitDynamics (mechanical models) or a rocket.RocketDynamics; bundle: a GradientBundle -- the linearisation is
        then fx_gb / fu_gb (src/gradient_bundle.jl:109-147, examples/planar_push.jl with GB = true) instead of the implicit gradients"""
        self.im, self.obj, self.T = im, objective, T
        self.bundle = bundle
        if isinstance(im, ImplicitDynamics):
            self.n, self.m = 2 * im.model.nq, im.model.nu
        else:
            self.n, self.m = im.n, im.m
        self.alphas = torch.tensor(alphas, dtype=torch.float64, device=im.device)
        self.reg, self.c1 = reg, c1
        self._dev = None

    # -- the three device steps ----------------------------------------------------------------
    def linearize(self, x1, U):
        if self.bundle is not None:
            X, st = self.im.rollout(x1, U, grads=False)[0], None
            A, Bm = self.linearize_at(X, U)
            return X, A, Bm, st
        X, A, Bm, st, it, _ = self.im.rollout(x1, U)
        return X, A, Bm, st

    def linearize_at(self, X, U):
        """fx / fu on the knots of a trajectory whose states are known already (the accepted candidates of the forward pass):
        T*B independent knots in one launch instead of a second time recursion.  -> A (n, n, T, B), Bm (n, m, T, B)"""
        n, m, T = self.n, self.m, self.T
        B = U.shape[-1]
        Xk, Uk = X[:, :-1].reshape(n, T * B), U.reshape(m, T * B)
        if self.bundle is not None:
            from .gradient_bundle import gradient_batch
            nq = n // 2
            dz, st = gradient_batch(self.im, self.bundle, Xk.contiguous(), Uk.contiguous())        # (nq, 2nq+nu, T*B)
            DX = torch.zeros(n, n, T * B, dtype=torch.float64, device=Xk.device)
            DU = torch.zeros(n, m, T * B, dtype=torch.float64, device=Xk.device)
            for i in range(nq):
                DX[i, nq + i] = 1.0
            DX[nq:] = dz[:, :n]
            DU[nq:] = dz[:, n:]
            self.last_linearisation_ok = (st != 0).view(T, B)
        elif isinstance(self.im, ImplicitDynamics):
            _, DX, DU, st, _ = self.im.step_grad(Xk, Uk)
            self.last_linearisation_ok = ((st & 3) == 3).view(T, B)
        else:
            DX, DU, st = self.im.linearize_knots(Xk, Uk, with_status=True)
            need = 0x33 if self.im.project else 0x3
            self.last_linearisation_ok = ((st & need) == need).view(T, B)
        return DX.unflatten(-1, (T, B)), DU.unflatten(-1, (T, B))

    def backward(self, A, Bm, quad, reg):
        lxx, luu, lux, lx, lu, VxxT, VxT = quad
        n, m, T = self.n, self.m, self.T
        B = A.shape[-1]
        im = self.im
        im._use_current_stream()
        # rollout returns (row, col, T, B) views of column-major buffers: flatten back to (n*n, T, B)
        Af = A.transpose(0, 1).reshape(n * n, T, B).contiguous()
        Bf = Bm.transpose(0, 1).reshape(n * m, T, B).contiguous()
        K = torch.empty(m * n, T, B, dtype=torch.float64, device=im.device)
        k = torch.empty(m, T, B, dtype=torch.float64, device=im.device)
        dV = torch.empty(2, B, dtype=torch.float64, device=im.device)
        st = torch.empty(B, dtype=torch.int32, device=im.device)
        im.lib.check(im.lib.cdll.od_ilqr_backward(im._h, B, T, n, m, _ptr(Af), _ptr(Bf), _ptr(lxx), _ptr(luu), _ptr(lux),
                                                  _ptr(lx), _ptr(lu), _ptr(VxxT), _ptr(VxT), float(reg), _ptr(K), _ptr(k), _ptr(dV), _ptr(st)))
        return K, k, dV, st

    def forward(self, x1, X, U, K, k):
        """all step sizes at once -> Xc (n, T+1, nalpha*B), Uc (m, T, nalpha*B)"""
        n, m, T = self.n, self.m, self.T
        B, na = x1.shape[-1], self.alphas.numel()
        im = self.im
        if not isinstance(im, ImplicitDynamics):
            return im.rollout_policy(x1, X, U, K, k, self.alphas)
        im._use_current_stream()
        Xc = torch.empty(n, T + 1, na * B, dtype=torch.float64, device=im.device)
        Uc = torch.empty(m, T, na * B, dtype=torch.float64, device=im.device)
        st = torch.empty(T, na * B, dtype=torch.int32, device=im.device)
        im.lib.check(im.lib.cdll.od_rollout_policy(im._h, B, T, na, _ptr(self.alphas), _ptr(x1.contiguous()), _ptr(X.contiguous()),
                                                   _ptr(U.contiguous()), _ptr(K), _ptr(k), _ptr(Xc), _ptr(Uc), _ptr(st), 0))
        return Xc, Uc, st

    # -- solver --------------------------------------------------------------------------------
    # -- the iteration on the device (od_ilqr_*): what `solve` runs --------------------------------------------------
    def device_solver(self, B, max_iter=50, max_al_iter=1, rho_init=1.0, rho_scale=10.0, con_tol=1e-3, obj_tol=1e-6, history=0):
        """an od_ilqr solver object for B problems with this objective (buffers allocated once; see DeviceILQR)"""
        return DeviceILQR(self, B, max_iter=max_iter, max_al_iter=max_al_iter, rho_init=rho_init, rho_scale=rho_scale,
                          con_tol=con_tol, obj_tol=obj_tol, history=history)

    def solve(self, x1, U0, max_iter=50, max_al_iter=1, rho_init=1.0, rho_scale=10.0, con_tol=1e-3, obj_tol=1e-6, verbose=False):
        """iLQR.solve!: every iteration runs on the device behind the C ABI (od_ilqr_solve), no host round trip per iteration.
        -> X (n, T+1, B), U (m, T, B), J (B,), history (list of (B,) costs, one per iteration)"""
        x1 = self.im._prep(x1)
        U0 = self.im._prep(U0)
        key = (x1.shape[-1], max_iter, max_al_iter, rho_init, rho_scale, con_tol, obj_tol)
        if self._dev is None or self._dev.key != key:
            self._dev = self.device_solver(x1.shape[-1], max_iter, max_al_iter, rho_init, rho_scale, con_tol, obj_tol)
            self._dev.key = key
        d = self._dev
        d.solve(x1, U0)
        X, U, J = d.get()
        hist = d.history()
        if verbose:
            i = d.info()
            print("od_ilqr_solve: %d iterations, %d multiplier updates, max dJ %.3g, max violation %.3g, reg %.3g" %
                  (i.iterations, i.al_iterations, i.max_dJ, i.max_violation, i.reg))
        return X, U, J, [h for h in hist]

    # -- the same loop composed from the separate entry points, decisions on the host (the checker of `solve`) -------
    def _backward_each(self, A, Bm, quad, reg, active):
        """backward pass with every trajectory's own regularisation (and its own retry sequence reg -> 10 reg -> ... -> 1e6, as
        k_ilqr_backward_row does inside the kernel): od_ilqr_backward takes one reg per call, so trajectories are grouped by value.
        -> K, k, dV, bad (never factorised: K = k = dV = 0)"""
        n, m, T = self.n, self.m, self.T
        B = A.shape[-1]
        dev = self.im.device
        K = torch.zeros(m * n, T, B, dtype=torch.float64, device=dev)
        k = torch.zeros(m, T, B, dtype=torch.float64, device=dev)
        dV = torch.zeros(2, B, dtype=torch.float64, device=dev)
        todo = active.clone()
        reg_try = reg.clone()
        bad = torch.zeros(B, dtype=torch.bool, device=dev)
        while todo.any():
            for r in torch.unique(reg_try[todo]).tolist():
                idx = (todo & (reg_try == r)).nonzero().flatten()
                Ks, ks, dVs, bs = self.backward(A[..., idx], Bm[..., idx], tuple(q[..., idx].contiguous() for q in quad), r)
                good = bs == 1
                gi = idx[good]
                K[..., gi] = Ks[..., good]; k[..., gi] = ks[..., good]; dV[:, gi] = dVs[:, good]
                todo[gi] = False
                fail = idx[~good]
                give_up = fail[reg_try[fail] >= 1e6]
                bad[give_up] = True
                todo[give_up] = False
                again = fail[reg_try[fail] < 1e6]
                reg_try[again] = torch.clamp(torch.clamp(reg_try[again], min=1e-8) * 10.0, max=1e6)
        return K, k, dV, bad

    def solve_stepwise(self, *args, **kw):
        """`_solve_stepwise` with the thrust-cone projection's stall exit switched on for the handle while it runs, as the device-resident
        solver does for its own launches (od_ilqr_options.proj_stall_exit; the handle's default is off, like the reference)"""
        if getattr(self.obj, "parameter_stage", None) is not None:
            raise NotImplementedError("solve_stepwise has no parameter stage (od_ilqr_set_parameter_stage): the device solver with a parameter stage is checked by the numpy restatement under tests (solve_stages)")
        info = getattr(self.im, "info", None)
        if info is not None and hasattr(info, "set_projection_stall_exit"):
            before = getattr(info, "projection_stall_exit", False)       # (a caller's own setting survives this call)
            info.set_projection_stall_exit(kw.pop("proj_stall_exit", True))
            try:
                return self._solve_stepwise(*args, **kw)
            finally:
                info.set_projection_stall_exit(before)
        kw.pop("proj_stall_exit", None)
        return self._solve_stepwise(*args, **kw)

    def _solve_stepwise(self, x1, U0, max_iter=50, max_al_iter=1, rho_init=1.0, rho_scale=10.0, con_tol=1e-3, obj_tol=1e-6, verbose=False,
                        reuse_forward_states=True, rho_max=1e8):
        """The iteration of `solve` composed from the separate entry points with every decision taken here, on the host: the
        checker of od_ilqr_*.  Like there, the B problems are independent solves in lockstep launches: each trajectory has its own
        regularisation schedule, penalty, convergence flag (`done`) and constraint flag (`al_done`).
        reuse_forward_states: the accepted candidate of the forward pass IS the new nominal trajectory (its states were
        computed by the same time recursion), so the iteration linearises on those states knot by knot instead of rolling the
        trajectory out a second time (False: the second rollout, as a check)"""
        im, obj = self.im, self.obj
        x1 = im._prep(x1)
        U = im._prep(U0).clone()
        B, na = x1.shape[-1], self.alphas.numel()
        dev = im.device
        z = lambda *sh: torch.zeros(*sh, dtype=torch.float64, device=dev)
        general = obj.stage is not None or obj.terminal is not None      # stage / terminal rows: multipliers in a dict
        lam = None
        if general:
            lam = dict(lam=None if obj.goal_idx is None else z(obj.goal_idx.numel(), B),
                       lam_s=None if obj.stage is None else z(obj.stage[0].shape[0], self.T, B),
                       lam_t=None if obj.terminal is None else z(obj.terminal[0].shape[0], B))
        elif obj.goal_idx is not None:
            lam = z(obj.goal_idx.numel(), B)
        rho = torch.full((B,), rho_init if lam is not None else 0.0, dtype=torch.float64, device=dev)
        reg = torch.full((B,), float(self.reg), dtype=torch.float64, device=dev)
        done = torch.zeros(B, dtype=torch.bool, device=dev)
        al_done = torch.zeros(B, dtype=torch.bool, device=dev)

        def rep(l):                                   # multipliers of the candidates = those of their nominal trajectory
            if l is None:
                return None
            if isinstance(l, dict):
                return {k: (None if v is None else v.repeat(*([1] * (v.dim() - 1)), na)) for k, v in l.items()}
            return l.repeat(1, na)

        history = []
        X, A, Bm, st = self.linearize(x1, U)
        for al in range(max_al_iter):
            J = obj.value(X, U, lam, rho, im=im)
            for it in range(max_iter):
                if done.all():
                    break
                act = ~done
                quad = obj.expansion(X, U, lam, rho)
                K, k, dV, bad = self._backward_each(A, Bm, quad, reg, act)
                Xc, Uc, cst = self.forward(x1, X, U, K, k)
                Jc = obj.value(Xc, Uc, rep(lam), rho.repeat(na), im=im).view(na, B)
                ok_roll = ((cst & 1) == 1).all(0).view(na, B)
                expected = self.alphas[:, None] * dV[0][None, :] + self.alphas[:, None] ** 2 * dV[1][None, :]
                # (a zeroed trajectory reproduces its nominal: Jc == J and expected == 0 would pass the Armijo test -- it has no step)
                accept = ok_roll & torch.isfinite(Jc) & (Jc <= J[None, :] + self.c1 * expected) & ~bad[None, :] & act[None, :]
                first = torch.where(accept.any(0), accept.float().argmax(0), torch.full((B,), -1, device=dev, dtype=torch.long))
                took = first >= 0
                sel = torch.clamp(first, min=0) * B + torch.arange(B, device=dev)
                U = torch.where(took[None, None, :], Uc[:, :, sel].double(), U)
                Jn = torch.where(took, Jc.reshape(-1)[sel], J)
                dJ = (J - Jn)
                if took.any():                                   # nothing moved: the linearisation is still valid
                    if reuse_forward_states:
                        X = torch.where(took[None, None, :], Xc[:, :, sel].double(), X)
                        A, Bm = self.linearize_at(X, U)
                        J = Jn
                    else:
                        X, A, Bm, st = self.linearize(x1, U)
                        J = obj.value(X, U, lam, rho, im=im)
                history.append(J.clone())
                # every trajectory's own bookkeeping (k_il_finish)
                stuck = act & ~took
                reg = torch.where(stuck, torch.clamp(reg * 10.0, max=1e6), reg)
                moved = act & took
                reg = torch.where(moved, torch.clamp(reg / 5.0, min=float(self.reg)), reg)
                done = done | (stuck & (reg >= 1e6)) | (moved & (dJ < obj_tol))
                if verbose:
                    print("al %d it %d  J mean %.6g  accepted %d/%d  converged %d  max dJ %.3g" % (al, it, J.mean().item(), int(took.sum()), B, int(done.sum()), dJ.max().item()))
            if lam is None:
                break
            viol = obj.violation(X, U)
            met = ~al_done & (viol < con_tol)
            al_done = al_done | met
            done = done | met
            if al + 1 == max_al_iter or al_done.all():
                break
            upd = ~al_done
            if general:
                if obj.goal_idx is not None:
                    lam["lam"] = torch.where(upd[None, :], lam["lam"] + rho * obj.constraint(X), lam["lam"])
                if obj.terminal is not None:
                    l = lam["lam_t"] + rho * obj.terminal_c(X); kk = obj.terminal[2]
                    l[:kk] = torch.where(l[:kk] > 0, l[:kk], torch.zeros_like(l[:kk]))
                    lam["lam_t"] = torch.where(upd[None, :], l, lam["lam_t"])
                if obj.stage is not None:
                    l = lam["lam_s"] + rho * obj.stage_c(X, U); kk = obj.stage[3]
                    l[:kk] = torch.where(l[:kk] > 0, l[:kk], torch.zeros_like(l[:kk]))
                    lam["lam_s"] = torch.where(upd[None, None, :], l, lam["lam_s"])
            else:
                lam = torch.where(upd[None, :], lam + rho * obj.constraint(X), lam)
            rho = torch.where(upd, torch.clamp(rho * rho_scale, max=rho_max), rho)
            reg = torch.where(upd, torch.full_like(reg, float(self.reg)), reg)
            done = torch.where(upd, torch.zeros_like(done), done)
        self.last_status = dict(done=done, al_done=al_done, rho=rho, reg=reg)
        return X, U, J, history


class DeviceILQR:
    """od_ilqr_* (include/od_mi355x.h): the whole iLQR iteration on the device, decisions included.  `iterate(n)` only enqueues
    kernels on the current stream (capturable in a HIP graph); `solve` is od_ilqr_solve."""

    def __init__(self, ilqr: ILQR, B, max_iter=50, max_al_iter=1, rho_init=1.0, rho_scale=10.0, con_tol=1e-3, obj_tol=1e-6, history=0,
                 proj_stall_exit=True, rho_max=1e8):
        import ctypes as C
        from . import _lib
        im, obj = ilqr.im, ilqr.obj
        self.im, self.lib, self.B, self.T, self.n, self.m = im, im.lib, int(B), ilqr.T, ilqr.n, ilqr.m
        o = _lib.IlqrOptions()
        self.lib.check(self.lib.cdll.od_ilqr_default_options(C.byref(o)))
        o.reg, o.c1, o.obj_tol, o.con_tol = float(ilqr.reg), float(ilqr.c1), float(obj_tol), float(con_tol)
        o.rho_init, o.rho_scale, o.max_iter, o.max_al_iter = float(rho_init), float(rho_scale), int(max_iter), int(max_al_iter)
        o.project = 1 if getattr(im, "project", True) else 0
        o.history = int(history)
        o.proj_stall_exit = 1 if proj_stall_exit else 0
        o.rho_max = float(rho_max)
        self.options = o
        al = np.ascontiguousarray(ilqr.alphas.cpu().numpy(), dtype=np.float64)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        h = C.c_void_p()
        im._use_current_stream()
        self.lib.check(self.lib.cdll.od_ilqr_create(im._h, self.B, self.T, al.size, dp(al), C.byref(o), C.byref(h)))
        self._s = h
        f = lambda t: np.ascontiguousarray(t.cpu().numpy(), dtype=np.float64)
        Q, R, QT, xr = f(obj.Q.T), f(obj.R.T), f(obj.QT.T), f(obj.x_ref)            # column-major flattenings
        if obj.goal_idx is not None:
            gi = np.ascontiguousarray(obj.goal_idx.cpu().numpy(), dtype=np.int32)
            g = f(obj.goal)
            self.lib.check(self.lib.cdll.od_ilqr_set_objective(self._s, dp(Q), dp(R), dp(QT), dp(xr), gi.size, gi.ctypes.data_as(C.POINTER(C.c_int)), dp(g)))
        else:
            self.lib.check(self.lib.cdll.od_ilqr_set_objective(self._s, dp(Q), dp(R), dp(QT), dp(xr), 0, None, None))
        if obj.stage is not None or obj.terminal is not None:
            cm = lambda M: np.ascontiguousarray(M.T).reshape(-1)               # column-major flattening
            ns = nt = nsi = nti = 0
            Cs = Ds = ds = Ct = dt = None
            if obj.stage is not None:
                Cs, Ds, ds, nsi = cm(obj.stage[0]), cm(obj.stage[1]), obj.stage[2], obj.stage[3]
                ns = obj.stage[2].size
            if obj.terminal is not None:
                Ct, dt, nti = cm(obj.terminal[0]), obj.terminal[1], obj.terminal[2]
                nt = obj.terminal[1].size
            q = lambda a: None if a is None else dp(a)
            self._keep = (Cs, Ds, ds, Ct, dt)
            self.lib.check(self.lib.cdll.od_ilqr_set_constraints(self._s, ns, nsi, q(Cs), q(Ds), q(ds), nt, nti, q(Ct), q(dt)))
        ps = getattr(obj, "parameter_stage", None)
        if ps is not None:
            q = _lib.IlqrParameterStage()
            cid = -1
            if ps["constraint"] is not None:
                cid = self.lib.cdll.od_constraint_id(ps["constraint"].encode())
                if cid < 0:
                    raise _lib.ODError("constraint function %r is not in this library (python -m optimization_dynamics_amd.codegen --add-constraint)" % ps["constraint"])
            cm = lambda M: np.ascontiguousarray(M.T).reshape(-1)
            keep = [ps["w_theta"], ps["p"]]
            q.constraint, q.n_p, q.p = cid, (0 if ps["p"] is None else ps["p"].size), (None if ps["p"] is None else dp(ps["p"]))
            q.w_theta, q.cost_const = dp(ps["w_theta"]), ps["cost_const"]
            if ps["terminal"] is not None:
                cx, cth, dtt, ni = ps["terminal"]
                keep += [cm(cx), cm(cth), dtt]
                q.nt, q.nt_ineq, q.Ct_x, q.Ct_theta, q.dt = dtt.size, ni, dp(keep[-3]), dp(keep[-2]), dp(keep[-1])
            self._keep_ps = keep
            self.lib.check(self.lib.cdll.od_ilqr_set_parameter_stage(self._s, C.byref(q)))
        if ilqr.bundle is not None:
            gb = ilqr.bundle
            self._keep_eta = np.ascontiguousarray(gb.eta.T, dtype=np.float64)           # (nzb x N) column-major == N rows of nzb
            self.lib.check(self.lib.cdll.od_ilqr_set_gradient_bundle(self._s, int(gb.N), self._keep_eta.ctypes.data_as(C.c_void_p)))
        self.max_hist = history if history > 0 else max_iter * max_al_iter

    def __del__(self):
        try:
            if getattr(self, "_s", None) and getattr(self.im, "_h", None):
                self.lib.cdll.od_ilqr_destroy(self._s)
            self._s = None
        except Exception:
            pass

    def _args(self, x1, U0):
        x1 = self.im._prep(x1)
        U0 = self.im._prep(U0)
        assert x1.shape == (self.n, self.B) and U0.shape == (self.m, self.T, self.B)
        self.im._use_current_stream()
        return x1, U0

    def init(self, x1, U0):
        x1, U0 = self._args(x1, U0)
        self.lib.check(self.lib.cdll.od_ilqr_init(self._s, _ptr(x1), _ptr(U0)))

    def iterate(self, n=1):
        self.im._use_current_stream()
        self.lib.check(self.lib.cdll.od_ilqr_iterate(self._s, int(n)))

    def al_update(self):
        self.im._use_current_stream()
        self.lib.check(self.lib.cdll.od_ilqr_al_update(self._s))

    def solve(self, x1, U0):
        x1, U0 = self._args(x1, U0)
        self.lib.check(self.lib.cdll.od_ilqr_solve(self._s, _ptr(x1), _ptr(U0)))

    def get(self, gains=False):
        dev = self.im.device
        X = torch.empty(self.n, self.T + 1, self.B, dtype=torch.float64, device=dev)
        U = torch.empty(self.m, self.T, self.B, dtype=torch.float64, device=dev)
        J = torch.empty(self.B, dtype=torch.float64, device=dev)
        K = torch.empty(self.m * self.n, self.T, self.B, dtype=torch.float64, device=dev) if gains else None
        k = torch.empty(self.m, self.T, self.B, dtype=torch.float64, device=dev) if gains else None
        self.im._use_current_stream()
        self.lib.check(self.lib.cdll.od_ilqr_get(self._s, _ptr(X), _ptr(U), _ptr(J), _ptr(K) if gains else None, _ptr(k) if gains else None))
        return (X, U, J, K, k) if gains else (X, U, J)

    def info(self):
        import ctypes as C
        from . import _lib
        i = _lib.IlqrInfo()
        self.im._use_current_stream()
        self.lib.check(self.lib.cdll.od_ilqr_get_info(self._s, C.byref(i)))
        return i

    def status(self):
        """per trajectory: flags (bit 0 inner loop converged, bit 1 constraints met), violation at the last multiplier round, penalty"""
        dev = self.im.device
        fl = torch.empty(self.B, dtype=torch.int32, device=dev)
        v = torch.empty(self.B, dtype=torch.float64, device=dev)
        r = torch.empty(self.B, dtype=torch.float64, device=dev)
        self.im._use_current_stream()
        self.lib.check(self.lib.cdll.od_ilqr_get_status(self._s, _ptr(fl), _ptr(v), _ptr(r)))
        return fl, v, r

    def trace(self):
        """the decisions of every iteration since init, (iterations, B) each: index of the accepted step size (-1 none, -2 the
        trajectory had converged before), regularisation after the iteration, penalty (od_ilqr_get_trace)"""
        dev = self.im.device
        sel = torch.empty(self.max_hist, self.B, dtype=torch.int32, device=dev)
        reg = torch.empty(self.max_hist, self.B, dtype=torch.float64, device=dev)
        rho = torch.empty(self.max_hist, self.B, dtype=torch.float64, device=dev)
        self.im._use_current_stream()
        rows = self.lib.cdll.od_ilqr_get_trace(self._s, _ptr(sel), _ptr(reg), _ptr(rho), self.max_hist)
        if rows < 0:
            self.lib.check(rows)
        return sel[:rows], reg[:rows], rho[:rows]

    def history(self):
        """(iterations, B): the costs after every iteration since init"""
        H = torch.empty(self.max_hist, self.B, dtype=torch.float64, device=self.im.device)
        self.im._use_current_stream()
        rows = self.lib.cdll.od_ilqr_get_history(self._s, _ptr(H), self.max_hist)
        if rows < 0:
            self.lib.check(rows)
        return H[:rows]
