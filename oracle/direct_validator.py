"""Comparison baseline of the reference, CPU side -- TEST INFRASTRUCTURE ONLY (the direct-method leg).

`examples/comparisons/hopper.jl` solves the hopper's gait task with a contact-implicit DIRECT method: one nonlinear programme over
the whole trajectory -- configurations, controls, contact impulses gamma, friction beta / psi / eta and a complementarity slack -- with
the time-stepping dynamics (:6-37) and the contact conditions (:57-162) as constraints, the slack penalised in the cost (:205-222),
handed to Ipopt through DirectTrajectoryOptimization.jl (:290-303); its solution is then priced under the iLQR example's cost
(:318-357) and set beside the result of the paper's method.  Ipopt, DirectTrajectoryOptimization.jl and MuJoCo (the other
comparison, examples/comparisons/acrobot) do not exist in this image and cannot be installed; scipy.optimize does.  What is built here
is that comparison on the CPU oracle:

  task      the gait problem as the device-resident solver poses it (tests/ilqr_checks.py::hopper_example: examples/hopper.jl:12-13,
            178-220 with the initial configuration fixed -- T = 21, h = 0.05, hop 0.5 m and land in the starting pose, |u| <= 10,
            cost 1/2 (x - x_ref)' 0.1 diag(1,10,1,10,1,10,1,10) (x - x_ref) + 1/2 0.1 |u|^2, terminal 1/2 |x_T - x_ref|^2);
  direct    unknowns per step t: the whole solution vector z_t = [q_{t+1} (4), gamma (4), s_gamma (4), psi (2), b (2), s_psi (2), s_b (2)]
            of the oracle's hopper residual, the control u_t (2) and one complementarity slack s_t >= 0 (the role of s_alpha in :88,
            :103-106); constraints: the 12 equality rows of the oracle's residual r(z_t; [q_{t-1}, q_t, u_t, mu, h]) = 0 (discrete
            Euler-Lagrange equations, signed distances, friction-cone and tangential-velocity rows: what :6-37 and :140-162 state), the
            8 complementarity rows within the slack, |r_bil(z_t)| <= s_t (:100-106,128-131), cone membership (orthant variables >= 0,
            psi >= |b|, s_psi >= |s_b|), control bounds (:262-263), the gait's terminal conditions (:282-296); cost: the iLQR cost of the
            trajectory + 1000 sum_t s_t (:209,217).  Solved by scipy.optimize.minimize(method="trust-constr") with analytic Jacobians
            (the oracle's rz / rtheta) and quasi-Newton Hessians, started from the iLQR solution;
  iLQR      oracle/ilqr_np.py::solve on the same task with the oracle's f / fx / fu (the numpy twin of od_ilqr_solve, which the GPU tier
            compares with the device solver decision by decision).

`compare()` returns both objectives under the iLQR cost (:318-357), both constraint violations and the consistency of the direct
solution with the time-stepping simulator (its controls rolled out through the oracle's f).  tests/test_fd_validator.py asserts
that the two methods agree on the task: both feasible, objectives within 10 %."""
import numpy as np

from . import ilqr_np as N
from . import oracle as O

NQ, NU, NZ = 4, 2, 20
EQ_ROWS = list(range(12))
BIL_ROWS = list(range(12, 20))
ORT = list(range(4, 12))                     # gamma, s_gamma >= 0
SOC = [(12, 14), (13, 15), (16, 18), (17, 19)]   # (psi_i, b_i), (s_psi_i, s_b_i): first >= |second|


def gait_task(h=0.05, T=20, foot_radius=0.05):
    q1 = np.array([0.0, 0.5 + foot_radius, 0.0, 0.5])
    q_ref = np.array([0.5, 0.75 + foot_radius, 0.0, 0.25])
    x1, x_ref = np.concatenate([q1, q1]), np.concatenate([q_ref, q_ref])
    w = np.array([1.0, 10.0, 1.0, 10.0] * 2)
    Q, R, QT = 0.1 * np.diag(w), 0.1 * np.eye(2), np.eye(8)
    Cs = np.zeros((4, 8)); Ds = np.vstack([-np.eye(2), np.eye(2)]); ds = np.full(4, 10.0)
    Ct = np.zeros((8, 8)); dt = np.zeros(8)
    Ct[0, 0] = -1.0; dt[0] = -(0.5 + x1[0])              # x_travel - (x[1] - theta[1]) <= 0   (:291-292)
    Ct[1, 4] = -1.0; dt[1] = -(0.5 + x1[4])
    for k, i in enumerate([1, 2, 3, 5, 6, 7]):           # the other coordinates as at the start   (:288-289)
        Ct[2 + k, i] = 1.0; dt[2 + k] = x1[i]
    return dict(h=h, T=T, x1=x1, x_ref=x_ref, Q=Q, R=R, QT=QT, stage=(Cs, Ds, ds, 4), terminal=(Ct, dt, 2), u_max=10.0)


def ilqr_cost(task, X, U):
    dx = X - task["x_ref"]
    return float(0.5 * np.einsum("ti,ij,tj->", dx[:-1], task["Q"], dx[:-1]) + 0.5 * np.einsum("ti,ij,tj->", U, task["R"], U)
                 + 0.5 * dx[-1] @ task["QT"] @ dx[-1])


def solve_ilqr(task, sim):
    step, lin = N.mechanical_dynamics(sim)
    p = N.Problem(step, lin, task["Q"], task["R"], task["QT"], task["x_ref"], stage=task["stage"], terminal=task["terminal"])
    T = task["T"]
    U0 = np.zeros((T, NU)); U0[:, 1] = 9.81 * 3.0 * 0.5 * task["h"]          # the standing control, examples/hopper.jl:270
    r = N.solve(p, task["x1"], U0, alphas=tuple(2.0 ** -i for i in range(17)), max_iter=10, max_al_iter=15, con_tol=1e-3, obj_tol=1e-3)
    return r


class DirectNLP:
    """w = [z_0 .. z_{T-1} | u_0 .. u_{T-1} | s_0 .. s_{T-1}]"""

    def __init__(self, task, sim):
        self.task, self.sim = task, sim
        self.T = T = task["T"]
        self.oz, self.ou, self.os, self.n = 0, NZ * T, NZ * T + NU * T, NZ * T + NU * T + T
        self.fric = np.array([sim.fric[0], sim.fric[1]])
        self.q0 = task["x1"][:NQ].copy()                      # q_{-1} = q_0 = the fixed initial configuration

    def split(self, w):
        T = self.T
        return w[:self.ou].reshape(T, NZ), w[self.ou:self.os].reshape(T, NU), w[self.os:]

    def qprev(self, Z, t):
        """(q_{t-1}, q_t) of step t: configurations -1 and 0 are fixed, q_{t} = z_{t-1}[0:4]"""
        qa = self.q0 if t < 2 else Z[t - 2, :NQ]
        qb = self.q0 if t < 1 else Z[t - 1, :NQ]
        return qa, qb

    def theta(self, Z, U, t):
        qa, qb = self.qprev(Z, t)
        return np.concatenate([qa, qb, U[t], self.fric, [self.task["h"]]])

    def trajectory(self, w):
        Z, U, s = self.split(w)
        q = [self.q0, self.q0] + [Z[t, :NQ] for t in range(self.T)]
        X = np.array([np.concatenate([q[t], q[t + 1]]) for t in range(self.T + 1)])
        return X, U

    # objective: the iLQR cost of the trajectory + 1000 sum s
    def cost(self, w):
        X, U = self.trajectory(w)
        return ilqr_cost(self.task, X, U) + 1000.0 * self.split(w)[2].sum()

    def cost_grad(self, w):
        Z, U, s = self.split(w)
        X, _ = self.trajectory(w)
        task, T = self.task, self.T
        gx = (X[:-1] - task["x_ref"]) @ task["Q"]             # d/dx_t, t < T
        gT = (X[-1] - task["x_ref"]) @ task["QT"]
        g = np.zeros(self.n)
        gz = g[:self.ou].reshape(T, NZ)
        G = np.vstack([gx, gT[None]])                         # d cost / d x_t, t = 0 .. T
        # x_t = [c_t, c_{t+1}] with c_0 = c_1 fixed and c_{t+2} = z_t[0:4]: z_t's configuration is the second half of x_{t+1} and the
        # first half of x_{t+2}
        for t in range(T):
            gz[t, :NQ] += G[t + 1, NQ:]
            if t + 2 <= T:
                gz[t, :NQ] += G[t + 2, :NQ]
        g[self.ou:self.os] = (U @ task["R"]).reshape(-1)
        g[self.os:] = 1000.0
        return g

    # equality constraints: the 12 equality rows of every step's residual
    def eq(self, w):
        Z, U, s = self.split(w)
        return np.concatenate([O.eval_r("hopper", Z[t], self.theta(Z, U, t), 0.0)[EQ_ROWS] for t in range(self.T)])

    def _jac_rows(self, w, rows, sign=1.0, slack=False):
        """Jacobian of r[rows] of every step w.r.t. w (dense (T * len(rows), n))"""
        Z, U, s = self.split(w)
        T, nr = self.T, len(rows)
        J = np.zeros((T * nr, self.n))
        for t in range(T):
            th = self.theta(Z, U, t)
            rz = O.eval_rz("hopper", Z[t], th)[rows]
            rth = O.eval_rth("hopper", Z[t], th)[rows]
            r0 = t * nr
            J[r0:r0 + nr, NZ * t:NZ * (t + 1)] = sign * rz
            if t >= 2:
                J[r0:r0 + nr, NZ * (t - 2):NZ * (t - 2) + NQ] += sign * rth[:, 0:4]
            if t >= 1:
                J[r0:r0 + nr, NZ * (t - 1):NZ * (t - 1) + NQ] += sign * rth[:, 4:8]
            J[r0:r0 + nr, self.ou + NU * t:self.ou + NU * (t + 1)] = sign * rth[:, 8:10]
            if slack:
                J[r0:r0 + nr, self.os + t] = 1.0
        return J

    def eq_jac(self, w):
        return self._jac_rows(w, EQ_ROWS)

    # inequalities (>= 0): s_t - r_bil >= 0 and s_t + r_bil >= 0; cone membership; terminal conditions of the gait
    def ineq(self, w):
        Z, U, s = self.split(w)
        rb = np.stack([O.eval_r("hopper", Z[t], self.theta(Z, U, t), 0.0)[BIL_ROWS] for t in range(self.T)])
        out = [(s[:, None] - rb).reshape(-1), (s[:, None] + rb).reshape(-1)]
        for a, b in SOC:
            out.append(Z[:, a] - Z[:, b]); out.append(Z[:, a] + Z[:, b])
        X, _ = self.trajectory(w)
        Ct, dt, k = self.task["terminal"]
        ct = Ct @ X[-1] - dt
        out.append(-ct[:k])
        return np.concatenate(out)

    def ineq_jac(self, w):
        T = self.T
        J1 = self._jac_rows(w, BIL_ROWS, sign=-1.0, slack=True)
        J2 = self._jac_rows(w, BIL_ROWS, sign=1.0, slack=True)
        rows = [J1, J2]
        for a, b in SOC:
            for sg in (-1.0, 1.0):
                J = np.zeros((T, self.n))
                for t in range(T):
                    J[t, NZ * t + a] = 1.0; J[t, NZ * t + b] = sg
                rows.append(J)
        rows.append(-self._terminal_jac()[: self.task["terminal"][2]])
        return np.vstack(rows)

    def _terminal_jac(self):
        """d (Ct x_T) / d w with x_T = [q_T, q_{T+1}] = [z_{T-2}[0:4], z_{T-1}[0:4]]"""
        Ct = self.task["terminal"][0]
        J = np.zeros((Ct.shape[0], self.n))
        T = self.T
        J[:, NZ * (T - 2):NZ * (T - 2) + NQ] = Ct[:, :NQ]
        J[:, NZ * (T - 1):NZ * (T - 1) + NQ] = Ct[:, NQ:]
        return J

    def term_eq(self, w):
        X, _ = self.trajectory(w)
        Ct, dt, k = self.task["terminal"]
        return (Ct @ X[-1] - dt)[k:]

    def term_eq_jac(self, w):
        return self._terminal_jac()[self.task["terminal"][2]:]

    def bounds(self):
        lo, hi = np.full(self.n, -np.inf), np.full(self.n, np.inf)
        for t in range(self.T):
            for i in ORT:
                lo[NZ * t + i] = 0.0
            for a, _ in SOC:
                lo[NZ * t + a] = 0.0
        lo[self.ou:self.os], hi[self.ou:self.os] = -self.task["u_max"], self.task["u_max"]
        lo[self.os:] = 0.0
        return list(zip(lo, hi))

    def start_from(self, X, U):
        """w from a trajectory of the time-stepping simulator: z_t = the oracle's whole solution of step t, s_t = its complementarity"""
        T = self.T
        Z = np.zeros((T, NZ)); s = np.zeros(T)
        for t in range(T):
            ok, z, dz, it = O.step_full(self.sim, X[t], U[t], self.sim.opts.kappa_tol, False)
            Z[t] = z
        w = np.concatenate([Z.reshape(-1), np.asarray(U).reshape(-1), s])
        Zs, Us, _ = self.split(w)
        for t in range(T):
            s[t] = np.abs(O.eval_r("hopper", Zs[t], self.theta(Zs, Us, t), 0.0)[BIL_ROWS]).max()
        w[self.os:] = s
        return w


def compare(maxiter=800, verbose=False):
    import scipy.sparse as sp
    from scipy.optimize import BFGS, Bounds, NonlinearConstraint, minimize
    task = gait_task()
    sim = O.make_sim("hopper", task["h"], kappa_tol=1e-4, kappa_grad_tol=1e-3, friction=[0.5, 0.5])
    r = solve_ilqr(task, sim)
    Xi, Ui = r["X"], r["U"]
    nlp = DirectNLP(task, sim)
    w0 = nlp.start_from(Xi, Ui)
    lo, hi = map(np.array, zip(*nlp.bounds()))
    cons = [NonlinearConstraint(nlp.eq, 0, 0, jac=lambda w: sp.csr_matrix(nlp.eq_jac(w)), hess=BFGS()),
            NonlinearConstraint(nlp.term_eq, 0, 0, jac=lambda w: sp.csr_matrix(nlp.term_eq_jac(w)), hess=BFGS()),
            NonlinearConstraint(nlp.ineq, 0, np.inf, jac=lambda w: sp.csr_matrix(nlp.ineq_jac(w)), hess=BFGS())]
    # (SLSQP leaves the feasible region of the complementarity rows at its first step and never returns -- measured; the
    # trust-region interior-point method of scipy, quasi-Newton Hessians, is the stand-in for Ipopt)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = minimize(nlp.cost, w0, jac=nlp.cost_grad, hess=BFGS(), bounds=Bounds(lo, hi), constraints=cons, method="trust-constr",
                       options=dict(maxiter=maxiter, verbose=1 if verbose else 0, gtol=1e-6, xtol=1e-10, initial_tr_radius=0.1))
    w = res.x
    Xd, Ud = nlp.trajectory(w)
    # the direct solution's controls through the time-stepping simulator: the two contact models must describe the same motion
    Xr, _, okr = N.rollout(N.Problem(*N.mechanical_dynamics(sim), task["Q"], task["R"], task["QT"], task["x_ref"]), task["x1"], Ud)
    Ct, dt, k = task["terminal"]
    ct = Ct @ Xd[-1] - dt
    out = dict(
        ilqr_objective=ilqr_cost(task, Xi, Ui), ilqr_violation=r["violation"], ilqr_iterations=len(r["log"]),
        direct_objective=ilqr_cost(task, Xd, Ud), direct_start_objective=ilqr_cost(task, *nlp.trajectory(w0)),
        direct_slack_sum=float(nlp.split(w)[2].sum()), direct_slack_max=float(nlp.split(w)[2].max()),
        direct_equality_violation=float(max(np.abs(nlp.eq(w)).max(), np.abs(nlp.term_eq(w)).max())),
        direct_inequality_violation=float(max(0.0, -nlp.ineq(w).min())),
        direct_terminal_violation=float(max(np.maximum(ct[:k], 0).max(), np.abs(ct[k:]).max())),
        direct_iterations=int(res.nit), direct_status=int(res.status), direct_message=str(res.message), direct_optimality=float(res.optimality),
        rollout_of_direct_controls_state_diff=float(np.abs(Xr - Xd).max()), rollout_converged=bool(okr),
        travel_ilqr=float(Xi[-1, 4]), travel_direct=float(Xd[-1, 4]))
    return out


if __name__ == "__main__":
    import json
    import time
    O.build()
    t0 = time.time()
    print(json.dumps(compare(verbose=True), indent=1), "%.1f s" % (time.time() - t0))
