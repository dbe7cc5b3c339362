"""The call sequences of julia/OptimizationDynamicsMI355X.jl, replayed through ctypes (Julia is absent here): every
function of the shim is one or two ccalls on host vectors; this file makes exactly those calls, with the same argument
order, NULLs and buffer shapes, against the host-emulation build (CPU tier) and checks the results against the batched
entry points / the oracle.  A wrong argument order or a missing symbol in the shim's ccalls shows up here."""
import ctypes as C
import re
import os

import numpy as np
import pytest
import torch

import parity_checks as P
import workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _p(a):
    return a.ctypes.data if a is not None else None


def test_every_ccall_of_the_shim_names_an_exported_symbol_with_matching_arity(emu_lib):
    from optimization_dynamics_amd import _lib
    src = open(os.path.join(ROOT, "julia", "OptimizationDynamicsMI355X.jl")).read()
    calls = re.findall(r"ccall\(\(:(\w+), LIB\), (\w+), \(([^)]*)\)", src)
    assert len(calls) >= 20
    for sym, ret, args in calls:
        assert hasattr(emu_lib.cdll, sym), sym
        if sym in _lib.SIGNATURES:
            jargs = [a.strip() for a in args.split(",") if a.strip()]
            assert len(jargs) == len(_lib.SIGNATURES[sym][1]), (sym, len(jargs), len(_lib.SIGNATURES[sym][1]))
            # ... and the same class of C type in every position (a Cint where the header says long reads garbage on x86-64)
            import ctypes as C
            def jclass(t):
                return "ptr" if t.startswith(("Ptr", "Ref", "Cstring")) else {"Cint": "int", "Clong": "long", "Cdouble": "double", "Csize_t": "size"}[t]
            def cclass(t):
                return {C.c_int: "int", C.c_long: "long", C.c_double: "double", C.c_size_t: "size"}.get(t, "ptr")
            assert [jclass(a) for a in jargs] == [cclass(t) for t in _lib.SIGNATURES[sym][1]], sym
            assert jclass(ret) == cclass(_lib.SIGNATURES[sym][0]), sym


def _c_struct_fields(header, name):
    """[(type, field)] of `typedef struct { ... } name;` in the header, in declaration order"""
    end = re.search(r"\}\s*%s;" % name, header).start()
    start = header.rindex("typedef struct {", 0, end) + len("typedef struct {")
    body = re.sub(r"/\*.*?\*/", "", header[start:end], flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            ty, names = decl.split(None, 1)
            out += [(ty, n.strip()) for n in names.split(",")]
    return out


def test_struct_mirrors_match_the_header_field_for_field():
    """the structs the C ABI passes by pointer (od_options, od_ilqr_options, od_ilqr_info) are mirrored twice -- ctypes
    (optimization_dynamics_amd/_lib.py) and Julia (julia/OptimizationDynamicsMI355X.jl) -- and nothing but this test holds the three
    declarations together: same fields, same order, same C types"""
    import ctypes as C
    from optimization_dynamics_amd import _lib
    header = open(os.path.join(ROOT, "include", "od_mi355x.h")).read()
    jl = open(os.path.join(ROOT, "julia", "OptimizationDynamicsMI355X.jl")).read()
    ctype = {"double": C.c_double, "int": C.c_int, "long": C.c_long}
    jtype = {"double": "Cdouble", "int": "Cint", "long": "Clong"}
    for cname, py, jname in (("od_options", _lib.Options, "ODOptions"), ("od_ilqr_options", _lib.IlqrOptions, "ILQROptions"),
                             ("od_ilqr_info", _lib.IlqrInfo, "ILQRInfo")):
        fields = _c_struct_fields(header, cname)
        assert len(fields) >= 9
        assert [(n, t) for n, t in py._fields_] == [(n, ctype[ty]) for ty, n in fields], cname
        m = re.search(r"struct %s\b[^\n]*\n(.*?)\nend" % jname, jl, re.S)
        jf = [tuple(f.strip().split("::")) for line in m.group(1).split("\n") for f in line.split("#")[0].split(";") if f.strip()]
        assert jf == [(n, jtype[ty]) for ty, n in fields], (jname, jf)


def _implicit_dynamics_callbacks_sequence(oracle, emu_lib, device):
    """f / fx / fu / ffxfu! (od_f_host, od_fx_host, od_fu_host, od_ffxfu_host) incl. the friction re-sync before each"""
    name = "cartpole_friction"
    X, U = W.knots(name, 3, seed=51)
    im = P.make_im(name, emu_lib, device)
    h = im._h
    mu = np.array([0.35, 0.35])
    n, nu = 4, 1
    for b in range(3):
        x, u = np.ascontiguousarray(X[:, b]), np.ascontiguousarray(U[:, b])
        d = np.zeros(n); dxb = np.zeros((n, n), order="F"); dub = np.zeros((n, nu), order="F")
        assert emu_lib.cdll.od_set_friction(h, mu.ctypes.data_as(C.POINTER(C.c_double)), 2) == 0
        assert emu_lib.cdll.od_f_host(h, _p(x), _p(u), _p(d)) == 0
        assert emu_lib.cdll.od_fx_host(h, _p(x), _p(u), _p(dxb)) == 0
        assert emu_lib.cdll.od_fu_host(h, _p(x), _p(u), _p(dub)) == 0
        d2 = np.zeros(n); dx2 = np.zeros((n, n), order="F"); du2 = np.zeros((n, nu), order="F")
        assert emu_lib.cdll.od_ffxfu_host(h, _p(x), _p(u), _p(d2), _p(dx2), _p(du2)) == 0
        assert np.array_equal(d, d2) and np.array_equal(dxb, dx2) and np.array_equal(dub, du2)
        sim = P.make_sim(oracle, name)
        so, do, _ = oracle.f(sim, x, u)
        _, dxo, _ = oracle.fx(sim, x, u)
        _, duo, _ = oracle.fu(sim, x, u)
        assert np.abs(d - do).max() < 1e-6 and np.abs(dxb - dxo).max() < 1e-4 * max(1, np.abs(dxo).max()) and np.abs(dub - duo).max() < 1e-4 * max(1, np.abs(duo).max())
    # partial outputs: NULLs as the shim passes them
    d3 = np.zeros(n)
    assert emu_lib.cdll.od_ffxfu_host(h, _p(x), _p(u), _p(d3), None, None) == 0 and np.array_equal(d3, d2)


def _gradient_bundle_sequence(oracle, emu_lib, device):
    """gradient! / fx_gb / fu_gb: od_bundle_grad_host(h, N, x, u, eta, dz, NULL)"""
    from optimization_dynamics_amd import gradient_bundle as gbm, models
    name = "hopper"
    X, U = W.knots(name, 2, seed=31)
    gb = gbm.GradientBundle(models.BY_NAME[name], N=50, eps=1e-4, seed=5)
    im = P.make_im(name, emu_lib, device, info=gb)
    nq, nzb = 4, 10
    dzb, st = gbm.gradient_batch(im, gb, torch.tensor(X), torch.tensor(U))
    for b in range(2):
        x, u = np.ascontiguousarray(X[:, b]), np.ascontiguousarray(U[:, b])
        dz = np.zeros((nq, nzb), order="F")
        eta = np.asfortranarray(gb.eta)
        assert emu_lib.cdll.od_bundle_grad_host(im._h, 50, _p(x), _p(u), _p(eta), _p(dz), None) == 0
        assert np.array_equal(dz, dzb[:, :, b].cpu().numpy())


def _rocket_sequence(oracle, emu_lib, device):
    """RocketInfo: od_create(rocket, NULL opts) + od_set_u_max; f/fx/fu_rocket(_proj) = od_rocket_host with NULLs;
    soc_projection(_gradient) = od_soc_project_host"""
    from optimization_dynamics_amd import models, rocket as rk
    hd = C.c_void_p()
    assert emu_lib.cdll.od_create(5, 0, None, C.c_double(0.05), C.byref(hd)) == 0
    assert emu_lib.cdll.od_set_u_max(hd, C.c_double(12.5)) == 0
    Xr, Ur = W.rocket_inputs(3, seed=1)
    info = rk.RocketInfo(models.rocket, 12.5, 0.05, device=device, lib=emu_lib)
    for project in (0, 1):
        Yb, DXb, DUb, UPb, stb = info.solve(torch.tensor(Xr), torch.tensor(Ur), project=bool(project), grads=True)
        for b in range(3):
            x, u = np.ascontiguousarray(Xr[:, b]), np.ascontiguousarray(Ur[:, b])
            y = np.zeros(12); dx = np.zeros((12, 12), order="F"); du = np.zeros((12, 3), order="F")
            assert emu_lib.cdll.od_rocket_host(hd, project, _p(x), _p(u), _p(y), None, None, None, None) == 0       # f_rocket(_proj)
            assert emu_lib.cdll.od_rocket_host(hd, project, _p(x), _p(u), None, _p(dx), None, None, None) == 0      # fx_rocket(_proj)
            assert emu_lib.cdll.od_rocket_host(hd, project, _p(x), _p(u), None, None, _p(du), None, None) == 0      # fu_rocket(_proj)
            assert np.array_equal(y, Yb[:, b].cpu().numpy()) and np.array_equal(dx, DXb[:, :, b].cpu().numpy()) and np.array_equal(du, DUb[:, :, b].cpu().numpy())
    up = np.zeros(3); dp = np.zeros((3, 3), order="F")
    u = np.ascontiguousarray(Ur[:, 0])
    assert emu_lib.cdll.od_soc_project_host(hd, _p(u), _p(up), None, None) == 0
    up2 = up.copy()
    assert emu_lib.cdll.od_soc_project_host(hd, _p(u), _p(up), _p(dp), None) == 0
    assert np.array_equal(up, up2)
    s, z, dzo, it = oracle.soc_projection(12.5, u, True)
    assert np.abs(up - z[:3]).max() < 2e-4 * max(1, np.abs(z[:3]).max())
    assert np.hypot(up[0], up[1]) <= up[2] + 2e-2            # examples/rocket.jl:151
    assert emu_lib.cdll.od_destroy(hd) == 0



def _communicator_sequence(lib, device):
    """Communicator / allgather_compact! of the shim: od_version check of __init__, od_comm_unique_id (128 bytes), od_comm_create with the
    handle, od_comm_info with a NULL device pointer, od_set_layout(1) + od_rollout_compact (od_rollout_compact!: Julia n x K matrices are
    batch-major), od_allgather_compact, od_comm_allgather with a byte count, od_comm_destroy -- one rank; block 0 of the gathered arrays
    is the rollout's output bit for bit (the layout is the handle's: the gather ships the arrays as they stand)"""
    import bench
    import parity_checks as P
    from optimization_dynamics_amd import _lib
    assert lib.cdll.od_version() == _lib.ABI_VERSION == 101
    src = open(os.path.join(ROOT, "julia", "OptimizationDynamicsMI355X.jl")).read()
    assert "const ABI_VERSION = %d" % _lib.ABI_VERSION in src and "zeros(UInt8, %d)" % _lib.COMM_ID_BYTES in src
    B, T = 6, 5
    im = P.make_im("hopper", lib, device)
    uid = (C.c_ubyte * 128)()
    assert lib.cdll.od_comm_unique_id(uid) == 0
    hd = C.c_void_p()
    assert lib.cdll.od_comm_create(im._h, uid, 0, 1, C.byref(hd)) == 0
    w, r = C.c_int(-1), C.c_int(-1)
    assert lib.cdll.od_comm_info(hd, C.byref(w), C.byref(r), None) == 0 and (w.value, r.value) == (1, 0)
    x1, U = bench.workload_slice(0, B, B, T)
    dev = torch.device(device)
    x1j = torch.tensor(np.ascontiguousarray(x1.T), device=dev)                      # Julia 8 x B column-major == B x 8 row-major
    Uj = torch.tensor(np.ascontiguousarray(U.reshape(2, T * B).T), device=dev)
    X = torch.zeros((T + 1) * B, 8, dtype=torch.float64, device=dev); G = torch.zeros(T * B, 40, dtype=torch.float64, device=dev)
    Xa, Ga = torch.zeros_like(X), torch.zeros_like(G)
    assert lib.cdll.od_set_layout(im._h, 1) == 0
    assert lib.cdll.od_rollout_compact(im._h, B, T, x1j.data_ptr(), Uj.data_ptr(), X.data_ptr(), G.data_ptr(), None, None) == 0
    assert lib.cdll.od_allgather_compact(im._h, hd, B, T, X.data_ptr(), G.data_ptr(), Xa.data_ptr(), Ga.data_ptr()) == 0
    k = torch.arange(11, dtype=torch.float64, device=dev); ka = torch.zeros_like(k)
    assert lib.cdll.od_comm_allgather(im._h, hd, k.data_ptr(), ka.data_ptr(), C.c_size_t(88)) == 0
    assert lib.cdll.od_synchronize(im._h) == 0
    assert torch.equal(Xa, X) and torch.equal(Ga, G) and torch.equal(ka, k) and X.abs().sum().item() > 0
    # the same numbers as the batch-minor rollout of the Python mirror
    assert lib.cdll.od_set_layout(im._h, 0) == 0
    Xm, Gm, st, it, out = im.rollout_compact(torch.tensor(x1, device=dev), torch.tensor(U, device=dev))
    assert torch.equal(X.T.reshape(8, T + 1, B), Xm)
    assert lib.cdll.od_comm_destroy(hd) == 0


def test_communicator_sequence(emu_lib):
    import glob
    try:
        _communicator_sequence(emu_lib, "cpu")
    finally:
        for f in glob.glob("/dev/shm/odemu_%d_*" % os.getpid()):
            os.remove(f)


@pytest.mark.gpu
def test_communicator_sequence_gpu(gpu_lib):
    buf = (C.c_ubyte * 128)()
    if gpu_lib.cdll.od_comm_unique_id(buf) == -2:
        pytest.skip("librccl not loadable on this box: " + gpu_lib.cdll.od_last_error().decode())
    _communicator_sequence(gpu_lib, "cuda:0")


# the three call sequences on the host build of the product sources (CPU tier) and on the shipped HIP library (-m gpu): the shim's
# *_host entry points are the boundary of BASELINE config 1
def test_implicit_dynamics_callbacks_sequence(oracle, emu_lib):
    _implicit_dynamics_callbacks_sequence(oracle, emu_lib, "cpu")


def test_gradient_bundle_sequence(oracle, emu_lib):
    _gradient_bundle_sequence(oracle, emu_lib, "cpu")


def test_rocket_sequence(oracle, emu_lib):
    _rocket_sequence(oracle, emu_lib, "cpu")


@pytest.mark.gpu
def test_implicit_dynamics_callbacks_sequence_gpu(oracle, gpu_lib):
    _implicit_dynamics_callbacks_sequence(oracle, gpu_lib, "cuda:0")


@pytest.mark.gpu
def test_gradient_bundle_sequence_gpu(oracle, gpu_lib):
    _gradient_bundle_sequence(oracle, gpu_lib, "cuda:0")


@pytest.mark.gpu
def test_rocket_sequence_gpu(oracle, gpu_lib):
    _rocket_sequence(oracle, gpu_lib, "cuda:0")


def test_rocket_host_entry_points_on_a_single_precision_handle(emu_lib):
    """od_rocket_host / od_soc_project_host take and return host DOUBLES whatever the handle computes in: on an OD_F32 handle
    the values are converted on the way in and out (they used to be reinterpreted: garbage with rc = 0)"""
    from optimization_dynamics_amd import models, rocket as rk
    hd = C.c_void_p()
    assert emu_lib.cdll.od_create(5, 1, None, C.c_double(0.05), C.byref(hd)) == 0        # OD_F32
    assert emu_lib.cdll.od_set_u_max(hd, C.c_double(12.5)) == 0
    from optimization_dynamics_amd._lib import Options
    oo = Options()
    assert emu_lib.cdll.od_get_options(hd, C.byref(oo)) == 0
    oo.r_tol = 1e-4                                                                      # tolerances reachable in float
    assert emu_lib.cdll.od_set_options(hd, C.byref(oo)) == 0
    Xr, Ur = W.rocket_inputs(3, seed=1)
    info64 = rk.RocketInfo(models.rocket, 12.5, 0.05, device="cpu", lib=emu_lib)
    for project in (0, 1):
        Y, DX, DU, UP, st = info64.solve(torch.tensor(Xr), torch.tensor(Ur), project=bool(project), grads=True)
        for b in range(3):
            x, u = np.ascontiguousarray(Xr[:, b]), np.ascontiguousarray(Ur[:, b])
            y = np.zeros(12); dx = np.zeros((12, 12), order="F"); du = np.zeros((12, 3), order="F"); up = np.zeros(3)
            stat = C.c_int(0)
            assert emu_lib.cdll.od_rocket_host(hd, project, _p(x), _p(u), _p(y), _p(dx), _p(du), _p(up), C.byref(stat)) == 0
            assert np.isfinite(y).all() and np.abs(y - Y[:, b].numpy()).max() < 5e-4 * max(1.0, np.abs(Y[:, b].numpy()).max())
            assert np.abs(dx - DX[:, :, b].numpy()).max() < 2e-2 * np.abs(DX[:, :, b].numpy()).max()
            if project:
                assert np.abs(up - UP[:, b].numpy()).max() < 5e-3 * max(1.0, np.abs(UP[:, b].numpy()).max())
    up = np.zeros(3); dp = np.zeros((3, 3), order="F")
    u = np.ascontiguousarray(Ur[:, 0])
    assert emu_lib.cdll.od_soc_project_host(hd, _p(u), _p(up), _p(dp), None) == 0
    assert np.isfinite(up).all() and np.isfinite(dp).all() and np.hypot(up[0], up[1]) <= up[2] + 2e-2
    assert emu_lib.cdll.od_destroy(hd) == 0


def test_grad_iterates_need_the_knot_count_of_the_last_pass(emu_lib):
    im = P.make_im("hopper", emu_lib, "cpu")
    X, U = W.knots("hopper", 12, seed=3)
    im.step_grad(torch.tensor(X), torch.tensor(U))
    assert im.grad_iterates(12).shape == (21, 12)
    buf = torch.empty(21 * 8, dtype=torch.float64)
    assert emu_lib.cdll.od_get_grad_iterates(im._h, 8, buf.data_ptr()) == -1          # another K: wrong stride, refused
    assert b"last gradient pass" in emu_lib.cdll.od_last_error()
