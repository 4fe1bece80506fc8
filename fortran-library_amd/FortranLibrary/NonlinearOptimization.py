"""Batched line-search optimisers on MI355X -- ctypes over libFL.so's C ABI.

Names follow the reference module ``NonlinearOptimization``
(/root/reference/source/NonlinearOptimization.f90): LBFGS (398), ConjugateGradient
(193), SteepestDescent (55); keyword names and defaults are the reference's optional
arguments (NO.f90:41-51).  Inputs are torch CUDA tensors (fp64, problem-major
[batch, n]); x is updated in place like the reference's ``x`` (intent inout).
torch is only the carrier of device memory and the stream -- all arithmetic runs in
the HIP kernels.
"""
import ctypes as C

from .basic import FL

QUARTIC, ROSENBROCK, DIAGQUAD = 0, 1, 2
CONVERGED, STEP_CONVERGED, MAXIT = 0, 1, 2
SD, CG, LBFGS_, BFGS_ = 0, 1, 2, 3
OK = 0


class Options(C.Structure):
    """struct fl_options (include/fl_nlopt.h)."""
    _fields_ = [("strong", C.c_int32), ("max_iteration", C.c_int32), ("precision", C.c_double),
                ("min_step_length", C.c_double), ("wolfe_c1", C.c_double), ("wolfe_c2", C.c_double),
                ("increment", C.c_double), ("memory", C.c_int32), ("cg_method", C.c_int32),
                ("fused_f_fd", C.c_int32), ("clamp", C.c_int32), ("exact_step", C.c_int32)]


_vp, _dp, _ip = C.c_void_p, C.c_void_p, C.c_void_p
FL.fl_version.restype = C.c_int
FL.fl_default_options.argtypes = [C.POINTER(Options), C.c_int]
FL.fl_default_options.restype = None
FL.fl_reduction_geometry.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
FL.fl_reduction_geometry_for.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
FL.fl_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
FL.fl_workspace_bytes.restype = C.c_size_t
FL.fl_lbfgs_batched.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.POINTER(Options), _vp, C.c_size_t, _dp,
                                _dp, _ip, _ip, _ip, _ip, _vp]
_cg_sd = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.POINTER(Options), _dp, _dp, _ip, _ip, _ip, _ip, _vp]
FL.fl_conjugate_gradient_batched.argtypes = _cg_sd
FL.fl_steepest_descent_batched.argtypes = _cg_sd
FL.fl_bfgs_batched.argtypes = FL.fl_lbfgs_batched.argtypes
FL.fl_augmented_lagrangian_batched.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp,
                                               C.c_double, C.POINTER(Options), _vp, C.c_size_t, _dp, _dp, _ip, _ip,
                                               _ip, _ip, _ip, _vp]
FL.fl_lbfgs_two_loop_batched.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _vp]
FL.fl_synth_uniform.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, _dp, _vp]
FL.fl_synth_diag_spectrum.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, _dp, _vp]


class FLError(RuntimeError):
    pass


def _check(rc, what):
    if rc != OK:
        names = {-1: "FL_ERR_INVALID_ARGUMENT", -2: "FL_ERR_UNSUPPORTED_SIZE", -3: "FL_ERR_WORKSPACE",
                 -4: "FL_ERR_NO_DEVICE", -5: "FL_ERR_LAUNCH"}
        raise FLError(f"{what} failed: {names.get(rc, rc)}")


def default_options(solver, **kw):
    """Reference defaults (NO.f90:73-86, 419-434), overridden by reference-named keywords:
    Strong, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment, Memory,
    Method ('DY'|'PR'), f_fd (bool: act as if f_fd was passed), clamp."""
    o = Options()
    FL.fl_default_options(C.byref(o), solver)
    ren = {"Strong": "strong", "MaxIteration": "max_iteration", "Precision": "precision",
           "MinStepLength": "min_step_length", "WolfeConst1": "wolfe_c1", "WolfeConst2": "wolfe_c2",
           "Increment": "increment", "Memory": "memory", "f_fd": "fused_f_fd", "clamp": "clamp",
           "ExactStep": "exact_step"}
    for k, v in kw.items():
        if v is None:
            continue
        if k == "Method":
            if v not in ("DY", "PR"):  # reference: "Program abort: unsupported conjugate gradient method" (NO.f90:345)
                raise ValueError("unsupported conjugate gradient method " + str(v))
            o.cg_method = 0 if v == "DY" else 1
        elif k in ren:
            setattr(o, ren[k], int(v) if isinstance(v, bool) else v)
        else:
            raise TypeError("unknown option " + k)
    return o


def reduction_geometry(n, solver=None):
    """(threads, elements per thread) of the kernels' fixed summation order for dimension n; with `solver` (SD | CG |
    LBFGS_ | BFGS_ | 4): of that solver's fused kernel (fl_reduction_geometry_for)"""
    t, e = C.c_int(), C.c_int()
    if solver is None:
        _check(FL.fl_reduction_geometry(n, C.byref(t), C.byref(e)), "fl_reduction_geometry")
    else:
        _check(FL.fl_reduction_geometry_for(int(solver), n, C.byref(t), C.byref(e)), "fl_reduction_geometry_for")
    return t.value, e.value


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _prep(x, d, b):
    import torch
    if not (x.is_cuda and x.dtype == torch.float64 and x.dim() == 2 and x.is_contiguous()):
        raise ValueError("x must be a contiguous CUDA float64 tensor [batch, n]")
    for t in (d, b):
        if t is not None and not (t.is_cuda and t.dtype == torch.float64 and t.shape == x.shape and t.is_contiguous()):
            raise ValueError("objective data must be contiguous CUDA float64 tensors shaped like x")
    B, n = x.shape
    out = dict(f=torch.empty(B, dtype=torch.float64, device=x.device),
               gg=torch.empty(B, dtype=torch.float64, device=x.device),
               iters=torch.empty(B, dtype=torch.int32, device=x.device),
               status=torch.empty(B, dtype=torch.int32, device=x.device),
               nf=torch.empty(B, dtype=torch.int32, device=x.device),
               ng=torch.empty(B, dtype=torch.int32, device=x.device))
    return B, n, out


def workspace(batch, n, memory, device):
    import torch
    nbytes = FL.fl_workspace_bytes(LBFGS_, batch, n, memory)
    return torch.empty(max(nbytes // 8, 1), dtype=torch.float64, device=device)


def LBFGS(objective, x, d=None, b=None, workspace_=None, options=None, **kw):
    """Batched L-BFGS (reference: subroutine LBFGS, NO.f90:398-625). Returns dict of per-problem outputs."""
    o = options if options is not None else default_options(LBFGS_, **kw)
    B, n, out = _prep(x, d, b)
    ws = workspace_ if workspace_ is not None else workspace(B, n, o.memory, x.device)
    _check(FL.fl_lbfgs_batched(objective, B, n, _ptr(x), _ptr(d), _ptr(b), C.byref(o), _ptr(ws),
                               ws.numel() * 8, _ptr(out["f"]), _ptr(out["gg"]), _ptr(out["iters"]),
                               _ptr(out["status"]), _ptr(out["nf"]), _ptr(out["ng"]), _stream()), "fl_lbfgs_batched")
    out["workspace"] = ws
    return out


def ConjugateGradient(objective, x, d=None, b=None, options=None, **kw):
    """Batched conjugate gradient, Method='DY'|'PR' (reference: NO.f90:193-394)."""
    o = options if options is not None else default_options(CG, **kw)
    B, n, out = _prep(x, d, b)
    _check(FL.fl_conjugate_gradient_batched(objective, B, n, _ptr(x), _ptr(d), _ptr(b), C.byref(o), _ptr(out["f"]),
                                            _ptr(out["gg"]), _ptr(out["iters"]), _ptr(out["status"]),
                                            _ptr(out["nf"]), _ptr(out["ng"]), _stream()),
           "fl_conjugate_gradient_batched")
    return out


def SteepestDescent(objective, x, d=None, b=None, options=None, **kw):
    """Batched steepest descent (reference: NO.f90:55-188)."""
    o = options if options is not None else default_options(SD, **kw)
    B, n, out = _prep(x, d, b)
    _check(FL.fl_steepest_descent_batched(objective, B, n, _ptr(x), _ptr(d), _ptr(b), C.byref(o), _ptr(out["f"]),
                                          _ptr(out["gg"]), _ptr(out["iters"]), _ptr(out["status"]), _ptr(out["nf"]),
                                          _ptr(out["ng"]), _stream()), "fl_steepest_descent_batched")
    return out


FL.fl_workspace_bytes_for.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Options)]
FL.fl_workspace_bytes_for.restype = C.c_size_t
FL.fl_newton_raphson_batched.argtypes = FL.fl_lbfgs_batched.argtypes
FL.fl_dposv_batched.argtypes = [C.c_int, C.c_int, _dp, _dp, _ip, _vp]
FL.fl_dsysv_batched.argtypes = [C.c_int, C.c_int, _dp, _dp, _ip, _vp]
FL.fl_dpotri_batched.argtypes = [C.c_int, C.c_int, _dp, _dp, _ip, _vp]
NEWTON_ = 4


def bfgs_workspace(batch, n, device, options=None, solver=BFGS_):
    import torch
    o = options if options is not None else default_options(solver, ExactStep=0)
    nbytes = FL.fl_workspace_bytes_for(solver, batch, n, C.byref(o))
    return torch.empty(max(nbytes // 8, 1), dtype=torch.float64, device=device)


def NewtonRaphson(objective, x, d=None, b=None, workspace_=None, options=None, **kw):
    """Batched Newton-Raphson with the objective's analytic Hessian (reference: NO.f90:1026-1271, fdd present)."""
    o = options if options is not None else default_options(NEWTON_, **kw)
    B, n, out = _prep(x, d, b)
    ws = workspace_ if workspace_ is not None else bfgs_workspace(B, n, x.device, o, NEWTON_)
    _check(FL.fl_newton_raphson_batched(objective, B, n, _ptr(x), _ptr(d), _ptr(b), C.byref(o), _ptr(ws),
                                        ws.numel() * 8, _ptr(out["f"]), _ptr(out["gg"]), _ptr(out["iters"]),
                                        _ptr(out["status"]), _ptr(out["nf"]), _ptr(out["ng"]), _stream()),
           "fl_newton_raphson_batched")
    out["workspace"] = ws
    return out


def dposv(A, b):
    """My_dposv for a batch: A [batch, n, ld] column-major SPD (overwritten by its Cholesky factor),
    b [batch, n] -> solution.  Returns info [batch] (0 = solved)."""
    import torch
    B, n = b.shape
    info = torch.empty(B, dtype=torch.int32, device=b.device)
    _check(FL.fl_dposv_batched(B, n, _ptr(A), _ptr(b), _ptr(info), _stream()), "fl_dposv_batched")
    return info


def dsysv(A, b):
    """My_dsysv for a batch: A [batch, n, ld] column-major symmetric INDEFINITE, lower triangle referenced
    (destroyed), b [batch, n] -> solution.  Returns info [batch] (0 = solved)."""
    import torch
    B, n = b.shape
    info = torch.empty(B, dtype=torch.int32, device=b.device)
    _check(FL.fl_dsysv_batched(B, n, _ptr(A), _ptr(b), _ptr(info), _stream()), "fl_dsysv_batched")
    return info


def dpotri(A):
    """My_dpotri + syL2U for a batch: A [batch, n, ld] column-major SPD -> A^{-1} (both triangles)."""
    import torch
    B, n = A.shape[0], A.shape[1]
    work = torch.empty_like(A)
    info = torch.empty(B, dtype=torch.int32, device=A.device)
    _check(FL.fl_dpotri_batched(B, n, _ptr(A), _ptr(work), _ptr(info), _stream()), "fl_dpotri_batched")
    return info


def BFGS(objective, x, d=None, b=None, workspace_=None, options=None, **kw):
    """Batched dense BFGS (subroutine BFGS, NO.f90:632-1022).  ExactStep (default 20 like the reference):
    every ExactStep iterations the exact inverse Hessian of the built-in objective replaces the
    quasi-Newton matrix (the reference's fdd branch); ExactStep <= 0: rank-2 updates only."""
    o = options if options is not None else default_options(BFGS_, **kw)
    B, n, out = _prep(x, d, b)
    ws = workspace_ if workspace_ is not None else bfgs_workspace(B, n, x.device, o)
    _check(FL.fl_bfgs_batched(objective, B, n, _ptr(x), _ptr(d), _ptr(b), C.byref(o), _ptr(ws), ws.numel() * 8,
                              _ptr(out["f"]), _ptr(out["gg"]), _ptr(out["iters"]), _ptr(out["status"]),
                              _ptr(out["nf"]), _ptr(out["ng"]), _stream()), "fl_bfgs_batched")
    out["workspace"] = ws
    return out


def AugmentedLagrangian(objective, x, M, d=None, b=None, UnconstrainedSolver="LBFGS", lambda0=None, miu0=1.0,
                        workspace_=None, options=None, **kw):
    """Batched augmented Lagrangian with M block-sphere equality constraints (reference: subroutine
    AugmentedLagrangian, NO.f90:2005-2241).  UnconstrainedSolver: 'LBFGS' | 'ConjugateGradient' | 'BFGS' | 'NewtonRaphson'
    (the last two with the analytic Hessian of L, n <= 4096; BFGS with ExactStep=0: quasi-Newton only).  lambda0: optional [batch, M] tensor, updated in place (returned as
    out['lambda'])."""
    import torch
    solvers = {"LBFGS": LBFGS_, "ConjugateGradient": CG, "BFGS": BFGS_, "NewtonRaphson": 4}
    if UnconstrainedSolver not in solvers:  # reference: "Program abort: unsupported unconstrained solver" (NO.f90:2186)
        raise ValueError("unsupported unconstrained solver on the device path: " + str(UnconstrainedSolver))
    solver = solvers[UnconstrainedSolver]
    o = options if options is not None else default_options(solver, **kw)
    B, n, out = _prep(x, d, b)
    lam = lambda0 if lambda0 is not None else torch.zeros(B, M, dtype=torch.float64, device=x.device)
    ws = workspace_
    if ws is None:
        if solver == LBFGS_:
            ws = workspace(B, n, o.memory, x.device)
        elif solver == BFGS_ or solver == 4:
            ws = bfgs_workspace(B, n, x.device, o, solver=solver)
        else:
            ws = torch.empty(1, dtype=torch.float64, device=x.device)
    out["outer"] = torch.empty(B, dtype=torch.int32, device=x.device)
    out["cnorm2"] = torch.empty(B, dtype=torch.float64, device=x.device)
    _check(FL.fl_augmented_lagrangian_batched(solver, objective, B, n, M, _ptr(x), _ptr(d), _ptr(b), _ptr(lam),
                                              miu0, C.byref(o), _ptr(ws), ws.numel() * 8, _ptr(out["f"]),
                                              _ptr(out["cnorm2"]), _ptr(out["iters"]), _ptr(out["outer"]),
                                              _ptr(out["status"]), _ptr(out["nf"]), _ptr(out["ng"]), _stream()),
           "fl_augmented_lagrangian_batched")
    out["lambda"] = lam
    out["workspace"] = ws
    del out["gg"]
    return out


FL.fl_bfgs_update_gemm_workspace_bytes.argtypes = [C.c_int, C.c_int]
FL.fl_bfgs_update_gemm_workspace_bytes.restype = C.c_size_t
FL.fl_bfgs_update_gemm_batched.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _vp, C.c_size_t, _vp]


def bfgs_update_gemm(H, s, y, chunk=None, workspace_=None):
    """H <- U^T (H U) + rho s s^T exactly as the reference writes it (two dense products on f64 MFMA).
    H: [batch, n, ld] column-major inverse Hessians (solver layout, ld = threads*ept), in place."""
    import torch
    B, n = s.shape
    c = chunk if chunk else B
    ws = workspace_
    if ws is None:
        ws = torch.empty(FL.fl_bfgs_update_gemm_workspace_bytes(c, n) // 8, dtype=torch.float64, device=s.device)
    _check(FL.fl_bfgs_update_gemm_batched(B, n, _ptr(H), _ptr(s), _ptr(y), _ptr(ws), ws.numel() * 8, _stream()),
           "fl_bfgs_update_gemm_batched")
    return ws


FL.fl_rci_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(Options), _vp]
FL.fl_rci_step.argtypes = [_vp, _dp, _dp, _dp, _ip]
FL.fl_rci_results.argtypes = [_vp, _dp, _dp, _ip, _ip, _ip, _ip]
FL.fl_rci_destroy.argtypes = [_vp]


if hasattr(FL, "fl_rci_step_flags"):
    FL.fl_rci_step_flags.argtypes = [_vp, _dp, _dp, _dp, _ip, C.c_int]
    FL.fl_rci_step_compact.argtypes = [_vp, _dp, _ip, C.c_int, _dp, _dp, _dp, _ip, C.c_int]
RCI_BOTH = 1


def _f64c(t):
    import torch
    return t if (t.dtype == torch.float64 and t.is_contiguous()) else t.to(torch.float64).contiguous()


REQ_H = 16
if hasattr(FL, "fl_rci_put_hessians"):
    FL.fl_rci_put_hessians.argtypes = [_vp, _dp, _ip]
    FL.fl_rci_auglag_miu.argtypes = [_vp, _dp]
    FL.fl_fd_points.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _vp]
    FL.fl_fd_column.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _dp, _vp]


def numerical_hessian(grad, x, eps=1e-8):
    """f'' of a batch by central differences of `grad(x) -> g[batch, n]` with MKL djacobi's step rule (fl_fd_points /
    fl_fd_column): what the reference does when no fdd is passed (NO.f90:676, 981, 1067).  2n gradient evaluations of
    the batch; returns H [batch, n, n] (H[k, j, :] = column j)."""
    import torch
    B, n = x.shape
    H = torch.empty(B, n, n, dtype=torch.float64, device=x.device)
    xp, xm = torch.empty_like(x), torch.empty_like(x)
    for j in range(n):
        _check(FL.fl_fd_points(B, n, j, eps, _ptr(x), _ptr(xp), _ptr(xm), _stream()), "fl_fd_points")
        gp, gm = _f64c(grad(xp)), _f64c(grad(xm))
        _check(FL.fl_fd_column(B, n, j, eps, _ptr(x), _ptr(gp), _ptr(gm), _ptr(H), _stream()), "fl_fd_column")
    return H


def minimize_rci(solver, x, fun, options=None, max_steps=10000000, check_every=8, mode="legacy", compact_below=0.5, hess=None, **kw):
    """Batched minimisation of a user objective by reverse communication (fl_rci_*); x [batch, n] is updated in place.
    The solver machines run in the HIP kernels; only the objective is the caller's.  solver: SD | CG | LBFGS_ | BFGS_.

    mode="legacy": `fun(x)` returns (f[batch], g[batch, n]) for the whole batch on every step, and every request of the
        reference's callback pattern -- f alone, then f' at the same point -- is a step of its own (fl_rci_step).
    mode="full":   `fun(x, request)` as above, but f AND g are taken at every point (FL_RCI_BOTH): one step per distinct
        trial point, like the fused kernels.  Finished problems cost one word per step on the device; the objective still
        sees the whole batch (request[k] == 0 marks the rows it may skip).
    mode="graph":  as "full", with one round -- the objective and the step kernel -- captured ONCE in a HIP graph and replayed: the
        ~10 launches of a round cost one graph launch (small and medium batches are launch-bound: tools/bench_configs.py rcigraph).
        `fun(x, request)` must be stream-capturable torch code (no host synchronisation, fixed shapes); `check_every` rounds
        run between two looks at the requests.  n <= 4096, no Hessians.
    mode="compact" (SD | CG | LBFGS_): `fun(xc, request, ids, epoch)` gets only the ACTIVE problems: xc [n_active, n], their
        problem ids [n_active] and a counter that changes whenever the list does (so that per-problem data can be
        gathered once per change, not per call); returns (f[n_active], g[n_active, n]).  The list is re-compacted when
        fewer than `compact_below` of its problems are still running: the tail of the slowest problems costs
        evaluations of those problems only.
    Hessians (solver 4 = NewtonRaphson, or BFGS_ with ExactStep > 0; mode "legacy"): `hess(x) -> f''[batch, n, n]` is
    called where a problem asks for it (the reference's fdd); hess="numerical": central differences of fun's gradient with
    djacobi's step rule, like the reference without fdd.
    The request vector is brought to the host only every `check_every` steps."""
    import torch
    o = options if options is not None else default_options(solver, **kw)
    B, n, out = _prep(x, None, None)
    if solver == BFGS_ and hess is None:
        o.exact_step = 0  # no Hessian source: quasi-Newton updates only (the reference would differentiate f' numerically: hess="numerical")
    wants_h = solver == 4 or (solver == BFGS_ and o.exact_step > 0)
    if wants_h and (hess is None or mode != "legacy"):
        raise ValueError("NewtonRaphson / BFGS with ExactStep > 0 by reverse communication: pass hess=callable | 'numerical' (mode 'legacy')")
    h = C.c_void_p()
    side = None
    if mode == "graph":  # the handle lives on a stream of its own: the one the step + objective sequence is captured on
        if wants_h or n > 4096:
            raise ValueError("mode 'graph': SD / CG / L-BFGS / quasi-Newton BFGS, n <= 4096")
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream())
        _check(FL.fl_rci_create(C.byref(h), solver, B, n, C.byref(o), C.c_void_p(side.cuda_stream)), "fl_rci_create")
    else:
        _check(FL.fl_rci_create(C.byref(h), solver, B, n, C.byref(o), _stream()), "fl_rci_create")
    try:
        req = torch.empty(B, dtype=torch.int32, device=x.device)
        steps = 0
        if mode == "graph":
            # One evaluation round -- the caller's objective (torch operations on fixed tensors) and the step kernel -- captured
            # ONCE in a HIP graph and replayed: the ~10 launches of a round cost one graph launch.  `fun` must be capturable
            # (no host synchronisation, no data-dependent shapes); every point asked for is evaluated for f AND f' (FL_RCI_BOTH).
            fbuf = torch.empty(B, dtype=torch.float64, device=x.device)
            gbuf = torch.empty(B, n, dtype=torch.float64, device=x.device)
            with torch.cuda.stream(side):
                _check(FL.fl_rci_step_flags(h, _ptr(x), None, None, _ptr(req), RCI_BOTH), "fl_rci_step_flags")
                for _ in range(2):  # warm-up rounds outside the capture (allocator, lazy initialisation), real steps of the solve
                    fn, gn = fun(x, req)
                    fbuf.copy_(fn)
                    gbuf.copy_(gn)
                    _check(FL.fl_rci_step_flags(h, _ptr(x), _ptr(fbuf), _ptr(gbuf), _ptr(req), RCI_BOTH), "fl_rci_step_flags")
                    steps += 1
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            rcs = []
            with torch.cuda.graph(graph, stream=side):
                fn, gn = fun(x, req)
                fbuf.copy_(fn)
                gbuf.copy_(gn)
                rcs.append(FL.fl_rci_step_flags(h, _ptr(x), _ptr(fbuf), _ptr(gbuf), _ptr(req), RCI_BOTH))
            _check(rcs[0], "fl_rci_step_flags (captured)")
            while steps < max_steps:
                if not bool((req != 0).any()):
                    break
                for _ in range(check_every):
                    graph.replay()
                steps += check_every
            torch.cuda.current_stream().wait_stream(side)
        elif mode == "compact":
            ids = torch.arange(B, dtype=torch.int32, device=x.device)
            xc = x.clone()
            na, epoch = B, 0
            _check(FL.fl_rci_step_compact(h, _ptr(x), _ptr(ids), na, _ptr(xc), None, None, _ptr(req), RCI_BOTH), "fl_rci_step_compact")
            while steps < max_steps:
                if steps % check_every == 0:
                    running = int((req[:na] != 0).sum())
                    if running == 0:
                        break
                    if running < compact_below * na:  # drop the finished problems from the list, moving their rows along
                        keep = torch.nonzero(req[:na]).flatten()
                        ids[:running] = ids[:na][keep]
                        xc[:running] = xc[:na][keep]
                        req[:running] = req[:na][keep]
                        na, epoch = running, epoch + 1
                fn, gn = fun(xc[:na], req[:na], ids[:na], epoch)
                _check(FL.fl_rci_step_compact(h, _ptr(x), _ptr(ids), na, _ptr(xc), _ptr(_f64c(fn)), _ptr(_f64c(gn)), _ptr(req),
                                              RCI_BOTH), "fl_rci_step_compact")
                steps += 1
        else:
            if mode == "full":
                def step(fp, gp):
                    return FL.fl_rci_step_flags(h, _ptr(x), fp, gp, _ptr(req), RCI_BOTH)
            else:
                def step(fp, gp):
                    return FL.fl_rci_step(h, _ptr(x), fp, gp, _ptr(req))
            _check(step(None, None), "fl_rci_step")
            while steps < max_steps:
                if steps % check_every == 0 and not bool((req != 0).any()):
                    break
                if wants_h and bool((req & REQ_H).any()):  # f''(x) for the problems that ask (their x is unchanged)
                    Hn = numerical_hessian(lambda xx: fun(xx)[1], x) if hess == "numerical" else _f64c(hess(x))
                    _check(FL.fl_rci_put_hessians(h, _ptr(Hn), _ptr(req)), "fl_rci_put_hessians")
                fn, gn = fun(x, req) if mode == "full" else fun(x)
                _check(step(_ptr(_f64c(fn)), _ptr(_f64c(gn))), "fl_rci_step")
                steps += 1
        _check(FL.fl_rci_results(h, _ptr(out["f"]), _ptr(out["gg"]), _ptr(out["iters"]), _ptr(out["status"]),
                                 _ptr(out["nf"]), _ptr(out["ng"])), "fl_rci_results")
        torch.cuda.synchronize()
        out["steps"] = steps
        if hasattr(FL, "fl_rci_cooperative_groups"):
            FL.fl_rci_cooperative_groups.argtypes = [_vp]
            out["cooperative_groups"] = int(FL.fl_rci_cooperative_groups(h))
    finally:
        FL.fl_rci_destroy(h)
    return out


def auglag_hessian(fdd, cdd, cd, c, lam, miu):
    """The Hessian of the augmented Lagrangian exactly as the reference's Ldd forms it (NO.f90:2229-2241):
        Ldd = (f'' + sum_j c_j'' (miu c_j - lambda_j)) + cd cd^T        -- no miu on the last term, as written.
    fdd [B, n, n], cdd [B, M, n, n] (or None: linear constraints), cd [B, M, n], c, lam [B, M], miu [B].  Element-wise
    operations in the reference's order, so a caller whose fdd / cdd have the oracle's bits gets the oracle's Hessian."""
    v = miu[:, None] * c - lam
    H = fdd
    if cdd is not None:
        t = 0.0 + cdd[:, 0] * v[:, 0, None, None]
        for j in range(1, cdd.shape[1]):
            t = t + cdd[:, j] * v[:, j, None, None]
        H = H + t
    pp = 0.0 + cd[:, 0, :, None] * cd[:, 0, None, :]
    for j in range(1, cd.shape[1]):
        pp = pp + cd[:, j, :, None] * cd[:, j, None, :]
    return H + pp


def minimize_rci_auglag(solver, x, fun, M, lambda0=None, miu0=1.0, options=None, max_steps=10000000, check_every=8, hess=None, **kw):
    """AugmentedLagrangian (NO.f90:2005) for a batch with the caller's objective AND equality constraints, by reverse
    communication: `fun(x)` returns (f[batch], g[batch, n], c[batch, M], cd[batch, M, n]) CUDA tensors for the whole
    batch (cd[k, j] = grad c_j at x_k).  solver (the inner one): LBFGS_ | CG | BFGS_ | 4 (NewtonRaphson); any n for the first three
    (BFGS_ to 16384), n <= 4096 with Hessians.  NewtonRaphson, and
    BFGS_ with ExactStep > 0, take the Hessian of L (the reference's fdd / cdd branch, NO.f90:2074-2148): pass
    hess=callable `hess(x) -> (fdd[batch, n, n], cdd[batch, M, n, n] | None)` -- Ldd is assembled from them as the reference
    does (auglag_hessian) -- or hess="numerical": central differences of grad L with djacobi's step rule, like the reference
    when fdd / cdd are absent.  x is updated in place; returns the usual outputs plus "lambda" [batch, M], "cnorm2", "outer"."""
    import torch
    o = options if options is not None else default_options(solver, **kw)
    B, n, out = _prep(x, None, None)
    if solver == BFGS_ and hess is None:
        o.exact_step = 0  # no Hessian source: quasi-Newton updates only
    wants_h = solver == 4 or (solver == BFGS_ and o.exact_step > 0)
    if wants_h and hess is None:
        raise ValueError("NewtonRaphson inside: pass hess=callable | 'numerical'")
    lam = torch.zeros(B, M, dtype=torch.float64, device=x.device) if lambda0 is None else \
        lambda0.to(torch.float64).contiguous().clone()
    FL.fl_rci_create_auglag.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double,
                                        C.POINTER(Options), C.c_void_p]
    FL.fl_rci_step_auglag.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    FL.fl_rci_results_auglag.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    h = C.c_void_p()
    _check(FL.fl_rci_create_auglag(C.byref(h), solver, B, n, M, _ptr(lam), float(miu0), C.byref(o), _stream()),
           "fl_rci_create_auglag")
    try:
        req = torch.empty(B, dtype=torch.int32, device=x.device)
        _check(FL.fl_rci_step_auglag(h, _ptr(x), None, None, None, None, _ptr(req)), "fl_rci_step_auglag")
        steps = 0
        while steps < max_steps:
            if steps % check_every == 0 and not bool((req != 0).any()):
                break
            fn, gn, cn, cdn = (t.to(torch.float64).contiguous() for t in fun(x))
            if wants_h and bool((req & REQ_H).any()):  # the Hessian of L at the (unchanged) current points
                miu = torch.empty(B, dtype=torch.float64, device=x.device)
                _check(FL.fl_rci_auglag_miu(h, _ptr(miu)), "fl_rci_auglag_miu")
                if hess == "numerical":
                    def grad_l(xx):  # grad L = f' + cd^T (miu c - lambda), the sum over the constraints in order (NO.f90:2205)
                        _, g2, c2, cd2 = fun(xx)
                        v = miu[:, None] * c2 - lam
                        t = 0.0 + cd2[:, 0] * v[:, 0, None]
                        for j in range(1, M):
                            t = t + cd2[:, j] * v[:, j, None]
                        return g2 + t
                    Hn = numerical_hessian(grad_l, x)
                else:
                    fdd, cdd = hess(x)
                    Hn = _f64c(auglag_hessian(fdd, cdd, cdn, cn, lam, miu))
                _check(FL.fl_rci_put_hessians(h, _ptr(Hn), _ptr(req)), "fl_rci_put_hessians")
            _check(FL.fl_rci_step_auglag(h, _ptr(x), _ptr(fn), _ptr(gn), _ptr(cn), _ptr(cdn), _ptr(req)),
                   "fl_rci_step_auglag")
            steps += 1
        _check(FL.fl_rci_results(h, _ptr(out["f"]), _ptr(out["gg"]), _ptr(out["iters"]), _ptr(out["status"]),
                                 _ptr(out["nf"]), _ptr(out["ng"])), "fl_rci_results")
        out["cnorm2"] = torch.empty(B, dtype=torch.float64, device=x.device)
        out["outer"] = torch.empty(B, dtype=torch.int32, device=x.device)
        _check(FL.fl_rci_results_auglag(h, _ptr(out["cnorm2"]), _ptr(out["outer"])), "fl_rci_results_auglag")
        torch.cuda.synchronize()
        out["steps"] = steps
        out["lambda"] = lam
    finally:
        FL.fl_rci_destroy(h)
    return out


def TrustRegion(x, fun, M, low=None, up=None, MaxIteration=1000, MaxStepIteration=100, Precision=1e-15,
                MinStepLength=1e-15, max_steps=100000, check_every=4):
    """TrustRegion (NO.f90:1728) for a batch on the device: `fun(x, request)` returns (r [batch, M], J [batch, N, M])
    CUDA tensors -- residuals f'(x) and the column-major M x N Jacobians -- for the whole batch (it may skip problems
    whose request bits do not ask).  x [batch, N] is updated in place; returns resnorm, iters, reason."""
    import torch
    B, N = x.shape
    FL.fl_trust_region_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_int, C.c_double, C.c_double, C.c_void_p]
    FL.fl_trust_region_step.argtypes = [C.c_void_p] * 5
    FL.fl_trust_region_results.argtypes = [C.c_void_p] * 4
    FL.fl_trust_region_destroy.argtypes = [C.c_void_p]
    h = C.c_void_p()
    _check(FL.fl_trust_region_create(C.byref(h), B, M, N, _ptr(low) if low is not None else None,
                                     _ptr(up) if up is not None else None, MaxIteration, MaxStepIteration, Precision,
                                     MinStepLength, _stream()), "fl_trust_region_create")
    try:
        req = torch.empty(B, dtype=torch.int32, device=x.device)
        _check(FL.fl_trust_region_step(h, _ptr(x), None, None, _ptr(req)), "fl_trust_region_step")
        steps = 0
        r = J = None
        while steps < max_steps:
            if steps % check_every == 0 and not bool((req != 0).any()):
                break
            r, J = fun(x, req)
            r, J = r.to(torch.float64).contiguous(), J.to(torch.float64).contiguous()
            _check(FL.fl_trust_region_step(h, _ptr(x), _ptr(r), _ptr(J), _ptr(req)), "fl_trust_region_step")
            steps += 1
        out = {"resnorm": torch.empty(B, dtype=torch.float64, device=x.device),
               "iters": torch.empty(B, dtype=torch.int32, device=x.device),
               "reason": torch.empty(B, dtype=torch.int32, device=x.device), "steps": steps}
        _check(FL.fl_trust_region_results(h, _ptr(out["resnorm"]), _ptr(out["iters"]), _ptr(out["reason"])),
               "fl_trust_region_results")
        torch.cuda.synchronize()
    finally:
        FL.fl_trust_region_destroy(h)
    return out


def dgemm(A, B, transA=False):
    """My_dgemm / My_dgemm_T on device tensors.  Column-major operands given as torch tensors of the TRANSPOSED shape
    (a row of the tensor = a column of the matrix): A [K, M] holds the M x K matrix (transA: A [M, K] holds the K x M
    one), B [N, K] holds K x N.  Returns C as a tensor [N, M] = the M x N product, column-major."""
    import torch
    FL.fl_dgemm.restype = C.c_int
    FL.fl_dgemm.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                            C.c_int, C.c_void_p]
    if transA:
        M, K = A.shape
        lda = K
    else:
        K, M = A.shape
        lda = M
    N = B.shape[0]
    assert B.shape[1] == K
    out = torch.empty(N, M, dtype=torch.float64, device=A.device)
    _check(FL.fl_dgemm(1 if transA else 0, M, K, N, _ptr(A), lda, _ptr(B), K, _ptr(out), M, _stream()), "fl_dgemm")
    return out


def dsyev(A, vectors=True):
    """My_dsyev on a device tensor: A [n, n] symmetric (only what is the LOWER triangle of the column-major matrix is read,
    i.e. A[j, i] for i >= j), overwritten.  Returns (w, V): the eigenvalues ascending and, with vectors, V [n, n] whose ROW k
    is the normalised eigenvector k (= column k of the column-major result; V is A itself) -- else (w, None).
    'V': fl_dsyev_vectors (tridiagonalisation + inverse iteration + Cholesky-QR + back-transformation), cyclic Jacobi when
    its device-side check of the basis fails; 'N': fl_dsyev_values.  n <= 6144."""
    import torch
    n = A.shape[0]
    assert A.shape == (n, n) and A.dtype == torch.float64 and A.is_contiguous()
    FL.fl_dsyev_vectors_workspace_bytes.restype = C.c_size_t
    FL.fl_dsyev_workspace_bytes.restype = C.c_size_t
    w = torch.empty(n, dtype=torch.float64, device=A.device)
    wsb = max(FL.fl_dsyev_vectors_workspace_bytes(n) if vectors else 0, FL.fl_dsyev_workspace_bytes(n))
    ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=A.device)
    if not vectors:
        _check(FL.fl_dsyev_values(C.c_int(n), C.c_void_p(A.data_ptr()), C.c_int(n), C.c_void_p(w.data_ptr()), C.c_void_p(ws.data_ptr()),
                                  C.c_size_t(wsb), _stream()), "fl_dsyev_values")
        return w, None
    keep = A.clone()
    rc = FL.fl_dsyev_vectors(C.c_int(n), C.c_void_p(A.data_ptr()), C.c_int(n), C.c_void_p(w.data_ptr()), C.c_void_p(ws.data_ptr()),
                             C.c_size_t(wsb), None, _stream())
    if rc == 1:  # the basis did not pass its check: Jacobi on the copy (unsorted: sort here)
        A.copy_(keep)
        sweeps = C.c_int(0)
        _check(FL.fl_dsyev_jacobi(C.c_char(b"V"), C.c_int(n), C.c_void_p(A.data_ptr()), C.c_int(n), C.c_void_p(w.data_ptr()),
                                  C.c_void_p(ws.data_ptr()), C.c_size_t(wsb), C.c_int(60), C.byref(sweeps), _stream()), "fl_dsyev_jacobi")
        V = ws[n * n: 2 * n * n].view(n, n)
        order = torch.argsort(w)
        A.copy_(V[order])
        return w[order].contiguous(), A
    _check(rc, "fl_dsyev_vectors")
    return w, A


def lbfgs_onchip_pairs(objective, n):
    """pairs of the (s, y) ring the fused L-BFGS kernel keeps in registers / LDS for this objective and dimension"""
    FL.fl_lbfgs_onchip_pairs.restype = C.c_int
    FL.fl_lbfgs_onchip_pairs.argtypes = [C.c_int, C.c_int]
    return int(FL.fl_lbfgs_onchip_pairs(int(objective), int(n)))


def two_loop(hist, rho, g, p, memory, recent):
    """Stand-alone batched two-loop recursion (Before(), NO.f90:586-608): p = -H g."""
    B, n = g.shape
    _check(FL.fl_lbfgs_two_loop_batched(B, n, memory, recent, _ptr(hist), _ptr(rho), _ptr(g), _ptr(p), _stream()),
           "fl_lbfgs_two_loop_batched")


def synth_uniform(seed, out, lo, hi):
    B, n = out.shape
    _check(FL.fl_synth_uniform(seed, B, n, lo, hi, _ptr(out), _stream()), "fl_synth_uniform")


def synth_diag_spectrum(seed, out, kappa_lo, kappa_hi):
    B, n = out.shape
    _check(FL.fl_synth_diag_spectrum(seed, B, n, kappa_lo, kappa_hi, _ptr(out), _stream()),
           "fl_synth_diag_spectrum")


def multi_solve(solver, objective, x, d=None, b=None, M=0, lambda0=None, miu0=1.0, nshards=0, interleaved=False, options=None, **kw):
    """A batch of independent problems over all the GPUs of the node from this one process (fl_multi_solve): numpy HOST
    arrays [batch, n] in, x updated in place, one host thread per shard.  M > 0: AugmentedLagrangian with M block-sphere
    constraints around `solver`.  nshards = 0: up to four shards per visible device (their transfers and solves overlap).  Returns the usual per-problem outputs."""
    import numpy as np
    o = options if options is not None else default_options(solver, **kw)
    if not (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.ndim == 2 and x.flags.c_contiguous):
        raise ValueError("x must be a C-contiguous float64 numpy array [batch, n]")
    B, n = x.shape
    dd = None if d is None else np.ascontiguousarray(d, dtype=np.float64)
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
    out = dict(f=np.zeros(B), gg=np.zeros(B), iters=np.zeros(B, np.int32), status=np.zeros(B, np.int32),
               nf=np.zeros(B, np.int32), ng=np.zeros(B, np.int32))
    lam = cn = outer = None
    if M:
        lam = np.zeros((B, M)) if lambda0 is None else np.ascontiguousarray(lambda0, dtype=np.float64).copy()
        cn, outer = np.zeros(B), np.zeros(B, np.int32)
        out.update({"lambda": lam, "cnorm2": cn, "outer": outer})

    def p(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None
    FL.fl_multi_solve.argtypes = [C.c_int] * 4 + [C.c_void_p] * 3 + [C.POINTER(Options), C.c_int, C.c_void_p, C.c_double] + \
        [C.c_void_p] * 8 + [C.c_int, C.c_int]
    _check(FL.fl_multi_solve(solver, objective, B, n, p(x), p(dd), p(bb), C.byref(o), M, p(lam), float(miu0), p(out["f"]),
                             p(out["gg"]), p(cn), p(out["iters"]), p(outer), p(out["status"]), p(out["nf"]), p(out["ng"]),
                             int(nshards), int(bool(interleaved))), "fl_multi_solve")
    return out


# ---- YOUR objective inside the fused kernels, compiled at run time (fl_user_compile: hiprtc) -------------------------------
FL.fl_user_compile.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
FL.fl_user_compile_check.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]
FL.fl_user_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
FL.fl_user_solve.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _vp, C.POINTER(Options), _vp, C.c_size_t, _dp, _dp, _ip, _ip, _ip, _ip, _vp]
FL.fl_user_destroy.argtypes = [C.c_void_p]
FL.fl_user_compile_auglag.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
FL.fl_user_solve_auglag.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _vp, _dp, C.c_double, C.POINTER(Options), _vp, C.c_size_t,
                                    _dp, _dp, _ip, _ip, _ip, _ip, _ip, _vp]
TUNE_NONE = 4


class CompiledObjective:
    """An objective given as HIP source text -- the functor of include/fl_user_objective.hpp:
        template <int NW, int EPT> struct Name { static constexpr int LDS_DOUBLES; init(A, prob, lds); eval(x, g, s0, s1, n, lds);
        static combine(s0, s1); } --
    compiled into the fused kernel of `solver` (SD | CG | LBFGS_ | BFGS_) for dimension n by hiprtc; .solve() then
    runs batches at the fused kernel's speed.  n > 4096 (vectors in HBM): class_name names a plain class with the STREAMING
    interface instead -- NEIGHBOURS, init(A, prob), pair(e, n, x, xa, xb, ta, tb, ua, ub, ga, gb), combine(s0, s1):
    include/fl_user_stream_objective.hpp, tests/user_sources.py STREAM_*.  tune_like: DIAGQUAD for an element-wise objective keeping at most two data
    vectors in registers, else TUNE_NONE.  The reference's interface for this is callbacks (NO.f90:33-38)."""

    def __init__(self, source, class_name, n, solver=LBFGS_, tune_like=TUNE_NONE, constrained=False, constraints_class=None):
        """constrained=True: inside the augmented Lagrangian; constraints_class: the caller's own constraints functor in the same
        source (partial / add_gradient, include/fl_nlopt.h), else the library's block spheres"""
        self.solver, self.n, self.constrained = int(solver), int(n), bool(constrained or constraints_class)
        h = C.c_void_p()
        log = C.create_string_buffer(1 << 16)
        if self.constrained:
            rc = FL.fl_user_compile_auglag(C.byref(h), source.encode(), class_name.encode(), (constraints_class or "").encode(), self.solver,
                                           self.n, int(tune_like), log, len(log))
        else:
            rc = FL.fl_user_compile(C.byref(h), source.encode(), class_name.encode(), self.solver, self.n, int(tune_like), log, len(log))
        self.log = log.value.decode(errors="replace")
        if rc != OK:
            raise FLError(f"fl_user_compile failed ({rc}):\n{self.log}")
        self._h = h
        t, e = C.c_int(), C.c_int()
        FL.fl_user_geometry(self._h, C.byref(t), C.byref(e))
        self.geometry = (t.value, e.value)

    def solve(self, x, data0=None, data1=None, params=None, workspace_=None, options=None, **kw):
        """x [batch, n] in/out; data0 / data1 [batch, n] and params (any CUDA tensor) reach the functor as A.d, A.b, A.user"""
        import torch
        B, n, out = _prep(x, data0, data1)
        if n != self.n:
            raise ValueError(f"compiled for n = {self.n}")
        o = options if options is not None else default_options(self.solver, **kw)
        ws = workspace_
        nbytes = FL.fl_workspace_bytes_for(self.solver, B, n, C.byref(o))
        if nbytes and (ws is None or ws.numel() * ws.element_size() < nbytes):
            ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=x.device)
        _check(FL.fl_user_solve(self._h, B, _ptr(x), _ptr(data0), _ptr(data1), _ptr(params), C.byref(o), _ptr(ws),
                                ws.numel() * ws.element_size() if ws is not None else 0, _ptr(out["f"]), _ptr(out["gg"]),
                                _ptr(out["iters"]), _ptr(out["status"]), _ptr(out["nf"]), _ptr(out["ng"]), _stream()), "fl_user_solve")
        out["workspace"] = ws
        return out

    def solve_auglag(self, x, M, data0=None, data1=None, params=None, lambda0=None, miu0=1.0, workspace_=None, options=None, **kw):
        """AugmentedLagrangian around the compiled objective with M block-sphere constraints (compiled with constrained=True)"""
        import torch
        B, n, out = _prep(x, data0, data1)
        o = options if options is not None else default_options(self.solver, **kw)
        lam = torch.zeros(B, M, dtype=torch.float64, device=x.device) if lambda0 is None else lambda0.clone()
        ws = workspace_
        nbytes = FL.fl_workspace_bytes_for(self.solver, B, n, C.byref(o))
        if nbytes and (ws is None or ws.numel() * ws.element_size() < nbytes):
            ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=x.device)
        out["cnorm2"] = torch.empty(B, dtype=torch.float64, device=x.device)
        out["outer"] = torch.empty(B, dtype=torch.int32, device=x.device)
        _check(FL.fl_user_solve_auglag(self._h, B, int(M), _ptr(x), _ptr(data0), _ptr(data1), _ptr(params), _ptr(lam), float(miu0), C.byref(o),
                                       _ptr(ws), ws.numel() * ws.element_size() if ws is not None else 0, _ptr(out["f"]), _ptr(out["cnorm2"]),
                                       _ptr(out["iters"]), _ptr(out["outer"]), _ptr(out["status"]), _ptr(out["nf"]), _ptr(out["ng"]), _stream()),
               "fl_user_solve_auglag")
        out["lambda"] = lam
        out["workspace"] = ws
        del out["gg"]
        return out

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and FL is not None:  # (at interpreter exit the module's globals may be gone already)
            FL.fl_user_destroy(h)
            self._h = None


def compile_objective(source, class_name, n, solver=LBFGS_, tune_like=TUNE_NONE, constrained=False, constraints_class=None):
    return CompiledObjective(source, class_name, n, solver, tune_like, constrained, constraints_class)


def compile_check(source, class_name, n, solver=LBFGS_, tune_like=TUNE_NONE, arch="gfx950"):
    """compile only (no GPU needed): (return code, compiler log)"""
    log = C.create_string_buffer(1 << 16)
    rc = FL.fl_user_compile_check(source.encode(), class_name.encode(), int(solver), int(n), int(tune_like), arch.encode(), log, len(log))
    return rc, log.value.decode(errors="replace")


FL.fl_augmented_lagrangian_launch_plan.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
FL.fl_cooperative_groups_for.argtypes = [C.c_int] * 4
FL.fl_bfgs_deferred_updates.argtypes = [C.c_int]


def bfgs_deferred_updates(n):
    """rank-2 updates the fused BFGS kernels of dimension n keep pending before folding them into H (0 for n <= 128): the update
    form of a bit-exact replay (oracle bfgs_form 100 + this, 1 if 0)"""
    return int(FL.fl_bfgs_deferred_updates(int(n)))



def cooperative_groups(solver, objective, batch, n):
    """workgroups that share one problem when SteepestDescent / ConjugateGradient / LBFGS run this batch on the current device
    (1: none; > 1 for few problems of n > 14336: the sums are then the oracle's tree order with `groups`; FL_COOP_GROUPS overrides)"""
    return int(FL.fl_cooperative_groups_for(int(solver), int(objective), int(batch), int(n)))


def augmented_lagrangian_launch_plan(solver, objective, batch, n, M):
    """[(waves per problem, hand over when at most this many problems are left; 0 = runs to the end), ...]: the launches
    fl_augmented_lagrangian_batched makes for this batch on the current device (results do not depend on it)"""
    w, p = (C.c_int * 3)(), (C.c_int * 3)()
    ns = FL.fl_augmented_lagrangian_launch_plan(int(solver), int(objective), int(batch), int(n), int(M), w, p, 3)
    if ns < 0:
        _check(ns, "fl_augmented_lagrangian_launch_plan")
    return [(w[k], p[k]) for k in range(ns)]
