"""ctypes binding of the CPU oracle (oracle/libfemshell_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libfemshell_oracle.so")

REF_Y21 = 1
REF_DRILL_MAX = 2
REF_DEFAULT = REF_Y21 | REF_DRILL_MAX


class Material(C.Structure):
    _fields_ = [("nu", C.c_double), ("E", C.c_double), ("thickness", C.c_double), ("flags", C.c_uint32)]


class Tri3Parts(C.Structure):
    _fields_ = [
        ("trafo", C.c_double * 9),
        ("transUV", C.c_double * 6),
        ("dphi", C.c_double * 6),
        ("area", C.c_double),
        ("Ke_m", C.c_double * 36),
        ("Ke_p", C.c_double * 81),
        ("K_local", C.c_double * 324),
        ("K_global_nm", C.c_double * 324),
    ]


class PcgInfo(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32),
        ("converged", C.c_int32),
        ("rel_residual", C.c_double),
        ("seconds", C.c_double),
    ]


def build():
    """(Re)build the oracle shared library if it is missing or stale."""
    src = os.path.join(ORACLE_DIR, "femshell_oracle.c")
    hdr = os.path.join(ORACLE_DIR, "femshell_oracle.h")
    if (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < max(
        os.path.getmtime(src), os.path.getmtime(hdr)
    ):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


FAST_LIB_PATH = os.path.join(ORACLE_DIR, "libfemshell_oracle_fast.so")
_lib = None
_use_fast = False


def use_fast_build(threads=1):
    """Switch this process to the CPU-baseline build of the same source (`make -C oracle fast`: -O3 -march=native
    -fopenmp, compiled on the machine that runs it) and set its thread count.  bench.py's cpu_baseline leg only."""
    global _lib, _use_fast
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "fast"])
    _use_fast = True
    _lib = None
    lib().fso_set_threads(int(threads))
    return lib().fso_threads()


def set_threads(n):
    lib().fso_set_threads(int(n))
    return lib().fso_threads()


def set_specht_polynomial(on):
    """True: rebuild the Specht curvatures from the polynomial derivation for every element (cross-check of the tables)."""
    lib().fso_set_specht_polynomial(1 if on else 0)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(FAST_LIB_PATH if _use_fast else LIB_PATH)
        L.fso_set_threads.argtypes = [C.c_int]
        L.fso_threads.restype = C.c_int
        L.fso_set_specht_polynomial.argtypes = [C.c_int]
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        bp = C.POINTER(C.c_uint8)
        L.fso_material_matrices.argtypes = [C.POINTER(Material), dp, dp]
        L.fso_tri3_specht_B.argtypes = [dp, C.c_double, C.c_double, dp, dp]
        L.fso_element_tri3.argtypes = [dp, C.POINTER(Material), dp, C.POINTER(Tri3Parts)]
        L.fso_element_tri3.restype = C.c_int
        L.fso_element_quad4.argtypes = [dp, C.POINTER(Material), dp, dp, dp, dp]
        L.fso_element_quad4.restype = C.c_int
        L.fso_bsr_pattern.argtypes = [C.c_int32, C.c_int32, ip, C.c_int32, ip, ip, ip]
        L.fso_bsr_pattern.restype = C.c_int64
        L.fso_assemble_bsr.argtypes = [C.c_int32, dp, C.c_int32, ip, C.c_int32, ip, C.POINTER(Material),
                                       bp, dp, ip, ip, dp, dp]
        L.fso_assemble_bsr.restype = C.c_int
        L.fso_bsr_spmv.argtypes = [C.c_int32, ip, ip, dp, dp, dp]
        L.fso_pcg_block_jacobi.argtypes = [C.c_int32, ip, ip, dp, dp, C.c_double, C.c_int32, dp, dp,
                                           C.POINTER(PcgInfo)]
        L.fso_pcg_block_jacobi.restype = C.c_int
        L.fso_time_assembly.argtypes = [C.c_int32, dp, C.c_int32, ip, C.POINTER(Material), bp, dp, ip, ip,
                                        dp, dp, C.c_int32]
        L.fso_time_assembly.restype = C.c_double
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _b(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None


def material(nu, E, t, flags=REF_DEFAULT):
    return Material(float(nu), float(E), float(t), int(flags))


def material_matrices(mat):
    Dm = np.zeros(9)
    Dp = np.zeros(9)
    lib().fso_material_matrices(C.byref(mat), _d(Dm), _d(Dp))
    return Dm.reshape(3, 3), Dp.reshape(3, 3)


def specht_B(C3, L1, L2, dphi):
    C3 = np.ascontiguousarray(C3, dtype=np.float64)
    dphi = np.ascontiguousarray(dphi, dtype=np.float64).reshape(-1)
    B = np.zeros(27)
    lib().fso_tri3_specht_B(_d(C3), float(L1), float(L2), _d(dphi), _d(B))
    return B.reshape(3, 9)


def element_tri3(xyz, mat, want_parts=False):
    """xyz: (3,3).  Returns Ke (18,18) in the reference's variable-major order."""
    X = np.ascontiguousarray(xyz, dtype=np.float64).reshape(9)
    Ke = np.zeros(324)
    parts = Tri3Parts()
    rc = lib().fso_element_tri3(_d(X), C.byref(mat), _d(Ke), C.byref(parts))
    if rc:
        raise ValueError("degenerate TRI3 element")
    Ke = Ke.reshape(18, 18)
    if not want_parts:
        return Ke
    p = {
        "trafo": np.array(parts.trafo).reshape(3, 3),
        "transUV": np.array(parts.transUV).reshape(3, 2),
        "dphi": np.array(parts.dphi).reshape(3, 2),
        "area": parts.area,
        "Ke_m": np.array(parts.Ke_m).reshape(6, 6),
        "Ke_p": np.array(parts.Ke_p).reshape(9, 9),
        "K_local": np.array(parts.K_local).reshape(18, 18),
        "K_global_nm": np.array(parts.K_global_nm).reshape(18, 18),
    }
    return Ke, p


def element_quad4(xyz, mat, want_parts=False):
    X = np.ascontiguousarray(xyz, dtype=np.float64).reshape(12)
    Ke = np.zeros(576)
    Km = np.zeros(64)
    Kp = np.zeros(144)
    Kg = np.zeros(576)
    rc = lib().fso_element_quad4(_d(X), C.byref(mat), _d(Ke), _d(Km), _d(Kp), _d(Kg))
    if rc:
        raise ValueError("degenerate QUAD4 element")
    Ke = Ke.reshape(24, 24)
    if not want_parts:
        return Ke
    return Ke, {"Ke_m": Km.reshape(8, 8), "Ke_p": Kp.reshape(12, 12), "K_global_nm": Kg.reshape(24, 24)}


def bsr_pattern(n_nodes, tri, quad):
    tri = np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 3)
    quad = np.ascontiguousarray(quad, dtype=np.int32).reshape(-1, 4)
    rowptr = np.zeros(n_nodes + 1, dtype=np.int32)
    nnzb = lib().fso_bsr_pattern(n_nodes, len(tri), _i(tri), len(quad), _i(quad), _i(rowptr), None)
    colidx = np.zeros(nnzb, dtype=np.int32)
    lib().fso_bsr_pattern(n_nodes, len(tri), _i(tri), len(quad), _i(quad), _i(rowptr), _i(colidx))
    return rowptr, colidx


def assemble(xyz, tri, quad, mat, dirichlet=None, loads=None, pattern=None):
    """Returns (rowptr, colidx, vals[nnzb,6,6], F[6*n_nodes])."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    n = len(xyz)
    tri = np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 3)
    quad = np.ascontiguousarray(quad, dtype=np.int32).reshape(-1, 4)
    if pattern is None:
        pattern = bsr_pattern(n, tri, quad)
    rowptr, colidx = pattern
    dirichlet = None if dirichlet is None else np.ascontiguousarray(dirichlet, dtype=np.uint8)
    loads = None if loads is None else np.ascontiguousarray(loads, dtype=np.float64).reshape(n, 6)
    vals = np.zeros((len(colidx), 6, 6))
    F = np.zeros(6 * n)
    rc = lib().fso_assemble_bsr(n, _d(xyz), len(tri), _i(tri), len(quad), _i(quad), C.byref(mat),
                                _b(dirichlet), _d(loads), _i(rowptr), _i(colidx), _d(vals), _d(F))
    if rc:
        raise ValueError("degenerate element %d" % (-rc - 1))
    return rowptr, colidx, vals, F


def spmv(rowptr, colidx, vals, x):
    n = len(rowptr) - 1
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(6 * n)
    lib().fso_bsr_spmv(n, _i(rowptr), _i(colidx), _d(vals), _d(x), _d(y))
    return y


def pcg(rowptr, colidx, vals, b, rtol=1e-10, max_it=10000, history=False):
    n = len(rowptr) - 1
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(6 * n)
    hist = np.zeros(max_it) if history else None
    info = PcgInfo()
    rc = lib().fso_pcg_block_jacobi(n, _i(rowptr), _i(colidx), _d(vals), _d(b), rtol, max_it, _d(x),
                                    _d(hist), C.byref(info))
    if rc:
        raise ValueError("oracle PCG setup failed (%d)" % rc)
    out = {"iterations": info.iterations, "converged": info.converged,
           "rel_residual": info.rel_residual, "seconds": info.seconds}
    if history:
        out["history"] = hist[: info.iterations]
    return x, out


def to_scipy(rowptr, colidx, vals):
    import scipy.sparse as sp

    n = len(rowptr) - 1
    return sp.bsr_matrix((vals, colidx, rowptr), shape=(6 * n, 6 * n)).tocsr()


def direct_solve(rowptr, colidx, vals, F):
    import scipy.sparse.linalg as spla

    K = to_scipy(rowptr, colidx, vals).tocsc()
    return spla.spsolve(K, F)


def refined_solve(rowptr, colidx, vals, F, sweeps=4, return_history=False):
    """Sparse LU followed by iterative refinement with the residual accumulated in extended precision
    (numpy longdouble, row sums by reduceat over the CSR arrays).  On the ill-conditioned shell systems plain LU
    is only good to ~kappa*eps (1e-10 on a 1k-element cantilever, 1e-8 on the 250k-element roof); this is the
    reference the displacement parity tests use."""
    import scipy.sparse.linalg as spla

    K = to_scipy(rowptr, colidx, vals)
    K.sort_indices()
    lu = spla.splu(K.tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    data = K.data.astype(np.longdouble)
    starts = K.indptr[:-1]
    assert np.all(np.diff(K.indptr) > 0)
    Fl = F.astype(np.longdouble)
    u = lu.solve(F).astype(np.longdouble)
    hist = []
    for _ in range(sweeps):
        res = Fl - np.add.reduceat(data * u[K.indices], starts)
        hist.append(float(np.sqrt((res * res).sum()) / np.sqrt((Fl * Fl).sum())))
        u = u + lu.solve(res.astype(np.float64)).astype(np.longdouble)
    if return_history:
        return u.astype(np.float64), hist
    return u.astype(np.float64)
