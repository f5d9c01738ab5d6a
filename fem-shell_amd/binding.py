"""ctypes mirror of include/femshell.h (no arithmetic here; see __init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")

REF_Y21 = 0x1
REF_DRILL_MAX = 0x2
REASSEMBLE_EACH_SOLVE = 0x4
REF_DEFAULT = REF_Y21 | REF_DRILL_MAX
REORDER_MORTON = 0x10
REORDER_RCM = 0x20

KERNEL_ASSEMBLE, KERNEL_SPMV, KERNEL_CG_UPDATE, KERNEL_CG_DIRECTION = 0, 1, 2, 3

SYMBOLS = [
    "femshell_create", "femshell_destroy", "femshell_last_error", "femshell_set_mesh",
    "femshell_set_dirichlet", "femshell_set_loads", "femshell_assemble", "femshell_assemble_async", "femshell_solve",
    "femshell_get_solution", "femshell_residual_history", "femshell_element_matrices",
    "femshell_nnz_blocks", "femshell_export_bsr", "femshell_spmv", "femshell_row_begin",
    "femshell_row_end", "femshell_comm_unique_id", "femshell_comm_init", "femshell_time_kernel",
    "femshell_sync", "femshell_pc_defaults", "femshell_set_preconditioner", "femshell_amg_levels", "femshell_amg_level",
    "femshell_amg_export", "femshell_residual", "femshell_comm_ranks", "femshell_amg_setup_stats", "femshell_amg_dense_stats", "femshell_amg_partition_info", "femshell_assembly_kernel",
    "femshell_amg_cycle_bytes", "femshell_comm_selftest", "femshell_comm_counters", "femshell_owned_nodes", "femshell_comm_bytes", "femshell_set_initial_guess",
    "femshell_amg_patch_info",
]


class FemShellError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("femshell error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("nu", C.c_double), ("E", C.c_double), ("thickness", C.c_double), ("flags", C.c_uint32),
                ("device", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32)]


class SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("rel_residual", C.c_double),
                ("true_rel_residual", C.c_double), ("assemble_seconds", C.c_double), ("setup_seconds", C.c_double), ("solve_seconds", C.c_double),
                ("bytes_per_iteration", C.c_double), ("pc_type", C.c_int32), ("amg_levels", C.c_int32),
                ("pc_setup_seconds", C.c_double), ("operator_complexity", C.c_double),
                ("refine_passes_done", C.c_int32), ("pc_fp64_fallback", C.c_int32), ("refine_correction_rel", C.c_double),
                ("refine_residual_reduction", C.c_double), ("error_estimate", C.c_double)]


class PcOptions(C.Structure):
    _fields_ = [("type", C.c_int32), ("cycle", C.c_int32), ("smoother_degree", C.c_int32), ("coarse_degree", C.c_int32),
                ("coarsest_nodes", C.c_int32), ("max_levels", C.c_int32), ("refine_passes", C.c_int32), ("reserved", C.c_int32),
                ("eig_ratio", C.c_double)]


class AmgLevelInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_coarse", C.c_int32), ("nnz_blocks", C.c_int64), ("p_blocks", C.c_int64),
                ("lambda_max", C.c_double)]


PC_BLOCK_JACOBI, PC_AMG = 0, 1
CYCLE_V, CYCLE_K = 0, 1


def library_path():
    return os.path.join(_HERE, "libfemshell.so")


def build_library(force=False):
    """Compile csrc/ for gfx950 with hipcc (in-tree, so the .so travels with the repo)."""
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", _CSRC, "-s"])
    return library_path()


_lib = None


def load_library():
    """Loads libfemshell.so; raises if it has not been built (there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    host_only = os.environ.get("FEMSHELL_HOST_LIBRARY")
    if host_only:
        # sanitizer builds of the host-side code only (`make -C fem-shell_amd/csrc san`, tools/run_sanitizers.sh): they
        # export include/femshell_plan.h and femshell_last_error, nothing that touches a GPU
        L = C.CDLL(host_only)
        L.femshell_last_error.restype = C.c_char_p
        _lib = L
        return L
    path = library_path()
    if not os.path.exists(path):
        raise FemShellError(-2, "libfemshell.so is missing: build it with `make -C fem-shell_amd/csrc` "
                                "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(path)
    dp, ip, bp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    vp = C.c_void_p
    L.femshell_last_error.restype = C.c_char_p
    L.femshell_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.femshell_destroy.argtypes = [vp]
    L.femshell_set_mesh.argtypes = [vp, C.c_int32, dp, C.c_int32, ip, C.c_int32, ip]
    L.femshell_set_dirichlet.argtypes = [vp, C.c_int32, ip, bp]
    L.femshell_set_loads.argtypes = [vp, C.c_int32, ip, dp]
    L.femshell_assemble.argtypes = [vp]
    L.femshell_assemble_async.argtypes = [vp]
    L.femshell_solve.argtypes = [vp, C.c_double, C.c_int32, dp, C.POINTER(SolveInfo)]
    L.femshell_get_solution.argtypes = [vp, dp]
    L.femshell_residual_history.argtypes = [vp, dp, C.c_int32]
    L.femshell_residual_history.restype = C.c_int32
    L.femshell_element_matrices.argtypes = [vp, C.c_int32, C.c_int32, dp]
    L.femshell_nnz_blocks.argtypes = [vp]
    L.femshell_nnz_blocks.restype = C.c_int64
    L.femshell_export_bsr.argtypes = [vp, ip, ip, dp, dp]
    L.femshell_spmv.argtypes = [vp, dp, dp]
    L.femshell_residual.argtypes = [vp, dp, dp]
    L.femshell_row_begin.argtypes = [vp]
    L.femshell_row_begin.restype = C.c_int32
    L.femshell_row_end.argtypes = [vp]
    L.femshell_row_end.restype = C.c_int32
    L.femshell_set_initial_guess.argtypes = [vp, dp]
    L.femshell_owned_nodes.argtypes = [vp, C.POINTER(C.c_int32)]
    L.femshell_owned_nodes.restype = C.c_int32
    L.femshell_comm_unique_id.argtypes = [bp]
    L.femshell_comm_init.argtypes = [vp, bp]
    L.femshell_comm_ranks.argtypes = [vp]
    L.femshell_comm_ranks.restype = C.c_int32
    L.femshell_time_kernel.argtypes = [vp, C.c_int, C.c_int32, dp, dp]
    L.femshell_sync.argtypes = [vp]
    L.femshell_pc_defaults.argtypes = [C.c_int32, C.POINTER(PcOptions)]
    L.femshell_set_preconditioner.argtypes = [vp, C.POINTER(PcOptions)]
    L.femshell_amg_levels.argtypes = [vp]
    L.femshell_amg_levels.restype = C.c_int32
    L.femshell_amg_level.argtypes = [vp, C.c_int32, C.POINTER(AmgLevelInfo)]
    L.femshell_amg_export.argtypes = [vp, C.c_int32, C.c_int32, C.c_void_p]
    L.femshell_amg_export.restype = C.c_int64
    L.femshell_amg_setup_stats.argtypes = [vp, dp]
    L.femshell_amg_dense_stats.argtypes = [vp, dp]
    L.femshell_amg_patch_info.argtypes = [vp, dp]
    L.femshell_amg_partition_info.argtypes = [vp, dp]
    L.femshell_assembly_kernel.argtypes = [vp]
    L.femshell_comm_selftest.argtypes = [vp, dp]
    L.femshell_comm_counters.argtypes = [vp, C.POINTER(C.c_int64), C.c_int32]
    L.femshell_comm_bytes.argtypes = [vp, C.POINTER(C.c_int64), C.c_int32]
    L.femshell_amg_cycle_bytes.argtypes = [vp, dp, C.c_int32]
    L.femshell_amg_cycle_bytes.restype = C.c_int32
    for name in SYMBOLS:
        if name != "femshell_last_error" and not name.startswith("femshell_nnz") and \
                not name.startswith("femshell_row") and name != "femshell_residual_history" and \
                name not in ("femshell_amg_levels", "femshell_amg_export", "femshell_comm_ranks"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _b(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None


def _check(rc):
    if rc != 0:
        raise FemShellError(rc, load_library().femshell_last_error().decode())


def comm_unique_id():
    buf = np.zeros(128, dtype=np.uint8)
    _check(load_library().femshell_comm_unique_id(_b(buf)))
    return buf


class FemShell:
    """One context = one GPU = one rank of the row partition."""

    def __init__(self, nu, E, thickness, flags=REF_DEFAULT, device=-1, rank=0, world_size=1):
        self._L = load_library()
        self._h = C.c_void_p()
        cfg = Config(float(nu), float(E), float(thickness), int(flags), int(device), int(rank), int(world_size))
        _check(self._L.femshell_create(C.byref(cfg), C.byref(self._h)))
        self.n_nodes = 0
        self.n_tri = 0
        self.n_quad = 0

    def close(self):
        if self._h:
            self._L.femshell_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def comm_init(self, unique_id):
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        assert uid.size == 128
        _check(self._L.femshell_comm_init(self._h, _b(uid)))

    def comm_ranks(self):
        return int(self._L.femshell_comm_ranks(self._h))

    def set_mesh(self, xyz, tri=None, quad=None):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        tri = np.zeros((0, 3), np.int32) if tri is None else np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 3)
        quad = np.zeros((0, 4), np.int32) if quad is None else np.ascontiguousarray(quad, dtype=np.int32).reshape(-1, 4)
        _check(self._L.femshell_set_mesh(self._h, len(xyz), _d(xyz), len(tri), _i(tri), len(quad), _i(quad)))
        self.n_nodes, self.n_tri, self.n_quad = len(xyz), len(tri), len(quad)

    def set_dirichlet(self, mask, node_ids=None):
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        ids = None if node_ids is None else np.ascontiguousarray(node_ids, dtype=np.int32)
        _check(self._L.femshell_set_dirichlet(self._h, len(mask), _i(ids), _b(mask)))

    def set_loads(self, f6, node_ids=None):
        f6 = np.ascontiguousarray(f6, dtype=np.float64).reshape(-1, 6)
        ids = None if node_ids is None else np.ascontiguousarray(node_ids, dtype=np.int32)
        _check(self._L.femshell_set_loads(self._h, len(f6), _i(ids), _d(f6)))

    def assemble(self, wait=True):
        """wait=False: femshell_assemble_async -- enqueued only; sync(), solve() ... report a failed element."""
        _check(self._L.femshell_assemble(self._h) if wait else self._L.femshell_assemble_async(self._h))

    def solve(self, rtol=1e-10, max_it=10000, fetch=True):
        u = np.zeros((self.n_nodes, 6)) if fetch else None
        info = SolveInfo()
        _check(self._L.femshell_solve(self._h, float(rtol), int(max_it), _d(u), C.byref(info)))
        d = {f[0]: getattr(info, f[0]) for f in SolveInfo._fields_}
        return u, d

    def get_solution(self):
        u = np.zeros((self.n_nodes, 6))
        _check(self._L.femshell_get_solution(self._h, _d(u)))
        return u

    def residual_history(self, cap=1 << 20):
        h = np.zeros(cap)
        n = self._L.femshell_residual_history(self._h, _d(h), cap)
        return h[:n].copy()

    def element_matrices(self, first, count):
        size = 324 if first < self.n_tri else 576
        out = np.zeros((count, size))
        _check(self._L.femshell_element_matrices(self._h, int(first), int(count), _d(out)))
        n = 18 if size == 324 else 24
        return out.reshape(count, n, n)

    def nnz_blocks(self):
        return int(self._L.femshell_nnz_blocks(self._h))

    def export_bsr(self):
        nb = self.nnz_blocks()
        rowptr = np.zeros(self.n_nodes + 1, dtype=np.int32)
        colidx = np.zeros(nb, dtype=np.int32)
        vals = np.zeros((nb, 6, 6))
        F = np.zeros(6 * self.n_nodes)
        _check(self._L.femshell_export_bsr(self._h, _i(rowptr), _i(colidx), _d(vals), _d(F)))
        return rowptr, colidx, vals, F

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        y = np.zeros_like(x)
        _check(self._L.femshell_spmv(self._h, _d(x), _d(y)))
        return y

    def residual(self, x):
        """F - K x with double-double products and row sums."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        r = np.zeros_like(x)
        _check(self._L.femshell_residual(self._h, _d(x), _d(r)))
        return r

    def row_range(self):
        return int(self._L.femshell_row_begin(self._h)), int(self._L.femshell_row_end(self._h))

    def set_initial_guess(self, u0=None):
        """The next solve starts from u0 (n_nodes x 6) -- None: from the previous solve's solution, where it lies in HBM."""
        if u0 is not None:
            u0 = np.ascontiguousarray(u0, dtype=np.float64).reshape(-1)
            if len(u0) != 6 * self.n_nodes:
                raise ValueError("u0 needs n_nodes x 6 entries")
        _check(self._L.femshell_set_initial_guess(self._h, _d(u0)))

    def owned_nodes(self):
        """the caller's ids of the node rows this rank owns, in the order export_bsr gives them"""
        n = int(self._L.femshell_owned_nodes(self._h, None))
        ids = np.zeros(n, dtype=np.int32)
        self._L.femshell_owned_nodes(self._h, _i(ids))
        return ids

    def time_kernel(self, which, reps=10):
        ms = C.c_double()
        by = C.c_double()
        _check(self._L.femshell_time_kernel(self._h, int(which), int(reps), C.byref(ms), C.byref(by)))
        return ms.value, by.value

    def sync(self):
        _check(self._L.femshell_sync(self._h))

    def set_preconditioner(self, kind="amg", **options):
        """kind: "jacobi" (6x6 block-Jacobi, the default) or "amg" (smoothed-aggregation multigrid);
        options: cycle ("V"/"K"), smoother_degree, coarse_degree, coarsest_nodes, max_levels, refine_passes, eig_ratio."""
        o = PcOptions()
        _check(self._L.femshell_pc_defaults(PC_AMG if kind == "amg" else PC_BLOCK_JACOBI, C.byref(o)))
        for k, v in options.items():
            if k == "cycle":
                v = CYCLE_K if str(v).upper() == "K" else CYCLE_V
            if not hasattr(o, k):
                raise TypeError("unknown preconditioner option " + k)
            setattr(o, k, v)
        _check(self._L.femshell_set_preconditioner(self._h, C.byref(o)))

    def amg_levels(self):
        """Per-level facts of the multigrid hierarchy of the last solve (empty with block-Jacobi)."""
        out = []
        for l in range(self._L.femshell_amg_levels(self._h)):
            info = AmgLevelInfo()
            _check(self._L.femshell_amg_level(self._h, l, C.byref(info)))
            out.append({f[0]: getattr(info, f[0]) for f in AmgLevelInfo._fields_})
        return out

    def comm_bytes(self, clear=False):
        """(bytes handed to the sends of halo exchanges, bytes contributed to all-reduces and broadcasts) since the last clear"""
        out = (C.c_int64 * 2)()
        rc = self._L.femshell_comm_bytes(self._h, out, 1 if clear else 0)
        if rc < 0:
            _check(rc)
        return int(out[0]), int(out[1])

    def comm_counters(self, clear=False):
        """Communication enqueued since the counters were cleared (femshell_comm_counters)."""
        out = (C.c_int64 * 4)()
        rc = self._L.femshell_comm_counters(self._h, out, 1 if clear else 0)
        if rc < 0:
            _check(rc)
        return {"halo_exchanges_on_the_halo_stream": int(out[0]), "halo_exchanges_on_the_main_stream": int(out[1]),
                "allreduces": int(out[2]), "row_gathers": int(out[3])}

    def comm_selftest(self):
        """Microseconds of the three communication patterns femshell_comm_init checked on first contact (None: no communicator)."""
        out = np.zeros(3)
        rc = self._L.femshell_comm_selftest(self._h, _d(out))
        if rc < 0:
            _check(rc)
        return None if rc == 0 else {"halo_send_recv_on_second_stream_beside_allreduce_us": float(out[0]),
                                     "grouped_broadcast_row_gather_us": float(out[1]), "allreduce_3_words_us": float(out[2])}

    def amg_cycle_bytes(self):
        """Algorithmic HBM bytes of one multigrid cycle per level (femshell_amg_cycle_bytes)."""
        out = np.zeros(32)
        n = self._L.femshell_amg_cycle_bytes(self._h, _d(out), 32)
        if n < 0:
            _check(n)
        return out[:n].copy()

    def amg_setup_stats(self):
        """Device timings of the first coarsening step of the last multigrid setup."""
        out = np.zeros(7)
        _check(self._L.femshell_amg_setup_stats(self._h, _d(out)))
        return {"prolongator_ms": out[0], "ap_ms": out[1], "restriction_ms": out[2], "galerkin_ms": out[3],
                "galerkin_useful_flops": out[4], "galerkin_mfma_flops_issued": out[5], "galerkin_on_matrix_cores": bool(out[6])}

    def amg_symbolic_info(self):
        """Coarsening steps of the last setup by where their patterns were built."""
        out = np.zeros(3, dtype=np.int32)
        self._L.femshell_amg_symbolic_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        _check(self._L.femshell_amg_symbolic_info(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return {"in_hbm": int(out[0]), "host_after_overflow": int(out[1]), "host_by_rule": int(out[2])}

    def assembly_kernel(self):
        """Name of the kernel femshell_assemble launches for this mesh."""
        k = self._L.femshell_assembly_kernel(self._h)
        if k < 0:
            _check(k)
        return "k_assemble_pipe" if k == 1 else "k_assemble"

    def amg_dense_stats(self):
        """The dense inverse of the coarsest operator when the matrix cores computed it (n = 0: host path)."""
        out = np.zeros(6)
        _check(self._L.femshell_amg_dense_stats(self._h, _d(out)))
        return {"n": int(out[0]), "ms": out[1], "mfma_flops_issued": out[2], "useful_flops": out[3], "dropped_directions": int(out[4]),
                "bytes": out[5]}

    def amg_patch_info(self):
        """The patch smoother of level 0 (csrc/amg_patch.hpp): rigid edges, clusters, nodes in them; all zero without rigid edges."""
        out = np.zeros(6)
        _check(self._L.femshell_amg_patch_info(self._h, _d(out)))
        return {"rigid_edges": int(out[0]), "clusters": int(out[1]), "nodes_in_clusters": int(out[2]), "not_positive_definite": int(out[3]),
                "tau": out[4], "max_nodes": int(out[5])}

    def amg_partition_info(self):
        """Row-partitioned hierarchy (femshell_amg_partition_info): levels split over the ranks, bytes that shrink with the
        rank count, bytes every rank holds in full."""
        out = np.zeros(6)
        _check(self._L.femshell_amg_partition_info(self._h, _d(out)))
        return {"partitioned_levels": int(out[0]), "bytes_partitioned": out[1], "bytes_replicated": out[2],
                "rows_on_last_partitioned_level": int(out[3]), "ghost_rows_on_last_partitioned_level": int(out[4]),
                "nodes_of_first_replicated_level": int(out[5])}

    def amg_export(self, level):
        """Host copies of a level (small problems): dict with agg, A (rowptr, cols, vals), P (rowptr, cols, vals)."""
        names = {"agg": (0, np.int32), "A_rowptr": (1, np.int64), "A_cols": (2, np.int32), "A_vals": (3, np.float64),
                 "P_rowptr": (4, np.int64), "P_cols": (5, np.int32), "P_vals": (6, np.float64),
                 "coarse_inverse": (7, np.float64),  # (the coarsest level only: its dense inverse, n x n)
                 "patch_labels": (8, np.int32)}      # (level 0: cluster of every node or -1; None without clusters)
        out = {}
        for name, (which, dt) in names.items():
            n = self._L.femshell_amg_export(self._h, level, which, None)
            if n < 0:
                out[name] = None
                continue
            a = np.zeros(n, dtype=dt)
            if n:
                self._L.femshell_amg_export(self._h, level, which, a.ctypes.data_as(C.c_void_p))
            out[name] = a.reshape(-1, 6, 6) if name.endswith("vals") else (a.reshape(int(round(np.sqrt(n))), -1) if name == "coarse_inverse" else a)
        return out


# ---- host-only plan inspection (include/femshell_plan.h); needs no GPU -----------------------

PLAN_INFO = ["n_own", "n_pad", "n_ghost", "n_slices", "n_ltri", "n_lquad", "total_slots", "n_pairs",
             "n_peers", "row_begin", "row_end", "nnz_blocks", "n_interior_slices", "n_items",
             "n_multi_round_slices", "max_slice_elems", "max_slice_width", "symmetric", "stored_blocks", "pipe"]
PLAN_ARRAYS = {
    "ghost_global": (0, np.int32), "tri_local": (1, np.int32), "tri_global_id": (2, np.int32),
    "quad_local": (3, np.int32), "quad_global_id": (4, np.int32), "slice_width": (5, np.int32),
    "slice_base": (6, np.int64), "cols": (7, np.int32), "pair_ptr": (8, np.int32), "pairs": (9, np.uint32),
    "xyz_local": (10, np.float64), "peer_ranks": (11, np.int32), "peer_recv_offset": (12, np.int32),
    "peer_recv_count": (13, np.int32), "peer_send_ptr": (14, np.int32), "peer_send_nodes": (15, np.int32),
    "spmv_order": (16, np.int32), "in_width": (17, np.int32), "in_base": (18, np.int64), "in_slots": (19, np.int32),
    "gat_slots": (20, np.int32), "loc_list": (21, np.uint8), "loc_index": (22, np.uint8),
    "item_ptr": (23, np.int32), "items": (24, np.uint32), "pairs16": (25, np.uint16), "slice_elem_ptr": (26, np.int32),
}


def build_plan(xyz, tri=None, quad=None, rank=0, world_size=1):
    """Returns the symbolic plan of one rank as a dict of numpy arrays (CPU only)."""
    L = load_library()
    L.femshell_plan_create.argtypes = [C.c_int32, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_int32),
                                       C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                                       C.POINTER(C.c_void_p)]
    L.femshell_plan_create.restype = C.c_int
    L.femshell_plan_destroy.argtypes = [C.c_void_p]
    L.femshell_plan_destroy.restype = None
    L.femshell_plan_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.femshell_plan_array.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.femshell_plan_array.restype = C.c_int64
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    tri = np.zeros((0, 3), np.int32) if tri is None else np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 3)
    quad = np.zeros((0, 4), np.int32) if quad is None else np.ascontiguousarray(quad, dtype=np.int32).reshape(-1, 4)
    h = C.c_void_p()
    _check(L.femshell_plan_create(len(xyz), _d(xyz), len(tri), _i(tri), len(quad), _i(quad), rank, world_size,
                                  C.byref(h)))
    try:
        info = np.zeros(len(PLAN_INFO), dtype=np.int64)
        _check(L.femshell_plan_info(h, info.ctypes.data_as(C.POINTER(C.c_int64))))
        out = {k: int(v) for k, v in zip(PLAN_INFO, info)}
        for name, (which, dt) in PLAN_ARRAYS.items():
            n = L.femshell_plan_array(h, which, None)
            a = np.zeros(n, dtype=dt)
            if n:
                L.femshell_plan_array(h, which, a.ctypes.data_as(C.c_void_p))
            out[name] = a
    finally:
        L.femshell_plan_destroy(h)
    return out


def plan_node_normals(xyz, tri=None, quad=None, from_gather_lists=True, rank=0, world_size=1):
    """Unit normals of the owned nodes of one rank's plan (CPU only): by the walk over all elements or from the gather lists."""
    L = load_library()
    L.femshell_plan_create.argtypes = [C.c_int32, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_int32),
                                       C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                                       C.POINTER(C.c_void_p)]
    L.femshell_plan_create.restype = C.c_int
    L.femshell_plan_destroy.argtypes = [C.c_void_p]
    L.femshell_plan_destroy.restype = None
    L.femshell_plan_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.femshell_plan_node_normals.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]
    L.femshell_plan_node_normals.restype = C.c_int
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    tri = np.zeros((0, 3), np.int32) if tri is None else np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 3)
    quad = np.zeros((0, 4), np.int32) if quad is None else np.ascontiguousarray(quad, dtype=np.int32).reshape(-1, 4)
    h = C.c_void_p()
    _check(L.femshell_plan_create(len(xyz), _d(xyz), len(tri), _i(tri), len(quad), _i(quad), rank, world_size, C.byref(h)))
    try:
        info = np.zeros(len(PLAN_INFO), dtype=np.int64)
        _check(L.femshell_plan_info(h, info.ctypes.data_as(C.POINTER(C.c_int64))))
        out = np.zeros((int(info[PLAN_INFO.index("n_own")]), 3))
        _check(L.femshell_plan_node_normals(h, 1 if from_gather_lists else 0, _d(out)))
    finally:
        L.femshell_plan_destroy(h)
    return out


# ---- host-only pieces of the multigrid setup (include/femshell_plan.h); need no GPU --------------------------

COARSEN_ARRAYS = {"agg": (0, np.int32), "P_rowptr": (1, np.int64), "P_cols": (2, np.int32), "P_vals": (3, np.float64),
                  "Ac_rowptr": (4, np.int64), "Ac_cols": (5, np.int32), "Ac_vals": (6, np.float64), "Bc": (7, np.float64)}


def amg_host_rbm(xyz, dmask=None):
    L = load_library()
    L.femshell_amg_host_rbm.argtypes = [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_double)]
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    dm = None if dmask is None else np.ascontiguousarray(dmask, dtype=np.uint8)
    B = np.zeros((len(xyz), 6, 6))
    _check(L.femshell_amg_host_rbm(len(xyz), _d(xyz), _b(dm), _d(B)))
    return B


def amg_host_coarsen(rowptr, colidx, vals, B, lambda_max):
    """One coarsening step of the library's host setup: dict with agg, P (BSR arrays), Ac (BSR arrays), Bc."""
    L = load_library()
    L.femshell_amg_host_coarsen.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                            C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_void_p)]
    L.femshell_amg_coarsening_destroy.argtypes = [C.c_void_p]
    L.femshell_amg_coarsening_destroy.restype = None
    L.femshell_amg_coarsening_array.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.femshell_amg_coarsening_array.restype = C.c_int64
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    h = C.c_void_p()
    _check(L.femshell_amg_host_coarsen(len(rowptr) - 1, _i(rowptr), _i(colidx), _d(vals), _d(B), float(lambda_max), C.byref(h)))
    try:
        out = {}
        for name, (which, dt) in COARSEN_ARRAYS.items():
            n = L.femshell_amg_coarsening_array(h, which, None)
            a = np.zeros(n, dtype=dt)
            if n:
                L.femshell_amg_coarsening_array(h, which, a.ctypes.data_as(C.c_void_p))
            out[name] = a
    finally:
        L.femshell_amg_coarsening_destroy(h)
    out["P_vals"] = out["P_vals"].reshape(-1, 6, 6)
    out["Ac_vals"] = out["Ac_vals"].reshape(-1, 6, 6)
    out["Bc"] = out["Bc"].reshape(-1, 6, 6)
    return out


def amg_host_aggregate(rowptr, colidx, visit=None):
    """The library's greedy aggregation of a node graph: (agg, n_aggregates); visit = order of the passes (optional)."""
    L = load_library()
    L.femshell_amg_host_aggregate.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                              C.POINTER(C.c_int32)]
    L.femshell_amg_host_aggregate.restype = C.c_int32
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    agg = np.zeros(len(rowptr) - 1, dtype=np.int32)
    v = None if visit is None else np.ascontiguousarray(visit, dtype=np.int32)
    na = L.femshell_amg_host_aggregate(len(rowptr) - 1, _i(rowptr), _i(colidx), None if v is None else _i(v), _i(agg))
    if na < 0:
        _check(-1)  # FEMSHELL_ERR_INVALID; the message is in femshell_last_error
    return agg, na


def amg_host_patch_clusters(rowptr, colidx, vals, tau=0.8, max_nodes=8):
    """Clusters of rigidly coupled nodes of a host matrix (csrc/amg_patch.hpp): (labels, n_clusters, rigid edges)."""
    L = load_library()
    L.femshell_amg_host_patch_clusters.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_double,
                                                   C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.femshell_amg_host_patch_clusters.restype = C.c_int32
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    labels = np.zeros(len(rowptr) - 1, dtype=np.int32)
    edges = C.c_int64(0)
    nc = L.femshell_amg_host_patch_clusters(len(rowptr) - 1, _i(rowptr), _i(colidx), _d(vals), tau, max_nodes, _i(labels), C.byref(edges))
    if nc < 0:
        _check(-1)
    return labels, nc, edges.value


def amg_host_aggregate_glued(rowptr, colidx, labels, visit=None):
    """The aggregation with the clusters glued into one node each: (agg, n_aggregates)."""
    L = load_library()
    L.femshell_amg_host_aggregate_glued.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.femshell_amg_host_aggregate_glued.restype = C.c_int32
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    agg = np.zeros(len(rowptr) - 1, dtype=np.int32)
    v = None if visit is None else np.ascontiguousarray(visit, dtype=np.int32)
    na = L.femshell_amg_host_aggregate_glued(len(rowptr) - 1, _i(rowptr), _i(colidx), _i(labels), None if v is None else _i(v), _i(agg))
    if na < 0:
        _check(-1)
    return agg, na


def amg_host_dense_inverse(rowptr, colidx, vals):
    L = load_library()
    L.femshell_amg_host_dense_inverse.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                                  C.POINTER(C.c_double)]
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    n = 6 * (len(rowptr) - 1)
    inv = np.zeros((n, n))
    _check(L.femshell_amg_host_dense_inverse(len(rowptr) - 1, _i(rowptr), _i(colidx), _d(vals), _d(inv)))
    return inv


def amg_host_pack(rowptr, colidx, vals, diag_first):
    """Sliced block ELL image of a BSR matrix: (slice_width, slice_base, cols, ell_vals)."""
    L = load_library()
    L.femshell_amg_host_pack.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int32,
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.femshell_amg_host_pack.restype = C.c_int64
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    n = len(rowptr) - 1
    total = L.femshell_amg_host_pack(n, _i(rowptr), _i(colidx), _d(vals), int(diag_first), None, None, None, None)
    ns = (n + 31) // 32
    sw = np.zeros(ns, dtype=np.int32)
    sb = np.zeros(ns + 1, dtype=np.int64)
    cols = np.zeros(total, dtype=np.int32)
    ev = np.zeros(total * 36)
    L.femshell_amg_host_pack(n, _i(rowptr), _i(colidx), _d(vals), int(diag_first), _i(sw),
                             sb.ctypes.data_as(C.POINTER(C.c_int64)), _i(cols), _d(ev))
    return sw, sb, cols, ev


def amg_host_pack_sym(rowptr, colidx, vals):
    """Symmetric-storage image of a square BSR matrix: dict of the ELL arrays and the in-lists."""
    L = load_library()
    i32, i64, dbl = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)
    L.femshell_amg_host_pack_sym.argtypes = [C.c_int32, i32, i32, dbl, i32, i64, i32, dbl, i32, i64, i32, i32, i64]
    L.femshell_amg_host_pack_sym.restype = C.c_int64
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    n = len(rowptr) - 1
    in_total = C.c_int64()
    total = L.femshell_amg_host_pack_sym(n, _i(rowptr), _i(colidx), _d(vals), None, None, None, None, None, None, None, None,
                                         C.byref(in_total))
    ns = (n + 31) // 32
    out = {"slice_width": np.zeros(ns, np.int32), "slice_base": np.zeros(ns + 1, np.int64), "cols": np.zeros(total, np.int32),
           "vals": np.zeros(total * 36), "in_width": np.zeros(ns, np.int32), "in_base": np.zeros(ns + 1, np.int64),
           "in_slots": np.zeros(in_total.value, np.int32), "in_rows": np.zeros(in_total.value, np.int32)}
    p64 = lambda a: a.ctypes.data_as(i64)  # noqa: E731
    L.femshell_amg_host_pack_sym(n, _i(rowptr), _i(colidx), _d(vals), _i(out["slice_width"]), p64(out["slice_base"]),
                                 _i(out["cols"]), _d(out["vals"]), _i(out["in_width"]), p64(out["in_base"]),
                                 _i(out["in_slots"]), _i(out["in_rows"]), C.byref(in_total))
    return out


def reorder_host(kind, xyz, tri=None, quad=None):
    """perm[new index] = caller's node id of the library's optional renumbering ("morton" or "rcm")."""
    L = load_library()
    i32, dbl = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    L.femshell_reorder_host.argtypes = [C.c_int32, C.c_int32, dbl, C.c_int32, i32, C.c_int32, i32, i32]
    L.femshell_reorder_host.restype = C.c_int
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    tri = np.zeros((0, 3), np.int32) if tri is None else np.ascontiguousarray(tri, dtype=np.int32)
    quad = np.zeros((0, 4), np.int32) if quad is None else np.ascontiguousarray(quad, dtype=np.int32)
    perm = np.zeros(len(xyz), np.int32)
    rc = L.femshell_reorder_host({"morton": 0, "rcm": 1}[kind], len(xyz), _d(xyz), len(tri), _i(tri) if len(tri) else None,
                                 len(quad), _i(quad) if len(quad) else None, _i(perm))
    if rc:
        raise FemShellError(-1, "femshell_reorder_host: invalid mesh")
    return perm
