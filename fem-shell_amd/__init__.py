"""fem-shell_amd -- MI355X (gfx950) implementation of fem-shell's hot path.

The product is the C-ABI library ``libfemshell.so`` (sources in ``csrc/``, interface in
``include/femshell.h``): per-element flat-shell stiffness assembly and the block-Jacobi CG
solve for nodal displacements, as hand-written HIP kernels.  ``FemShell`` below is a thin
ctypes mirror of that C ABI for the Python tests and the benchmark; it adds no arithmetic
and has no fallback -- if the library or a GPU is missing, calls raise.

The directory name contains a hyphen; import it with
``importlib.import_module("fem-shell_amd")``.
"""
from .binding import (  # noqa: F401
    FemShell,
    FemShellError,
    KERNEL_ASSEMBLE,
    KERNEL_CG_DIRECTION,
    KERNEL_CG_UPDATE,
    KERNEL_SPMV,
    REASSEMBLE_EACH_SOLVE,
    REF_DEFAULT,
    REF_DRILL_MAX,
    REF_Y21,
    REORDER_MORTON,
    REORDER_RCM,
    build_library,
    comm_unique_id,
    library_path,
    load_library,
    build_plan,
    plan_node_normals,
    reorder_host,
)
