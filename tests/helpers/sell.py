"""Numpy model of the sliced-block-ELL layout (fem-shell_amd/csrc/plan.hpp), used by the
CPU tests to check the plan's gather lists and partition with the oracle's arithmetic."""
import numpy as np

SLICE = 32


def assemble_from_plan(plan, mat, oracle, dmask_local=None):
    """Row-owner gather on the CPU: every block slot sums the oracle's element node blocks
    listed in its gather list.  Returns dict slot -> (row_local, col_local, 6x6)."""
    xyz = plan["xyz_local"].reshape(-1, 3)
    tri = plan["tri_local"].reshape(-1, 3)
    quad = plan["quad_local"].reshape(-1, 4)
    n_ltri = plan["n_ltri"]
    cache = {}

    def elem_blocks(le):
        if le not in cache:
            if le < n_ltri:
                _, parts = oracle.element_tri3(xyz[tri[le]], mat, want_parts=True)
                cache[le] = (parts["K_global_nm"], 3)
            else:
                _, parts = oracle.element_quad4(xyz[quad[le - n_ltri]], mat, want_parts=True)
                cache[le] = (parts["K_global_nm"], 4)
        return cache[le]

    out = {}
    for s in range(plan["n_slices"]):
        base = int(plan["slice_base"][s])
        for k in range(int(plan["slice_width"][s])):
            for n in range(SLICE):
                slot = base + k * SLICE + n
                p0, p1 = plan["pair_ptr"][slot], plan["pair_ptr"][slot + 1]
                if p1 == p0:
                    continue
                row = s * SLICE + n
                col = int(plan["cols"][slot])
                blk = np.zeros((6, 6))
                for pr in plan["pairs"][p0:p1]:
                    le, ia, ib = int(pr >> 4), int((pr >> 2) & 3), int(pr & 3)
                    Kg, _ = elem_blocks(le)
                    blk += Kg[6 * ia:6 * ia + 6, 6 * ib:6 * ib + 6]
                if dmask_local is not None:
                    mr, mc = int(dmask_local[row]), int(dmask_local[col])
                    for i in range(6):
                        for j in range(6):
                            if (mr >> i) & 1 or (mc >> j) & 1:
                                blk[i, j] = 0.0
                    if row == col:
                        for i in range(6):
                            if (mr >> i) & 1:
                                blk[i, i] = float(p1 - p0)
                out[slot] = (row, col, blk)
    if plan.get("symmetric"):
        # symmetric storage: of a pair of owned nodes only one row holds the block; the other one is
        # its transpose (keys beyond the slot range)
        nxt = int(plan["slice_base"][-1])
        for slot, (row, col, blk) in list(out.items()):
            if col != row and col < plan["n_own"]:
                out[nxt] = (col, row, blk.T.copy())
                nxt += 1
    return out


def to_global_id(plan, local):
    if local < plan["n_pad"]:
        return plan["row_begin"] + local
    return int(plan["ghost_global"][local - plan["n_pad"]])
