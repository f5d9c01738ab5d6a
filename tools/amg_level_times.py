#!/usr/bin/env python3
"""Per-level milliseconds of one outer iteration of the multigrid-preconditioned solve, from the by-(kernel, grid) summary of a
rocprofv3 kernel trace (tools/kernel_trace_by_grid.py) and the probe's own output (tools/amg_probe.py: level sizes, iterations):
    amg_level_times.py <kernels_by_level.txt> <probe.txt> <out.json>
A launch belongs to the level whose slice count its grid matches (per-slice kernels: 192 threads per slice; node kernels and
k_spmv_sym: 64 threads per pair of slices; vector passes: by their length); the setup kernels and what cannot be told (single-
workgroup scalar steps) are listed as unassigned.  A restriction is a product over the rows of the COARSER level and is booked there
(as femshell_amg_cycle_bytes books its bytes).  bench.py attaches the result as roofline_amg_iteration.by_level_ms while the
kernel sources are the ones the trace was taken with."""
import hashlib
import json
import os
import re
import sys


def kernel_source_digest():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "fem-shell_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".h", ".cpp")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    table, probe, out = sys.argv[1:4]
    text = open(probe).read()
    info = json.loads(re.search(r"^\{.*\}$", text, re.M).group(0))
    nodes = [int(x) for x in re.findall(r"'n_nodes': (\d+)", text)]
    # two solves in the probe (the second reuses the hierarchy) + the setup's power iterations; iterations of ONE solve:
    its = info["iterations"]
    solves = 2
    slices = [(n + 31) // 32 for n in nodes]

    def level_of(name, wgs):
        if name.startswith("k_dense_gemv") or (name.startswith("k_spmv<") and wgs == ((slices[-1] + 7) // 8) * 8):
            return len(nodes) - 1 if name.startswith("k_dense_gemv") else len(nodes) - 2  # (R onto the coarsest level: work of the level above)
        best, err = None, 0.2
        for l, s in enumerate(slices):
            if name.startswith("k_kcyc") and l == 0:
                continue  # (the K cycle's vector passes exist on levels >= 1 only; their grids are capped at 128 / 2048 workgroups)
            for g in (s, (s + 1) // 2, min(6 * 32 * s // 256 + 1, 2048), min(6 * 32 * s // 4096 + 1, 128), min((6 * 32 * s // 2 + 255) // 256, 4096)):
                e = abs(wgs - g) / max(g, 1)
                if e < err:
                    best, err = l, e
        return best

    per_level = [0.0] * len(nodes)
    unassigned, setup = 0.0, 0.0
    detail = []
    for line in open(table).read().splitlines()[2:]:
        m = re.match(r"(\S.*?)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)$", line)
        if not m:
            continue
        name, wgs, calls, total = m.group(1).strip(), int(m.group(2)), int(m.group(3)), float(m.group(4))
        if name.startswith(("k_amg_", "k_dense_update", "k_dense_panels", "k_dense_pivot", "k_dense_finish", "k_dense_scatter", "k_dense_prepare",
                            "k_assemble", "k_item_flags", "k_block_jacobi", "k_to_f32", "k_minv_apply_norm", "k_fill_hash", "__amd", "k_residual_dd")):
            setup += total
            continue
        l = level_of(name, wgs)
        if l is None:
            unassigned += total
        else:
            per_level[l] += total
        detail.append({"kernel": name, "workgroups": wgs, "calls": calls, "total_ms": total, "level": l})
    denom = float(its * solves)
    res = {"kernel_source_digest": kernel_source_digest(), "iterations_per_solve": its, "solves_in_trace": solves, "nodes_by_level": nodes,
           "ms_per_iteration_by_level": [t / denom for t in per_level], "ms_per_iteration_unassigned": unassigned / denom,
           "setup_and_one_off_kernels_ms_total": setup, "kernels": detail}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print("per-level ms per outer iteration:", " ".join("%.3f" % (t / denom) for t in per_level), "| unassigned %.3f" % (unassigned / denom))


if __name__ == "__main__":
    main()
