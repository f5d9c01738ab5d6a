"""Row-partitioned multigrid over the test transport (ranks share the box's GPU): iterations and solutions against one rank.
python tools/lab/dist_amg_gpu_probe.py [kind ...]"""
import os, sys, tempfile, pathlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_multirank_gpu import run_ranks

kinds = sys.argv[1:] or ["panel", "cylinder"]
for kind in kinds:
    with tempfile.TemporaryDirectory() as td:
        td = pathlib.Path(td)
        (td / "one").mkdir()
        single = run_ranks(1, kind, td / "one", pc="amg")[0]
        print(kind, "single: iterations", int(single["iterations"]), "levels", int(single["levels"]), flush=True)
        for dist_min in ("60000", "100"):
            os.environ["FEMSHELL_AMG_DIST_MIN"] = dist_min
            for world in (2, 3, 4):
                d = td / ("w%d_%s" % (world, dist_min))
                d.mkdir()
                try:
                    ranks = run_ranks(world, kind, d, pc="amg")
                except Exception as ex:  # noqa: BLE001
                    print(kind, "world", world, "dist_min", dist_min, "FAILED", str(ex)[-1500:], flush=True)
                    continue
                err = np.linalg.norm(ranks[0]["u"] - single["u"]) / np.linalg.norm(single["u"])
                print(kind, "world", world, "dist_min", dist_min, "iterations", [int(r["iterations"]) for r in ranks], "levels",
                      int(ranks[0]["levels"]), "converged", [int(r["converged"]) for r in ranks], "error vs single %.2e" % err, flush=True)
