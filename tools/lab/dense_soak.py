"""The look-ahead of the dense inverse, again and again: N contexts one after the other (half of them kept alive, so that streams pile
up in the process), each with a coarsest operator of 7.8k dofs -- times of the inverse, iteration counts, solutions.
python tools/lab/dense_soak.py [N=100]"""
import importlib
import sys

import numpy as np

sys.path.insert(0, ".")
from tests.helpers import meshes  # noqa: E402

pkg = importlib.import_module("fem-shell_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
m = meshes.structured(105, 105, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
keep, ms, its, ref = [], [], [], None
for k in range(N):
    fs = pkg.FemShell(0.3, 1e7, 0.5)
    fs.set_mesh(m.xyz, m.tri, m.quad)
    fs.set_dirichlet(m.dirichlet_mask())
    fs.set_loads(m.loads)
    fs.set_preconditioner("amg")
    u, info = fs.solve(rtol=1e-10, max_it=300)
    assert info["converged"] == 1
    ms.append(fs.amg_dense_stats()["ms"])
    its.append(info["iterations"])
    if ref is None:
        ref = u
    else:
        assert np.array_equal(u, ref), "solution differs from the first run's in run %d" % k
    if k % 2 == 0 and len(keep) < 12:
        keep.append(fs)
    else:
        fs.close()
ms = np.array(ms)
print("%d inverses: %.2f / %.2f / %.2f ms (min / median / max), %d above 1.3 x the median; iterations %s; every solution bitwise the first"
      % (N, ms.min(), np.median(ms), ms.max(), int((ms > 1.3 * np.median(ms)).sum()), sorted(set(its))))
