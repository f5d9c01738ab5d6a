"""Probe: libfemshell's RCCL path inside a process that imported torch first (bench.py's situation: the
loader then resolves librccl.so.1 / libamdhip64.so.7 to the copies bundled with torch)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first, like bench.py)
os.environ["FEMSHELL_FORCE_COMM"] = "1"
from tests.helpers import meshes
pkg = importlib.import_module("fem-shell_amd")
torch.cuda.set_device(0)
m = meshes.load_example("test_C_w_tA16")
fs = pkg.FemShell(0.3, 10.92, 1.0, device=0, rank=0, world_size=1)
fs.comm_init(pkg.comm_unique_id())
fs.set_mesh(m.xyz, m.tri, m.quad); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
u, info = fs.solve(rtol=1e-12, max_it=20000)
print("torch", torch.__version__, "w(144) =", u[144, 2], info["iterations"], "iterations through a 1-rank RCCL communicator")
maps = open("/proc/self/maps").read()
print([l.split()[-1] for l in maps.splitlines() if "librccl" in l or "libamdhip64" in l][:4])
