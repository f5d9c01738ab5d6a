"""Does the first kernel after an idle gap run slow, or the first kernel after a host-to-device copy?  (lab probe;
run under rocprofv3 --kernel-trace: tools/lab/idle_prof.sh)"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module("fem-shell_amd")
from tests.helpers import meshes
m = meshes.structured(1414, 1414, 0, 0, 10, 10, kind="t", ul_lr=True, bcids=(0, 0, 0, 0), factor=300.0, loading=2)
fs = pkg.FemShell(0.3, 1e7, 0.5)
fs.set_mesh(m.xyz, m.tri); fs.set_dirichlet(m.dirichlet_mask()); fs.set_loads(m.loads)
for _ in range(30):
    fs.assemble()
fs.sync()
for gap in (0.0, 0.02, 0.1, 0.5):
    time.sleep(gap)
    fs.assemble(); fs.assemble(); fs.sync()      # two launches after an idle gap of `gap` seconds
time.sleep(0.2)
fs.set_loads(m.loads)                              # a 96 MB upload from pageable memory, then two launches
fs.assemble(); fs.assemble(); fs.sync()
