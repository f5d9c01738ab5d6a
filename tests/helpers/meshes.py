"""Mesh fixtures of the tests: the Python mirror of the reference's meshGen and XDA/_f readers lives in the
product package (fem-shell_amd/meshgen.py, also used by bench.py); re-exported here under its old name."""
import importlib as _importlib

_m = _importlib.import_module("fem-shell_amd.meshgen")
globals().update({k: getattr(_m, k) for k in dir(_m) if not k.startswith("__")})
