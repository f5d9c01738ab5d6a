"""Loads the product package (directory name has a hyphen) for the tests."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pkg = importlib.import_module("fem-shell_amd")


def ensure_built():
    """Build libfemshell.so in-tree if it is missing (hipcc cross-compiles without a GPU)."""
    if not os.path.exists(pkg.library_path()):
        pkg.build_library()
    return pkg
