"""SURVEY section 5 (race / memory-error detection): the multi-threaded host code of the library -- the symbolic plan, the
renumbering, the host part of the multigrid setup -- is built with -fsanitize=address,undefined and, separately, with
-fsanitize=thread (`make -C fem-shell_amd/csrc san`), and the CPU tests that drive it (tests/test_plan_cpu.py,
tests/test_amg_host.py, including the 1 / 3 / 8-thread digest test) run against both.  CPU only: sanitizers are not
available for GPU code on this pool, and the GPU box never runs this."""
import os
import shutil
import subprocess

import pytest

from tests.helpers.product import ROOT


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_host_code_is_clean_under(kind):
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU box: sanitizer runs belong to the CPU container")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run([os.path.join(ROOT, "tools", "run_sanitizers.sh"), kind], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "sanitizer runs clean" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
