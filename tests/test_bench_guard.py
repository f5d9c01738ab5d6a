"""bench.py --gpus N > 1 must end with a message, never hang: the rank's work runs in a child process and
the parent -- which never touches the GPU -- bounds the wait.  Here: rank 0 of a two-rank run whose second rank never starts."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_parent_ends_a_rank_whose_peer_never_arrives():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571",
               FEMSHELL_BENCH_TIMEOUT="12")
    env.pop("FEMSHELL_BENCH_CHILD", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--nx", "32",
                        "--no-cpu-baseline", "--no-full-parity"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert time.time() - t0 < 90
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout[-500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    # (on a box with a GPU the child waits for its peer in the rendezvous and the parent's limit ends it; without one it
    #  fails on its own a moment earlier -- either way the launcher gets one line that says so and a non-zero status)
    assert d["value"] is None and d["n_gpus"] == 2 and "retry_hint" in d
    assert "no result after 12 s" in d["error"] or "ended with status" in d["error"], d["error"]
