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


def _canned_detail():
    """Round 5's 20.7 KB record (profiles/r05_bench.json: the line the driver could not parse) in the shape of this round's
    detail record."""
    with open(os.path.join(ROOT, "profiles", "r05_bench.json")) as f:
        d = json.load(f)
    d["config"]["assembly_step"] = "one femshell_assemble: launch + status round trip"
    d["config"]["warmup_step"] = "one femshell_assemble + cg_iters CG iterations"
    d["config"]["time_to_solution_preconditioner"] = "SA multigrid, K cycle, mixed precision"
    return d


def test_bench_line_is_compact_and_complete(tmp_path):
    """the last stdout line is a compact JSON object (< 4 KB) with metric, value, ms_per_step, config,
    roofline, cpu_baseline and the four scalars; everything else goes to bench_detail.json."""
    sys.path.insert(0, ROOT)
    import bench

    d = _canned_detail()
    line = bench.compact_line(d)
    assert len(line) < 4096 and "\n" not in line
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype",
              "config", "roofline", "cpu_baseline", "cg_iters_per_s", "time_to_solution_s", "time_to_solution_iterations",
              "parity_max_rel"):
        assert k in rec, k
    assert rec["value"] == d["value"] and rec["ms_per_step"] == d["ms_per_step"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "ms_per_launch", "algorithmic_bytes_per_launch"):
        assert k in rec["roofline"], k
    assert abs(rec["roofline"]["frac"] - rec["roofline"]["achieved"] / rec["roofline"]["peak"]) < 1e-12
    for k in ("value", "unit", "cores", "cpu_model", "cg_iters_per_s", "kind"):
        assert k in rec["cpu_baseline"], k
    assert rec["cpu_baseline"]["value"] > 0 and len(rec["config"]) <= 10 and "workload" in rec["config"]
    assert rec["time_to_solution_iterations"] == d["time_to_solution"]["iterations"]
    # worst of the K / solver-term figures of the canned record: the cylinder's manufactured solve (2.6e-11)
    assert 1e-16 < rec["parity_max_rel"] < 1e-10
    # strings that grow are cut, numbers stay
    d["config"]["workload"] = "x" * 5000
    assert len(bench.compact_line(d)) < 4096
    # as the process prints it: one line on stdout, the last one, and the detail in its file
    env = dict(os.environ, FEMSHELL_BENCH_DETAIL_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-c", "import json, sys; sys.path.insert(0, %r); import bench; "
                        "bench.emit(json.load(open(%r)))" % (ROOT, os.path.join(ROOT, "profiles", "r05_bench.json"))],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-800:]
    out = r.stdout.splitlines()
    assert len(out) == 1 and len(out[0]) < 4096 and json.loads(out[0])["roofline"]["frac"] > 0
    with open(tmp_path / "bench_detail.json") as f:
        assert "time_to_solution" in json.load(f)
