"""The host logic of bench.py, checked without a GPU: the argument defaults the driver relies
on, the table of baseline configurations, the CPU-baseline leg on a small sample, how counter
traffic is tied to the build that is timed, the watchdog around the extra multi-GPU check, and
that the benchmark refuses to run without a GPU (there is no CPU path to fall back on)."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

import psim_testlib as T


def load_bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(T.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_defaults_are_one_gpu_and_the_headline_configuration(monkeypatch):
    bench = load_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    args = bench.parse_args()
    assert args.gpus == 1 and args.config == 3 and args.scaling == "weak"
    assert 1 <= args.steps <= 1000 and 0 <= args.warmup <= 100      # finishes within minutes
    assert 0 < args.cpu_seconds <= 30 and args.sustained_seconds is None   # 6 s, 0 in a rehearsal
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    args = bench.parse_args()                                       # the driver's own flags
    assert (args.gpus, args.steps, args.warmup) == (8, 20, 5)


def test_configurations_are_the_baselines():
    """CONFIGS[k] is configs[k - 1] of BASELINE.json: sizes as written there, the 8-GPU
    configurations divided by 8, the exchange only in the last."""
    bench = load_bench()
    with open(os.path.join(T.ROOT, "BASELINE.json")) as f:
        named = json.load(f)["configs"]
    assert len(named) == len(bench.CONFIGS) == 5
    per_gpu = {k: v[1] for k, v in bench.CONFIGS.items()}
    total = {k: v[2] for k, v in bench.CONFIGS.items()}
    assert named[0].startswith("100k") and per_gpu[1] == 100_000
    assert named[1].startswith("10M") and per_gpu[2] == 10_000_000
    assert named[2].startswith("100M") and per_gpu[3] == 100_000_000
    assert named[3].startswith("1B") and total[4] == 1_000_000_000 and per_gpu[4] * 8 == total[4]
    assert named[4].startswith("50M") and total[5] == 50_000_000 and per_gpu[5] * 8 == total[5]
    assert [v[3] for v in bench.CONFIGS.values()] == ["single"] * 4 + ["multi"]
    assert bench.ALGORITHMIC_BYTES_PER_UPDATE == 48 and bench.HBM_PEAK_GBPS == 8000.0
    for k, v in bench.CONFIGS.items():
        assert v[0].startswith(f"configs[{k - 1}]")


def test_cpu_baseline_leg_on_a_small_sample():
    bench = load_bench()
    out = bench.cpu_baseline(50_000, 0.2)
    assert out["kind"] == "port" and out["unit"] == "particle-updates/s"
    assert out["cores"] == T.load_oracle().psim_oracle_max_threads() >= 1
    assert out["value"] > 1e6 and "50000 particles" in out["sample"]


def test_counter_traffic_is_reported_only_for_the_profiled_build():
    bench = load_bench()
    with open(os.path.join(T.ROOT, "profiles", "pmc_summary.json")) as f:
        s = json.load(f)
    n = int(s["particles"])
    same_library = {"library_sha256": s["library_sha256"], "kernel_sources_sha256": "x"}
    same_sources = {"library_sha256": "x", "kernel_sources_sha256": s["kernel_sources_sha256"]}
    other = {"library_sha256": "x", "kernel_sources_sha256": "y"}
    got = bench.load_traffic(n, same_library)
    assert got["traffic_build_match"] == "library" and got["traffic"] == s["hbm_bytes_per_launch"]
    assert got["traffic_profiled_git_sha"] == s["git_sha"]
    got = bench.load_traffic(n, same_sources)
    assert got["traffic_build_match"] == "kernel sources" and got["traffic"] > 0
    got = bench.load_traffic(n, other)
    assert got["traffic"] is None and got["traffic_build_match"].startswith("none")
    got = bench.load_traffic(n // 2, same_library)                   # another size: not comparable
    assert got["traffic"] is None and "particles" in got["traffic_build_match"]
    # the committed counters say what the design claims: no wasted re-reads
    assert abs(s["hbm_bytes_per_launch"] / s["algorithmic_bytes_per_launch"] - 1.0) < 0.01
    assert s["algorithmic_bytes_per_launch"] == 48 * n


def test_build_identity_hashes_the_library_and_the_kernel_sources():
    bench = load_bench()
    ident = bench.build_identity(T.psim)
    assert len(ident["library_sha256"]) == 64 and len(ident["kernel_sources_sha256"]) == 64
    csrc = os.path.dirname(T.psim.LIBRARY_PATH)
    for name in bench.KERNEL_SOURCES:
        assert os.path.exists(os.path.join(csrc, name)), name


WATCHDOG = """
import importlib.util, os, sys, time
spec = importlib.util.spec_from_file_location("bench", {bench!r})
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
def gave_up(why):
    print("gave up:", why, flush=True)
guard = bench.Watchdog({seconds}, gave_up, 3)
{body}
"""


def run_watchdog(seconds, body):
    code = WATCHDOG.format(bench=os.path.join(T.ROOT, "bench.py"), seconds=seconds, body=body)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)


def test_watchdog_ends_a_check_that_does_not_return_with_a_failure_status():
    r = run_watchdog(0.2, "time.sleep(30)")
    assert r.returncode == 3 and "gave up: no result after" in r.stdout


def test_watchdog_takes_the_launchers_sigterm_for_a_lost_rank():
    r = run_watchdog(30, "import signal; os.kill(os.getpid(), signal.SIGTERM); time.sleep(30)")
    assert r.returncode == 3 and "terminated by the launcher" in r.stdout


def test_a_cancelled_watchdog_leaves_the_process_alone():
    r = run_watchdog(0.2, "guard.cancel(); time.sleep(0.5); print('still here')")
    assert r.returncode == 0 and "still here" in r.stdout and "gave up" not in r.stdout


@pytest.mark.skipif(T.gpu_available(), reason="the machine has a GPU")
def test_without_a_gpu_the_benchmark_refuses_to_run():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), "--steps", "1"],
                       capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode != 0 and "no GPU visible" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
