"""GPU test of bench.py's output contract: exactly one JSON line on stdout with the agreed
fields (run at a reduced size so it takes seconds; the headline run uses the defaults)."""
import json
import os
import subprocess
import sys

import pytest

import psim_testlib as T

pytestmark = pytest.mark.gpu


def run_bench(*extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), "--steps", "5",
                        "--warmup", "2", "--particles", "2000000", *extra],
                       capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must hold exactly one line, got {len(lines)}: {r.stdout[:500]}"
    return json.loads(lines[0])


def test_single_workload_line():
    out = run_bench("--cpu-sample", "1000000", "--cpu-seconds", "1")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 5 and out["warmup"] == 2
    assert out["unit"] == "particle-updates/s" and out["dtype"] == "f32"
    assert out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["vs_baseline"] is None and out["data"] == "synthetic"
    assert "workload" in out["config"] and "model" not in out["config"]
    # value = particles x steps / wall time
    assert abs(out["value"] - 2_000_000 * 5 / (out["ms_per_step"] * 5e-3)) / out["value"] < 1e-6
    roof = out["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert roof["algorithmic_bytes_per_launch"] == 48 * 2_000_000
    assert roof["traffic"] is None            # the committed counters are for the 100 M size
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0
    assert cpu["unit"] == "particle-updates/s" and "particles" in cpu["sample"]


def test_two_rank_control_flow_rehearsal():
    """The N > 1 path of bench.py as the driver launches it (torch.distributed.run, one
    process per rank), rehearsed on one GPU: both ranks share it and talk over gloo.  Checks
    the aggregation (value over all ranks, max-over-ranks time, one line from rank 0), not
    the speed."""
    env = dict(os.environ, PSIM_BENCH_REHEARSAL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    n = 1_000_000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(T.ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "5", "--warmup", "2", "--particles", str(n)],
                       capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[:1000]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "cpu_baseline" not in out
    assert out["config"]["total_particles"] == 2 * n and out["config"]["particles_per_gpu"] == n
    assert abs(out["value"] - 2 * n * 5 / (out["ms_per_step"] * 5e-3)) / out["value"] < 1e-6


def test_one_rank_under_the_launcher_uses_rccl():
    """The driver's launch line with one rank: the process group is RCCL ("nccl"), the
    barriers carry device ids and the max-over-ranks reduction runs on a GPU tensor -- the
    same calls the N > 1 runs make."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PSIM_BENCH_REHEARSAL"):
        env.pop(k, None)
    n = 1_000_000
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29543", os.path.join(T.ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "5", "--warmup", "2", "--particles", str(n),
                        "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[:1000]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["total_particles"] == n
    assert out["value"] > 0 and out["roofline"]["achieved"] > 0


@pytest.mark.parametrize("pacing", ["host", "device"])
def test_multi_workload_line(pacing):
    out = run_bench("--workload", "multi", "--pacing", pacing, "--no-cpu-baseline")
    assert out["n_gpus"] == 1 and "cpu_baseline" not in out
    assert "all-gather" in out["config"]["sharding"]
    assert "psim_exchange_step" in out["config"]["sharding"]
    assert f"{pacing}-paced" in out["config"]["sharding"]
    assert out["value"] > 0 and out["roofline"]["traffic"] is None


@pytest.mark.parametrize("config,per_gpu,workload", [(1, 100_000, "single"),
                                                      (2, 10_000_000, "single"),
                                                      (3, 100_000_000, "single"),
                                                      (4, 125_000_000, "single"),
                                                      (5, 6_250_000, "multi")])
def test_baseline_config_presets(config, per_gpu, workload):
    """--config N names the baseline's configuration N as written.  Run at the preset's own
    size when that takes a moment (1, 2, 5), at a reduced size otherwise; the preset sizes are
    checked against the table either way."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(T.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.CONFIGS[config][1] == per_gpu and bench.CONFIGS[config][3] == workload
    full_size = config in (1, 2, 5)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(T.ROOT, "bench.py"), "--steps", "5", "--warmup", "2",
           "--config", str(config), "--no-cpu-baseline"]
    if not full_size:
        cmd += ["--particles", "3000000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert out["config"]["baseline_config"] == config
    assert out["config"]["workload"].startswith(f"configs[{config - 1}]")
    assert out["config"]["particles_per_gpu"] == (per_gpu if full_size else 3_000_000)
    assert ("all-gather" in out["config"]["sharding"]) == (workload == "multi")
    assert out["per_rank_ms_per_step"]["min"] <= out["per_rank_ms_per_step"]["max"]
    assert out["roofline"]["per_rank_launch_ms"]["max"] == out["roofline"]["launch_ms"]
    assert len(out["roofline"]["library_sha256"]) == 64


def test_strong_scaling_rehearsal():
    """--scaling strong: the total is fixed and divided over the ranks (uneven by one when
    it does not divide).  Two ranks on one GPU over gloo, as in the rehearsal above."""
    env = dict(os.environ, PSIM_BENCH_REHEARSAL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    total = 2_000_001
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29545", os.path.join(T.ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "5", "--warmup", "2", "--scaling", "strong",
                        "--particles", str(total)],
                       capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[:1000]
    out = json.loads(lines[0])
    assert out["scaling"] == "strong" and out["n_gpus"] == 2
    assert out["config"]["total_particles"] == total and out["config"]["particles_per_gpu"] == 1_000_000
    assert abs(out["value"] - total * 5 / (out["ms_per_step"] * 5e-3)) / out["value"] < 1e-6
    assert "exchange_check" not in out               # a rehearsal cannot run RCCL


def test_multi_gpu_exchange_check_rehearsal():
    """What bench.py appends to a multi-GPU run of a single-attractor configuration -- the
    config-5 exchange over all ranks, outside the timed region -- rehearsed with two ranks on
    one GPU: gloo between the ranks and tests/cpp/libfake_rccl.so (a shared-memory test double,
    RCCL refuses two ranks on one device) under the library's exchange.  Checks that the line
    carries the check and that the lists the ranks received are the closed-form paths."""
    cpp_dir = os.path.join(T.ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", cpp_dir, "libfake_rccl.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, PSIM_BENCH_REHEARSAL="1",
               PSIM_RCCL_LIBRARY=os.path.join(cpp_dir, "libfake_rccl.so"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", os.path.join(T.ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "5", "--warmup", "2", "--particles", "1000000"],
                       capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[:1000]
    out = json.loads(lines[0])
    check = out["exchange_check"]
    assert "error" not in check, check
    assert check["lists_match_closed_form"] is True
    assert check["us_per_step_device"] > 0 and check["us_per_step_kernel_only"] > 0
    assert out["n_gpus"] == 2 and out["value"] > 0


def test_config_5_two_rank_rehearsal():
    """bench.py --config 5 as the driver launches it on several GPUs -- every rank steps its
    shard through psim_exchange_step, the communicator id travelling over the process group --
    rehearsed with two ranks on one GPU (gloo between the ranks, the shared-memory test double
    under the exchange)."""
    cpp_dir = os.path.join(T.ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", cpp_dir, "libfake_rccl.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, PSIM_BENCH_REHEARSAL="1",
               PSIM_RCCL_LIBRARY=os.path.join(cpp_dir, "libfake_rccl.so"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    for pacing in ("host", "device"):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                            "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", "29549", os.path.join(T.ROOT, "bench.py"),
                            "--gpus", "2", "--config", "5", "--steps", "40", "--warmup", "5",
                            "--particles", "500000", "--pacing", pacing],
                           capture_output=True, text=True, timeout=280, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
        assert len(lines) == 1, r.stdout[:1000]
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["config"]["baseline_config"] == 5
        assert out["config"]["total_particles"] == 1_000_000
        assert f"{pacing}-paced" in out["config"]["sharding"]
        assert "2 moving attractors" in out["config"]["workload"]
        assert abs(out["value"] - 1_000_000 * 40 / (out["ms_per_step"] * 40e-3)) / out["value"] < 1e-6
