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
                        "--warmup", "2", "--particles", "2000000",
                        *(() if "--sustained-seconds" in extra else ("--sustained-seconds", "0")),
                        *extra],
                       capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must hold exactly one line, got {len(lines)}: {r.stdout[:500]}"
    return json.loads(lines[0])


def test_single_workload_line():
    out = run_bench("--cpu-sample", "1000000", "--cpu-seconds", "1", "--sustained-seconds", "0.5")
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
    held = out["sustained"]
    assert held["steps"] >= 5 and 0.2 < held["seconds"] < 5.0 and held["unit"] == out["unit"]
    assert abs(held["value"] - 2_000_000 * held["steps"] / held["seconds"]) / held["value"] < 1e-6
    assert 0 < held["launch_ms"] <= held["ms_per_step"] * 1.001
    copy = roof["device_copy"]
    assert copy["bytes"] == 48 * 2_000_000 and copy["unit"] == "GB/s"
    assert abs(copy["achieved"] - copy["bytes"] / (copy["ms"] / 1e3) / 1e9) < 1e-6 * copy["achieved"]
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
                        "--no-cpu-baseline", "--sustained-seconds", "0.3"],
                       capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[:1000]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["total_particles"] == n
    assert out["value"] > 0 and out["roofline"]["achieved"] > 0
    assert out["sustained"]["value"] > 0          # its reduction over the ranks ran on RCCL too


@pytest.mark.parametrize("pacing", ["host", "device", "pipelined"])
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
           "--config", str(config), "--no-cpu-baseline", "--sustained-seconds", "0"]
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
    assert check["us_per_step_device_host_paced"] > 0 and check["us_per_step_device_pipelined"] > 0
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


def test_config_5_two_ranks_hold_the_step_for_the_same_number_of_steps():
    """The step held after the timed steps (`sustained`) on several ranks of --config 5, where
    every step is a collective: all ranks must take the same number of steps -- derived from
    the reduced time, not from each rank's own -- or they would wait for each other for ever."""
    r, lines = launch_ranks(2, "--config", "5", "--steps", "20", "--warmup", "5", "--particles",
                            "500000", "--sustained-seconds", "0.4", port=29599)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[:1000]
    held = json.loads(lines[0])["sustained"]
    assert held["steps"] >= 20 and 0.05 < held["seconds"] < 30.0
    assert abs(held["value"] - 1_000_000 * held["steps"] / held["seconds"]) / held["value"] < 1e-6


# ---- readiness for a multi-GPU node: worlds of 4 and 8 rehearsed on this machine's GPU --------

def launch_ranks(world, *args, port, env_extra=None, timeout=280):
    """bench.py under torch.distributed.run exactly as the driver launches it, `world` ranks
    sharing the GPUs that exist (PSIM_BENCH_REHEARSAL: gloo between the ranks, the shared-memory
    test double under the library's exchange)."""
    cpp_dir = os.path.join(T.ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-C", cpp_dir, "libfake_rccl.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, PSIM_BENCH_REHEARSAL="1",
               PSIM_RCCL_LIBRARY=os.path.join(cpp_dir, "libfake_rccl.so"), OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(T.ROOT, "bench.py"),
                        "--gpus", str(world), *args],
                       capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    return r, lines


@pytest.mark.parametrize("world", [4, 8])
def test_weak_scaling_line_and_exchange_check_at_four_and_eight_ranks(world):
    """--config 3 (reduced per-GPU count) as `--gpus 4` / `--gpus 8`: the whole-job value over
    all ranks, one line from rank 0, and the appended exchange check -- the 50 M-particle
    configuration's own step, 6.25 M particles on every rank with `world` attractors --
    delivering the closed-form lists on every rank."""
    n = 1_000_000
    r, lines = launch_ranks(world, "--config", "3", "--steps", "5", "--warmup", "2",
                            "--particles", str(n), port=29551 + world)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[:1000]
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["scaling"] == "weak" and "cpu_baseline" not in out
    assert out["config"]["total_particles"] == world * n
    assert abs(out["value"] - world * n * 5 / (out["ms_per_step"] * 5e-3)) / out["value"] < 1e-6
    assert out["per_rank_ms_per_step"]["max"] * (1 + 1e-9) >= out["ms_per_step"] * 0.5
    check = out["exchange_check"]
    assert "error" not in check, check
    assert check["lists_match_closed_form"] is True
    assert f"{world} moving attractors" in check["workload"]


@pytest.mark.parametrize("world", [4, 8])
def test_billion_particle_configuration_line_at_four_and_eight_ranks(world):
    """--config 4 names 125 M particles per GPU; rehearsed at a reduced per-GPU count, every
    rank a shard at its own offset of the one global array."""
    n = 2_000_000
    r, lines = launch_ranks(world, "--config", "4", "--steps", "5", "--warmup", "2",
                            "--particles", str(n), "--no-exchange-check", port=29561 + world)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(lines[0])
    assert len(lines) == 1 and out["config"]["baseline_config"] == 4 and out["n_gpus"] == world
    assert out["config"]["total_particles"] == world * n and "exchange_check" not in out
    assert "no data-path collective" in out["config"]["sharding"]


@pytest.mark.parametrize("world,pacing", [(4, "host"), (8, "host"), (8, "device"), (8, "pipelined")])
def test_multi_attractor_configuration_line_at_four_and_eight_ranks(world, pacing):
    """--config 5 at its own per-GPU size (6.25 M particles) on 4 and 8 ranks: every rank
    steps its shard through psim_exchange_step."""
    r, lines = launch_ranks(world, "--config", "5", "--steps", "40", "--warmup", "5",
                            "--pacing", pacing, port=29571 + world)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[:1000]
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["config"]["baseline_config"] == 5
    assert out["config"]["particles_per_gpu"] == 6_250_000
    assert out["config"]["total_particles"] == world * 6_250_000
    assert f"{world} moving attractors" in out["config"]["workload"]
    assert f"{pacing}-paced" in out["config"]["sharding"]


@pytest.mark.parametrize("world", [4, 8])
def test_strong_scaling_line_at_four_and_eight_ranks(world):
    """--scaling strong: a fixed total that does not divide evenly, over 4 and 8 ranks."""
    total = 8_000_003
    r, lines = launch_ranks(world, "--scaling", "strong", "--steps", "5", "--warmup", "2",
                            "--particles", str(total), "--no-exchange-check", port=29581 + world)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(lines[0])
    assert len(lines) == 1 and out["scaling"] == "strong" and out["n_gpus"] == world
    assert out["config"]["total_particles"] == total
    assert abs(out["value"] - total * 5 / (out["ms_per_step"] * 5e-3)) / out["value"] < 1e-6


def test_an_abandoned_exchange_check_is_reported_and_fails_the_run():
    """The check after the headline may hang or fail on a machine where the headline ran: the
    line must still go out (with the error in it), and the run must NOT report success.  Here
    the watchdog is given 50 ms, less than the check needs to set itself up."""
    r, lines = launch_ranks(2, "--steps", "5", "--warmup", "2", "--particles", "1000000",
                            port=29591, env_extra={"PSIM_BENCH_EXCHANGE_CHECK_SECONDS": "0.05"})
    assert r.returncode != 0, "an abandoned check must not exit with status 0"
    assert len(lines) == 1, r.stdout[:1000]
    out = json.loads(lines[0])
    assert out["value"] > 0 and out["n_gpus"] == 2          # the headline is intact
    assert "abandoned" in out["exchange_check"]["error"]
    assert "exchange check abandoned" in r.stderr           # every rank leaves a trace


def test_one_rank_under_the_launcher_measures_what_the_plain_run_measures():
    """`python bench.py` and the driver's launch line with one rank (process group over RCCL,
    device barriers) at the headline size: the same workload, and rates within noise."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PSIM_BENCH_REHEARSAL"):
        env.pop(k, None)
    args = ["--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--sustained-seconds", "0"]
    plain = subprocess.run([sys.executable, os.path.join(T.ROOT, "bench.py"), *args],
                           capture_output=True, text=True, timeout=280, env=env)
    assert plain.returncode == 0, plain.stderr[-3000:]
    under = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                            "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                            "--master-port", "29593", os.path.join(T.ROOT, "bench.py"),
                            "--gpus", "1", *args],
                           capture_output=True, text=True, timeout=280, env=env)
    assert under.returncode == 0, under.stderr[-3000:]
    a = json.loads([ln for ln in plain.stdout.splitlines() if ln.strip().startswith("{")][0])
    b = json.loads([ln for ln in under.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert a["config"] == b["config"] and a["config"]["particles_per_gpu"] == 100_000_000
    assert abs(a["value"] - b["value"]) / a["value"] < 0.05, (a["value"], b["value"])
    assert abs(a["roofline"]["launch_ms"] - b["roofline"]["launch_ms"]) / a["roofline"]["launch_ms"] < 0.05
    print(f"plain {a['value']:.4g} updates/s, under the launcher {b['value']:.4g}")
