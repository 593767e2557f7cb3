import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
psim = importlib.import_module("particle-simulation-3d_amd")
n = 100_000_000
sim = psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM)
ref = sim.checksum()
for b in (64, 128, 192, 256, 320, 384, 512, 768):
    psim.debug_set_tuning("checksum_blocks", b)
    assert sim.checksum() == ref
    t0 = time.perf_counter()
    for _ in range(10): sim.checksum()
    dt = (time.perf_counter() - t0) / 10
    print(f"blocks {b:6d}: {dt*1e3:.4f} ms  {24*n/dt/1e9:.0f} GB/s", flush=True)
