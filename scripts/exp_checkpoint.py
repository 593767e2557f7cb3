#!/usr/bin/env python3
"""Rate of psim_checkpoint_write / psim_checkpoint_read (GB/s of the file) at a given size;
the file goes to the directory named by the second argument (default: /tmp)."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
psim = importlib.import_module("particle-simulation-3d_amd")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
path = os.path.join(sys.argv[2] if len(sys.argv) > 2 else "/tmp", f"psim_exp_{os.getpid()}.ckp")
sim = psim.HipParticleSimulation.new(n, psim.GEN_UNIFORM)
try:
    for rep in range(3):
        t0 = time.perf_counter()
        sim.save_checkpoint(path, step=rep)
        t1 = time.perf_counter()
        sim.load_checkpoint(path)
        t2 = time.perf_counter()
        gb = 64 * n / 1e9
        print(f"{n} particles ({gb:.2f} GB): write {t1 - t0:.3f} s = {gb / (t1 - t0):.2f} GB/s, "
              f"read {t2 - t1:.3f} s = {gb / (t2 - t1):.2f} GB/s", flush=True)
finally:
    if os.path.exists(path):
        os.unlink(path)
