"""The Rust side of the boundary (rust/psim-sys, rust/hip_backend.rs) cannot be compiled here
(no Rust toolchain in this environment), so it is checked as text: the generator reproduces
the committed bindings, and -- parsed on their own, without the generator's code -- the
`extern "C"` declarations have the names, argument counts and pointer / scalar kinds of the
prototypes in include/psim.h and of the symbols the built library exports."""
import os
import re
import subprocess
import sys

import psim_testlib as T
from psim_testlib import psim

HEADER = os.path.join(T.ROOT, "include", "psim.h")
LIB_RS = os.path.join(T.ROOT, "rust", "psim-sys", "src", "lib.rs")
BACKEND = os.path.join(T.ROOT, "rust", "hip_backend.rs")


def header_prototypes():
    """name -> list of 'ptr' / 'int' / 'float' per argument, straight from the C header."""
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+;", "", text, flags=re.S)
    protos = {}
    for name, args in re.findall(r"\b(psim_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = " ".join(args.split())
        kinds = []
        if args and args != "void":
            for a in args.split(","):
                kinds.append("ptr" if ("*" in a or "[" in a) else
                             ("float" if re.search(r"\bfloat\b", a) else "int"))
        protos[name] = kinds
    return protos


def rust_declarations():
    text = open(LIB_RS).read()
    block = text[text.index('extern "C" {'):]
    decls = {}
    for name, args in re.findall(r"pub fn (psim_[a-z0-9_]+)\(([^)]*)\)", block):
        kinds = []
        for a in filter(None, (x.strip() for x in args.split(","))):
            ty = a.split(":", 1)[1].strip()
            kinds.append("ptr" if ty.startswith("*") else ("float" if ty in ("f32", "f64") else "int"))
        decls[name] = kinds
    return decls


def test_generator_reproduces_the_committed_bindings():
    r = subprocess.run([sys.executable, os.path.join(T.ROOT, "scripts", "gen_rust_sys.py"), "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_declarations_match_the_header():
    protos, decls = header_prototypes(), rust_declarations()
    assert sorted(decls) == sorted(protos)
    for name, kinds in protos.items():
        assert decls[name] == kinds, f"{name}: header {kinds}, lib.rs {decls[name]}"


def test_declarations_match_the_exported_symbols():
    out = subprocess.check_output(["nm", "-D", "--defined-only", psim.LIBRARY_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert {s for s in exported if s.startswith("psim_")} == set(rust_declarations())


def test_records_and_constants():
    text = open(LIB_RS).read()
    for size, name in ((48, "PsimSimParams"), (64, "PsimParticle"), (32, "PsimAttractor"),
                       (64, "PsimCheckpointHeader")):
        assert f"size_of::<{name}>() == {size}" in text
    header = open(HEADER).read()
    for name, value in re.findall(r"(PSIM_[A-Z_]+)\s*=\s*(\d+)", header):
        assert re.search(rf"pub const {name}: c_int = {value};", text), name
    assert "pub const PSIM_MAX_ATTRACTORS: u32 = 64;" in text
    assert "pub const PSIM_EXCHANGE_ID_BYTES: u32 = 128;" in text
    assert "pub const PSIM_MAX_PARTICLES: u32 = 0xFFFFFC00;" in text


def test_backend_uses_declared_entry_points_only_and_covers_the_trait():
    """rust/hip_backend.rs implements every method of `trait ParticleSimulation`
    (src/simulation/mod.rs:21-50) and calls nothing psim-sys does not declare."""
    text = re.sub(r"//.*", "", open(BACKEND).read())           # code only
    decls = rust_declarations()
    used = set(re.findall(r"\b(psim_[a-z0-9_]+)\s*\(", text))
    assert used and used <= set(decls), sorted(used - set(decls))
    for method in ("fn new(", "fn update(", "fn resize_buffer(", "fn get_particle_buffer(",
                   "fn get_method(", "fn get_particle_count(", "fn reset(", "fn is_paused(",
                   "fn set_paused("):
        assert method in text, method
    # argument counts of the calls
    for name, args in re.findall(r"\b(psim_[a-z0-9_]+)\s*\(((?:[^()]|\([^()]*\))*)\)", text):
        depth, count, cur = 0, 0, ""
        for ch in args:
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
            if ch == "," and depth == 0:
                count += 1 if cur.strip() else 0
                cur = ""
            else:
                cur += ch
        count += 1 if cur.strip() else 0
        assert count == len(decls[name]), f"{name} called with {count} arguments"
