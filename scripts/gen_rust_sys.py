#!/usr/bin/env python3
"""Derives rust/psim-sys/src/lib.rs from include/psim.h: one `extern "C"` declaration per
function prototype of the header, the records and constants beside them.  The Rust is not
compiled in this repository's environment (no Rust toolchain); tests/test_rust_shim.py runs
this generator again and compares, and checks names and arity against the header on its own.

usage: gen_rust_sys.py [--check]     (--check: exit 1 when the committed file differs)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "psim.h")
OUT = os.path.join(ROOT, "rust", "psim-sys", "src", "lib.rs")

OPAQUE = {"psim_handle": "PsimHandle", "psim_group": "PsimGroup", "psim_exchange": "PsimExchange"}
RECORDS = {"psim_sim_params": "PsimSimParams", "psim_particle": "PsimParticle",
           "psim_attractor": "PsimAttractor", "psim_soa_view": "PsimSoaView",
           "psim_checkpoint_header": "PsimCheckpointHeader",
           "psim_launch_shape_info": "PsimLaunchShapeInfo"}
SCALARS = {"int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "float": "f32",
           "char": "c_char", "void": "c_void"}

PRELUDE = '''//! Raw bindings of `libpsim_hip.so`, the MI355X (gfx950) backend of the particle update.
//!
//! GENERATED from `include/psim.h` by `scripts/gen_rust_sys.py`; do not edit by hand.
//! One declaration per prototype of the header, with the same names.  The records are the
//! reference's own `#[repr(C)]` layouts (`src/simulation/mod.rs:52-67, 86-98`), so
//! `&SimParams` / `&[Particle]` of the application cross the boundary by pointer cast.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

macro_rules! opaque {
    ($name:ident) => {
        #[repr(C)]
        pub struct $name {
            _private: [u8; 0],
        }
    };
}
opaque!(PsimHandle);
opaque!(PsimGroup);
opaque!(PsimExchange);

/// `struct SimParams`, 48 bytes (`src/simulation/mod.rs:52-67`).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct PsimSimParams {
    pub delta_time: f32,
    pub gravity: f32,
    pub color_mode: u32,
    pub mouse_force: f32,
    pub mouse_radius: f32,
    pub is_mouse_dragging: u32,
    pub damping: f32,
    pub max_dist_for_color: f32,
    pub mouse_position: [f32; 3],
    pub _padding2: u32,
}

/// `struct Particle`, 64 bytes (`src/simulation/mod.rs:86-98`).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct PsimParticle {
    pub position: [f32; 3],
    pub padding1: f32,
    pub velocity: [f32; 3],
    pub padding2: f32,
    pub color: [f32; 4],
    pub initial_color: [f32; 4],
}

/// One attractor of the multi-attractor extension, 32 bytes.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct PsimAttractor {
    pub position: [f32; 3],
    pub force: f32,
    pub radius: f32,
    pub active: u32,
    pub _pad: [u32; 2],
}

/// Raw device pointers of a handle's structure-of-arrays storage.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PsimSoaView {
    pub px: *mut f32,
    pub py: *mut f32,
    pub pz: *mut f32,
    pub vx: *mut f32,
    pub vy: *mut f32,
    pub vz: *mut f32,
    pub color: *mut f32,
    pub initial_color: *mut f32,
    pub padding: *mut f32,
    pub count: u32,
    pub capacity: u32,
}

/// The 64 bytes at the start of a checkpoint file.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PsimCheckpointHeader {
    pub magic: [c_char; 8],
    pub version: u32,
    pub record_bytes: u32,
    pub particle_count: u64,
    pub step: u64,
    pub generation_mode: u32,
    pub color_semantics: u32,
    pub reserved: [u8; 24],
}

/// The launch shape the library picks for a particle count.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PsimLaunchShapeInfo {
    pub access: c_int,
    pub workgroup_size: u32,
    pub workgroups: u32,
    pub reserve_lds: u32,
    pub reserve_lds_colour: u32,
    pub resident_per_cu: u32,
}

const _: () = assert!(std::mem::size_of::<PsimSimParams>() == 48);
const _: () = assert!(std::mem::size_of::<PsimParticle>() == 64);
const _: () = assert!(std::mem::size_of::<PsimAttractor>() == 32);
const _: () = assert!(std::mem::size_of::<PsimCheckpointHeader>() == 64);
'''


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def constants(text):
    out = []
    for name, value in re.findall(r"^#define\s+(PSIM_[A-Z_]+)\s+(0x[0-9A-Fa-f]+u?|\d+u?)\s*$",
                                  text, flags=re.M):
        v = value.rstrip("u")
        n = int(v, 0)
        out.append(f"pub const {name}: {'u32' if n <= 0xFFFFFFFF else 'u64'} = {v};")
    for body in re.findall(r"enum\s*\{(.*?)\};", text, flags=re.S):
        for name, value in re.findall(r"(PSIM_[A-Z_]+)\s*=\s*(\d+)", body):
            out.append(f"pub const {name}: c_int = {value};")
    return out


def prototypes(text):
    """(return type, name, [(type, name)]) of every function prototype."""
    body = text[text.index('extern "C" {'):]
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+;", "", body, flags=re.S)
    body = re.sub(r"enum\s*\{.*?\};", "", body, flags=re.S)
    out = []
    for ret, name, args in re.findall(r"^([A-Za-z_][\w \*]*?)\s*\b(psim_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;",
                                      body, flags=re.M | re.S):
        args = " ".join(args.split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                m = re.match(r"^(.*?)(\w+)(\[[^\]]*\])?$", a)
                ctype, pname, array = m.group(1).strip(), m.group(2), m.group(3)
                if array:
                    ctype += " *"
                params.append((ctype, pname))
        out.append((" ".join(ret.split()), name, params))
    return out


def rust_type(ctype):
    ctype = ctype.strip()
    stars = ctype.count("*")
    base = ctype.replace("*", " ").split()
    const = "const" in base
    base = [b for b in base if b != "const"]
    assert len(base) == 1, ctype
    name = base[0]
    rust = SCALARS.get(name) or OPAQUE.get(name) or RECORDS.get(name)
    assert rust, f"unknown C type {ctype!r}"
    if stars == 0:
        return rust
    # the innermost level carries the const of `const T *`; outer levels are `*mut`
    inner = ("*const " if const else "*mut ") + rust
    return "*mut " * (stars - 1) + inner


def generate():
    text = strip_comments(open(HEADER).read())
    lines = [PRELUDE, "// ---- constants -------------------------------------------------"]
    lines += constants(text)
    lines += ["", '#[link(name = "psim_hip")]', 'extern "C" {']
    for ret, name, params in prototypes(text):
        args = ", ".join(f"{'r#type' if p == 'type' else p}: {rust_type(t)}" for t, p in params)
        arrow = "" if ret == "void" else f" -> {rust_type(ret)}"
        lines.append(f"    pub fn {name}({args}){arrow};")
    lines += ["}", ""]
    return "\n".join(lines)


def main():
    new = generate()
    if "--check" in sys.argv:
        old = open(OUT).read() if os.path.exists(OUT) else ""
        if old != new:
            print(f"{OUT} is out of date: run scripts/gen_rust_sys.py")
            return 1
        return 0
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(new)
    print(f"wrote {OUT}: {len(prototypes(strip_comments(open(HEADER).read())))} functions")
    return 0


if __name__ == "__main__":
    sys.exit(main())
