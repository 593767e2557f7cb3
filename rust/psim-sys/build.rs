// Links libpsim_hip.so (built by `make -C particle-simulation-3d_amd/csrc`).
// PSIM_HIP_LIB_DIR names the directory that holds it; the run-time search path is set to the
// same directory and to ROCm's, where libamdhip64.so lives.
fn main() {
    println!("cargo:rerun-if-env-changed=PSIM_HIP_LIB_DIR");
    let dir = std::env::var("PSIM_HIP_LIB_DIR")
        .expect("set PSIM_HIP_LIB_DIR to the directory that holds libpsim_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=psim_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rustc-link-arg=-Wl,-rpath,/opt/rocm/lib");
}
