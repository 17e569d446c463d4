// Link against libpathtrace_amd.so built by `make -C pathtrace_amd/csrc` (gfx950 only).
// PATHTRACE_AMD_DIR = directory that holds the library (default: ../../pathtrace_amd of this repository).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("PATHTRACE_AMD_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../pathtrace_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=pathtrace_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=PATHTRACE_AMD_DIR");
}
