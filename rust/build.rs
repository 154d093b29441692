// Link against libqn_hip.so.  QN_HIP_LIB_DIR overrides the in-tree location (../optimization-solvers_amd/lib, where
// `make -C optimization-solvers_amd/csrc` and __graft_entry__.build() put it).  NOT COMPILED in the build image (no cargo).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("QN_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("optimization-solvers_amd").join("lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=qn_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=QN_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/qn_hip.h");
}
