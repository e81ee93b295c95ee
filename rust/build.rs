// build.rs -- link libsemtools_hip.so when the crate is built with `--features hip`.
// SEMTOOLS_HIP_LIB_DIR = directory holding libsemtools_hip.so (semtools_amd/lib in the semtools-hip repository).
fn main() {
    if std::env::var_os("CARGO_FEATURE_HIP").is_none() {
        return;
    }
    let dir = std::env::var("SEMTOOLS_HIP_LIB_DIR").expect("set SEMTOOLS_HIP_LIB_DIR to the directory of libsemtools_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=semtools_hip"); // needs libamdhip64.so.7 (ROCm 7) at run time
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=SEMTOOLS_HIP_LIB_DIR");
}
