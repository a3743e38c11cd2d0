// build.rs — only does something with `--features hip`.
// DAACHORSE_AMD_LIB_DIR = directory holding libdaachorse_amd.so (this repository: daachorse_amd/lib);
// ROCM_PATH (default /opt/rocm) for libamdhip64, which the library itself depends on.
fn main() {
    if std::env::var_os("CARGO_FEATURE_HIP").is_none() {
        return;
    }
    let lib = std::env::var("DAACHORSE_AMD_LIB_DIR").expect("set DAACHORSE_AMD_LIB_DIR to the directory of libdaachorse_amd.so");
    let rocm = std::env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    println!("cargo:rustc-link-search=native={lib}");
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=daachorse_amd");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{lib}");
    println!("cargo:rerun-if-env-changed=DAACHORSE_AMD_LIB_DIR");
}
