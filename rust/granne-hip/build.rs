// Links libgranne_hip.so the way the reference links its optional BLAS (build.rs:1-6 there):
// one rustc-link-lib line, plus where to find the library. GRANNE_HIP_LIB_DIR overrides the
// in-tree location (granne_amd/lib/, where `python -m granne_amd.build` puts it).
fn main() {
    let dir = std::env::var("GRANNE_HIP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{}/../../granne_amd/lib", here)
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=granne_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=GRANNE_HIP_LIB_DIR");
}
