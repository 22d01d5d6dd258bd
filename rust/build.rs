// build.rs for the reference crate: link libdoomgpu.so (built by `make -C doom-rust-renderer_amd/csrc`).
// DOOMGPU_DIR = directory that holds libdoomgpu.so (default: ../doom-rust-renderer_amd relative to the crate).
fn main() {
    let dir = std::env::var("DOOMGPU_DIR").unwrap_or_else(|_| "../doom-rust-renderer_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=doomgpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=DOOMGPU_DIR");
}
