// Links libcapgpu.so (built by `make -C cap_amd/csrc`): CAPGPU_LIB_DIR names the directory that holds it.
fn main() {
    println!("cargo:rerun-if-env-changed=CAPGPU_LIB_DIR");
    if let Ok(dir) = std::env::var("CAPGPU_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=capgpu");
}
