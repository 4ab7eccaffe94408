// Links libhfgpu.so (built by `python huffman_amd/build.py`, hipcc, gfx950).
// HFGPU_LIB_DIR names the directory that holds it.
fn main() {
    let dir = std::env::var("HFGPU_LIB_DIR")
        .expect("set HFGPU_LIB_DIR to the directory of libhfgpu.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=hfgpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=HFGPU_LIB_DIR");
}
