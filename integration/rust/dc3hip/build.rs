// Links the prebuilt libdc3hip.so (built by `make -C stringsearch_amd/csrc` in the dc3hip repository).
// Counterpart of crates/cdivsufsort/build.rs:1-29, which compiles the C sources with the `cc` crate;
// here the native side is HIP, so it is built by hipcc outside cargo and only linked.
fn main() {
    let dir = std::env::var("DC3HIP_LIB_DIR").expect("set DC3HIP_LIB_DIR to the directory holding libdc3hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=dc3hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=DC3HIP_LIB_DIR");
}
