// Links libapd_hip.so (the C ABI of include/apd.h).  UNCOMPILED: no Rust toolchain in the build image.
fn main() {
    let dir = std::env::var("APD_LIB_DIR").expect("set APD_LIB_DIR to the directory that holds libapd_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=apd_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=APD_LIB_DIR");
}
