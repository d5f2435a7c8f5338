// libzkhip.so is built by `python -c "import __graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950) into
// <repo>/zkevm-circuits_amd; ZKHIP_LIB_DIR names that directory.
fn main() {
    let dir = std::env::var("ZKHIP_LIB_DIR").expect("set ZKHIP_LIB_DIR to <repo>/zkevm-circuits_amd");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=zkhip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=ZKHIP_LIB_DIR");
}
