// TAPSTARK_LIB_DIR = directory holding libtapstark_hip.so (tap-stark_amd/lib after
// `python -m tapstark_amd.build`, i.e. __graft_entry__.build()).
fn main() {
    let dir = std::env::var("TAPSTARK_LIB_DIR").unwrap_or_else(|_| "../../../tap-stark_amd/lib".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=tapstark_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=TAPSTARK_LIB_DIR");
}
