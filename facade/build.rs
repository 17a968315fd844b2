// Links libbppp_hip.so (built in-tree by `python -c "import __graft_entry__ as g; g.build()"`).  BPPP_LIB_DIR overrides the
// search directory.  Without the `gpu` feature nothing is linked (the fixture emitter runs the reference only).
fn main() {
    if std::env::var_os("CARGO_FEATURE_GPU").is_none() {
        return;
    }
    let dir = std::env::var("BPPP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{here}/../bp_pp_amd")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=bppp_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=BPPP_LIB_DIR");
}
