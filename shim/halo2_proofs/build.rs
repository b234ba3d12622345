// Links libzkmi355.so (the C ABI of include/zkmi355.h).  ZKMI355_LIB_DIR = directory that holds it
// (zkevm-circuits_amd/lib of the zkmi355 tree).
fn main() {
    if std::env::var("CARGO_FEATURE_ZKMI355").is_err() {
        return;
    }
    let dir = std::env::var("ZKMI355_LIB_DIR").expect("set ZKMI355_LIB_DIR to the directory holding libzkmi355.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=zkmi355");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=ZKMI355_LIB_DIR");
}
