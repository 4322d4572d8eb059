// Links libtinyorb.so (built by `python -m tinyslam_amd.build`).  TINYORB_LIB_DIR = directory holding it.
fn main() {
    if let Ok(dir) = std::env::var("TINYORB_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=tinyorb");
    println!("cargo:rerun-if-env-changed=TINYORB_LIB_DIR");
}
