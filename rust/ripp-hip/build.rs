fn main() {
    if std::env::var("CARGO_FEATURE_FFI").is_ok() {
        if let Ok(dir) = std::env::var("RIPP_HIP_LIB_DIR") {
            println!("cargo:rustc-link-search=native={dir}");
            println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
        }
        println!("cargo:rustc-link-lib=dylib=ripp_hip");
    }
    println!("cargo:rerun-if-env-changed=RIPP_HIP_LIB_DIR");
}
