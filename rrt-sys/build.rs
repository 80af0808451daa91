// Links librrt_hip.so.  RRT_LIB_DIR names the directory that holds it (rust_raytracer_amd/ of the MI355X build);
// the library's own dependencies (the HIP runtime, zlib) are recorded in it and need no mention here.
fn main() {
    println!("cargo:rerun-if-env-changed=RRT_LIB_DIR");
    if let Ok(dir) = std::env::var("RRT_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=rrt_hip");
}
