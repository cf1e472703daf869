// libvittrack_hip.so is built by `python __graft_entry__.py` (hipcc --offload-arch=gfx950); point
// VITTRACK_HIP_DIR at the directory that holds it (gstreamer-vit-tracker_amd/ in the source tree).
fn main() {
    let dir = std::env::var("VITTRACK_HIP_DIR")
        .expect("set VITTRACK_HIP_DIR to the directory of libvittrack_hip.so");
    println!("cargo:rerun-if-env-changed=VITTRACK_HIP_DIR");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=vittrack_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
