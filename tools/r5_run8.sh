set -e
cd /root/repo
export VITTRACK_HIP_LIB=$PWD/gstreamer-vit-tracker_amd/libvittrack_hip_stamps.so
for a in "30 24 128 128 1 3 2" "30 24 128 128 1 4 2" "1 24 128 128 1 1 1" "1 24 128 128 1 3 2" "30 24 768 128 0 3 2"; do echo "one_headconv $a:"; python tools/one_headconv.py $a 20 2>&1 | grep -v amdgpu; done
