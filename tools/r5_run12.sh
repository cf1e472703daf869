set -e
cd /root/repo
for r in 1 2 3; do
python tools/one_headconv.py 30 24 128 128 1 3 2 50 2>&1 | grep -v amdgpu
VITTRACK_HIP_LIB=$PWD/gstreamer-vit-tracker_amd/libvittrack_hip_headil.so python tools/one_headconv.py 30 24 128 128 1 3 2 50 2>&1 | grep -v amdgpu
done
VITTRACK_HIP_LIB=$PWD/gstreamer-vit-tracker_amd/libvittrack_hip_headil.so python tools/one_headconv.py 30 24 768 128 0 3 2 50 2>&1 | grep -v amdgpu
VITTRACK_HIP_LIB=$PWD/gstreamer-vit-tracker_amd/libvittrack_hip_headil.so python tools/one_headconv.py 1 24 128 128 1 1 1 50 2>&1 | grep -v amdgpu
