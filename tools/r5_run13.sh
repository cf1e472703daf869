cd /root/repo
for v in "" _noread _nomfma _headil; do
echo "variant '$v'"; VITTRACK_HIP_LIB=$PWD/gstreamer-vit-tracker_amd/libvittrack_hip$v.so python tools/one_headconv.py 30 24 128 128 1 3 2 50 2>&1 | grep -v amdgpu
VITTRACK_HIP_LIB=$PWD/gstreamer-vit-tracker_amd/libvittrack_hip$v.so python tools/one_headconv.py 1 24 128 128 1 3 2 50 2>&1 | grep -v amdgpu
done
