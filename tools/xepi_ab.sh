#!/bin/bash
# what parts of the X-epilogue cost: builds of the library alternating in ONE process (tools/lib_ab.py)
#   python gstreamer-vit-tracker_amd/build.py --variant neither -DVT_AB_NOSTATS -DVT_AB_NOSPLIT   (no chunk statistics, no second half of the pair)
#   python gstreamer-vit-tracker_amd/build.py --variant nofinal -DVT_AB_NOFINALIZE               (row terms not finalized by the last workgroup of a panel)
P=gstreamer-vit-tracker_amd
for shape in "21600 768 768" "21600 768 3072"; do
    python3 tools/lib_ab.py $shape 1 18 $P/libvittrack_hip.so,$P/libvittrack_hip_nofinal.so,$P/libvittrack_hip_neither.so 9 2>&1 | grep -v amdgpu.ids
done
