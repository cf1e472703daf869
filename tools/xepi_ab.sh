#!/bin/bash
# what parts of the X-epilogue cost (tuning builds: python gstreamer-vit-tracker_amd/build.py --variant NAME -DMACRO):
#   neither: -DVT_AB_NOSTATS -DVT_AB_NOSPLIT (no chunk statistics, no second half of the pair)
#   nofinal: -DVT_AB_NOFINALIZE (row terms not finalized by the last workgroup of a panel)
for v in "" _neither _nofinal; do
  for shape in "21600 768 768" "21600 768 3072"; do
    echo -n "lib$v $shape: "
    VITTRACK_HIP_LIB=gstreamer-vit-tracker_amd/libvittrack_hip$v.so python3 tools/one_gemm.py $shape 1 18 40 2>&1 | grep us
  done
done
