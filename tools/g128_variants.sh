#!/bin/bash
# fc1 shape (30 cfg3 streams) on config 16 under each tuning build of k_gemm128.hip, and config 19 beside it
for v in "" _noprio _skew _skewnp; do
    export VITTRACK_HIP_LIB=$PWD/gstreamer-vit-tracker_amd/libvittrack_hip$v.so
    echo "lib '$v': cfg16 $(python3 tools/one_gemm.py 21600 3072 768 2 16 30 2>/dev/null)   cfg19 $(python3 tools/one_gemm.py 21600 3072 768 2 19 30 2>/dev/null)"
done
