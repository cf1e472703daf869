set -e
cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -k "head_band" 2>&1 | tail -2
python -m pytest tests/test_gpu_pipeline.py -x -q -k "head_band" 2>&1 | tail -2
python tools/headconv_bench.py 1,30 2>&1 | grep -v amdgpu | cut -c1-200
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_r05b/conv3 -- python3 tools/one_headconv.py 30 24 128 128 1 0 0 20 > gpurun_out/pmc_r05b.log 2>&1
python - <<'PY'
import csv,glob
acc={};n={}
for f in glob.glob('gpurun_out/pmc_r05b/conv3/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'head_conv' in r['Kernel_Name']:
            acc[r['Counter_Name']]=acc.get(r['Counter_Name'],0)+float(r['Counter_Value']); n[r['Counter_Name']]=n.get(r['Counter_Name'],0)+1
print({k:round(acc[k]/n[k]) for k in acc})
PY
export VITTRACK_HIP_LIB=$PWD/gstreamer-vit-tracker_amd/libvittrack_hip_stamps.so
for a in "30 24 128 128 1 3 2" "30 24 768 128 0 3 2"; do echo "one_headconv $a:"; python tools/one_headconv.py $a 20 2>&1 | grep -v amdgpu; done
