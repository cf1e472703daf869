python tools/gemm_ab.py 0,1,3,17,19 4,8,16 3
python tools/split_bench.py cfg3 8 4+4 16 8+8 20 10+10
