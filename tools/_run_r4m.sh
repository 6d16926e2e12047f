cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4m && bash tools/gpu_ab_libs.sh r4m 2
timeout -k 10 200 python tools/dp_bench.py 200000 35 94 > gpurun_out/r4m/dp_base.txt 2>&1
cp gpu_exp/lib_waves4.so pansvr_amd/libpsvr_engine.so && timeout -k 10 200 python tools/dp_bench.py 200000 35 94 > gpurun_out/r4m/dp_waves4.txt 2>&1
cp gpu_exp/lib_base.so pansvr_amd/libpsvr_engine.so
tail -n 2 gpurun_out/r4m/dp_base.txt gpurun_out/r4m/dp_waves4.txt
timeout -k 10 300 python tools/dp_bench_wide.py 16384 300 3100 > gpurun_out/r4m/dp_wide16k.txt 2>&1; tail -n 3 gpurun_out/r4m/dp_wide16k.txt
