set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01m
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r01m/tests.log 2>&1
tail -3 $R/gpurun_out/r01m/tests.log
timeout -k 10 400 python bench.py > $R/gpurun_out/r01m/bench.json 2> $R/gpurun_out/r01m/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01m/ks -o ks -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 > $R/gpurun_out/r01m/ks.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01m/pf -o pf -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-pairs 0 > $R/gpurun_out/r01m/pf.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01m/pw -o pw -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-pairs 0 > $R/gpurun_out/r01m/pw.log 2>&1
ls -R $R/gpurun_out/r01m | head -40
