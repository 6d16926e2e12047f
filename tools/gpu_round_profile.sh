set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01n
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r01n/tests.log 2>&1
tail -3 $R/gpurun_out/r01n/tests.log
timeout -k 10 400 python bench.py > $R/gpurun_out/r01n/bench.json 2> $R/gpurun_out/r01n/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01n/ks -o ks -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-pairs 0 > $R/gpurun_out/r01n/ks.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01n/pf -o pf -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-pairs 0 > $R/gpurun_out/r01n/pf.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01n/pw -o pw -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-pairs 0 > $R/gpurun_out/r01n/pw.log 2>&1
ls -R $R/gpurun_out/r01n | head -40
