cd $GRAFT_REPO_ROOT
nproc; python3 -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; free -g | head -2
mkdir -p gpurun_out/r5b /tmp/e2e && cd /tmp/e2e
g++ -O2 -std=c++17 -o gen $GRAFT_REPO_ROOT/tests/tools/gen_signal_fastq.cpp -lpthread
./gen anchors 10000 11 > anchors.fa; mkdir -p idx; $GRAFT_REPO_ROOT/pansvr_amd/bin/panSVR index -k 22 --sparse-hash anchors.fa idx/ 2>/dev/null
D=/dev/shm/e2e_$$; mkdir -p $D; ./gen reads 10000 11 1000000 17 8 > $D/reads.fq; printf "@SQ\tSN:chr1\tLN:250000000\n@SQ\tSN:chr2\tLN:250000000\n" > header.sam
for t in 8 16 32; do PSVR_CLI_TIMING=1 $GRAFT_REPO_ROOT/pansvr_amd/bin/panSVR aln -S -t $t -o $D/o.sam -p $D/p.sam idx $D/reads.fq header.sam 2> err_$t.txt; echo "t=$t $(grep -o 'wall_s.:[0-9.]*' err_$t.txt) $(grep -o 'read_parse_s.:[0-9.]*' err_$t.txt) $(grep -o 'engine_s.:[0-9.]*' err_$t.txt) $(grep -o 'format_s.:[0-9.]*' err_$t.txt) $(grep -o 'write_s.:[0-9.]*' err_$t.txt)"; done
grep "batch " err_16.txt
for t in 16 48; do $GRAFT_REPO_ROOT/pansvr_amd/bin/panSVR aln -t $t -o $D/o.bam -p $D/p.bam idx $D/reads.fq header.sam 2> errb_$t.txt; echo "BAM t=$t $(grep -o 'wall_s.:[0-9.]*' errb_$t.txt) $(grep -o 'format_s.:[0-9.]*' errb_$t.txt) $(grep -o 'write_s.:[0-9.]*' errb_$t.txt)"; done
rm -rf $D
