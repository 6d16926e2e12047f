# GPU box: the drop-in command end to end on 1 M generated pairs (SAM at several thread counts, the BAM routes), CPU seconds beside the wall
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-e2e}
mkdir -p $O /tmp/e2e && cd /tmp/e2e
echo "nproc $(nproc) cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" | tee $O/probe.txt
g++ -O2 -std=c++17 -o gen $GRAFT_REPO_ROOT/tests/tools/gen_signal_fastq.cpp -lpthread
./gen anchors 10000 11 > anchors.fa; mkdir -p idx; $GRAFT_REPO_ROOT/pansvr_amd/bin/panSVR index -k 22 --sparse-hash anchors.fa idx/ 2>/dev/null
D=/dev/shm/e2e_$$; mkdir -p $D; ./gen reads 10000 11 1000000 17 8 > $D/reads.fq; printf "@SQ\tSN:chr1\tLN:250000000\n@SQ\tSN:chr2\tLN:250000000\n" > header.sam
run() { # label, args...
  l=$1; shift
  TIMEFORMAT="%U %S"
  rm -f $D/o.* $D/p.* $D/o0.* $D/p0.*                     # (opening an existing 0.9 GB file for writing frees its pages first: 0.1 s)
  { time $BIN aln "$@" idx $D/reads.fq header.sam 2> err.txt; } 2> tm.txt
  grep -q e2e_json err.txt || { echo "$l FAILED"; tail -n 3 err.txt; return; }
  echo "$l $(grep -o 'wall_s.:[0-9.]*' err.txt) $(grep -o 'read_parse_s.:[0-9.]*' err.txt) $(grep -o 'engine_s.:[0-9.]*' err.txt) $(grep -o 'format_s.:[0-9.]*' err.txt) $(grep -o 'write_s.:[0-9.]*' err.txt) cpu(user sys) $(cat tm.txt)" | tee -a $O/probe.txt
}
BIN=$GRAFT_REPO_ROOT/pansvr_amd/bin/panSVR
for rep in 1; do
for t in 16; do run "SAM t=$t" -S -t $t -o $D/o.sam -p $D/p.sam; done
md5sum $D/o.sam $D/p.sam | tee -a $O/probe.txt
run "SAM t=16 --sub-batch 32768" -S -t 16 --sub-batch 32768 -o $D/o.sam -p $D/p.sam
md5sum $D/o.sam $D/p.sam | tee -a $O/probe.txt
done
rm -f $D/o.* $D/p.*
PSVR_CLI_TIMING=1 $BIN aln -S -t 16 -o $D/o.sam -p $D/p.sam idx $D/reads.fq header.sam 2> $O/timeline.txt
md5sum $D/o.sam $D/p.sam | tee -a $O/probe.txt
if [ "$2" = "bam-block" ]; then
  for b in 16384 8192 4096; do export PSVR_BGZF_BLOCK=$b; run "BAM device t=16 block $b" -t 16 --bgzf-device -o $D/o.bam -p $D/p.bam; ls -l $D/o.bam | awk '{print "  o.bam bytes", $5}' | tee -a $O/probe.txt; done
  unset PSVR_BGZF_BLOCK
  rm -rf $D; exit 0
fi
if [ "$2" = "bam-device" ]; then
  PSVR_CLI_TIMING=1 $BIN aln -t 16 --bgzf-device -o $D/o.bam -p $D/p.bam idx $D/reads.fq header.sam 2> $O/timeline_bamdev.txt
  rm -f $D/o.* $D/p.*
  PSVR_CLI_TIMING=1 $BIN aln -t 16 --compress-level 1 -o $D/o.bam -p $D/p.bam idx $D/reads.fq header.sam 2> $O/timeline_bam1.txt
  rm -rf $D; exit 0
fi
if [ "$2" = "hip-trace" ]; then
  rm -f $D/o.* $D/p.*
  export TMPDIR=/tmp
  timeout -k 10 200 rocprofv3 --hip-trace --stats --output-format csv -d $O/hip -o hip -- $BIN aln -S -t 16 -o $D/o.sam -p $D/p.sam idx $D/reads.fq header.sam > $O/hip.log 2>&1
  rm -rf $D; exit 0
fi
[ "$2" = "sam-only" ] && { rm -rf $D; exit 0; }
run "BAM t=48" -t 48 -o $D/o.bam -p $D/p.bam
run "BAM level1 t=48" -t 48 --compress-level 1 -o $D/o.bam -p $D/p.bam
ls -l $D/o.bam | awk '{print "  o.bam bytes", $5}' | tee -a $O/probe.txt
run "BAM fast t=48" -t 48 --bgzf-fast -o $D/o.bam -p $D/p.bam
ls -l $D/o.bam | awk '{print "  o.bam bytes", $5}' | tee -a $O/probe.txt
run "BAM fast t=16" -t 16 --bgzf-fast -o $D/o.bam -p $D/p.bam
run "BAM device t=48" -t 48 --bgzf-device -o $D/o.bam -p $D/p.bam
run "BAM device t=16" -t 16 --bgzf-device -o $D/o.bam -p $D/p.bam
rm -rf $D
