set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_aln_gpu.py -x -q 2>&1 | tail -2
python - <<'PY'
import os, subprocess, sys, tempfile, filecmp
sys.path.insert(0,'tests')
import aln_common as ac
w=ac.workdir("fx2"); tmp=tempfile.mkdtemp()
cli=os.path.join(ac.ROOT,"pansvr_amd","bin","panSVR")
base=[os.path.join(ac.golden_dir("fx2"),"idx"), os.path.join(w,"reads150.fq"), os.path.join(w,"header.sam")]
outs=[]
for tag,extra in (("one",[]),("many",["--batch","97"])):
    o=os.path.join(tmp,tag)
    r=subprocess.run([cli,"aln","-S","-t","4","-o",o+".sam","-p",o+".ori.sam","--records",o+".jsonl"]+extra+base,stdout=subprocess.PIPE,stderr=subprocess.PIPE)
    assert r.returncode==0, r.stderr.decode()[-500:]
    outs.append(o)
for ext in (".sam",".ori.sam",".jsonl"):
    same=filecmp.cmp(outs[0]+ext, outs[1]+ext, shallow=False)
    print(ext, "identical" if same else "DIFFER", os.path.getsize(outs[0]+ext))
    assert same
PY
timeout -k 10 500 python tools/cli_e2e.py 1000000 16
