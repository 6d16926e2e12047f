"""Host-side BAM/BGZF encoder of the CLI (pansvr_amd/csrc/bam_writer.h) against the independent reader in
tests/bam_reader.py: header, reference list, every fixed field, CIGAR, 4-bit sequence, qualities, bin, integer tag
widths, string/char tags, multi-block BGZF with per-block sizes and the EOF marker."""
import os
import subprocess
import tempfile

import aln_common as ac
import bam_reader

DRIVER = r'''
#include "bam_writer.h"
int main(int argc, char **argv)
{
	psvr::BamWriter w;
	std::vector<psvr::BamRef> refs = {{"chr1", 1000000}, {"chr2", 2000000}};
	const int level = argc > 2 ? atoi(argv[2]) : Z_DEFAULT_COMPRESSION;        // -2: the built-in encoder (--bgzf-fast)
	if (!w.open(argv[1], "@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:1000000\n@SQ\tSN:chr2\tLN:2000000\n", refs, 3, level)) return 2;
	for (int i = 0; i < 3000; ++i) {
		psvr::SamFields f;
		f.qname = "r" + std::to_string(i), f.flag = i % 2 ? 0x50 : 0x83, f.tid = i % 2, f.pos1 = 100 + i * 337, f.mapq = i % 61;
		f.cigar = i % 3 ? "40S100M5I5M2D" : "150M";
		f.mtid = i % 5 ? f.tid : (i % 7 ? 1 - f.tid : -1), f.mpos1 = 500 + i, f.isize = (i % 2 ? 1 : -1) * (300 + i);
		f.seq = std::string(150 - i % 2, "ACGTN"[i % 5]), f.qual = std::string(150 - i % 2, (char)(33 + i % 40));
		f.tags = "\tAS:i:" + std::to_string(i * 97 - 1000) + "\tOS:i:300\tOA:Z:0,1,2,3,M;\tRC:Z:comment_" + std::to_string(i) + "\tXX:A:Q\tNM:i:70000\tYY:i:-40000\tZZ:i:-7";
		if (!w.write(f)) return 3;
	}
	return w.close() ? 0 : 4;
}
'''


import pytest


@pytest.mark.parametrize("level", [-1, 1, -2])
def test_bam_writer_round_trip(level):
    d = tempfile.mkdtemp(prefix="psvr_bamw_")
    open(os.path.join(d, "t.cpp"), "w").write(DRIVER)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ac.ROOT, "pansvr_amd", "csrc"), "-o", os.path.join(d, "t"), os.path.join(d, "t.cpp"), "-lz", "-lpthread"])
    subprocess.check_call([os.path.join(d, "t"), os.path.join(d, "x.bam"), str(level)])
    assert bam_reader.check_bgzf(os.path.join(d, "x.bam")) > 2          # several 0xff00-byte blocks + EOF
    text, refs, recs = bam_reader.read_bam(os.path.join(d, "x.bam"))
    assert text.startswith("@HD") and refs == [("chr1", 1000000), ("chr2", 2000000)] and len(recs) == 3000
    for i, r in enumerate(recs):
        tid = i % 2
        mtid = tid if i % 5 else ((1 - tid) if i % 7 else -1)
        want = ["r%d" % i, str(0x50 if i % 2 else 0x83), "chr%d" % (tid + 1), str(100 + i * 337), str(i % 61), "40S100M5I5M2D" if i % 3 else "150M",
                "*" if mtid < 0 else ("=" if mtid == tid else "chr%d" % (mtid + 1)), str(500 + i), str((1 if i % 2 else -1) * (300 + i)),
                "ACGTN"[i % 5] * (150 - i % 2), chr(33 + i % 40) * (150 - i % 2),
                "AS:i:%d" % (i * 97 - 1000), "OS:i:300", "OA:Z:0,1,2,3,M;", "RC:Z:comment_%d" % i, "XX:A:Q", "NM:i:70000", "YY:i:-40000", "ZZ:i:-7"]
        assert r == want, (i, r, want)
