"""Host code under AddressSanitizer / UBSan (CPU build only): the BAM reader and the `signal` step on malformed records --
an integer tag cut off behind its type byte, a Z tag without terminator, a read name without NUL, l_qname = 0, a B array
that claims more elements than the record holds, an empty file.  Every case must end with a clean error or clean output,
never with a sanitizer report (aux values were read past the record)."""
import os
import struct
import subprocess

import pytest

import test_signal as ts

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "asan", "signal_asan")


@pytest.fixture(scope="module")
def exe():
    src = os.path.join(HERE, "asan", "signal_asan_main.cpp")
    deps = [src] + [os.path.join(ts.ROOT, "pansvr_amd", "csrc", f) for f in ("signal_step.h", "bam_reader.h", "fastq_batch.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-DPSVR_NO_ENGINE_LIB", "-o", EXE, src, "-lz", "-lpthread"])
    return EXE


def raw_record(name_bytes, l_qname, flag, aux, seq_len=20, n_cigar=1):
    cg = struct.pack("<I", seq_len << 4) * n_cigar
    body = struct.pack("<iiBBHHHiiii", 0, 100, l_qname, 20, 4680, n_cigar, flag, seq_len, 0, 300, 250) + name_bytes + cg + bytes((seq_len + 1) // 2) + bytes([30] * seq_len) + aux
    return struct.pack("<i", len(body)) + body


GOOD1 = raw_record(b"p1\0", 3, 0x41, b"NMC\x02")
GOOD2 = raw_record(b"p1\0", 3, 0x81, b"NMC\x01")
CASES = {
    "int_tag_cut_off": [GOOD1, raw_record(b"p1\0", 3, 0x81, b"NMi\x01")],
    "short_tag_cut_off": [GOOD1, raw_record(b"p1\0", 3, 0x81, b"NMS\x01")],
    "z_tag_unterminated": [GOOD1, raw_record(b"p1\0", 3, 0x81, b"XAZchr1,+100,20M,0;chr2")],
    "z_tag_unterminated_before_nm": [raw_record(b"p1\0", 3, 0x41, b"SAZabcdef"), GOOD2],
    "b_array_overclaims": [GOOD1, raw_record(b"p1\0", 3, 0x81, b"ZBBc" + struct.pack("<I", 1 << 30) + b"\x01\x02NMC\x01")],
    "qname_without_nul": [raw_record(b"p1x", 3, 0x41, b""), GOOD2],
    "qname_empty": [raw_record(b"", 0, 0x41, b""), GOOD2],
    "unknown_tag_type": [GOOD1, raw_record(b"p1\0", 3, 0x81, b"XX?\x01\x02")],
    "empty_file": [],
    "valid": [GOOD1, GOOD2],
}


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("mode", ["-N", "pos"])
def test_malformed_records_end_cleanly(exe, tmp_path, case, mode):
    bam = str(tmp_path / "x.bam")
    ts.write_bam(bam, CASES[case], [("chr1", 1000000)])
    args = [exe, "signal", "-D", "-H", str(tmp_path / "h.sam"), "-S", str(tmp_path / "s.txt")] + (["-N"] if mode == "-N" else []) + [bam]
    r = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    err = r.stderr.decode(errors="replace")
    assert "AddressSanitizer" not in err and "runtime error" not in err, err[-3000:]
    assert r.returncode in (0, 1), (r.returncode, err[-1000:])          # clean output or a clean refusal, not a signal
    if case == "valid":
        assert r.returncode == 0 and r.stdout.count(b"\n") == 8
