"""-m gpu: the HIP DP kernels (through the C ABI, seam B2) against
(1) the committed reference known-answer vectors and (2) the oracle on fresh random inputs."""
import numpy as np
import pytest

from ksw_cases import rand_seq, case, mat5, random_cases
from ksw_ref import run_oracle
from test_oracle_ksw import diff, load_kat

pytestmark = pytest.mark.gpu


def run_gpu(cases, variant):
    from pansvr_amd import ksw
    # group by parameter set: one batch (one plan) per distinct parameter tuple
    groups = {}
    for i, c in enumerate(cases):
        key = (c["m"], c["match"], c["mismatch"], c["q"], c["e"], c["q2"], c["e2"], c["w"], c["zdrop"], c["end_bonus"], c["flag"])
        groups.setdefault(key, []).append(i)
    out = [None] * len(cases)
    for key, ids in groups.items():
        m, match, mismatch, q, e, q2, e2, w, zdrop, end_bonus, flag = key
        p = ksw.make_params(m, mat5(match, mismatch), q, e, q2, e2, w, zdrop, end_bonus, flag)
        res = ksw.ext_batch([cases[i]["query"] for i in ids], [cases[i]["target"] for i in ids], p, variant)
        for i, r in zip(ids, res):
            out[i] = r
    return out


@pytest.mark.parametrize("variant", ["extd2", "extz2"])
def test_gpu_matches_reference_kat(variant):
    recs = load_kat()
    got = run_gpu(recs, variant)
    bad = [(i, recs[i]["flag"], len(recs[i]["query"]), len(recs[i]["target"]), diff(recs[i][variant], g))
           for i, g in enumerate(got) if g != recs[i][variant]]
    assert not bad, "%d/%d mismatches, first: %r" % (len(bad), len(recs), bad[:3])


@pytest.mark.parametrize("variant", ["extd2", "extz2"])
def test_gpu_matches_oracle_random(variant):
    cases = random_cases(777, 4000, 300)
    got = run_gpu(cases, variant)
    bad = []
    for i, (c, g) in enumerate(zip(cases, got)):
        want = run_oracle(c, variant)
        if g != want:
            bad.append((i, len(c["query"]), len(c["target"]), diff(want, g)))
    assert not bad, "%d mismatches, first: %r" % (len(bad), bad[:3])


def test_gpu_tiny_problems_match_oracle():
    """qlen, tlen <= 16 go to the one-thread-per-alignment kernel (extd2_tiny_kernel): every shape 1..16 x 1..16 several
    times over, with N bases, the flags that kernel implements (extension-only, reversed CIGAR, score-only) and z-drop values
    small enough to trigger on a 16-base matrix."""
    rng = np.random.RandomState(4242)
    cases = []
    for rep in range(6):
        for ql in range(1, 17):
            for tl in range(1, 17):
                q = rand_seq(rng, ql)
                t = (list(q[:min(ql, tl)]) + rand_seq(rng, tl))[:tl] if rng.randint(2) else rand_seq(rng, tl)
                t = [int(x) for x in t]
                for _ in range(rng.randint(3)):
                    t[rng.randint(tl)] = int(rng.randint(5))           # substitutions, some to N (4)
                if rng.randint(8) == 0:
                    q[rng.randint(ql)] = 4
                flag = int(rng.choice([0, 0, 0, 0x40, 0x80, 0xC0, 0x01, 0x41]))
                cases.append(case(q, t, flag=flag, zdrop=int(rng.choice([400, 400, 10, 3])), end_bonus=int(rng.choice([-1, 0, 5]))))
    got = run_gpu(cases, "extd2")
    bad = []
    for i, (c, g) in enumerate(zip(cases, got)):
        want = run_oracle(c, "extd2")
        if g != want:
            bad.append((i, c["flag"], c["zdrop"], len(c["query"]), len(c["target"]), diff(want, g)))
    assert not bad, "%d/%d mismatches, first: %r" % (len(bad), len(cases), bad[:3])


def test_gpu_team_kernel_edge_shapes_match_oracle():
    """The team kernel (extd2_team_kernel + extd2_team_finish_kernel: strips of 16 target columns, two lanes of eight columns, a row per
    step) at the shapes where its bookkeeping has corners: queries shorter than the lanes' stagger, a last strip of 1, 7, 8, 9 or 16
    columns, the largest target the band of these cases lets it take (201; 207 goes to the wavefront kernels), targets of a single partial strip under long queries; with N bases, the flags it
    implements and z-drop values that stop the rules early.  (qlen, tlen <= 16 is the thread-per-alignment kernel's, tested above.)"""
    rng = np.random.RandomState(9091)
    shapes = [(ql, tl) for ql in (1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 31, 32, 33, 64) for tl in (17, 18, 23, 24, 25, 31, 32, 33, 40, 47, 48, 49, 96, 193, 200, 201, 207)]
    shapes += [(ql, tl) for ql in range(17, 41) for tl in (1, 2, 7, 8, 9, 15, 16)]
    shapes += [(ql, tl) for ql in (190, 199, 200) for tl in (1, 16, 17, 32, 33, 150, 201)]
    cases = []
    for rep in range(3):
        for ql, tl in shapes:
            q = rand_seq(rng, ql)
            t = (list(q[:min(ql, tl)]) + rand_seq(rng, tl))[:tl] if rng.randint(3) else rand_seq(rng, tl)
            t = [int(x) for x in t]
            for _ in range(rng.randint(4)):
                t[rng.randint(tl)] = int(rng.randint(5))               # substitutions, some to N (4)
            if rng.randint(8) == 0:
                q[rng.randint(ql)] = 4
            flag = int(rng.choice([0, 0, 0, 0x40, 0x80, 0xC0, 0x01, 0x41]))
            cases.append(case(q, t, flag=flag, zdrop=int(rng.choice([400, 400, 30, 10, 3])), end_bonus=int(rng.choice([-1, 0, 5]))))
    got = run_gpu(cases, "extd2")
    bad = []
    for i, (c, g) in enumerate(zip(cases, got)):
        want = run_oracle(c, "extd2")
        if g != want:
            bad.append((i, c["flag"], c["zdrop"], len(c["query"]), len(c["target"]), diff(want, g)))
    assert not bad, "%d/%d mismatches, first: %r" % (len(bad), len(cases), bad[:3])


def test_gpu_random_scoring_and_bands_match_oracle():
    """Scoring parameters and band widths other than the path's (the CLI exposes -M -m -O -E -P -F -z): every kernel family gets
    exercised -- small bands clip the matrix (wavefront kernels with the SSE artefacts), large penalties leave the int8-safe
    regime (no lean sweep), swapped gap pairs hit the pre-swap qe quirk."""
    rng = np.random.RandomState(9001)
    cases = []
    for _ in range(14):
        match = int(rng.randint(1, 4))
        mismatch = int(rng.randint(1, 31))
        q, e = int(rng.randint(2, 33)), int(rng.randint(0, 4))
        q2, e2 = int(rng.randint(2, 41)), int(rng.randint(0, 3))
        w = int(rng.choice([200, 200, 200, 64, 16, 5, -1]))
        zdrop = int(rng.choice([400, 400, 50, 10, -1]))
        for c in random_cases(int(rng.randint(1 << 30)), 90, 140):
            c.update(match=match, mismatch=mismatch, q=q, e=e, q2=q2, e2=e2, w=w, zdrop=zdrop, flag=int(rng.choice([0, 0, 0x40, 0x80])))
            cases.append(c)
    got = run_gpu(cases, "extd2")
    bad = []
    for i, (c, g) in enumerate(zip(cases, got)):
        want = run_oracle(c, "extd2")
        if g != want:
            bad.append((i, {k: c[k] for k in ("match", "mismatch", "q", "e", "q2", "e2", "w", "zdrop", "flag")}, len(c["query"]), len(c["target"]), diff(want, g)))
    assert not bad, "%d/%d mismatches, first: %r" % (len(bad), len(cases), bad[:3])


def test_gpu_empty_and_degenerate():
    from pansvr_amd import ksw
    p = ksw.make_params(5, mat5(2, 12), 16, 1, 32, 0, 200, 400, -1, 0)
    assert ksw.ext_batch([], [], p) == []
    res = ksw.ext_batch([[0], [], [1, 2]], [[0], [1, 2, 3], []], p)
    assert res[0]["score"] == 2 and res[0]["cigar"] == [1 << 4]
    for r in res[1:]:   # qlen<=0 or tlen<=0: the reference returns right after ksw_reset_extz
        assert r["n_cigar"] == 0 and r["score"] == -0x40000000 and r["max"] == 0


def test_gpu_lean_team_kernel_matches_oracle(monkeypatch):
    """The variant of the team kernel the engine's own DP launches run when the z-drop rule cannot trigger (e2 == 0, zdrop >= 2 q2: the
    reference's defaults 2/-12, 16+1k | 32+0k, zdrop 400, band 200): no per-anti-diagonal maximum is kept, so ez.max / max_q / max_t are
    not produced -- every other field and the CIGAR must be the oracle's.  PSVR_DP_FORCE_LEAN routes psvr_extd2_batch through it."""
    monkeypatch.setenv("PSVR_DP_FORCE_LEAN", "1")
    from ksw_cases import mutate
    rng = np.random.RandomState(99)
    cases = []
    for k in range(3000):
        ql = int(rng.randint(17, 200))
        q = rand_seq(rng, ql)
        t = mutate(rng, q, 0.04, 0.02, 0.02)
        if k % 3:                                       # an extension: the window is the piece + 30
            t = (t + rand_seq(rng, ql + 30))[:min(ql + 30, 201)]
        else:                                           # an end-to-end piece
            t = t[:201] or rand_seq(rng, 3)
        cases.append(case(q, t))                        # the defaults of ksw_cases.DEF are the aln path's parameters
    got = run_gpu(cases, "extd2")
    skip = ("max", "max_q", "max_t")
    bad = []
    for i, (c, g) in enumerate(zip(cases, got)):
        want = run_oracle(c, "extd2")
        assert want["zdropped"] == 0                   # what makes the variant legal
        a = {k: v for k, v in want.items() if k not in skip}
        b = {k: v for k, v in g.items() if k not in skip}
        if a != b:
            bad.append((i, len(c["query"]), len(c["target"]), diff(want, g)))
    assert not bad, "%d mismatches, first: %r" % (len(bad), bad[:3])


def test_gpu_ring_kernels_match_oracle():
    """Matrices wider than the register-resident kernels' 320 columns whose band fits a ring of 192 / 256 columns (extd2_ring_kernel<3 / 4>):
    `fc_sv`'s contig re-alignment (2/-10, 24+2k | 32+1k, w = zdrop = 132, SignalAssembly.hpp:418-420,463) and the `aln` path's own parameters at
    w = 200 on long reads; band widths either side of the two ring sizes' limits (w + 33 <= 192: 159 | 160; <= 256: 223 | 224, the latter
    goes to the general kernel), extensions with a far longer target (the band runs along the query and off the matrix), z-drops in the middle,
    N bases, every flag of the fast path."""
    from ksw_cases import mutate
    rng = np.random.RandomState(31337)
    cases = []
    shapes = [(321, 321), (340, 500), (500, 340), (777, 801), (1200, 1150), (1599, 1629), (2048, 2047), (3100, 3000), (330, 3100), (3100, 330), (640, 641)]
    for k in range(150):
        ql, tl = shapes[k % len(shapes)] if k < 44 else (int(rng.randint(200, 1800)), int(rng.randint(321, 1800)))
        q = rand_seq(rng, ql)
        t = mutate(rng, q, 0.03, 0.01, 0.01, maxindel=int(rng.choice([8, 40, 150])))
        t = (t + rand_seq(rng, tl))[:tl] if len(t) < tl else t[:tl]
        if k % 7 == 0:                                   # a diverged stretch: z-drop candidates
            a = int(rng.randint(0, max(1, min(ql, tl) - 120)))
            t[a:a + 100] = rand_seq(rng, len(t[a:a + 100]))
        if k % 5 == 0:
            q[int(rng.randint(ql))] = 4
            t[int(rng.randint(tl))] = 4
        flag = int(rng.choice([0, 0, 0x40, 0x80, 0xC0, 0x01]))
        if k % 2:
            cases.append(case(q, t, match=2, mismatch=10, q=24, e=2, q2=32, e2=1, w=132, zdrop=132, flag=flag))
        else:
            cases.append(case(q, t, w=int(rng.choice([200, 200, 159, 160, 223, 224, 64, 20])), zdrop=int(rng.choice([400, 400, 100, -1])), flag=flag))
    got = run_gpu(cases, "extd2")
    bad = []
    for i, (c, g) in enumerate(zip(cases, got)):
        want = run_oracle(c, "extd2", cap=16384)
        if g != want:
            bad.append((i, c["w"], c["flag"], len(c["query"]), len(c["target"]), diff(want, g)))
    assert not bad, "%d/%d mismatches, first: %r" % (len(bad), len(cases), bad[:3])
