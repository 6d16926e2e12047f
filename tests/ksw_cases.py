"""Deterministic DP test-case generator (inputs only) shared by the golden-vector
script and the parity tests.  Sequences are uint8 codes 0..3 (4 = N)."""
import numpy as np

DEF = dict(m=5, match=2, mismatch=12, q=16, e=1, q2=32, e2=0, w=200, zdrop=400, end_bonus=-1, flag=0)


def mat5(match, mismatch, scN=0):
    """KSW_ALN_handler::ksw_gen_mat_D (read_realignment.cpp:829-843)."""
    m = []
    for l in range(4):
        for k in range(4):
            m.append(match if l == k else -mismatch)
        m.append(scN)
    m += [scN] * 5
    return m


def mutate(rng, s, sub=0.03, ins=0.01, dele=0.01, maxindel=8):
    out = []
    i = 0
    n = len(s)
    while i < n:
        r = rng.random_sample()
        if r < sub:
            out.append((s[i] + 1 + rng.randint(3)) % 4)
            i += 1
        elif r < sub + ins:
            out.extend(rng.randint(0, 4, size=1 + rng.randint(maxindel)).tolist())
        elif r < sub + ins + dele:
            i += 1 + rng.randint(maxindel)
        else:
            out.append(s[i])
            i += 1
    return out


def rand_seq(rng, n):
    return rng.randint(0, 4, size=n).tolist()


def case(qs, ts, **kw):
    c = dict(DEF)
    c.update(kw)
    c["query"] = [int(x) for x in qs]
    c["target"] = [int(x) for x in ts]
    return c


def fixed_cases(seed=20241008):
    """The KAT list committed as tests/golden/ksw_kat.json.gz."""
    rng = np.random.RandomState(seed)
    cs = []
    # 1. realistic right/left extensions: tlen = qlen + 30
    for _ in range(60):
        ql = 5 + rng.randint(90)
        q = rand_seq(rng, ql)
        t = mutate(rng, q, 0.04, 0.02, 0.02)
        t = (t + rand_seq(rng, ql + 30))[: ql + 30]
        cs.append(case(q, t))
    # 2. small end-to-end gaps between seeds
    for _ in range(60):
        ql = 1 + rng.randint(32)
        q = rand_seq(rng, ql)
        t = mutate(rng, q, 0.1, 0.08, 0.08, 6)
        if not t:
            t = rand_seq(rng, 1)
        cs.append(case(q, t[:40]))
    # 3. length grid, related and unrelated
    grid = [1, 2, 15, 16, 17, 31, 32, 33, 94, 124, 150, 180, 250, 280, 500, 1599]
    for ql in grid:
        for tl in grid:
            if ql * tl > 1000000 or (ql > 300 and tl > 300 and (ql, tl) != (500, 500)):
                continue
            q = rand_seq(rng, ql)
            t = mutate(rng, q, 0.03, 0.01, 0.01)
            t = (t + rand_seq(rng, tl))[:tl]
            cs.append(case(q, t))
    for ql, tl in [(16, 16), (33, 150), (150, 33), (94, 124), (250, 280), (180, 150)]:
        cs.append(case(rand_seq(rng, ql), rand_seq(rng, tl)))
    # 4. long gaps around the long_thres switch (ksw2_extd2_sse.c:95-98)
    for g in [1, 2, 13, 14, 15, 16, 17, 18, 25, 40, 60]:
        q = rand_seq(rng, 120)
        cs.append(case(q, q[:60] + rand_seq(rng, g) + q[60:]))          # deletion of g
        cs.append(case(q[:60] + rand_seq(rng, g) + q[60:], q))          # insertion of g
        cs.append(case(q, rand_seq(rng, g) + q))                        # leading gap
        cs.append(case(q + rand_seq(rng, g), q))                        # trailing gap
    # 5. band-edge: |qlen-tlen| in {199,200,201,260}
    for d in [199, 200, 201, 260]:
        q = rand_seq(rng, 60)
        cs.append(case(q, q + rand_seq(rng, d)))
        cs.append(case(q + rand_seq(rng, d), q))
        cs.append(case(q, q + rand_seq(rng, d), w=50))
    for ql, tl in [(400, 420), (520, 500), (700, 900)]:
        q = rand_seq(rng, ql)
        t = (mutate(rng, q, 0.02, 0.01, 0.01, 30) + rand_seq(rng, tl))[:tl]
        cs.append(case(q, t))
        cs.append(case(q, t, w=30))
    # 6. z-drop triggers
    for zd in [10, 30, 100, 400]:
        for _ in range(4):
            q = rand_seq(rng, 150)
            k = 30 + rng.randint(60)
            t = q[:k] + rand_seq(rng, 180 - k)
            cs.append(case(q, t, zdrop=zd))
            cs.append(case(q, t, zdrop=zd, flag=0x40))
    # 7. N bases, homopolymers, all-mismatch
    for _ in range(10):
        q = rand_seq(rng, 80)
        t = mutate(rng, q)[:100]
        for i in rng.randint(0, len(q), size=3):
            q[i] = 4
        for i in rng.randint(0, len(t), size=3):
            t[i] = 4
        cs.append(case(q, t))
    cs.append(case([0] * 100, [0] * 130))
    cs.append(case([0] * 100, [1] * 130))
    cs.append(case([0, 1] * 50, [0, 1] * 65))
    cs.append(case([0, 1] * 50, [1, 0] * 65))
    cs.append(case([2] * 40 + [3] * 40, [2] * 55 + [3] * 30))
    # 8. flags (path uses only 0; others for completeness)
    flags = [0x01, 0x02, 0x04, 0x08, 0x18, 0x40, 0x80, 0xC0, 0x42, 0x82, 0x09]
    for fl in flags:
        for _ in range(6):
            ql = 10 + rng.randint(150)
            q = rand_seq(rng, ql)
            t = (mutate(rng, q, 0.05, 0.03, 0.03) + rand_seq(rng, 40))[: ql + rng.randint(40)]
            if not t:
                t = [0]
            cs.append(case(q, t, flag=fl, end_bonus=int(rng.choice([-1, 0, 10]))))
    # 9. scoring-parameter variants
    params = [dict(q=4, e=2, q2=24, e2=1, match=2, mismatch=4), dict(q=32, e=0, q2=16, e2=1),
              dict(q=6, e=2, q2=6, e2=2, match=1, mismatch=2), dict(q=5, e=3, q2=20, e2=1, match=3, mismatch=5),
              dict(w=-1), dict(w=10), dict(w=0), dict(q=16, e=1, q2=32, e2=0, match=1, mismatch=30)]
    for pr in params:
        for _ in range(6):
            ql = 10 + rng.randint(180)
            q = rand_seq(rng, ql)
            t = (mutate(rng, q, 0.05, 0.03, 0.03, 20) + rand_seq(rng, 40))[: max(1, ql + rng.randint(-8, 40))]
            cs.append(case(q, t, **pr))
    # 10. the second user of ksw_extd2_sse, fc_sv's contig re-alignment (SignalAssembly.hpp:418-420,463): match 2 / mismatch 10,
    #     gaps 24+2k | 32+1k, bandwidth = zdrop = gap_open2 + 100 = 132, contig-length queries against an anchor region
    sv = dict(match=2, mismatch=10, q=24, e=2, q2=32, e2=1, w=132, zdrop=132)
    for ql, tl in [(300, 350), (600, 640), (1000, 1100), (1500, 1500), (2000, 2100), (3000, 3100), (1000, 1300), (2500, 2400)]:
        t = rand_seq(rng, tl)
        base = t[: min(ql, tl)]
        q = mutate(rng, base, 0.01, 0.003, 0.003, 12)
        cs.append(case((q + rand_seq(rng, ql))[:ql], t, **sv))
        k = len(base) // 2                                               # an SV-sized event in the middle of the contig
        g = 40 + rng.randint(80)
        cs.append(case((base[:k] + rand_seq(rng, g) + base[k:])[:ql], t, **sv))      # insertion allele
        cs.append(case((base[:k] + base[k + g:] + rand_seq(rng, ql))[:ql], t, **sv))  # deletion allele
    cs.append(case(rand_seq(rng, 800), rand_seq(rng, 900), **sv))            # unrelated: z-drop at 132
    return cs


def random_cases(seed, n, maxlen=200):
    rng = np.random.RandomState(seed)
    cs = []
    for _ in range(n):
        ql = 1 + rng.randint(maxlen)
        q = rand_seq(rng, ql)
        kind = rng.randint(4)
        if kind == 0:
            t = (mutate(rng, q, 0.04, 0.02, 0.02) + rand_seq(rng, ql + 30))[: ql + 30]
        elif kind == 1:
            t = mutate(rng, q, 0.08, 0.05, 0.05, 12) or [0]
        elif kind == 2:
            t = rand_seq(rng, 1 + rng.randint(maxlen))
        else:
            k = rng.randint(ql + 1)
            t = q[:k] + rand_seq(rng, 1 + rng.randint(60))
        cs.append(case(q, t))
    return cs
