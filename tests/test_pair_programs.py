"""CPU execution of the two-lanes-per-signature GCN programs (fabric-mod_amd/csrc/gen_pair_gcn.py) in the reference
interpreter of gcn_dsl.py: the exact instruction lists the pair kernel runs, with 32/64-bit wrap-around semantics and an
explicit (even, odd) lane pair, checked against big-integer point arithmetic of the oracle.  Chained so that the lazy
limb ranges the kernel produces (L(Y) = 1 after a doubling or an addition, 2 after a mixed addition) are what the next program eats."""
import os
import random
import sys

import pytest

import bccsp_sw_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fabric-mod_amd", "csrc"))
import gen_pair_gcn as gp  # noqa: E402

P = po.P
R = 1 << 261
RI = pow(R, -1, P)


def bal(x):
    d = []
    for _ in range(8):
        t = x & ((1 << 29) - 1)
        if t >> 28:
            t -= 1 << 29
        d.append(t)
        x = (x - t) >> 29
    d.append(x)
    return d


def to_fe(x):
    return bal(x * R % P)


def val(regs, name):
    return sum(regs["%s.%d" % (name, i)] << (29 * i) for i in range(9)) * RI % P


def put(regs, name, digits):
    for i in range(9):
        regs["%s.%d" % (name, i)] = digits[i]


def jac_of(pt, rng):
    z = rng.randrange(1, P)
    return (pt[0] * z * z % P, pt[1] * z * z * z % P, z)


def affine(X, Y, Z):
    zi = pow(Z, -1, P)
    return (X * zi * zi % P, Y * zi * zi * zi % P)


class Pair:
    """Lane-pair state between programs: E holds A = X, B = Y; O holds B = Z."""

    def __init__(self, X, Y, Z, rng):
        self.e, self.o = {}, {}
        put(self.e, "A", to_fe(X)); put(self.e, "B", to_fe(Y))
        put(self.o, "A", [rng.randrange(-(1 << 28), 1 << 28) for _ in range(9)]); put(self.o, "B", to_fe(Z))

    def point(self):
        return affine(val(self.e, "A"), val(self.e, "B"), val(self.o, "B"))

    def run(self, prog, extra_e=None, extra_o=None):
        e = {k: v for k, v in self.e.items() if k[:2] in ("A.", "B.")}
        o = {k: v for k, v in self.o.items() if k[:2] in ("A.", "B.")}
        e.update(extra_e or {}); o.update(extra_o or {})
        self.e, self.o = prog.run(e, o)
        for regs in (self.e, self.o):          # every limb must still be a 32-bit value
            assert all(-(1 << 31) <= v < (1 << 31) for v in regs.values())
            if "AO.0" in regs:                 # the additions work out of place: their sum is the next state
                for i in range(9):
                    regs["A.%d" % i], regs["B.%d" % i] = regs["AO.%d" % i], regs["BO.%d" % i]


@pytest.fixture(scope="module")
def progs():
    return {"dbl": gp.build_pair_dbl(), "add": gp.build_pair_add(), "madd": gp.build_pair_madd()}


def test_program_sizes(progs):
    sizes = {}
    for k, p in progs.items():
        text, st = p.emit_asm({n: "(%s)" % n for n in p.order})
        sizes[k] = st
        assert "s_nop" not in text or st["nops"] < 8
    # the point of the exercise: well under the one-lane streams (1339 / 2678 / 1900 instructions)
    assert sizes["dbl"]["instructions"] < 850 and sizes["add"]["instructions"] < 1600 and sizes["madd"]["instructions"] < 1300


def test_field_product_and_square_programs():
    """FE29_GCN_MUL / FE29_GCN_SQR (the single-lane streams of fe29.h) come from the same DSL: run them in the interpreter."""
    rng = random.Random(76)
    mul, sqr = gp.build_fe_mul(), gp.build_fe_sqr()
    for _ in range(40):
        a, b = rng.randrange(P), rng.randrange(P)
        e, o = {}, {}
        put(e, "A", to_fe(a)); put(e, "B", to_fe(b)); put(o, "A", to_fe(b)); put(o, "B", to_fe(a))
        re, ro = mul.run(e, o)
        assert val(re, "R") == a * b % P and val(ro, "R") == a * b % P
        lazy = [x + y for x, y in zip(to_fe(a), to_fe(b))]            # L = 2 operand: (a + b)^2
        e2, o2 = {}, {}
        put(e2, "A", lazy); put(o2, "A", to_fe(a))
        re, ro = sqr.run(e2, o2)
        assert val(re, "R") == (a + b) * (a + b) % P and val(ro, "R") == a * a % P
        assert all(-(1 << 28) <= re["R.%d" % i] <= (1 << 28) for i in range(8))   # balanced output digits


def test_bn_field_product_and_square_programs():
    """BN29_GCN_MUL / BN29_GCN_SQR (bn29.h on the device): the DSL's generic-prime reduction, run in the interpreter against big
    integers mod the FP256BN prime, with lazy operands up to the bound of that field (L(a) L(b) <= 12)."""
    import gen_bn_consts as bc
    BP = bc.P
    BRI = pow(R, -1, BP)

    def to_bn(x):
        return bal(x * R % BP)

    def bval(regs, name):
        return sum(regs["%s.%d" % (name, i)] << (29 * i) for i in range(9)) * BRI % BP
    rng = random.Random(77)
    mul, sqr = gp.build_bn_mul(), gp.build_bn_sqr()
    assert sum(1 for i in mul.ins if i[0] in ("mad", "mad0")) == 162 and sum(1 for i in sqr.ins if i[0] in ("mad", "mad0")) == 126
    edge = [0, 1, BP - 1, BP - 2, (1 << 255) % BP]
    cases = [(a, b) for a in edge for b in edge] + [(rng.randrange(BP), rng.randrange(BP)) for _ in range(60)]
    for a, b in cases:
        e, o = {}, {}
        put(e, "A", to_bn(a)); put(e, "B", to_bn(b)); put(o, "A", to_bn(b)); put(o, "B", to_bn(a))
        re, ro = mul.run(e, o)
        assert bval(re, "R") == a * b % BP and bval(ro, "R") == a * b % BP
        l3 = [3 * x for x in to_bn(a)]                                 # L = 3 operand, squared: 9 <= 12
        l4 = [4 * x for x in to_bn(a)]
        e2, o2 = {}, {}
        put(e2, "A", l3); put(o2, "A", to_bn(b))
        re, ro = sqr.run(e2, o2)
        assert bval(re, "R") == 9 * a * a % BP and bval(ro, "R") == b * b % BP
        assert all(-(1 << 28) <= re["R.%d" % i] <= (1 << 28) for i in range(8))
        e3, o3 = {}, {}
        put(e3, "A", l4); put(e3, "B", l3); put(o3, "A", l3); put(o3, "B", l4)   # L = 4 x 3 = 12: the edge of the bound
        re, ro = mul.run(e3, o3)
        assert bval(re, "R") == 12 * a * a % BP


def test_pair_scalar_multiplication_chain(progs):
    """Left-to-right double-and-add of a random 48-bit scalar with Jacobian table entries (pair add) and affine ones
    (pair madd), against the oracle after every step."""
    rng = random.Random(77)
    G = (po.GX, po.GY)
    for trial in range(3):
        base = po.pt_mul(rng.randrange(1, po.N), G)
        k = rng.randrange(1 << 47, 1 << 48)
        st = Pair(*jac_of(base, rng), rng)
        acc = base
        for bit in bin(k)[3:]:
            st.run(progs["dbl"])
            acc = po.pt_add(acc, acc)
            assert st.point() == acc
            if bit == "1":
                if rng.random() < 0.5:
                    X2, Y2, Z2 = jac_of(base, rng)
                    ce, co = {}, {}
                    put(ce, "C", to_fe(Z2)); put(ce, "D", [0] * 9)
                    put(co, "C", to_fe(X2)); put(co, "D", to_fe(Y2))
                    st.run(progs["add"], ce, co)
                    # H (E: h, O: -h) and RR on E are the caller's exceptional-case probes: non-zero here
                    assert val(st.e, "H") != 0 and (val(st.o, "H") + val(st.e, "H")) % P == 0 and val(st.e, "RR") != 0
                else:
                    ce, co = {}, {}
                    put(ce, "C", to_fe(base[0])); put(ce, "D", [0] * 9)
                    put(co, "C", [0] * 9); put(co, "D", to_fe(base[1]))
                    st.run(progs["madd"], ce, co)
                acc = po.pt_add(acc, base)
                assert st.point() == acc


def test_bn_pair_programs_scalar_multiplication_chain():
    """PAIRBN_DBL / PAIRBN_ADD / PAIRBN_MADD (pair29_bn_gcn.h: the point operations of the four-lanes-per-signature idemix kernel,
    bn_quad29.h): left-to-right double-and-add on FP256BN's G1 in the interpreter, against big-integer point arithmetic after
    every step, on a fixture base point of the reference."""
    import gen_bn_consts as bc
    import idemix_oracle as io
    from idemix_common import fixtures
    BP = bc.P
    BRI = pow(R, -1, BP)
    progs = {"dbl": gp.build_bn_pair_dbl(), "add": gp.build_bn_pair_add(), "madd": gp.build_bn_pair_madd()}
    sizes = {k: pr.emit_asm({n: "(%s)" % n for n in pr.order})[1]["instructions"] for k, pr in progs.items()}
    assert sizes["dbl"] < 1000 and sizes["add"] < 1950 and sizes["madd"] < 1550        # one lane: ~1500 / ~3500 / ~2400

    def tb(x):
        return bal(x * R % BP)

    def bv(regs, name):
        return sum(regs["%s.%d" % (name, i)] << (29 * i) for i in range(9)) * BRI % BP

    def jac(pt, rng):
        z = rng.randrange(1, BP)
        return (pt[0] * z * z % BP, pt[1] * z * z * z % BP, z)

    class BnPair:
        def __init__(self, X, Y, Z, rng):
            self.e, self.o = {}, {}
            put(self.e, "A", tb(X)); put(self.e, "B", tb(Y))
            put(self.o, "A", [rng.randrange(-(1 << 28), 1 << 28) for _ in range(9)]); put(self.o, "B", tb(Z))

        def point(self):
            zi = pow(bv(self.o, "B"), -1, BP)
            return (bv(self.e, "A") * zi * zi % BP, bv(self.e, "B") * zi * zi * zi % BP)

        def run(self, prog, ee=None, eo=None):
            e = {k: v for k, v in self.e.items() if k[:2] in ("A.", "B.")}
            o = {k: v for k, v in self.o.items() if k[:2] in ("A.", "B.")}
            e.update(ee or {}); o.update(eo or {})
            self.e, self.o = prog.run(e, o)
            for regs in (self.e, self.o):
                assert all(-(1 << 31) <= v < (1 << 31) for v in regs.values())
                if "AO.0" in regs:
                    for i in range(9):
                        regs["A.%d" % i], regs["B.%d" % i] = regs["AO.%d" % i], regs["BO.%d" % i]
    rng = random.Random(83)
    base = fixtures()["MSP1OU1"]["ipk"].h_sk
    for trial in range(2):
        k = rng.randrange(1 << 39, 1 << 40)
        st = BnPair(*jac(base, rng), rng)
        acc = base
        for bit in bin(k)[3:]:
            st.run(progs["dbl"])
            acc = io.g1_add(acc, acc)
            assert st.point() == acc
            if bit == "1":
                if rng.random() < 0.5:
                    X2, Y2, Z2 = jac(base, rng)
                    ce, co = {}, {}
                    put(ce, "C", tb(Z2)); put(ce, "D", [0] * 9)
                    put(co, "C", tb(X2)); put(co, "D", tb(Y2))
                    st.run(progs["add"], ce, co)
                else:
                    ce, co = {}, {}
                    put(ce, "C", tb(base[0])); put(ce, "D", [0] * 9)
                    put(co, "C", [0] * 9); put(co, "D", tb(base[1]))
                    st.run(progs["madd"], ce, co)
                acc = io.g1_add(acc, base)
                assert st.point() == acc


def test_limb_contracts_are_closed_under_every_program():
    """The proof that no 64-bit column and no 32-bit limb of the pair programs can wrap, whatever sequence of operations a kernel runs:
    gcn_dsl.Program.run_intervals executes each program on RANGES (interval arithmetic over the very instruction list the kernel runs) and
    the outputs of every program lie inside the state contract its inputs were drawn from (gen_pair_gcn.STATE_P256 / STATE_BN).  The
    generator refuses to emit headers otherwise; this is the same check, plus the one-lane consumers of a state (pair_x_equals_r29 squares Z,
    fe_is_zero multiplies H / RR by one)."""
    p256 = {"dbl": gp.build_pair_dbl(), "add": gp.build_pair_add(), "madd": gp.build_pair_madd()}
    bn = {"dbl": gp.build_bn_pair_dbl(), "add": gp.build_bn_pair_add(), "madd": gp.build_bn_pair_madd()}
    for progs_, C in ((p256, gp.STATE_P256), (bn, gp.STATE_BN)):
        U = gp.contracts_closed(progs_, C)
        # a state that starts as an affine point is inside the contract too
        assert all(C[k][0] <= gp.AFFINE[0] and gp.AFFINE[1] <= C[k][1] and C[k][2] <= gp.AFFINE[2] and gp.AFFINE[3] <= C[k][3] for k in "XYZ")
        assert U["X"][1] < 1 << 30 and U["Y"][1] <= 1 << 29
    # ... and a contract that is too generous is refused: the tool can say no
    wide = dict(gp.STATE_P256, Y=(-4 << 28, 4 << 28, -4 << 24, 3 << 24))
    with pytest.raises(OverflowError):
        gp.contracts_closed(p256, wide)
    # one-lane consumers: Z^2 (any lane's B), r Z^2, and a * 1 of fe_is_zero on differences of states
    sq, mu = gp.build_fe_sqr(), gp.build_fe_mul()
    for c in (gp.STATE_P256["Z"], gp.STATE_P256["Y"], gp.STATE_BN["Z"]):
        e = gp.fe_range("A", c)
        sq.run_intervals(e, e)
    d3 = (-3 << 28, 3 << 28, -8 << 24, 8 << 24)
    e = dict(gp.fe_range("A", d3)); e.update(gp.fe_range("B", gp.AFFINE))
    mu.run_intervals(e, e)
    # FP256BN: bn_nym_quad_part2 multiplies a state's X and Y by powers of 1 / Z (balanced digits)
    bmu = gp.build_bn_mul()
    for c in (gp.STATE_BN["X"], gp.STATE_BN["Y"], gp.STATE_BN["Z"]):
        e = dict(gp.fe_range("A", c)); e.update(gp.fe_range("B", (-(1 << 28), 1 << 28, -(2 << 24), 2 << 24)))
        bmu.run_intervals(e, e)


def test_the_interval_proof_rejects_the_withdrawn_doubling():
    """Round 6 committed a 739-instruction doubling for one commit: E squares gamma instead of 2 gamma and the factor rides in the addend's
    coefficient - Y3 = alpha (beta4 - X3) - 8 gamma^2.  It computed the right point on every input the suites have (this test runs it on a
    chain too), and it has NO closed contract: a Montgomery product leaves a b / R plus up to p, the -8 multiplies that slack by eight where
    the kept form multiplies it by two, and the worst-case value of Y grows from doubling to doubling.  The proof is what found it; this
    keeps the finding: the same instruction list, correct on numbers, refused on ranges."""
    from gcn_dsl import Program

    def withdrawn():
        p = Program("PAIR29_DBL_WITHDRAWN")
        A = p.fe("A", "io"); B = p.fe("B", "io")
        U1, U2, U3, W3, P1, P2, T0, T1, TD = (p.fe(n, "tmp") for n in ("U1", "U2", "U3", "W3", "P1", "P2", "T0", "T1", "TD"))
        p.sqr(U1, B, TD)                                  # E: gamma          O: delta
        p.swp_sub(P1, A, U1); p.swp_add(P2, A, U1)
        p.shl(T0, U1, 2)                                  # E: 4 gamma
        p.sel(P1, P1, A); p.sel(P2, P2, T0)
        p.mul(U2, P1, P2)                                 # E: beta4          O: m
        p.shladd(T0, U2, 1, U2)                           #                   O: alpha
        p.sel(W3, T0, U1)                                 # E: gamma
        p.swp(T0, U2)
        p.lane_const(TD[8], -2, 0)
        p.sqr(U3, W3, TD, T0, TD[8])                      # E: gamma^2        O: X3
        p.swp(A, U3); p.sub(T0, U2, A); p.swp(T1, W3); p.swp(P1, B); p.shl(P1, P1, 1)
        p.sel(P1, P1, T1); p.sel(P2, B, T0)
        p.lane_const(TD[0], 0, -8)
        p.mul(B, P1, P2, U3, TD[0])                       # E: Y3 = yy - 8 gamma^2     O: Z3
        return p
    bad = withdrawn()
    # right on numbers ...
    rng = random.Random(82)
    pt = po.pt_mul(rng.randrange(1, po.N), (po.GX, po.GY))
    st = Pair(*jac_of(pt, rng), rng)
    acc = pt
    for _ in range(12):
        st.run(bad)
        acc = po.pt_add(acc, acc)
        assert st.point() == acc
    # ... and without a contract: iterating the outputs back into the inputs never closes (the kept doubling closes in a few rounds)
    def closes(dbl):
        progs_ = {"dbl": dbl, "add": gp.build_pair_add(), "madd": gp.build_pair_madd()}
        C = {"X": gp.AFFINE, "Y": gp.AFFINE, "Z": gp.AFFINE}
        try:
            for _ in range(40):
                U = gp.state_outputs(progs_, C)
                new = {k: (min(C[k][0], U[k][0]), max(C[k][1], U[k][1]), min(C[k][2], U[k][2]), max(C[k][3], U[k][3])) for k in C}
                if new == C:
                    return True
                C = new
        except OverflowError:
            return False
        return False
    assert closes(gp.build_pair_dbl()) and not closes(bad)


def test_pair_programs_at_the_edge_of_their_limb_contracts(progs):
    """Every input digit at an extreme of the state contract (signs: all high, all low, alternating, random): the 64-bit column sums must
    not wrap - the interpreter's arithmetic is exact 64-bit, so a wrap shows as a wrong value against the formulas in big integers
    (evaluated on the VALUES the digits represent: such inputs are not curve points, the formulas do not care).  The interval proof above
    says this cannot fail; this is the same statement on concrete numbers."""
    rng = random.Random(80)
    S = gp.STATE_P256

    def edge(c, pattern):
        pick = {"hi": lambda i: 1, "lo": lambda i: 0, "alt": lambda i: i & 1, "rnd": lambda i: rng.randrange(2)}[pattern]
        return [c[1] if pick(i) else c[0] for i in range(8)] + [c[3] if pick(8) else c[2]]

    def v(d):
        return sum(x << (29 * i) for i, x in enumerate(d)) * RI % P

    def inside(regs, name, c):
        return all(c[0] <= regs["%s.%d" % (name, i)] <= c[1] for i in range(8)) and c[2] <= regs["%s.8" % name] <= c[3]
    negY = (-S["Y"][1], -S["Y"][0], -S["Y"][3], -S["Y"][2])
    for pattern in ("hi", "lo", "alt", "rnd", "rnd", "rnd", "rnd", "rnd"):
        X, Y, Z = edge(S["X"], pattern), edge(S["Y"], pattern), edge(S["Z"], pattern)
        e, o = {}, {}
        put(e, "A", X); put(e, "B", Y); put(o, "A", edge(gp.AFFINE, "rnd")); put(o, "B", Z)
        re, ro = progs["dbl"].run(e, o)
        x, y, z = v(X), v(Y), v(Z)
        g, d = y * y % P, z * z % P
        b4, al = 4 * x * g % P, 3 * (x - d) * (x + d) % P
        x3 = (al * al - 2 * b4) % P
        assert val(re, "A") == x3 and val(re, "B") == (al * (b4 - x3) - 8 * g * g) % P and val(ro, "B") == 2 * y * z % P
        assert inside(re, "A", S["X"]) and inside(re, "B", S["Y"]) and inside(ro, "B", S["Z"])
        X1, Y1, Z1, X2, Y2, Z2 = edge(S["X"], pattern), edge(S["Y"], pattern), edge(S["Z"], pattern), edge(S["X"], "rnd"), edge(negY, pattern), edge(S["Z"], "rnd")
        e, o = {}, {}
        put(e, "A", X1); put(e, "B", Y1); put(e, "C", Z2); put(e, "D", [0] * 9)
        put(o, "A", edge(gp.AFFINE, "rnd")); put(o, "B", Z1); put(o, "C", X2); put(o, "D", Y2)
        re, ro = progs["add"].run(e, o)
        x1, y1, z1, x2, y2, z2 = v(X1), v(Y1), v(Z1), v(X2), v(Y2), v(Z2)
        u1, u2 = x1 * z2 * z2 % P, x2 * z1 * z1 % P
        s1, s2 = y1 * z2 ** 3 % P, y2 * z1 ** 3 % P
        h, rr = (u2 - u1) % P, (s2 - s1) % P
        x3 = (rr * rr - h ** 3 - 2 * u1 * h * h) % P
        y3 = (rr * (u1 * h * h - x3) - s1 * h ** 3) % P
        z3 = z1 * z2 * h % P
        assert (val(re, "AO"), val(re, "BO"), val(ro, "BO")) == (x3, (-y3) % P, (-z3) % P), pattern           # (X, -Y, -Z) is the same point
        assert inside(re, "AO", S["X"]) and inside(re, "BO", S["Y"]) and inside(ro, "BO", S["Z"])
        X1, Y1, Z1, X2, Y2 = edge(S["X"], pattern), edge(S["Y"], pattern), edge(S["Z"], pattern), edge(gp.AFFINE, "rnd"), edge(gp.AFFINE, pattern)
        e, o = {}, {}
        put(e, "A", X1); put(e, "B", Y1); put(e, "C", X2); put(e, "D", [0] * 9)
        put(o, "A", edge(gp.AFFINE, "rnd")); put(o, "B", Z1); put(o, "C", [0] * 9); put(o, "D", Y2)
        re, ro = progs["madd"].run(e, o)
        x1, y1, z1, x2, y2 = v(X1), v(Y1), v(Z1), v(X2), v(Y2)
        u2, s2 = x2 * z1 * z1 % P, y2 * z1 ** 3 % P
        h, rr = (u2 - x1) % P, (s2 - y1) % P
        x3 = (rr * rr - h ** 3 - 2 * x1 * h * h) % P
        assert (val(re, "AO"), val(re, "BO"), val(ro, "BO")) == (x3, (rr * (x1 * h * h - x3) - y1 * h ** 3) % P, z1 * h % P), pattern
        assert inside(re, "AO", S["X"]) and inside(re, "BO", S["Y"]) and inside(ro, "BO", S["Z"])


def test_one_lane_programs_chain_and_contract():
    """one29_gcn.h (gen_pair_gcn.py one): the device's pt_dbl29 / pt_add29 / pt_add_mixed29 for one lane per signature.  The instruction
    lists behind the generated C++ run in the interpreter: a double-and-add chain against the oracle after every step (Jacobian and affine
    addends, negated ones too), the probes h / rr of the exceptional cases, and the interval proof that the one-lane state contract is closed."""
    rng = random.Random(81)
    G = (po.GX, po.GY)
    dbl, add, madd = gp.build_one_dbl(), gp.build_one_add(), gp.build_one_madd()
    gp.one_contracts_closed({"dbl": dbl, "add": add, "madd": madd}, gp.STATE_ONE)
    with pytest.raises(OverflowError):
        gp.one_contracts_closed({"dbl": dbl, "add": add, "madd": madd}, dict(gp.STATE_ONE, Y=(-4 << 28, 4 << 28, -4 << 24, 3 << 24)))

    def state_of(regs, names):
        return {("XYZ"[i] + ".%d" % l): regs["%s.%d" % (n, l)] for i, n in enumerate(names) for l in range(9)}
    for trial in range(3):
        base = po.pt_mul(rng.randrange(1, po.N), G)
        k = rng.randrange(1 << 39, 1 << 40)
        X, Y, Z = jac_of(base, rng)
        st = {}
        put(st, "X", to_fe(X)); put(st, "Y", to_fe(Y)); put(st, "Z", to_fe(Z))
        acc = base
        for bit in bin(k)[3:]:
            out, _ = dbl.run(st, st)
            st = state_of(out, ("X3", "Y3", "Z3"))
            acc = po.pt_add(acc, acc)
            assert affine(val(st, "X"), val(st, "Y"), val(st, "Z")) == acc
            if bit == "1":
                sign = rng.choice((1, -1))
                addend = (base[0], base[1] * sign % P)
                e = {k2.replace("X.", "X1.").replace("Y.", "Y1.").replace("Z.", "Z1."): v for k2, v in st.items()}
                if rng.random() < 0.5:
                    X2, Y2, Z2 = jac_of(base, rng)
                    y2 = to_fe(Y2) if sign == 1 else [-d for d in to_fe(Y2)]          # negated the way the kernels do it: digit by digit
                    put(e, "X2", to_fe(X2)); put(e, "Y2", y2); put(e, "Z2", to_fe(Z2))
                    out, _ = add.run(e, e)
                else:
                    put(e, "X2", to_fe(addend[0])); put(e, "Y2", to_fe(addend[1]))
                    out, _ = madd.run(e, e)
                assert val(out, "H") != 0 and val(out, "RR") != 0
                st = state_of(out, ("X3", "Y3", "Z3"))
                acc = po.pt_add(acc, addend)
                assert affine(val(st, "X"), val(st, "Y"), val(st, "Z")) == acc
            assert all(gp.STATE_ONE[c][0] <= st["%s.%d" % (c, l)] <= gp.STATE_ONE[c][1] for c in "XYZ" for l in range(8))
    # P == +-Q: h == 0, and rr == 0 exactly for the doubling case
    pt = po.pt_mul(rng.randrange(1, po.N), G)
    for sign in (1, -1):
        X1, Y1, Z1 = jac_of(pt, rng)
        X2, Y2, Z2 = jac_of((pt[0], pt[1] * sign % P), rng)
        e = {}
        for n, v_ in (("X1", X1), ("Y1", Y1), ("Z1", Z1), ("X2", X2), ("Y2", Y2), ("Z2", Z2)):
            put(e, n, to_fe(v_))
        out, _ = add.run(e, e)
        assert val(out, "H") == 0 and (val(out, "RR") == 0) == (sign == 1)
        e = {}
        for n, v_ in (("X1", X1), ("Y1", Y1), ("Z1", Z1), ("X2", pt[0]), ("Y2", pt[1] * sign % P)):
            put(e, n, to_fe(v_))
        out, _ = madd.run(e, e)
        assert val(out, "H") == 0 and (val(out, "RR") == 0) == (sign == 1)


def test_pair_add_reports_the_exceptional_cases(progs):
    rng = random.Random(78)
    G = (po.GX, po.GY)
    pt = po.pt_mul(rng.randrange(1, po.N), G)
    for sign in (1, -1):
        st = Pair(*jac_of(pt, rng), rng)
        X2, Y2, Z2 = jac_of((pt[0], pt[1] * sign % P), rng)
        ce, co = {}, {}
        put(ce, "C", to_fe(Z2)); put(ce, "D", [0] * 9)
        put(co, "C", to_fe(X2)); put(co, "D", to_fe(Y2))
        st.run(progs["add"], ce, co)
        assert val(st.e, "H") == 0                      # same x: P == +-Q
        assert (val(st.e, "RR") == 0) == (sign == 1)    # and same y: the doubling case


@pytest.mark.gpu
def test_generated_streams_on_gpu_match_the_interpreter_register_for_register(progs):
    """One wavefront (32 lane pairs) runs each generated asm statement; every output limb of every lane - including the
    don't-care lanes - must equal what gcn_dsl.Program.run() computes for the same inputs."""
    import ctypes

    import numpy as np
    lib = ctypes.CDLL(os.path.join(ROOT, "fabric-mod_amd", "lib", "libfabgpu_gputest.so"))
    rng = random.Random(79)
    G = (po.GX, po.GY)
    for op, name in ((0, "dbl"), (1, "add"), (2, "madd")):
        inp = np.zeros((64, 36), dtype=np.int32)
        want = {}
        for k in range(32):
            p1 = po.pt_mul(rng.randrange(1, po.N), G)
            p2 = po.pt_mul(rng.randrange(1, po.N), G)
            X1, Y1, Z1 = jac_of(p1, rng)
            X2, Y2, Z2 = jac_of(p2, rng)
            junk = lambda: [rng.randrange(-(1 << 28), 1 << 28) for _ in range(9)]
            S = gp.STATE_P256
            anyof = lambda c: [rng.choice((c[0], c[1], rng.randrange(c[0], c[1] + 1))) for _ in range(8)] + [rng.randrange(c[2], c[3] + 1)]
            if k & 1:      # not curve points: digits anywhere in the state contract, its extremes included (the comparison is with the interpreter)
                e = {"A": anyof(S["X"]), "B": anyof(S["Y"])}
                o = {"A": junk(), "B": anyof(S["Z"])}
                if name == "add":
                    e.update(C=anyof(S["Z"]), D=junk()); o.update(C=anyof(S["X"]), D=anyof(S["Y"]))
            else:
                e = {"A": to_fe(X1), "B": to_fe(Y1)}
                o = {"A": junk(), "B": to_fe(Z1)}
                if name == "add":
                    e.update(C=to_fe(Z2), D=junk()); o.update(C=to_fe(X2), D=to_fe(Y2))
            if name == "madd":
                e.update(C=to_fe(p2[0]), D=junk()); o.update(C=junk(), D=to_fe(p2[1]))
            elif name == "dbl":
                e.update(C=junk(), D=junk()); o.update(C=junk(), D=junk())
            for lane, regs in ((2 * k, e), (2 * k + 1, o)):
                inp[lane] = regs["A"] + regs["B"] + regs["C"] + regs["D"]
            re, ro = {}, {}
            for nm in "ABCD":
                put(re, nm, e[nm]); put(ro, nm, o[nm])
            want[k] = progs[name].run(re, ro)
        out = np.zeros((64, 36), dtype=np.int32)
        rc = lib.gputest_pair_op(op, inp.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        names = (["A", "B"] if name == "dbl" else ["AO", "BO"]) + (["H", "RR"] if name == "add" else [])     # (the sums are out of place)
        bad = set()
        for k in range(32):
            for lane, regs in ((2 * k, want[k][0]), (2 * k + 1, want[k][1])):
                for j, nm in enumerate(names):
                    got = [int(v) for v in out[lane, 9 * j:9 * j + 9]]
                    exp = [regs["%s.%d" % (nm, i)] for i in range(9)]
                    if got != exp:
                        bad.add((nm, "odd" if lane & 1 else "even", tuple(i for i in range(9) if got[i] != exp[i])))
        assert not bad, (name, sorted(bad))


@pytest.mark.gpu
def test_bn_pair_streams_on_gpu_match_the_interpreter_register_for_register():
    """PAIRBN_DBL / ADD / MADD (the point operations of bn_quad29.h): one wavefront runs each generated asm statement, every output
    limb of every lane must equal the interpreter's."""
    import ctypes

    import numpy as np
    import gen_bn_consts as bc
    import idemix_oracle as io
    from idemix_common import fixtures
    BP = bc.P
    lib = ctypes.CDLL(os.path.join(ROOT, "fabric-mod_amd", "lib", "libfabgpu_gputest.so"))
    progs = {"dbl": gp.build_bn_pair_dbl(), "add": gp.build_bn_pair_add(), "madd": gp.build_bn_pair_madd()}
    rng = random.Random(91)
    base = fixtures()["MSP2OU1"]["ipk"].h_rand

    def tb(x):
        return bal(x * R % BP)

    def jac(pt):
        z = rng.randrange(1, BP)
        return (pt[0] * z * z % BP, pt[1] * z * z * z % BP, z)
    for op, name in ((4, "dbl"), (5, "add"), (6, "madd")):
        inp = np.zeros((64, 36), dtype=np.int32)
        want = {}
        for k in range(32):
            p1 = io.g1_mul(base, rng.randrange(1, io.R))
            p2 = io.g1_mul(base, rng.randrange(1, io.R))
            X1, Y1, Z1 = jac(p1)
            X2, Y2, Z2 = jac(p2)
            junk = lambda: [rng.randrange(-(1 << 28), 1 << 28) for _ in range(9)]
            e = {"A": tb(X1), "B": tb(Y1)}
            o = {"A": junk(), "B": tb(Z1)}
            if name == "add":
                e.update(C=tb(Z2), D=junk()); o.update(C=tb(X2), D=tb(Y2))
            elif name == "madd":
                e.update(C=tb(p2[0]), D=junk()); o.update(C=junk(), D=tb(p2[1]))
            else:
                e.update(C=junk(), D=junk()); o.update(C=junk(), D=junk())
            for lane, regs in ((2 * k, e), (2 * k + 1, o)):
                inp[lane] = regs["A"] + regs["B"] + regs["C"] + regs["D"]
            re, ro = {}, {}
            for nm in "ABCD":
                put(re, nm, e[nm]); put(ro, nm, o[nm])
            want[k] = progs[name].run(re, ro)
        out = np.zeros((64, 36), dtype=np.int32)
        assert lib.gputest_pair_op(op, inp.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)) == 0
        names = (["A", "B"] if name == "dbl" else ["AO", "BO"]) + (["H", "RR"] if name == "add" else [])     # (the sums are out of place)
        bad = set()
        for k in range(32):
            for lane, regs in ((2 * k, want[k][0]), (2 * k + 1, want[k][1])):
                for j, nm in enumerate(names):
                    got = [int(v) for v in out[lane, 9 * j:9 * j + 9]]
                    exp = [regs["%s.%d" % (nm, i)] for i in range(9)]
                    if got != exp:
                        bad.add((nm, "odd" if lane & 1 else "even", tuple(i for i in range(9) if got[i] != exp[i])))
        assert not bad, (name, sorted(bad))


def test_tracked_generated_headers_are_what_the_generators_produce():
    """pair29_gcn.h, one29_gcn.h, pair29_bn_gcn.h, fe29_gcn.h, bn29_gcn.h and bn29_consts.h are tracked AND have Makefile rules: a header that is
    older than its generator by content (not by timestamp) would compile silently.  Regenerate each and compare byte for byte."""
    import subprocess
    csrc = os.path.join(ROOT, "fabric-mod_amd", "csrc")
    for args, header in ((["gen_pair_gcn.py", "field"], "fe29_gcn.h"), (["gen_pair_gcn.py"], "pair29_gcn.h"), (["gen_pair_gcn.py", "one"], "one29_gcn.h"), (["gen_pair_gcn.py", "bnfield"], "bn29_gcn.h"),
                         (["gen_pair_gcn.py", "bnpair"], "pair29_bn_gcn.h"), (["gen_bn_consts.py"], "bn29_consts.h")):
        fresh = subprocess.run([sys.executable] + args, cwd=csrc, check=True, capture_output=True).stdout
        assert fresh == open(os.path.join(csrc, header), "rb").read(), "%s is stale: run make -C fabric-mod_amd/csrc %s" % (header, header)
