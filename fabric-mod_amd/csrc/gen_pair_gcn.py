#!/usr/bin/env python3
"""Generates pair29_gcn.h: point doubling / addition / mixed addition on secp256r1 with TWO LANES PER SIGNATURE.

Why: BASELINE config 2 (30 000 signatures) is 469 wavefronts for 1024 SIMDs and every wave issues one VALU instruction per
~4.2 cycles whatever it is - the one-signature-per-lane kernel is bound by the LENGTH of its instruction stream.  The
eight field products of a doubling have dependency depth four, the sixteen of an addition depth eight: on a lane pair
(even lane "E", odd lane "O", exchanged with DPP quad_perm:[1,0,3,2]) the stream is 732 / 1364 / 1104 instructions (round 6; 787 / 1463 / 1185 in rounds 2-5)
instead of 1339 / 2678 / 1900.

Lane roles are static.  Between operations   E holds A = X, B = Y   and   O holds B = Z   (O's A is don't-care).
Every step is ONE field product executed by both lanes on lane-specific operands (v_cndmask on the lane-parity mask
builds them); squarings are paired with squarings so the cheaper square stream is kept.  The formulas and limb bounds
are exactly those of p256_verify29.h (pt_dbl29 / pt_add29 / pt_add_mixed29).

tests/test_pair_programs.py runs these very programs in gcn_dsl.Program.run() against big-integer point arithmetic.

Run:  python3 gen_pair_gcn.py > pair29_gcn.h ; python3 gen_pair_gcn.py field > fe29_gcn.h   (the Makefile does this)
The single-lane field product / square of fe29.h are generated from the same DSL (build_fe_mul / build_fe_sqr).
"""
from gcn_dsl import GenericField, Program

# Product steps (numbered in program order) whose output digits are left unsigned - [0, 2^29), two instructions per high column instead
# of three.  Which steps can is decided by Program.run_intervals over the closed state contract (tests/test_pair_programs.py).
UNSIGNED_DBL = (2, 3)
UNSIGNED_ADD = (0, 1, 2, 3, 4, 5, 6, 7)
UNSIGNED_MADD = (0, 1, 3, 4, 5)
UNSIGNED_BN_DBL = (2, 3)
UNSIGNED_BN_ADD = (0, 1, 2, 3, 4, 5, 6, 7)
UNSIGNED_BN_MADD = (0, 1, 3, 4, 5)


def build_pair_dbl(uns=UNSIGNED_DBL):
    """(A, B) <- 2 * (A, B).   in: L(X) <= 2, L(Y) <= 3, L(Z) <= 2 (2Y x Z: 6 x 2);  out: L(X) = 1, L(Y) = 1, L(Z) = 1.
    Round 6: X3 = alpha^2 - 2 beta4 and Y3 = yy - 2 gg leave their PRODUCTS already subtracted and with balanced digits (nine more
    MACs each on the high columns, Program._columns' addend) instead of a subtraction, a carry pass and a select after them: 746
    instructions, were 787."""
    p = Program("PAIR29_DBL")
    A = p.fe("A", "io")
    B = p.fe("B", "io")
    U1 = p.fe("U1", "tmp")
    U2 = p.fe("U2", "tmp")
    U3 = p.fe("U3", "tmp")
    W3 = p.fe("W3", "tmp")
    P1 = p.fe("P1", "tmp")
    P2 = p.fe("P2", "tmp")
    T0 = p.fe("T0", "tmp")
    T1 = p.fe("T1", "tmp")
    TD = p.fe("TD", "tmp")
    p.sqr(U1, B, TD, unsigned=0 in uns)                 # E: gamma = Y^2            O: delta = Z^2
    p.swp_sub(P1, A, U1)             #                           O: X - delta
    p.swp_add(P2, A, U1)             #                           O: X + delta
    p.shl(T0, A, 2)                  # E: 4X
    p.sel(P1, P1, T0)
    p.sel(P2, P2, U1)                # E: gamma
    p.mul(U2, P1, P2, unsigned=1 in uns)                # E: beta4 = 4 X gamma      O: m = (X - delta)(X + delta)
    p.shladd(T0, U2, 1, U2)          #                           O: alpha = 3m
    p.shl(T1, U1, 1)                 # E: 2 gamma
    p.sel(W3, T0, T1)
    p.swp(T0, U2)                    #                           O: beta4
    p.lane_const(TD[8], -2, 0)       # (the square's scratch uses TD[0..7])
    p.sqr(U3, W3, TD, T0, TD[8], unsigned=2 in uns)     # E: gg = 4 gamma^2         O: X3 = alpha^2 - 2 beta4
    p.swp(A, U3)                     # E: X3                     (O: gg - its A is don't-care)
    p.sub(T0, U2, A)                 # E: beta4 - X3         (L2)
    p.swp(T1, W3)                    # E: alpha
    p.swp(P1, B)                     #                           O: Y
    p.shl(P1, P1, 1)                 #                           O: 2Y
    p.sel(P1, P1, T1)
    p.sel(P2, B, T0)                 #                           O: Z
    p.lane_const(TD[0], 0, -2)
    p.mul(B, P1, P2, U3, TD[0], unsigned=3 in uns)      # E: Y3 = alpha (beta4 - X3) - 2 gg      O: Z3 = 2 Y Z
    return p


def build_pair_add(name="PAIR29_ADD", field=None, uns=UNSIGNED_ADD):
    """(AO, BO) <- (A, B) + P2 with P2 handed over CROSSED:  E: C = Z2,  O: C = X2, D = Y2.
    in: L(X1) <= 2, L(Y1) <= 3, L(Z1) = 1, L(X2) = 1, L(Y2) <= 3, L(Z2) = 1;  out: L(X) = 1, L(Y) = 1, L(Z) = 1
    (round 6: Y3 leaves its last product already subtracted, Program._columns' addend).
    H (E: h = u2 - u1, O: -h) and RR (E: s2 - s1) are left for the caller's P == +-Q test (a test for zero)."""
    p = Program(name, field)
    AO = p.fe("AO", "tmp")         # the sum, out of place (round 6): the caller keeps the addend-free state for digits that are zero
    BO = p.fe("BO", "tmp")         # without copying it first
    A = p.fe("A", "in")
    B = p.fe("B", "in")
    H = p.fe("H", "tmp")
    RR = p.fe("RR", "tmp")
    W = p.fe("W", "tmp")
    U1 = p.fe("U1", "tmp")
    U2 = p.fe("U2", "tmp")
    U3 = p.fe("U3", "tmp")
    U4 = p.fe("U4", "tmp")
    U6 = p.fe("U6", "tmp")
    P1 = p.fe("P1", "tmp")
    P2 = p.fe("P2", "tmp")
    T0 = p.fe("T0", "tmp")
    T1 = p.fe("T1", "tmp")
    TD = p.fe("TD", "tmp")
    C = p.fe("C", "in")
    D = p.fe("D", "in")
    p.sel(W, B, C)                   # E: Z2                     O: Z1
    p.sqr(U1, W, TD, unsigned=0 in uns)                 # E: z2z2                   O: z1z1
    p.sel(P1, C, A)                  # E: X1                     O: X2
    p.mul(U2, P1, U1, unsigned=1 in uns)                # E: u1 = X1 z2z2           O: u2 = X2 z1z1
    p.mul(U3, W, U1, unsigned=2 in uns)                 # E: Z2^3                   O: Z1^3
    p.sel(P1, D, B)                  # E: Y1                     O: Y2
    p.mul(U4, P1, U3, unsigned=3 in uns)                # E: s1                     O: s2
    p.swp_sub(H, U2, U2)             # E: h = u2 - u1            O: -h     (round 6: O keeps -h; what it costs is a sign, see the last step)
    p.swp_sub(RR, U4, U4)            # E: rr = s2 - s1           O: -rr
    p.sel(P1, RR, H)
    U5 = W                           # W is dead
    p.sqr(U5, P1, TD, unsigned=4 in uns)                # E: hh                     O: r2 = rr^2
    p.sel(P1, H, U2)                 # E: u1                     O: -h
    p.bce(P2, U5)                    # hh on both lanes
    p.mul(U6, P1, P2, unsigned=5 in uns)                # E: v = u1 hh              O: -hhh
    p.shl(T0, U6, 1)                 # E: 2v
    p.swp_sub(T0, U6, T0)            # E: -hhh - 2v
    p.swp_add(T1, U5, T0)            # E: r2 - hhh - 2v      (L4)
    p.wnorm(AO, T1, TD)               # E: X3
    p.sub(T0, AO, U6)                 # E: X3 - v             (L2)
    p.swp(T1, C)                     #                           O: Z2
    p.swp(P2, U6)                    # E: -hhh
    p.sel(P1, B, U4)                 # E: s1                     O: Z1
    p.sel(P2, T1, P2)
    U7 = U1                          # U1 is dead
    p.mul(U7, P1, P2, unsigned=6 in uns)                # E: -y2 = s1 (-hhh)        O: zz = Z1 Z2
    p.sel(P1, U7, RR)                # E: rr                     O: zz
    p.sel(P2, H, T0)                 # E: X3 - v                 O: -h
    p.lane_const(TD[0], 0, -1)
    p.mul(BO, P1, P2, U7, TD[0], unsigned=7 in uns)      # E: -Y3 = rr (X3 - v) + y2    O: -Z3 = zz (-h)        (X3, -Y3, -Z3) is the same point as (X3, Y3, Z3)
    return p


def build_pair_madd(name="PAIR29_MADD", field=None, uns=UNSIGNED_MADD):
    """(AO, BO) <- (A, B) + (x2, y2) affine, handed over as  E: C = x2,  O: D = y2.
    in: L(X1) = 1, L(Y1) <= 2, L(Z1) = 1, x2 / y2 normalised;  out: L(X) = 1, L(Y) = 2, L(Z) = 1."""
    p = Program(name, field)
    AO = p.fe("AO", "tmp")         # out of place, as in the addition
    BO = p.fe("BO", "tmp")
    A = p.fe("A", "in")
    B = p.fe("B", "in")
    U1 = p.fe("U1", "tmp")
    U2 = p.fe("U2", "tmp")
    U3 = p.fe("U3", "tmp")
    U4 = p.fe("U4", "tmp")
    H = p.fe("H", "tmp")
    RR = p.fe("RR", "tmp")
    P1 = p.fe("P1", "tmp")
    P2 = p.fe("P2", "tmp")
    T0 = p.fe("T0", "tmp")
    T1 = p.fe("T1", "tmp")
    TD = p.fe("TD", "tmp")
    C = p.fe("C", "in")
    D = p.fe("D", "in")
    p.sqr(U1, B, TD, unsigned=0 in uns)                 #                           O: z1z1
    p.sel(P1, B, C)                  # E: x2                     O: Z1
    p.bco(P2, U1)                    # z1z1 on both lanes
    p.mul(U2, P1, P2, unsigned=1 in uns)                # E: u2 = x2 z1z1           O: Z1^3
    p.sub(H, U2, A)                  # E: h = u2 - X1        (L2)
    p.sel(P1, D, H)                  # E: h                      O: y2
    p.sel(P2, U2, H)                 # E: h                      O: Z1^3
    p.mul(U3, P1, P2, unsigned=2 in uns)                # E: hh                     O: s2
    p.swp_sub(RR, B, U3)             #                           O: Y1 - s2 = -rr   (L3: round 6, every producer of a state leaves L(Y) <= 2 - no carry pass)
    p.sel(P1, RR, U3)                # E: hh                     O: -rr
    p.sel(P2, RR, H)                 # E: h                      O: -rr
    p.mul(U4, P1, P2, unsigned=3 in uns)                # E: hhh                    O: r2
    p.swp(T0, H)                     #                           O: h
    p.sel(P1, B, A)                  # E: X1                     O: Z1
    p.sel(P2, T0, U3)                # E: hh                     O: h
    U5 = U1                          # U1 is dead
    p.mul(U5, P1, P2, unsigned=4 in uns)                # E: v = X1 hh              O: Z3 = Z1 h
    p.shl(T0, U5, 1)                 # E: 2v
    p.swp_sub(T1, U4, U4)            # E: r2 - hhh
    p.sub(T1, T1, T0)                # E: r2 - hhh - 2v      (L4)
    p.wnorm(AO, T1, TD)               # E: X3
    p.sub(T0, AO, U5)                 # E: X3 - v             (L2)
    p.swp(T1, T0)                    #                           O: X3 - v
    p.sel(P1, RR, B)                 # E: Y1                     O: -rr
    p.sel(P2, T1, U4)                # E: hhh                    O: X3 - v
    U6 = U3                          # U3 is dead
    p.mul(U6, P1, P2, unsigned=5 in uns)                # E: y2 = Y1 hhh            O: y1 = (-rr)(X3 - v)
    p.swp_sub(T0, U6, U6)            # E: Y3 = y1 - y2       (L2)
    p.sel(BO, U5, T0)                 #                           O: Z3
    return p


def build_fe_mul():
    """R = A * B / 2^261 mod p (fe29.h fe_mul_body); R must not alias A or B."""
    p = Program("FE29_GCN_MUL")
    R = p.fe("R", "tmp")
    A = p.fe("A", "in")
    B = p.fe("B", "in")
    p.mul(R, A, B)
    return p


def build_fe_sqr():
    """R = A * A / 2^261 mod p (fe29.h fe_sqr_body); T: scratch for the doubled limbs."""
    p = Program("FE29_GCN_SQR")
    R = p.fe("R", "tmp")
    T = p.fe("T", "tmp")
    A = p.fe("A", "in")
    p.sqr(R, A, T)
    return p


def bn_field():
    import gen_bn_consts as c
    return GenericField(c.balanced(c.P), (-pow(c.P, -1, 1 << 29)) % (1 << 29))


def build_bn_mul():
    """R = A * B / 2^261 mod the FP256BN prime (bn29.h); R must not alias A or B."""
    p = Program("BN29_GCN_MUL", bn_field())
    R = p.fe("R", "tmp")
    A = p.fe("A", "in")
    B = p.fe("B", "in")
    p.mul(R, A, B)
    return p


def build_bn_sqr():
    p = Program("BN29_GCN_SQR", bn_field())
    R = p.fe("R", "tmp")
    T = p.fe("T", "tmp")
    A = p.fe("A", "in")
    p.sqr(R, A, T)
    return p


# ---- limb contracts, proved by interval arithmetic (Program.run_intervals) -------------------------------------------------
# The STATE of a lane pair between point operations: digits 0..7 of X (E's A), Y (E's B), Z (O's B) as (lo, hi) and the top digit - which
# carries the value's magnitude - as (top lo, top hi).  state_outputs() runs the three programs on ranges: every table entry is a state that
# was stored earlier (Y possibly negated), a comb entry is an affine point in Montgomery form.  contracts_closed() is the proof that
# whatever sequence of operations a kernel runs, no 64-bit column and no 32-bit limb ever wraps: the outputs of every program lie inside
# the contract its inputs were drawn from.  emit() refuses to generate headers otherwise; tests/test_pair_programs.py runs it too.
B28, B24 = 1 << 28, 1 << 24
STATE_P256 = {"X": (-(B28 + 4), 2 * B28 - 1, -5 * B24, 4 * B24), "Y": (-2 * B28, 2 * B28, -4 * B24, 3 * B24), "Z": (-B28, 2 * B28 - 1, -B24, 2 * B24)}
STATE_BN = {"X": (-(B28 + 4), 2 * B28 - 1, -5 * B24, 4 * B24), "Y": (-2 * B28, 2 * B28, -4 * B24, 3 * B24), "Z": (-2 * B28, 2 * B28, -B24, 3 * B24)}
AFFINE = (-B28, B28 - 1, -B24, 2 * B24)  # fe_to_mont of a residue (comb entries, the key): balanced digits, a value in (-p, 2p) - generous


def fe_range(name, c):
    d = {"%s.%d" % (name, i): (c[0], c[1]) for i in range(8)}
    d["%s.8" % name] = (c[2], c[3])
    return d


def state_outputs(progs, C):
    """progs: {"dbl", "add", "madd"}; C: a state contract.  Returns {"X" | "Y" | "Z": (lo, hi, top lo, top hi)}, the union over the three
    programs of what they leave; raises OverflowError where a column or a limb can wrap."""
    def state():
        e, o = {}, {}
        e.update(fe_range("A", C["X"])); e.update(fe_range("B", C["Y"])); o.update(fe_range("B", C["Z"]))
        return e, o
    y = C["Y"]
    ny = (min(y[0], -y[1]), max(y[1], -y[0]), min(y[2], -y[3]), max(y[3], -y[2]))
    outs = []
    e, o = state()
    outs.append(progs["dbl"].run_intervals(e, o))
    e, o = state()
    e.update(fe_range("C", C["Z"])); o.update(fe_range("C", C["X"])); o.update(fe_range("D", ny))
    outs.append(progs["add"].run_intervals(e, o))
    e, o = state()
    e.update(fe_range("C", AFFINE)); o.update(fe_range("D", AFFINE))
    outs.append(progs["madd"].run_intervals(e, o))
    U = {}
    for re_, ro in outs:
        oa, ob = ("AO", "BO") if "AO.0" in re_ else ("A", "B")
        for nm, regs, fe in (("X", re_, oa), ("Y", re_, ob), ("Z", ro, ob)):
            lo = min(regs["%s.%d" % (fe, i)][0] for i in range(8))
            hi = max(regs["%s.%d" % (fe, i)][1] for i in range(8))
            t = regs[fe + ".8"]
            cur = U.get(nm)
            U[nm] = (lo, hi, t[0], t[1]) if cur is None else (min(cur[0], lo), max(cur[1], hi), min(cur[2], t[0]), max(cur[3], t[1]))
    return U


def contracts_closed(progs, C):
    U = state_outputs(progs, C)
    for k in ("X", "Y", "Z"):
        u, c = U[k], C[k]
        if not (c[0] <= u[0] and u[1] <= c[1] and c[2] <= u[2] and u[3] <= c[3]):
            raise OverflowError("%s leaves the state contract: %r not inside %r" % (k, u, c))
    return U


# ---- one lane per signature: the point operations of p256_verify29.h / ec29.h for the device (round 6) ----------------------------------
# The same formulas as the C++ bodies (which stay: host build, specification), with this round's shortcuts - subtractions made inside the
# products, no carry passes, unsigned digits where proved safe - emitted by Program.emit_cxx as device functions whose products are one
# asm statement each.  Which product steps leave unsigned digits: UNSIGNED_ONE_* (decided by one_contracts_closed, as for the pairs).
UNSIGNED_ONE_DBL = (0, 3, 4, 5, 6, 7)
UNSIGNED_ONE_ADD = tuple(range(16))
UNSIGNED_ONE_MADD = tuple(range(11))


def build_one_dbl(uns=None):
    """(X3, Y3, Z3) = 2 (X, Y, Z), a = -3 (pt_dbl29): 4M + 4S."""
    uns = UNSIGNED_ONE_DBL if uns is None else uns
    p = Program("ONE29_DBL")
    X3, Y3, Z3 = p.fe("X3", "tmp"), p.fe("Y3", "tmp"), p.fe("Z3", "tmp")
    X, Y, Z = p.fe("X", "in"), p.fe("Y", "in"), p.fe("Z", "in")
    DL, GM, T1, T2, M, AL, B4, GG, TD = (p.fe(n, "tmp") for n in ("DL", "GM", "T1", "T2", "M", "AL", "B4", "GG", "TD"))
    p.sqr(DL, Z, TD, unsigned=0 in uns)                  # delta = Z^2
    p.sqr(GM, Y, TD, unsigned=1 in uns)                  # gamma = Y^2
    p.sub(T1, X, DL)
    p.add(T2, X, DL)
    p.mul(M, T1, T2, unsigned=2 in uns)                  # (X - delta)(X + delta)
    p.shladd(AL, M, 1, M)                                # alpha = 3 m
    p.shl(T1, X, 2)                                      # 4X
    p.mul(B4, T1, GM, unsigned=3 in uns)                 # beta4 = 4 X gamma
    p.sqr(X3, AL, TD, [(B4, -2)], unsigned=4 in uns)     # X3 = alpha^2 - 2 beta4
    p.shl(T2, Y, 1)
    p.mul(Z3, T2, Z, unsigned=5 in uns)                  # Z3 = 2 Y Z
    p.shl(T1, GM, 1)
    p.sqr(GG, T1, TD, unsigned=6 in uns)                 # 4 gamma^2
    p.sub(T1, B4, X3)
    p.mul(Y3, AL, T1, [(GG, -2)], unsigned=7 in uns)     # Y3 = alpha (beta4 - X3) - 8 gamma^2
    return p


def build_one_add(uns=None):
    """(X3, Y3, Z3) = (X1, Y1, Z1) + (X2, Y2, Z2) (pt_add29): 12M + 4S; H = u2 - u1 and RR = s2 - s1 for the caller's P == +-Q test."""
    uns = UNSIGNED_ONE_ADD if uns is None else uns
    p = Program("ONE29_ADD")
    X3, Y3, Z3, H, RR = (p.fe(n, "tmp") for n in ("X3", "Y3", "Z3", "H", "RR"))
    X1, Y1, Z1, X2, Y2, Z2 = (p.fe(n, "in") for n in ("X1", "Y1", "Z1", "X2", "Y2", "Z2"))
    Z11, Z22, U1, U2, S1, S2, T, HH, HHH, V, TD = (p.fe(n, "tmp") for n in ("Z11", "Z22", "U1", "U2", "S1", "S2", "T", "HH", "HHH", "V", "TD"))
    p.sqr(Z11, Z1, TD, unsigned=0 in uns)
    p.sqr(Z22, Z2, TD, unsigned=1 in uns)
    p.mul(U1, X1, Z22, unsigned=2 in uns)
    p.mul(U2, X2, Z11, unsigned=3 in uns)
    p.mul(T, Z2, Z22, unsigned=4 in uns)
    p.mul(S1, Y1, T, unsigned=5 in uns)
    T2 = p.fe("T2", "tmp")
    p.mul(T2, Z1, Z11, unsigned=6 in uns)
    p.mul(S2, Y2, T2, unsigned=7 in uns)
    p.sub(H, U2, U1)
    p.sub(RR, S2, S1)
    p.sqr(HH, H, TD, unsigned=8 in uns)
    p.mul(HHH, HH, H, unsigned=9 in uns)
    p.mul(V, U1, HH, unsigned=10 in uns)
    p.sqr(X3, RR, TD, [(HHH, -1), (V, -2)], unsigned=11 in uns)      # X3 = rr^2 - h^3 - 2 v
    p.sub(T, V, X3)
    Y2P = p.fe("Y2P", "tmp")
    p.mul(Y2P, S1, HHH, unsigned=12 in uns)
    p.mul(Y3, RR, T, [(Y2P, -1)], unsigned=13 in uns)                # Y3 = rr (v - X3) - s1 h^3
    ZZ = p.fe("ZZ", "tmp")
    p.mul(ZZ, Z1, Z2, unsigned=14 in uns)
    p.mul(Z3, ZZ, H, unsigned=15 in uns)
    return p


def build_one_madd(uns=None):
    """(X3, Y3, Z3) = (X1, Y1, Z1) + (x2, y2) affine (pt_add_mixed29): 8M + 3S."""
    uns = UNSIGNED_ONE_MADD if uns is None else uns
    p = Program("ONE29_MADD")
    X3, Y3, Z3, H, RR = (p.fe(n, "tmp") for n in ("X3", "Y3", "Z3", "H", "RR"))
    X1, Y1, Z1, X2, Y2 = (p.fe(n, "in") for n in ("X1", "Y1", "Z1", "X2", "Y2"))
    Z11, U2, S2, T, HH, HHH, V, Y2P, TD = (p.fe(n, "tmp") for n in ("Z11", "U2", "S2", "T", "HH", "HHH", "V", "Y2P", "TD"))
    p.sqr(Z11, Z1, TD, unsigned=0 in uns)
    p.mul(U2, X2, Z11, unsigned=1 in uns)
    p.mul(T, Z1, Z11, unsigned=2 in uns)
    p.mul(S2, Y2, T, unsigned=3 in uns)
    p.sub(H, U2, X1)
    p.sub(RR, S2, Y1)                                                # (no carry pass: every producer leaves |Y| digits <= 2 * 2^28)
    p.sqr(HH, H, TD, unsigned=4 in uns)
    p.mul(HHH, HH, H, unsigned=5 in uns)
    p.mul(V, X1, HH, unsigned=6 in uns)
    p.sqr(X3, RR, TD, [(HHH, -1), (V, -2)], unsigned=7 in uns)
    p.sub(T, V, X3)
    p.mul(Y2P, Y1, HHH, unsigned=8 in uns)
    p.mul(Y3, RR, T, [(Y2P, -1)], unsigned=9 in uns)
    p.mul(Z3, Z1, H, unsigned=10 in uns)
    return p


STATE_ONE = {"X": (-B28, 2 * B28 - 1, -5 * B24, 4 * B24), "Y": (-B28, 2 * B28 - 1, -4 * B24, 3 * B24), "Z": (-B28, 2 * B28 - 1, -B24, 2 * B24)}


def one_state_outputs(progs, C):
    """As state_outputs, for the one-lane programs: all three coordinates on the same lane."""
    y = C["Y"]
    ny = (min(y[0], -y[1]), max(y[1], -y[0]), min(y[2], -y[3]), max(y[3], -y[2]))
    outs = []
    e = {}
    e.update(fe_range("X", C["X"])); e.update(fe_range("Y", C["Y"])); e.update(fe_range("Z", C["Z"]))
    outs.append(progs["dbl"].run_intervals(e, e)[0])
    for y2 in (C["Y"], ny):
        e = {}
        e.update(fe_range("X1", C["X"])); e.update(fe_range("Y1", C["Y"])); e.update(fe_range("Z1", C["Z"]))
        e.update(fe_range("X2", C["X"])); e.update(fe_range("Y2", y2)); e.update(fe_range("Z2", C["Z"]))
        outs.append(progs["add"].run_intervals(e, e)[0])
    e = {}
    e.update(fe_range("X1", C["X"])); e.update(fe_range("Y1", C["Y"])); e.update(fe_range("Z1", C["Z"]))
    e.update(fe_range("X2", AFFINE)); e.update(fe_range("Y2", AFFINE))
    outs.append(progs["madd"].run_intervals(e, e)[0])
    U = {}
    for regs in outs:
        for nm, fe in (("X", "X3"), ("Y", "Y3"), ("Z", "Z3")):
            lo = min(regs["%s.%d" % (fe, i)][0] for i in range(8))
            hi = max(regs["%s.%d" % (fe, i)][1] for i in range(8))
            t = regs[fe + ".8"]
            cur = U.get(nm)
            U[nm] = (lo, hi, t[0], t[1]) if cur is None else (min(cur[0], lo), max(cur[1], hi), min(cur[2], t[0]), max(cur[3], t[1]))
    return U


def one_contracts_closed(progs, C):
    U = one_state_outputs(progs, C)
    for k in ("X", "Y", "Z"):
        u, c = U[k], C[k]
        if not (c[0] <= u[0] and u[1] <= c[1] and c[2] <= u[2] and u[3] <= c[3]):
            raise OverflowError("%s leaves the one-lane state contract: %r not inside %r" % (k, u, c))
    return U


FIELD_PROGRAMS = [build_fe_mul, build_fe_sqr]
BN_FIELD_PROGRAMS = [build_bn_mul, build_bn_sqr]

def build_bn_pair_dbl(uns=UNSIGNED_BN_DBL):
    """(A, B) <- 2 * (A, B) on a curve with a = 0 (FP256BN's G1), two lanes per point, formulas of bn_nym29.h::pt_dbl29:
        A2 = X^2, Bq = Y^2, c4 = (2 Bq)^2, D = 4 X Bq, E = 3 A2, F = E^2, X3 = F - 2 D, Y3 = E (D - X3) - 2 c4, Z3 = 2 Y Z.
    Seven field operations in four paired steps (the last one has an idle odd slot).
    in: L(X) = 1, L(Y) <= 3, L(Z) <= 2;  out: L(X) = 1, L(Y) = 1, L(Z) = 2.   E holds A = X, B = Y; O holds B = Z."""
    p = Program("PAIRBN_DBL", bn_field())
    A = p.fe("A", "io")
    B = p.fe("B", "io")
    U1 = p.fe("U1", "tmp")
    U2 = p.fe("U2", "tmp")
    U3 = p.fe("U3", "tmp")
    U4 = p.fe("U4", "tmp")
    P1 = p.fe("P1", "tmp")
    P2 = p.fe("P2", "tmp")
    T0 = p.fe("T0", "tmp")
    T1 = p.fe("T1", "tmp")
    TD = p.fe("TD", "tmp")
    p.swp(T0, A)                     #                           O: X
    p.sel(P1, T0, B)                 # E: Y                      O: X
    p.sqr(U1, P1, TD, unsigned=0 in uns)                # E: Bq = Y^2   [3x3]       O: A2 = X^2   [1x1]
    p.shl(T0, A, 2)                  # E: 4X    (L4)
    p.swp(T1, B)                     #                           O: Y
    p.sel(P1, T1, T0)                # E: 4X                     O: Y
    p.sel(P2, B, U1)                 # E: Bq                     O: Z
    p.mul(U2, P1, P2, unsigned=1 in uns)                # E: D = 4 X Bq [4x1]       O: yz = Y Z   [3x2]
    p.shl(T0, U1, 1)                 # E: 2 Bq  (L2)
    p.shladd(T1, U1, 1, U1)          #                           O: E3 = 3 A2  (L3)
    p.sel(P1, T1, T0)                # E: 2 Bq                   O: E3
    p.swp(T0, U2)                    #                           O: D
    p.lane_const(TD[8], -2, 0)
    p.sqr(U3, P1, TD, T0, TD[8], unsigned=2 in uns)     # E: c4 = 4 Bq^2 [2x2]      O: X3 = E3^2 - 2 D   [3x3]   (round 6: subtracted inside the square)
    p.swp(A, U3)                     # E: X3
    p.sub(T0, U2, A)                 # E: D - X3    (L2)
    p.swp(T1, P1)                    # E: E3 (O's P1)
    p.lane_const(TD[0], 0, -2)
    p.mul(U4, T1, T0, U3, TD[0], unsigned=3 in uns)     # E: Y3 = E3 (D - X3) - 2 c4  [3x2]      O: (idle slot: product of leftovers)
    p.shl(T1, U2, 1)                 #                           O: Z3 = 2 yz  (L2)
    p.sel(B, T1, U4)
    return p


def build_bn_pair_add(uns=UNSIGNED_BN_ADD):
    return build_pair_add("PAIRBN_ADD", bn_field(), uns)


def build_bn_pair_madd(uns=UNSIGNED_BN_MADD):
    return build_pair_madd("PAIRBN_MADD", bn_field(), uns)


BN_PAIR_PROGRAMS = [build_bn_pair_dbl, build_bn_pair_add, build_bn_pair_madd]

PROGRAMS = [build_pair_dbl, build_pair_add, build_pair_madd]

ALIGN_NOTE = """// Every instruction below is 8 bytes (VOP3, VOP2 + DPP, or VOP2 + 32-bit literal): a block started on an 8-byte boundary stays
// on 8-byte boundaries throughout (8-byte instructions at 4-mod-8 addresses cost 8.5 % of kernel time, measured).
#ifndef FE29_GCN_ALIGN
#define FE29_GCN_ALIGN ".p2align 3\\n\\t"
#endif
"""


def emit(path_kind):
    progs = {"field": FIELD_PROGRAMS, "bnfield": BN_FIELD_PROGRAMS, "bnpair": BN_PAIR_PROGRAMS}.get(path_kind, PROGRAMS)
    if path_kind == "one":
        one_contracts_closed({"dbl": build_one_dbl(), "add": build_one_add(), "madd": build_one_madd()}, STATE_ONE)
        print("// GENERATED by gen_pair_gcn.py one - do not edit.  The point operations of p256_verify29.h / ec29.h for one lane per signature on the")
        print("// device: the formulas of pt_dbl29 / pt_add29 / pt_add_mixed29 with the subtractions made inside the products and unsigned digits")
        print("// where gen_pair_gcn.one_contracts_closed proves the headroom; one asm statement per field product (gcn_dsl.Program.emit_cxx).")
        print("#pragma once")
        print('#include "fe29_gcn.h"   // FE29_GCN_ALIGN')
        print()
        print("namespace fab {")
        print()
        for fn, b, params in (("one29_dbl", build_one_dbl, [("X3", "out"), ("Y3", "out"), ("Z3", "out"), ("X", "in"), ("Y", "in"), ("Z", "in")]),
                              ("one29_add", build_one_add, [("X3", "out"), ("Y3", "out"), ("Z3", "out"), ("H", "out"), ("RR", "out"), ("X1", "in"), ("Y1", "in"),
                                                            ("Z1", "in"), ("X2", "in"), ("Y2", "in"), ("Z2", "in")]),
                              ("one29_madd", build_one_madd, [("X3", "out"), ("Y3", "out"), ("Z3", "out"), ("H", "out"), ("RR", "out"), ("X1", "in"), ("Y1", "in"),
                                                              ("Z1", "in"), ("X2", "in"), ("Y2", "in")])):
            text, st = b().emit_cxx(fn, params)
            print(text)
        print("}  // namespace fab")
        return
    if path_kind == "bnpair":
        contracts_closed({"dbl": build_bn_pair_dbl(), "add": build_bn_pair_add(), "madd": build_bn_pair_madd()}, STATE_BN)
    elif path_kind == "pair":
        contracts_closed({"dbl": build_pair_dbl(), "add": build_pair_add(), "madd": build_pair_madd()}, STATE_P256)
    if path_kind == "bnpair":
        print("// GENERATED by gen_pair_gcn.py bnpair - do not edit.  Two-lanes-per-point operations on FP256BN's G1 (a = 0): the point")
        print("// operations of the four-lanes-per-signature idemix kernel (bn_quad29.h); verified in the DSL interpreter against big integers")
        print("// (tests/test_pair_programs.py) and register for register on the MI355X (gputest.hip ops 4-6).")
        print("#pragma once")
        print('#include "fe29_gcn.h"   // FE29_GCN_ALIGN')
        print()
    elif path_kind == "bnfield":
        print("// GENERATED by gen_pair_gcn.py bnfield - do not edit.  gfx950 instruction streams of the FP256BN field product (bn29.h).")
        print("#pragma once")
        print('#include "fe29_gcn.h"   // FE29_GCN_ALIGN')
        print()
    elif path_kind == "field":
        print("// GENERATED by gen_pair_gcn.py field - do not edit.  gfx950 instruction streams of fe_mul / fe_sqr (fe29.h).")
        print("#pragma once")
        print(ALIGN_NOTE)
    else:
        print("// GENERATED by gen_pair_gcn.py - do not edit.  Two-lanes-per-signature point operations (see gen_pair_gcn.py).")
        print("#pragma once")
        print('#include "fe29_gcn.h"   // FE29_GCN_ALIGN')
        print()
    for b in progs:
        prog = b()
        args = {n: "(%s)" % n for n in prog.order}
        text, stats = prog.emit_asm(args)
        print(text)


def main():
    import sys
    emit(sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("field", "bnfield", "bnpair", "one") else "pair")


if __name__ == "__main__":
    main()
