#!/usr/bin/env python3
"""A tiny straight-line "assembler DSL" for the gfx950 instruction streams of the fe29 field / point arithmetic.

A Program is a list of abstract instructions over symbolic registers:
    fe registers   nine 32-bit VGPR limbs each, named "<fe>.<i>"
    "acc"          the 64-bit column accumulator, hard-wired to v[0:1] (inline asm cannot name half of a 64-bit operand)
    constants      wave-uniform SGPR operands (the Montgomery multipliers 2^9, 2^18, -2^21, 2^24, the rounding 2^28,
                   and the lane-parity mask of the two-lanes-per-signature programs)
Four consumers of the same list:
    emit_asm()     one GNU inline-asm statement (operands numbered outputs-first), every instruction encoded in 8 bytes
                   (VOP3, VOP2+DPP, or VOP2+literal) so that a `.p2align 3` block never leaves 8-byte alignment, with the
                   gfx940-family hazard "VALU writes a VGPR, a DPP instruction reads it as its DPP source: 2 wait states"
                   padded by s_nop where the program order does not already provide the distance (only v_mov / v_add / v_sub
                   are used in DPP form: v_subrev_u32_dpp did not compute src1 - dpp(src0) on MI355X, probed in gputest.hip);
    emit_cxx()     (round 6) for programs without lane routing: a device function whose field products are one asm statement
                   each and whose limb-wise operations are C++ - the compiler keeps the register allocation (one29_gcn.h);
    run()          a reference interpreter with exact 32/64-bit wrap-around semantics on an explicit (even, odd) lane
                   pair - tests/test_pair_programs.py executes the very programs the kernels run against big-integer
                   point arithmetic, without a GPU;
    run_intervals()  (round 6) the same interpreter on RANGES: the proof that no 64-bit column and no 32-bit limb can wrap for
                   any input inside a contract (gen_pair_gcn.contracts_closed / one_contracts_closed; the generator refuses
                   to emit a header whose contract is not closed).
A field product may take addends (r = a b + c d + ...: the addend's digits join the high columns) and may leave unsigned
digits (two instructions per high column instead of three) - both decided per product by the generator (gen_pair_gcn.py).
"""

M32 = (1 << 32) - 1
M64 = (1 << 64) - 1


def s32(x):
    x &= M32
    return x - (1 << 32) if x >> 31 else x


def s64(x):
    x &= M64
    return x - (1 << 64) if x >> 63 else x


CONSTS = {"c9": 512, "c18": 262144, "cm21": -2097152, "c24": 16777216, "c28": 268435456}
RED = [(3, "c9"), (6, "c18"), (7, "cm21"), (8, "c24")]


class GenericField:
    """A prime without structure: Montgomery reduction by quotient digits q = column * n0 mod 2^29 and nine MACs q * p_j with p
    as balanced digits (bn29.h).  The constants travel as SGPR operands pb0..pb8 and n0."""

    def __init__(self, p_balanced, n0):
        assert len(p_balanced) == 9
        self.pb = list(p_balanced)
        self.n0 = n0

    def consts(self):
        d = {"pb%d" % j: self.pb[j] for j in range(9)}
        d["n0"] = self.n0
        return d


class Program:
    def __init__(self, name, field=None):
        self.name = name
        self.field = field     # None: the P-256 prime (reduction by the four constants of RED); GenericField otherwise
        self.consts = dict(CONSTS)
        if field is not None:
            self.consts.update(field.consts())
        self.ins = []          # (op, dst, *srcs)
        self.fes = {}          # fe name -> kind: "io" | "tmp" | "in"
        self.order = []        # declaration order

    # ---- declarations -------------------------------------------------------------------------------------------
    def fe(self, name, kind):
        assert kind in ("io", "tmp", "in") and name not in self.fes
        self.fes[name] = kind
        self.order.append(name)
        return [f"{name}.{i}" for i in range(9)]

    # ---- elementwise helpers (nine limbs) ---------------------------------------------------------------------
    def add(self, d, a, b):
        for i in range(9):
            self.ins.append(("add", d[i], a[i], b[i]))

    def sub(self, d, a, b):
        for i in range(9):
            self.ins.append(("sub", d[i], a[i], b[i]))

    def neg(self, d, a):
        for i in range(9):
            self.ins.append(("sub", d[i], 0, a[i]))

    def shl(self, d, a, n):
        for i in range(9):
            self.ins.append(("shl", d[i], a[i], n))

    def shladd(self, d, a, n, b):             # d = (a << n) + b
        for i in range(9):
            self.ins.append(("shladd", d[i], a[i], n, b[i]))

    def sel(self, d, odd, even):              # d = lane is odd ? odd : even
        for i in range(9):
            self.ins.append(("sel", d[i], odd[i], even[i]))

    def mov(self, d, a):
        for i in range(9):
            self.ins.append(("mov", d[i], a[i]))

    def swp(self, d, a):                      # d = partner's a
        for i in range(9):
            self.ins.append(("swp_mov", d[i], a[i]))

    def bce(self, d, a):                      # d = the EVEN lane's a, on both lanes of the pair (quad_perm:[0,0,2,2])
        for i in range(9):
            self.ins.append(("bc_mov", d[i], a[i], 0))

    def bco(self, d, a):                      # d = the ODD lane's a, on both lanes (quad_perm:[1,1,3,3])
        for i in range(9):
            self.ins.append(("bc_mov", d[i], a[i], 1))

    def lane_const(self, d, odd, even):       # one register: lane is odd ? odd : even  (inline constants)
        self.ins.append(("sel", d, odd, even))

    def swp_sub(self, d, a, b):               # d = partner's a - own b
        for i in range(9):
            self.ins.append(("swp_sub", d[i], a[i], b[i]))

    def swp_add(self, d, a, b):               # d = partner's a + own b
        for i in range(9):
            self.ins.append(("swp_add", d[i], a[i], b[i]))

    def wnorm(self, d, a, c):
        """One parallel carry pass (fe_weak_norm): d = a with digits back in [-2^28, 2^28] (+-carry).  c: scratch fe."""
        for i in range(8):
            self.ins.append(("addc", c[i], a[i], "c28"))          # a + 2^28
            self.ins.append(("ashr", c[i], c[i], 29))             # carry out of limb i
        self.ins.append(("bfe29", d[0], a[0]))
        for i in range(1, 8):
            self.ins.append(("bfe29", c[8], a[i]))                # c[8] is free scratch
            self.ins.append(("add", d[i], c[8], c[i - 1]))
        self.ins.append(("add", d[8], a[8], c[7]))

    # ---- Montgomery product / square (fe29.h FE29_REDUCE_COLUMN) ---------------------------------------------------
    def _columns(self, r, products, addend=None, coef=None, unsigned=False):
        """unsigned: the output digits are the low 29 bits of their columns as they are, in [0, 2^29) - two instructions per high column
        instead of three (no rounding add); for products whose consumers have the headroom (Program.run_intervals proves it).
        addend / coef: r = product + coef * addend (Montgomery domain: the addend's digit j joins column 9 + j, its top digit the
        last carry) - coef is a register, possibly lane-specific; the output digits are balanced whatever was added."""
        addends = [] if addend is None else (list(addend) if coef is None else [(addend, coef)])   # [(fe, coefficient), ...]
        self.ins.append(("prod_begin",))       # (markers for emit_cxx: one asm statement per product; the other back ends skip them)
        first = True
        for k in range(17):
            for (x, y) in products(k):
                self.ins.append(("mad0" if first else "mad", "acc", x, y))
                first = False
            if k >= 9:
                for (ad, cf) in addends:
                    self.ins.append(("mad", "acc", ad[k - 9], cf))
            if self.field is None:
                for (dd, c) in RED:
                    if k >= dd and k - dd <= 8:
                        self.ins.append(("mad", "acc", r[k - dd], c))
            else:
                for i in range(9):
                    j = k - i
                    if i < k and 0 <= j < 9:
                        self.ins.append(("mad", "acc", r[i], "pb%d" % j))
            if k <= 8:
                if self.field is None:
                    self.ins.append(("q29", r[k], "acc"))         # r[k] = acc & (2^29-1)   (quotient digit: p = -1 mod 2^29)
                else:
                    self.ins.append(("qn0", r[k], "acc"))         # r[k] = (acc * n0) & (2^29-1)
                    self.ins.append(("mad", "acc", r[k], "pb0"))
                self.ins.append(("ashr64", "acc", 29))
            else:
                if unsigned:
                    self.ins.append(("q29", r[k - 9], "acc"))     # output digit in [0, 2^29)
                else:
                    self.ins.append(("bfe29", r[k - 9], "acc"))   # balanced output digit
                    self.ins.append(("round28", "acc"))           # acc += 2^28
                self.ins.append(("ashr64", "acc", 29))
                if k == 16:
                    for (ad, cf) in addends:
                        self.ins.append(("mad", "acc", ad[8], cf))
                    self.ins.append(("movacc", r[8], "acc"))
        self.ins.append(("prod_end",))

    def mul(self, r, a, b, addend=None, coef=None, unsigned=False):
        assert r[0] != a[0] and r[0] != b[0], "mul destination must not alias a source"
        self._columns(r, lambda k: [(a[i], b[k - i]) for i in range(9) if 0 <= k - i < 9], addend, coef, unsigned)

    def sqr(self, r, a, t, addend=None, coef=None, unsigned=False):
        """t: scratch fe for the doubled limbs (8 used)."""
        assert r[0] != a[0] and t[0] != a[0] and t[0] != r[0]
        for i in range(8):
            self.ins.append(("shl", t[i], a[i], 1))

        def prods(k):
            out = []
            for i in range(9):
                j = k - i
                if j > i and j < 9:
                    out.append((t[i], a[j]))
                if j == i:
                    out.append((a[i], a[i]))
            return out
        self._columns(r, prods, addend, coef, unsigned)

    # ---- interpreter -------------------------------------------------------------------------------------------------
    def run(self, regs_even, regs_odd):
        """regs_*: dict "fe.i" -> int (io / in registers must be present).  Executes on the lane pair, returns the dicts."""
        L = [dict(regs_even), dict(regs_odd)]
        acc = [0, 0]

        def val(lane, x):
            if isinstance(x, int):
                return x
            if x in self.consts:
                return self.consts[x]
            return L[lane][x]
        for ins in self.ins:
            op = ins[0]
            if op in ("prod_begin", "prod_end"):
                continue
            if op == "bc_mov":
                v = val(ins[3], ins[2])
                for lane in (0, 1):
                    L[lane][ins[1]] = v
                continue
            if op in ("swp_mov", "swp_sub", "swp_add"):
                new = []
                for lane in (0, 1):
                    p = val(1 - lane, ins[2])
                    if op == "swp_mov":
                        v = p
                    elif op == "swp_sub":
                        v = p - val(lane, ins[3])
                    else:
                        v = p + val(lane, ins[3])
                    new.append(s32(v))
                for lane in (0, 1):
                    L[lane][ins[1]] = new[lane]
                continue
            for lane in (0, 1):
                R = L[lane]
                if op == "add":
                    R[ins[1]] = s32(val(lane, ins[2]) + val(lane, ins[3]))
                elif op == "addc":
                    R[ins[1]] = s32(val(lane, ins[2]) + val(lane, ins[3]))
                elif op == "sub":
                    R[ins[1]] = s32(val(lane, ins[2]) - val(lane, ins[3]))
                elif op == "shl":
                    R[ins[1]] = s32(val(lane, ins[2]) << ins[3])
                elif op == "shladd":
                    R[ins[1]] = s32((val(lane, ins[2]) << ins[3]) + val(lane, ins[4]))
                elif op == "ashr":
                    R[ins[1]] = s32(val(lane, ins[2])) >> ins[3]
                elif op == "sel":
                    R[ins[1]] = val(lane, ins[2]) if lane == 1 else val(lane, ins[3])
                elif op == "mov":
                    R[ins[1]] = val(lane, ins[2])
                elif op == "bfe29":
                    x = acc[lane] if ins[2] == "acc" else val(lane, ins[2])
                    x &= (1 << 29) - 1
                    R[ins[1]] = x - (1 << 29) if x >> 28 else x
                elif op == "mad0":
                    acc[lane] = s64(s32(val(lane, ins[2])) * s32(val(lane, ins[3])))
                elif op == "mad":
                    acc[lane] = s64(acc[lane] + s32(val(lane, ins[2])) * s32(val(lane, ins[3])))
                elif op == "q29":
                    R[ins[1]] = acc[lane] & ((1 << 29) - 1)
                elif op == "qn0":
                    R[ins[1]] = ((acc[lane] & M32) * self.consts["n0"]) & ((1 << 29) - 1)
                elif op == "ashr64":
                    acc[lane] = s64(acc[lane]) >> ins[2]
                elif op == "round28":
                    acc[lane] = s64(acc[lane] + (1 << 28))
                elif op == "movacc":
                    R[ins[1]] = s32(acc[lane])
                else:
                    raise ValueError(op)
        return L[0], L[1]

    # ---- interval interpreter: the proof that no accumulator and no limb ever wraps ---------------------------------------
    def run_intervals(self, regs_even, regs_odd):
        """regs_*: dict "fe.i" -> (lo, hi), the range of the register on that lane; a register that is absent is GARBAGE (a lane's
        don't-care value: whatever is computed from it is garbage too and is not checked - it must not reach anything that matters,
        which the caller sees as a missing output).  Executes the program on ranges: every 64-bit accumulator value must stay inside
        int64 and every 32-bit result inside int32 for ALL inputs in the given ranges (interval arithmetic: sound, not tight).
        Returns the dicts of output ranges."""
        L = [dict(regs_even), dict(regs_odd)]
        acc = [None, None]
        I32 = (-(1 << 31), (1 << 31) - 1)
        I64 = (-(1 << 63), (1 << 63) - 1)

        def val(lane, x):
            if isinstance(x, int):
                return (x, x)
            if x in self.consts:
                return (self.consts[x], self.consts[x])
            return L[lane].get(x)

        def fit(iv, box, what):
            if iv is not None and not (box[0] <= iv[0] and iv[1] <= box[1]):
                raise OverflowError("%s: %s leaves [%d, %d]: [%d, %d]" % (self.name, what, box[0], box[1], iv[0], iv[1]))
            return iv

        def mul(a, b):
            if a is None or b is None:
                return None
            c = (a[0] * b[0], a[0] * b[1], a[1] * b[0], a[1] * b[1])
            return (min(c), max(c))

        def add(a, b):
            return None if a is None or b is None else (a[0] + b[0], a[1] + b[1])

        def sub(a, b):
            return None if a is None or b is None else (a[0] - b[1], a[1] - b[0])

        def low29(a, signed):
            if a is None:
                return None
            box = (-(1 << 28), (1 << 28) - 1) if signed else (0, (1 << 29) - 1)
            return a if box[0] <= a[0] and a[1] <= box[1] else box
        for n, ins in enumerate(self.ins):
            op = ins[0]
            if op in ("prod_begin", "prod_end"):
                continue
            what = "%s #%d -> %s" % (op, n, ins[1])
            if op == "bc_mov":
                v = val(ins[3], ins[2])
                for lane in (0, 1):
                    L[lane][ins[1]] = v
                continue
            if op in ("swp_mov", "swp_sub", "swp_add"):
                new = []
                for lane in (0, 1):
                    pv = val(1 - lane, ins[2])
                    v = pv if op == "swp_mov" else (sub(pv, val(lane, ins[3])) if op == "swp_sub" else add(pv, val(lane, ins[3])))
                    new.append(fit(v, I32, what))
                for lane in (0, 1):
                    L[lane][ins[1]] = new[lane]
                continue
            for lane in (0, 1):
                R = L[lane]
                if op in ("add", "addc"):
                    R[ins[1]] = fit(add(val(lane, ins[2]), val(lane, ins[3])), I32, what)
                elif op == "sub":
                    R[ins[1]] = fit(sub(val(lane, ins[2]), val(lane, ins[3])), I32, what)
                elif op == "shl":
                    R[ins[1]] = fit(mul(val(lane, ins[2]), (1 << ins[3], 1 << ins[3])), I32, what)
                elif op == "shladd":
                    R[ins[1]] = fit(add(mul(val(lane, ins[2]), (1 << ins[3], 1 << ins[3])), val(lane, ins[4])), I32, what)
                elif op == "ashr":
                    a = val(lane, ins[2])
                    R[ins[1]] = None if a is None else (a[0] >> ins[3], a[1] >> ins[3])
                elif op == "sel":
                    R[ins[1]] = val(lane, ins[2]) if lane == 1 else val(lane, ins[3])
                elif op == "mov":
                    R[ins[1]] = val(lane, ins[2])
                elif op == "bfe29":
                    R[ins[1]] = low29(acc[lane] if ins[2] == "acc" else val(lane, ins[2]), True)
                elif op == "mad0":
                    acc[lane] = fit(mul(val(lane, ins[2]), val(lane, ins[3])), I64, what)
                elif op == "mad":
                    acc[lane] = fit(add(acc[lane], mul(val(lane, ins[2]), val(lane, ins[3]))), I64, what)
                elif op == "q29":
                    R[ins[1]] = low29(acc[lane], False)
                elif op == "qn0":
                    R[ins[1]] = None if acc[lane] is None else (0, (1 << 29) - 1)
                elif op == "ashr64":
                    acc[lane] = None if acc[lane] is None else (acc[lane][0] >> ins[2], acc[lane][1] >> ins[2])
                elif op == "round28":
                    acc[lane] = fit(add(acc[lane], (1 << 28, 1 << 28)), I64, what)
                elif op == "movacc":
                    R[ins[1]] = fit(acc[lane], I32, what)
                else:
                    raise ValueError(op)
        return L[0], L[1]

    # ---- C++ back end: one asm statement per PRODUCT, limb-wise operations as C++ --------------------------------------------
    def emit_cxx(self, fn_name, params, ftype="fe"):
        """For one-lane programs (no lane routing): a __device__ function whose field products are one asm statement each - the same
        columns, constants and digits as emit_asm would produce - and whose limb-wise operations are plain C++, so that the compiler
        allocates registers and schedules around the products as it does for hand-written code (a whole point operation as ONE asm
        statement pins every temporary to a register of its own: fine at one wavefront per SIMD, not at two).
        params: [(fe name, "in" | "out")] in signature order; every other fe is a local.  Returns (text, stats)."""
        kinds = dict(params)

        def cx(reg):
            name, i = reg.rsplit(".", 1)
            return "%s.v[%s]" % (name, i)

        def ev(x):
            return str(x) if isinstance(x, int) else cx(x)
        body = []
        n_asm_ins = 0
        n_cxx = 0
        i = 0
        ins_list = self.ins
        cnames = ("c9", "c18", "cm21", "c24") if self.field is None else tuple("pb%d" % j for j in range(9)) + ("n0",)
        while i < len(ins_list):
            ins = ins_list[i]
            op = ins[0]
            if op == "prod_begin":
                j = i + 1
                while ins_list[j][0] != "prod_end":
                    j += 1
                group = ins_list[i + 1:j]
                written, read = [], []
                for g in group:
                    if g[0] in ("q29", "qn0", "bfe29", "movacc") and g[1] != "acc" and g[1] not in written:
                        written.append(g[1])
                for g in group:
                    for x in g[2:]:
                        if isinstance(x, str) and "." in x and x not in written and x not in read:
                            read.append(x)
                num = {}
                outs = []
                for r in written:
                    num[r] = len(outs)
                    outs.append('"=&v"(%s)' % cx(r))
                ins_ = []
                for r in read:
                    num[r] = len(outs) + len(ins_)
                    ins_.append('"v"(%s)' % cx(r))
                for c in cnames:
                    num[c] = len(outs) + len(ins_)
                    ins_.append('"s"(%d)' % self.consts[c])
                num["c28q"] = len(outs) + len(ins_)
                ins_.append('"s"((int64_t)268435456)')

                def o(x):
                    return str(x) if isinstance(x, int) else "%%%d" % num[x]
                lines = []
                for g in group:
                    gop = g[0]
                    if gop == "mad0":
                        lines.append("v_mad_i64_i32 v[0:1], vcc, %s, %s, 0" % (o(g[2]), o(g[3])))
                    elif gop == "mad":
                        lines.append("v_mad_i64_i32 v[0:1], vcc, %s, %s, v[0:1]" % (o(g[2]), o(g[3])))
                    elif gop == "q29":
                        lines.append("v_and_b32 %s, 0x1fffffff, v0" % o(g[1]))
                    elif gop == "qn0":
                        lines.append("v_mul_lo_u32 %s, v0, %s" % (o(g[1]), o("n0")))
                        lines.append("v_and_b32 %s, 0x1fffffff, %s" % (o(g[1]), o(g[1])))
                    elif gop == "bfe29":
                        assert g[2] == "acc"
                        lines.append("v_bfe_i32 %s, v0, 0, 29" % o(g[1]))
                    elif gop == "ashr64":
                        lines.append("v_ashrrev_i64 v[0:1], %d, v[0:1]" % g[2])
                    elif gop == "round28":
                        lines.append("v_lshl_add_u64 v[0:1], v[0:1], 0, %s" % o("c28q"))
                    elif gop == "movacc":
                        lines.append("v_mov_b32_e64 %s, v0" % o(g[1]))
                    else:
                        raise ValueError(gop)
                n_asm_ins += len(lines)
                body.append("    asm(FE29_GCN_ALIGN")
                for l in lines:
                    body.append('        "%s\\n\\t"' % l)
                body.append("        : %s" % ", ".join(outs))
                body.append("        : %s" % ", ".join(ins_))
                body.append('        : "v0", "v1", "vcc");')
                i = j + 1
                continue
            n_cxx += 1
            if op in ("add", "addc"):
                body.append("    %s = %s + %s;" % (cx(ins[1]), ev(ins[2]) if ins[2] not in self.consts else str(self.consts[ins[2]]), ev(ins[3]) if ins[3] not in self.consts else str(self.consts[ins[3]])))
            elif op == "sub":
                body.append("    %s = %s - %s;" % (cx(ins[1]), ev(ins[2]), ev(ins[3])))
            elif op == "shl":
                body.append("    %s = (int32_t)((uint32_t)%s << %d);" % (cx(ins[1]), ev(ins[2]), ins[3]))
            elif op == "shladd":
                body.append("    %s = (int32_t)(((uint32_t)%s << %d) + (uint32_t)%s);" % (cx(ins[1]), ev(ins[2]), ins[3], ev(ins[4])))
            elif op == "ashr":
                body.append("    %s = %s >> %d;" % (cx(ins[1]), ev(ins[2]), ins[3]))
            elif op == "mov":
                body.append("    %s = %s;" % (cx(ins[1]), ev(ins[2])))
            elif op == "bfe29":
                body.append("    %s = (int32_t)((uint32_t)%s << 3) >> 3;" % (cx(ins[1]), ev(ins[2])))
            else:
                raise ValueError("emit_cxx: %s has no one-lane C++ form" % op)
            i += 1
        sig = ", ".join(("const %s& %s" if k == "in" else "%s& %s") % (ftype, n) for n, k in params)
        local = [n for n in self.order if n not in kinds]
        text = ["// %s: %d instructions in %d product statements + %d limb-wise C++ operations" % (
            fn_name, n_asm_ins, sum(1 for x in self.ins if x[0] == "prod_begin"), n_cxx)]
        text.append("FAB_D void %s(%s) {" % (fn_name, sig))
        if local:
            text.append("    %s %s;" % (ftype, ", ".join(local)))
        text += body
        text.append("}")
        return "\n".join(text) + "\n", {"instructions": n_asm_ins + n_cxx}

    # ---- asm back end -------------------------------------------------------------------------------------------------
    def emit_asm(self, macro_args):
        """macro_args: dict fe name -> C expression of the `fe` lvalue.  Returns (text of the #define, stats)."""
        outs, ins_ = [], []
        num = {}
        for name in self.order:
            if self.fes[name] in ("io", "tmp"):
                for i in range(9):
                    num[f"{name}.{i}"] = len(outs)
                    outs.append('"%sv"((%s).v[%d])' % ("+" if self.fes[name] == "io" else "=&", macro_args[name], i))
        for name in self.order:
            if self.fes[name] == "in":
                for i in range(9):
                    num[f"{name}.{i}"] = len(outs) + len(ins_)
                    ins_.append('"v"((%s).v[%d])' % (macro_args[name], i))
        cnames = ("c9", "c18", "cm21", "c24", "c28") if self.field is None else tuple("pb%d" % j for j in range(9)) + ("n0", "c28")
        for c in cnames:
            num[c] = len(outs) + len(ins_)
            ins_.append('"s"(%d)' % self.consts[c])
        num["c28q"] = len(outs) + len(ins_)
        ins_.append('"s"((int64_t)268435456)')
        num["mask"] = len(outs) + len(ins_)
        ins_.append('"s"((uint64_t)0xAAAAAAAAAAAAAAAAull)')

        def o(x):
            if isinstance(x, int):
                return str(x)
            return "%%%d" % num[x]
        DPP = " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        DPP_BC = (" quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf", " quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf")
        lines = []
        last_write = {}
        nops = 0
        count = 0

        def put(text, writes=None, dpp_src=None):
            nonlocal nops, count
            if dpp_src is not None and dpp_src in last_write:
                between = count - last_write[dpp_src] - 1
                if between < 2:
                    lines.append("s_nop %d" % (2 - between - 1))
                    nops += 1
                    count += 1
            lines.append(text)
            if writes is not None:
                last_write[writes] = count
            count += 1
        for ins in self.ins:
            op = ins[0]
            if op in ("prod_begin", "prod_end"):
                continue
            if op == "add" or op == "addc":
                put("v_add_u32_e64 %s, %s, %s" % (o(ins[1]), o(ins[2]), o(ins[3])), ins[1])
            elif op == "sub":
                put("v_sub_u32_e64 %s, %s, %s" % (o(ins[1]), o(ins[2]), o(ins[3])), ins[1])
            elif op == "shl":
                put("v_lshlrev_b32_e64 %s, %d, %s" % (o(ins[1]), ins[3], o(ins[2])), ins[1])
            elif op == "shladd":
                put("v_lshl_add_u32 %s, %s, %d, %s" % (o(ins[1]), o(ins[2]), ins[3], o(ins[4])), ins[1])
            elif op == "ashr":
                put("v_ashrrev_i32_e64 %s, %d, %s" % (o(ins[1]), ins[3], o(ins[2])), ins[1])
            elif op == "sel":
                put("v_cndmask_b32_e64 %s, %s, %s, %s" % (o(ins[1]), o(ins[3]), o(ins[2]), o("mask")), ins[1])
            elif op == "mov":
                put("v_mov_b32_e64 %s, %s" % (o(ins[1]), o(ins[2])), ins[1])
            elif op == "swp_mov":
                put("v_mov_b32_dpp %s, %s%s" % (o(ins[1]), o(ins[2]), DPP), ins[1], dpp_src=ins[2])
            elif op == "bc_mov":
                put("v_mov_b32_dpp %s, %s%s" % (o(ins[1]), o(ins[2]), DPP_BC[ins[3]]), ins[1], dpp_src=ins[2])
            elif op == "swp_sub":
                put("v_sub_u32_dpp %s, %s, %s%s" % (o(ins[1]), o(ins[2]), o(ins[3]), DPP), ins[1], dpp_src=ins[2])
            elif op == "swp_add":
                put("v_add_u32_dpp %s, %s, %s%s" % (o(ins[1]), o(ins[2]), o(ins[3]), DPP), ins[1], dpp_src=ins[2])
            elif op == "bfe29":
                put("v_bfe_i32 %s, %s, 0, 29" % (o(ins[1]), "v0" if ins[2] == "acc" else o(ins[2])), ins[1])
            elif op == "mad0":
                put("v_mad_i64_i32 v[0:1], vcc, %s, %s, 0" % (o(ins[2]), o(ins[3])))
            elif op == "mad":
                put("v_mad_i64_i32 v[0:1], vcc, %s, %s, v[0:1]" % (o(ins[2]), o(ins[3])))
            elif op == "q29":
                put("v_and_b32 %s, 0x1fffffff, v0" % o(ins[1]), ins[1])
            elif op == "qn0":
                put("v_mul_lo_u32 %s, v0, %s" % (o(ins[1]), o("n0")), ins[1])
                put("v_and_b32 %s, 0x1fffffff, %s" % (o(ins[1]), o(ins[1])), ins[1])
            elif op == "ashr64":
                put("v_ashrrev_i64 v[0:1], %d, v[0:1]" % ins[2])
            elif op == "round28":
                put("v_lshl_add_u64 v[0:1], v[0:1], 0, %s" % o("c28q"))
            elif op == "movacc":
                put("v_mov_b32_e64 %s, v0" % o(ins[1]), ins[1])
            else:
                raise ValueError(op)
        args = ", ".join(macro_args[n].strip("()") for n in self.order)
        text = ["// %s: %d instructions (%d v_mad_i64_i32, %d hazard s_nop)" % (
            self.name, len(lines), sum(l.startswith("v_mad") for l in lines), nops)]
        text.append("#define %s(%s) \\" % (self.name, args))
        text.append("    asm( \\")
        text.append("        FE29_GCN_ALIGN \\")
        for l in lines:
            text.append('        "%s\\n\\t" \\' % l)
        text.append("        : %s \\" % ", ".join(outs))
        text.append("        : %s \\" % ", ".join(ins_))
        text.append('        : "v0", "v1", "vcc")')
        return "\n".join(text) + "\n", {"instructions": len(lines), "nops": nops}
