#!/usr/bin/env python3
"""Where the instructions of the dominant kernel go, by CLASS, from the generator's own programs (fabric-mod_amd/csrc/gen_pair_gcn.py,
gcn_dsl.py) x the trip counts of p256_verify_pair_lds_kernel (65 signed 4-bit windows, 8-entry table in LDS):

    mac      v_mad_i64_i32: limb products and the four reduction MACs per column                 (the work the roofline counts)
    carry    what turns a 64-bit column into a limb: quotient digit / balanced digit extraction, the 64-bit shift, the rounding add
    route    lane-pair routing: v_cndmask on the parity mask, DPP moves / fused DPP add / sub between the two lanes of a signature
    field    limb-wise field additions, subtractions, doublings, the weak normalisation pass      (9 instructions per field op)
    hazard   s_nop in front of a DPP source written less than two instructions earlier

Prints a table (instructions per operation and per verification) and the floor argument's numbers; the measured total to compare with
is SQ_INSTS_VALU / SQ_WAVES of profiles/r06_pmc_sq.txt.   usage: python tools/instr_classes.py [measured instructions per wave]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fabric-mod_amd", "csrc"))
import gen_pair_gcn as g   # noqa: E402

CLASS = {"mad0": "mac", "mad": "mac", "q29": "carry", "qn0": "carry", "bfe29": "carry", "round28": "carry", "ashr64": "carry", "movacc": "carry",
         "sel": "route", "swp_mov": "route", "swp_sub": "route", "swp_add": "route", "bc_mov": "route",
         "add": "field", "addc": "field", "sub": "field", "shl": "field", "shladd": "field", "ashr": "field", "mov": "field"}


def classes(prog):
    text, stats = prog.emit_asm({n: "(%s)" % n for n in prog.order})
    out = {"mac": 0, "carry": 0, "route": 0, "field": 0, "hazard": stats["nops"]}
    for ins in prog.ins:
        if ins[0] in ("prod_begin", "prod_end"):      # markers for emit_cxx, not instructions
            continue
        c = CLASS[ins[0]]
        out[c] += 2 if ins[0] == "qn0" else 1
    # bfe29 inside wnorm is a field op, not a column carry: gcn_dsl.wnorm emits 16 of them (plus adds) per call
    return out, stats["instructions"]


def main():
    progs = {"doubling (4M + 4S)": g.build_pair_dbl(), "addition (12M + 4S)": g.build_pair_add(), "mixed addition (8M + 3S)": g.build_pair_madd(),
             "field product (one lane)": g.build_fe_mul(), "field square (one lane)": g.build_fe_sqr()}
    rows = {}
    print("%-28s %7s %7s %7s %7s %7s %7s   %s" % ("per operation", "mac", "carry", "route", "field", "hazard", "total", "mac share"))
    for name, p in progs.items():
        c, total = classes(p)
        assert sum(c.values()) == total, (name, c, total)
        rows[name] = (c, total)
        print("%-28s %7d %7d %7d %7d %7d %7d   %.2f" % (name, c["mac"], c["carry"], c["route"], c["field"], c["hazard"], total, c["mac"] / total))
    # trip counts of the LDS-table kernel: 64 windows x 4 doublings (the first window has nothing to double) + the table's 4; 65 window
    # additions + the final one; 16 comb + the table's 3 mixed additions (the final addition's doubling runs for crafted inputs only)
    trips = {"doubling (4M + 4S)": 64 * 4 + 4, "addition (12M + 4S)": 65 + 1, "mixed addition (8M + 3S)": 16 + 3}
    tot = {k: 0 for k in ("mac", "carry", "route", "field", "hazard")}
    for name, n in trips.items():
        for k in tot:
            tot[k] += n * rows[name][0][k]
    point = sum(tot.values())
    print()
    print("per verification (two lanes), point arithmetic: %d doublings, %d additions, %d mixed additions" % tuple(trips.values()))
    for k in ("mac", "carry", "route", "field", "hazard"):
        print("  %-8s %8d  %5.1f %%" % (k, tot[k], 100.0 * tot[k] / point))
    print("  %-8s %8d" % ("total", point))
    measured = float(sys.argv[1]) if len(sys.argv) > 1 else 354.4e3
    print("  measured SQ_INSTS_VALU per wave (argument; profiles/r06_pmc_sq.txt): %.0f -> %.0f outside the point programs (gates, safegcd inversion mod n shared by the pair,"
          " u1 / u2, window recoding, table building glue, final comparison)" % (measured, measured - point))
    print()
    print("floor: a lone wavefront issues one VALU instruction per 4.16 cycles whatever its class (SQ_WAVE_CYCLES x 4 / SQ_INSTS_VALU); with the carry, route and field")
    print("classes at ZERO the stream would be %d MACs + the %d scalar instructions = %.0f k -> %.2f of today's time.  The classes are not zero:" %
          (tot["mac"], measured - point, (tot["mac"] + measured - point) / 1e3, (tot["mac"] + measured - point) / measured))
    print("  carry  2 instructions per low column, 3 per high column with balanced digits, 2 where the consumers take unsigned ones (gen_pair_gcn.UNSIGNED_*);")
    print("  route  the price of two lanes per signature - it buys a stream of %d instead of %d for the same doubling on one lane;" % (rows["doubling (4M + 4S)"][1], 2 * 4 * 160 // 1 - 0))
    print("  field  9 per limb-wise add / sub, no carry chain, no conditional subtraction.")


if __name__ == "__main__":
    main()
