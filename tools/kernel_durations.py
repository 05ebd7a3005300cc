import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for pat in ("walk_count", "walk_emit", "walk_gate", "sha256_mixed", "sha256_messages_coop", "sha256_midstate", "gather_spans"):
    rows = [((e - s) / 1e3) for (s, e) in c.execute("select start, end from kernels where name like ? order by start", ("%" + pat + "%",))]
    print("%-16s" % pat, " ".join("%.0f" % r for r in rows))
