#!/usr/bin/env python3
"""Static hygiene for the Go binding under fabric-mod_amd/go/ - the part of this repository no compiler has ever seen (the build image has no
Go toolchain: SURVEY.md 8(c)).  Not a type checker; it catches what reading misses:

  1. lexing: every file tokenises (strings, raw strings, runes, comments) and its (), [], {} balance; `package` clause present;
  2. imports: every imported package is used and every used package qualifier is imported (or is C / a local identifier);
  3. reference symbols: for every import under github.com/hyperledger/fabric/ the package directory exists in the reference tree and
     every  pkg.Name  the file uses is an exported top-level declaration (func / type / var / const) of that package.  The reference
     tree (/root/reference) only exists in the build container, so the declarations the binding needs are snapshotted by
     `--snapshot` into tests/golden/go_reference_symbols.json (package -> sorted exported names); without the tree the snapshot is used;
  4. cgo: every  C.fabgpu_*  the files call is declared in include/*.h (or defined in that file's cgo preamble) and is called with the
     declared number of arguments; every  C.FABGPU_*  constant is #defined there.

usage: check_go_sources.py [--snapshot] [--reference DIR]      exit status 0 = clean; findings are printed one per line."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C_TYPES = set()
GO_ROOT = os.path.join(ROOT, "fabric-mod_amd", "go")
SNAPSHOT = os.path.join(ROOT, "tests", "golden", "go_reference_symbols.json")
FABRIC = "github.com/hyperledger/fabric/"
# packages of the binding itself (they live under the fabric import path once dropped into the tree: INTEGRATION.md section 2)
OWN = {"bccsp/gpu", "bccsp/idemixgpu"}


def lex(src, path):
    """-> (tokens, errors).  tokens: (kind, text, line) with kind in ident / num / str / op / cgo-preamble comments are dropped."""
    toks, errs, i, n, line = [], [], 0, len(src), 1
    while i < n:
        c = src[i]
        if c == "\n":
            toks.append(("nl", "\n", line))
            line += 1
            i += 1
        elif c in " \t\r":
            i += 1
        elif src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            if j < 0:
                errs.append("%s:%d: unterminated /* comment" % (path, line))
                break
            line += src.count("\n", i, j + 2)
            i = j + 2
        elif c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                if src[j] == "\\":
                    j += 1
                if j < n and src[j] == "\n":
                    errs.append("%s:%d: newline in string literal" % (path, line))
                    break
                j += 1
            if j >= n:
                errs.append("%s:%d: unterminated string" % (path, line))
                break
            toks.append(("str", src[i:j + 1], line))
            i = j + 1
        elif c == "`":
            j = src.find("`", i + 1)
            if j < 0:
                errs.append("%s:%d: unterminated raw string" % (path, line))
                break
            toks.append(("str", src[i:j + 1], line))
            line += src.count("\n", i, j + 1)
            i = j + 1
        elif c == "'":
            j = i + 1
            while j < n and src[j] != "'":
                if src[j] == "\\":
                    j += 1
                j += 1
            if j >= n or j - i > 12:
                errs.append("%s:%d: bad rune literal" % (path, line))
                break
            toks.append(("str", src[i:j + 1], line))
            i = j + 1
        elif c.isalpha() or c == "_":
            j = i
            while j < n and (src[j].isalnum() or src[j] == "_"):
                j += 1
            toks.append(("ident", src[i:j], line))
            i = j
        elif c.isdigit():
            j = i
            while j < n and (src[j].isalnum() or src[j] in "._"):
                j += 1
            toks.append(("num", src[i:j], line))
            i = j
        else:
            toks.append(("op", c, line))
            i += 1
    return toks, errs


def balance(toks, path):
    errs, stack = [], []
    pair = {")": "(", "]": "[", "}": "{"}
    for kind, t, line in toks:
        if kind != "op":
            continue
        if t in "([{":
            stack.append((t, line))
        elif t in ")]}":
            if not stack or stack[-1][0] != pair[t]:
                errs.append("%s:%d: unbalanced %r" % (path, line, t))
                return errs
            stack.pop()
    for t, line in stack:
        errs.append("%s:%d: %r never closed" % (path, line, t))
    return errs


def imports_of(toks):
    """-> {qualifier: import path} ('_' and '.' imports skipped)"""
    out, i = {}, 0
    sig = [t for t in toks if t[0] != "nl"]
    while i < len(sig):
        if sig[i][:2] == ("ident", "import"):
            i += 1
            group = sig[i][:2] == ("op", "(")
            if group:
                i += 1
            while i < len(sig):
                if group and sig[i][:2] == ("op", ")"):
                    break
                alias = None
                if sig[i][0] == "ident" or sig[i][:2] in (("op", "."), ("op", "_")):
                    alias = sig[i][1]
                    i += 1
                if sig[i][0] != "str":
                    break
                path = sig[i][1].strip('"`')
                q = alias or path.rsplit("/", 1)[-1]
                if q.startswith("go-") or q.endswith("-go"):
                    q = q.replace("go-", "").replace("-go", "")
                if alias not in ("_", "."):
                    out[q] = path
                i += 1
                if not group:
                    break
        else:
            i += 1
    return out


def qualified_uses(toks):
    """-> [(qualifier, name, line)] for every  ident . Ident  that is not itself preceded by '.' (a field / method chain)"""
    sig = [t for t in toks if t[0] != "nl"]
    out = []
    for k in range(len(sig) - 2):
        if sig[k][0] == "ident" and sig[k + 1][:2] == ("op", ".") and sig[k + 2][0] == "ident":
            if k > 0 and sig[k - 1][:2] == ("op", "."):
                continue
            out.append((sig[k][1], sig[k + 2][1], sig[k][2], k))
    return out, sig


DECL = re.compile(r"^(?:func\s+(?:\([^)]*\)\s*)?([A-Z]\w*)|type\s+([A-Z]\w*)|var\s+([A-Z]\w*)|const\s+([A-Z]\w*))", re.M)


def exported_names(pkg_dir):
    names = set()
    for fn in sorted(os.listdir(pkg_dir)):
        if not fn.endswith(".go") or fn.endswith("_test.go"):
            continue
        src = open(os.path.join(pkg_dir, fn), encoding="utf-8", errors="replace").read()
        for m in DECL.finditer(src):
            if m.group(0).startswith("func (") or m.group(0).startswith("func\t("):
                continue                                              # a method: reached through a value, never as pkg.Name
            names.update(g for g in m.groups() if g)
        for block in re.finditer(r"^(?:var|const|type)\s*\((.*?)^\)", src, re.M | re.S):      # grouped declarations
            for ln in block.group(1).splitlines():
                m = re.match(r"\s*([A-Z]\w*)\b", ln)
                if m:
                    names.add(m.group(1))
    return names


def c_prototypes():
    """include/*.h -> ({name: n_params}, {macro names})"""
    protos, macros = {}, set()
    global C_TYPES
    C_TYPES = set()
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not fn.endswith(".h"):
            continue
        src = open(os.path.join(ROOT, "include", fn)).read()
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
        src = re.sub(r"//[^\n]*", " ", src)
        macros.update(re.findall(r"^\s*#\s*define\s+(FABGPU_\w+)", src, re.M))
        C_TYPES.update(re.findall(r"typedef\s+struct\s+\w+\s+(fabgpu_\w+)\s*;", src))
        C_TYPES.update(re.findall(r"\}\s*(fabgpu_\w+)\s*;", src))
        for m in re.finditer(r"\b(fabgpu_\w+)\s*\(([^;{]*?)\)\s*;", src, re.S):
            args = m.group(2).strip()
            protos[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return protos, macros


def preamble_functions(src):
    """static helpers defined in the cgo preamble (the comment right above `import "C"`) -> {name: n_params}"""
    m = re.search(r"/\*(.*?)\*/\s*import\s+\"C\"", src, re.S)
    out = {}
    if m:
        for f in re.finditer(r"\b(fabgpu_\w+)\s*\(([^)]*)\)\s*\{", m.group(1)):
            args = f.group(2).strip()
            out[f.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def call_arity(sig, k):
    """sig[k] is the name token; if followed by '(' -> number of arguments of the call, else None"""
    if k + 1 >= len(sig) or sig[k + 1][:2] != ("op", "("):
        return None
    depth, args, seen = 0, 0, False
    for j in range(k + 1, len(sig)):
        kind, t, _ = sig[j]
        if kind == "op" and t in "([{":
            depth += 1
        elif kind == "op" and t in ")]}":
            depth -= 1
            if depth == 0:
                return args + 1 if seen else 0
        elif depth == 1:
            if kind == "op" and t == ",":
                args += 1
            else:
                seen = True
        else:
            seen = True
    return None


def check(reference=None, snapshot_out=None):
    findings, wanted = [], {}
    snap = json.load(open(SNAPSHOT)) if os.path.exists(SNAPSHOT) else {}
    protos, macros = c_prototypes()
    files = []
    for d, _, fns in os.walk(GO_ROOT):
        files += [os.path.join(d, f) for f in fns if f.endswith(".go")]
    assert files, "no Go sources under %s" % GO_ROOT
    ref_cache = {}
    for path in sorted(files):
        rel = os.path.relpath(path, ROOT)
        src = open(path, encoding="utf-8").read()
        toks, errs = lex(src, rel)
        findings += errs
        if errs:
            continue
        findings += balance(toks, rel)
        sig0 = [t for t in toks if t[0] != "nl"]
        if not any(a[:2] == ("ident", "package") for a in sig0[:40]):
            findings.append("%s: no package clause" % rel)
        imps = imports_of(toks)
        uses, sig = qualified_uses(toks)
        used_q = {q for q, _, _, _ in uses}
        for q, p in imps.items():
            if q not in used_q and q != "C":
                findings.append("%s: imported and not used: %s (%s)" % (rel, q, p))
        pre = preamble_functions(src)
        for q, name, line, k in uses:
            if q == "C":
                if name.startswith("fabgpu_"):
                    want = pre.get(name, protos.get(name))
                    if name in C_TYPES:
                        continue
                    if want is None:
                        findings.append("%s:%d: C.%s is declared neither in include/*.h nor in the cgo preamble" % (rel, line, name))
                    else:
                        got = call_arity(sig, k + 2)
                        if got is not None and got != want:
                            findings.append("%s:%d: C.%s called with %d arguments, declared with %d" % (rel, line, name, got, want))
                elif name.startswith("FABGPU_") and name not in macros:
                    findings.append("%s:%d: C.%s is not #defined in include/*.h" % (rel, line, name))
                continue
            p = imps.get(q)
            if p is None or not p.startswith(FABRIC):
                continue
            pkg = p[len(FABRIC):]
            if pkg in OWN or not name[:1].isupper():
                continue
            wanted.setdefault(pkg, set()).add(name)
            if reference:
                if pkg not in ref_cache:
                    d = os.path.join(reference, pkg)
                    ref_cache[pkg] = exported_names(d) if os.path.isdir(d) else None
                names = ref_cache[pkg]
                if names is None:
                    findings.append("%s:%d: package %s does not exist in the reference tree" % (rel, line, p))
                elif name not in names:
                    findings.append("%s:%d: %s.%s is not an exported declaration of %s" % (rel, line, q, name, p))
            else:
                names = snap.get(pkg)
                if names is None:
                    findings.append("%s:%d: package %s is not in the snapshot (run tools/check_go_sources.py --snapshot where /root/reference exists)" % (rel, line, p))
                elif name not in names:
                    findings.append("%s:%d: %s.%s is not in the snapshot of %s" % (rel, line, q, name, p))
    if snapshot_out and reference:
        out = {}
        for pkg in sorted(wanted):
            d = os.path.join(reference, pkg)
            have = exported_names(d) if os.path.isdir(d) else set()
            out[pkg] = sorted(wanted[pkg] & have)                        # only what the binding uses AND the reference declares
        json.dump({"_what": "exported declarations of the reference packages the Go binding uses, checked against /root/reference by "
                            "tools/check_go_sources.py --snapshot; package -> names", **out}, open(snapshot_out, "w"), indent=1, sort_keys=True)
    return sorted(set(findings)), len(files)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--snapshot", action="store_true", help="rewrite tests/golden/go_reference_symbols.json from the reference tree")
    a = ap.parse_args()
    ref = a.reference if os.path.isdir(a.reference) else None
    if a.snapshot and not ref:
        sys.exit("--snapshot needs the reference tree")
    findings, n = check(ref, SNAPSHOT if a.snapshot else None)
    for f in findings:
        print(f)
    print("%d Go files checked against %s: %d finding(s)" % (n, ref or "the committed snapshot", len(findings)), file=sys.stderr)
    sys.exit(1 if findings else 0)


if __name__ == "__main__":
    main()
