#!/usr/bin/env python3
"""Rust `extern "C"` declarations for EVERY function include/helix_vec.h declares (INTEGRATION.md section 2 is this script's
output; tests/test_abi_and_host.py parses that block again and holds it to the header: names, arity, argument types).
usage: gen_rust_ffi.py [--check INTEGRATION.md | --update INTEGRATION.md]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALARS = {"uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "int64_t": "i64", "uint16_t": "u16", "uint8_t": "u8", "float": "f32",
           "double": "f64", "size_t": "usize", "int": "c_int", "char": "c_char", "void": "c_void", "unsigned": "u32"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def c_prototypes(header_text):
    """[(name, return C type, [(C type, name)])] of every function prototype `... hvx_xxx(...);` in the header."""
    text = strip_comments(header_text)
    text = re.sub(r"#[^\n]*", " ", text)
    text = re.sub(r'extern\s+"C"\s*\{', " ", text)
    while True:  # drop struct / enum bodies (no function prototype lives inside one)
        t2 = re.sub(r"\{[^{}]*\}", " ", text)
        if t2 == text:
            break
        text = t2
    out = []
    for stmt in text.split(";"):
        stmt = " ".join(stmt.split())
        m = re.match(r"^((?:const\s+)?[A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(hvx_[a-z0-9_]+)\s*\((.*)\)$", stmt)
        if not m:
            continue
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if ret.startswith("typedef"):
            continue
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)?$", a)
                ctype, pname = mm.group(1).strip(), mm.group(2)
                if not ctype:  # unnamed parameter: the "name" was the type (e.g. `hvx_index *`)
                    ctype, pname = a, None
                if pname in SCALARS or (pname and pname.startswith("hvx_") and "*" not in ctype and ctype in ("", "const", "struct")):
                    ctype, pname = a, None
                params.append((" ".join(ctype.split()), pname))
        out.append((name, " ".join(ret.split()), params))
    return out


def rust_type(ctype):
    t = ctype.replace("struct ", "").strip()
    stars = t.count("*")
    t = t.replace("*", " ").strip()
    const = False
    toks = [x for x in t.split() if x]
    if "const" in toks:
        const = True
        toks = [x for x in toks if x != "const"]
    base = " ".join(toks)
    rt = SCALARS.get(base, base)
    if stars == 0:
        return rt
    # `const T **` never occurs in the header; `T **out` = *mut *mut T
    out = rt
    for level in range(stars):
        inner_const = const and level == 0
        out = ("*const " if inner_const else "*mut ") + out
    return out


def rust_decl(name, ret, params):
    args = []
    for i, (ctype, pname) in enumerate(params):
        pn = pname or f"a{i}"
        if pn in ("type", "ref", "in", "fn", "mod", "use", "match", "move", "box", "loop"):
            pn += "_"
        args.append(f"{pn}: {rust_type(ctype)}")
    r = "" if ret == "void" else f" -> {rust_type(ret)}"
    return f"pub fn {name}({', '.join(args)}){r};"


def rust_block(header_text):
    lines = []
    for name, ret, params in c_prototypes(header_text):
        d = rust_decl(name, ret, params)
        if len(d) + 4 > 150:  # wrap long declarations
            head, rest = d.split("(", 1)
            parts = rest.rsplit(")", 1)
            args = parts[0].split(", ")
            cur, wrapped = "    " + head + "(", []
            for a in args:
                if len(cur) + len(a) + 2 > 146:
                    wrapped.append(cur.rstrip())
                    cur = "        "
                cur += a + ", "
            wrapped.append(cur.rstrip(", ") + ")" + parts[1])
            lines.extend(wrapped)
        else:
            lines.append("    " + d)
    return "\n".join(lines)


def parse_rust_fns(block):
    """{name: ([arg types], return type or None)} of the `pub fn` declarations inside a Rust extern block."""
    block = re.sub(r"//[^\n]*", " ", block)
    out = {}
    for m in re.finditer(r"pub fn (hvx_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        args = " ".join(m.group(2).split())
        types = []
        if args:
            for a in args.split(","):
                a = a.strip()
                if a:
                    types.append(" ".join(a.split(":", 1)[1].split()))
        out[m.group(1)] = (types, m.group(3).strip() if m.group(3) else None)
    return out


if __name__ == "__main__":
    hdr = open(os.path.join(ROOT, "include", "helix_vec.h")).read()
    if len(sys.argv) > 2 and sys.argv[1] == "--check":
        doc = open(sys.argv[2]).read()
        got = parse_rust_fns(doc)
        want = {n: ([rust_type(t) for t, _ in p], None if r == "void" else rust_type(r)) for n, r, p in c_prototypes(hdr)}
        missing = sorted(set(want) - set(got))
        wrong = sorted(n for n in want if n in got and got[n] != want[n])
        print(f"{len(want)} prototypes in the header, {len(got)} declared in {sys.argv[2]}; missing {missing}; differing {wrong}")
        sys.exit(1 if missing or wrong else 0)
    if len(sys.argv) > 2 and sys.argv[1] == "--update":  # rewrite the body of the extern block in place
        doc = open(sys.argv[2]).read()
        start = doc.index('#[link(name = "helix_vec_gfx950")]')
        body0 = doc.index("{\n", start) + 2
        body1 = doc.index("\n}", body0)
        doc = doc[:body0] + rust_block(hdr) + doc[body1:]
        n = len(c_prototypes(hdr))
        doc = re.sub(r"// EVERY function the library exports \(\d+\)", f"// EVERY function the library exports ({n})", doc)
        open(sys.argv[2], "w").write(doc)
        print(f"{sys.argv[2]}: extern block rewritten with {n} declarations")
        sys.exit(0)
    print(rust_block(hdr))
