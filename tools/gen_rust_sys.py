#!/usr/bin/env python
"""Rust `extern "C"` declarations from include/granne_hip.h (for INTEGRATION.md's src/gpu.rs).
usage: python tools/gen_rust_sys.py [name ...]   (no names: every entry point)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TYPES = {"void": "c_void", "char": "c_char", "int": "c_int", "float": "f32", "uint64_t": "u64", "uint32_t": "u32",
         "int8_t": "i8", "int64_t": "i64"}


def rust_type(c):
    c = c.strip()
    m = re.match(r"^(const\s+)?([A-Za-z_0-9]+)\s*((?:\*\s*(?:const\s*)?)*)$", c)
    assert m, c
    const, base, stars = bool(m.group(1)), m.group(2), m.group(3)
    t = TYPES.get(base, base)
    ptrs = re.findall(r"\*\s*(const)?", stars)
    for i, pc in enumerate(ptrs):  # innermost pointer first
        inner_const = const if i == 0 else bool(ptrs[i - 1])
        t = ("*const " if inner_const else "*mut ") + t
    return t


def protos():
    h = open(os.path.join(ROOT, "include", "granne_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    h = re.sub(r"//[^\n]*", "", h)
    for ret, name, args in re.findall(r"\n\s*([A-Za-z_][A-Za-z0-9_ \*]*?)\b(granne_hip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", h):
        params = []
        if args.strip() != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
                params.append((m.group(2), rust_type(m.group(1))))
        yield name, ret.strip(), params


def decl(name, ret, params):
    r = "" if ret == "void" else " -> " + rust_type(ret)
    s = "    fn %s(%s)%s;" % (name, ", ".join("%s: %s" % p for p in params), r)
    if len(s) > 118:  # wrap like the hand-written block
        lines, cur = [], "    fn %s(" % name
        for i, p in enumerate(params):
            piece = "%s: %s" % p + (", " if i + 1 < len(params) else "")
            if len(cur) + len(piece) > 116:
                lines.append(cur.rstrip())
                cur = "        "
            cur += piece
        lines.append(cur + ")" + r + ";")
        s = "\n".join(lines)
    return s


if __name__ == "__main__":
    want = set(sys.argv[1:])
    for name, ret, params in protos():
        if not want or name in want:
            print(decl(name, ret, params))
