#!/usr/bin/env python3
"""Emits the Rust `extern "C"` declarations of every function include/gsearch_amd.h declares (INTEGRATION.md section 1 is generated
with it; tests/test_abi_cpu.py checks that the document still names every exported symbol).  usage: gen_rust_extern.py > block.rs"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "include", "gsearch_amd.h")).read()
src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
src = re.sub(r"//[^\n]*", "", src)
body = src[src.index('extern "C" {') + 12:src.rindex("#ifdef __cplusplus")]
body = re.sub(r"^\s*#.*$", "", body, flags=re.M)
body = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", body, flags=re.S)
body = re.sub(r"typedef\s+struct\s+\w+\s+\w+\s*;", "", body)
body = re.sub(r"enum\s*\{.*?\}\s*;", "", body, flags=re.S)
TY = {"int": "c_int", "void": "c_void", "char": "c_char", "float": "c_float", "double": "c_double", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32",
      "uint16_t": "u16", "uint8_t": "u8", "int64_t": "i64", "int32_t": "i32", "gs_ctx": "GsCtx", "gs_index": "GsIndex", "gs_comm": "GsComm",
      "gs_sketch_params": "GsSketchParams", "gs_index_params": "GsIndexParams"}


def rust_type(c):
    c = c.strip()
    const = "const" in c.split("*")[0].split()
    base = [w for w in c.replace("*", " ").split() if w not in ("const", "unsigned")][0]
    t = TY[base]
    for i in range(c.count("*")):
        t = ("*const " if (const and i == 0) else "*mut ") + t
    return t


def functions():
    for m in re.finditer(r"([\w\s\*]+?)\b(gs_\w+)\s*\(([^;]*?)\)\s*;", body, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for i, a in enumerate(args.split(",")):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)(\[\d*\])?$", a)
                if a.endswith("*") or not mm:             # unnamed pointer parameter
                    ty, nm, arr = a, "arg%d" % i, None
                else:
                    ty, nm, arr = mm.group(1).strip(), mm.group(2), mm.group(3)
                    if not ty or nm in TY:                # unnamed value parameter
                        ty, nm = a, "arg%d" % i
                if arr:
                    ty += " *"
                if nm.startswith("arg"):
                    rt = rust_type(ty + (" *" if arr else ""))
                    nm = {"GsCtx": "ctx", "GsIndex": "ix", "GsComm": "comm", "GsSketchParams": "p", "GsIndexParams": "prm"}.get(rt.split()[-1], nm)
                params.append((nm if nm not in ("type", "in", "ref", "box", "fn", "mod", "move") else nm + "_", rust_type(ty)))
        yield name, params, (None if ret == "void" else rust_type(ret))


if __name__ == "__main__":
    for name, params, ret in functions():
        line = "    pub fn %s(%s)%s;" % (name, ", ".join("%s: %s" % p for p in params), (" -> " + ret) if ret else "")
        print(line)
