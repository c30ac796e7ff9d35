#!/usr/bin/env python3
"""Rewrites the `pub fn gs_*` lines of INTEGRATION.md's Rust extern block from include/gsearch_amd.h (tools/gen_rust_extern.py), wrapped at 150 columns."""
import importlib.util, os, textwrap
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen", os.path.join(ROOT, "tools", "gen_rust_extern.py"))
gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
path = os.path.join(ROOT, "INTEGRATION.md")
lines = open(path).read().split("\n")
first = next(i for i, l in enumerate(lines) if l.startswith("    pub fn gs_"))
last = next(i for i in range(first, len(lines)) if lines[i] == "}")
out = []
for name, params, ret in gen.functions():
    decl = "pub fn %s(%s)%s;" % (name, ", ".join("%s: %s" % p for p in params), (" -> " + ret) if ret else "")
    out += textwrap.wrap(decl, 146, initial_indent="    ", subsequent_indent="        ", break_long_words=False)
open(path, "w").write("\n".join(lines[:first] + out + lines[last:]))
print("INTEGRATION.md: %d functions" % len(list(gen.functions())))
