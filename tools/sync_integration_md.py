#!/usr/bin/env python3
"""Rewrites the `pub fn` lines of INTEGRATION.md's `extern "C"` block from include/gsearch_amd.h (tools/gen_rust_extern.py), wrapped at 150 columns,
leaving the comment lines at the top of the block alone. Run after every change to the header:  python tools/sync_integration_md.py"""
import importlib.util, os, textwrap
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_rust_extern", os.path.join(ROOT, "tools", "gen_rust_extern.py"))
gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
path = os.path.join(ROOT, "INTEGRATION.md")
lines = open(path).read().split("\n")
a = next(i for i, l in enumerate(lines) if l.startswith('extern "C" {'))
first = next(i for i in range(a, len(lines)) if lines[i].lstrip().startswith("pub fn "))
end = next(i for i in range(first, len(lines)) if lines[i].startswith("}"))
out = []
for name, params, ret in gen.functions():
    decl = "pub fn %s(%s)%s;" % (name, ", ".join("%s: %s" % p for p in params), (" -> " + ret) if ret else "")
    out.extend(textwrap.wrap(decl, width=146, initial_indent="    ", subsequent_indent="        ", break_long_words=False, break_on_hyphens=False))
lines[first:end] = out
open(path, "w").write("\n".join(lines))
print("INTEGRATION.md: %d functions" % len(list(gen.functions())))
