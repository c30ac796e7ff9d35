#!/usr/bin/env python3
"""Static instruction mix of the sketch kernel's per-k-mer hot path: disassembles gs_sketch.hip for gfx950, takes the
interior-word fast loop of k_sketch_min<DNA, LDS table, optdens, 64-bit values, u32 keys> (the span between two consecutive ds_min_u32 of
the unrolled-by-2 loop that carries the fewest 64-bit compares, i.e. no bounds test) and counts instructions per class. Issue cycles per
wave64 instruction: simple VALU 2 (MI355X_MICROARCH.md:52-54); 32-bit integer multiplies / v_mad_u64_u32 weighted by the ratio measured by
tools/ubench_valu (profiles/r02_ubench_valu.txt) when that file is given, else the quarter-rate assumption (x4).
usage: isa_mix.py [ubench_valu.txt] > profiles/r05_sketch_isa_mix.json"""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gsearch_amd", "csrc", "gs_sketch.hip")
asm = os.path.join(tempfile.gettempdir(), "gs_sketch_isa.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", asm, src],
                      stderr=subprocess.DEVNULL)
text = open(asm).read()
a = text.index("_ZN2gs12k_sketch_minILb0ELb1ELi4ELi64EjLb0ELi0EEE")      # <DNA, LDS table, optdens, 64-bit values, u32 keys, unfiltered emitter, canonical compiled in>
body = text[a:text.index("s_endpgm", a)].splitlines()
ins = [l.split()[0] for l in body if l.startswith("\t") and l.strip() and not l.strip().startswith((";", "."))]
mins = [i for i, op in enumerate(ins) if op == "ds_min_u32"]
best = None
for x, y in zip(mins, mins[1:]):
    ops = ins[x + 1:y + 1]
    if len(ops) > 400:
        continue
    n_cmp64 = sum(1 for op in ops if op.startswith("v_cmp_") and "_u64" in op)
    if best is None or (n_cmp64, len(ops)) < (best[0], len(best[1])):
        best = (n_cmp64, ops)
ops = best[1]
# per-opcode issue cycles: measured by tools/ubench_valu when its output is given (cycles at the clock it ran at), else 2 / 8
UB = {}
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    for line in open(sys.argv[1]):
        m = re.match(r"(v_\w+)\s+[0-9.]+ ms\s+->\s+([0-9.]+) ns per wave-instr per SIMD", line)
        if m:
            UB[m.group(1)] = float(m.group(2))
base_ns = UB.get("v_xor_b32")


def cost(op):
    """issue cost of one wave64 instruction in units of a simple VALU op (v_xor_b32 = 1)"""
    o = re.sub(r"_e(32|64)$", "", op)
    if UB and base_ns:
        for k in (o, "v_mul_lo_u32" if o.startswith("v_mul_") else None, "v_cmp_lt_u64" if o.startswith("v_cmp_") and "_u64" in o else None,
                  "v_lshlrev_b64" if o in ("v_lshlrev_b64", "v_ashrrev_i64") else None):
            if k and k in UB:
                return UB[k] / base_ns
        return 1.0
    return 4.0 if o.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32")) else 1.0


cls, weighted = collections.Counter(), 0.0
for op in ops:
    if op.startswith("v_"):
        key = "valu_mul32" if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32")) else ("valu_64bit" if re.search(r"_[bui]64", op) else "valu_simple")
        cls[key] += 1
        weighted += cost(op)
    elif op.startswith("ds_"):
        cls["lds"] += 1
    elif op.startswith("s_"):
        cls["salu_or_branch"] += 1
    else:
        cls["other"] += 1
valu = cls["valu_mul32"] + cls["valu_simple"] + cls["valu_64bit"]
simple_cycles = 2.0 if not (UB and base_ns) else base_ns * 2.4       # cycles of a simple op: guide value, or measured ns at 2.4 GHz
print(json.dumps({"kernel": "k_sketch_min<DNA, LDS table, optdens, u32 keys>: interior-word loop, one k-mer (one half of the unroll-by-2 body)",
                  "instructions_per_kmer": len(ops), "classes": dict(cls), "valu_per_kmer": valu, "valu_weighted_in_simple_ops": weighted,
                  "cycles_per_simple_valu_op": simple_cycles, "issue_cycles_per_kmer": weighted * simple_cycles,
                  "per_opcode_ns_per_wave_instr_per_simd": UB or None,
                  "model": "issue cycles per wave64 group of 64 k-mers = sum over VALU instructions of their measured issue time (tools/ubench_valu, one op stream per SIMD "
                           "at 8 waves/SIMD) when given, else 2 cycles per simple op and 8 per 32-bit multiply (MI355X_MICROARCH.md:52-54); SALU / branch / LDS issue on other ports"}, indent=1))
