#!/usr/bin/env python3
"""time the dense DistHamming tile kernel (gs_hamming_qxc_dev) on synthetic signatures resident in HBM"""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsearch_amd as G
from gsearch_amd import _lib
nq, nc, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = int(sys.argv[4]) if len(sys.argv) > 4 else 3
esz = 8 if kind == 2 else 4
ctx = G.Context(0); L = ctx.L
dq, dc, do = ctx.alloc(nq * m * esz), ctx.alloc(nc * m * esz), ctx.alloc(nq * nc * 4)
_lib.check(L.gs_synth_sigs_dev(ctx.h, kind, m, 1, 0, nq, 100, 0.3, 0.9, dq))
_lib.check(L.gs_synth_sigs_dev(ctx.h, kind, m, 1, 1000000, nc, 100, 0.3, 0.9, dc))
ctx.sync()
for it in range(3):
    ctx.timer_start()
    _lib.check(L.gs_hamming_qxc_dev(ctx.h, kind, m, dq, nq, dc, nc, do))
    ms = ctx.timer_stop()
    pe = nq * nc * m
    print("nq=%d nc=%d m=%d kind=%d: %.2f ms  %.3e pair-elements/s  algorithmic %.1f TB/s" % (nq, nc, m, kind, ms, pe / ms * 1e3, nq * nc * m * esz / ms * 1e3 / 1e12))
