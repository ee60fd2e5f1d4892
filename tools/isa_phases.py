#!/usr/bin/env python3
"""Instruction mix of a kernel's phases (before the first s_barrier / between first and last / after the last).
usage: isa_phases.py file.s symbol-substring"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and sys.argv[2] in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
ins = [l.strip() for l in lines[start + 1:end] if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().split(";")[0].strip().endswith(":")]
bars = [i for i, l in enumerate(ins) if l.startswith("s_barrier")]
def count(seg):
    c = {}
    for l in seg:
        op = l.split()[0]
        k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
             "vmem" if op.startswith(("buffer_", "global_", "flat_")) else "lds" if op.startswith("ds_") else "other")
        c[k] = c.get(k, 0) + 1
    return c
print(len(ins), "instructions, barriers at", bars)
print("prologue", count(ins[:bars[0]]))
print("loop    ", count(ins[bars[0]:bars[-1]]))
print("epilogue", count(ins[bars[-1]:]))
