#!/usr/bin/env python3
"""Instruction mix of the step loop of fill_ring_kernel<M,1,false> in a hipcc -S dump.
usage: count_loop.py file.s [M]   (classes per tools/ubench_ops.hip: A = full-rate VALU, B = half-rate)"""
import re, sys
args = [a for a in sys.argv[1:] if not a.startswith("-")]
M = int(args[1]) if len(args) > 1 else 3
txt = open(args[0]).read()
name = "_ZN3cvx16fill_ring_kernelILi%dELb0ELi0EEEvNS_8FillArgsE" % M      # <M, WRAP = false, MODE = kFillTwoPhase>
body = txt[txt.index(name + ":"):]
body = body[:body.index("s_endpgm")].split("\n")
# the step loop = the longest stretch between a label and a backward branch to it
labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
best = (0, 0, 0)
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        j = labels[m.group(1)]
        # the step loop is the smallest loop holding all 8*M plane updates
        n_addc = sum(1 for q in body[j:i] if "v_addc_co_u32_e64" in q)
        if n_addc >= 8 * M and (best[0] == 0 or i - j < best[0]): best = (i - j, j, i)
_, j, i = best
A = ("v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32_e32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
     "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_f16", "v_mac_f32")
na = nb = ns = nm = 0
ops = {}
for l in body[j:i + 1]:
    t = l.strip().split()
    if not t or t[0].startswith((";", ".")): continue
    op = t[0]
    if op.startswith("v_"):
        if op.startswith(A) and "dpp" not in op and "sdwa" not in op: na += 1
        else: nb += 1
    elif op.startswith("s_"): ns += 1
    elif op.startswith(("global_", "ds_", "buffer_", "scratch_", "flat_")): nm += 1
    ops[op] = ops.get(op, 0) + 1
cells = 4 * M
print("loop lines %d..%d: A %d  B %d  SALU %d  mem %d  | per cell: A %.2f B %.2f S %.2f" % (j, i, na, nb, ns, nm, na / cells, nb / cells, ns / cells))
if "-v" in sys.argv:
    for k, v in sorted(ops.items(), key=lambda kv: -kv[1]): print("  %-28s %d" % (k, v))
