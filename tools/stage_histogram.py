#!/usr/bin/env python3
"""Instruction mix of a hand-scheduled forward kernel between consecutive s_barrier instructions (one barrier per stage / row):
MFMA, VALU, LDS, vector memory, scalar, s_nop (with its idle cycles), s_waitcnt counts - and the issue-model estimate
max(16.4, 11.3 + 4.2 VALU/MFMA) cycles per MFMA (profiles/r04_microbench_wino_issue_model.txt).

    python tools/stage_histogram.py build/obj/net_forward_w1dband.hip.o w1dband_kernelILb0"""
import sys, os, re
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_check import disassemble, kernels


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"): return "lane"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if op == "s_nop": return "nop"
    if op == "s_waitcnt": return "wait"
    if op == "s_barrier": return "barrier"
    if op.startswith("s_cbranch") or op == "s_branch": return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    obj, flt = sys.argv[1], sys.argv[2]
    ks = kernels(disassemble(obj))
    for name, ins in ks.items():
        if flt not in name: continue
        print(name)
        seg, segs = {}, []
        for addr, op, args, tgt in ins:
            c = classify(op)
            seg[c] = seg.get(c, 0) + 1
            if c == "nop":
                seg["nopcyc"] = seg.get("nopcyc", 0) + int(args.strip() or 0) + 1
            if c == "barrier":
                segs.append(seg); seg = {}
        segs.append(seg)
        keys = ["mfma", "valu", "lane", "lds", "vmem", "salu", "branch", "nop", "nopcyc", "wait"]
        print("seg  " + " ".join(f"{k:>6s}" for k in keys) + "   model")
        for i, s in enumerate(segs):
            m = s.get("mfma", 0)
            v = s.get("valu", 0) + s.get("lane", 0)
            model = m * max(16.4, 11.3 + 4.2 * v / m) if m else 4.5 * v
            print(f"{i:3d}  " + " ".join(f"{s.get(k, 0):6d}" for k in keys) + f"  {model:7.0f}")


if __name__ == "__main__":
    main()
