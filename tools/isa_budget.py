#!/usr/bin/env python3
"""Issue-slot budget of a kernel from its gfx950 assembly.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o conv.s savsr_amd/csrc/conv_mfma.hip
    python3 tools/isa_budget.py conv.s 'conv_bf16x3_kernelILi3ELi2ELi2ELb0' [--top 12] [--dump LABEL]

Splits the kernel into basic blocks (labels), classifies every instruction (MFMA / VALU / SALU / DS read / DS write /
VMEM load / VMEM store / LDS-DMA / waitcnt / barrier / branch / nop) and prices each class with the measured issue costs of
MI355X_MICROARCH.md ("vector-instruction ISSUE cost": plain VALU 4 cycles per wave-instruction, v_cvt_pk_bf16_f32 4-5,
transcendental 8, an MFMA holds the vector issue port for 8 of its 32 cycles, s_nop 4; scalar instructions ~1 issue cycle
each on the scalar unit; DS per the LDS table: ds_read_b128 4 array cycles, ds_write_b64 6).  Prints the blocks holding MFMAs
(the K-phase bodies) and the non-MFMA instructions named by opcode, so the ones that are not operand movement can be removed.
"""
import argparse
import collections
import re
import sys

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "MFMA"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "DS_RD"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "DS_WR"
    if op.startswith("ds_"):
        return "DS_OTHER"
    if op.startswith("global_load_lds") or (op.startswith("buffer_load") and "lds" in op):
        return "LDS_DMA"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "VMEM_LD"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
        return "VMEM_ST"
    if op == "s_waitcnt":
        return "WAIT"
    if op == "s_barrier":
        return "BARRIER"
    if op == "s_nop":
        return "NOP"
    if op.startswith(("s_cbranch", "s_branch")):
        return "BRANCH"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("v_"):
        return "VALU"
    return "OTHER"


def issue_cost(op, cls):
    if cls == "MFMA":
        return 8
    if cls == "VALU":
        if op.startswith(TRANS):
            return 8
        if op.startswith("v_pk_") and "f32" in op:
            return 8            # packed fp32: an anti-lever beside MFMAs (guide: +22..26 cycles per pair vs plain)
        return 4
    if cls == "NOP":
        return 4
    if cls in ("SALU", "SMEM", "BRANCH", "WAIT", "BARRIER"):
        return 1
    if cls == "DS_RD":
        return 4
    if cls == "DS_WR":
        return 6 if "b64" in op else (13 if "b128" in op else 4)
    if cls in ("VMEM_LD", "VMEM_ST"):
        return 4
    if cls == "LDS_DMA":
        return 60
    return 1


def parse(path, kernel_pat):
    lines = open(path).read().splitlines()
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\w*%s\w*:" % re.escape(kernel_pat), l):
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    if start is None:
        sys.exit("kernel not found: " + kernel_pat)
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1:end]:
        s = l.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        blocks[cur].append((parts[0], parts[1] if len(parts) > 1 else ""))
    return blocks


def summarise(ins):
    c = collections.Counter()
    cost = collections.Counter()
    ops = collections.Counter()
    for op, _ in ins:
        cls = classify(op)
        c[cls] += 1
        cost[cls] += issue_cost(op, cls)
        if cls not in ("MFMA",):
            ops[op] += 1
    return c, cost, ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--top", type=int, default=8)
    ap.add_argument("--dump", default=None, help="print the instructions of one block")
    ap.add_argument("--ops", type=int, default=25, help="opcodes listed per block")
    a = ap.parse_args()
    blocks = parse(a.asm, a.kernel)
    if a.dump:
        for op, args in blocks[a.dump]:
            print(f"    {op} {args}")
        return
    tot, totcost, _ = summarise([i for b in blocks.values() for i in b])
    print(f"kernel {a.kernel}: {len(blocks)} blocks, {sum(tot.values())} instructions")
    print("  static totals:", dict(tot))
    ranked = sorted(blocks.items(), key=lambda kv: -summarise(kv[1])[0]["MFMA"])
    for name, ins in ranked[:a.top]:
        c, cost, ops = summarise(ins)
        if not c["MFMA"]:
            break
        n = c["MFMA"]
        print(f"\nblock {name}: {len(ins)} instructions, {n} MFMAs")
        print("  per MFMA: " + ", ".join(f"{k} {c[k] / n:.2f}" for k in ("VALU", "SALU", "SMEM", "DS_RD", "DS_WR", "VMEM_LD", "VMEM_ST", "LDS_DMA", "WAIT", "NOP", "BARRIER", "BRANCH") if c[k]))
        vec = cost["MFMA"] + cost["VALU"] + cost["NOP"] + cost["DS_RD"] + cost["DS_WR"] + cost["VMEM_LD"] + cost["VMEM_ST"] + cost["LDS_DMA"]
        print(f"  priced issue cycles of ONE wave per MFMA: vector port {vec / n:.1f} (MFMA 8 + VALU {cost['VALU'] / n:.1f} + nop {cost['NOP'] / n:.1f} + DS {(cost['DS_RD'] + cost['DS_WR']) / n:.1f}"
              f" + VMEM {(cost['VMEM_LD'] + cost['VMEM_ST']) / n:.1f} + DMA {cost['LDS_DMA'] / n:.1f}); scalar {(cost['SALU'] + cost['SMEM'] + cost['BRANCH'] + cost['WAIT'] + cost['BARRIER']) / n:.1f}"
              f"  -> two waves per SIMD need {2 * vec / n:.1f} of the 64 cycles two MFMAs take")
        print("  opcodes: " + ", ".join(f"{op}×{k}" for op, k in ops.most_common(a.ops)))


if __name__ == "__main__":
    main()
