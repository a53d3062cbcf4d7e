#!/usr/bin/env python3
"""Issue-cost census of one kernel in the hipcc ISA listing (make -C muscle_amd/csrc asm), weighted with the costs
measured by diag/pkbench on MI355X (profiles/r02_pkbench.log): unit = one v_add_f32 (2 cycles per wave on a SIMD).
  full rate (1.0): add/sub/mul f32, and/or/xor, add_u32/sub_u32, mov, cmp
  ~1.67          : min/max (f32/u32), med3, floor/ceil/cvt, shifts, bfe, lshl_add, cndmask, DPP forms, v_pk_*, readlane(+)
  1.57           : fma
usage: scripts/isa_cost.py <listing.s> <kernel-name-substring> [--blocks]
Prints, per basic block (label), instruction counts by class and weighted VALU units; blocks sorted by units."""
import collections
import re
import sys

FULL = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_mov_b32", "v_cmp", "v_add_co", "v_accvgpr", "v_not_b32", "v_nop")


def cost(op, line):
    if not op.startswith("v_"):
        return 0.0
    if "dpp" in line or "sdwa" in op or "_sdwa" in line:
        base = 1.67 if "dpp" in line else None
        if base:
            return base
    if op.startswith("v_pk_"):
        return 1.8
    if op.startswith("v_fma") or op.startswith("v_mad") or op.startswith("v_fmac"):
        return 1.57
    for f in FULL:
        if op.startswith(f):
            return 1.0
    return 1.67


def mean_cost(path, key, walk=False):
    """Mean issue cost (units of one full-rate VALU op) of the VALU instructions of kernel `key` in listing `path`: the whole
    kernel, or — walk=True — what lies between the first and the last hand-scheduled merge loop (.Lrv_step_*), i.e. one step of
    the relax walk. A static mean: every instruction of the region counts once."""
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l and "@" in l)
    body = []
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        body.append(s)
    if walk:
        marks = [i for i, s in enumerate(body) if re.match(r"\.Lrv_step_\d+:", s)]
        if marks:
            body = body[marks[0]:marks[-1] + 60]
    units = n = 0
    in_loop = walk  # whole kernel: only the basic blocks the compiler marks as part of a loop count (prologue and epilogue run once)
    for s in body:
        if s.startswith(".LBB") or s.startswith("; %bb."):
            in_loop = walk or "in Loop:" in s or "Loop Header" in s
            continue
        if s.startswith(";") and ("Parent Loop" in s or "Inner Loop Header" in s or "Loop Header" in s):
            in_loop = True
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        if in_loop and op.startswith("v_"):
            units += cost(op, s)
            n += 1
    return units / max(n, 1), n


def main():
    if sys.argv[1] == "--json":  # --json <listing.s> <out.json>: the mean issue costs bench.py's roofline uses, for the default kernels
        import json
        path, out = sys.argv[2], sys.argv[3]
        res = {}
        for name, key, walk in (("relax_band_kernel", "relax_band_kernelILi1024ELi15ELi2ELi0E11MpcRbWinAsm", True),
                                ("relax_band_kernel/MpcRbBlocksAsm", "relax_band_kernelILi1024ELi13ELi2ELi0E14MpcRbBlocksAsm", True),
                                ("relax_var_kernel", "relax_var_kernelILi1024ELi13ELi2ELi0E14MpcRvBlocksAsm", True),
                                ("fb_chain_kernel", "fb_chain_kernelILi7E", False), ("fb_kernel", "fb_kernelILi7ELb0ELb0E", False)):
            try:
                c, n = mean_cost(path, key, walk)
                res[name] = {"mean_issue_cost": round(c, 4), "valu_instructions": n, "region": "one step of the walk" if walk else "the kernel's loops", "mangled": key}
            except StopIteration:
                pass
        json.dump(res, open(out, "w"), indent=1, sort_keys=True)
        print(json.dumps(res))
        return
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end") or s.startswith("s_endpgm"):
            break
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        blocks[cur].append(s)
    rows = []
    for name, ins in blocks.items():
        cls = collections.Counter()
        units = 0.0
        for s in ins:
            op = s.split()[0]
            c = cost(op, s)
            units += c
            if op.startswith("v_"):
                k = "dpp" if "dpp" in s else op.split("_e32")[0].split("_e64")[0]
                cls[k] += 1
            elif op.startswith("ds_"):
                cls[op] += 1
            elif op.startswith("global_") or op.startswith("buffer_"):
                cls[op] += 1
            elif op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_cbranch"):
                cls[op.split("_vccz")[0]] += 1
            else:
                cls["salu/other"] += 1
        rows.append((units, name, len(ins), cls))
    tot = sum(r[0] for r in rows)
    print("kernel %s: %d blocks, %d instructions, %.0f VALU units" % (key, len(rows), sum(r[2] for r in rows), tot))
    for units, name, n, cls in sorted(rows, key=lambda r: -r[0])[:int(sys.argv[4]) if len(sys.argv) > 4 else 8]:
        print("\n%s: %d instr, %.0f units" % (name, n, units))
        print("  " + ", ".join("%s %d" % kv for kv in cls.most_common(40)))


if __name__ == "__main__":
    main()
