#!/usr/bin/env python3
"""Static view of one kernel of a `hipcc -S --cuda-device-only` listing: its basic blocks with instruction counts by class, and its
loops (backward branches) with the instructions between branch target and branch.  Costs no GPU time: a lone wave issues one
instruction per ~5.4 cycles (profiles/r01_ubench_pkfma.txt), so the instruction count of a loop body is its duration to first order.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -fno-slp-vectorize -S --cuda-device-only -o /tmp/isa/pbre_capi.s pbre_capi.hip
    python tools/isa_blocks.py /tmp/isa/pbre_capi.s 'k_row_list<7>'
"""
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z[\w]*:", l)]
    dm = demangle([n for _, n in starts])
    sel = [(i, n) for i, n in starts if pat in dm[n]]
    if not sel:
        sys.exit("no kernel matches %r" % pat)
    for i0, name in sel:
        print("==", dm[name][:120])
        end = next(j for j in range(i0, len(lines)) if lines[j].startswith(".Lfunc_end"))      # (a kernel may hold several s_endpgm)
        # walk: labels and instructions
        blocks, cur, pos = [], None, {}
        idx = 0
        insts = []
        for j in range(i0 + 1, end + 1):
            l = lines[j].split(";")[0].strip()
            if not l or l.startswith("."):
                if re.match(r"^\.LBB[\w]*:$", l):
                    pos[l[:-1]] = idx
                continue
            m = re.match(r"^(\.LBB[\w]*):$", l)
            if m:
                pos[m.group(1)] = idx
                continue
            op = l.split()[0]
            insts.append((op, l))
            idx += 1
        tot = {}
        for op, _ in insts:
            k = classify(op)
            tot[k] = tot.get(k, 0) + 1
        print("   instructions:", len(insts), tot)
        dpp = sum(1 for op, l in insts if "dpp" in l or "row_" in l or "quad_perm" in l)
        scr = sum(1 for op, _ in insts if op.startswith("scratch_"))
        print("   dpp:", dpp, " scratch:", scr)
        loops = []
        for k, (op, l) in enumerate(insts):
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = l.split()[-1]
                if tgt in pos and pos[tgt] <= k:
                    loops.append((pos[tgt], k))
        for a, b in sorted(loops, key=lambda t: t[0] - t[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
            cnt = {}
            for op, l in insts[a:b + 1]:
                c = classify(op)
                cnt[c] = cnt.get(c, 0) + 1
            d = sum(1 for op, l in insts[a:b + 1] if "dpp" in l or "row_" in l or "quad_perm" in l)
            print("   loop [%6d .. %6d] %6d instr %s dpp %d" % (a, b, b - a + 1, cnt, d))


if __name__ == "__main__":
    main()
