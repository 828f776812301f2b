#!/usr/bin/env python3
"""Static check of the 4-wave GEMM's hand-placed MFMA hazards on a `hipcc --save-temps` .s (no GPU).

The kernel's MFMAs are `asm volatile`: the compiler's hazard recognizer does not see them, so correctness rests on (a) no
compiler-generated instruction touching an accumulator register inside the K loop, and (b) the `s_nop 15 ; s_nop 15` fence
standing between the last MFMA of a tile and the first compiler-generated access to any accumulator.  For every
gemm_nt256w4_kernel instantiation this script collects the accumulator registers (the destinations of the asm MFMAs), then
walks the instruction stream: after an asm MFMA, any instruction OUTSIDE an asm block that names one of those registers before
the fence (or before the next loop iteration's MFMAs have all issued) is reported.

    tools/check_asm_mfma.py file.s            -> exit 1 and a list of offending lines, or "clean: N kernels"
"""
import re, sys


def regs_of(tok):
    """'a[4:7]' -> {('a',4..7)}, 'v12' -> {('v',12)}"""
    out = set()
    for kind, a, b, single in re.findall(r"\b([av])\[(\d+):(\d+)\]|\b([av]\d+)\b", tok):
        if single:
            out.add((single[0], int(single[1:])))
        else:
            out.update((kind, r) for r in range(int(a), int(b) + 1))
    return out


def check_kernel(name, lines):
    bad, in_asm, acc = [], False, set()
    # pass 1: accumulators = destinations of MFMAs inside asm blocks with the "dst == srcC" form of the main loop
    for l in lines:
        s = l.strip()
        if s.startswith(";;#ASMSTART"): in_asm = True
        elif s.startswith(";;#ASMEND"): in_asm = False
        elif in_asm and s.startswith("v_mfma"):
            ops = [o.strip() for o in s.split(None, 1)[1].split(",")]
            if len(ops) >= 4 and ops[0] == ops[3]:
                acc |= regs_of(ops[0])
    if not acc:
        return bad, 0
    # pass 2 (no CFG needed).  The compiler DOES touch accumulators between the asm MFMAs — it renumbers them with v_accvgpr_mov /
    # _write where a loop version hands over to the next (tail iterations, persistent tile loop).  That is harmless when the
    # value it reads was produced long ago and the value it writes is consumed late enough; it is the round-3 bug class when it
    # is not.  Inside every basic block that holds accumulator MFMAs, for each instruction outside an asm statement that names an
    # accumulator register r:
    #   * reading r: the nearest asm MFMA in front of it (same block) that WRITES r must be >= MIN_READ instructions back (an
    #     8-pass MFMA's result is readable ~11 wait states later; other MFMAs in between each count 4);
    #   * writing r: the nearest asm MFMA behind it that takes r (accumulator in / out) must be >= MIN_WRITE instructions on
    #     (VALU write -> MFMA srcC: 2 wait states);
    # and the fence statement (s_nop 15 x 2) resets the read rule: behind it the epilogue reads every accumulator.
    MIN_READ, MIN_WRITE = 12, 2
    labels = [0] + [i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)] + [len(lines)]
    def instrs(lo, hi):
        out, in_asm = [], False
        for n in range(lo, hi):
            s = lines[n].strip()
            if s.startswith(";;#ASMSTART"): in_asm = True; continue
            if s.startswith(";;#ASMEND"): in_asm = False; continue
            if not s or s.startswith((";", ".")) or s.endswith(":"):
                continue
            out.append((n, s, in_asm))
        return out
    for lo, hi in zip(labels, labels[1:]):
        ins = instrs(lo, hi)
        mf = [k for k, (n, s, ia) in enumerate(ins) if ia and s.startswith("v_mfma") and regs_of(s.split(None, 1)[1].split(",")[0]) & acc]
        if not mf:
            continue
        fence = [k for k, (n, s, ia) in enumerate(ins) if ia and s.startswith("s_nop 15")]
        for k, (n, s, ia) in enumerate(ins):
            if ia or " " not in s or s.startswith("s_"):
                continue
            ops = [o.strip() for o in s.split(None, 1)[1].split(",")]
            dst, src = regs_of(ops[0]) & acc, set().union(*[regs_of(o) for o in ops[1:]]) & acc if len(ops) > 1 else set()
            if s.startswith(("global_store", "ds_write", "buffer_store")):
                dst, src = set(), regs_of(s.split(None, 1)[1]) & acc
            for r in src:
                prev = [j for j in mf if j < k and r in regs_of(ins[j][1].split(None, 1)[1].split(",")[0])]
                if prev and not any(prev[-1] < f < k for f in fence):
                    dist = sum(4 if ins[j][1].startswith("v_mfma") else 1 for j in range(prev[-1] + 1, k))
                    if dist < MIN_READ:
                        bad.append(f"{name}: line {n}: `{s[:80]}` reads {r[0]}{r[1]} {dist} wait states behind the asm MFMA that writes it (line {ins[prev[-1]][0]})")
            for r in dst:
                nxt = [j for j in mf if j > k and r in regs_of(ins[j][1].split(None, 1)[1])]
                if nxt and nxt[0] - k - 1 < MIN_WRITE:
                    bad.append(f"{name}: line {n}: `{s[:80]}` writes {r[0]}{r[1]} {nxt[0] - k - 1} instructions in front of the asm MFMA that takes it (line {ins[nxt[0]][0]})")
    return sorted(set(bad)), len(acc)


def main(path):
    txt = open(path).read().split("\n")
    starts = [i for i, l in enumerate(txt) if re.match(r"^_Z\w*gemm_nt256w4_kernel\w*:", l)]
    total, bad = 0, []
    for st in starts:
        end = next(i for i in range(st, len(txt)) if "s_endpgm" in txt[i])
        b, nacc = check_kernel(txt[st].split(":")[0][:60], txt[st:end])
        if nacc:
            total += 1
            bad += b
    if bad:
        print("\n".join(bad[:40]))
        print(f"{len(bad)} hazards in {total} kernels")
        return 1
    print(f"clean: {total} kernels with asm MFMAs; every compiler-generated accumulator access keeps its distance from them")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
