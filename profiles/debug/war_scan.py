"""Static scan of a gfx950 disassembly (llvm-objdump -d) for the pattern DESIGN.md round 1 blamed for a
one-off wrong result: an LDS/VMEM load whose destination VGPRs are the SrcA/SrcB registers of an MFMA
issued shortly BEFORE it (write-after-read on an MFMA source), and for MFMAs whose destination overlaps
their own SrcA/SrcB.  Prints every such pair with its distance in instructions and the number of MFMAs
issued in between.  CPU-only; usage: war_scan.py listing.s [max_distance]"""
import re, sys

def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()

def main():
    maxd = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    ins = []
    for line in open(sys.argv[1]):
        line = line.split("//")[0].strip()
        if not line or line.endswith(":") or line.startswith(("<", ".", ";")):
            continue
        parts = line.replace(",", " ").split()
        if re.match(r"^[0-9a-f]+$", parts[0]) or parts[0].endswith(">:"):
            continue
        ins.append(parts)
    pairs, overlap = [], []
    for i, p in enumerate(ins):
        if not p[0].startswith("v_mfma"):
            continue
        d, a, b = regs(p[1]), regs(p[2]), regs(p[3])
        if d & (a | b):
            overlap.append((i, " ".join(p)))
        nm = 0
        for j in range(i + 1, min(i + 1 + maxd, len(ins))):
            q = ins[j]
            if q[0].startswith(("s_cbranch", "s_branch", "s_endpgm")):
                break
            if q[0].startswith("v_mfma"):
                nm += 1
                continue
            if q[0].startswith(("ds_read", "global_load", "buffer_load", "scratch_load")):
                w = regs(q[1])
                if w & a or w & b:
                    pairs.append((i, j - i - 1, nm, " ".join(p), " ".join(q), "SrcA" if w & a else "SrcB"))
    print("MFMAs: %d" % sum(1 for p in ins if p[0].startswith("v_mfma")))
    print("load overwrites a source of an MFMA issued <= %d instructions earlier: %d" % (maxd, len(pairs)))
    by = {}
    for _, dist, nm, *_r in pairs:
        by[(dist, nm)] = by.get((dist, nm), 0) + 1
    for k in sorted(by):
        print("   distance %2d instructions, %d MFMAs in between: %d pairs" % (k[0], k[1], by[k]))
    for p in pairs[:6]:
        print("   e.g. #%d  %s  ->  %s  (%s)" % (p[0], p[3], p[4], p[5]))
    print("MFMA destination overlaps its own SrcA/SrcB: %d" % len(overlap))
    for o in overlap[:4]:
        print("   e.g. #%d  %s" % o)

main()
