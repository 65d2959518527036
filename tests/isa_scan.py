"""Static check of the built gfx950 code objects for the store-data hazard hipcc under-pads
(dfx_device.cuh, tools/probe/probe_store_war.hip): a VALU instruction must not overwrite a data
register of a 16-byte global/buffer/flat store fewer than 2 wait states after it.  Follows both
sides of branches inside the window.  Pure text processing of `llvm-objdump -d` output."""
import os
import re
import subprocess
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
STORE16 = re.compile(r"^(global|buffer|flat|scratch)_store_dwordx[34]\b")
NEED = 2


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def code_objects(lib):
    """extract the gfx950 code objects embedded in a HIP fat binary -> list of paths (temp dir)"""
    tmp = tempfile.mkdtemp(prefix="dfx_isa_")
    local = os.path.join(tmp, os.path.basename(lib))
    with open(lib, "rb") as f, open(local, "wb") as g:
        g.write(f.read())
    subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(os.path.join(tmp, n) for n in os.listdir(tmp) if "gfx950" in n)


def scan_object(path):
    """-> (number of 16-byte stores, list of violations (kernel, store line, offending line))"""
    out = subprocess.run([OBJDUMP, "-d", path], capture_output=True, text=True).stdout
    ins, kernel_of, addr_index = [], [], {}
    kernel = "?"
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            kernel = m.group(1)
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-F]+):", line)
        if not m:
            continue
        addr_index[int(m.group(2), 16)] = len(ins)
        ins.append((int(m.group(2), 16), m.group(1)))
        kernel_of.append(kernel)
    nstores, bad = 0, []

    def weight(text):
        m = re.match(r"^s_nop (\d+)", text)
        return int(m.group(1)) + 1 if m else 1

    def walk(i, data, have, seen):
        """scan from instruction i with `have` wait states already behind the store"""
        while i < len(ins) and have < NEED:
            if (i, have) in seen:
                return None
            seen.add((i, have))
            addr, text = ins[i]
            op = text.split()[0]
            if op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_readlane") \
                    and not op.startswith("v_readfirstlane"):
                dst = regs(text.split()[1]) if len(text.split()) > 1 else set()
                if dst & data:
                    return text
            if op in ("s_endpgm",):
                return None
            m = re.match(r"^s_(c?branch\w*) (\d+)", text)
            if m:
                simm = int(m.group(2))
                simm -= 65536 if simm >= 32768 else 0
                tgt = addr_index.get(addr + 4 + 4 * simm)
                if tgt is not None:
                    r = walk(tgt, data, have + 1, seen)
                    if r:
                        return r
                if op == "s_branch":
                    return None
            have += weight(text)
            i += 1
        return None

    for i, (addr, text) in enumerate(ins):
        if not STORE16.match(text):
            continue
        nstores += 1
        toks = [t for t in re.split(r"[ ,]+", text) if t]
        # data operand: the first multi-register vector operand that is 3 or 4 registers wide
        data = set()
        for t in toks[1:]:
            r = regs(t)
            if len(r) in (3, 4):
                data = r
                break
        off = walk(i + 1, data, 0, set())
        if off:
            bad.append((kernel_of[i], text, off))
    return nstores, bad


def packed_f32_opsel_forms(path):
    """-> Counter of (opcode, op_sel text) of every packed-f32 instruction in the code object"""
    import collections
    out = subprocess.run([OBJDUMP, "-d", path], capture_output=True, text=True).stdout
    c = collections.Counter()
    for line in out.splitlines():
        m = re.search(r"\b(v_pk_(?:add|mul|fma)_f32)\b([^/]*)", line)
        if m:
            sel = re.findall(r"op_sel(?:_hi)?:\[[01,]+\]", m.group(2))
            c[(m.group(1), " ".join(sel))] += 1
    return c
