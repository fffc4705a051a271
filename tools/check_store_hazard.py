"""Scans the device code of libunflow_hip.so (or assembly listings) for a store-data hazard of gfx950 that LLVM's hazard
recognizer does not cover: a VALU instruction that overwrites a data register of a VMEM store of more than 64 bits within
two instructions of the store.  Measured (tools/ubench/store_data_hazard.hip, profiles/ubench_store_data_hazard.txt):
`buffer_store_dwordx4 v[8:11], v0, s[8:11], s6 offen ; v_mov_b32 v8, 0` stores the NEW v8 in dword 0 of lanes 12-15
(mod 16) in 0.5 % of the stores; with an immediate soffset even one s_nop in between leaves 0.4 %; two wait states are
clean.  The compiler emits the SGPR-soffset form without any wait state (it takes it to be safe) -- it bit two of
round 3's experimental kernels; the shipped kernels are checked by this script (tests/test_codegen_hazards.py).

usage: python tools/check_store_hazard.py [libunflow_hip.so | file.s ...]      exit code 1 = found"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
STORE = re.compile(r"^\s*((?:buffer|global|flat|scratch)_store_(?:dwordx[34]|b96|b128))\s+(.*)$")
WINDOW = 2       # wait states after the store in which its data registers must not be written (s_nop N = N + 1 states)


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def _valu_dst(text):
    m = re.match(r"(v_\w+)\s+([^,]+)", text)
    if not m:
        return set()
    op = m.group(1)
    if op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_mfma", "v_smfmac")) and not op.startswith("v_cmpx"):
        return _regs(m.group(2).strip()) if op.startswith(("v_mfma", "v_smfmac")) else set()
    return _regs(m.group(2).strip())


def scan(lines, name):
    """Walks every path of up to WINDOW wait states behind a wide store: straight on, and through branches whose target
    the listing gives (objdump prints the address of every instruction; a branch's operand is its 16-bit dword offset)."""
    found = []
    kern = "?"
    code = []      # (kernel, text, address or None)
    for raw in lines:
        parts = raw.split("//")
        t = parts[0].rstrip()
        m = re.match(r"^(?:[0-9a-f]+ )?<?(_Z\w+)>?:", t.strip())     # kernel label (.s) or symbol line (objdump)
        if m:
            kern = m.group(1)
            continue
        s = t.strip()
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        am = re.match(r"\s*([0-9A-Fa-f]+):", parts[1]) if len(parts) > 1 else None
        code.append((kern, s, int(am.group(1), 16) if am else None))
    at = {a: i for i, (_, _, a) in enumerate(code) if a is not None}

    def walk(store, data, i, states, seen):
        while states < WINDOW and i < len(code):
            nxt = code[i][1]
            if _valu_dst(nxt) & data:
                found.append("%s [%s]: %s   -> %s" % (name, code[store][0][:80], code[store][1], nxt))
                return
            if nxt.startswith("s_endpgm"):
                return
            bm = re.match(r"s_(c?branch\w*)\s+(\d+)", nxt)
            if bm or re.match(r"s_(setpc|swappc)", nxt):
                tgt = None
                if bm and code[i][2] is not None:
                    off = int(bm.group(2))
                    tgt = at.get(code[i][2] + 4 + 4 * (off - 65536 if off >= 32768 else off))
                if tgt is None:
                    found.append("%s [%s]: %s   -> %s (branch inside the hazard window, target unknown)" % (name, code[store][0][:80], code[store][1], nxt))
                    return
                if (tgt, states) not in seen:
                    seen.add((tgt, states))
                    walk(store, data, tgt, states + 1, seen)
                if nxt.startswith("s_branch"):
                    return
            m2 = re.match(r"s_nop\s+(\d+)", nxt)
            states += int(m2.group(1)) + 1 if m2 else 1
            i += 1

    for i, (kern, s, _) in enumerate(code):
        m = STORE.match(s)
        if not m:
            continue
        ops = [x.strip() for x in m.group(2).split(",")]
        data = _regs(ops[0]) if m.group(1).startswith("buffer") else (_regs(ops[1]) if len(ops) > 1 else set())
        walk(i, data, i + 1, 0, set())
    return found


LDS_DMA = re.compile(r"^\s*(buffer_load_dword\w*\s.*\blds\b|global_load_lds_\w+)")
M0_WRITE = re.compile(r"^\s*s_\w+\s+m0\b")


def scan_m0(lines, name):
    """For COMPILER LISTINGS (hipcc -S: they carry the ;;#ASMSTART / ;;#ASMEND markers).  Every LDS-DMA instruction
    takes its LDS destination from M0.  The hand-written instruction blocks write M0 themselves in front of each of
    theirs, and the compiler does not model that (an "m0" clobber only draws a warning: the register is reserved), so
    a compiler-issued LDS-DMA that relied on an M0 value from BEFORE such a block would land in the wrong place.  The
    invariant that rules it out: walking back from every LDS-DMA instruction, the nearest write of M0 comes before any
    label and on the same side of the asm markers (a block's DMA: inside the same block; the compiler's: no block in
    between).  Returns (findings, number of LDS-DMA instructions)."""
    found, kern, code = [], "?", []        # (kernel, kind, text): kind = "i" instruction, "l" label, "s" ASMSTART, "e" ASMEND
    for raw in lines:
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            code.append((kern, "s", t))
        elif t.startswith(";;#ASMEND"):
            code.append((kern, "e", t))
        elif re.match(r"^[\w.$]+:", t):
            m = re.match(r"^(_Z\w+):", t)
            if m:
                kern = m.group(1)
            if not re.match(r"^\d+:", t):          # numeric local labels belong to the blocks themselves
                code.append((kern, "l", t))
        elif not t or t.startswith((";", ".", "//")):
            continue
        else:
            code.append((kern, "i", t.split(";")[0].strip()))
    n = 0
    for i, (kern, kind, s) in enumerate(code):
        if kind != "i" or not LDS_DMA.match(s):
            continue
        n += 1
        j, why = i - 1, None
        while j >= 0:
            k, t = code[j][1], code[j][2]
            if k == "i" and M0_WRITE.match(t):
                break
            if k in ("l", "s", "e"):
                why = {"l": "a label", "s": "the start of an asm block", "e": "the end of an asm block"}[k]
                break
            j -= 1
        if why or j < 0:
            found.append("%s [%s]: %s   <- %s between it and the nearest write of m0" % (name, kern[:80], s, why or "nothing"))
    return found, n


def device_listings(so):
    tmp = tempfile.mkdtemp(prefix="unflow_disasm_")
    try:
        local = os.path.join(tmp, os.path.basename(so))
        shutil.copy(so, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:
                out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
                yield f, out.split("\n")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def check(paths, m0=None):
    """m0: a list that receives [findings, number of LDS-DMA instructions] of scan_m0."""
    found, n_stores = [], 0
    m0_found, n_dma = [], 0
    for p in paths:
        if p.endswith(".so") or p.endswith(".o") or p.endswith(".co"):
            listings = device_listings(p)
        else:
            listings = [(os.path.basename(p), open(p).read().split("\n"))]
        for name, lines in listings:
            n_stores += sum(1 for l in lines if STORE.match(l.split("//")[0]))
            found += scan(lines, name)
            if p.endswith(".s"):
                f, n = scan_m0(lines, name)
                m0_found += f
                n_dma += n
    if m0 is not None:
        m0[:] = [m0_found, n_dma]
    return found, n_stores


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sys.argv[1:] or [os.path.join(root, "unflow_amd", "libunflow_hip.so")]
    m0 = []
    found, n = check(paths, m0)
    for f in found + m0[0]:
        print(f)
    print("%d wide VMEM stores scanned, %d with a data register overwritten within %d wait states" % (n, len(found), WINDOW))
    if m0[1]:
        print("%d LDS-DMA instructions scanned, %d whose m0 is not written in their own stretch of code" % (m0[1], len(m0[0])))
    sys.exit(1 if found or m0[0] else 0)
