"""Writes unflow_amd/csrc/corr_fwd2_asm.inc: the main loop of corr_fwd2_kernel (correlation_fwd2.hip) -- the whole
channel sweep of the FlowNetC correlation forward, from the first LDS-DMA to the last MFMA -- as ONE inline-assembly
statement on fixed vector registers, plus the layout constants the C++ around it shares with it.

Why one statement: every instruction the compiler puts between two MFMA groups costs the matrix pipe issue time
(profiles/ubench_mfma_lds.txt), and what it put there in corr_mfma_fwd_half_kernel was 2.1 scalar + 1.2 vector
instructions per MFMA.  Here a K-step of 4 channels is 18 MFMAs, 10 ds_read_b64, 2-3 LDS-DMA and ~12 scalar
instructions, nothing else.

Work decomposition (see correlation_fwd2.hip): workgroup = 8 waves = (tile column tx, tile row ty, half nh of the
N-tile columns); a wave owns BOTH column classes of its 4x4 tile of class positions, so the two floats of one
ds_read_b64 (class 0, class 1 of one full-resolution column pair) feed two MFMAs.

Staging: stages of FOUR channels (one K-step) in a ring of four, a barrier per K-step, the LDS-DMA of K-step j + 4
issued behind the barrier of K-step j: three barrier intervals between issue and use (profiles/r04_variants.txt: with
two stages of eight channels a wave waited for its own DMA 8-19 % of the sweep, with the ring 2-3 %).  Wave w stages
channel w & 3: waves 0..3 the f1 pieces 0, 1, 2 (whole 1-KiB pieces), waves 4..7 the half piece 3 and the f0 plane
(half a piece each, lanes 0..31).  The statement holds the sweep twice, once per kind: their wait counts differ.

Register map (128 registers, two workgroups per CU):
  v[ACC0 + 4t : +3]   accumulator tile t = (r * 3 + col) * 2 + c     (r: N-tile row 0..2, col: 0..2, c: column class)
  v[FB(p) + 0, +1]    A fragment (class 0, 1) of fragment buffer p (K-step parity)
  v[FB(p) + 2 + 2 * (r * 3 + col) + c]  B fragments
Run after changing anything here:  python tools/gen_corr_fwd_asm.py"""
import os
import sys

# Diagnostic knobs (environment; tools/fwd2_variants.sh builds such variants into unflow_amd/variants/ and the kernel
# reports them through unflow_build_flags()).  The shipped include is generated with none set.
#   FWD2_STAMPS=1  cycles a wave spends waiting for its DMA / at the barriers (tools/stamps_fwd2.py)
#   FWD2_NODMA=1   timing only: no LDS-DMA at all
#   FWD2_FULL=1    experiment: the FULL window per workgroup -- six N-tile rows, 36 accumulator tiles per
#                  wave, 256 registers, ONE workgroup per CU -- instead of two half-window workgroups
KNOBS = {k: os.environ.get("FWD2_" + k, "") for k in ("STAMPS", "NODMA", "FULL", "OUT", "PRIO")}
FULL = bool(KNOBS["FULL"])

# ---- LDS layout of a stage (floats), shared with correlation_fwd2.hip through the generated constants
ROWF = 56                 # one staged window row at full resolution (both column classes)
ROWS = 28 if FULL else 16 # staged window rows of a half-window workgroup
PLANE = ROWS * ROWF       # 896
PL = PLANE if FULL else PLANE + 32   # plane stride 928 (full window: 1568) = 32 mod 64: the two channels of a half-wave take disjoint halves of the 64 banks
assert PL % 64 == 32
F0PL = 8 * 16             # [row 8][x 16] of one channel of f0
F0S = F0PL + 8            # 136: the same for the A reads
KS = 4                    # channels per stage = K of the MFMA
F0OFF = KS * PL           # 3712: the f0 planes behind the f1 planes of a stage
STAGE = KS * (PL + F0S)   # 4256 floats
SB = STAGE * 4            # bytes
NSTAGE = 4                # ring of four: 68096 bytes

NROW, NCOL = (6 if FULL else 3), 3
NT = NROW * NCOL * 2      # 18 accumulator tiles
FRAG = 2 + 2 * NROW * NCOL  # 20 registers per fragment buffer
FB0 = 16
ACC0 = FB0 + 2 * FRAG     # 56
LAST = ACC0 + 4 * NT - 1
assert LAST == (235 if FULL else 127)
# LDS-DMA instructions per K-step of the two kinds of waves: (operand letter, lanes that take part)
PIECES = ([[("A", 64), ("B", 64), ("C", 64), ("D", 64)], [("A", 64), ("B", 64), ("C", 8), ("D", 32)]] if FULL else
          [[("A", 64), ("B", 64), ("C", 64)], [("A", 32), ("B", 32)]])


def fb(p):
    return FB0 + FRAG * p


def acc(r, col, c):
    return ACC0 + 4 * ((r * NCOL + col) * 2 + c)


def lit(lines):
    return " \\\n".join('  "%s\\n\\t"' % l for l in lines)


def mfma(p, r, col, c):
    a = acc(r, col, c)
    return "v_mfma_f32_16x16x4_f32 v[%d:%d], v%d, v%d, v[%d:%d]" % (a, a + 3, fb(p) + c, fb(p) + 2 + 2 * (r * NCOL + col) + c, a, a + 3)


def base_and_offset(name, stage, off):
    """ds_read offsets are 16 bits: the full-window form (27 KB stages) reads stages 2, 3 from a second base register."""
    if FULL and stage >= 2:
        name, stage = name + "2", stage - 2
    off += stage * SB
    assert off < 65536
    return "%%[%s] offset:%d" % (name, off)


def read_a(p, stage):
    return "ds_read_b64 v[%d:%d], %s" % (fb(p), fb(p) + 1, base_and_offset("aaddr", stage, 0))


def read_b(p, stage, r, col):
    reg = fb(p) + 2 + 2 * (r * NCOL + col)
    return "ds_read_b64 v[%d:%d], %s" % (reg, reg + 1, base_and_offset("baddr", stage, r * 4 * ROWF * 4 + col * 8 * 4))


def group(p, r, nstage, label, reads=True):
    L = ["s_bitcmp0_b32 %%[mask], %d" % r, "s_cbranch_scc1 %df" % label]
    k = 0
    for col in range(NCOL):
        for c in range(2):
            L.append(mfma(p, r, col, c))
            if k < NCOL and reads:
                L.append(read_b(1 - p, nstage, r, k))
            k += 1
    L.append("%d:" % label)
    return L


def set_m0(which, stage):
    return ["s_add_u32 m0, %%[m%s], %d" % (which, stage * SB)]


def load(which):
    if KNOBS["NODMA"]:
        return []
    return ["buffer_load_dwordx4 %%[v%s], %%[r%s], %%[soff] offen lds" % (which, which)]


def narrow(lanes):
    return {64: [], 32: ["s_mov_b32 exec_hi, 0"], 8: ["s_mov_b32 exec_hi, 0", "s_mov_b32 exec_lo, 0xff"]}[lanes]


def widen(lanes):
    return {64: [], 32: ["s_mov_b32 exec_hi, -1"], 8: ["s_mov_b32 exec_lo, -1", "s_mov_b32 exec_hi, -1"]}[lanes]


def dma(which, stage, lanes):
    # (prologue form) a write of M0 needs one wait state before the LDS-DMA that uses it
    return set_m0(which, stage) + (narrow(lanes) or ["s_nop 0"]) + load(which) + widen(lanes)


def kstep(j, base, dma, wait, spec, reads=True):
    """K-step j of a group of four (stage j, fragment buffer j & 1).  Behind its barrier every wave has read stage j for
    the last time and K-step j + 1 has landed: its fragments are read from stage j + 1 between the MFMAs, and (dma) the
    LDS-DMA of K-step j + 4 goes into stage j -- one piece behind each of the first MFMA groups, its M0 written in
    front of the group (the group's branch test is the wait state a skipped group leaves).  spec: the wave kind's
    pieces (operand letter, lanes): whole pieces, or half pieces with EXEC narrowed around each."""
    p, nstage = j & 1, (j + 1) & 3
    if KNOBS["STAMPS"]:
        L = ["s_memtime s[90:91]", "s_waitcnt vmcnt(%d) lgkmcnt(0)" % wait, "s_memtime s[92:93]", "s_waitcnt lgkmcnt(0)",
             "s_sub_u32 s94, s92, s90", "s_add_u32 %[accv], %[accv], s94", "s_barrier", "s_memtime s[90:91]", "s_waitcnt lgkmcnt(0)",
             "s_sub_u32 s94, s90, s92", "s_add_u32 %[accb], %[accb], s94"]
    else:
        L = ["s_waitcnt vmcnt(%d) lgkmcnt(0)" % wait, "s_barrier"]
    if KNOBS["PRIO"]:      # experiment: an issue priority per K-step of the group of four, e.g. FWD2_PRIO=3210
        L.append("s_setprio %d" % int(KNOBS["PRIO"][j]))
    if reads:
        L.append(read_a(1 - p, nstage))
    for r in range(NROW):
        issue = dma and r < len(spec)
        if issue:
            L += set_m0(spec[r][0], j)
        L += group(p, r, nstage, base + r, reads)
        if issue:
            L += narrow(spec[r][1]) + load(spec[r][0]) + widen(spec[r][1])
    if dma:
        L.append("s_add_u32 %[soff], %[soff], %[kstep]")
    return L


def sweep(spec, lb):
    """The sweep for one kind of wave (spec: its DMA pieces per K-step; waves 0..3 / waves 4..7)."""
    npk = len(spec)
    L = ["s_mov_b32 %[soff], 0"]
    for j in range(4):
        for which, lanes in spec:
            L += dma(which, j, lanes)
        L += ["s_add_u32 %[soff], %[soff], %[kstep]"]
    L += ["s_sub_u32 %[cnt], %[ngroups], 1", "s_waitcnt vmcnt(%d)" % (3 * npk), "s_barrier"]
    L += [read_a(0, 0)] + [read_b(0, 0, r, col) for r in range(NROW) for col in range(NCOL)]
    L += ["s_cmp_eq_u32 %[cnt], 0", "s_cbranch_scc1 %df" % (lb + 91), "%d:" % (lb + 90)]
    for j in range(4):
        L += kstep(j, lb + 20 + 8 * j, True, 2 * npk, spec)
    L += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 %db" % (lb + 90), "%d:" % (lb + 91)]
    for j, wait in zip(range(4), (2 * npk, npk, 0, 0)):
        L += kstep(j, lb + 52 + 8 * j, False, wait, spec, reads=j < 3)
    return L


def main_loop():
    L = ["v_mov_b32 v%d, 0" % r for r in range(ACC0, LAST + 1)]
    L += ["s_cmp_eq_u32 %[ehi], 0", "s_cbranch_scc1 300f"] + sweep(PIECES[0], 100) + ["s_branch 301f", "300:"] + sweep(PIECES[1], 200) + ["301:"]
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15"]
    return L


out = ["// GENERATED by tools/gen_corr_fwd_asm.py -- do not edit.  Main loop of corr_fwd2_kernel (correlation_fwd2.hip).",
       "// Register map: fragment buffers v[%d:%d] / v[%d:%d], accumulators v[%d:%d] (tile (r, col, c) at %d + 4 * ((r * 3 + col) * 2 + c))." % (
           fb(0), fb(0) + FRAG - 1, fb(1), fb(1) + FRAG - 1, ACC0, LAST, ACC0)]
if FULL:
    out.append("#define UNFLOW_FWD2_FULLWIN 1")
for name, val in (("NROW", NROW), ("ROWF", ROWF), ("ROWS", ROWS), ("PL", PL), ("F0S", F0S), ("F0OFF", F0OFF), ("STAGE", STAGE), ("NSTAGE", NSTAGE), ("NT", NT)):
    out.append("#define UNFLOW_FWD2_%s %d" % (name, val))
out.append("#define UNFLOW_FWD2_FRAG_CLOBBERS " + ", ".join('"v%d"' % r for r in range(FB0, ACC0)))
out.append("// the accumulators leave the statement as its outputs, bound to their registers: no move, and the compiler owns them afterwards")
out.append("#define UNFLOW_FWD2_ACC_OUTPUTS(A) " + ", ".join('"=&{v[%d:%d]}"(A[%d])' % (ACC0 + 4 * t, ACC0 + 4 * t + 3, t) for t in range(NT)))
out.append("// operands: [aaddr] [baddr] [vA] [vB] [vC] v; [rA] [rB] [rC] s (128 bit); [mask] [mA] [mB] [mC] [ehi] [kstep] [ngroups] s; [soff] [cnt] =&s;")
out.append("// clobbers the fragment registers, m0, scc, memory; EXEC is all ones on entry and exit")
if KNOBS["STAMPS"]:
    out.append('#define UNFLOW_FWD2_STAMP_OPERANDS , [accv] "+s"(accv), [accb] "+s"(accb)')
    out.append('#define UNFLOW_FWD2_STAMP_CLOBBERS , "s90", "s91", "s92", "s93", "s94"')
else:
    out.append("#define UNFLOW_FWD2_STAMP_OPERANDS")
    out.append("#define UNFLOW_FWD2_STAMP_CLOBBERS")
out.append("#define UNFLOW_FWD2_MAIN \\\n%s" % lit(main_loop()))
path = KNOBS["OUT"] or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unflow_amd", "csrc", "corr_fwd2_asm.inc")
flags = " ".join("FWD2_%s=%s" % (k, v) for k, v in sorted(KNOBS.items()) if v and k != "OUT")
if flags:
    out.insert(1, '#define UNFLOW_FWD2_GENERATOR_KNOBS "%s"' % flags)
open(path, "w").write("\n".join(out) + "\n")
print("wrote", path)
