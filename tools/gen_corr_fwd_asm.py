"""Writes unflow_amd/csrc/corr_fwd2_asm.inc: the main loop of corr_fwd2_kernel (correlation_fwd2.hip) -- the whole
channel sweep of the FlowNetC correlation forward, from the first LDS-DMA to the last MFMA -- as ONE inline-assembly
statement on fixed vector registers, plus the layout constants the C++ around it shares with it.

Why one statement: every instruction the compiler puts between two MFMA groups costs the matrix pipe issue time
(profiles/ubench_mfma_lds.txt), and what it put there in corr_mfma_fwd_half_kernel was 2.1 scalar + 1.2 vector
instructions per MFMA.  Here a chunk of 8 channels is 36 MFMAs, 20 ds_read_b64, 5 LDS-DMA and ~30 scalar
instructions, nothing else.

Work decomposition (see correlation_fwd2.hip): workgroup = 8 waves = (tile column tx, tile row ty, half nh of the
N-tile columns); a wave owns BOTH column classes of its 4x4 tile of class positions, so the two floats of one
ds_read_b64 (class 0, class 1 of one full-resolution column pair) feed two MFMAs.

Register map (128 registers, two workgroups per CU):
  v[ACC0 + 4t : +3]   accumulator tile t = (r * 3 + col) * 2 + c     (r: N-tile row 0..2, col: 0..2, c: column class)
  v[FB(p) + 0, +1]    A fragment (class 0, 1) of fragment buffer p (K-step parity)
  v[FB(p) + 2 + 2 * (r * 3 + col) + c]  B fragments
Run after changing anything here:  python tools/gen_corr_fwd_asm.py"""
import os
import sys

# Diagnostic / schedule knobs (environment; tools/fwd2_variants.sh builds such variants into unflow_amd/variants/ and the
# kernel reports them through unflow_build_flags()).  The shipped include is generated with none set.
KNOBS = {k: os.environ.get("FWD2_" + k, "") for k in ("NODMA", "NOREADS", "NOBARRIER", "NOSKIP", "DMAFIRST", "PRIO", "STAMPS", "OUT")}

# ---- LDS layout of a stage (floats), shared with correlation_fwd2.hip through the generated constants
ROWF = 56                 # one staged window row at full resolution (both column classes)
ROWS = 16                 # staged window rows of a half-window workgroup
PLANE = ROWS * ROWF       # 896
PL = PLANE + 32           # plane stride 928 = 32 mod 64: the two channels of a half-wave take disjoint halves of the 64 banks
F0PL = 8 * 16             # [row 8][x 16] of one channel of f0
F0S = F0PL + 8            # 136: the same for the A reads
KC = 8                    # channels per chunk = waves
F0OFF = KC * PL           # 7424
STAGE = F0OFF + KC * F0S  # 8512 floats
SB = STAGE * 4            # bytes

NROW, NCOL = 3, 3
NT = NROW * NCOL * 2      # 18 accumulator tiles
FRAG = 2 + 2 * NROW * NCOL  # 20 registers per fragment buffer
FB0 = 16
ACC0 = FB0 + 2 * FRAG     # 56
LAST = ACC0 + 4 * NT - 1
assert LAST == 127


def fb(p):
    return FB0 + FRAG * p


def acc(r, col, c):
    return ACC0 + 4 * ((r * NCOL + col) * 2 + c)


def lit(lines):
    return " \\\n".join('  "%s\\n\\t"' % l for l in lines)


def read_a(p, stage, kk, first=False):
    if KNOBS["NOREADS"] and not first:
        return "s_nop 0"
    return "ds_read_b64 v[%d:%d], %%[aaddr] offset:%d" % (fb(p), fb(p) + 1, stage * SB + kk * 4 * F0S * 4)


def read_b(p, stage, kk, r, col, first=False):
    if KNOBS["NOREADS"] and not first:
        return "s_nop 0"
    reg = fb(p) + 2 + 2 * (r * NCOL + col)
    off = stage * SB + kk * 4 * PL * 4 + r * 4 * ROWF * 4 + col * 8 * 4
    assert off < 65536
    return "ds_read_b64 v[%d:%d], %%[baddr] offset:%d" % (reg, reg + 1, off)


def mfma(p, r, col, c):
    a = acc(r, col, c)
    return "v_mfma_f32_16x16x4_f32 v[%d:%d], v%d, v%d, v[%d:%d]" % (a, a + 3, fb(p) + c, fb(p) + 2 + 2 * (r * NCOL + col) + c, a, a + 3)


def group(p, r, nxt, label):
    """The 6 MFMAs of N-tile row r on fragment buffer p, with the reads of the same row for the NEXT K-step
    (nxt = (stage, kk), into buffer 1 - p) behind the first three; skipped as a whole when the row lies in the padding."""
    L = [] if KNOBS["NOSKIP"] else ["s_bitcmp0_b32 %%[mask], %d" % r, "s_cbranch_scc1 %df" % label]
    k = 0
    for col in range(NCOL):
        for c in range(2):
            L.append(mfma(p, r, col, c))
            if k < NCOL:
                L.append(read_b(1 - p, nxt[0], nxt[1], r, k))
            k += 1
    L.append("%d:" % label)
    return L


def dma_f1(j, stage):
    if KNOBS["NODMA"] == "1" or (KNOBS["NODMA"] == "half" and j in (1, 3)):
        return []
    return ["s_add_u32 m0, %%[m1], %d" % (stage * SB + j * 1024), "s_nop 0",
            "buffer_load_dwordx4 %%[v%d], %%[r1], %%[soff] offen lds" % j]


def dma_f0(stage):
    if KNOBS["NODMA"] == "1":
        return []
    return ["s_add_u32 m0, %%[m0f], %d" % (stage * SB), "s_nop 0",
            "buffer_load_dwordx4 %[v4], %[r0], %[soff] offen lds"]


def dma_all(stage):
    return (dma_f1(0, stage) + dma_f1(1, stage) + dma_f1(2, stage) + ["s_mov_b32 exec_hi, 0"] + dma_f1(3, stage) + dma_f0(stage) +
            ["s_mov_b32 exec_hi, -1"])


def chunk(s, base):
    """One chunk of 8 channels in stage s.  K-step 0 runs on buffer 0 and reads K-step 1's fragments (stage s, channels
    4..7) into buffer 1.  The chunk's barrier stands in front of K-step 1: behind it every wave has read this stage for
    the last time and the next chunk has landed in the other stage, so K-step 1 reads the next chunk's first fragments
    from there and issues the LDS-DMA of the chunk after next into this stage."""
    L = ["s_waitcnt lgkmcnt(0)", read_a(1, s, 1)]
    for r in range(NROW):
        L += group(0, r, (s, 1), base + r)
    if KNOBS["STAMPS"]:     # cycles a wave spends waiting for its DMA / at the barrier (s[90:94] are named as clobbers)
        L += ["s_memtime s[90:91]", "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_memtime s[92:93]", "s_waitcnt lgkmcnt(0)",
              "s_sub_u32 s94, s92, s90", "s_add_u32 %[accv], %[accv], s94", "s_barrier", "s_memtime s[90:91]", "s_waitcnt lgkmcnt(0)",
              "s_sub_u32 s94, s90, s92", "s_add_u32 %[accb], %[accb], s94"]
    else:
        L += ["s_waitcnt vmcnt(0) lgkmcnt(0)"] + ([] if KNOBS["NOBARRIER"] else ["s_barrier"])
    L += [read_a(0, 1 - s, 0),
          "s_cmp_lg_u32 %[cnt], 1", "s_cbranch_scc0 %df" % (base + 9)]
    # with DMA (a chunk after next exists)
    if KNOBS["DMAFIRST"]:
        L += dma_all(s) + ["s_add_u32 %[soff], %[soff], %[cstep]"]
        for r in range(NROW):
            L += group(1, r, (1 - s, 0), base + 3 + r)
    else:
        L += group(1, 0, (1 - s, 0), base + 3) + dma_f1(0, s) + dma_f1(1, s)
        L += group(1, 1, (1 - s, 0), base + 4) + dma_f1(2, s)
        L += group(1, 2, (1 - s, 0), base + 5) + ["s_mov_b32 exec_hi, 0"] + dma_f1(3, s) + dma_f0(s) + ["s_mov_b32 exec_hi, -1",
                                                                                                           "s_add_u32 %[soff], %[soff], %[cstep]"]
    L += ["s_branch %df" % (base + 10), "%d:" % (base + 9)]
    # without (the last two chunks)
    for r in range(NROW):
        L += group(1, r, (1 - s, 0), base + 6 + r)
    L += ["%d:" % (base + 10)]
    return L


def main_loop():
    L = ["v_mov_b32 v%d, 0" % r for r in range(ACC0, LAST + 1)]
    L += ["s_mov_b32 %[soff], 0"] + dma_all(0) + ["s_add_u32 %[soff], %[soff], %[cstep]"] + dma_all(1)
    L += ["s_add_u32 %[soff], %[soff], %[cstep]", "s_mov_b32 %[cnt], %[npairs]", "s_waitcnt vmcnt(5)", "s_barrier"]
    L += [read_a(0, 0, 0, True)] + [read_b(0, 0, 0, r, col, True) for r in range(NROW) for col in range(NCOL)]
    L += ["90:"] + chunk(0, 20) + chunk(1, 40)
    L += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 90b"]
    # MFMA results -> any other reader: 11 wait states for an 8-pass MFMA; the compiler knows nothing of what is in flight
    L += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 15"]
    return L


out = ["// GENERATED by tools/gen_corr_fwd_asm.py -- do not edit.  Main loop of corr_fwd2_kernel (correlation_fwd2.hip).",
       "// Register map: fragment buffers v[%d:%d] / v[%d:%d], accumulators v[%d:%d] (tile (r, col, c) at %d + 4 * ((r * 3 + col) * 2 + c))." % (
           fb(0), fb(0) + FRAG - 1, fb(1), fb(1) + FRAG - 1, ACC0, LAST, ACC0)]
for name, val in (("ROWF", ROWF), ("ROWS", ROWS), ("PL", PL), ("F0S", F0S), ("F0OFF", F0OFF), ("STAGE", STAGE), ("NT", NT)):
    out.append("#define UNFLOW_FWD2_%s %d" % (name, val))
out.append("#define UNFLOW_FWD2_FRAG_CLOBBERS " + ", ".join('"v%d"' % r for r in range(FB0, ACC0)))
out.append("// the accumulators leave the statement as its outputs, bound to their registers: no move, and the compiler owns them afterwards")
out.append("#define UNFLOW_FWD2_ACC_OUTPUTS(A) " + ", ".join('"=&{v[%d:%d]}"(A[%d])' % (ACC0 + 4 * t, ACC0 + 4 * t + 3, t) for t in range(NT)))
out.append("// operands: [aaddr] [baddr] [v0..v4] v; [r0] [r1] s (128 bit); [mask] [m1] [m0f] [cstep] [npairs] s; [soff] [cnt] =&s;")
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
