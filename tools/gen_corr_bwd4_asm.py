"""Writes unflow_amd/csrc/corr_bwd4_asm.inc: the hand-written instruction blocks of corr_mfma_bwd4_kernel
(correlation_mfma.hip) as string-literal macros.  The kernel keeps its accumulators and fragment buffers in FIXED
vector registers that only these blocks touch (the compiler is told through the clobber lists and stays below
them); everything else of the kernel is C++.  Run after changing the register map:  python tools/gen_corr_bwd4_asm.py"""
import os

# B4_WAVES=16 (experiment, round 4): the same blocks for a SIXTEEN-wave workgroup -- a wave holds half of the channels of its
# tiles (4 M-tiles: 48 accumulator registers, 128 VGPRs in all, four waves per SIMD) -> corr_bwd16_asm.inc, UNFLOW_BWD16_*
W16 = os.environ.get("B4_WAVES", "8") == "16"
PFX = "UNFLOW_BWD16_" if W16 else "UNFLOW_BWD4_"
RING, MTS = 3, (4 if W16 else 8)
ACS = 60                      # A image: floats per channel row (correlation_mfma.hip)
if W16:
    AB = [64, 68]             # A fragment buffers: 4 registers each
    BB = [72, 75]             # B fragment buffers: 3 registers each
    ACC0 = 80                 # accumulators: v[80:127]
else:
    AB = [136, 144]           # A fragment buffers: 8 registers each
    BB = [152, 155]           # B fragment buffers: 3 registers each
    ACC0 = 160                # accumulators: v[160:255], tile (slot, m) at 160 + (slot * 8 + m) * 4
FIRST, LAST = AB[0], ACC0 + RING * MTS * 4 - 1
assert LAST == (127 if W16 else 255)


def acc(sl, m):
    return ACC0 + (sl * MTS + m) * 4


def lit(lines):
    return " \\\n".join('  "%s\\n\\t"' % l for l in lines)


def mfmas(p, sl):
    out = []
    for m in range(MTS):
        a = acc(sl, m)
        out.append("v_mfma_f32_16x16x4_f32 v[%d:%d], v%d, v%d, v[%d:%d]" % (a, a + 3, AB[p] + m, BB[p] + sl, a, a + 3))
    return out


def reads(n):
    out = []
    for m in range(MTS):
        out.append("ds_read_b32 v%d, %%[a_rd] offset:%d+%%[ia]" % (AB[n] + m, m * 16 * ACS * 4))
    for sl in range(RING):
        out.append("ds_read_b32 v%d, %%[b%d] offset:%%[ib]" % (BB[n] + sl, sl))
    return out


AG_BYTES = 8 * ACS * 4                      # a group of 8 channel rows of the A image
NW = 16 if W16 else 8
A_STEP = (NW // 2) * AG_BYTES                # LDS distance of a wave's consecutive A pieces (pieces p, p + NW: NW / 2 groups on)
T_STEP = NW * 256 * 4                        # ... of its consecutive top-image pieces (NW pieces of 1 KiB on)
TV = 78 if W16 else 158                      # scratch registers (two) for a top-image piece's offset


def dma_a(j):
    # one 1-KiB piece of the A image by LDS-DMA: M0 = LDS destination, row in the scalar offset (always inside the image)
    return ["s_add_u32 m0, %%[a_m0], %d" % (j * A_STEP), "s_nop 0",
            "buffer_load_dwordx4 %%[av%d], %%[rin], %%[asoff] offen lds" % j]


def dma_t(j):
    # one piece of the top image: per-lane base (out of range for cells that must be zero) + the row's byte offset
    tv = TV + (j & 1)      # (two scratch registers in turn: the next piece's offset is not written into the register of a load just issued)
    return ["v_add_u32 v%d, %%[twoff], %%[tb%d]" % (tv, j), "s_add_u32 m0, %%[t_m0], %d" % (j * T_STEP), "s_nop 0",
            "buffer_load_dwordx4 v%d, %%[rtop], 0 offen lds" % tv]


SOFTBAR = os.environ.get("B4_SOFTBAR", "1") != "0" and not W16   # the row barrier as an LDS arrival counter (kstep5_row)
PRIO = os.environ.get("B4_PRIO", "1") != "0" and not W16   # balanced issue priorities of the two waves of a SIMD
SPREAD = os.environ.get("B4_SPREAD", "1") != "0"   # the next K-step's fragment reads one per MFMA gap (0: in one clump)


def kstep(p, dma=None):
    """dma None: plain; 'A': K-step 0 of a row (the four A pieces and top pieces 0, 1 of the NEXT row's stage);
    'T': K-step 1 (top pieces 2, 3).  [fetch] = 0 skips them (last row of the last item).
    The 11 fragment reads of the next K-step go ONE PER MFMA GAP into slot 0's group (the eight A reads) and slot 1's
    (the three B reads): a wave that runs alone on its SIMD -- its partner waits at the row barrier for about half
    of every row -- hides at most ~5 single-issue instructions behind an MFMA, and a clump of 11 LDS instructions
    behind the first group left the matrix pipe idle.  A dead slot's group is skipped as a whole; its share of the
    reads is then issued in the skip path."""
    n = 1 - p
    L = []
    if PRIO:
        # issue priority by progress: the two waves of a SIMD would otherwise be arbitrated by age, the older one running
        # ahead and then waiting at the row barrier while the younger one works off its rest alone (stamps: a quarter of
        # every row).  [pa] / [pb] fall along the row, the upper four waves' (bit 8 of [live]) one K-step later: the wave
        # that is behind wins, at equal progress they take turns.
        L += ["s_setprio %[pa]", "s_bitcmp0_b32 %[live], 8", "s_cbranch_scc1 10f", "s_setprio %[pb]", "10:"]
    L += ["s_waitcnt lgkmcnt(0)"]
    rd = reads(n)
    share = {0: rd[:MTS], 1: rd[MTS:], 2: []} if SPREAD else {0: [], 1: [], 2: []}
    for sl in range(RING):
        mm = mfmas(p, sl)
        L += ["s_bitcmp0_b32 %%[live], %d" % sl, "s_cbranch_scc1 %df" % (sl + 1)]
        if share[sl]:
            for i, m in enumerate(mm):
                L.append(m)
                if i < len(share[sl]):
                    L.append(share[sl][i])
            L += ["s_branch %df" % (sl + 4), "%d:" % (sl + 1)] + share[sl] + ["%d:" % (sl + 4)]
        else:
            L += mm + ["%d:" % (sl + 1)]
        if sl == 0 and not SPREAD:
            L += rd
        # (every group is guarded on its own and leaves EXEC complete: the MFMAs that follow need all lanes)
        if dma == "A" and W16:   # sixteen waves: two A pieces and two top pieces per wave and row
            if sl == 0:
                L += ["s_cmp_eq_u32 %[fetch], 0", "s_cbranch_scc1 7f", "s_mov_b64 vcc, exec", "s_and_b64 exec, exec, %[amask]"] + dma_a(0) + dma_a(1) + ["s_mov_b64 exec, vcc", "7:"]
            elif sl == 1:
                L += ["s_cmp_eq_u32 %[fetch], 0", "s_cbranch_scc1 8f"] + dma_t(0) + ["8:"]
        elif dma == "T" and W16:
            if sl == 1:
                L += ["s_cmp_eq_u32 %[t3], 0", "s_cbranch_scc1 8f"] + dma_t(1) + ["8:"]
        elif dma == "A":
            if sl == 0:       # lanes that take part in an A piece: the same mask for the wave's four pieces
                L += ["s_cmp_eq_u32 %[fetch], 0", "s_cbranch_scc1 7f", "s_mov_b64 vcc, exec", "s_and_b64 exec, exec, %[amask]"] + dma_a(0) + dma_a(1) + ["s_mov_b64 exec, vcc", "7:"]
            elif sl == 1:
                L += ["s_cmp_eq_u32 %[fetch], 0", "s_cbranch_scc1 8f", "s_mov_b64 vcc, exec", "s_and_b64 exec, exec, %[amask]"] + dma_a(2) + dma_a(3) + ["s_mov_b64 exec, vcc", "8:"]
            else:
                L += ["s_cmp_eq_u32 %[fetch], 0", "s_cbranch_scc1 9f"] + dma_t(0) + dma_t(1) + ["9:"]
        elif dma == "T":
            if sl == 1:
                L += ["s_cmp_eq_u32 %[fetch], 0", "s_cbranch_scc1 8f"] + dma_t(2) + ["8:"]
            elif sl == 2:
                L += ["s_cmp_eq_u32 %[t3], 0", "s_cbranch_scc1 9f"] + dma_t(3) + ["9:"]
    return L


def row_step():
    """The sweep state one row further (see the kernel): per ring slot the row j of the tile's life (wraps 23 -> 0 into the
    next tile of the slot: u += 3, the slot's live bit follows), the lanes' top-image row address (one image row up, or
    23 down at the wrap, plus the change of stage) and the address the fragments are read at (the all-zero cell for lanes
    outside the 21 rows in the three rows at either end of a tile's life)."""
    L = ["s_add_u32 %[zero], %[zero], %[dstage]", "v_add_u32 %[aaddr], %[dstage], %[aaddr]"]
    for sl in range(RING):
        a, b, c, d, e, f = (10 * (sl + 1) + k for k in range(6))
        L += ["s_add_u32 %%[j%d], %%[j%d], 1" % (sl, sl), "s_cmp_lt_u32 %%[j%d], 24" % sl, "s_cbranch_scc1 %df" % a,
              "s_mov_b32 %%[j%d], 0" % sl, "s_add_i32 %%[u%d], %%[u%d], 3" % (sl, sl), "s_bitset0_b32 %%[live], %d" % sl,
              "s_cmp_lt_i32 %%[u%d], %%[tmax]" % sl, "s_cbranch_scc0 %df" % b, "s_bitset1_b32 %%[live], %d" % sl, "%d:" % b,
              "v_add_u32 %%[br%d], %%[dwrap], %%[br%d]" % (sl, sl), "s_branch %df" % c,
              "%d:" % a, "v_add_u32 %%[br%d], %%[dstep], %%[br%d]" % (sl, sl), "%d:" % c,
              "s_cmp_lt_u32 %%[j%d], 3" % sl, "s_cbranch_scc1 %df" % d, "s_cmp_gt_u32 %%[j%d], 20" % sl, "s_cbranch_scc1 %df" % e,
              "v_mov_b32 %%[bu%d], %%[br%d]" % (sl, sl), "s_branch %df" % f,
              "%d:" % d, "v_cmp_ge_u32 vcc, %%[j%d], %%[oyl]" % sl, "v_mov_b32 %%[bu%d], %%[zero]" % sl, "v_cndmask_b32 %%[bu%d], %%[bu%d], %%[br%d], vcc" % (sl, sl, sl), "s_branch %df" % f,
              "%d:" % e, "s_sub_u32 %%[tmp], %%[j%d], 20" % sl, "v_cmp_le_u32 vcc, %[tmp], %[oyl]", "v_mov_b32 %%[bu%d], %%[zero]" % sl, "v_cndmask_b32 %%[bu%d], %%[bu%d], %%[br%d], vcc" % (sl, sl, sl),
              "%d:" % f]
    return L


def row_chunks(lb):
    """row_step() as straight-line pieces that may stand between any two MFMAs (only the rare wrap of a slot -- one row
    in 24 -- is a branch, and that piece is scalar; lb: its local labels).  Besides [bu*] (the address K-steps 1..5 of
    the next row read the top image at) it leaves [bf*], the same with the lanes that have no valid dx' in K-step 0 on
    the all-zero cell ([vfirst], a lane mask): what the first fragment reads of the next row take.  v{TV} holds the zero
    cell's address."""
    ch = [["s_add_u32 %[zero], %[zero], %[dstage]", "v_add_u32 %[aaddr], %[dstage], %[aaddr]", "v_mov_b32 v%d, %%[zero]" % TV]]
    for sl in range(RING):
        ch.append(["s_add_u32 %%[j%d], %%[j%d], 1" % (sl, sl), "s_cmp_lt_u32 %%[j%d], 24" % sl,
                   "s_cselect_b32 %[tmp], %[dstep], %[dwrap]", "v_add_u32 %%[br%d], %%[tmp], %%[br%d]" % (sl, sl)])
        ch.append(["s_cmp_lt_u32 %%[j%d], 24" % sl, "s_cbranch_scc1 %df" % (lb + sl),
                   "s_mov_b32 %%[j%d], 0" % sl, "s_add_i32 %%[u%d], %%[u%d], 3" % (sl, sl), "s_bitset0_b32 %%[lvn], %d" % sl,
                   "s_cmp_lt_i32 %%[u%d], %%[tmax]" % sl, "s_cbranch_scc0 %df" % (lb + sl), "s_bitset1_b32 %%[lvn], %d" % sl,
                   "%d:" % (lb + sl)])
        # rows j of a tile's life whose top-image row 20 + oyl - j lies in 0..20: 0 <= j - oyl <= 20
        ch.append(["v_sub_u32 %%[bf%d], %%[j%d], %%[oyl]" % (sl, sl), "v_cmp_ge_u32 vcc, 20, %%[bf%d]" % sl,
                   "v_cndmask_b32 %%[bu%d], v%d, %%[br%d], vcc" % (sl, TV, sl)])
        ch.append(["s_and_b64 vcc, vcc, %[vfirst]", "s_nop 0", "v_cndmask_b32 %%[bf%d], v%d, %%[br%d], vcc" % (sl, TV, sl)])
    return ch


def kstep5_row():
    """K-step 5 behind the row barrier WITH the sweep state's step to the next row between its MFMAs (slots 0, 1) and the
    next row's first fragment reads between those of slot 2: executed as a block of its own behind the barrier, the
    state's ~40 instructions met an empty matrix pipe in all eight waves at once (3.5 % of the kernel).  [live]: this
    row's; the pieces a dead slot's group would have carried are issued in its skip path."""
    p, n = 1, 0
    L = []
    if PRIO:
        L += ["s_setprio %[pa]", "s_bitcmp0_b32 %[live], 8", "s_cbranch_scc1 10f", "s_setprio %[pb]", "10:"]
    L += ["s_waitcnt lgkmcnt(0)"]
    rd = ["ds_read_b32 v%d, %%[aaddr] offset:%d" % (AB[n] + m, m * 16 * ACS * 4) for m in range(MTS)]
    rd += ["ds_read_b32 v%d, %%[bf%d] offset:%%[ib]" % (BB[n] + sl, sl) for sl in range(RING)]
    rdp = [rd[0:2], rd[2:4], rd[4:6], rd[6:7], rd[7:8], rd[8:9], rd[9:10], rd[10:11]]
    share = [{0: c[:7], 1: c[7:], 2: rdp} for c in (row_chunks(40), row_chunks(50))]   # [0]: between MFMAs, [1]: skip path
    for sl in range(RING):
        if SOFTBAR and sl == 1:
            # The row barrier is an arrival counter in LDS ([bar]: its address; every wave adds 1 behind K-step 4, when
            # its DMA pieces of the next row have landed and its last reads of this row's stage are on their way -- LDS
            # serves a wave's operations in order).  Nothing in front of this point needs the other waves: the counter is
            # read here, behind slot 0's MFMAs, and looked at in front of slot 2's, whose gaps carry the first reads of
            # the next row's stage -- the s_barrier this replaces cost 4.9 % of the kernel with every wave standing still.
            L += ["ds_read_b32 v%d, %%[bar]" % (TV + 1)]
        if SOFTBAR and sl == 2:
            L += ["s_waitcnt lgkmcnt(0)", "v_readfirstlane_b32 %%[tmp], v%d" % (TV + 1), "s_cmp_ge_u32 %[tmp], %[target]", "s_cbranch_scc1 61f",
                  "60:", "ds_read_b32 v%d, %%[bar]" % (TV + 1), "s_waitcnt lgkmcnt(0)", "v_readfirstlane_b32 %%[tmp], v%d" % (TV + 1),
                  "s_cmp_lt_u32 %[tmp], %[target]", "s_cbranch_scc1 60b", "61:"]
        L += ["s_bitcmp0_b32 %%[live], %d" % sl, "s_cbranch_scc1 %df" % (sl + 1)]
        for i, m in enumerate(mfmas(p, sl)):
            L.append(m)
            if i < len(share[0][sl]):
                L += share[0][sl][i]
        L += ["s_branch %df" % (sl + 4), "%d:" % (sl + 1)]
        for c in share[1][sl]:
            L += c
        L += ["%d:" % (sl + 4)]
    return L


def dump(sl):
    # XDL write -> LDS data read of the same register needs software wait states (the compiler inserts them for its
    # own code; 8-pass MFMA: 11): 24 here, once per finished tile
    L = ["s_nop 15", "s_nop 7"] * int(os.environ.get("B4_NOPS", "1"))
    for m in (range(MTS - 1, -1, -1) if os.environ.get("B4_REV") else range(MTS)):
        for j in range(4):
            L.append("ds_write_b32 %%[addr], v%d offset:%d" % (acc(sl, m) + j, (m * 16 + j) * 4 * 16 * 4))
    L.append("s_waitcnt lgkmcnt(0)")
    for m in range(MTS):
        for j in range(4):
            L.append("v_mov_b32 v%d, 0" % (acc(sl, m) + j))
    L.append("s_nop 1")
    return L


out = ["// GENERATED by tools/gen_corr_bwd4_asm.py -- do not edit.  Instruction blocks of corr_mfma_bwd4_kernel.",
       "// Register map: A fragments v[%d:%d] / v[%d:%d], B fragments v[%d:%d] / v[%d:%d], accumulators v[%d:%d]." % (
           AB[0], AB[0] + 7, AB[1], AB[1] + 7, BB[0], BB[0] + 2, BB[1], BB[1] + 2, ACC0, LAST),
       "#define " + PFX + "FIRST_FIXED_VGPR %d" % FIRST,
       "#define " + PFX + "CLOBBERS " + ", ".join('"v%d"' % r for r in range(FIRST, LAST + 1)),
       "// one K-step: wait for its fragments, 8 MFMAs per live ring slot (parity P buffers), the next K-step's 11 fragment",
       "// reads (other buffers) behind slot 0.  Operands: [live] s, [a_rd] [b0] [b1] [b2] v, [ia] [ib] i"]
for p in (0, 1):
    out.append("#define " + PFX + "KSTEP_P%d \\\n%s" % (p, lit(kstep(p))))
out.append("// K-step 0 of a row + the LDS-DMA of the next row's A image and top pieces 0, 1: [fetch] [a_m0] [asoff] [t_m0] [twoff] s,")
out.append("// [amask] s (64 bit), [rin] [rtop] s (128 bit), [av0..3] [tb0] [tb1] v; clobbers vcc, m0")
out.append("#define " + PFX + "KSTEP0_DMA \\\n%s" % lit(kstep(0, "A")))
out.append("// K-step 1 + top pieces 2, 3 ([t3] = 0: the wave has no fourth piece)")
out.append("#define " + PFX + "KSTEP1_DMA \\\n%s" % lit(kstep(1, "T")))
if W16:   # (the eight-wave kernel takes the state's step between the MFMAs of K-step 5: KSTEP5_ROW below)
    out.append("// the sweep state one row further: [zero] [j0..2] [u0..2] [live] +s, [tmp] =&s, [aaddr] [br0..2] +v, [bu0..2] =v,")
    out.append("// [dstage] [dstep] [dwrap] [tmax] s, [oyl] v; clobbers vcc, scc")
    out.append("#define " + PFX + "ROW_STEP \\\n%s" % lit(row_step()))
if not W16:
    out.append("// K-step 5 + the sweep state one row further + the next row's first fragment reads: [live] s (this row's), [lvn] +s (the")
    out.append("// next row's), [zero] [j0..2] [u0..2] +s, [tmp] =&s, [aaddr] [br0..2] +v, [bu0..2] [bf0..2] =&v, [dstage] [dstep] [dwrap]")
    out.append("// [tmax] s, [vfirst] s (64 bit), [oyl] v, [ib] [pa] [pb] i, [bar] v [target] s (the row barrier's counter); clobbers vcc, scc, v%d, v%d" % (TV, TV + 1))
    out.append("#define " + PFX + "KSTEP5_ROW \\\n%s" % lit(kstep5_row()))
out.append("// the first fragments of an item (into the parity-0 buffers)")
out.append("#define " + PFX + "FIRST_READS \\\n%s" % lit(reads(0)))
out.append("// a finished tile: accumulators of ring slot S to the epilogue image ([addr] v: byte address of the lane's")
out.append("// (channel 4k, row, column) cell; + (m*16 + j) rows of 256 bytes), then zeroed")
for sl in range(RING):
    out.append("#define " + PFX + "DUMP_S%d \\\n%s" % (sl, lit(dump(sl))))
out.append("#define " + PFX + "ZERO \\\n%s" % lit(["v_mov_b32 v%d, 0" % r for r in range(ACC0, LAST + 1)] + ["s_nop 1"]))
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unflow_amd", "csrc", "corr_bwd16_asm.inc" if W16 else "corr_bwd4_asm.inc")
open(path, "w").write("\n".join(out) + "\n")
print("wrote", path)
