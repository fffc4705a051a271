"""profiles/rNN_pmc_corr.txt from tools/profile_round.sh's raw counter summary.
usage: python tools/pmc_corr_report.py gpurun_out/profiles_r02/r02_pmc_corr_raw.txt > profiles/r02_pmc_corr.txt"""
import re, sys
raw = open(sys.argv[1]).read()


def grab(block, name):
    return float(re.search(name + r"\s+([0-9.]+)", block).group(1))


blocks = raw.split("== ")[1:]
fwd = [b for b in blocks if b.startswith("corr_fwd2_kernel")][0]
bwd = [b for b in blocks if b.startswith("corr_mfma_bwd4_kernel")][0]


def line(tag, b):
    busy = grab(b, "SQ_VALU_MFMA_BUSY_CYCLES") / 1024
    dur = grab(b, "GRBM_GUI_ACTIVE") / 8
    return ("  %s: %d/1024 = %.1fK of %.1fK cycles = %.1f %%   (%.2f M v_mfma_f32_16x16x4_f32 x 32 cycles); LDS bank conflicts %.1f %% of the "
            "LDS cycles; waves parked (SQ_WAIT_ANY) %.1f %% of the wave cycles; L2 hit rate %.1f %%" % (
                tag, grab(b, "SQ_VALU_MFMA_BUSY_CYCLES"), busy / 1e3, dur / 1e3, 100 * busy / dur, grab(b, "SQ_INSTS_MFMA") / 1e6,
                100 * grab(b, "SQ_LDS_BANK_CONFLICT") / grab(b, "SQ_LDS_IDX_ACTIVE"), 100 * grab(b, "SQ_WAIT_ANY") / grab(b, "SQ_WAVE_CYCLES"),
                100 * grab(b, "TCC_HIT_sum") / (grab(b, "TCC_HIT_sum") + grab(b, "TCC_MISS_sum"))))


print("""PMC counters of the correlation kernels (tools/profile_round.sh: rocprofv3 --pmc in 6 separate passes on tools/corr_only.py, B=32 [32,256,48,160]); values are per launch, summed over the chip
MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs):
%s
%s
(round 3, profiles/r03_pmc_corr.txt: forward 63.1 %% busy, 33 %% LDS conflicts; backward 69.8 %% busy, 28 %% LDS conflicts)

---- forward
== %s
---- backward (mean of the two gradient launches of one call; FETCH_SIZE / WRITE_SIZE in KB)
== %s""" % (line("forward (corr_fwd2_kernel)", fwd), line("backward, one gradient (corr_mfma_bwd4_kernel<WRT1>)", bwd), fwd, bwd))
