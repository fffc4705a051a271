"""Per-kernel summary of the tools/pmc_iwim.sh passes: the image gradient of image_warp (per template instance) and
downsample_pyramid_kernel.  FETCH_SIZE is doubled (gfx950 tallies a 128-byte request of a wide coalesced read at 64 bytes,
MI355X_MICROARCH.md section HBM); both size counters are in KB.  Normalisations as tools/pmc_loss_report.py: kernel cycles =
GRBM_GUI_ACTIVE / 8, VALU issue = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * cycles), LDS active = SQ_LDS_IDX_ACTIVE / (256 CUs * cycles).  usage: python tools/pmc_iwim_report.py gpurun_out/pmc_iwim_*/"""
import collections, csv, glob, os, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            m = re.search(r"(image_warp_bwd_im_tiled_kernel<[^>]*>|downsample_pyramid_kernel<[^>]*>)", k)
            if m:
                acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
ALG = {"image_warp_bwd_im_tiled_kernel<3": 32 * 384 * 1280 * 32.0, "image_warp_bwd_im_tiled_kernel<2": 32 * 384 * 1280 * 24.0,
       "downsample_pyramid_kernel<3": 2 * 32 * 384 * 1280 * 12.0 * (1 + 2 * (1 / 16 + 1 / 64 + 1 / 256 + 1 / 1024 + 1 / 4096))}
for k, cs in sorted(acc.items()):
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    g = lambda n: c.get(n, float("nan"))
    waves = g("SQ_WAVES")
    alg = next((v for p, v in ALG.items() if k.startswith(p)), float("nan"))
    hbm = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024
    cyc = g("GRBM_GUI_ACTIVE") / 8                      # (the counter sums the 8 XCDs)
    print("%-52s cycles %9.0f  waves %7.0f | VALU issue %4.1f %%  LDS active %4.1f %% (conflicts %4.1f %% of it)  waiting on an instruction %4.1f %% of wave cycles"
          % (k, cyc, waves, 100 * g("SQ_ACTIVE_INST_VALU") * 4 / (1024 * cyc), 100 * g("SQ_LDS_IDX_ACTIVE") / (256 * cyc),
             100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else float("nan"),
             100 * g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else float("nan")))
    print("%-52s insts / wave: VALU %6.0f  SALU %6.0f  LDS %5.0f  VMEM rd %4.0f  wr %4.0f | HBM read %7.1f MB (2 x FETCH_SIZE)  written %7.1f MB  = %.2f x the algorithmic %.1f MB"
          % ("", g("SQ_INSTS_VALU") / waves, g("SQ_INSTS_SALU") / waves, g("SQ_INSTS_LDS") / waves, g("SQ_INSTS_VMEM_RD") / waves,
             g("SQ_INSTS_VMEM_WR") / waves, 2 * g("FETCH_SIZE") * 1024 / 1e6, g("WRITE_SIZE") * 1024 / 1e6, hbm / alg, alg / 1e6))
    print("   raw: " + "  ".join("%s=%.4g" % (n, v) for n, v in sorted(c.items())))
