"""profiles/r03_pmc_loss.txt and .json from the PMC passes of tools/pmc_loss_step.sh: per loss kernel of the bench
step its bound, the fraction of that bound it reaches and the counters behind it.
  census (ternary_*_pair_kernel<3,16,true>): VALU issue.  frac = SQ_ACTIVE_INST_VALU (quad-cycles, summed over the
      waves) * 4 / (1024 SIMDs * kernel cycles); beside it the instruction-count model: a wave walks 19 rows x 24 pair
      taps with 3 transcendentals each (8 issue cycles; every other VALU instruction 4).
  forward_warp_fwd_kernel: LDS integer atomics.  frac = waves * 81 taps * 5.3 cycles (conflict-free ds_add_u32 per CU,
      profiles/ubench_lds_atomic.txt) / (256 CUs * kernel cycles).
  warps / occlusion / scatter: HBM (PMC bytes and algorithmic bytes per launch against 8 TB/s).
Kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs)."""
import csv, glob, json, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            if "unflow::" not in k:
                continue
            k = k.split("unflow::(anonymous namespace)::")[-1]
            # cut the argument list: the first "(" outside the template brackets
            depth, cut = 0, len(k)
            for i, ch in enumerate(k):
                depth += ch == "<"
                depth -= ch == ">"
                if ch == "(" and depth == 0:
                    cut = i
                    break
            k = k[:cut].replace("<HIP_vector_type<float, 2u> >", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
NPX = 32 * 384 * 1280
ALG = {"ternary_fwd_pair_kernel<3, 16, true>": 28 + 8 + 8, "ternary_bwd_pair_kernel<3, 16, true>": 8 + 4 + 12 + 8 + 8,
       "forward_warp_fwd_kernel": 12, "fb_occlusion_warp_tile_kernel": 24, "sym_loss_kernel": 8,
       "image_warp_bwd_im_tiled_kernel<3, 8>": 32, "image_warp_bwd_im_tiled_kernel<2, 16>": 24}
out = {}
print(__doc__)
for k in sorted(acc):
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    if "GRBM_GUI_ACTIVE" not in c or c.get("SQ_WAVES", 0) < 64:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    waves = c.get("SQ_WAVES", 0)
    rec = {"kernel_cycles": cyc, "waves": waves,
           "valu_issue_frac": c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * cyc),
           "lds_active_frac": c.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc),
           "lds_conflict_frac_of_lds": c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1),
           "wait_any_frac_of_wave_cycles": c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
           "valu_insts_per_wave": c.get("SQ_INSTS_VALU", 0) / max(waves, 1), "lds_insts_per_wave": c.get("SQ_INSTS_LDS", 0) / max(waves, 1),
           "hbm_read_MB": 2 * c.get("FETCH_SIZE", 0) / 1e3, "hbm_write_MB": c.get("WRITE_SIZE", 0) / 1e3}
    if k.startswith("ternary_"):
        trans = 19 * 24 * 3
        model_cycles = waves * ((rec["valu_insts_per_wave"] - trans) * 4 + trans * 8) / 1024
        rec.update(bound="valu issue", frac=rec["valu_issue_frac"], model_issue_cycles=model_cycles, model_frac=model_cycles / cyc)
    elif k.startswith("forward_warp_fwd"):
        floor = waves * 81 * 5.3 / 256
        rec.update(bound="lds integer atomics", frac=floor / cyc, floor_cycles=floor)
    elif k in ALG:
        rec.update(bound="hbm", frac=None)
    if k in ALG:
        rec["algorithmic_MB"] = ALG[k] * NPX / 1e6
    out[k] = rec
    print("%-46s cycles %9.0f  waves %7d | VALU issue %5.1f %%  LDS active %5.1f %% (conflicts %4.1f %% of it)  parked %4.1f %% | insts/wave VALU %6.0f LDS %5.0f | HBM rd %7.1f wr %7.1f MB%s%s" % (
        k[:46], cyc, waves, 100 * rec["valu_issue_frac"], 100 * rec["lds_active_frac"], 100 * rec["lds_conflict_frac_of_lds"],
        100 * rec["wait_any_frac_of_wave_cycles"], rec["valu_insts_per_wave"], rec["lds_insts_per_wave"], rec["hbm_read_MB"], rec["hbm_write_MB"],
        (" (algorithmic %.1f MB)" % rec["algorithmic_MB"]) if "algorithmic_MB" in rec else "",
        (" | bound: %s, frac %.3f" % (rec["bound"], rec["frac"])) if rec.get("frac") is not None else ""))
json.dump(out, open(os.path.join(os.path.dirname(sys.argv[1].rstrip("/")), "r03_pmc_loss.json"), "w"), indent=1)
