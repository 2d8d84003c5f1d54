"""profiles/<round>_traffic.json from the rocprofv3 --pmc summaries in profiles/<round>_pmc/ (tools/pmc_r03.sh + tools/rocpd_pmc.py).
usage: python tools/make_traffic_json.py <round: r02 | r03> [precision ...]   (default: every summary_<prec>_{fetch,write,sq}.txt found)"""
import glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r03"
PMC = os.path.join(ROOT, "profiles", RND + "_pmc")


def parse(path):
    out, k = {}, None
    for line in open(path):
        if not line.startswith(" "):
            k = line.strip()
            out[k] = {}
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+mean=(\S+)", line)
            if m:
                out[k][m.group(1)] = float(m.group(2))
    return out


precs = sys.argv[2:] or sorted({os.path.basename(f).split("_")[1] for f in glob.glob(os.path.join(PMC, "summary_*_sq.txt"))})
doc = {"_comment": "HBM traffic and SQ counters per launch from rocprofv3 --pmc (tools/pmc_" + RND + ".sh: separate passes for "
                   "FETCH_SIZE+GRBM_GUI_ACTIVE, WRITE_SIZE and the SQ set; bench.py --steps 10 --warmup 3 --no-settle; mean over "
                   "dispatches after the first 3; counters summed over the instances of a dispatch). FETCH_SIZE / WRITE_SIZE in "
                   "KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half the bytes of wide coalesced "
                   "reads, so hbm_bytes_corrected = 2 x FETCH + WRITE. SQ_* in the units rocprofv3 reports (SQ_WAVE_CYCLES / "
                   "SQ_WAIT_* / SQ_ACTIVE_INST_* quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs). Raw summaries: "
                   "profiles/" + RND + "_pmc/.  Built by tools/make_traffic_json.py.",
       "workload": "B=65536 d=128 L=10 Reg_VAE kl_reg fused step", "precisions": {}}
for p in precs:
    f, w, s = (parse(os.path.join(PMC, f"summary_{p}_{x}.txt")) for x in ("fetch", "write", "sq"))
    ks = {}
    for k in sorted(s):
        if not any(t in k for t in ("dec8_kernel", "enc_bwd_kernel", "enc_fwd_kernel", "step_bf16_kernel", "draw_step_kernel",
                                    "reduce_step")):
            continue
        e = {"FETCH_SIZE_KiB": f[k]["FETCH_SIZE"], "WRITE_SIZE_KiB": w[k]["WRITE_SIZE"]}
        e["hbm_bytes_corrected"] = int(round((2 * e["FETCH_SIZE_KiB"] + e["WRITE_SIZE_KiB"]) * 1024))
        e["GRBM_GUI_ACTIVE"] = f[k].get("GRBM_GUI_ACTIVE")
        e.update(s[k])
        e["mfma_busy_cycles_per_simd"] = s[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
        e["wait_inst_frac_of_wave_cycles"] = round(s[k]["SQ_WAIT_INST_ANY"] / s[k]["SQ_WAVE_CYCLES"], 3)
        e["wait_any_frac_of_wave_cycles"] = round(s[k]["SQ_WAIT_ANY"] / s[k]["SQ_WAVE_CYCLES"], 3)
        ks[k] = e
    doc["precisions"][p] = ks
json.dump(doc, open(os.path.join(ROOT, "profiles", RND + "_traffic.json"), "w"), indent=1)
print("wrote profiles/" + RND + "_traffic.json for", precs)
