"""Print VGPR / scratch / LDS / occupancy per kernel from hipcc -Rpass-analysis=kernel-resource-usage output."""
import glob, os, re, subprocess, sys
d = sys.argv[1] if len(sys.argv) > 1 else "vae-posterior-consistency_amd/csrc/build"
for f in sorted(glob.glob(os.path.join(d, "*.resource.txt"))):
    txt = open(f).read()
    for blk in txt.split("Function Name:")[1:]:
        name = blk.split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        g = lambda k: re.search(re.escape(k) + r":\s*(\d+)", blk)
        vals = {k: (g(k).group(1) if g(k) else "?") for k in
                ["VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs", "LDS Size [bytes/block]"]}
        short = re.sub(r"vpc::|\(vpc::\w+\)", "", dem)[:70]
        print(f"{short:70s} vgpr={vals['VGPRs']:>4} agpr={vals['AGPRs']:>4} sgpr={vals['SGPRs']:>4} scratch={vals['ScratchSize [bytes/lane]']:>5} occ={vals['Occupancy [waves/SIMD]']}")
