"""register / LDS / scratch usage of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage, no GPU
needed): python tools/kernel_resources.py df-vo_amd/csrc/conv_igemm_f32.hip [name filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
names = []
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    names.append((name, g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                  g(r"LDS Size \[bytes/block\]"), g("SGPRs")))
dem = subprocess.run(["c++filt"], input="\n".join(n[0] for n in names), capture_output=True, text=True).stdout.splitlines()
for d, n in zip(dem, names):
    if flt in d:
        print("%-100s vgpr %s agpr %s scratch %s occ %s lds %s sgpr %s" % ((re.sub(r"\(.*", "", d)[-100:],) + n[1:]))
