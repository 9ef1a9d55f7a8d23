"""condensed view of a kernel's main loop in the gfx950 ISA (no GPU needed):
    python tools/isa_loop.py df-vo_amd/csrc/conv_igemm_f32.hip '<mangled-name prefix>'
prints the instruction stream between the loop header that contains the MFMAs and its back edge, run-length encoded, with the
operands of waits / memory instructions kept."""
import re
import subprocess
import sys

src, key = sys.argv[1], sys.argv[2]
asm = "/tmp/isa_loop.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-S",
                "--cuda-device-only", "-o", asm, src], stderr=subprocess.DEVNULL, check=True)
lines = open(asm).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(key) and l.rstrip().split(";")[0].strip().endswith(":")][0]
end = [i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm")][0]
body = lines[start:end + 1]
mf = [i for i, l in enumerate(body) if "v_mfma" in l]
heads = [i for i, l in enumerate(body) if "Loop Header" in l and i < mf[0]]
lo = heads[-1] if heads else 0
hi = mf[-1]
while hi < len(body) and "s_cbranch" not in body[hi]:
    hi += 1
keep = ("s_waitcnt", "global_load", "ds_read", "ds_write", "s_cbranch", "s_barrier", "scratch_", "buffer_")
out = []
for l in body[lo:hi + 1]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    m = t.split()[0]
    if m.startswith("v_mfma"):
        m = "MFMA"
    elif m.startswith(keep):
        m = m + " " + t.split(None, 1)[1].split(";")[0].strip()[:44] if len(t.split(None, 1)) > 1 else m
    out.append(m)
res, prev, cnt = [], None, 0
for o in out:
    if o == prev:
        cnt += 1
    else:
        if prev is not None:
            res.append(prev if cnt == 1 else "%s x%d" % (prev, cnt))
        prev, cnt = o, 1
res.append(prev if cnt == 1 else "%s x%d" % (prev, cnt))
print("%d instructions, %d MFMAs, %d scratch ops in the loop (lines %d..%d of %d)" % (
    len(out), sum(1 for o in out if o == "MFMA"), sum(1 for o in out if o.startswith("scratch_")), lo, hi, len(body)))
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
print(" | ".join(res)[:limit])
