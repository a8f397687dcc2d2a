"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin or file) per kernel."""
import re
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
KEYS = [("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("SGPR", r"SGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("sspill", r"SGPRs Spill"), ("vspill", r"VGPRs Spill"),
        ("LDS", r"LDS Size \[bytes/block\]")]
for b in txt.split("Function Name: ")[1:]:
    name = b.split()[0]
    vals = []
    for label, key in KEYS:
        m = re.search(key + r": (\S+)", b)
        vals.append("%s=%s" % (label, m.group(1) if m else "?"))
    print("%-72s %s" % (name[:72], " ".join(vals)))
