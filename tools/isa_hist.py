"""Instruction histogram of a kernel's MFMA region in a hipcc -save-temps .s file (development aid)."""
import re
import sys
from collections import Counter

path, pat = sys.argv[1], sys.argv[2]
txt = open(path).read()
m = re.search(r"^(%s\S*):.*?s_endpgm" % pat, txt, flags=re.S | re.M)
lines = m.group(0).split("\n")
idx = [i for i, l in enumerate(lines) if "v_mfma" in l]
print(m.group(1)[:100], "| lines", len(lines), "| mfma", len(idx), "first", idx[0], "last", idx[-1])
c = Counter()
for l in lines[max(0, idx[0] - 150): idx[-1] + 250]:
    l = l.strip()
    if not l or l[0] in ";." or l.endswith(":"):
        continue
    c[l.split()[0]] += 1
print("  ".join(f"{k}={v}" for k, v in c.most_common(40)))
mf = [l for l in lines if "v_mfma" in l]
print("mfma with AGPR dst:", sum(1 for l in mf if re.search(r"v_mfma\S+ a\[", l)), "of", len(mf))
