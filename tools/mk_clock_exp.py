"""Development aid: derive exp/clock/ (whole-kernel stamps + in-kernel clock) from the product kernel header."""
import os

s = open("salamander_amd/csrc/salnmf_kernels.h").read()
s = s.replace("    int64_t ntiles;\n};\n\n// natural log", "    int64_t ntiles;\n    unsigned long long* dbg;\n};\n\n// natural log", 1)
anchor = "    const int64_t N = p.N;\n\n    double* Wl = lds;"
assert anchor in s
s = s.replace(anchor, "    const int64_t N = p.N;\n    unsigned long long T0 = __builtin_amdgcn_s_memtime(), R0 = __builtin_amdgcn_s_memrealtime();\n\n    double* Wl = lds;", 1)
anchor = "    if (tile < p.ntiles) load_tile(tile);\n    for (; tile < p.ntiles; tile += tstride) process_tile(tile);"
assert anchor in s
s = s.replace(anchor, "    unsigned long long T1 = __builtin_amdgcn_s_memtime();\n    if (tile < p.ntiles) load_tile(tile);\n    for (; tile < p.ntiles; tile += tstride) process_tile(tile);\n    unsigned long long T2 = __builtin_amdgcn_s_memtime();", 1)
marker = "// ----------------------------------------------------------------------------------------------\n// Forward pass + objective"
i = s.index(marker)
j = s.rfind("}\n", 0, i)
s = s[:j] + "    { unsigned long long T3 = __builtin_amdgcn_s_memtime(), R3 = __builtin_amdgcn_s_memrealtime(); if (p.dbg && lane == 0) { unsigned long long* d = p.dbg + ((int64_t)blockIdx.x * WAVES + wave) * 8; d[0] = T1 - T0; d[1] = T2 - T1; d[2] = T3 - T2; d[3] = R3 - R0; d[4] = T3 - T0; d[5] = R0; d[6] = R3; } }\n" + s[j:]
os.makedirs("exp/clock", exist_ok=True)
open("exp/clock/salnmf_kernels.h", "w").write(s)
print("exp/clock/salnmf_kernels.h written")
