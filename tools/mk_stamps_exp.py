"""Development aid: derive exp/stamps/ (per-phase s_memtime stamps inside the tile loop) from the product header."""
import os

s = open("salamander_amd/csrc/salnmf_kernels.h").read()


def rep(anchor, new):
    global s
    assert anchor in s, anchor
    s = s.replace(anchor, new, 1)


rep("    unsigned* abort_host;   // pinned host word, set to 1 when a wait gave up\n};", "    unsigned* abort_host;   // pinned host word, set to 1 when a wait gave up\n    unsigned long long* dbg;\n};")
macro = '''#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t1_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory"); __builtin_amdgcn_sched_barrier(0); ph[i] += t1_ - t0_; t0_ = t1_; } while (0)
'''
rep("template <int KS, int KTM, int KR, bool DO_G, bool DO_U, bool DO_STATS, bool WTS = false, bool PERSIST = false>\n__global__", macro + "template <int KS, int KTM, int KR, bool DO_G, bool DO_U, bool DO_STATS, bool WTS = false, bool PERSIST = false>\n__global__")
rep("    d2 hpre[HV];\n    double x[VT][4];\n\n    auto load_tile",
    "    unsigned long long ph[8] = {0,0,0,0,0,0,0,0}; unsigned long long t0_;\n    d2 hpre[HV];\n    double x[VT][4];\n\n    auto load_tile")
rep("        const int64_t n0 = tile * 16;\n        // ---- stage the H tile", "        const int64_t n0 = tile * 16;\n        STAMP(7);\n        // ---- stage the H tile")
rep("        // G-phase A operands (H^T): issue the LDS reads now", "        STAMP(0);\n        // G-phase A operands (H^T): issue the LDS reads now")
rep("        // prefetch the next tile: X and the staging registers are free from here on", "        STAMP(1);\n        // prefetch the next tile: X and the staging registers are free from here on")
rep("        if (DO_U) {\n            // ---- transpose R through LDS", "        STAMP(3);\n        if (DO_U) {\n            // ---- transpose R through LDS")
rep("        if (DO_U) {\n            // ---- U = R . W^T", "        STAMP(2);\n        if (DO_U) {\n            // ---- U = R . W^T")
rep("            // ---- H update (_utils_klnmf.py:343-361), rows n = q+4r", "            STAMP(4);\n            // ---- H update (_utils_klnmf.py:343-361), rows n = q+4r")
rep("    };\n\n    if (tile < p.ntiles) load_tile(tile);", "        STAMP(5);\n    };\n\n    if (tile < p.ntiles) load_tile(tile);\n    asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t0_) :: \"memory\");")
rep("    // ---- workgroup reductions, fixed order (deterministic)", "    if (p.dbg && lane == 0) { for (int i = 0; i < 8; ++i) p.dbg[((int64_t)blockIdx.x * WAVES + wave) * 8 + i] = ph[i]; }\n    // ---- workgroup reductions, fixed order (deterministic)")
os.makedirs("exp/stamps", exist_ok=True)
open("exp/stamps/salnmf_kernels.h", "w").write(s)
print("exp/stamps/salnmf_kernels.h written")
