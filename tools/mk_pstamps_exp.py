"""Development aid: derive exp/pstamps/salnmf_kernels.h: the persistent kernel records s_memrealtime per workgroup and
step at its synchronisation points (timed by tools/pstamps_bench.hip)."""
import os

s = open("salamander_amd/csrc/salnmf_kernels.h").read()


def rep(anchor, new, count=1):
    global s
    assert anchor in s, anchor
    s = s.replace(anchor, new, count)


rep("    unsigned* abort_host;   // pinned host word, set to 1 when a wait gave up\n};", "    unsigned* abort_host;   // pinned host word, set to 1 when a wait gave up\n    unsigned long long* dbg;\n};")
rep("// End of one step of the persistent kernel (kept out of line",
    '#define PSTAMP(i) do { if (dbg_ && tid == 0) { dbg_[i] = __builtin_amdgcn_s_memrealtime(); } } while (0)\n// End of one step of the persistent kernel (kept out of line')
rep("double* W, int K, int V, int n_given, double* lds, int step, int tid) {\n    // publish the slab",
    "double* W, int K, int V, int n_given, double* lds, int step, int tid, unsigned long long* dbg_) {\n    PSTAMP(4);\n    // publish the slab")
rep("    if (tid == 0) bump_counter(sync, SYNC_SLABS);\n", "    PSTAMP(5);\n    if (tid == 0) bump_counter(sync, SYNC_SLABS);\n")
rep("            waited = true;\n", "            waited = true;\n            PSTAMP(6);\n")
rep("        if (tid == 0) bump_counter(sync, SYNC_WROWS);\n    }\n", "        PSTAMP(7);\n        if (tid == 0) bump_counter(sync, SYNC_WROWS);\n    }\n")
rep("    if (PERSIST) asm volatile(\"\" : \"+v\"(tid));\n", "    if (PERSIST) asm volatile(\"\" : \"+v\"(tid));\n    unsigned long long* dbg_ = (PERSIST && p.dbg) ? p.dbg + ((size_t)step * gridDim.x + blockIdx.x) * 8 : nullptr;\n    PSTAMP(0);\n")
rep("        if (step > 0 && !persist_wait_W(p.sync, p.abort_host, (unsigned)step * (unsigned)K, lds, tid)) return;\n",
    "        if (step > 0 && !persist_wait_W(p.sync, p.abort_host, (unsigned)step * (unsigned)K, lds, tid)) return;\n        PSTAMP(1);\n")
rep("        stage_W<G_::WROWS, true>(Wl, p.Wmut, K, V, tid);  // sc1 loads: rows published by other workgroups\n        __syncthreads();\n",
    "        stage_W<G_::WROWS, true>(Wl, p.Wmut, K, V, tid);  // sc1 loads: rows published by other workgroups\n        __syncthreads();\n        PSTAMP(2);\n")
rep("    __syncthreads();  // every wave is done with the LDS copy of W\n", "    __syncthreads();  // every wave is done with the LDS copy of W\n    PSTAMP(3);\n")
rep("p.Wmut, K, V, p.n_given, lds, step, tid)) return;", "p.Wmut, K, V, p.n_given, lds, step, tid, dbg_)) return;")
os.makedirs("exp/pstamps", exist_ok=True)
open("exp/pstamps/salnmf_kernels.h", "w").write(s)
print("exp/pstamps/salnmf_kernels.h written")
