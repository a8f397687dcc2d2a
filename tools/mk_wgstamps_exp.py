"""Development aid: derive exp/wgstamps/salnmf_kernels.h from the product header: every wave of the fused kernel
records s_memrealtime (100 MHz, chip-wide clock) at entry / W staged / first tile done / loop done / after the
barrier in front of the reductions / kernel end.  Timed by tools/wgstamps_bench.hip."""
import os

s = open("salamander_amd/csrc/salnmf_kernels.h").read()


def rep(anchor, new):
    global s
    assert anchor in s, anchor
    s = s.replace(anchor, new, 1)


rep("    int64_t ntiles;\n};\n\n// log(x / p)", "    int64_t ntiles;\n    unsigned long long* dbg;\n};\n\n// log(x / p)")
macro = '''#define RSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t1_; asm volatile("s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory"); __builtin_amdgcn_sched_barrier(0); st_[i] = t1_; } while (0)
'''
rep("template <int KS, int KTM, int KR, bool DO_G, bool DO_U, bool DO_STATS, bool WTS = false>\n__global__", macro + "template <int KS, int KTM, int KR, bool DO_G, bool DO_U, bool DO_STATS, bool WTS = false>\n__global__")
rep("    const int tid = threadIdx.x;\n    const int lane = tid & 63;\n    const int wave = tid >> 6;\n    const int c16 = lane & 15;\n    const int q = lane >> 4;\n    const int V = p.V, K = p.K;\n    const int64_t N = p.N;\n    const double* __restrict__ const wkl",
    "    unsigned long long st_[8] = {0,0,0,0,0,0,0,0};\n    RSTAMP(0);\n    const int tid = threadIdx.x;\n    const int lane = tid & 63;\n    const int wave = tid >> 6;\n    const int c16 = lane & 15;\n    const int q = lane >> 4;\n    const int V = p.V, K = p.K;\n    const int64_t N = p.N;\n    const double* __restrict__ const wkl")
rep("    stage_W<G_::WROWS>(Wl, p.W, K, V, tid);\n    __syncthreads();\n    for (; tile < p.ntiles; tile += tstride) process_tile(tile);",
    "    stage_W<G_::WROWS>(Wl, p.W, K, V, tid);\n    __syncthreads();\n    RSTAMP(1);\n    { bool first_ = true; for (; tile < p.ntiles; tile += tstride) { process_tile(tile); if (first_) { RSTAMP(2); first_ = false; } } }\n    RSTAMP(3);")
rep("    __syncthreads();  // every wave is done with the LDS copy of W\n", "    __syncthreads();  // every wave is done with the LDS copy of W\n    RSTAMP(4);\n")
# end of the fused kernel: the DO_STATS && DO_G block is the last statement
rep("        if (tid == 0) p.KLpart[blockIdx.x] = Ks[0];\n    }\n}", "        if (tid == 0) p.KLpart[blockIdx.x] = Ks[0];\n    }\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    RSTAMP(5);\n    if (p.dbg && lane == 0) { for (int i = 0; i < 8; ++i) p.dbg[((int64_t)blockIdx.x * WAVES + wave) * 8 + i] = st_[i]; }\n}")
rep("    int nparts;\n};\n\nconstexpr int TAIL_PARTS", "    int nparts;\n    unsigned long long* dbg;\n};\n\nconstexpr int TAIL_PARTS")
rep("    const int k = blockIdx.x;\n    const int v = threadIdx.x % VMAX;", "    unsigned long long ts0_; asm volatile(\"s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(ts0_) :: \"memory\");\n    const int k = blockIdx.x;\n    const int v = threadIdx.x % VMAX;")
rep("        p.W[k * V + v] = w;\n    }\n}", "        p.W[k * V + v] = w;\n    }\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    if (p.dbg && threadIdx.x == 0) { unsigned long long ts1_; asm volatile(\"s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(ts1_) :: \"memory\"); p.dbg[2 * k] = ts0_; p.dbg[2 * k + 1] = ts1_; }\n}")
os.makedirs("exp/wgstamps", exist_ok=True)
open("exp/wgstamps/salnmf_kernels.h", "w").write(s)
print("exp/wgstamps/salnmf_kernels.h written")
