"""Development aid: derive exp/wgstamps/salnmf_kernels.h from the product header: every wave of the fused kernel
records s_memrealtime (100 MHz, chip-wide clock) at entry / W staged / first tile done / loop done / after the
barrier in front of the reductions / kernel end.  Timed by tools/wgstamps_bench.hip."""
import os

s = open("salamander_amd/csrc/salnmf_kernels.h").read()


def rep(anchor, new):
    global s
    assert anchor in s, anchor
    s = s.replace(anchor, new, 1)


rep("    unsigned* abort_host;   // pinned host word, set to 1 when a wait gave up\n};", "    unsigned* abort_host;   // pinned host word, set to 1 when a wait gave up\n    unsigned long long* dbg;\n};")
macro = '''#define RSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t1_; asm volatile("s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory"); __builtin_amdgcn_sched_barrier(0); st_[i] = t1_; } while (0)
'''
rep("template <int KS, int KTM, int KR, bool DO_G, bool DO_U, bool DO_STATS, bool WTS = false, bool PERSIST = false>\n__global__", macro + "template <int KS, int KTM, int KR, bool DO_G, bool DO_U, bool DO_STATS, bool WTS = false, bool PERSIST = false>\n__global__")
rep("    int tid = threadIdx.x;\n    if (PERSIST) asm volatile(\"\" : \"+v\"(tid));\n", "    unsigned long long st_[8] = {0,0,0,0,0,0,0,0};\n    RSTAMP(0);\n    int tid = threadIdx.x;\n    if (PERSIST) asm volatile(\"\" : \"+v\"(tid));\n")
rep("        stage_W<G_::WROWS>(Wl, p.W, K, V, tid);\n        __syncthreads();\n        if (tile < nfull) load_tile(tile);\n    }\n    for (; tile < nfull; tile += tstride) process_tile(tile);\n    if (COOP && coop && (int64_t)blockIdx.x < nleft) process_tile_coop(nfull + blockIdx.x);",
    "        stage_W<G_::WROWS>(Wl, p.W, K, V, tid);\n        __syncthreads();\n        RSTAMP(1);\n        if (tile < nfull) load_tile(tile);\n    }\n    { bool first_ = true; for (; tile < nfull; tile += tstride) { process_tile(tile); if (first_) { RSTAMP(2); first_ = false; } } }\n    RSTAMP(3);\n    if (COOP && coop && (int64_t)blockIdx.x < nleft) process_tile_coop(nfull + blockIdx.x);\n    RSTAMP(6);")
rep("    __syncthreads();  // every wave is done with the LDS copy of W\n", "    __syncthreads();  // every wave is done with the LDS copy of W\n    RSTAMP(4);\n")
# end of the fused kernel: the DO_STATS && DO_G block is the last statement
rep("        if (tid == 0) p.KLpart[blockIdx.x] = Ks[0];\n    }\n    }  // step\n}", "        if (tid == 0) p.KLpart[blockIdx.x] = Ks[0];\n    }\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    RSTAMP(5);\n    if (p.dbg && lane == 0) { for (int i = 0; i < 8; ++i) p.dbg[((int64_t)blockIdx.x * WAVES + wave) * 8 + i] = st_[i]; }\n    }  // step\n}")
rep("    int nparts;\n};\n\n__global__ void __launch_bounds__(TAIL_BLOCK) tail_kernel", "    int nparts;\n    unsigned long long* dbg;\n};\n\n__global__ void __launch_bounds__(TAIL_BLOCK) tail_kernel")
rep("    __shared__ TailScratch S;\n    const int k = blockIdx.x;\n", "    __shared__ TailScratch S;\n    unsigned long long ts0_; asm volatile(\"s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(ts0_) :: \"memory\");\n    const int k = blockIdx.x;\n")
rep("    tail_row<TAIL_BLOCK, false>(S, threadIdx.x, k, p.Gpart, p.nslabs, p.G, p.W, p.V, K, p.n_given, p.clip_mode, p.do_tail != 0);\n}", "    tail_row<TAIL_BLOCK, false>(S, threadIdx.x, k, p.Gpart, p.nslabs, p.G, p.W, p.V, K, p.n_given, p.clip_mode, p.do_tail != 0);\n    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n    if (p.dbg && threadIdx.x == 0) { unsigned long long ts1_; asm volatile(\"s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(ts1_) :: \"memory\"); p.dbg[2 * k] = ts0_; p.dbg[2 * k + 1] = ts1_; }\n}")
os.makedirs("exp/wgstamps", exist_ok=True)
open("exp/wgstamps/salnmf_kernels.h", "w").write(s)
print("exp/wgstamps/salnmf_kernels.h written")
