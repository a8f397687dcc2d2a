// Parameter block of the CorrNMF sample-embedding solves, shared by the one-wavefront-per-sample kernel
// (salnmf_corr_kernels.h) and the batched lockstep kernel (salnmf_corr_batched.h, its own translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace salnmf {

constexpr int CORR_MODS = 4;     // modalities per joint solve
constexpr int CORR_TERMS = 128;  // signatures of all modalities together

struct SampleEmbeddingParams {
    const double* aux[CORR_MODS];    // [Np][KP_mod]
    const double* alpha[CORR_MODS];  // [Np]
    const double* beta[CORR_MODS];   // [K_mod]
    const double* L[CORR_MODS];      // [K_mod][dim]
    int K[CORR_MODS], KP[CORR_MODS];
    int n_mod;
    double* U;                       // [N][dim]  in / out (shared by the modalities)
    int* status;                     // [N] or null: ncg::Status of every solve
    double variance;
    int64_t N;
    int dim, maxiter;
};

// Batched solves (salnmf_corr_batched.hip): true if a kernel instantiation covers (terms, dim) and was launched.
bool launch_sample_embeddings_batched(const SampleEmbeddingParams& p, int terms, hipStream_t stream);

}  // namespace salnmf
