/*
 * salnmf.h -- C ABI of the MI355X-native KL-NMF update engine.
 *
 * The reference (parklab/Salamander @ 2024_10_08) has no FFI: its hot path is a set
 * of NumPy/numba functions called from Python.  This header is therefore the boundary
 * a maintainer would bind (ctypes); each entry point names the
 * reference function (file:line under the reference tree) whose arithmetic it
 * replaces.  Plain pointers and sizes only; no torch / numpy types.
 *
 * Conventions
 *   - All matrices are float64, row-major, in AnnData's storage layout:
 *       X  (n_samples  x n_features)    adata.X
 *       H  (n_samples  x n_signatures)  adata.obsm["exposures"]
 *       W  (n_signatures x n_features)  asignatures.X
 *     (the reference passes the .T views of exactly these buffers,
 *      src/salamander/models/klnmf.py:97-104).
 *   - Every function returns 0 on success, non-zero on failure; the message is
 *     available from salnmf_last_error().  Nothing aborts.
 *   - Host pointers are borrowed for the duration of the call only.  The engine owns
 *     all device memory.  A handle is not re-entrant.
 *   - One engine = one GPU = one shard of the sample axis.
 *
 * Limits of this build (the reference has none; every BASELINE.json configuration fits):
 *   n_features <= 3072, n_signatures <= 512 (salnmf_create: what is available beyond 96 features / 64 signatures),
 *   dim_embeddings <= 64, joint sample solves over at most 4 modalities / 128 signatures.  Anything larger is refused
 *   with a message by salnmf_create / salnmf_corr_configure / salnmf_corr_update_sample_embeddings_multi -- never
 *   silently truncated.  They come from the register / LDS plan of the fused kernel.
 */
#ifndef SALNMF_H
#define SALNMF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct salnmf_engine salnmf_engine;

/* clip floor used everywhere on the path: np.finfo(np.float32).eps as a double
 * (src/salamander/models/_utils_klnmf.py:7). */
#define SALNMF_EPSILON 1.1920928955078125e-07

/* W-tail variants */
#define SALNMF_CLIP_ALL 0       /* update_WH: clip every column, _utils_klnmf.py:341 */
#define SALNMF_CLIP_NON_GIVEN 1 /* update_W: clip only non-given columns, _utils_klnmf.py:215 */

/* device buffers that can be exposed to a caller-side collective (salnmf_device_ptr).
 * G, W, OBJ and RED are compact; X and H are in the engine's padded device layout
 * (X [16*ceil(N/16)][96], H [16*ceil(N/16)][16*ceil(K/16)... see salnmf_kernels.h]). */
#define SALNMF_BUF_G 0    /* (n_signatures x n_features) numerator aux@H.T, local shard */
#define SALNMF_BUF_W 1
#define SALNMF_BUF_H 2
#define SALNMF_BUF_X 3
#define SALNMF_BUF_OBJ 4  /* scalar objective partial of the local shard */
#define SALNMF_BUF_RED 5  /* MvNMF reduction vector: [G (K*V) | rowsums_H (K) | KL (1)] */

const char* salnmf_last_error(void);
int salnmf_version(void);
/* Optional parts compiled into this library: SALNMF_BUILD_PERSISTENT = the persistent multi-step KL kernel
 * (a measured negative; only in builds made with SALNMF_WITH_PERSISTENT=1). */
#define SALNMF_BUILD_PERSISTENT 1
int salnmf_build_flags(void);
int salnmf_device_count(void);

/* Create an engine for a shard of n_samples rows on HIP device `device`.
 * Limits of this build: n_signatures <= 512; n_features <= 3072.  Up to
 * 96 features and 64 signatures (every BASELINE configuration) the whole API is available.  Wider catalogues (SBS-288,
 * SBS-1536, ...) run the KLNMF entry points -- upload / download, salnmf_kl_step (both halves from the old state, as
 * update_WH), salnmf_update_H / _W, the objectives (blocking and queued), salnmf_samplewise_kl, salnmf_reconstruct,
 * per-sample weights -- one 96-feature block of X and W per launch (U = R W^T accumulated over the blocks).  More than 64
 * signatures run the same entry points per chunk of <= 64 signatures: the product H W is accumulated over the chunks by a
 * chain of forward launches, the last of which forms the ratio X / (H W) (or the objective), and the update passes run
 * once per chunk on that ratio.  On feature blocks also: MvNMF (the step in its plain form, salnmf_mv_wide_kernels.h),
 * CorrNMF (its two passes over X block by block; everything else is K- and dim-sized) and the device-side
 * initialisation incl. the separableNMF selection.  On signature chunks also: MvNMF (round 5: the same plain form, its
 * K x K algebra -- Gram matrix, elimination, log det, A and B -- in global memory; mvnmf.py:116-126 has no limit on
 * n_signatures).  Both together (round 5): per feature block the chain over the chunks forms that block's ratio, every chunk
 * runs its numerator pass and its share of U = R W^T on it (accumulated over the blocks); the KLNMF entry points and MvNMF.
 * Sample shards (salnmf_comm_init / salnmf_p2p_connect) of either kind run the KLNMF entry points, the
 * device-side initialisation, MvNMF and -- on feature blocks -- CorrNMF: the numerators of all blocks / chunks cross the
 * ranks in ONE all-reduce of K * V doubles per W update (through RCCL beyond the peer inbox's 16 384 doubles).  The fp32
 * fast mode answers with an error in both cases, and so does CorrNMF on more than 64 signatures.  The device-side
 * initialisation runs everywhere (more than 64 signatures: projection and post-processing chunk by chunk). */
int salnmf_create(int device, int n_features, int64_t n_samples, int n_signatures,
                  salnmf_engine** out);
void salnmf_destroy(salnmf_engine* e);

/* Upload the count matrix.  clip != 0 applies X.clip(EPSILON) on the way, as
 * SignatureNMF._setup_adata does (src/salamander/models/signature_nmf.py:281).
 * Ingest is pinned, chunked and overlapped: the caller's array is copied in 32 MB chunks into two pinned staging
 * buffers by a few host threads, each chunk crosses PCIe by asynchronous DMA and is converted / clipped / padded
 * into the engine's layout by a kernel while the next chunk is being staged.  The typed entry point takes raw count
 * matrices as they come (float32 / int32 / int64 / uint16): the conversion to float64 happens on the device. */
#define SALNMF_F64 0
#define SALNMF_F32 1
#define SALNMF_I32 2
#define SALNMF_I64 3
#define SALNMF_U16 4
int salnmf_upload_X(salnmf_engine* e, const double* X, int clip);
int salnmf_upload_X_typed(salnmf_engine* e, const void* X, int dtype /* SALNMF_F64 ... */, int clip);
int salnmf_upload_W(salnmf_engine* e, const double* W);
int salnmf_upload_H(salnmf_engine* e, const double* H);
/* Per-sample weights (each n_samples long) or NULL to disable
 * (KLNMF._setup_fitting_parameters, klnmf.py:128-153). */
int salnmf_set_weights(salnmf_engine* e, const double* weights_kl, const double* weights_lhalf);
/* normalize_WH's exposure side (utils.py:155-158) + clip, applied lazily: from now on H is read as
 * clip(H * scale[k], EPSILON) by every pass (scale: n_signatures values, the column sums of the raw signatures), and the
 * first pass that rewrites H stores it that way -- the initialisation's 40 MB pass over the exposures disappears into
 * the first update (initialize.py:116-118 with init_method="custom" inside fit). */
int salnmf_set_H_scale(salnmf_engine* e, const double* scale);
int salnmf_download_W(salnmf_engine* e, double* W);
int salnmf_download_H(salnmf_engine* e, double* H);

/* n_steps joint KLNMF updates, device resident: update_WH, _utils_klnmf.py:281-361.
 * The first n_given rows of W are never changed.
 * Opt-in (salnmf_set_persistent, in builds compiled with -DSALNMF_WITH_PERSISTENT): with n_steps >= 2, no per-sample weights and no communicator
 * attached, the steps run in ONE persistent launch (workgroups stay resident; the W tail and its hand-offs happen
 * inside the kernel) -- same bits as step-by-step launches, measured 10 % slower on MI355X, hence off
 * by default.  A wait inside that kernel that gives up (another process holding CUs) is reported by the next
 * synchronising call as an error; the resident state is then invalid. */
int salnmf_kl_step(salnmf_engine* e, int n_steps, int n_given);
/* salnmf_kl_step that keeps the state it starts from: the first step writes the new H / W into second buffers (no
 * copy), which then change roles.  salnmf_kl_rollback returns to the kept state (once, until the next _keep call); the
 * steps it discards may still be running.  For a caller that queues the next block of steps BEFORE it has read the
 * objective that decides convergence (signature_nmf.py:373-380): the decision's host round trip hides behind the block. */
int salnmf_kl_step_keep(salnmf_engine* e, int n_steps, int n_given);
/* Small cohorts: with at most `max_tiles` tiles of 16 samples (default 8 = 128 samples, at most 64), at most 16
 * signatures, no weights, fp64, one device, salnmf_kl_step runs ALL n_steps of a call in ONE launch of ONE workgroup
 * (csrc/salnmf_small.hip: W in LDS, workgroup barriers only) instead of two launches per step: 2.1x / 1.5x the per-step
 * path's rate at 64 / 128 samples.  One workgroup has one CU's matrix pipes, so from 12 tiles on (the reference's own
 * data/pcawg_breast_sbs.csv: 192 samples) the per-step path is as fast; hence the default.  Bit for bit the same W, H as
 * the per-step path at every size (the summation orders are reproduced).  0 turns it off. */
int salnmf_set_small_cohort_tiles(salnmf_engine* e, int max_tiles);
int salnmf_kl_rollback(salnmf_engine* e);
/* The objective (klnmf.py:64-80) of the resident state into slot `slot` of the objective ring (as salnmf_objective_async),
 * then n_steps >= 0 joint updates (keep != 0: as salnmf_kl_step_keep).  This is what SignatureNMF.fit does at every
 * convergence test that does not end the fit (signature_nmf.py:358-385): the first update forms P = H W of exactly the
 * state the objective is about, so -- unweighted, unsharded, n_features <= 96, n_given < n_signatures -- the divergence is
 * evaluated inside that update's launch (one logarithm per entry on top of it) and reduced by a spare workgroup of its
 * W tail; no forward pass of its own.  Otherwise, and for n_steps == 0, the calls named above in sequence.  The value
 * equals salnmf_objective's to rounding (another summation order), not bit for bit. */
int salnmf_kl_step_objective(salnmf_engine* e, int slot, int n_steps, int n_given, int keep);
/* Switch the persistent multi-step launch of salnmf_kl_step on (1) or off (0, the default).  Measurement aid: the
 * default build does not carry that kernel (measured 10 % slower) and refuses on = 1 with an error; build with
 * SALNMF_WITH_PERSISTENT=1 (__graft_entry__.py) to get it. */
int salnmf_set_persistent(salnmf_engine* e, int on);
/* update_H with the current W: _utils_klnmf.py:220-278. */
int salnmf_update_H(salnmf_engine* e);
/* update_W (clip_mode = SALNMF_CLIP_NON_GIVEN): _utils_klnmf.py:164-217. */
int salnmf_update_W(salnmf_engine* e, int n_given, int clip_mode);

/* KLNMF.objective_function: weighted kl_divergence (+ l-half penalty if set),
 * klnmf.py:64-80 and _utils_klnmf.py:11-55.  With a communicator attached the value
 * is all-reduced over the shards. */
int salnmf_objective(salnmf_engine* e, double* out);
/* The same without a host round trip: queue the objective of the resident state into slot `slot` of a device ring
 * (SALNMF_OBJECTIVE_SLOTS entries, pinned host memory the reducing kernel writes directly), read a range of slots later:
 * the read waits for those objectives only, not for work queued behind them.
 * SignatureNMF.fit evaluates the objective every conv_test_freq iterations (signature_nmf.py:373-383) but cannot stop
 * before min_iterations: until then the values are only queued and come back in one read. */
#define SALNMF_OBJECTIVE_SLOTS 256
int salnmf_objective_async(salnmf_engine* e, int slot);
int salnmf_objective_read(salnmf_engine* e, int first, int count, double* out);
/* samplewise_kl_divergence, _utils_klnmf.py:58-97 (unweighted); out has n_samples. */
int salnmf_samplewise_kl(salnmf_engine* e, double* out);
/* H @ W (n_samples x n_features): SignatureNMF.compute_reconstruction,
 * signature_nmf.py:221-224. */
int salnmf_reconstruct(salnmf_engine* e, double* out);

/* MvNMF (src/salamander/models/mvnmf.py).  One call = n_steps times
 * _update_parameters (:197-210): update_H (:162-165), update_W_unconstrained (:37-66),
 * line_search (:69-92).  gamma_inout carries MvNMF._gamma across calls. */
int salnmf_mv_step(salnmf_engine* e, int n_steps, int n_given, double lam, double delta,
                   double* gamma_inout);
/* The pieces of the W step as the reference exposes them (function-level API of mvnmf.py):
 *   salnmf_mv_logdet                  volume_logdet (mvnmf.py:19-24) of the resident W
 *   salnmf_mv_update_W_unconstrained  update_W_unconstrained (:37-66) from the resident (X, W, H) -> Wunc_out [K][V];
 *                                     the resident state is not changed
 *   salnmf_mv_line_search             line_search (:69-92) from the resident (X, W, H) with the caller's
 *                                     W_unconstrained [K][V]: leaves the accepted W and the rescaled H resident, updates gamma */
int salnmf_mv_logdet(salnmf_engine* e, double delta, double* out);
int salnmf_mv_update_W_unconstrained(salnmf_engine* e, int n_given, double lam, double delta, double* Wunc_out);
int salnmf_mv_line_search(salnmf_engine* e, double lam, double delta, double* gamma_inout, const double* Wunc);
/* salnmf_mv_step that also returns MvNMF.objective_function (mvnmf.py:149-156) of the state it leaves behind: the line
 * search of the last step has evaluated exactly that (its accepted f, mvnmf.py:82-89), so a fit loop that tests
 * convergence after a block of steps (signature_nmf.py:373-380) needs no salnmf_mv_objective call -- one forward pass, one
 * log det and one host round trip less per test.  Equal to salnmf_mv_objective's value to rounding.
 * more_follows != 0: the caller expects to go on with another salnmf_mv_step* call (same delta, n_given).  Inside a call
 * every step but the last already runs the first half of its successor speculatively while the host waits for the
 * line-search scalars; with more_follows the last step does so too and the engine keeps that half step across the calls
 * (W is the accepted trial, H already the successor's update).  Any other entry point first steps back to the plain
 * accepted state (no kernel: the pre-update exposures are still there), so results never depend on the flag. */
/* MvNMF steps of an unsharded engine are QUEUED AHEAD of the host (default): per step a tail launch and one pass over the
 * samples (update_H with the trial + the numerator of the next W step: fused_kernel<.., MVJ>); the line-search decision
 * of step i (mvnmf.py:84) is taken on the device, in the prologue of step i + 1's tail; the host reads a flag and the
 * scalars once per call and resolves a rejected first trial on the classic path.  0 = the classic form: one host decision
 * per step, two passes over the samples.  Same results bit for bit. */
int salnmf_set_mv_queued(salnmf_engine* e, int on);
/* The update passes copy W (38 KB at K = 50) into every workgroup's LDS once per launch.  Default (1): by LDS-DMA
 * (global_load_lds_dwordx4, no registers) where n_features == 96 and W is 16-byte aligned; 0: through registers (the
 * form of rounds 1-4; any other layout takes it anyway).  Same results bit for bit (W is copied, not computed). */
int salnmf_set_w_dma(salnmf_engine* e, int on);
int salnmf_mv_step_objective(salnmf_engine* e, int n_steps, int n_given, double lam, double delta,
                             double* gamma_inout, double* objective_out, int more_follows);
/* only MvNMF._update_W (:190-195) / only _update_H (= salnmf_update_H) for the
 * reference's single-step tests (tests/test_mvnmf.py:70-76). */
int salnmf_mv_update_W(salnmf_engine* e, int n_given, double lam, double delta,
                       double* gamma_inout);
/* kl_divergence_penalized, mvnmf.py:27-34. */
int salnmf_mv_objective(salnmf_engine* e, double lam, double delta, double* out);

/* ---- Correlated NMF, dense pieces.  The exposure matrix is not a free
 * parameter but exp(signature scaling + sample scaling + <signature embedding, sample
 * embedding>) (src/salamander/models/corrnmf.py:66-77).  The engine keeps the scalings and
 * embeddings next to X / W / H; H holds the exposures.  Both families of embedding solves
 * (scipy.optimize.minimize(method="Newton-CG") in the reference, _utils_corrnmf.py:354-410) run on
 * the device: salnmf_corr_update_signature_embeddings and salnmf_corr_update_sample_embeddings.
 * One CorrNMFDet._update_parameters (corrnmf_det.py:157-169) is the sequence
 *   update_sample_scalings, compute_exposures, compute_aux, update_signature_scalings,
 *   update_signature_embeddings, update_sample_embeddings, [variance: host scalar], update_signatures. */
#define SALNMF_CORR_SIGNATURE_SCALINGS 0   /* [n_signatures]                       */
#define SALNMF_CORR_SAMPLE_SCALINGS 1      /* [n_samples]                          */
#define SALNMF_CORR_SIGNATURE_EMBEDDINGS 2 /* [n_signatures][dim_embeddings]       */
#define SALNMF_CORR_SAMPLE_EMBEDDINGS 3    /* [n_samples][dim_embeddings]          */
#define SALNMF_CORR_AUX 4                  /* [n_samples][n_signatures] = aux.T    */
/* Allocate the CorrNMF state for embeddings of dimension dim_embeddings (1..64); all zero. */
int salnmf_corr_configure(salnmf_engine* e, int dim_embeddings);
int salnmf_corr_upload(salnmf_engine* e, int which, const double* src);
int salnmf_corr_download(salnmf_engine* e, int which, double* dst);
/* update_sample_scalings, _utils_corrnmf.py:141-179 (CorrNMFDet.update_sample_scalings,
 * corrnmf_det.py:32-44). */
int salnmf_corr_update_sample_scalings(salnmf_engine* e);
/* compute_exposures, _utils_corrnmf.py:11-25: overwrites H. */
int salnmf_corr_compute_exposures(salnmf_engine* e);
/* compute_aux, _utils_corrnmf.py:28-52: aux[n][k] = H[n][k] * sum_v W[k][v] X[n][v] / (HW)[n][v]
 * from the current X, W, H.  The same pass leaves the reduced numerator of update_W in
 * SALNMF_BUF_G for salnmf_corr_update_signatures. */
int salnmf_corr_compute_aux(salnmf_engine* e);
/* update_signature_scalings, _utils_corrnmf.py:103-138, from the aux buffer. */
int salnmf_corr_update_signature_scalings(salnmf_engine* e);
/* CorrNMFDet.update_signatures (corrnmf_det.py:71-86) = update_W, _utils_klnmf.py:164-217, with
 * the numerator of the last salnmf_corr_compute_aux (W and the exposures are unchanged in
 * between in CorrNMFDet._update_parameters). */
int salnmf_corr_update_signatures(salnmf_engine* e, int n_given);
/* CorrNMFDet.update_sample_embeddings (corrnmf_det.py:115-141): one Newton-CG solve per sample
 * (update_embedding, _utils_corrnmf.py:354-410, objective :182-239, gradient :242-293, Hessian
 * :296-351) from the current sample embeddings, using the aux buffer, both scalings and the
 * signature embeddings on the device.  maxiter <= 0 selects SciPy's default (200 * dim);
 * CorrNMFDet passes 3.  status_out (n_samples ints, or NULL) receives per solve 0 = converged,
 * 1 = iteration limit, 2 = line search failed, 3 = CG failed (SciPy's warnflag values).
 * The arithmetic restates SciPy's `_minimize_newtoncg` + `_line_search_wolfe12`
 * (salamander_amd/csrc/salnmf_newtoncg.h). */
int salnmf_corr_update_sample_embeddings(salnmf_engine* e, double variance, int maxiter,
                                         int* status_out);
/* MultimodalCorrNMF.update_sample_embeddings (mmcorrnmf.py:398-428): the sample embeddings are
 * shared by the modalities, so one solve per sample sees the signatures (embeddings, scalings, aux)
 * of all of them and that modality's sample scaling per term.  engines[i] is modality i (1..4
 * engines on one device with equal n_samples and dim_embeddings, at most 128 signatures in total);
 * the result is written to every engine's sample embeddings. */
int salnmf_corr_update_sample_embeddings_multi(salnmf_engine* const* engines, int n_engines,
                                               double variance, int maxiter, int* status_out);
/* CorrNMFDet.update_signature_embeddings (corrnmf_det.py:88-113): one Newton-CG solve per signature
 * (same objective with the roles of signatures and samples exchanged; every evaluation is a pass
 * over all samples, one workgroup per signature).  maxiter <= 0 selects SciPy's default, which is
 * what the reference uses here; status_out has n_signatures ints or is NULL. */
int salnmf_corr_update_signature_embeddings(salnmf_engine* e, double variance, int maxiter,
                                            int* status_out);
/* From 2 048 samples on, the signature solves run in LOCKSTEP (csrc/salnmf_corr_lockstep.h): one evaluation round
 * per launch over (chunks x signatures) workgroups, the K Newton-CG solvers replayed from their evaluation logs
 * between rounds -- 3x faster than one workgroup per signature at c5, and shardable: a sample-sharded engine
 * all-reduces the 66 + dim^2 sums per signature of every round (through the peer exchange where they fit its inbox).  0 forces the single-kernel form; the two agree
 * to rounding of the sums. */
int salnmf_set_lockstep(salnmf_engine* e, int on);
/* The sample solves (both entry points above) run BATCHED where the shape allows (at most 80 signatures over all
 * modalities, dim_embeddings <= 48; csrc/salnmf_corr_batched.hip): sixteen solves per wavefront advance in lockstep
 * rounds, every round's evaluations as two fp64 MFMA products with the signature embeddings, the solver being the
 * resumable form of the same Newton-CG (csrc/salnmf_ncg_machine.h).  0 forces one wavefront per sample
 * (csrc/salnmf_corr_kernels.h), which larger shapes always use; the two agree to rounding of the sums (same solver,
 * same problems, another summation order). */
int salnmf_set_batched_sample_solves(salnmf_engine* e, int on);

/* Opt-in fast mode of salnmf_kl_step: the joint update_WH step on the fp32 matrix cores (fp32 copies of X and H are made
 * on the device; W, the numerator's cross-workgroup sums and the W tail stay fp64; H is converted back at the end of every
 * salnmf_kl_step call, so every other entry point sees the usual fp64 state).  Unweighted steps only.  Tolerance: W, H
 * within 1e-5 rel-L2 of the fp64 step after 20 steps, within 1e-3 after 500 (the reference computes in fp64,
 * _utils_klnmf.py:7-9; the default SALNMF_PRECISION_F64 is the path all parity claims and benchmarks refer to). */
#define SALNMF_PRECISION_F64 0
#define SALNMF_PRECISION_F32_FAST 1
int salnmf_set_precision(salnmf_engine* e, int precision);
/* The same solves with the sample-side inputs handed over by the caller: n_all samples (of ALL shards, in
 * global order) with embeddings U_all (n_all x dim), scalings alpha_all (n_all) and aux_all (n_all x
 * n_signatures, compact), host pointers.  This is the exchange point of a sample-sharded CorrNMF update
 * (mmcorrnmf.py:347-396 / corrnmf_det.py:88-113 need sums over all samples): with a communicator attached
 * salnmf_corr_update_signature_embeddings gathers these three arrays itself over RCCL; this entry point lets a
 * host-side collective (any torch.distributed backend) do the gather instead.  Result: the engine's
 * signature embeddings. */
int salnmf_corr_update_signature_embeddings_from(salnmf_engine* e, int64_t n_all, const double* U_all,
                                                 const double* alpha_all, const double* aux_all,
                                                 double variance, int maxiter, int* status_out);
/* out2[0] = sum of squares of the signature embeddings, out2[1] = of the sample embeddings: what
 * update_variance (corrnmf_det.py:60-69) and the prior terms of elbo_corrnmf
 * (_utils_corrnmf.py:93-98) need from the device-resident embeddings. */
int salnmf_corr_embedding_sumsq(salnmf_engine* e, double* out2);
/* poisson_llh, _utils_klnmf.py:98-160 (the data term of elbo_corrnmf, _utils_corrnmf.py:92). */
int salnmf_corr_poisson_llh(salnmf_engine* e, double* out);

/* ---- Initialisation on the device: the deterministic methods of
 * src/salamander/initialization/methods.py on the resident X, so that a default fit does not spend its time in a
 * host SVD.  The host layer (salamander_amd/device_init.py) drives them; the signature side (n_features <= 96)
 * stays on the host.
 *   init_gram:    gram_out (n_features x n_features) = X^T X and xsum_out = sum of all entries of X (both over ALL
 *                 shards when a communicator is attached).  The right singular vectors of X are its eigenvectors:
 *                 what sklearn's _initialize_nmf (methods.py:83) obtains from a randomized SVD.
 *   init_project: H <- X B^T for B (n_signatures x n_features, host), i.e. the left singular vectors for
 *                 B = V^T / sigma; posneg_out[0..K) / [K..2K) = squared norms of the positive / negative parts of
 *                 every column (all shards): all the NNDSVD sign split needs from the left factor.
 *   init_finish:  H <- clip(fill(threshold(scale_j * part_j(H))) * post_j): the chosen sign part (column 0: |.|),
 *                 sklearn's `W[W < eps] = 0`, the "nndsvda" fill (0 = none), the exposure scaling of normalize_WH
 *                 (initialize.py:116) and the EPSILON clip (:117).
 *   init_flat:    methods.py:58-66 + the same post-processing: H[n][j] = clip(rowsum(X_n) / n_signatures * post_j). */
int salnmf_init_gram(salnmf_engine* e, double* gram_out, double* xsum_out);
int salnmf_init_project(salnmf_engine* e, const double* B, double* posneg_out);
int salnmf_init_finish(salnmf_engine* e, const double* scale, const int* take_neg, const double* post,
                       double zero_below, double fill);
int salnmf_init_flat(salnmf_engine* e, const double* post /* n_signatures */);
/* separableNMF (methods.py:112-135), the signature side: n_select rounds of successive projection on the resident X
 * (every sample normalised to sum 1; argmax of the squared norms with np.argmax's tie rule; rank-1 deflation) ->
 * chosen_out[n_select] = the selected sample indices in selection order.  The exposures of that method come from the
 * host's legacy RNG (methods.py:133) and stay with the host layer.  Unsharded engines only.  norms_out receives the winning squared norm of every round: after deflation by a
 * rank-deficient or duplicated selection the remaining norms are at rounding level and the device's argmax (fused
 * multiply-adds, another summation order) may then differ from np.argmax on the host -- the caller compares
 * norms[k] with norms[0] and keeps the host's selection for such inputs (salamander_amd/models/standard_nmf.py). */
int salnmf_init_separable(salnmf_engine* e, int n_select, int64_t* chosen_out, double* norms_out /* n_select or NULL */);

/* Multi-GPU: one engine per process/GPU, sample axis sharded; the only exchange is an
 * RCCL all-reduce of the (K x V) numerator per W update (+ scalars for objectives; CorrNMF adds one gather
 * of the sample-side inputs of the signature-embedding solves per update).
 * Rank 0 obtains an id, the host layer broadcasts it, every rank calls comm_init. */
#define SALNMF_UNIQUE_ID_BYTES 128
int salnmf_comm_unique_id(char* out_id /* SALNMF_UNIQUE_ID_BYTES */);
int salnmf_comm_init(salnmf_engine* e, const char* id, int n_ranks, int rank);
/* Number of ranks, this engine's rank, and the number of samples over all shards (n_samples of this engine
 * when no communicator is attached).  Any output may be NULL. */
int salnmf_comm_info(salnmf_engine* e, int* n_ranks, int* rank, int64_t* n_samples_total);
/* What the exchange layers THEMSELVES report about this engine, as opposed to what the caller passed to comm_init /
 * p2p_connect (the reference has no counterpart: SURVEY.md 8e; a benchmark line that claims N ranks quotes these):
 *   rccl_nranks, rccl_rank, rccl_device : ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the attached
 *       communicator; -1 without a communicator (or if the RCCL in the process lacks the query)
 *   p2p_nranks        : ranks of the connected peer-to-peer exchange (0 when not connected)
 *   p2p_inboxes_mapped: inboxes this engine holds a mapping of, its own included (== p2p_nranks when connected)
 *   peer_devices      : SALNMF_P2P_MAX_RANKS ints, the HIP device (as this process numbers them) on which rank r's inbox
 *       lives, -1 for ranks beyond p2p_nranks or where the runtime cannot tell for an IPC mapping
 *   device, pci_bus_id: this engine's HIP device and its PCI bus id ("0000:05:00.0", SALNMF_PCI_BUS_ID_BYTES incl. NUL)
 * Any output may be NULL. */
#define SALNMF_P2P_MAX_RANKS 8
#define SALNMF_PCI_BUS_ID_BYTES 32
int salnmf_comm_observed(salnmf_engine* e, int* rccl_nranks, int* rccl_rank, int* rccl_device, int* p2p_nranks, int* p2p_inboxes_mapped,
                         int* peer_devices /* SALNMF_P2P_MAX_RANKS */, int* device, char* pci_bus_id /* SALNMF_PCI_BUS_ID_BYTES */);

/* Peer-to-peer exchange for the small all-reduces (the K x V numerator, the MvNMF line-search sums, objective scalars):
 * every rank stores its vector straight into an inbox on each peer of the node (hipIpc-mapped uncached memory, xGMI
 * peer stores) and adds the n vectors in rank order, so all ranks hold bit-identical sums.  It replaces the library
 * all-reduce where that is pure latency (38 KB at K = 50); larger exchanges keep using RCCL.  One node, <= 8 ranks.
 *   export : allocates this engine's inbox for vectors of up to max_count doubles (<= 16384) and returns its IPC handle
 *   connect: maps the peers' inboxes; `handles` = n_ranks * SALNMF_P2P_HANDLE_BYTES in rank order (the host layer
 *            all-gathers them); n_samples_total = samples over all shards (used when no RCCL communicator is attached)
 *   set_p2p: switch the exchange off / on again (off: the all-reduces go through RCCL, which must then be attached)
 *   set_p2p_timeout_ms: how long a rank waits for a peer inside an exchange before it gives up (default 20 000 ms:
 *            the ranks' host threads may be that far apart)
 * Every rank must issue the same sequence of calls on its engine.  A rank that waits 20 s for a peer gives up: the
 * next download / objective call on that engine fails, and its later exchanges return at once.  Destroy the engines only after all ranks are done. */
#define SALNMF_P2P_HANDLE_BYTES 64
int salnmf_p2p_export(salnmf_engine* e, int n_ranks, int64_t max_count, char* handle_out /* SALNMF_P2P_HANDLE_BYTES */);
int salnmf_p2p_connect(salnmf_engine* e, int rank, int n_ranks, const char* handles, int64_t n_samples_total);
int salnmf_set_p2p(salnmf_engine* e, int on);
int salnmf_set_p2p_timeout_ms(salnmf_engine* e, int64_t timeout_ms);

/* The joint step split at the exchange point, for a caller-side collective
 * (e.g. torch.distributed on a wrapped device pointer):
 *   partial: fused pass + local reduction -> SALNMF_BUF_G, H updated in place
 *   finish : W tail from SALNMF_BUF_G (after the caller's all-reduce) */
int salnmf_kl_step_partial(salnmf_engine* e);
int salnmf_kl_step_finish(salnmf_engine* e, int n_given, int clip_mode);
/* Device address of one of the engine's buffers (SALNMF_BUF_*).  Query it when needed instead of caching it:
 * an MvNMF step exchanges the W buffer with its trial buffer rather than copying. */
void* salnmf_device_ptr(salnmf_engine* e, int which);
void* salnmf_stream(salnmf_engine* e);
int salnmf_sync(salnmf_engine* e);

/* Measurement: run n_steps joint steps on the engine's stream; every sample_stride-th step carries HIP events bound
 * to its two dispatches (start / stop of the fused update kernel and of the W tail: the kernels' own durations, the
 * figure rocprofv3 reports -- events recorded around a launch would add their barrier packets to it).  Outputs (any may
 * be NULL): total ms over the n_steps (events recorded before the first and after the last launch), and the average
 * duration in ms of the fused update kernel and of the W tail over the samples. */
int salnmf_profile_kl_steps(salnmf_engine* e, int n_steps, int n_given, int sample_stride,
                            double* total_ms, double* fused_avg_ms, double* tail_avg_ms);
/* Same for the forward (W@H) + objective kernel: average duration over n_calls. */
/* Where a SHARDED joint step's microseconds go, on this rank: n_steps steps with HIP events bound to the two dispatches of
 * every step and s_memrealtime stamps inside the tail's exchange (csrc/salnmf_p2p_kernels.h: tail_p2p_kernel).  out9, in
 * microseconds: [0] step (stream time / n_steps) [1] fused pass [2] tail + exchange launch; per row workgroup of the tail,
 * averaged over rows and steps: [3] local slab reduction [4] stores to the peers [5] wait for the peers' rows (the payload
 * carries its arrival tag: salnmf_p2p_kernels.h) [6] sum of the rows [7] W row finish; [8] the longest such wait of any row
 * and step.  Needs the
 * peer-to-peer exchange (a one-rank world included); every rank of the world must call it with the same n_steps. */
int salnmf_profile_sharded_steps(salnmf_engine* e, int n_steps, int n_given, double* out9);
int salnmf_profile_objective(salnmf_engine* e, int n_calls, double* avg_ms);
/* Same for the forward kernel alone (H @ W written to a scratch buffer, no objective terms). */
int salnmf_profile_reconstruct(salnmf_engine* e, int n_calls, double* avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* SALNMF_H */
