/*
 * icsg3d.h -- C ABI of libicsg3d_hip.so, the MI355X (gfx950) engine for the ICSG3D hot path.
 *
 * The reference (by256/icsg3d) has no FFI: its boundary is the Python class surface of
 * unet/unet.py (AtomUnet) and vae/lattice_vae.py (LatticeDFCVAE) over Keras.  This header is what
 * those classes bind instead of Keras; each entry point cites the reference call it stands in for
 * (paths relative to the reference root).  icsg3d_amd/_lib.py is the ctypes binding;
 * INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: every function returns 0 on success, non-zero on failure with the message in
 * ics_last_error() (thread-local).  Host tensors are caller-owned, dense, row-major, channels-last
 * (B,D,H,W,C) float32 unless stated; the library copies them.  All device memory is owned by the
 * handle and released by *_destroy.  A handle is bound to the HIP device current at creation and is
 * not thread-safe (the reference drives training from one thread: unet/unet.py:370).
 */
#ifndef ICSG3D_H
#define ICSG3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- library / device */
const char* ics_last_error(void);
const char* ics_version(void);
/* Device kernels this process has enqueued through the library so far (every launch site counts itself; the
 * runtime's copy / fill kernels and RCCL's kernels are not included).  bench.py reports the per-step difference as
 * kernel_launches_per_step; rocprofv3 --kernel-trace of the same command shows the same number of ics:: kernels. */
long long ics_kernel_launches(void);
int ics_device_count(int* count);
int ics_set_device(int device);
/* HIP device properties the benchmark reports (name is a 256-byte buffer). */
int ics_device_info(char* name, int* compute_units, size_t* hbm_bytes);

typedef struct ics_net ics_net; /* opaque: a U-Net or a DFC-VAE engine */

/* ---------------------------------------------------------------- AtomUnet
 * AtomUnet.__init__ / unet_3d_multiclass            (unet/unet.py:235-355)
 * loss = weighted_categorical_crossentropy(num_classes) + binary_crossentropy, Adam(lr)
 *                                                   (unet/unet.py:245-259) */
typedef struct ics_unet_config {
  int in_channels;   /* C of input_shape=(d,d,d,C): 1 or 4          (unet/unet.py:240) */
  int num_classes;   /* 95                                          (unet/unet.py:237) */
  int d;             /* grid edge, power of two >= 8 (32; 64 is a build extension) */
  int max_batch;     /* largest batch a single call may carry                             */
  float lr;          /* Adam learning rate                          (unet/unet.py:241,245) */
  float loss_weight; /* scalar class weight; <=0 selects float(num_classes)  */
  int pool_ties_all; /* 1: TF-CPU MaxPool3DGrad tie rule (default), 0: first max only     */
  int bn_unbias;     /* 1: Keras moving-variance n/(n-(1+eps)) rescale (default)          */
  int bce_from_logits; /* "binary_crossentropy" of the sig head (unet/unet.py:254-256): 0 (default) = the clipped-probability
                        * form -[t log p + (1-t) log(1-p)], p in [1e-7, 1-1e-7]; 1 = tf.keras.backend.binary_crossentropy's
                        * short-circuit for a Sigmoid producer op, sigmoid_cross_entropy_with_logits (TF 2.1; SURVEY App. B,
                        * confidence M).  The two agree to ~1e-7 relative except at saturated probabilities.       */
} ics_unet_config;

int ics_unet_create(const ics_unet_config* cfg, ics_net** out);

/* model.predict(X) -> [soft (B,d,d,d,num_classes), sig (B,d,d,d,1)]   (unet/unet.py:98,383-385;
 * callers generate.py:220, eval.py:166).  Eval-mode BatchNorm. */
int ics_unet_predict(ics_net* net, const float* x, int batch, float* soft, float* sig);
/* Fused inference tail (generate.py:220-225): argmax species + (sig >= thresh) mask, uint8. */
int ics_unet_predict_labels(ics_net* net, const float* x, int batch, float thresh,
                            uint8_t* species, uint8_t* mask);
/* model.train_on_batch / the per-batch step inside fit_generator (unet/unet.py:370): forward in
 * training mode, loss, backward, BN moving-stat update, Adam.  labels are uint8 class ids
 * (B,d,d,d) -- the one-hot of unet/data.py:89 is never built; the sigmoid target is labels != 0
 * (unet/data.py:87).  metrics = [Loss, lsoft, lsig, f1, wr] (unet/unet.py:250). */
int ics_unet_train_step(ics_net* net, const float* x, const uint8_t* labels, int batch,
                        float metrics[5]);
/* model.test_on_batch / validation pass: eval-mode BN, same metrics, no update. */
int ics_unet_test_step(ics_net* net, const float* x, const uint8_t* labels, int batch,
                       float metrics[5]);
/* The K.sum(...) terms behind the ratios of r_m / p_m / f1_m / wr_m (unet/unet.py:159-193) and the two loss means of
 * the LAST train / test step (data parallel: after the all-reduce, i.e. over the global batch):
 * sums = [sum_voxels wcce, sum_voxels bce, true_positives, predicted_positives, wr true_positives, wr possible_positives,
 * voxel count (= possible_positives of r_m: y_true is one-hot)].  The counts are integers held in doubles.
 * Valid only after a step that RETURNED metrics (ics_unet_train_step / _test_step, or _train_step_resident with a non-NULL
 * metrics pointer): a resident step enqueued without metrics skips the reduction over the ranks under data parallelism, and
 * the sums then still hold the last metric-returning step's values. */
int ics_unet_metric_sums(ics_net* net, double sums[7]);

/* Benchmark path: batch resident in HBM, steps enqueued back-to-back on the engine's stream. */
int ics_unet_upload_batch(ics_net* net, const float* x, const uint8_t* labels, int batch);
int ics_unet_train_step_resident(ics_net* net, float metrics_or_null[5]);
/* model.predict on the resident batch (unet/unet.py:383-385; BASELINE configs[0]), enqueued only: eval-mode forward, the
 * outputs stay in HBM -- probabilities, or with labels_only the uint8 argmax / (sig >= thresh) volumes (generate.py:221-225). */
int ics_unet_predict_resident(ics_net* net, int labels_only, float thresh);

/* ---------------------------------------------------------------- LatticeDFCVAE
 * LatticeDFCVAE.__init__/_set_model/build_encoder/build_decoder (vae/lattice_vae.py:89-230);
 * loss = mse + alpha*perceptual + beta*kld (vae/lattice_vae.py:232-270). The perceptual U-Net is
 * an ics_net created by ics_unet_create whose weights stay frozen (vae/lattice_vae.py:120). */
typedef struct ics_vae_config {
  int in_channels;  /* C                                           (vae/lattice_vae.py:91)  */
  int cond_shape;   /* 10                                          (vae/lattice_vae.py:102) */
  int latent_dim;   /* 256                                         (vae/lattice_vae.py:95)  */
  int filters[4];   /* 16,32,64,128                                (vae/lattice_vae.py:94)  */
  int d;            /* 32 (decoder seed (d/8)^3*4, encoder flatten (d/16)^3*4)              */
  int max_batch;
  float lr;         /* Adam(5e-4)                                  (vae/lattice_vae.py:98)  */
  float alpha;      /* 0.5  */
  float beta;       /* 3e-4 */
  float pm_layer_weights[4]; /* 1,1,1,1 for re_lu_2/4/6/8          (vae/lattice_vae.py:100-101) */
  int bn_unbias;
} ics_vae_config;

int ics_vae_create(const ics_vae_config* cfg, ics_net* perceptual_unet, ics_net** out);

/* encoder.predict([M, cond]) -> (z_mean, z_log_var, z)  (generate.py:196, interpolate.py:50-59);
 * eps is the N(0,1) draw of sampling() (vae/lattice_vae.py:53-66), injected by the caller. */
int ics_vae_encode(ics_net* net, const float* x, const float* cond, const float* eps, int batch,
                   float* z_mean, float* z_log_var, float* z);
/* decoder.predict([z, cond]) -> (B,d,d,d,C)             (generate.py:208, lattice_vae.py:351) */
int ics_vae_decode(ics_net* net, const float* z, const float* cond, int batch, float* recon);
/* model.train_on_batch([M,cond], M) (vae/lattice_vae.py:296) -> [Loss, PM, MSE, KLD]; the
 * perceptual U-Net runs BatchNorm on batch statistics, frozen. */
int ics_vae_train_step(ics_net* net, const float* x, const float* cond, const float* eps, int batch,
                       float metrics[4]);
/* model.test_on_batch (vae/lattice_vae.py:307): learning phase 0 everywhere. */
int ics_vae_test_step(ics_net* net, const float* x, const float* cond, const float* eps, int batch,
                      float metrics[4]);
int ics_vae_upload_batch(ics_net* net, const float* x, const float* cond, const float* eps, int batch);
int ics_vae_train_step_resident(ics_net* net, float metrics_or_null[4]);

/* Fused inference tail (generate.py:204-225, eval.py:163-175): decoder.predict([z, cond]) ->
 * unet.model.predict -> np.argmax(soft, -1) / (sig >= thresh), device-resident end to end (the
 * reconstruction is never copied to the host).  `unet` is any U-Net engine with the VAE's grid and
 * channel count.  Outputs, each optional (NULL to skip): species/mask uint8 (B,d,d,d); density = channel 0
 * of the reconstruction, float (B,d,d,d) (watershed input, generate.py:232); coord_minmax float (B,3,2) =
 * per-sample {min,max} of channels 1..3 -- all that to_lattice_params (utils.py:160-178) reads of the
 * coordinate channels (zeros when in_channels == 1). */
int ics_vae_decode_to_unet_labels(ics_net* vae, ics_net* unet, const float* z, const float* cond, int batch,
                                  float thresh, uint8_t* species, uint8_t* mask, float* density,
                                  float* coord_minmax);

/* Connected-component post-processing on the device: the integer part of watershed_clustering
 * (/root/reference/watershed.py:190-203) that follows the fused tail in generate.py:228-236 / eval.py.
 *   - segment_nuclei step 1 (watershed.py:52-56): measure.label(binary, connectivity=1) = 6-connected components
 *     numbered in raster order of their first voxel; components with <= min_voxels voxels are dropped (reference: 3);
 *   - R as segment_nuclei returns it when every kept component passes `convexity >= min_convexity`
 *     (watershed.py:85-92): kept components renumbered 1..n in ascending label order, 0 elsewhere;
 *   - centroids / majority_vote (watershed.py:153-187) per region: the most frequent non-zero species (equal counts:
 *     the larger id, as the reference's stable sort leaves it) and the integer coordinate sums of ALL its voxels.
 * The convexity test (watershed.py:80) and the split of non-convex components (watershed.py:95-150) continue from
 * these regions, counts and bounding boxes: ics_op_label_boxes / ics_op_watershed_split below, driven by
 * icsg3d_amd/watershed.py (the recursion and the Qhull convexity test run on the host, as in the reference).
 * Outputs: regions int32 (B,d,d,d) or NULL; counts int32 (B,2) = {components, kept components};
 * atom_stats int32 (B,max_atoms,11) = {species, voxels, sum_z, sum_y, sum_x, z0, y0, x0, z1, y1, x1} (axes 0,1,2 of the
 * volume -- the reference calls them x,y,z; bounding box half-open), rows >= kept components are empty.
 * A sample with more than max_atoms kept components reports counts[b][1] > max_atoms: its rows are truncated and the
 * caller must skip it (the reference skips a failed sample too, generate.py:228-236); the call itself succeeds.
 * convexity_bounds int64 (B,max_atoms,8) or NULL = {P, sum zz, sum yy, sum xx, sum zy, sum zx, sum yx, 0} per region:
 * P = the number of grid points of the region's bounding box inside its 26-direction polytope (directions in {-1,0,1}^3,
 * the voxels offset by +-0.5 along one axis at a time as convex_hull_image does).  The convex hull is a subset of that
 * polytope, so voxels / P is a LOWER bound of the convexity watershed.py:80-83 tests: where it reaches min_convexity no
 * hull is needed; the second moments decide coplanarity (the 3 x 3 scatter matrix is singular), i.e. the components for
 * which the reference stack's Qhull call fails.  Exact integer arithmetic (icsg3d_amd/watershed.py: refine_atoms).
 * Integer atomics only: results are bit-exact. */
int ics_op_segment_atoms(const uint8_t* mask, const uint8_t* species, int batch, int d, int min_voxels, int max_atoms,
                         int num_species, int32_t* regions, int32_t* counts, int32_t* atom_stats,
                         int64_t* convexity_bounds);
/* ---- segment_nuclei's non-convex branch and recursion (watershed.py:40-150) on small boxes.  A box is a dense int32
 * volume [D][H][W] (extents 1..64) -- a component cropped to its bounding box, or a watershed result that is segmented
 * again; `nbox` boxes lie back to back in `vols` / `boxes`, dims = [nbox][3].  HOST pointers; one workgroup per box.
 * The scikit-image 0.17.2 routines these replace (requirements.txt:95) are absent from the image: PARITY UNPINNED, the
 * algorithms are restated in oracle/watershed_ref.py and the kernels are held to that restatement bit for bit.
 *   ics_op_label_boxes   = skimage.measure.label(box, connectivity) (watershed.py:52, :103): maximal sets of voxels of
 *     EQUAL non-zero value, numbered from 1 in raster order of their first voxel; connectivity 1 = 6 neighbours,
 *     3 = 26.  nlabels[nbox]; stats (or NULL) [nbox][max_labels][7] = {voxels, z0, y0, x0, z1, y1, x1} (half-open).
 *   ics_op_watershed_split = watershed.py:95-110 for a box with values {0, cls[b]}: fg / bg = erosion / dilation with
 *     ball(1) (out-of-box neighbours ignored), markers = label(fg) + 1 with `markers[(bg - fg) == 1] = 0` (the
 *     reference compares against 1, so the shell opens only for the component labelled 1), the priority flood of
 *     segmentation.watershed (two image levels; a binary heap on (level, age) with skimage's tie behaviour when
 *     tie = 0, FIFO when tie = 1), `wss[wss == 1] = 0`.  wss: labels 2.. or 0, before the reference's max_class shift. */
int ics_op_label_boxes(const int32_t* vols, const int32_t* dims, int nbox, int connectivity, int max_labels,
                       int32_t* labels, int32_t* nlabels, int32_t* stats);
int ics_op_watershed_split(const int32_t* boxes, const int32_t* dims, const int32_t* cls, int nbox, int tie, int32_t* wss);
/* Round 6: ics_op_watershed_split runs on HOST threads by default (one box per thread: the flood is a chain of dependent heap
 * operations, which one CPU core walks ~30x faster than one GPU lane and which cannot be spread over lanes without changing the
 * pop order the result depends on); ICSG3D_WS_DEVICE=1 selects the kernel.  Both are held to oracle/watershed_ref.py.
 * ics_op_component_bounds: the exact-integer bounds the convexity test of watershed.py:80-83 is decided from wherever they
 * are conclusive, for every component (labels 1..nlabels[b], more than min_voxels voxels) of the label volumes ics_op_label_boxes
 * returned (same dims / stats layout): bounds [nbox][max_labels][5] = {voxels, P, F, flat, H} with P >= count_nonzero(
 * convex_hull_image(component)) >= F -- P = bounding-box grid points inside the component's 26-direction polytope, F = the
 * component closed under axis-parallel line fills -- flat = 1 for coplanar / collinear components (the reference stack's
 * Qhull call fails there), and H = that count itself, computed exactly (integer gift wrapping over the +-0.5 offset voxel
 * set on doubled coordinates; a grid point on a facet counts as inside, as under the reference's 1e-10 tolerance) only where
 * hull_threshold > 0 and voxels / P < hull_threshold <= voxels / F, else 0.  Host threads; no device work. */
int ics_op_component_bounds(const int32_t* labels, const int32_t* dims, int nbox, const int32_t* nlabels, const int32_t* stats,
                            int max_labels, int min_voxels, double hull_threshold, int64_t* bounds);
/* The three box-level entry points below keep one stream and one grow-only device scratch buffer per host thread (they are
 * called tens of times per sample from the host recursion of segment_nuclei); this releases the calling thread's. */
int ics_release_caches(void);
/* centroids / majority_vote (watershed.py:153-187) for an arbitrary region volume R [D][H][W] (labels 1..num_labels, e.g.
 * what segment_nuclei returns after splits): stats [num_labels][11] as ics_op_segment_atoms; labels that do not occur
 * keep voxels = 0.  Pinned by the reference's own two functions (tests/golden/watershed_golden.npz). */
int ics_op_region_stats(const int32_t* R, const uint8_t* species, int D, int H, int W, int num_labels, int num_species,
                        int32_t* stats);
/* ics_vae_decode_to_unet_labels continued on the device through the component labelling above: the mask and species
 * volumes never leave HBM between the U-Net and the region statistics.  species/mask/density/coord_minmax/regions
 * and convexity_bounds are optional (NULL to skip). */
int ics_vae_decode_to_unet_atoms(ics_net* vae, ics_net* unet, const float* z, const float* cond, int batch,
                                 float thresh, int min_voxels, int max_atoms, uint8_t* species, uint8_t* mask,
                                 float* density, float* coord_minmax, int32_t* regions, int32_t* counts,
                                 int32_t* atom_stats, int64_t* convexity_bounds);

/* ---------------------------------------------------------------- common to both engines */
int ics_net_destroy(ics_net* net);
int ics_net_sync(ics_net* net);
/* model.save_weights / load_weights / get_weights (unet/unet.py:261-264,378; lattice_vae.py:149,339):
 * named tensors in Keras layouts (conv kernel (3,3,3,Cin,Cout), dense (in,out)); BatchNorm moving
 * statistics are the non-trainable entries. */
int ics_net_num_tensors(ics_net* net, int* count);
int ics_net_tensor_info(ics_net* net, int index, const char** name, int* ndim, int64_t dims[5],
                        int* trainable);
int ics_net_set_tensor(ics_net* net, const char* name, const float* host, size_t count);
int ics_net_get_tensor(ics_net* net, const char* name, float* host, size_t count);
/* gradient of the last train step w.r.t. a trainable tensor (parity tests) */
int ics_net_get_grad(ics_net* net, const char* name, float* host, size_t count);
/* stored pre-BatchNorm activation s = pre_act(conv+b) of a conv layer ("c1".."c18", "e0".., "d0"..)
 * from the most recent forward, (B*S^3*Cout) floats.  Parity tests use it to pin ReLU masks whose
 * pre-activation is within fp32 rounding of the kink. */
int ics_net_get_activation(ics_net* net, const char* layer, float* host, size_t count);
/* the fp32 BatchNorm affine (scale = gamma*rstd, shift = beta - mean*scale) the most recent forward
 * applied to that layer's output; lets a test reproduce the engine's max-pool routing bit-exactly. */
int ics_net_get_bn_affine(ics_net* net, const char* layer, float* scale, float* shift, size_t count);
/* Diagnosis aid (round 6): with ICSG3D_DEBUG_CANARY=1 in the environment when the handle is created, every device buffer of
 * the handle is followed by 8 KB of guard bytes; this reports how many guards were written to (kernels running past the end of
 * their buffer) and describes the first few in ics_last_error(). */
int ics_net_check_canaries(ics_net* net, int* dirty);
int ics_net_set_lr(ics_net* net, float lr);
/* optimizer step counter (Adam t) and reset of its moments */
int ics_net_reset_optimizer(ics_net* net);
/* Adam moments over the flat parameter buffer (same order as the trainable tensors) + step count: lets the
 * host keep the optimizer state of keras.optimizers.Adam (unet/unet.py:245) when it re-creates an engine
 * for a larger batch.  count = ics_net_num_params. */
int ics_net_num_params(ics_net* net, size_t* count);
int ics_net_get_optimizer_state(ics_net* net, float* m, float* v, size_t count, int* step);
int ics_net_set_optimizer_state(ics_net* net, const float* m, const float* v, size_t count, int step);

/* Per-kernel timing with HIP events on the engine's stream (bench.py roofline): enable, run steps,
 * then read back rows {label, launches, total_ms, total_flop, total_bytes}. */
/* Two engines of one process on ONE stream: `net` enqueues on `other`'s stream from now on (its own is kept for destruction),
 * so their steps alternate in program order with no events and no host waits.  New (the reference trains its two nets in
 * separate runs): joint U-Net + DFC-VAE training on one GPU.  `other` must outlive `net`'s use of it. */
int ics_net_share_stream(ics_net* net, ics_net* other);
/* Device-clock bracket on the engine's stream: start records an event; stop records another, waits for it and returns the
 * milliseconds between them (bench.py's gpu_active_s, the self-check of ms_per_step). */
int ics_net_timer_start(ics_net* net);
int ics_net_timer_stop(ics_net* net, double* ms);
/* Measurement aid (no reference counterpart): the resident train step eagerly and as a replayed
 * hipGraph, milliseconds per step over `iters` steps each, and the number of nodes the captured graph holds.  The replay
 * repeats the captured step's host-computed Adam step size: for timing, not for training. */
int ics_net_graph_probe(ics_net* net, int iters, double* eager_ms, double* graph_ms, int* graph_nodes);
int ics_net_profile_enable(ics_net* net, int on);
/* Restrict the events to the launch sites whose label starts with `prefix` -- or with one of several prefixes separated by
 * ';' -- (NULL / "" = every launch): bench.py times its K steps with events on the dominant kernel's launch sites only, so
 * that the timed region is the training job. */
int ics_net_profile_filter(ics_net* net, const char* prefix);
int ics_net_profile_count(ics_net* net, int* rows);
int ics_net_profile_row(ics_net* net, int row, const char** label, int64_t* launches, double* total_ms,
                        double* total_flop, double* total_bytes);

/* ---------------------------------------------------------------- data parallel (new)
 * One process per GPU; the flat fp32 gradient buffer is summed over the ranks in ~4 buckets (last layer
 * first) on a second HIP stream while the backward pass continues, and scaled by 1/nranks inside Adam.
 * Loss / metric numerators and denominators are all-reduced when a step returns metrics (every rank must
 * ask on the same steps).  uid is an ncclUniqueId (128 bytes) from rank 0. */
int ics_comm_unique_id(char uid[128]);
int ics_net_comm_init(ics_net* net, int rank, int nranks, const char uid[128]);
int ics_net_comm_allreduce_max(ics_net* net, double* value); /* barrier + max over ranks */
/* rank `root`'s parameters, BN moving statistics, Adam moments and step count overwrite everyone's:
 * replicas start identical (the class API initialises from an unseeded RNG).  No-op without a communicator. */
int ics_net_comm_broadcast_state(ics_net* net, int root);
/* 1: BatchNorm batch statistics (forward) and their gradient sums (backward) are exchanged over the
 * communicator, so N replicas x B grids normalise exactly like one process at N*B (the reference is
 * single-process: vae/lattice_vae.py:296, unet/unet.py:370).  0 (default): per-replica statistics; the
 * moving statistics are then averaged over the ranks after every step. */
int ics_net_set_sync_bn(ics_net* net, int on);
/* nranks = 0 without a communicator; buckets_last_step = gradient all-reduce messages of the last step */
int ics_net_comm_info(ics_net* net, int* rank, int* nranks, int* buckets_last_step);

/* ---------------------------------------------------------------- single-op entry points
 * (kernel parity tests against oracle/; host buffers, NDHWC) */
int ics_op_conv3d_forward(const float* x, const float* w, const float* bias, int B, int S, int Cin,
                          int Cout, int taps, int pre_act, float* y);
int ics_op_conv3d_backward(const float* x, const float* w, const float* dy, int B, int S, int Cin,
                           int Cout, int taps, float* dx, float* dw);

/* The two 1x1x1 heads with their losses and metrics on a GIVEN trunk output (kernel parity tests against the reference's own
 * formulas): soft = softmax(x wsoft + bsoft), sig = sigmoid(x wsig + bsig) (unet/unet.py:339-352), then
 * weighted_categorical_crossentropy / binary_crossentropy / f1_m / wr_m (unet/unet.py:159-221,252-259).
 * x [M][128], wsoft [128][ncls], bsoft [ncls], wsig [128], bsig [1], labels uint8 [M]; loss_weight <= 0: the scalar ncls.
 * mode 0: out [M][ncls+1] = probabilities (soft | sig); 1: metrics[5] and sums[7] (ics_unet_metric_sums) only;
 * 2: also out = dLoss/dlogits; + 4: binary_crossentropy in the logits form (ics_unet_config.bce_from_logits).  fused != 0: the GEMM inside the loss kernel (M % 16 == 0, ncls == 95), else GEMM + loss kernel. */
int ics_op_unet_head(const float* x, const float* wsoft, const float* bsoft, const float* wsig, const float* bsig,
                     const uint8_t* labels, size_t M, int ncls, float loss_weight, int mode, int fused, float* out,
                     float metrics[5], double sums[7]);
/* kernel micro-benchmark on device-resident constant data, HIP-event timed: mode 0 forward
 * (ablate 0 = full kernel, 1 = MFMA+LDS reads only, 2 = MFMA only), 1 backward-data, 2 backward-weight. */
int ics_op_conv3d_bench(int B, int S, int Cin, int Cout, int taps, int mode, int ablate, int iters,
                        float* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* ICSG3D_H */
