// Declarations for elementwise.hip (BatchNorm / pool / heads / losses / Adam launchers).
#pragma once
#include "common.h"

#include <rccl/rccl.h>

namespace ics {

struct BnParams {
  const float* gamma;
  const float* beta;
  float* moving_mean;
  float* moving_var;
  float* mean;    // batch (or moving, in eval) mean
  float* rstd;    // 1/sqrt(var+eps)
  float* scale;   // gamma*rstd
  float* shift;   // beta - mean*scale
  float* xs = nullptr;   // optional [2][C]: xhat as an affine of the stored activation (rstd, -mean*rstd); written by the
                         // training finalize without SyncBN only (conv_bnfuse_kernel's weight-gradient GEMM reads it)
};

// SyncBN exchange (opt-in data-parallel mode): `local` holds 3*C doubles (forward: n, mean, M2) or 2*C
// (backward: sum d, sum d*xhat); `gathered` nranks * 3*C.
struct BnSync {
  ncclComm_t comm;
  int nranks;
  double* local;
  double* gathered;
};

enum GradSrcKind : int { GS_NONE = 0, GS_DIRECT = 1, GS_UP = 2, GS_POOL = 3 };

// Where the gradient w.r.t. a layer's OUTPUT o comes from: the dA buffer of a consumer conv.
struct GradSrc {
  const float* p;                 // consumer's input-gradient buffer [rows][ld]
  int ld, off;                    // leading dimension and first channel of this layer inside it
  int kind;                       // GS_DIRECT same resolution; GS_UP consumer at 2S (sum 8 children);
                                  // GS_POOL consumer at S/2 behind MaxPool3D
  const float* pooled;            // GS_POOL: pooled forward values [rows/8][C]
  const unsigned char* pool_idx;  // GS_POOL: argmax index (first max) [rows/8][C]
};

// Everything needed to push a gradient back through  o = post_act(BN(s)), s = pre_act(conv + b).
struct LayerBwd {
  const float* s;                 // stored pre-BN activations [M][C]
  const float* scale; const float* shift; const float* mean; const float* rstd;
  const float* dtap;              // optional extra gradient w.r.t. s
  const float* tap_ref;           // perceptual tap: the same layer's s in the pass over y_true; adds
  float tap_coef;                 //   tap_coef * (s - tap_ref) to the gradient w.r.t. s and writes the block's
  double* tap_partial;            //   sum (s - tap_ref)^2 to tap_partial[blockIdx.x]  (bn_bwd_num_blocks entries)
  GradSrc g0, g1;
  int B, S, lgS, C;
  int has_bn, pre_act, post_act;
  int pool_ties_all;              // 1: TF-CPU MaxPool3DGrad tie rule, 0: first max only
  int flags;                      // ConvFlags of the owning handle
};

int launch_bn_finalize(hipStream_t st, const float* partial, int nblk, int Npad, const BnParams& bn,
                       int C, int update_moving, int unbias, const BnSync* sync = nullptr);
int launch_bn_eval_prepare(hipStream_t st, const BnParams& bn, int C);
// tie_mask / tie_ssum (optional, [rows/8][C]): bit k of a window's mask = element k receives the window's gradient in the
// backward pass (ties_all: all within kPoolTieTol of the maximum; else the first maximum); the sum of those elements'
// stored activations
int launch_pool_fwd(hipStream_t st, const float* s, const float* scale, const float* shift, int act,
                    int B, int S, int C, float* out, unsigned char* idx, unsigned char* tie_mask = nullptr,
                    float* tie_ssum = nullptr, int ties_all = 1);
// per-block (sum d, sum d*xhat) already produced by the backward-data launch that wrote dO (BwdStat in common.h):
// [2][ld][nblk]; nblk == 0 -> not available, run bn_bwd_reduce
struct BwdPre {
  const float* partial;
  int nblk, ld;
};
int launch_layer_bwd(hipStream_t st, const LayerBwd& L, float* dy, float* ws_partial, float* c1c2,
                     float* dgamma, float* dbeta, float* dbias, const BnSync* sync = nullptr,
                     const BwdPre* pre = nullptr, float* db_partial_own = nullptr, int* db_blocks = nullptr);
// deferred bias-gradient finalizes (out[c] = sum over nblk partial rows [nblk][C]), up to 24 per launch
__host__ __device__ inline int colsum_blocks(int C) { return C % 4 == 0 ? C / 4 : C; }   // blocks of colsum_batch_kernel per job
struct ColsumJobs {
  const float* partial[24];
  float* out[24];
  int nblk[24], C[24], blk0[25];
  int n;
};
int launch_colsum_batch(hipStream_t st, const ColsumJobs& J);
size_t layer_bwd_workspace_floats(const LayerBwd& L);
int bn_bwd_num_blocks(const LayerBwd& L, int* rows_per_block);
// perceptual-loss partial sums handed to vae_loss: n[l] partials for tap l, each tap's per-sample element count
// and layer weight (vae/lattice_vae.py:100-101,266-269)
struct PmSums {
  int n[4];
  double per[4];
  float w[4];
};
bool head_fused_ok(int ncls, int cin, size_t M, int act, int flags);     // flags: the handle's ConvFlags
bool head_dgrad_ok(int ncls, int cin, size_t M, const BwdStat* bs, int flags);
int launch_head_dgrad(hipStream_t st, const float* dz, const float* wsoft_k, const float* wsig_k, float* dx, int ldo, size_t M,
                      const BwdStat* bs, int Npad, int* blocks, const float* c1c2 = nullptr, float* db_partial = nullptr);
// the head -> c18 BatchNorm-backward fusion (round 4; elementwise.hip, head_dgrad_kernel<true>)
int launch_xhat_affine(hipStream_t st, const float* mean, const float* rstd, int C, float* xs);
int launch_head_bnfuse(hipStream_t st, const float* Q, const float* dzsum, const float* wsoft, const float* wsig,
                       const float* gamma, const float* beta, double n, int ncls, float* dwsoft, float* dwsig, float* c1c2,
                       float* dgamma, float* dbeta, const BnSync* sync = nullptr);
// BatchNorm-backward constants of a 3x3x3 layer's producer from the layer's own (xhat-sourced) weight-gradient GEMM:
// see conv_bnfuse_kernel
bool conv_bnfuse_ok(int S, int Cin, int N);
size_t conv_bnfuse_partial_floats(int B, int S, int C);
int launch_conv_bnfuse(hipStream_t st, const float* dy, int B, int S, int Cin, int CinTot, int N, const float* db_partial,
                       int db_nblk, const float* W, float* G, const float* gamma, const float* beta, const float* mean,
                       const float* rstd, const float* scale, float* abc, float* c1c2, float* dgamma, float* dbeta,
                       float* ws_partial, size_t ws_partial_floats, double* ws_R, double* sums_out = nullptr);
// second gradient source behind a MaxPool3D: its share of the two sums, added to sums_conv (launch_conv_bnfuse's sums_out)
// a layer whose only gradient source is its max-pool: the BatchNorm-backward sums on the pooled grid, as BwdPre partials
bool pool_presum_ok(int C, int ldg);
int pool_presum_blocks(size_t pooled_rows);
int launch_pool_presum(hipStream_t st, const float* g, int ldg, const unsigned char* mask, const float* ssum, size_t pooled_rows,
                       int C, const float* mean, const float* rstd, float* pre, size_t pre_floats, int* blocks);
size_t pool_bnfuse_partial_doubles(size_t pooled_rows, int C);
int launch_pool_bnfuse(hipStream_t st, const float* g, int ldg, const unsigned char* mask, const float* ssum, size_t pooled_rows,
                       int C, double cnt, const double* sums_conv, const float* mean, const float* rstd, const float* scale,
                       float* abc, float* c1c2, float* dgamma, float* dbeta, double* ws_partial, size_t ws_partial_doubles);
int launch_head_fused(hipStream_t st, const float* x, int ldx, const float* scale, const float* shift, const float* wsoft_k,
                      const float* wsig_k, const float* bsoft, const float* bsig, float* z, const unsigned char* labels,
                      size_t M, int mode, int want_grad, float wsoft, double* partial, int partial_blocks, float* metrics,
                      int* nblk_out = nullptr, float* dz_colsum = nullptr, double* keep = nullptr,
                      float thresh = 0.f, unsigned char* species = nullptr, unsigned char* mask = nullptr);
// mode 0: probabilities -> z; 1: loss / metrics (+ dz with want_grad); 2: uint8 argmax species + (sig >= thresh) mask only
int launch_head(hipStream_t st, float* z, int ldz, int ncls, const unsigned char* labels, size_t M,
                int mode, int want_grad, float wsoft, double* partial, int partial_blocks,
                float* metrics, int* nblk_out = nullptr, float* dz_colsum = nullptr, double* keep = nullptr);
// dz_colsum: optional [blocks][ncls+1] per-block column sums of the dz the kernel writes (head bias gradients)
// phase 0: reduce block partials + finalize; 1: reduce only -> sums[7]; 2: finalize from sums[7]
// keep: optional [7], receives the (global) sums the ratios were formed from in phases 0 and 2
int launch_head_metrics(hipStream_t st, const double* partial, int nblk, double M, float* metrics, double* sums,
                        int phase, double* keep = nullptr);
int launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, size_t n, float lr_t,
                float gscale);
int launch_bn_apply(hipStream_t st, const float* s, const float* scale, const float* shift, int act,
                    size_t n, int C, float* out);
int launch_sqdiff(hipStream_t st, const float* a, const float* b, int B, size_t per_sample,
                  int blocks_per_sample, double* partial, float* grad, float coef, int accumulate);
int launch_sampling(hipStream_t st, const float* mulv, int ld, int latent, const float* eps,
                    const float* cond, int ncond, int B, float* z, float* zc);
int launch_vae_loss(hipStream_t st, const float* mulv, int ld, int latent, int B, const double* mse_partial,
                    int n_mse, double n_elems, const double* pm_partial, const PmSums& pmc, float alpha, float beta,
                    float* metrics, double* sums = nullptr, int phase = 0);
int launch_vae_dz(hipStream_t st, const float* mulv, int ld, int latent, int B, const float* eps,
                  const float* dzc, int ldzc, float beta, float* dmulv);
// spatially constant input channels folded into a position-dependent bias / region sums (see elementwise.hip)
int launch_cond_bias_table(hipStream_t st, const float* w, const float* bias, const float* cond, int C, int ncond,
                           int Cin_tot, int Cout, int B, float* T);
size_t cond_wgrad_workspace_doubles(int B, int Cout);
int launch_cond_wgrad(hipStream_t st, const float* dy, int B, int S, int Cout, const float* cond, int C0, int nfold,
                      int ncond, int Cin_tot, float* dw, double* ws, size_t ws_doubles);
int launch_relu_bwd(hipStream_t st, const float* a, float* g, size_t n);
int launch_colsum_small(hipStream_t st, const float* a, int rows, int cols, int ld, float* out);
int launch_axpy(hipStream_t st, float* y, const float* x, size_t n, float a);
int launch_pool27(hipStream_t st, const float* dy, int B, int S, int N, float* out, int ldo);
int launch_permute_up_dw(hipStream_t st, const float* tmp, int Cu, int N, int Cin, int c_off, float* dw);

}  // namespace ics
