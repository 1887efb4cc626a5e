// HBM-bound kernels around the convolutions: BatchNorm statistics merge / backward, MaxPool3D,
// fused softmax+sigmoid heads with the U-Net losses and metrics, DFC-VAE loss terms, Adam.
// Semantics follow /root/reference/unet/unet.py:159-221,276-352 and
// /root/reference/vae/lattice_vae.py:53-66,232-270 plus the Keras op semantics of SURVEY.md App. B.
// All reductions are block partials + fixed-order merges (no float atomics): run-to-run bit-stable.
#include "common.h"
#include "elementwise.h"

namespace ics {

// ------------------------------------------------------------------------------------------
// block reduction helpers (256 threads)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// sum over the whole 256-thread block; result valid in every thread
__device__ __forceinline__ double block_sum_d(double v, double* sh /*[4]*/) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
// any block size up to 1024 threads (sh[16]); the waves' sums are added in wave order
__device__ __forceinline__ double block_sum_dn(double v, double* sh /*[16]*/) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = sh[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += sh[w];
  return r;
}

// ------------------------------------------------------------------------------------------
// BatchNorm forward statistics: merge per-block (count, mean, M2) -> mean/var -> scale/shift,
// moving-average update.  One block per channel, fp64 merge (Chan et al.).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                            int Npad, BnParams bn, int update_moving,
                                                            float momentum, int unbias) {
  __shared__ double sh[16];
  const int c = blockIdx.x;
  // partial layout [3][Npad][nblk]: this channel's (count, mean, M2) rows are contiguous.  ONE pass in fp64:
  // M2 = sum (M2_b + n_b m_b^2) - N mean^2 (the inputs are fp32: 29 spare bits for the cancellation), one trip
  // through the block reduction for the three sums -- the kernel sits between every convolution and its consumer.
  const float* pn = partial + (size_t)c * nblk;
  const float* pm = partial + ((size_t)Npad + c) * nblk;
  const float* pq = partial + ((size_t)2 * Npad + c) * nblk;
  double n = 0.0, s = 0.0, q = 0.0;
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
    const double nb = (double)pn[b], mb = (double)pm[b];
    n += nb;
    s += nb * mb;
    q += (double)pq[b] + nb * mb * mb;
  }
  n = wave_sum_d(n); s = wave_sum_d(s); q = wave_sum_d(q);
  __shared__ double sh3[3][16];
  if ((threadIdx.x & 63) == 0) { sh3[0][threadIdx.x >> 6] = n; sh3[1][threadIdx.x >> 6] = s; sh3[2][threadIdx.x >> 6] = q; }
  __syncthreads();
  n = sh3[0][0]; s = sh3[1][0]; q = sh3[2][0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { n += sh3[0][w]; s += sh3[1][w]; q += sh3[2][w]; }
  const double mean = s / n;
  double m2 = q - n * mean * mean;
  if (m2 < 0.0) m2 = 0.0;
  (void)sh;
  if (threadIdx.x == 0) {
    const double var = m2 / n;   // biased, as tf.nn.moments
    const float rstd = (float)(1.0 / sqrt(var + (double)kBnEps));
    const float sc = bn.gamma[c] * rstd;
    bn.mean[c] = (float)mean;
    bn.rstd[c] = rstd;
    bn.scale[c] = sc;
    bn.shift[c] = bn.beta[c] - (float)mean * sc;
    if (bn.xs) { bn.xs[c] = rstd; bn.xs[gridDim.x + c] = -(float)mean * rstd; }   // as xhat_affine_kernel
    if (update_moving) {
      // keras 2.3.1 BatchNormalization.call: variance fed to the moving average is rescaled by
      // n/(n-(1+eps)); plain EMA with momentum 0.99
      const double v = unbias ? var * (n / (n - (1.0 + (double)kBnEps))) : var;
      bn.moving_mean[c] = bn.moving_mean[c] * momentum + (float)mean * (1.f - momentum);
      bn.moving_var[c] = bn.moving_var[c] * momentum + (float)v * (1.f - momentum);
    }
  }
}

__global__ void bn_eval_prepare_kernel(BnParams bn, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.f / sqrtf(bn.moving_var[c] + kBnEps);
  const float sc = bn.gamma[c] * rstd;
  bn.mean[c] = bn.moving_mean[c];
  bn.rstd[c] = rstd;
  bn.scale[c] = sc;
  bn.shift[c] = bn.beta[c] - bn.moving_mean[c] * sc;
}

// ---- SyncBN (data parallel, opt-in): every rank merges its block partials into per-channel
// (n, mean, M2) in fp64, the ranks exchange those 3*C doubles with one all-gather, and each rank merges
// them in RANK ORDER (Chan et al.) -- the same result on every rank, equal to the statistics of the
// global batch.
__global__ __launch_bounds__(256) void bn_local_merge_kernel(const float* __restrict__ partial, int nblk,
                                                              int Npad, int C, double* __restrict__ out) {
  __shared__ double sh[4];
  const int c = blockIdx.x;
  double n = 0.0, s = 0.0;
  // partial layout [3][Npad][nblk]: this channel's (count, mean, M2) rows are contiguous
  const float* pn = partial + (size_t)c * nblk;
  const float* pm = partial + ((size_t)Npad + c) * nblk;
  const float* pq = partial + ((size_t)2 * Npad + c) * nblk;
  for (int b = threadIdx.x; b < nblk; b += 256) {
    n += (double)pn[b];
    s += (double)pn[b] * (double)pm[b];
  }
  n = block_sum_d(n, sh);
  s = block_sum_d(s, sh);
  const double mean = s / n;
  double m2 = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) {
    const double d = (double)pm[b] - mean;
    m2 += (double)pq[b] + (double)pn[b] * d * d;
  }
  m2 = block_sum_d(m2, sh);
  if (threadIdx.x == 0) { out[c] = n; out[C + c] = mean; out[2 * C + c] = m2; }
}
__global__ void bn_sync_finalize_kernel(const double* __restrict__ gathered, int nranks, int C, BnParams bn,
                                        int update_moving, float momentum, int unbias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int r = 0; r < nranks; ++r) {
    const double* g = gathered + (size_t)r * 3 * C;
    const double nb = g[c], mb = g[C + c], qb = g[2 * C + c];
    const double nt = n + nb, d = mb - mean;
    m2 += qb + d * d * n * nb / nt;
    mean += d * nb / nt;
    n = nt;
  }
  const double var = m2 / n;
  const float rstd = (float)(1.0 / sqrt(var + (double)kBnEps));
  const float sc = bn.gamma[c] * rstd;
  bn.mean[c] = (float)mean;
  bn.rstd[c] = rstd;
  bn.scale[c] = sc;
  bn.shift[c] = bn.beta[c] - (float)mean * sc;
  if (update_moving) {
    const double v = unbias ? var * (n / (n - (1.0 + (double)kBnEps))) : var;
    bn.moving_mean[c] = bn.moving_mean[c] * momentum + (float)mean * (1.f - momentum);
    bn.moving_var[c] = bn.moving_var[c] * momentum + (float)v * (1.f - momentum);
  }
}

int launch_bn_finalize(hipStream_t st, const float* partial, int nblk, int Npad, const BnParams& bn,
                       int C, int update_moving, int unbias, const BnSync* sync) {
  if (sync != nullptr) {
    ICS_LAUNCH(bn_local_merge_kernel, dim3(C), dim3(256), 0, st, partial, nblk, Npad, C, sync->local);
    ICS_HIP(hipGetLastError());
    ncclResult_t r = ncclAllGather(sync->local, sync->gathered, (size_t)3 * C, ncclDouble, sync->comm, st);
    ICS_CHECK(r == ncclSuccess, std::string("ncclAllGather(SyncBN): ") + ncclGetErrorString(r));
    ICS_LAUNCH(bn_sync_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, st, sync->gathered, sync->nranks, C,
                       bn, update_moving, 0.99f, unbias);
    ICS_HIP(hipGetLastError());
    return 0;
  }
  ICS_LAUNCH(bn_finalize_kernel, dim3(C), dim3(nblk >= 2048 ? 1024 : 256), 0, st, partial, nblk, Npad, bn,
                     update_moving, 0.99f, unbias);
  ICS_HIP(hipGetLastError());
  return 0;
}
int launch_bn_eval_prepare(hipStream_t st, const BnParams& bn, int C) {
  ICS_LAUNCH(bn_eval_prepare_kernel, dim3((C + 255) / 256), dim3(256), 0, st, bn, C);
  ICS_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// MaxPool3D(2) over o = act(s*scale+shift)   (pooling follows BN: unet.py:282, lattice_vae.py:176)
// ------------------------------------------------------------------------------------------
__global__ void pool_fwd_kernel(const float* __restrict__ s, const float* __restrict__ scale,
                                const float* __restrict__ shift, int act, int B, int S, int C,
                                float* __restrict__ out, unsigned char* __restrict__ idx) {
  const int Sh = S >> 1;
  const size_t total = (size_t)B * Sh * Sh * Sh * C;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = i % C;
  size_t v = i / C;
  const int x = v % Sh; v /= Sh;
  const int y = v % Sh; v /= Sh;
  const int z = v % Sh;
  const int b = v / Sh;
  const float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f;
  float best = -INFINITY;
  int bi = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int zz = 2 * z + (k >> 2), yy = 2 * y + ((k >> 1) & 1), xx = 2 * x + (k & 1);
    const float o = act_fwd(fmaf(s[((((size_t)b * S + zz) * S + yy) * S + xx) * C + c], sc, sh), act);
    if (o > best) { best = o; bi = k; }   // first maximum in (dz,dy,dx) scan order
  }
  out[i] = best;
  idx[i] = (unsigned char)bi;
}

// four channels per thread (C % 4 == 0): eight 16-byte loads in flight, one 16-byte and one 4-byte store
__global__ __launch_bounds__(256) void pool_fwd4_kernel(const float* __restrict__ s, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int act, int B, int S, int C,
                                                         float* __restrict__ out, unsigned char* __restrict__ idx,
                                                         unsigned char* __restrict__ tie_mask,
                                                         float* __restrict__ tie_ssum, int ties_all) {
  typedef float qv4 __attribute__((ext_vector_type(4)));
  const int Sh = S >> 1, C4 = C >> 2;
  const size_t total = (size_t)B * Sh * Sh * Sh * C4;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C4) * 4;
  size_t v = i / C4;
  const int x = v % Sh; v /= Sh;
  const int y = v % Sh; v /= Sh;
  const int z = v % Sh;
  const int b = (int)(v / Sh);
  qv4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (scale) { sc = *reinterpret_cast<const qv4*>(scale + c); sh = *reinterpret_cast<const qv4*>(shift + c); }
  qv4 q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int zz = 2 * z + (k >> 2), yy = 2 * y + ((k >> 1) & 1), xx = 2 * x + (k & 1);
    q[k] = *reinterpret_cast<const qv4*>(s + ((((size_t)b * S + zz) * S + yy) * S + xx) * C + c);
  }
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  unsigned bi[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float o = act_fwd(fmaf(q[k][j], sc[j], sh[j]), act);
      if (o > best[j]) { best[j] = o; bi[j] = (unsigned)k; }   // first maximum in (dz,dy,dx) scan order
    }
  *reinterpret_cast<qv4*>(out + i * 4) = qv4{best[0], best[1], best[2], best[3]};
  *reinterpret_cast<unsigned*>(idx + i * 4) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
  if (tie_mask != nullptr) {
    // which window elements the backward pass routes the gradient to (bit k), and the sum of their stored activations:
    // what gather_do / the fast BatchNorm-backward kernels decide per voxel, once per window (conv_wino64.hip FOLD = 3,
    // pool_sums_kernel).  ties_all: every element within kPoolTieTol of the maximum (TF-CPU MaxPool3DGrad), else the first.
    unsigned mk[4] = {0, 0, 0, 0};
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float o = act_fwd(fmaf(q[k][j], sc[j], sh[j]), act);
        const bool hit = ties_all ? fabsf(o - best[j]) < kPoolTieTol : bi[j] == (unsigned)k;
        if (hit) { mk[j] |= 1u << k; ss[j] += q[k][j]; }
      }
    *reinterpret_cast<unsigned*>(tie_mask + i * 4) = mk[0] | (mk[1] << 8) | (mk[2] << 16) | (mk[3] << 24);
    *reinterpret_cast<qv4*>(tie_ssum + i * 4) = qv4{ss[0], ss[1], ss[2], ss[3]};
  }
}

int launch_pool_fwd(hipStream_t st, const float* s, const float* scale, const float* shift, int act,
                    int B, int S, int C, float* out, unsigned char* idx, unsigned char* tie_mask, float* tie_ssum,
                    int ties_all) {
  if (C % 4 == 0 && ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(idx)) & 15) == 0) {
    const size_t total4 = (size_t)B * (S / 2) * (S / 2) * (S / 2) * (C / 4);
    ICS_LAUNCH(pool_fwd4_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, s, scale, shift, act, B,
                       S, C, out, idx, tie_mask, tie_ssum, ties_all);
    ICS_HIP(hipGetLastError());
    return 0;
  }
  ICS_CHECK(tie_mask == nullptr, "pool forward: the tie masks need the four-channel kernel");
  const size_t total = (size_t)B * (S / 2) * (S / 2) * (S / 2) * C;
  ICS_LAUNCH(pool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s, scale,
                     shift, act, B, S, C, out, idx);
  ICS_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Backward through  o = post_act(BN(s)),  s = pre_act(conv):  gather dO from the consumers,
// reduce (sum d, sum d*xhat), then dy = pre_act'(s) * [ scale*(d - c1 - xhat*c2) + dtap ].
// ------------------------------------------------------------------------------------------
struct ElemCtx {
  int b, z, y, x, c;
};

// VW = 1 or 4 consecutive channels per thread (float4 global accesses when C % 4 == 0)
template <int VW>
struct Vec {
  float v[VW];
};
template <int VW>
__device__ __forceinline__ Vec<VW> vload(const float* p) {
  Vec<VW> r;
  if (VW == 4) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    r.v[0] = q.x; r.v[1 % VW] = q.y; r.v[2 % VW] = q.z; r.v[3 % VW] = q.w;
  } else {
    r.v[0] = p[0];
  }
  return r;
}
template <int VW>
__device__ __forceinline__ void vstore(float* p, const Vec<VW>& r) {
  if (VW == 4) *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1 % VW], r.v[2 % VW], r.v[3 % VW]);
  else p[0] = r.v[0];
}

template <int VW>
__device__ __forceinline__ Vec<VW> gather_do(const GradSrc& g, const LayerBwd& L, size_t row,
                                             const ElemCtx& e, const Vec<VW>& o) {
  Vec<VW> r;
#pragma unroll
  for (int k = 0; k < VW; ++k) r.v[k] = 0.f;
  if (g.kind == GS_DIRECT) return vload<VW>(g.p + row * g.ld + g.off + e.c);
  if (g.kind == GS_UP) {
    // consumer ran at 2S on the nearest-upsampled tensor: sum the 8 children (UpSampling3D bwd)
    const int S2 = L.S * 2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int zz = 2 * e.z + (k >> 2), yy = 2 * e.y + ((k >> 1) & 1), xx = 2 * e.x + (k & 1);
      const Vec<VW> q = vload<VW>(g.p + ((((size_t)e.b * S2 + zz) * S2 + yy) * S2 + xx) * g.ld + g.off + e.c);
#pragma unroll
      for (int j = 0; j < VW; ++j) r.v[j] += q.v[j];
    }
    return r;
  }
  if (g.kind == GS_POOL) {
    const int Sh = L.S >> 1;
    const size_t prow = (((size_t)e.b * Sh + (e.z >> 1)) * Sh + (e.y >> 1)) * Sh + (e.x >> 1);
    const Vec<VW> up = vload<VW>(g.p + prow * g.ld + g.off + e.c);
    if (L.pool_ties_all) {
      // TF CPU MaxPool3DGrad: every window element within 1e-5 of the max receives the gradient
      const Vec<VW> ymax = vload<VW>(g.pooled + prow * L.C + e.c);
#pragma unroll
      for (int j = 0; j < VW; ++j) r.v[j] = fabsf(o.v[j] - ymax.v[j]) < kPoolTieTol ? up.v[j] : 0.f;
      return r;
    }
    const int k = ((e.z & 1) << 2) | ((e.y & 1) << 1) | (e.x & 1);
#pragma unroll
    for (int j = 0; j < VW; ++j) r.v[j] = g.pool_idx[prow * L.C + e.c + j] == k ? up.v[j] : 0.f;
    return r;
  }
  return r;
}

// d = dO * post_act'(bn_out);  also returns xhat
template <int VW>
__device__ __forceinline__ Vec<VW> elem_d(const LayerBwd& L, size_t row, const ElemCtx& e, const Vec<VW>& sv,
                                          Vec<VW>* xhat) {
  Vec<VW> bnout = sv, xh, o;
#pragma unroll
  for (int j = 0; j < VW; ++j) xh.v[j] = 0.f;
  if (L.has_bn) {
    const Vec<VW> sc = vload<VW>(L.scale + e.c), sh = vload<VW>(L.shift + e.c);
    const Vec<VW> mu = vload<VW>(L.mean + e.c), rs = vload<VW>(L.rstd + e.c);
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      bnout.v[j] = fmaf(sv.v[j], sc.v[j], sh.v[j]);
      xh.v[j] = (sv.v[j] - mu.v[j]) * rs.v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VW; ++j) o.v[j] = act_fwd(bnout.v[j], L.post_act);
  Vec<VW> d = gather_do<VW>(L.g0, L, row, e, o);
  if (L.g1.kind != GS_NONE) {
    const Vec<VW> d1 = gather_do<VW>(L.g1, L, row, e, o);
#pragma unroll
    for (int j = 0; j < VW; ++j) d.v[j] += d1.v[j];
  }
#pragma unroll
  for (int j = 0; j < VW; ++j) d.v[j] *= act_grad(bnout.v[j], L.post_act);
  *xhat = xh;
  return d;
}

__device__ __forceinline__ ElemCtx decode_elem(size_t row, int c, int S, int lg) {
  ElemCtx e;
  e.c = c;
  e.x = row & (S - 1);
  e.y = (row >> lg) & (S - 1);
  e.z = (row >> (2 * lg)) & (S - 1);
  e.b = row >> (3 * lg);
  return e;
}

// Thread layout for [M][C] tensors with power-of-two C: CB = min(C/VW, 256) column groups across
// threads, RPP = 256/CB rows per pass; column groups of 256*VW looped when C/VW > 256.
template <int VW>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(LayerBwd L, int rows_per_block,
                                                             float* __restrict__ partial) {
  __shared__ float sh1[256 * VW], sh2[256 * VW];
  const int C = L.C, CV = C / VW, CB = CV < 256 ? CV : 256, RPP = 256 / CB;
  const int t = threadIdx.x, tc = t % CB, tr = t / CB;
  const size_t M = (size_t)L.B << (3 * L.lgS);
  const size_t r0 = (size_t)blockIdx.x * rows_per_block;
  const size_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  for (int cg = 0; cg < CV; cg += 256) {
    const int c = (cg + tc) * VW;
    float a1[VW], a2[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) a1[j] = a2[j] = 0.f;
    for (size_t row = r0 + tr; row < r1; row += RPP) {
      const ElemCtx e = decode_elem(row, c, L.S, L.lgS);
      Vec<VW> xh;
      const Vec<VW> d = elem_d<VW>(L, row, e, vload<VW>(L.s + row * C + c), &xh);
#pragma unroll
      for (int j = 0; j < VW; ++j) { a1[j] += d.v[j]; a2[j] += d.v[j] * xh.v[j]; }
    }
#pragma unroll
    for (int j = 0; j < VW; ++j) { sh1[t * VW + j] = a1[j]; sh2[t * VW + j] = a2[j]; }
    __syncthreads();
    for (int col = t; col < CB * VW; col += 256) {
      // column `col` of this group: thread (r, tc = col / VW) holds it at [.. * VW + col % VW]
      const int tcc = col / VW, jj = col % VW;
      float s1 = 0.f, s2 = 0.f;
      for (int r = 0; r < RPP; ++r) { s1 += sh1[(r * CB + tcc) * VW + jj]; s2 += sh2[(r * CB + tcc) * VW + jj]; }
      // layout [2][C][blocks] (block index fastest): the finalize kernel reads a channel's partials contiguously
      partial[((size_t)0 * C + cg * VW + col) * gridDim.x + blockIdx.x] = s1;
      partial[((size_t)1 * C + cg * VW + col) * gridDim.x + blockIdx.x] = s2;
    }
    __syncthreads();
  }
}

// merges the block partials: c1 = sum d / n, c2 = sum d*xhat / n ; writes dgamma/dbeta if asked.
// sums != nullptr (SyncBN): the raw fp64 sums go to sums[0..C) / sums[C..2C) instead of c1/c2; after
// the all-reduce over ranks bn_bwd_sync_c_kernel divides by the GLOBAL element count.  dgamma/dbeta
// stay the local sums (the gradient all-reduce adds the ranks later).
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial,
                                                               int nblk, int C, int ld, double n,
                                                               float* __restrict__ c1,
                                                               float* __restrict__ c2,
                                                               float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta,
                                                               double* __restrict__ sums) {
  __shared__ double sh[16];
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  const float* p1 = partial + (size_t)c * nblk;                 // [2][ld][nblk]
  const float* p2 = partial + ((size_t)ld + c) * nblk;
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
    s1 += (double)p1[b];
    s2 += (double)p2[b];
  }
  s1 = block_sum_dn(s1, sh);
  s2 = block_sum_dn(s2, sh);
  if (threadIdx.x == 0) {
    if (sums) { sums[c] = s1; sums[C + c] = s2; }
    else { c1[c] = (float)(s1 / n); c2[c] = (float)(s2 / n); }
    if (dgamma) { dgamma[c] = (float)s2; dbeta[c] = (float)s1; }
  }
}
__global__ void bn_bwd_sync_c_kernel(const double* __restrict__ sums, int C, double n_global,
                                     float* __restrict__ c1, float* __restrict__ c2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  c1[c] = (float)(sums[c] / n_global);
  c2[c] = (float)(sums[C + c] / n_global);
}

// TAP: fused perceptual-tap loss term / gradient (compile-time: the fp64 accumulator and the extra load cost
// registers the plain variant, 97 % of the launches, must not pay for)
template <int VW, bool TAP>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(LayerBwd L, int rows_per_block,
                                                            const float* __restrict__ c1,
                                                            const float* __restrict__ c2,
                                                            float* __restrict__ dy,
                                                            float* __restrict__ db_partial) {
  __shared__ float sh1[256 * VW];
  __shared__ double shd[4];
  const int C = L.C, CV = C / VW, CB = CV < 256 ? CV : 256, RPP = 256 / CB;
  const int t = threadIdx.x, tc = t % CB, tr = t / CB;
  const size_t M = (size_t)L.B << (3 * L.lgS);
  const size_t r0 = (size_t)blockIdx.x * rows_per_block;
  const size_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  double tap_sq = 0.0;       // sum (s - tap_ref)^2 over this block's elements (perceptual loss term)
  for (int cg = 0; cg < CV; cg += 256) {
    const int c = (cg + tc) * VW;
    Vec<VW> k1, k2, sg;
#pragma unroll
    for (int j = 0; j < VW; ++j) { k1.v[j] = 0.f; k2.v[j] = 0.f; sg.v[j] = 1.f; }
    if (L.has_bn) { k1 = vload<VW>(c1 + c); k2 = vload<VW>(c2 + c); sg = vload<VW>(L.scale + c); }
    float acc[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) acc[j] = 0.f;
    for (size_t row = r0 + tr; row < r1; row += RPP) {
      const ElemCtx e = decode_elem(row, c, L.S, L.lgS);
      const Vec<VW> sv = vload<VW>(L.s + row * C + c);
      Vec<VW> xh, g;
      const Vec<VW> d = elem_d<VW>(L, row, e, sv, &xh);
      Vec<VW> tap;
#pragma unroll
      for (int j = 0; j < VW; ++j) tap.v[j] = 0.f;
      if (L.dtap) tap = vload<VW>(L.dtap + row * C + c);
      if (TAP) {
        // perceptual tap (vae/lattice_vae.py:257-270): d/ds of tap_coef/2 * (s - ref)^2, ref = the same layer's
        // activation in the pass over y_true; the loss term itself is summed on the way (no separate pass)
        const Vec<VW> rf = vload<VW>(L.tap_ref + row * C + c);
#pragma unroll
        for (int j = 0; j < VW; ++j) {
          const float df = sv.v[j] - rf.v[j];
          tap.v[j] += L.tap_coef * df;
          tap_sq += (double)df * (double)df;
        }
      }
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        float ds = L.has_bn ? sg.v[j] * (d.v[j] - k1.v[j] - xh.v[j] * k2.v[j]) : d.v[j];
        ds += tap.v[j];
        g.v[j] = ds * act_grad(sv.v[j], L.pre_act);
        acc[j] += g.v[j];
      }
      vstore<VW>(dy + row * C + c, g);
    }
    if (db_partial) {
#pragma unroll
      for (int j = 0; j < VW; ++j) sh1[t * VW + j] = acc[j];
      __syncthreads();
      for (int col = t; col < CB * VW; col += 256) {
        const int tcc = col / VW, jj = col % VW;
        float sacc = 0.f;
        for (int r = 0; r < RPP; ++r) sacc += sh1[(r * CB + tcc) * VW + jj];
        db_partial[(size_t)blockIdx.x * C + cg * VW + col] = sacc;
      }
      __syncthreads();
    }
  }
  if (TAP) {
    const double tot = block_sum_d(tap_sq, shd);
    if (threadIdx.x == 0) L.tap_partial[blockIdx.x] = tot;
  }
}

// Fast path of the apply pass (VW = 4, BatchNorm present, no extra tap gradient, consumers at the same resolution
// and/or behind the max-pool -- every U-Net layer and the plain VAE layers): the generic kernel above keeps ONE
// float4 pair per thread in flight at 4 waves/SIMD (126 VGPRs of run-time source dispatch), 32 KB per CU, and gives
// each block its own 256 KB region: 3.3 TB/s.  Here the source kinds are compile-time, the per-channel constants
// live in registers, UNR rows per thread are loaded before the first is used, and the blocks walk the tensor
// interleaved (pass p of block b = rows (b + p*gridDim)*RPP ...), so all resident blocks read one compact window.
// Element arithmetic is the generic kernel's (up to fma contraction); only the order of the per-block bias-gradient
// and tap-loss partial sums differs.
__device__ __forceinline__ float4 ldf4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int G0, int G1, bool TIES, int UNR, bool TAP = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_fast_kernel(LayerBwd L, const float* __restrict__ c1,
                                                                 const float* __restrict__ c2, float* __restrict__ dy,
                                                                 float* __restrict__ db_partial) {
  __shared__ float sh1[1024];
  __shared__ double shd[4];
  double tap_sq = 0.0;
  const int C = L.C, CB = C >> 2, RPP = 256 / CB;
  const int t = threadIdx.x, tc = t % CB, tr = t / CB, c = tc * 4;
  const size_t M = (size_t)L.B << (3 * L.lgS);
  const size_t npass = M / RPP, stride = gridDim.x;
  const float4 sc = ldf4(L.scale + c), shf = ldf4(L.shift + c), mu = ldf4(L.mean + c), rs = ldf4(L.rstd + c);
  const float4 k1 = ldf4(c1 + c), k2 = ldf4(c2 + c);
  const int post = L.post_act, pre = L.pre_act;
  const int lg = L.lgS, Sm = L.S - 1, Sh = L.S >> 1;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};

  auto body = [&](size_t p0, auto unr_tag) {
    constexpr int U = decltype(unr_tag)::value;
    float4 sv[U], d0[U], d1[U], ym[U], rf[U];
    unsigned pi[U];
    size_t row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      row[u] = (p0 + (size_t)u * stride) * RPP + tr;
      sv[u] = ldf4(L.s + row[u] * C + c);
      if (G0 == GS_DIRECT) {
        d0[u] = ldf4(L.g0.p + row[u] * L.g0.ld + L.g0.off + c);
      } else {   // GS_POOL
        const size_t r = row[u];
        const int x = (int)(r & Sm), y = (int)((r >> lg) & Sm), z = (int)((r >> (2 * lg)) & Sm);
        const size_t b = r >> (3 * lg);
        const size_t prow = ((b * Sh + (z >> 1)) * Sh + (y >> 1)) * Sh + (x >> 1);
        d0[u] = ldf4(L.g0.p + prow * L.g0.ld + L.g0.off + c);
        if (TIES) ym[u] = ldf4(L.g0.pooled + prow * C + c);
        else {
          pi[u] = *reinterpret_cast<const unsigned*>(L.g0.pool_idx + prow * C + c);
          const unsigned k = (unsigned)(((z & 1) << 2) | ((y & 1) << 1) | (x & 1));
          pi[u] ^= k * 0x01010101u;          // byte j == 0  <=>  this voxel is channel c+j's (first) maximum
        }
      }
      if (G1 == GS_DIRECT) d1[u] = ldf4(L.g1.p + row[u] * L.g1.ld + L.g1.off + c);
      if (TAP) rf[u] = ldf4(L.tap_ref + row[u] * C + c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float s4[4] = {sv[u].x, sv[u].y, sv[u].z, sv[u].w};
      const float a4[4] = {sc.x, sc.y, sc.z, sc.w}, b4[4] = {shf.x, shf.y, shf.z, shf.w};
      const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, r4[4] = {rs.x, rs.y, rs.z, rs.w};
      const float k14[4] = {k1.x, k1.y, k1.z, k1.w}, k24[4] = {k2.x, k2.y, k2.z, k2.w};
      const float g04[4] = {d0[u].x, d0[u].y, d0[u].z, d0[u].w};
      float g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bnout = fmaf(s4[j], a4[j], b4[j]);
        const float xh = (s4[j] - m4[j]) * r4[j];
        float d;
        if (G0 == GS_DIRECT) d = g04[j];
        else if (TIES) {
          const float y4[4] = {ym[u].x, ym[u].y, ym[u].z, ym[u].w};
          d = fabsf(act_fwd(bnout, post) - y4[j]) < kPoolTieTol ? g04[j] : 0.f;
        } else d = ((pi[u] >> (8 * j)) & 0xffu) == 0u ? g04[j] : 0.f;
        if (G1 == GS_DIRECT) {
          const float g14[4] = {d1[u].x, d1[u].y, d1[u].z, d1[u].w};
          d += g14[j];
        }
        d *= act_grad(bnout, post);
        float ds = a4[j] * (d - k14[j] - xh * k24[j]);
        if (TAP) {   // perceptual tap: see bn_bwd_apply_kernel<VW, true>
          const float r4f[4] = {rf[u].x, rf[u].y, rf[u].z, rf[u].w};
          const float df = s4[j] - r4f[j];
          ds += L.tap_coef * df;
          tap_sq += (double)df * (double)df;
        }
        g[j] = ds * act_grad(s4[j], pre);
        acc[j] += g[j];
      }
      *reinterpret_cast<float4*>(dy + row[u] * C + c) = make_float4(g[0], g[1], g[2], g[3]);
    }
  };
  size_t p = blockIdx.x;
  for (; p + (size_t)(UNR - 1) * stride < npass; p += (size_t)UNR * stride) body(p, std::integral_constant<int, UNR>{});
  for (; p < npass; p += stride) body(p, std::integral_constant<int, 1>{});

  if (db_partial) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sh1[t * 4 + j] = acc[j];
    __syncthreads();
    for (int col = t; col < C; col += 256) {
      const int tcc = col >> 2, jj = col & 3;
      float sacc = 0.f;
      for (int r = 0; r < RPP; ++r) sacc += sh1[(r * CB + tcc) * 4 + jj];
      db_partial[(size_t)blockIdx.x * C + col] = sacc;
    }
  }
  if (TAP) {
    const double tot = block_sum_d(tap_sq, shd);
    if (threadIdx.x == 0) L.tap_partial[blockIdx.x] = tot;
  }
}

// the reduce pass in the same shape (layers whose dO has two sources or sits behind the max-pool keep this pass;
// everywhere else the backward-data epilogue already summed these)
template <int G0, int G1, bool TIES, int UNR>
__global__ __launch_bounds__(256) void bn_bwd_reduce_fast_kernel(LayerBwd L, float* __restrict__ partial) {
  __shared__ float sh1[1024], sh2[1024];
  const int C = L.C, CB = C >> 2, RPP = 256 / CB;
  const int t = threadIdx.x, tc = t % CB, tr = t / CB, c = tc * 4;
  const size_t M = (size_t)L.B << (3 * L.lgS);
  const size_t npass = M / RPP, stride = gridDim.x;
  const float4 sc = ldf4(L.scale + c), shf = ldf4(L.shift + c), mu = ldf4(L.mean + c), rs = ldf4(L.rstd + c);
  const int post = L.post_act;
  const int lg = L.lgS, Sm = L.S - 1, Sh = L.S >> 1;
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};

  auto body = [&](size_t p0, auto unr_tag) {
    constexpr int U = decltype(unr_tag)::value;
    float4 sv[U], d0[U], d1[U], ym[U];
    unsigned pi[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t r = (p0 + (size_t)u * stride) * RPP + tr;
      sv[u] = ldf4(L.s + r * C + c);
      if (G0 == GS_DIRECT) {
        d0[u] = ldf4(L.g0.p + r * L.g0.ld + L.g0.off + c);
      } else {
        const int x = (int)(r & Sm), y = (int)((r >> lg) & Sm), z = (int)((r >> (2 * lg)) & Sm);
        const size_t b = r >> (3 * lg);
        const size_t prow = ((b * Sh + (z >> 1)) * Sh + (y >> 1)) * Sh + (x >> 1);
        d0[u] = ldf4(L.g0.p + prow * L.g0.ld + L.g0.off + c);
        if (TIES) ym[u] = ldf4(L.g0.pooled + prow * C + c);
        else {
          pi[u] = *reinterpret_cast<const unsigned*>(L.g0.pool_idx + prow * C + c);
          pi[u] ^= (unsigned)(((z & 1) << 2) | ((y & 1) << 1) | (x & 1)) * 0x01010101u;
        }
      }
      if (G1 == GS_DIRECT) d1[u] = ldf4(L.g1.p + r * L.g1.ld + L.g1.off + c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float s4[4] = {sv[u].x, sv[u].y, sv[u].z, sv[u].w};
      const float a4[4] = {sc.x, sc.y, sc.z, sc.w}, b4[4] = {shf.x, shf.y, shf.z, shf.w};
      const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, r4[4] = {rs.x, rs.y, rs.z, rs.w};
      const float g04[4] = {d0[u].x, d0[u].y, d0[u].z, d0[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float bnout = fmaf(s4[j], a4[j], b4[j]);
        const float xh = (s4[j] - m4[j]) * r4[j];
        float d;
        if (G0 == GS_DIRECT) d = g04[j];
        else if (TIES) {
          const float y4[4] = {ym[u].x, ym[u].y, ym[u].z, ym[u].w};
          d = fabsf(act_fwd(bnout, post) - y4[j]) < kPoolTieTol ? g04[j] : 0.f;
        } else d = ((pi[u] >> (8 * j)) & 0xffu) == 0u ? g04[j] : 0.f;
        if (G1 == GS_DIRECT) {
          const float g14[4] = {d1[u].x, d1[u].y, d1[u].z, d1[u].w};
          d += g14[j];
        }
        d *= act_grad(bnout, post);
        a1[j] += d;
        a2[j] += d * xh;
      }
    }
  };
  size_t p = blockIdx.x;
  for (; p + (size_t)(UNR - 1) * stride < npass; p += (size_t)UNR * stride) body(p, std::integral_constant<int, UNR>{});
  for (; p < npass; p += stride) body(p, std::integral_constant<int, 1>{});
#pragma unroll
  for (int j = 0; j < 4; ++j) { sh1[t * 4 + j] = a1[j]; sh2[t * 4 + j] = a2[j]; }
  __syncthreads();
  for (int col = t; col < C; col += 256) {
    const int tcc = col >> 2, jj = col & 3;
    float s1 = 0.f, s2 = 0.f;
    for (int r = 0; r < RPP; ++r) { s1 += sh1[(r * CB + tcc) * 4 + jj]; s2 += sh2[(r * CB + tcc) * 4 + jj]; }
    partial[((size_t)0 * C + col) * gridDim.x + blockIdx.x] = s1;
    partial[((size_t)1 * C + col) * gridDim.x + blockIdx.x] = s2;
  }
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial,
                                                               int nblk, int C, float* __restrict__ out) {
  __shared__ double sh[4];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += (double)partial[(size_t)b * C + c];
  s = block_sum_d(s, sh);
  if (threadIdx.x == 0) out[c] = (float)s;
}

int bn_bwd_num_blocks(const LayerBwd& L, int* rows_per_block) {
  const size_t M = (size_t)L.B << (3 * L.lgS);
  // aim for ~2048 blocks, at least 64 rows each
  size_t rpb = (M + 2047) / 2048;
  const size_t lo = M / 256 < 8 ? 8 : (M / 256 > 64 ? 64 : M / 256);   // S = 4 layers: 256 blocks, not 32
  if (rpb < lo) rpb = lo;
  const int VW = (L.C % 4 == 0) ? 4 : 1;
  const int CV = L.C / VW, CB = CV < 256 ? CV : 256, RPP = 256 / CB;
  rpb = (rpb + RPP - 1) / RPP * RPP;
  *rows_per_block = (int)rpb;
  return (int)((M + rpb - 1) / rpb);
}

int launch_layer_bwd(hipStream_t st, const LayerBwd& L, float* dy, float* ws_partial, float* c1c2,
                     float* dgamma, float* dbeta, float* dbias, const BnSync* sync, const BwdPre* pre,
                     float* db_partial_own, int* db_blocks) {
  int rpb;
  const int nblk = bn_bwd_num_blocks(L, &rpb);
  if (db_blocks) *db_blocks = nblk;
  const double n = (double)((size_t)L.B << (3 * L.lgS));
  float* c1 = c1c2;
  float* c2 = c1c2 + L.C;
  const bool v4 = (L.C % 4 == 0);
  const int cb = L.C >> 2;
  const bool fast = v4 && L.has_bn && !L.dtap && (!L.tap_ref || L.g1.kind == GS_NONE) && cb <= 256 && (cb & (cb - 1)) == 0 &&
                    ((size_t)n % (size_t)(256 / (cb ? cb : 1))) == 0 &&
                    (L.g0.kind == GS_DIRECT || L.g0.kind == GS_POOL) &&
                    (L.g1.kind == GS_NONE || L.g1.kind == GS_DIRECT) && !(L.flags & CF_NO_FAST_BNBWD);
  if (L.has_bn) {
    const float* red = ws_partial;
    int red_n = nblk, red_ld = L.C;
    if (pre != nullptr && pre->nblk > 0) {
      // the backward-data launch that produced dO already summed d and d*xhat per block (BwdStat)
      red = pre->partial; red_n = pre->nblk; red_ld = pre->ld;
    } else if (fast) {
#define ICS_BNR(G0_, G1_, T_, U_) \
      ICS_LAUNCH((bn_bwd_reduce_fast_kernel<G0_, G1_, T_, U_>), dim3(nblk), dim3(256), 0, st, L, ws_partial)
      const bool g1 = L.g1.kind == GS_DIRECT, ties = L.pool_ties_all != 0;
      if (L.g0.kind == GS_DIRECT) { if (g1) ICS_BNR(GS_DIRECT, GS_DIRECT, false, 4); else ICS_BNR(GS_DIRECT, GS_NONE, false, 4); }
      else if (ties) { if (g1) ICS_BNR(GS_POOL, GS_DIRECT, true, 2); else ICS_BNR(GS_POOL, GS_NONE, true, 2); }
      else { if (g1) ICS_BNR(GS_POOL, GS_DIRECT, false, 2); else ICS_BNR(GS_POOL, GS_NONE, false, 2); }
#undef ICS_BNR
      ICS_HIP(hipGetLastError());
    } else {
      if (v4) ICS_LAUNCH(bn_bwd_reduce_kernel<4>, dim3(nblk), dim3(256), 0, st, L, rpb, ws_partial);
      else ICS_LAUNCH(bn_bwd_reduce_kernel<1>, dim3(nblk), dim3(256), 0, st, L, rpb, ws_partial);
      ICS_HIP(hipGetLastError());
    }
    ICS_LAUNCH(bn_bwd_finalize_kernel, dim3(L.C), dim3(red_n >= 2048 ? 1024 : 256), 0, st, red, red_n, L.C, red_ld, n, c1,
                       c2, dgamma, dbeta, sync ? sync->local : nullptr);
    ICS_HIP(hipGetLastError());
    if (sync) {   // every rank holds the same number of rows (equal shards): n_global = n * nranks
      ncclResult_t r = ncclAllReduce(sync->local, sync->local, (size_t)2 * L.C, ncclDouble, ncclSum, sync->comm, st);
      ICS_CHECK(r == ncclSuccess, std::string("ncclAllReduce(SyncBN bwd): ") + ncclGetErrorString(r));
      ICS_LAUNCH(bn_bwd_sync_c_kernel, dim3((L.C + 63) / 64), dim3(64), 0, st, sync->local, L.C,
                         n * (double)sync->nranks, c1, c2);
      ICS_HIP(hipGetLastError());
    }
  }
  // db_partial_own: the bias-gradient partials go to the layer's own buffer and the caller finalizes them later, together
  // with every other layer's (launch_colsum_batch): one launch per step instead of one per layer
  float* dbp = dbias ? (db_partial_own ? db_partial_own : ws_partial) : nullptr;
  if (fast) {
    // M = B * S^3 with S >= 4 a power of two and RPP <= 64 rows per pass: whole passes
#define ICS_BNF(G0_, G1_, T_, U_) \
    ICS_LAUNCH((bn_bwd_apply_fast_kernel<G0_, G1_, T_, U_>), dim3(nblk), dim3(256), 0, st, L, c1, c2, dy, dbp)
    const bool g1 = L.g1.kind == GS_DIRECT, ties = L.pool_ties_all != 0;
    if (L.tap_ref) {                               // perceptual tap layers: behind the max-pool, or straight from a consumer
      if (L.g0.kind == GS_DIRECT) ICS_LAUNCH((bn_bwd_apply_fast_kernel<GS_DIRECT, GS_NONE, false, 4, true>), dim3(nblk), dim3(256), 0, st, L, c1, c2, dy, dbp);
      else if (ties) ICS_LAUNCH((bn_bwd_apply_fast_kernel<GS_POOL, GS_NONE, true, 2, true>), dim3(nblk), dim3(256), 0, st, L, c1, c2, dy, dbp);
      else ICS_LAUNCH((bn_bwd_apply_fast_kernel<GS_POOL, GS_NONE, false, 2, true>), dim3(nblk), dim3(256), 0, st, L, c1, c2, dy, dbp);
    }
    else if (L.g0.kind == GS_DIRECT) { if (g1) ICS_BNF(GS_DIRECT, GS_DIRECT, false, 4); else ICS_BNF(GS_DIRECT, GS_NONE, false, 4); }
    else if (ties) { if (g1) ICS_BNF(GS_POOL, GS_DIRECT, true, 2); else ICS_BNF(GS_POOL, GS_NONE, true, 2); }
    else { if (g1) ICS_BNF(GS_POOL, GS_DIRECT, false, 2); else ICS_BNF(GS_POOL, GS_NONE, false, 2); }
#undef ICS_BNF
  } else if (L.tap_ref) {
    if (v4) ICS_LAUNCH((bn_bwd_apply_kernel<4, true>), dim3(nblk), dim3(256), 0, st, L, rpb, c1, c2, dy, dbp);
    else ICS_LAUNCH((bn_bwd_apply_kernel<1, true>), dim3(nblk), dim3(256), 0, st, L, rpb, c1, c2, dy, dbp);
  } else {
    if (v4) ICS_LAUNCH((bn_bwd_apply_kernel<4, false>), dim3(nblk), dim3(256), 0, st, L, rpb, c1, c2, dy, dbp);
    else ICS_LAUNCH((bn_bwd_apply_kernel<1, false>), dim3(nblk), dim3(256), 0, st, L, rpb, c1, c2, dy, dbp);
  }
  ICS_HIP(hipGetLastError());
  if (dbias && !db_partial_own) {
    ICS_LAUNCH(colsum_finalize_kernel, dim3(L.C), dim3(256), 0, st, ws_partial, nblk, L.C, dbias);
    ICS_HIP(hipGetLastError());
  }
  return 0;
}
// every pending bias-gradient finalize of a step in ONE launch: block -> (job, channel group) through the prefix table.
// A block owns 4 channels (one float4 of every partial row, 256 row groups) when the job's channel count allows, else one
// channel (colsum_blocks).  c17's partials of a Winograd backward-data launch are 8192 rows: the one-channel form (4 of
// every 64 bytes, 32 dependent iterations) took 28 us, 16 channels per block (128 iterations) 67.
__global__ __launch_bounds__(256) void colsum_batch_kernel(ColsumJobs J) {
  __shared__ double sh[4];
  __shared__ double sh4[4][4];
  int j = 0;
  while (j + 1 < J.n && (int)blockIdx.x >= J.blk0[j + 1]) ++j;
  const int C = J.C[j], nblk = J.nblk[j];
  const float* partial = J.partial[j];
  if (C % 4 == 0) {
    const int c0 = ((int)blockIdx.x - J.blk0[j]) * 4;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int b = threadIdx.x;
    for (; b + 768 < nblk; b += 1024) {           // four rows in flight per thread (same order of additions)
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(partial + (size_t)(b + 256 * u) * C + c0);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a0 += (double)v[u].x; a1 += (double)v[u].y; a2 += (double)v[u].z; a3 += (double)v[u].w; }
    }
    for (; b < nblk; b += 256) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)b * C + c0);
      a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
    }
    a0 = wave_sum_d(a0); a1 = wave_sum_d(a1); a2 = wave_sum_d(a2); a3 = wave_sum_d(a3);
    if ((threadIdx.x & 63) == 0) {
      const int w = threadIdx.x >> 6;
      sh4[w][0] = a0; sh4[w][1] = a1; sh4[w][2] = a2; sh4[w][3] = a3;
    }
    __syncthreads();
    if (threadIdx.x < 4) J.out[j][c0 + threadIdx.x] = (float)((sh4[0][threadIdx.x] + sh4[1][threadIdx.x]) +
                                                               (sh4[2][threadIdx.x] + sh4[3][threadIdx.x]));
    return;
  }
  const int c = (int)blockIdx.x - J.blk0[j];
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += (double)partial[(size_t)b * C + c];
  s = block_sum_d(s, sh);
  if (threadIdx.x == 0) J.out[j][c] = (float)s;
}
int launch_colsum_batch(hipStream_t st, const ColsumJobs& J) {
  if (J.n == 0) return 0;
  ICS_LAUNCH(colsum_batch_kernel, dim3((unsigned)J.blk0[J.n]), dim3(256), 0, st, J);
  ICS_HIP(hipGetLastError());
  return 0;
}
size_t layer_bwd_workspace_floats(const LayerBwd& L) {
  int rpb;
  const int nblk = bn_bwd_num_blocks(L, &rpb);
  return (size_t)nblk * 2 * L.C;
}

// ------------------------------------------------------------------------------------------
// U-Net heads: z[M][ldz] holds 95 softmax logits + 1 sigmoid logit per voxel (one 1x1x1 GEMM).
// One wave per voxel row; lane l owns columns l and l+64.
//   mode 0 (predict): overwrite z with probabilities.
//   mode 1 (train/test): accumulate losses + metrics; if want_grad overwrite z with dLoss/dz.
// weighted_categorical_crossentropy (unet.py:196-221) with scalar weight, keras binary_crossentropy,
// f1_m / wr_m (unet.py:159-193).
// ------------------------------------------------------------------------------------------
// Four voxels per wave: 16 lanes own one voxel's logits (classes sl, sl+16, ... : up to 8 per lane, ncls <= 128),
// so the three row reductions are 4 xor-shuffle steps inside a 16-lane group instead of a 64-lane wave reduction
// per voxel (the one-wave-per-voxel version was instruction-bound at 1.8 TB/s).
__global__ __launch_bounds__(256) void head_kernel(float* __restrict__ z, int ldz, int ncls,
                                                   const unsigned char* __restrict__ labels, size_t M,
                                                   int rows_per_block, int mode, int want_grad,
                                                   float wsoft, float inv_bv, double* __restrict__ partial,
                                                   float* __restrict__ dz_colsum) {
  constexpr int J = 8;
  // want_grad bit 1: binary_crossentropy in TF 2.1's logits form (tf.keras.backend.binary_crossentropy short-circuits to
  // sigmoid_cross_entropy_with_logits when the prediction's producer op is a Sigmoid -- SURVEY App. B, confidence M):
  // max(z, 0) - z t + log1p(exp(-|z|)), gradient sigmoid(z) - t with no clip kink.  Default: the clipped-probability form.
  const bool bce_z = (want_grad & 2) != 0;
  want_grad &= 1;
  __shared__ double shd[16][6];
  __shared__ float shc[4][132];
  float cacc[J], cacc_sig = 0.f;          // column sums of dz over this lane's voxels (head bias gradients)
#pragma unroll
  for (int j = 0; j < J; ++j) cacc[j] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sl = lane & 15, grp = lane >> 4;           // class slot, voxel slot inside the wave
  const int nj = (ncls + 15) >> 4;                     // class columns per lane actually used (6 for 95)
  const size_t r0 = (size_t)blockIdx.x * rows_per_block;
  const size_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  double a_ls = 0, a_lg = 0, a_tp = 0, a_pred = 0, a_tpw = 0, a_posw = 0;
  auto gsum = [](float v) {
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
  };
  auto gmax = [](float v) {
    v = fmaxf(v, __shfl_xor(v, 8)); v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 1));
    return v;
  };
  for (size_t rowb = r0 + (size_t)wave * 4; rowb < r1; rowb += 16) {
    const size_t row = rowb + grp;
    const bool rv = row < r1;
    float* zr = z + (rv ? row : r0) * ldz;
    float zz[J], p[J];
    float mxl = -INFINITY;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int c = sl + 16 * j;
      zz[j] = (j < nj && c < ncls) ? zr[c] : -INFINITY;
      mxl = fmaxf(mxl, zz[j]);
    }
    const float zsig = zr[ncls];
    const float mx = gmax(mxl);
    float sl_sum = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int c = sl + 16 * j;
      p[j] = (j < nj && c < ncls) ? expf(zz[j] - mx) : 0.f;
      sl_sum += p[j];
    }
    const float sum = gsum(sl_sum);
    float psl = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) { p[j] = p[j] / sum; psl += p[j]; }
    const float ps = 1.f / (1.f + expf(-zsig));
    if (mode == 0) {
      if (rv) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int c = sl + 16 * j;
          if (j < nj && c < ncls) zr[c] = p[j];
        }
        if (sl == 0) zr[ncls] = ps;
      }
      continue;
    }
    const int lab = rv ? labels[row] : 0;
    // renormalise (p /= sum p), clip, -w*log  (only the true class contributes)
    const float psum = gsum(psl);
    float mine = 0.f, cnt = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int c = sl + 16 * j;
      mine = (c == lab) ? p[j] : mine;
      // predicted positives: round(clip(p,0,1)) == 1  <=>  p > 0.5 (round-half-even: 0.5 -> 0)
      cnt += (j < nj && c < ncls && p[j] > 0.5f) ? 1.f : 0.f;
    }
    const float pt_raw = gsum(mine);
    const float npred = gsum(cnt);
    const float qt = pt_raw / psum;
    const bool inside = qt >= kKEps && qt <= 1.f - kKEps;
    const float qc = fminf(fmaxf(qt, kKEps), 1.f - kKEps);
    const float tsig = lab != 0 ? 1.f : 0.f;
    const bool inside_s = ps >= kKEps && ps <= 1.f - kKEps;
    const float pc = fminf(fmaxf(ps, kKEps), 1.f - kKEps);
    if (sl == 0 && rv) {
      a_ls += (double)(-wsoft * logf(qc));
      a_lg += bce_z ? (double)(fmaxf(zsig, 0.f) - zsig * tsig + log1pf(expf(-fabsf(zsig))))
                    : (double)(-(tsig * logf(pc) + (1.f - tsig) * logf(1.f - pc)));
      const bool hit = pt_raw > 0.5f;
      a_tp += hit ? 1.0 : 0.0;
      a_pred += (double)npred;
      if (lab != 0) { a_posw += 1.0; a_tpw += hit ? 1.0 : 0.0; }
    }
    if (want_grad && rv) {
      // d lsoft/dz = w*(p - y)/(B*V) when the true-class probability is not clipped, else 0
      const float gs = inside ? wsoft * inv_bv : 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int c = sl + 16 * j;
        const float dzv = gs * (p[j] - (c == lab ? 1.f : 0.f));
        if (j < nj && c < ncls) { zr[c] = dzv; cacc[j] += dzv; }
      }
      const float dzs = (inside_s || bce_z) ? (ps - tsig) * inv_bv : 0.f;
      if (sl == 0) { zr[ncls] = dzs; cacc_sig += dzs; }
    }
  }
  if (mode == 0) return;
  if (want_grad && dz_colsum != nullptr) {
    // per-block column sums of dz (soft | sig bias gradients): 4 voxel slots x 4 waves, fixed order
#pragma unroll
    for (int j = 0; j < J; ++j) { cacc[j] += __shfl_xor(cacc[j], 16); cacc[j] += __shfl_xor(cacc[j], 32); }
    cacc_sig += __shfl_xor(cacc_sig, 16); cacc_sig += __shfl_xor(cacc_sig, 32);
    if (grp == 0) {
#pragma unroll
      for (int j = 0; j < J; ++j) shc[wave][sl + 16 * j] = cacc[j];
      if (sl == 0) shc[wave][128] = cacc_sig;
    }
    __syncthreads();
    if (threadIdx.x <= ncls) {
      const int c = threadIdx.x == ncls ? 128 : threadIdx.x;
      dz_colsum[(size_t)blockIdx.x * (ncls + 1) + threadIdx.x] = shc[0][c] + shc[1][c] + shc[2][c] + shc[3][c];
    }
  }
  if (sl == 0) {
    const int q = wave * 4 + grp;
    shd[q][0] = a_ls; shd[q][1] = a_lg; shd[q][2] = a_tp;
    shd[q][3] = a_pred; shd[q][4] = a_tpw; shd[q][5] = a_posw;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    double acc = 0.0;
    for (int q = 0; q < 16; ++q) acc += shd[q][k];
    partial[(size_t)blockIdx.x * 6 + k] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// Fused head: the 1x1x1 GEMM (128 -> 95 softmax + 1 sigmoid logits) AND head_kernel's loss / gradient / metrics in one
// pass over the trunk output -- the logits never go to memory (two launches, 0.61 ms and 1.75 GB of traffic per step
// at B = 32 before; one pass over 0.54 GB in, 0.40 GB out now).
//   * The GEMM runs TRANSPOSED on v_mfma_f32_16x16x4_f32: D[class][voxel] = sum_ch Wt[class][ch] * x[voxel][ch].  A wave
//     owns 16 voxels; a lane (n = lane & 15, g = lane >> 4) ends up with the 24 logits 16 t + 4 g + r (t < 6, r < 4) of
//     ITS voxel n: the softmax reductions are 24 in-lane values + two xor-shuffles (16, 32), and dz leaves as six
//     float4 stores.  The B operand is the voxel's own row, read straight from global memory (a 16-byte load = the
//     operand of four k-steps); only the weights sit in LDS, class-major, pitch 132 (conflict-free ds_read_b128).
//   * The producer's BatchNorm affine is folded into the weights: W' = W * scale[ch], b' = b + sum_ch W[ch] * shift[ch]
//     (a 1x1x1 "convolution" has no padding, so this is exact algebra; rounding differs from the unfused path at 1e-7).
//   * Waves are independent (no barrier after the weights are staged): while one wave does its softmax the other
//     waves of the SIMD keep the matrix cores busy.  Loss arithmetic = head_kernel's, line for line.
// Requires ncls + 1 == 96, 128 input channels, rows % 16 == 0, no activation between the BatchNorm and the head.
// ------------------------------------------------------------------------------------------
typedef float hv4 __attribute__((ext_vector_type(4)));
constexpr int kHeadWP = 132;             // LDS pitch of a class row of Wt (floats)

__global__ __launch_bounds__(256, 2) void head_fused_kernel(const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ wsoft_k, const float* __restrict__ wsig_k,
                                                         const float* __restrict__ bsoft, const float* __restrict__ bsig,
                                                         float* __restrict__ z, const unsigned char* __restrict__ labels,
                                                         int ntiles, int mode, int want_grad, float wsoft, float inv_bv,
                                                         double* __restrict__ partial, float* __restrict__ dz_colsum,
                                                         float thresh, unsigned char* __restrict__ species,
                                                         unsigned char* __restrict__ mask) {
  constexpr int NC = 95, NZ = 96, CH = 128;
  const bool bce_z = (want_grad & 2) != 0;                 // binary_crossentropy in the logits form (see head_kernel)
  want_grad &= 1;
  __shared__ __attribute__((aligned(16))) float Wt[NZ * kHeadWP];
  __shared__ __attribute__((aligned(16))) float biasp[NZ];
  __shared__ float shc[4][NZ];
  __shared__ double shd[4][6];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NZ * CH; i += 256) {
    const int cls = i >> 7, ch = i & 127;
    const float w = cls < NC ? wsoft_k[ch * NC + cls] : wsig_k[ch];
    Wt[cls * kHeadWP + ch] = scale ? w * scale[ch] : w;
  }
  if (tid < NZ) {
    float b = tid < NC ? bsoft[tid] : bsig[0];
    if (shift) {
      for (int ch = 0; ch < CH; ++ch) b += (tid < NC ? wsoft_k[ch * NC + tid] : wsig_k[ch]) * shift[ch];
    }
    biasp[tid] = b;
  }
  __syncthreads();

  const int n = lane & 15, g = lane >> 4;
  const int nwaves = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  float cacc[6][4];
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) cacc[t][r] = 0.f;
  double a_ls = 0, a_lg = 0, a_tp = 0, a_pred = 0, a_tpw = 0, a_posw = 0;
  auto xsum = [](float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; };
  auto xmax = [](float v) { v = fmaxf(v, __shfl_xor(v, 16)); v = fmaxf(v, __shfl_xor(v, 32)); return v; };

  hv4 xb[8];
  auto xload = [&](int tl) {
    const float* xr = x + ((size_t)tl * 16 + n) * ldx + 4 * g;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) xb[kk] = *reinterpret_cast<const hv4*>(xr + 16 * kk);
  };
  if (tile < ntiles) xload(tile);
  for (; tile < ntiles; tile += nwaves) {
    hv4 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) acc[t] = *reinterpret_cast<const hv4*>(&biasp[16 * t + 4 * g]);
    hv4 xc[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) xc[kk] = xb[kk];
    if (tile + nwaves < ntiles) xload(tile + nwaves);       // next tile's rows under this tile's work
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const hv4 a4 = *reinterpret_cast<const hv4*>(&Wt[(16 * t + n) * kHeadWP + 16 * kk + 4 * g]);
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[sidx], xc[kk][sidx], acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);                     // or the scheduler hoists all 48 weight reads (192 registers)
    }
    // ---- this lane: voxel `row`, logits of classes c(t, r) = 16 t + 4 g + r; c == 95 is the sigmoid logit
    const size_t row = (size_t)tile * 16 + n;
    float mxl = -INFINITY;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * t + 4 * g + r;
        if (c < NC) mxl = fmaxf(mxl, acc[t][r]);
      }
    const float zsig = __shfl(acc[5][3], 48 + n);            // class 95 lives in the g == 3 lane of the voxel
    const float mx = xmax(mxl);
    float p[6][4], sl_sum = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * t + 4 * g + r;
        p[t][r] = c < NC ? expf(acc[t][r] - mx) : 0.f;
        sl_sum += p[t][r];
      }
    const float sum = xsum(sl_sum);
    const float rsum = 1.f / sum;                            // one division per voxel (head_kernel divides per class)
    float psl = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { p[t][r] = p[t][r] * rsum; psl += p[t][r]; }
    const float ps = 1.f / (1.f + expf(-zsig));
    float* zr = z + row * NZ + 4 * g;
    if (mode == 2) {
      // generate.py:221-225 without the probabilities ever leaving the registers: np.argmax over the 95 class
      // probabilities (first maximum = lowest class among equals) and sig >= thresh, one byte each per voxel.
      // The same p values mode 0 stores, so the labels equal argmax / threshold of a mode-0 output bit for bit.
      float bv = -1.f;
      int bc = 0;
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * t + 4 * g + r;               // ascending within the lane
          if (c < NC && p[t][r] > bv) { bv = p[t][r]; bc = c; }
        }
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oc = __shfl_xor(bc, off);
        if (ov > bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
      }
      if (g == 0) {
        species[row] = (unsigned char)bc;
        mask[row] = ps >= thresh ? 1 : 0;
      }
      continue;
    }
    if (mode == 0) {
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        hv4 o = {p[t][0], p[t][1], p[t][2], p[t][3]};
        if (t == 5 && g == 3) o[3] = ps;
        *reinterpret_cast<hv4*>(zr + 16 * t) = o;
      }
      continue;
    }
    const int lab = labels[row];
    const float psum = xsum(psl);
    float mine = 0.f, cnt = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * t + 4 * g + r;
        mine = (c == lab) ? p[t][r] : mine;                  // lab < 95: the padding slot (p = 0) never matches a label
        cnt += (c < NC && p[t][r] > 0.5f) ? 1.f : 0.f;
      }
    const float pt_raw = xsum(mine);
    const float npred = xsum(cnt);
    const float qt = pt_raw / psum;
    const bool inside = qt >= kKEps && qt <= 1.f - kKEps;
    const float qc = fminf(fmaxf(qt, kKEps), 1.f - kKEps);
    const float tsig = lab != 0 ? 1.f : 0.f;
    const bool inside_s = ps >= kKEps && ps <= 1.f - kKEps;
    const float pc = fminf(fmaxf(ps, kKEps), 1.f - kKEps);
    if (g == 0) {
      a_ls += (double)(-wsoft * logf(qc));
      a_lg += bce_z ? (double)(fmaxf(zsig, 0.f) - zsig * tsig + log1pf(expf(-fabsf(zsig))))
                    : (double)(-(tsig * logf(pc) + (1.f - tsig) * logf(1.f - pc)));
      const bool hit = pt_raw > 0.5f;
      a_tp += hit ? 1.0 : 0.0;
      a_pred += (double)npred;
      if (lab != 0) { a_posw += 1.0; a_tpw += hit ? 1.0 : 0.0; }
    }
    if (want_grad) {
      const float gs = inside ? wsoft * inv_bv : 0.f;
      const float dzs = (inside_s || bce_z) ? (ps - tsig) * inv_bv : 0.f;
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        hv4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * t + 4 * g + r;
          float dzv = gs * (p[t][r] - (c == lab ? 1.f : 0.f));
          if (c == NC) dzv = dzs;
          o[r] = dzv;
          cacc[t][r] += dzv;
        }
        *reinterpret_cast<hv4*>(zr + 16 * t) = o;
      }
    }
  }
  if (mode == 0 || mode == 2) return;
  if (want_grad && dz_colsum != nullptr) {
    // per-block column sums of dz (soft | sig bias gradients): 16 voxel lanes, then the 4 waves, fixed order
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = cacc[t][r];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        if (n == 0) shc[wave][16 * t + 4 * g + r] = v;
      }
  }
  double am[6] = {a_ls, a_lg, a_tp, a_pred, a_tpw, a_posw};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double v = am[k];                                         // only the g == 0 lanes hold non-zero sums
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    if (lane == 0) shd[wave][k] = v;
  }
  __syncthreads();
  if (want_grad && dz_colsum != nullptr && tid < NZ)
    dz_colsum[(size_t)blockIdx.x * NZ + tid] = shc[0][tid] + shc[1][tid] + shc[2][tid] + shc[3][tid];
  if (tid < 6) partial[(size_t)blockIdx.x * 6 + tid] = shd[0][tid] + shd[1][tid] + shd[2][tid] + shd[3][tid];
}

bool head_fused_ok(int ncls, int cin, size_t M, int act, int flags) {
  return ncls == 95 && cin == 128 && M % 16 == 0 && act == ACT_NONE && !(flags & CF_NO_FUSED_HEAD);
}
int launch_head_fused(hipStream_t st, const float* x, int ldx, const float* scale, const float* shift, const float* wsoft_k,
                      const float* wsig_k, const float* bsoft, const float* bsig, float* z, const unsigned char* labels,
                      size_t M, int mode, int want_grad, float wsoft, double* partial, int partial_blocks, float* metrics,
                      int* nblk_out, float* dz_colsum, double* keep, float thresh, unsigned char* species,
                      unsigned char* mask) {
  ICS_CHECK(mode != 2 || (species != nullptr && mask != nullptr), "fused head, label mode: null output");
  ICS_CHECK(M % 16 == 0 && ldx % 4 == 0, "fused head: rows must come in sixteens, float4-aligned");
  const int ntiles = (int)(M / 16);
  int nblk = (ntiles + 3) / 4;
  const int cap = partial_blocks < 512 ? partial_blocks : 512;   // 2 workgroups per CU, whole rounds of tiles at B = 32
  if (nblk > cap) nblk = cap;
  ICS_LAUNCH(head_fused_kernel, dim3(nblk), dim3(256), 0, st, x, ldx, scale, shift, wsoft_k, wsig_k, bsoft, bsig, z,
                     labels, ntiles, mode, want_grad, wsoft, (float)(1.0 / (double)M), partial, dz_colsum, thresh, species, mask);
  ICS_HIP(hipGetLastError());
  if (nblk_out) *nblk_out = nblk;
  if (mode == 1 && metrics != nullptr) ICS_TRY(launch_head_metrics(st, partial, nblk, (double)M, metrics, nullptr, 0, keep));
  return 0;
}

// Backward-data of the same head in the same transposed shape: dX^T[ch][voxel] = sum_cls W[ch][cls] dZ^T[cls][voxel].
// The B operand is the voxel's own dz row from global memory (six 16-byte loads = 24 k-steps), the weights sit in LDS
// channel-major (pitch 100), a lane ends with the channels 16 c + 4 g + r of its voxel (eight float4 stores), and the
// producer's BatchNorm-backward sums (sum d, sum d * xhat per channel -- what bn_bwd_reduce would compute) are
// accumulated per lane over the wave's voxels and reduced once at the end: [2][Npad][blocks], block index fastest.
constexpr int kHeadDP = 100;
// BNF (round 4): the producer's whole BatchNorm-backward APPLY in the epilogue.  d = dz W^T is linear in dz, so the two
// batch sums the apply needs -- sum_v d and sum_v d * xhat -- follow from quantities the head already has (its bias and
// weight gradients: head_bnfuse_kernel below) BEFORE this kernel runs; the epilogue then writes
//   dy = relu'(s) * scale * (d - c1 - xhat * c2)          (Conv -> ReLU -> BN blocks, unet.py:276-278)
// straight into the producer's dy, and the per-block column sums of dy (its bias gradient) where the folded sums went:
// c18's separate BatchNorm-backward pass over 3 x 537 MB disappears.  c1c2: [2][128]; db_partial: [gridDim.x][128].
template <bool BNF>
__global__ __launch_bounds__(256, 2) void head_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ wsoft_k,
                                                            const float* __restrict__ wsig_k, float* __restrict__ dx,
                                                            int ldo, int ntiles, BwdStat bs, int Npad,
                                                            const float* __restrict__ c1c2, float* __restrict__ db_partial) {
  constexpr int NC = 95, NZ = 96, CH = 128;
  __shared__ __attribute__((aligned(16))) float Wc[CH * kHeadDP];
  __shared__ __attribute__((aligned(16))) float mu_s[CH], rs_s[CH];
  __shared__ __attribute__((aligned(16))) float sc_s[BNF ? CH : 4], k1_s[BNF ? CH : 4], k2_s[BNF ? CH : 4];
  __shared__ float fold[4][2][CH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool stat = BNF || bs.partial != nullptr;
  if (BNF && tid < CH) { sc_s[tid] = bs.scale[tid]; k1_s[tid] = c1c2[tid]; k2_s[tid] = c1c2[CH + tid]; }
  for (int i = tid; i < CH * NZ; i += 256) {
    const int ch = i / NZ, cls = i - ch * NZ;
    Wc[ch * kHeadDP + cls] = cls < NC ? wsoft_k[ch * NC + cls] : wsig_k[ch];
  }
  if (stat && tid < CH) { mu_s[tid] = bs.mean[tid]; rs_s[tid] = bs.rstd[tid]; }
  __syncthreads();
  const int n = lane & 15, g = lane >> 4;
  const int nwaves = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  hv4 f1[8], f2[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { f1[c] = hv4{0.f, 0.f, 0.f, 0.f}; f2[c] = hv4{0.f, 0.f, 0.f, 0.f}; }
  hv4 db[6];
  auto dload = [&](int tl) {
    const float* zr = dz + ((size_t)tl * 16 + n) * NZ + 4 * g;
#pragma unroll
    for (int t = 0; t < 6; ++t) db[t] = *reinterpret_cast<const hv4*>(zr + 16 * t);
  };
  if (tile < ntiles) dload(tile);
  for (; tile < ntiles; tile += nwaves) {
    const size_t row = (size_t)tile * 16 + n;
    hv4 dc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) dc[t] = db[t];
    if (tile + nwaves < ntiles) dload(tile + nwaves);
    hv4 sv[8];
    if (stat) {
      const float* sr = bs.s + row * bs.ld + 4 * g;
#pragma unroll
      for (int c = 0; c < 8; ++c) sv[c] = *reinterpret_cast<const hv4*>(sr + 16 * c);
    }
    hv4 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = hv4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 6; ++t) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const hv4 a4 = *reinterpret_cast<const hv4*>(&Wc[(16 * c + n) * kHeadDP + 16 * t + 4 * g]);
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[sidx], dc[t][sidx], acc[c], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float* xr = dx + row * ldo + 4 * g;
    if (BNF) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const hv4 mu = *reinterpret_cast<const hv4*>(&mu_s[16 * c + 4 * g]);
        const hv4 rs = *reinterpret_cast<const hv4*>(&rs_s[16 * c + 4 * g]);
        const hv4 sc = *reinterpret_cast<const hv4*>(&sc_s[16 * c + 4 * g]);
        const hv4 k1 = *reinterpret_cast<const hv4*>(&k1_s[16 * c + 4 * g]);
        const hv4 k2 = *reinterpret_cast<const hv4*>(&k2_s[16 * c + 4 * g]);
        hv4 gy;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float xh = (sv[c][r] - mu[r]) * rs[r];
          const float ds = sc[r] * (acc[c][r] - k1[r] - xh * k2[r]);
          gy[r] = sv[c][r] > 0.f ? ds : 0.f;           // ReLU'(s) on the stored (post-ReLU) activation, as act_grad does
        }
        *reinterpret_cast<hv4*>(xr + 16 * c) = gy;
        f1[c] += gy;
      }
    } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) *reinterpret_cast<hv4*>(xr + 16 * c) = acc[c];
    if (stat) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const hv4 mu = *reinterpret_cast<const hv4*>(&mu_s[16 * c + 4 * g]);
        const hv4 rs = *reinterpret_cast<const hv4*>(&rs_s[16 * c + 4 * g]);
        f1[c] += acc[c];
        f2[c] += acc[c] * ((sv[c] - mu) * rs);
      }
    }
    }
  }
  if (!stat) return;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = f1[c][r], b = f2[c][r];
      a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4); a += __shfl_xor(a, 8);
      b += __shfl_xor(b, 1); b += __shfl_xor(b, 2); b += __shfl_xor(b, 4); b += __shfl_xor(b, 8);
      if (n == 0) { fold[wave][0][16 * c + 4 * g + r] = a; fold[wave][1][16 * c + 4 * g + r] = b; }
    }
  __syncthreads();
  {
    const int which = tid >> 7, ch = tid & 127;
    const float v = fold[0][which][ch] + fold[1][which][ch] + fold[2][which][ch] + fold[3][which][ch];
    if (BNF) { if (which == 0) db_partial[(size_t)blockIdx.x * CH + ch] = v; }
    else bs.partial[((size_t)which * Npad + ch) * gridDim.x + blockIdx.x] = v;
  }
}

// xhat = (s - mean) * rstd as an affine of the stored activation: scale = rstd, shift = -mean * rstd
__global__ void xhat_affine_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, int C,
                                   float* __restrict__ xs) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  xs[c] = rstd[c];
  xs[C + c] = -mean[c] * rstd[c];
}
// Q[ch][cls] = sum_v xhat[v][ch] dz[v][cls] (the head's weight-gradient GEMM run on xhat instead of on gamma xhat + beta),
// dzsum[cls] = sum_v dz[v][cls] (the head's bias gradients).  Per channel:
//   head dW[ch][cls] = gamma Q + beta dzsum                           (what the GEMM on the BatchNorm output would have given)
//   sum_v d[v][ch]        = sum_cls W[ch][cls] dzsum[cls]  -> c1 = / n, the producer's dbeta
//   sum_v d xhat [v][ch]  = sum_cls W[ch][cls] Q[ch][cls]  -> c2 = / n, the producer's dgamma
// (class sums in fp64, fixed order: wave butterfly, then the two waves)
__global__ __launch_bounds__(128) void head_bnfuse_kernel(const float* __restrict__ Q, const float* __restrict__ dzsum,
                                                          const float* __restrict__ wsoft, const float* __restrict__ wsig,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          double n, int ncls, float* __restrict__ dwsoft,
                                                          float* __restrict__ dwsig, float* __restrict__ c1c2,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          double* __restrict__ sums) {
  // one block per channel, one thread per class (first version: one thread per channel walking 96 classes with a
  // 384-byte stride between lanes -- 52 us for 48 KB)
  __shared__ double sh1[2], sh2[2];
  const int ch = blockIdx.x, cls = threadIdx.x, nz = ncls + 1;
  const float g = gamma[ch], b = beta[ch];
  double pd = 0.0, pq = 0.0;
  if (cls < nz) {
    const float q = Q[ch * nz + cls], zs = dzsum[cls];
    const float w = cls < ncls ? wsoft[ch * ncls + cls] : wsig[ch];
    pd = (double)w * (double)zs;
    pq = (double)w * (double)q;
    const float dw = fmaf(g, q, b * zs);
    if (cls < ncls) dwsoft[ch * ncls + cls] = dw; else dwsig[ch] = dw;
  }
  pd = wave_sum_d(pd); pq = wave_sum_d(pq);
  if ((threadIdx.x & 63) == 0) { sh1[threadIdx.x >> 6] = pd; sh2[threadIdx.x >> 6] = pq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double sd = sh1[0] + sh1[1], sq = sh2[0] + sh2[1];
    if (sums) { sums[ch] = sd; sums[128 + ch] = sq; }       // SyncBN: all-reduced, then bn_bwd_sync_c_kernel forms c1 / c2
    else { c1c2[ch] = (float)(sd / n); c1c2[128 + ch] = (float)(sq / n); }
    dgamma[ch] = (float)sq;
    dbeta[ch] = (float)sd;
  }
}
int launch_xhat_affine(hipStream_t st, const float* mean, const float* rstd, int C, float* xs) {
  ICS_LAUNCH(xhat_affine_kernel, dim3((C + 127) / 128), dim3(128), 0, st, mean, rstd, C, xs);
  ICS_HIP(hipGetLastError());
  return 0;
}
int launch_head_bnfuse(hipStream_t st, const float* Q, const float* dzsum, const float* wsoft, const float* wsig,
                       const float* gamma, const float* beta, double n, int ncls, float* dwsoft, float* dwsig, float* c1c2,
                       float* dgamma, float* dbeta, const BnSync* sync) {
  ICS_CHECK(ncls + 1 <= 128, "head BN-fuse: at most 127 classes");
  ICS_LAUNCH(head_bnfuse_kernel, dim3(128), dim3(128), 0, st, Q, dzsum, wsoft, wsig, gamma, beta, n, ncls, dwsoft, dwsig, c1c2,
             dgamma, dbeta, sync ? sync->local : nullptr);
  ICS_HIP(hipGetLastError());
  if (sync) {   // global-batch statistics: the two sums over ALL ranks' voxels (every rank holds the same number of rows)
    ncclResult_t r = ncclAllReduce(sync->local, sync->local, (size_t)256, ncclDouble, ncclSum, sync->comm, st);
    ICS_CHECK(r == ncclSuccess, std::string("ncclAllReduce(SyncBN head fuse): ") + ncclGetErrorString(r));
    ICS_LAUNCH(bn_bwd_sync_c_kernel, dim3(2), dim3(64), 0, st, sync->local, 128, n * (double)sync->nranks, c1c2, c1c2 + 128);
    ICS_HIP(hipGetLastError());
  }
  return 0;
}
// ------------------------------------------------------------------------------------------
// BatchNorm-backward apply of a 3x3x3 layer's PRODUCER inside that layer's backward-data launch (round 4, the general form
// of the head fusion above).  Layer L (weights W[t][c][n], output gradient dy) reads P's BatchNorm output
// gamma xhat + beta (zero outside the grid).  d = dgrad_L(dy) is linear in dy, so the two batch sums P's BatchNorm backward
// needs follow from L's own weight-gradient GEMM, run on xhat instead of on gamma xhat + beta:
//   G[t][c][n]  = sum_v xhat[v + t][c] dy[v][n]             (the Winograd backward-weight kernel, source affine = xhat's)
//   S_t[n]      = sum over the voxels v with v + t inside the grid of dy[v][n]   (27 border-class sums, as cond_wgrad above)
//   dW[t][c][n] = gamma_c G + beta_c S_t[n]                 (what the GEMM on the BatchNorm output would have given)
//   sum_v d[v][c]      = sum_{t,n} W[t][c][n] S_t[n]   -> c1 = / n, P's dbeta
//   sum_v d xhat[v][c] = sum_{t,n} W[t][c][n] G[t][c][n] -> c2 = / n, P's dgamma
// all known BEFORE L's backward-data kernel runs, whose epilogue then writes P's dy directly (conv_wino64.hip FOLD = 2):
// P's separate pass over 3 x its activation bytes disappears.  Exact reassociations; the class sums in fp64, fixed order.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool tap_valid_for_class(int tapd /*0,1,2 = -1,0,+1*/, int cls /*0 first,1 interior,2 last*/) {
  return !((cls == 0 && tapd == 0) || (cls == 2 && tapd == 2));
}
struct ClassBlocks { int blk0[28]; };   // first block of each of the 27 border classes (class 13, the interior, has none)
constexpr int kClassRows = 64;          // voxel rows per block
constexpr int kTotalSplit = 16;         // blocks that share the sum over all voxels (slot 13 and 27 .. 41 of R)
// partial[blk][C]: sums of dy over the block's rows of its class; a class = {first plane, interior, last plane}^3
__global__ __launch_bounds__(256) void class_sums_kernel(const float* __restrict__ dy, int B, int S, int C, ClassBlocks cb,
                                                         float* __restrict__ partial) {
  __shared__ hv4 sh[256];
  int cls = 0;
  while (cls < 26 && (int)blockIdx.x >= cb.blk0[cls + 1]) ++cls;
  const int cz = cls / 9, cy = (cls / 3) % 3, cx = cls % 3;
  const int C4 = C >> 2, c4 = threadIdx.x % C4, vr = threadIdx.x / C4, VR = 256 / C4;
  auto lo = [&](int k) { return k == 0 ? 0 : (k == 1 ? 1 : S - 1); };
  auto cnt = [&](int k) { return k == 1 ? S - 2 : 1; };
  const int nz = cnt(cz), ny = cnt(cy), nx = cnt(cx), per = nz * ny * nx;
  const int total = B * per;
  const int r0 = ((int)blockIdx.x - cb.blk0[cls]) * kClassRows;
  const int r1 = min(r0 + kClassRows, total);
  hv4 acc = hv4{0.f, 0.f, 0.f, 0.f};
  for (int i = r0 + vr; i < r1; i += VR) {
    const int b = i / per, q = i - b * per;
    const int ix = q % nx, iy = (q / nx) % ny, iz = q / (nx * ny);
    const size_t v = (((size_t)b * S + (lo(cz) + iz)) * S + (lo(cy) + iy)) * S + (lo(cx) + ix);
    acc += *reinterpret_cast<const hv4*>(dy + v * C + c4 * 4);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (vr == 0) {
    hv4 t = sh[c4];
    for (int r = 1; r < VR; ++r) t += sh[r * C4 + c4];
    *reinterpret_cast<hv4*>(partial + (size_t)blockIdx.x * C + c4 * 4) = t;
  }
}
// R[cls][n]: the class sums (cls != 13); the sum over ALL voxels (L's bias-gradient partials [nblk][C], up to a few
// thousand rows) is shared by kTotalSplit blocks, slot 13 and 27 .. 41 of R, which conv_bnfuse_kernel adds up.
// One block per slot, (C / 4 channel quads) x (1024 / (C / 4) row groups); the row groups are merged in a fixed order.
struct dbl4 { double x, y, z, w; };
__global__ __launch_bounds__(1024) void class_reduce_kernel(const float* __restrict__ partial, ClassBlocks cb, int C,
                                                            const float* __restrict__ db_partial, int db_nblk,
                                                            double* __restrict__ R) {
  __shared__ dbl4 sh[1024];
  const int slot = blockIdx.x;
  const bool tot = slot == 13 || slot >= 27;
  const int part = slot == 13 ? 0 : slot - 26;                       // stripe of the total
  const int per = (db_nblk + kTotalSplit - 1) / kTotalSplit;
  const int t0 = min(part * per, db_nblk), t1 = min(t0 + per, db_nblk);
  const float* p = tot ? db_partial + (size_t)t0 * C : partial + (size_t)cb.blk0[slot] * C;
  const int nb = tot ? t1 - t0 : cb.blk0[slot + 1] - cb.blk0[slot];
  const int C4 = C >> 2, c4 = threadIdx.x % C4, rg = threadIdx.x / C4, RG = 1024 / C4;
  dbl4 a{0.0, 0.0, 0.0, 0.0};
  for (int k = rg; k < nb; k += RG) {
    const hv4 v = *reinterpret_cast<const hv4*>(p + (size_t)k * C + c4 * 4);
    a.x += (double)v[0]; a.y += (double)v[1]; a.z += (double)v[2]; a.w += (double)v[3];
  }
  sh[threadIdx.x] = a;
  __syncthreads();
  if (rg == 0) {
    for (int r = 1; r < RG; ++r) {
      const dbl4 b = sh[r * C4 + c4];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    double* o = R + (size_t)slot * C + c4 * 4;
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
  }
}
// one block per input channel c of L (= channel of P)
// CinTot: row pitch of W / G in channels (an up-split layer's skip channels are its first gridDim.x of CinTot);
// sums_out != nullptr: only the weight-gradient fix and the two sums [2][gridDim.x] (the producer has a second gradient
// source whose sums are added later: pool_bnfuse_kernel)
__global__ __launch_bounds__(256) void conv_bnfuse_kernel(const double* __restrict__ R, const float* __restrict__ W,
                                                          float* __restrict__ G, int CinTot, int N,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ scale, double cnt,
                                                          float* __restrict__ abc, float* __restrict__ c1c2,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          double* __restrict__ sums_out) {
  extern __shared__ double St[];          // [27][N]
  __shared__ double sh1[4], sh2[4];
  const int c = blockIdx.x, Cin = gridDim.x;
  constexpr int PF = 14;                  // this thread's (t, n) entries, requested ahead of the class-sum phase (N <= 128)
  float wpf[PF], qpf[PF];
  const bool pf = 27 * N <= PF * 256;
  if (pf) {
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int i = threadIdx.x + k * 256;
      const int t = i / N, n = i - t * N;
      const size_t idx = ((size_t)t * CinTot + c) * N + n;
      wpf[k] = i < 27 * N ? W[idx] : 0.f;
      qpf[k] = i < 27 * N ? G[idx] : 0.f;
    }
  }
  for (int n = threadIdx.x; n < N; n += 256) {
    double r[27], border = 0.0;
#pragma unroll
    for (int cls = 0; cls < 27; ++cls) {
      r[cls] = R[(size_t)cls * N + n];
      if (cls != 13) border += r[cls];
    }
    for (int k = 27; k < 26 + kTotalSplit; ++k) r[13] += R[(size_t)k * N + n];   // the other stripes of the total
    r[13] -= border;                                   // interior class = sum over all voxels - border classes
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      double sv = 0.0;
#pragma unroll
      for (int cls = 0; cls < 27; ++cls)
        if (tap_valid_for_class(t / 9, cls / 9) && tap_valid_for_class((t / 3) % 3, (cls / 3) % 3) &&
            tap_valid_for_class(t % 3, cls % 3))
          sv += r[cls];
      St[t * N + n] = sv;
    }
  }
  __syncthreads();
  const float g = gamma[c], b = beta[c];
  double pd = 0.0, pq = 0.0;
  if (pf) {
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int i = threadIdx.x + k * 256;
      if (i < 27 * N) {
        const int t = i / N, n = i - t * N;
        const size_t idx = ((size_t)t * CinTot + c) * N + n;
        const float w = wpf[k], q = qpf[k];
        const double sv = St[i];
        pd += (double)w * sv;
        pq += (double)w * (double)q;
        G[idx] = (float)((double)g * (double)q + (double)b * sv);
      }
    }
  } else {
  for (int i = threadIdx.x; i < 27 * N; i += 256) {
    const int t = i / N, n = i - t * N;
    const size_t idx = ((size_t)t * CinTot + c) * N + n;
    const float w = W[idx], q = G[idx];
    const double sv = St[i];
    pd += (double)w * sv;
    pq += (double)w * (double)q;
    G[idx] = (float)((double)g * (double)q + (double)b * sv);
  }
  }
  pd = wave_sum_d(pd); pq = wave_sum_d(pq);
  if ((threadIdx.x & 63) == 0) { sh1[threadIdx.x >> 6] = pd; sh2[threadIdx.x >> 6] = pq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double sd = (sh1[0] + sh1[1]) + (sh1[2] + sh1[3]), sq = (sh2[0] + sh2[1]) + (sh2[2] + sh2[3]);
    if (sums_out != nullptr) { sums_out[c] = sd; sums_out[Cin + c] = sq; return; }
    const double k1 = sd / cnt, k2 = sq / cnt, sc = (double)scale[c], rs = (double)rstd[c], mu = (double)mean[c];
    abc[c] = (float)sc;
    abc[Cin + c] = (float)(-sc * k2 * rs);
    abc[2 * Cin + c] = (float)(sc * (k2 * rs * mu - k1));
    c1c2[c] = (float)k1; c1c2[Cin + c] = (float)k2;
    dgamma[c] = (float)sq;
    dbeta[c] = (float)sd;
  }
}
static ClassBlocks class_blocks(int B, int S, int* total) {
  ClassBlocks cb{};
  int nb = 0;
  for (int cls = 0; cls < 27; ++cls) {
    cb.blk0[cls] = nb;
    if (cls == 13) continue;
    auto cnt = [&](int k) { return k == 1 ? S - 2 : 1; };
    const long long rows = (long long)B * cnt(cls / 9) * cnt((cls / 3) % 3) * cnt(cls % 3);
    nb += (int)((rows + kClassRows - 1) / kClassRows);
  }
  cb.blk0[27] = nb;
  *total = nb;
  return cb;
}
size_t conv_bnfuse_partial_floats(int B, int S, int C) {
  int nb = 0;
  class_blocks(B, S, &nb);
  return (size_t)nb * C;
}
bool conv_bnfuse_ok(int S, int Cin, int N) {
  const int C4 = N / 4;
  return S >= 3 && N % 4 == 0 && C4 <= 256 && 256 % C4 == 0 && (C4 & (C4 - 1)) == 0 && (size_t)27 * N * sizeof(double) <= 60 * 1024 && Cin > 0;
}
// dy [B S^3][N] -> G (in: the xhat-sourced weight-gradient GEMM; out: L's weight gradient), abc / c1c2 / dgamma / dbeta of P
int launch_conv_bnfuse(hipStream_t st, const float* dy, int B, int S, int Cin, int CinTot, int N, const float* db_partial,
                       int db_nblk, const float* W, float* G, const float* gamma, const float* beta, const float* mean,
                       const float* rstd, const float* scale, float* abc, float* c1c2, float* dgamma, float* dbeta,
                       float* ws_partial, size_t ws_partial_floats, double* ws_R, double* sums_out) {
  ICS_CHECK(conv_bnfuse_ok(S, Cin, N), "conv BN-fuse: unsupported shape");
  int nb = 0;
  const ClassBlocks cb = class_blocks(B, S, &nb);
  ICS_CHECK((size_t)nb * N <= ws_partial_floats, "conv BN-fuse: class-sum workspace too small");
  ICS_LAUNCH(class_sums_kernel, dim3(nb), dim3(256), 0, st, dy, B, S, N, cb, ws_partial);
  ICS_LAUNCH(class_reduce_kernel, dim3(26 + kTotalSplit), dim3(1024), 0, st, ws_partial, cb, N, db_partial, db_nblk, ws_R);
  ICS_LAUNCH(conv_bnfuse_kernel, dim3(Cin), dim3(256), (size_t)27 * N * sizeof(double), st, ws_R, W, G, CinTot, N, gamma, beta,
             mean, rstd, scale, (double)B * S * S * S, abc, c1c2, dgamma, dbeta, sums_out);
  ICS_HIP(hipGetLastError());
  return 0;
}
// The MaxPool3D consumer's share of the producer's two sums, on the pooled grid: a window hands its gradient g to the
// k elements of its tie mask, so  sum_v d = sum_w k g  and  sum_v d xhat = sum_w g (ssum - k mean) rstd  with ssum the
// sum of the stored activations of those elements (launch_pool_fwd).  partial [blocks][2][C] doubles.
constexpr int kPoolSumRows = 64;          // pooled rows per block
// pre != nullptr: the block sums go out as floats in bn_bwd_finalize_kernel's layout [2][C][gridDim.x] (block index
// fastest) -- the layer's ONLY gradient source is the max-pool (the perceptual U-Net's tap layers in a DFC-VAE step), so
// the sums on the pooled grid replace the BatchNorm-backward reduce pass over the fine grid (launch_pool_presum).
__global__ __launch_bounds__(256) void pool_sums_kernel(const float* __restrict__ g, int ldg, const unsigned char* __restrict__ mask,
                                                        const float* __restrict__ ssum, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, size_t rows, int C,
                                                        double* __restrict__ partial, float* __restrict__ pre) {
  __shared__ dbl4 sh[2][256];
  const int C4 = C >> 2, c4 = threadIdx.x % C4, rg = threadIdx.x / C4, RG = 256 / C4, c = c4 * 4;
  const size_t r0 = (size_t)blockIdx.x * kPoolSumRows;
  const size_t r1 = r0 + kPoolSumRows < rows ? r0 + kPoolSumRows : rows;
  const hv4 mu = *reinterpret_cast<const hv4*>(mean + c), rs = *reinterpret_cast<const hv4*>(rstd + c);
  dbl4 a{0.0, 0.0, 0.0, 0.0}, q{0.0, 0.0, 0.0, 0.0};
  for (size_t r = r0 + rg; r < r1; r += RG) {
    const hv4 gv = *reinterpret_cast<const hv4*>(g + r * ldg + c);
    const hv4 sv = *reinterpret_cast<const hv4*>(ssum + r * C + c);
    const unsigned m = *reinterpret_cast<const unsigned*>(mask + r * C + c);
    float kf[4], xs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      kf[j] = (float)__popc((m >> (8 * j)) & 0xffu);
      xs[j] = (sv[j] - kf[j] * mu[j]) * rs[j];
    }
    a.x += (double)(gv[0] * kf[0]); a.y += (double)(gv[1] * kf[1]); a.z += (double)(gv[2] * kf[2]); a.w += (double)(gv[3] * kf[3]);
    q.x += (double)(gv[0] * xs[0]); q.y += (double)(gv[1] * xs[1]); q.z += (double)(gv[2] * xs[2]); q.w += (double)(gv[3] * xs[3]);
  }
  sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = q;
  __syncthreads();
  if (rg == 0) {
    for (int r = 1; r < RG; ++r) {
      const dbl4 b1 = sh[0][r * C4 + c4], b2 = sh[1][r * C4 + c4];
      a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
      q.x += b2.x; q.y += b2.y; q.z += b2.z; q.w += b2.w;
    }
    if (pre != nullptr) {
      const size_t nb = gridDim.x;
      const double av[4] = {a.x, a.y, a.z, a.w}, qv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pre[(size_t)(c + j) * nb + blockIdx.x] = (float)av[j];
        pre[((size_t)C + c + j) * nb + blockIdx.x] = (float)qv[j];
      }
      return;
    }
    double* o = partial + (size_t)blockIdx.x * 2 * C;
    o[c] = a.x; o[c + 1] = a.y; o[c + 2] = a.z; o[c + 3] = a.w;
    o[C + c] = q.x; o[C + c + 1] = q.y; o[C + c + 2] = q.z; o[C + c + 3] = q.w;
  }
}
// one block per channel: (conv consumer's sums) + (pool consumer's block partials) -> the apply's constants
__global__ __launch_bounds__(256) void pool_bnfuse_kernel(const double* __restrict__ sums_conv, const double* __restrict__ partial,
                                                          int nblk, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ scale, double cnt, float* __restrict__ abc,
                                                          float* __restrict__ c1c2, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta) {
  __shared__ double sh[4];
  const int c = blockIdx.x, C = gridDim.x;
  double a = 0.0, q = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 256) { a += partial[(size_t)k * 2 * C + c]; q += partial[(size_t)k * 2 * C + C + c]; }
  a = block_sum_d(a, sh);
  q = block_sum_d(q, sh);
  if (threadIdx.x == 0) {
    const double sd = sums_conv[c] + a, sq = sums_conv[C + c] + q;
    const double k1 = sd / cnt, k2 = sq / cnt, sc = (double)scale[c], rs = (double)rstd[c], mu = (double)mean[c];
    abc[c] = (float)sc;
    abc[C + c] = (float)(-sc * k2 * rs);
    abc[2 * C + c] = (float)(sc * (k2 * rs * mu - k1));
    c1c2[c] = (float)k1; c1c2[C + c] = (float)k2;
    dgamma[c] = (float)sq;
    dbeta[c] = (float)sd;
  }
}
bool pool_presum_ok(int C, int ldg) { return C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0 && ldg % 4 == 0; }
int pool_presum_blocks(size_t pooled_rows) { return (int)((pooled_rows + kPoolSumRows - 1) / kPoolSumRows); }
// pre: [2][C][blocks] floats (BwdPre with ld = C)
int launch_pool_presum(hipStream_t st, const float* g, int ldg, const unsigned char* mask, const float* ssum, size_t pooled_rows,
                       int C, const float* mean, const float* rstd, float* pre, size_t pre_floats, int* blocks) {
  ICS_CHECK(pool_presum_ok(C, ldg), "pool pre-sum: unsupported channel count");
  const int nblk = pool_presum_blocks(pooled_rows);
  ICS_CHECK((size_t)2 * C * nblk <= pre_floats, "pool pre-sum: workspace too small");
  ICS_LAUNCH(pool_sums_kernel, dim3((unsigned)nblk), dim3(256), 0, st, g, ldg, mask, ssum, mean, rstd, pooled_rows, C,
             static_cast<double*>(nullptr), pre);
  ICS_HIP(hipGetLastError());
  *blocks = nblk;
  return 0;
}
size_t pool_bnfuse_partial_doubles(size_t pooled_rows, int C) { return (pooled_rows + kPoolSumRows - 1) / kPoolSumRows * 2 * C; }
int launch_pool_bnfuse(hipStream_t st, const float* g, int ldg, const unsigned char* mask, const float* ssum, size_t pooled_rows,
                       int C, double cnt, const double* sums_conv, const float* mean, const float* rstd, const float* scale,
                       float* abc, float* c1c2, float* dgamma, float* dbeta, double* ws_partial, size_t ws_partial_doubles) {
  const int C4 = C / 4;
  ICS_CHECK(C % 4 == 0 && C4 <= 256 && 256 % C4 == 0 && ldg % 4 == 0, "pool BN-fuse: unsupported channel count");
  const size_t nblk = (pooled_rows + kPoolSumRows - 1) / kPoolSumRows;
  ICS_CHECK(nblk * 2 * C <= ws_partial_doubles, "pool BN-fuse: workspace too small");
  ICS_LAUNCH(pool_sums_kernel, dim3((unsigned)nblk), dim3(256), 0, st, g, ldg, mask, ssum, mean, rstd, pooled_rows, C, ws_partial,
             nullptr);
  ICS_LAUNCH(pool_bnfuse_kernel, dim3(C), dim3(256), 0, st, sums_conv, ws_partial, (int)nblk, mean, rstd, scale, cnt, abc, c1c2,
             dgamma, dbeta);
  ICS_HIP(hipGetLastError());
  return 0;
}

bool head_dgrad_ok(int ncls, int cin, size_t M, const BwdStat* bs, int flags) {
  return ncls == 95 && cin == 128 && M % 16 == 0 && (bs == nullptr || bs->partial == nullptr || bs->post_act == ACT_NONE) &&
         !(flags & CF_NO_FUSED_HEAD);
}
int launch_head_dgrad(hipStream_t st, const float* dz, const float* wsoft_k, const float* wsig_k, float* dx, int ldo, size_t M,
                      const BwdStat* bs, int Npad, int* blocks, const float* c1c2, float* db_partial) {
  ICS_CHECK(M % 16 == 0 && ldo % 4 == 0, "head backward-data: rows must come in sixteens, float4-aligned");
  const int ntiles = (int)(M / 16);
  int nblk = (ntiles + 3) / 4;
  if (nblk > 512) nblk = 512;
  const BwdStat b = bs ? *bs : BwdStat{};
  if (c1c2 != nullptr) {       // fused BatchNorm-backward apply: dx receives the producer's dy, db_partial [nblk][128]
    ICS_CHECK(bs != nullptr && b.s && b.mean && b.rstd && b.scale && db_partial && b.ld == 128 && b.post_act == ACT_NONE,
              "fused head backward: the producer's activations, statistics and scale are needed");
    ICS_LAUNCH(head_dgrad_kernel<true>, dim3(nblk), dim3(256), 0, st, dz, wsoft_k, wsig_k, dx, ldo, ntiles, b, Npad, c1c2, db_partial);
    conv_set_last_kernel_id("head_dgrad_kernel<true>");
    if (blocks) *blocks = nblk;
  } else {
    ICS_LAUNCH(head_dgrad_kernel<false>, dim3(nblk), dim3(256), 0, st, dz, wsoft_k, wsig_k, dx, ldo, ntiles, b, Npad, nullptr, nullptr);
    conv_set_last_kernel_id("head_dgrad_kernel");
    if (blocks) *blocks = b.partial ? nblk : 0;
  }
  ICS_HIP(hipGetLastError());
  return 0;
}

// metrics[5] = [Loss, lsoft, lsig, f1, wr].  Data parallel: the six sums and the voxel count are the
// numerators / denominators SURVEY 8(e) asks to all-reduce (not the ratios):
//   phase 0: reduce the block partials and finalize (single GPU);
//   phase 1: reduce only -> sums[0..5], sums[6] = M;   phase 2: finalize from (all-reduced) sums.
__global__ __launch_bounds__(256) void head_finalize_kernel(const double* __restrict__ partial, int nblk,
                                                            double M, float* __restrict__ metrics,
                                                            double* __restrict__ sums, int phase,
                                                            double* __restrict__ keep) {
  __shared__ double sh[4];
  double acc[6];
  if (phase == 2) {
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = sums[k];
    M = sums[6];
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double s = 0.0;
      for (int b = threadIdx.x; b < nblk; b += 256) s += partial[(size_t)b * 6 + k];
      acc[k] = block_sum_d(s, sh);
    }
  }
  if (threadIdx.x != 0) return;
  if (phase == 1) {
#pragma unroll
    for (int k = 0; k < 6; ++k) sums[k] = acc[k];
    sums[6] = M;
    return;
  }
  if (keep) {   // the global sums behind the ratios (ics_unet_metric_sums): [sum lsoft, sum lsig, tp, predicted, wr tp, wr possible, M]
#pragma unroll
    for (int k = 0; k < 6; ++k) keep[k] = acc[k];
    keep[6] = M;
  }
  const double eps = 1e-7;
  const double lsoft = acc[0] / M, lsig = acc[1] / M;
  const double tp = acc[2], predicted = acc[3], possible = M;
  const double precision = tp / (predicted + eps), recall = tp / (possible + eps);
  const double f1 = 2.0 * ((precision * recall) / (precision + recall + eps));
  const double wr = acc[4] / (acc[5] + eps);
  metrics[0] = (float)(lsoft + lsig);
  metrics[1] = (float)lsoft;
  metrics[2] = (float)lsig;
  metrics[3] = (float)f1;
  metrics[4] = (float)wr;
}
int launch_head_metrics(hipStream_t st, const double* partial, int nblk, double M, float* metrics, double* sums,
                        int phase, double* keep) {
  ICS_LAUNCH(head_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nblk, M, metrics, sums, phase, keep);
  ICS_HIP(hipGetLastError());
  return 0;
}

int launch_head(hipStream_t st, float* z, int ldz, int ncls, const unsigned char* labels, size_t M,
                int mode, int want_grad, float wsoft, double* partial, int partial_blocks,
                float* metrics, int* nblk_out, float* dz_colsum, double* keep) {
  ICS_CHECK(ncls <= 128, "head kernel supports at most 128 classes");
  int rpb = (int)((M + partial_blocks - 1) / partial_blocks);
  rpb = (rpb + 15) / 16 * 16;
  const int nblk = (int)((M + rpb - 1) / rpb);
  ICS_LAUNCH(head_kernel, dim3(nblk), dim3(256), 0, st, z, ldz, ncls, labels, M, rpb, mode,
                     want_grad, wsoft, (float)(1.0 / (double)M), partial, dz_colsum);
  ICS_HIP(hipGetLastError());
  if (nblk_out) *nblk_out = nblk;
  if (mode != 0 && metrics != nullptr) ICS_TRY(launch_head_metrics(st, partial, nblk, (double)M, metrics, nullptr, 0, keep));
  return 0;
}

// ------------------------------------------------------------------------------------------
// Up-split backward (see engine.hip "split_up"): for a conv whose input channels come from a
// nearest-upsampled tensor xl, dW[tap] = xl^T * dyS[tap] and dxl = sum_tap dyS[tap] * W[tap]^T with
//   dyS[vl][tap][n] = sum over the <= 8 output voxels v with (v + off(tap)) >> 1 == vl of dy[v][n].
// Per axis, tap offset d and low-res coordinate a: v in {2a-d, 2a+1-d}; with slots s = 0..3 <->
// coordinate 2a-1+s:  d=+1 -> slots {0,1}, d=0 -> {1,2}, d=-1 -> {2,3}.  Separable sums in registers.
// ------------------------------------------------------------------------------------------
// One thread = one low-res voxel x FOUR channels (float4 loads / stores: the scalar version issued 64 4-byte loads and
// 27 4-byte stores per output and sat at 3.7 TB/s).
typedef float pf4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void pool27_kernel(const float* __restrict__ dy, int B, int S, int N, size_t total,
                                                     float* __restrict__ out, int ldo) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int Sh = S >> 1, N4 = N >> 2;
  const int n = (int)(i % N4) * 4;
  size_t vl = i / N4;
  const int c = vl % Sh; size_t r = vl / Sh;
  const int bq = r % Sh; r /= Sh;
  const int a = r % Sh;
  const int b = (int)(r / Sh);
  pf4 o[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) o[k] = pf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sz = 0; sz < 4; ++sz) {
    const int z = 2 * a - 1 + sz;
    if ((unsigned)z >= (unsigned)S) continue;
    pf4 lx[4][3];
#pragma unroll
    for (int sy = 0; sy < 4; ++sy) {
      const int y = 2 * bq - 1 + sy;
      pf4 v[4];
#pragma unroll
      for (int sx = 0; sx < 4; ++sx) {
        const int x = 2 * c - 1 + sx;
        const bool ok = (unsigned)y < (unsigned)S && (unsigned)x < (unsigned)S;
        v[sx] = ok ? *reinterpret_cast<const pf4*>(dy + ((((size_t)b * S + z) * S + y) * S + x) * N + n) : pf4{0.f, 0.f, 0.f, 0.f};
      }
      lx[sy][2] = v[0] + v[1];   // dx = +1  (tap index dx+1 = 2)
      lx[sy][1] = v[1] + v[2];   // dx =  0
      lx[sy][0] = v[2] + v[3];   // dx = -1
    }
    pf4 P[3][3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      P[2][dx] = lx[0][dx] + lx[1][dx];   // dy = +1
      P[1][dx] = lx[1][dx] + lx[2][dx];   // dy =  0
      P[0][dx] = lx[2][dx] + lx[3][dx];   // dy = -1
    }
#pragma unroll
    for (int dyi = 0; dyi < 3; ++dyi)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        if (sz <= 1) o[2 * 9 + dyi * 3 + dx] += P[dyi][dx];             // dz = +1: slots 0,1
        if (sz >= 1 && sz <= 2) o[1 * 9 + dyi * 3 + dx] += P[dyi][dx];  // dz =  0: slots 1,2
        if (sz >= 2) o[0 * 9 + dyi * 3 + dx] += P[dyi][dx];            // dz = -1: slots 2,3
      }
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) *reinterpret_cast<pf4*>(out + vl * ldo + (size_t)k * N + n) = o[k];
}
int launch_pool27(hipStream_t st, const float* dy, int B, int S, int N, float* out, int ldo) {
  ICS_CHECK(N % 4 == 0 && ldo % 4 == 0, "pool27: channel counts must be multiples of 4");
  const size_t total = (size_t)B * (S / 2) * (S / 2) * (S / 2) * (N / 4);
  ICS_LAUNCH(pool27_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, B, S, N, total, out, ldo);
  ICS_HIP(hipGetLastError());
  return 0;
}
// dw[(tap*Cin + c_off + c)*N + n] = tmp[c*(27N) + tap*N + n]
__global__ void permute_up_dw_kernel(const float* __restrict__ tmp, int Cu, int N, int Cin, int c_off,
                                     size_t total, float* __restrict__ dw) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int n = (int)(i % N);
  size_t r = i / N;
  const int c = (int)(r % Cu);
  const int tap = (int)(r / Cu);
  dw[((size_t)tap * Cin + c_off + c) * N + n] = tmp[(size_t)c * 27 * N + (size_t)tap * N + n];
}
int launch_permute_up_dw(hipStream_t st, const float* tmp, int Cu, int N, int Cin, int c_off, float* dw) {
  const size_t total = (size_t)27 * Cu * N;
  ICS_LAUNCH(permute_up_dw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, tmp, Cu, N, Cin,
                     c_off, total, dw);
  ICS_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Adam (keras 2.3.1): m,v update, p -= lr_t * m / (sqrt(v) + 1e-7)
// ------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float lr_t, float b1, float b2, float omb1,
                            float omb2, float eps, float gscale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  const float mi = b1 * m[i] + omb1 * gi;        // omb = 1-beta evaluated in double on the host
  const float vi = b2 * v[i] + omb2 * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
}
int launch_adam(hipStream_t st, float* p, const float* g, float* m, float* v, size_t n, float lr_t,
                float gscale) {
  ICS_LAUNCH(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n,
                     lr_t, 0.9f, 0.999f, (float)(1.0 - 0.9), (float)(1.0 - 0.999), 1e-7f, gscale);
  ICS_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// DFC-VAE pieces (vae/lattice_vae.py:53-66, 232-270)
// ------------------------------------------------------------------------------------------
// o = act(s*scale+shift) materialised (decoder output / any tensor that must leave the device)
__global__ void bn_apply_kernel(const float* __restrict__ s, const float* __restrict__ scale,
                                const float* __restrict__ shift, int act, size_t n, int C,
                                float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = i % C;
  out[i] = act_fwd(scale ? fmaf(s[i], scale[c], shift[c]) : s[i], act);
}
int launch_bn_apply(hipStream_t st, const float* s, const float* scale, const float* shift, int act,
                    size_t n, int C, float* out) {
  ICS_LAUNCH(bn_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s, scale, shift,
                     act, n, C, out);
  ICS_HIP(hipGetLastError());
  return 0;
}

// sum over elements of (a-b)^2 per sample -> partial[b][blk]; optional grad: dst = coef*(b - a)
// (coef carries the 2/N and loss weights; sign chosen for d/d b of (a-b)^2 = 2 (b-a)).
__global__ __launch_bounds__(256) void sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      size_t per_sample, int blocks_per_sample,
                                                      double* __restrict__ partial, float* __restrict__ grad,
                                                      float coef, int accumulate) {
  __shared__ double sh[4];
  const int smp = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
  size_t chunk = (per_sample + blocks_per_sample - 1) / blocks_per_sample;
  chunk = (chunk + 3) & ~(size_t)3;
  const size_t i0 = (size_t)blk * chunk, i1 = i0 + chunk < per_sample ? i0 + chunk : per_sample;
  const size_t base = (size_t)smp * per_sample;
  double acc = 0.0;
  const bool v4 = (per_sample & 3) == 0 && grad == nullptr;     // eval path of the big taps: 16-byte loads
  if (v4) {
    for (size_t i = i0 + 4 * (size_t)threadIdx.x; i < i1; i += 1024) {
      const float4 x = *reinterpret_cast<const float4*>(a + base + i);
      const float4 y = *reinterpret_cast<const float4*>(b + base + i);
      const float d0 = y.x - x.x, d1 = y.y - x.y, d2 = y.z - x.z, d3 = y.w - x.w;
      acc += (double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2 + (double)d3 * d3;
    }
  } else {
    for (size_t i = i0 + threadIdx.x; i < i1; i += 256) {
      const float d = b[base + i] - a[base + i];
      acc += (double)d * (double)d;
      if (grad) {
        const float gv = coef * d;
        grad[base + i] = accumulate ? grad[base + i] + gv : gv;
      }
    }
  }
  acc = block_sum_d(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
int launch_sqdiff(hipStream_t st, const float* a, const float* b, int B, size_t per_sample,
                  int blocks_per_sample, double* partial, float* grad, float coef, int accumulate) {
  ICS_LAUNCH(sqdiff_kernel, dim3(B * blocks_per_sample), dim3(256), 0, st, a, b, per_sample,
                     blocks_per_sample, partial, grad, coef, accumulate);
  ICS_HIP(hipGetLastError());
  return 0;
}

// z = mu + exp(0.5*logvar)*eps ; zc = [z | cond]
__global__ void sampling_kernel(const float* __restrict__ mulv, int ld, int latent,
                                const float* __restrict__ eps, const float* __restrict__ cond, int ncond,
                                int B, float* __restrict__ z, float* __restrict__ zc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int W = latent + ncond;
  if (i >= B * W) return;
  const int b = i / W, j = i % W;
  if (j < latent) {
    const float mu = mulv[b * ld + j], lv = mulv[b * ld + latent + j];
    const float zz = mu + expf(0.5f * lv) * eps[b * latent + j];
    z[b * latent + j] = zz;
    zc[i] = zz;
  } else {
    zc[i] = cond[b * ncond + (j - latent)];
  }
}
int launch_sampling(hipStream_t st, const float* mulv, int ld, int latent, const float* eps,
                    const float* cond, int ncond, int B, float* z, float* zc) {
  const int n = B * (latent + ncond);
  ICS_LAUNCH(sampling_kernel, dim3((n + 255) / 256), dim3(256), 0, st, mulv, ld, latent, eps, cond,
                     ncond, B, z, zc);
  ICS_HIP(hipGetLastError());
  return 0;
}

// VAE loss assembly + gradient w.r.t. (mu, logvar) given dL/dz.
//   kld_b = -0.5*sum_j(1 + lv - mu^2 - exp(lv));  Loss = mse + alpha*mean_b(pm_b) + beta*mean_b(kld_b)
// metrics[4] = [Loss, PM, MSE, KLD]; dmulv[b][0:latent] = dz + beta*mu/B ;
// dmulv[b][latent:] = dz*eps*0.5*exp(0.5 lv) + beta*(-0.5)*(1-exp(lv))/B
// phase 0: everything; phase 1: raw sums only -> sums[5] = {kl, squared error, weighted pm, B, n_elems};
// phase 2: metrics from the (all-reduced) sums  (data parallel: numerators / denominators)
__global__ __launch_bounds__(1024) void vae_loss_kernel(const float* __restrict__ mulv, int ld, int latent,
                                                       int B, const double* __restrict__ mse_partial,
                                                       int n_mse, double n_elems,
                                                       const double* __restrict__ pm_partial,
                                                       PmSums pmc, float alpha,
                                                       float beta, float* __restrict__ metrics,
                                                       double* __restrict__ sums, int phase) {
  // 1024 threads (round 6: with 256 the 8192 fp64 exp() of the KL term alone took 31 us on the step's critical chain)
  __shared__ double sh[16];
  const int nt = (int)blockDim.x;
  double kl = 0.0, mse = 0.0, pm = 0.0, nb_tot = (double)B;
  if (phase == 2) {
    kl = sums[0]; mse = sums[1]; pm = sums[2]; nb_tot = sums[3]; n_elems = sums[4];
  } else {
    for (int i = threadIdx.x; i < B * latent; i += nt) {
      const int b = i / latent, j = i % latent;
      const double mu = mulv[b * ld + j], lv = mulv[b * ld + latent + j];
      kl += -0.5 * (1.0 + lv - mu * mu - exp(lv));
    }
    kl = block_sum_dn(kl, sh);
    for (int i = threadIdx.x; i < n_mse; i += nt) mse += mse_partial[i];
    mse = block_sum_dn(mse, sh);
    size_t off = 0;
    for (int l = 0; l < 4; ++l) {
      const int nb = pmc.n[l];
      double s = 0.0;
      for (int i = threadIdx.x; i < nb; i += nt) s += pm_partial[off + i];
      s = block_sum_dn(s, sh);
      pm += (double)pmc.w[l] * s / pmc.per[l];
      off += nb;
    }
  }
  if (threadIdx.x != 0) return;
  if (phase == 1) {
    sums[0] = kl; sums[1] = mse; sums[2] = pm; sums[3] = nb_tot; sums[4] = n_elems;
    return;
  }
  const double mse_mean = mse / n_elems, pm_mean = pm / nb_tot, kl_mean = kl / nb_tot;
  metrics[0] = (float)(mse_mean + alpha * pm_mean + beta * kl_mean);
  metrics[1] = (float)pm_mean;
  metrics[2] = (float)mse_mean;
  metrics[3] = (float)kl_mean;
}
__global__ void vae_dz_kernel(const float* __restrict__ mulv, int ld, int latent, int B,
                              const float* __restrict__ eps, const float* __restrict__ dzc, int ldzc,
                              float beta, float* __restrict__ dmulv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * latent) return;
  const int b = i / latent, j = i % latent;
  const float mu = mulv[b * ld + j], lv = mulv[b * ld + latent + j];
  const float dz = dzc[b * ldzc + j];
  dmulv[b * ld + j] = dz + beta * mu / (float)B;
  dmulv[b * ld + latent + j] =
      dz * eps[b * latent + j] * 0.5f * expf(0.5f * lv) + beta * (-0.5f) * (1.f - expf(lv)) / (float)B;
}
int launch_vae_loss(hipStream_t st, const float* mulv, int ld, int latent, int B, const double* mse_partial,
                    int n_mse, double n_elems, const double* pm_partial, const PmSums& pmc, float alpha, float beta,
                    float* metrics, double* sums, int phase) {
  ICS_LAUNCH(vae_loss_kernel, dim3(1), dim3(1024), 0, st, mulv, ld, latent, B, mse_partial, n_mse,
                     n_elems, pm_partial, pmc, alpha, beta, metrics, sums, phase);
  ICS_HIP(hipGetLastError());
  return 0;
}
int launch_vae_dz(hipStream_t st, const float* mulv, int ld, int latent, int B, const float* eps,
                  const float* dzc, int ldzc, float beta, float* dmulv) {
  const int n = B * latent;
  ICS_LAUNCH(vae_dz_kernel, dim3((n + 255) / 256), dim3(256), 0, st, mulv, ld, latent, B, eps, dzc,
                     ldzc, beta, dmulv);
  ICS_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Spatially constant input channels (the VAE encoder's condition: Reshape -> K.tile -> Concatenate,
// lattice_vae.py:167-169).  A 3x3x3 "same" conv over a channel that is constant in space gives, per sample, one
// value per BORDER CLASS of the output voxel (first / interior / last plane on each axis: 27 classes) -- the zero
// padding removes the taps that reach outside.  So instead of materialising C*cond extra input channels:
//   forward : pos_bias[b][cls][co] = bias[co] + sum_{taps valid for cls} sum_j W[tap][C+j][co] * cond[b][j % ncond]
//   backward: dW[tap][C+j][co] = sum_b cond[b][j % ncond] * D[b][tap][co],   D[b][tap] = sum of dy over the voxels
//             whose tap-shifted position lies inside the grid = sums of the 27 REGION sums R[b][cls] of dy.
// Exact reassociations of the same sums (the border regions are summed directly, the interior one is total - rest).
// ------------------------------------------------------------------------------------------
__global__ void cond_bias_table_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                       const float* __restrict__ cond, int C, int ncond, int Cin_tot, int Cout, int B,
                                       float* __restrict__ T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 27 * Cout) return;
  const int co = i % Cout, cls = (i / Cout) % 27, b = i / (27 * Cout);
  const int cz = cls / 9, cy = (cls / 3) % 3, cx = cls % 3;
  double acc = bias ? (double)bias[co] : 0.0;
  for (int tap = 0; tap < 27; ++tap) {
    if (!tap_valid_for_class(tap / 9, cz) || !tap_valid_for_class((tap / 3) % 3, cy) || !tap_valid_for_class(tap % 3, cx))
      continue;
    for (int j = 0; j < C * ncond; ++j)
      acc += (double)w[((size_t)tap * Cin_tot + C + j) * Cout + co] * (double)cond[b * ncond + (j % ncond)];
  }
  T[i] = (float)acc;
}
int launch_cond_bias_table(hipStream_t st, const float* w, const float* bias, const float* cond, int C, int ncond,
                           int Cin_tot, int Cout, int B, float* T) {
  const int n = B * 27 * Cout;
  ICS_LAUNCH(cond_bias_table_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, bias, cond, C, ncond, Cin_tot,
                     Cout, B, T);
  ICS_HIP(hipGetLastError());
  return 0;
}

// partial[b][blk][co] = sum of dy over chunk blk of sample b (per-sample totals, first stage)
__global__ __launch_bounds__(256) void sample_colsum_kernel(const float* __restrict__ dy, size_t per_sample, int C,
                                                             int blocks_per_sample, double* __restrict__ partial) {
  __shared__ double sh[256];
  const int b = blockIdx.x / blocks_per_sample, blk = blockIdx.x % blocks_per_sample;
  const size_t chunk = (per_sample + blocks_per_sample - 1) / blocks_per_sample;
  const size_t v0 = (size_t)blk * chunk, v1 = v0 + chunk < per_sample ? v0 + chunk : per_sample;
  const int c = threadIdx.x % C, vr = threadIdx.x / C, VR = 256 / C;        // C divides 256 (16 or 32)
  double acc = 0.0;
  const float* base = dy + (size_t)b * per_sample * C;
  for (size_t v = v0 + vr; v < v1; v += VR) acc += (double)base[v * C + c];
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (vr == 0) {
    double s = 0.0;
    for (int r = 0; r < VR; ++r) s += sh[r * C + c];
    partial[((size_t)b * blocks_per_sample + blk) * C + c] = s;
  }
}
// R[b][cls][co] for the 26 border classes (cls != 13): direct sums over the region's voxels
__global__ __launch_bounds__(256) void border_region_sums_kernel(const float* __restrict__ dy, int S, int C,
                                                                  double* __restrict__ R) {
  __shared__ double sh[256];
  const int b = blockIdx.x / 27, cls = blockIdx.x % 27;
  const int cz = cls / 9, cy = (cls / 3) % 3, cx = cls % 3;
  const int c = threadIdx.x % C, vr = threadIdx.x / C, VR = 256 / C;
  auto lo = [&](int k) { return k == 0 ? 0 : (k == 1 ? 1 : S - 1); };
  auto cnt = [&](int k) { return k == 1 ? S - 2 : 1; };
  double acc = 0.0;
  if (cls != 13) {
    const int nz = cnt(cz), ny = cnt(cy), nx = cnt(cx);
    const int total = nz * ny * nx;
    const float* base = dy + (size_t)b * S * S * S * C;
    for (int i = vr; i < total; i += VR) {
      const int ix = i % nx, iy = (i / nx) % ny, iz = i / (nx * ny);
      const size_t v = ((size_t)(lo(cz) + iz) * S + (lo(cy) + iy)) * S + (lo(cx) + ix);
      acc += (double)base[v * C + c];
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (vr == 0 && cls != 13) {
    double s = 0.0;
    for (int r = 0; r < VR; ++r) s += sh[r * C + c];
    R[((size_t)b * 27 + cls) * C + c] = s;
  }
}
// one block per tap: D[b][co] for this tap from the region sums, then dW[tap][C0 + j][co] = sum_b cond[b][j%ncond]*D
__global__ __launch_bounds__(256) void cond_wgrad_kernel(const double* __restrict__ R, const double* __restrict__ tot_partial,
                                                          int blocks_per_sample, const float* __restrict__ cond, int B,
                                                          int C0, int nfold, int ncond, int Cin_tot, int Cout,
                                                          float* __restrict__ dw) {
  extern __shared__ double D[];            // [B][Cout]
  const int tap = blockIdx.x;
  const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
  for (int i = threadIdx.x; i < B * Cout; i += 256) {
    const int b = i / Cout, co = i % Cout;
    double tot = 0.0, border = 0.0, d = 0.0;
    for (int k = 0; k < blocks_per_sample; ++k) tot += tot_partial[((size_t)b * blocks_per_sample + k) * Cout + co];
    for (int cls = 0; cls < 27; ++cls)
      if (cls != 13) border += R[((size_t)b * 27 + cls) * Cout + co];
    for (int cls = 0; cls < 27; ++cls) {
      if (!tap_valid_for_class(tz, cls / 9) || !tap_valid_for_class(ty, (cls / 3) % 3) || !tap_valid_for_class(tx, cls % 3))
        continue;
      d += cls == 13 ? tot - border : R[((size_t)b * 27 + cls) * Cout + co];
    }
    D[i] = d;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nfold * Cout; i += 256) {
    const int j = i / Cout, co = i % Cout;
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += (double)cond[b * ncond + (j % ncond)] * D[b * Cout + co];
    dw[((size_t)tap * Cin_tot + C0 + j) * Cout + co] = (float)s;
  }
}
size_t cond_wgrad_workspace_doubles(int B, int Cout) { return (size_t)B * (64 + 27) * Cout; }
int launch_cond_wgrad(hipStream_t st, const float* dy, int B, int S, int Cout, const float* cond, int C0, int nfold,
                      int ncond, int Cin_tot, float* dw, double* ws, size_t ws_doubles) {
  ICS_CHECK(256 % Cout == 0 && S >= 3, "cond wgrad: unsupported shape");
  const int bps = 64;
  ICS_CHECK(cond_wgrad_workspace_doubles(B, Cout) <= ws_doubles, "cond wgrad workspace too small");
  double* tot_partial = ws;                                  // [B][64][Cout]
  double* R = ws + (size_t)B * bps * Cout;                   // [B][27][Cout]
  const size_t per = (size_t)S * S * S;
  ICS_LAUNCH(sample_colsum_kernel, dim3(B * bps), dim3(256), 0, st, dy, per, Cout, bps, tot_partial);
  ICS_LAUNCH(border_region_sums_kernel, dim3(B * 27), dim3(256), 0, st, dy, S, Cout, R);
  ICS_LAUNCH(cond_wgrad_kernel, dim3(27), dim3(256), (size_t)B * Cout * sizeof(double), st, R, tot_partial, bps,
                     cond, B, C0, nfold, ncond, Cin_tot, Cout, dw);
  ICS_HIP(hipGetLastError());
  return 0;
}

// relu backward for the Dense(256, relu): g *= (a > 0), where a is the stored relu OUTPUT
__global__ void relu_bwd_kernel(const float* __restrict__ a, float* __restrict__ g, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !(a[i] > 0.f)) g[i] = 0.f;
}
int launch_relu_bwd(hipStream_t st, const float* a, float* g, size_t n) {
  ICS_LAUNCH(relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, g, n);
  ICS_HIP(hipGetLastError());
  return 0;
}
// column sums of a small [rows][cols] matrix (dense-layer bias gradients)
__global__ void colsum_small_kernel(const float* __restrict__ a, int rows, int cols, int ld,
                                    float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int r = 0; r < rows; ++r) s += a[(size_t)r * ld + c];
  out[c] = s;
}
int launch_colsum_small(hipStream_t st, const float* a, int rows, int cols, int ld, float* out) {
  ICS_LAUNCH(colsum_small_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, a, rows, cols, ld, out);
  ICS_HIP(hipGetLastError());
  return 0;
}
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n, float a) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += a * x[i];
}
int launch_axpy(hipStream_t st, float* y, const float* x, size_t n, float a) {
  ICS_LAUNCH(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, x, n, a);
  ICS_HIP(hipGetLastError());
  return 0;
}

}  // namespace ics
