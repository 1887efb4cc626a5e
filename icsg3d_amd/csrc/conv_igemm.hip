// Conv3D forward / backward-data / backward-weight as implicit GEMMs on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32: bit-identical to an fmaf chain, issued at the fp32 vector peak; see
// /opt/skills/guides/cdna_hip_programming.md §3).  Replaces keras.layers.Conv3D as used at
// /root/reference/unet/unet.py:276-352 and /root/reference/vae/lattice_vae.py:173,178,213,219.
//
// GEMM view (NDHWC):  out[m][n] = sum_k A[m][k] * W[k][n],  m = voxel (b,z,y,x), k = (tap,ci), n = co.
// A is never materialised: the tile loader gathers the tap-shifted voxel rows, applies the
// producer's BatchNorm affine + activation, nearest-upsampling and channel concat on the fly, and
// writes zeros for the "same" padding.  Tiles are register-staged (global -> VGPR -> LDS) with two
// LDS buffers; fragments are read with ds_read_b128 using a fixed k-permutation inside each 8-deep
// k group (lane half h supplies k = 4h..4h+3), which only reorders the fp32 summation.
#include "common.h"

#ifndef ICS_GEMM_DEEP_FWD_MIN
#define ICS_GEMM_DEEP_FWD_MIN 1  // accumulator tiles per wave from which the forward / backward-data GEMM path prefetches two chunks ahead
                                 // (with the copying form the 64 x 64 tiles lost 4 %; with swapping sets: c15.up / c13.up dgrad -6 / -7 %)
#endif
#ifndef ICS_GEMM_DEEP_MIN
#define ICS_GEMM_DEEP_MIN 3  // accumulator tiles per wave from which the backward-weight GEMM prefetches two chunks ahead
#endif
#ifndef ICS_GEMM_DEEP
#define ICS_GEMM_DEEP 1      // 1: chunks are requested TWO ahead (a second register set): these GEMMs stream both operands
#endif                       //    from HBM with no reuse in L2, one chunk of prefetch left the loads exposed

#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace ics {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));   // first-class vector: stays in VGPRs

#ifndef ICS_KLDA
#define ICS_KLDA 36
#endif
constexpr int kLDA = ICS_KLDA;   // A-tile row stride in floats: 32 + 4 pad -> conflict-free ds_read_b128

__device__ __forceinline__ int xcd_swizzle(int bid, int nblk) {
  // blocks are dispatched round-robin over the 8 XCDs; give each XCD a contiguous range of tiles
  // so neighbouring M-tiles (shared halo) and the N-tiles of one M-tile hit the same L2.
  const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Exact template instantiation (the name rocprofv3 prints) of the last conv GEMM kernel this thread launched.
static thread_local const char* g_last_kernel_id = "";
const char* conv_last_kernel_id() { return g_last_kernel_id; }
void conv_set_last_kernel_id(const char* id) { g_last_kernel_id = id; }
static const char* tf(bool b) { return b ? "true" : "false"; }

int conv_flags_from_env() {
  int f = 0;
  if (getenv("ICSG3D_NO_REUSE")) f |= CF_NO_REUSE;
  if (getenv("ICSG3D_NO_WGRAD3")) f |= CF_NO_WGRAD3;
  if (getenv("ICSG3D_NO_WGRAD3S")) f |= CF_NO_WGRAD3S;
  if (getenv("ICSG3D_NO_FWD_SPLITK")) f |= CF_NO_FWD_SPLITK;
  if (getenv("ICSG3D_NO_THIN_N")) f |= CF_NO_THIN_N;
  if (getenv("ICSG3D_NO_UPSPLIT")) f |= CF_NO_UPSPLIT;
  if (getenv("ICSG3D_NO_THIN_C")) f |= CF_NO_THIN_C;
  if (getenv("ICSG3D_NO_COND_FOLD")) f |= CF_NO_COND_FOLD;
  if (getenv("ICSG3D_NO_WINO")) f |= CF_NO_WINO;
  if (getenv("ICSG3D_NO_WINO_WGRAD")) f |= CF_NO_WINO_WGRAD;
  if (getenv("ICSG3D_NO_WINO64")) f |= CF_NO_WINO64;
  if (getenv("ICSG3D_NO_UP3")) f |= CF_NO_UP3;
  if (getenv("ICSG3D_NO_THIN1_2STAGE")) f |= CF_NO_THIN1_2STAGE;
  if (getenv("ICSG3D_NO_FAST_BNBWD")) f |= CF_NO_FAST_BNBWD;
  if (getenv("ICSG3D_NO_FUSED_HEAD")) f |= CF_NO_FUSED_HEAD;
  if (getenv("ICSG3D_NO_BWD_FOLD")) f |= CF_NO_BWD_FOLD;
  if (getenv("ICSG3D_NO_WINOG")) f |= CF_NO_WINOG;
  if (getenv("ICSG3D_NO_HEAD_BNFUSE")) f |= CF_NO_HEAD_BNFUSE;
  if (getenv("ICSG3D_NO_UP3N")) f |= CF_NO_UP3N;
  if (getenv("ICSG3D_NO_DGRAD_BNFUSE")) f |= CF_NO_DGRAD_BNFUSE;
  if (getenv("ICSG3D_NO_POOL_PRESUM")) f |= CF_NO_POOL_PRESUM;
  if (getenv("ICSG3D_NO_HEAD_LABELS")) f |= CF_NO_HEAD_LABELS;
  { const char* e = getenv("ICSG3D_UP3_BIG_MIN_WG"); if (e && *e && strtoul(e, nullptr, 10) <= 1) f |= CF_UP3_BIG_ALWAYS; }
  return f;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember per (kernel instantiation, device)
struct DevOnce {
  bool done[64] = {false};
  bool need(int* dev) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) { *dev = -1; return true; }
    *dev = d;
    return !done[d];
  }
  void mark(int dev) { if (dev >= 0) done[dev] = true; }
};

// Sources without a BatchNorm affine get pointers to constant ones/zeros so that the tile loaders
// never branch on "has affine" (fma(v,1,0) == v exactly).
static int identity_affine(const float** ones, const float** zeros) {
  constexpr int kN = 16384;
  static float* buf[64] = {nullptr};
  int dev = 0;
  ICS_HIP(hipGetDevice(&dev));
  ICS_CHECK(dev >= 0 && dev < 64, "device index out of range");
  if (!buf[dev]) {
    float* p = nullptr;
    ICS_HIP(hipMalloc(&p, 2 * kN * sizeof(float)));
    std::vector<float> h(2 * kN, 0.f);
    for (int i = 0; i < kN; ++i) h[i] = 1.f;
    ICS_HIP(hipMemcpy(p, h.data(), 2 * kN * sizeof(float), hipMemcpyHostToDevice));
    buf[dev] = p;
  }
  *ones = buf[dev];
  *zeros = buf[dev] + kN;
  return 0;
}
static int fix_src(ConvSrc& s) {
  if (s.scale == nullptr) {
    ICS_CHECK(s.C <= 16384, "source wider than the identity-affine buffer");
    ICS_TRY(identity_affine(&s.scale, &s.shift));
    s.act = ACT_NONE;
  }
  return 0;
}

struct RowPos {
  int b, z, y, x;
};
__device__ __forceinline__ RowPos decode_row(int m, int S, int lg) {
  RowPos r;
  r.x = m & (S - 1);
  r.y = (m >> lg) & (S - 1);
  r.z = (m >> (2 * lg)) & (S - 1);
  r.b = m >> (3 * lg);
  return r;
}

// element offset of voxel (b,z,y,x) (already shifted, in bounds) inside source s
__device__ __forceinline__ size_t src_voxel(const ConvSrc& s, int b, int z, int y, int x, int S) {
  if (s.up) {
    const int Sh = S >> 1;
    return ((((size_t)b * Sh + (z >> 1)) * Sh + (y >> 1)) * Sh + (x >> 1)) * (size_t)s.C;
  }
  return ((((size_t)b * S + z) * S + y) * S + x) * (size_t)s.C;
}

// Branch-free activation: slope 1 = identity, 0 = ReLU, 0.3 = LeakyReLU.  max(v, v*slope)
// is exact for all three, and keeps the tile loaders a single basic block
// so the scheduler can run the next tile's address math and loads under the current tile's MFMAs.
__host__ __device__ __forceinline__ float act_slope_of(int act) {
  return act == ACT_RELU ? 0.f : (act == ACT_LRELU ? kLeaky : 1.f);
}
__device__ __forceinline__ float act_apply(float v, float slope) {
  return fmaxf(v, v * slope);   // slope in [0,1]: v>0 -> v ; v<0 -> v*slope ; exact for 0, 0.3, 1
}
// NOACT: the source's activation sits BEFORE its BatchNorm (every U-Net layer: BN(ReLU(conv))), so the loader
// applies the affine only: 1 instead of 3 vector ops per element, as a compile-time variant (a run-time
// `slope == 1` test splits the loop's basic block and costs 7 % - measured).
template <bool NOACT>
__device__ __forceinline__ v4f affine_only_or_act4(v4f v, v4f sc, v4f sh, float slope);
__device__ __forceinline__ v4f affine_act4(v4f v, v4f sc, v4f sh, float slope) {
  v4f r;
  r.x = act_apply(fmaf(v.x, sc.x, sh.x), slope); r.y = act_apply(fmaf(v.y, sc.y, sh.y), slope);
  r.z = act_apply(fmaf(v.z, sc.z, sh.z), slope); r.w = act_apply(fmaf(v.w, sc.w, sh.w), slope);
  return r;
}
template <bool NOACT>
__device__ __forceinline__ v4f affine_only_or_act4(v4f v, v4f sc, v4f sh, float slope) {
  if (!NOACT) return affine_act4(v, sc, sh, slope);
  v4f r;
  r.x = fmaf(v.x, sc.x, sh.x); r.y = fmaf(v.y, sc.y, sh.y); r.z = fmaf(v.z, sc.z, sh.z); r.w = fmaf(v.w, sc.w, sh.w);
  return r;
}
// element offset of the (clamped) voxel in a source; u = 1 for a nearest-upsampled source
__device__ __forceinline__ unsigned voxel_off(int b, int z, int y, int x, int S, int u, int C) {
  const int Ss = S >> u;
  return ((((unsigned)b * Ss + (z >> u)) * Ss + (y >> u)) * Ss + (x >> u)) * (unsigned)C;
}
__device__ __forceinline__ int clampi(int v, int hi) { return min(max(v, 0), hi); }

// field-wise select (a reference to `cond ? s0 : s1` would force both kernel-arg structs to scratch)
__device__ __forceinline__ ConvSrc pick_src(const ConvSrc& s0, const ConvSrc& s1, bool first) {
  ConvSrc s;
  s.p = first ? s0.p : s1.p;
  s.scale = first ? s0.scale : s1.scale;
  s.shift = first ? s0.shift : s1.shift;
  s.C = first ? s0.C : s1.C;
  s.up = first ? s0.up : s1.up;
  s.act = first ? s0.act : s1.act;
  s.bcast = first ? s0.bcast : s1.bcast;
  return s;
}

// scalar gather of one element of the virtual input (used by the thin-channel paths)
__device__ __forceinline__ float gather_scalar(const ConvSrc& s0, const ConvSrc& s1, int ci, int b,
                                               int z, int y, int x, int S) {
  const bool first = ci < s0.C;
  const ConvSrc s = pick_src(s0, s1, first);
  const int cl = first ? ci : ci - s0.C;
  float v;
  if (s.bcast > 0) {
    v = s.p[(size_t)b * s.bcast + (cl % s.bcast)];
  } else {
    v = s.p[src_voxel(s, b, z, y, x, S) + cl];
  }
  if (s.scale) v = act_fwd(fmaf(v, s.scale[cl], s.shift[cl]), s.act);
  return v;
}

// =====================================================================================
// Forward / backward-data kernel
// =====================================================================================
// ABL (ablation, benchmarking only): 0 = full kernel; 1 = no global loads / LDS stores in the loop
// (MFMA + LDS reads only); 2 = MFMA only.  Results of ABL != 0 are meaningless by construction.
// AFF: some source carries a BatchNorm affine/activation; UP: some source is nearest-upsampled
// (both compile the corresponding loader work out when false: backward-data launches are <false,false>).
// THIN (VEC only): Cin is a power of two < 32 (4, 8, 16), single source: a 32-wide K chunk then spans
// 32/Cin taps and every thread's float4 column carries its OWN tap (per-thread instead of uniform).
// REUSE (VEC, 27 taps, 4 <= S <= BM): an M tile is a whole number of x-lines; in LDS every line is
// followed by a zero row, so the dx = -1/0/+1 taps of a (dz,dy) pair read the SAME staged A tile at
// row offsets -1/0/+1.  A tiles are then loaded, BN-transformed and stored once per 3 chunks.
// PAR (VEC, taps == 8): one of the 8 output-parity classes of a 3x3x3 conv over a nearest-upsampled
// input, run on the LOW-RES grid: class p = blockIdx.y = (pz,py,px) reads the 2x2x2 neighbourhood
// a + e + p - 1 (e in {0,1}^3) with pre-summed weights wp[p] and scatters row a to fine voxel 2a + p.
// accumulate: the epilogue adds the previous contents of out before bias / activation / statistics.
// FOLD (backward-data launches only): the epilogue also sums the NEXT layer's BatchNorm-backward terms over the tile
// (BwdStat) -- a compile-time variant so that the forward instantiations keep their register budget.
template <int WM, int WN, int TM, int TN, bool VEC, int ABL = 0, bool AFF = true, bool UP = true, bool THIN = false,
          bool REUSE = false, bool PAR = false, bool NOACT = false, bool FOLD = false>
__global__ __launch_bounds__(256, FOLD ? 2 : ((REUSE && PAR) ? 3 : 1)) void conv_fwd_kernel(ConvGeom g, ConvSrc s0, ConvSrc s1,
                                                        const float* __restrict__ wp,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ out, int ldo, int pre_act,
                                                        float* __restrict__ stat_partial, int gridM,
                                                        int gridN, int accumulate, BwdStat bs) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int B_FLOATS = 32 * BN;
  constexpr int RA = BM / 32;    // VEC: float4 A loads per thread per chunk
  constexpr int RS = BM / 8;     // SCALAR: scalar A loads per thread per chunk
  constexpr int NB = BN / 32;    // float4 B loads per thread per chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // REUSE: BM rows + one zero row after each of the BM/S lines + one leading zero row
  const int A_FLOATS = (REUSE ? BM + (BM >> g.lgS) + 1 : BM) * kLDA;
  float* As = smem;                      // [2][rows][36]
  constexpr int NABUF = 1;   // one A buffer, restaged behind a barrier: 52 instead of 70 KB of LDS -> 3 blocks/CU
  float* Bs = smem + NABUF * A_FLOATS;   // [2][8][BN][4]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int sw = xcd_swizzle(blockIdx.x, gridM * gridN);
  const int mb = sw / gridN, nb = sw % gridN;
  const int S = g.S, lg = g.lgS;
  const int M = g.B << (3 * lg);
  const int n0 = nb * BN;
  const int nchunks = g.Kpad >> 5;
  const int cpt = (VEC && !THIN) ? (g.Cin >> 5) : 1;   // 32-channel chunks per tap (VEC only)
  const int cls = PAR ? (int)blockIdx.y : 0;
  const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
  if (PAR) wp += (size_t)cls * g.Kpad * g.Npad;
  // CF_ZBATCH (launch_gemm_zbatch): gridDim.z INDEPENDENT plain GEMMs, one per z -- the 64 frequency images of a
  // Winograd-domain evaluation (conv_winog.hip): A, the packed weights and the output advance by whole matrices; no split-K
  const bool zbatch = (g.flags & CF_ZBATCH) != 0;
  if (zbatch) {
    s0.p += (size_t)blockIdx.z * (size_t)M * (size_t)s0.C;
    wp += (size_t)blockIdx.z * (size_t)g.Kpad * (size_t)g.Npad;
    out += (size_t)blockIdx.z * (size_t)M * (size_t)ldo;
  }
  const int zsplit = zbatch ? 1 : (int)gridDim.z, zidx = zbatch ? 0 : (int)blockIdx.z;

  // ---- per-thread row bookkeeping
  const int mrow_base = mb * BM + (VEC ? (t >> 3) : (t >> 5));   // + 32*r (VEC) / 8*r (SCALAR)
  const float slope0 = act_slope_of(s0.act), slope1 = act_slope_of(s1.act);
  const float pre_slope = act_slope_of(pre_act);

  // VEC path per-row state, computed once: bits 0..26 = "tap t reads inside the grid", bits 27..29 =
  // parity of (z,y,x); rvh = voxel index of the row in a half-resolution (upsampled) source.
  // Per chunk the same-resolution voxel index is simply m + (dz*S+dy)*S+dx.
  unsigned rmask[RA];
  int rvh[RA];
  if (VEC) {
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int m = mrow_base + 32 * r;
      const RowPos rp = decode_row(m, S, lg);
      unsigned mk = 0;
      if (m < M) {
        if (PAR) {
#pragma unroll
          for (int tp = 0; tp < 8; ++tp) {
            const int zz = rp.z + (tp >> 2) + pz - 1, yy = rp.y + ((tp >> 1) & 1) + py - 1, xx = rp.x + (tp & 1) + px - 1;
            const bool ok = (unsigned)zz < (unsigned)S && (unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S;
            mk |= (ok ? 1u : 0u) << tp;
          }
#pragma unroll
          for (int gz = 0; gz < 4; ++gz) {   // bits 8..11: the (ez,ey) pair reads inside the grid at dx = 0
            const int zz = rp.z + (gz >> 1) + pz - 1, yy = rp.y + (gz & 1) + py - 1;
            mk |= (((unsigned)zz < (unsigned)S && (unsigned)yy < (unsigned)S) ? 1u : 0u) << (8 + gz);
          }
        } else if (g.taps == 27) {
          const unsigned zm = (rp.z > 0 ? 1u : 0u) | 2u | (rp.z < S - 1 ? 4u : 0u);
          const unsigned ym = (rp.y > 0 ? 1u : 0u) | 2u | (rp.y < S - 1 ? 4u : 0u);
          const unsigned xm = (rp.x > 0 ? 1u : 0u) | 2u | (rp.x < S - 1 ? 4u : 0u);
#pragma unroll
          for (int tp = 0; tp < 27; ++tp)
            mk |= (((zm >> (tp / 9)) & (ym >> ((tp / 3) % 3)) & (xm >> (tp % 3))) & 1u) << tp;
        } else {
          mk = 1u;
        }
      }
      mk |= ((unsigned)(rp.z & 1) << 29) | ((unsigned)(rp.y & 1) << 28) | ((unsigned)(rp.x & 1) << 27);
      rmask[r] = mk;
      const int Sh = S >> 1;
      rvh[r] = ((rp.b * Sh + (rp.z >> 1)) * Sh + (rp.y >> 1)) * Sh + (rp.x >> 1);
    }
  }

  v4f ra4[RA];
  float ras[RS];
  v4f rb[NB];

  // ---- B tile of K rows [krow4*4, krow4*4+32): packed weights [Kpad/4][Npad][4].
  // Address = wave-uniform row pointer (scalar registers) + a per-thread byte offset fixed for the whole kernel:
  // no vector address arithmetic per chunk (the 64-bit mads it replaced cost ~2 % of the MFMA issue slots -
  // fp32 VALU work and fp32 MFMAs serialise on a SIMD, scripts/probes/pipe_probe.hip).
  unsigned bofs[NB];
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const int idx = t + 256 * r;
    bofs[r] = (unsigned)((idx / BN) * g.Npad + n0 + idx % BN) * 16u;
  }
  auto load_b = [&](int krow4) {
    const char* wrow = reinterpret_cast<const char*>(wp) + (size_t)krow4 * (size_t)g.Npad * 16;
#pragma unroll
    for (int r = 0; r < NB; ++r) rb[r] = *reinterpret_cast<const v4f*>(wrow + bofs[r]);
  };
  auto store_b = [&](int buf) {
    float* Bw = Bs + buf * B_FLOATS;
#pragma unroll
    for (int r = 0; r < NB; ++r) *reinterpret_cast<v4f*>(Bw + (t + 256 * r) * 4) = rb[r];
  };
  // REUSE: A tile of group G = (dz*3+dy)*cpt + cc, i.e. the dx = 0 rows of 32 channels
  // (PAR: G = (ez*2+ey)*cpt + cc with dz = ez+pz-1, dy = ey+py-1; two dx chunks per group instead of three)
  unsigned rowB0[RA], rowB1[RA];   // byte offset of this thread's rows in source 0 / 1 (same-resolution sources)
  if (VEC) {
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      rowB0[r] = (unsigned)(mrow_base + 32 * r) * (unsigned)s0.C * 4u;
      rowB1[r] = (unsigned)(mrow_base + 32 * r) * (unsigned)s1.C * 4u;
    }
  }
  auto load_a_group = [&](int gzy, int cc) {
    const int ci0 = cc << 5;
    const int dz = PAR ? (gzy >> 1) + pz - 1 : gzy / 3 - 1, dy = PAR ? (gzy & 1) + py - 1 : gzy % 3 - 1;
    const bool first = ci0 < s0.C;
    const char* sp = reinterpret_cast<const char*>(first ? s0.p : s1.p);
    const float* sscale = first ? s0.scale : s1.scale;
    const float* sshift = first ? s0.shift : s1.shift;
    const int sC = first ? s0.C : s1.C, su = first ? s0.up : s1.up;
    const float slope = first ? slope0 : slope1;
    const int cl = (first ? ci0 : ci0 - s0.C) + (t & 7) * 4;
    const v4f sc = *reinterpret_cast<const v4f*>(sscale + cl);
    const v4f sh = *reinterpret_cast<const v4f*>(sshift + cl);
    const int sdelta = (dz * S + dy) * S;
    const unsigned goffb = (unsigned)(sdelta * sC + cl) * 4u;     // wraps for negative shifts; rowB + goffb is exact
    const int Sh = S >> 1;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const unsigned mk = rmask[r];
      const bool inb = (mk >> (PAR ? 8 + gzy : gzy * 3 + 1)) & 1u;   // validity of the centre (dx = 0) tap
      unsigned offb = (first ? rowB0[r] : rowB1[r]) + goffb;
      if (UP) {
        const int ez = (dz + (int)((mk >> 29) & 1u)) >> 1;
        const int ey = (dy + (int)((mk >> 28) & 1u)) >> 1;
        const unsigned up_b = ((unsigned)(rvh[r] + (ez * Sh + ey) * Sh) * (unsigned)sC + (unsigned)cl) * 4u;
        offb = su ? up_b : offb;
      }
      offb = inb ? offb : (unsigned)cl * 4u;
      v4f v = *reinterpret_cast<const v4f*>(sp + offb);
      if (AFF) v = affine_only_or_act4<NOACT>(v, sc, sh, slope);
      ra4[r] = inb ? v : v4f{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_a_group = [&](int buf) {
    float* A = As + buf * A_FLOATS;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int row = (t >> 3) + 32 * r;
      *reinterpret_cast<v4f*>(A + (row + (row >> lg) + 1) * kLDA + (t & 7) * 4) = ra4[r];
    }
  };

  auto load_chunk = [&](int c) {
    load_b(c * 8);
    // ---- A tile
    if (VEC) {
      int tap, ci0;
      bool kvalid = true;
      if (THIN) {
        const int kq = (c << 5) + (t & 7) * 4;          // this thread's flattened k = tap*Cin + ci
        const int lgC = 31 - __clz(g.Cin);
        tap = kq >> lgC; ci0 = kq & (g.Cin - 1);
        kvalid = tap < g.taps;                           // K is padded up to a multiple of 32
      } else {
        tap = c / cpt; ci0 = (c - tap * cpt) << 5;
      }
      const bool t27 = g.taps == 27;
      int dz = t27 ? tap / 9 - 1 : 0, dy = t27 ? (tap / 3) % 3 - 1 : 0, dx = t27 ? tap % 3 - 1 : 0;
      if (PAR) { dz = (tap >> 2) + pz - 1; dy = ((tap >> 1) & 1) + py - 1; dx = (tap & 1) + px - 1; }
      const bool first = THIN ? true : (ci0 < s0.C);
      const float* sp = first ? s0.p : s1.p;
      const float* sscale = first ? s0.scale : s1.scale;
      const float* sshift = first ? s0.shift : s1.shift;
      const int sC = first ? s0.C : s1.C, su = first ? s0.up : s1.up;
      const float slope = first ? slope0 : slope1;
      const int cl = THIN ? ci0 : (first ? ci0 : ci0 - s0.C) + (t & 7) * 4;
      const v4f sc = *reinterpret_cast<const v4f*>(sscale + cl);
      const v4f sh = *reinterpret_cast<const v4f*>(sshift + cl);
      const int sdelta = (dz * S + dy) * S + dx;   // scalar: same-resolution voxel index shift
      const int Sh = S >> 1;
#pragma unroll
      for (int r = 0; r < RA; ++r) {
        const unsigned mk = rmask[r];
        const bool inb = kvalid && ((mk >> (tap & 31)) & 1u);
        int idx = mrow_base + 32 * r + sdelta;
        if (UP) {
          // (z+dz)>>1 = (z>>1) + ((dz + (z&1)) >> 1), likewise y, x
          const int ez = (dz + (int)((mk >> 29) & 1u)) >> 1;
          const int ey = (dy + (int)((mk >> 28) & 1u)) >> 1;
          const int ex = (dx + (int)((mk >> 27) & 1u)) >> 1;
          const int idx_up = rvh[r] + (ez * Sh + ey) * Sh + ex;
          idx = su ? idx_up : idx;
        }
        const unsigned off = inb ? (unsigned)idx * (unsigned)sC + cl : (unsigned)cl;
        if (ABL == 3) {   // trivial addressing, no affine/act/select: isolates the address-math cost
          ra4[r] = *reinterpret_cast<const v4f*>(sp + (((unsigned)(mrow_base + 32 * r) * sC + c * 32 + cl) & 0xffffffu));
          continue;
        }
        if (ABL == 8) {   // real (clamp-free) addresses, no zero-select
          ra4[r] = *reinterpret_cast<const v4f*>(sp + ((unsigned)max(idx, 0) * (unsigned)sC + cl));
          continue;
        }
        if (ABL == 9) {   // dense addresses + the real mask/select
          const v4f vv = *reinterpret_cast<const v4f*>(sp + (((unsigned)(mrow_base + 32 * r) * sC + c * 32 + cl) & 0xffffffu));
          ra4[r] = inb ? vv : v4f{0.f, 0.f, 0.f, 0.f};
          continue;
        }
        v4f v = *reinterpret_cast<const v4f*>(sp + off);
        if (AFF && ABL != 4) v = affine_act4(v, sc, sh, slope);
        ra4[r] = inb ? v : v4f{0.f, 0.f, 0.f, 0.f};   // zero "same" padding applies after BN/act
      }
    } else {
      const int kf = (c << 5) + (t & 31);
      const int tap = kf / g.Cin, ci = kf - tap * g.Cin;
      const bool kvalid = tap < g.taps;
      int dz = 0, dy = 0, dx = 0;
      if (g.taps == 27) { dz = tap / 9 - 1; dy = (tap / 3) % 3 - 1; dx = tap % 3 - 1; }
#pragma unroll
      for (int r = 0; r < RS; ++r) {
        const int m = mrow_base + 8 * r;
        const RowPos rp = decode_row(m, S, lg);
        const int zz = rp.z + dz, yy = rp.y + dy, xx = rp.x + dx;
        const bool inb = kvalid && m < M && (unsigned)zz < (unsigned)S &&
                         (unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S;
        ras[r] = inb ? gather_scalar(s0, s1, ci, rp.b, zz, yy, xx, S) : 0.f;
      }
    }
  };

  // taps == 1 over same-resolution sources is a plain [M][K] x [K][N] GEMM (dense layers, the 1x1x1 heads, the
  // coarse-grid GEMMs of the up-split backward): row byte offsets are fixed, the channel offset is wave-uniform,
  // and rows >= M just re-read row M-1 (their outputs are never stored): no address or mask arithmetic per chunk.
  unsigned rowG0[RA], rowG1[RA];
  if (VEC) {
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const unsigned mc = (unsigned)min(mrow_base + 32 * r, M - 1);
      rowG0[r] = (mc * (unsigned)s0.C + (unsigned)(t & 7) * 4u) * 4u;
      rowG1[r] = (mc * (unsigned)s1.C + (unsigned)(t & 7) * 4u) * 4u;
    }
  }
  auto load_chunk_gemm = [&](int c) {
    load_b(c * 8);
    const int ci0 = c << 5;
    const bool first = ci0 < s0.C;
    const int cbase = first ? ci0 : ci0 - s0.C;
    const char* sp = reinterpret_cast<const char*>(first ? s0.p : s1.p) + (size_t)cbase * 4;
    const float slope = first ? slope0 : slope1;
    v4f sc = v4f{1.f, 1.f, 1.f, 1.f}, sh = v4f{0.f, 0.f, 0.f, 0.f};
    if (AFF) {
      sc = *reinterpret_cast<const v4f*>((first ? s0.scale : s1.scale) + cbase + (t & 7) * 4);
      sh = *reinterpret_cast<const v4f*>((first ? s0.shift : s1.shift) + cbase + (t & 7) * 4);
    }
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      v4f v = *reinterpret_cast<const v4f*>(sp + (first ? rowG0[r] : rowG1[r]));
      if (AFF) v = affine_only_or_act4<NOACT>(v, sc, sh, slope);
      ra4[r] = v;
    }
  };
  auto store_a_chunk = [&]() {
    float* A = As;
    if (VEC) {
#pragma unroll
      for (int r = 0; r < RA; ++r)
        *reinterpret_cast<v4f*>(A + ((t >> 3) + 32 * r) * kLDA + (t & 7) * 4) = ra4[r];
    } else {
#pragma unroll
      for (int r = 0; r < RS; ++r) A[((t >> 5) + 8 * r) * kLDA + (t & 31)] = ras[r];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // accumulate: the accumulators START from the previous contents of out (the parity-class pass of an up-split
  // layer), so the read's latency hides under the prologue instead of stalling the epilogue
  if (accumulate && !PAR && zsplit == 1) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * TN * 32 + j * 32 + li;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb * BM + wm * TM * 32 + 4 * lh + i * 32 + (r & 3) + 8 * (r >> 2);
          if (m < M && n < g.Cout) acc[i][j][r] = out[(size_t)m * ldo + n];
        }
    }
  }

  // generic MFMA stream over one staged chunk; arow0 = LDS row of this lane's first A row
  auto compute_at = [&](const float* Abase, int arow_stride_rows, int bbuf) {
    const float* Bw = Bs + bbuf * B_FLOATS + (lh * BN + wn * TN * 32 + li) * 4;
#pragma unroll
    for (int g8 = 0; g8 < 4; ++g8) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const float4*>(Abase + i * arow_stride_rows * kLDA + g8 * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(Bw + (g8 * 2 * BN + j * 32) * 4);
      // rotate over the TM*TN accumulators inside each k step: the next MFMA on the same
      // accumulator is TM*TN issues away, so interleaved VALU/LDS never sits between dependents
#pragma unroll
      for (int tk = 0; tk < 4; ++tk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float av = tk == 0 ? a[i].x : tk == 1 ? a[i].y : tk == 2 ? a[i].z : a[i].w;
            const float bv = tk == 0 ? b[j].x : tk == 1 ? b[j].y : tk == 2 ? b[j].z : b[j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
    }
  };

  if (REUSE) {
    // ---- dx-reuse pipeline: groups G = ((dz*3+dy), 32-channel slice); chunks c = NDX*G + i, i = dx index
    constexpr int NDX = PAR ? 2 : 3;
    const int nG = (PAR ? 4 : 9) * cpt;
    const int dx_first = PAR ? px - 1 : -1;                        // dx of chunk i is dx_first + i
    const int lines = BM >> lg;
    for (int i = t; i < NABUF * (lines + 1) * kLDA; i += 256) {      // the separator rows stay zero
      const int b = i / ((lines + 1) * kLDA), rem = i - b * (lines + 1) * kLDA;
      As[b * A_FLOATS + (rem / kLDA) * (S + 1) * kLDA + rem % kLDA] = 0.f;
    }
    // this lane's A rows inside the padded image (a 32-row MFMA tile may span lines when S < 32, so every
    // tile row gets its own padded index)
    int arow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * TM * 32 + i * 32 + li;
      arow[i] = row + (row >> lg) + 1;
    }
    auto b_rows = [&](int gzy, int cc, int dxi) {   // first packed K row (in units of 4) of a chunk
      return (((gzy * NDX + dxi) * g.Cin + (cc << 5)) >> 2);
    };
    auto compute_reuse = [&](int abuf, int bbuf, int dx) {
      const float* Bw = Bs + bbuf * B_FLOATS + (lh * BN + wn * TN * 32 + li) * 4;
      const float* A = As + abuf * A_FLOATS + lh * 4 + dx * kLDA;
#pragma unroll
      for (int g8 = 0; g8 < 4; ++g8) {
        float4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(A + arow[i] * kLDA + g8 * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(Bw + (g8 * 2 * BN + j * 32) * 4);
#pragma unroll
        for (int tk = 0; tk < 4; ++tk)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const float av = tk == 0 ? a[i].x : tk == 1 ? a[i].y : tk == 2 ? a[i].z : a[i].w;
              const float bv = tk == 0 ? b[j].x : tk == 1 ? b[j].y : tk == 2 ? b[j].z : b[j].w;
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
            }
      }
    };
    // a tile inside one z-plane (BM <= S*S) at the z = 0 / S-1 face reads only padding through the
    // dz = -1 / +1 groups: skip them (block-uniform loop bounds; 2/(3S) of the MFMA work)
    int G0 = 0, G1 = nG;
    if (!PAR && BM <= (S << lg)) {
      const int zb = ((mb * BM) >> (2 * lg)) & (S - 1);
      if (zb == 0) G0 = 3 * cpt;
      if (zb == S - 1) G1 = 6 * cpt;
    }
    if (zsplit > 1) {   // split-K: this block's slice of the groups; partial sums go to the workspace
      const int per = (nG + zsplit - 1) / zsplit;
      G0 = max(G0, zidx * per);
      G1 = min(G1, (zidx + 1) * per);
    }
    if (G0 < G1) {
    // (gzy, cc) of the current / next group are carried incrementally: no integer division in the loop
    int gzy = G0 / cpt, cc = G0 - gzy * cpt;
    int par = 0;                                   // B buffer holding the current chunk
    load_a_group(gzy, cc);
    store_a_group(0);
    load_b(b_rows(gzy, cc, 0));
    store_b(par);
    __syncthreads();
    for (int G = G0; G + 1 < G1; ++G) {
      int ngzy = gzy, ncc = cc + 1;
      if (ncc == cpt) { ncc = 0; ++ngzy; }
      // first dx chunk: also fetch the next group's A rows (kept in registers until the group is done)
      load_a_group(ngzy, ncc);
#pragma unroll
      for (int i = 0; i < NDX; ++i) {
        load_b(i + 1 < NDX ? b_rows(gzy, cc, i + 1) : b_rows(ngzy, ncc, 0));
        compute_reuse(0, par, dx_first + i);
        par ^= 1;
        store_b(par);
        __syncthreads();
      }
      // restage the (single) A buffer with the next group's tile
      store_a_group(0);
      __syncthreads();
      gzy = ngzy; cc = ncc;
    }
    {
#pragma unroll
      for (int i = 0; i + 1 < NDX; ++i) {
        load_b(b_rows(gzy, cc, i + 1));
        compute_reuse(0, par, dx_first + i);
        par ^= 1;
        store_b(par);
        __syncthreads();
      }
      compute_reuse(0, par, dx_first + NDX - 1);
      __syncthreads();
    }
    }   // G0 < G1
  } else {
  int cb = 0, ce = nchunks;
  if (zsplit > 1) {
    const int per = (nchunks + zsplit - 1) / zsplit;
    cb = zidx * per;
    ce = min(nchunks, cb + per);
  }
  const bool gemm = VEC && !THIN && !PAR && !UP && ABL == 0 && g.taps == 1;   // block-uniform
  if (cb < ce && gemm) {
    auto compute_g = [&](int buf) { compute_at(As + (wm * TM * 32 + li) * kLDA + lh * 4, 32, buf); };
    load_chunk_gemm(cb);
    store_b(cb & 1);
    store_a_chunk();
    __syncthreads();
#if ICS_GEMM_DEEP
    if (!AFF && TM * TN >= ICS_GEMM_DEEP_FWD_MIN) {
      // two register sets that swap roles (loop unrolled by two): chunk c + 1 is in set A when an iteration starts,
      // chunk c + 2 is requested into set B.  (First version: one set copied into the other, 16 64-bit moves per chunk.)
      v4f raA[RA], rbA[NB], raB[RA], rbB[NB];
      auto load_set = [&](int c, v4f (&xa)[RA], v4f (&xb)[NB]) {
        const char* wrow = reinterpret_cast<const char*>(wp) + (size_t)(c * 8) * (size_t)g.Npad * 16;
#pragma unroll
        for (int r = 0; r < NB; ++r) xb[r] = *reinterpret_cast<const v4f*>(wrow + bofs[r]);
        const int ci0 = c << 5;
        const bool first = ci0 < s0.C;
        const char* sp = reinterpret_cast<const char*>(first ? s0.p : s1.p) + (size_t)(first ? ci0 : ci0 - s0.C) * 4;
#pragma unroll
        for (int r = 0; r < RA; ++r) xa[r] = *reinterpret_cast<const v4f*>(sp + (first ? rowG0[r] : rowG1[r]));
      };
      auto store_set = [&](int buf, const v4f (&xa)[RA], const v4f (&xb)[NB]) {   // B now, A behind the barrier
        float* Bw = Bs + buf * B_FLOATS;
#pragma unroll
        for (int r = 0; r < NB; ++r) *reinterpret_cast<v4f*>(Bw + (t + 256 * r) * 4) = xb[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RA; ++r) *reinterpret_cast<v4f*>(As + ((t >> 3) + 32 * r) * kLDA + (t & 7) * 4) = xa[r];
        __syncthreads();
      };
      const int lastc = ce - 1;
      if (cb + 1 < ce) load_set(cb + 1, raA, rbA);
      int c = cb;
      for (; c + 2 < ce; c += 2) {
        load_set(c + 2, raB, rbB);
        compute_g(c & 1);
        store_set((c + 1) & 1, raA, rbA);
        load_set(c + 3 < ce ? c + 3 : lastc, raA, rbA);    // past the end: the last chunk again (never stored)
        compute_g((c + 1) & 1);
        store_set(c & 1, raB, rbB);                        // chunk c + 2
      }
      if (c + 1 < ce) {                                    // chunk c + 1 is still in set A
        compute_g(c & 1);
        store_set((c + 1) & 1, raA, rbA);
      }
      compute_g((ce - 1) & 1);
      __syncthreads();
    } else
#endif
    {
    for (int c = cb; c + 1 < ce; ++c) {
      load_chunk_gemm(c + 1);
      compute_g(c & 1);
      store_b((c + 1) & 1);
      __syncthreads();
      store_a_chunk();
      __syncthreads();
    }
    compute_g((ce - 1) & 1);
    __syncthreads();
    }
  } else if (cb < ce) {
  load_chunk(cb);
  store_b(cb & 1);
  store_a_chunk();
  __syncthreads();

  auto compute = [&](int buf) {
    compute_at(As + (wm * TM * 32 + li) * kLDA + lh * 4, 32, buf);
  };
  auto compute_regs_only = [&]() {
    float av = (float)li, bv = (float)lh;
#pragma unroll
    for (int g8 = 0; g8 < 16; ++g8)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
  };
  if (ABL == 0 || ABL >= 3) {
    for (int c = cb; c + 1 < ce; ++c) {
      // (sched_barrier fences around the MFMA stream were measured: -4 %; hipcc's own interleave wins)
      load_chunk(c + 1);
      compute(c & 1);
      store_b((c + 1) & 1);
      __syncthreads();
      store_a_chunk();
      __syncthreads();
    }
  } else {
    for (int c = 0; c + 1 < nchunks; ++c) {
      if (ABL == 1) compute(c & 1);
      else compute_regs_only();
      __syncthreads();
    }
  }
  compute((ce - 1) & 1);
  __syncthreads();
  }   // cb < ce
  }   // !REUSE
  if (zsplit > 1) out += (size_t)zidx * (size_t)M * ldo;

  // ---- epilogue: bias + activation, store, per-block BatchNorm partial statistics
  const int mrow0 = mb * BM + wm * TM * 32 + 4 * lh;
  if (FOLD) {
    // backward-data tile = dO of the producer layer P: store it and add this tile's share of P's BatchNorm-backward
    // sums  S1 = sum d,  S2 = sum d * xhat  (d = dO * post_act'(BN(s)), xhat = (s - mean) * rstd).  P's activations
    // are fetched 16 rows at a time (the compiler barrier keeps at most 16 loads in flight: register budget).
    float* red1 = smem;               // [WM][BN]
    float* red2 = smem + WM * BN;     // [WM][BN]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * TN * 32 + j * 32 + li;
      const bool nvalid = n < g.Cout;
      const int nc = nvalid ? n : 0;
      const float b_mu = bs.mean[nc], b_rs = bs.rstd[nc];
      const float b_sc = bs.post_act != ACT_NONE ? bs.scale[nc] : 1.f;
      const float b_sh = bs.post_act != ACT_NONE ? bs.shift[nc] : 0.f;
      float a1 = 0.f, a2 = 0.f;
      // 32-bit byte offsets from wave-uniform bases (sources are < 4 GiB, checked at launch): one VGPR per address
      const char* sbase = reinterpret_cast<const char*>(bs.s);
      char* obase = reinterpret_cast<char*>(out);
      const unsigned s_ldb = (unsigned)bs.ld * 4u, o_ldb = (unsigned)ldo * 4u, colb = (unsigned)nc * 4u;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned m = (unsigned)(mrow0 + i * 32 + (r & 3) + 8 * (r >> 2));
          sv[r] = *reinterpret_cast<const float*>(sbase + min(m, (unsigned)(M - 1)) * s_ldb + colb);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned m = (unsigned)(mrow0 + i * 32 + (r & 3) + 8 * (r >> 2));
          const bool ok = m < (unsigned)M && nvalid;
          const float v = acc[i][j][r];
          const float d = ok ? v * act_grad(fmaf(sv[r], b_sc, b_sh), bs.post_act) : 0.f;
          a1 += d;
          a2 = fmaf(d, (sv[r] - b_mu) * b_rs, a2);
          if (ok) *reinterpret_cast<float*>(obase + m * o_ldb + colb) = v;
        }
        asm volatile("" ::: "memory");
      }
      a1 += __shfl_xor(a1, 32);
      a2 += __shfl_xor(a2, 32);
      if (lh == 0) {
        red1[wm * BN + wn * TN * 32 + j * 32 + li] = a1;
        red2[wm * BN + wn * TN * 32 + j * 32 + li] = a2;
      }
    }
    __syncthreads();
    if (t < BN) {
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { a1 += red1[w * BN + t]; a2 += red2[w * BN + t]; }
      float* sp = bs.partial + (size_t)(n0 + t) * gridM + mb;
      sp[0] = a1;
      sp[(size_t)g.Npad * gridM] = a2;
    }
    return;
  }
  float colsum[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * TN * 32 + j * 32 + li;
    const bool nvalid = n < g.Cout;
    const float bv = (bias != nullptr && nvalid) ? bias[n] : 0.f;
    float cs = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
        const bool ok = m < M;
        size_t mo = m;
        if (PAR) {
          const RowPos rp = decode_row(m, S, lg);
          mo = ((((size_t)rp.b * (2 * S) + 2 * rp.z + pz) * (2 * S) + 2 * rp.y + py) * (2 * S)) + 2 * rp.x + px;
        }
        float a = acc[i][j][r] + bv;
        if (PAR && accumulate && ok && nvalid) a += out[mo * ldo + n];
        float v = act_apply(a, pre_slope);
        if (!ok) v = 0.f;
        acc[i][j][r] = v;
        cs += v;
        if (ok && nvalid) out[mo * ldo + n] = v;
      }
    colsum[j] = cs;
  }
  if (stat_partial == nullptr) return;

  // block-level (count, mean, M2) per column, two-pass inside the block -> exact Chan merge later
  float* red = smem;              // [WM][BN]
  float* bmean = smem + WM * BN;  // [BN]
  const int nvalid_rows = min(BM, M - mb * BM);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    float cs = colsum[j];
    cs += __shfl_xor(cs, 32);
    if (lh == 0) red[wm * BN + wn * TN * 32 + j * 32 + li] = cs;
  }
  __syncthreads();
  if (t < BN) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) s += red[w * BN + t];
    bmean[t] = s / (float)nvalid_rows;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const float mu = bmean[wn * TN * 32 + j * 32 + li];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mrow0 + i * 32 + (r & 3) + 8 * (r >> 2);
        const float d = acc[i][j][r] - mu;
        q += (m < M) ? d * d : 0.f;
      }
    q += __shfl_xor(q, 32);
    colsum[j] = q;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < TN; ++j)
    if (lh == 0) red[wm * BN + wn * TN * 32 + j * 32 + li] = colsum[j];
  __syncthreads();
  if (t < BN) {
    float q = 0.f;
#pragma unroll
    for (int w = 0; w < WM; ++w) q += red[w * BN + t];
    // layout [3][Npad][nstat] (block index fastest): bn_finalize then reads each channel's partials contiguously
    const size_t nstat = (size_t)(PAR ? 8 : 1) * gridM;                             // PAR: 8 x gridM blocks
    float* sp = stat_partial + (size_t)(n0 + t) * nstat + (size_t)(cls * gridM + mb);
    sp[0] = (float)nvalid_rows;
    sp[(size_t)g.Npad * nstat] = bmean[t];
    sp[(size_t)2 * g.Npad * nstat] = q;
  }
}

template <int WM, int WN, int TM, int TN, bool VEC, int ABL = 0, bool AFF = true, bool UP = true, bool THIN = false,
          bool REUSE = false, bool PAR = false, bool NOACT = false, bool FOLD = false>
static int launch_fwd_cfg(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const ConvSrc& s1,
                          const float* wp, const float* bias, float* out, int ldo, int pre_act,
                          float* stat_partial, int* rows_per_block, int accumulate = 0, int ksplit = 1,
                          const BwdStat* bwd = nullptr) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int M = g.B << (3 * g.lgS);
  const int gridM = (M + BM - 1) / BM, gridN = g.Npad / BN;
  const int arows = REUSE ? BM + (BM >> g.lgS) + 1 : BM;
  const size_t lds = (size_t)(arows * kLDA + 2 * 32 * BN) * sizeof(float);
  auto kern = conv_fwd_kernel<WM, WN, TM, TN, VEC, ABL, AFF, UP, THIN, REUSE, PAR, NOACT, FOLD>;
  static DevOnce attr;
  int dev;
  if (attr.need(&dev)) {
    const size_t lds_max = (size_t)((REUSE ? BM + BM / 4 + 1 : BM) * kLDA + 2 * 32 * BN) * sizeof(float);
    ICS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    attr.mark(dev);
  }
  if (rows_per_block) *rows_per_block = BM;
  static const std::string id = std::string("conv_fwd_kernel<") + std::to_string(WM) + ", " + std::to_string(WN) + ", " +
                                std::to_string(TM) + ", " + std::to_string(TN) + ", " + tf(VEC) + ", " +
                                std::to_string(ABL) + ", " + tf(AFF) + ", " + tf(UP) + ", " + tf(THIN) + ", " +
                                tf(REUSE) + ", " + tf(PAR) + ", " + tf(NOACT) + ", " + tf(FOLD) + ">";
  g_last_kernel_id = id.c_str();
  BwdStat bs_arg;
  if (FOLD) bs_arg = *bwd;
  ICS_LAUNCH(kern, dim3(gridM * gridN, PAR ? 8 : 1, ksplit), dim3(256), lds, st, g, s0, s1, wp, bias, out, ldo,
                     pre_act, stat_partial, gridM, gridN, accumulate, bs_arg);
  ICS_HIP(hipGetLastError());
  return 0;
}

// tile choice for a geometry. Available layouts (4 waves): BM=128 x BN in {128,96,64,32}; 64x64.
static void pick_fwd_tile(const ConvGeom& g, int* bm, int* bn) {
  const int M = g.B << (3 * g.lgS);
  int BN = (g.Npad % 128 == 0) ? 128 : (g.Npad % 96 == 0) ? 96 : (g.Npad % 64 == 0) ? 64 : 32;
  int BM = 128;
  // keep >= ~1.5 blocks per CU in flight on the low-resolution layers
  if (g.Npad % 64 == 0 && (long)((M + 127) / 128) * (g.Npad / BN) < 384) {
    BM = 64;
    BN = 64;
  }
  *bm = BM;
  *bn = BN;
}

static bool fwd_is_vec(const ConvGeom& g, const ConvSrc& s0, const ConvSrc& s1) {
  return (g.Cin % 32 == 0) && (s0.C % 32 == 0) && s0.bcast == 0 && s1.bcast == 0;
}
// thin vector path: one source, Cin in {4, 8, 16}
static bool conv_is_thin(const ConvGeom& g, const ConvSrc& s0, int nsrc) {
  return nsrc == 1 && s0.bcast == 0 && s0.C == g.Cin && (g.Cin == 4 || g.Cin == 8 || g.Cin == 16);
}


// nz independent plain GEMMs out_z[M][N] = A_z[M][K] x W_z[K][N] (A row-major, W in the packed [K/4][N][4] layout, both
// advancing by whole matrices per z) on the 64 x 64 tile of conv_fwd_kernel: the frequency images of conv_winog.hip
int launch_gemm_zbatch(hipStream_t st, int nz, int M, int K, int N, const float* A, const float* Wp, float* out, int flags) {
  ICS_CHECK(nz >= 1 && nz <= 65535 && M >= 1 && K % 32 == 0 && N % 64 == 0, "z-batched GEMM: K % 32, N % 64 required");
  ICS_CHECK((size_t)M * (size_t)std::max(K, N) < ((size_t)1 << 30), "z-batched GEMM: matrix too large for 32-bit offsets");
  const ConvGeom g{M, 1, 0, K, N, 1, K, N, flags | CF_ZBATCH};
  const ConvSrc s0{A, nullptr, nullptr, K, 0, ACT_NONE, 0}, s1{};
  // 128 x 128 tiles when they still give every CU two workgroups (ICS_ZB_BIG=0: always 64 x 64)
#ifndef ICS_ZB_BIG
#define ICS_ZB_BIG 1
#endif
  if (ICS_ZB_BIG && M % 128 == 0 && N % 128 == 0 && (long)(M / 128) * (N / 128) * nz >= 512)
    return launch_fwd_cfg<2, 2, 2, 2, true, 0, false, false, false, false, false, true, false>(
        st, g, s0, s1, Wp, nullptr, out, N, ACT_NONE, nullptr, nullptr, 0, nz);
  return launch_fwd_cfg<2, 2, 1, 1, true, 0, false, false, false, false, false, true, false>(
      st, g, s0, s1, Wp, nullptr, out, N, ACT_NONE, nullptr, nullptr, 0, nz);
}

int conv_fwd_rows_per_block(const ConvGeom& g) {
  int bm, bn;
  pick_fwd_tile(g, &bm, &bn);
  // the thin-N direct kernel (Cout <= 4) has 256 / (Cin/4) voxels per block; size for the smaller of the two
  if (g.taps == 27 && g.Cout <= 4 && g.Cin % 4 == 0 && g.Cin >= 4 && g.Cin <= 64) {
    bm = std::min(bm, g.Cin <= 16 ? 256 : 64);
  }
  return bm;
}

// =====================================================================================
// Thin-C direct forward kernel: few input channels (1, 4, 16) into 16 or 32 output channels at full resolution -- the
// first convs of both nets (U-Net c1: 32^3 x {1,4} -> 32, VAE e0: 32^3 x 11(16) -> 16, e1: 16^3 x 16 -> 32).  These
// layers are HBM-bound (AI 13-25 FLOP/B): as implicit GEMMs they ran at 14-28 TFLOP/s with most of a
// 32-wide MFMA tile being padding.  Here: a 3x3x3 stencil on the vector ALU, weights staged in LDS; a thread owns
// 4 consecutive x voxels x 4 output channels (16 accumulators), so each weight float4 read from LDS feeds 16 FMAs
// and each (dz,dy) row needs 6 input positions instead of 12.  LPV = COUT/4 lanes share a voxel group: a wave's
// stores cover whole 64/128-byte output rows.  Epilogue: bias, activation, store, per-block BatchNorm partials.
// CIN = channels read per voxel; wstride = channels per tap in the packed weights (>= CIN: c1 at C = 1 reads the
// un-padded input but the weights are packed for the 4-channel padded problem the backward-weight kernel uses).
// =====================================================================================
#ifndef ICS_THINC_UNROLL
#define ICS_THINC_UNROLL 3   // (dz, dy) rows whose loads are in flight together
#endif
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_thin_c_fwd_kernel(ConvGeom g, const float* __restrict__ x, int wstride,
                                                               const float* __restrict__ wp,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ out, int ldo, int pre_act,
                                                               float* __restrict__ stat_partial,
                                                               const float* __restrict__ pos_bias) {
  constexpr int LPV = COUT / 4;            // lanes per voxel group
  constexpr int GPB = 256 / LPV;           // voxel groups (of 4 voxels) per block
  constexpr int VPB = 4 * GPB;             // voxels per block
  constexpr int C4 = CIN >= 4 ? CIN / 4 : 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w = smem;                         // [27 * wstride][COUT]
  const int t = threadIdx.x;
  const int S = g.S, lg = g.lgS;
  const int M = g.B << (3 * lg);
  // CIN < 4 reads only channel 0 of each tap's wstride packed rows: stage those 27 rows (a block's staging is a chain of
  // dependent L2 round trips in front of its barrier: 14 per thread for all 27 x wstride rows, 4 for these)
  constexpr bool WCH = CIN < 4;
  const int wrow = WCH ? 1 : wstride;
  const int K = 27 * wrow;
  for (int i = t; i < K * COUT; i += 256) {
    const int kk = i / COUT, n = i - kk * COUT;
    const int k = WCH ? kk * wstride : kk;
    w[i] = wp[((size_t)(k >> 2) * g.Npad + n) * 4 + (k & 3)];
  }
  __syncthreads();
  const int q = t % LPV, grp = t / LPV;
  const int m0 = ((int)blockIdx.x * GPB + grp) * 4;            // first voxel of this thread's group (x0 % 4 == 0)
  const bool gvalid = m0 < M;
  const int x0 = m0 & (S - 1), y = (m0 >> lg) & (S - 1), z = (m0 >> (2 * lg)) & (S - 1);
  float acc[4][4];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[v][j] = 0.f;
  const float* wq = w + 4 * q;
#pragma unroll ICS_THINC_UNROLL
  for (int gzy = 0; gzy < 9; ++gzy) {
    const int dz = gzy / 3 - 1, dy = gzy % 3 - 1;
    const bool rowok = gvalid && (unsigned)(z + dz) < (unsigned)S && (unsigned)(y + dy) < (unsigned)S;
    const long rowbase = (long)(m0 + (dz * S + dy) * S - 1) * CIN;          // element offset of position x0-1
#pragma unroll
    for (int c4 = 0; c4 < C4; ++c4) {
      float xv[6][4];
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const bool ok = rowok && (unsigned)(x0 + p - 1) < (unsigned)S;     // zero "same" padding
        if (CIN >= 4) {
          v4f tv = v4f{0.f, 0.f, 0.f, 0.f};
          if (ok) tv = *reinterpret_cast<const v4f*>(x + rowbase + (long)p * CIN + c4 * 4);
          xv[p][0] = tv.x; xv[p][1] = tv.y; xv[p][2] = tv.z; xv[p][3] = tv.w;
        } else {
          xv[p][0] = ok ? x[rowbase + p] : 0.f;
          xv[p][1] = xv[p][2] = xv[p][3] = 0.f;
        }
      }
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float* wk = wq + (size_t)((gzy * 3 + dx) * wrow + c4 * 4) * COUT;
#pragma unroll
        for (int j = 0; j < (CIN >= 4 ? 4 : 1); ++j) {
          const v4f wv = *reinterpret_cast<const v4f*>(wk + j * COUT);
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const float xs = xv[v + dx][j];
            acc[v][0] = fmaf(xs, wv.x, acc[v][0]); acc[v][1] = fmaf(xs, wv.y, acc[v][1]);
            acc[v][2] = fmaf(xs, wv.z, acc[v][2]); acc[v][3] = fmaf(xs, wv.w, acc[v][3]);
          }
        }
      }
    }
  }
  // ---- epilogue: bias, activation, store (each lane one float4 per voxel: LPV lanes = one full output row)
  const float pre_slope = act_slope_of(pre_act);
  v4f bv = v4f{0.f, 0.f, 0.f, 0.f};       // (scalar loads: a tensor inside the flat parameter buffer need not be 16-B aligned)
  if (bias != nullptr) { bv.x = bias[4 * q]; bv.y = bias[4 * q + 1]; bv.z = bias[4 * q + 2]; bv.w = bias[4 * q + 3]; }
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  // pos_bias [B][27][COUT]: position-dependent bias -- the contribution of spatially constant input channels (the
  // K.tile'd condition of the VAE encoder, lattice_vae.py:167-169), which differs only between the 27 border classes
  // (first / interior / last plane per axis) because of the zero "same" padding; it already includes the layer bias
  const int bsmp = m0 >> (3 * lg);
  const int czy = ((z == 0 ? 0 : (z == S - 1 ? 2 : 1)) * 3 + (y == 0 ? 0 : (y == S - 1 ? 2 : 1))) * 3;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    if (pos_bias != nullptr && gvalid) {
      const int xx = x0 + v;
      const int cls = czy + (xx == 0 ? 0 : (xx == S - 1 ? 2 : 1));
      const float* pb = pos_bias + ((size_t)bsmp * 27 + cls) * COUT + 4 * q;
      bv.x = pb[0]; bv.y = pb[1]; bv.z = pb[2]; bv.w = pb[3];
    }
    acc[v][0] = act_apply(acc[v][0] + bv.x, pre_slope); acc[v][1] = act_apply(acc[v][1] + bv.y, pre_slope);
    acc[v][2] = act_apply(acc[v][2] + bv.z, pre_slope); acc[v][3] = act_apply(acc[v][3] + bv.w, pre_slope);
    if (gvalid) {
      *reinterpret_cast<v4f*>(out + (size_t)(m0 + v) * ldo + 4 * q) = v4f{acc[v][0], acc[v][1], acc[v][2], acc[v][3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) csum[j] += acc[v][j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[v][j] = 0.f;
    }
  }
  if (stat_partial == nullptr) return;
  // block (count, mean, M2) per column, two passes inside the block (as the MFMA kernels): lanes with equal q hold
  // the same 4 columns; reduce over the wave's voxel groups (xor strides LPV .. 32), then over the 4 waves via LDS
  __syncthreads();                         // the weights in LDS are no longer needed
  float* red = smem;                       // [4 waves][COUT]
  float* bmean = smem + 4 * COUT;          // [COUT]
  const int lane = t & 63, wave = t >> 6;
  const int nvalid = min(VPB, M - (int)blockIdx.x * VPB);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float a = csum[j];
    for (int o = LPV; o < 64; o <<= 1) a += __shfl_xor(a, o);
    if (lane < LPV) red[wave * COUT + 4 * lane + j] = a;
  }
  __syncthreads();
  if (t < COUT) bmean[t] = (red[t] + red[COUT + t] + red[2 * COUT + t] + red[3 * COUT + t]) / (float)nvalid;
  __syncthreads();
  float mu[4], qs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) mu[j] = bmean[4 * q + j];
  if (gvalid) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = acc[v][j] - mu[j]; qs[j] = fmaf(d, d, qs[j]); }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float a = qs[j];
    for (int o = LPV; o < 64; o <<= 1) a += __shfl_xor(a, o);
    if (lane < LPV) red[wave * COUT + 4 * lane + j] = a;
  }
  __syncthreads();
  if (t < COUT) {
    const size_t nstat = gridDim.x;
    float* sp = stat_partial + (size_t)t * nstat + blockIdx.x;           // layout [3][Npad][nstat]
    sp[0] = (float)nvalid;
    sp[(size_t)g.Npad * nstat] = bmean[t];
    sp[(size_t)2 * g.Npad * nstat] = red[t] + red[COUT + t] + red[2 * COUT + t] + red[3 * COUT + t];
  }
}

static int launch_reduce_splits(hipStream_t st, const float* ws, int nsplit, size_t n_elems, int N, float* dw, int ldw,
                                int sub_rows, int row_pitch, int row_off);

// Backward-weight of the same layers for a single-channel input:  dW[tap][co] = sum_v x[v + off(tap)] * dy[v][co]
// (27 x COUT numbers reduced over all voxels).  Same thread shape as the forward stencil -- a thread owns 4 x-voxels
// x 4 output channels and keeps its 27 x 4 partial sums in registers over a strip of voxel groups; per group it needs
// 4 float4 of dy and the 9 x 6 input positions around it (432 FMAs).  Partials are combined across the lanes holding
// the same channel quad (xor shuffles), the waves (LDS) and the blocks (ws[split][27][COUT] + reduce_splits).
template <int COUT>
__global__ __launch_bounds__(256) void conv_thin_c_wgrad_kernel(ConvGeom g, const float* __restrict__ x,
                                                                 const float* __restrict__ dy, int ldy,
                                                                 float* __restrict__ ws, int groups_per_block) {
  constexpr int LPV = COUT / 4;
  constexpr int GPW = 256 / LPV;            // voxel groups handled per pass of the block
  __shared__ float red[4][LPV][27 * 4];
  const int t = threadIdx.x;
  const int S = g.S, lg = g.lgS;
  const int M = g.B << (3 * lg);
  const int q = t % LPV, slot = t / LPV;
  float acc[27][4];
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;
  const int g_begin = (int)blockIdx.x * groups_per_block;
  const int g_end = min(g_begin + groups_per_block, M >> 2);
  for (int gi = g_begin + slot; gi < g_end; gi += GPW) {
    const int m0 = gi << 2;
    const int x0 = m0 & (S - 1), y = (m0 >> lg) & (S - 1), z = (m0 >> (2 * lg)) & (S - 1);
    v4f d[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) d[v] = *reinterpret_cast<const v4f*>(dy + (size_t)(m0 + v) * ldy + 4 * q);
#pragma unroll
    for (int gzy = 0; gzy < 9; ++gzy) {
      const int dz = gzy / 3 - 1, dyo = gzy % 3 - 1;
      const bool rowok = (unsigned)(z + dz) < (unsigned)S && (unsigned)(y + dyo) < (unsigned)S;
      const long rowbase = (long)m0 + (long)(dz * S + dyo) * S - 1;
      float xv[6];
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const bool ok = rowok && (unsigned)(x0 + p - 1) < (unsigned)S;
        xv[p] = ok ? x[rowbase + p] : 0.f;
      }
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float xs = xv[v + dx];
          acc[gzy * 3 + dx][0] = fmaf(xs, d[v].x, acc[gzy * 3 + dx][0]);
          acc[gzy * 3 + dx][1] = fmaf(xs, d[v].y, acc[gzy * 3 + dx][1]);
          acc[gzy * 3 + dx][2] = fmaf(xs, d[v].z, acc[gzy * 3 + dx][2]);
          acc[gzy * 3 + dx][3] = fmaf(xs, d[v].w, acc[gzy * 3 + dx][3]);
        }
    }
  }
  // lanes with equal q (stride LPV inside the wave) hold partials of the same 4 columns: fixed xor tree
  const int lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = acc[k][j];
      for (int o = LPV; o < 64; o <<= 1) a += __shfl_xor(a, o);
      if (lane < LPV) red[wave][lane][k * 4 + j] = a;
    }
  __syncthreads();
  float* wsp = ws + (size_t)blockIdx.x * 27 * COUT;
  for (int i = t; i < 27 * COUT; i += 256) {
    const int k = i / COUT, co = i - k * COUT;
    const int qq = co >> 2, j = co & 3;
    wsp[i] = red[0][qq][k * 4 + j] + red[1][qq][k * 4 + j] + red[2][qq][k * 4 + j] + red[3][qq][k * 4 + j];
  }
}

static int thin_c_wgrad_blocks(const ConvGeom& g) {
  const long groups = ((long)g.B << (3 * g.lgS)) >> 2;
  return (int)std::max(1L, std::min(2048L, groups / 256));     // >= 256 voxel groups (1024 voxels) per block
}
size_t conv_thin_c_wgrad_workspace_floats(const ConvGeom& g) { return (size_t)thin_c_wgrad_blocks(g) * 27 * g.Cout; }
// dw[(tap * row_pitch) * ldw + co] for the single input channel (row_pitch = input channels per tap of dw's layout)
int launch_conv_wgrad_thin_c(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* dy, int ldy, float* dw,
                             int ldw, int row_pitch, float* ws, size_t ws_floats, int phase) {
  ICS_CHECK(conv_thin_c_ok(g, s0, 1, 1) && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0,
            "thin-C wgrad: unsupported shape");
  const int nb = thin_c_wgrad_blocks(g);
  ICS_CHECK((size_t)nb * 27 * g.Cout <= ws_floats, "wgrad workspace too small");
  const long groups = ((long)g.B << (3 * g.lgS)) >> 2;
  const int gpb = (int)((groups + nb - 1) / nb);
  if (phase != 2) {
    if (g.Cout == 32) {
      g_last_kernel_id = "conv_thin_c_wgrad_kernel<32>";
      ICS_LAUNCH(conv_thin_c_wgrad_kernel<32>, dim3(nb), dim3(256), 0, st, g, s0.p, dy, ldy, ws, gpb);
    } else {
      g_last_kernel_id = "conv_thin_c_wgrad_kernel<16>";
      ICS_LAUNCH(conv_thin_c_wgrad_kernel<16>, dim3(nb), dim3(256), 0, st, g, s0.p, dy, ldy, ws, gpb);
    }
    ICS_HIP(hipGetLastError());
  }
  if (phase != 1)   // rows k = tap (sub_rows = 1) land on row tap * row_pitch of dw
    ICS_TRY(launch_reduce_splits(st, ws, nb, (size_t)27 * g.Cout, g.Cout, dw, ldw, 1, row_pitch, 0));
  return 0;
}

// plain (no affine, same resolution) single-channel source; Cout in {16, 32}.  The kernel is generic in CIN, but on
// MI355X only CIN = 1 beats the MFMA path: measured c1 (1 -> 32, 32^3, B = 32) 0.133 -> 0.062 ms and the backward-data
// of decoder_output (1 -> 16) 0.167 -> 0.026 ms, while e0 (16 -> 16) 0.356 -> 0.420 ms and e1 (16 -> 32 at 16^3)
// 0.055 -> 0.119 ms got slower (216 float4 loads per thread: L1-bound) -- so wider inputs stay on the MFMA kernels.
static bool thin_c_shape_ok(const ConvGeom& g, int cin_log, int cout) {
  return g.taps == 27 && g.S >= 4 && (cout == 16 || cout == 32) && cin_log == 1 && !(g.flags & CF_NO_THIN_C);
}
bool conv_thin_c_ok(const ConvGeom& g, const ConvSrc& s0, int nsrc, int cin_log) {
  return nsrc == 1 && s0.scale == nullptr && !s0.up && !s0.bcast && s0.C == cin_log && thin_c_shape_ok(g, cin_log, g.Cout);
}
int launch_conv_fwd_thin_c(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, int cin_log, int wstride,
                           const float* wp, const float* bias, float* out, int ldo, int pre_act, float* stat_partial,
                           int* rows_per_block, const float* pos_bias) {
  ICS_CHECK(conv_thin_c_ok(g, s0, 1, cin_log) && wstride >= cin_log && ldo % 4 == 0, "thin-C conv: unsupported shape");
  const int M = g.B << (3 * g.lgS);
  const int lpv = g.Cout / 4, vpb = 4 * (256 / lpv);
  const dim3 grid((M + vpb - 1) / vpb);
  const size_t lds = (size_t)std::max(27 * wstride * g.Cout, 5 * g.Cout) * sizeof(float);
  if (rows_per_block) *rows_per_block = vpb;
#define ICS_TC(CINV, COUTV)                                                                                          \
  do {                                                                                                               \
    auto kern = conv_thin_c_fwd_kernel<CINV, COUTV>;                                                                 \
    static DevOnce attr;                                                                                             \
    int dev;                                                                                                         \
    if (attr.need(&dev)) {                                                                                           \
      ICS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                  27 * 8 * COUTV * 4));                                                              \
      attr.mark(dev);                                                                                                \
    }                                                                                                                \
    g_last_kernel_id = "conv_thin_c_fwd_kernel<" #CINV ", " #COUTV ">";                                              \
    ICS_LAUNCH(kern, grid, dim3(256), lds, st, g, s0.p, wstride, wp, bias, out, ldo, pre_act, stat_partial, \
                       pos_bias);                                                                                     \
  } while (0)
  if (g.Cout == 32) ICS_TC(1, 32); else ICS_TC(1, 16);
#undef ICS_TC
  ICS_HIP(hipGetLastError());
  return 0;
}

// =====================================================================================
// Thin-N direct kernels (Cout <= 4: the VAE's decoder_output conv, backward-data of the first conv).
// A 32-wide MFMA tile would be >= 87 % padding here; these are HBM/L1-bound stencils on the vector ALU:
// one thread per output voxel (forward) / one thread per weight row (backward-weight).
// =====================================================================================
static bool thin_n_ok(const ConvGeom& g, const ConvSrc& s0, int nsrc) {
  return !(g.flags & CF_NO_THIN_N) && nsrc == 1 && g.taps == 27 && g.Cout <= 4 && g.Cin % 4 == 0 && g.Cin >= 4 && g.Cin <= 64 &&
         s0.C == g.Cin && !s0.up && !s0.bcast && g.S >= 2;
}

template <int NOUT, bool AFF, int lgLPV>
__global__ __launch_bounds__(256) void conv_thin_n_fwd_kernel(ConvGeom g, ConvSrc s0, const float* __restrict__ wp,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ out, int ldo, int pre_act,
                                                               float* __restrict__ stat_partial, int accumulate) {
  // LPV = 2^lgLPV lanes share one voxel (1 for Cin <= 16, 4 above), each owning the float4 channel columns
  // cq, cq+LPV, ...: wider rows then still load whole cache lines per wave instruction (measured on c1's
  // backward-data, Cin = 32: LPV 1 / 2 / 4 / 8 -> 0.41 / 0.45 / 0.23 / 0.26 ms).
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* w = smem;                                   // [27*Cin][NOUT]
  const int t = threadIdx.x;
  const int S = g.S, lg = g.lgS, C = g.Cin;
  const int M = g.B << (3 * lg);
  const int K = 27 * C;
  for (int i = t; i < K * NOUT; i += 256) {
    const int k = i / NOUT, n = i - k * NOUT;
    w[i] = n < g.Cout ? wp[((size_t)(k >> 2) * g.Npad + n) * 4 + (k & 3)] : 0.f;
  }
  __syncthreads();
  constexpr int LPV = 1 << lgLPV, VPB = 256 >> lgLPV;    // lanes per voxel, voxels per block
  const int cq = t & (LPV - 1);
  const int m = blockIdx.x * VPB + (t >> lgLPV);
  const bool mvalid = m < M;
  const RowPos rp = decode_row(mvalid ? m : 0, S, lg);
  const float slope = act_slope_of(s0.act);
  float acc[NOUT];
#pragma unroll
  for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
  for (int tap = 0; tap < 27; ++tap) {
    const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
    const bool inb = mvalid && (unsigned)(rp.z + dz) < (unsigned)S && (unsigned)(rp.y + dy) < (unsigned)S &&
                     (unsigned)(rp.x + dx) < (unsigned)S;
    const unsigned base = inb ? (unsigned)(m + (dz * S + dy) * S + dx) * (unsigned)C : 0u;
    const float* wk = w + tap * C * NOUT;
    for (int c = cq * 4; c < C; c += 4 * LPV) {
      v4f v = *reinterpret_cast<const v4f*>(s0.p + base + c);
      if (AFF) {
        const v4f sc = *reinterpret_cast<const v4f*>(s0.scale + c);
        const v4f sh = *reinterpret_cast<const v4f*>(s0.shift + c);
        v = affine_act4(v, sc, sh, slope);
      }
      if (!inb) v = v4f{0.f, 0.f, 0.f, 0.f};     // zero "same" padding applies after BN / activation
#pragma unroll
      for (int n = 0; n < NOUT; ++n) {
        acc[n] = fmaf(v.x, wk[(c + 0) * NOUT + n], acc[n]);
        acc[n] = fmaf(v.y, wk[(c + 1) * NOUT + n], acc[n]);
        acc[n] = fmaf(v.z, wk[(c + 2) * NOUT + n], acc[n]);
        acc[n] = fmaf(v.w, wk[(c + 3) * NOUT + n], acc[n]);
      }
    }
  }
  const float pre_slope = act_slope_of(pre_act);
  const bool owner = mvalid && cq == 0;
#pragma unroll
  for (int n = 0; n < NOUT; ++n) {
    float a = acc[n];
    for (int o = LPV >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o);     // fixed tree: deterministic
    a += (bias != nullptr && n < g.Cout) ? bias[n] : 0.f;
    if (accumulate && owner && n < g.Cout) a += out[(size_t)m * ldo + n];
    a = act_apply(a, pre_slope);
    if (!owner) a = 0.f;
    acc[n] = a;
    if (owner && n < g.Cout) out[(size_t)m * ldo + n] = a;
  }
  if (stat_partial == nullptr) return;
  // block (count, mean, M2) per column: two passes inside the block, as the MFMA kernels do
  __syncthreads();
  float* red = smem;   // [4 waves][NOUT]   (weights are no longer needed)
  const int nvalid_rows = min(VPB, M - (int)blockIdx.x * VPB);
  const int lane = t & 63, wave = t >> 6;
  float mu[NOUT];
#pragma unroll
  for (int n = 0; n < NOUT; ++n) {
    float v = acc[n];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[wave * NOUT + n] = v;
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NOUT; ++n)
    mu[n] = (red[n] + red[NOUT + n] + red[2 * NOUT + n] + red[3 * NOUT + n]) / (float)nvalid_rows;
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NOUT; ++n) {
    const float d = acc[n] - mu[n];
    float q = owner ? d * d : 0.f;
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) red[wave * NOUT + n] = q;
  }
  __syncthreads();
  if (t < NOUT && t < g.Npad) {
    const size_t nstat = gridDim.x;
    float* sp = stat_partial + (size_t)t * nstat + blockIdx.x;
    sp[0] = (float)nvalid_rows;
    sp[(size_t)g.Npad * nstat] = mu[t];
    sp[(size_t)2 * g.Npad * nstat] = red[t] + red[NOUT + t] + red[2 * NOUT + t] + red[3 * NOUT + t];
  }
}

// Cout == 1 in two stages (the stencil above moves 27 x Cin floats per output voxel through L1: c1's backward-data,
// Cin = 32, 0.23 ms at 0.6 TB/s):
//   A. T[tap][m] = sum_c x[m][c] w[tap][c] for all 27 taps at once -- a [M x Cin] x [Cin x 27] product, so it runs on
//      v_mfma_f32_16x16x4_f32 transposed (D[tap][voxel]): the B operand is the voxel's own row from global memory, the
//      27 x Cin weights live in registers; one pass over x, 27 planes out;
//   B. out[m] = act(sum_tap T[tap][m + off(tap)] + b): 27 coalesced plane reads per voxel, zero padding by predicate.
// Traffic (M = 1 M voxels, Cin = 32): 134 MB in + 113 MB out, then 113 MB in: ~0.08 ms.  Same sums, other order.
typedef float tv4 __attribute__((ext_vector_type(4)));
template <int CK, bool AFF>                        // CK = Cin / 16; AFF: producer's BatchNorm affine + activation on load
__global__ __launch_bounds__(256) void thin1_taps_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ wp,
                                                         int Npad, int M, float* __restrict__ T,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         float slope) {
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  constexpr int C = CK * 16;
  tv4 wa[2][CK];                                   // A operand: w[tap = 16 tt + n][c = 16 kk + 4 g + s]
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int kk = 0; kk < CK; ++kk) {
      const int tap = 16 * tt + n;
      const int k4 = (tap * C + 16 * kk + 4 * g) >> 2;       // packed [K/4][Npad][4], column 0
      wa[tt][kk] = tap < 27 ? *reinterpret_cast<const tv4*>(wp + (size_t)k4 * Npad * 4) : tv4{0.f, 0.f, 0.f, 0.f};
    }
  tv4 asc[CK], ash[CK];
#pragma unroll
  for (int kk = 0; kk < CK; ++kk) {
    asc[kk] = AFF ? *reinterpret_cast<const tv4*>(scale + 16 * kk + 4 * g) : tv4{1.f, 1.f, 1.f, 1.f};
    ash[kk] = AFF ? *reinterpret_cast<const tv4*>(shift + 16 * kk + 4 * g) : tv4{0.f, 0.f, 0.f, 0.f};
  }
  const int ntiles = M >> 4, nw = gridDim.x * 4;
  for (int tile = blockIdx.x * 4 + (threadIdx.x >> 6); tile < ntiles; tile += nw) {
    const size_t m = (size_t)tile * 16 + n;
    tv4 xb[CK];
#pragma unroll
    for (int kk = 0; kk < CK; ++kk) {
      xb[kk] = *reinterpret_cast<const tv4*>(x + m * ldx + 16 * kk + 4 * g);
      if (AFF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) xb[kk][e] = act_apply(fmaf(xb[kk][e], asc[kk][e], ash[kk][e]), slope);
      }
    }
    tv4 acc[2] = {tv4{0.f, 0.f, 0.f, 0.f}, tv4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < CK; ++kk)
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0][kk][sidx], xb[kk][sidx], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1][kk][sidx], xb[kk][sidx], acc[1], 0, 0, 0);
      }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tap = 16 * tt + 4 * g + r;                 // D[tap local = 4 g + r][voxel n]
        if (tap < 27) T[(size_t)tap * M + m] = acc[tt][r];
      }
  }
}
__global__ __launch_bounds__(256) void thin1_gather_kernel(const float* __restrict__ T, int B, int S, int lg, int M,
                                                           const float* __restrict__ bias, float slope,
                                                           float* __restrict__ out, int ldo,
                                                           float* __restrict__ stat_partial, int Npad) {
  __shared__ float red[4];
  const int m = blockIdx.x * 256 + threadIdx.x;
  const bool mv = m < M;
  const RowPos rp = decode_row(mv ? m : 0, S, lg);
  float a = 0.f;
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) {
    const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
    const bool inb = mv && (unsigned)(rp.z + dz) < (unsigned)S && (unsigned)(rp.y + dy) < (unsigned)S &&
                     (unsigned)(rp.x + dx) < (unsigned)S;
    const float v = T[(size_t)tap * M + (inb ? m + (dz * S + dy) * S + dx : 0)];
    a += inb ? v : 0.f;
  }
  if (bias != nullptr) a += bias[0];
  a = mv ? act_apply(a, slope) : 0.f;
  if (mv) out[(size_t)m * ldo] = a;
  if (stat_partial == nullptr) return;
  // block (count, mean, M2) of column 0: two passes inside the block, as conv_thin_n_fwd_kernel does
  const int nvalid = min(256, M - (int)blockIdx.x * 256);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float v = a;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  const float mu = (red[0] + red[1] + red[2] + red[3]) / (float)nvalid;
  __syncthreads();
  const float d = a - mu;
  float q = mv ? d * d : 0.f;
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  if (lane == 0) red[wave] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    const size_t nstat = gridDim.x;
    float* sp = stat_partial + blockIdx.x;
    sp[0] = (float)nvalid;
    sp[(size_t)Npad * nstat] = mu;
    sp[(size_t)2 * Npad * nstat] = red[0] + red[1] + red[2] + red[3];
  }
}
static bool thin1_two_stage_ok(const ConvGeom& g, const ConvSrc& s0) {
  const int M = g.B << (3 * g.lgS);
  return g.Cin % 16 == 0 && g.Cin <= 64 && M % 16 == 0 && (s0.scale != nullptr || s0.act == ACT_NONE) &&
         !(g.flags & CF_NO_THIN1_2STAGE);
}
static int launch_thin1_two_stage(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* wp, const float* bias,
                                  float* out, int ldo, int pre_act, float* ws, float* stat_partial) {
  const int M = g.B << (3 * g.lgS);
  int nblk = (M / 16 + 3) / 4;
  if (nblk > 1024) nblk = 1024;
  const bool aff = s0.scale != nullptr;
  const float in_slope = act_slope_of(s0.act);
#define ICS_T1T(CKV)                                                                                                   \
  do {                                                                                                                 \
    if (aff) ICS_LAUNCH((thin1_taps_kernel<CKV, true>), dim3(nblk), dim3(256), 0, st, s0.p, s0.C, wp, g.Npad, M, ws, \
                                s0.scale, s0.shift, in_slope);                                                         \
    else ICS_LAUNCH((thin1_taps_kernel<CKV, false>), dim3(nblk), dim3(256), 0, st, s0.p, s0.C, wp, g.Npad, M, ws,    \
                            nullptr, nullptr, 1.f);                                                                    \
  } while (0)
  switch (g.Cin / 16) { case 1: ICS_T1T(1); break; case 2: ICS_T1T(2); break; case 3: ICS_T1T(3); break; default: ICS_T1T(4); break; }
#undef ICS_T1T
  ICS_HIP(hipGetLastError());
  ICS_LAUNCH(thin1_gather_kernel, dim3((M + 255) / 256), dim3(256), 0, st, ws, g.B, g.S, g.lgS, M, bias,
                     act_slope_of(pre_act), out, ldo, stat_partial, g.Npad);
  ICS_HIP(hipGetLastError());
  g_last_kernel_id = "thin1_taps_kernel + thin1_gather_kernel";
  return 0;
}

static int thin_n_lg_lpv(const ConvGeom& g) { return g.Cin <= 16 ? 0 : 2; }
static int launch_thin_n_fwd(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* wp, const float* bias,
                             float* out, int ldo, int pre_act, float* stat_partial, int* rows_per_block,
                             int accumulate, float* ws = nullptr, size_t ws_floats = 0) {
  const int M = g.B << (3 * g.lgS);
  if (g.Cout == 1 && !accumulate && ws != nullptr && ws_floats >= (size_t)27 * M && thin1_two_stage_ok(g, s0)) {
    if (rows_per_block) *rows_per_block = 256;
    return launch_thin1_two_stage(st, g, s0, wp, bias, out, ldo, pre_act, ws, stat_partial);
  }
  const int lgv = thin_n_lg_lpv(g), vpb = 256 >> lgv;
  const dim3 grid((M + vpb - 1) / vpb);
  const size_t lds = (size_t)std::max(27 * g.Cin * 4, 64) * sizeof(float);
  const bool aff = s0.scale != nullptr;
  ConvSrc s = s0;
  if (rows_per_block) *rows_per_block = vpb;
#define ICS_TN(NOUT, AFFV, LGV)                                                                                   \
  do {                                                                                                            \
    g_last_kernel_id = "conv_thin_n_fwd_kernel<" #NOUT ", " #AFFV ", " #LGV ">";                                  \
    ICS_LAUNCH((conv_thin_n_fwd_kernel<NOUT, AFFV, LGV>), grid, dim3(256), lds, st, g, s, wp, bias, out,   \
                       ldo, pre_act, stat_partial, accumulate);                                                   \
  } while (0)
#define ICS_TN_L(NOUT, AFFV)                                          \
  do {                                                                \
    if (lgv == 0) ICS_TN(NOUT, AFFV, 0); else ICS_TN(NOUT, AFFV, 2);  \
  } while (0)
  if (g.Cout == 1) { if (aff) ICS_TN_L(1, true); else ICS_TN_L(1, false); }
  else { if (aff) ICS_TN_L(4, true); else ICS_TN_L(4, false); }
#undef ICS_TN_L
#undef ICS_TN
  ICS_HIP(hipGetLastError());
  return 0;
}

// Backward-weight for Cout <= 4:  ws[split][k][n] = sum_{m in split} A[m + off(tap_k)][c_k] * dy[m][n].
// Thread t owns weight rows k = t, t + 256, ... (lanes = consecutive channels: coalesced A reads); the voxel
// loop is block-uniform, so the row decode and the dy row come from scalar registers.
template <int NOUT, bool AFF, int KPT>
__global__ __launch_bounds__(256) void conv_thin_n_wgrad_kernel(ConvGeom g, ConvSrc s0, const float* __restrict__ dy,
                                                                 int ldy, float* __restrict__ ws, int rows_per_split) {
  const int t = threadIdx.x;
  const int S = g.S, lg = g.lgS, C = g.Cin;
  const int M = g.B << (3 * lg);
  const int K = 27 * C;
  const int m_begin = blockIdx.x * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  const float slope = act_slope_of(s0.act);
  int koff[KPT], kdz[KPT], kdy[KPT], kdx[KPT], kc[KPT];
  float ksc[KPT], ksh[KPT];
  bool kvalid[KPT];
  float acc[KPT][NOUT];
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int k = t + 256 * i;
    kvalid[i] = k < K;
    const int tap = kvalid[i] ? k / C : 0;
    kc[i] = kvalid[i] ? k - tap * C : 0;
    kdz[i] = tap / 9 - 1; kdy[i] = (tap / 3) % 3 - 1; kdx[i] = tap % 3 - 1;
    koff[i] = ((kdz[i] * S + kdy[i]) * S + kdx[i]) * C + kc[i];
    ksc[i] = AFF ? s0.scale[kc[i]] : 1.f;
    ksh[i] = AFF ? s0.shift[kc[i]] : 0.f;
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[i][n] = 0.f;
  }
#pragma unroll 8
  for (int m = m_begin; m < m_end; ++m) {
    const int x = m & (S - 1), y = (m >> lg) & (S - 1), z = (m >> (2 * lg)) & (S - 1);
    float d[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) d[n] = n < g.Cout ? dy[(size_t)m * ldy + n] : 0.f;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const bool inb = kvalid[i] && (unsigned)(z + kdz[i]) < (unsigned)S && (unsigned)(y + kdy[i]) < (unsigned)S &&
                       (unsigned)(x + kdx[i]) < (unsigned)S;
      float a = s0.p[inb ? (size_t)((long)m * C + koff[i]) : (size_t)kc[i]];
      if (AFF) a = act_apply(fmaf(a, ksc[i], ksh[i]), slope);
      a = inb ? a : 0.f;
#pragma unroll
      for (int n = 0; n < NOUT; ++n) acc[i][n] = fmaf(a, d[n], acc[i][n]);
    }
  }
  float* wsp = ws + (size_t)blockIdx.x * K * g.Cout;
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int k = t + 256 * i;
    if (k < K)
#pragma unroll
      for (int n = 0; n < NOUT; ++n)
        if (n < g.Cout) wsp[(size_t)k * g.Cout + n] = acc[i][n];
  }
}

// Cout == 1 backward-weight on the matrix cores: dW[tap][c] = sum_w x'[w][c] dy[w - off(tap)] is a [Cin x M] x [M x 27]
// product with K = all voxels.  v_mfma_f32_16x16x4_f32: A[c][k = voxel] = the (BatchNorm + activation applied) input, a
// 64-byte row piece per lane group; B[k = voxel][tap] = the output gradient GATHERED at the tap's offset (dy is one float
// per voxel: 4 MB, cache resident; out-of-grid positions are zero).  A wave keeps its [Cin x 27] accumulators over all its
// 16-voxel tiles and writes one split.  (The stencil above, a thread per weight row looping over voxels: 0.23 ms for the
// decoder_output layer at 0.3 TB/s.)
template <int CK, bool AFF>
__global__ __launch_bounds__(256) void thin1_wgrad_kernel(ConvGeom g, ConvSrc s0, const float* __restrict__ dy, int ldy,
                                                          float* __restrict__ ws) {
  const int lane = threadIdx.x & 63, n = lane & 15, gq = lane >> 4;
  const int S = g.S, lg = g.lgS, M = g.B << (3 * lg);
  constexpr int C = CK * 16;
  const float slope = act_slope_of(s0.act);
  float sc[CK], sh[CK];
#pragma unroll
  for (int ct = 0; ct < CK; ++ct) { sc[ct] = AFF ? s0.scale[16 * ct + n] : 1.f; sh[ct] = AFF ? s0.shift[16 * ct + n] : 0.f; }
  int tdz[2], tdy[2], tdx[2], toff[2];
  bool tv[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int tap = 16 * tt + n;
    tv[tt] = tap < 27;
    const int tp = tv[tt] ? tap : 13;
    tdz[tt] = tp / 9 - 1; tdy[tt] = (tp / 3) % 3 - 1; tdx[tt] = tp % 3 - 1;
    toff[tt] = (tdz[tt] * S + tdy[tt]) * S + tdx[tt];
  }
  tv4 acc[CK][2];
#pragma unroll
  for (int ct = 0; ct < CK; ++ct) { acc[ct][0] = tv4{0.f, 0.f, 0.f, 0.f}; acc[ct][1] = tv4{0.f, 0.f, 0.f, 0.f}; }
  const int ntiles = M >> 4, wave_g = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  for (int tile = wave_g; tile < ntiles; tile += nw) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int w = tile * 16 + 4 * j + gq;                  // this lane group's voxel of the k-step
      const int x = w & (S - 1), y = (w >> lg) & (S - 1), z = (w >> (2 * lg)) & (S - 1);
      float bv[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const bool inb = tv[tt] && (unsigned)(z - tdz[tt]) < (unsigned)S && (unsigned)(y - tdy[tt]) < (unsigned)S &&
                         (unsigned)(x - tdx[tt]) < (unsigned)S;
        const float v = dy[(size_t)(inb ? w - toff[tt] : w) * ldy];
        bv[tt] = inb ? v : 0.f;
      }
#pragma unroll
      for (int ct = 0; ct < CK; ++ct) {
        float a = s0.p[(size_t)w * s0.C + 16 * ct + n];
        if (AFF) a = act_apply(fmaf(a, sc[ct], sh[ct]), slope);
        acc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[0], acc[ct][0], 0, 0, 0);
        acc[ct][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[1], acc[ct][1], 0, 0, 0);
      }
    }
  }
  // D[c local = 4 gq + r][tap local = n]  ->  ws[split][k = tap * C + c]
  float* wsp = ws + (size_t)wave_g * 27 * C;
#pragma unroll
  for (int ct = 0; ct < CK; ++ct)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tap = 16 * tt + n;
        if (tap < 27) wsp[tap * C + 16 * ct + 4 * gq + r] = acc[ct][tt][r];
      }
}
static bool thin1_wgrad_ok(const ConvGeom& g, const ConvSrc& s0) {
  const int M = g.B << (3 * g.lgS);
  return g.Cout == 1 && g.Cin % 16 == 0 && g.Cin <= 64 && M % 16 == 0 && !(g.flags & CF_NO_THIN1_2STAGE);
}
static int thin1_wgrad_blocks(const ConvGeom& g) {
  const int M = g.B << (3 * g.lgS);
  return std::max(1, std::min(512, (M / 16 + 3) / 4));     // x 4 waves = splits
}

static int thin_n_wgrad_splits(const ConvGeom& g) {
  if (g.Cout == 1 && g.Cin % 16 == 0 && g.Cin <= 64 && ((g.B << (3 * g.lgS)) % 16) == 0)
    return std::max(4 * thin1_wgrad_blocks(g), (int)std::max(1L, std::min(4096L, ((long)g.B << (3 * g.lgS)) / 128)));

  const long M = (long)g.B << (3 * g.lgS);
  // >= 4 resident blocks per CU, but at least 256 voxels per block
  return (int)std::max(1L, std::min(4096L, M / 128));
}


// Split-K for launches that cannot fill the chip (64x64 tiles on the S <= 8 layers): ksplit blocks share a
// tile, each sums a slice of K into ws[split][M][Npad]; splitk_finish_kernel adds the slices in fixed order
// and runs the epilogue (bias, activation, store, BatchNorm partial statistics of 64-row blocks).
static int fwd_splitk_plan(const ConvGeom& g, const ConvSrc& s0, const ConvSrc& s1, int nsrc) {
  if ((g.flags & CF_NO_FWD_SPLITK) || !fwd_is_vec(g, s0, s1) || conv_is_thin(g, s0, nsrc) || thin_n_ok(g, s0, nsrc) ||
      conv_thin_c_ok(g, s0, nsrc, g.Cin))
    return 1;
  int bm, bn;
  pick_fwd_tile(g, &bm, &bn);
  // 64x64 tiles (S <= 8 layers) and the narrow 128x32 tile when it yields only a handful of blocks (the VAE's e4:
  // 2^3 x 128 -> 4 and d0's backward-data: 4^3 x 128 -> 4 are 2..16 blocks walking K = 3456 alone: 0.13 ms of latency)
  const long M = (long)g.B << (3 * g.lgS);
  const bool narrow = bm == 128 && bn == 32 && (M + 127) / 128 * (g.Npad / 32) <= 64;
  if (bm != 64 && !narrow) return 1;
  const long blocks = narrow ? (M + 127) / 128 * (g.Npad / 32) : ((M + 63) / 64) * (g.Npad / 64);
  const bool reuse = g.taps == 27 && g.S >= 4 && g.S <= bm;
  const long units = reuse ? 9L * (g.Cin / 32) : g.Kpad / 32;     // groups (3 chunks) or chunks
  const long min_units = reuse ? 3 : 8;
  long ks = std::min(768 / std::max(blocks, 1L), units / min_units);
  return (int)std::max(1L, std::min(ks, 16L));
}
size_t conv_fwd_workspace_floats(const ConvGeom& g, const ConvSrc* src, int nsrc) {
  ConvSrc s0 = src[0], s1 = src[nsrc > 1 ? 1 : 0];
  if (nsrc == 1) s1.C = 0;
  const int ks = fwd_splitk_plan(g, s0, s1, nsrc);
  size_t need = ks > 1 ? (size_t)ks * ((size_t)g.B << (3 * g.lgS)) * g.Npad : 0;
  if (thin_n_ok(g, s0, nsrc) && g.Cin % 16 == 0) need = std::max(need, (size_t)27 * ((size_t)g.B << (3 * g.lgS)));   // thin1 two-stage planes
  return need;
}

__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ ws, int nsplit, int M, int N,
                                                             int Npad, const float* __restrict__ bias,
                                                             float slope, float* __restrict__ out, int ldo,
                                                             int accumulate, float* __restrict__ stat_partial) {
  __shared__ float red[4][64];
  __shared__ float bmean[64];
  const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
  const int col = blockIdx.y * 64 + tx;
  const int row0 = blockIdx.x * 64 + ty * 16;
  const bool cvalid = col < N;
  const float bv = (bias != nullptr && cvalid) ? bias[col] : 0.f;
  float v[16];
  float sum = 0.f;
  // split-outer, row-inner: the 16 rows of a split are 16 independent loads in flight (row-outer, the first form, walked
  // nsplit dependent loads per row: 55 us for the VAE's e4 with 4 blocks in the grid).  Every element still adds its
  // slices in ascending split order: the same bits as before.
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 0.f;
  const int mlast = M - 1;
  for (int p = 0; p < nsplit; ++p) {
    const float* wsp = ws + (size_t)p * M * Npad + (cvalid ? col : 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] += wsp[(size_t)min(row0 + i, mlast) * Npad];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = row0 + i;
    float a = 0.f;
    if (m < M && cvalid) {
      a = v[i] + bv;
      if (accumulate) a += out[(size_t)m * ldo + col];
      a = act_apply(a, slope);
      out[(size_t)m * ldo + col] = a;
    }
    v[i] = a;
    sum += a;
  }
  if (stat_partial == nullptr) return;
  const int nvalid_rows = min(64, M - (int)blockIdx.x * 64);
  red[ty][tx] = sum;
  __syncthreads();
  if (ty == 0) bmean[tx] = (red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]) / (float)nvalid_rows;
  __syncthreads();
  const float mu = bmean[tx];
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float d = v[i] - mu;
    q += (row0 + i < M) ? d * d : 0.f;
  }
  __syncthreads();
  red[ty][tx] = q;
  __syncthreads();
  if (ty == 0 && col < Npad) {
    const size_t nstat = gridDim.x;
    float* sp = stat_partial + (size_t)col * nstat + blockIdx.x;
    sp[0] = (float)nvalid_rows;
    sp[(size_t)Npad * nstat] = mu;
    sp[(size_t)2 * Npad * nstat] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
  }
}

static int launch_conv_fwd_inner(hipStream_t st, const ConvGeom& g, const ConvSrc* src, int nsrc,
                                 const float* wp, const float* bias, float* out, int ldo, int pre_act,
                                 float* stat_partial, int* rows_per_block, int accumulate, int ksplit,
                                 const BwdStat* bwd = nullptr, float* ws = nullptr, size_t ws_floats = 0);

int launch_conv_fwd(hipStream_t st, const ConvGeom& g, const ConvSrc* src, int nsrc,
                    const float* wp, const float* bias, float* out, int ldo, int pre_act,
                    float* stat_partial, int* rows_per_block, int accumulate, float* ws, size_t ws_floats,
                    const BwdStat* bwd, int* bwd_blocks) {
  ConvSrc s0 = src[0], s1 = src[nsrc > 1 ? 1 : 0];
  if (nsrc == 1) s1.C = 0;
  int ks = ws ? fwd_splitk_plan(g, s0, s1, nsrc) : 1;
  // workspaces are sized at the handle's maximum batch; a smaller batch may plan more splits than fit: run unsplit
  if (ks > 1 && (size_t)ks * ((size_t)g.B << (3 * g.lgS)) * g.Npad > ws_floats) ks = 1;
  if (bwd_blocks) *bwd_blocks = 0;
  if (ks <= 1) {
    const bool fold = bwd != nullptr && bwd->partial != nullptr && !thin_n_ok(g, src[0], nsrc) &&
                      fwd_is_vec(g, s0, s1) && !conv_is_thin(g, s0, nsrc) && s0.scale == nullptr && !s0.up &&
                      (nsrc < 2 || (s1.scale == nullptr && !s1.up)) && bias == nullptr && stat_partial == nullptr &&
                      !accumulate;
    int rpb = 0;
    ICS_TRY(launch_conv_fwd_inner(st, g, src, nsrc, wp, bias, out, ldo, pre_act, stat_partial, &rpb, accumulate, 1,
                                  fold ? bwd : nullptr, ws, ws_floats));
    if (rows_per_block) *rows_per_block = rpb;
    if (fold && bwd_blocks) *bwd_blocks = (int)((((size_t)g.B << (3 * g.lgS)) + rpb - 1) / rpb);
    return 0;
  }
  const int M = g.B << (3 * g.lgS);
  ICS_CHECK((size_t)ks * M * g.Npad <= ws_floats, "forward split-K workspace too small");
  ICS_TRY(launch_conv_fwd_inner(st, g, src, nsrc, wp, nullptr, ws, g.Npad, ACT_NONE, nullptr, nullptr, 0, ks));
  ICS_LAUNCH(splitk_finish_kernel, dim3((M + 63) / 64, (g.Npad + 63) / 64), dim3(256), 0, st, ws, ks, M, g.Cout,
                     g.Npad, bias, act_slope_of(pre_act), out, ldo, accumulate, stat_partial);
  ICS_HIP(hipGetLastError());
  if (rows_per_block) *rows_per_block = 64;
  return 0;
}

static int launch_conv_fwd_inner(hipStream_t st, const ConvGeom& g, const ConvSrc* src, int nsrc,
                                 const float* wp, const float* bias, float* out, int ldo, int pre_act,
                                 float* stat_partial, int* rows_per_block, int accumulate, int ksplit,
                                 const BwdStat* bwd, float* ws, size_t ws_floats) {
  if (ksplit == 1 && thin_n_ok(g, src[0], nsrc))
    return launch_thin_n_fwd(st, g, src[0], wp, bias, out, ldo, pre_act, stat_partial, rows_per_block, accumulate, ws,
                             ws_floats);
  if (ksplit == 1 && !accumulate && bwd == nullptr && conv_thin_c_ok(g, src[0], nsrc, g.Cin) && ldo % 4 == 0)
    return launch_conv_fwd_thin_c(st, g, src[0], g.Cin, g.Cin, wp, bias, out, ldo, pre_act, stat_partial,
                                  rows_per_block);
  ConvSrc s0 = src[0], s1 = src[nsrc > 1 ? 1 : 0];
  if (nsrc == 1) s1.C = 0;
  // loader variants: 0 = plain sources (backward-data, pooled inputs), 1 = BN affine/activation,
  // 2 = affine + a nearest-upsampled source (the U-Net's concat layers, the VAE decoder)
  const bool aff = s0.scale != nullptr || (nsrc > 1 && s1.scale != nullptr);
  const bool up = s0.up || (nsrc > 1 && s1.up);
  const int variant = up ? 2 : (aff ? 1 : 0);
  // every affine source applies its activation before the BatchNorm: affine-only loader
  const bool noact = aff && (s0.scale == nullptr || s0.act == ACT_NONE) &&
                     (nsrc < 2 || s1.scale == nullptr || s1.act == ACT_NONE);
  ICS_TRY(fix_src(s0));
  ICS_TRY(fix_src(s1));
  bool vec = fwd_is_vec(g, s0, s1);
  const bool thin = conv_is_thin(g, s0, nsrc);
  // the tile loaders address a source with 32-bit byte offsets
  ICS_CHECK(((size_t)g.B << (3 * g.lgS)) * (size_t)std::max(s0.C, s1.C) * 4 <= 0xffffffffull,
            "conv source larger than 4 GiB");
  const bool no_reuse = (g.flags & CF_NO_REUSE) != 0;   // A/B switch (tests, benchmarking)
  // BatchNorm-backward sums folded into the epilogue: only the plain-source vector instantiations carry the variant
  const bool fold = bwd != nullptr && bwd->partial != nullptr && ksplit == 1 && vec && !thin && variant == 0 &&
                    bias == nullptr && stat_partial == nullptr && !accumulate;
  int bm, bn;
  pick_fwd_tile(g, &bm, &bn);
#define ICS_FWD_ARGS st, g, s0, s1, wp, bias, out, ldo, pre_act, stat_partial, rows_per_block, accumulate, ksplit, bwd
#define ICS_FWD(WM, WN, TM, TN)                                                                 \
  do {                                                                                          \
    if (thin) return launch_fwd_cfg<WM, WN, TM, TN, true, 0, true, true, true>(ICS_FWD_ARGS);   \
    if (!vec) return launch_fwd_cfg<WM, WN, TM, TN, false, 0, true, true>(ICS_FWD_ARGS);        \
    if (g.taps == 27 && g.S >= 4 && g.S <= (WM) * (TM) * 32 && !no_reuse) {                      \
      if (variant == 0 && fold)                                                                 \
        return launch_fwd_cfg<WM, WN, TM, TN, true, 0, false, false, false, true, false, false, true>(ICS_FWD_ARGS); \
      if (variant == 0) return launch_fwd_cfg<WM, WN, TM, TN, true, 0, false, false, false, true>(ICS_FWD_ARGS); \
      if (variant == 1 && noact)                                                                \
        return launch_fwd_cfg<WM, WN, TM, TN, true, 0, true, false, false, true, false, true>(ICS_FWD_ARGS);     \
      if (variant == 1) return launch_fwd_cfg<WM, WN, TM, TN, true, 0, true, false, false, true>(ICS_FWD_ARGS);  \
      return launch_fwd_cfg<WM, WN, TM, TN, true, 0, true, true, false, true>(ICS_FWD_ARGS);    \
    }                                                                                           \
    if (variant == 0 && fold)                                                                   \
      return launch_fwd_cfg<WM, WN, TM, TN, true, 0, false, false, false, false, false, false, true>(ICS_FWD_ARGS); \
    if (variant == 0) return launch_fwd_cfg<WM, WN, TM, TN, true, 0, false, false>(ICS_FWD_ARGS); \
    if (variant == 1) return launch_fwd_cfg<WM, WN, TM, TN, true, 0, true, false>(ICS_FWD_ARGS);  \
    return launch_fwd_cfg<WM, WN, TM, TN, true, 0, true, true>(ICS_FWD_ARGS);                   \
  } while (0)
  if (bm == 64) { ICS_FWD(2, 2, 1, 1); }
  if (bn == 128) { ICS_FWD(2, 2, 2, 2); }
  if (bn == 96) { ICS_FWD(4, 1, 1, 3); }
  if (bn == 64) { ICS_FWD(2, 2, 2, 1); }
  ICS_FWD(4, 1, 1, 1);
#undef ICS_FWD_ARGS
#undef ICS_FWD
}

// The 8 parity-class GEMMs of a 3x3x3 conv over a nearest-upsampled source (see PAR above).
// g: geometry of the LOW-RES grid with taps = 8, Cin = channels of the source, Kpad = 8*Cin;
// wp: [8][Kpad/4][Npad][4] pre-summed weights (launch_pack_par); out: fine-grid rows [8*M][ldo], raw sums.
// block count heuristics of pick_fwd_tile see one class; the launch has 8x the blocks: prefer the large tile
static void pick_par_tile(const ConvGeom& g, int* bm, int* bn) {
  pick_fwd_tile(g, bm, bn);
  if (*bm == 64 && g.Npad % 128 == 0 && (long)(((g.B << (3 * g.lgS)) + 127) / 128) * (g.Npad / 128) * 8 >= 384) {
    *bm = 128; *bn = 128;
  }
}
int launch_conv_fwd_par(hipStream_t st, const ConvGeom& g, const ConvSrc& src, const float* wp, float* out,
                        int ldo, const float* bias, int pre_act, float* stat_partial, int* stat_blocks) {
  ConvSrc s0 = src, s1 = src;
  s1.C = 0;
  ICS_CHECK(g.taps == 8 && g.Cin % 32 == 0 && s0.C == g.Cin && !s0.up && !s0.bcast && g.Kpad == 8 * g.Cin,
            "parity-class conv needs a direct 32k-channel source");
  ICS_TRY(fix_src(s0));
  ICS_TRY(fix_src(s1));
  int bm, bn;
  pick_par_tile(g, &bm, &bn);
  int rpb = 0;
#define ICS_PAR_ARGS st, g, s0, s1, wp, bias, out, ldo, pre_act, stat_partial, &rpb
  int rc;
  const bool no_reuse = (g.flags & CF_NO_REUSE) != 0;
  const bool reuse = !no_reuse && g.S >= 4 && g.S <= bm;   // dx-reuse: the two ex taps share a staged A tile
  const bool noact = src.scale != nullptr && src.act == ACT_NONE;   // affine-only loader (see NOACT)
#define ICS_PAR(WM, WN, TM, TN)                                                                              \
  (reuse ? (noact ? launch_fwd_cfg<WM, WN, TM, TN, true, 0, true, false, false, true, true, true>(ICS_PAR_ARGS) \
                  : launch_fwd_cfg<WM, WN, TM, TN, true, 0, true, false, false, true, true>(ICS_PAR_ARGS))   \
         : launch_fwd_cfg<WM, WN, TM, TN, true, 0, true, false, false, false, true>(ICS_PAR_ARGS))
  if (bm == 64) rc = ICS_PAR(2, 2, 1, 1);
  else if (bn == 128) rc = ICS_PAR(2, 2, 2, 2);
  else if (bn == 96) rc = ICS_PAR(4, 1, 1, 3);
  else if (bn == 64) rc = ICS_PAR(2, 2, 2, 1);
  else rc = ICS_PAR(4, 1, 1, 1);
#undef ICS_PAR
#undef ICS_PAR_ARGS
  if (stat_blocks) *stat_blocks = 8 * (((g.B << (3 * g.lgS)) + rpb - 1) / rpb);
  return rc;
}

// ---- one-launch packing.  After every Adam step ~50 packed weight images are rebuilt (forward, tap-flipped
// backward-data, skip / parity-class / up-split variants per layer); as separate launches they cost ~5 us each
// whatever their size.  The engine records the jobs once (the launch_pack_* calls below append to g_pack_rec
// instead of launching) and replays them with ONE pack_table_kernel launch per step.
struct PackJob {
  int kind;                 // 0 fwd, 1 bwd, 2 sub, 3 fwd_sub, 4 par, 5 wino, 6 up3, 7 up3n
  const float* w;
  float* dst;
  int a[9];
  unsigned long long total; // elements of dst this job covers
  unsigned blk0;            // first block of the job in the table launch
};
static thread_local std::vector<PackJob>* g_pack_rec = nullptr;

// pre-summed parity weights: dst[cls][k = e*Cu + c][n] = sum over taps d with (per axis)
//   p=0: e=0 -> {-1}, e=1 -> {0,+1};  p=1: e=0 -> {-1,0}, e=1 -> {+1}   of w[d][c_off + c][n]
__device__ __forceinline__ float pack_par_value(size_t i, const float* __restrict__ w, int Cin_total, int Cout, int c_off,
                                                int Cu, int Kpad, int Npad) {
  const size_t per = (size_t)Kpad * Npad;
  const int cls = (int)(i / per);
  const size_t j = i - (size_t)cls * per;
  const int tq = j & 3;
  const size_t rest = j >> 2;
  const int n = rest % Npad;
  const int k = (int)(rest / Npad) * 4 + tq;
  const int e = k / Cu, c = k - e * Cu;
  float v = 0.f;
  if (e < 8 && n < Cout) {
    const int p3[3] = {cls >> 2, (cls >> 1) & 1, cls & 1};
    const int e3[3] = {e >> 2, (e >> 1) & 1, e & 1};
    int lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (p3[a] == 0) { lo[a] = e3[a] ? 1 : 0; hi[a] = e3[a] ? 2 : 0; }
      else { lo[a] = e3[a] ? 2 : 0; hi[a] = e3[a] ? 2 : 1; }
    }
    for (int dz = lo[0]; dz <= hi[0]; ++dz)
      for (int dy = lo[1]; dy <= hi[1]; ++dy)
        for (int dx = lo[2]; dx <= hi[2]; ++dx)
          v += w[((size_t)((dz * 3 + dy) * 3 + dx) * Cin_total + c_off + c) * Cout + n];
  }
  return v;
}
__global__ void pack_par_kernel(const float* __restrict__ w, int Cin_total, int Cout, int c_off, int Cu,
                                float* __restrict__ dst, int Kpad, int Npad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 8 * (size_t)Kpad * Npad) return;
  dst[i] = pack_par_value(i, w, Cin_total, Cout, c_off, Cu, Kpad, Npad);
}
int launch_pack_par(hipStream_t st, const float* w, int Cin_total, int Cout, int c_off, int Cu, float* dst,
                    int Kpad, int Npad) {
  if (g_pack_rec) { g_pack_rec->push_back(PackJob{4, w, dst, {Cin_total, Cout, c_off, Cu, Kpad, Npad, 0, 0, 0}, 8ull * Kpad * Npad, 0}); return 0; }
  const size_t total = (size_t)8 * Kpad * Npad;
  ICS_LAUNCH(pack_par_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, Cin_total, Cout,
                     c_off, Cu, dst, Kpad, Npad);
  ICS_HIP(hipGetLastError());
  return 0;
}
// forward weights of a channel subset: dst[k = tap*Csub + c][n] = w[tap][c_off + c][n]
__device__ __forceinline__ float pack_fwd_sub_value(size_t i, const float* __restrict__ w, int taps, int Cin_total,
                                                    int Cout, int c_off, int Csub, int Npad) {
  const int tq = i & 3;
  const size_t rest = i >> 2;
  const int n = rest % Npad;
  const int k = (int)(rest / Npad) * 4 + tq;
  const int tap = k / Csub, c = k - tap * Csub;
  return (tap < taps && n < Cout) ? w[((size_t)tap * Cin_total + c_off + c) * Cout + n] : 0.f;
}
__global__ void pack_fwd_sub_kernel(const float* __restrict__ w, int taps, int Cin_total, int Cout, int c_off,
                                    int Csub, float* __restrict__ dst, int Kpad, int Npad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)Kpad * Npad) return;
  dst[i] = pack_fwd_sub_value(i, w, taps, Cin_total, Cout, c_off, Csub, Npad);
}
int launch_pack_fwd_sub(hipStream_t st, const float* w, int taps, int Cin_total, int Cout, int c_off, int Csub,
                        float* dst, int Kpad, int Npad) {
  if (g_pack_rec) { g_pack_rec->push_back(PackJob{3, w, dst, {taps, Cin_total, Cout, c_off, Csub, Npad, 0, 0, 0}, (unsigned long long)Kpad * Npad, 0}); return 0; }
  const size_t total = (size_t)Kpad * Npad;
  ICS_LAUNCH(pack_fwd_sub_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, taps,
                     Cin_total, Cout, c_off, Csub, dst, Kpad, Npad);
  ICS_HIP(hipGetLastError());
  return 0;
}

// benchmarking-only: the 128x128 vector-path instantiation with part of the loop removed
int launch_conv_fwd_ablate(hipStream_t st, const ConvGeom& g, const ConvSrc* src, int nsrc, const float* wp,
                           float* out, int ldo, int ablate) {
  ConvSrc s0 = src[0], s1 = src[nsrc > 1 ? 1 : 0];
  if (nsrc == 1) s1.C = 0;
  ICS_TRY(fix_src(s0));
  ICS_TRY(fix_src(s1));
  ICS_CHECK(fwd_is_vec(g, s0, s1) && g.Npad % 128 == 0, "ablation bench needs the vector 128x128 configuration");
#define ICS_ABL_ARGS st, g, s0, s1, wp, nullptr, out, ldo, 0, nullptr, nullptr
  if (ablate == 1) return launch_fwd_cfg<2, 2, 2, 2, true, 1, true, false>(ICS_ABL_ARGS);
  if (ablate == 2) return launch_fwd_cfg<2, 2, 2, 2, true, 2, true, false>(ICS_ABL_ARGS);
  if (ablate == 3) return launch_fwd_cfg<2, 2, 2, 2, true, 3, true, false>(ICS_ABL_ARGS);
  if (ablate == 4) return launch_fwd_cfg<2, 2, 2, 2, true, 4, true, false>(ICS_ABL_ARGS);
  if (ablate == 8) return launch_fwd_cfg<2, 2, 2, 2, true, 8, false, false>(ICS_ABL_ARGS);
  if (ablate == 9) return launch_fwd_cfg<2, 2, 2, 2, true, 9, false, false>(ICS_ABL_ARGS);
  if (ablate == 5) return launch_fwd_cfg<2, 2, 2, 2, true, 0, true, false>(ICS_ABL_ARGS);   // affine, no upsample
  if (ablate == 6) return launch_fwd_cfg<2, 2, 2, 2, true, 0, true, true>(ICS_ABL_ARGS);    // affine + upsample math
  return launch_fwd_cfg<2, 2, 2, 2, true, 0, false, false>(ICS_ABL_ARGS);
#undef ICS_ABL_ARGS
}


// =====================================================================================
// Backward-weight kernel: ws[split][k][n] = sum_{m in split} A[m][k] * dy[m][n]
// =====================================================================================
template <int WM, int WN, int TM, int TN, bool VEC, bool DYVEC, bool AFF = true, bool UP = true, bool THIN = false,
          int ABL = 0>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(ConvGeom g, ConvSrc s0, ConvSrc s1,
                                                          const float* __restrict__ dy, int ldy,
                                                          int n_load,
                                                          float* __restrict__ ws, int ktiles,
                                                          int ntiles, int rows_per_split) {
  constexpr int KT = WM * TM * 32, NT = WN * TN * 32;
  constexpr int A_FLOATS = 32 * KT, D_FLOATS = 32 * NT;
  constexpr int AF4 = KT / 4, ATOT = 32 * AF4, APASS = ATOT / 256;   // 256 % AF4 == 0
  constexpr int DF4 = NT / 4, DTOT = 32 * DF4, DPASS = DTOT / 256;
  static_assert(ATOT % 256 == 0 && DTOT % 256 == 0, "tile loads must divide evenly over 256 threads");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][32][KT]
  float* Ds = smem + 2 * A_FLOATS;     // [2][32][NT]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int S = g.S, lg = g.lgS;
  const int M = g.B << (3 * lg);
  const int K = g.taps * g.Cin;

  int bid = blockIdx.x;
  const int nt_i = bid % ntiles; bid /= ntiles;
  const int kt_i = bid % ktiles; bid /= ktiles;
  const int split = bid;
  const int n0 = nt_i * NT;
  const int m_begin = split * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  const int nchunks = (m_end - m_begin + 31) >> 5;

  // k tiles are runs of KT consecutive flattened k = tap*Cin + ci.  VEC (Cin % 32 == 0): a thread's
  // float4 column sits inside one 32-channel group, hence one tap and one source -- all per-thread
  // constants for the whole block, so tiles may span taps (thin Cin) and sources (concat layers).
  const int k0 = kt_i * KT;
  int tdz = 0, tdy = 0, tdx = 0, sdelta = 0, xbad = -1, ybad = -1, zbad = -1;
  bool kvalid = false;
  const float* sp = s0.p;
  int sC = s0.C, su = 0, cl0 = 0;
  float slope = 1.f;
  v4f sc = v4f{1.f, 1.f, 1.f, 1.f}, sh = v4f{0.f, 0.f, 0.f, 0.f};
  if (VEC) {
    const int kf = k0 + (t % AF4) * 4;
    const int tap = kf / g.Cin, ci = kf - tap * g.Cin;
    kvalid = kf < K;
    if (g.taps == 27 && kvalid) { tdz = tap / 9 - 1; tdy = (tap / 3) % 3 - 1; tdx = tap % 3 - 1; }
    sdelta = (tdz * S + tdy) * S + tdx;
    // a shifted coordinate leaves the grid iff it starts on the face the tap points away from
    xbad = tdx < 0 ? 0 : (tdx > 0 ? S - 1 : -1);
    ybad = tdy < 0 ? 0 : (tdy > 0 ? S - 1 : -1);
    zbad = tdz < 0 ? 0 : (tdz > 0 ? S - 1 : -1);
    const bool first = ci < s0.C;
    const ConvSrc sv = pick_src(s0, s1, first);
    sp = sv.p; sC = sv.C; su = sv.up;
    cl0 = kvalid ? (first ? ci : ci - s0.C) : 0;
    slope = act_slope_of(sv.act);
    if (AFF) {
      sc = *reinterpret_cast<const v4f*>(sv.scale + cl0);
      sh = *reinterpret_cast<const v4f*>(sv.shift + cl0);
    }
  }
  // SCALAR: this thread's fixed k
  int s_tap = 0, s_ci = 0; bool s_kvalid = false; int sdz = 0, sdy = 0, sdx = 0;
  if (!VEC) {
    const int kf = k0 + (t & 31);
    s_tap = kf / g.Cin; s_ci = kf - s_tap * g.Cin; s_kvalid = kf < K;
    if (g.taps == 27) { sdz = s_tap / 9 - 1; sdy = (s_tap / 3) % 3 - 1; sdx = s_tap % 3 - 1; }
  }

  v4f ra[VEC ? APASS : 1];
  float ras[4];
  v4f rd[DPASS];

  auto load_chunk = [&](int c) {
    const int mbase = m_begin + (c << 5);
    if (VEC) {
      const int Sh = S >> 1;
#pragma unroll
      for (int p = 0; p < APASS; ++p) {
        const int m = mbase + (t + 256 * p) / AF4;
        const int x = m & (S - 1), y = (m >> lg) & (S - 1), z = (m >> (2 * lg)) & (S - 1);
        const bool inb = kvalid && m < m_end && x != xbad && y != ybad && z != zbad;
        int idx = m + sdelta;            // same-resolution voxel index of the tap-shifted row
        if (UP) {
          const int b = m >> (3 * lg);
          const int idx_up = ((b * Sh + ((z + tdz) >> 1)) * Sh + ((y + tdy) >> 1)) * Sh + ((x + tdx) >> 1);
          idx = su ? idx_up : idx;
        }
        const unsigned off = inb ? (unsigned)idx * (unsigned)sC + cl0 : (unsigned)cl0;
        v4f v = *reinterpret_cast<const v4f*>(sp + off);
        if (AFF) v = affine_act4(v, sc, sh, slope);
        ra[p] = inb ? v : v4f{0.f, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int m = mbase + (t >> 5) + 8 * p;
        float v = 0.f;
        if (s_kvalid && m < m_end) {
          const RowPos r = decode_row(m, S, lg);
          const int zz = r.z + sdz, yy = r.y + sdy, xx = r.x + sdx;
          if ((unsigned)zz < (unsigned)S && (unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S)
            v = gather_scalar(s0, s1, s_ci, r.b, zz, yy, xx, S);
        }
        ras[p] = v;
      }
    }
#pragma unroll
    for (int p = 0; p < DPASS; ++p) {
      const int idx = t + 256 * p;
      const int m = mbase + idx / DF4;
      const int n = n0 + (idx % DF4) * 4;
      const float* q = dy + (size_t)min(m, M - 1) * ldy;
      v4f v;
      if (DYVEC) {
        v = *reinterpret_cast<const v4f*>(q + min(n, n_load - 4));
        if (!(m < m_end && n < n_load)) v = v4f{0.f, 0.f, 0.f, 0.f};
      } else {
        v.x = q[min(n + 0, n_load - 1)]; v.y = q[min(n + 1, n_load - 1)];
        v.z = q[min(n + 2, n_load - 1)]; v.w = q[min(n + 3, n_load - 1)];
        const bool rowok = m < m_end;
        v.x = (rowok && n + 0 < n_load) ? v.x : 0.f; v.y = (rowok && n + 1 < n_load) ? v.y : 0.f;
        v.z = (rowok && n + 2 < n_load) ? v.z : 0.f; v.w = (rowok && n + 3 < n_load) ? v.w : 0.f;
      }
      rd[p] = v;
    }
  };
  // taps == 1 over a same-resolution source with whole 32-row chunks and whole N tiles (the coarse-grid GEMMs of the
  // up-split backward, the heads): a plain A^T x dY.  Wave-uniform chunk pointers + per-thread byte offsets fixed
  // for the kernel, no validity arithmetic (a thread whose k >= K loads column 0; its rows are never written).
  const bool gemm = VEC && DYVEC && !UP && !THIN && ABL == 0 && g.taps == 1 && (M & 31) == 0 &&
                    (rows_per_split & 31) == 0 && n0 + NT <= n_load;
  const bool noact_g = slope == 1.f;
  unsigned aoffg[VEC ? APASS : 1], doffg[DPASS];
  if (VEC) {
#pragma unroll
    for (int p = 0; p < APASS; ++p) aoffg[p] = (unsigned)(((t + 256 * p) / AF4) * sC + cl0) * 4u;
  }
#pragma unroll
  for (int p = 0; p < DPASS; ++p) {
    const int idx = t + 256 * p;
    doffg[p] = (unsigned)((idx / DF4) * ldy + n0 + (idx % DF4) * 4) * 4u;
  }
  auto load_chunk_gemm = [&](int c, auto noact_tag) {
    constexpr bool NA = decltype(noact_tag)::value;
    const int mbase = m_begin + (c << 5);
    const char* abase = reinterpret_cast<const char*>(sp) + (size_t)mbase * (size_t)sC * 4;
    const char* dbase = reinterpret_cast<const char*>(dy) + (size_t)mbase * (size_t)ldy * 4;
    if (VEC) {
#pragma unroll
      for (int p = 0; p < APASS; ++p) {
        v4f v = *reinterpret_cast<const v4f*>(abase + aoffg[p]);
        if (AFF) v = affine_only_or_act4<NA>(v, sc, sh, slope);
        ra[p] = v;
      }
    }
#pragma unroll
    for (int p = 0; p < DPASS; ++p) rd[p] = *reinterpret_cast<const v4f*>(dbase + doffg[p]);
  };
  auto store_chunk = [&](int buf) {
    float* A = As + buf * A_FLOATS;
    float* D = Ds + buf * D_FLOATS;
    if (VEC) {
#pragma unroll
      for (int p = 0; p < APASS; ++p) {
        const int idx = t + 256 * p;
        *reinterpret_cast<v4f*>(A + (idx / AF4) * KT + (idx % AF4) * 4) = ra[p];
      }
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) A[((t >> 5) + 8 * p) * KT + (t & 31)] = ras[p];
    }
#pragma unroll
    for (int p = 0; p < DPASS; ++p) {
      const int idx = t + 256 * p;
      *reinterpret_cast<v4f*>(D + (idx / DF4) * NT + (idx % DF4) * 4) = rd[p];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nchunks > 0) {
    if (gemm && noact_g) load_chunk_gemm(0, std::true_type{});
    else if (gemm) load_chunk_gemm(0, std::false_type{});
    else load_chunk(0);
    store_chunk(0);
  }
  __syncthreads();
  // LDS keeps the natural [voxel m][channel] images (b128 stores straight from the global loads).
  // Operand fetch is COMPONENT-SPLIT: lane i reads TM (TN) consecutive channels k = TM*i + tm with one
  // ds_read_b64/b32 and component tm feeds the accumulator that owns the row set {TM*i + tm}; the
  // contraction index of one MFMA is the voxel pair (2s, 2s+1) selected by the lane half.  No
  // transposition anywhere; 32 b64 reads per 64 MFMAs for the 128x128 tile.  T = 3 falls back to the
  // block mapping k = 32*tm + i (three b32 reads) because ds_read_b96 needs 16-byte alignment.
  constexpr bool SPLIT_M = (TM != 3), SPLIT_N = (TN != 3);
  auto compute = [&](int buf) {
    const float* A = As + buf * A_FLOATS + lh * KT + wm * TM * 32 + (SPLIT_M ? TM * li : li);
    const float* D = Ds + buf * D_FLOATS + lh * NT + wn * TN * 32 + (SPLIT_N ? TN * li : li);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float a[TM], b[TN];
      if (SPLIT_M && TM == 2) {
        const float2 q = *reinterpret_cast<const float2*>(A + s * 2 * KT);
        a[0] = q.x; a[1 % TM] = q.y;
      } else if (SPLIT_M && TM == 4) {
        const float4 q = *reinterpret_cast<const float4*>(A + s * 2 * KT);
        a[0] = q.x; a[1 % TM] = q.y; a[2 % TM] = q.z; a[3 % TM] = q.w;
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = A[s * 2 * KT + i * 32];
      }
      if (SPLIT_N && TN == 2) {
        const float2 q = *reinterpret_cast<const float2*>(D + s * 2 * NT);
        b[0] = q.x; b[1 % TN] = q.y;
      } else if (SPLIT_N && TN == 4) {
        const float4 q = *reinterpret_cast<const float4*>(D + s * 2 * NT);
        b[0] = q.x; b[1 % TN] = q.y; b[2 % TN] = q.z; b[3 % TN] = q.w;
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = D[s * 2 * NT + j * 32];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };
#if ICS_GEMM_DEEP
  if (gemm && VEC && TM * TN >= ICS_GEMM_DEEP_MIN) {
    // two chunks ahead through TWO raw register sets that swap roles (the loop is unrolled by two; the affine is applied
    // when a set is stored): both operands stream from HBM with no reuse, one chunk of prefetch left the loads exposed
    // (see conv_fwd_kernel).  The first version rotated one set into the other: 24 register moves per 64 MFMAs.
    v4f raA[APASS], rdA[DPASS], raB[APASS], rdB[DPASS];
    auto loadS = [&](int c, v4f (&xa)[APASS], v4f (&xd)[DPASS]) {
      const int mbase = m_begin + (c << 5);
      const char* abase = reinterpret_cast<const char*>(sp) + (size_t)mbase * (size_t)sC * 4;
      const char* dbase = reinterpret_cast<const char*>(dy) + (size_t)mbase * (size_t)ldy * 4;
#pragma unroll
      for (int p = 0; p < APASS; ++p) xa[p] = *reinterpret_cast<const v4f*>(abase + aoffg[p]);
#pragma unroll
      for (int p = 0; p < DPASS; ++p) xd[p] = *reinterpret_cast<const v4f*>(dbase + doffg[p]);
    };
    auto storeS = [&](int buf, const v4f (&xa)[APASS], const v4f (&xd)[DPASS]) {
      float* A = As + buf * A_FLOATS;
      float* D = Ds + buf * D_FLOATS;
#pragma unroll
      for (int p = 0; p < APASS; ++p) {
        const int idx = t + 256 * p;
        v4f v = xa[p];
        if (AFF) v = noact_g ? affine_only_or_act4<true>(v, sc, sh, slope) : affine_only_or_act4<false>(v, sc, sh, slope);
        *reinterpret_cast<v4f*>(A + (idx / AF4) * KT + (idx % AF4) * 4) = v;
      }
#pragma unroll
      for (int p = 0; p < DPASS; ++p) {
        const int idx = t + 256 * p;
        *reinterpret_cast<v4f*>(D + (idx / DF4) * NT + (idx % DF4) * 4) = xd[p];
      }
    };
    const int last = nchunks - 1;
    if (nchunks > 1) loadS(1, raA, rdA);
    int c = 0;
    for (; c + 2 < nchunks; c += 2) {                      // set A holds chunk c + 1
      loadS(c + 2, raB, rdB);
      compute(c & 1);
      storeS((c + 1) & 1, raA, rdA);
      __syncthreads();
      loadS(c + 3 < nchunks ? c + 3 : last, raA, rdA);     // past the end: the last chunk again (never stored)
      compute((c + 1) & 1);
      storeS(c & 1, raB, rdB);                             // chunk c + 2
      __syncthreads();
    }
    if (c + 1 < nchunks) {                                 // an even number of chunks: chunk c + 1 is still in set A
      compute(c & 1);
      storeS((c + 1) & 1, raA, rdA);
      __syncthreads();
    }
  } else
#endif
  if (gemm) {
    for (int c = 0; c + 1 < nchunks; ++c) {
      if (noact_g) load_chunk_gemm(c + 1, std::true_type{}); else load_chunk_gemm(c + 1, std::false_type{});
      compute(c & 1);
      store_chunk((c + 1) & 1);
      __syncthreads();
    }
  } else {
  for (int c = 0; c + 1 < nchunks; ++c) {
    if (ABL == 0 || ABL == 2) load_chunk(ABL == 2 ? 1 : c + 1);   // ABL 2: same rows every chunk (L1/L2-hot)
    compute(c & 1);
    if (ABL == 0 || ABL == 2) store_chunk((c + 1) & 1);
    __syncthreads();
  }
  }
  if (nchunks > 0) compute((nchunks - 1) & 1);

  float* wsp = ws + (size_t)split * K * g.Cout;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;     // MFMA output row held in register r
      const int k = k0 + wm * TM * 32 + (SPLIT_M ? TM * row + i : 32 * i + row);
      if (k >= K) continue;
      if (SPLIT_N && TN == 2) {
        const int n = n0 + wn * TN * 32 + 2 * li;
        float* q = wsp + (size_t)k * g.Cout + n;
        if (n + 1 < g.Cout && (g.Cout & 1) == 0) *reinterpret_cast<float2*>(q) = make_float2(acc[i][0][r], acc[i][1 % TN][r]);
        else {
          if (n < g.Cout) q[0] = acc[i][0][r];
          if (n + 1 < g.Cout) q[1] = acc[i][1 % TN][r];
        }
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * TN * 32 + (SPLIT_N ? TN * li + j : 32 * j + li);
          if (n < g.Cout) wsp[(size_t)k * g.Cout + n] = acc[i][j][r];
        }
      }
    }
}

// =====================================================================================
// Backward-weight with dx-reuse ("wgrad3"): one block owns the three dx taps of a (dz,dy) pair for a
// group of 64 input channels and 128 output channels: dW[(dz,dy,dx)][c0..c0+63][n0..n0+127].  The A
// image holds the dx = 0 voxel lines (natural [m][64] layout) with a zero row after every x-line, so
// tap dx reads the SAME image at row offset dx; per 32-voxel chunk we stage 8 KB of A + 16 KB of dy
// for 96 MFMAs per wave (the plain kernel stages 32 KB for 64).  Waves split N (32 columns each);
// lane i fetches channels 2i,2i+1 with one ds_read_b64 per tap (component-split), 6 accumulators.
// Requires 27 taps, S <= 32 (chunks are whole x-lines), Cin % 64 == 0 with sources 64-aligned.
// =====================================================================================
template <bool AFF, bool UP, bool HALO>
__global__ __launch_bounds__(256, HALO ? 2 : 3) void conv_wgrad3_kernel(ConvGeom g, ConvSrc s0, ConvSrc s1,
                                                           const float* __restrict__ dy, int ldy, int n_load,
                                                           float* __restrict__ ws, int cgroups, int ntiles,
                                                           int rows_per_split) {
  constexpr int CK = 64, NT = 128;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int S = g.S, lg = g.lgS;
  // S <= 32: a chunk is 32/S whole x-lines, each followed by a zero row.  S = 64: a chunk is half a line; rows
  // 0 / 33 then hold the real neighbours x0-1 / x0+32 (zero at the line ends), loaded with every chunk.
  constexpr bool halo = HALO;   // S = 64
  const int arows = 32 + (halo ? 1 : (32 >> lg)) + 1;
  const int A_FLOATS = arows * CK;
  constexpr int D_FLOATS = 32 * NT;
  float* As = smem;                    // [2][arows][64]
  float* Ds = smem + 2 * A_FLOATS;     // [2][32][128]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int M = g.B << (3 * lg);
  const int K = g.taps * g.Cin;

  int bid = blockIdx.x;
  const int nt_i = bid % ntiles; bid /= ntiles;
  const int cg = bid % cgroups; bid /= cgroups;
  const int gzy = bid % 9; bid /= 9;
  const int split = bid;
  const int n0 = nt_i * NT;
  const int c0 = cg * CK;
  const int m_begin = split * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  const int nchunks = (m_end - m_begin + 31) >> 5;

  // block-uniform tap pair and source
  const int dz = gzy / 3 - 1, dyy = gzy % 3 - 1;
  const int sdelta = (dz * S + dyy) * S;
  const int ybad = dyy < 0 ? 0 : (dyy > 0 ? S - 1 : -1);
  const int zbad = dz < 0 ? 0 : (dz > 0 ? S - 1 : -1);
  const bool first = c0 < s0.C;
  const float* sp = first ? s0.p : s1.p;
  const int sC = first ? s0.C : s1.C, su = first ? s0.up : s1.up;
  const float slope = act_slope_of(first ? s0.act : s1.act);
  const int ac4 = t & 15;                                  // this thread's float4 column of the A tile
  const int cl = (first ? c0 : c0 - s0.C) + ac4 * 4;
  // the block's 64 BN scale/shift values live in LDS (re-read per chunk): 8 fewer live VGPRs, which is what
  // lets the affine variants fit 3 waves/SIMD (168 registers) without spilling
  float* Aff = Ds + 2 * D_FLOATS;      // [2][64]
  if (AFF && t < 32) {
    const float* src = (t < 16) ? (first ? s0.scale : s1.scale) : (first ? s0.shift : s1.shift);
    *reinterpret_cast<v4f*>(Aff + t * 4) = *reinterpret_cast<const v4f*>(src + (first ? c0 : c0 - s0.C) + (t & 15) * 4);
  }

  v4f ra[2], rd[4], rah = v4f{0.f, 0.f, 0.f, 0.f};
  const int hside = (t >> 4) & 1;   // halo loaders: threads 0..15 left row, 16..31 right row
  auto load_chunk = [&](int c) {
    const int mbase = m_begin + (c << 5);
    const int Sh = S >> 1;
    if (halo) {
      const int x0 = mbase & (S - 1);
      const int m = hside ? mbase + 32 : mbase - 1;
      const int y = (mbase >> lg) & (S - 1), z = (mbase >> (2 * lg)) & (S - 1);
      const bool inb = t < 32 && mbase < m_end && (hside ? x0 == 0 : x0 != 0) && y != ybad && z != zbad;
      int idx = m + sdelta;
      if (UP) {
        const int x = m & (S - 1), b = mbase >> (3 * lg);
        const int idx_up = ((b * Sh + ((z + dz) >> 1)) * Sh + ((y + dyy) >> 1)) * Sh + (x >> 1);
        idx = su ? idx_up : idx;
      }
      const unsigned off = inb ? (unsigned)idx * (unsigned)sC + cl : (unsigned)cl;
      v4f v = *reinterpret_cast<const v4f*>(sp + off);
      if (AFF) v = affine_act4(v, *reinterpret_cast<const v4f*>(Aff + ac4 * 4), *reinterpret_cast<const v4f*>(Aff + 64 + ac4 * 4), slope);
      rah = inb ? v : v4f{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int m = mbase + (t >> 4) + 16 * p;
      const int y = (m >> lg) & (S - 1), z = (m >> (2 * lg)) & (S - 1);
      const bool inb = m < m_end && y != ybad && z != zbad;
      int idx = m + sdelta;
      if (UP) {
        const int x = m & (S - 1), b = m >> (3 * lg);
        const int idx_up = ((b * Sh + ((z + dz) >> 1)) * Sh + ((y + dyy) >> 1)) * Sh + (x >> 1);
        idx = su ? idx_up : idx;
      }
      const unsigned off = inb ? (unsigned)idx * (unsigned)sC + cl : (unsigned)cl;
      v4f v = *reinterpret_cast<const v4f*>(sp + off);
      if (AFF) v = affine_act4(v, *reinterpret_cast<const v4f*>(Aff + ac4 * 4), *reinterpret_cast<const v4f*>(Aff + 64 + ac4 * 4), slope);
      ra[p] = inb ? v : v4f{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = mbase + (t >> 5) + 8 * p;
      const int n = n0 + (t & 31) * 4;
      const float* q = dy + (size_t)min(m, M - 1) * ldy;
      v4f v = *reinterpret_cast<const v4f*>(q + max(min(n, n_load - 4), 0));
      if (!(m < m_end && n < n_load)) v = v4f{0.f, 0.f, 0.f, 0.f};
      rd[p] = v;
    }
  };
  auto store_chunk = [&](int buf) {
    float* A = As + buf * A_FLOATS;
    float* D = Ds + buf * D_FLOATS;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = (t >> 4) + 16 * p;
      *reinterpret_cast<v4f*>(A + (r + (r >> lg) + 1) * CK + ac4 * 4) = ra[p];
    }
    if (halo && t < 32) *reinterpret_cast<v4f*>(A + (hside ? 33 : 0) * CK + ac4 * 4) = rah;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<v4f*>(D + ((t >> 5) + 8 * p) * NT + (t & 31) * 4) = rd[p];
  };

  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // zero separator rows (row 0 and the row after every x-line) of both A buffers
  {
    const int lines = 32 >> lg;
    for (int i = t; i < 2 * (lines + 1) * CK; i += 256) {
      const int b = i / ((lines + 1) * CK), rem = i - b * (lines + 1) * CK;
      As[b * A_FLOATS + (rem / CK) * (S + 1) * CK + rem % CK] = 0.f;
    }
  }
  __syncthreads();   // the BN affine vectors in LDS are read by the very first load_chunk
  if (nchunks > 0) {
    load_chunk(0);
    store_chunk(0);
  }
  __syncthreads();
  auto compute = [&](int buf) {
    const float* A = As + buf * A_FLOATS + 2 * li;
    const float* D = Ds + buf * D_FLOATS + 32 * wave + li;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int prow = 2 * s + ((2 * s) >> lg) + 1 + lh;     // padded A row of voxel 2s+lh (dx = 0)
      const float b = D[(2 * s + lh) * NT];
      float2 a[3];
#pragma unroll
      for (int dxi = 0; dxi < 3; ++dxi) a[dxi] = *reinterpret_cast<const float2*>(A + (prow + dxi - 1) * CK);
#pragma unroll
      for (int dxi = 0; dxi < 3; ++dxi) {
        acc[dxi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dxi].x, b, acc[dxi][0], 0, 0, 0);
        acc[dxi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dxi].y, b, acc[dxi][1], 0, 0, 0);
      }
    }
  };
  for (int c = 0; c + 1 < nchunks; ++c) {
    load_chunk(c + 1);
    compute(c & 1);
    store_chunk((c + 1) & 1);
    __syncthreads();
  }
  if (nchunks > 0) compute((nchunks - 1) & 1);

  float* wsp = ws + (size_t)split * K * g.Cout;
  const int n = n0 + 32 * wave + li;
#pragma unroll
  for (int dxi = 0; dxi < 3; ++dxi)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int k = (gzy * 3 + dxi) * g.Cin + c0 + 2 * row + j;
        if (n < g.Cout) wsp[(size_t)k * g.Cout + n] = acc[dxi][j][r];
      }
}

// sub_rows > 0: the GEMM covered a channel subset: its row k = tap*sub_rows + c lands on row
// tap*row_pitch + row_off + c of the full [taps*Cin][N] weight-gradient tensor.
__global__ void reduce_splits_kernel(const float* __restrict__ ws, int nsplit, size_t n_elems,
                                     int N, float* __restrict__ dw, int ldw, int sub_rows, int row_pitch,
                                     int row_off) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_elems) return;
  float s = 0.f;
  for (int p = 0; p < nsplit; ++p) s += ws[(size_t)p * n_elems + i];   // fixed order: deterministic
  size_t k = i / N;
  const size_t n = i - k * N;
  if (sub_rows > 0) k = (k / sub_rows) * row_pitch + row_off + k % sub_rows;
  dw[k * ldw + n] = s;
}

// Many splits of a small weight tensor (thin layers: n_elems ~ 1e4, nsplit up to 512): 64 elements per
// block, the split loop shared by 4 thread rows, combined in fixed order (deterministic).
__global__ __launch_bounds__(256) void reduce_splits_wide_kernel(const float* __restrict__ ws, int nsplit,
                                                                  size_t n_elems, int N, float* __restrict__ dw,
                                                                  int ldw, int sub_rows, int row_pitch, int row_off) {
  __shared__ float part[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + tx;
  float s = 0.f;
  if (i < n_elems) {
#pragma unroll 4
    for (int p = ty; p < nsplit; p += 8) s += ws[(size_t)p * n_elems + i];
  }
  part[ty][tx] = s;
  __syncthreads();
  if (ty != 0 || i >= n_elems) return;
  s = part[0][tx];
#pragma unroll
  for (int q = 1; q < 8; ++q) s += part[q][tx];   // fixed order: deterministic
  size_t k = i / N;
  const size_t n = i - k * N;
  if (sub_rows > 0) k = (k / sub_rows) * row_pitch + row_off + k % sub_rows;
  dw[k * ldw + n] = s;
}
// four consecutive columns per thread (N, ldw multiples of 4, 16-byte aligned buffers): 16-byte loads, four of them in
// flight per thread -- the scalar kernel moved 2.5 TB/s on the large layers
__global__ __launch_bounds__(256) void reduce_splits4_kernel(const float* __restrict__ ws, int nsplit, size_t n_elems,
                                                              int N, float* __restrict__ dw, int ldw, int sub_rows,
                                                              int row_pitch, int row_off) {
  typedef float rv4 __attribute__((ext_vector_type(4)));
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n_elems) return;
  rv4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int p = 0; p < nsplit; ++p) s += *reinterpret_cast<const rv4*>(ws + (size_t)p * n_elems + i);   // fixed order
  size_t k = i / N;
  const size_t n = i - k * N;
  if (sub_rows > 0) k = (k / sub_rows) * row_pitch + row_off + k % sub_rows;
  *reinterpret_cast<rv4*>(dw + k * ldw + n) = s;
}
static int launch_reduce_splits(hipStream_t st, const float* ws, int nsplit, size_t n_elems, int N, float* dw, int ldw,
                                int sub_rows, int row_pitch, int row_off) {
  const bool v4 = N % 4 == 0 && ldw % 4 == 0 && n_elems % 4 == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(dw) & 15) == 0;
  if (v4 && !(nsplit >= 16 && n_elems * 4 < (size_t)1 << 20))
    ICS_LAUNCH(reduce_splits4_kernel, dim3((unsigned)((n_elems / 4 + 255) / 256)), dim3(256), 0, st, ws, nsplit,
                       n_elems, N, dw, ldw, sub_rows, row_pitch, row_off);
  else if (nsplit >= 16 && n_elems * 4 < (size_t)1 << 20)
    ICS_LAUNCH(reduce_splits_wide_kernel, dim3((unsigned)((n_elems + 31) / 32)), dim3(256), 0, st, ws, nsplit,
                       n_elems, N, dw, ldw, sub_rows, row_pitch, row_off);
  else
    ICS_LAUNCH(reduce_splits_kernel, dim3((unsigned)((n_elems + 255) / 256)), dim3(256), 0, st, ws, nsplit,
                       n_elems, N, dw, ldw, sub_rows, row_pitch, row_off);
  ICS_HIP(hipGetLastError());
  return 0;
}

int launch_wgrad_reduce_splits(hipStream_t st, const float* ws, int nsplit, size_t n_elems, int N, float* dw, int ldw,
                               int sub_rows, int row_pitch, int row_off) {
  return launch_reduce_splits(st, ws, nsplit, n_elems, N, dw, ldw, sub_rows, row_pitch, row_off);
}

struct WgradPlan {
  int kt, nt, ktiles, ntiles, ksplit, rows_per_split;
  bool vec, thin;
};

static int pick_ksplit(long base, long M, long slots, long r0 = 1);
static WgradPlan plan_wgrad(const ConvGeom& g, const ConvSrc& s0, int nsrc, const ConvSrc& s1) {
  WgradPlan p;
  const int M = g.B << (3 * g.lgS);
  p.thin = conv_is_thin(g, s0, nsrc);
  p.vec = p.thin || ((g.Cin % 32 == 0) && (s0.C % 32 == 0) && s0.bcast == 0 && (nsrc < 2 || s1.bcast == 0));
  if (p.vec) {
    p.kt = (g.taps * g.Cin >= 128) ? 128 : (g.taps * g.Cin >= 64 ? 64 : 32);
    p.ktiles = (g.taps * g.Cin + p.kt - 1) / p.kt;
  } else {
    p.kt = 32;
    p.ktiles = g.Kpad / 32;
  }
  // available (kt, nt) layouts: 128x{128,96,64,32}, 64x{128,64}, 32x128; nt need not divide Npad
  const int n32 = g.Npad;
  if (p.kt == 128) p.nt = (n32 % 128 == 0) ? 128 : (n32 == 96) ? 96 : (n32 >= 128) ? 128 : (n32 == 64) ? 64 : 32;
  else if (p.kt == 64) p.nt = (n32 <= 64) ? 64 : 128;
  else p.nt = 128;
  p.ntiles = (g.Npad + p.nt - 1) / p.nt;
  // resident blocks per CU follow the LDS footprint 2*32*(kt+nt)*4 B: 2 for the 128x128 tile, up to 4 for thin ones
  const long per_cu = std::max(1L, std::min(4L, 163840L / (2L * 32 * (p.kt + p.nt) * 4)));
  const long want = pick_ksplit((long)p.ktiles * p.ntiles, M, 256 * per_cu);
  int rows = (int)(((M + want - 1) / want + 31) / 32 * 32);
  p.rows_per_split = rows;
  p.ksplit = (M + rows - 1) / rows;
  return p;
}

// Split-K count: base*ks blocks should fill whole "rounds" of the resident block slots (1539 blocks on 512
// slots = 3.006 rounds leave a nearly empty 4th round: measured -20 %).  Every split costs a pass of the
// reduction over the weight tensor, so take the FEWEST rounds whose last round is >= 95 % full (c18: 170 splits
// filled 6 rounds and cost a 300 MB reduction; 42 splits fill one), else the best fill; splits keep >= 256 rows.
static int pick_ksplit(long base, long M, long slots, long r0) {
  const long ks_max = std::max<long>(1, M / 256);
  long want = 1;
  double best = -1.0;
  for (long r = r0; r <= 6; ++r) {
    long ks = (r * slots) / base;
    if (ks < 1) ks = 1;
    if (ks > 512) ks = 512;
    if (ks > ks_max) { if (r > r0) break; ks = ks_max; }   // small M: as many >= 256-row splits as there are
    const long total = base * ks;
    const double fill = (double)total / (double)(((total + slots - 1) / slots) * slots);
    if (fill > best + 1e-9) { best = fill; want = ks; }
    if (fill >= 0.95) break;
  }
  return (int)want;
}

// wgrad3 with wave-uniform loaders (the hot variant): S = 2^LGC in {16, 32, 64}, M % 32 == 0, one same-resolution
// source per 64-channel group, Cout % 128 == 0.  A 32-voxel chunk then lies inside one x-line (or two, S = 16), so
// its (y, z) validity is a SCALAR; every global address is a wave-uniform base plus a per-thread byte offset fixed
// for the whole kernel, LDS offsets are immediates, and no row / column bound checks remain:
// 134 -> ~35 vector ALU instructions per 96 MFMAs (fp32 VALU and fp32 MFMA share the issue pipe).
template <bool AFF, bool NOACT, int LGC>
__global__ __launch_bounds__(256, LGC == 6 ? 2 : 3) void conv_wgrad3s_kernel(ConvGeom g, ConvSrc s0, ConvSrc s1,
                                                                              const float* __restrict__ dy, int ldy,
                                                                              float* __restrict__ ws, int cgroups,
                                                                              int ntiles, int rows_per_split) {
  constexpr int CK = 64, NT = 128;
  constexpr int lg = LGC, S = 1 << LGC;
  constexpr bool halo = LGC == 6;
  constexpr int lines = halo ? 1 : (32 >> lg);           // x-lines (zero separators) per chunk
  constexpr int arows = 32 + lines + 1;
  constexpr int A_FLOATS = arows * CK, D_FLOATS = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][arows][64]
  float* Ds = smem + 2 * A_FLOATS;     // [2][32][128]
  float* Aff = Ds + 2 * D_FLOATS;      // [2][64] BN scale | shift of this block's channels

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int M = g.B << (3 * lg);
  const int K = g.taps * g.Cin;

  int bid = blockIdx.x;
  const int nt_i = bid % ntiles; bid /= ntiles;
  const int cg = bid % cgroups; bid /= cgroups;
  const int gzy = bid % 9; bid /= 9;
  const int split = bid;
  const int n0 = nt_i * NT;
  const int c0 = cg * CK;
  const int m_begin = split * rows_per_split;            // multiples of 32 (M % 32 == 0, rows_per_split % 32 == 0)
  const int m_end = min(M, m_begin + rows_per_split);
  const int nchunks = (m_end - m_begin) >> 5;

  const int dz = gzy / 3 - 1, dyy = gzy % 3 - 1;
  const int sdelta = (dz * S + dyy) * S;
  const int ybad = dyy < 0 ? 0 : (dyy > 0 ? S - 1 : -1);
  const int zbad = dz < 0 ? 0 : (dz > 0 ? S - 1 : -1);
  const bool first = c0 < s0.C;
  const char* sp = reinterpret_cast<const char*>(first ? s0.p : s1.p);
  const int sC = first ? s0.C : s1.C;
  const float slope = act_slope_of(first ? s0.act : s1.act);
  const int ac4 = t & 15;
  const int cl = (first ? c0 : c0 - s0.C) + ac4 * 4;
  if (AFF && t < 32) {
    const float* src = (t < 16) ? (first ? s0.scale : s1.scale) : (first ? s0.shift : s1.shift);
    *reinterpret_cast<v4f*>(Aff + t * 4) = *reinterpret_cast<const v4f*>(src + (first ? c0 : c0 - s0.C) + (t & 15) * 4);
  }
  // per-thread byte offsets, fixed for the kernel
  const unsigned aoff = (unsigned)((t >> 4) * sC + cl) * 4u;                 // row (t>>4) of a 16-row half
  const unsigned hoff = (unsigned)cl * 4u;                                   // halo row (S = 64)
  unsigned doff[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) doff[p] = (unsigned)(((t >> 5) + 8 * p) * ldy + n0 + (t & 31) * 4) * 4u;
  const int hside = (t >> 4) & 1;

  v4f ra[2], rd[4], rah = v4f{0.f, 0.f, 0.f, 0.f};
  auto load_chunk = [&](int c) {
    const int mbase = m_begin + (c << 5);
    v4f sc = v4f{1.f, 1.f, 1.f, 1.f}, sh = v4f{0.f, 0.f, 0.f, 0.f};
    if (AFF) { sc = *reinterpret_cast<const v4f*>(Aff + ac4 * 4); sh = *reinterpret_cast<const v4f*>(Aff + 64 + ac4 * 4); }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row0 = mbase + 16 * p;                                       // uniform
      const int y = (row0 >> lg) & (S - 1), z = (row0 >> (2 * lg)) & (S - 1);
      const bool inb = (y != ybad) & (z != zbad);                            // scalar
      const float msk = inb ? 1.f : 0.f;
      const char* base = sp + (inb ? (long)(row0 + sdelta) * sC * 4 : 0L);   // wave-uniform pointer
      v4f v = *reinterpret_cast<const v4f*>(base + aoff);
      if (AFF) v = affine_only_or_act4<NOACT>(v, sc, sh, slope);
      ra[p] = v4f{v.x * msk, v.y * msk, v.z * msk, v.w * msk};               // zero padding AFTER BN / activation
    }
    if (halo) {   // S = 64: rows 0 / 33 of the tile are the real neighbours x0-1 / x0+32 (zero at the line ends)
      const int x0 = mbase & (S - 1);
      const int y = (mbase >> lg) & (S - 1), z = (mbase >> (2 * lg)) & (S - 1);
      const bool lineok = (y != ybad) & (z != zbad);
      const bool inl = lineok & (x0 != 0), inr = lineok & (x0 == 0);         // scalars
      const char* bl = sp + (inl ? (long)(mbase - 1 + sdelta) * sC * 4 : 0L);
      const char* br = sp + (inr ? (long)(mbase + 32 + sdelta) * sC * 4 : 0L);
      const float msk = (hside ? inr : inl) ? 1.f : 0.f;
      v4f v = *reinterpret_cast<const v4f*>((hside ? br : bl) + hoff);
      if (AFF) v = affine_only_or_act4<NOACT>(v, sc, sh, slope);
      rah = v4f{v.x * msk, v.y * msk, v.z * msk, v.w * msk};
    }
    const char* dbase = reinterpret_cast<const char*>(dy) + (size_t)mbase * (size_t)ldy * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) rd[p] = *reinterpret_cast<const v4f*>(dbase + doff[p]);
  };
  // LDS positions of this thread's stores (constant)
  float* a_st[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = (t >> 4) + 16 * p;
    a_st[p] = As + (r + (halo ? 0 : (r >> lg)) + 1) * CK + ac4 * 4;
  }
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<v4f*>(a_st[p] + buf * A_FLOATS) = ra[p];
    if (halo && t < 32) *reinterpret_cast<v4f*>(As + buf * A_FLOATS + (hside ? 33 : 0) * CK + ac4 * 4) = rah;
    float* D = Ds + buf * D_FLOATS;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<v4f*>(D + ((t >> 5) + 8 * p) * NT + (t & 31) * 4) = rd[p];
  };

  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (!halo) {   // zero separator rows (row 0 and the row after every x-line) of both A buffers
    for (int i = t; i < 2 * (lines + 1) * CK; i += 256) {
      const int b = i / ((lines + 1) * CK), rem = i - b * (lines + 1) * CK;
      As[b * A_FLOATS + (rem / CK) * (S + 1) * CK + rem % CK] = 0.f;
    }
  }
  __syncthreads();   // the BN affine vectors in LDS are read by the very first load_chunk
  if (nchunks > 0) {
    load_chunk(0);
    store_chunk(0);
  }
  __syncthreads();
  auto compute = [&](int buf) {
    const float* A = As + buf * A_FLOATS + 2 * li + lh * CK;
    const float* D = Ds + buf * D_FLOATS + 32 * wave + li + lh * NT;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      constexpr int dummy = 0; (void)dummy;
      const int prow = 2 * s + (halo ? 0 : ((2 * s) >> lg)) + 1;   // compile-time: padded A row of voxel 2s (+lh above)
      const float b = D[(2 * s) * NT];
      float2 a[3];
#pragma unroll
      for (int dxi = 0; dxi < 3; ++dxi) a[dxi] = *reinterpret_cast<const float2*>(A + (prow + dxi - 1) * CK);
#pragma unroll
      for (int dxi = 0; dxi < 3; ++dxi) {
        acc[dxi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dxi].x, b, acc[dxi][0], 0, 0, 0);
        acc[dxi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dxi].y, b, acc[dxi][1], 0, 0, 0);
      }
    }
  };
  for (int c = 0; c + 1 < nchunks; ++c) {
    load_chunk(c + 1);
    compute(c & 1);
    store_chunk((c + 1) & 1);
    __syncthreads();
  }
  if (nchunks > 0) compute((nchunks - 1) & 1);

  float* wsp = ws + (size_t)split * K * g.Cout;
  const int n = n0 + 32 * wave + li;
#pragma unroll
  for (int dxi = 0; dxi < 3; ++dxi)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int k = (gzy * 3 + dxi) * g.Cin + c0 + 2 * row + j;
        wsp[(size_t)k * g.Cout + n] = acc[dxi][j][r];
      }
}

// can the dx-reuse kernel run this geometry?  (27 taps, whole x-lines per 32-voxel chunk, 64-aligned
// channel groups inside one source, full 128-wide vectorisable dy rows)
static bool wgrad3_ok(const ConvGeom& g, const ConvSrc& s0, int nsrc, const ConvSrc& s1) {
  if ((g.flags & CF_NO_WGRAD3) || g.taps != 27 || g.S > 64 || g.S < 2 || g.Cin % 64 != 0 || g.Cout % 128 != 0) return false;
  if (s0.bcast || (nsrc > 1 && s1.bcast)) return false;
  if (nsrc > 1 && (s0.C % 64 != 0)) return false;
  return true;
}
struct Wgrad3Plan { int cgroups, ntiles, ksplit, rows_per_split; };
static Wgrad3Plan plan_wgrad3(const ConvGeom& g) {
  Wgrad3Plan p;
  const long M = (long)g.B << (3 * g.lgS);
  p.cgroups = g.Cin / 64;
  p.ntiles = g.Cout / 128;
  // 3 blocks/CU (2 at S = 64); two rounds: the second round evens out the tail of the first (c18: 130.5 ->
  // 132.2 TFLOP/s) for one more pass of the split reduction (+0.12 ms/step over all layers) - a wash on the
  // step, taken for the better-balanced dominant kernel
  const int want = pick_ksplit(9L * p.cgroups * p.ntiles, M, g.lgS > 5 ? 512 : 768, 2);
  p.rows_per_split = (int)(((M + want - 1) / want + 31) / 32 * 32);
  p.ksplit = (int)((M + p.rows_per_split - 1) / p.rows_per_split);
  return p;
}


size_t conv_wgrad_workspace_floats(const ConvGeom& g, const ConvSrc* src, int nsrc) {
  if (thin_n_ok(g, src[0], nsrc)) return (size_t)thin_n_wgrad_splits(g) * g.taps * g.Cin * g.Cout;
  ConvSrc s0 = src[0], s1 = src[nsrc > 1 ? 1 : 0];
  WgradPlan p = plan_wgrad(g, s0, nsrc, s1);
  int ks = p.ksplit;
  if (wgrad3_ok(g, s0, nsrc, s1)) ks = std::max(ks, plan_wgrad3(g).ksplit);
  return (size_t)ks * g.taps * g.Cin * g.Cout;
}

template <int WM, int WN, int TM, int TN, bool VEC, bool DYVEC, bool AFF = true, bool UP = true, bool THIN = false,
          int ABL = 0>
static int launch_wgrad_cfg(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const ConvSrc& s1,
                            const float* dy, int ldy, int n_load, float* ws, const WgradPlan& p) {
  constexpr int KT = WM * TM * 32, NT = WN * TN * 32;
  const size_t lds = (size_t)2 * 32 * (KT + NT) * sizeof(float);
  auto kern = conv_wgrad_kernel<WM, WN, TM, TN, VEC, DYVEC, AFF, UP, THIN, ABL>;
  static DevOnce attr;
  int dev;
  if (attr.need(&dev)) {
    ICS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr.mark(dev);
  }
  static const std::string id = std::string("conv_wgrad_kernel<") + std::to_string(WM) + ", " + std::to_string(WN) +
                                ", " + std::to_string(TM) + ", " + std::to_string(TN) + ", " + tf(VEC) + ", " +
                                tf(DYVEC) + ", " + tf(AFF) + ", " + tf(UP) + ", " + tf(THIN) + ", " +
                                std::to_string(ABL) + ">";
  g_last_kernel_id = id.c_str();
  ICS_LAUNCH(kern, dim3(p.ktiles * p.ntiles * p.ksplit), dim3(256), lds, st, g, s0, s1, dy,
                     ldy, n_load, ws, p.ktiles, p.ntiles, p.rows_per_split);
  ICS_HIP(hipGetLastError());
  return 0;
}

// benchmarking-only: 128x128 vector wgrad with the staging removed (1) or made cache-hot (2)
int launch_conv_wgrad_ablate(hipStream_t st, const ConvGeom& g, const ConvSrc* src, const float* dy, int ldy,
                             float* workspace, int ablate) {
  ConvSrc s0 = src[0], s1 = src[0];
  s1.C = 0;
  ICS_TRY(fix_src(s0));
  ICS_TRY(fix_src(s1));
  WgradPlan p = plan_wgrad(g, s0, 1, s1);
  ICS_CHECK(p.vec && p.kt == 128 && p.nt == 128, "ablation bench needs the vector 128x128 configuration");
  if (ablate == 1) return launch_wgrad_cfg<2, 2, 2, 2, true, true, false, false, false, 1>(st, g, s0, s1, dy, ldy, g.Cout, workspace, p);
  return launch_wgrad_cfg<2, 2, 2, 2, true, true, false, false, false, 2>(st, g, s0, s1, dy, ldy, g.Cout, workspace, p);
}

int launch_conv_wgrad(hipStream_t st, const ConvGeom& g, const ConvSrc* src, int nsrc,
                      const float* dy, int ldy, float* dw, int ldw, float* workspace,
                      size_t workspace_floats, int sub_rows, int row_pitch, int row_off, int phase) {
  if (thin_n_ok(g, src[0], nsrc)) {
    const int ns = thin_n_wgrad_splits(g);
    const size_t n_el = (size_t)g.taps * g.Cin * g.Cout;
    ICS_CHECK((size_t)ns * n_el <= workspace_floats, "wgrad workspace too small");
    const int M = g.B << (3 * g.lgS);
    const int rows = (M + ns - 1) / ns;
    const int K = 27 * g.Cin;
    const bool aff = src[0].scale != nullptr;
    if (thin1_wgrad_ok(g, src[0])) {
      const int nb = thin1_wgrad_blocks(g);
      if (phase != 2) {
#define ICS_T1W(CKV)                                                                                                 \
  do {                                                                                                               \
    if (aff) ICS_LAUNCH((thin1_wgrad_kernel<CKV, true>), dim3(nb), dim3(256), 0, st, g, src[0], dy, ldy, workspace); \
    else ICS_LAUNCH((thin1_wgrad_kernel<CKV, false>), dim3(nb), dim3(256), 0, st, g, src[0], dy, ldy, workspace);    \
  } while (0)
        switch (g.Cin / 16) { case 1: ICS_T1W(1); break; case 2: ICS_T1W(2); break; case 3: ICS_T1W(3); break; default: ICS_T1W(4); break; }
#undef ICS_T1W
        ICS_HIP(hipGetLastError());
        g_last_kernel_id = "thin1_wgrad_kernel";
      }
      if (phase != 1) ICS_TRY(launch_reduce_splits(st, workspace, 4 * nb, n_el, g.Cout, dw, ldw, sub_rows, row_pitch, row_off));
      return 0;
    }
    if (phase != 2) {
#define ICS_TNW(NOUT, AFFV, KPT, NAME)                                                                         \
  do {                                                                                                         \
    g_last_kernel_id = NAME;                                                                                   \
    ICS_LAUNCH((conv_thin_n_wgrad_kernel<NOUT, AFFV, KPT>), dim3((M + rows - 1) / rows), dim3(256), 0,  \
                       st, g, src[0], dy, ldy, workspace, rows);                                               \
  } while (0)
#define ICS_TNW_K(NOUT, AFFV, TAG)                                                                  \
  do {                                                                                              \
    if (K <= 512) ICS_TNW(NOUT, AFFV, 2, "conv_thin_n_wgrad_kernel<" TAG ", 2>");                   \
    else if (K <= 1024) ICS_TNW(NOUT, AFFV, 4, "conv_thin_n_wgrad_kernel<" TAG ", 4>");             \
    else ICS_TNW(NOUT, AFFV, 7, "conv_thin_n_wgrad_kernel<" TAG ", 7>");                            \
  } while (0)
      if (g.Cout == 1) { if (aff) ICS_TNW_K(1, true, "1, true"); else ICS_TNW_K(1, false, "1, false"); }
      else { if (aff) ICS_TNW_K(4, true, "4, true"); else ICS_TNW_K(4, false, "4, false"); }
#undef ICS_TNW_K
#undef ICS_TNW
      ICS_HIP(hipGetLastError());
    }
    if (phase != 1)
      ICS_TRY(launch_reduce_splits(st, workspace, (M + rows - 1) / rows, n_el, g.Cout, dw, ldw, sub_rows, row_pitch,
                                   row_off));
    return 0;
  }
  ConvSrc s0 = src[0], s1 = src[nsrc > 1 ? 1 : 0];
  if (nsrc == 1) s1.C = 0;
  ICS_TRY(fix_src(s0));
  ICS_TRY(fix_src(s1));
  WgradPlan p = plan_wgrad(g, s0, nsrc, s1);
  const size_t n_elems = (size_t)g.taps * g.Cin * g.Cout;
  // dy may be a column slice of a wider matrix (callers pass it already offset); vector loads need
  // 16-byte aligned rows and a multiple-of-4 column count that stays inside the row.
  const int n_load4 = (g.Cout + 3) / 4 * 4;
  const bool dy_vec = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(dy) & 15) == 0) && (n_load4 <= ldy);
  const int n_load = dy_vec ? n_load4 : g.Cout;
  if (!dy_vec) {   // scalar dy loads exist only for the smallest N tile of each K tile
    p.nt = (p.kt == 128) ? 32 : (p.kt == 64) ? 64 : 128;
    p.ntiles = (g.Npad + p.nt - 1) / p.nt;
  }
  const bool aff = src[0].scale != nullptr || (nsrc > 1 && src[1].scale != nullptr);
  const bool up = src[0].up || (nsrc > 1 && src[1].up);
  if (dy_vec && wgrad3_ok(g, s0, nsrc, s1)) {
    const Wgrad3Plan q = plan_wgrad3(g);
    ICS_CHECK((size_t)q.ksplit * n_elems <= workspace_floats, "wgrad workspace too small");
    const int arows = 32 + std::max(32 >> g.lgS, 1) + 1;
    const size_t lds = (size_t)(2 * (arows * 64 + 32 * 128) + 128) * sizeof(float);
    const dim3 grid(9 * q.cgroups * q.ntiles * q.ksplit);
    const int M3 = g.B << (3 * g.lgS);
    const bool fast = !up && g.lgS >= 4 && g.lgS <= 6 && M3 % 32 == 0 && n_load == g.Cout &&
                      !(g.flags & CF_NO_WGRAD3S);
    if (phase != 2 && fast) {   // wave-uniform loaders (conv_wgrad3s_kernel)
      const bool noact = aff && (src[0].scale == nullptr || src[0].act == ACT_NONE) &&
                         (nsrc < 2 || src[1].scale == nullptr || src[1].act == ACT_NONE);
      const int ar = 32 + (g.lgS == 6 ? 1 : (32 >> g.lgS)) + 1;
      const size_t lds3 = (size_t)(2 * (ar * 64 + 32 * 128) + 128) * sizeof(float);
#define ICS_W3S(AFFV, NOACTV, LGV)                                                                              \
  do {                                                                                                          \
    g_last_kernel_id = "conv_wgrad3s_kernel<" #AFFV ", " #NOACTV ", " #LGV ">";                                 \
    ICS_LAUNCH((conv_wgrad3s_kernel<AFFV, NOACTV, LGV>), grid, dim3(256), lds3, st, g, s0, s1, dy, ldy,  \
                       workspace, q.cgroups, q.ntiles, q.rows_per_split);                                       \
  } while (0)
#define ICS_W3S_L(AFFV, NOACTV)                                                       \
  do {                                                                                \
    if (g.lgS == 4) ICS_W3S(AFFV, NOACTV, 4);                                         \
    else if (g.lgS == 5) ICS_W3S(AFFV, NOACTV, 5);                                    \
    else ICS_W3S(AFFV, NOACTV, 6);                                                    \
  } while (0)
      if (!aff) ICS_W3S_L(false, false);
      else if (noact) ICS_W3S_L(true, true);
      else ICS_W3S_L(true, false);
#undef ICS_W3S_L
#undef ICS_W3S
      ICS_HIP(hipGetLastError());
    } else if (phase != 2) {
#define ICS_W3(AFFV, UPV, HALOV)                                                                                 \
  do {                                                                                                           \
    g_last_kernel_id = "conv_wgrad3_kernel<" #AFFV ", " #UPV ", " #HALOV ">";                                    \
    ICS_LAUNCH((conv_wgrad3_kernel<AFFV, UPV, HALOV>), grid, dim3(256), lds, st, g, s0, s1, dy, ldy,      \
                       n_load, workspace, q.cgroups, q.ntiles, q.rows_per_split);                                \
  } while (0)
      if (g.lgS > 5) {
        if (up) ICS_W3(true, true, true); else if (aff) ICS_W3(true, false, true); else ICS_W3(false, false, true);
      } else {
        if (up) ICS_W3(true, true, false); else if (aff) ICS_W3(true, false, false); else ICS_W3(false, false, false);
      }
#undef ICS_W3
      ICS_HIP(hipGetLastError());
    }
    if (phase != 1) {
      ICS_TRY(launch_reduce_splits(st, workspace, q.ksplit, n_elems, g.Cout, dw, ldw, sub_rows, row_pitch, row_off));
    }
    return 0;
  }
  ICS_CHECK((size_t)p.ksplit * n_elems <= workspace_floats, "wgrad workspace too small");
  if (phase != 2) {
#define ICS_WG_ARGS st, g, s0, s1, dy, ldy, n_load, workspace, p
#define ICS_WG(WM, WN, TM, TN, DV)                                                                 \
  do {                                                                                             \
    if (p.thin) ICS_TRY((launch_wgrad_cfg<WM, WN, TM, TN, true, DV, true, true, true>(ICS_WG_ARGS)));  \
    else if (!p.vec) ICS_TRY((launch_wgrad_cfg<WM, WN, TM, TN, false, DV, true, true>(ICS_WG_ARGS)));   \
    else if (up) ICS_TRY((launch_wgrad_cfg<WM, WN, TM, TN, true, DV, true, true>(ICS_WG_ARGS)));   \
    else if (aff) ICS_TRY((launch_wgrad_cfg<WM, WN, TM, TN, true, DV, true, false>(ICS_WG_ARGS))); \
    else ICS_TRY((launch_wgrad_cfg<WM, WN, TM, TN, true, DV, false, false>(ICS_WG_ARGS)));         \
  } while (0)
  if (!dy_vec) {
    if (p.kt == 128) ICS_WG(4, 1, 1, 1, false);
    else if (p.kt == 64) ICS_WG(2, 2, 1, 1, false);
    else ICS_WG(1, 4, 1, 1, false);
  } else if (p.kt == 128) {
    if (p.nt == 128) ICS_WG(2, 2, 2, 2, true);
    else if (p.nt == 96) ICS_WG(4, 1, 1, 3, true);
    else if (p.nt == 64) ICS_WG(4, 1, 1, 2, true);
    else ICS_WG(4, 1, 1, 1, true);
  } else if (p.kt == 64) {
    if (p.nt == 128) ICS_WG(2, 2, 1, 2, true);
    else ICS_WG(2, 2, 1, 1, true);
  } else {
    ICS_WG(1, 4, 1, 1, true);
  }
#undef ICS_WG_ARGS
#undef ICS_WG
  }
  if (phase != 1) {
    ICS_TRY(launch_reduce_splits(st, workspace, p.ksplit, n_elems, g.Cout, dw, ldw, sub_rows, row_pitch, row_off));
  }
  return 0;
}

// =====================================================================================
// Weight packing
// =====================================================================================
// dst[(kq*Npad + n)*4 + t] = w[(k_src)*N + n_src] with k = kq*4+t; region outside is zero-filled
// only when zero_first (the head packs two weight tensors into one buffer).
// cin_phys > 0: the GEMM runs on an input zero-padded from cin_log to cin_phys channels per tap
// (K = taps*cin_phys); rows of the padding channels are zero.
__device__ __forceinline__ bool pack_fwd_value(size_t i, const float* __restrict__ w, int K, int N, int Npad, int k_off,
                                               int n_off, int cin_log, int cin_phys, float* v) {
  const int tq = i & 3;
  const size_t rest = i >> 2;
  const int n = rest % Npad;
  const int k = (int)(rest / Npad) * 4 + tq;
  int ks = k - k_off;
  const int ns = n - n_off;
  bool ok = ks >= 0 && ks < K && ns >= 0 && ns < N;
  if (ok && cin_phys > 0) {
    const int tap = ks / cin_phys, ci = ks - tap * cin_phys;
    ok = ci < cin_log;
    ks = tap * cin_log + ci;
  }
  *v = ok ? w[(size_t)ks * N + ns] : 0.f;
  return ok;
}
__global__ void pack_fwd_kernel(const float* __restrict__ w, int K, int N, float* __restrict__ dst,
                                int Kpad, int Npad, int k_off, int n_off, int zero_first, int cin_log,
                                int cin_phys) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)Kpad * Npad;
  if (i >= total) return;
  float v;
  if (pack_fwd_value(i, w, K, N, Npad, k_off, n_off, cin_log, cin_phys, &v) || zero_first) dst[i] = v;
}

// materialise the virtual (concatenated / broadcast / BN-applied) input zero-padded to CinG channels
__global__ void materialize_input_kernel(ConvSrc s0, ConvSrc s1, int Cin, int CinG, int S, int lg, size_t total,
                                         float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % CinG);
  const RowPos r = decode_row((int)(i / CinG), S, lg);
  out[i] = c < Cin ? gather_scalar(s0, s1, c, r.b, r.z, r.y, r.x, S) : 0.f;
}
int launch_materialize_input(hipStream_t st, const ConvSrc* src, int nsrc, int Cin, int CinG, int B, int S,
                             float* out) {
  ConvSrc s0 = src[0], s1 = src[nsrc > 1 ? 1 : 0];
  if (nsrc == 1) s1.C = 0;
  int lg = 0;
  while ((1 << lg) < S) ++lg;
  const size_t total = (size_t)B * S * S * S * CinG;
  ICS_LAUNCH(materialize_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s0, s1, Cin,
                     CinG, S, lg, total, out);
  ICS_HIP(hipGetLastError());
  return 0;
}

// backward-data weights: k' = t'*cout_total + co_off + co, n' = ci, value = w[26-t'][ci][co]
__device__ __forceinline__ bool pack_bwd_value(size_t i, const float* __restrict__ w, int taps, int Cin, int Cout,
                                               int Npad, int cout_total, int co_off, float* v) {
  const int tq = i & 3;
  const size_t rest = i >> 2;
  const int n = rest % Npad;
  const int k = (int)(rest / Npad) * 4 + tq;
  const int tp = k / cout_total, co = k - tp * cout_total - co_off;
  const bool ok = tp < taps && co >= 0 && co < Cout && n < Cin;
  *v = ok ? w[((size_t)(taps - 1 - tp) * Cin + n) * Cout + co] : 0.f;
  return ok;
}
__global__ void pack_bwd_kernel(const float* __restrict__ w, int taps, int Cin, int Cout,
                                float* __restrict__ dst, int Kpad, int Npad, int cout_total,
                                int co_off, int zero_first) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)Kpad * Npad;
  if (i >= total) return;
  float v;
  if (pack_bwd_value(i, w, taps, Cin, Cout, Npad, cout_total, co_off, &v) || zero_first) dst[i] = v;
}

// Packed GEMM weights over a channel SUBSET [c_off, c_off+Csub) of w[taps][Cin_total][Cout]:
//   k = tp*Cout + co, n = c:  dst[k][n] = w[flip ? taps-1-tp : tp][c_off + c][co]
// flip = 1: backward-data weights of the subset; flip = 0: the "up-split" GEMM dxl = dyS x W.
__device__ __forceinline__ float pack_sub_value(size_t i, const float* __restrict__ w, int taps, int Cin_total, int Cout,
                                                int c_off, int Csub, int flip, int Npad) {
  const int tq = i & 3;
  const size_t rest = i >> 2;
  const int n = rest % Npad;
  const int k = (int)(rest / Npad) * 4 + tq;
  const int tp = k / Cout, co = k - tp * Cout;
  float v = 0.f;
  if (tp < taps && n < Csub) v = w[((size_t)(flip ? taps - 1 - tp : tp) * Cin_total + c_off + n) * Cout + co];
  return v;
}
__global__ void pack_sub_kernel(const float* __restrict__ w, int taps, int Cin_total, int Cout, int c_off,
                                int Csub, int flip, float* __restrict__ dst, int Kpad, int Npad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)Kpad * Npad;
  if (i >= total) return;
  dst[i] = pack_sub_value(i, w, taps, Cin_total, Cout, c_off, Csub, flip, Npad);
}
int launch_pack_sub(hipStream_t st, const float* w, int taps, int Cin_total, int Cout, int c_off, int Csub,
                    int flip, float* dst, int Kpad, int Npad) {
  if (g_pack_rec) { g_pack_rec->push_back(PackJob{2, w, dst, {taps, Cin_total, Cout, c_off, Csub, flip, Npad, 0, 0}, (unsigned long long)Kpad * Npad, 0}); return 0; }
  const size_t total = (size_t)Kpad * Npad;
  ICS_LAUNCH(pack_sub_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, taps, Cin_total,
                     Cout, c_off, Csub, flip, dst, Kpad, Npad);
  ICS_HIP(hipGetLastError());
  return 0;
}

int launch_pack_fwd(hipStream_t st, const float* w, int K, int N, float* dst, int Kpad, int Npad,
                    int k_off, int n_off, int zero_first, int cin_log, int cin_phys) {
  if (g_pack_rec) { g_pack_rec->push_back(PackJob{0, w, dst, {K, N, Npad, k_off, n_off, cin_log, cin_phys, 0, 0}, (unsigned long long)Kpad * Npad, 0}); return 0; }
  const size_t total = (size_t)Kpad * Npad;
  ICS_LAUNCH(pack_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, K, N,
                     dst, Kpad, Npad, k_off, n_off, zero_first, cin_log, cin_phys);
  ICS_HIP(hipGetLastError());
  return 0;
}
int launch_pack_bwd(hipStream_t st, const float* w, int taps, int Cin, int Cout, float* dst,
                    int Kpad, int Npad, int cout_total, int co_off, int zero_first) {
  if (g_pack_rec) { g_pack_rec->push_back(PackJob{1, w, dst, {taps, Cin, Cout, Npad, cout_total, co_off, 0, 0, 0}, (unsigned long long)Kpad * Npad, 0}); return 0; }
  const size_t total = (size_t)Kpad * Npad;
  ICS_LAUNCH(pack_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, taps,
                     Cin, Cout, dst, Kpad, Npad, cout_total, co_off, zero_first);
  ICS_HIP(hipGetLastError());
  return 0;
}

// Winograd weight transform (conv_wino.hip's operand layout dst[Nn/32][K/4][64 f][2 h][32 n][2 j], k = c4*4 + h*2 + j).
// Thread t of a job = one (k, n) pair, t = (((nchunk * K/4 + c4) * 2 + h) * 32 + n32) * 2 + j: it reads the pair's 27 taps
// once, applies U = (G (x) G (x) G) g with G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1] one axis at a time and writes the
// 64 frequencies (a half-wave writes 128 contiguous floats per frequency).
// layout 1 (conv_wino64.hip): dst[Nn/64][K/4][64 f][4 k][16 n][4 column blocks]; t = ((nchunk * K/4 + c4) * 256 +
// (kk * 16 + n16) * 4 + nb: consecutive threads write consecutive floats of one frequency image.
__device__ __forceinline__ void pack_wino_pair(size_t t, const float* __restrict__ w, int Cin_total, int Cout, int c_off,
                                               int Csub, int bwd, float* __restrict__ dst, int layout) {
  const int K = bwd ? Cout : Csub;
  if (layout == 2) {
    // conv_winog.hip: 64 plain GEMM images dst[f][K/4][Nn][4] (the packed-weight layout of conv_fwd_kernel, one per
    // frequency); t = ((k >> 2) * Nn + n) * 4 + (k & 3): consecutive threads write consecutive floats of every image
    const int Nn = bwd ? Csub : Cout;
    const int k2 = (int)(t / ((size_t)4 * Nn)) * 4 + (int)(t & 3), n2 = (int)((t >> 2) % (size_t)Nn);
    float g2[27];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap)
      g2[tap] = bwd ? w[((size_t)(26 - tap) * Cin_total + c_off + n2) * Cout + k2]
                    : w[((size_t)tap * Cin_total + c_off + k2) * Cout + n2];
    const size_t fs2 = (size_t)K * Nn;
    float* d2 = dst + t;
#pragma unroll
    for (int fz = 0; fz < 4; ++fz)
#pragma unroll
      for (int fy = 0; fy < 4; ++fy)
#pragma unroll
        for (int fx = 0; fx < 4; ++fx) {
          // U[fz][fy][fx] = sum_{a,b,c} G[fz][a] G[fy][b] G[fx][c] g[a][b][c],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
          float acc = 0.f;
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const float ga = fz == 0 ? (a == 0 ? 1.f : 0.f) : fz == 3 ? (a == 2 ? 1.f : 0.f) : (fz == 2 && a == 1 ? -0.5f : 0.5f);
            if (ga == 0.f) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
              const float gb = fy == 0 ? (b == 0 ? 1.f : 0.f) : fy == 3 ? (b == 2 ? 1.f : 0.f) : (fy == 2 && b == 1 ? -0.5f : 0.5f);
              if (gb == 0.f) continue;
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                const float gc = fx == 0 ? (c == 0 ? 1.f : 0.f) : fx == 3 ? (c == 2 ? 1.f : 0.f) : (fx == 2 && c == 1 ? -0.5f : 0.5f);
                if (gc == 0.f) continue;
                acc += (ga * gb * gc) * g2[(a * 3 + b) * 3 + c];
              }
            }
          }
          d2[(size_t)((fz * 4 + fy) * 4 + fx) * fs2] = acc;
        }
    return;
  }
  const int j = (int)(t & 1), n32 = (int)((t >> 1) & 31), h = (int)((t >> 6) & 1);
  const size_t rest = layout ? t >> 8 : t >> 7;
  const int c4 = (int)(rest % (size_t)(K / 4)), nchunk = (int)(rest / (size_t)(K / 4));
  int k = c4 * 4 + h * 2 + j, n = nchunk * 32 + n32;
  if (layout) {
    const int nbk = (int)(t & 3), n16 = (int)((t >> 2) & 15), kk = (int)((t >> 6) & 3);
    k = c4 * 4 + kk; n = nchunk * 64 + nbk * 16 + n16;
  }
  float g[27];
#pragma unroll
  for (int tap = 0; tap < 27; ++tap)
    g[tap] = bwd ? w[((size_t)(26 - tap) * Cin_total + c_off + n) * Cout + k]
                 : w[((size_t)tap * Cin_total + c_off + k) * Cout + n];
  float gx[9][4];                                  // [(a, b)][fx]
#pragma unroll
  for (int ab = 0; ab < 9; ++ab) {
    const float g0 = g[ab * 3], g1 = g[ab * 3 + 1], g2 = g[ab * 3 + 2];
    gx[ab][0] = g0; gx[ab][1] = 0.5f * (g0 + g1 + g2); gx[ab][2] = 0.5f * (g0 - g1 + g2); gx[ab][3] = g2;
  }
  float gy[3][4][4];                               // [a][fy][fx]
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int fx = 0; fx < 4; ++fx) {
      const float g0 = gx[a * 3][fx], g1 = gx[a * 3 + 1][fx], g2 = gx[a * 3 + 2][fx];
      gy[a][0][fx] = g0; gy[a][1][fx] = 0.5f * (g0 + g1 + g2); gy[a][2][fx] = 0.5f * (g0 - g1 + g2); gy[a][3][fx] = g2;
    }
  const int fs = layout ? 256 : 128;                                        // floats per frequency image
  float* d = layout ? dst + (size_t)rest * 64 * 256 + (t & 255)
                    : dst + ((size_t)rest * 64 * 2 + h) * 64 + n32 * 2 + j;    // + f * fs
#pragma unroll
  for (int fy = 0; fy < 4; ++fy)
#pragma unroll
    for (int fx = 0; fx < 4; ++fx) {
      const float g0 = gy[0][fy][fx], g1 = gy[1][fy][fx], g2 = gy[2][fy][fx];
      d[(0 * 16 + fy * 4 + fx) * fs] = g0;
      d[(1 * 16 + fy * 4 + fx) * fs] = 0.5f * (g0 + g1 + g2);
      d[(2 * 16 + fy * 4 + fx) * fs] = 0.5f * (g0 - g1 + g2);
      d[(3 * 16 + fy * 4 + fx) * fs] = g2;
    }
}
__global__ void pack_wino_kernel(const float* __restrict__ w, int Cin_total, int Cout, int c_off, int Csub, int bwd,
                                 float* __restrict__ dst, size_t pairs, int layout) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= pairs) return;
  pack_wino_pair(t, w, Cin_total, Cout, c_off, Csub, bwd, dst, layout);
}
// The FULL predicate of the 64-channel kernel, size bound included: a layer whose geometry at the maximum batch fails
// it keeps layout 0 and the 32 x 32 kernel at EVERY batch size (the caller stores the answer and hands it to
// launch_conv_fwd_wino; deciding it again from the launch geometry once read layout-1 weights with the layout-0 kernel
// in the window 2^29 <= B*S^3*C < 2^31).
int conv_wino_layout(const ConvGeom& g) {
  const ConvSrc s{nullptr, nullptr, nullptr, g.Cin, 0, ACT_NONE, 0};
  return conv_wino64_ok(g, &s, 1) ? 1 : 0;
}
int launch_pack_wino(hipStream_t st, const float* w, int Cin_total, int Cout, int c_off, int Csub, int bwd, float* dst,
                     int layout) {
  const size_t pairs = (size_t)Csub * Cout;
  if (g_pack_rec) { g_pack_rec->push_back(PackJob{5, w, dst, {Cin_total, Cout, c_off, Csub, bwd, layout, 0, 0, 0}, (unsigned long long)pairs, 0}); return 0; }
  ICS_LAUNCH(pack_wino_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, w, Cin_total, Cout,
                     c_off, Csub, bwd, dst, pairs, layout);
  ICS_HIP(hipGetLastError());
  return 0;
}

// Weights of the 27-product upsampled-input kernel (conv_up3.hip): dst[Cout/64][Cu/4][27 f][4 k][16 n][4 column blocks] =
// (g (x) g (x) g) applied to w[tap][c_off + k][n] along z, y, x with g = [1 0 0; 1 1 1; 0 0 1].
// Thread t = (nchunk * Cu/4 + c4) * 256 + (kk * 16 + n16) * 4 + nb: one (k, n) pair.
__device__ __forceinline__ void pack_up3_pair(size_t t, const float* __restrict__ w, int Cin_total, int Cout, int c_off,
                                              int Cu, float* __restrict__ dst) {
  const size_t rest = t >> 8;
  const int c4 = (int)(rest % (size_t)(Cu / 4)), nchunk = (int)(rest / (size_t)(Cu / 4));
  const int nbk = (int)(t & 3), n16 = (int)((t >> 2) & 15), kk = (int)((t >> 6) & 3);
  const int k = c4 * 4 + kk, n = nchunk * 64 + nbk * 16 + n16;
  float g[27];
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) g[tap] = w[((size_t)tap * Cin_total + c_off + k) * Cout + n];
  float gx[9][3];
#pragma unroll
  for (int ab = 0; ab < 9; ++ab) {
    gx[ab][0] = g[ab * 3]; gx[ab][1] = g[ab * 3] + g[ab * 3 + 1] + g[ab * 3 + 2]; gx[ab][2] = g[ab * 3 + 2];
  }
  float gy[3][3][3];                               // [a][fy][fx]
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int fx = 0; fx < 3; ++fx) {
      gy[a][0][fx] = gx[a * 3][fx]; gy[a][1][fx] = gx[a * 3][fx] + gx[a * 3 + 1][fx] + gx[a * 3 + 2][fx];
      gy[a][2][fx] = gx[a * 3 + 2][fx];
    }
  float* d = dst + (size_t)rest * 27 * 256 + (t & 255);
#pragma unroll
  for (int fy = 0; fy < 3; ++fy)
#pragma unroll
    for (int fx = 0; fx < 3; ++fx) {
      const float sgn = fy == 1 ? -1.f : 1.f;      // conv_up3.hip builds -r1 for the fy = 1 frequencies (one fma per column)
      d[(0 * 9 + fy * 3 + fx) * 256] = sgn * gy[0][fy][fx];
      d[(1 * 9 + fy * 3 + fx) * 256] = sgn * (gy[0][fy][fx] + gy[1][fy][fx] + gy[2][fy][fx]);
      d[(2 * 9 + fy * 3 + fx) * 256] = sgn * gy[2][fy][fx];
    }
}
__global__ void pack_up3_kernel(const float* __restrict__ w, int Cin_total, int Cout, int c_off, int Cu,
                                float* __restrict__ dst, size_t pairs) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < pairs) pack_up3_pair(t, w, Cin_total, Cout, c_off, Cu, dst);
}
// weights: dst[f][Cu/4][4 k][Cout] = (g (x) g (x) g) applied to w[tap][c_off + k][n]; one thread per (k, n) pair
__device__ __forceinline__ void pack_up3n_pair_dev(size_t t, const float* __restrict__ w, int Cin_total, int Cout, int c_off,
                                                   int Cu, float* __restrict__ dst) {
  const int n = (int)(t % (size_t)Cout), k = (int)(t / (size_t)Cout);
  float g[3][3][3];
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) g[tap / 9][(tap / 3) % 3][tap % 3] = w[((size_t)tap * Cin_total + c_off + k) * Cout + n];
  // g rows: (w[-1], w[-1] + w[0] + w[+1], w[+1]) along x, y, z
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) g[i][j][1] += g[i][j][0] + g[i][j][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) g[i][1][kx] += g[i][0][kx] + g[i][2][kx];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) g[1][j][kx] += g[0][j][kx] + g[2][j][kx];
  const size_t fs = (size_t)Cu * Cout;
#pragma unroll
  for (int f = 0; f < 27; ++f) dst[(size_t)f * fs + (size_t)k * Cout + n] = g[f / 9][(f / 3) % 3][f % 3];
}
__global__ void pack_up3n_kernel(const float* __restrict__ w, int Cin_total, int Cout, int c_off, int Cu,
                                 float* __restrict__ dst, size_t pairs) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < pairs) pack_up3n_pair_dev(t, w, Cin_total, Cout, c_off, Cu, dst);
}
int launch_pack_up3n(hipStream_t st, const float* w, int Cin_total, int Cout, int c_off, int Cu, float* dst) {
  const size_t pairs = (size_t)Cu * Cout;
  if (g_pack_rec) { g_pack_rec->push_back(PackJob{7, w, dst, {Cin_total, Cout, c_off, Cu, 0, 0, 0, 0, 0}, (unsigned long long)pairs, 0}); return 0; }
  ICS_LAUNCH(pack_up3n_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, w, Cin_total, Cout, c_off, Cu, dst, pairs);
  ICS_HIP(hipGetLastError());
  return 0;
}
size_t conv_up3_weight_floats(int Cu, int Cout) { return (size_t)27 * Cu * Cout; }
int launch_pack_up3(hipStream_t st, const float* w, int Cin_total, int Cout, int c_off, int Cu, float* dst) {
  const size_t pairs = (size_t)Cu * Cout;
  if (g_pack_rec) { g_pack_rec->push_back(PackJob{6, w, dst, {Cin_total, Cout, c_off, Cu, 0, 0, 0, 0, 0}, (unsigned long long)pairs, 0}); return 0; }
  ICS_LAUNCH(pack_up3_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, w, Cin_total, Cout, c_off,
                     Cu, dst, pairs);
  ICS_HIP(hipGetLastError());
  return 0;
}

// One launch for a whole table of pack jobs.  Destination buffers are zero-initialised at allocation and the
// padding of a packed image never changes, so jobs only write their valid elements (two jobs may share a
// destination: the head packs soft | sig side by side) -- no ordering between jobs is needed.
__global__ __launch_bounds__(256) void pack_table_kernel(const PackJob* __restrict__ jobs, int njobs) {
  __shared__ int sj;
  if (threadIdx.x == 0) {
    int lo = 0, hi = njobs - 1;              // last job whose first block is <= blockIdx.x
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].blk0 <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    sj = lo;
  }
  __syncthreads();
  const PackJob J = jobs[sj];
  const int* a = J.a;
  size_t i = (size_t)(blockIdx.x - J.blk0) * 256 + threadIdx.x;
  if (a[8] > 0) {
    // Winograd / 27-product jobs: blocks are dispatched round-robin over the 8 XCDs, and in the backward images the eight
    // launch-order neighbours (k quads c4 .. c4 + 7 of the same n rows) read different 16-byte pieces of the SAME 128-byte
    // lines of w -- through eight different L2s.  Give every XCD a contiguous range of the job's groups instead (blk0 and
    // the block count a[8] are multiples of 8): the neighbours in k then follow each other on one XCD
    const unsigned bj = blockIdx.x - J.blk0, per = (unsigned)a[8] >> 3;
    i = ((size_t)(bj & 7u) * per + (bj >> 3)) * 256 + threadIdx.x;
  }
  if (i >= J.total) return;
  float v;
  switch (J.kind) {
    case 0: if (pack_fwd_value(i, J.w, a[0], a[1], a[2], a[3], a[4], a[5], a[6], &v)) J.dst[i] = v; break;
    case 1: if (pack_bwd_value(i, J.w, a[0], a[1], a[2], a[3], a[4], a[5], &v)) J.dst[i] = v; break;
    case 2: J.dst[i] = pack_sub_value(i, J.w, a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
    case 3: J.dst[i] = pack_fwd_sub_value(i, J.w, a[0], a[1], a[2], a[3], a[4], a[5]); break;
    case 5: pack_wino_pair(i, J.w, a[0], a[1], a[2], a[3], a[4], J.dst, a[5]); break;
    case 6: pack_up3_pair(i, J.w, a[0], a[1], a[2], a[3], J.dst); break;
    case 7: pack_up3n_pair_dev(i, J.w, a[0], a[1], a[2], a[3], J.dst); break;
    default: J.dst[i] = pack_par_value(i, J.w, a[0], a[1], a[2], a[3], a[4], a[5]); break;
  }
}

// record the launch_pack_* calls made by `fn` instead of executing them
struct PackTableHost { std::vector<PackJob> jobs; };
void* pack_table_record_begin() {
  auto* t = new PackTableHost();
  g_pack_rec = &t->jobs;
  return t;
}
// finishes the recording: assigns block ranges, returns the table bytes to upload (caller copies them to the
// device) and the total block count
int pack_table_record_end(void* handle, std::vector<unsigned char>* bytes, int* njobs, unsigned* nblocks) {
  g_pack_rec = nullptr;
  auto* t = static_cast<PackTableHost*>(handle);
  unsigned blk = 0;
  for (auto& j : t->jobs) {
    const bool xcd = (j.kind == 5 && j.a[5] == 1) || j.kind == 6;     // see pack_table_kernel
    if (xcd) blk = (blk + 7u) & ~7u;
    j.blk0 = blk;
    const unsigned nb = (unsigned)((j.total + 255) / 256);
    j.a[8] = xcd ? (int)((nb + 7u) & ~7u) : 0;
    blk += xcd ? (unsigned)j.a[8] : nb;
  }
  bytes->resize(t->jobs.size() * sizeof(PackJob));
  if (!t->jobs.empty()) std::memcpy(bytes->data(), t->jobs.data(), bytes->size());
  *njobs = (int)t->jobs.size();
  *nblocks = blk;
  delete t;
  return 0;
}
int launch_pack_table(hipStream_t st, const void* d_jobs, int njobs, unsigned nblocks) {
  if (njobs == 0) return 0;
  ICS_LAUNCH(pack_table_kernel, dim3(nblocks), dim3(256), 0, st, static_cast<const PackJob*>(d_jobs), njobs);
  ICS_HIP(hipGetLastError());
  return 0;
}

}  // namespace ics
