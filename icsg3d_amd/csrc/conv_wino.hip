// Winograd F(2x2x2, 3x3x3) convolution on the fp32 matrix cores (gfx950 / MI355X only).
//
// Same operator as conv_igemm.hip's 27-tap implicit GEMM -- Keras Conv3D(3x3x3, padding="same") and its backward-data
// (/root/reference/unet/unet.py:283-336, vae/lattice_vae.py:170-222) -- evaluated with 64 multiplies per 2x2x2 output
// tile instead of 216.  Y = A^T [ (G g G^T) .* (B^T d B) ] A per axis with the standard F(2,3) matrices
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
// applied along z, y and x.  The products over input channels are 64 independent GEMMs (one per frequency
// f = (fz, fy, fx)), tiles x Cin x Cout, which is where the MFMAs go.
//
// One workgroup (8 waves, two per SIMD) = 32 tiles (2x4x4 tiles = 4x8x8 voxels) x 32 output channels x all 64
// frequencies: wave w owns the 8 frequencies (fz = w >> 1, fy in {2 (w & 1), 2 (w & 1) + 1}, fx = 0..3), one 32x32 fp32
// accumulator each (v_mfma_f32_32x32x2_f32).  K loop over input channels in chunks of 16: the halo block
// [6][10][10] voxels x 16 channels is read once per chunk by 400 threads, each owning one (y, x, channel quad) column:
// BatchNorm affine of the producer, zero padding after it, and the z rows of B^T are applied THERE, once per voxel, and
// the eight z-combined planes (tile z, fz) go to LDS (double buffered).  Each wave then reads three rows of its plane
// per column and finishes the y / x transforms in registers, directly in the MFMA A-operand layout (lane = tile,
// half-wave = channel pair): 12 ds_read_b64 + 32 VALU per 16 MFMAs.  B operands (the transformed weights, pre-arranged
// so that one sub-step of one wave is 4 KB contiguous) stream from L2 one sub-step ahead.  The transform of sub-step
// g+1 is interleaved, instruction by instruction, with the MFMAs of sub-step g.  On this chip the fp32 MFMA shares the
// SIMD's VALU issue (SQ_VALU_MFMA_COEXEC_CYCLES = 0; scripts/probes/mfma_filler.hip: a v_fma_f32 between MFMAs costs
// 5.2 cycles with one wave per SIMD, 3.6 with two, and the bare MFMA issues every 75 / 69.5 cycles): every VALU
// instruction is MFMA time lost, hence two waves per SIMD, scalar-base addressing, immediates and no packed math in
// the main loop.  LDS layout: voxel pitch 18 floats, row pitch 10 voxels, plane pitch 100 voxels, the two channel pairs
// of each 4-channel group swapped where (hy >> 1) is odd: the 32 tiles of a half-wave hit 32 distinct bank pairs.
// Epilogue (two passes of 8 accumulator registers): fx -> dx and the wave's share of fy -> dy in registers, the sum
// over the eight waves (fz -> dz) through LDS, then bias / accumulate / activation / store as float4 and the per-block
// BatchNorm partials (count, mean, M2) of conv_igemm.hip's layout.
#include "common.h"

#include <type_traits>

#include <algorithm>

#ifdef ICS_WG_TIMELINE
// variant builds only (scripts/variants.sh conv_wino "wgtl:-DICS_WG_TIMELINE"): per workgroup of conv_wino_wgrad_kernel
// wall-clock stamps {0 entry, 1 first block staged + first operands built (main loop starts), 2..9 main loop done on wave
// 0..7, 10 after the epilogue's last store, 11 blocks in this split} + HW_ID / XCC_ID, fetched with
// ics_debug_wgrad_timeline (scripts/wgrad_timeline.py).  The stamps are scalar stores from lane 0 of a wave.
__device__ unsigned long long ics_wg_tl[16 * 8192];
extern "C" int ics_debug_wgrad_timeline(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ics_wg_tl), (size_t)n * 8);
}
#define ICS_WGTL(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 8192) ics_wg_tl[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#define ICS_WGTL0(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) ics_wg_tl[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define ICS_WGTL(i)
#define ICS_WGTL0(i)
#endif

namespace ics {

typedef float wf2 __attribute__((ext_vector_type(2)));
typedef float wf4 __attribute__((ext_vector_type(4)));
typedef float wf16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int KC = 16;                                   // input channels per LDS chunk
constexpr int HZ = 6, HY = 10, HX = 10, NV = HZ * HY * HX;
constexpr int P3 = 18, PY3 = 10, PZ3 = 100, BUF3 = 8 * PZ3 * P3;       // floats per buffer (57 600 B): 8 z-combined planes
constexpr int kRowsPerBlock = 256;                       // voxels per workgroup

__device__ __forceinline__ float wact(float v, float slope) { return fmaxf(v, v * slope); }
__host__ __device__ __forceinline__ float wslope(int act) { return act == ACT_RELU ? 0.f : (act == ACT_LRELU ? kLeaky : 1.f); }
}  // namespace

// wt layout: [Cout/32][Cin/4][64 f][2 h][32 n][2 j]   (input channel = c4*4 + h*2 + j)
// AFF: the source carries a per-channel affine (+ activation of slope in_slope); NOACT: affine only.
// FOLD (backward-data launches): the tile just produced is dO of the producer layer P; the block also adds its share of
// P's BatchNorm-backward sums (BwdStat, common.h) -- the pass bn_bwd_reduce_kernel would otherwise make over dO and s.
// FOLD = 2: the producer's whole BatchNorm-backward apply in the epilogue (see conv_wino64.hip; BwdStat::abc / db_partial)
template <bool AFF, bool NOACT, int FOLD>
__global__ __launch_bounds__(512) void conv_wino_kernel(const float* __restrict__ x, int ldx,
                                                        const float* __restrict__ in_scale,
                                                        const float* __restrict__ in_shift, float in_slope,
                                                        const float* __restrict__ wt, const float* __restrict__ bias,
                                                        float* __restrict__ y, int ldo, float pre_slope, int accumulate,
                                                        float* __restrict__ stat_partial, int Npad, int S, int Cin,
                                                        int Cout, BwdStat bs) {
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF3];   // 115 200 B; the epilogue reuses it
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fz = w >> 1, fyh = w & 1;            // this wave: frequencies (fz, 2 fyh + {0,1}, 0..3)
  const int m = lane & 31, h = lane >> 5;
  const int nchunks = Cout >> 5;
  const int nb = blockIdx.x % nchunks;           // the n-chunks of one tile block are neighbours in launch order:
  const int tblk = blockIdx.x / nchunks;         // with Cout/32 <= 8 each XCD keeps ONE n-chunk's weights hot in its L2
  int tb = tblk;
  const int nbx = S >> 3, nby = S >> 3, nbz = S >> 2;
  const int bx = tb % nbx; tb /= nbx;
  const int by = tb % nby; tb /= nby;
  const int bz = tb % nbz;
  const int b = tb / nbz;
  const int oz = bz * 4, oy = by * 8, ox = bx * 8, n0 = nb * 32;
  const bool edge = bx == 0 || by == 0 || bz == 0 || bx == nbx - 1 || by == nby - 1 || bz == nbz - 1;   // uniform

  // ---- staging: thread t < 400 owns one (hy, hx, channel quad) column of the halo block: six raw z values in,
  // the eight z-combined planes (tz, fz) out -- the z rows of B^T are applied ONCE per voxel here instead of once per
  // overlapping tile in the waves
  const int cmb = tid < 400 ? tid : 399;
  const int q = cmb & 3, hx = (cmb >> 2) % HX, hy = (cmb >> 2) / HX;
  wf4 stage[6];
  unsigned soff[6];                              // unsigned: scalar base + 32-bit lane offset addressing
  unsigned okmask = 0;
  {
    const int gy = oy - 1 + hy, gx = ox - 1 + hx;
    const bool okyx = gy >= 0 && gy < S && gx >= 0 && gx < S;
    const int cy = min(max(gy, 0), S - 1), cx = min(max(gx, 0), S - 1);
#pragma unroll
    for (int hz = 0; hz < 6; ++hz) {
      const int gz = oz - 1 + hz;
      okmask |= (okyx && gz >= 0 && gz < S) ? (1u << hz) : 0u;
      const int cz = min(max(gz, 0), S - 1);
      soff[hz] = (unsigned)((((b * S + cz) * S + cy) * S + cx) * ldx + q * 4);
    }
  }
  const int sw = ((hy >> 1) & 1) * 2;
  const int dbase = (hy * PY3 + hx) * P3 + q * 4;            // + plane * PZ3 * P3
  wf4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  auto gload = [&](int c0) {
    const float* xc = x + c0;                  // uniform base
#pragma unroll
    for (int i = 0; i < 6; ++i) stage[i] = *reinterpret_cast<const wf4*>(xc + soff[i]);
    if (AFF) {
      sc4 = *reinterpret_cast<const wf4*>(in_scale + c0 + q * 4);
      sh4 = *reinterpret_cast<const wf4*>(in_shift + c0 + q * 4);
    }
  };
  auto sstore = [&](const int bo) {            // bo: a compile-time constant after unrolling
    if (AFF) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        wf4 t = stage[i];
        t.x = fmaf(t.x, sc4.x, sh4.x); t.y = fmaf(t.y, sc4.y, sh4.y);
        t.z = fmaf(t.z, sc4.z, sh4.z); t.w = fmaf(t.w, sc4.w, sh4.w);
        if (!NOACT) { t.x = wact(t.x, in_slope); t.y = wact(t.y, in_slope); t.z = wact(t.z, in_slope); t.w = wact(t.w, in_slope); }
        stage[i] = t;
      }
    }
    if (edge) {                                // "same" padding: zeros AFTER the producer's affine / activation
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if (!((okmask >> i) & 1)) stage[i] = wf4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < 400) {
#pragma unroll
      for (int tz = 0; tz < 2; ++tz) {
        const wf4 d0 = stage[2 * tz], d1 = stage[2 * tz + 1], d2 = stage[2 * tz + 2], d3 = stage[2 * tz + 3];
        const wf4 c[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int o = bo + (tz * 4 + f) * (PZ3 * P3) + dbase;
          *reinterpret_cast<wf2*>(&lds[o + sw]) = wf2{c[f].x, c[f].y};
          *reinterpret_cast<wf2*>(&lds[o + 2 - sw]) = wf2{c[f].z, c[f].w};
        }
      }
    }
  };

  // ---- per-lane read geometry.  tile m = (tz, ty, tx).  The wave's two fy rows of B^T need rows (a, b, c) of the
  // combined plane:  fy = 2 fyh:  R_a - R_b,   fy = 2 fyh + 1:  R_b + sg R_c   with (a,b,c,sg) = (0,2,1,+) / (2,1,3,-)
  const int tz = m >> 4, ty = (m >> 2) & 3, tx = m & 3;
  const float sg = fyh ? -1.f : 1.f;
  auto rowbase = [&](int iy) {
    const int hyy = 2 * ty + iy;
    return ((tz * 4 + fz) * PZ3 + hyy * PY3 + 2 * tx) * P3 + 2 * (h ^ ((hyy >> 1) & 1));
  };
  const int Ra = rowbase(fyh ? 2 : 0), Rb = rowbase(fyh ? 1 : 2), Rc = rowbase(fyh ? 3 : 1);

  const int nsub = Cin >> 2;
  constexpr int wstride_f = 128;
  constexpr int wsub = 64 * 128;
  const float* wu = wt + ((size_t)nb * nsub * 64 + fz * 16 + fyh * 8) * 128;      // uniform
  const int wlane = h * 64 + m * 2;
  wf2 wreg[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) wreg[f] = *reinterpret_cast<const wf2*>(wu + f * wstride_f + wlane);

  wf16 acc[8];                                   // frequency (fy local, fx) = acc[fyl * 4 + fx]
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  wf2 u[8], tn[2][4], qa, qb, qc;
  auto rd = [&](const int bo, const int sub, const int col) {
    const int off = col * P3 + 4 * sub + bo;
    qa = *reinterpret_cast<const wf2*>(&lds[Ra + off]);
    qb = *reinterpret_cast<const wf2*>(&lds[Rb + off]);
    qc = *reinterpret_cast<const wf2*>(&lds[Rc + off]);
  };
  auto xform = [&]() {
#pragma unroll
    for (int fy = 0; fy < 2; ++fy) {
      u[fy * 4 + 0] = tn[fy][0] - tn[fy][2];
      u[fy * 4 + 1] = tn[fy][1] + tn[fy][2];
      u[fy * 4 + 2] = tn[fy][2] - tn[fy][1];
      u[fy * 4 + 3] = tn[fy][1] - tn[fy][3];
    }
  };

  gload(0);
  sstore(0);
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    rd(0, 0, g);
    tn[0][g] = qa - qb;
    tn[1][g] = qb + sg * qc;
  }
  xform();
  rd(0, 1, 0);                                   // column 0 of sub-step 1

  const int nch = Cin / KC;
  // ST = false: the last chunk stages nothing (its read-ahead columns come from a buffer that is never consumed)
  auto chunk = [&](const int ch, const int cur, const int nxt, auto stage_tag) {
    constexpr bool ST = decltype(stage_tag)::value;
    if (ST) gload((ch + 1 < nch ? ch + 1 : ch) * KC);   // not peeled: past the end the last chunk again (never consumed)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      int gs = ch * 4 + s + 1;
      gs = gs < nsub ? gs : nsub - 1;
      const float* wn = wu + (size_t)gs * wsub;  // uniform
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const wf2 a = qa, bq = qb, c = qc;
        if (ST && s == 2 && g == 3) {            // the next chunk must be visible before its first column is read
          sstore(nxt);
          __syncthreads();
        }
        // reads of the next column: column g+1 of sub-step gs, or column 0 of sub-step gs+1
        if (g < 3) rd(s == 3 ? nxt : cur, (s + 1) & 3, g + 1);
        else rd(s >= 2 ? nxt : cur, (s + 2) & 3, 0);
        __builtin_amdgcn_sched_barrier(0);
#define ICS_WMF(F, C) acc[F] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[F].C, wreg[F].C, acc[F], 0, 0, 0)
#define ICS_WFN __builtin_amdgcn_sched_barrier(0)
        // consecutive MFMAs on different accumulators; the column math of the next sub-step in between
        ICS_WMF(g, x); tn[0][g] = a - bq; ICS_WFN;
        ICS_WMF(4 + g, x); tn[1][g] = bq + sg * c; ICS_WFN;
        ICS_WMF(g, y); ICS_WFN;
        ICS_WMF(4 + g, y); ICS_WFN;
#undef ICS_WMF
        wreg[g] = *reinterpret_cast<const wf2*>(wn + g * wstride_f + wlane);
        wreg[4 + g] = *reinterpret_cast<const wf2*>(wn + (4 + g) * wstride_f + wlane);
        ICS_WFN;
#undef ICS_WFN
      }
      xform();
    }
  };
  // Cin % 32 == 0: an even number of chunks (a conditional second chunk makes the register allocator spill the
  // accumulators).  Without a BatchNorm-affine source the last chunk is a separate copy that stages nothing (-2.6 %;
  // with it the extra copy costs 25 spilled registers and 5 %: measured, not used)
  if (!AFF) {
    for (int ch = 0; ch < nch - 2; ch += 2) {
      chunk(ch, 0, BUF3, std::true_type{});
      chunk(ch + 1, BUF3, 0, std::true_type{});
    }
    chunk(nch - 2, 0, BUF3, std::true_type{});
    chunk(nch - 1, BUF3, 0, std::false_type{});
  } else {
    for (int ch = 0; ch < nch; ch += 2) {
      chunk(ch, 0, BUF3, std::true_type{});
      chunk(ch + 1, BUF3, 0, std::true_type{});
    }
  }

  // ---------------------------------------------------------------- epilogue, two passes of 8 accumulator registers
  float* part = lds;                             // [8 w][32 slots = rr*4 + dy*2 + dx][64 lanes]  (64 KB)
  float* red = lds + 16384;                      // [8 w][32] + [32]
  const int k = tid & 7;                         // this thread's output-channel quad in the final stage
  wf4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) bv = *reinterpret_cast<const wf4*>(bias + n0 + 4 * k);
  wf4 val[4];
  wf4 csum = {0.f, 0.f, 0.f, 0.f};
  wf4 f1 = {0.f, 0.f, 0.f, 0.f}, f2s = {0.f, 0.f, 0.f, 0.f};    // FOLD: sum d, sum d * xhat of this thread's 4 columns
  wf4 b_mu = f1, b_rs = f1, b_sc = {1.f, 1.f, 1.f, 1.f}, b_sh = f1;
  if (FOLD == 2) {                                 // b_mu / b_rs / b_sc hold the apply's a / b / c
    b_mu = *reinterpret_cast<const wf4*>(bs.abc + n0 + 4 * k);
    b_rs = *reinterpret_cast<const wf4*>(bs.abc + Cout + n0 + 4 * k);
    b_sc = *reinterpret_cast<const wf4*>(bs.abc + 2 * Cout + n0 + 4 * k);
  } else if (FOLD) {
    b_mu = *reinterpret_cast<const wf4*>(bs.mean + n0 + 4 * k);
    b_rs = *reinterpret_cast<const wf4*>(bs.rstd + n0 + 4 * k);
    if (bs.post_act != ACT_NONE) {
      b_sc = *reinterpret_cast<const wf4*>(bs.scale + n0 + 4 * k);
      b_sh = *reinterpret_cast<const wf4*>(bs.shift + n0 + 4 * k);
    }
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = pass * 8 + rr;
      float qv[2][2];                            // [fy local][dx]
#pragma unroll
      for (int fy = 0; fy < 2; ++fy) {
        qv[fy][0] = acc[fy * 4 + 0][r] + acc[fy * 4 + 1][r] + acc[fy * 4 + 2][r];
        qv[fy][1] = acc[fy * 4 + 1][r] - acc[fy * 4 + 2][r] - acc[fy * 4 + 3][r];
      }
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        // rows of A^T: the wave with fy 0,1 gives dy0 += q0 + q1, dy1 += q1; the one with fy 2,3: dy0 += q0, dy1 -= q0 + q1
        const float d0 = fyh ? qv[0][dx] : qv[0][dx] + qv[1][dx];
        const float d1 = fyh ? -qv[0][dx] - qv[1][dx] : qv[1][dx];
        part[(w * 32 + rr * 4 + 0 + dx) * 64 + lane] = d0;
        part[(w * 32 + rr * 4 + 2 + dx) * 64 + lane] = d1;
      }
    }
    __syncthreads();
    // one task per thread: t = tid >> 3: hh = t & 1, o = (t >> 1) & 3 (= dy*2 + dx), rr = t >> 3; tile = (r>>2)*8 + hh*4 + (r&3)
    const int t = tid >> 3;
    const int hh = t & 1, o = (t >> 1) & 3, rr = t >> 3;
    const int r = pass * 8 + rr;
    const int mt = (r >> 2) * 8 + hh * 4 + (r & 3);
    const int ttz = mt >> 4, tty = (mt >> 2) & 3, ttx = mt & 3;
    const int slot = (rr * 4 + o) * 64 + hh * 32 + 4 * k;
    wf4 p[4];
#pragma unroll
    for (int z = 0; z < 4; ++z)
      p[z] = *reinterpret_cast<const wf4*>(&part[(2 * z) * 2048 + slot]) +
             *reinterpret_cast<const wf4*>(&part[(2 * z + 1) * 2048 + slot]);
    const int vy = oy + 2 * tty + (o >> 1), vx = ox + 2 * ttx + (o & 1), vz = oz + 2 * ttz;
    const size_t o0 = ((((size_t)b * S + vz) * S + vy) * S + vx) * ldo + n0 + 4 * k;
    const size_t o1 = o0 + (size_t)S * S * ldo;
    wf4 e0 = p[0] + p[1] + p[2] + bv, e1 = p[1] - p[2] - p[3] + bv;
    if (accumulate) {
      e0 += *reinterpret_cast<const wf4*>(y + o0);
      e1 += *reinterpret_cast<const wf4*>(y + o1);
    }
    e0.x = wact(e0.x, pre_slope); e0.y = wact(e0.y, pre_slope); e0.z = wact(e0.z, pre_slope); e0.w = wact(e0.w, pre_slope);
    e1.x = wact(e1.x, pre_slope); e1.y = wact(e1.y, pre_slope); e1.z = wact(e1.z, pre_slope); e1.w = wact(e1.w, pre_slope);
    if (FOLD == 2) {
      const size_t s0 = ((((size_t)b * S + vz) * S + vy) * S + vx) * bs.ld + n0 + 4 * k;
      const wf4 sv0 = *reinterpret_cast<const wf4*>(bs.s + s0);
      const wf4 sv1 = *reinterpret_cast<const wf4*>(bs.s + s0 + (size_t)S * S * bs.ld);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        e0[r4] = sv0[r4] > 0.f ? fmaf(b_mu[r4], e0[r4], fmaf(b_rs[r4], sv0[r4], b_sc[r4])) : 0.f;
        e1[r4] = sv1[r4] > 0.f ? fmaf(b_mu[r4], e1[r4], fmaf(b_rs[r4], sv1[r4], b_sc[r4])) : 0.f;
      }
      f1 += e0 + e1;
    }
    *reinterpret_cast<wf4*>(y + o0) = e0;
    *reinterpret_cast<wf4*>(y + o1) = e1;
    val[2 * pass] = e0; val[2 * pass + 1] = e1;
    csum += e0 + e1;
    if (FOLD == 1) {
      const size_t s0 = ((((size_t)b * S + vz) * S + vy) * S + vx) * bs.ld + n0 + 4 * k;
      const wf4 sv0 = *reinterpret_cast<const wf4*>(bs.s + s0);
      const wf4 sv1 = *reinterpret_cast<const wf4*>(bs.s + s0 + (size_t)S * S * bs.ld);
      wf4 d0 = e0, d1 = e1;
      if (bs.post_act != ACT_NONE) {
        const wf4 z0 = sv0 * b_sc + b_sh, z1 = sv1 * b_sc + b_sh;
        d0.x *= act_grad(z0.x, bs.post_act); d0.y *= act_grad(z0.y, bs.post_act);
        d0.z *= act_grad(z0.z, bs.post_act); d0.w *= act_grad(z0.w, bs.post_act);
        d1.x *= act_grad(z1.x, bs.post_act); d1.y *= act_grad(z1.y, bs.post_act);
        d1.z *= act_grad(z1.z, bs.post_act); d1.w *= act_grad(z1.w, bs.post_act);
      }
      f1 += d0 + d1;
      f2s += d0 * ((sv0 - b_mu) * b_rs) + d1 * ((sv1 - b_mu) * b_rs);
    }
  }
  auto colreduce = [&](wf4 v) -> wf4 {             // columns 4k..4k+3 live in the lanes with the same (tid & 7)
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) {
      v.x += __shfl_xor(v.x, d); v.y += __shfl_xor(v.y, d); v.z += __shfl_xor(v.z, d); v.w += __shfl_xor(v.w, d);
    }
    return v;
  };
  auto sum8 = [&](int i) {
    float sacc = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) sacc += red[ww * 32 + i];
    return sacc;
  };
  if (FOLD == 2) {                                 // column sums of the written dy_P: [blocks][Cout]
    f1 = colreduce(f1);
    __syncthreads();
    if (lane < 8) *reinterpret_cast<wf4*>(&red[w * 32 + 4 * lane]) = f1;
    __syncthreads();
    if (tid < 32) bs.db_partial[(size_t)tblk * Cout + n0 + tid] = sum8(tid);
    return;
  }
  if (FOLD) {                                      // [2][Npad][blocks] (block index fastest), as conv_igemm.hip's FOLD
    f1 = colreduce(f1); f2s = colreduce(f2s);
    __syncthreads();
    if (lane < 8) {
      *reinterpret_cast<wf4*>(&red[w * 32 + 4 * lane]) = f1;
      *reinterpret_cast<wf4*>(&red[512 + w * 32 + 4 * lane]) = f2s;
    }
    __syncthreads();
    if (tid < 32) {
      const size_t nstat = gridDim.x / nchunks;
      float a2 = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) a2 += red[512 + ww * 32 + tid];
      float* sp = bs.partial + (size_t)(n0 + tid) * nstat + tblk;
      sp[0] = sum8(tid);
      sp[(size_t)Npad * nstat] = a2;
    }
    return;
  }
  if (stat_partial == nullptr) return;

  // block-level (count, mean, M2) per column, two passes inside the block (conv_igemm.hip's layout
  // [3][Npad][nblocks], block index fastest)
  csum = colreduce(csum);
  __syncthreads();
  if (lane < 8) *reinterpret_cast<wf4*>(&red[w * 32 + 4 * lane]) = csum;
  __syncthreads();
  if (tid < 32) red[256 + tid] = sum8(tid) * (1.f / kRowsPerBlock);
  __syncthreads();
  const wf4 mu = *reinterpret_cast<const wf4*>(&red[256 + 4 * k]);
  wf4 qs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const wf4 d = val[i] - mu;
    qs += d * d;
  }
  qs = colreduce(qs);
  __syncthreads();
  if (lane < 8) *reinterpret_cast<wf4*>(&red[w * 32 + 4 * lane]) = qs;
  __syncthreads();
  if (tid < 32) {
    const size_t nstat = gridDim.x / nchunks;
    float* sp = stat_partial + (size_t)(n0 + tid) * nstat + tblk;
    sp[0] = (float)kRowsPerBlock;
    sp[(size_t)Npad * nstat] = red[256 + tid];
    sp[(size_t)2 * Npad * nstat] = sum8(tid);
  }
}

// ================================================================================================================
// Backward-weight in the Winograd domain.  From Y = A^T[(G g) .* (B^T d)]:
//   dL/dg = G^T [ sum over tiles (A dY) .* (B^T d) ]     (per axis; dY = the 2x2x2 gradient tile, A = [1 0; 1 1; 1 -1; 0 -1])
// i.e. 64 independent GEMMs  dW^_f[ci][co] = sum_t U_f[t][ci] V_f[t][co]  reduced over ALL tiles, then the 4 -> 3
// contraction with G per axis.  One workgroup = 32 input channels x 32 output channels x 64 frequencies over a range
// of tile blocks (split-K over the tiles); 8 waves, wave w owns (fz = w >> 1, fy in {2 (w&1), 2 (w&1) + 1}, fx = 0..3),
// one 32x32 accumulator each.  A block = 2x2x4 tiles (4x4x8 voxels): the x halo [6][6][10] x 32 ci is read by 480
// threads, each owning one (y, x, channel quad) column and writing the eight z-combined planes (tile z, fz) (BatchNorm
// affine of the producer and zero padding applied first), dy [4][4][8] x 32 co is staged as is; both CHANNEL-major in
// LDS (pitch 482 / 130 floats per channel), double buffered.  MFMA k = 2 tiles: lanes 0-31 tile 2p, lanes 32-63 tile
// 2p+1, lane & 31 = channel -- every lane transforms its own (tile, channel) pair from conflict-free ds_read_b64
// (two x-neighbours per read) at immediate offsets of one base register.  The transform of k-step p+1 is interleaved
// with the 8 MFMAs of step p.  V is built without the negations of A's last row; the epilogue puts the signs back while
// contracting fx -> c and the wave's share of fy -> b in registers and the rest (fy halves, fz -> a) across the waves,
// and writes ws[split][27*Cin][Cout], the layout conv_igemm.hip's split reduction consumes.
namespace {
constexpr int GZ = 6, GY = 6, GX = 10;
constexpr int GXP = 482, GYP = 130;                                        // floats per channel: 8 planes x 60 (+2), 128 (+2)
constexpr int GXF = 32 * GXP, GYF = 32 * GYP;                              // floats: x part 61 696 B, dy part 16 640 B
// LDS = [x buffer 0][x buffer 1][dy buffer 0][dy buffer 1]: the second buffer of each part is < 64 KB from the first,
// so both are immediate offsets of ONE base register (a per-buffer base costs 4 more VGPRs, and spilled bases are
// reloaded through scratch behind the HBM-latency staging loads: 43 % of the wave cycles were such waits)
}  // namespace

template <bool AFF, bool NOACT>
__global__ __launch_bounds__(512) void conv_wino_wgrad_kernel(const float* __restrict__ x, int ldx,
                                                              const float* __restrict__ in_scale,
                                                              const float* __restrict__ in_shift, float in_slope,
                                                              const float* __restrict__ dy, int ldy,
                                                              float* __restrict__ ws, int S, int Cin, int Cout,
                                                              int per_split) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (GXF + GYF)];      // 156 672 B
  ICS_WGTL0(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fz = w >> 1, fyh = w & 1;
  const int c = lane & 31, hl = lane >> 5;
  const int nco = Cout >> 5, nci = Cin >> 5;
  // workgroups are dispatched round-robin over the 8 XCDs: give every XCD whole splits, so that the nci x nco
  // workgroups that read the same tiles (different channel chunks) share one L2
  int lin = blockIdx.x;
  const int npairs = nco * nci, nsplit = gridDim.x / npairs;
  if ((nsplit & 7) == 0) lin = ((blockIdx.x >> 3) / npairs * 8 + (blockIdx.x & 7)) * npairs + (blockIdx.x >> 3) % npairs;
  const int cob = lin % nco;
  const int cib = (lin / nco) % nci;
  const int split = lin / npairs;
  const int ci0 = cib * 32, co0 = cob * 32;
  const int blk_lo = split * per_split, blk_hi = blk_lo + per_split;
  const int nbx = S >> 3, nby = S >> 2, nbz = S >> 2;

  // ---- staging.  x: thread t < 480 owns (hy, hx, channel quad q): six raw z values -> eight combined planes.
  // thread -> (position, channel quad): a 32-lane store group holds FOUR quads x eight positions.  The channel pitch is
  // = 2 mod 32 floats, so quad q lands on bank 8 q + position: with eight quads per group (q = lane & 7, the first
  // version) q and q + 4 collided on every staging store -- 16 % of the kernel's LDS cycles were bank conflicts (PMC).
  const int sq = (tid & 3) | (((tid >> 5) & 1) << 2);                    // channel quad 0..7
  const int spos = (tid >> 6) * 8 + ((tid >> 2) & 7);                    // 0..63; the halo has 60 positions
  const bool sact = spos < GY * GX;
  const int cmb = sact ? spos : GY * GX - 1;
  const int q = sq, hx = cmb % GX, hy = cmb / GX;
  wf4 xs[6], ys[2];
  // Buffer loads (one 32-bit per-lane byte offset, everything block-dependent in the scalar offset operand): the halo
  // positions are CLAMPED into the grid per block -- a handful of VALU per block -- and zeroed at store time.  The
  // earlier form kept six per-lane offsets plus a "safe" one, selected per block through a scratch-resident table: two
  // scratch round trips ahead of every block's loads.
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, 0x7fffffff, 0x00020000);
  unsigned yrel[2];                                // bytes, relative to the block origin
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int v = spos + i * 64;                                           // 0..127 = (vz, vy, vx)
    yrel[i] = (unsigned)((((v >> 5) * S + ((v >> 3) & 3)) * S + (v & 7)) * ldy + co0 + sq * 4) * 4u;
  }
  const int xw = (q * 4) * GXP + hy * GX + hx;                             // LDS write base (floats): + j*GXP + plane*60
  bool okyx_cur = true;                            // this lane's (y, x) halo position lies inside the grid (current block)
  unsigned okz_cur = 0x3f;                         // uniform: bit hz set = plane oz - 1 + hz inside the grid
  auto gload = [&](int blk) {
    int t = blk;
    const int bx = t % nbx; t /= nbx;
    const int by = t % nby; t /= nby;
    const int bz = t % nbz;
    const int b = t / nbz;
    const int gy = 4 * by - 1 + hy, gx = 8 * bx - 1 + hx;
    const int cy = min(max(gy, 0), S - 1), cx = min(max(gx, 0), S - 1);
    okyx_cur = gy == cy && gx == cx;
    const unsigned voff = (unsigned)((cy * S + cx) * ldx + ci0 + q * 4) * 4u;
    unsigned okz = 0;
#pragma unroll
    for (int hz = 0; hz < 6; ++hz) {
      const int gz = 4 * bz - 1 + hz, cz = min(max(gz, 0), S - 1);
      okz |= (gz == cz) ? (1u << hz) : 0u;
      xs[hz] = __builtin_bit_cast(wf4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)voff, (int)((unsigned)((b * S + cz) * S * S * ldx) * 4u), 0));
    }
    okz_cur = okz;
    const unsigned org = (unsigned)((((b * S + 4 * bz) * S + 4 * by) * S + 8 * bx) * ldy) * 4u;   // block origin (uniform)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      ys[i] = __builtin_bit_cast(wf4, __builtin_amdgcn_raw_buffer_load_b128(yrs, (int)yrel[i], (int)org, 0));
  };
  auto sstore = [&](const int bo) {            // bo: buffer index 0 / 1
    if (AFF) {
      // reloaded per block (L1 hits): 8 registers less to keep live through the main loop
      const wf4 sc4 = *reinterpret_cast<const wf4*>(in_scale + ci0 + q * 4);
      const wf4 sh4 = *reinterpret_cast<const wf4*>(in_shift + ci0 + q * 4);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        wf4 t = xs[i];
        t.x = fmaf(t.x, sc4.x, sh4.x); t.y = fmaf(t.y, sc4.y, sh4.y);
        t.z = fmaf(t.z, sc4.z, sh4.z); t.w = fmaf(t.w, sc4.w, sh4.w);
        if (!NOACT) { t.x = wact(t.x, in_slope); t.y = wact(t.y, in_slope); t.z = wact(t.z, in_slope); t.w = wact(t.w, in_slope); }
        xs[i] = t;
      }
    }
    {                                                                      // padding zeros come AFTER the affine
      const unsigned okz = okz_cur;
#pragma unroll
      for (int hz = 0; hz < 6; ++hz)
        if (!(okyx_cur && ((okz >> hz) & 1))) xs[hz] = wf4{0.f, 0.f, 0.f, 0.f};
    }
    if (sact) {
#pragma unroll
      for (int tz = 0; tz < 2; ++tz) {
        const wf4 d0 = xs[2 * tz], d1 = xs[2 * tz + 1], d2 = xs[2 * tz + 2], d3 = xs[2 * tz + 3];
        const wf4 cz[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int o = bo * GXF + xw + (tz * 4 + f) * 60;
          lds[o] = cz[f].x; lds[o + GXP] = cz[f].y; lds[o + 2 * GXP] = cz[f].z; lds[o + 3 * GXP] = cz[f].w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = 2 * GXF + bo * GYF + (sq * 4) * GYP + spos + i * 64;
      lds[o] = ys[i].x; lds[o + GYP] = ys[i].y; lds[o + 2 * GYP] = ys[i].z; lds[o + 3 * GYP] = ys[i].w;
    }
  };

  // The same store in pieces, for the main loop: spread over the MFMA slots of k-steps 2..5 so that it runs under the
  // matrix pipe instead of in front of the block's barrier
  wf4 sc4v = {1.f, 1.f, 1.f, 1.f}, sh4v = {0.f, 0.f, 0.f, 0.f};
  auto ss_pre = [&]() {
    if (AFF) {
      sc4v = *reinterpret_cast<const wf4*>(in_scale + ci0 + q * 4);
      sh4v = *reinterpret_cast<const wf4*>(in_shift + ci0 + q * 4);
    }
  };
  auto ss_aff = [&](const int i) {
    wf4 t = xs[i];
    if (AFF) {
      t.x = fmaf(t.x, sc4v.x, sh4v.x); t.y = fmaf(t.y, sc4v.y, sh4v.y);
      t.z = fmaf(t.z, sc4v.z, sh4v.z); t.w = fmaf(t.w, sc4v.w, sh4v.w);
      if (!NOACT) { t.x = wact(t.x, in_slope); t.y = wact(t.y, in_slope); t.z = wact(t.z, in_slope); t.w = wact(t.w, in_slope); }
    }
    // zero padding as a bit mask (a select per component compiled to a branch per piece: 12 divergent regions per block)
    const unsigned msk = (okyx_cur ? ~0u : 0u) & (((okz_cur >> i) & 1) ? ~0u : 0u);
    t.x = __uint_as_float(__float_as_uint(t.x) & msk); t.y = __uint_as_float(__float_as_uint(t.y) & msk);
    t.z = __uint_as_float(__float_as_uint(t.z) & msk); t.w = __uint_as_float(__float_as_uint(t.w) & msk);
    xs[i] = t;
  };
  auto ss_x = [&](const int bo, const int tz, const int f) {
    // no `if (sact)`: the 32 threads past the halo's 60 positions were clamped to position 59 and hold the same values
    // as its owner -- a duplicate store of identical data instead of an exec-masked region per piece
    const wf4 d0 = xs[2 * tz], d1 = xs[2 * tz + 1], d2 = xs[2 * tz + 2], d3 = xs[2 * tz + 3];
    const wf4 czf = f == 0 ? d0 - d2 : (f == 1 ? d1 + d2 : (f == 2 ? d2 - d1 : d1 - d3));
    const int o = bo * GXF + xw + (tz * 4 + f) * 60;
    lds[o] = czf.x; lds[o + GXP] = czf.y; lds[o + 2 * GXP] = czf.z; lds[o + 3 * GXP] = czf.w;
  };
  auto ss_y = [&](const int bo, const int i) {
    const int o = 2 * GXF + bo * GYF + (sq * 4) * GYP + spos + i * 64;
    lds[o] = ys[i].x; lds[o + GYP] = ys[i].y; lds[o + 2 * GYP] = ys[i].z; lds[o + 3 * GYP] = ys[i].w;
  };
  // slot = the MFMA of k-step pp after which the piece is issued (slots 0..2 carry the y / dy transforms, slot 3 the
  // operand reads).  Measured on top of this: everything one or two k-steps earlier +-0, the staging loads in a slot +-0
  auto ss_slot = [&](const int pp, const int slot, const int nxt) {
    if (pp == 2 && slot == 5) ss_y(nxt, 0);
    if (pp == 2 && slot == 6) ss_y(nxt, 1);
    if (pp == 2 && slot == 7) ss_pre();
    if (pp == 3 && slot >= 4 && slot <= 6) { ss_aff(2 * (slot - 4)); ss_aff(2 * (slot - 4) + 1); }
    if (pp == 4 && slot >= 4) ss_x(nxt, 0, slot - 4);
    if (pp == 5 && slot >= 4) ss_x(nxt, 1, slot - 4);
  };

  // ---- per-lane read bases.  x rows (a, b, c) of the wave's two fy rows:  fy = 2 fyh: R_a - R_b,  fy = 2 fyh + 1:
  // R_b + sg R_c  with (a, b, c, sg) = (0,2,1,+) / (2,1,3,-);  dy: rows fz of A as (ca, cb), rows fy as (a1, b0)
  const float sg = fyh ? -1.f : 1.f;
  const int xbase = c * GXP + fz * 60 + 2 * hl;
  const int XA = xbase + (fyh ? 2 : 0) * GX, XB = xbase + (fyh ? 1 : 2) * GX, XC = xbase + (fyh ? 3 : 1) * GX;
  const int YB = 2 * GXF + c * GYP + 2 * hl;
  const float ca = (fz == 3) ? 0.f : 1.f, cb = (fz == 0) ? 0.f : (fz == 2 ? -1.f : 1.f);   // fz = 3 un-negated
  const float a1 = fyh ? -1.f : 0.f, b0 = fyh ? 0.f : 1.f;   // rlo = g0 + a1 g1, rhi = b0 g0 + g1 (fy = 3 un-negated)

  wf16 acc[8];
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  float u[2][8], vv[2][8];             // operands of k-step p in [p & 1]: f = fy_local * 4 + fx
  wf2 xa[2], xb2[2], xc[2];            // raw reads of a later k-step: rows a, b, c, x pairs (0,1) (2,3)
  wf2 e0[2], e1[2];                    // dy: z = 2tz / 2tz+1, [dyy] = (dx0, dx1)
  auto rd = [&](const int bo, const int p) {
    const int tz = p >> 2, ty = (p >> 1) & 1, txp = p & 1;
    const int ox = bo * GXF + tz * 240 + (2 * ty) * GX + 4 * txp;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      xa[j] = *reinterpret_cast<const wf2*>(&lds[XA + ox + 2 * j]);
      xb2[j] = *reinterpret_cast<const wf2*>(&lds[XB + ox + 2 * j]);
      xc[j] = *reinterpret_cast<const wf2*>(&lds[XC + ox + 2 * j]);
    }
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int oy = bo * GYF + ((2 * tz) * 4 + 2 * ty + d) * 8 + 4 * txp;
      e0[d] = *reinterpret_cast<const wf2*>(&lds[YB + oy]);
      e1[d] = *reinterpret_cast<const wf2*>(&lds[YB + oy + 32]);
    }
  };
  // The transforms run on PACKED fp32 (v_pk_add_f32 / v_pk_fma_f32: two values per lane and instruction at the scalar
  // rate -- fp32 VALU shares the issue port with the MFMAs, so every instruction saved is matrix time): the x pairs as
  // read, op_sel / neg modifiers for the butterflies, 16 instead of 32 instructions per k-step.  The fx = 3 row of B^T
  // comes out negated (t3 - t1) -- which cancels the missing negation of A's last row on the dy side.
  wf2 t0p[2], t1p[2], rlp, rhp;
  const wf2 sg2 = {sg, sg}, ca2 = {ca, ca}, cb2 = {cb, cb}, a12 = {a1, a1}, b02 = {b0, b0};   // wave-uniform: SGPR pairs
  // (hipcc scalarises most of the same expressions written on float2 values: 767 VALU per 128 MFMAs instead of 688)
#define ICS_PK2(out, text, A, B) asm(text : "=v"(out) : "v"(A), "v"(B))
  auto pk_fma_s = [&](wf2 sc, wf2 x, wf2 y) { wf2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "s"(sc), "v"(x), "v"(y)); return r; };
  auto pk_mul_s = [&](wf2 sc, wf2 x) { wf2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "s"(sc), "v"(x)); return r; };
  auto tr_y = [&](int j) {             // x pair j: columns 2j, 2j+1
    ICS_PK2(t0p[j], "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]", xa[j], xb2[j]);          // a - b
    t1p[j] = pk_fma_s(sg2, xc[j], xb2[j]);                                                         // b + sg c
  };
  auto tr_x = [&](const wf2* t, float* un) {
    wf2 lo, hi;
    ICS_PK2(lo, "v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]", t[0], t[1]);               // (t0 - t2, t1 + t2)
    ICS_PK2(hi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]", t[1], t[0]);  // (t2 - t1, t3 - t1)
    un[0] = lo.x; un[1] = lo.y; un[2] = hi.x; un[3] = hi.y;                                        // un[3] = -(B^T d)_3
  };
  auto tr_d = [&]() {                  // dy: z rows then the wave's two y rows
    const wf2 g0 = pk_fma_s(cb2, e1[0], pk_mul_s(ca2, e0[0]));             // dyy = 0: dx 0, 1
    const wf2 g1 = pk_fma_s(cb2, e1[1], pk_mul_s(ca2, e0[1]));             // dyy = 1
    rlp = pk_fma_s(a12, g1, g0);
    rhp = pk_fma_s(b02, g0, g1);
  };
  auto tr_v = [&](const wf2 r, float* vn) {
    wf2 m;
    ICS_PK2(m, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]", r, r);        // (r0 + r1, r0 - r1)
    vn[0] = r.x; vn[1] = m.x; vn[2] = m.y; vn[3] = r.y;
  };

  // per_split is even and every split is full (launcher): no conditional blocks -- a conditional one makes the
  // register allocator spill the accumulators
  gload(blk_lo);
  sstore(0);
  __syncthreads();
  rd(0, 0);
  tr_y(0); tr_y(1); tr_x(t0p, u[0]); tr_x(t1p, u[0] + 4); tr_d(); tr_v(rlp, vv[0]); tr_v(rhp, vv[0] + 4);
  rd(0, 1);
  ICS_WGTL0(1);
#ifndef ICS_WG_ABL
#define ICS_WG_ABL 0        // ablation (scripts/variants.sh): 1 no transforms / operand reads / staging stores, 4 no staging,
                            // 8 no MFMA, 16 no barrier, 64 no staging loads
#endif
  auto block = [&](const int blk, const int cur, const int nxt) {
#if !(ICS_WG_ABL & (4 | 64))
    gload(blk + 1 < blk_hi ? blk + 1 : blk);             // past the end: this block again (never consumed)
#endif
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      float* uc = u[p & 1];  float* vc = vv[p & 1];      // this step's operands
      float* un = u[(p + 1) & 1];  float* vn = vv[(p + 1) & 1];
      // raw reads in xa/xb2/xc/e0/e1 belong to step p+1 (step 0 of the next block when p == 7); its transform is
      // interleaved with this step's MFMAs (measured: doing it after them costs 15 %)
#define ICS_GMF(F) acc[F] = __builtin_amdgcn_mfma_f32_32x32x2f32(uc[F], vc[F], acc[F], 0, 0, 0)
#define ICS_GFN __builtin_amdgcn_sched_barrier(0)
#if ICS_WG_ABL & 8
#undef ICS_GMF
#define ICS_GMF(F)
#endif
#if ICS_WG_ABL & 1
      ICS_GMF(0); ICS_GMF(4); ICS_GMF(1); ICS_GMF(5); ICS_GMF(2); ICS_GMF(6); ICS_GMF(3); ICS_GMF(7); ICS_GFN;
#else
      // One fence per MFMA.  Slots 0..2: the y / dy math of step p+1.  Slot 3: the raw reads of step p+2 (of this block,
      // or step (p+2)-8 of the next one) -- the raw registers are free from here on, and the reads (ds_read2_b64, 8 LDS
      // cycles each) complete under five MFMAs; at the end of the k-step, where they sat first, the next step's first
      // transform waited for them (-3.2 %).  Slots 4..7: a piece of the next block's staging store (k-steps 2..5; as one
      // burst in front of the barrier: +3.3 %).  s_setprio per wave or around the MFMAs: +-0.
      ICS_GMF(0); tr_y(0); ICS_GFN;
      ICS_GMF(4); tr_y(1); ICS_GFN;
      ICS_GMF(1); tr_d(); ICS_GFN;
      ICS_GMF(5); if (p < 6) rd(cur, p + 2); else rd(nxt, p - 6); ICS_GFN;
      ICS_GMF(2); ss_slot(p, 4, nxt); ICS_GFN;
      ICS_GMF(6); ss_slot(p, 5, nxt); ICS_GFN;
      ICS_GMF(3); ss_slot(p, 6, nxt); ICS_GFN;
      ICS_GMF(7); ss_slot(p, 7, nxt); ICS_GFN;
#endif
#undef ICS_GMF
      // the x transform of step p+1 as a burst here: it runs under the other wave's MFMAs (in the slots above: +1.5 %)
#if !(ICS_WG_ABL & 1)
      tr_x(t0p, un); tr_x(t1p, un + 4); tr_v(rlp, vn); tr_v(rhp, vn + 4);
#endif
      // the barrier of the block: every read of `cur` is issued (slot 3 of step 5 was the last), the next block's last
      // store (slot 7 of step 5) is behind us, and step 6 reads the next block's step 0.  No wave waits here any more
      // (a build without the barrier: +-0)
#if !(ICS_WG_ABL & (4 | 16))
      if (p == 5) __syncthreads();
#endif
      ICS_GFN;
#undef ICS_GFN
    }
  };
  for (int blk = blk_lo; blk < blk_hi; blk += 2) {
    block(blk, 0, 1);
    block(blk + 1, 1, 0);
  }
  ICS_WGTL(2 + w);

  // ---------------------------------------------------------------- epilogue: G^T contraction, signs of the f = 3 rows
  // in registers: fx -> c and this wave's share of fy -> b;  part[w 8][9 (b,c)][4 rr][64 lanes]  (73 728 B per pass)
  float* part = lds;
  float* wsp = ws + (size_t)split * 27 * Cin * Cout;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = pass * 4 + rr;
      float yc[2][3];                                    // [fy local][c]
#pragma unroll
      for (int fy = 0; fy < 2; ++fy) {
        const float X0 = acc[fy * 4 + 0][r], X1 = acc[fy * 4 + 1][r], X2 = acc[fy * 4 + 2][r], X3 = acc[fy * 4 + 3][r];
        const float h1 = 0.5f * (X1 + X2), h2 = 0.5f * (X1 - X2);
        yc[fy][0] = X0 + h1; yc[fy][1] = h2; yc[fy][2] = h1 + X3;      // X3 carries its true sign (packed transform)
      }
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        // rows of G^T over fy: b0 = Y0 + (Y1 + Y2)/2, b1 = (Y1 - Y2)/2, b2 = (Y1 + Y2)/2 - Y3'
        const float lo = yc[0][cc], hi = yc[1][cc];      // fyh = 0: (Y0, Y1); fyh = 1: (Y2, Y3')
        const float v0 = fyh ? 0.5f * lo : lo + 0.5f * hi;
        const float v1 = fyh ? -0.5f * lo : 0.5f * hi;
        const float v2 = fyh ? 0.5f * lo - hi : 0.5f * hi;
        part[((w * 9 + 0 * 3 + cc) * 4 + rr) * 64 + lane] = v0;
        part[((w * 9 + 1 * 3 + cc) * 4 + rr) * 64 + lane] = v1;
        part[((w * 9 + 2 * 3 + cc) * 4 + rr) * 64 + lane] = v2;
      }
    }
    __syncthreads();
    // task = ((bc * 4 + rr) * 2 + hh) * 8 + quad: 576 per pass; each gives the three taps a = 0,1,2 as float4
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int task = tid + 512 * i;
      if (task < 576) {
        const int quad = task & 7, hh = (task >> 3) & 1, rr = (task >> 4) & 3, bc = task >> 6;
        const int r = pass * 4 + rr;
        const int row = (r >> 2) * 8 + hh * 4 + (r & 3);
        const int slot = (bc * 4 + rr) * 64 + hh * 32 + 4 * quad;
        wf4 pz[4];
#pragma unroll
        for (int z = 0; z < 4; ++z)
          pz[z] = *reinterpret_cast<const wf4*>(&part[(2 * z) * 2304 + slot]) +
                  *reinterpret_cast<const wf4*>(&part[(2 * z + 1) * 2304 + slot]);
        const wf4 h1 = 0.5f * (pz[1] + pz[2]), h2 = 0.5f * (pz[1] - pz[2]);
        float* o = wsp + ((size_t)(bc)*Cin + ci0 + row) * Cout + co0 + 4 * quad;   // tap = a*9 + bc
        *reinterpret_cast<wf4*>(o) = pz[0] + h1;
        *reinterpret_cast<wf4*>(o + (size_t)9 * Cin * Cout) = h2;
        *reinterpret_cast<wf4*>(o + (size_t)18 * Cin * Cout) = h1 - pz[3];
      }
    }
  }
#ifdef ICS_WG_TIMELINE
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
    unsigned long long* r = ics_wg_tl + (size_t)blockIdx.x * 16;
    r[10] = wall_clock64();
    r[11] = (unsigned long long)per_split;
    r[14] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
    r[15] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
  }
#endif
}

// ---------------------------------------------------------------- host side
bool conv_wino_ok(const ConvGeom& g, const ConvSrc* src, int nsrc) {
  if (g.flags & CF_NO_WINO) return false;
  if (g.taps != 27 || nsrc != 1 || g.S < 8 || g.lgS < 3) return false;
  const ConvSrc& s = src[0];
  if (s.up || s.bcast || s.C != g.Cin) return false;
  if (g.Cin % (2 * KC) != 0 || g.Cout % 32 != 0) return false;
  if ((long long)g.B * g.S * g.S * g.S * (long long)std::max(g.Cin, g.Cout) >= (1ll << 31)) return false;  // 32-bit offsets
  return true;
}
size_t conv_wino_weight_floats(int Cin, int Cout) { return (size_t)64 * Cin * Cout; }

int launch_conv_fwd_wino(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* wt, const float* bias,
                         float* out, int ldo, int pre_act, float* stat_partial, int* rows_per_block, int accumulate,
                         int layout, const BwdStat* bwd, int* bwd_blocks) {
  ICS_CHECK(conv_wino_ok(g, &s0, 1), "shape not served by the Winograd kernel");
  ICS_CHECK(layout == 0 || layout == 1, "unknown Winograd weight layout");
  if (layout == 1)       // the 16-tile x 64-channel shape: the layout the weights were packed in decides, not this launch
    return launch_conv_fwd_wino64(st, g, s0, wt, bias, out, ldo, pre_act, stat_partial, rows_per_block, accumulate, bwd,
                                  bwd_blocks);
  ICS_CHECK(ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(bias) & 15) == 0 && (reinterpret_cast<uintptr_t>(s0.p) & 15) == 0,
            "Winograd kernel: float4 accesses need 16-byte aligned tensors");
  const unsigned grid = (unsigned)(g.B * (g.S / 4) * (g.S / 8) * (g.S / 8) * (g.Cout / 32));
  if (rows_per_block) *rows_per_block = kRowsPerBlock;
  const bool aff = s0.scale != nullptr;
  const bool noact = s0.act == ACT_NONE;
  const float in_slope = wslope(s0.act), pre_slope = wslope(pre_act);
  // the folded BatchNorm-backward sums: plain backward-data launches only (no source affine, no statistics, no
  // activation); the producer's tensors must be float4-addressable like the output
  const bool plain = bwd != nullptr && !aff && stat_partial == nullptr && bias == nullptr && pre_act == ACT_NONE &&
                     !accumulate && bwd->ld % 4 == 0;
  const bool apply = plain && bwd->abc != nullptr && bwd->pool_d == nullptr;
  ICS_CHECK(bwd == nullptr || bwd->abc == nullptr || (apply && bwd->db_partial != nullptr && bwd->s != nullptr),
            "Winograd backward-data with the fused BatchNorm-backward apply: not a plain launch (or a pooled source: 64-channel kernel only)");
  const bool fold = plain && !apply && bwd->partial != nullptr;
  if (bwd_blocks) *bwd_blocks = (fold || apply) ? (int)(grid / (unsigned)(g.Cout / 32)) : 0;
  const BwdStat bs = (fold || apply) ? *bwd : BwdStat{};
#define ICS_WINO_LAUNCH(AFFV, NOACTV, FOLDV)                                                                        \
  do {                                                                                                              \
    ICS_LAUNCH((conv_wino_kernel<AFFV, NOACTV, FOLDV>), dim3(grid), dim3(512), 0, st, s0.p, s0.C, s0.scale, \
                       s0.shift, in_slope, wt, bias, out, ldo, pre_slope, accumulate, stat_partial, g.Npad, g.S,    \
                       g.Cin, g.Cout, bs);                                                                          \
    conv_set_last_kernel_id("conv_wino_kernel<" #AFFV ", " #NOACTV ", " #FOLDV ">");                                \
  } while (0)
  if (apply) ICS_WINO_LAUNCH(false, true, 2);
  else if (fold) ICS_WINO_LAUNCH(false, true, 1);
  else if (!aff) ICS_WINO_LAUNCH(false, true, 0);
  else if (noact) ICS_WINO_LAUNCH(true, true, 0);
  else ICS_WINO_LAUNCH(true, false, 0);
#undef ICS_WINO_LAUNCH
  ICS_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- backward-weight launcher
// blocks per workgroup: a power of two >= 2 that divides the number of blocks (every split full, even count), small
// enough for ~1 workgroup per CU, large enough for the workspace.  A split is a contiguous range of the linear block
// index (b, bz, by, bx) and may span samples: capping it at one sample's blocks (round 2) left the S = 8 layers with 32
// splits of 4 blocks where 4 splits of 32 were wanted -- 8x the fixed cost and 8x the partial sums (c14: 0.62 ms).
#ifndef ICS_WG_WANT
#define ICS_WG_WANT 256
#endif
static int wino_wgrad_per_split(const ConvGeom& g, size_t ws_floats) {
  const int nblocks = g.B * (g.S / 4) * (g.S / 4) * (g.S / 8);
  const int pairs = (g.Cin / 32) * (g.Cout / 32);
  // Workgroups wanted per launch (one per CU; the kernel takes a whole CU).  Per U-Net step, round 2: 256 -> 36.96 ms,
  // 512 -> 37.05, 1536 -> 37.15, 3072 -> 37.7; end of round 3: 256 -> 30.5, 384 -> 30.5, 512 -> 30.66 (half the
  // epilogues and half the partial sums for reduce_splits: 0.32 -> 0.28 ms).  Round 2 kept 512 so that a communication
  // kernel holding a few CUs costs a third round instead of a second; the gradient buckets (125 MB per step) are
  // estimated to keep RCCL kernels live for a few per cent of a step, so that insurance buys less than the 0.15 ms it
  // costs (single-rank communicator forced on: scripts/forcedist_bench.sh).  The split count fixes the
  // summation order: it must not depend on whether a communicator is attached (tests/test_gpu_dp.py compares bit for bit).
  const int want = std::max((ICS_WG_WANT + pairs - 1) / pairs, 1);  // splits wanted
  const size_t per = (size_t)27 * g.Cin * g.Cout;
  int ps = 2;
  while (nblocks % (2 * ps) == 0 && ((nblocks / ps) > want || (size_t)(nblocks / ps) * per > ws_floats)) ps *= 2;
  if (nblocks % ps != 0 || (size_t)(nblocks / ps) * per > ws_floats) return 0;
  return ps;
}
size_t conv_wino_wgrad_workspace_floats(const ConvGeom& g) {
  const int ps = wino_wgrad_per_split(g, ~(size_t)0);
  return (size_t)(g.B * (g.S / 4) * (g.S / 4) * (g.S / 8) / ps) * 27 * g.Cin * g.Cout;
}
bool conv_wino_wgrad_ok(const ConvGeom& g, const ConvSrc* src, int nsrc) {
  if (g.flags & (CF_NO_WINO | CF_NO_WINO_WGRAD)) return false;
  if (g.taps != 27 || nsrc != 1 || g.S < 8) return false;
  const ConvSrc& s = src[0];
  if (s.up || s.bcast || s.C != g.Cin) return false;
  if (g.Cin % 32 != 0 || g.Cout % 32 != 0) return false;
  if ((long long)g.B * g.S * g.S * g.S * (long long)std::max(g.Cin, g.Cout) >= (1ll << 29)) return false;   // 32-bit BYTE offsets
  return true;
}
int launch_conv_wgrad_wino(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* dy, int ldy, float* dw,
                           int ldw, float* ws, size_t ws_floats, int sub_rows, int row_pitch, int row_off, int phase) {
  ICS_CHECK(conv_wino_wgrad_ok(g, &s0, 1), "shape not served by the Winograd backward-weight kernel");
  const int per_split = wino_wgrad_per_split(g, ws_floats);
  ICS_CHECK(per_split >= 2, "wgrad workspace too small");
  const int nblocks = g.B * (g.S / 4) * (g.S / 4) * (g.S / 8);
  const int nsplit = nblocks / per_split;
  if (phase != 2) {
    const unsigned grid = (unsigned)(nsplit * (g.Cin / 32) * (g.Cout / 32));
    const bool aff = s0.scale != nullptr, noact = s0.act == ACT_NONE;
    const float in_slope = wslope(s0.act);
#define ICS_WG_LAUNCH(AFFV, NOACTV)                                                                                   \
  do {                                                                                                                \
    ICS_LAUNCH((conv_wino_wgrad_kernel<AFFV, NOACTV>), dim3(grid), dim3(512), 0, st, s0.p, s0.C, s0.scale,    \
                       s0.shift, in_slope, dy, ldy, ws, g.S, g.Cin, g.Cout, per_split);                               \
    conv_set_last_kernel_id("conv_wino_wgrad_kernel<" #AFFV ", " #NOACTV ">");                                        \
  } while (0)
    if (!aff) ICS_WG_LAUNCH(false, true);
    else if (noact) ICS_WG_LAUNCH(true, true);
    else ICS_WG_LAUNCH(true, false);
#undef ICS_WG_LAUNCH
    ICS_HIP(hipGetLastError());
  }
  if (phase != 1)
    ICS_TRY(launch_wgrad_reduce_splits(st, ws, nsplit, (size_t)27 * g.Cin * g.Cout, g.Cout, dw, ldw, sub_rows, row_pitch,
                                       row_off));
  return 0;
}

}  // namespace ics
