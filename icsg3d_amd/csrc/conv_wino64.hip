// Winograd F(2x2x2, 3x3x3) forward / backward-data, second kernel shape (gfx950 / MI355X only): one workgroup =
// 16 tiles (2x2x4 tiles = 4x4x8 voxels) x 64 output channels x all 64 frequencies on v_mfma_f32_16x16x4_f32.
//
// Same operator and same algebra as conv_wino.hip (Keras Conv3D 3x3x3 "same", /root/reference/unet/unet.py:283-336);
// what changes is the shape of the per-frequency GEMM tile, 16 tiles x 64 channels instead of 32 x 32:
//   * the transformed input (the MFMA A operand) is built ONCE per (tile, input channel, frequency) and feeds FOUR
//     16x16x4 MFMAs (four 16-channel column blocks) instead of one 32-wide tile: the transform's VALU instructions per
//     MFMA cycle halve (fp32 VALU and fp32 MFMA share the issue port on this chip: every VALU instruction is MFMA
//     time lost -- scripts/probes/mfma_filler.hip);
//   * the halo block is [6][6][10] voxels for 128 outputs x 64 channels instead of [6][10][10] for 256 x 32: 0.6x the
//     staging work and 0.6x the input re-reads per MFMA (Cout/64 instead of Cout/32 n-chunks);
//   * a buffer of 32 input channels is 64 KB, so the K loop runs in chunks of 32 channels, double buffered: half the
//     barriers per channel and twice the distance between a chunk's global loads and their first use.
// 8 waves, two per SIMD: wave w owns the 8 frequencies (fz = w >> 1, fy in {2 (w & 1), 2 (w & 1) + 1}, fx = 0..3) x 4
// column blocks = 32 accumulators of 4 registers.  MFMA lane l: A[tile = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15],
// D[tile = 4 (l >> 4) + reg][n = l & 15].  Staging as in conv_wino.hip: thread t < 480 owns one (y, x, channel quad)
// column of the halo, applies the producer's BatchNorm affine, the zero padding after it and the z rows of B^T, and
// writes the eight z-combined planes (tile z, fz).  LDS: voxel pitch 33 floats, row pitch 332, plane pitch 1996:
// the 32 lanes of a ds_read_b32 group (16 tiles x 2 channels) hit 32 distinct banks for every (fz, row, column,
// sub-step) -- checked exhaustively.  Weights: [Cout/64][Cin/4][64 f][4 k][16 n][4 column blocks],
// one ds-free global_load_dwordx4 per frequency and sub-step, a wave's sub-step = 8 KB contiguous.
#include "common.h"

#include <algorithm>
#include <type_traits>

#ifndef ICS_W64_FENCE
#define ICS_W64_FENCE 1      // 2: a scheduling fence after every MFMA pair; 1: one per column; 0: none.  Measured, c18
                             // forward: 2.59 / 2.53 / 2.72 ms (with ICS_W64_COLTOP: 2.51 / 2.51 / 2.77)
#endif
#ifndef ICS_W64_COLTOP
#define ICS_W64_COLTOP 1     // 1: the two column-math instructions ahead of a column's MFMAs instead of between them
                             // (no copies of the three operands, one MFMA -> VALU -> MFMA switch per column less)
#endif
#ifndef ICS_W64_PRIO
#define ICS_W64_PRIO 0       // 1: s_setprio(1) around a column's MFMAs (measured: +1 % time)
#endif

#ifdef ICS_W64_TIMELINE
// variant builds only (scripts/variants.sh conv_wino64 "tl:-DICS_W64_TIMELINE"): per workgroup {entry, first chunk staged,
// main loop done, exit; 4..: inside the epilogue} wall-clock stamps + HW_ID / XCC_ID, fetched with ics_debug_w64_timeline (scripts/w64_timeline.py)
__device__ unsigned long long ics_w64_tl[16 * 32768];
extern "C" int ics_debug_w64_timeline(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ics_w64_tl), (size_t)n * 8);
}
#define ICS_TL(i) do { if (threadIdx.x == 0 && blockIdx.x < 32768) ics_w64_tl[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define ICS_TL(i)
#endif

namespace ics {

typedef float vf4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int KC = 32;                                   // input channels per LDS chunk
constexpr int VX = KC + 1, RP = 10 * VX + 2, PP = 6 * RP + 4, BUF = 8 * PP;   // floats: 33 / 332 / 1996 / 15 968 (63 872 B)
constexpr int kRows = 128;                               // voxels per workgroup

__device__ __forceinline__ float wact(float v, float slope) { return fmaxf(v, v * slope); }
__host__ __device__ __forceinline__ float wslope(int act) { return act == ACT_RELU ? 0.f : (act == ACT_LRELU ? kLeaky : 1.f); }
}  // namespace

// FOLD: 0 plain; 1 the consumer-side BatchNorm-backward sums of the producer folded into the epilogue (BwdStat::partial);
// 2 the producer's whole BatchNorm-backward APPLY in the epilogue (BwdStat::abc, round 4): the launch writes
//   dy_P = relu'(s) * (a * d + b * s + c),   a = scale, b = -scale c2 rstd, c = scale (c2 rstd mean - c1)
// (= scale (d - c1 - xhat c2), the three per-channel constants formed in fp64 by conv_bnfuse_kernel) and the per-block
// column sums of dy_P (the producer's bias gradient) to BwdStat::db_partial [blocks][Cout];
// 3 = 2 with the producer's second consumer, a MaxPool3D, added to d first: d += pool_d[v >> 1] where this voxel's bit of
//   the window's tie mask is set (BwdStat::pool_d / pool_mask; the masks are written by the forward pooling kernel).
template <bool AFF, bool NOACT, int FOLD>
__global__ __launch_bounds__(512) void conv_wino64_kernel(const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ in_scale,
                                                          const float* __restrict__ in_shift, float in_slope,
                                                          const float* __restrict__ wt, const float* __restrict__ bias,
                                                          float* __restrict__ y, int ldo, float pre_slope,
                                                          int accumulate, float* __restrict__ stat_partial, int Npad,
                                                          int S, int Cin, int Cout, BwdStat bs) {
  // 2 buffers 127 744 B (the epilogue reuses them) + per-thread constants 4 KB + AFF: scale / shift 8 KB
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
  __shared__ unsigned park[1536];
  __shared__ __attribute__((aligned(16))) float aff[AFF ? 2048 : 4];
  ICS_TL(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fz = w >> 1, fyh = w & 1;            // this wave: frequencies (fz, 2 fyh + {0,1}, 0..3)
  const int m = lane & 15, kq = lane >> 4;
  const int nchunks = Cout >> 6;
  const int nb = blockIdx.x % nchunks;           // the n-chunks of one tile block are neighbours in launch order: an XCD
  const int tblk = blockIdx.x / nchunks;         // (blockIdx mod 8) keeps ONE n-chunk's weights hot in its L2
  int tb = tblk;
  const int nbx = S >> 3, nby = S >> 2, nbz = S >> 2;
  const int bx = tb % nbx; tb /= nbx;
  const int by = tb % nby; tb /= nby;
  const int bz = tb % nbz;
  const int b = tb / nbz;
  const int oz = bz * 4, oy = by * 4, ox = bx * 8, n0 = nb * 64;
  const bool edge = bx == 0 || by == 0 || bz == 0 || bx == nbx - 1 || by == nby - 1 || bz == nbz - 1;   // uniform

  // ---- staging: thread t < 480 owns one (hy, hx, channel quad) column of the halo block [6][6][10] x 32 channels
  const int cmb = tid < 480 ? tid : 479;
  const int q = cmb & 7, hx = (cmb >> 3) % 10, hy = (cmb >> 3) / 10;
  // Buffer loads: ONE per-lane byte offset (the clamped (y, x) position and the channel quad) in a VGPR; the
  // clamped z plane of each of the six loads and the channel chunk go into the scalar offset operand (six 64-bit
  // per-lane pointers cost 12 VGPRs, which the allocator spilled).
  vf4 stage[6];
  unsigned zoff[6];                              // uniform: byte offset of sample b, plane clamp(oz - 1 + hz)
  unsigned okmask = 0;
  unsigned voff;
  {
    const int gy = oy - 1 + hy, gx = ox - 1 + hx;
    const bool okyx = gy >= 0 && gy < S && gx >= 0 && gx < S;
    const int cy = min(max(gy, 0), S - 1), cx = min(max(gx, 0), S - 1);
    voff = (unsigned)((cy * S + cx) * ldx + q * 4) * 4u;
#pragma unroll
    for (int hz = 0; hz < 6; ++hz) {
      const int gz = oz - 1 + hz;
      okmask |= (okyx && gz >= 0 && gz < S) ? (1u << hz) : 0u;
      const int cz = min(max(gz, 0), S - 1);
      zoff[hz] = (unsigned)((b * S + cz) * S * S * ldx) * 4u;
    }
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x7fffffff, 0x00020000);
  const int dbase = hy * RP + hx * VX + q * 4;   // + plane * PP + buffer
  // The two per-thread staging constants are PARKED in LDS and read back where a chunk is loaded / stored: kept in
  // registers they are what the allocator spills, and a scratch reload is a vector-memory load whose s_waitcnt
  // vmcnt(0) drains the whole weight prefetch at every chunk (measured: +9 % on the BatchNorm-affine variant).
  park[tid] = voff;
  park[512 + tid] = (unsigned)dbase;
  park[1024 + tid] = (unsigned)(q * 4);
  // LDS byte address of the wave's 64 entries (scalar); the lane part is rebuilt from v_mbcnt at every read: a per-lane
  // address register is itself one more value for the allocator to spill (it did, with a vmcnt(0) reload per read)
  const unsigned park_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&park[0] + (unsigned)w * 256u);
  auto unpark = [&](const int which) -> int {    // an LDS read the compiler can neither hoist nor keep in a register
    int v;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshl_add_u32 %0, %0, 2, %1\n\t"
                 "ds_read_b32 %0, %0 offset:%2\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v) : "s"(park_wave), "i"(which * 2048) : "memory");
    return v;
  };
  // AFF: the producer's per-channel scale / shift of ALL input channels sit in LDS behind the two buffers (Cin <= 1024:
  // 8 KB), read back per chunk at store time -- eight registers less to keep live through the main loop
  // (requested together with the first halo block and the first weights, below: as a load -> wait -> store loop up here
  // it put a full memory round trip in front of every other load of the prologue)
  auto gload = [&](int c0) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
      stage[i] = __builtin_bit_cast(vf4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)voff, (int)(zoff[i] + (unsigned)c0 * 4u), 0));
  };
  auto sstore = [&](const int bo, const int c0) {
    if (AFF) {
      const vf4 sc4 = *reinterpret_cast<const vf4*>(&aff[c0 + q * 4]);
      const vf4 sh4 = *reinterpret_cast<const vf4*>(&aff[1024 + c0 + q * 4]);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        vf4 t = stage[i];
        t.x = fmaf(t.x, sc4.x, sh4.x); t.y = fmaf(t.y, sc4.y, sh4.y);
        t.z = fmaf(t.z, sc4.z, sh4.z); t.w = fmaf(t.w, sc4.w, sh4.w);
        if (!NOACT) { t.x = wact(t.x, in_slope); t.y = wact(t.y, in_slope); t.z = wact(t.z, in_slope); t.w = wact(t.w, in_slope); }
        stage[i] = t;
      }
    }
    if (edge) {                                  // "same" padding: zeros AFTER the producer's affine / activation
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if (!((okmask >> i) & 1)) stage[i] = vf4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < 480) {
      float* o = &lds[bo + dbase];
#pragma unroll
      for (int tz = 0; tz < 2; ++tz) {
        const vf4 d0 = stage[2 * tz], d1 = stage[2 * tz + 1], d2 = stage[2 * tz + 2], d3 = stage[2 * tz + 3];
#pragma unroll
        for (int f = 0; f < 4; ++f) {            // one plane at a time: four live temporaries, not sixteen
          const vf4 c = f == 0 ? d0 - d2 : (f == 1 ? d1 + d2 : (f == 2 ? d2 - d1 : d1 - d3));
          float* op = o + (tz * 4 + f) * PP;     // odd voxel pitch: four dword stores
          op[0] = c.x; op[1] = c.y; op[2] = c.z; op[3] = c.w;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };

  // In the main loop the halo of the next chunk is staged in TWO phases of four z rows each -- rows 0..3 -> the planes of
  // tile z 0, rows 2..5 -> tile z 1 (rows 2, 3 are read twice, from L1 / L2) -- so that 16 instead of 24 registers are in
  // flight next to the accumulators; the prologue, where nothing else is live yet, uses the six-row form above.
  vf4 hs[4];
  auto hload = [&](const int tzh, int c0) {
    const int vo = unpark(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      hs[i] = __builtin_bit_cast(vf4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, (int)(zoff[2 * tzh + i] + (unsigned)c0 * 4u), 0));
  };
  auto hstore = [&](const int tzh, const int bo, const int c0) {
    vf4 r[4] = {hs[0], hs[1], hs[2], hs[3]};     // local copies: updating hs in place can send it to scratch (compiler)
    if (AFF) {
      const int q4 = unpark(2);
      typedef float vf2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {           // two channels at a time: four instead of eight live scale / shift registers
        const vf2 sc2 = *reinterpret_cast<const vf2*>(&aff[c0 + q4 + 2 * hh]);
        const vf2 sh2 = *reinterpret_cast<const vf2*>(&aff[1024 + c0 + q4 + 2 * hh]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float t0 = fmaf(r[i][2 * hh], sc2.x, sh2.x), t1 = fmaf(r[i][2 * hh + 1], sc2.y, sh2.y);
          if (!NOACT) { t0 = wact(t0, in_slope); t1 = wact(t1, in_slope); }
          r[i][2 * hh] = t0; r[i][2 * hh + 1] = t1;
        }
      }
    }
    if (edge) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (!((okmask >> (2 * tzh + i)) & 1)) r[i] = vf4{0.f, 0.f, 0.f, 0.f};
    }
    // !AFF: threads 480..511 (clamped to thread 479's column: same data) store too -- a duplicate store of identical values
    // instead of an exec-masked region in the main loop (backward-data -0.9 %).  The AFF variants keep the predicate: without
    // it they spill four more registers (forward +3 %).
    if (!AFF || tid < 480) {
      float* o = &lds[bo + unpark(1) + tzh * 4 * PP];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const vf4 c = f == 0 ? r[0] - r[2] : (f == 1 ? r[1] + r[2] : (f == 2 ? r[2] - r[1] : r[1] - r[3]));
        float* op = o + f * PP;
        op[0] = c.x; op[1] = c.y; op[2] = c.z; op[3] = c.w;
      }
    }
  };

  // ---- per-lane read geometry.  tile m = (tz, ty, tx).  The wave's two fy rows of B^T need rows (a, b, c) of the
  // combined plane:  fy = 2 fyh:  R_a - R_b,   fy = 2 fyh + 1:  R_b + sg R_c   with (a,b,c,sg) = (0,2,1,+) / (2,1,3,-)
  const int tz = m >> 3, ty = (m >> 2) & 1, tx = m & 3;
  const float sg = fyh ? -1.f : 1.f;
  auto rowbase = [&](int iy) { return (tz * 4 + fz) * PP + (2 * ty + iy) * RP + 2 * tx * VX + kq; };
  const int Ra0 = rowbase(fyh ? 2 : 0), Rb0 = rowbase(fyh ? 1 : 2), Rc0 = rowbase(fyh ? 3 : 1);

  const int nsub = Cin >> 2;
  constexpr int wstride_f = 256;                 // floats per frequency of one sub-step: [4 k][16 n][4 column blocks]
  constexpr int wsub = 64 * 256;
  // buffer loads again: one per-lane byte offset, everything else (n-chunk, wave, sub-step, frequency) scalar
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wt + ((size_t)nb * nsub * 64 + fz * 16 + fyh * 8) * 256), 0, 0x7fffffff, 0x00020000);
  const int wlane = lane * 16;                   // bytes
  auto wload = [&](int gs, int f) {              // sub-step gs, local frequency f
    return __builtin_bit_cast(vf4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane, (gs * wsub + f * wstride_f) * 4, 0));
  };
  vf4 wreg[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) wreg[f] = wload(0, f);

  vf4 acc[8][4];                                 // [frequency fy_local * 4 + fx][column block]
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[f][j] = vf4{0.f, 0.f, 0.f, 0.f};

  float u[8], tn[2][4], qa, qb, qc;
  auto xform = [&]() {
#pragma unroll
    for (int fy = 0; fy < 2; ++fy) {
      u[fy * 4 + 0] = tn[fy][0] - tn[fy][2];
      u[fy * 4 + 1] = tn[fy][1] + tn[fy][2];
      u[fy * 4 + 2] = tn[fy][2] - tn[fy][1];
      u[fy * 4 + 3] = tn[fy][1] - tn[fy][3];
    }
  };

  float a_sc[2] = {0.f, 0.f}, a_sh[2] = {0.f, 0.f};
  if (AFF) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + 512 * k < Cin) { a_sc[k] = in_scale[tid + 512 * k]; a_sh[k] = in_shift[tid + 512 * k]; }
  }
  gload(0);
  if (AFF) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + 512 * k < Cin) { aff[tid + 512 * k] = a_sc[k]; aff[1024 + tid + 512 * k] = a_sh[k]; }
    __syncthreads();                             // scale / shift visible
  }
  sstore(0, 0);
  __syncthreads();
  ICS_TL(1);

  auto rd = [&](const int ra, const int rb, const int rc, const int sub, const int col) {
    const int off = col * VX + 4 * sub;          // compile-time after unrolling
    qa = lds[ra + off]; qb = lds[rb + off]; qc = lds[rc + off];
  };
  int Ra = Ra0, Rb = Rb0, Rc = Rc0;              // row bases of the buffer being read
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    rd(Ra, Rb, Rc, 0, g);
    tn[0][g] = qa - qb;
    tn[1][g] = fmaf(sg, qc, qb);
  }
  xform();
  rd(Ra, Rb, Rc, 1, 0);                          // column 0 of sub-step 1

  const int nch = Cin / KC;
  int nxt = BUF;                                 // float offset of the buffer being filled (the other one is consumed)
#ifndef ICS_W64_PEEL
#define ICS_W64_PEEL 1       // 1: the last chunk is a second copy of the loop body without the staging of a next chunk
#endif
  // ST = false: nothing is staged (the last chunk; its read-ahead columns wrap around inside the buffer being consumed
  // and are never used)
  auto chunk = [&](const int ch, auto stage_tag) {
    constexpr bool ST = decltype(stage_tag)::value;
#if ICS_W64_PEEL
    const int cn = (ch + 1) * KC;
#else
    const int cn = (ch + 1 < nch ? ch + 1 : ch) * KC;                   // past the end: the last chunk again (never consumed)
#endif
    const int dlt = 2 * nxt - BUF;               // + BUF / - BUF: from the buffer being consumed to the other one
    if (ST) hload(0, cn);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      int gs = ch * 8 + s + 1;
      gs = gs < nsub ? gs : nsub - 1;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#if ICS_W64_COLTOP
        tn[0][g] = qa - qb;                      // column g of sub-step s+1, read one column ago
        tn[1][g] = fmaf(sg, qc, qb);
#else
        const float a = qa, bq = qb, c = qc;
#endif
        if (ST && s == 3 && g == 3) {            // first half of the next chunk's planes; then the second half's rows
          hstore(0, nxt, cn);
          hload(1, cn);
        }
        if (ST && s == 6 && g == 3) {            // the next chunk must be visible before its first column is read
          hstore(1, nxt, cn);
          __syncthreads();
          Ra += dlt; Rb += dlt; Rc += dlt;       // every read from here on is in the other buffer
        }
        // reads of the next column: column g+1 of sub-step s+1, or column 0 of sub-step s+2
        if (g < 3) rd(Ra, Rb, Rc, (s + 1) & 7, g + 1);
        else rd(Ra, Rb, Rc, (s + 2) & 7, 0);
#if ICS_W64_FENCE != 0
        __builtin_amdgcn_sched_barrier(0);
#endif
#define ICS_WMF(F, J) acc[F][J] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[F], wreg[F][J], acc[F][J], 0, 0, 0)
#if ICS_W64_FENCE == 2
#define ICS_WFN __builtin_amdgcn_sched_barrier(0)
#define ICS_WFC __builtin_amdgcn_sched_barrier(0)
#elif ICS_W64_FENCE == 1
#define ICS_WFN
#define ICS_WFC __builtin_amdgcn_sched_barrier(0)
#else
#define ICS_WFN
#define ICS_WFC
#endif
        // consecutive MFMAs on different accumulators; the column math of the next sub-step in between
#if ICS_W64_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#if ICS_W64_COLTOP
        ICS_WMF(g, 0); ICS_WMF(4 + g, 0); ICS_WFN;
        ICS_WMF(g, 1); ICS_WMF(4 + g, 1); ICS_WFN;
#else
        ICS_WMF(g, 0); ICS_WMF(4 + g, 0); tn[0][g] = a - bq; ICS_WFN;
        ICS_WMF(g, 1); ICS_WMF(4 + g, 1); tn[1][g] = fmaf(sg, c, bq); ICS_WFN;
#endif
        ICS_WMF(g, 2); ICS_WMF(4 + g, 2); ICS_WFN;
        ICS_WMF(g, 3); ICS_WMF(4 + g, 3); ICS_WFC;
#if ICS_W64_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#undef ICS_WMF
        wreg[g] = wload(gs, g);
        wreg[4 + g] = wload(gs, 4 + g);
        ICS_WFC;
#undef ICS_WFN
#undef ICS_WFC
      }
      xform();
    }
    nxt = BUF - nxt;
  };
#if ICS_W64_PEEL
  for (int ch = 0; ch < nch - 1; ++ch) chunk(ch, std::true_type{});
  chunk(nch - 1, std::false_type{});
#else
  for (int ch = 0; ch < nch; ++ch) chunk(ch, std::true_type{});
#endif

  ICS_TL(2);
#ifdef ICS_W64_TIMELINE
  auto tl_flush = [&]() {
    if (threadIdx.x == 0 && blockIdx.x < 32768) {
      unsigned long long* r = ics_w64_tl + (size_t)blockIdx.x * 16;
      r[3] = wall_clock64();
      r[14] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
      r[15] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    }
  };
#define ICS_TL_FLUSH() tl_flush()
#else
#define ICS_TL_FLUSH()
#endif
  // ---------------------------------------------------------------- epilogue, two passes of two column blocks each
  // lane l holds D[tile = 4 kq + i][n = l & 15]: tile = (tz, ty, tx) with (tz, ty) = kq, tx = i
  // [8 w][32 slots = ((i * 4 + dy * 2 + dx) * 2 + jl)][64 lanes + 16]: the final stage reads 16 bytes per lane with
  // lane bits (cq, jl | o | tile) -- with a 64-float slot pitch and jl in a high slot bit every lane of a wave landed in
  // the same four 16-byte bank groups (half the LDS bandwidth); pitch 80 with jl as the lowest slot bit gives the eight
  // lanes of a group eight different ones
  constexpr int PS = 80, PW = 32 * PS;           // floats per slot / per wave: 8 x 2560 x 4 B = 80 KB
  float* part = lds;
  float* red = lds + 8 * PW;                     // 2 x [8 w][64]
  // final-stage task of this thread: cq = tid & 3 (channel quad of a column block), jl = (tid >> 2) & 1, o = (tid >> 3) & 3
  // (= dy * 2 + dx), tile = tid >> 5
  const int cq = tid & 3, jl = (tid >> 2) & 1, o = (tid >> 3) & 3, tile = tid >> 5;
  const int ttz = tile >> 3, tty = (tile >> 2) & 1, ttx = tile & 3;
  const int vz = oz + 2 * ttz, vy = oy + 2 * tty + (o >> 1), vx = ox + 2 * ttx + (o & 1);
  const size_t vox0 = (((size_t)b * S + vz) * S + vy) * S + vx;
  const int slot_rd = ((ttx * 4 + o) * 2 + jl) * PS + (tile >> 2) * 16 + cq * 4;
  vf4 val[2][2];
  vf4 f1[2], f2s[2];
  // packed fp32 (full rate when no MFMA is in flight): subtraction through the neg modifiers, wave-uniform scalars from SGPRs
  typedef float vf2 __attribute__((ext_vector_type(2)));
  auto pk_sub = [](vf2 a, vf2 b) { vf2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; };
  auto sub4 = [&](vf4 a, vf4 b) { const vf2 lo = pk_sub(a.xy, b.xy), hi = pk_sub(a.zw, b.zw); return vf4{lo.x, lo.y, hi.x, hi.y}; };
  auto pk_fma_s = [](vf2 sc, vf2 x, vf2 y) { vf2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "s"(sc), "v"(x), "v"(y)); return r; };
  auto pk_mul_s = [](vf2 sc, vf2 x) { vf2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "s"(sc), "v"(x)); return r; };
  auto fma4s = [&](vf2 sc, vf4 x, vf4 y) { const vf2 lo = pk_fma_s(sc, x.xy, y.xy), hi = pk_fma_s(sc, x.zw, y.zw); return vf4{lo.x, lo.y, hi.x, hi.y}; };
  auto mul4s = [&](vf2 sc, vf4 x) { const vf2 lo = pk_mul_s(sc, x.xy), hi = pk_mul_s(sc, x.zw); return vf4{lo.x, lo.y, hi.x, hi.y}; };
  const float k1f = fyh ? 0.f : 1.f, k2f = fyh ? -1.f : 0.f, k3f = fyh ? -1.f : 1.f;
  const vf2 k1 = {k1f, k1f}, k2 = {k2f, k2f}, k3 = {k3f, k3f}, half2 = {0.5f, 0.5f};
#ifndef ICS_W64_EPI_PREFETCH
#define ICS_W64_EPI_PREFETCH 1   // 1: everything the final stage reads from global memory is requested before the first
#endif                           //    pass, under the output transform, instead of after each pass' second barrier
  static_assert(ICS_W64_EPI_PREFETCH || FOLD < 2, "the FOLD >= 2 epilogues (BatchNorm-backward fold, pooled gradient) exist only "
                                                  "in the prefetching final stage");
#if ICS_W64_EPI_PREFETCH
  vf4 pbias[2], pacc[2][2], psv[2][2], pmu[2], prs[2], pk[FOLD >= 2 ? 2 : 1], ppool[FOLD == 3 ? 2 : 1];
  unsigned pmask[2] = {0u, 0u};
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int nn = n0 + (pass * 2 + jl) * 16 + cq * 4;
    pbias[pass] = vf4{0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) pbias[pass] = *reinterpret_cast<const vf4*>(bias + nn);
    if (accumulate) {
      const size_t o0 = vox0 * ldo + nn;
      pacc[pass][0] = *reinterpret_cast<const vf4*>(y + o0);
      pacc[pass][1] = *reinterpret_cast<const vf4*>(y + o0 + (size_t)S * S * ldo);
    }
    if (FOLD == 3) {
      const int Sh = S >> 1;
      const size_t prow = (((size_t)b * Sh + (vz >> 1)) * Sh + (vy >> 1)) * Sh + (vx >> 1);
      ppool[pass] = *reinterpret_cast<const vf4*>(bs.pool_d + prow * bs.pool_ld + nn);
      pmask[pass] = *reinterpret_cast<const unsigned*>(bs.pool_mask + prow * Cout + nn);
    }
    if (FOLD >= 2) {                               // pmu / prs / pk hold a / b / c
      pmu[pass] = *reinterpret_cast<const vf4*>(bs.abc + nn);
      prs[pass] = *reinterpret_cast<const vf4*>(bs.abc + Cout + nn);
      pk[pass] = *reinterpret_cast<const vf4*>(bs.abc + 2 * Cout + nn);
    } else if (FOLD) {
      pmu[pass] = *reinterpret_cast<const vf4*>(bs.mean + nn);
      prs[pass] = *reinterpret_cast<const vf4*>(bs.rstd + nn);
    }
    if (FOLD) {
      const size_t s0 = vox0 * bs.ld + nn;
      psv[pass][0] = *reinterpret_cast<const vf4*>(bs.s + s0);
      psv[pass][1] = *reinterpret_cast<const vf4*>(bs.s + s0 + (size_t)S * S * bs.ld);
    }
  }
#endif
#ifndef ICS_W64_EPI_EARLY
#define ICS_W64_EPI_EARLY 0     // 1: a pass' transform arithmetic ahead of the barrier that frees the LDS region it is written to
                                // (a wave that leaves the main loop early would use its wait): measured +-0 on all nine
                                // layers, forward and backward-data -- its VALU work competes with the late wave's MFMAs
#endif
  vf4 dd[2][2][2];              // [jj][dx][dy row]: the wave's share of the output transform of one pass
  auto xf = [&](const int pass) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = pass * 2 + jj;
      // Whole accumulator quads (the four tiles i of a lane) at a time: the epilogue is VALU issue time on two waves per
      // SIMD (575 instructions after the last MFMA when this was written element by element), and on float4 values the
      // compiler selects v_pk_add_f32 -- full rate when no MFMA is in flight.
      vf4 qv[2][2];                              // [fy local][dx]
#pragma unroll
      for (int fy = 0; fy < 2; ++fy) {
        qv[fy][0] = acc[fy * 4 + 0][j] + acc[fy * 4 + 1][j] + acc[fy * 4 + 2][j];
        qv[fy][1] = sub4(sub4(acc[fy * 4 + 1][j], acc[fy * 4 + 2][j]), acc[fy * 4 + 3][j]);
      }
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        // rows of A^T: the wave with fy 0,1 gives dy0 = q0 + q1, dy1 = q1; the one with fy 2,3: dy0 = q0, dy1 = -(q0 + q1)
        // -- as d0 = q0 + k1 q1, d1 = k2 q0 + k3 q1 with wave-uniform coefficients in SGPR pairs (the selects and
        // negations were 112 VALU instructions per thread)
        dd[jj][dx][0] = fma4s(k1, qv[1][dx], qv[0][dx]);
        dd[jj][dx][1] = fma4s(k3, qv[1][dx], mul4s(k2, qv[0][dx]));
      }
    }
  };
#if ICS_W64_EPI_EARLY
  xf(0);                        // a wave that leaves the main loop early does this while the last one still runs MFMAs
#endif
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#if !ICS_W64_EPI_EARLY
    xf(pass);
#endif
    __syncthreads();
    ICS_TL(4 + pass * 3);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          part[w * PW + ((i * 4 + 0 + dx) * 2 + jj) * PS + lane] = dd[jj][dx][0][i];
          part[w * PW + ((i * 4 + 2 + dx) * 2 + jj) * PS + lane] = dd[jj][dx][1][i];
        }
#if ICS_W64_EPI_EARLY
    if (pass == 0) xf(1);
#endif
    __syncthreads();
    ICS_TL(5 + pass * 3);
    const int nn = n0 + (pass * 2 + jl) * 16 + cq * 4;             // this thread's four output channels
    vf4 p[4];
#pragma unroll
    for (int z = 0; z < 4; ++z)
      p[z] = *reinterpret_cast<const vf4*>(&part[(2 * z) * PW + slot_rd]) +
             *reinterpret_cast<const vf4*>(&part[(2 * z + 1) * PW + slot_rd]);
    const size_t o0 = vox0 * ldo + nn;
    const size_t o1 = o0 + (size_t)S * S * ldo;
#if ICS_W64_EPI_PREFETCH
    const vf4 bv = pbias[pass];
    vf4 e0 = p[0] + p[1] + p[2] + bv, e1 = sub4(sub4(p[1], p[2]), p[3]) + bv;
    if (accumulate) { e0 += pacc[pass][0]; e1 += pacc[pass][1]; }
#else
    vf4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) bv = *reinterpret_cast<const vf4*>(bias + nn);
    vf4 e0 = p[0] + p[1] + p[2] + bv, e1 = sub4(sub4(p[1], p[2]), p[3]) + bv;
    if (accumulate) {
      e0 += *reinterpret_cast<const vf4*>(y + o0);
      e1 += *reinterpret_cast<const vf4*>(y + o1);
    }
#endif
    e0.x = wact(e0.x, pre_slope); e0.y = wact(e0.y, pre_slope); e0.z = wact(e0.z, pre_slope); e0.w = wact(e0.w, pre_slope);
    e1.x = wact(e1.x, pre_slope); e1.y = wact(e1.y, pre_slope); e1.z = wact(e1.z, pre_slope); e1.w = wact(e1.w, pre_slope);
#if ICS_W64_EPI_PREFETCH
    if (FOLD == 3) {                               // vz is even: the voxel pair (vz, vz + 1) shares its pooling window
      const int k0 = ((vy & 1) << 1) | (vx & 1);
      const vf4 pd = ppool[pass];
      const unsigned m0 = pmask[pass] >> k0, m1 = m0 >> 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        e0[r] += ((m0 >> (8 * r)) & 1u) ? pd[r] : 0.f;
        e1[r] += ((m1 >> (8 * r)) & 1u) ? pd[r] : 0.f;
      }
    }
    if (FOLD >= 2) {
      const vf4 ka = pmu[pass], kb = prs[pass], kc = pk[pass], sv0 = psv[pass][0], sv1 = psv[pass][1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        e0[r] = sv0[r] > 0.f ? fmaf(ka[r], e0[r], fmaf(kb[r], sv0[r], kc[r])) : 0.f;
        e1[r] = sv1[r] > 0.f ? fmaf(ka[r], e1[r], fmaf(kb[r], sv1[r], kc[r])) : 0.f;
      }
      f1[pass] = e0 + e1;
    }
#endif
    *reinterpret_cast<vf4*>(y + o0) = e0;
    *reinterpret_cast<vf4*>(y + o1) = e1;
    val[pass][0] = e0; val[pass][1] = e1;
    ICS_TL(6 + pass * 3);
    if (FOLD == 1) {
#if ICS_W64_EPI_PREFETCH
      const vf4 b_mu = pmu[pass], b_rs = prs[pass], sv0 = psv[pass][0], sv1 = psv[pass][1];
#else
      const vf4 b_mu = *reinterpret_cast<const vf4*>(bs.mean + nn), b_rs = *reinterpret_cast<const vf4*>(bs.rstd + nn);
      const size_t s0 = vox0 * bs.ld + nn;
      const vf4 sv0 = *reinterpret_cast<const vf4*>(bs.s + s0);
      const vf4 sv1 = *reinterpret_cast<const vf4*>(bs.s + s0 + (size_t)S * S * bs.ld);
#endif
      vf4 d0 = e0, d1 = e1;
      if (bs.post_act != ACT_NONE) {
        const vf4 b_sc = *reinterpret_cast<const vf4*>(bs.scale + nn), b_sh = *reinterpret_cast<const vf4*>(bs.shift + nn);
        const vf4 z0 = sv0 * b_sc + b_sh, z1 = sv1 * b_sc + b_sh;
        d0.x *= act_grad(z0.x, bs.post_act); d0.y *= act_grad(z0.y, bs.post_act);
        d0.z *= act_grad(z0.z, bs.post_act); d0.w *= act_grad(z0.w, bs.post_act);
        d1.x *= act_grad(z1.x, bs.post_act); d1.y *= act_grad(z1.y, bs.post_act);
        d1.z *= act_grad(z1.z, bs.post_act); d1.w *= act_grad(z1.w, bs.post_act);
      }
      f1[pass] = d0 + d1;
      f2s[pass] = d0 * ((sv0 - b_mu) * b_rs) + d1 * ((sv1 - b_mu) * b_rs);
    }
  }
  // column reductions over the block's 128 voxels: a column quad (pass, jl, cq) lives in the threads with the same
  // (tid & 7): three xor-shuffle steps inside the wave (16 voxels), then the eight waves through LDS.
  // red[0..512) / red[512..1024): [w][64], channel index (pass * 2 + jl) * 16 + cq * 4 + e
  const int cidx = jl * 16 + cq * 4;
  const size_t nstat = gridDim.x / nchunks;
#ifndef ICS_W64_STATS_LDS
#define ICS_W64_STATS_LDS 1   // 1: per-thread values through LDS, merged by 256 + 64 threads; 0: xor-shuffle trees
#endif
#if ICS_W64_STATS_LDS
  // (see the statistics epilogue below: the xor-shuffle trees were VALU issue time on two waves per SIMD)
  float* smf = lds + 8 * PW + 1024;              // [64 voxel pairs][16 channel quads] float4 (x 2 quantities): 32 KB
  if (FOLD) {
    const int e = tid >> 3;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int qd = pass * 8 + (tid & 7);
      *reinterpret_cast<vf4*>(&smf[(e * 16 + qd) * 4]) = f1[pass];
      if (FOLD == 1) *reinterpret_cast<vf4*>(&smf[4096 + (e * 16 + qd) * 4]) = f2s[pass];
    }
    __syncthreads();
    if (tid < 256) {
      const int c = tid & 63, sub = tid >> 6;
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        a += smf[(sub * 16 + i) * 64 + c];
        if (FOLD == 1) q += smf[4096 + (sub * 16 + i) * 64 + c];
      }
      red[sub * 64 + c] = a;
      if (FOLD == 1) red[512 + sub * 64 + c] = q;
    }
    __syncthreads();
    if (FOLD >= 2) {                               // column sums of the written dy_P only: [blocks][Cout]
      if (tid < 64) bs.db_partial[(size_t)tblk * Cout + n0 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
    } else if (tid < 128) {                        // [2][Npad][blocks] (block index fastest), as conv_igemm.hip's FOLD
      const int c = tid & 63, which = tid >> 6;
      const float* r = red + which * 512;
      bs.partial[((size_t)which * Npad + n0 + c) * nstat + tblk] = (r[c] + r[64 + c]) + (r[128 + c] + r[192 + c]);
    }
    ICS_TL_FLUSH();
    return;
  }
#else
  if (FOLD >= 2) {                                 // column sums of the written dy_P only: [blocks][Cout]
#pragma unroll
    for (int d = 8; d < 64; d <<= 1)
#pragma unroll
      for (int pass = 0; pass < 2; ++pass)
#pragma unroll
        for (int e = 0; e < 4; ++e) f1[pass][e] += __shfl_xor(f1[pass][e], d);
    if (lane < 8) {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) *reinterpret_cast<vf4*>(&red[w * 64 + pass * 32 + cidx]) = f1[pass];
    }
    __syncthreads();
    if (tid < 64) {
      float sacc = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) sacc += red[ww * 64 + tid];
      bs.db_partial[(size_t)tblk * Cout + n0 + tid] = sacc;
    }
    ICS_TL_FLUSH();
    return;
  }
  if (FOLD) {                                      // [2][Npad][blocks] (block index fastest), as conv_igemm.hip's FOLD
#pragma unroll
    for (int d = 8; d < 64; d <<= 1)
#pragma unroll
      for (int pass = 0; pass < 2; ++pass)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f1[pass][e] += __shfl_xor(f1[pass][e], d);
          f2s[pass][e] += __shfl_xor(f2s[pass][e], d);
        }
    if (lane < 8) {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        *reinterpret_cast<vf4*>(&red[w * 64 + pass * 32 + cidx]) = f1[pass];
        *reinterpret_cast<vf4*>(&red[512 + w * 64 + pass * 32 + cidx]) = f2s[pass];
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int c = tid & 63, which = tid >> 6;
      float sacc = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) sacc += red[which * 512 + ww * 64 + c];
      bs.partial[((size_t)which * Npad + n0 + c) * nstat + tblk] = sacc;
    }
    ICS_TL_FLUSH();
    return;
  }
#endif
  if (stat_partial == nullptr) { ICS_TL_FLUSH(); return; }

  // block-level (count, mean, M2) per column (conv_igemm.hip's layout [3][Npad][nblocks], block index fastest).
#ifndef ICS_W64_STATS_LDS
#define ICS_W64_STATS_LDS 1   // 1: pair statistics through LDS, merged by 256 + 64 threads; 0: the xor-shuffle Chan tree
#endif
#if ICS_W64_STATS_LDS
  // The statistics epilogue cost every forward launch 1.5 us per workgroup (0.30 ms per U-Net step over the nine layers;
  // the BatchNorm-affine source costs nothing: scripts/wino_bench.py WINO_FEAT=8 / 16) -- almost all of it VALU issue on
  // two waves per SIMD: 48 xor-shuffles with their address arithmetic and 40 Chan merges per thread.  Now a thread only
  // forms the exact (mean, M2) of its two voxels per channel and parks them: [64 voxel pairs][16 channel quads] float4,
  // behind the output-transform buffers (no barrier needed in front).  Four waves then merge 16 pairs each per channel
  // (mean of means, M2 = sum M2_i + 2 sum (mean_i - m)^2: the same two-level form as before, no cancellation), 64 threads
  // merge the four partials.
  float* sm = lds + 8 * PW + 1024;               // [64][16] float4 means | [64][16] float4 M2: 32 KB
  {
    const int e = tid >> 3;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const vf4 dlt = sub4(val[pass][1], val[pass][0]);
      const vf4 hd = mul4s(half2, dlt);
      const vf4 mnv = val[pass][0] + hd;
      const vf4 m2v = hd * dlt;
      const int qd = pass * 8 + (tid & 7);
      *reinterpret_cast<vf4*>(&sm[(e * 16 + qd) * 4]) = mnv;
      *reinterpret_cast<vf4*>(&sm[4096 + (e * 16 + qd) * 4]) = m2v;
    }
  }
  __syncthreads();
  if (tid < 256) {
    const int c = tid & 63, sub = tid >> 6;
    float mi[16], msum = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      mi[i] = sm[(sub * 16 + i) * 64 + c];
      msum += mi[i];
      q += sm[4096 + (sub * 16 + i) * 64 + c];
    }
    const float m = msum * (1.f / 16.f);
    float dev = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) dev += (mi[i] - m) * (mi[i] - m);
    red[sub * 64 + c] = m;
    red[512 + sub * 64 + c] = q + 2.f * dev;       // 32 voxels
  }
  __syncthreads();
  if (tid < 64) {
    float mw[4], msum = 0.f, q = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { mw[ww] = red[ww * 64 + tid]; msum += mw[ww]; q += red[512 + ww * 64 + tid]; }
    const float mean_t = msum * 0.25f;
    float dev = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) dev += (mw[ww] - mean_t) * (mw[ww] - mean_t);
    float* sp = stat_partial + (size_t)(n0 + tid) * nstat + tblk;
    sp[0] = (float)kRows;
    sp[(size_t)Npad * nstat] = mean_t;
    sp[(size_t)2 * Npad * nstat] = q + dev * (float)(kRows / 4);
  }
#else
  // a tree of equal-count Chan merges: thread (2 voxels) -> wave (16) -> block (128); one trip through LDS
  vf4 mn[2], m2[2];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const vf4 dlt = val[pass][1] - val[pass][0];
    mn[pass] = val[pass][0] + 0.5f * dlt;
    m2[pass] = 0.5f * dlt * dlt;
  }
  float nh = 1.f;                                  // half the element count of the groups being merged
#pragma unroll
  for (int d = 8; d < 64; d <<= 1) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float mo = __shfl_xor(mn[pass][e], d), qo = __shfl_xor(m2[pass][e], d);
        const float dl = mo - mn[pass][e];
        m2[pass][e] = m2[pass][e] + qo + dl * dl * nh;
        mn[pass][e] = mn[pass][e] + 0.5f * dl;
      }
    nh *= 2.f;
  }
  if (lane < 8) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      *reinterpret_cast<vf4*>(&red[w * 64 + pass * 32 + cidx]) = mn[pass];
      *reinterpret_cast<vf4*>(&red[512 + w * 64 + pass * 32 + cidx]) = m2[pass];
    }
  }
  __syncthreads();
  if (tid < 64) {
    float mw[8], msum = 0.f, q = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) { mw[ww] = red[ww * 64 + tid]; msum += mw[ww]; q += red[512 + ww * 64 + tid]; }
    const float mean_t = msum * 0.125f;
    float dev = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) dev += (mw[ww] - mean_t) * (mw[ww] - mean_t);
    float* sp = stat_partial + (size_t)(n0 + tid) * nstat + tblk;
    sp[0] = (float)kRows;
    sp[(size_t)Npad * nstat] = mean_t;
    sp[(size_t)2 * Npad * nstat] = q + dev * (float)(kRows / 8);
  }
#endif
  ICS_TL_FLUSH();
}

// ---------------------------------------------------------------- host side
bool conv_wino64_ok(const ConvGeom& g, const ConvSrc* src, int nsrc) {
  if (g.flags & (CF_NO_WINO | CF_NO_WINO64)) return false;
  if (g.taps != 27 || nsrc != 1 || g.S < 8 || g.lgS < 3) return false;
  const ConvSrc& s = src[0];
  if (s.up || s.bcast || s.C != g.Cin) return false;
  if (g.Cin % KC != 0 || g.Cout % 64 != 0) return false;
  if ((long long)g.B * g.S * g.S * g.S * (long long)std::max(g.Cin, g.Cout) >= (1ll << 29)) return false;  // 32-bit BYTE offsets
  return true;
}

int launch_conv_fwd_wino64(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* wt, const float* bias,
                           float* out, int ldo, int pre_act, float* stat_partial, int* rows_per_block, int accumulate,
                           const BwdStat* bwd, int* bwd_blocks) {
  ICS_CHECK(conv_wino64_ok(g, &s0, 1), "shape not served by the 64-channel Winograd kernel");
  ICS_CHECK(ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(bias) & 15) == 0 && (reinterpret_cast<uintptr_t>(s0.p) & 15) == 0,
            "Winograd kernel: float4 accesses need 16-byte aligned tensors");
  const unsigned grid = (unsigned)(g.B * (g.S / 4) * (g.S / 4) * (g.S / 8) * (g.Cout / 64));
  if (rows_per_block) *rows_per_block = kRows;
  const bool aff = s0.scale != nullptr;
  const bool noact = s0.act == ACT_NONE;
  const float in_slope = wslope(s0.act), pre_slope = wslope(pre_act);
  const bool plain = bwd != nullptr && !aff && stat_partial == nullptr && bias == nullptr && pre_act == ACT_NONE &&
                     !accumulate && bwd->ld % 4 == 0;
  const bool apply = plain && bwd->abc != nullptr;           // the producer's BatchNorm-backward apply in the epilogue
  const bool pool = apply && bwd->pool_d != nullptr;
  ICS_CHECK(!pool || (bwd->pool_mask != nullptr && bwd->pool_ld % 4 == 0 && g.S % 2 == 0),
            "Winograd backward-data with a pooled second gradient source: mask / leading dimension");
  ICS_CHECK(bwd == nullptr || bwd->abc == nullptr || (apply && bwd->db_partial != nullptr && bwd->s != nullptr),
            "Winograd backward-data with the fused BatchNorm-backward apply: not a plain launch");
  const bool fold = plain && !apply && bwd->partial != nullptr;
  if (bwd_blocks) *bwd_blocks = (fold || apply) ? (int)(grid / (unsigned)(g.Cout / 64)) : 0;
  const BwdStat bs = (fold || apply) ? *bwd : BwdStat{};
#define ICS_WINO_LAUNCH(AFFV, NOACTV, FOLDV)                                                                          \
  do {                                                                                                                \
    ICS_LAUNCH((conv_wino64_kernel<AFFV, NOACTV, FOLDV>), dim3(grid), dim3(512), 0, st, s0.p, s0.C, s0.scale, \
                       s0.shift, in_slope, wt, bias, out, ldo, pre_slope, accumulate, stat_partial, g.Npad, g.S,      \
                       g.Cin, g.Cout, bs);                                                                            \
    conv_set_last_kernel_id("conv_wino64_kernel<" #AFFV ", " #NOACTV ", " #FOLDV ">");                                \
  } while (0)
  if (pool) ICS_WINO_LAUNCH(false, true, 3);
  else if (apply) ICS_WINO_LAUNCH(false, true, 2);
  else if (fold) ICS_WINO_LAUNCH(false, true, 1);
  else if (!aff) ICS_WINO_LAUNCH(false, true, 0);
  else if (noact) ICS_WINO_LAUNCH(true, true, 0);
  else ICS_WINO_LAUNCH(true, false, 0);
#undef ICS_WINO_LAUNCH
  ICS_HIP(hipGetLastError());
  return 0;
}

}  // namespace ics
