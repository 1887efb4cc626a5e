// 3x3x3 "same" convolution over a nearest-upsampled input with 27 instead of 64 multiplies per low-res voxel
// (gfx950 / MI355X only) -- the upsampled channels of the U-Net's [skip | UpSampling3D(x)] convolutions
// (/root/reference/unet/unet.py:309-336) and the VAE decoder's upsampled layers (vae/lattice_vae.py:211-217).
//
// Along one axis the fine outputs 2a, 2a+1 of low-res position a see x[a-1], x[a], x[a+1]:
//   y[2a]   = W-1 x[a-1] + (W0 + W+1) x[a]   = P0 + P1        P0 = W-1 (x[a-1] - x[a])
//   y[2a+1] = (W-1 + W0) x[a] + W+1 x[a+1]   = P1 + P2        P1 = (W-1 + W0 + W+1) x[a]      P2 = W+1 (x[a+1] - x[a])
// three products instead of the four of the parity-class form (conv_igemm.hip's PAR kernels: 8 GEMMs with K = 8 Cu).
// In three dimensions: 27 "frequencies" (fz, fy, fx), a transformed input D = (B (x) B (x) B) x over the 3x3x3 low-res
// neighbourhood with B = [1 -1 0; 0 1 0; 0 -1 1], weights G = (g (x) g (x) g) w with g = [1 0 0; 1 1 1; 0 0 1], 27 GEMMs
// voxels x Cu x Cout, and Y = (A (x) A (x) A) P with A = [1 1 0; 0 1 1] -- exact in exact arithmetic, the same
// structure as the Winograd kernels (conv_wino64.hip) with 3 instead of 4 points per axis and tiles one low-res voxel
// apart.  Zero padding of the fine grid is zero padding of the low-res grid.
//
// One workgroup = 32 low-res voxels (2x4x4: an 4x8x8 block of fine outputs) x 64 output channels x 27 frequencies on
// v_mfma_f32_16x16x4_f32 (lane l: A[voxel l & 15][k = l >> 4]; the two 16-voxel halves are two y rows apart and share
// every B operand), EIGHT waves, two per SIMD: wave w owns the (fz, fy) pair w with fx = 0..2 and all four 16-channel
// column blocks (2 x 12 accumulators of 4 registers), and the 24 tiles of the ninth pair (2, 2) are dealt over all eight
// waves -- column block w & 3 of voxel half w >> 2, fx = 0..2: 24 + 3 = 27 MFMAs per wave and sub-step, 54 on every SIMD
// (the kernel header below has the history).  Staging: thread t < 288 owns (y, x, channel quad) of the halo [4][6][6] x
// 32 channels, in two phases of three z rows (12 registers in flight): the producer's BatchNorm affine + activation,
// zero padding, the z rows of B, three planes out per phase.  LDS: voxel pitch 34 floats, row pitch 208, plane pitch
// 1256 (3 x 1256 = 24 mod 32 as with the 16-voxel tile's 840: conflict-free ds_read_b32 for every (fz, row, column,
// sub-step), checked exhaustively).  Weights [Cout/64][Cu/4][27 f][4 k][16 n][4 column blocks]; the fy = 1 frequencies
// carry the opposite sign (one fma per column in the y transform).  The 16-voxel tile (NH = 1, small launches) keeps the
// ninth pair on the waves 0..3 (column block w) and the four-row staging.
#include "common.h"

#include <cstdlib>
#include <type_traits>

#include <algorithm>

namespace ics {

typedef float uf4 __attribute__((ext_vector_type(4)));
typedef float uf2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int KC = 32;                                   // input channels per LDS chunk
#ifndef ICS_UP3_BIG_MIN_WG
#define ICS_UP3_BIG_MIN_WG 256                           // workgroups of the 32-voxel tile below which the 16-voxel one runs
#endif
constexpr int VX = 34, RP = 208;                         // floats; plane pitch and buffer size depend on the tile (kernel)

__device__ __forceinline__ float uact(float v, float slope) { return fmaxf(v, v * slope); }
__host__ __device__ __forceinline__ float uslope(int act) { return act == ACT_RELU ? 0.f : (act == ACT_LRELU ? kLeaky : 1.f); }
}  // namespace

// x: low-res source [B][Sl][Sl][Sl][ldx]; y: fine output [B][2 Sl]^3 [ldo].  AFF / NOACT as in conv_wino64.hip.
// STATS: per-block BatchNorm statistics of the stored values (the VAE decoder's layers; the U-Net's launches leave bias,
// activation and statistics to the skip-channel pass that accumulates on top).
// History of the wave layout (c17.up, ms per U-Net step): nine waves, one pair each, sat 3 + 2 + 2 + 2 on the SIMDs and
// the main loop ran at the pace of SIMD 0's 36 MFMAs per sub-step (1.36; per-wave time stamps in an instrumented build, and an ablation without
// weight loads and operand reads still took 1.24); twelve waves x three triples 1.36 (two operand sets per wave); the
// layout below with the extra work under wave-uniform branches INSIDE the loop 1.39 (conservative waits at every join);
// with the two roles as two straight-line copies of the loop behind one branch: 1.24; 32 instead of 16 voxels per
// workgroup on top (half the weight traffic, halo 4.5 instead of 6 positions per voxel, one epilogue per 32): 1.15 --
// but the four waves with the whole ninth pair (24 more accumulator registers) spilled the four staged rows right behind
// their loads and waited for HBM four times per chunk; the ninth pair dealt over ALL eight waves (one loop, one role) and
// the staging in two three-row phases: no spill, 1.06.
// NH: 16-voxel halves per workgroup -- 2 (2x4x4 low-res voxels) or 1 (2x2x4, for launches that would not fill the chip
// with the large tile: the S = 4 layers and the VAE decoder).
template <bool AFF, bool NOACT, bool STATS, int NH>
__global__ __launch_bounds__(512) void conv_up3_kernel(const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ in_scale,
                                                          const float* __restrict__ in_shift, float in_slope,
                                                          const float* __restrict__ wt, const float* __restrict__ bias,
                                                          float* __restrict__ y, int ldo, float pre_slope, int accumulate,
                                                          float* __restrict__ stat_partial, int Npad, int Sl, int Cin,
                                                          int Cout) {
  constexpr int HY = 2 + 2 * NH;                 // halo rows in y
  constexpr int PP = HY * RP + 8, BUF = 6 * PP;  // plane pitch 1256 / 840: 3 PP = 24 mod 32 either way
  constexpr int NST = HY * 6 * 8;                // staging threads: 288 / 192
  constexpr int kRows = 128 * NH;                // fine voxels per workgroup
  // NH = 2: 92 160 B -- two buffers (60 288 B), the epilogue's [9][32][80]; NH = 1: 46 080 B
  __shared__ __attribute__((aligned(16))) float lds[9 * 16 * NH * 80];
  __shared__ unsigned park[3 * 512];
  __shared__ __attribute__((aligned(16))) float aff[AFF ? 2048 : 4];
  __shared__ float red[9 * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // EIGHT waves, two per SIMD.  Wave w owns the (fz, fy) pair w (fx = 0..2, four column blocks); the ninth pair (2, 2)
  // is split over the waves so that every SIMD issues the same number of MFMAs per sub-step.  (Nine waves sat 3 + 2 + 2
  // + 2: the main loop ran at the pace of SIMD 0's 36: per-wave time stamps.)  Where only some waves carry extra tiles
  // (NH = 1) the two roles are two straight-line copies of the main loop behind ONE wave-uniform branch: branches around
  // the extra MFMAs inside the loop cost all of the gain (conservative waits at every join).
  const int fz = w / 3, fy = w - 3 * fz;
  // NH = 2: EVERY wave takes three tiles of the ninth pair -- column block w & 3 of voxel half w >> 2 (the 24 tiles of the
  // pair over eight waves); NH = 1: the waves 0..3, column block w
  const bool extra = NH == 2 || w < 4;
  const int xh = NH == 2 ? (w >> 2) : 0;         // voxel half of this wave's extra tiles
  const int m = lane & 15, kq = lane >> 4;
  const int nchunks = Cout >> 6;
  const int nb = blockIdx.x % nchunks;
  const int tblk = blockIdx.x / nchunks;
  int tb = tblk;
  const int nbx = Sl >> 2, nby = Sl >> NH, nbz = Sl >> 1;
  const int bx = tb % nbx; tb /= nbx;
  const int by = tb % nby; tb /= nby;
  const int bz = tb % nbz;
  const int b = tb / nbz;
  const int oz = bz * 2, oy = by * 2 * NH, ox = bx * 4, n0 = nb * 64; // low-res origin of the block (2 x 2 NH x 4 voxels)
  const bool edge = bx == 0 || by == 0 || bz == 0 || bx == nbx - 1 || by == nby - 1 || bz == nbz - 1;   // uniform
  const int S = 2 * Sl;

  // ---- staging: thread t < NST owns (hy, hx, channel quad) of the halo [4][HY][6]: four z rows in, six planes out
  const int cmb = tid < NST ? tid : NST - 1;
  const int q = cmb & 7, hx = (cmb >> 3) % 6, hy = (cmb >> 3) / 6;
  uf4 hs[4];
  unsigned zoff[4];                              // uniform: byte offset of sample b, low-res plane clamp(oz - 1 + hz)
  unsigned okz = 0;                              // uniform: bit hz = plane inside the grid
  {
    const int gy = oy - 1 + hy, gx = ox - 1 + hx;
    const int cy = min(max(gy, 0), Sl - 1), cx = min(max(gx, 0), Sl - 1);
    const bool okyx = gy == cy && gx == cx;
    park[tid] = (unsigned)((cy * Sl + cx) * ldx + q * 4) * 4u;
    park[512 + tid] = (unsigned)(hy * RP + hx * VX + q * 4);
    park[1024 + tid] = (unsigned)(q * 4) | ((okyx ? 1u : 0u) << 8);
#pragma unroll
    for (int hz = 0; hz < 4; ++hz) {
      const int gz = oz - 1 + hz, cz = min(max(gz, 0), Sl - 1);
      okz |= gz == cz ? (1u << hz) : 0u;
      zoff[hz] = (unsigned)((b * Sl + cz) * Sl * Sl * ldx) * 4u;
    }
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x7fffffff, 0x00020000);
  // per-thread staging constants parked in LDS (conv_wino64.hip explains why)
  const unsigned park_addr = (unsigned)(uintptr_t)&park[tid];
  auto unpark = [&](const int which) -> int {
    int v;
    asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(park_addr), "i"(which * 2048) : "memory");
    return v;
  };
  // (the producer's scale / shift table is requested together with the first halo rows, below: as a load -> wait -> store
  // loop up here it put a full memory round trip in front of every other load of the prologue)
  auto hload = [&](int c0) {
    const int vo = unpark(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      hs[i] = __builtin_bit_cast(uf4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, (int)(zoff[i] + (unsigned)c0 * 4u), 0));
  };
  auto hstore = [&](const int bo, const int c0) {
    // two channels at a time (few temporaries); every z row gets the affine ONCE although rows 1, 2 feed both tile z
    const int pk2 = unpark(2);
    const int q4 = pk2 & 255;
    const bool okyx = ((pk2 >> 8) & 1) != 0;
    float* o = &lds[bo + unpark(1)];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      uf2 r[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = uf2{hs[i][2 * hh], hs[i][2 * hh + 1]};
      if (AFF) {
        const uf2 sc2 = *reinterpret_cast<const uf2*>(&aff[c0 + q4 + 2 * hh]);
        const uf2 sh2 = *reinterpret_cast<const uf2*>(&aff[1024 + c0 + q4 + 2 * hh]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float t0 = fmaf(r[i].x, sc2.x, sh2.x), t1 = fmaf(r[i].y, sc2.y, sh2.y);
          if (!NOACT) { t0 = uact(t0, in_slope); t1 = uact(t1, in_slope); }
          r[i] = uf2{t0, t1};
        }
      }
      if (edge) {                                // zero padding AFTER the producer's affine / activation
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (!(okyx && ((okz >> i) & 1))) r[i] = uf2{0.f, 0.f};
      }
      if (tid < NST) {
#pragma unroll
        for (int tzq = 0; tzq < 2; ++tzq) {       // tile z: rows tzq, tzq + 1, tzq + 2
          *reinterpret_cast<uf2*>(o + (tzq * 3 + 0) * PP + 2 * hh) = r[tzq] - r[tzq + 1];
          *reinterpret_cast<uf2*>(o + (tzq * 3 + 1) * PP + 2 * hh) = r[tzq + 1];
          *reinterpret_cast<uf2*>(o + (tzq * 3 + 2) * PP + 2 * hh) = r[tzq + 2] - r[tzq + 1];
        }
      }
    }
  };

  // NH = 2, main loop: the same staging in TWO phases of three z rows -- rows 0..2 -> the planes of tile z 0, rows 1..3 ->
  // tile z 1 (rows 1, 2 come a second time, from L1 / L2) -- so that 12 instead of 16 registers are in flight next to the
  // 132 accumulator registers: with four rows the allocator spilled one right behind its load, i.e. every wave waited for
  // HBM once per chunk (and the four waves that carried the whole ninth pair in the first version: four times)
  auto hloadP = [&](const int tzq, int c0) {
    const int vo = unpark(0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
      hs[i] = __builtin_bit_cast(uf4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, (int)(zoff[tzq + i] + (unsigned)c0 * 4u), 0));
  };
  auto hstoreP = [&](const int tzq, const int bo, const int c0) {
    const int pk2 = unpark(2);
    const int q4 = pk2 & 255;
    const bool okyx = ((pk2 >> 8) & 1) != 0;
    float* o = &lds[bo + unpark(1)];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      uf2 r[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) r[i] = uf2{hs[i][2 * hh], hs[i][2 * hh + 1]};
      if (AFF) {
        const uf2 sc2 = *reinterpret_cast<const uf2*>(&aff[c0 + q4 + 2 * hh]);
        const uf2 sh2 = *reinterpret_cast<const uf2*>(&aff[1024 + c0 + q4 + 2 * hh]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          float t0 = fmaf(r[i].x, sc2.x, sh2.x), t1 = fmaf(r[i].y, sc2.y, sh2.y);
          if (!NOACT) { t0 = uact(t0, in_slope); t1 = uact(t1, in_slope); }
          r[i] = uf2{t0, t1};
        }
      }
      if (edge) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (!(okyx && ((okz >> (tzq + i)) & 1))) r[i] = uf2{0.f, 0.f};
      }
      if (tid < NST) {
        *reinterpret_cast<uf2*>(o + (tzq * 3 + 0) * PP + 2 * hh) = r[0] - r[1];
        *reinterpret_cast<uf2*>(o + (tzq * 3 + 1) * PP + 2 * hh) = r[1];
        *reinterpret_cast<uf2*>(o + (tzq * 3 + 2) * PP + 2 * hh) = r[2] - r[1];
      }
    }
  };

  // ---- per-lane read geometry: voxel m = (tz, ty, tx); rows of the wave's fy: D0 = r0 - r1, D1 = r1, D2 = r2 - r1
  const int tz = m >> 3, ty = (m >> 2) & 1, tx = m & 3;
  // t = sa * qa - qb: ONE fma per column.  For fy = 1 that is -r1 instead of r1: the packed weights of the fy = 1
  // frequencies carry the opposite sign (pack_up3_pair), the products are unchanged
  const float sa = fy == 1 ? 0.f : 1.f;
  const int R0 = (tz * 3 + fz) * PP + ty * RP + tx * VX + kq;
  int Ra = R0 + (fy == 2 ? 2 : 0) * RP, Rb = R0 + RP;
  int Rxa = (tz * 3 + 2) * PP + ty * RP + tx * VX + kq + 2 * RP + xh * 2 * RP, Rxb = Rxa - RP;   // pair (2, 2): rows 2 and 1 (of half xh)

  const int nsub = Cin >> 2;
  constexpr int wstride_f = 256;                 // floats per frequency of one sub-step: [4 k][16 n][4 column blocks]
  constexpr int wsub = 27 * 256;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wt + ((size_t)nb * nsub * 27 + (fz * 3 + fy) * 3) * 256), 0, 0x7fffffff, 0x00020000);
  const int wlane = lane * 16;                   // bytes
  auto wload = [&](int gs, int f) {
    return __builtin_bit_cast(uf4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane, (gs * wsub + f * wstride_f) * 4, 0));
  };
  uf4 wreg[3];
#pragma unroll
  for (int f = 0; f < 3; ++f) wreg[f] = wload(0, f);
  // the extra triple's weights: frequency (2, 2, fx) = 24 + fx relative to frequency 0, column block w: one dword per lane
  const __amdgpu_buffer_rsrc_t wrx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wt + ((size_t)nb * nsub * 27 + 24) * 256 + (w & 3)), 0, 0x7fffffff, 0x00020000);
  auto wloadx = [&](int gs, int f) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrx, wlane, (gs * wsub + f * wstride_f) * 4, 0));
  };
  float wx[3];
#pragma unroll
  for (int f = 0; f < 3; ++f) wx[f] = wloadx(0, f);
  uf4 accx[3];
#pragma unroll
  for (int f = 0; f < 3; ++f) accx[f] = uf4{0.f, 0.f, 0.f, 0.f};
  float ux[3], tx3[3], qxa, qxb;

  uf4 acc[NH][3][4];                             // [voxel half][fx][column block]
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int f = 0; f < 3; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][f][j] = uf4{0.f, 0.f, 0.f, 0.f};

  float u[NH][3], tn[NH][3], qa[NH], qb[NH];
  auto rd = [&](const int sub, const int col, auto xt) {
    const int off = col * VX + 4 * sub;          // compile-time after unrolling
#pragma unroll
    for (int h = 0; h < NH; ++h) {                // second half: two low-res rows further
      qa[h] = lds[Ra + off + h * 2 * RP]; qb[h] = lds[Rb + off + h * 2 * RP];
    }
    if (decltype(xt)::value) { qxa = lds[Rxa + off]; qxb = lds[Rxb + off]; }
  };
  auto tstep = [&](const int g, auto xt) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      tn[h][g] = fmaf(sa, qa[h], -qb[h]);
    }
    if (decltype(xt)::value) tx3[g] = qxa - qxb;              // fy = 2: r2 - r1
  };
  auto xform = [&](auto xt) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      u[h][0] = tn[h][0] - tn[h][1]; u[h][1] = tn[h][1]; u[h][2] = tn[h][2] - tn[h][1];
    }
    if (decltype(xt)::value) { ux[0] = tx3[0] - tx3[1]; ux[1] = tx3[1]; ux[2] = tx3[2] - tx3[1]; }
  };

  float a_sc[2] = {0.f, 0.f}, a_sh[2] = {0.f, 0.f};
  if (AFF) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + 512 * k < Cin) { a_sc[k] = in_scale[tid + 512 * k]; a_sh[k] = in_shift[tid + 512 * k]; }
  }
  hload(0);
  if (AFF) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tid + 512 * k < Cin) { aff[tid + 512 * k] = a_sc[k]; aff[1024 + tid + 512 * k] = a_sh[k]; }
    __syncthreads();                             // scale / shift visible
  }
  hstore(0, 0);
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    rd(0, g, std::true_type{});
    tstep(g, std::true_type{});
  }
  xform(std::true_type{});
  rd(1, 0, std::true_type{});                    // column 0 of sub-step 1

  const int nch = Cin / KC;
  int nxt = BUF;                                 // float offset of the buffer being filled
  // ST = false: the last chunk, a second copy of the loop body that stages nothing (its read-ahead columns wrap around
  // inside the buffer being consumed and are never used) -- as in conv_wino64.hip
  auto chunk = [&](const int ch, auto stage_tag, auto xt) {
    constexpr bool ST = decltype(stage_tag)::value;
    constexpr bool XT = decltype(xt)::value;
    const int cn = (ch + 1) * KC;
    const int dlt = 2 * nxt - BUF;
    if (ST) { if (NH == 2) hloadP(0, cn); else hload(cn); }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      int gs = ch * 8 + s + 1;
      gs = gs < nsub ? gs : nsub - 1;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        tstep(g, xt);                            // column g of sub-step s+1, read one column ago
        if (ST && NH == 2 && s == 3 && g == 2) { hstoreP(0, nxt, cn); hloadP(1, cn); }
        if (ST && s == 6 && g == 2) {            // the next chunk must be visible before its first column is read
          if (NH == 2) hstoreP(1, nxt, cn); else hstore(nxt, cn);
          __syncthreads();
          Ra += dlt; Rb += dlt; Rxa += dlt; Rxb += dlt;
        }
        if (g < 2) rd((s + 1) & 7, g + 1, xt);
        else rd((s + 2) & 7, 0, xt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int h = 0; h < NH; ++h)
            acc[h][g][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[h][g], wreg[g][j], acc[h][g][j], 0, 0, 0);
        }
        if (XT) accx[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ux[g], wx[g], accx[g], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wreg[g] = wload(gs, g);
        if (XT) wx[g] = wloadx(gs, g);
        __builtin_amdgcn_sched_barrier(0);
      }
      xform(xt);
    }
    nxt = BUF - nxt;
  };
  if (NH == 2 || extra) {                        // NH = 2: one role, one copy of the loop
    for (int ch = 0; ch < nch - 1; ++ch) chunk(ch, std::true_type{}, std::true_type{});
    chunk(nch - 1, std::false_type{}, std::true_type{});
  } else {
    for (int ch = 0; ch < nch - 1; ++ch) chunk(ch, std::true_type{}, std::false_type{});
    chunk(nch - 1, std::false_type{}, std::false_type{});
  }

  // ---------------------------------------------------------------- epilogue, two passes of two column blocks each
  // lane l holds P[voxel = 16 h + 4 kq + i][n = l & 15] for its (fz, fy) and fx = 0..2.  fx -> dx in registers (dx0 = P0 + P1,
  // dx1 = P1 + P2); fy -> dy and fz -> dz across the waves through LDS: output (dz, dy, dx) = sum over fz in {dz, dz+1},
  // fy in {dy, dy+1}.
  // [9 w][32 slots = ((h * 4 + i) * 2 + dx) * 2 + jl][64 lanes + 16]: slot pitch 80 with jl as the lowest slot bit, so that the
  // eight lanes of a 16-byte read group (cq, jl) hit eight different bank groups (conv_wino64.hip has the arithmetic)
  constexpr int PS = 80, PW = 16 * NH * PS;
  float* part = lds;
  const int cq = tid & 3, jl = (tid >> 2) & 1, o = (tid >> 3) & 3, tile = (tid >> 5) & 15;
  const int dyo = o >> 1, dxo = o & 1;
  const int ttz = tile >> 3, tty = (tile >> 2) & 1, ttx = tile & 3;
  const int vz = 2 * (oz + ttz), vy = 2 * (oy + tty) + dyo, vx = 2 * (ox + ttx) + dxo;
  const size_t vox0 = (((size_t)b * S + vz) * S + vy) * S + vx;         // half h: + 4 fine rows
  const int slot_rd = ((ttx * 2 + dxo) * 2 + jl) * PS + (tile >> 2) * 16 + cq * 4;
  uf4 val[2][NH][2];                             // [pass][half][dz]
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = pass * 2 + jj;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          part[w * PW + (((h * 4 + i) * 2 + 0) * 2 + jj) * PS + lane] = acc[h][0][j][i] + acc[h][1][j][i];
          part[w * PW + (((h * 4 + i) * 2 + 1) * 2 + jj) * PS + lane] = acc[h][1][j][i] + acc[h][2][j][i];
        }
      }
    if (extra && ((w & 3) >> 1) == pass) {       // pair (2, 2), column block w & 3, voxel half xh
      const int jj = w & 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        part[8 * PW + (((xh * 4 + i) * 2 + 0) * 2 + jj) * PS + lane] = accx[0][i] + accx[1][i];
        part[8 * PW + (((xh * 4 + i) * 2 + 1) * 2 + jj) * PS + lane] = accx[1][i] + accx[2][i];
      }
    }
    __syncthreads();
    {
      const int nn = n0 + (pass * 2 + jl) * 16 + cq * 4;           // this thread's four output channels
      uf4 bv = {0.f, 0.f, 0.f, 0.f};
      if (bias != nullptr) bv = *reinterpret_cast<const uf4*>(bias + nn);
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        uf4 r[3];                                // sum over fy in {dy, dy+1} for fz = 0, 1, 2
#pragma unroll
        for (int z = 0; z < 3; ++z)
          r[z] = *reinterpret_cast<const uf4*>(&part[(z * 3 + dyo) * PW + h * 16 * PS + slot_rd]) +
                 *reinterpret_cast<const uf4*>(&part[(z * 3 + dyo + 1) * PW + h * 16 * PS + slot_rd]);
        const size_t o0 = (vox0 + (size_t)h * 4 * S) * ldo + nn;
        const size_t o1 = o0 + (size_t)S * S * ldo;
        uf4 e0 = r[0] + r[1] + bv, e1 = r[1] + r[2] + bv;
        if (accumulate) {
          e0 += *reinterpret_cast<const uf4*>(y + o0);
          e1 += *reinterpret_cast<const uf4*>(y + o1);
        }
        e0.x = uact(e0.x, pre_slope); e0.y = uact(e0.y, pre_slope); e0.z = uact(e0.z, pre_slope); e0.w = uact(e0.w, pre_slope);
        e1.x = uact(e1.x, pre_slope); e1.y = uact(e1.y, pre_slope); e1.z = uact(e1.z, pre_slope); e1.w = uact(e1.w, pre_slope);
        *reinterpret_cast<uf4*>(y + o0) = e0;
        *reinterpret_cast<uf4*>(y + o1) = e1;
        if (STATS) { val[pass][h][0] = e0; val[pass][h][1] = e1; }
      }
    }
  }
  if (!STATS) return;

  // block-level (count, mean, M2) per column over the block's 256 fine voxels (conv_igemm.hip's layout
  // [3][Npad][nblocks], block index fastest); the 512 reader threads are waves 0..7, four voxels each per pass
  auto colreduce = [&](uf4 v) -> uf4 {
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) {
      v.x += __shfl_xor(v.x, d); v.y += __shfl_xor(v.y, d); v.z += __shfl_xor(v.z, d); v.w += __shfl_xor(v.w, d);
    }
    return v;
  };
  const int cidx = jl * 16 + cq * 4;
  auto put = [&](const uf4 (&v)[2]) {
    uf4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
    r0 = colreduce(v[0]); r1 = colreduce(v[1]);
    __syncthreads();
    if (lane < 8) {
      *reinterpret_cast<uf4*>(&red[w * 64 + cidx]) = r0;
      *reinterpret_cast<uf4*>(&red[w * 64 + 32 + cidx]) = r1;
    }
    __syncthreads();
  };
  auto sum8 = [&](int i) {
    float sacc = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) sacc += red[ww * 64 + i];
    return sacc;
  };
  const size_t nstat = gridDim.x / nchunks;
  uf4 csum[2];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    csum[pass] = val[pass][0][0] + val[pass][0][1];
    if (NH == 2) csum[pass] += val[pass][NH - 1][0] + val[pass][NH - 1][1];
  }
  put(csum);
  if (tid < 64) red[512 + tid] = sum8(tid) * (1.f / kRows);
  __syncthreads();
  uf4 qs[2];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const uf4 mu = *reinterpret_cast<const uf4*>(&red[512 + pass * 32 + cidx]);
    uf4 qacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int dzz = 0; dzz < 2; ++dzz) { const uf4 dd = val[pass][h][dzz] - mu; qacc += dd * dd; }
    qs[pass] = qacc;
  }
  float mean_t = 0.f;
  if (tid < 64) mean_t = red[512 + tid];
  put(qs);
  if (tid < 64) {
    float* sp = stat_partial + (size_t)(n0 + tid) * nstat + tblk;
    sp[0] = (float)kRows;
    sp[(size_t)Npad * nstat] = mean_t;
    sp[(size_t)2 * Npad * nstat] = sum8(tid);
  }
}

// ---------------------------------------------------------------- host side
// g: geometry of the LOW-RES problem as geom_par_fwd gives it (S = low-res extent, Cin = Cu, Cout)
bool conv_up3_ok(const ConvGeom& g, const ConvSrc& s) {
  if (g.flags & (CF_NO_UPSPLIT | CF_NO_UP3)) return false;
  if (g.S < 4 || g.S % 4 != 0 || s.up || s.bcast || s.C != g.Cin) return false;     // g.S: the LOW-RES side
  if (g.Cin % KC != 0 || g.Cin > 1024 || g.Cout % 64 != 0) return false;
  if ((long long)g.B * g.S * g.S * g.S * 8ll * (long long)std::max(g.Cin, g.Cout) >= (1ll << 29)) return false;   // byte offsets
  return true;
}

int launch_conv_fwd_up3(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* wt, const float* bias,
                        float* out, int ldo, int pre_act, float* stat_partial, int* stat_blocks, int accumulate) {
  ICS_CHECK(conv_up3_ok(g, s0), "shape not served by the 27-product upsampled-input kernel");
  ICS_CHECK(ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(s0.p) & 15) == 0,
            "upsampled-input kernel: float4 accesses need 16-byte aligned tensors");
  // 32-voxel workgroups when they still give every CU several (c17.up 4096, c15.up 1024); 16-voxel ones below that
  const unsigned grid32 = (unsigned)(g.B * (g.S / 2) * (g.S / 4) * (g.S / 4) * (g.Cout / 64));
  // ICSG3D_UP3_BIG_MIN_WG overrides the threshold (tests: 1 = the 32-voxel tile wherever the shape allows it)
  const unsigned big_min = (g.flags & CF_UP3_BIG_ALWAYS) ? 1u : (unsigned)ICS_UP3_BIG_MIN_WG;
  const bool big = grid32 >= big_min;
  const unsigned grid = big ? grid32 : 2u * grid32;
  if (stat_blocks) *stat_blocks = (int)(grid / (unsigned)(g.Cout / 64));
  const bool aff = s0.scale != nullptr, noact = s0.act == ACT_NONE;
  const float in_slope = uslope(s0.act), pre_slope = uslope(pre_act);
#define ICS_UP3_LAUNCH(AFFV, NOACTV, STATSV)                                                                      \
  do {                                                                                                            \
    if (big)                                                                                                      \
      ICS_LAUNCH((conv_up3_kernel<AFFV, NOACTV, STATSV, 2>), dim3(grid), dim3(512), 0, st, s0.p, s0.C,     \
                         s0.scale, s0.shift, in_slope, wt, bias, out, ldo, pre_slope, accumulate, stat_partial,   \
                         g.Npad, g.S, g.Cin, g.Cout);                                                             \
    else                                                                                                          \
      ICS_LAUNCH((conv_up3_kernel<AFFV, NOACTV, STATSV, 1>), dim3(grid), dim3(512), 0, st, s0.p, s0.C,     \
                         s0.scale, s0.shift, in_slope, wt, bias, out, ldo, pre_slope, accumulate, stat_partial,   \
                         g.Npad, g.S, g.Cin, g.Cout);                                                             \
    conv_set_last_kernel_id(big ? "conv_up3_kernel<" #AFFV ", " #NOACTV ", " #STATSV ", 2>"                         \
                                : "conv_up3_kernel<" #AFFV ", " #NOACTV ", " #STATSV ", 1>");                       \
  } while (0)
  if (stat_partial) {
    if (!aff) ICS_UP3_LAUNCH(false, true, true);
    else if (noact) ICS_UP3_LAUNCH(true, true, true);
    else ICS_UP3_LAUNCH(true, false, true);
  } else {
    if (!aff) ICS_UP3_LAUNCH(false, true, false);
    else if (noact) ICS_UP3_LAUNCH(true, true, false);
    else ICS_UP3_LAUNCH(true, false, false);
  }
#undef ICS_UP3_LAUNCH
  ICS_HIP(hipGetLastError());
  return 0;
}

}  // namespace ics
