// EXPERIMENT, not built (round 3): conv_up3.hip as twelve waves (three per SIMD) x 32 low-res voxels per workgroup:
// every wave owns two column blocks of one (fz, fy) pair and one of another, for both 16-voxel halves (18 accumulators),
// original weight layout.  Correct (tests/test_gpu_conv.py, test_gpu_switches.py).  ms per U-Net step, c17.up / c15.up /
// c13.up: shipped nine-wave kernel 1.362 / 0.634 / 0.318, this file 1.377 / 0.645 / 0.338.  Three more layouts measured the
// same +-1 %: twelve waves x 16 voxels with re-laid-out weights (1.361), nine waves x 32 voxels (1.394), eight waves +
// the ninth pair split over four of them (1.387) -- although scripts/up3_timeline.py shows the shipped main loop running
// at the pace of the three waves that share SIMD 0.  The in-loop spills of this file (168 registers at three waves per
// SIMD: the staged rows go through scratch) are part of the reason; the rest is not understood.
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
// One workgroup = 32 low-res voxels (2x4x4: a 4x8x8 block of fine outputs) x 64 output channels x 27 frequencies on
// v_mfma_f32_16x16x4_f32 (lane l: A[voxel l & 15][k = l >> 4]), TWELVE waves, three per SIMD.  The 27 frequencies x 4
// column blocks are 36 triples ((fz, fy) pair, column block) of three fx; every wave owns three triples -- two column
// blocks of one pair and one of another -- for both 16-voxel halves of the block: 18 accumulators of 4 registers, each
// weight load feeding two MFMAs per half.  The first version (nine waves, one pair x four column blocks x 16 voxels each)
// sat 3 + 2 + 2 + 2 on the SIMDs: its main loop ran at the pace of SIMD 0's 36 MFMAs per sub-step (scripts/up3_timeline.py).
// Staging: thread t < 576 owns (hy, hx, tile z, channel quad) of the halo [4][6][6] x 32 channels: three z rows in, the
// producer's BatchNorm affine + activation, zero padding, the z rows of B, three planes out.  LDS: voxel pitch 34 floats,
// row pitch 208, plane pitch 1256 (conflict-free ds_read_b32 for every (fz, row, column, sub-step): the pattern of the
// 16-voxel layout, the second half is a constant offset of two rows).  Weights [Cout/64][Cu/4][27 f][4 k][16 n][4 column blocks].
#include "common.h"

#include <type_traits>

#include <algorithm>

namespace ics {

typedef float uf4 __attribute__((ext_vector_type(4)));
typedef float uf2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int KC = 32;                                   // input channels per LDS chunk
constexpr int VX = 34, RP = 208, PP = 1256, BUF = 6 * PP; // floats; one buffer = 30 144 B
constexpr int kRows = 256;                               // fine voxels per workgroup

__device__ __forceinline__ float uact(float v, float slope) { return fmaxf(v, v * slope); }
__host__ __device__ __forceinline__ float uslope(int act) { return act == ACT_RELU ? 0.f : (act == ACT_LRELU ? kLeaky : 1.f); }
}  // namespace

// x: low-res source [B][Sl][Sl][Sl][ldx]; y: fine output [B][2 Sl]^3 [ldo].  AFF / NOACT as in conv_wino64.hip.
// STATS: per-block BatchNorm statistics of the stored values (the VAE decoder's layers; the U-Net's launches leave bias,
// activation and statistics to the skip-channel pass that accumulates on top).
// One twelve-wave workgroup per CU.  Every MFMA takes a fresh 256-byte B operand from L2 (16 voxels per weight load):
// ~18 TB/s over the chip, half of the L2's 34.5 TB/s, the same rate conv_wino64.hip runs at.
template <bool AFF, bool NOACT, bool STATS>
__global__ __launch_bounds__(768) void conv_up3_kernel(const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ in_scale,
                                                          const float* __restrict__ in_shift, float in_slope,
                                                          const float* __restrict__ wt, const float* __restrict__ bias,
                                                          float* __restrict__ y, int ldo, float pre_slope, int accumulate,
                                                          float* __restrict__ stat_partial, int Npad, int Sl, int Cin,
                                                          int Cout) {
  __shared__ __attribute__((aligned(16))) float lds[9 * 32 * 80];   // 92 160 B: two buffers (60 288 B); the epilogue's [9][32][80]
  __shared__ unsigned park[3 * 768];
  __shared__ __attribute__((aligned(16))) float aff[AFF ? 2048 : 4];
  __shared__ float red[9 * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this wave: two column blocks (cbA, cbA + 1) of the (fz, fy) pair pA and one column block cbB of the pair pB, all
  // three fx each -- 36 (pair, column block) triples over 12 waves.  Pairs 0..5 give column blocks 0, 1 to wave p and
  // 2 / 3 as the single block of waves (p + 5) % 6 / p + 6; pairs 6..8 give (0, 1) and (2, 3) to waves 6 + 2 j, 7 + 2 j.
  const int pA = w < 6 ? w : 6 + ((w - 6) >> 1), cbA = w < 6 ? 0 : 2 * ((w - 6) & 1);
  const int pB = w < 6 ? (w + 1) % 6 : w - 6, cbB = w < 6 ? 2 : 3;
  const int fzA = pA / 3, fyA = pA - 3 * fzA, fzB = pB / 3, fyB = pB - 3 * fzB;
  const int m = lane & 15, kq = lane >> 4;
  const int nchunks = Cout >> 6;
  const int nb = blockIdx.x % nchunks;
  const int tblk = blockIdx.x / nchunks;
  int tb = tblk;
  const int nbx = Sl >> 2, nby = Sl >> 2, nbz = Sl >> 1;
  const int bx = tb % nbx; tb /= nbx;
  const int by = tb % nby; tb /= nby;
  const int bz = tb % nbz;
  const int b = tb / nbz;
  const int oz = bz * 2, oy = by * 4, ox = bx * 4, n0 = nb * 64;      // low-res origin of the block
  const bool edge = bx == 0 || by == 0 || bz == 0 || bx == nbx - 1 || by == nby - 1 || bz == nbz - 1;   // uniform
  const int S = 2 * Sl;

  // ---- staging: thread t < 576 owns (hy, hx, tile z tzh, channel quad): z rows tzh, tzh+1, tzh+2 of the halo [4][6][6]
  const int scol = tid < 576 ? tid : 575;
  const int q = scol & 7, tzh = (scol >> 3) & 1, hx = (scol >> 4) % 6, hy = (scol >> 4) / 6;
  uf4 hs[3];
  unsigned zoff[4];                              // uniform: byte offset of sample b, low-res plane clamp(oz - 1 + hz)
  unsigned okz = 0;                              // uniform: bit hz = plane inside the grid
  {
    const int gy = oy - 1 + hy, gx = ox - 1 + hx;
    const int cy = min(max(gy, 0), Sl - 1), cx = min(max(gx, 0), Sl - 1);
    const bool okyx = gy == cy && gx == cx;
    park[tid] = (unsigned)((cy * Sl + cx) * ldx + q * 4) * 4u;
    park[768 + tid] = (unsigned)(tzh * 3 * PP + hy * RP + hx * VX + q * 4);
    park[1536 + tid] = (unsigned)(q * 4) | ((unsigned)tzh << 8) | ((okyx ? 1u : 0u) << 9);
#pragma unroll
    for (int hz = 0; hz < 4; ++hz) {
      const int gz = oz - 1 + hz, cz = min(max(gz, 0), Sl - 1);
      okz |= gz == cz ? (1u << hz) : 0u;
      zoff[hz] = (unsigned)((b * Sl + cz) * Sl * Sl * ldx) * 4u;
    }
  }
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x7fffffff, 0x00020000);
  // per-thread staging constants parked in LDS (conv_wino64.hip explains why)
  const unsigned park_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&park[0] + (unsigned)w * 256u);
  auto unpark = [&](const int which) -> int {
    int v;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshl_add_u32 %0, %0, 2, %1\n\t"
                 "ds_read_b32 %0, %0 offset:%2\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v) : "s"(park_wave), "i"(which * 3072) : "memory");
    return v;
  };
  if (AFF) {
    for (int i = tid; i < Cin; i += 768) { aff[i] = in_scale[i]; aff[1024 + i] = in_shift[i]; }
  }
  auto hload = [&](int c0) {
    const int vo = unpark(0);
    const bool tz1 = ((unpark(2) >> 8) & 1) != 0; // per lane: the tile z of this column
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const unsigned zo = tz1 ? zoff[1 + i] : zoff[i];
      hs[i] = __builtin_bit_cast(uf4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)((unsigned)vo + zo), (int)((unsigned)c0 * 4u), 0));
    }
  };
  auto hstore = [&](const int bo, const int c0) {
    // two channels at a time: 14 instead of 28 temporaries (the kernel runs three waves per SIMD: 168 registers)
    const int pk2 = unpark(2);
    const int q4 = pk2 & 255;
    const unsigned okzz = ((pk2 >> 8) & 1) ? okz >> 1 : okz;
    const bool okyx = ((pk2 >> 9) & 1) != 0;
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
      if (edge) {                                // zero padding AFTER the producer's affine / activation
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (!(okyx && ((okzz >> i) & 1))) r[i] = uf2{0.f, 0.f};
      }
      if (tid < 576) {
        *reinterpret_cast<uf2*>(o + 2 * hh) = r[0] - r[1];
        *reinterpret_cast<uf2*>(o + PP + 2 * hh) = r[1];
        *reinterpret_cast<uf2*>(o + 2 * PP + 2 * hh) = r[2] - r[1];
      }
    }
  };

  // ---- per-lane read geometry: voxel m = (tz, ty, tx); rows of a pair's fy: D0 = r0 - r1, D1 = r1, D2 = r2 - r1
  const int tz = m >> 3, ty = (m >> 2) & 1, tx = m & 3;
  const float saA = fyA == 1 ? 0.f : 1.f, saB = fyB == 1 ? 0.f : 1.f;    // t = sa * qa - qb (fy = 1: sign folded into the weights)
  const int R0A = (tz * 3 + fzA) * PP + ty * RP + tx * VX + kq, R0B = (tz * 3 + fzB) * PP + ty * RP + tx * VX + kq;
  int RaA = R0A + (fyA == 2 ? 2 : 0) * RP, RbA = R0A + RP, RaB = R0B + (fyB == 2 ? 2 : 0) * RP, RbB = R0B + RP;

  const int nsub = Cin >> 2;
  constexpr int wstride_f = 256;                 // floats per frequency of one sub-step: [4 k][16 n][4 column blocks]
  constexpr int wsub = 27 * 256;
  const __amdgpu_buffer_rsrc_t wrsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wt + ((size_t)nb * nsub * 27 + pA * 3) * 256 + cbA), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wt + ((size_t)nb * nsub * 27 + pB * 3) * 256 + cbB), 0, 0x7fffffff, 0x00020000);
  const int wlane = lane * 16;                   // bytes
  auto wloadA = [&](int gs, int g) {             // sub-step gs, fx g: column blocks cbA, cbA + 1
    return __builtin_bit_cast(uf2, __builtin_amdgcn_raw_buffer_load_b64(wrsA, wlane, (gs * wsub + g * wstride_f) * 4, 0));
  };
  auto wloadB = [&](int gs, int g) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsB, wlane, (gs * wsub + g * wstride_f) * 4, 0));
  };
  uf2 wA[3];
  float wB[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) { wA[g] = wloadA(0, g); wB[g] = wloadB(0, g); }

  uf4 acc[2][3][3];                              // [voxel half][triple][fx]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[h][i][g] = uf4{0.f, 0.f, 0.f, 0.f};

  float uA[2][3], uB[2][3], tnA[2][3], tnB[2][3], qaA[2], qbA[2], qaB[2], qbB[2];
  auto rd = [&](const int sub, const int col) {
    const int off = col * VX + 4 * sub;          // compile-time after unrolling
#pragma unroll
    for (int h = 0; h < 2; ++h) {                // second half: two low-res rows further
      qaA[h] = lds[RaA + off + h * 2 * RP]; qbA[h] = lds[RbA + off + h * 2 * RP];
      qaB[h] = lds[RaB + off + h * 2 * RP]; qbB[h] = lds[RbB + off + h * 2 * RP];
    }
  };
  auto tstep = [&](const int g) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      tnA[h][g] = fmaf(saA, qaA[h], -qbA[h]);
      tnB[h][g] = fmaf(saB, qaB[h], -qbB[h]);
    }
  };
  auto xform = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uA[h][0] = tnA[h][0] - tnA[h][1]; uA[h][1] = tnA[h][1]; uA[h][2] = tnA[h][2] - tnA[h][1];
      uB[h][0] = tnB[h][0] - tnB[h][1]; uB[h][1] = tnB[h][1]; uB[h][2] = tnB[h][2] - tnB[h][1];
    }
  };

  hload(0);
  if (AFF) __syncthreads();                      // scale / shift visible
  hstore(0, 0);
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    rd(0, g);
    tstep(g);
  }
  xform();
  rd(1, 0);                                      // column 0 of sub-step 1

  const int nch = Cin / KC;
  int nxt = BUF;                                 // float offset of the buffer being filled
  // ST = false: the last chunk, a second copy of the loop body that stages nothing (its read-ahead columns wrap around
  // inside the buffer being consumed and are never used) -- as in conv_wino64.hip
  auto chunk = [&](const int ch, auto stage_tag) {
    constexpr bool ST = decltype(stage_tag)::value;
    const int cn = (ch + 1) * KC;
    const int dlt = 2 * nxt - BUF;
    if (ST) hload(cn);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      int gs = ch * 8 + s + 1;
      gs = gs < nsub ? gs : nsub - 1;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        tstep(g);                                // column g of sub-step s+1, read one column ago
        if (ST && s == 6 && g == 2) {            // the next chunk must be visible before its first column is read
          hstore(nxt, cn);
          __syncthreads();
          RaA += dlt; RbA += dlt; RaB += dlt; RbB += dlt;
        }
        if (g < 2) rd((s + 1) & 7, g + 1);
        else rd((s + 2) & 7, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          acc[h][0][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(uA[h][g], wA[g].x, acc[h][0][g], 0, 0, 0);
          acc[h][1][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(uA[h][g], wA[g].y, acc[h][1][g], 0, 0, 0);
          acc[h][2][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(uB[h][g], wB[g], acc[h][2][g], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        wA[g] = wloadA(gs, g);
        wB[g] = wloadB(gs, g);
        __builtin_amdgcn_sched_barrier(0);
      }
      xform();
    }
    nxt = BUF - nxt;
  };
  for (int ch = 0; ch < nch - 1; ++ch) chunk(ch, std::true_type{});
  chunk(nch - 1, std::false_type{});

  // ---------------------------------------------------------------- epilogue, two passes of two column blocks each
  // lane l holds P[voxel = 16 h + 4 kq + i][n = l & 15] for its (fz, fy) and fx = 0..2.  fx -> dx in registers (dx0 = P0 + P1,
  // dx1 = P1 + P2); fy -> dy and fz -> dz across the waves through LDS: output (dz, dy, dx) = sum over fz in {dz, dz+1},
  // fy in {dy, dy+1}.
  // [9 w][32 slots = ((h * 4 + i) * 2 + dx) * 2 + jl][64 lanes + 16]: slot pitch 80 with jl as the lowest slot bit, so that the
  // eight lanes of a 16-byte read group (cq, jl) hit eight different bank groups (conv_wino64.hip has the arithmetic)
  constexpr int PS = 80, PW = 32 * PS;
  float* part = lds;
  const int cq = tid & 3, jl = (tid >> 2) & 1, o = (tid >> 3) & 3, tile = (tid >> 5) & 15;
  const int dyo = o >> 1, dxo = o & 1;
  const int ttz = tile >> 3, tty = (tile >> 2) & 1, ttx = tile & 3;
  const int vz = 2 * (oz + ttz), vy = 2 * (oy + tty) + dyo, vx = 2 * (ox + ttx) + dxo;
  const size_t vox0 = (((size_t)b * S + vz) * S + vy) * S + vx;         // half h: + 4 fine rows
  const int slot_rd = ((ttx * 2 + dxo) * 2 + jl) * PS + (tile >> 2) * 16 + cq * 4;
  uf4 val[2][2][2];                              // [pass][half][dz]
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int cmb = i < 2 ? pA : pB, cbk = i < 2 ? cbA + i : cbB;  // uniform
      if ((cbk >> 1) == pass) {
        const int jj = cbk & 1;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            part[cmb * PW + (((h * 4 + r) * 2 + 0) * 2 + jj) * PS + lane] = acc[h][i][0][r] + acc[h][i][1][r];
            part[cmb * PW + (((h * 4 + r) * 2 + 1) * 2 + jj) * PS + lane] = acc[h][i][1][r] + acc[h][i][2][r];
          }
      }
    }
    __syncthreads();
    if (tid < 512) {
      const int nn = n0 + (pass * 2 + jl) * 16 + cq * 4;           // this thread's four output channels
      uf4 bv = {0.f, 0.f, 0.f, 0.f};
      if (bias != nullptr) bv = *reinterpret_cast<const uf4*>(bias + nn);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
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
    if (w < 8) { r0 = colreduce(v[0]); r1 = colreduce(v[1]); }
    __syncthreads();
    if (w < 8 && lane < 8) {
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
  for (int pass = 0; pass < 2; ++pass) csum[pass] = (val[pass][0][0] + val[pass][0][1]) + (val[pass][1][0] + val[pass][1][1]);
  put(csum);
  if (tid < 64) red[512 + tid] = sum8(tid) * (1.f / kRows);
  __syncthreads();
  uf4 qs[2];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const uf4 mu = *reinterpret_cast<const uf4*>(&red[512 + pass * 32 + cidx]);
    uf4 qacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h)
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
  if (g.S < 4 || s.up || s.bcast || s.C != g.Cin) return false;
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
  const unsigned grid = (unsigned)(g.B * (g.S / 2) * (g.S / 4) * (g.S / 4) * (g.Cout / 64));
  if (stat_blocks) *stat_blocks = (int)(grid / (unsigned)(g.Cout / 64));
  const bool aff = s0.scale != nullptr, noact = s0.act == ACT_NONE;
  const float in_slope = uslope(s0.act), pre_slope = uslope(pre_act);
#define ICS_UP3_LAUNCH(AFFV, NOACTV, STATSV)                                                                      \
  do {                                                                                                            \
    hipLaunchKernelGGL((conv_up3_kernel<AFFV, NOACTV, STATSV>), dim3(grid), dim3(768), 0, st, s0.p, s0.C,          \
                       s0.scale, s0.shift, in_slope, wt, bias, out, ldo, pre_slope, accumulate, stat_partial,     \
                       g.Npad, g.S, g.Cin, g.Cout);                                                               \
    conv_set_last_kernel_id("conv_up3_kernel<" #AFFV ", " #NOACTV ", " #STATSV ">");                              \
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
