// 3x3x3 "same" convolution over a nearest-upsampled input, 27 products per low-res voxel, NARROW outputs (gfx950 / MI355X
// only), round 4: the VAE decoder's last upsampled layers d2 (64 -> 32 channels, 8^3 -> 16^3) and d3 (32 -> 16, 16^3 -> 32^3)
// (/root/reference/vae/lattice_vae.py:211-217).  Same algebra as conv_up3.hip (its header has the derivation):
//   D = (B (x) B (x) B) x over the 3x3x3 low-res neighbourhood, B = [1 -1 0; 0 1 0; 0 -1 1]
//   P_f = D_f . G_f,  G = (g (x) g (x) g) w,  g = [1 0 0; 1 1 1; 0 0 1]          27 products per low-res voxel
//   Y = (A (x) A (x) A) P,  A = [1 1 0; 0 1 1]                                    the 2x2x2 fine outputs
// conv_up3.hip's tile is 64 output channels wide; with 16 / 32 outputs these layers stayed on the 8-tap parity GEMMs with
// a 128 x 32 tile (half of it padding at Cout = 16, 64 instead of 27 products: d3 0.20 ms at 43 TFLOP/s executed, the
// largest single item of the VAE's own layers).  Here the MFMA tile is v_mfma_f32_16x16x4_f32's natural 16 voxels x 16
// channels: one workgroup = 4x4x4 low-res voxels (an 8^3 block of fine outputs) x 16 output channels, four waves, wave w
// owns the z slice w (16 voxels = the MFMA's M) and all 27 frequencies (27 accumulators of 4 registers).  The halo
// [6][6][6] x Cu is staged once into LDS (producer's BatchNorm affine + activation, zero padding after it); per k-step of
// four channels a lane reads its 27 neighbours, applies B along x, y, z in registers and issues 27 MFMAs against weights
// streamed from L1 / L2 ([27 f][Cu/4][4 k][Cout]: one sub-step of one frequency and column block = 256 contiguous bytes).
// Epilogue in the lane: D[voxel 4 (l >> 4) + r][channel l & 15] holds all 27 P_f of a (voxel, channel) pair -> A along
// x, y, z, bias, activation, 8 stores; BatchNorm (count, mean, M2) partials per workgroup (512 fine voxels) by equal-count
// Chan merges.  Column blocks (Cout = 32: two) are separate workgroups (gridDim.y).
#include "common.h"

#include <algorithm>

namespace ics {

typedef float nf4 __attribute__((ext_vector_type(4)));

namespace {
__host__ __device__ constexpr int up3n_row_pitch(int Cu) {      // smallest Py >= 6 (Cu + 2) with Py = 8 (mod 32)
  return 6 * (Cu + 2) + ((8 - (6 * (Cu + 2)) % 32) + 32) % 32;
}
__device__ __forceinline__ float nact(float v, float slope) { return fmaxf(v, v * slope); }
__host__ __device__ __forceinline__ float nslope(int act) { return act == ACT_RELU ? 0.f : (act == ACT_LRELU ? kLeaky : 1.f); }
}  // namespace

// x: low-res source [B][Sl]^3[ldx] (Cu channels used); y: fine output [B][2 Sl]^3[ldo]; wt: [27][Cu/4][4][Cout];
// stat_partial: per-workgroup BatchNorm partials or nullptr.
template <bool AFF, int Cu>
__global__ __launch_bounds__(256, 2) void conv_up3n_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ in_scale,
                                                        const float* __restrict__ in_shift, float in_slope,
                                                        const float* __restrict__ wt, const float* __restrict__ bias,
                                                        float* __restrict__ y, int ldo, float pre_slope,
                                                        float* __restrict__ stat_partial, int Npad, int Sl, int Cout) {
  // halo[hz][hy][hx][channel] with pitches Px = Cu + 2 (= 2 mod 32), Py = 8 mod 32, Pz = 6 Py: the 32 lanes of a half-wave
  // (voxel (my, mx), k index kq in {0, 1} / {2, 3}) read banks 2 mx + 8 my + kq -- all different
  extern __shared__ __attribute__((aligned(16))) float halo[];
  __shared__ float s_mean[4][16], s_m2[4][16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nt = Sl >> 2;
  int tb = blockIdx.x;
  const int bx = tb % nt; tb /= nt;
  const int by = tb % nt; tb /= nt;
  const int bz = tb % nt;
  const int b = tb / nt;
  const int n0 = blockIdx.y * 16;
  constexpr int Px = Cu + 2, Py = up3n_row_pitch(Cu), Pz = 6 * Py;
  // ---- stage the halo: positions (hz, hy, hx) in [0, 6)^3 <-> low-res voxel (4 bz - 1 + hz, ...), float4 per thread.
  // ALL of a thread's loads are requested before the first is stored (first version: load -> store per iteration, seven
  // HBM round trips in a row and 10 of a workgroup's 15 us)
  constexpr int q4 = Cu >> 2, NIT = (216 * q4 + 255) / 256;
  nf4 sv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + 256 * it;
    const int pos = i / q4, cq = i - pos * q4;
    const int hx = pos % 6, hy = (pos / 6) % 6, hz = pos / 36;
    const int gz = 4 * bz - 1 + hz, gy = 4 * by - 1 + hy, gx = 4 * bx - 1 + hx;
    const bool ok = i < 216 * q4 && (unsigned)gz < (unsigned)Sl && (unsigned)gy < (unsigned)Sl && (unsigned)gx < (unsigned)Sl;
    sv[it] = nf4{0.f, 0.f, 0.f, 0.f};
    if (ok) sv[it] = *reinterpret_cast<const nf4*>(x + ((((size_t)b * Sl + gz) * Sl + gy) * Sl + gx) * ldx + cq * 4);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + 256 * it;
    const int pos = i / q4, cq = i - pos * q4;
    const int hx = pos % 6, hy = (pos / 6) % 6, hz = pos / 36;
    const int gz = 4 * bz - 1 + hz, gy = 4 * by - 1 + hy, gx = 4 * bx - 1 + hx;
    const bool ok = (unsigned)gz < (unsigned)Sl && (unsigned)gy < (unsigned)Sl && (unsigned)gx < (unsigned)Sl;
    nf4 v = sv[it];
    if (AFF && ok) {
      const nf4 sc = *reinterpret_cast<const nf4*>(in_scale + cq * 4), sh = *reinterpret_cast<const nf4*>(in_shift + cq * 4);
      v.x = nact(fmaf(v.x, sc.x, sh.x), in_slope); v.y = nact(fmaf(v.y, sc.y, sh.y), in_slope);
      v.z = nact(fmaf(v.z, sc.z, sh.z), in_slope); v.w = nact(fmaf(v.w, sc.w, sh.w), in_slope);
    }
    if (i < 216 * q4) {
      float* h = halo + hz * Pz + hy * Py + hx * Px + cq * 4;
      h[0] = v.x; h[1] = v.y; h[2] = v.z; h[3] = v.w;
    }
  }
  __syncthreads();
  // ---- main loop: lane = (voxel m = (my, mx) of the wave's z slice, k index kq)
  const int m = lane & 15, kq = lane >> 4, my = m >> 2, mx = m & 3;
  const float* hb = halo + w * Pz + my * Py + mx * Px + kq;         // neighbour (dz, dy, dx) in 0..2: + dz Pz + dy Py + dx Px
  nf4 acc[27];
#pragma unroll
  for (int f = 0; f < 27; ++f) acc[f] = nf4{0.f, 0.f, 0.f, 0.f};
  const float* wl = wt + (size_t)kq * Cout + n0 + (lane & 15);      // + ((f * q4 + s) * 4) * Cout
  // the weights of step s + 1 are requested while step s multiplies (two register sets that swap roles, loop unrolled by
  // two); the neighbourhood comes from LDS at the top of each step -- the other wave of the SIMD covers that latency
  float d[27], bA[27], bB[27];
  auto load_w = [&](int s, float (&bw)[27]) {
#pragma unroll
    for (int f = 0; f < 27; ++f) bw[f] = wl[(size_t)((f * q4 + s) * 4) * Cout];
  };
  auto mul_step = [&](int s, const float (&bw)[27]) {
#pragma unroll
    for (int f = 0; f < 27; ++f) d[f] = hb[(f / 9) * Pz + ((f / 3) % 3) * Py + (f % 3) * Px + 4 * s];
    // rows of B: (x[-1] - x[0], x[0], x[+1] - x[0]) along x, y, z
#pragma unroll
    for (int i = 0; i < 9; ++i) { d[3 * i] -= d[3 * i + 1]; d[3 * i + 2] -= d[3 * i + 1]; }
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) { d[dz * 9 + dx] -= d[dz * 9 + 3 + dx]; d[dz * 9 + 6 + dx] -= d[dz * 9 + 3 + dx]; }
#pragma unroll
    for (int i = 0; i < 9; ++i) { d[i] -= d[9 + i]; d[18 + i] -= d[9 + i]; }
#pragma unroll
    for (int f = 0; f < 27; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[f], bw[f], acc[f], 0, 0, 0);
  };
  load_w(0, bA);
#pragma unroll 1
  for (int s = 0; s < q4; s += 2) {                 // q4 is even (Cu % 32 == 0)
    load_w(s + 1, bB);
    mul_step(s, bA);
    if (s + 2 < q4) load_w(s + 2, bA);
    mul_step(s + 1, bB);
  }
  // ---- epilogue: this lane holds P_f of voxels (w, row >> 2, row & 3), row = 4 kq + r, channel n0 + (lane & 15)
  const int n = n0 + (lane & 15);
  const float bv = bias != nullptr ? bias[n] : 0.f;
  const int S = 2 * Sl;
  float vsum = 0.f, vals[4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * kq + r, ry = row >> 2, rx = row & 3;
    const int az = 4 * bz + w, ay = 4 * by + ry, ax = 4 * bx + rx;
    float p[3][3][3];
#pragma unroll
    for (int f = 0; f < 27; ++f) p[f / 9][(f / 3) % 3][f % 3] = acc[f][r];
    // rows of A: (P0 + P1, P1 + P2) along x, y, z
    float ox[3][3][2], oy[3][2][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) { ox[i][j][0] = p[i][j][0] + p[i][j][1]; ox[i][j][1] = p[i][j][1] + p[i][j][2]; }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) { oy[i][0][k] = ox[i][0][k] + ox[i][1][k]; oy[i][1][k] = ox[i][1][k] + ox[i][2][k]; }
#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const float v = nact(oy[pz][py][px] + oy[pz + 1][py][px] + bv, pre_slope);
          vals[r][(pz * 2 + py) * 2 + px] = v;
          vsum += v;
          y[((((size_t)b * S + 2 * az + pz) * S + 2 * ay + py) * S + 2 * ax + px) * ldo + n] = v;
        }
  }
  if (stat_partial == nullptr) return;    // (a compile-time variant without the statistics spilled 128 registers at Cu = 32)
  // (count, mean, M2) of the workgroup's 512 fine voxels per channel: lane (32 values) -> the 4 lanes of a channel -> 4 waves
  float mean = vsum * (1.f / 32.f), m2 = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int e = 0; e < 8; ++e) m2 += (vals[r][e] - mean) * (vals[r][e] - mean);
  float cnt = 32.f;
#pragma unroll
  for (int dlt = 16; dlt < 64; dlt <<= 1) {          // equal-count Chan merge with the lane dlt away
    const float mo = __shfl_xor(mean, dlt), qo = __shfl_xor(m2, dlt);
    const float dl = mo - mean;
    m2 = m2 + qo + dl * dl * (cnt * 0.5f);
    mean = mean + 0.5f * dl;
    cnt *= 2.f;
  }
  if (lane < 16) { s_mean[w][lane] = mean; s_m2[w][lane] = m2; }
  __syncthreads();
  if (tid < 16) {
    float ms = 0.f, q = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { ms += s_mean[ww][tid]; q += s_m2[ww][tid]; }
    const float mt = ms * 0.25f;
    float dev = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) dev += (s_mean[ww][tid] - mt) * (s_mean[ww][tid] - mt);
    const size_t nblk = gridDim.x;
    float* sp = stat_partial + (size_t)(n0 + tid) * nblk + blockIdx.x;
    sp[0] = 512.f;
    sp[(size_t)Npad * nblk] = mt;
    sp[(size_t)2 * Npad * nblk] = q + 128.f * dev;
  }
}

// ---------------------------------------------------------------- host side
// g = the LOW-RES geometry (S = low-res extent, Cin = Cu, Cout); s0 = the low-res source
bool conv_up3n_ok(const ConvGeom& g, const ConvSrc& s0) {
  if (g.flags & (CF_NO_UP3 | CF_NO_UP3N)) return false;
  if (s0.bcast || s0.C != g.Cin || s0.up) return false;
  return (g.Cout == 16 || g.Cout == 32) && (g.Cin == 32 || g.Cin == 64) && g.S >= 4 && g.S % 4 == 0;
}
size_t conv_up3n_weight_floats(int Cu, int Cout) { return (size_t)27 * Cu * Cout; }
int launch_conv_fwd_up3n(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* wt, const float* bias, float* out,
                         int ldo, int pre_act, float* stat_partial, int* stat_blocks) {
  ICS_CHECK(conv_up3n_ok(g, s0), "shape not served by the narrow 27-product upsampled-input kernel");
  const int nt = g.S / 4;
  const unsigned gx = (unsigned)(g.B * nt * nt * nt);
  const size_t lds = (size_t)36 * up3n_row_pitch(g.Cin) * sizeof(float);
  const bool aff = s0.scale != nullptr;
  const float in_slope = aff ? nslope(s0.act) : 1.f, pre_slope = nslope(pre_act);
  if (stat_blocks) *stat_blocks = (int)gx;
#define ICS_UP3N(A_)                                                                                                  \
  do {                                                                                                                \
    if (g.Cin == 32)                                                                                                  \
      ICS_LAUNCH((conv_up3n_kernel<A_, 32>), dim3(gx, (unsigned)(g.Cout / 16)), dim3(256), lds, st, s0.p, s0.C, s0.scale,   \
                 s0.shift, in_slope, wt, bias, out, ldo, pre_slope, stat_partial, g.Npad, g.S, g.Cout);                \
    else                                                                                                              \
      ICS_LAUNCH((conv_up3n_kernel<A_, 64>), dim3(gx, (unsigned)(g.Cout / 16)), dim3(256), lds, st, s0.p, s0.C, s0.scale,   \
                 s0.shift, in_slope, wt, bias, out, ldo, pre_slope, stat_partial, g.Npad, g.S, g.Cout);                \
  } while (0)
  if (aff) ICS_UP3N(true); else ICS_UP3N(false);
#undef ICS_UP3N
  ICS_HIP(hipGetLastError());
  conv_set_last_kernel_id("conv_up3n_kernel");
  return 0;
}

}  // namespace ics
