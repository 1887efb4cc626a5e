// Prototype: fused Winograd F(2x2x2, 3x3x3) forward convolution on fp32 MFMA (gfx950).
// Standalone probe (not part of the library): hipcc --offload-arch=gfx950 -O3 -o wino_proto wino_proto.hip
//   ./wino_proto check            small problem against a direct fp64 convolution on the host
//   ./wino_proto time B S Cin Cout  timing (default 32 32 128 128 = the U-Net's c18)
//
// One workgroup (4 waves) = 32 output tiles (2x4x4 tiles of 2x2x2 voxels = 4x8x8 voxels) x 32 output channels, all 64
// Winograd frequencies: wave w owns the 16 frequencies with fz = w, one 32x32 fp32 accumulator each (256 AGPRs).
// K loop over input channels: a raw halo chunk [6][10][10] voxels x 16 channels is staged in LDS; every wave reads the
// two z-planes its fz row of B^T combines, finishes the y/x transforms in registers directly in the MFMA A-operand
// layout (lane = tile, half-wave = channel group), and streams its B operands (pre-transformed weights) from L2.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int KC = 16;             // input channels per LDS chunk
constexpr int P = 20;              // LDS floats per halo voxel (16 + pad)
constexpr int HZ = 6, HY = 10, HX = 10, NV = HZ * HY * HX;

#define HIPCHECK(e)                                                                           \
  do {                                                                                        \
    hipError_t _e = (e);                                                                      \
    if (_e != hipSuccess) {                                                                   \
      fprintf(stderr, "%s -> %s @%d\n", #e, hipGetErrorString(_e), __LINE__);                 \
      exit(1);                                                                                \
    }                                                                                         \
  } while (0)

// wt layout: [64 f][Cin/8][2 h][Cout][4 j]  (channel = c8*8 + h*4 + j)
__global__ __launch_bounds__(256) void wino_fwd(const float* __restrict__ x, const float* __restrict__ wt,
                                                float* __restrict__ y, int S, int Cin, int Cout) {
  __shared__ __attribute__((aligned(16))) float lds[2 * NV * P];   // 96 000 B; the epilogue reuses it (64 KB)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m = lane & 31, h = lane >> 5;
  const int nchunks = Cout >> 5;
  const int nb = blockIdx.x % nchunks;
  int tb = blockIdx.x / nchunks;
  const int nbx = S >> 3, nby = S >> 3, nbz = S >> 2;
  const int bx = tb % nbx; tb /= nbx;
  const int by = tb % nby; tb /= nby;
  const int bz = tb % nbz;
  const int b = tb / nbz;
  const int oz = bz * 4, oy = by * 8, ox = bx * 8, n0 = nb * 32;

  // ---- staging: NV*4 float4 per chunk, 256 threads
  constexpr int NLD = (NV * 4 + 255) / 256;   // 10
  f4 stage[NLD];
  int soff[NLD];        // global float offset of (voxel, quad) without the channel-chunk term, or -1
  int doff[NLD];        // LDS float offset
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int e = tid + i * 256;
    int v = e >> 2, q = e & 3;
    int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
    int gz = oz - 1 + hz, gy = oy - 1 + hy, gx = ox - 1 + hx;
    bool ok = e < NV * 4 && gz >= 0 && gz < S && gy >= 0 && gy < S && gx >= 0 && gx < S;
    soff[i] = ok ? ((((b * S + gz) * S + gy) * S + gx) * Cin + q * 4) : -1;
    doff[i] = e < NV * 4 ? v * P + q * 4 : -1;
  }
  auto gload = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      f4 z = {0.f, 0.f, 0.f, 0.f};
      stage[i] = soff[i] >= 0 ? *reinterpret_cast<const f4*>(x + soff[i] + c0) : z;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (doff[i] >= 0) *reinterpret_cast<f4*>(&lds[buf * NV * P + doff[i]]) = stage[i];
  };

  // ---- per-lane transform geometry: tile m = (tz, ty, tx); wave w = fz picks planes (za, zb) and a sign
  const int tz = m >> 4, ty = (m >> 2) & 3, tx = m & 3;
  const int za = (w == 0) ? 0 : (w == 2 ? 2 : 1);
  const int zb = (w == 0) ? 2 : (w == 1 ? 2 : (w == 2 ? 1 : 3));
  const float sg = (w == 1) ? 1.f : -1.f;
  const int la = (((2 * tz + za) * HY + 2 * ty) * HX + 2 * tx) * P + h * 4;
  const int lb = (((2 * tz + zb) * HY + 2 * ty) * HX + 2 * tx) * P + h * 4;

  // ---- B operand stream
  const int nsub = Cin >> 3;
  const size_t wstride_f = (size_t)nsub * 2 * Cout * 4;                       // floats per frequency
  const float* wbase = wt + (size_t)(w * 16) * wstride_f + (size_t)h * Cout * 4 + (size_t)(n0 + m) * 4;
  f4 wreg[16];
#pragma unroll
  for (int f = 0; f < 16; ++f) wreg[f] = *reinterpret_cast<const f4*>(wbase + f * wstride_f);

  f16v acc[16];
#pragma unroll
  for (int f = 0; f < 16; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  gload(0);
  sstore(0);
  __syncthreads();

  const int nch = Cin / KC;
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nch) gload((ch + 1) * KC);
    const float* L = lds + buf * NV * P;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f4 u[4][4];
      {
        f4 t[4][4];
#pragma unroll
        for (int ix = 0; ix < 4; ++ix) {
          f4 v[4];
#pragma unroll
          for (int iy = 0; iy < 4; ++iy) {
            f4 a = *reinterpret_cast<const f4*>(L + la + (iy * HX + ix) * P + s * 8);
            f4 bb = *reinterpret_cast<const f4*>(L + lb + (iy * HX + ix) * P + s * 8);
            v[iy] = a + sg * bb;
          }
          t[0][ix] = v[0] - v[2];
          t[1][ix] = v[1] + v[2];
          t[2][ix] = v[2] - v[1];
          t[3][ix] = v[1] - v[3];
        }
#pragma unroll
        for (int fy = 0; fy < 4; ++fy) {
          u[fy][0] = t[fy][0] - t[fy][2];
          u[fy][1] = t[fy][1] + t[fy][2];
          u[fy][2] = t[fy][2] - t[fy][1];
          u[fy][3] = t[fy][1] - t[fy][3];
        }
      }
      const int ss = ch * 2 + s + 1;                                   // next sub-step's weights
      const bool more = ss < nsub;
#pragma unroll
      for (int f = 0; f < 16; ++f) {
        const f4 a = u[f >> 2][f & 3];
        const f4 bw = wreg[f];
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bw.x, acc[f], 0, 0, 0);
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bw.y, acc[f], 0, 0, 0);
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bw.z, acc[f], 0, 0, 0);
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bw.w, acc[f], 0, 0, 0);
        if (more) wreg[f] = *reinterpret_cast<const f4*>(wbase + f * wstride_f + (size_t)ss * 2 * Cout * 4);
      }
    }
    if (ch + 1 < nch) sstore(buf ^ 1);
    __syncthreads();
  }

  // ---- output transform.  In-wave: (fy, fx) -> (dy, dx) with A^T = [[1,1,1,0],[0,1,-1,-1]]; across waves: fz -> dz.
  // part[w][r*4 + o][lane], o = dy*2+dx
  float* part = lds;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float q[2][4];   // [dy][fx]
#pragma unroll
    for (int fx = 0; fx < 4; ++fx) {
      q[0][fx] = acc[0 * 4 + fx][r] + acc[1 * 4 + fx][r] + acc[2 * 4 + fx][r];
      q[1][fx] = acc[1 * 4 + fx][r] - acc[2 * 4 + fx][r] - acc[3 * 4 + fx][r];
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      part[(w * 64 + r * 4 + dy * 2 + 0) * 64 + lane] = q[dy][0] + q[dy][1] + q[dy][2];
      part[(w * 64 + r * 4 + dy * 2 + 1) * 64 + lane] = q[dy][1] - q[dy][2] - q[dy][3];
    }
  }
  __syncthreads();
  // wave w finishes accumulator registers r in [4w, 4w+4): rows (tiles) (r/4)*8 + h*4 + (r%4) = w*8 + h*4 + rr
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = w * 4 + rr;
    const int mt = w * 8 + h * 4 + rr;
    const int ttz = mt >> 4, tty = (mt >> 2) & 3, ttx = mt & 3;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const float p0 = part[(0 * 64 + r * 4 + o) * 64 + lane];
      const float p1 = part[(1 * 64 + r * 4 + o) * 64 + lane];
      const float p2 = part[(2 * 64 + r * 4 + o) * 64 + lane];
      const float p3 = part[(3 * 64 + r * 4 + o) * 64 + lane];
      const int vy = oy + 2 * tty + (o >> 1), vx = ox + 2 * ttx + (o & 1);
      const int vz = oz + 2 * ttz;
      const size_t o0 = ((((size_t)b * S + vz) * S + vy) * S + vx) * Cout + n0 + m;
      y[o0] = p0 + p1 + p2;
      y[o0 + (size_t)S * S * Cout] = p1 - p2 - p3;
    }
  }
}


// ---------------------------------------------------------------- v2: 8 waves (2 per SIMD), 8 frequencies per wave
// wave w: fz = w >> 1, fy in {2*(w&1), 2*(w&1)+1}; 8 accumulators (128 AGPRs); the other wave on the SIMD hides the
// transform / load latency.  Branch-free loads (out-of-range halo voxels read a zero page).
__global__ __launch_bounds__(512) void wino_fwd8(const float* __restrict__ x, const float* __restrict__ wt,
                                                 const float* __restrict__ zeros, float* __restrict__ y, int S, int Cin,
                                                 int Cout) {
  __shared__ __attribute__((aligned(16))) float lds[32768];   // 128 KB: staging uses 2*NV*P*4 = 96 000 B, the epilogue all
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m = lane & 31, h = lane >> 5;
  const int nchunks = Cout >> 5;
  const int nb = blockIdx.x % nchunks;
  int tb = blockIdx.x / nchunks;
  const int nbx = S >> 3, nby = S >> 3, nbz = S >> 2;
  const int bx = tb % nbx; tb /= nbx;
  const int by = tb % nby; tb /= nby;
  const int bz = tb % nbz;
  const int b = tb / nbz;
  const int oz = bz * 4, oy = by * 8, ox = bx * 8, n0 = nb * 32;

  constexpr int NLD = (NV * 4 + 511) / 512;   // 5
  f4 stage[NLD];
  const float* sptr[NLD];
  int doff[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int e = tid + i * 512;
    if (e >= NV * 4) e = NV * 4 - 1;          // duplicates of the last slot: harmless
    int v = e >> 2, q = e & 3;
    int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
    int gz = oz - 1 + hz, gy = oy - 1 + hy, gx = ox - 1 + hx;
    bool ok = gz >= 0 && gz < S && gy >= 0 && gy < S && gx >= 0 && gx < S;
    sptr[i] = ok ? x + ((((size_t)b * S + gz) * S + gy) * S + gx) * Cin + q * 4 : zeros;
    doff[i] = v * P + q * 4;
  }
  const int cstep_ok = 1;
  auto gload = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) stage[i] = *reinterpret_cast<const f4*>(sptr[i] + c0);
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) *reinterpret_cast<f4*>(&lds[buf * NV * P + doff[i]]) = stage[i];
  };
  (void)cstep_ok;

  const int fz = w >> 1, fyh = w & 1;
  const int tz = m >> 4, ty = (m >> 2) & 3, tx = m & 3;
  const int za = (fz == 0) ? 0 : (fz == 2 ? 2 : 1);
  const int zb = (fz == 0) ? 2 : (fz == 1 ? 2 : (fz == 2 ? 1 : 3));
  const float sg = (fz == 1) ? 1.f : -1.f;
  const int la = (((2 * tz + za) * HY + 2 * ty) * HX + 2 * tx) * P + h * 4;
  const int lb = (((2 * tz + zb) * HY + 2 * ty) * HX + 2 * tx) * P + h * 4;

  const int nsub = Cin >> 3;
  const size_t wstride_f = (size_t)nsub * 2 * Cout * 4;
  const float* wbase = wt + (size_t)(fz * 16 + fyh * 8) * wstride_f + (size_t)h * Cout * 4 + (size_t)(n0 + m) * 4;
  f4 wreg[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) wreg[f] = *reinterpret_cast<const f4*>(wbase + f * wstride_f);

  f16v acc[8];
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  gload(0);
  sstore(0);
  __syncthreads();

  const int nch = Cin / KC;
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    // the zero page is Cin floats long, so the chunk offset applies to it too; past the end re-read the last chunk
    gload((ch + 1 < nch ? ch + 1 : ch) * KC);
    const float* L = lds + buf * NV * P;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f4 u[2][4];
      {
        f4 t[2][4];
#pragma unroll
        for (int ix = 0; ix < 4; ++ix) {
          f4 v[4];
#pragma unroll
          for (int iy = 0; iy < 4; ++iy) {
            f4 a = *reinterpret_cast<const f4*>(L + la + (iy * HX + ix) * P + s * 8);
            f4 bb = *reinterpret_cast<const f4*>(L + lb + (iy * HX + ix) * P + s * 8);
            v[iy] = a + sg * bb;
          }
          // fy rows of B^T: 0: v0-v2, 1: v1+v2, 2: v2-v1, 3: v1-v3
          t[0][ix] = fyh ? v[2] - v[1] : v[0] - v[2];
          t[1][ix] = fyh ? v[1] - v[3] : v[1] + v[2];
        }
#pragma unroll
        for (int fy = 0; fy < 2; ++fy) {
          u[fy][0] = t[fy][0] - t[fy][2];
          u[fy][1] = t[fy][1] + t[fy][2];
          u[fy][2] = t[fy][2] - t[fy][1];
          u[fy][3] = t[fy][1] - t[fy][3];
        }
      }
      int ss = ch * 2 + s + 1;
      ss = ss < nsub ? ss : nsub - 1;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const f4 a = u[f >> 2][f & 3];
        const f4 bw = wreg[f];
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bw.x, acc[f], 0, 0, 0);
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bw.y, acc[f], 0, 0, 0);
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bw.z, acc[f], 0, 0, 0);
        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bw.w, acc[f], 0, 0, 0);
        wreg[f] = *reinterpret_cast<const f4*>(wbase + f * wstride_f + (size_t)ss * 2 * Cout * 4);
      }
    }
    sstore(buf ^ 1);
    __syncthreads();
  }

  // ---- output transform: x in registers (fx -> dx), this wave's two fy rows -> its share of (dy0, dy1);
  // part[w][r*4 + dy*2 + dx][lane]; the sum over the 8 waves carries the z signs
  float* part = lds;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float q0[2], q1[2];   // [dx] for local fy 0 / 1
    q0[0] = acc[0][r] + acc[1][r] + acc[2][r];
    q0[1] = acc[1][r] - acc[2][r] - acc[3][r];
    q1[0] = acc[4][r] + acc[5][r] + acc[6][r];
    q1[1] = acc[5][r] - acc[6][r] - acc[7][r];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      // fyh=0 holds fy 0,1: dy0 += q0+q1, dy1 += q1;  fyh=1 holds fy 2,3: dy0 += q0, dy1 += -q0-q1
      const float d0 = fyh ? q0[dx] : q0[dx] + q1[dx];
      const float d1 = fyh ? -q0[dx] - q1[dx] : q1[dx];
      part[(w * 64 + r * 4 + 0 + dx) * 64 + lane] = d0;
      part[(w * 64 + r * 4 + 2 + dx) * 64 + lane] = d1;
    }
  }
  __syncthreads();
  // wave w finishes accumulator registers r = 2w, 2w+1: tiles (r/4)*8 + h*4 + (r%4)
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int r = w * 2 + rr;
    const int mt = (r >> 2) * 8 + h * 4 + (r & 3);
    const int ttz = mt >> 4, tty = (mt >> 2) & 3, ttx = mt & 3;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float p[4];
#pragma unroll
      for (int z = 0; z < 4; ++z)
        p[z] = part[((2 * z) * 64 + r * 4 + o) * 64 + lane] + part[((2 * z + 1) * 64 + r * 4 + o) * 64 + lane];
      const int vy = oy + 2 * tty + (o >> 1), vx = ox + 2 * ttx + (o & 1);
      const int vz = oz + 2 * ttz;
      const size_t o0 = ((((size_t)b * S + vz) * S + vy) * S + vx) * Cout + n0 + m;
      y[o0] = p[0] + p[1] + p[2];
      y[o0 + (size_t)S * S * Cout] = p[1] - p[2] - p[3];
    }
  }
}


// ---------------------------------------------------------------- v3: 4 waves, 2 channels per lane per sub-step,
// software pipelined: while the 32 MFMAs of sub-step g run, the wave reads and transforms sub-step g+1.
// LDS layout: voxel pitch 18 floats, row pitch 10 voxels, plane pitch 104 voxels, the two channel pairs of every
// 4-channel group swapped where (hy >> 1) is odd -> the 32 tiles of a half-wave hit 32 distinct bank pairs.
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int P3 = 18, PY3 = 10, PZ3 = 104, BUF3 = HZ * PZ3 * P3;      // floats per buffer (11232 = 44 928 B)
// wt3 layout: [64 f][Cin/4][2 h][Cout][2 j]   (channel = c4*4 + h*2 + j)
template <int EXP>
__global__ __launch_bounds__(256) void wino_fwd3(const float* __restrict__ x, const float* __restrict__ wt,
                                                 const float* __restrict__ zeros, float* __restrict__ y, int S, int Cin,
                                                 int Cout, int flags) {
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF3];   // 89 856 B; the epilogue reuses 64 KB of it
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m = lane & 31, h = lane >> 5;
  const int nchunks = Cout >> 5;
  const int nb = blockIdx.x % nchunks;
  int tb = blockIdx.x / nchunks;
  const int nbx = S >> 3, nby = S >> 3, nbz = S >> 2;
  const int bx = tb % nbx; tb /= nbx;
  const int by = tb % nby; tb /= nby;
  const int bz = tb % nbz;
  const int b = tb / nbz;
  const int oz = bz * 4, oy = by * 8, ox = bx * 8, n0 = nb * 32;

  constexpr int NLD = (NV * 4 + 255) / 256;   // 10
  f4 stage[NLD];
  const float* sptr[NLD];
  int doff[NLD];                               // LDS float offset of the 4-channel group | swap flag in bit 0
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int e = tid + i * 256;
    if (e >= NV * 4) e = NV * 4 - 1;
    int v = e >> 2, q = e & 3;
    int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
    int gz = oz - 1 + hz, gy = oy - 1 + hy, gx = ox - 1 + hx;
    bool ok = gz >= 0 && gz < S && gy >= 0 && gy < S && gx >= 0 && gx < S;
    sptr[i] = (ok && !(flags & 1)) ? x + ((((size_t)b * S + gz) * S + gy) * S + gx) * Cin + q * 4 : zeros;
    doff[i] = ((hz * PZ3 + hy * PY3 + hx) * P3 + q * 4) | ((hy >> 1) & 1);
  }
  auto gload = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) stage[i] = *reinterpret_cast<const f4*>(sptr[i] + c0);
  };
  auto sstore = [&](int bo) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int sw = (doff[i] & 1) * 2, base = (doff[i] & ~1) + bo;
      f2 lo = {stage[i].x, stage[i].y}, hi = {stage[i].z, stage[i].w};
      *reinterpret_cast<f2*>(&lds[base + sw]) = lo;
      *reinterpret_cast<f2*>(&lds[base + 2 - sw]) = hi;
    }
  };

  const int tz = m >> 4, ty = (m >> 2) & 3, tx = m & 3;
  const int za = (w == 0) ? 0 : (w == 2 ? 2 : 1);
  const int zb = (w == 0) ? 2 : (w == 1 ? 2 : (w == 2 ? 1 : 3));
  const float sg = (w == 1) ? 1.f : -1.f;
  const int la = ((2 * tz + za) * PZ3 + 2 * ty * PY3 + 2 * tx) * P3;
  const int lb = ((2 * tz + zb) * PZ3 + 2 * ty * PY3 + 2 * tx) * P3;
  const int hs0 = 2 * (h ^ (ty & 1)), hs1 = 2 * (h ^ ((ty + 1) & 1));     // iy in {0,1} / {2,3}
  const int A0 = la + hs0, A1 = la + hs1, B0 = lb + hs0, B1 = lb + hs1;

  const int nsub = Cin >> 2;
  const size_t wstride_f = (size_t)nsub * 2 * Cout * 2;                     // floats per frequency
  const size_t wsub = (size_t)2 * Cout * 2;                                 // floats per sub-step
  const float* wbase = wt + (size_t)(w * 16) * wstride_f + (size_t)h * Cout * 2 + (size_t)(n0 + m) * 2;
  f2 wreg[16];
#pragma unroll
  for (int f = 0; f < 16; ++f) wreg[f] = *reinterpret_cast<const f2*>(wbase + f * wstride_f);

  f16v acc[16];
#pragma unroll
  for (int f = 0; f < 16; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  f2 u[16];
  // transform of (buffer offset bo, sub-step sub) -> u, one column at a time
  auto column = [&](int bo, int sub, int g, f2 (&tn)[4][4]) {
    f2 v[4];
#pragma unroll
    for (int iy = 0; iy < 4; ++iy) {
      const int off = (iy * PY3 + g) * P3 + 4 * sub + bo;
      f2 a = *reinterpret_cast<const f2*>(&lds[(iy < 2 ? A0 : A1) + off]);
      f2 bb = *reinterpret_cast<const f2*>(&lds[(iy < 2 ? B0 : B1) + off]);
      v[iy] = a + sg * bb;
    }
    tn[0][g] = v[0] - v[2];
    tn[1][g] = v[1] + v[2];
    tn[2][g] = v[2] - v[1];
    tn[3][g] = v[1] - v[3];
  };
  auto xform = [&](f2 (&tn)[4][4]) {
#pragma unroll
    for (int fy = 0; fy < 4; ++fy) {
      u[fy * 4 + 0] = tn[fy][0] - tn[fy][2];
      u[fy * 4 + 1] = tn[fy][1] + tn[fy][2];
      u[fy * 4 + 2] = tn[fy][2] - tn[fy][1];
      u[fy * 4 + 3] = tn[fy][1] - tn[fy][3];
    }
  };

  gload(0);
  sstore(0);
  __syncthreads();
  {
    f2 tn[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) column(0, 0, g, tn);
    xform(tn);
  }

  const int nch = Cin / KC;
  for (int ch = 0; ch < nch; ++ch) {
    const int cur = (ch & 1) * BUF3, nxt = BUF3 - cur;
    if (!(EXP & 8)) gload((ch + 1 < nch ? ch + 1 : ch) * KC);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s == 3) {
        if (!(EXP & 8)) sstore(nxt);
        __syncthreads();
      }
      int gs = ch * 4 + s + 1;                       // next global sub-step (weights)
      gs = gs < nsub ? gs : nsub - 1;
      const float* wn = wbase + (size_t)gs * wsub;
      const int bo = (s == 3) ? nxt : cur;
      const int sn = (s + 1) & 3;
      f2 tn[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int fx = 0; fx < 4; ++fx) {
          const int f = g * 4 + fx;
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[f].x, wreg[f].x, acc[f], 0, 0, 0);
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[f].y, wreg[f].y, acc[f], 0, 0, 0);
          if (!(EXP & 2)) wreg[f] = *reinterpret_cast<const f2*>(wn + f * wstride_f);
        }
        if (!(EXP & 4)) column(bo, sn, g, tn);
        else { tn[0][g] = u[g]; tn[1][g] = u[g + 4]; tn[2][g] = u[g + 8]; tn[3][g] = u[g + 12]; }
      }
      xform(tn);
    }
  }

  float* part = lds;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float q[2][4];
#pragma unroll
    for (int fx = 0; fx < 4; ++fx) {
      q[0][fx] = acc[0 * 4 + fx][r] + acc[1 * 4 + fx][r] + acc[2 * 4 + fx][r];
      q[1][fx] = acc[1 * 4 + fx][r] - acc[2 * 4 + fx][r] - acc[3 * 4 + fx][r];
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      part[(w * 64 + r * 4 + dy * 2 + 0) * 64 + lane] = q[dy][0] + q[dy][1] + q[dy][2];
      part[(w * 64 + r * 4 + dy * 2 + 1) * 64 + lane] = q[dy][1] - q[dy][2] - q[dy][3];
    }
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = w * 4 + rr;
    const int mt = w * 8 + h * 4 + rr;
    const int ttz = mt >> 4, tty = (mt >> 2) & 3, ttx = mt & 3;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const float p0 = part[(0 * 64 + r * 4 + o) * 64 + lane];
      const float p1 = part[(1 * 64 + r * 4 + o) * 64 + lane];
      const float p2 = part[(2 * 64 + r * 4 + o) * 64 + lane];
      const float p3 = part[(3 * 64 + r * 4 + o) * 64 + lane];
      const int vy = oy + 2 * tty + (o >> 1), vx = ox + 2 * ttx + (o & 1);
      const int vz = oz + 2 * ttz;
      const size_t o0 = ((((size_t)b * S + vz) * S + vy) * S + vx) * Cout + n0 + m;
      y[o0] = p0 + p1 + p2;
      y[o0 + (size_t)S * S * Cout] = p1 - p2 - p3;
    }
  }
}


// ---------------------------------------------------------------- v4: v3 with the transform stream hand-interleaved
// into the MFMA stream: in every group of 8 MFMAs the wave issues the 8 LDS reads of the NEXT column, and does the 8
// packed VALU ops of the column read one group earlier, one op after each MFMA.
template <int PIN, int EXP>
__global__ __launch_bounds__(256) void wino_fwd4(const float* __restrict__ x, const float* __restrict__ wt,
                                                 const float* __restrict__ zeros, float* __restrict__ y, int S, int Cin,
                                                 int Cout) {
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF3];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m = lane & 31, h = lane >> 5;
  const int nchunks = Cout >> 5;
  const int nb = blockIdx.x % nchunks;
  int tb = blockIdx.x / nchunks;
  const int nbx = S >> 3, nby = S >> 3, nbz = S >> 2;
  const int bx = tb % nbx; tb /= nbx;
  const int by = tb % nby; tb /= nby;
  const int bz = tb % nbz;
  const int b = tb / nbz;
  const int oz = bz * 4, oy = by * 8, ox = bx * 8, n0 = nb * 32;

  constexpr int NLD = (NV * 4 + 255) / 256;   // 10
  f4 stage[NLD];
  int soff[NLD];                               // float offset into x, or -1: zero page
  int doff[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int e = tid + i * 256;
    if (e >= NV * 4) e = NV * 4 - 1;
    int v = e >> 2, q = e & 3;
    int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
    int gz = oz - 1 + hz, gy = oy - 1 + hy, gx = ox - 1 + hx;
    bool ok = gz >= 0 && gz < S && gy >= 0 && gy < S && gx >= 0 && gx < S;
    soff[i] = ok ? ((((b * S + gz) * S + gy) * S + gx) * Cin + q * 4) : -1;
    doff[i] = ((hz * PZ3 + hy * PY3 + hx) * P3 + q * 4) | ((hy >> 1) & 1);
  }
  auto gload = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const float* p = soff[i] >= 0 ? x + soff[i] + c0 : zeros;
      stage[i] = *reinterpret_cast<const f4*>(p);
    }
  };
  auto sstore = [&](int bo) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int sw = (doff[i] & 1) * 2, base = (doff[i] & ~1) + bo;
      f2 lo = {stage[i].x, stage[i].y}, hi = {stage[i].z, stage[i].w};
      *reinterpret_cast<f2*>(&lds[base + sw]) = lo;
      *reinterpret_cast<f2*>(&lds[base + 2 - sw]) = hi;
    }
  };

  const int tz = m >> 4, ty = (m >> 2) & 3, tx = m & 3;
  const int za = (w == 0) ? 0 : (w == 2 ? 2 : 1);
  const int zb = (w == 0) ? 2 : (w == 1 ? 2 : (w == 2 ? 1 : 3));
  const float sg = (w == 1) ? 1.f : -1.f;
  const int la = ((2 * tz + za) * PZ3 + 2 * ty * PY3 + 2 * tx) * P3;
  const int lb = ((2 * tz + zb) * PZ3 + 2 * ty * PY3 + 2 * tx) * P3;
  const int hs0 = 2 * (h ^ (ty & 1)), hs1 = 2 * (h ^ ((ty + 1) & 1));
  const int A0 = la + hs0, A1 = la + hs1, B0 = lb + hs0, B1 = lb + hs1;

  const int nsub = Cin >> 2;
  // wt4 layout: [Cout/32][Cin/4 ss][64 f][2 h][32 n][2 j]: one sub-step of one wave is 8 KB contiguous (the 64 KB
  // frequency stride of the v3 layout put every load of every CU on the same L2 channel)
  constexpr int wstride_f = 128;
  constexpr int wsub = 64 * 128;
  const float* wbase = wt + ((size_t)nb * nsub * 64 + w * 16) * 128 + h * 64 + m * 2;
  f2 wreg[16];
#pragma unroll
  for (int f = 0; f < 16; ++f) wreg[f] = *reinterpret_cast<const f2*>(wbase + f * wstride_f);

  f16v acc[16];
#pragma unroll
  for (int f = 0; f < 16; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  f2 u[16], tn[4][4], ra[4], rb[4];
  auto rd = [&](int bo, int sub, int col) {
#pragma unroll
    for (int iy = 0; iy < 4; ++iy) {
      const int off = (iy * PY3 + col) * P3 + 4 * sub + bo;
      ra[iy] = *reinterpret_cast<const f2*>(&lds[(iy < 2 ? A0 : A1) + off]);
      rb[iy] = *reinterpret_cast<const f2*>(&lds[(iy < 2 ? B0 : B1) + off]);
    }
  };
  auto colmath = [&](int g) {
    f2 v0 = ra[0] + sg * rb[0], v1 = ra[1] + sg * rb[1], v2 = ra[2] + sg * rb[2], v3 = ra[3] + sg * rb[3];
    tn[0][g] = v0 - v2;
    tn[1][g] = v1 + v2;
    tn[2][g] = v2 - v1;
    tn[3][g] = v1 - v3;
  };
  auto xform = [&]() {
#pragma unroll
    for (int fy = 0; fy < 4; ++fy) {
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
    colmath(g);
  }
  xform();
  rd(0, 1, 0);                                   // column 0 of sub-step 1

  const int nch = Cin / KC;
  for (int ch = 0; ch < nch; ++ch) {
    const int cur = (ch & 1) * BUF3, nxt = BUF3 - cur;
    if (!(EXP & 8)) gload((EXP & 128) ? 0 : (ch + 1 < nch ? ch + 1 : ch) * KC);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      int gs = ch * 4 + s + 1;
      gs = gs < nsub ? gs : nsub - 1;
      if (EXP & 64) gs = 0;
      const float* wn = wbase + gs * wsub;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // v/tn math of column g (read one group ago) interleaved with this group's 8 MFMAs
        f2 v0, v1, v2, v3;
        f2 a0 = ra[0], a1 = ra[1], a2 = ra[2], a3 = ra[3], b0 = rb[0], b1 = rb[1], b2 = rb[2], b3 = rb[3];
        if (s == 2 && g == 3) {                  // next chunk must be visible before its first column is read
          if (!(EXP & 8)) sstore(nxt);
          __syncthreads();
        }
        // reads of the next column: column g+1 of sub-step gs+1, or column 0 of sub-step gs+2
        // EXP & 256: spread the LDS reads and W loads over the MFMA gaps instead of issuing them in bursts
        const int rbo = (g < 3) ? (s == 3 ? nxt : cur) : (s >= 2 ? nxt : cur);
        const int rsub = (g < 3) ? ((s + 1) & 3) : ((s + 2) & 3);
        const int rcol = (g < 3) ? g + 1 : 0;
        if (!(EXP & 4) && !(EXP & 256)) rd(rbo, rsub, rcol);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
#define MF(F, C) acc[F] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[F].C, wreg[F].C, acc[F], 0, 0, 0)
#define FENCE if (PIN) __builtin_amdgcn_sched_barrier(0)
#define RD1(IY)                                                                                              \
  if (EXP & 256) {                                                                                           \
    const int off = (IY * PY3 + rcol) * P3 + 4 * rsub + rbo;                                                  \
    ra[IY] = *reinterpret_cast<const f2*>(&lds[(IY < 2 ? A0 : A1) + off]);                                    \
    rb[IY] = *reinterpret_cast<const f2*>(&lds[(IY < 2 ? B0 : B1) + off]);                                    \
  }
#define WL1(FX)                                                                                              \
  if ((EXP & 256) && !(EXP & 2)) wreg[g * 4 + FX] = *reinterpret_cast<const f2*>(wn + (g * 4 + FX) * wstride_f)
        MF(g * 4 + 0, x); v0 = a0 + sg * b0; RD1(0); FENCE;
        MF(g * 4 + 1, x); v1 = a1 + sg * b1; RD1(1); FENCE;
        MF(g * 4 + 2, x); v2 = a2 + sg * b2; RD1(2); FENCE;
        MF(g * 4 + 3, x); v3 = a3 + sg * b3; RD1(3); FENCE;
        MF(g * 4 + 0, y); tn[0][g] = v0 - v2; WL1(0); FENCE;
        MF(g * 4 + 1, y); tn[1][g] = v1 + v2; WL1(1); FENCE;
        MF(g * 4 + 2, y); tn[2][g] = v2 - v1; WL1(2); FENCE;
        MF(g * 4 + 3, y); tn[3][g] = v1 - v3; WL1(3); FENCE;
#pragma unroll
        for (int fx = 0; fx < 4; ++fx)
          if (!(EXP & 2) && !(EXP & 256)) wreg[g * 4 + fx] = *reinterpret_cast<const f2*>(wn + (g * 4 + fx) * wstride_f);
        FENCE;
      }
      if (!(EXP & 16)) xform();
    }
  }

  float* part = lds;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float q[2][4];
#pragma unroll
    for (int fx = 0; fx < 4; ++fx) {
      q[0][fx] = acc[0 * 4 + fx][r] + acc[1 * 4 + fx][r] + acc[2 * 4 + fx][r];
      q[1][fx] = acc[1 * 4 + fx][r] - acc[2 * 4 + fx][r] - acc[3 * 4 + fx][r];
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      part[(w * 64 + r * 4 + dy * 2 + 0) * 64 + lane] = q[dy][0] + q[dy][1] + q[dy][2];
      part[(w * 64 + r * 4 + dy * 2 + 1) * 64 + lane] = q[dy][1] - q[dy][2] - q[dy][3];
    }
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = w * 4 + rr;
    const int mt = w * 8 + h * 4 + rr;
    const int ttz = mt >> 4, tty = (mt >> 2) & 3, ttx = mt & 3;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const float p0 = part[(0 * 64 + r * 4 + o) * 64 + lane];
      const float p1 = part[(1 * 64 + r * 4 + o) * 64 + lane];
      const float p2 = part[(2 * 64 + r * 4 + o) * 64 + lane];
      const float p3 = part[(3 * 64 + r * 4 + o) * 64 + lane];
      const int vy = oy + 2 * tty + (o >> 1), vx = ox + 2 * ttx + (o & 1);
      const int vz = oz + 2 * ttz;
      const size_t o0 = ((((size_t)b * S + vz) * S + vy) * S + vx) * Cout + n0 + m;
      y[o0] = p0 + p1 + p2;
      y[o0 + (size_t)S * S * Cout] = p1 - p2 - p3;
    }
  }
}


// ---------------------------------------------------------------- v6: v5 on a VALU diet.  The fp32 MFMA shares the
// SIMD's VALU issue (SQ_VALU_MFMA_COEXEC_CYCLES = 0: every VALU instruction is MFMA time lost), so: W loads take a
// scalar base + per-lane offset + immediates (no 64-bit pointer arithmetic), LDS addresses are VGPR + immediate (chunk
// loop unrolled by two so the buffer offset is static), halo zero-fill is a select only in boundary blocks, and the
// epilogue reads float4 and stores float4.
template <int EXP>
__global__ __launch_bounds__(256) void wino_fwd6(const float* __restrict__ x, const float* __restrict__ wt,
                                                 float* __restrict__ y, int S, int Cin, int Cout) {
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF3];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  const int nchunks = Cout >> 5;
  const int nb = blockIdx.x % nchunks;
  int tb = blockIdx.x / nchunks;
  const int nbx = S >> 3, nby = S >> 3, nbz = S >> 2;
  const int bx = tb % nbx; tb /= nbx;
  const int by = tb % nby; tb /= nby;
  const int bz = tb % nbz;
  const int b = tb / nbz;
  const int oz = bz * 4, oy = by * 8, ox = bx * 8, n0 = nb * 32;
  const bool edge = bx == 0 || by == 0 || bz == 0 || bx == nbx - 1 || by == nby - 1 || bz == nbz - 1;   // uniform

  constexpr int NLD = (NV * 4 + 255) / 256;   // 10
  f4 stage[NLD];
  int soff[NLD];                               // float offset into x of the (clamped) voxel + channel quad
  int dlo[NLD], dhi[NLD];                      // LDS float offsets of the two channel pairs
  unsigned okmask = 0;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    int e = tid + i * 256;
    if (e >= NV * 4) e = NV * 4 - 1;
    int v = e >> 2, q = e & 3;
    int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
    int gz = oz - 1 + hz, gy = oy - 1 + hy, gx = ox - 1 + hx;
    bool ok = gz >= 0 && gz < S && gy >= 0 && gy < S && gx >= 0 && gx < S;
    okmask |= ok ? (1u << i) : 0u;
    gz = min(max(gz, 0), S - 1); gy = min(max(gy, 0), S - 1); gx = min(max(gx, 0), S - 1);
    soff[i] = (((b * S + gz) * S + gy) * S + gx) * Cin + q * 4;
    const int base = (hz * PZ3 + hy * PY3 + hx) * P3 + q * 4, sw = ((hy >> 1) & 1) * 2;
    dlo[i] = base + sw;
    dhi[i] = base + 2 - sw;
  }
  auto gload = [&](int c0) {
    const float* xc = x + c0;                  // uniform base
#pragma unroll
    for (int i = 0; i < NLD; ++i) stage[i] = *reinterpret_cast<const f4*>(xc + soff[i]);
  };
  auto sstore = [&](const int bo) {            // bo: compile-time after unrolling
    if (edge) {
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        if (!((okmask >> i) & 1)) stage[i] = f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      f2 lo = {stage[i].x, stage[i].y}, hi = {stage[i].z, stage[i].w};
      *reinterpret_cast<f2*>(&lds[bo + dlo[i]]) = lo;
      *reinterpret_cast<f2*>(&lds[bo + dhi[i]]) = hi;
    }
  };

  const int tz = m >> 4, ty = (m >> 2) & 3, tx = m & 3;
  const int za = (w == 0) ? 0 : (w == 2 ? 2 : 1);
  const int zb = (w == 0) ? 2 : (w == 1 ? 2 : (w == 2 ? 1 : 3));
  const float sg = (w == 1) ? 1.f : -1.f;
  const int la = ((2 * tz + za) * PZ3 + 2 * ty * PY3 + 2 * tx) * P3;
  const int lb = ((2 * tz + zb) * PZ3 + 2 * ty * PY3 + 2 * tx) * P3;
  const int hs0 = 2 * (h ^ (ty & 1)), hs1 = 2 * (h ^ ((ty + 1) & 1));
  const int A0 = la + hs0, A1 = la + hs1, B0 = lb + hs0, B1 = lb + hs1;

  const int nsub = Cin >> 2;
  constexpr int wstride_f = 128;               // wt4 layout [Cout/32][Cin/4][64 f][2 h][32 n][2]
  constexpr int wsub = 64 * 128;
  const float* wu = wt + ((size_t)nb * nsub * 64 + w * 16) * 128;     // uniform
  const int wlane = h * 64 + m * 2;
  f2 wreg[16];
#pragma unroll
  for (int f = 0; f < 16; ++f) wreg[f] = *reinterpret_cast<const f2*>(wu + f * wstride_f + wlane);

  f16v acc[16];
#pragma unroll
  for (int f = 0; f < 16; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  f2 u[16], tn[4][4], ra[4], rb[4];
  auto rd = [&](const int bo, const int sub, const int col) {
#pragma unroll
    for (int iy = 0; iy < 4; ++iy) {
      const int off = (iy * PY3 + col) * P3 + 4 * sub + bo;
      ra[iy] = *reinterpret_cast<const f2*>(&lds[(iy < 2 ? A0 : A1) + off]);
      rb[iy] = *reinterpret_cast<const f2*>(&lds[(iy < 2 ? B0 : B1) + off]);
    }
  };
  auto colmath = [&](int g) {
    f2 v0 = ra[0] + sg * rb[0], v1 = ra[1] + sg * rb[1], v2 = ra[2] + sg * rb[2], v3 = ra[3] + sg * rb[3];
    tn[0][g] = v0 - v2;
    tn[1][g] = v1 + v2;
    tn[2][g] = v2 - v1;
    tn[3][g] = v1 - v3;
  };
  auto xform = [&]() {
#pragma unroll
    for (int fy = 0; fy < 4; ++fy) {
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
    colmath(g);
  }
  xform();
  rd(0, 1, 0);

  const int nch = Cin / KC;
  auto chunk = [&](const int ch, const int cur, const int nxt) {
    gload((ch + 1 < nch ? ch + 1 : ch) * KC);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      int gs = ch * 4 + s + 1;
      gs = gs < nsub ? gs : nsub - 1;
      const float* wn = wu + (size_t)gs * wsub;          // uniform
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f2 v0, v1, v2, v3;
        f2 a0 = ra[0], a1 = ra[1], a2 = ra[2], a3 = ra[3], b0 = rb[0], b1 = rb[1], b2 = rb[2], b3 = rb[3];
        if (s == 2 && g == 3) {
          sstore(nxt);
          __syncthreads();
        }
        if (g < 3) rd(s == 3 ? nxt : cur, (s + 1) & 3, g + 1);
        else rd(s >= 2 ? nxt : cur, (s + 2) & 3, 0);
        __builtin_amdgcn_sched_barrier(0);
#define MF6(F, C) acc[F] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[F].C, wreg[F].C, acc[F], 0, 0, 0)
#define FN6 __builtin_amdgcn_sched_barrier(0)
        MF6(g * 4 + 0, x); v0 = a0 + sg * b0; FN6;
        MF6(g * 4 + 1, x); v1 = a1 + sg * b1; FN6;
        MF6(g * 4 + 2, x); v2 = a2 + sg * b2; FN6;
        MF6(g * 4 + 3, x); v3 = a3 + sg * b3; FN6;
        MF6(g * 4 + 0, y); tn[0][g] = v0 - v2; FN6;
        MF6(g * 4 + 1, y); tn[1][g] = v1 + v2; FN6;
        MF6(g * 4 + 2, y); tn[2][g] = v2 - v1; FN6;
        MF6(g * 4 + 3, y); tn[3][g] = v1 - v3; FN6;
#pragma unroll
        for (int fx = 0; fx < 4; ++fx)
          wreg[g * 4 + fx] = *reinterpret_cast<const f2*>(wn + (g * 4 + fx) * wstride_f + wlane);
        FN6;
      }
      xform();
    }
  };
  for (int ch = 0; ch < nch; ch += 2) {
    chunk(ch, 0, BUF3);
    chunk(ch + 1, BUF3, 0);
  }

  // ---- epilogue
  float* part = lds;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float q[2][4];
#pragma unroll
    for (int fx = 0; fx < 4; ++fx) {
      q[0][fx] = acc[0 * 4 + fx][r] + acc[1 * 4 + fx][r] + acc[2 * 4 + fx][r];
      q[1][fx] = acc[1 * 4 + fx][r] - acc[2 * 4 + fx][r] - acc[3 * 4 + fx][r];
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      part[(w * 64 + r * 4 + dy * 2 + 0) * 64 + lane] = q[dy][0] + q[dy][1] + q[dy][2];
      part[(w * 64 + r * 4 + dy * 2 + 1) * 64 + lane] = q[dy][1] - q[dy][2] - q[dy][3];
    }
  }
  __syncthreads();
  // thread task (i = 0..3): n-quad k = tid & 7, q = (tid >> 3) + 32 i: h = q & 1, o = (q >> 1) & 3, r = q >> 3
  {
    const int k = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = (tid >> 3) + 32 * i;
      const int hh = q & 1, o = (q >> 1) & 3, r = q >> 3;
      const int mt = (r >> 2) * 8 + hh * 4 + (r & 3);
      const int ttz = mt >> 4, tty = (mt >> 2) & 3, ttx = mt & 3;
      const int slot = (r * 4 + o) * 64 + hh * 32 + 4 * k;
      const f4 p0 = *reinterpret_cast<const f4*>(&part[0 * 4096 + slot]);
      const f4 p1 = *reinterpret_cast<const f4*>(&part[1 * 4096 + slot]);
      const f4 p2 = *reinterpret_cast<const f4*>(&part[2 * 4096 + slot]);
      const f4 p3 = *reinterpret_cast<const f4*>(&part[3 * 4096 + slot]);
      const int vy = oy + 2 * tty + (o >> 1), vx = ox + 2 * ttx + (o & 1), vz = oz + 2 * ttz;
      const size_t o0 = ((((size_t)b * S + vz) * S + vy) * S + vx) * Cout + n0 + 4 * k;
      *reinterpret_cast<f4*>(y + o0) = p0 + p1 + p2;
      *reinterpret_cast<f4*>(y + o0 + (size_t)S * S * Cout) = p1 - p2 - p3;
    }
  }
}


// ---------------------------------------------------------------- v7: 8 waves (2 per SIMD: VALU 3.6 instead of 5.2
// cycles, MFMA 69.5 instead of 75), z-combination done ONCE at staging time (8 combined planes (tz, fz) in LDS instead
// of 6 raw ones), wave w: fz = w >> 1, fy rows {2 (w&1), 2 (w&1) + 1}: 12 LDS reads + 32 VALU per 16 MFMAs.
__device__ unsigned long long* g_tl;   // EXP & 64: per-wave wall-clock stamps (100 MHz)
constexpr int P7 = 18, PY7 = 10, PZ7 = 100, BUF7 = 8 * PZ7 * P7;          // floats per buffer (57 600 B)
template <int EXP>
__global__ __launch_bounds__(512) void wino_fwd7(const float* __restrict__ x, const float* __restrict__ wt,
                                                 float* __restrict__ y, int S, int Cin, int Cout) {
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF7];   // 115 200 B
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned long long tl0 = 0, tl1 = 0, tl2 = 0;
  if (EXP & 64) tl0 = wall_clock64();
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fz = w >> 1, fyh = w & 1;
  const int m = lane & 31, h = lane >> 5;
  const int nchunks = Cout >> 5;
  const int nb = blockIdx.x % nchunks;
  int tb = blockIdx.x / nchunks;
  const int nbx = S >> 3, nby = S >> 3, nbz = S >> 2;
  const int bx = tb % nbx; tb /= nbx;
  const int by = tb % nby; tb /= nby;
  const int bz = tb % nbz;
  const int b = tb / nbz;
  const int oz = bz * 4, oy = by * 8, ox = bx * 8, n0 = nb * 32;
  const bool edge = bx == 0 || by == 0 || bz == 0 || bx == nbx - 1 || by == nby - 1 || bz == nbz - 1;

  // ---- staging: thread t < 400 owns (hy, hx, q): six raw z values -> eight combined planes
  const int cmb = tid < 400 ? tid : 399;
  const int q = cmb & 3, hx = (cmb >> 2) % HX, hy = (cmb >> 2) / HX;
  f4 stage[6];
  int soff[6];
  unsigned okmask = 0;
  {
    const int gy = oy - 1 + hy, gx = ox - 1 + hx;
    const bool okyx = gy >= 0 && gy < S && gx >= 0 && gx < S;
    const int cy = min(max(gy, 0), S - 1), cx = min(max(gx, 0), S - 1);
#pragma unroll
    for (int hz = 0; hz < 6; ++hz) {
      const int gz = oz - 1 + hz;
      const bool ok = okyx && gz >= 0 && gz < S;
      okmask |= ok ? (1u << hz) : 0u;
      const int cz = min(max(gz, 0), S - 1);
      soff[hz] = (EXP & 16) ? q * 4 : (((b * S + cz) * S + cy) * S + cx) * Cin + q * 4;
    }
  }
  const int sw = ((hy >> 1) & 1) * 2;
  const int dbase = (hy * PY7 + hx) * P7 + q * 4;           // + plane * PZ7 * P7
  auto gload = [&](int c0) {
    const float* xc = x + c0;
#pragma unroll
    for (int i = 0; i < 6; ++i) stage[i] = *reinterpret_cast<const f4*>(xc + soff[i]);
  };
  auto sstore = [&](const int bo) {
    if (edge) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if (!((okmask >> i) & 1)) stage[i] = f4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < 400) {
#pragma unroll
      for (int tz = 0; tz < 2; ++tz) {
        const f4 d0 = stage[2 * tz], d1 = stage[2 * tz + 1], d2 = stage[2 * tz + 2], d3 = stage[2 * tz + 3];
        const f4 c[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int o = bo + (tz * 4 + f) * (PZ7 * P7) + dbase;
          *reinterpret_cast<f2*>(&lds[o + sw]) = f2{c[f].x, c[f].y};
          *reinterpret_cast<f2*>(&lds[o + 2 - sw]) = f2{c[f].z, c[f].w};
        }
      }
    }
  };

  // ---- per-lane read geometry: rows (a, b, cc) of the y transform for this wave's two fy rows
  const int tz = m >> 4, ty = (m >> 2) & 3, tx = m & 3;
  const int ra_ = fyh ? 2 : 0, rb_ = fyh ? 1 : 2, rc_ = fyh ? 3 : 1;
  const float sg = fyh ? -1.f : 1.f;                       // t1 = R_b + sg * R_c
  auto rowbase = [&](int iy) {
    const int hyy = 2 * ty + iy;
    return ((tz * 4 + fz) * PZ7 + hyy * PY7 + 2 * tx) * P7 + 2 * (h ^ ((hyy >> 1) & 1));
  };
  const int Ra = rowbase(ra_), Rb = rowbase(rb_), Rc = rowbase(rc_);

  const int nsub = Cin >> 2;
  constexpr int wstride_f = 128;
  constexpr int wsub = 64 * 128;
  const float* wu = wt + ((size_t)nb * nsub * 64 + fz * 16 + fyh * 8) * 128;       // uniform
  const int wlane = h * 64 + m * 2;
  f2 wreg[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) wreg[f] = *reinterpret_cast<const f2*>(wu + f * wstride_f + wlane);

  f16v acc[8];
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  f2 u[8], tn[2][4], qa, qb, qc;
  auto rd = [&](const int bo, const int sub, const int col) {
    const int off = col * P7 + 4 * sub + bo;
    qa = *reinterpret_cast<const f2*>(&lds[Ra + off]);
    qb = *reinterpret_cast<const f2*>(&lds[Rb + off]);
    qc = *reinterpret_cast<const f2*>(&lds[Rc + off]);
  };
  auto colmath = [&](int g) {
    tn[0][g] = qa - qb;
    tn[1][g] = qb + sg * qc;
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
    colmath(g);
  }
  xform();
  rd(0, 1, 0);
  if (EXP & 64) tl1 = wall_clock64();

  const int nch = Cin / KC;
  // ablations (v7): 128 no weight reloads, 256 no LDS operand reads, 512 no transform VALU, 1024 no staging (global loads,
  // LDS stores, barrier), 2048 no MFMAs
  auto chunk = [&](const int ch, const int cur, const int nxt) {
    if (!(EXP & 1024)) gload((ch + 1 < nch ? ch + 1 : ch) * KC);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      int gs = ch * 4 + s + 1;
      gs = gs < nsub ? gs : nsub - 1;
      const float* wn = wu + (size_t)gs * wsub;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f2 a = qa, bq = qb, c = qc;
        if (s == 2 && g == 3 && !(EXP & 1024)) {
          sstore(nxt);
          __syncthreads();
        }
        if (!(EXP & 256)) {
          if (g < 3) rd(s == 3 ? nxt : cur, (s + 1) & 3, g + 1);
          else rd(s >= 2 ? nxt : cur, (s + 2) & 3, 0);
        }
        if (!(EXP & 1)) __builtin_amdgcn_sched_barrier(0);
#define MF7(F, C) if (!(EXP & 2048)) acc[F] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[F].C, wreg[F].C, acc[F], 0, 0, 0)
#define FN7 if (!(EXP & 1)) __builtin_amdgcn_sched_barrier(0)
        if (EXP & 512) {
          MF7(g, x); FN7; MF7(4 + g, x); FN7; MF7(g, y); FN7; MF7(4 + g, y); FN7;
        } else if (EXP & 2) {          // colmath as a burst after the group's MFMAs
          MF7(g, x); FN7; MF7(4 + g, x); FN7; MF7(g, y); FN7; MF7(4 + g, y); FN7;
          tn[0][g] = a - bq; tn[1][g] = bq + sg * c;
        } else if (EXP & 4) {   // colmath after MFMAs 2 and 3
          MF7(g, x); FN7; MF7(4 + g, x); FN7;
          MF7(g, y); tn[0][g] = a - bq; FN7;
          MF7(4 + g, y); tn[1][g] = bq + sg * c; FN7;
        } else {
          MF7(g, x); tn[0][g] = a - bq; FN7;
          MF7(4 + g, x); tn[1][g] = bq + sg * c; FN7;
          MF7(g, y); FN7;
          MF7(4 + g, y); FN7;
        }
        if (!(EXP & 128)) {
          wreg[g] = *reinterpret_cast<const f2*>(wn + g * wstride_f + wlane);
          wreg[4 + g] = *reinterpret_cast<const f2*>(wn + (4 + g) * wstride_f + wlane);
        }
        FN7;
      }
      if (!(EXP & 512)) xform();
    }
  };
  for (int ch = 0; ch < nch; ch += 2) {
    chunk(ch, 0, BUF7);
    chunk(ch + 1, BUF7, 0);
  }
  if (EXP & 64) tl2 = wall_clock64();

  // EXP & 32: while the epilogue runs, pull the first chunk of the tile block this CU will most likely get next
  // (blockIdx + 256: same XCD) into L2 with LDS-DMA loads into a dead LDS region
  if (EXP & 32) {
    const int tbn = (int)(blockIdx.x + 256) / nchunks;
    const int total_tb = (int)gridDim.x / nchunks;
    if (tbn < total_tb) {
      int t2 = tbn;
      const int bx2 = t2 % nbx; t2 /= nbx;
      const int by2 = t2 % nby; t2 /= nby;
      const int bz2 = t2 % nbz;
      const int b2 = t2 / nbz;
      const int gy = min(max(by2 * 8 - 1 + hy, 0), S - 1), gx = min(max(bx2 * 8 - 1 + hx, 0), S - 1);
#pragma unroll
      for (int hz = 0; hz < 6; ++hz) {
        const int gz = min(max(bz2 * 4 - 1 + hz, 0), S - 1);
        const float* p = x + ((((size_t)b2 * S + gz) * S + gy) * S + gx) * Cin + q * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(lds + 20480 + w * 256), 16, 0, 0);
      }
    }
  }
  // ---- epilogue, two passes of 8 accumulator registers
  float* part = lds;                              // [8 w][32 = rr*4 + dy*2 + dx][64 lanes]  (64 KB)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = pass * 8 + rr;
      float qv[2][2];                             // [fy local][dx]
#pragma unroll
      for (int fy = 0; fy < 2; ++fy) {
        qv[fy][0] = acc[fy * 4 + 0][r] + acc[fy * 4 + 1][r] + acc[fy * 4 + 2][r];
        qv[fy][1] = acc[fy * 4 + 1][r] - acc[fy * 4 + 2][r] - acc[fy * 4 + 3][r];
      }
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        // fyh = 0 holds fy 0,1: dy0 += q0 + q1, dy1 += q1;   fyh = 1 holds fy 2,3: dy0 += q0, dy1 += -q0 - q1
        const float d0 = fyh ? qv[0][dx] : qv[0][dx] + qv[1][dx];
        const float d1 = fyh ? -qv[0][dx] - qv[1][dx] : qv[1][dx];
        part[(w * 32 + rr * 4 + 0 + dx) * 64 + lane] = d0;
        part[(w * 32 + rr * 4 + 2 + dx) * 64 + lane] = d1;
      }
    }
    __syncthreads();
    // task: k = tid & 7 (channel quad), t = tid >> 3 (0..63): hh = t & 1, o = (t >> 1) & 3, rr = t >> 3
    {
      const int k = tid & 7, t = tid >> 3;
      const int hh = t & 1, o = (t >> 1) & 3, rr = t >> 3;
      const int r = pass * 8 + rr;
      const int mt = (r >> 2) * 8 + hh * 4 + (r & 3);
      const int ttz = mt >> 4, tty = (mt >> 2) & 3, ttx = mt & 3;
      const int slot = (rr * 4 + o) * 64 + hh * 32 + 4 * k;
      f4 p[4];
#pragma unroll
      for (int z = 0; z < 4; ++z)
        p[z] = *reinterpret_cast<const f4*>(&part[(2 * z) * 2048 + slot]) +
               *reinterpret_cast<const f4*>(&part[(2 * z + 1) * 2048 + slot]);
      const int vy = oy + 2 * tty + (o >> 1), vx = ox + 2 * ttx + (o & 1), vz = oz + 2 * ttz;
      const size_t o0 = ((((size_t)b * S + vz) * S + vy) * S + vx) * Cout + n0 + 4 * k;
      if (!(EXP & 8) || (p[0].x == 12345.f)) {
        *reinterpret_cast<f4*>(y + o0) = p[0] + p[1] + p[2];
        *reinterpret_cast<f4*>(y + o0 + (size_t)S * S * Cout) = p[1] - p[2] - p[3];
      }
    }
  }
  if (EXP & 64) {                                   // one record per WAVE: start, prologue end, main-loop end, end, ids
    if (lane == 0) {
      unsigned long long* r = g_tl + ((size_t)blockIdx.x * 8 + w) * 6;
      r[0] = tl0; r[1] = tl1; r[2] = tl2; r[3] = wall_clock64();
      r[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
      r[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
  }
}

// ---------------------------------------------------------------- host
static void transform_weights(const std::vector<float>& w, int Cin, int Cout, std::vector<float>& wt, int cw, int lay) {
  // w: [3][3][3][Cin][Cout] -> wt: [64][Cin/8][2][Cout][4]
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  wt.assign((size_t)64 * Cin * Cout, 0.f);
  for (int ci = 0; ci < Cin; ++ci)
    for (int co = 0; co < Cout; ++co) {
      double g[3][3][3];
      for (int a = 0; a < 3; ++a)
        for (int bq = 0; bq < 3; ++bq)
          for (int c = 0; c < 3; ++c) g[a][bq][c] = w[((((size_t)a * 3 + bq) * 3 + c) * Cin + ci) * Cout + co];
      for (int fz = 0; fz < 4; ++fz)
        for (int fy = 0; fy < 4; ++fy)
          for (int fx = 0; fx < 4; ++fx) {
            double s = 0;
            for (int a = 0; a < 3; ++a)
              for (int bq = 0; bq < 3; ++bq)
                for (int c = 0; c < 3; ++c) s += G[fz][a] * G[fy][bq] * G[fx][c] * g[a][bq][c];
            int f = (fz * 4 + fy) * 4 + fx;
            const int g2 = 2 * cw;   // channels per sub-step: 8 (4 per lane) or 4 (2 per lane)
            size_t idx = ((((size_t)f * (Cin / g2) + ci / g2) * 2 + ((ci % g2) / cw)) * Cout + co) * cw + (ci % cw);
            if (lay == 4)
              idx = (((((size_t)(co / 32) * (Cin / 4) + ci / 4) * 64 + f) * 2 + ((ci % 4) / 2)) * 32 + co % 32) * 2 + ci % 2;
            wt[idx] = (float)s;
          }
    }
}

int main(int argc, char** argv) {
  bool check = argc > 1 && !strcmp(argv[1], "check");
  int B = 32, S = 32, Cin = 128, Cout = 128;
  if (check) { B = 2; S = 8; Cin = 32; Cout = 32; }
  if (argc > 5) { B = atoi(argv[2]); S = atoi(argv[3]); Cin = atoi(argv[4]); Cout = atoi(argv[5]); }
  size_t nx = (size_t)B * S * S * S * Cin, ny = (size_t)B * S * S * S * Cout, nw = (size_t)27 * Cin * Cout;
  std::vector<float> hx(nx), hw(nw), hwt;
  unsigned st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.f - 0.5f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = rnd() * 0.2f;
  int variant = getenv("WINO_V") ? atoi(getenv("WINO_V")) : 3;
  int flags = getenv("WINO_FLAGS") ? atoi(getenv("WINO_FLAGS")) : 0;
  float *dx, *dwt, *dy, *dz;
  transform_weights(hw, Cin, Cout, hwt, variant >= 3 ? 2 : 4, variant >= 4 ? 4 : 0);
  HIPCHECK(hipMalloc(&dz, (Cin + 64) * 4)); HIPCHECK(hipMemset(dz, 0, (Cin + 64) * 4));
  HIPCHECK(hipMalloc(&dx, nx * 4)); HIPCHECK(hipMalloc(&dwt, hwt.size() * 4)); HIPCHECK(hipMalloc(&dy, ny * 4));
  HIPCHECK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(dwt, hwt.data(), hwt.size() * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemset(dy, 0, ny * 4));
  int grid = B * (S / 4) * (S / 8) * (S / 8) * (Cout / 32);
  auto launch = [&]() {
    if (variant == 0) hipLaunchKernelGGL(wino_fwd, dim3(grid), dim3(256), 0, 0, dx, dwt, dy, S, Cin, Cout);
    else if (variant == 3) {
      int ex = flags >> 1;
#define L3(E) hipLaunchKernelGGL(wino_fwd3<E>, dim3(grid), dim3(256), 0, 0, dx, dwt, dz, dy, S, Cin, Cout, flags)
      if (ex == 0) L3(0); else if (ex == 1) L3(2); else if (ex == 2) L3(4); else if (ex == 3) L3(6); else if (ex == 4) L3(8);
      else if (ex == 5) L3(10); else if (ex == 6) L3(12); else L3(14);
    }
    else if (variant == 7) {
#define L7(E) case E: hipLaunchKernelGGL((wino_fwd7<E>), dim3(grid), dim3(512), 0, 0, dx, dwt, dy, S, Cin, Cout); break
      switch (flags) { L7(0); L7(1); L7(2); L7(4); L7(8); L7(16); L7(24); L7(32); L7(64); L7(128); L7(256); L7(512); L7(1024);
        L7(2048); L7(384); L7(896); L7(1920); L7(1040); L7(1152); L7(2048 + 16); L7(640); L7(1536);
        default: fprintf(stderr, "v7: flags %d not instantiated\n", flags); exit(2); }
    }
    else if (variant == 6) hipLaunchKernelGGL((wino_fwd6<0>), dim3(grid), dim3(256), 0, 0, dx, dwt, dy, S, Cin, Cout);
    else if (variant == 4) hipLaunchKernelGGL((wino_fwd4<0, 0>), dim3(grid), dim3(256), 0, 0, dx, dwt, dz, dy, S, Cin, Cout);
    else if (variant == 5) {
#define L5(E) hipLaunchKernelGGL((wino_fwd4<1, E>), dim3(grid), dim3(256), 0, 0, dx, dwt, dz, dy, S, Cin, Cout)
      switch (flags) { case 0: L5(0); break; case 2: L5(2); break; case 4: L5(4); break; case 8: L5(8); break;
        case 16: L5(16); break; case 14: L5(14); break; case 30: L5(30); break; case 6: L5(6); break; case 64: L5(64); break; case 128: L5(128); break; case 192: L5(192); break; case 256: L5(256); break; default: L5(0); }
    }
    else hipLaunchKernelGGL(wino_fwd8, dim3(grid), dim3(512), 0, 0, dx, dwt, dz, dy, S, Cin, Cout);
  };
  unsigned long long* dtl = nullptr;
  if (variant == 7 && flags == 64) {
    HIPCHECK(hipMalloc(&dtl, (size_t)grid * 8 * 6 * 8));
    HIPCHECK(hipMemset(dtl, 0, (size_t)grid * 8 * 6 * 8));
    HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &dtl, sizeof(dtl)));
    launch(); launch();                       // warm (clocks, L2)
    HIPCHECK(hipDeviceSynchronize());
  }
  launch();
  HIPCHECK(hipDeviceSynchronize());
  if (dtl) {
    // timeline analysis: per CU, workgroups in start order; all times in 10 ns ticks of the 100 MHz wall clock
    std::vector<unsigned long long> tl((size_t)grid * 8 * 6);
    HIPCHECK(hipMemcpy(tl.data(), dtl, tl.size() * 8, hipMemcpyDeviceToHost));
    struct WG { unsigned long long s, p, m, e; int cu; int blk; };
    std::vector<WG> wgs(grid);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int g = 0; g < grid; ++g) {
      WG q{~0ull, 0, 0, 0, 0, g};
      for (int w = 0; w < 8; ++w) {
        const unsigned long long* r = &tl[((size_t)g * 8 + w) * 6];
        q.s = r[0] < q.s ? r[0] : q.s; q.p = r[1] > q.p ? r[1] : q.p; q.m = r[2] > q.m ? r[2] : q.m; q.e = r[3] > q.e ? r[3] : q.e;
        const unsigned hw = (unsigned)r[4], xcc = (unsigned)r[5] & 15;
        q.cu = (int)((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15));
      }
      wgs[g] = q; t0 = q.s < t0 ? q.s : t0; t1 = q.e > t1 ? q.e : t1;
    }
    std::sort(wgs.begin(), wgs.end(), [](const WG& a, const WG& b) { return a.cu != b.cu ? a.cu < b.cu : a.s < b.s; });
    double sp = 0, sm = 0, se = 0, sg = 0; long ng = 0; int ncu = 0, overlaps = 0;
    for (int i = 0; i < grid; ++i) {
      sp += wgs[i].p - wgs[i].s; sm += wgs[i].m - wgs[i].p; se += wgs[i].e - wgs[i].m;
      if (i == 0 || wgs[i].cu != wgs[i - 1].cu) { ++ncu; continue; }
      if (wgs[i].s < wgs[i - 1].e) ++overlaps;
      sg += (double)wgs[i].s - (double)wgs[i - 1].e; ++ng;
    }
    printf("timeline: %d workgroups on %d CUs, kernel span %.1f us; per workgroup (us): prologue %.3f  main loop %.3f  epilogue %.3f  "
           "gap to the next workgroup on the same CU %.3f (%d overlapping)\n", grid, ncu, (t1 - t0) * 0.01, sp / grid * 0.01,
           sm / grid * 0.01, se / grid * 0.01, sg / ng * 0.01, overlaps);
    for (int i = 0; i < 12 && i < grid; ++i)
      printf("  cu %04x blk %5d: start %.2f  +pro %.2f  +main %.2f  +epi %.2f us\n", wgs[i].cu, wgs[i].blk, (wgs[i].s - t0) * 0.01,
             (wgs[i].p - wgs[i].s) * 0.01, (wgs[i].m - wgs[i].p) * 0.01, (wgs[i].e - wgs[i].m) * 0.01);
    double dm = 0, de = 0;                    // how far apart do the 8 waves of a workgroup leave the main loop / finish?
    for (int g = 0; g < grid; ++g) {
      unsigned long long mlo = ~0ull, mhi = 0, elo = ~0ull, ehi = 0;
      for (int w = 0; w < 8; ++w) {
        const unsigned long long* r = &tl[((size_t)g * 8 + w) * 6];
        mlo = r[2] < mlo ? r[2] : mlo; mhi = r[2] > mhi ? r[2] : mhi; elo = r[3] < elo ? r[3] : elo; ehi = r[3] > ehi ? r[3] : ehi;
      }
      dm += mhi - mlo; de += ehi - elo;
    }
    printf("  wave spread: main-loop exit %.3f us, end %.3f us\n", dm / grid * 0.01, de / grid * 0.01);
  }
  if (check || (size_t)B * S * S * S <= 4096) {
    std::vector<float> hy(ny);
    HIPCHECK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int b = 0; b < B; ++b)
      for (int z = 0; z < S; ++z)
        for (int yy = 0; yy < S; ++yy)
          for (int xx = 0; xx < S; ++xx)
            for (int co = 0; co < Cout; ++co) {
              double s = 0;
              for (int a = 0; a < 3; ++a)
                for (int bq = 0; bq < 3; ++bq)
                  for (int c = 0; c < 3; ++c) {
                    int iz = z + a - 1, iy = yy + bq - 1, ix = xx + c - 1;
                    if (iz < 0 || iz >= S || iy < 0 || iy >= S || ix < 0 || ix >= S) continue;
                    const float* xp = &hx[((((size_t)b * S + iz) * S + iy) * S + ix) * Cin];
                    const float* wp = &hw[((((size_t)a * 3 + bq) * 3 + c) * Cin) * Cout + co];
                    for (int ci = 0; ci < Cin; ++ci) s += (double)xp[ci] * wp[(size_t)ci * Cout];
                  }
              double got = hy[((((size_t)b * S + z) * S + yy) * S + xx) * Cout + co];
              maxerr = fmax(maxerr, fabs(got - s));
              maxref = fmax(maxref, fabs(s));
            }
    printf("check B=%d S=%d Cin=%d Cout=%d: max|err| %.3e  max|ref| %.3e  rel %.3e\n", B, S, Cin, Cout, maxerr, maxref,
           maxerr / maxref);
    if (check) return maxerr / maxref < 1e-5 ? 0 : 1;
  }
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
  const int iters = 10;
  HIPCHECK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch();
  HIPCHECK(hipEventRecord(e1));
  HIPCHECK(hipEventSynchronize(e1));
  float ms;
  HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  double direct_flops = 2.0 * 27 * (double)B * S * S * S * Cin * Cout;
  printf("v%d time B=%d S=%d Cin=%d Cout=%d: %.3f ms  = %.1f direct-equivalent TFLOP/s (%.1f Winograd TFLOP/s)\n", variant, B, S, Cin,
         Cout, ms, direct_flops / ms / 1e9, direct_flops * 64.0 / 216.0 / ms / 1e9);
  return 0;
}
