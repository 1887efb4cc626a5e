// Winograd F(2x2x2, 3x3x3) for the S = 4 layers (gfx950 / MI355X only), round 4: the U-Net's bottom convolutions c9 / c10
// (Keras Conv3D 3x3x3 "same", /root/reference/unet/unet.py:303-308), 256 / 512 -> 512 channels on 4^3 voxels.
//
// The fused Winograd kernels (conv_wino*.hip) need a 4 x 4 x 8 voxel block inside one sample; at S = 4 a sample IS 4^3
// voxels, so these layers ran on the 27-tap implicit GEMM (c10 forward 0.25 ms, 29 GFLOP at 113 TFLOP/s, plus a split-K
// finish).  Here the activations are tiny (B x 64 voxels x 512 channels = 4 MB) and the weights large, so the UNFUSED form
// pays: the transformed tensors cost 8 x 4 MB of HBM traffic per layer, nothing next to the 188 GFLOP of multiplies saved:
//   V[f][tile][ci]  = (B^T d B) of the 4^3 input patch of each 2^3 output tile           winog_in_kernel   (HBM-bound)
//   M[f][tile][co]  = V[f] x U[f],  U[f][ci][co] = (G g G^T) of the weights, packed once   launch_gemm_zbatch: 64 plain
//                     per parameter update (pack_wino_pair layout 2)                        GEMMs on conv_fwd_kernel's
//                                                                                           64 x 64 fp32-MFMA tile
//   y[voxel][co]    = A^T M A + bias, activation, per-block BatchNorm partials             winog_out_kernel  (HBM-bound)
// 64/216 of the multiplies, exact in exact arithmetic (the same identity as conv_wino.hip).  Backward-data is the same
// chain on dy with the tap-flipped, transposed weights; backward-weight is dU[f][ci][co] = sum_tiles V[f][t][ci] Z[f][t][co]
// with Z = (A dy A^T) -- again 64 plain GEMMs, reduction length = tiles -- followed by dg = G^T dU G.
// Measured on MI355X at B = 32: see DESIGN.md section 4 (round 4).
#include "common.h"

namespace ics {

typedef float vf4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float wg_act(float v, float slope) { return fmaxf(v, v * slope); }
__host__ __device__ __forceinline__ float wg_slope(int act) { return act == ACT_RELU ? 0.f : (act == ACT_LRELU ? kLeaky : 1.f); }

// rows of B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1] applied to (d0, d1, d2, d3) in place
#define ICS_WG_BT(d0, d1, d2, d3)            \
  do {                                       \
    const float t0_ = d0 - d2, t1_ = d1 + d2, t2_ = d2 - d1, t3_ = d1 - d3; \
    d0 = t0_; d1 = t1_; d2 = t2_; d3 = t3_;  \
  } while (0)

// one thread = one (tile, channel): the 4^3 patch around the tile's 2^3 outputs (zero outside the sample, AFTER the
// producer's BatchNorm affine / activation, as Keras pads), B^T along z, y, x, 64 frequency values.
// V[f][tile][c] (coalesced over c).
template <bool AFF, bool NOACT>
__global__ __launch_bounds__(256) void winog_in_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float slope, int S, int T, int C,
                                                       float* __restrict__ V) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)T * C) return;
  const int c = (int)(idx % (size_t)C), t = (int)(idx / (size_t)C);
  const int Th = S >> 1;
  const int tx = t % Th, ty = (t / Th) % Th, tz = (t / (Th * Th)) % Th, b = t / (Th * Th * Th);
  float sc = 1.f, sh = 0.f;
  if (AFF) { sc = scale[c]; sh = shift[c]; }
  float d[4][4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int z = 2 * tz - 1 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int y = 2 * ty - 1 + j;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int xx = 2 * tx - 1 + k;
        const bool ok = (unsigned)z < (unsigned)S && (unsigned)y < (unsigned)S && (unsigned)xx < (unsigned)S;
        float v = 0.f;
        if (ok) {
          v = x[((((size_t)b * S + z) * S + y) * S + xx) * ldx + c];
          if (AFF) { v = fmaf(v, sc, sh); if (!NOACT) v = wg_act(v, slope); }
        }
        d[i][j][k] = v;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) ICS_WG_BT(d[0][j][k], d[1][j][k], d[2][j][k], d[3][j][k]);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) ICS_WG_BT(d[i][0][k], d[i][1][k], d[i][2][k], d[i][3][k]);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) ICS_WG_BT(d[i][j][0], d[i][j][1], d[i][j][2], d[i][j][3]);
  const size_t fs = (size_t)T * C;
  float* vp = V + (size_t)t * C + c;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) vp[(size_t)((i * 4 + j) * 4 + k) * fs] = d[i][j][k];
}

// V[f][T][C] -> Vt[f][C][T] (the A operand of the backward-weight GEMMs: rows = input channels, reduction = tiles) through
// 32 x 32 LDS tiles, both sides coalesced.  (Written straight from winog_in_kernel -- one 4-byte store per lane at a stride of
// T floats -- the same 33 MB took 85 us on c10; as its own pass 17.)
__global__ __launch_bounds__(256) void winog_transpose_kernel(const float* __restrict__ V, int T, int C, float* __restrict__ Vt) {
  __shared__ float tile[32][33];
  const int tbx = C >> 5;                                  // 32-wide tiles along C
  const int f = blockIdx.y, bt = blockIdx.x / tbx, bc = blockIdx.x % tbx;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;  // 8 rows per pass
  const float* src = V + (size_t)f * T * C;
  float* dst = Vt + (size_t)f * T * C;
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[ly + 8 * r][lx] = src[(size_t)(bt * 32 + ly + 8 * r) * C + bc * 32 + lx];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) dst[(size_t)(bc * 32 + ly + 8 * r) * T + bt * 32 + lx] = tile[lx][ly + 8 * r];
}

// rows of A^T = [1 1 1 0; 0 1 -1 -1]: (m0..m3) -> (o0, o1)
#define ICS_WG_AT(m0, m1, m2, m3, o0, o1) \
  do { o0 = m0 + m1 + m2; o1 = m1 - m2 - m3; } while (0)

// one block = 8 consecutive tiles (64 output voxels; at S = 4 one sample) x 32 channels; thread = (tile, channel).
// y = A^T M A (+ bias, activation), float stores coalesced over the channel; per-block BatchNorm (count, mean, M2) partials in
// conv_igemm.hip's layout [3][Npad][nblk] (block index fastest, 64 rows per block) as equal-count Chan merges.
__global__ __launch_bounds__(256) void winog_out_kernel(const float* __restrict__ Mt, const float* __restrict__ bias,
                                                        float pre_slope, float* __restrict__ y, int ldo, int S, int T, int N,
                                                        float* __restrict__ stat_partial, int Npad) {
  __shared__ float s_mean[8][33], s_m2[8][33];
  const int nchunks = N >> 5;
  const int tb = blockIdx.x / nchunks, nc = blockIdx.x % nchunks;
  const int tl = threadIdx.x >> 5, cl = threadIdx.x & 31;
  const int t = tb * 8 + tl, n = nc * 32 + cl;
  const int Th = S >> 1;
  const int tx = t % Th, ty = (t / Th) % Th, tz = (t / (Th * Th)) % Th, b = t / (Th * Th * Th);
  const size_t fs = (size_t)T * N;
  const float* mp = Mt + (size_t)t * N + n;
  float a[2][4][4];                                   // after the z rows of A^T: [dz][fy][fx]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float m0 = mp[(size_t)((0 * 4 + j) * 4 + k) * fs], m1 = mp[(size_t)((1 * 4 + j) * 4 + k) * fs],
                  m2 = mp[(size_t)((2 * 4 + j) * 4 + k) * fs], m3 = mp[(size_t)((3 * 4 + j) * 4 + k) * fs];
      ICS_WG_AT(m0, m1, m2, m3, a[0][j][k], a[1][j][k]);
    }
  float o[2][2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float r[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ICS_WG_AT(a[i][0][k], a[i][1][k], a[i][2][k], a[i][3][k], r[0][k], r[1][k]);
#pragma unroll
    for (int j = 0; j < 2; ++j) ICS_WG_AT(r[j][0], r[j][1], r[j][2], r[j][3], o[i][j][0], o[i][j][1]);
  }
  const float bv = bias != nullptr ? bias[n] : 0.f;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float v = wg_act(o[i][j][k] + bv, pre_slope);
        o[i][j][k] = v;
        sum += v;
        y[((((size_t)b * S + 2 * tz + i) * S + 2 * ty + j) * S + 2 * tx + k) * ldo + n] = v;
      }
  if (stat_partial == nullptr) return;
  const float mean8 = sum * 0.125f;
  float m2 = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k) m2 += (o[i][j][k] - mean8) * (o[i][j][k] - mean8);
  s_mean[tl][cl] = mean8; s_m2[tl][cl] = m2;
  __syncthreads();
  if (tl == 0) {
    float ms = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { ms += s_mean[w][cl]; q += s_m2[w][cl]; }
    const float mean_t = ms * 0.125f;
    float dev = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) dev += (s_mean[w][cl] - mean_t) * (s_mean[w][cl] - mean_t);
    const size_t nblk = (size_t)(T >> 3);
    float* sp = stat_partial + (size_t)n * nblk + tb;
    sp[0] = 64.f;
    sp[(size_t)Npad * nblk] = mean_t;
    sp[(size_t)2 * Npad * nblk] = q + 8.f * dev;
  }
}

// backward-weight, the dy operand: Z = A dy A^T per tile (A = [1 0; 1 1; 1 -1; 0 -1]), written as the packed B operand of the
// frequency GEMMs, Zp[f][T/4][N][4] (k = tile): one thread = one (tile quad, channel), one float4 per frequency.
__global__ __launch_bounds__(256) void winog_dy_kernel(const float* __restrict__ dy, int ldy, int S, int T, int N,
                                                       float* __restrict__ Zp) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)(T >> 2) * N) return;
  const int n = (int)(idx % (size_t)N), q = (int)(idx / (size_t)N);
  const int Th = S >> 1;
  float d[4][2][2][2];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int t = q * 4 + e;
    const int tx = t % Th, ty = (t / Th) % Th, tz = (t / (Th * Th)) % Th, b = t / (Th * Th * Th);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k)
          d[e][i][j][k] = dy[((((size_t)b * S + 2 * tz + i) * S + 2 * ty + j) * S + 2 * tx + k) * ldy + n];
  }
  const size_t fs = (size_t)T * N;                   // floats per frequency image: (T/4) * N * 4
  float* zp = Zp + ((size_t)q * N + n) * 4;
  // rows of A: f = 0: d0, 1: d0 + d1, 2: d0 - d1, 3: -d1
#define ICS_WG_A(f, d0, d1) ((f) == 0 ? (d0) : (f) == 1 ? (d0) + (d1) : (f) == 2 ? (d0) - (d1) : -(d1))
#pragma unroll
  for (int fz = 0; fz < 4; ++fz) {
    float zy[4][4][2];                               // [tile][fy][x] after the z and y rows
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float c0 = ICS_WG_A(fz, d[e][0][0][k], d[e][1][0][k]), c1 = ICS_WG_A(fz, d[e][0][1][k], d[e][1][1][k]);
#pragma unroll
        for (int fy = 0; fy < 4; ++fy) zy[e][fy][k] = ICS_WG_A(fy, c0, c1);
      }
#pragma unroll
    for (int fy = 0; fy < 4; ++fy)
#pragma unroll
      for (int fx = 0; fx < 4; ++fx) {
        vf4 v;
        v.x = ICS_WG_A(fx, zy[0][fy][0], zy[0][fy][1]); v.y = ICS_WG_A(fx, zy[1][fy][0], zy[1][fy][1]);
        v.z = ICS_WG_A(fx, zy[2][fy][0], zy[2][fy][1]); v.w = ICS_WG_A(fx, zy[3][fy][0], zy[3][fy][1]);
        *reinterpret_cast<vf4*>(zp + (size_t)((fz * 4 + fy) * 4 + fx) * fs) = v;
      }
  }
#undef ICS_WG_A
}

// backward-weight, the last step: dg = G^T dU G per (ci, co) pair (G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]); one thread per
// pair, coalesced over co; dw[(tap * row_pitch + row_off + ci) * ldw + co]
#define ICS_WG_GT(u0, u1, u2, u3, w0, w1, w2) \
  do { const float h_ = 0.5f * (u1 + u2); w0 = u0 + h_; w1 = 0.5f * (u1 - u2); w2 = h_ + u3; } while (0)
__global__ __launch_bounds__(256) void winog_dw_kernel(const float* __restrict__ dU, int K, int N, float* __restrict__ dw,
                                                       int ldw, int row_pitch, int row_off) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)K * N) return;
  const int co = (int)(idx % (size_t)N), ci = (int)(idx / (size_t)N);
  const size_t fs = (size_t)K * N;
  const float* up = dU + idx;
  float a[3][4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float u0 = up[(size_t)((0 * 4 + j) * 4 + k) * fs], u1 = up[(size_t)((1 * 4 + j) * 4 + k) * fs],
                  u2 = up[(size_t)((2 * 4 + j) * 4 + k) * fs], u3 = up[(size_t)((3 * 4 + j) * 4 + k) * fs];
      ICS_WG_GT(u0, u1, u2, u3, a[0][j][k], a[1][j][k], a[2][j][k]);
    }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float r[3][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ICS_WG_GT(a[i][0][k], a[i][1][k], a[i][2][k], a[i][3][k], r[0][k], r[1][k], r[2][k]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float w0, w1, w2;
      ICS_WG_GT(r[j][0], r[j][1], r[j][2], r[j][3], w0, w1, w2);
      const int tap = (i * 3 + j) * 3;
      dw[((size_t)(tap + 0) * row_pitch + row_off + ci) * ldw + co] = w0;
      dw[((size_t)(tap + 1) * row_pitch + row_off + ci) * ldw + co] = w1;
      dw[((size_t)(tap + 2) * row_pitch + row_off + ci) * ldw + co] = w2;
    }
  }
}

}  // namespace

// ---------------------------------------------------------------- host side
bool conv_winog_ok(const ConvGeom& g, const ConvSrc* src, int nsrc) {
  if (g.flags & (CF_NO_WINO | CF_NO_WINOG)) return false;
  if (g.taps != 27 || nsrc != 1 || g.S != 4) return false;
  const ConvSrc& s = src[0];
  if (s.up || s.bcast || s.C != g.Cin) return false;
  // 64 x 64 GEMM tiles want a real K and N
  return g.Cin % 32 == 0 && g.Cin >= 64 && g.Cout % 64 == 0;
}
// backward-weight additionally needs the tile count (the GEMMs' reduction length) to be a multiple of 32
bool conv_winog_wgrad_ok(const ConvGeom& g, const ConvSrc* src, int nsrc) {
  return conv_winog_ok(g, src, nsrc) && (g.B * 8) % 32 == 0;
}
size_t conv_winog_weight_floats(int Cin, int Cout) { return (size_t)64 * Cin * Cout; }
// scratch floats: v = transformed input [64][T][max(Cin, Cout)] (also dy for backward-data), m = GEMM results
// [64][T][max(Cin, Cout)] or dU [64][Cin][Cout], z = packed dy operand [64][T][Cout]
void conv_winog_scratch_floats(const ConvGeom& g, size_t* v, size_t* m, size_t* z) {
  const size_t T = (size_t)g.B * 8, cm = (size_t)std::max(g.Cin, g.Cout);
  *v = 64 * T * cm;
  *m = std::max(64 * T * cm, (size_t)64 * g.Cin * g.Cout);
  *z = 64 * T * (size_t)g.Cout;
}

// forward (or backward-data with s0 = dy and the tap-flipped transposed weights): out[voxel][g.Cout]
int launch_conv_fwd_winog(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* wg, const float* bias, float* out,
                          int ldo, int pre_act, float* stat_partial, int* rows_per_block, float* v_scratch, float* m_scratch,
                          float* vt_keep) {
  ICS_CHECK(conv_winog_ok(g, &s0, 1), "shape not served by the Winograd-domain GEMM path");
  const int T = g.B * 8;
  const size_t tot = (size_t)T * g.Cin;
  const unsigned gi = (unsigned)((tot + 255) / 256);
  const bool aff = s0.scale != nullptr, noact = s0.act == ACT_NONE;
  if (!aff) ICS_LAUNCH((winog_in_kernel<false, true>), dim3(gi), dim3(256), 0, st, s0.p, s0.C, s0.scale, s0.shift, 1.f, g.S, T,
                       g.Cin, v_scratch);
  else if (noact) ICS_LAUNCH((winog_in_kernel<true, true>), dim3(gi), dim3(256), 0, st, s0.p, s0.C, s0.scale, s0.shift, 1.f, g.S,
                             T, g.Cin, v_scratch);
  else ICS_LAUNCH((winog_in_kernel<true, false>), dim3(gi), dim3(256), 0, st, s0.p, s0.C, s0.scale, s0.shift, wg_slope(s0.act),
                  g.S, T, g.Cin, v_scratch);
  if (vt_keep != nullptr) {
    ICS_CHECK(T % 32 == 0, "the transposed transform needs B % 4 == 0");
    ICS_LAUNCH(winog_transpose_kernel, dim3((unsigned)((T / 32) * (g.Cin / 32)), 64), dim3(256), 0, st, v_scratch, T, g.Cin, vt_keep);
  }
  ICS_HIP(hipGetLastError());
  ICS_TRY(launch_gemm_zbatch(st, 64, T, g.Cin, g.Cout, v_scratch, wg, m_scratch, g.flags));
  conv_set_last_kernel_id("conv_winog (transforms + 64 x conv_fwd_kernel<2, 2, 1, 1> GEMMs)");
  if (rows_per_block) *rows_per_block = 64;
  ICS_LAUNCH(winog_out_kernel, dim3((unsigned)((T / 8) * (g.Cout / 32))), dim3(256), 0, st, m_scratch, bias, wg_slope(pre_act), out,
             ldo, g.S, T, g.Cout, stat_partial, g.Npad);
  ICS_HIP(hipGetLastError());
  return 0;
}

// backward-weight: vt = the forward's transposed transform [64][Cin][T]; dw rows as launch_conv_wgrad (row_pitch = total
// input channels of the weight tensor, row_off = first of this layer's)
int launch_conv_wgrad_winog(hipStream_t st, const ConvGeom& g, const float* vt, const float* dy, int ldy, float* dw, int ldw,
                            int row_pitch, int row_off, float* z_scratch, float* m_scratch) {
  const int T = g.B * 8;
  ICS_CHECK(T % 32 == 0 && g.S == 4, "backward-weight GEMMs need B % 4 == 0");
  ICS_LAUNCH(winog_dy_kernel, dim3((unsigned)(((size_t)(T / 4) * g.Cout + 255) / 256)), dim3(256), 0, st, dy, ldy, g.S, T, g.Cout,
             z_scratch);
  ICS_HIP(hipGetLastError());
  ICS_TRY(launch_gemm_zbatch(st, 64, g.Cin, T, g.Cout, vt, z_scratch, m_scratch, g.flags));
  conv_set_last_kernel_id("conv_winog (transforms + 64 x conv_fwd_kernel<2, 2, 1, 1> GEMMs)");
  ICS_LAUNCH(winog_dw_kernel, dim3((unsigned)(((size_t)g.Cin * g.Cout + 255) / 256)), dim3(256), 0, st, m_scratch, g.Cin, g.Cout,
             dw, ldw, row_pitch ? row_pitch : g.Cin, row_off);
  ICS_HIP(hipGetLastError());
  return 0;
}

}  // namespace ics
