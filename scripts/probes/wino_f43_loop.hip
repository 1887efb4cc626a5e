// Probe (round 5): would F(4,3) on the x axis -- F(4x2x2, 3x3x3): 96 products per 16 outputs instead of 64 per 8 -- pay in the
// MAIN LOOP of the fused Winograd forward kernel, given the register file?  Not a convolution: the instruction mix of one
// k-sub-step (4 input channels) of one wave, per design, issued exactly as the shipped kernel issues it (16x16x4 fp32 MFMAs on
// independent accumulators, the transform VALU ops and the LDS operand reads between them, the weight loads from L2), timed
// with 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o wino_f43_loop wino_f43_loop.hip && ./wino_f43_loop
//
//   design            tiles x channels x freq / workgroup      acc regs / wave   MFMA   VALU   LDS b32   weight loads   work / sub-step
//   A  shipped F(2,3)^3      16 x 64 x 64 (128 voxels)              128            32     21      16       8 x b128         4096
//   B  F(4,3)x, 4 col blocks 16 x 64 x 96 (256 voxels)              192            48     39      24      12 x b128         8192
//   C  F(4,3)x, 2 col blocks 16 x 32 x 96 (256 voxels)               96            24     39      24      12 x b64          4096
// work = voxels x output channels x input channels finished per wave and sub-step (1/8 of the workgroup's).  VALU counts: the
// shipped kernel issues 0.67 VALU per MFMA (PMC, profiles/r4_pmc_summary.txt) = 21 per sub-step, 8 of them the x rows of B^T
// (4 per (fz, fy) line, 2 lines); F(4,3)'s x rows are 13 per line with common subexpressions (a = d4 - 4 d2, b = d3 - 4 d1, ...)
// instead of 4, the rest (y rows, addressing) is unchanged: 13 + 26 = 39.  B needs 192 accumulator + ~100 other registers
// per lane at 2 waves per SIMD (256 available): it cannot be built -- it is timed here to show what it would have given.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NACC, int NVALU, int NLDS, int NWL, bool WIDE>
__global__ __launch_bounds__(512) void loop_kernel(float* out, const float* in, const float* wt, int iters) {
  __shared__ float lds[16384];
  const int tid = threadIdx.x;
  for (int i = tid; i < 16384; i += 512) lds[i] = in[i & 1023];
  __syncthreads();
  f4 acc[NACC];
#pragma unroll
  for (int f = 0; f < NACC; ++f) acc[f] = f4{0.f, 0.f, 0.f, 0.f};
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = in[tid + 64 * j];
  const float* wp = wt + (size_t)(blockIdx.x & 7) * 65536 + (tid & 63) * 4;
  f4 wreg[NWL];
#pragma unroll
  for (int j = 0; j < NWL; ++j) wreg[j] = f4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    // weight operands of the NEXT sub-step: requested up front, consumed one iteration later (as the shipped kernel does)
    f4 wnext[NWL];
#pragma unroll
    for (int j = 0; j < NWL; ++j) {
      const float* p = wp + ((it * NWL + j) & 255) * 256;
      if (WIDE) wnext[j] = *reinterpret_cast<const f4*>(p);
      else { const f2 t = *reinterpret_cast<const f2*>(p); wnext[j] = f4{t.x, t.y, t.x, t.y}; }
    }
    constexpr int PER = NACC / NWL;                    // MFMAs per weight register set (column blocks)
#pragma unroll
    for (int f = 0; f < NACC; ++f) {
      // the LDS reads and the transform ops spread evenly between the MFMAs
#pragma unroll
      for (int l = f * NLDS / NACC; l < (f + 1) * NLDS / NACC; ++l) v[l & 7] += lds[(tid * 3 + l * 517 + it * 33) & 16383];
#pragma unroll
      for (int q = f * NVALU / NACC; q < (f + 1) * NVALU / NACC; ++q)
        v[q & 7] = __builtin_fmaf(v[(q + 1) & 7], v[(q + 3) & 7], v[(q + 5) & 7]);
      acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[f & 7], wreg[f / PER][f % PER], acc[f], 0, 0, 0);
      if ((f & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NWL; ++j) wreg[j] = wnext[j];
  }
  float s = 0.f;
#pragma unroll
  for (int f = 0; f < NACC; ++f) s += acc[f][0] + acc[f][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j];
  out[(size_t)blockIdx.x * 512 + tid] = s;
}

template <int NACC, int NVALU, int NLDS, int NWL, bool WIDE>
static double run(const char* name, double work, float* dout, float* din, float* dw) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((loop_kernel<NACC, NVALU, NLDS, NWL, WIDE>), dim3(256), dim3(512), 0, 0, dout, din, dw, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((loop_kernel<NACC, NVALU, NLDS, NWL, WIDE>), dim3(256), dim3(512), 0, 0, dout, din, dw, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&loop_kernel<NACC, NVALU, NLDS, NWL, WIDE>));
  // one SIMD runs 2 waves: cycles per sub-step of ONE wave = elapsed / (2 * iters) (the two share the MFMA pipe)
  const double cyc = ms * 1e-3 * 2.4e9 / (2.0 * iters);
  const double ideal = NACC * 32.0;                                 // 16x16x4 fp32: 32 cycles per MFMA
  printf("%-26s %3d MFMA %2d VALU %2d LDS %2d loads: %7.1f cycles per sub-step (MFMA alone %6.0f: %.2f busy), %6.3f cycles per "
         "unit of work, %d VGPRs, scratch %d B\n", name, NACC, NVALU, NLDS, NWL, cyc, ideal, ideal / cyc, cyc / work * 1024.0,
         fa.numRegs, (int)fa.localSizeBytes);
  return cyc / work;
}

int main() {
  float *dout, *din, *dw;
  hipMalloc(&dout, (size_t)256 * 512 * 4); hipMalloc(&din, 4096 * 4); hipMalloc(&dw, (size_t)8 * 65536 * 4 + 65536 * 4);
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (i % 17) * 0.01f - 0.05f;
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemset(dw, 0, (size_t)8 * 65536 * 4 + 65536 * 4);
  const double a = run<32, 21, 16, 8, true>("A shipped F(2,3)^3", 4096, dout, din, dw);
  const double a0 = run<32, 0, 0, 8, true>("A MFMA + weight loads only", 4096, dout, din, dw);
  const double b = run<48, 39, 24, 12, true>("B F(4,3)x 4 col blocks", 8192, dout, din, dw);
  const double c = run<24, 39, 24, 12, false>("C F(4,3)x 2 col blocks", 4096, dout, din, dw);
  const double c2 = run<24, 30, 24, 12, false>("C with 30 VALU (optimistic)", 4096, dout, din, dw);
  printf("main-loop time per unit of work relative to A: A' (no transform) %.3f, B %.3f, C %.3f, C optimistic %.3f\n", a0 / a, b / a,
         c / a, c2 / a);
  printf("whole kernel, if the 18 %% outside the main loop (prologue, output transform, statistics) stayed the same per voxel: "
         "B %.3f, C %.3f of A\n", 0.82 * b / a + 0.18, 0.82 * c / a + 0.18);
  return 0;
}
