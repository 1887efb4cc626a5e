// Shared declarations for the icsg3d_amd HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace ics {

// Every kernel launch of the library goes through ICS_LAUNCH: a process-wide count of device kernels enqueued, the
// number rocprofv3 --kernel-trace reports for the same run minus the runtime's own copy / fill kernels and RCCL's
// (ics_kernel_launches; bench.py's kernel_launches_per_step).
extern std::atomic<long long> g_kernel_launches;
// Per-launch timing (engine.hip Profiler): while a LaunchTimer is installed on this thread, every launch carries its own
// start / stop event pair in the dispatch itself (hipExtLaunchKernelGGL) instead of two hipEventRecord markers on the
// stream around it -- the markers cost the timed region ~10 us per bracket (0.16 ms per U-Net step in round 4's bench).
struct LaunchTimer {
  virtual void next(hipEvent_t* start, hipEvent_t* stop) = 0;
  virtual ~LaunchTimer() = default;
};
extern thread_local LaunchTimer* tl_launch_timer;
template <typename... P, typename... A>
inline void ics_launch(void (*kernel)(P...), dim3 grid, dim3 block, size_t shmem, hipStream_t st, A&&... args) {
  static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
  if (tl_launch_timer != nullptr) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    tl_launch_timer->next(&e0, &e1);
    hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)shmem, st, e0, e1, 0u, static_cast<P>(args)...);
  } else {
    hipLaunchKernelGGL(kernel, grid, block, shmem, st, static_cast<P>(args)...);
  }
}
#define ICS_LAUNCH(...)                                      \
  do {                                                       \
    ::ics::g_kernel_launches.fetch_add(1, std::memory_order_relaxed); \
    ::ics::ics_launch(__VA_ARGS__);                          \
  } while (0)

// ---------------------------------------------------------------- error plumbing
void set_error(const std::string& msg);
#define ICS_HIP(expr)                                                                     \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ::ics::set_error(std::string(#expr) + " -> " + hipGetErrorString(_e) + " @" +       \
                       __FILE__ + ":" + std::to_string(__LINE__));                        \
      return -1;                                                                          \
    }                                                                                     \
  } while (0)
#define ICS_CHECK(cond, msg)                                                              \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      ::ics::set_error(std::string(msg) + " (" #cond ") @" + __FILE__ + ":" +             \
                       std::to_string(__LINE__));                                         \
      return -1;                                                                          \
    }                                                                                     \
  } while (0)
#define ICS_TRY(expr)                                                                     \
  do {                                                                                    \
    if ((expr) != 0) return -1;                                                           \
  } while (0)

// ---------------------------------------------------------------- activations
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2 };
constexpr float kLeaky = 0.3f;     // keras LeakyReLU default alpha (vae/lattice_vae.py:175)
constexpr float kBnEps = 1e-3f;    // keras BatchNormalization default epsilon
constexpr float kKEps = 1e-7f;     // keras.backend.epsilon()
constexpr float kPoolTieTol = 1e-5f;

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_LRELU) return v > 0.f ? v : kLeaky * v;
  return v;
}
// derivative given the activation OUTPUT (or input: same sign for these monotone acts)
__device__ __forceinline__ float act_grad(float v, int act) {
  if (act == ACT_RELU) return v > 0.f ? 1.f : 0.f;
  if (act == ACT_LRELU) return v > 0.f ? 1.f : kLeaky;
  return 1.f;
}

// ---------------------------------------------------------------- A-operand sources of a conv
// One input source of a convolution's (virtual) input tensor.  The virtual input is the channel
// concatenation of up to two sources, each optionally (a) BatchNorm-applied on the fly
// (v*scale[c]+shift[c], then `act`), (b) nearest-upsampled by 2, (c) a per-sample broadcast vector
// (the K.tile'd condition of vae/lattice_vae.py:167-169).  Zero "same" padding applies AFTER that.
struct ConvSrc {
  const float* p;       // NDHWC tensor at the source's own resolution (or [B][bc_n] if bcast)
  const float* scale;   // per-channel affine or nullptr
  const float* shift;
  int C;                // channels this source contributes
  int up;               // 1: source is at S/2 and is nearest-upsampled
  int act;              // activation after the affine
  int bcast;            // >0: broadcast mode, value = p[b*bcast + (c % bcast)]
};

// A/B switches that route a layer through the general kernels instead of a fast path (tests/test_gpu_switches.py,
// benchmarking).  They are read from the environment ONCE PER HANDLE at creation (conv_flags_from_env) and travel
// with the geometry -- no process-global state decides which kernel a launch takes.
enum ConvFlags : int {
  CF_NO_REUSE = 1,        // ICSG3D_NO_REUSE: no dx-reuse forward / backward-data pipeline
  CF_NO_WGRAD3 = 2,       // ICSG3D_NO_WGRAD3: no dx-reuse backward-weight kernels
  CF_NO_WGRAD3S = 4,      // ICSG3D_NO_WGRAD3S: no wave-uniform-loader variant of it
  CF_NO_FWD_SPLITK = 8,   // ICSG3D_NO_FWD_SPLITK
  CF_NO_THIN_N = 16,      // ICSG3D_NO_THIN_N: Cout <= 4 layers through the MFMA kernels
  CF_NO_UPSPLIT = 32,     // ICSG3D_NO_UPSPLIT: direct 27-tap evaluation of upsampled inputs
  CF_NO_THIN_C = 64,      // ICSG3D_NO_THIN_C: single-channel-input layers through the MFMA kernels
  CF_NO_COND_FOLD = 128,  // ICSG3D_NO_COND_FOLD: the VAE encoder's K.tile'd condition as materialised input channels
  CF_NO_WINO = 256,       // ICSG3D_NO_WINO: 3x3x3 layers through the 27-tap implicit GEMM instead of Winograd F(2,3)
  CF_NO_WINO_WGRAD = 512, // ICSG3D_NO_WINO_WGRAD: backward-weight through the direct kernels, Winograd forward/backward-data kept
  CF_NO_WINO64 = 1024,    // ICSG3D_NO_WINO64: Winograd forward/backward-data through the 32-tile x 32-channel kernel only
  CF_NO_UP3 = 2048,       // ICSG3D_NO_UP3: upsampled channels forward through the 8 parity-class GEMMs (64 products per low-res
                          // voxel) instead of the 27-product kernel
  CF_NO_THIN1_2STAGE = 4096,   // ICSG3D_NO_THIN1_2STAGE: Cout = 1 layers through the direct stencils
  CF_NO_FAST_BNBWD = 8192,     // ICSG3D_NO_FAST_BNBWD: BatchNorm backward through the generic (run-time source) kernels
  CF_NO_FUSED_HEAD = 16384,    // ICSG3D_NO_FUSED_HEAD: 1x1x1 head GEMM + loss kernel / implicit-GEMM head backward-data
  CF_NO_BWD_FOLD = 32768,      // ICSG3D_NO_BWD_FOLD: BatchNorm-backward sums in their own pass, not in the consumer's dgrad
  CF_UP3_BIG_ALWAYS = 65536,   // ICSG3D_UP3_BIG_MIN_WG=1: 32-voxel conv_up3 workgroups wherever the shape allows (tests)
  CF_NO_WINOG = 262144,        // ICSG3D_NO_WINOG: S = 4 layers through the 27-tap kernels instead of the Winograd-domain
                               // batched GEMMs of conv_winog.hip (round 4)
  CF_NO_HEAD_BNFUSE = 524288,  // ICSG3D_NO_HEAD_BNFUSE: c18's BatchNorm backward as its own pass, not in the head's backward-data
  CF_NO_UP3N = 1 << 21,        // ICSG3D_NO_UP3N: the VAE decoder's narrow upsampled layers through the 8-tap parity GEMMs
  CF_NO_DGRAD_BNFUSE = 1 << 22,  // ICSG3D_NO_DGRAD_BNFUSE: c17 / c15 BatchNorm backward as its own pass behind c18 / c16 backward-data
  CF_NO_POOL_PRESUM = 1 << 23,   // ICSG3D_NO_POOL_PRESUM: pool-only layers (perceptual taps) keep the BatchNorm-backward reduce pass
  CF_NO_HEAD_LABELS = 1 << 24,   // ICSG3D_NO_HEAD_LABELS: inference labels from the stored probabilities (labels_kernel), not from the fused head's registers
  CF_ZBATCH = 1 << 20,         // internal: conv_fwd_kernel runs gridDim.z independent GEMMs (launch_gemm_zbatch)
  // (131072 was CF_NO_TICKET, the per-layer bias-gradient finalize of round 3: removed in round 6 with the other switches
  // DESIGN.md records as rejected -- ICSG3D_SIDE_STREAM, ICSG3D_NO_ARENA)
};
int conv_flags_from_env();

struct ConvGeom {
  int B, S, lgS;        // batch, cubic spatial extent (power of two) and its log2
  int Cin, Cout;        // logical channel counts of this GEMM (Cin = sum of source C)
  int taps;             // 27 (3x3x3 same) or 1 (1x1x1)
  int Kpad, Npad;       // padded GEMM K (= taps*Cin rounded up to 32) and N (Cout rounded up to 32)
  int flags = 0;        // ConvFlags of the owning handle
};

// Backward-data launches can fold the NEXT layer's BatchNorm-backward reductions into their epilogue: the tile a
// block just produced is dO of the producer layer P; with P's stored activations s and batch statistics at hand
// the block adds its share of  sum d  and  sum d*xhat  (d = dO * post_act'(BN(s)), xhat = (s - mean) * rstd)
// per channel -- the pass bn_bwd_reduce_kernel would otherwise make over dO and s.
struct BwdStat {
  const float* s = nullptr;       // P's stored activations [M][ld]
  const float* mean = nullptr;
  const float* rstd = nullptr;
  const float* scale = nullptr;   // P's BN affine (only read when post_act != ACT_NONE)
  const float* shift = nullptr;
  float* partial = nullptr;       // out: [2][Npad][gridM] per-block column sums (block index fastest)
  int post_act = ACT_NONE;
  int ld = 0;
  // round 4: P's whole BatchNorm-backward apply in the launch's epilogue (conv_wino64.hip FOLD = 2) instead of the sums:
  const float* abc = nullptr;     // [3][N]: dy_P = relu'(s) (a d + b s + c), from conv_bnfuse_kernel; the launch's `out` is P's dy
  float* db_partial = nullptr;    // out: [blocks][N] column sums of the written dy_P
  // with abc: P's second consumer is a MaxPool3D whose input gradient is added to d first (FOLD = 3)
  const float* pool_d = nullptr;            // [rows/8][pool_ld] gradient w.r.t. the pooled tensor
  int pool_ld = 0;
  const unsigned char* pool_mask = nullptr; // [rows/8][N] bit k: window element k receives the gradient (launch_pool_fwd)
};

// ---------------------------------------------------------------- kernel launchers (conv_igemm.hip)
// out[m*ldo + n] = pre_act( sum_k A[m][k] * W[k][n] + bias[n] ),  m over B*S^3 voxels.
// stat_partial: optional [3][Npad][gridM] (count, mean, M2) of the stored values per block column (block index fastest).
int launch_conv_fwd(hipStream_t st, const ConvGeom& g, const ConvSrc* src, int nsrc,
                    const float* wpacked, const float* bias, float* out, int ldo, int pre_act,
                    float* stat_partial, int* stat_rows_per_block, int accumulate = 0,
                    float* splitk_ws = nullptr, size_t splitk_ws_floats = 0, const BwdStat* bwd = nullptr,
                    int* bwd_blocks = nullptr);
// bwd / bwd_blocks: see BwdStat; *bwd_blocks = number of row blocks written to bwd->partial, or 0 when this launch
// could not fold the reductions (split-K or thin-N path) and the caller must run them separately
size_t conv_fwd_workspace_floats(const ConvGeom& g, const ConvSrc* src, int nsrc);
// Thin-C direct stencil forward (see conv_igemm.hip): plain single source with cin_log in {1,4,16} channels, Cout in
// {16,32}; wstride = input channels per tap in the packed weights (>= cin_log)
bool conv_thin_c_ok(const ConvGeom& g, const ConvSrc& s0, int nsrc, int cin_log);
// backward-weight of a single-channel-input layer: dw[(tap*row_pitch)*ldw + co]; phase as launch_conv_wgrad
size_t conv_thin_c_wgrad_workspace_floats(const ConvGeom& g);
int launch_conv_wgrad_thin_c(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* dy, int ldy, float* dw,
                             int ldw, int row_pitch, float* ws, size_t ws_floats, int phase);
int launch_conv_fwd_thin_c(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, int cin_log, int wstride,
                           const float* wp, const float* bias, float* out, int ldo, int pre_act, float* stat_partial,
                           int* rows_per_block, const float* pos_bias = nullptr);
// pos_bias: optional [B][27][Cout] position-dependent bias (border class = first/interior/last plane per axis)
int launch_conv_fwd_par(hipStream_t st, const ConvGeom& g_lowres, const ConvSrc& src, const float* wpar, float* out,
                        int ldo, const float* bias = nullptr, int pre_act = ACT_NONE, float* stat_partial = nullptr,
                        int* stat_blocks = nullptr);
int launch_pack_par(hipStream_t st, const float* w, int Cin_total, int Cout, int c_off, int Cu, float* dst, int Kpad,
                    int Npad);
int launch_pack_fwd_sub(hipStream_t st, const float* w, int taps, int Cin_total, int Cout, int c_off, int Csub,
                        float* dst, int Kpad, int Npad);
// partial[split][k][n] = sum_{m in split} A[m][k] * dy[m*ldy + n];  then reduced into dw[k*ldw+n].
int launch_conv_wgrad(hipStream_t st, const ConvGeom& g, const ConvSrc* src, int nsrc,
                      const float* dy, int ldy, float* dw, int ldw, float* workspace,
                      size_t workspace_floats, int sub_rows = 0, int row_pitch = 0, int row_off = 0,
                      int phase = 0);   // 0: GEMM + split reduction; 1: GEMM only; 2: reduction only
// exact template instantiation (as rocprofv3 names it) of the last conv GEMM kernel launched by this thread
const char* conv_last_kernel_id();
void conv_set_last_kernel_id(const char* id);
int launch_pack_sub(hipStream_t st, const float* w, int taps, int Cin_total, int Cout, int c_off, int Csub,
                    int flip, float* dst, int Kpad, int Npad);
size_t conv_wgrad_workspace_floats(const ConvGeom& g, const ConvSrc* src, int nsrc);
int conv_fwd_rows_per_block(const ConvGeom& g);
int launch_conv_wgrad_ablate(hipStream_t st, const ConvGeom& g, const ConvSrc* src, const float* dy, int ldy,
                             float* workspace, int ablate);
int launch_conv_fwd_ablate(hipStream_t st, const ConvGeom& g, const ConvSrc* src, int nsrc, const float* wp,
                           float* out, int ldo, int ablate);
// weight packing: Keras [taps][Cin][Cout] -> [Kpad/4][Npad][4]
int launch_pack_fwd(hipStream_t st, const float* w, int K, int N, float* dst, int Kpad, int Npad,
                    int k_off, int n_off, int zero_first, int cin_log = 0, int cin_phys = 0);
// One-launch packing: between record_begin and record_end the launch_pack_* functions append jobs to a table
// instead of launching; the engine uploads the table once and replays it with launch_pack_table after every
// parameter change (destination buffers must be zero-initialised: jobs write only their valid elements).
void* pack_table_record_begin();
int pack_table_record_end(void* handle, std::vector<unsigned char>* bytes, int* njobs, unsigned* nblocks);
int launch_pack_table(hipStream_t st, const void* d_jobs, int njobs, unsigned nblocks);
int launch_materialize_input(hipStream_t st, const ConvSrc* src, int nsrc, int Cin, int CinG, int B, int S,
                             float* out);
int launch_pack_bwd(hipStream_t st, const float* w, int taps, int Cin, int Cout, float* dst,
                    int Kpad, int Npad, int cout_total, int co_off, int zero_first);

// ---------------------------------------------------------------- Winograd F(2x2x2, 3x3x3) path (conv_wino.hip)
// forward / backward-data of a 3x3x3 "same" convolution with one plain (optionally BatchNorm-affine) source,
// Cin % 32 == 0, Cout % 32 == 0, S >= 8.  wt: weights transformed by launch_pack_wino.  stat_partial / accumulate:
// as launch_conv_fwd; *rows_per_block = 256.
bool conv_wino_ok(const ConvGeom& g, const ConvSrc* src, int nsrc);
size_t conv_wino_weight_floats(int Cin, int Cout);
int launch_conv_fwd_wino(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* wt, const float* bias,
                         float* out, int ldo, int pre_act, float* stat_partial, int* rows_per_block, int accumulate,
                         int layout, const BwdStat* bwd = nullptr, int* bwd_blocks = nullptr);   // bwd / bwd_blocks: as launch_conv_fwd
// layout: the layout `wt` was packed in (conv_wino_layout at the geometry the weights were packed for, i.e. the
// maximum batch); 1 selects conv_wino64.hip's kernel, 0 conv_wino.hip's.
// dst[Nn/32][K/4][64 f][2][32][2] = (G (x) G (x) G) applied to the 27 taps of
//   bwd = 0: w[tap][c_off + k][n]          (forward: K = Csub input channels of Cin_total, Nn = Cout)
//   bwd = 1: w[26 - tap][c_off + n][k]     (backward-data: K = Cout, Nn = Csub input channels)
// layout 0: conv_wino.hip's operand; layout 1: conv_wino64.hip's dst[Nn/64][K/4][64 f][4 k][16 n][4 column blocks].
// conv_wino_layout(g): the layout to pack for GEMM geometry g (g.Cin = K, g.Cout = Nn, g.B = the LARGEST batch the
// weights will serve): 1 iff conv_wino64_ok holds there (it then holds at every smaller batch).
int launch_pack_wino(hipStream_t st, const float* w, int Cin_total, int Cout, int c_off, int Csub, int bwd, float* dst,
                     int layout = 0);
int conv_wino_layout(const ConvGeom& g);
// the 16-tile x 64-channel kernel shape (conv_wino64.hip): Cout % 64 == 0; launch_conv_fwd_wino dispatches to it
bool conv_wino64_ok(const ConvGeom& g, const ConvSrc* src, int nsrc);
int launch_conv_fwd_wino64(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* wt, const float* bias,
                           float* out, int ldo, int pre_act, float* stat_partial, int* rows_per_block, int accumulate,
                           const BwdStat* bwd = nullptr, int* bwd_blocks = nullptr);
// backward-weight in the Winograd domain; workspace / sub_rows / row_pitch / row_off / phase as launch_conv_wgrad
bool conv_wino_wgrad_ok(const ConvGeom& g, const ConvSrc* src, int nsrc);
size_t conv_wino_wgrad_workspace_floats(const ConvGeom& g);
int launch_conv_wgrad_wino(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* dy, int ldy, float* dw,
                           int ldw, float* ws, size_t ws_floats, int sub_rows, int row_pitch, int row_off, int phase);
// ---------------------------------------------------------------- upsampled input, 27 products per low-res voxel (conv_up3.hip)
// g = the LOW-RES geometry (geom_par_fwd: S = low-res extent, Cin = Cu upsampled channels, Cout); s0 = the low-res source.
// *stat_blocks = row blocks (128 fine voxels each) written to stat_partial.
bool conv_up3_ok(const ConvGeom& g, const ConvSrc& s0);
size_t conv_up3_weight_floats(int Cu, int Cout);
int launch_pack_up3(hipStream_t st, const float* w, int Cin_total, int Cout, int c_off, int Cu, float* dst);
int launch_conv_fwd_up3(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* wt, const float* bias,
                        float* out, int ldo, int pre_act, float* stat_partial, int* stat_blocks, int accumulate);
// ---------------------------------------------------------------- Winograd-domain GEMMs for the S = 4 layers (conv_winog.hip)
// 3x3x3 "same" convolution on 4^3 grids as input transform -> 64 plain GEMMs (launch_gemm_zbatch) -> output transform.
// wg: weights transformed by launch_pack_wino(..., layout 2): [64][K/4][Nn][4].  v / m / z scratch: conv_winog_scratch_floats.
bool conv_winog_ok(const ConvGeom& g, const ConvSrc* src, int nsrc);
bool conv_winog_wgrad_ok(const ConvGeom& g, const ConvSrc* src, int nsrc);
size_t conv_winog_weight_floats(int Cin, int Cout);
void conv_winog_scratch_floats(const ConvGeom& g, size_t* v, size_t* m, size_t* z);
// *rows_per_block = 64 (BatchNorm partials).  vt_keep (or nullptr): the transposed transform [64][Cin][T] backward-weight reads
int launch_conv_fwd_winog(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* wg, const float* bias, float* out,
                          int ldo, int pre_act, float* stat_partial, int* rows_per_block, float* v_scratch, float* m_scratch,
                          float* vt_keep);
int launch_conv_wgrad_winog(hipStream_t st, const ConvGeom& g, const float* vt, const float* dy, int ldy, float* dw, int ldw,
                            int row_pitch, int row_off, float* z_scratch, float* m_scratch);
// nz independent plain GEMMs out_z[M][N] = A_z[M][K] x W_z[K][N] (conv_igemm.hip; W_z packed [K/4][N][4])
int launch_gemm_zbatch(hipStream_t st, int nz, int M, int K, int N, const float* A, const float* Wp, float* out, int flags);
// the same 27-product form for NARROW outputs (Cout 16 / 32: the VAE decoder's d2 / d3), conv_up3n.hip; *stat_blocks = blocks
// of 512 fine voxels written to stat_partial
bool conv_up3n_ok(const ConvGeom& g, const ConvSrc& s0);
size_t conv_up3n_weight_floats(int Cu, int Cout);
int launch_pack_up3n(hipStream_t st, const float* w, int Cin_total, int Cout, int c_off, int Cu, float* dst);
int launch_conv_fwd_up3n(hipStream_t st, const ConvGeom& g, const ConvSrc& s0, const float* wt, const float* bias, float* out,
                         int ldo, int pre_act, float* stat_partial, int* stat_blocks);
// fixed-order reduction of split-K weight-gradient partials ws[split][k][n] into dw (conv_igemm.hip)
int launch_wgrad_reduce_splits(hipStream_t st, const float* ws, int nsplit, size_t n_elems, int N, float* dw, int ldw,
                               int sub_rows, int row_pitch, int row_off);

}  // namespace ics
