// Host-side engine: wires the HIP kernels into the two reference graphs
//   AtomUnet.unet_3d_multiclass      /root/reference/unet/unet.py:272-355
//   LatticeDFCVAE encoder/decoder    /root/reference/vae/lattice_vae.py:160-230
// and their training steps (unet.py:252-259,370 ; lattice_vae.py:232-270,296), and exports the
// C ABI declared in include/icsg3d.h.  One engine = one HIP stream on one device; everything a
// step needs lives in HBM for the life of the handle (activations for B=32, d=32 are ~7 GB of 288).
#include "common.h"
#include "elementwise.h"
#include "segment.h"
#include "../../include/icsg3d.h"

#include <rccl/rccl.h>
#include <roctracer/roctx.h>

#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace ics {

// roctx ranges around the phases of a step: visible in `rocprofv3 --marker-trace` next to the kernel trace, free
// when no tool is attached
struct Range {
  explicit Range(const char* name) { roctxRangePushA(name); }
  ~Range() { roctxRangePop(); }
};

std::atomic<long long> g_kernel_launches{0};
thread_local LaunchTimer* tl_launch_timer = nullptr;
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

static int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------
struct Tensor {
  std::string name;
  std::vector<int64_t> dims;
  size_t count = 0;
  bool trainable = false;
  size_t off = 0;        // offset into the flat parameter buffer when trainable
  float* ptr = nullptr;  // device pointer (state tensors own theirs; trainable: P + off)
};

struct Profiler : LaunchTimer {
  struct Row { std::string label; int64_t launches = 0; double ms = 0, flop = 0, bytes = 0; };
  struct Pending { int row; std::vector<hipEvent_t> ev; std::string label; double flop, bytes; };   // ev: (start, stop) pairs
  bool on = false;
  bool active = false;                  // the bracket being measured passed the filter
  // default: every kernel launched inside a bracket carries its own event pair (common.h LaunchTimer) and the bracket's
  // time is the sum of its kernels' durations; ICSG3D_PROF_BRACKET=1: two hipEventRecord markers around the bracket (also
  // sees what is not a kernel of this library -- RCCL calls, copies -- and costs ~10 us of stream time per bracket)
  bool bracket = getenv("ICSG3D_PROF_BRACKET") != nullptr;
  std::string filter;                   // non-empty: only launch sites whose label starts with one of its ';'-separated prefixes get events
  std::vector<Row> rows;
  std::map<std::string, int> index;
  std::vector<Pending> pending;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
  }
  void next(hipEvent_t* a, hipEvent_t* b) override {
    if (!active || pending.empty()) { *a = *b = nullptr; return; }   // a bracket left open by an error path: plain launch
    *a = get(); *b = get();
    pending.back().ev.push_back(*a); pending.back().ev.push_back(*b);
  }
  // A label ending in '|' gets the exact template instantiation of the conv kernel launched inside the
  // bracket appended at end() (conv_last_kernel_id(): the name rocprofv3 prints for the same launch).
  void begin(hipStream_t st, const std::string& label, double flop, double bytes) {
    if (!on) return;
    active = filter.empty();
    for (size_t i = 0; !active && i < filter.size();) {          // ';'-separated list of label prefixes
      size_t j = filter.find(';', i);
      if (j == std::string::npos) j = filter.size();
      active = j > i && label.compare(0, j - i, filter, i, j - i) == 0;
      i = j + 1;
    }
    if (!active) return;
    pending.push_back(Pending{-1, {}, label, flop, bytes});
    if (bracket) { hipEvent_t a = get(); (void)hipEventRecord(a, st); pending.back().ev.push_back(a); }
    else tl_launch_timer = this;
  }
  void end(hipStream_t st) {
    if (!on || !active) return;
    active = false;
    tl_launch_timer = nullptr;
    Pending& p = pending.back();
    if (bracket) { hipEvent_t b = get(); (void)hipEventRecord(b, st); p.ev.push_back(b); }
    if (!p.label.empty() && p.label.back() == '|') p.label += conv_last_kernel_id();
    auto it = index.find(p.label);
    int r;
    if (it == index.end()) { r = (int)rows.size(); rows.push_back(Row{p.label}); index[p.label] = r; }
    else r = it->second;
    rows[r].launches += 1; rows[r].flop += p.flop; rows[r].bytes += p.bytes;
    p.row = r;
  }
  void resolve() {   // call after a stream sync
    if (tl_launch_timer == this) tl_launch_timer = nullptr;
    active = false;
    for (auto& p : pending) {
      for (size_t i = 0; i + 1 < p.ev.size(); i += 2) {
        float ms = 0.f;
        if (p.row >= 0 && hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]) == hipSuccess) rows[p.row].ms += ms;
      }
      for (hipEvent_t e : p.ev) pool.push_back(e);
    }
    pending.clear();
  }
  void reset() { resolve(); rows.clear(); index.clear(); }
  ~Profiler() {
    if (tl_launch_timer == this) tl_launch_timer = nullptr;
    for (auto& p : pending) for (hipEvent_t e : p.ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
  }
};

struct ConvLayer {
  std::string name;
  int Cin = 0, Cout = 0, taps = 27, S = 1;
  int flags = 0;                        // ConvFlags of the owning handle (carried into every ConvGeom)
  bool cond_fold = false;               // VAE e0 at C = 1: the K.tile'd condition channels are folded into a
  float* cond_T = nullptr;              // position-dependent bias [maxB][27][Cout] (forward) / region sums (backward)
  int bwd_pre_nblk = 0, bwd_pre_ld = 0; // > 0: this layer's BN-backward sums are in ws_bwd2 (set by the consumer's
                                        // backward-data launch, consumed by conv_backward)
  int CinG = 0;                         // GEMM input channels: Cin, or Cin zero-padded to 4/16/32k when the
                                        // virtual input is materialised (thin / broadcast inputs: c1, e0)
  float* pad_in = nullptr;              // [M][CinG] materialised input (padded layers only)
  float* dw_phys = nullptr;             // [taps*CinG][Cout] weight gradient before un-padding
  ConvSrc vsrc[2];                      // the virtual sources feeding pad_in
  int nvsrc = 0;
  int pre_act = ACT_NONE, has_bn = 0, post_act = ACT_NONE;
  int Kpad = 0, Npad = 0, Kpad_b = 0, Npad_b = 0;
  int t_w = -1, t_b = -1, t_gamma = -1, t_beta = -1;   // tensor indices
  // BatchNorm state
  float *mm = nullptr, *mv = nullptr, *mean = nullptr, *rstd = nullptr, *scale = nullptr, *shift = nullptr;
  float *wp = nullptr, *wf = nullptr;   // packed forward / backward-data weights
  float *ww = nullptr, *wwb = nullptr;  // Winograd-transformed forward / backward-data weights (conv_wino.hip); of the
                                        // skip channels only in an up-split layer.  nullptr: layer not served
  int ww_layout = 0, wwb_layout = 0;    // decided once at the maximum batch (conv_wino_layout), used by pack AND launch
  bool wino_w = false;                  // backward-weight in the Winograd domain
  // S = 4 layers (conv_winog.hip): Winograd-domain batched GEMMs.  wg / wgb: transformed forward / backward-data weights
  // [64][K/4][N][4]; wg_v / wg_m: transform and GEMM-result scratch; wg_vt: the forward's transposed transform, kept for
  // backward-weight (wg_w); wg_z / wg_du: backward-weight scratch (its own: the weight gradients may run on the side stream)
  int wg_vt_batch = 0;                  // batch whose transform wg_vt holds (0: none)
  float *wg = nullptr, *wgb = nullptr, *wg_v = nullptr, *wg_m = nullptr, *wg_vt = nullptr, *wg_z = nullptr, *wg_du = nullptr;
  bool wg_w = false;
  float* s = nullptr;                   // stored output [M][Cout]
  float* dy = nullptr;                  // grad w.r.t. conv output [M][Cout]
  float* dA = nullptr;                  // grad w.r.t. virtual input [M][Cin]
  float* c1c2 = nullptr;
  float* db_partial = nullptr;          // [blocks][Cout] bias-gradient partials of the BatchNorm-backward apply pass (own buffer:
                                        // finalized with every other layer's in one launch, Net::colsum)
  int db_blocks = 0;                    // rows of db_partial the last backward pass wrote
  size_t db_rows = 0;                   // rows db_partial holds
  // round 4 (conv_bnfuse_kernel): this layer's BatchNorm-backward apply inside its CONSUMER's backward-data launch
  float* xs = nullptr;                  // [2][Cout] xhat as an affine of the stored activation
  float* abc = nullptr;                 // [3][Cout] the apply's per-channel constants
  bool dy_ready = false;                // dy was written by the consumer's backward-data launch: skip bn_act_bwd
  // ... and the same for a layer with TWO consumers, a skip concat and a MaxPool3D (c2): the skip consumer's backward-data
  // launch is deferred until the pool consumer's gradient exists and then writes this layer's dy (conv_wino64.hip FOLD = 3)
  unsigned char* tie_mask = nullptr;    // [M/8][Cout] which window elements receive the pooled gradient (launch_pool_fwd)
  int tie_batch = 0;                    // batch whose masks / sums the buffers hold (0: none)
  float* tie_ssum = nullptr;            // [M/8][Cout] sum of their stored activations
  double* bn_sums = nullptr;            // [2][Cout] the skip consumer's share of (sum d, sum d xhat)
  ConvLayer* skip_prod = nullptr;       // on the CONSUMER (c17): the layer its skip channels come from
  ConvLayer* deferred_skip = nullptr;   // on the PRODUCER (c2): consumer whose skip backward-data launch is pending
  float* pooled = nullptr;              // MaxPool3D(o) if a pool follows
  unsigned char* pool_idx = nullptr;
  ConvSrc src[2];
  int nsrc = 1;
  // "up-split" backward for [skip | nearest-upsampled] concat inputs (c13/c15/c17): the up channels'
  // gradients are GEMMs over the LOW-RES grid against dy pooled per tap (27 -> 27/8 of the FLOPs).
  bool split_up = false;
  bool wf_stale = false;                       // up-split layer: wf not rebuilt since the last parameter change
  int Cs = 0, Cu = 0;
  float *wf_skip = nullptr, *w_up = nullptr;   // packed: dgrad of the skip channels; dxl = dyS x W_up
  float *wp_skip = nullptr, *wp_par = nullptr; // packed forward: skip channels (27 taps); 8 parity classes x 8 taps
  float* w_up3 = nullptr;                      // upsampled channels, 27-product form (conv_up3.hip): replaces wp_par
  float* w_up3n = nullptr;                     // the same for narrow outputs (conv_up3n.hip; layers without skip channels)
  float *dyS = nullptr;                        // [M/8][ldS] tap-pooled dy, ldS = 27*Cout rounded up to 32 (pad = 0)
  int ldS = 0;
  float *dA_skip = nullptr, *dxl = nullptr;    // [M][Cs] grad of the skip input; [M/8][Cu] grad of the low-res input
  float *dw_up = nullptr;                      // [Cu][27*Cout] GEMM result before the permute into G
};

struct Net {
  int kind = 0;   // 0 U-Net, 1 VAE
  int device = 0;
  int flags = 0;  // ConvFlags, read from the environment when the handle is created
  hipStream_t st = nullptr;
  // Second stream for the weight-gradient GEMMs (they need only a layer's dy, and the chain bn_bwd -> backward-data -> next
  // layer never waits for them until the gradients are consumed).  The DFC-VAE engine uses it (round 5: 5.71 -> 5.44 ms);
  // for the U-Net engine it measured -0.5 % in rounds 1 and 3 at the price of meaningless per-kernel durations and of the
  // BatchNorm-backward fusions (which need the weight-gradient GEMM's result in stream order): rejected, and the opt-in
  // switch (ICSG3D_SIDE_STREAM) was removed in round 6.
  hipStream_t st2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_on = true, side_dirty = false;
  bool pm_side = false;               // VAE engine: the perceptual y_true pass on st2 (vae_step)
  int maxB = 0, d = 0, C = 0;
  std::vector<void*> allocs;
  std::vector<Tensor> tensors;
  std::map<std::string, int> tindex;
  size_t nparams = 0;
  float *P = nullptr, *G = nullptr, *Mo = nullptr, *Vo = nullptr;
  int adam_t = 0;
  float lr = 1e-6f;
  bool packed_valid = false;
  void* d_pack_jobs = nullptr;       // device copy of the recorded pack-job table (one launch per step)
  int pack_njobs = -1;               // -1: not recorded yet
  unsigned pack_nblocks = 0;
  int pool_ties_all = 1, bn_unbias = 1;
  int bce_from_logits = 0;            // binary_crossentropy: 0 clipped probabilities (default), 1 TF 2.1's logits short-circuit
  Profiler prof;
  std::vector<std::unique_ptr<ConvLayer>> layers;
  // shared workspaces
  float* ws_stat = nullptr;  size_t ws_stat_n = 0;
  float* ws_bwd = nullptr;   size_t ws_bwd_n = 0;
  float* ws_bwd2 = nullptr;  size_t ws_bwd2_n = 0;    // BN-backward sums folded into a backward-data epilogue
  float* ws_wgrad = nullptr; size_t ws_wgrad_n = 0;
  float* ws_fwd = nullptr;   size_t ws_fwd_n = 0;     // forward / backward-data split-K partial sums
  bool want_tie_stats = false;      // forward passes write the max-pool tie masks / sums although no weight gradients follow
                                    // (the perceptual U-Net's pass over the reconstruction inside a DFC-VAE train step)
  bool want_wgrad_inputs = false;   // forward passes keep what a backward-WEIGHT pass needs (conv_winog.hip's transposed
                                    // transform): only inside the engine's own train step
  bool splitk = false;       // split-K only inside train steps: its plan depends on the batch size, and
                             // inference keeps "a sample's output does not depend on its batch" bit-exact
  float* fws() const { return splitk ? ws_fwd : nullptr; }
  double* ws_dbl = nullptr;  size_t ws_dbl_n = 0;
  float* d_metrics = nullptr;
  // data parallel: gradient buckets are all-reduced on comm_st while the backward pass
  // continues on st; BN moving statistics are averaged over the ranks every step; sync_bn = global-batch
  // BatchNorm statistics (per-layer all-gather / all-reduce of the per-channel sums on st)
  ncclComm_t comm = nullptr;
  ncclComm_t comm_bn = nullptr;      // SyncBN's own communicator (ncclCommSplit of comm): its small per-layer collectives
                                     // on st no longer serialise behind the gradient buckets on comm_st
  ncclComm_t small() const { return comm_bn ? comm_bn : comm; }   // the communicator of the small collectives on st
  int rank = 0, nranks = 1;
  double* d_red = nullptr;           // [16] small reductions (timing max, metric sums); [8..14] = the last head metric sums
  hipStream_t comm_st = nullptr;
  hipEvent_t ev_grad = nullptr, ev_comm = nullptr;
  size_t bucket_hi = 0, bucket_min = 0;   // gradients at offsets >= bucket_hi are already being reduced
  int buckets_issued = 0;
  bool overlap = true;               // ICSG3D_DP_NO_OVERLAP=1: one all-reduce on st after the backward pass
  int sync_bn = 0;
  float* bn_slab = nullptr;          // all BN moving means / variances, contiguous (one all-reduce)
  size_t bn_slab_n = 0, bn_slab_used = 0;
  double* sync_local = nullptr;      // [3*Cmax]
  double* sync_gathered = nullptr;   // [nranks][3*Cmax]
  int sync_cmax = 0;
  BnSync bn_sync{};
  const BnSync* sync() const { return (sync_bn && comm) ? &bn_sync : nullptr; }
  int head_nblk = 0;
  ColsumJobs colsum{};               // bias-gradient finalizes pending since the last flush (colsum_flush)
  float* ws_cls = nullptr; size_t ws_cls_n = 0;   // conv_bnfuse: per-block border-class sums of a dy
  double* ws_R = nullptr;                         // [27][Cmax] class sums
  double* ws_pool = nullptr; size_t ws_pool_n = 0;  // pool_sums_kernel's block partials
  size_t bnfuse_min_bytes = (size_t)64 << 20;     // producer activation size from which the fusion pays (ICSG3D_DGRAD_BNFUSE_MIN)

  // U-Net specifics
  int ncls = 95;
  float loss_weight = 95.f;
  float* x_in = nullptr;              // [maxB][d^3][C]
  unsigned char* labels = nullptr;    // [maxB][d^3]
  ConvLayer* head = nullptr;
  float* head_bias_grad = nullptr;
  float* head_dw_tmp = nullptr;       // [128][ncls+1] head weight gradient before the soft | sig split
  float* head_xs = nullptr;           // [2][128] c18's xhat as an affine of its stored activation (head BN-fuse)
  float* tap_copy[4] = {nullptr, nullptr, nullptr, nullptr};   // perceptual taps of the pass over y_true
  const float* tap_ref[4] = {nullptr, nullptr, nullptr, nullptr};   // set per VAE step: fused tap loss/gradient
  float tap_coef[4] = {0, 0, 0, 0};
  double* tap_partial[4] = {nullptr, nullptr, nullptr, nullptr};
  int resident_batch = 0;
  hipEvent_t timer_ev[2] = {nullptr, nullptr};   // ics_net_timer_start / _stop
  hipStream_t st_own = nullptr;                  // ics_net_share_stream: the stream this handle created (st then points at another handle's)
  int last_batch = 0;                 // batch of the most recent forward (activation export)

  // VAE specifics
  Net* pm = nullptr;
  int ncond = 10, latent = 256, filters[4] = {16, 32, 64, 128};
  float alpha = 0.5f, beta = 3e-4f, pm_w[4] = {1, 1, 1, 1};
  float *cond_in = nullptr, *eps_in = nullptr, *z_buf = nullptr, *zc = nullptr, *recon = nullptr,
        *drecon = nullptr, *dmulv = nullptr;
  ConvLayer *enc_dense = nullptr, *zmulv = nullptr, *dec_dense = nullptr;

  ~Net() {
    if (canary_on && !canaries.empty()) {      // ICSG3D_DEBUG_CANARY=1: every handle -- the temporary ones of the single-op entry
      if (st) (void)hipStreamSynchronize(st);  // points too -- reports overwritten guard bytes when it goes away
      std::string msg;
      const int bad = canaries_dirty(&msg);
      if (bad) fprintf(stderr, "[icsg3d] CANARY DIRTY (engine kind %d, d %d, max batch %d, %d buffers): %s\n", kind, d, maxB, bad, msg.c_str());
    }
    if (comm_bn) ncclCommDestroy(comm_bn);
    if (comm) ncclCommDestroy(comm);
    for (void* p : allocs) (void)hipFree(p);
    for (hipEvent_t e : timer_ev) if (e) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (st2) (void)hipStreamDestroy(st2);
    if (ev_grad) (void)hipEventDestroy(ev_grad);
    if (ev_comm) (void)hipEventDestroy(ev_comm);
    if (comm_st) (void)hipStreamDestroy(comm_st);
    if (st_own) (void)hipStreamDestroy(st_own);
    else if (st) (void)hipStreamDestroy(st);
  }
  // Device memory comes out of a few large slabs per handle: buffers below kArenaBig are carved out of chunks (128 MB, doubling to 1 GB) at 2 MB alignment (256 B for small ones), larger ones get their
  // own allocation.  A handle's few hundred buffers then sit in a handful of contiguous, large-page-backed ranges whatever
  // the process allocated or freed before -- see DESIGN.md section 10 (the DFC-VAE's two-stream schedule ran 10 % slower when
  // its buffers had been allocated one by one after a U-Net training step).
  static constexpr size_t kArenaChunk = (size_t)1 << 30, kArenaBig = (size_t)1 << 28;
  char* arena_cur = nullptr;
  size_t arena_left = 0, arena_next = (size_t)1 << 27;
  bool arena_on = true;
  // ICSG3D_DEBUG_CANARY=1 (diagnosis aid, round 6): every buffer is followed by 8 KB of 0x5A that nothing may touch;
  // ics_net_check_canaries reports the buffers whose guard was written to (a kernel running past the end of its output)
  struct Canary { char* p; size_t len, payload; int index; };
  std::vector<Canary> canaries;
  bool canary_on = getenv("ICSG3D_DEBUG_CANARY") != nullptr;
  // number of guards that were written to; *msg describes the first few
  int canaries_dirty(std::string* msg) const {
    std::vector<unsigned char> host;
    int bad = 0;
    for (const auto& c : canaries) {
      host.resize(c.len);
      if (hipMemcpy(host.data(), c.p, c.len, hipMemcpyDeviceToHost) != hipSuccess) continue;
      size_t first = c.len;
      for (size_t i = 0; i < c.len; ++i)
        if (host[i] != 0x5A) { first = i; break; }
      if (first < c.len) {
        ++bad;
        if (bad <= 8 && msg)
          *msg += "alloc #" + std::to_string(c.index) + " payload " + std::to_string(c.payload) + " B: guard dirty from +" + std::to_string(first) + "; ";
      }
    }
    return bad;
  }
  template <typename T>
  int alloc(T** out, size_t n) {
    const size_t guard = canary_on ? 8192 : 0;
    const size_t bytes = n * sizeof(T) + 256 + guard;
    void* p = nullptr;
    if (!arena_on || bytes >= kArenaBig) {
      ICS_HIP(hipMalloc(&p, bytes));
      allocs.push_back(p);
    } else {
      const size_t align = bytes >= ((size_t)1 << 21) ? ((size_t)1 << 21) : 256;
      size_t pad = (align - (reinterpret_cast<uintptr_t>(arena_cur) & (align - 1))) & (align - 1);
      if (arena_cur == nullptr || pad + bytes > arena_left) {
        void* c = nullptr;
        const size_t chunk = std::max(arena_next, bytes);
        ICS_HIP(hipMalloc(&c, chunk));
        allocs.push_back(c);
        arena_cur = static_cast<char*>(c);
        arena_left = chunk;
        arena_next = std::min(kArenaChunk, arena_next * 2);
        pad = 0;
      }
      p = arena_cur + pad;
      arena_cur += pad + bytes;
      arena_left -= pad + bytes;
    }
    ICS_HIP(hipMemsetAsync(p, 0, bytes, st));
    if (guard) {
      char* g = static_cast<char*>(p) + n * sizeof(T);
      ICS_HIP(hipMemsetAsync(g, 0x5A, guard + 256, st));
      canaries.push_back(Canary{g, guard + 256, n * sizeof(T), (int)canaries.size()});
    }
    *out = reinterpret_cast<T*>(p);
    return 0;
  }
  int add_tensor(const std::string& name, std::vector<int64_t> dims, bool trainable) {
    Tensor t;
    t.name = name; t.dims = dims; t.trainable = trainable;
    t.count = 1;
    for (auto v : dims) t.count *= (size_t)v;
    if (trainable) { t.off = nparams; nparams += t.count; }
    tensors.push_back(t);
    tindex[name] = (int)tensors.size() - 1;
    return (int)tensors.size() - 1;
  }
  float* tp(int ti) const { return P + tensors[ti].off; }
  float* tg(int ti) const { return G + tensors[ti].off; }
  size_t rows(const ConvLayer& L, int B) const { return (size_t)B * L.S * L.S * L.S; }
};

// ------------------------------------------------------------------------------------------
// layer construction
// ------------------------------------------------------------------------------------------
// channel count the fast (vector) loaders accept for a given logical Cin
static int padded_cin(int Cin) {
  if (Cin % 32 == 0 || Cin == 4 || Cin == 8 || Cin == 16) return Cin;
  return Cin < 4 ? 4 : (Cin < 8 ? 8 : (Cin < 16 ? 16 : round_up(Cin, 32)));
}

static ConvLayer* add_conv(Net& n, const std::string& name, int Cin, int Cout, int taps, int S, int pre_act,
                           int has_bn, int post_act, bool dense, bool register_params = true, bool pad_input = false) {
  auto L = std::make_unique<ConvLayer>();
  L->name = name; L->Cin = Cin; L->Cout = Cout; L->taps = taps; L->S = S;
  L->flags = n.flags;
  L->CinG = pad_input ? padded_cin(Cin) : Cin;
  L->pre_act = pre_act; L->has_bn = has_bn; L->post_act = post_act;
  L->Kpad = round_up(taps * L->CinG, 32); L->Npad = round_up(Cout, 32);
  L->Kpad_b = round_up(taps * Cout, 32); L->Npad_b = round_up(L->CinG, 32);
  if (register_params) {
    if (dense) L->t_w = n.add_tensor(name + "/kernel", {Cin, Cout}, true);
    else L->t_w = n.add_tensor(name + "/kernel", {taps == 27 ? 3 : 1, taps == 27 ? 3 : 1, taps == 27 ? 3 : 1, Cin, Cout}, true);
    L->t_b = n.add_tensor(name + "/bias", {Cout}, true);
    if (has_bn) {
      L->t_gamma = n.add_tensor(name + "/gamma", {Cout}, true);
      L->t_beta = n.add_tensor(name + "/beta", {Cout}, true);
    }
  }
  n.layers.push_back(std::move(L));
  return n.layers.back().get();
}

static int alloc_layer(Net& n, ConvLayer& L, bool need_bwd, bool pooled) {
  const size_t M = n.rows(L, n.maxB);
  ICS_TRY(n.alloc(&L.wp, (size_t)L.Kpad * L.Npad));
  ICS_TRY(n.alloc(&L.s, M * L.Cout));
  if (need_bwd) {
    ICS_TRY(n.alloc(&L.wf, (size_t)L.Kpad_b * L.Npad_b));
    ICS_TRY(n.alloc(&L.dy, M * L.Cout));
    ICS_TRY(n.alloc(&L.dA, M * L.CinG));
    ICS_TRY(n.alloc(&L.c1c2, (size_t)2 * L.Cout));
    if ((L.Cout & (L.Cout - 1)) == 0 || L.Cout % 4 == 0) {
      // 512: the head's backward-data launch; M / 128: a Winograd backward-data launch (one row per tile block); the
      // BatchNorm-backward pass: its block count is NOT monotone in the batch (rows per block are rounded up to what one
      // block covers: 16 channels at 32^3 take 1280 blocks for 5 grids and 1536 for 3), so every batch size the handle
      // accepts is asked -- sizing by max_batch alone let a 3-grid step on a 5-grid handle write 16 KB past the end
      // (found by tests/tools/fuzz_steps.py, round 6)
      size_t rows = std::max((size_t)512, M / 128 + 1);
      for (int b = 1; b <= n.maxB; ++b) {
        LayerBwd lb{};
        lb.B = b; lb.S = L.S; lb.lgS = ilog2(L.S); lb.C = L.Cout;
        int rpb;
        rows = std::max(rows, (size_t)bn_bwd_num_blocks(lb, &rpb));
      }
      L.db_rows = rows;
      ICS_TRY(n.alloc(&L.db_partial, rows * L.Cout));
    }
    if (L.has_bn) { ICS_TRY(n.alloc(&L.xs, (size_t)2 * L.Cout)); ICS_TRY(n.alloc(&L.abc, (size_t)3 * L.Cout)); }
  }
  if (L.has_bn) {
    if (!n.bn_slab) {   // one slab for every layer's moving statistics (data parallel: one all-reduce)
      size_t tot = 0;
      for (auto& q : n.layers) if (q->has_bn) tot += 2 * (size_t)q->Cout;
      ICS_TRY(n.alloc(&n.bn_slab, tot));
      n.bn_slab_n = tot;
    }
    ICS_CHECK(n.bn_slab_used + 2 * (size_t)L.Cout <= n.bn_slab_n, "BN slab overflow");
    L.mm = n.bn_slab + n.bn_slab_used; L.mv = L.mm + L.Cout;
    n.bn_slab_used += 2 * (size_t)L.Cout;
    n.sync_cmax = std::max(n.sync_cmax, L.Cout);
    ICS_TRY(n.alloc(&L.mean, (size_t)L.Cout)); ICS_TRY(n.alloc(&L.rstd, (size_t)L.Cout));
    ICS_TRY(n.alloc(&L.scale, (size_t)L.Cout)); ICS_TRY(n.alloc(&L.shift, (size_t)L.Cout));
    int ti = n.add_tensor(L.name + "/moving_mean", {L.Cout}, false);
    n.tensors[ti].ptr = L.mm;
    ti = n.add_tensor(L.name + "/moving_var", {L.Cout}, false);
    n.tensors[ti].ptr = L.mv;
  }
  if (L.cond_fold) {
    ICS_TRY(n.alloc(&L.cond_T, (size_t)n.maxB * 27 * L.Cout));
  } else if (L.CinG != L.Cin) {
    ICS_TRY(n.alloc(&L.pad_in, M * L.CinG));
    if (need_bwd) ICS_TRY(n.alloc(&L.dw_phys, (size_t)L.taps * L.CinG * L.Cout));
  }
  if (pooled) {
    ICS_TRY(n.alloc(&L.pooled, M / 8 * L.Cout));
    ICS_TRY(n.alloc(&L.pool_idx, M / 8 * L.Cout));
    if (need_bwd && L.has_bn && L.Cout % 4 == 0) {
      ICS_TRY(n.alloc(&L.tie_mask, M / 8 * L.Cout));
      ICS_TRY(n.alloc(&L.tie_ssum, M / 8 * L.Cout));
      ICS_TRY(n.alloc(&L.bn_sums, (size_t)2 * L.Cout));
    }
  }
  return 0;
}

static ConvSrc src_plain(const float* p, int C) { return ConvSrc{p, nullptr, nullptr, C, 0, ACT_NONE, 0}; }
static ConvSrc src_layer(const ConvLayer& L, int up) {
  if (L.has_bn) return ConvSrc{L.s, L.scale, L.shift, L.Cout, up, L.post_act, 0};
  return ConvSrc{L.s, nullptr, nullptr, L.Cout, up, ACT_NONE, 0};
}

static ConvGeom geom_fwd(const ConvLayer& L, int B) {
  return ConvGeom{B, L.S, ilog2(L.S), L.CinG, L.Cout, L.taps, L.Kpad, L.Npad, L.flags};
}
static ConvGeom geom_bwd(const ConvLayer& L, int B) {
  return ConvGeom{B, L.S, ilog2(L.S), L.Cout, L.CinG, L.taps, L.Kpad_b, L.Npad_b, L.flags};
}

// up-split GEMM geometries (see ConvLayer::split_up)
static ConvGeom geom_up_wgrad(const ConvLayer& L, int B) {   // dw_up[Cu][27N] = xl^T x dyS over the S/2 grid
  return ConvGeom{B, L.S / 2, ilog2(L.S / 2), L.Cu, 27 * L.Cout, 1, round_up(L.Cu, 32), L.ldS, L.flags};
}
static ConvGeom geom_up_dgrad(const ConvLayer& L, int B) {   // dxl[M/8][Cu] = dyS x W_up
  return ConvGeom{B, L.S / 2, ilog2(L.S / 2), L.ldS, L.Cu, 1, L.ldS, round_up(L.Cu, 32), L.flags};
}
static ConvGeom geom_skip_wgrad(const ConvLayer& L, int B) {
  return ConvGeom{B, L.S, ilog2(L.S), L.Cs, L.Cout, L.taps, round_up(L.taps * L.Cs, 32), L.Npad, L.flags};
}
static ConvGeom geom_skip_dgrad(const ConvLayer& L, int B) {
  return ConvGeom{B, L.S, ilog2(L.S), L.Cout, L.Cs, L.taps, L.Kpad_b, round_up(L.Cs, 32), L.flags};
}
static ConvGeom geom_skip_fwd(const ConvLayer& L, int B) {
  return ConvGeom{B, L.S, ilog2(L.S), L.Cs, L.Cout, L.taps, round_up(L.taps * L.Cs, 32), L.Npad, L.flags};
}
static ConvGeom geom_par_fwd(const ConvLayer& L, int B) {    // one parity class over the S/2 grid
  return ConvGeom{B, L.S / 2, ilog2(L.S / 2), L.Cu, L.Cout, 8, 8 * L.Cu, L.Npad, L.flags};
}
static ConvSrc src_lowres(const ConvLayer& L) { ConvSrc u = L.src[L.nsrc - 1]; u.up = 0; return u; }

// Layers whose (last) source is nearest-upsampled: [skip | up] concat convs (U-Net c13/c15/c17) and the VAE
// decoder's d1..d3 (Cs = 0).
static int enable_split_up(Net& n, ConvLayer& L) {
  if (n.flags & CF_NO_UPSPLIT) return 0;
  const ConvSrc& up = L.src[L.nsrc - 1];
  const int Cs = L.nsrc == 2 ? L.src[0].C : 0;
  if (!up.up || (L.nsrc == 2 && L.src[0].up) || L.taps != 27 || L.S < 2 || L.CinG != L.Cin || Cs % 32 || up.C % 32 ||
      up.bcast)
    return 0;
  L.split_up = true;
  L.Cs = Cs; L.Cu = up.C;
  L.ldS = round_up(27 * L.Cout, 32);
  const size_t M = n.rows(L, n.maxB);
  ICS_TRY(n.alloc(&L.w_up, (size_t)L.ldS * round_up(L.Cu, 32)));
  ICS_TRY(n.alloc(&L.dyS, M / 8 * L.ldS));
  ICS_HIP(hipMemsetAsync(L.dyS, 0, M / 8 * L.ldS * sizeof(float), n.st));   // pad columns stay zero
  ICS_TRY(n.alloc(&L.dxl, M / 8 * L.Cu));
  ICS_TRY(n.alloc(&L.dw_up, (size_t)L.Cu * 27 * L.Cout));
  if (conv_up3_ok(geom_par_fwd(L, n.maxB), src_lowres(L)))
    ICS_TRY(n.alloc(&L.w_up3, conv_up3_weight_floats(L.Cu, L.Cout)));
  else if (L.Cs == 0 && conv_up3n_ok(geom_par_fwd(L, n.maxB), src_lowres(L)))
    ICS_TRY(n.alloc(&L.w_up3n, conv_up3n_weight_floats(L.Cu, L.Cout)));
  else
    ICS_TRY(n.alloc(&L.wp_par, (size_t)8 * 8 * L.Cu * L.Npad));
  if (L.Cs) {
    ICS_TRY(n.alloc(&L.wf_skip, (size_t)L.Kpad_b * round_up(L.Cs, 32)));
    ICS_TRY(n.alloc(&L.dA_skip, M * L.Cs));
    ICS_TRY(n.alloc(&L.wp_skip, (size_t)round_up(L.taps * L.Cs, 32) * L.Npad));
  }
  return 0;
}

// Winograd F(2,3) forward / backward-data for the 3x3x3 layers conv_wino_ok accepts (in an up-split layer: the skip
// channels; the upsampled channels keep their 8-tap parity GEMMs, which already do 8/27 of the work).
static ConvGeom geom_wino_fwd(const ConvLayer& L, int B) { return L.split_up ? geom_skip_fwd(L, B) : geom_fwd(L, B); }
static ConvGeom geom_wino_bwd(const ConvLayer& L, int B) {
  if (L.split_up) return geom_skip_dgrad(L, B);
  ConvGeom g = geom_bwd(L, B);
  g.Cout = L.Cin;
  return g;
}
static int enable_wino(Net& n, ConvLayer& L, bool need_bwd) {
  if ((n.flags & CF_NO_WINO) || L.taps != 27 || L.pad_in || L.cond_fold || L.CinG != L.Cin) return 0;
  if (L.split_up ? L.Cs == 0 : L.nsrc != 1) return 0;
  const int K = L.split_up ? L.Cs : L.Cin;
  if (conv_wino_ok(geom_wino_fwd(L, n.maxB), L.src, 1)) {
    ICS_TRY(n.alloc(&L.ww, conv_wino_weight_floats(K, L.Cout)));
    L.ww_layout = conv_wino_layout(geom_wino_fwd(L, n.maxB));
  }
  const ConvSrc sdy = src_plain(nullptr, L.Cout);
  if (need_bwd && conv_wino_ok(geom_wino_bwd(L, n.maxB), &sdy, 1)) {
    ICS_TRY(n.alloc(&L.wwb, conv_wino_weight_floats(K, L.Cout)));
    L.wwb_layout = conv_wino_layout(geom_wino_bwd(L, n.maxB));
  }
  L.wino_w = need_bwd && conv_wino_wgrad_ok(L.split_up ? geom_skip_wgrad(L, n.maxB) : geom_fwd(L, n.maxB), L.src, 1);
  return 0;
}

// S = 4 layers: conv_winog.hip
static int enable_winog(Net& n, ConvLayer& L, bool need_bwd) {
  if (L.ww || L.taps != 27 || L.pad_in || L.cond_fold || L.CinG != L.Cin || L.split_up || L.nsrc != 1) return 0;
  const ConvGeom g = geom_fwd(L, n.maxB);
  if (!conv_winog_ok(g, L.src, 1)) return 0;
  size_t v, m, z;
  conv_winog_scratch_floats(g, &v, &m, &z);
  ICS_TRY(n.alloc(&L.wg, conv_winog_weight_floats(L.Cin, L.Cout)));
  ICS_TRY(n.alloc(&L.wg_v, v));
  ICS_TRY(n.alloc(&L.wg_m, m));
  if (need_bwd) {
    ConvGeom gb = geom_bwd(L, n.maxB);
    gb.Cout = L.Cin;
    const ConvSrc sdy = src_plain(nullptr, L.Cout);
    if (conv_winog_ok(gb, &sdy, 1)) ICS_TRY(n.alloc(&L.wgb, conv_winog_weight_floats(L.Cin, L.Cout)));
    L.wg_w = true;        // whether a given batch qualifies (B % 4 == 0) is decided per launch
    ICS_TRY(n.alloc(&L.wg_vt, (size_t)64 * n.maxB * 8 * L.Cin));
    ICS_TRY(n.alloc(&L.wg_z, z));
    ICS_TRY(n.alloc(&L.wg_du, (size_t)64 * L.Cin * L.Cout));
  }
  return 0;
}

// workspace sizing over all layers (max batch)
static int alloc_workspaces(Net& n, bool need_bwd) {
  size_t stat = 0, bwd = 0, wg = 0, fw = 0;
  // Every plan below (tiles, split counts, blocks per pass) is a function of the batch and not necessarily a monotone
  // one, and a handle runs any batch up to max_batch (the last batch of an epoch): each workspace is sized for the
  // largest demand over ALL of them, not for max_batch alone.
  for (int b = 1; b <= n.maxB; ++b)
  for (auto& Lp : n.layers) {
    ConvLayer& L = *Lp;
    const size_t M = n.rows(L, b);
    const ConvGeom g = geom_fwd(L, b);
    // per-block BatchNorm partials [3][Npad][blocks]: the split-K finish pass (and conv_winog.hip) write 64-row blocks
    // whatever the tile of the GEMM launch -- a narrow 128 x 32 tile that splits K (the VAE decoder's thin layers when
    // ICSG3D_NO_UPSPLIT routes them through the direct kernels) needs M / 64 columns, not M / 128: sizing by the tile
    // alone left splitk_finish_kernel writing past the end (found by tests/test_gpu_switches.py once other allocations moved)
    const int rpb = std::min(conv_fwd_rows_per_block(g), 64);
    stat = std::max(stat, (M + rpb - 1) / rpb * 3 * (size_t)L.Npad);
    if (L.split_up) stat = std::max(stat, 8 * ((M / 8 + 63) / 64) * 3 * (size_t)L.Npad);   // parity launch: 8 x gridM blocks
    fw = std::max(fw, conv_fwd_workspace_floats(g, L.src, L.nsrc));
    if (need_bwd) {
      const ConvSrc sdy = src_plain(L.dy, L.Cout);
      fw = std::max(fw, conv_fwd_workspace_floats(geom_bwd(L, b), &sdy, 1));
      if (L.split_up) {
        const ConvSrc sd = src_plain(L.dyS, L.ldS);
        fw = std::max(fw, conv_fwd_workspace_floats(geom_up_dgrad(L, b), &sd, 1));
        if (L.Cs) fw = std::max(fw, conv_fwd_workspace_floats(geom_skip_dgrad(L, b), &sdy, 1));
      }
      LayerBwd lb{};
      lb.B = b; lb.S = L.S; lb.lgS = ilog2(L.S); lb.C = L.Cout;
      // Cout may be a non power of two only for the head, which never goes through layer_bwd
      if ((L.Cout & (L.Cout - 1)) == 0) bwd = std::max(bwd, layer_bwd_workspace_floats(lb));
      wg = std::max(wg, conv_wgrad_workspace_floats(g, L.src, L.nsrc));
      if (L.wino_w)
        wg = std::max(wg, conv_wino_wgrad_workspace_floats(L.split_up ? geom_skip_wgrad(L, b) : g));
      if ((L.Cin == 1 || L.cond_fold) && (L.Cout == 16 || L.Cout == 32))
        wg = std::max(wg, conv_thin_c_wgrad_workspace_floats(g));
      if (L.split_up) {
        const ConvSrc lo = src_lowres(L);
        wg = std::max(wg, conv_wgrad_workspace_floats(geom_up_wgrad(L, b), &lo, 1));
        if (L.Cs) wg = std::max(wg, conv_wgrad_workspace_floats(geom_skip_wgrad(L, b), L.src, 1));
      }
    }
  }
  n.ws_stat_n = stat; n.ws_bwd_n = std::max(bwd, (size_t)4096 * 128); n.ws_wgrad_n = wg;
  ICS_TRY(n.alloc(&n.ws_stat, stat + 16));
  // ks*M*Npad <= 768 slots x 64 x 64 floats whatever the batch: keep at least that much so that batches below
  // max_batch (which plan more splits) can still split
  n.ws_fwd_n = std::max(fw, (size_t)768 * 64 * 64);
  ICS_TRY(n.alloc(&n.ws_fwd, n.ws_fwd_n + 16));
  if (need_bwd) {
    ICS_TRY(n.alloc(&n.ws_bwd, n.ws_bwd_n + 16));
    ICS_TRY(n.alloc(&n.ws_wgrad, wg + 16));
    size_t b2 = 0;    // [2][Npad][ceil(M/64)] for the widest layer
    for (auto& Lp : n.layers)
      if (Lp->has_bn) b2 = std::max(b2, (size_t)2 * Lp->Npad * ((n.rows(*Lp, n.maxB) + 63) / 64));
    n.ws_bwd2_n = b2;
    ICS_TRY(n.alloc(&n.ws_bwd2, b2 + 16));
    size_t cls = 0; int cmax = 0;
    for (auto& Lp : n.layers)
      if (Lp->taps == 27 && Lp->S >= 3 && Lp->Cout % 4 == 0) {
        for (int b = 1; b <= n.maxB; ++b) cls = std::max(cls, conv_bnfuse_partial_floats(b, Lp->S, Lp->Cout));
        cmax = std::max(cmax, Lp->Cout);
      }
    n.ws_cls_n = cls;
    ICS_TRY(n.alloc(&n.ws_cls, cls + 16));
    ICS_TRY(n.alloc(&n.ws_R, (size_t)48 * cmax + 16));
    size_t pl = 0;
    for (auto& Lp : n.layers)
      if (Lp->tie_mask) pl = std::max(pl, pool_bnfuse_partial_doubles(n.rows(*Lp, n.maxB) / 8, Lp->Cout));
    n.ws_pool_n = pl;
    ICS_TRY(n.alloc(&n.ws_pool, pl + 16));
    if (const char* e = getenv("ICSG3D_DGRAD_BNFUSE_MIN")) n.bnfuse_min_bytes = (size_t)atoll(e);
  }
  n.ws_dbl_n = 1 << 16;
  for (auto& Lp : n.layers)
    if (Lp->cond_fold) n.ws_dbl_n = std::max(n.ws_dbl_n, cond_wgrad_workspace_doubles(n.maxB, Lp->Cout));
  ICS_TRY(n.alloc(&n.ws_dbl, n.ws_dbl_n));
  ICS_TRY(n.alloc(&n.d_metrics, (size_t)16));
  ICS_TRY(n.alloc(&n.d_red, (size_t)16));
  return 0;
}

static int alloc_params(Net& n) {
  ICS_TRY(n.alloc(&n.P, n.nparams)); ICS_TRY(n.alloc(&n.G, n.nparams));
  ICS_TRY(n.alloc(&n.Mo, n.nparams)); ICS_TRY(n.alloc(&n.Vo, n.nparams));
  for (auto& t : n.tensors)
    if (t.trainable) t.ptr = n.P + t.off;
  return 0;
}

// BN defaults: gamma 1, moving_var 1 (keras initialisers); everything else zero
__global__ void fill_kernel(float* p, size_t n, float v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
static int fill(Net& n, float* p, size_t cnt, float v) {
  ICS_LAUNCH(fill_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, n.st, p, cnt, v);
  ICS_HIP(hipGetLastError());
  return 0;
}
static int init_bn_defaults(Net& n) {
  for (auto& Lp : n.layers)
    if (Lp->has_bn) {
      ICS_TRY(fill(n, n.tp(Lp->t_gamma), Lp->Cout, 1.f));
      ICS_TRY(fill(n, Lp->mv, Lp->Cout, 1.f));
    }
  return 0;
}

// ------------------------------------------------------------------------------------------
// weight packing (after every parameter change)
// ------------------------------------------------------------------------------------------
static int pack_layer(Net& n, ConvLayer& L, bool need_bwd) {
  // a layer served by the Winograd kernels (decided once, at the maximum batch) never reads the direct-path image of
  // the same weights: ww replaces wp (wp_skip in an up-split layer), wwb replaces wf (wf_skip)
  // An up-split layer never reads the full images either: its forward always takes the [skip | up] pair, and the only
  // reader of wf is the direct backward-data path (input gradient WITHOUT parameter gradients), which no engine runs on
  // these layers today -- it re-packs on demand (wf_stale).  c13 / c15 / c17: 111 MB per step not written.
  const bool wino_f = L.ww != nullptr || L.wg != nullptr, wino_b = L.wwb != nullptr || L.wgb != nullptr;
  if (L.wg) ICS_TRY(launch_pack_wino(n.st, n.tp(L.t_w), L.Cin, L.Cout, 0, L.Cin, 0, L.wg, 2));
  if (need_bwd && L.wgb) ICS_TRY(launch_pack_wino(n.st, n.tp(L.t_w), L.Cin, L.Cout, 0, L.Cin, 1, L.wgb, 2));
  if (!wino_f && !L.split_up)
    ICS_TRY(launch_pack_fwd(n.st, n.tp(L.t_w), L.taps * L.CinG, L.Cout, L.wp, L.Kpad, L.Npad, 0, 0, 1,
                            L.CinG != L.Cin ? L.Cin : 0, L.CinG != L.Cin ? L.CinG : 0));
  if (need_bwd && L.wf && !wino_b && !L.split_up)
    ICS_TRY(launch_pack_bwd(n.st, n.tp(L.t_w), L.taps, L.Cin, L.Cout, L.wf, L.Kpad_b, L.Npad_b, L.Cout, 0, 1));
  if (L.split_up) {
    if (L.Cs && !wino_f)
      ICS_TRY(launch_pack_fwd_sub(n.st, n.tp(L.t_w), L.taps, L.Cin, L.Cout, 0, L.Cs, L.wp_skip,
                                  round_up(L.taps * L.Cs, 32), L.Npad));
    if (L.w_up3) ICS_TRY(launch_pack_up3(n.st, n.tp(L.t_w), L.Cin, L.Cout, L.Cs, L.Cu, L.w_up3));
    else if (L.w_up3n) ICS_TRY(launch_pack_up3n(n.st, n.tp(L.t_w), L.Cin, L.Cout, L.Cs, L.Cu, L.w_up3n));
    else ICS_TRY(launch_pack_par(n.st, n.tp(L.t_w), L.Cin, L.Cout, L.Cs, L.Cu, L.wp_par, 8 * L.Cu, L.Npad));
  }
  if (L.ww)
    ICS_TRY(launch_pack_wino(n.st, n.tp(L.t_w), L.Cin, L.Cout, 0, L.split_up ? L.Cs : L.Cin, 0, L.ww, L.ww_layout));
  if (need_bwd && L.wwb)
    ICS_TRY(launch_pack_wino(n.st, n.tp(L.t_w), L.Cin, L.Cout, 0, L.split_up ? L.Cs : L.Cin, 1, L.wwb, L.wwb_layout));
  if (need_bwd && L.split_up) {
    if (L.Cs && !wino_b)
      ICS_TRY(launch_pack_sub(n.st, n.tp(L.t_w), L.taps, L.Cin, L.Cout, 0, L.Cs, 1, L.wf_skip, L.Kpad_b,
                              round_up(L.Cs, 32)));
    ICS_TRY(launch_pack_sub(n.st, n.tp(L.t_w), L.taps, L.Cin, L.Cout, L.Cs, L.Cu, 0, L.w_up, L.ldS, round_up(L.Cu, 32)));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------
// generic layer forward / backward
// ------------------------------------------------------------------------------------------
static int conv_forward(Net& n, ConvLayer& L, int B, bool training, bool update_moving, const float* bias) {
  n.last_batch = B;
  const ConvGeom g = geom_fwd(L, B);
  const size_t M = n.rows(L, B);
  const bool stats = L.has_bn && training;
  int rpb = 128, par_blocks = 0;
  // single-channel input served by the direct stencils (forward AND backward-weight): the padded copy is never read
  const bool thin1_all = L.pad_in && L.nvsrc == 1 && L.Cin == 1 && conv_thin_c_ok(g, L.vsrc[0], 1, 1) && L.Cout % 4 == 0;
  if (L.pad_in && !thin1_all) {
    n.prof.begin(n.st, "materialize_input", 0, 4.0 * M * (L.Cin + L.CinG));
    ICS_TRY(launch_materialize_input(n.st, L.vsrc, L.nvsrc, L.Cin, L.CinG, B, L.S, L.pad_in));
    n.prof.end(n.st);
  }
  if (L.split_up) {
    // [skip | upsampled] input: the upsampled channels are 8 parity-class GEMMs over the low-res grid with
    // pre-summed 2x2x2 weights (8/27 of the FLOPs); the skip channels then accumulate on top and finish
    // the epilogue (bias, activation, BatchNorm statistics).
    const ConvGeom gp = geom_par_fwd(L, B), gs = geom_skip_fwd(L, B);
    const ConvSrc lo = src_lowres(L);
    const bool only_up = L.Cs == 0;   // no skip part: the parity launch carries bias, activation and statistics
    // profile rows carry the MFMA work executed: 27 products per low-res voxel (conv_up3.hip) or 64 (8 parity GEMMs)
    const bool p27 = L.w_up3 != nullptr || L.w_up3n != nullptr;
    n.prof.begin(n.st, "conv_fwd:" + L.name + ".up|", 2.0 * M / 8 * (p27 ? 27 : 64) * L.Cu * L.Cout,
                 4.0 * (M / 8 * L.Cu + M * L.Cout + (p27 ? 27.0 : 64.0) * L.Cu * L.Cout));
    if (L.w_up3n)
      ICS_TRY(launch_conv_fwd_up3n(n.st, gp, lo, L.w_up3n, bias, L.s, L.Cout, L.pre_act, stats ? n.ws_stat : nullptr,
                                   &par_blocks));
    else if (L.w_up3)
      ICS_TRY(launch_conv_fwd_up3(n.st, gp, lo, L.w_up3, only_up ? bias : nullptr, L.s, L.Cout,
                                  only_up ? L.pre_act : ACT_NONE, (only_up && stats) ? n.ws_stat : nullptr, &par_blocks, 0));
    else
      ICS_TRY(launch_conv_fwd_par(n.st, gp, lo, L.wp_par, L.s, L.Cout, only_up ? bias : nullptr,
                                  only_up ? L.pre_act : ACT_NONE, (only_up && stats) ? n.ws_stat : nullptr, &par_blocks));
    n.prof.end(n.st);
    if (!only_up) {
      n.prof.begin(n.st, "conv_fwd:" + L.name + ".skip|", 2.0 * M * 27 * L.Cs * L.Cout,
                   4.0 * (M * L.Cs + 2 * M * L.Cout + 27.0 * L.Cs * L.Cout));
      if (L.ww && conv_wino_ok(gs, L.src, 1))
        ICS_TRY(launch_conv_fwd_wino(n.st, gs, L.src[0], L.ww, bias, L.s, L.Cout, L.pre_act,
                                     stats ? n.ws_stat : nullptr, &rpb, 1, L.ww_layout));
      else
      ICS_TRY(launch_conv_fwd(n.st, gs, L.src, 1, L.wp_skip, bias, L.s, L.Cout, L.pre_act, stats ? n.ws_stat : nullptr,
                              &rpb, 1));
      n.prof.end(n.st);
      par_blocks = 0;
    }
  } else {
  n.prof.begin(n.st, "conv_fwd:" + L.name + "|", 2.0 * M * L.taps * L.Cin * L.Cout,
               4.0 * (M * L.Cin + M * L.Cout + (double)L.taps * L.Cin * L.Cout));
  if (L.cond_fold) {
    // x (one channel) through the direct stencil; the spatially constant condition channels enter as a bias that
    // depends only on the sample and on the voxel's border class (see launch_cond_bias_table)
    const ConvSrc& cs = L.vsrc[1];
    ICS_TRY(launch_cond_bias_table(n.st, n.tp(L.t_w), bias, cs.p, L.vsrc[0].C, cs.bcast, L.Cin, L.Cout, B, L.cond_T));
    ICS_TRY(launch_conv_fwd_thin_c(n.st, g, L.vsrc[0], L.vsrc[0].C, L.CinG, L.wp, nullptr, L.s, L.Cout, L.pre_act,
                                   stats ? n.ws_stat : nullptr, &rpb, L.cond_T));
  } else if (L.pad_in && L.nvsrc == 1 && conv_thin_c_ok(g, L.vsrc[0], 1, L.Cin) && L.Cout % 4 == 0) {
    // single-channel input (c1 at C = 1): the direct stencil reads the un-padded tensor; the packed weights keep the
    // padded layout (CinG channels per tap) the backward-weight kernel is built for
    ICS_TRY(launch_conv_fwd_thin_c(n.st, g, L.vsrc[0], L.Cin, L.CinG, L.wp, bias, L.s, L.Cout, L.pre_act,
                                   stats ? n.ws_stat : nullptr, &rpb));
  } else if (L.ww && conv_wino_ok(g, L.src, L.nsrc)) {
    ICS_TRY(launch_conv_fwd_wino(n.st, g, L.src[0], L.ww, bias, L.s, L.Cout, L.pre_act, stats ? n.ws_stat : nullptr,
                                 &rpb, 0, L.ww_layout));
  } else if (L.wg && conv_winog_ok(g, L.src, L.nsrc)) {
    // the transposed transform is only written when a backward-weight pass will read it (not in the perceptual passes)
    const bool keep = n.want_wgrad_inputs && L.wg_w && conv_winog_wgrad_ok(g, L.src, L.nsrc);
    ICS_TRY(launch_conv_fwd_winog(n.st, g, L.src[0], L.wg, bias, L.s, L.Cout, L.pre_act, stats ? n.ws_stat : nullptr, &rpb,
                                  L.wg_v, L.wg_m, keep ? L.wg_vt : nullptr));
    L.wg_vt_batch = keep ? B : 0;
  } else {
    ICS_TRY(launch_conv_fwd(n.st, g, L.src, L.nsrc, L.wp, bias, L.s, L.Cout, L.pre_act,
                            stats ? n.ws_stat : nullptr, &rpb, 0, n.fws(), n.ws_fwd_n));
  }
  n.prof.end(n.st);
  }
  if (L.has_bn) {
    BnParams bn{n.tp(L.t_gamma), n.tp(L.t_beta), L.mm, L.mv, L.mean, L.rstd, L.scale, L.shift};
    if (training && n.sync() == nullptr) bn.xs = L.xs;
    if (training) {
      const int nblk = par_blocks ? par_blocks : (int)((M + rpb - 1) / rpb);
      ICS_TRY(launch_bn_finalize(n.st, n.ws_stat, nblk, L.Npad, bn, L.Cout, update_moving ? 1 : 0, n.bn_unbias, n.sync()));
    } else {
      ICS_TRY(launch_bn_eval_prepare(n.st, bn, L.Cout));
    }
  }
  if (L.pooled) {
    n.prof.begin(n.st, "pool_fwd", 0, 4.0 * M * L.Cout * 1.125);
    const bool ties = training && L.tie_mask != nullptr && (n.want_wgrad_inputs || n.want_tie_stats);   // a backward pass follows
    L.tie_batch = ties ? B : 0;
    ICS_TRY(launch_pool_fwd(n.st, L.s, L.has_bn ? L.scale : nullptr, L.shift, L.has_bn ? L.post_act : ACT_NONE,
                            B, L.S, L.Cout, L.pooled, L.pool_idx, ties ? L.tie_mask : nullptr, ties ? L.tie_ssum : nullptr,
                            n.pool_ties_all));
    n.prof.end(n.st);
  }
  return 0;
}

// dw_phys[(tap*CinG + ci)*N + n] -> dw[(tap*Cin + ci)*N + n]
__global__ void unpad_dw_kernel(const float* __restrict__ src, int Cin, int CinG, int N, size_t total,
                                float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t row = i / N, n = i - row * N;
  const size_t tap = row / Cin, ci = row - tap * Cin;
  dst[i] = src[(tap * CinG + ci) * N + n];
}
// y[m*C + c] += x[m*ldx + c]
__global__ void axpy_strided_kernel(float* __restrict__ y, const float* __restrict__ x, size_t rows, int C, int ldx) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  const size_t m = i / C, c = i - m * C;
  y[i] += x[m * ldx + c];
}
// route a padded layer through its materialised input buffer
static void use_padded_input(ConvLayer& L) {
  if (L.CinG == L.Cin) return;
  L.vsrc[0] = L.src[0]; L.vsrc[1] = L.src[1]; L.nvsrc = L.nsrc;
  L.src[0] = ConvSrc{L.pad_in, nullptr, nullptr, L.CinG, 0, ACT_NONE, 0};
  L.nsrc = 1;
}

static GradSrc gs_none() { return GradSrc{nullptr, 0, 0, GS_NONE, nullptr, nullptr}; }
static GradSrc gs_direct(const float* p, int ld, int off) { return GradSrc{p, ld, off, GS_DIRECT, nullptr, nullptr}; }
static GradSrc gs_up(const float* p, int ld, int off) { return GradSrc{p, ld, off, GS_UP, nullptr, nullptr}; }
static GradSrc gs_pool(const float* p, int ld, const ConvLayer& producer) {
  return GradSrc{p, ld, 0, GS_POOL, producer.pooled, producer.pool_idx};
}

// side stream: everything issued on st so far is visible to work issued on the returned stream
static hipStream_t side_begin(Net& n) {
  // (one stream while every launch is timed: an event bracket must not time the other stream's kernels)
  if (!n.side_on || (n.prof.on && n.prof.filter.empty())) return n.st;
  (void)hipEventRecord(n.ev_fork, n.st);
  (void)hipStreamWaitEvent(n.st2, n.ev_fork, 0);
  n.side_dirty = true;
  return n.st2;
}
// st waits for everything issued on the side stream (before gradients are consumed / buffers are reused)
static int side_join(Net& n) {
  if (!n.side_dirty) return 0;
  ICS_HIP(hipEventRecord(n.ev_join, n.st2));
  ICS_HIP(hipStreamWaitEvent(n.st, n.ev_join, 0));
  n.side_dirty = false;
  return 0;
}

// BwdStat for the backward-data launch whose output is dO of `next` (its only consumer); empty when not applicable
static BwdStat bwd_stat_for(Net& n, ConvLayer* next, int B) {
  BwdStat bs;
  if (next == nullptr || !next->has_bn || (n.flags & CF_NO_BWD_FOLD)) return bs;
  if ((size_t)2 * next->Npad * ((n.rows(*next, B) + 63) / 64) > n.ws_bwd2_n) return bs;
  bs.s = next->s; bs.mean = next->mean; bs.rstd = next->rstd; bs.scale = next->scale; bs.shift = next->shift;
  bs.partial = n.ws_bwd2; bs.post_act = next->post_act; bs.ld = next->Cout;
  return bs;
}
static void bwd_stat_done(ConvLayer* next, const BwdStat& bs, int blocks, int npad) {
  if (next == nullptr) return;
  next->bwd_pre_nblk = bs.partial ? blocks : 0;
  next->bwd_pre_ld = npad;
}

// Up-split backward of a [skip | upsampled] concat conv: with dyS = dy pooled per tap onto the low-res
// grid (pool27), the up channels need only  dW_up = xl^T dyS  and  dxl = dyS W_up  on M/8 rows; the skip
// channels run the ordinary kernels on a Cs-channel problem.  Exact (a reassociation of the same sums).
// Round 4, second form of the BatchNorm-backward fusion (see dgrad_bnfuse_ok below): the producer P of L's skip channels has a
// second consumer behind a MaxPool3D (c2: c17's skip channels and c3).  L's share of P's two sums comes from L's skip
// weight-gradient GEMM (on xhat) right here; L's skip backward-data launch waits until the pool consumer's gradient
// exists (conv_backward of P), takes the constants, adds the routed pool gradient and writes P's dy: P's reduce AND
// apply passes disappear (c2: 0.13 + 0.16 ms).
static bool skip_bnfuse_ok(const Net& n, const ConvLayer& L, int B) {
  const ConvLayer* P = L.skip_prod;
  if (P == nullptr || !L.split_up || L.Cs == 0 || (n.flags & CF_NO_DGRAD_BNFUSE) || n.sync() != nullptr || n.side_on) return false;
  if (!P->has_bn || P->pre_act != ACT_RELU || P->post_act != ACT_NONE || P->tie_mask == nullptr || P->xs == nullptr ||
      P->db_partial == nullptr || !P->pooled || P->Cout != L.Cs || L.src[0].p != P->s || L.src[0].scale == nullptr ||
      L.src[0].act != ACT_NONE || !n.want_wgrad_inputs || P->tie_batch != B)
    return false;
  if (!L.wino_w || !L.wwb || L.wwb_layout != 1) return false;
  if (n.rows(L, B) * (size_t)P->Cout * sizeof(float) < 2 * n.bnfuse_min_bytes) return false;
  const ConvGeom gw = geom_skip_wgrad(L, B), gd = geom_skip_dgrad(L, B);
  const ConvSrc sdy = src_plain(L.dy, L.Cout);
  return conv_bnfuse_ok(L.S, L.Cs, L.Cout) && conv_wino_wgrad_ok(gw, L.src, 1) && conv_wino_ok(gd, &sdy, 1) &&
         conv_bnfuse_partial_floats(B, L.S, L.Cout) <= n.ws_cls_n && L.Cs % 4 == 0 &&
         pool_bnfuse_partial_doubles(n.rows(L, B) / 8, P->Cout) <= n.ws_pool_n;
}
static int conv_grads_split_up(Net& n, ConvLayer& L, int B, ConvLayer* next) {
  const size_t M = n.rows(L, B);
  const double fl_skip = 2.0 * M * 27 * L.Cs * L.Cout, fl_up = 2.0 * (M / 8) * 27 * L.Cu * L.Cout;
  const bool skipfuse = skip_bnfuse_ok(n, L, B);
  ConvLayer* P = L.skip_prod;
  n.prof.begin(n.st, "pool27:" + L.name, 0, 4.0 * M * L.Cout * (1 + 27.0 / 8));
  ICS_TRY(launch_pool27(n.st, L.dy, B, L.S, L.Cout, L.dyS, L.ldS));
  n.prof.end(n.st);
  hipStream_t ws = side_begin(n);   // weight gradients: off the critical path (see Net::st2)
  if (L.Cs) {
    const ConvGeom g = geom_skip_wgrad(L, B);
    n.prof.begin(ws, "conv_wgrad:" + L.name + ".skip|", fl_skip,
                 4.0 * (M * L.Cs + M * L.Cout + 27.0 * L.Cs * L.Cout));
    const bool wino = L.wino_w && conv_wino_wgrad_ok(g, L.src, 1);
    ConvSrc sx = L.src[0];
    if (skipfuse) { sx.scale = P->xs; sx.shift = P->xs + P->Cout; }   // the GEMM on xhat (conv_bnfuse_kernel)
    if (wino)
      ICS_TRY(launch_conv_wgrad_wino(ws, g, sx, L.dy, L.Cout, n.tg(L.t_w), L.Cout, n.ws_wgrad, n.ws_wgrad_n,
                                     L.Cs, L.Cin, 0, 1));
    else
    ICS_TRY(launch_conv_wgrad(ws, g, L.src, 1, L.dy, L.Cout, n.tg(L.t_w), L.Cout, n.ws_wgrad, n.ws_wgrad_n, L.Cs,
                              L.Cin, 0, 1));
    n.prof.end(ws);
    n.prof.begin(ws, "wgrad_reduce_splits", 0, 0);
    if (wino)
      ICS_TRY(launch_conv_wgrad_wino(ws, g, sx, L.dy, L.Cout, n.tg(L.t_w), L.Cout, n.ws_wgrad, n.ws_wgrad_n,
                                     L.Cs, L.Cin, 0, 2));
    else
    ICS_TRY(launch_conv_wgrad(ws, g, L.src, 1, L.dy, L.Cout, n.tg(L.t_w), L.Cout, n.ws_wgrad, n.ws_wgrad_n, L.Cs,
                              L.Cin, 0, 2));
    n.prof.end(ws);
    if (skipfuse) {     // the skip rows of dW become final here (the gradient buckets may leave before P's backward)
      const bool pend = L.db_blocks > 0;
      n.prof.begin(ws, "bnfuse:" + L.name + ".skip", 0, 0);
      ICS_TRY(launch_conv_bnfuse(ws, L.dy, B, L.S, L.Cs, L.Cin, L.Cout, pend ? L.db_partial : n.tg(L.t_b), pend ? L.db_blocks : 1,
                                 n.tp(L.t_w), n.tg(L.t_w), n.tp(P->t_gamma), n.tp(P->t_beta), P->mean, P->rstd, P->scale,
                                 nullptr, nullptr, nullptr, nullptr, n.ws_cls, n.ws_cls_n, n.ws_R, P->bn_sums));
      n.prof.end(ws);
      P->deferred_skip = &L;
    }
  }
  {
    const ConvGeom g = geom_up_wgrad(L, B);
    const ConvSrc lo = src_lowres(L);
    n.prof.begin(ws, "conv_wgrad:" + L.name + ".up|", fl_up,
                 4.0 * (M / 8 * L.Cu + M / 8 * 27.0 * L.Cout + 27.0 * L.Cu * L.Cout));
    ICS_TRY(launch_conv_wgrad(ws, g, &lo, 1, L.dyS, L.ldS, L.dw_up, 27 * L.Cout, n.ws_wgrad, n.ws_wgrad_n, 0, 0, 0, 1));
    n.prof.end(ws);
    n.prof.begin(ws, "wgrad_reduce_splits", 0, 0);
    ICS_TRY(launch_conv_wgrad(ws, g, &lo, 1, L.dyS, L.ldS, L.dw_up, 27 * L.Cout, n.ws_wgrad, n.ws_wgrad_n, 0, 0, 0, 2));
    ICS_TRY(launch_permute_up_dw(ws, L.dw_up, L.Cu, L.Cout, L.Cin, L.Cs, n.tg(L.t_w)));
    n.prof.end(ws);
  }
  if (L.Cs && !skipfuse) {
    const ConvGeom g = geom_skip_dgrad(L, B);
    ConvSrc sdy = src_plain(L.dy, L.Cout);
    n.prof.begin(n.st, "conv_dgrad:" + L.name + ".skip|", fl_skip,
                 4.0 * (M * L.Cs + M * L.Cout + 27.0 * L.Cs * L.Cout));
    if (L.wwb && conv_wino_ok(g, &sdy, 1))
      ICS_TRY(launch_conv_fwd_wino(n.st, g, sdy, L.wwb, nullptr, L.dA_skip, L.Cs, ACT_NONE, nullptr, nullptr, 0,
                                   L.wwb_layout));
    else
    ICS_TRY(launch_conv_fwd(n.st, g, &sdy, 1, L.wf_skip, nullptr, L.dA_skip, L.Cs, ACT_NONE, nullptr, nullptr, 0,
                            n.fws(), n.ws_fwd_n));
    n.prof.end(n.st);
  }
  {
    const ConvGeom g = geom_up_dgrad(L, B);
    ConvSrc sd = src_plain(L.dyS, L.ldS);
    n.prof.begin(n.st, "conv_dgrad:" + L.name + ".up|", fl_up,
                 4.0 * (M / 8 * L.Cu + M / 8 * 27.0 * L.Cout + 27.0 * L.Cu * L.Cout));
    const BwdStat bs = bwd_stat_for(n, next, B);    // next = the producer of the upsampled channels
    int blocks = 0;
    ICS_TRY(launch_conv_fwd(n.st, g, &sd, 1, L.w_up, nullptr, L.dxl, L.Cu, ACT_NONE, nullptr, nullptr, 0, n.fws(),
                            n.ws_fwd_n, &bs, &blocks));
    bwd_stat_done(next, bs, blocks, g.Npad);
    n.prof.end(n.st);
  }
  return 0;
}

// weight / input gradients given L.dy
static int colsum_push(Net& n, const float* partial, int nblk, int C, float* out);
// Round 4: `next`'s BatchNorm backward inside L's backward-data launch (elementwise.hip conv_bnfuse_kernel, conv_wino64.hip
// FOLD = 2).  For the Conv -> ReLU -> BN blocks of unet.py:276-336 whose gradient comes from ONE 3x3x3 consumer served by
// the Winograd kernels, from the size on where the removed pass (3 x the activation bytes) outweighs the four small
// launches the constants cost: c18 -> c17 and c16 -> c15 at B = 32.  Not with SyncBN (the two sums would need their own
// all-reduce) and not with the weight gradients on the side stream (the backward-data launch waits for them here).
static bool dgrad_bnfuse_ok(const Net& n, const ConvLayer& L, const ConvLayer* next, int B, bool need_dA, bool param_grads) {
  if (next == nullptr || !need_dA || !param_grads || (n.flags & CF_NO_DGRAD_BNFUSE) || n.sync() != nullptr || n.side_on)
    return false;
  if (!next->has_bn || next->pre_act != ACT_RELU || next->post_act != ACT_NONE || next->xs == nullptr ||
      next->db_partial == nullptr || next->dy == nullptr)
    return false;
  if (L.split_up || L.nsrc != 1 || L.taps != 27 || !L.wino_w || !L.wwb || L.CinG != L.Cin ||
      L.Cin != next->Cout || L.src[0].p != next->s || L.src[0].scale == nullptr || L.src[0].act != ACT_NONE || L.dw_phys)
    return false;
  if (n.rows(L, B) * (size_t)next->Cout * sizeof(float) < n.bnfuse_min_bytes) return false;
  const ConvGeom g = geom_fwd(L, B);
  ConvGeom gb = geom_bwd(L, B);
  gb.Cout = L.Cin;
  const ConvSrc sdy = src_plain(L.dy, L.Cout);
  return conv_bnfuse_ok(L.S, L.Cin, L.Cout) && conv_wino_wgrad_ok(g, L.src, 1) && conv_wino_ok(gb, &sdy, 1) &&
         conv_bnfuse_partial_floats(B, L.S, L.Cout) <= n.ws_cls_n && L.Cin % 4 == 0;
}
static int conv_grads_from_dy(Net& n, ConvLayer& L, int B, bool need_dA, bool param_grads, ConvLayer* next) {
  if (L.split_up && need_dA && param_grads) return conv_grads_split_up(n, L, B, next);
  if (L.split_up) next = nullptr;   // direct path of an up-split layer: dA covers [skip | up] channels
  const ConvGeom g = geom_fwd(L, B);
  const size_t M = n.rows(L, B);
  const bool bnfuse = dgrad_bnfuse_ok(n, L, next, B, need_dA, param_grads);
  if (param_grads) {
    hipStream_t ws = side_begin(n);   // weight gradients: off the critical path (Net::st2); ALL of them, the
                                      // split-K workspace ws_wgrad is only ever touched from that stream
                                      // (bnfuse implies the side stream is off: ws == n.st)
    n.prof.begin(ws, "conv_wgrad:" + L.name + "|", 2.0 * M * L.taps * L.Cin * L.Cout,
                 4.0 * (M * L.Cin + M * L.Cout + (double)L.taps * L.Cin * L.Cout));
    float* dw = L.dw_phys ? L.dw_phys : n.tg(L.t_w);
    // single-channel input (c1 at C = 1): direct stencil reduction on the un-padded tensor, straight into G
    const bool thin1 = L.pad_in && L.nvsrc == 1 && L.Cin == 1 && conv_thin_c_ok(g, L.vsrc[0], 1, 1) && L.Cout % 4 == 0;
    if (L.cond_fold) {
      // rows of the x channel: stencil reduction; rows of the condition channels: region sums of dy x cond
      const ConvSrc& cs = L.vsrc[1];
      const int C0 = L.vsrc[0].C;
      ICS_TRY(launch_conv_wgrad_thin_c(ws, g, L.vsrc[0], L.dy, L.Cout, n.tg(L.t_w), L.Cout, L.Cin, n.ws_wgrad,
                                       n.ws_wgrad_n, 0));
      ICS_TRY(launch_cond_wgrad(ws, L.dy, B, L.S, L.Cout, cs.p, C0, cs.C, cs.bcast, L.Cin, n.tg(L.t_w), n.ws_dbl,
                                n.ws_dbl_n));
      n.prof.end(ws);
    } else if (thin1) {
      ICS_TRY(launch_conv_wgrad_thin_c(ws, g, L.vsrc[0], L.dy, L.Cout, n.tg(L.t_w), L.Cout, L.Cin, n.ws_wgrad,
                                       n.ws_wgrad_n, 1));
      n.prof.end(ws);
      n.prof.begin(ws, "wgrad_reduce_splits", 0, 0);
      ICS_TRY(launch_conv_wgrad_thin_c(ws, g, L.vsrc[0], L.dy, L.Cout, n.tg(L.t_w), L.Cout, L.Cin, n.ws_wgrad,
                                       n.ws_wgrad_n, 2));
      n.prof.end(ws);
    } else if (L.wg_w && L.wg_vt_batch == B && conv_winog_wgrad_ok(g, L.src, L.nsrc)) {
      ICS_TRY(launch_conv_wgrad_winog(ws, g, L.wg_vt, L.dy, L.Cout, dw, L.Cout, 0, 0, L.wg_z, L.wg_du));
      n.prof.end(ws);
    } else if (L.wino_w && !L.split_up && conv_wino_wgrad_ok(g, L.src, L.nsrc)) {
      ConvSrc sx = L.src[0];
      if (bnfuse) {     // the GEMM on xhat: conv_bnfuse_kernel turns its result into dW and next's BatchNorm-backward constants
        sx.scale = next->xs; sx.shift = next->xs + next->Cout;
      }
      ICS_TRY(launch_conv_wgrad_wino(ws, g, sx, L.dy, L.Cout, dw, L.Cout, n.ws_wgrad, n.ws_wgrad_n, 0, 0, 0, 1));
      n.prof.end(ws);
      n.prof.begin(ws, "wgrad_reduce_splits", 0, 0);
      ICS_TRY(launch_conv_wgrad_wino(ws, g, sx, L.dy, L.Cout, dw, L.Cout, n.ws_wgrad, n.ws_wgrad_n, 0, 0, 0, 2));
      n.prof.end(ws);
      if (bnfuse) {
        // L's total column sums of dy: its bias-gradient partials if their finalize is still pending, else the gradient
        const bool pend = L.db_blocks > 0;
        n.prof.begin(ws, "bnfuse:" + L.name, 0, 0);
        ICS_TRY(launch_conv_bnfuse(ws, L.dy, B, L.S, L.Cin, L.Cin, L.Cout, pend ? L.db_partial : n.tg(L.t_b), pend ? L.db_blocks : 1,
                                   n.tp(L.t_w), dw, n.tp(next->t_gamma), n.tp(next->t_beta), next->mean, next->rstd,
                                   next->scale, next->abc, next->c1c2, n.tg(next->t_gamma), n.tg(next->t_beta), n.ws_cls,
                                   n.ws_cls_n, n.ws_R));
        n.prof.end(ws);
      }
    } else {
    ICS_TRY(launch_conv_wgrad(ws, g, L.src, L.nsrc, L.dy, L.Cout, dw, L.Cout, n.ws_wgrad, n.ws_wgrad_n, 0, 0, 0, 1));
    n.prof.end(ws);
    n.prof.begin(ws, "wgrad_reduce_splits", 0, 0);
    ICS_TRY(launch_conv_wgrad(ws, g, L.src, L.nsrc, L.dy, L.Cout, dw, L.Cout, n.ws_wgrad, n.ws_wgrad_n, 0, 0, 0, 2));
    if (L.dw_phys) {
      const size_t cnt = (size_t)L.taps * L.Cin * L.Cout;
      ICS_LAUNCH(unpad_dw_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ws, L.dw_phys, L.Cin,
                         L.CinG, L.Cout, cnt, n.tg(L.t_w));
      ICS_HIP(hipGetLastError());
    }
    n.prof.end(ws);
    }
  }
  if (need_dA) {
    ConvGeom gb = geom_bwd(L, B);
    gb.Cout = L.Cin;   // a padded input (CinG > Cin) only needs its logical gradient columns
    ConvSrc sdy = src_plain(L.dy, L.Cout);
    n.prof.begin(n.st, "conv_dgrad:" + L.name + "|", 2.0 * M * L.taps * L.Cin * L.Cout,
                 4.0 * (M * L.Cin + M * L.Cout + (double)L.taps * L.Cin * L.Cout));
    int blocks = 0;
    if (bnfuse) {
      BwdStat ba{};
      ba.s = next->s; ba.ld = next->Cout; ba.abc = next->abc; ba.db_partial = next->db_partial;
      ICS_TRY(launch_conv_fwd_wino(n.st, gb, sdy, L.wwb, nullptr, next->dy, next->Cout, ACT_NONE, nullptr, nullptr, 0,
                                   L.wwb_layout, &ba, &blocks));
      n.prof.end(n.st);
      ICS_CHECK(blocks > 0, "fused BatchNorm-backward apply: the backward-data launch did not take it");
      ICS_CHECK((size_t)blocks <= next->db_rows, "bias-gradient partial buffer too small for this batch");
      next->dy_ready = true;
      next->db_blocks = 0;
      ICS_TRY(colsum_push(n, next->db_partial, blocks, next->Cout, n.tg(next->t_b)));
      next->db_blocks = blocks;
      return 0;
    }
    const BwdStat bs = bwd_stat_for(n, next, B);
    if (L.wwb && !L.split_up && conv_wino_ok(gb, &sdy, 1))
      ICS_TRY(launch_conv_fwd_wino(n.st, gb, sdy, L.wwb, nullptr, L.dA, L.CinG, ACT_NONE, nullptr, nullptr, 0,
                                   L.wwb_layout, &bs, &blocks));
    else if (L.wgb && conv_winog_ok(gb, &sdy, 1))       // no folded BatchNorm-backward sums here: blocks stays 0
      ICS_TRY(launch_conv_fwd_winog(n.st, gb, sdy, L.wgb, nullptr, L.dA, L.CinG, ACT_NONE, nullptr, nullptr, L.wg_v, L.wg_m,
                                    nullptr));
    else {
      if (L.split_up && L.wf_stale) {              // see pack_layer
        ICS_TRY(launch_pack_bwd(n.st, n.tp(L.t_w), L.taps, L.Cin, L.Cout, L.wf, L.Kpad_b, L.Npad_b, L.Cout, 0, 1));
        L.wf_stale = false;
      }
      ICS_TRY(launch_conv_fwd(n.st, gb, &sdy, 1, L.wf, nullptr, L.dA, L.CinG, ACT_NONE, nullptr, nullptr, 0, n.fws(),
                              n.ws_fwd_n, &bs, &blocks));
    }
    bwd_stat_done(next, bs, blocks, gb.Npad);
    n.prof.end(n.st);
  }
  return 0;
}

static int colsum_push(Net& n, const float* partial, int nblk, int C, float* out) {
  ColsumJobs& J = n.colsum;
  if (J.n == 24) { ICS_TRY(launch_colsum_batch(n.st, J)); J.n = 0; }
  if (J.n == 0) J.blk0[0] = 0;
  J.partial[J.n] = partial; J.out[J.n] = out; J.nblk[J.n] = nblk; J.C[J.n] = C;
  J.blk0[J.n + 1] = J.blk0[J.n] + colsum_blocks(C);
  J.n += 1;
  return 0;
}
static int colsum_flush(Net& n) {
  ICS_TRY(launch_colsum_batch(n.st, n.colsum));
  n.colsum.n = 0;
  return 0;
}
static int conv_backward(Net& n, ConvLayer& L, int B, GradSrc g0, GradSrc g1, const float* dtap, bool need_dA,
                         bool param_grads, int tap = -1, ConvLayer* next = nullptr) {
  // next: the layer whose ONLY gradient source is this layer's backward-data output (same resolution, no pooling /
  // second consumer in between): its BatchNorm-backward sums are folded into that launch
  LayerBwd lb{};
  lb.s = L.s; lb.scale = L.scale; lb.shift = L.shift; lb.mean = L.mean; lb.rstd = L.rstd;
  lb.dtap = dtap; lb.g0 = g0; lb.g1 = g1;
  if (tap >= 0) { lb.tap_ref = n.tap_ref[tap]; lb.tap_coef = n.tap_coef[tap]; lb.tap_partial = n.tap_partial[tap]; }
  lb.B = B; lb.S = L.S; lb.lgS = ilog2(L.S); lb.C = L.Cout;
  lb.has_bn = L.has_bn; lb.pre_act = L.pre_act; lb.post_act = L.post_act;
  lb.pool_ties_all = n.pool_ties_all;
  lb.flags = n.flags;
  const size_t M = n.rows(L, B);
  if (L.dy_ready) {                  // written by the consumer's backward-data launch (conv_grads_from_dy, bnfuse)
    L.dy_ready = false;
    return conv_grads_from_dy(n, L, B, need_dA, param_grads, next);
  }
  if (L.deferred_skip != nullptr) {  // skip_bnfuse_ok: g0 is the pool consumer's gradient, g1 the pending skip launch
    ConvLayer& K = *L.deferred_skip;
    L.deferred_skip = nullptr;
    ICS_CHECK(g0.kind == GS_POOL && g0.off == 0 && param_grads && tap < 0 && dtap == nullptr,
              "deferred skip backward-data: the producer's other gradient source must be its MaxPool3D");
    n.prof.begin(n.st, "bnfuse:" + L.name + ".pool", 0, 0);
    ICS_TRY(launch_pool_bnfuse(n.st, g0.p, g0.ld, L.tie_mask, L.tie_ssum, M / 8, L.Cout, (double)M, L.bn_sums, L.mean, L.rstd,
                               L.scale, L.abc, L.c1c2, n.tg(L.t_gamma), n.tg(L.t_beta), n.ws_pool, n.ws_pool_n));
    n.prof.end(n.st);
    const ConvGeom g = geom_skip_dgrad(K, B);
    ConvSrc sdy = src_plain(K.dy, K.Cout);
    n.prof.begin(n.st, "conv_dgrad:" + K.name + ".skip|", 2.0 * M * 27 * K.Cs * K.Cout,
                 4.0 * (M * K.Cs + M * K.Cout + 27.0 * K.Cs * K.Cout));
    BwdStat ba{};
    ba.s = L.s; ba.ld = L.Cout; ba.abc = L.abc; ba.db_partial = L.db_partial;
    ba.pool_d = g0.p; ba.pool_ld = g0.ld; ba.pool_mask = L.tie_mask;
    int blocks = 0;
    ICS_TRY(launch_conv_fwd_wino(n.st, g, sdy, K.wwb, nullptr, L.dy, L.Cout, ACT_NONE, nullptr, nullptr, 0, K.wwb_layout, &ba,
                                 &blocks));
    n.prof.end(n.st);
    ICS_CHECK(blocks > 0, "deferred skip backward-data: the launch did not take the fused apply");
    ICS_CHECK((size_t)blocks <= L.db_rows, "bias-gradient partial buffer too small for this batch");
    L.db_blocks = 0;
    ICS_TRY(colsum_push(n, L.db_partial, blocks, L.Cout, n.tg(L.t_b)));
    L.db_blocks = blocks;
    return conv_grads_from_dy(n, L, B, need_dA, param_grads, next);
  }
  BwdPre pre{n.ws_bwd2, L.bwd_pre_nblk, L.bwd_pre_ld};
  L.bwd_pre_nblk = 0;
  // the max-pool is the ONLY gradient source (plus, for a perceptual tap layer, a term that enters behind the BatchNorm):
  // the two BatchNorm-backward sums from the pooled grid (pool_sums_kernel) instead of a reduce pass over the fine grid
  if (pre.nblk == 0 && L.has_bn && L.post_act == ACT_NONE && g0.kind == GS_POOL && g0.off == 0 && g1.kind == GS_NONE &&
      dtap == nullptr && L.tie_mask != nullptr && L.tie_batch == B && !(n.flags & CF_NO_POOL_PRESUM) &&
      pool_presum_ok(L.Cout, g0.ld) && (size_t)2 * L.Cout * pool_presum_blocks(M / 8) <= n.ws_bwd2_n) {
    int pb = 0;
    n.prof.begin(n.st, "pool_presum:" + L.name, 0, 0);
    ICS_TRY(launch_pool_presum(n.st, g0.p, g0.ld, L.tie_mask, L.tie_ssum, M / 8, L.Cout, L.mean, L.rstd, n.ws_bwd2, n.ws_bwd2_n,
                               &pb));
    n.prof.end(n.st);
    pre.nblk = pb; pre.ld = L.Cout;
  }
  n.prof.begin(n.st, "bn_act_bwd:" + L.name, 0, 4.0 * M * L.Cout * (L.has_bn ? (pre.nblk ? 3.0 : 5.0) : 3.0));
  const bool defer = param_grads && L.db_partial != nullptr;
  int db_blocks = 0;
  {
    int rpb;
    ICS_CHECK(!defer || (size_t)bn_bwd_num_blocks(lb, &rpb) <= L.db_rows, "bias-gradient partial buffer too small for this batch");
    ICS_CHECK(!L.has_bn || layer_bwd_workspace_floats(lb) <= n.ws_bwd_n, "BatchNorm-backward workspace too small for this batch");
  }
  ICS_TRY(launch_layer_bwd(n.st, lb, L.dy, n.ws_bwd, L.c1c2,
                           (param_grads && L.has_bn) ? n.tg(L.t_gamma) : nullptr,
                           (param_grads && L.has_bn) ? n.tg(L.t_beta) : nullptr,
                           param_grads ? n.tg(L.t_b) : nullptr, n.sync(), &pre, defer ? L.db_partial : nullptr, &db_blocks));
  L.db_blocks = 0;
  if (defer) { ICS_TRY(colsum_push(n, L.db_partial, db_blocks, L.Cout, n.tg(L.t_b))); L.db_blocks = db_blocks; }
  n.prof.end(n.st);
  return conv_grads_from_dy(n, L, B, need_dA, param_grads, next);
}

// ---- data parallel gradient exchange.  The flat gradient buffer is laid out in layer order and the
// backward pass walks the layers last-to-first, so "all gradients at offsets >= lo are final" holds after
// each layer; a bucket [lo, bucket_hi) is handed to RCCL on comm_st (behind an event on st) as soon as it
// holds >= bucket_min floats, and overlaps with the rest of the backward pass.  Adam waits for comm_st.
static int grads_begin(Net& n) {
  n.bucket_hi = n.nparams;
  n.buckets_issued = 0;
  if (!n.comm || n.nranks < 1) return 0;
  if (!n.sync_bn && n.bn_slab_n) {
    // local BN: average the moving statistics the forward pass just updated (every rank then checkpoints
    // the same values; the update is linear, so this is the moving average of the rank-mean statistics)
    ICS_HIP(hipEventRecord(n.ev_grad, n.st));
    ICS_HIP(hipStreamWaitEvent(n.comm_st, n.ev_grad, 0));
    n.prof.begin(n.comm_st, "rccl_allreduce_bn_moving", 0, 4.0 * n.bn_slab_n);
    ncclResult_t r = ncclAllReduce(n.bn_slab, n.bn_slab, n.bn_slab_n, ncclFloat, ncclAvg, n.comm, n.comm_st);
    n.prof.end(n.comm_st);
    ICS_CHECK(r == ncclSuccess, std::string("ncclAllReduce(bn moving): ") + ncclGetErrorString(r));
  }
  return 0;
}
static int grads_ready(Net& n, size_t lo) {
  if (!n.comm || !n.overlap) return 0;
  if (lo >= n.bucket_hi) return 0;
  if (lo != 0 && n.bucket_hi - lo < n.bucket_min) return 0;
  ICS_TRY(colsum_flush(n));   // the bucket's bias gradients
  ICS_TRY(side_join(n));     // the bucket's weight gradients were computed on the side stream
  ICS_HIP(hipEventRecord(n.ev_grad, n.st));
  ICS_HIP(hipStreamWaitEvent(n.comm_st, n.ev_grad, 0));
  n.prof.begin(n.comm_st, "rccl_allreduce_grads", 0, 4.0 * (n.bucket_hi - lo));
  ncclResult_t r = ncclAllReduce(n.G + lo, n.G + lo, n.bucket_hi - lo, ncclFloat, ncclSum, n.comm, n.comm_st);
  n.prof.end(n.comm_st);
  ICS_CHECK(r == ncclSuccess, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
  n.bucket_hi = lo;
  n.buckets_issued += 1;
  return 0;
}
static size_t layer_lo(const Net& n, const ConvLayer& L) { return n.tensors[L.t_w].off; }

static int adam_step(Net& n) {
  float gscale = 1.f;
  ICS_TRY(colsum_flush(n));
  ICS_TRY(side_join(n));
  if (n.comm) {
    if (n.overlap) {
      ICS_TRY(grads_ready(n, 0));
    } else {
      n.prof.begin(n.st, "rccl_allreduce_grads", 0, 4.0 * n.nparams);
      ncclResult_t r = ncclAllReduce(n.G, n.G, n.nparams, ncclFloat, ncclSum, n.comm, n.st);
      n.prof.end(n.st);
      ICS_CHECK(r == ncclSuccess, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
      n.buckets_issued = 1;
    }
    ICS_HIP(hipEventRecord(n.ev_comm, n.comm_st));
    ICS_HIP(hipStreamWaitEvent(n.st, n.ev_comm, 0));
    gscale = 1.f / (float)n.nranks;
  }
  n.adam_t += 1;
  const double b1 = 0.9, b2 = 0.999;
  const double lr_t = (double)n.lr * std::sqrt(1.0 - std::pow(b2, n.adam_t)) / (1.0 - std::pow(b1, n.adam_t));
  n.prof.begin(n.st, "adam", 0, 4.0 * 7 * n.nparams);
  ICS_TRY(launch_adam(n.st, n.P, n.G, n.Mo, n.Vo, n.nparams, (float)lr_t, gscale));
  n.prof.end(n.st);
  n.packed_valid = false;
  return 0;
}

// ------------------------------------------------------------------------------------------
// U-Net
// ------------------------------------------------------------------------------------------
struct UnetRefs {
  ConvLayer *c1, *c2, *c3, *c4, *c5, *c6, *c9, *c10, *c13, *c14, *c15, *c16, *c17, *c18;
};
static UnetRefs unet_refs(Net& n) {
  UnetRefs r;
  ConvLayer** f[] = {&r.c1, &r.c2, &r.c3, &r.c4, &r.c5, &r.c6, &r.c9, &r.c10, &r.c13, &r.c14, &r.c15, &r.c16, &r.c17, &r.c18};
  for (int i = 0; i < 14; ++i) *f[i] = n.layers[i].get();
  return r;
}

static int unet_build(Net& n, const ics_unet_config& cfg) {
  n.side_on = false;     // the U-Net's weight gradients stay on the engine's stream (see Net::st2)
  n.kind = 0; n.maxB = cfg.max_batch; n.d = cfg.d; n.C = cfg.in_channels; n.ncls = cfg.num_classes;
  n.lr = cfg.lr; n.loss_weight = cfg.loss_weight > 0 ? cfg.loss_weight : (float)cfg.num_classes;
  n.pool_ties_all = cfg.pool_ties_all; n.bn_unbias = cfg.bn_unbias; n.bce_from_logits = cfg.bce_from_logits ? 1 : 0;
  const int d = cfg.d;
  struct Spec { const char* name; int cin, cout, S; bool pooled; };
  const Spec specs[14] = {
      {"c1", cfg.in_channels, 32, d, false}, {"c2", 32, 64, d, true},
      {"c3", 64, 64, d / 2, false},          {"c4", 64, 128, d / 2, true},
      {"c5", 128, 128, d / 4, false},        {"c6", 128, 256, d / 4, true},
      {"c9", 256, 512, d / 8, false},        {"c10", 512, 512, d / 8, false},
      {"c13", 768, 512, d / 4, false},       {"c14", 512, 256, d / 4, false},
      {"c15", 384, 256, d / 2, false},       {"c16", 256, 128, d / 2, false},
      {"c17", 192, 128, d, false},           {"c18", 128, 128, d, false}};
  for (auto& s : specs)
    add_conv(n, s.name, s.cin, s.cout, 27, s.S, ACT_RELU, 1, ACT_NONE, false, true, /*pad_input=*/&s == &specs[0]);
  // heads: one 1x1x1 GEMM with N = num_classes + 1; params keep the reference's two layers
  n.head = add_conv(n, "head", 128, cfg.num_classes + 1, 1, d, ACT_NONE, 0, ACT_NONE, false, false);
  const int t_sw = n.add_tensor("soft/kernel", {1, 1, 1, 128, cfg.num_classes}, true);
  const int t_gw = n.add_tensor("sig/kernel", {1, 1, 1, 128, 1}, true);
  const int t_sb = n.add_tensor("soft/bias", {cfg.num_classes}, true);   // soft/bias | sig/bias contiguous
  n.add_tensor("sig/bias", {1}, true);
  n.head->t_w = t_sw; n.head->t_b = t_sb; n.head->t_gamma = t_gw;   // t_gamma slot reused: sig kernel index
  ICS_TRY(alloc_params(n));
  const size_t M = (size_t)n.maxB * d * d * d;
  ICS_TRY(n.alloc(&n.x_in, M * n.C));
  ICS_TRY(n.alloc(&n.labels, M));
  for (int i = 0; i < 14; ++i) ICS_TRY(alloc_layer(n, *n.layers[i], true, specs[i].pooled));
  ICS_TRY(alloc_layer(n, *n.head, true, false));
  ICS_TRY(n.alloc(&n.head_bias_grad, (size_t)256));
  ICS_TRY(n.alloc(&n.head_dw_tmp, (size_t)128 * (n.ncls + 1)));
  ICS_TRY(n.alloc(&n.head_xs, (size_t)256));
  UnetRefs r = unet_refs(n);
  r.c1->src[0] = src_plain(n.x_in, n.C);
  r.c2->src[0] = src_layer(*r.c1, 0);
  r.c3->src[0] = src_plain(r.c2->pooled, 64);
  r.c4->src[0] = src_layer(*r.c3, 0);
  r.c5->src[0] = src_plain(r.c4->pooled, 128);
  r.c6->src[0] = src_layer(*r.c5, 0);
  r.c9->src[0] = src_plain(r.c6->pooled, 256);
  r.c10->src[0] = src_layer(*r.c9, 0);
  r.c13->src[0] = src_layer(*r.c6, 0); r.c13->src[1] = src_layer(*r.c10, 1); r.c13->nsrc = 2;   // [skip | up]
  r.c14->src[0] = src_layer(*r.c13, 0);
  r.c15->src[0] = src_layer(*r.c4, 0); r.c15->src[1] = src_layer(*r.c14, 1); r.c15->nsrc = 2;
  r.c16->src[0] = src_layer(*r.c15, 0);
  r.c17->src[0] = src_layer(*r.c2, 0); r.c17->src[1] = src_layer(*r.c16, 1); r.c17->nsrc = 2;
  r.c18->src[0] = src_layer(*r.c17, 0);
  n.head->src[0] = src_layer(*r.c18, 0);
  r.c13->skip_prod = r.c6; r.c15->skip_prod = r.c4; r.c17->skip_prod = r.c2;
  use_padded_input(*r.c1);
  ICS_TRY(enable_split_up(n, *r.c13)); ICS_TRY(enable_split_up(n, *r.c15)); ICS_TRY(enable_split_up(n, *r.c17));
  for (int i = 0; i < 14; ++i) ICS_TRY(enable_wino(n, *n.layers[i], true));
  for (int i = 0; i < 14; ++i) ICS_TRY(enable_winog(n, *n.layers[i], true));
  ICS_TRY(alloc_workspaces(n, true));
  ICS_TRY(init_bn_defaults(n));
  return 0;
}

static int unet_pack_jobs(Net& n) {
  for (int i = 0; i < 14; ++i) ICS_TRY(pack_layer(n, *n.layers[i], true));
  ConvLayer& H = *n.head;
  const int ncls = n.ncls;
  const float* wsoft = n.tp(H.t_w);
  const float* wsig = n.tp(H.t_gamma);
  ICS_TRY(launch_pack_fwd(n.st, wsoft, 128, ncls, H.wp, H.Kpad, H.Npad, 0, 0, 1));
  ICS_TRY(launch_pack_fwd(n.st, wsig, 128, 1, H.wp, H.Kpad, H.Npad, 0, ncls, 0));
  ICS_TRY(launch_pack_bwd(n.st, wsoft, 1, 128, ncls, H.wf, H.Kpad_b, H.Npad_b, ncls + 1, 0, 1));
  ICS_TRY(launch_pack_bwd(n.st, wsig, 1, 128, 1, H.wf, H.Kpad_b, H.Npad_b, ncls + 1, ncls, 0));
  return 0;
}
// The packed weight images are rebuilt after every parameter change with ONE launch: the per-layer pack calls
// are recorded into a job table the first time (pointers and shapes never change for the life of the handle).
static int run_pack_table(Net& n, int (*jobs)(Net&)) {
  if (n.packed_valid) return 0;
  if (n.pack_njobs < 0) {
    void* rec = pack_table_record_begin();
    const int rc = jobs(n);
    std::vector<unsigned char> bytes;
    int nj = 0; unsigned nb = 0;
    pack_table_record_end(rec, &bytes, &nj, &nb);
    ICS_TRY(rc);
    unsigned char* d = nullptr;
    ICS_TRY(n.alloc(&d, bytes.size() + 16));
    ICS_HIP(hipMemcpyAsync(d, bytes.data(), bytes.size(), hipMemcpyHostToDevice, n.st));
    ICS_HIP(hipStreamSynchronize(n.st));     // `bytes` is a host temporary
    n.d_pack_jobs = d; n.pack_njobs = nj; n.pack_nblocks = nb;
  }
  n.prof.begin(n.st, "pack_weights", 0, 0);
  ICS_TRY(launch_pack_table(n.st, n.d_pack_jobs, n.pack_njobs, n.pack_nblocks));
  n.prof.end(n.st);
  n.packed_valid = true;
  for (auto& L : n.layers) if (L->split_up) L->wf_stale = true;   // the table skips their full backward image
  return 0;
}
static int unet_pack(Net& n) { return run_pack_table(n, unet_pack_jobs); }

// trunk forward; upto_c10 for the perceptual sub-model (lattice_vae.py:257-270)
static int unet_forward_trunk(Net& n, int B, bool training, bool update_moving, bool upto_c10, const float* input) {
  ICS_TRY(unet_pack(n));
  UnetRefs r = unet_refs(n);
  (r.c1->pad_in ? r.c1->vsrc[0] : r.c1->src[0]).p = input;
  const int last = upto_c10 ? 8 : 14;
  for (int i = 0; i < last; ++i) {
    ConvLayer& L = *n.layers[i];
    ICS_TRY(conv_forward(n, L, B, training, update_moving, n.tp(L.t_b)));
  }
  return 0;
}

static int unet_head_forward(Net& n, int B) {
  ConvLayer& H = *n.head;
  return conv_forward(n, H, B, false, false, n.tp(H.t_b));
}

// mode 1 metrics: single GPU -> finalized on the spot.  With a communicator the six sums and the voxel
// count are all-reduced first (numerators / denominators, not ratios: SURVEY 8(e)); that costs a small
// collective on st, so it only happens when the caller reads the metrics (want_metrics) -- every rank
// must then ask on the same steps.
static int unet_loss(Net& n, int B, int mode, int want_grad, bool want_metrics = true) {
  const size_t M = n.rows(*n.head, B);
  n.prof.begin(n.st, "head_softmax_loss", 0, 4.0 * M * (n.ncls + 1) * 2);
  ICS_TRY(launch_head(n.st, n.head->s, n.ncls + 1, n.ncls, n.labels, M, mode, want_grad | (n.bce_from_logits << 1), n.loss_weight,
                      n.ws_dbl, 2048, n.comm ? nullptr : n.d_metrics, &n.head_nblk, want_grad ? n.ws_bwd : nullptr, n.d_red + 8));
  n.prof.end(n.st);
  if (mode != 0 && n.comm && want_metrics) {
    ICS_TRY(launch_head_metrics(n.st, n.ws_dbl, n.head_nblk, (double)M, n.d_metrics, n.d_red, 1));
    ncclResult_t r = ncclAllReduce(n.d_red, n.d_red, 7, ncclDouble, ncclSum, n.small(), n.st);
    ICS_CHECK(r == ncclSuccess, std::string("ncclAllReduce(metrics): ") + ncclGetErrorString(r));
    ICS_TRY(launch_head_metrics(n.st, n.ws_dbl, n.head_nblk, (double)M, n.d_metrics, n.d_red, 2, n.d_red + 8));
  }
  return 0;
}

// head GEMM + loss: one fused pass over the trunk output where the shape allows it (elementwise.hip head_fused_kernel:
// the logits never go to memory), else the 1x1x1 GEMM followed by head_kernel.  H.s receives what head_kernel would
// have left there: probabilities (mode 0) or dLoss/dz (mode 1 with want_grad).
static int unet_head_loss(Net& n, int B, int mode, int want_grad, bool want_metrics = true) {
  ConvLayer& H = *n.head;
  const size_t M = n.rows(H, B);
  const ConvSrc& s0 = H.src[0];
  if (H.nsrc != 1 || s0.up || s0.bcast || !head_fused_ok(n.ncls, s0.C, M, s0.act, n.flags)) {
    ICS_TRY(unet_head_forward(n, B));
    return unet_loss(n, B, mode, want_grad, want_metrics);
  }
  ICS_TRY(unet_pack(n));
  n.last_batch = B;
  n.prof.begin(n.st, "head_fused", 2.0 * M * 128 * (n.ncls + 1), 4.0 * M * (128 + (mode == 1 && !want_grad ? 0 : n.ncls + 1)));
  ICS_TRY(launch_head_fused(n.st, s0.p, s0.C, s0.scale, s0.shift, n.tp(H.t_w), n.tp(H.t_gamma), n.tp(H.t_b),
                            n.tp(H.t_b) + n.ncls, H.s, n.labels, M, mode, want_grad | (n.bce_from_logits << 1), n.loss_weight, n.ws_dbl, 2048,
                            n.comm ? nullptr : n.d_metrics, &n.head_nblk, want_grad ? n.ws_bwd : nullptr, n.d_red + 8));
  n.prof.end(n.st);
  if (mode != 0 && n.comm && want_metrics) {
    ICS_TRY(launch_head_metrics(n.st, n.ws_dbl, n.head_nblk, (double)M, n.d_metrics, n.d_red, 1));
    ncclResult_t r = ncclAllReduce(n.d_red, n.d_red, 7, ncclDouble, ncclSum, n.small(), n.st);
    ICS_CHECK(r == ncclSuccess, std::string("ncclAllReduce(metrics): ") + ncclGetErrorString(r));
    ICS_TRY(launch_head_metrics(n.st, n.ws_dbl, n.head_nblk, (double)M, n.d_metrics, n.d_red, 2, n.d_red + 8));
  }
  return 0;
}

__global__ void labels_kernel(const float* __restrict__ p, int ld, int ncls, size_t M, float thresh,
                              unsigned char* __restrict__ species, unsigned char* __restrict__ mask);
// generate.py:220-225: model.predict -> argmax / (sig >= thresh).  Where the fused head serves the shape the labels come
// straight out of its registers (the [M][96] probability tensor is neither written nor read back: 384 -> 2 bytes per voxel);
// otherwise GEMM + softmax kernel + labels_kernel.  Both give argmax / threshold of the SAME fp32 probabilities.
static int unet_head_labels(Net& n, int B, float thresh, unsigned char* d_species, unsigned char* d_mask, hipStream_t st) {
  ConvLayer& H = *n.head;
  const size_t M = n.rows(H, B);
  const ConvSrc& s0 = H.src[0];
  if (H.nsrc == 1 && !s0.up && !s0.bcast && head_fused_ok(n.ncls, s0.C, M, s0.act, n.flags) && !(n.flags & CF_NO_HEAD_LABELS)) {
    ICS_TRY(unet_pack(n));
    n.last_batch = B;
    n.prof.begin(st, "head_fused", 2.0 * M * 128 * (n.ncls + 1), 4.0 * M * 128 + 2.0 * M);
    ICS_TRY(launch_head_fused(st, s0.p, s0.C, s0.scale, s0.shift, n.tp(H.t_w), n.tp(H.t_gamma), n.tp(H.t_b),
                              n.tp(H.t_b) + n.ncls, H.s, n.labels, M, 2, 0, n.loss_weight, n.ws_dbl, 2048, nullptr, nullptr,
                              nullptr, nullptr, thresh, d_species, d_mask));
    n.prof.end(st);
    return 0;
  }
  ICS_TRY(unet_head_loss(n, B, 0, 0));
  n.prof.begin(st, "labels", 0, 4.0 * M * (n.ncls + 1) + 2.0 * M);
  ICS_LAUNCH(labels_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, H.s, n.ncls + 1, n.ncls, M, thresh,
                     d_species, d_mask);
  n.prof.end(st);
  ICS_HIP(hipGetLastError());
  return 0;
}

// merges per-block column partials [nblk][C] (head bias gradients; partials come from head_kernel)
__global__ __launch_bounds__(256) void colsum_merge_kernel(const float* __restrict__ partial, int nblk, int C,
                                                            float* __restrict__ out) {
  __shared__ double shd[256];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += (double)partial[(size_t)b * C + c];
  shd[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = (float)shd[0];
}
// tmp[K][na + nb] -> a[K][na], b[K][nb]
__global__ void split_cols_kernel(const float* __restrict__ tmp, int K, int N, int na, float* __restrict__ a,
                                  float* __restrict__ b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * N) return;
  const int k = i / N, c = i - k * N;
  if (c < na) a[k * na + c] = tmp[i];
  else b[k * (N - na) + (c - na)] = tmp[i];
}

// Cross-layer backward state (a consumer's launch leaves work or results for its producer's backward) is cleared on the
// success path as it is consumed; an error between a consumer and its producer would leave it set for the NEXT pass
// (a stale dy taken for a fresh one, finalize jobs pointing at old partials).  Every backward pass starts clean.
static void reset_backward_state(Net& n) {
  for (auto& L : n.layers) { L->dy_ready = false; L->deferred_skip = nullptr; L->db_blocks = 0; }
  n.colsum.n = 0;
}

static int unet_backward(Net& n, int B) {
  reset_backward_state(n);
  UnetRefs r = unet_refs(n);
  ConvLayer& H = *n.head;
  const size_t M = n.rows(H, B);
  const int nc1 = n.ncls + 1;
  // head: dz is in H.s (written by the loss kernel)
  const ConvGeom gh = geom_fwd(H, B);
  // Round 4: c18's BatchNorm backward inside the head's backward-data kernel (elementwise.hip head_dgrad_kernel<true>):
  // the head's weight-gradient GEMM runs on xhat (c18's normalised activation) instead of on gamma xhat + beta, and from
  // its result Q and the head's bias gradients a 128-thread kernel forms the head's weight gradients AND c18's
  // (c1, c2, dgamma, dbeta); the backward-data kernel then writes c18's dy directly.  With SyncBN the two sums are
  // all-reduced in between (256 doubles).  Only for the Conv -> ReLU -> BN shape the kernel hard-codes.
  const bool bnfuse = !(n.flags & CF_NO_HEAD_BNFUSE) && r.c18->has_bn && r.c18->pre_act == ACT_RELU &&
                      r.c18->post_act == ACT_NONE && r.c18->Cout == 128 && H.nsrc == 1 && head_dgrad_ok(n.ncls, 128, M, nullptr, n.flags);
  {
    // one GEMM over the [soft | sig] columns (one pass over c18's activations), then split into the two tensors
    ConvGeom gs = gh; gs.Cout = nc1; gs.Npad = round_up(nc1, 32);
    hipStream_t ws = bnfuse ? n.st : side_begin(n);     // fused: the backward-data kernel needs this GEMM's result
    ConvSrc hsrc = H.src[0];
    if (bnfuse) {
      // xhat as an affine of c18's stored activation: written by the forward's BatchNorm finalize, except under SyncBN
      // (its statistics come from the rank-merge kernels)
      float* xs = n.sync() ? n.head_xs : r.c18->xs;
      if (n.sync()) ICS_TRY(launch_xhat_affine(n.st, r.c18->mean, r.c18->rstd, 128, xs));
      hsrc.scale = xs; hsrc.shift = xs + 128;
    }
    n.prof.begin(ws, "conv_wgrad:head|", 2.0 * M * 128 * nc1, 4.0 * M * (128 + nc1));
    ICS_TRY(launch_conv_wgrad(ws, gs, &hsrc, 1, H.s, nc1, n.head_dw_tmp, nc1, n.ws_wgrad, n.ws_wgrad_n, 0, 0, 0, 1));
    n.prof.end(ws);
    n.prof.begin(ws, "wgrad_reduce_splits", 0, 0);
    ICS_TRY(launch_conv_wgrad(ws, gs, &hsrc, 1, H.s, nc1, n.head_dw_tmp, nc1, n.ws_wgrad, n.ws_wgrad_n, 0, 0, 0, 2));
    if (!bnfuse) {
      ICS_LAUNCH(split_cols_kernel, dim3((128 * nc1 + 255) / 256), dim3(256), 0, ws, n.head_dw_tmp, 128, nc1,
                         n.ncls, n.tg(H.t_w), n.tg(H.t_gamma));
      ICS_HIP(hipGetLastError());
    }
    n.prof.end(ws);
    // soft/bias | sig/bias (contiguous): the loss kernel left per-block column sums of dz in ws_bwd
    ICS_LAUNCH(colsum_merge_kernel, dim3(nc1), dim3(256), 0, n.st, n.ws_bwd, n.head_nblk, nc1, n.tg(H.t_b));
    ICS_HIP(hipGetLastError());
    if (bnfuse)
      ICS_TRY(launch_head_bnfuse(n.st, n.head_dw_tmp, n.tg(H.t_b), n.tp(H.t_w), n.tp(H.t_gamma), n.tp(r.c18->t_gamma),
                                 n.tp(r.c18->t_beta), (double)M, n.ncls, n.tg(H.t_w), n.tg(H.t_gamma), r.c18->c1c2,
                                 n.tg(r.c18->t_gamma), n.tg(r.c18->t_beta), n.sync()));
    const ConvGeom gb = geom_bwd(H, B);
    ConvSrc sdz = src_plain(H.s, nc1);
    n.prof.begin(n.st, "conv_dgrad:head|", 2.0 * M * 128 * nc1, 4.0 * M * (128 + nc1));
    BwdStat bs = bwd_stat_for(n, r.c18, B);
    int blocks = 0;
    if (bnfuse) {
      bs.s = r.c18->s; bs.mean = r.c18->mean; bs.rstd = r.c18->rstd; bs.scale = r.c18->scale; bs.shift = r.c18->shift;
      bs.partial = nullptr; bs.post_act = ACT_NONE; bs.ld = 128;
      ICS_TRY(launch_head_dgrad(n.st, H.s, n.tp(H.t_w), n.tp(H.t_gamma), r.c18->dy, 128, M, &bs, gb.Npad, &blocks,
                                r.c18->c1c2, r.c18->db_partial));
      n.prof.end(n.st);
      ICS_CHECK((size_t)blocks <= r.c18->db_rows, "bias-gradient partial buffer too small for this batch");
      ICS_TRY(colsum_push(n, r.c18->db_partial, blocks, 128, n.tg(r.c18->t_b)));
      r.c18->db_blocks = blocks;
    } else {
    if (head_dgrad_ok(n.ncls, 128, M, &bs, n.flags))
      ICS_TRY(launch_head_dgrad(n.st, H.s, n.tp(H.t_w), n.tp(H.t_gamma), H.dA, 128, M, &bs, gb.Npad, &blocks));
    else
      ICS_TRY(launch_conv_fwd(n.st, gb, &sdz, 1, H.wf, nullptr, H.dA, 128, ACT_NONE, nullptr, nullptr, 0, nullptr, 0, &bs,
                              &blocks));
    bwd_stat_done(r.c18, bs, blocks, gb.Npad);
    n.prof.end(n.st);
    }
  }
  // last argument: the layer fed ONLY by this layer's backward-data output (BatchNorm-backward sums folded in)
  auto bw = [&](ConvLayer* L, GradSrc g0, GradSrc g1, bool need_dA, ConvLayer* next = nullptr) {
    return conv_backward(n, *L, B, g0, g1, nullptr, need_dA, true, -1, next);
  };
  // gradient of a concat consumer w.r.t. its skip / upsampled producer
  auto g_skip = [&](ConvLayer* c) { return c->split_up ? gs_direct(c->dA_skip, c->Cs, 0) : gs_direct(c->dA, c->Cin, 0); };
  auto g_up = [&](ConvLayer* c) { return c->split_up ? gs_direct(c->dxl, c->Cu, 0) : gs_up(c->dA, c->Cin, c->src[0].C); };
  // after each layer its gradients (and everything behind them in the flat buffer) are final: grads_ready
  ICS_TRY(grads_ready(n, layer_lo(n, H)));
  if (bnfuse) ICS_TRY(conv_grads_from_dy(n, *r.c18, B, true, true, r.c17));     // c18's dy is already in place
  else ICS_TRY(bw(r.c18, gs_direct(H.dA, 128, 0), gs_none(), true, r.c17));
  ICS_TRY(bw(r.c17, gs_direct(r.c18->dA, 128, 0), gs_none(), true, r.c17->split_up ? r.c16 : nullptr));
  ICS_TRY(bw(r.c16, g_up(r.c17), gs_none(), true, r.c15));
  ICS_TRY(bw(r.c15, gs_direct(r.c16->dA, 256, 0), gs_none(), true, r.c15->split_up ? r.c14 : nullptr));
  ICS_TRY(grads_ready(n, layer_lo(n, *r.c15)));
  ICS_TRY(bw(r.c14, g_up(r.c15), gs_none(), true, r.c13));
  ICS_TRY(grads_ready(n, layer_lo(n, *r.c14)));
  ICS_TRY(bw(r.c13, gs_direct(r.c14->dA, 512, 0), gs_none(), true, r.c13->split_up ? r.c10 : nullptr));
  ICS_TRY(grads_ready(n, layer_lo(n, *r.c13)));
  ICS_TRY(bw(r.c10, g_up(r.c13), gs_none(), true, r.c9));
  ICS_TRY(grads_ready(n, layer_lo(n, *r.c10)));
  ICS_TRY(bw(r.c9, gs_direct(r.c10->dA, 512, 0), gs_none(), true));
  ICS_TRY(grads_ready(n, layer_lo(n, *r.c9)));
  ICS_TRY(bw(r.c6, gs_pool(r.c9->dA, 256, *r.c6), g_skip(r.c13), true, r.c5));
  ICS_TRY(bw(r.c5, gs_direct(r.c6->dA, 128, 0), gs_none(), true));
  ICS_TRY(bw(r.c4, gs_pool(r.c5->dA, 128, *r.c4), g_skip(r.c15), true, r.c3));
  ICS_TRY(bw(r.c3, gs_direct(r.c4->dA, 64, 0), gs_none(), true));
  ICS_TRY(bw(r.c2, gs_pool(r.c3->dA, 64, *r.c2), g_skip(r.c17), true, r.c1));
  ICS_TRY(bw(r.c1, gs_direct(r.c2->dA, 32, 0), gs_none(), false));
  return 0;
}

// perceptual sub-model backward: gradients w.r.t. the input given dtap (weights frozen)
static int unet_pm_backward(Net& n, int B) {
  UnetRefs r = unet_refs(n);
  // the taps' squared-difference loss terms and their gradients are formed inside the BN-backward pass of the
  // tap layers (tap >= 0): no separate sqdiff pass over the four tap tensors, no dtap buffers
  auto bw = [&](ConvLayer* L, GradSrc g0, int tap, bool need_dA, ConvLayer* next = nullptr) {
    return conv_backward(n, *L, B, g0, gs_none(), nullptr, need_dA, false, tap, next);
  };
  ICS_TRY(bw(r.c10, gs_none(), 3, true, r.c9));
  ICS_TRY(bw(r.c9, gs_direct(r.c10->dA, 512, 0), -1, true));
  ICS_TRY(bw(r.c6, gs_pool(r.c9->dA, 256, *r.c6), 2, true, r.c5));
  ICS_TRY(bw(r.c5, gs_direct(r.c6->dA, 128, 0), -1, true));
  ICS_TRY(bw(r.c4, gs_pool(r.c5->dA, 128, *r.c4), 1, true, r.c3));
  ICS_TRY(bw(r.c3, gs_direct(r.c4->dA, 64, 0), -1, true));
  ICS_TRY(bw(r.c2, gs_pool(r.c3->dA, 64, *r.c2), 0, true, r.c1));
  ICS_TRY(bw(r.c1, gs_direct(r.c2->dA, 32, 0), -1, true));
  return 0;
}

static int unet_train_resident_impl(Net& n, int B, float* metrics);
static int unet_train_resident(Net& n, int B, float* metrics) {
  n.splitk = true;
  n.want_wgrad_inputs = true;
  const int rc = unet_train_resident_impl(n, B, metrics);
  n.splitk = false;
  n.want_wgrad_inputs = false;
  return rc;
}
static int unet_train_resident_impl(Net& n, int B, float* metrics) {
  Range step("icsg3d.unet.train_step");
  {
    Range r("icsg3d.unet.forward");
    ICS_TRY(unet_forward_trunk(n, B, true, true, false, n.x_in));
    ICS_TRY(unet_head_loss(n, B, 1, 1, metrics != nullptr));
  }
  {
    Range r("icsg3d.unet.backward");
    ICS_TRY(grads_begin(n));
    ICS_TRY(unet_backward(n, B));
  }
  {
    Range r("icsg3d.unet.adam");
    ICS_TRY(adam_step(n));
  }
  if (metrics) {
    ICS_HIP(hipMemcpyAsync(metrics, n.d_metrics, 5 * sizeof(float), hipMemcpyDeviceToHost, n.st));
    ICS_HIP(hipStreamSynchronize(n.st));
    n.prof.resolve();
  }
  return 0;
}

static int unet_upload(Net& n, const float* x, const unsigned char* labels, int B) {
  ICS_CHECK(B >= 1 && B <= n.maxB, "batch exceeds max_batch");
  const size_t M = (size_t)B * n.d * n.d * n.d;
  ICS_HIP(hipMemcpyAsync(n.x_in, x, M * n.C * sizeof(float), hipMemcpyHostToDevice, n.st));
  if (labels) ICS_HIP(hipMemcpyAsync(n.labels, labels, M, hipMemcpyHostToDevice, n.st));
  n.resident_batch = B;
  return 0;
}

// ------------------------------------------------------------------------------------------
// VAE
// ------------------------------------------------------------------------------------------
struct VaeRefs {
  ConvLayer *e[4], *e4, *encd, *zml, *decd, *dl[4], *dout;
};
static VaeRefs vae_refs(Net& n) {
  VaeRefs r;
  int i = 0;
  for (int k = 0; k < 4; ++k) r.e[k] = n.layers[i++].get();
  r.e4 = n.layers[i++].get(); r.encd = n.layers[i++].get(); r.zml = n.layers[i++].get();
  r.decd = n.layers[i++].get();
  for (int k = 0; k < 4; ++k) r.dl[k] = n.layers[i++].get();
  r.dout = n.layers[i++].get();
  return r;
}

static int vae_build(Net& n, const ics_vae_config& cfg, Net* pm) {
  n.kind = 1; n.maxB = cfg.max_batch; n.d = cfg.d; n.C = cfg.in_channels; n.pm = pm;
  n.pm_side = getenv("ICSG3D_NO_PM_SIDE") == nullptr;
  // Round 5: the VAE's weight-gradient launches (small GEMMs, split reductions, the condition-weight sums: ~0.7 ms) leave the
  // critical chain bn_bwd -> backward-data -> next layer for the second stream, as the U-Net engine can (opt-in there):
  // 5.71 -> 5.44 ms per step.  (Round 1 measured the opposite, 12.59 -> 12.71 ms: the kernels were 2x longer and filled
  // the chip.)  ICSG3D_NO_VAE_SIDE_WGRAD=1: one stream.
  n.side_on = getenv("ICSG3D_NO_VAE_SIDE_WGRAD") == nullptr;
  n.ncond = cfg.cond_shape; n.latent = cfg.latent_dim; n.lr = cfg.lr; n.alpha = cfg.alpha; n.beta = cfg.beta;
  n.bn_unbias = cfg.bn_unbias; n.pool_ties_all = pm ? pm->pool_ties_all : 1;
  for (int i = 0; i < 4; ++i) { n.filters[i] = cfg.filters[i]; n.pm_w[i] = cfg.pm_layer_weights[i]; }
  const int d = cfg.d, C = cfg.in_channels, lat = cfg.latent_dim;
  ICS_CHECK(d >= 16 && (d & (d - 1)) == 0, "d must be a power of two >= 16");
  int cin = C + C * cfg.cond_shape;   // K.tile quirk: cond channels = C*cond_shape
  int S = d;
  for (int i = 0; i < 4; ++i) {
    ConvLayer* L = add_conv(n, "e" + std::to_string(i), cin, cfg.filters[i], 27, S, ACT_NONE, 1, ACT_LRELU, false, true, i == 0);
    // e0 at C = 1: condition channels folded analytically (exact; needs the single-channel stencil kernels)
    if (i == 0 && C == 1 && !(n.flags & (CF_NO_COND_FOLD | CF_NO_THIN_C)) && (cfg.filters[0] == 16 || cfg.filters[0] == 32))
      L->cond_fold = true;
    cin = cfg.filters[i]; S /= 2;
  }
  add_conv(n, "e4", cin, 4, 27, S, ACT_LRELU, 0, ACT_NONE, false);        // S = d/16
  const int flat = S * S * S * 4;
  add_conv(n, "enc_dense", flat, lat, 1, 1, ACT_RELU, 0, ACT_NONE, true);
  // z_mean | z_log_var as one GEMM with N = 2*latent; parameters stay two reference layers
  ConvLayer* zml = add_conv(n, "zmulv", lat, 2 * lat, 1, 1, ACT_NONE, 0, ACT_NONE, true, false);
  const int t_zm = n.add_tensor("z_mean/kernel", {lat, lat}, true);
  const int t_zl = n.add_tensor("z_log_var/kernel", {lat, lat}, true);
  const int t_zb = n.add_tensor("z_mean/bias", {lat}, true);   // z_mean/bias | z_log_var/bias contiguous
  n.add_tensor("z_log_var/bias", {lat}, true);
  zml->t_w = t_zm; zml->t_gamma = t_zl; zml->t_b = t_zb;
  const int seedS = d / 8, seed = seedS * seedS * seedS * 4;
  add_conv(n, "dec_dense", lat + cfg.cond_shape, seed, 1, 1, ACT_NONE, 0, ACT_NONE, true);
  cin = 4; S = seedS;
  for (int i = 0; i < 4; ++i) {
    add_conv(n, "d" + std::to_string(i), cin, cfg.filters[3 - i], 27, S, ACT_NONE, 1, ACT_LRELU, false);
    cin = cfg.filters[3 - i];
    if (i < 3) S *= 2;
  }
  add_conv(n, "dout", cin, C, 27, S, ACT_NONE, 1, ACT_RELU, false);
  ICS_TRY(alloc_params(n));
  const size_t M = (size_t)n.maxB * d * d * d;
  ICS_TRY(n.alloc(&n.x_in, M * C));
  ICS_TRY(n.alloc(&n.cond_in, (size_t)n.maxB * cfg.cond_shape));
  ICS_TRY(n.alloc(&n.eps_in, (size_t)n.maxB * lat));
  ICS_TRY(n.alloc(&n.z_buf, (size_t)n.maxB * lat));
  ICS_TRY(n.alloc(&n.zc, (size_t)n.maxB * (lat + cfg.cond_shape)));
  ICS_TRY(n.alloc(&n.recon, M * C));
  ICS_TRY(n.alloc(&n.drecon, M * C));
  ICS_TRY(n.alloc(&n.dmulv, (size_t)n.maxB * 2 * lat));
  VaeRefs r = vae_refs(n);
  for (int i = 0; i < 4; ++i) ICS_TRY(alloc_layer(n, *r.e[i], true, true));
  ICS_TRY(alloc_layer(n, *r.e4, true, false));
  ICS_TRY(alloc_layer(n, *r.encd, true, false));
  ICS_TRY(alloc_layer(n, *r.zml, true, false));
  ICS_TRY(alloc_layer(n, *r.decd, true, false));
  for (int i = 0; i < 4; ++i) ICS_TRY(alloc_layer(n, *r.dl[i], true, false));
  ICS_TRY(alloc_layer(n, *r.dout, true, false));
  // sources
  r.e[0]->src[0] = src_plain(n.x_in, C);
  r.e[0]->src[1] = ConvSrc{n.cond_in, nullptr, nullptr, C * cfg.cond_shape, 0, ACT_NONE, cfg.cond_shape};
  r.e[0]->nsrc = 2;
  for (int i = 1; i < 4; ++i) r.e[i]->src[0] = src_plain(r.e[i - 1]->pooled, r.e[i - 1]->Cout);
  r.e4->src[0] = src_plain(r.e[3]->pooled, r.e[3]->Cout);
  r.encd->src[0] = src_plain(r.e4->s, flat);
  r.zml->src[0] = src_plain(r.encd->s, lat);
  r.decd->src[0] = src_plain(n.zc, lat + cfg.cond_shape);
  r.dl[0]->src[0] = src_plain(r.decd->s, 4);
  for (int i = 1; i < 4; ++i) r.dl[i]->src[0] = src_layer(*r.dl[i - 1], 1);   // UpSampling3D after d0..d2
  r.dout->src[0] = src_layer(*r.dl[3], 0);
  use_padded_input(*r.e[0]);
  for (int i = 1; i < 4; ++i) ICS_TRY(enable_split_up(n, *r.dl[i]));
  for (auto& Lp : n.layers) ICS_TRY(enable_winog(n, *Lp, true));      // e3 at d = 32 (4^3 x 64 -> 128)
  ICS_TRY(alloc_workspaces(n, true));
  ICS_TRY(init_bn_defaults(n));
  if (pm) {
    // perceptual tap copies / gradients live with the U-Net (sized for ITS max batch)
    UnetRefs u = unet_refs(*pm);
    ConvLayer* taps[4] = {u.c2, u.c4, u.c6, u.c10};
    ICS_CHECK(pm->maxB >= n.maxB && pm->d == d && pm->C == C, "perceptual U-Net shape mismatch");
    for (int l = 0; l < 4; ++l) {
      const size_t cnt = pm->rows(*taps[l], pm->maxB) * taps[l]->Cout;
      if (!pm->tap_copy[l]) ICS_TRY(pm->alloc(&pm->tap_copy[l], cnt));
    }
    ICS_HIP(hipStreamSynchronize(n.st));
    ICS_HIP(hipStreamSynchronize(pm->st));
  }
  return 0;
}

static int vae_pack_jobs(Net& n) {
  VaeRefs r = vae_refs(n);
  for (auto& Lp : n.layers) {
    ConvLayer& L = *Lp;
    if (&L == r.zml) continue;
    ICS_TRY(pack_layer(n, L, true));
  }
  ConvLayer& Z = *r.zml;
  const int lat = n.latent;
  ICS_TRY(launch_pack_fwd(n.st, n.tp(Z.t_w), lat, lat, Z.wp, Z.Kpad, Z.Npad, 0, 0, 1));
  ICS_TRY(launch_pack_fwd(n.st, n.tp(Z.t_gamma), lat, lat, Z.wp, Z.Kpad, Z.Npad, 0, lat, 0));
  ICS_TRY(launch_pack_bwd(n.st, n.tp(Z.t_w), 1, lat, lat, Z.wf, Z.Kpad_b, Z.Npad_b, 2 * lat, 0, 1));
  ICS_TRY(launch_pack_bwd(n.st, n.tp(Z.t_gamma), 1, lat, lat, Z.wf, Z.Kpad_b, Z.Npad_b, 2 * lat, lat, 0));
  return 0;
}
static int vae_pack(Net& n) { return run_pack_table(n, vae_pack_jobs); }

static int vae_encode_fwd(Net& n, int B, bool training) {
  ICS_TRY(vae_pack(n));
  VaeRefs r = vae_refs(n);
  for (int i = 0; i < 4; ++i) ICS_TRY(conv_forward(n, *r.e[i], B, training, training, n.tp(r.e[i]->t_b)));
  ICS_TRY(conv_forward(n, *r.e4, B, training, training, n.tp(r.e4->t_b)));
  ICS_TRY(conv_forward(n, *r.encd, B, training, training, n.tp(r.encd->t_b)));
  ICS_TRY(conv_forward(n, *r.zml, B, training, training, n.tp(r.zml->t_b)));
  ICS_TRY(launch_sampling(n.st, r.zml->s, 2 * n.latent, n.latent, n.eps_in, n.cond_in, n.ncond, B, n.z_buf, n.zc));
  return 0;
}

static int vae_decode_fwd(Net& n, int B, bool training) {
  ICS_TRY(vae_pack(n));
  VaeRefs r = vae_refs(n);
  ICS_TRY(conv_forward(n, *r.decd, B, training, training, n.tp(r.decd->t_b)));
  for (int i = 0; i < 4; ++i) ICS_TRY(conv_forward(n, *r.dl[i], B, training, training, n.tp(r.dl[i]->t_b)));
  ICS_TRY(conv_forward(n, *r.dout, B, training, training, n.tp(r.dout->t_b)));
  const size_t cnt = n.rows(*r.dout, B) * n.C;
  ICS_TRY(launch_bn_apply(n.st, r.dout->s, r.dout->scale, r.dout->shift, ACT_RELU, cnt, n.C, n.recon));
  return 0;
}

// both engines share one stream in a VAE step: the U-Net is driven on the VAE's stream
static int vae_step(Net& n, int B, bool training, float* metrics) {
  ICS_CHECK(n.pm != nullptr, "VAE engine has no perceptual U-Net");
  Range step(training ? "icsg3d.vae.train_step" : "icsg3d.vae.test_step");
  Net& u = *n.pm;
  hipStream_t saved = u.st;
  u.st = n.st;
  n.splitk = u.splitk = training;
  n.want_wgrad_inputs = training;      // the VAE's own layers get parameter gradients; the perceptual U-Net's do not
  // SyncBN covers the perceptual U-Net's batch-statistics BatchNorm too (it borrows the VAE's communicator)
  ncclComm_t saved_comm = u.comm;
  const int saved_sync = u.sync_bn;
  const BnSync saved_bs = u.bn_sync;
  if (n.sync()) { u.comm = n.comm; u.sync_bn = 1; u.bn_sync = n.bn_sync; }
  int rc = 0;
  reset_backward_state(n);
  reset_backward_state(u);
  do {
    VaeRefs r = vae_refs(n);
    UnetRefs ur = unet_refs(u);
    ConvLayer* taps[4] = {ur.c2, ur.c4, ur.c6, ur.c10};
    const size_t M = (size_t)B * n.d * n.d * n.d;
    // perceptual pass on y_true: the four tap layers write straight into tap_copy (nothing in the c1..c10 trunk
    // reads a tap layer's s except its own pooling, which takes the pointer at launch time); then the pass on
    // y_pred into the layers' own buffers (state kept for the backward pass)
    // (Measured on MI355X: running this pass on a second stream, concurrently with the encoder / decoder, overlaps
    // 3.8 ms of kernels but each runs slower while sharing the chip -- step 12.69 vs 12.59 ms: not done.)
    // Round 5: the y_true pass needs nothing but x and produces nothing the encoder / decoder read, and those are ~60
    // small-grid launches at the 5 us floor that leave most of the chip idle: the pass runs on the VAE's second stream
    // next to them and is joined before the y_pred pass (which reuses the U-Net's layer buffers).  Not under SyncBN (the
    // pass's statistics collectives would interleave with the encoder's on one communicator).  ICSG3D_NO_PM_SIDE=1: serial.
    // ... and not while every launch is being timed (profiler on, no site filter): a kernel's event bracket would then
    // time the kernels of the other stream it shares the chip with -- the per-kernel table comes from a serial step.
    const bool timing_all = (n.prof.on && n.prof.filter.empty()) || (u.prof.on && u.prof.filter.empty());
    const bool pm_side = n.pm_side && !n.sync() && !u.sync() && !timing_all;
    if (pm_side) {
      if (hipEventRecord(n.ev_fork, n.st) != hipSuccess || hipStreamWaitEvent(n.st2, n.ev_fork, 0) != hipSuccess) {
        set_error("side stream fork failed"); rc = -1; break;
      }
      u.st = n.st2;
    }
    float* own_s[4];
    for (int l = 0; l < 4; ++l) { own_s[l] = taps[l]->s; taps[l]->s = u.tap_copy[l]; }
    rc = unet_forward_trunk(u, B, training, false, true, n.x_in);
    for (int l = 0; l < 4; ++l) taps[l]->s = own_s[l];
    u.st = n.st;
    if (rc) break;
    if ((rc = vae_encode_fwd(n, B, training))) break;
    if ((rc = vae_decode_fwd(n, B, training))) break;
    if (pm_side && (hipEventRecord(n.ev_join, n.st2) != hipSuccess || hipStreamWaitEvent(n.st, n.ev_join, 0) != hipSuccess)) {
      set_error("side stream join failed"); rc = -1; break;
    }
    u.want_tie_stats = training;       // this pass is differentiated (unet_pm_backward)
    rc = unet_forward_trunk(u, B, training, false, true, n.recon);
    u.want_tie_stats = false;
    if (rc) break;
    // loss terms (+ gradients when training)
    double* mse_part = n.ws_dbl;
    const int mse_bps = 16;
    size_t off = (size_t)B * mse_bps;
    if ((rc = launch_sqdiff(n.st, n.x_in, n.recon, B, M / B * n.C, mse_bps, mse_part, training ? n.drecon : nullptr,
                            (float)(2.0 / ((double)M * n.C)), 0))) break;
    double* pm_part = n.ws_dbl + off;
    PmSums pmc{};
    size_t poff = 0;
    for (int l = 0; l < 4; ++l) {
      const size_t per = (size_t)taps[l]->S * taps[l]->S * taps[l]->S * taps[l]->Cout;
      pmc.per[l] = (double)per; pmc.w[l] = n.pm_w[l];
      if (training) {
        // formed inside the BN-backward pass of the tap layer (unet_pm_backward): partials per block of that pass
        LayerBwd lb{};
        lb.B = B; lb.S = taps[l]->S; lb.lgS = ilog2(taps[l]->S); lb.C = taps[l]->Cout;
        int rpb;
        pmc.n[l] = bn_bwd_num_blocks(lb, &rpb);
        u.tap_ref[l] = u.tap_copy[l];
        u.tap_coef[l] = (float)(2.0 * n.alpha * n.pm_w[l] / ((double)per * B));
        u.tap_partial[l] = pm_part + poff;
      } else {
        const int bps = (int)std::min<size_t>(256, std::max<size_t>(8, per / 16384));
        pmc.n[l] = B * bps;
        if ((rc = launch_sqdiff(n.st, u.tap_copy[l], taps[l]->s, B, per, bps, pm_part + poff, nullptr, 0.f, 0))) break;
      }
      poff += (size_t)pmc.n[l];
    }
    if (rc) break;
    if (off + poff > n.ws_dbl_n) { set_error("loss partial workspace too small"); rc = -1; break; }
    if (training) {
      if ((rc = grads_begin(n))) break;
      if ((rc = unet_pm_backward(u, B))) break;
    }
    if (!n.comm) {
      // the loss VALUE is not an input of the backward pass: on the second stream (when the step uses one) it leaves the
      // critical chain; adam_step joins the streams before the metrics can be read (round 6)
      hipStream_t ls = training ? side_begin(n) : n.st;
      if ((rc = launch_vae_loss(ls, r.zml->s, 2 * n.latent, n.latent, B, mse_part, B * mse_bps, (double)M * n.C,
                                pm_part, pmc, n.alpha, n.beta, n.d_metrics))) break;
    } else if (metrics) {
      // data parallel: all-reduce the sums (numerators / denominators), then form the means
      if ((rc = launch_vae_loss(n.st, r.zml->s, 2 * n.latent, n.latent, B, mse_part, B * mse_bps, (double)M * n.C,
                                pm_part, pmc, n.alpha, n.beta, n.d_metrics, n.d_red, 1))) break;
      if (ncclAllReduce(n.d_red, n.d_red, 5, ncclDouble, ncclSum, n.small(), n.st) != ncclSuccess) {
        set_error("ncclAllReduce(vae metrics) failed"); rc = -1; break;
      }
      if ((rc = launch_vae_loss(n.st, r.zml->s, 2 * n.latent, n.latent, B, mse_part, B * mse_bps, (double)M * n.C,
                                pm_part, pmc, n.alpha, n.beta, n.d_metrics, n.d_red, 2))) break;
    }
    if (training) {
      ICS_LAUNCH(axpy_strided_kernel, dim3((unsigned)((M * n.C + 255) / 256)), dim3(256), 0, n.st, n.drecon,
                         ur.c1->dA, M, n.C, ur.c1->CinG);
      // decoder
      if ((rc = conv_backward(n, *r.dout, B, gs_direct(n.drecon, n.C, 0), gs_none(), nullptr, true, true))) break;
      if ((rc = conv_backward(n, *r.dl[3], B, gs_direct(r.dout->dA, r.dout->Cin, 0), gs_none(), nullptr, true, true))) break;
      for (int i = 2; i >= 0; --i)
        if ((rc = conv_backward(n, *r.dl[i], B,
                                r.dl[i + 1]->split_up ? gs_direct(r.dl[i + 1]->dxl, r.dl[i + 1]->Cu, 0)
                                                      : gs_up(r.dl[i + 1]->dA, r.dl[i + 1]->Cin, 0),
                                gs_none(), nullptr, true, true))) break;
      if (rc) break;
      if ((rc = conv_backward(n, *r.decd, B, gs_direct(r.dl[0]->dA, r.decd->Cout, 0), gs_none(), nullptr, true, true))) break;
      if ((rc = grads_ready(n, layer_lo(n, *r.decd)))) break;   // decoder gradients (the tail of the flat buffer)
      // sampling + KL
      if ((rc = launch_vae_dz(n.st, r.zml->s, 2 * n.latent, n.latent, B, n.eps_in, r.decd->dA, r.decd->Cin, n.beta, n.dmulv))) break;
      // zmulv given dy = dmulv
      ConvLayer& Z = *r.zml;
      const int lat = n.latent;
      {
        ConvGeom g1 = geom_fwd(Z, B); g1.Cout = lat; g1.Npad = round_up(lat, 32);
        hipStream_t ws = side_begin(n);
        if ((rc = launch_conv_wgrad(ws, g1, Z.src, 1, n.dmulv, 2 * lat, n.tg(Z.t_w), lat, n.ws_wgrad, n.ws_wgrad_n))) break;
        if ((rc = launch_conv_wgrad(ws, g1, Z.src, 1, n.dmulv + lat, 2 * lat, n.tg(Z.t_gamma), lat, n.ws_wgrad, n.ws_wgrad_n))) break;
        if ((rc = launch_colsum_small(n.st, n.dmulv, B, 2 * lat, 2 * lat, n.tg(Z.t_b)))) break;
        const ConvGeom gb = geom_bwd(Z, B);
        ConvSrc sd = src_plain(n.dmulv, 2 * lat);
        if ((rc = launch_conv_fwd(n.st, gb, &sd, 1, Z.wf, nullptr, Z.dA, lat, ACT_NONE, nullptr, nullptr))) break;
      }
      if ((rc = conv_backward(n, *r.encd, B, gs_direct(Z.dA, lat, 0), gs_none(), nullptr, true, true))) break;
      if ((rc = conv_backward(n, *r.e4, B, gs_direct(r.encd->dA, 4, 0), gs_none(), nullptr, true, true))) break;
      for (int i = 3; i >= 0; --i) {
        ConvLayer* cons = (i == 3) ? r.e4 : r.e[i + 1];
        if ((rc = conv_backward(n, *r.e[i], B, gs_pool(cons->dA, cons->Cin, *r.e[i]), gs_none(), nullptr, i > 0, true))) break;
      }
      if (rc) break;
      if ((rc = adam_step(n))) break;
    }
    if (metrics) {
      if (hipMemcpyAsync(metrics, n.d_metrics, 4 * sizeof(float), hipMemcpyDeviceToHost, n.st) != hipSuccess ||
          hipStreamSynchronize(n.st) != hipSuccess) { set_error("metrics copy failed"); rc = -1; break; }
      n.prof.resolve();
    }
  } while (0);
  u.st = saved;
  u.comm = saved_comm; u.sync_bn = saved_sync; u.bn_sync = saved_bs;
  n.splitk = u.splitk = false;
  n.want_wgrad_inputs = false;
  return rc;
}

static int vae_upload(Net& n, const float* x, const float* cond, const float* eps, int B) {
  ICS_CHECK(B >= 1 && B <= n.maxB, "batch exceeds max_batch");
  const size_t M = (size_t)B * n.d * n.d * n.d;
  if (x) ICS_HIP(hipMemcpyAsync(n.x_in, x, M * n.C * sizeof(float), hipMemcpyHostToDevice, n.st));
  if (cond) ICS_HIP(hipMemcpyAsync(n.cond_in, cond, (size_t)B * n.ncond * sizeof(float), hipMemcpyHostToDevice, n.st));
  if (eps) ICS_HIP(hipMemcpyAsync(n.eps_in, eps, (size_t)B * n.latent * sizeof(float), hipMemcpyHostToDevice, n.st));
  n.resident_batch = B;
  return 0;
}

// compact columns [col0, col0+ncols) of a [M][ld] matrix into a dense [M][ncols] buffer
__global__ void gather_cols_kernel(const float* __restrict__ src, int ld, int col0, int ncols, size_t M,
                                   float* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * (size_t)ncols) return;
  const size_t row = i / ncols;
  const int c = (int)(i - row * ncols);
  dst[i] = src[row * ld + col0 + c];
}

// argmax species + thresholded mask from head probabilities (generate.py:221-225)
__global__ void labels_kernel(const float* __restrict__ p, int ld, int ncls, size_t M, float thresh,
                              unsigned char* __restrict__ species, unsigned char* __restrict__ mask) {
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= M) return;
  const float* pr = p + row * ld;
  int best = 0;
  float bv = pr[0];
  for (int c = 1; c < ncls; ++c)
    if (pr[c] > bv) { bv = pr[c]; best = c; }   // np.argmax: first maximum
  species[row] = (unsigned char)best;
  mask[row] = pr[ncls] >= thresh ? 1 : 0;
}

// per-sample min / max of channels [c0, c0+nch) of an NDHWC tensor: out[(b*nch + j)*2 + {0,1}]
// (to_lattice_params, /root/reference/utils.py:160-178, needs nothing else of the coordinate channels)
__global__ __launch_bounds__(256) void chan_minmax_kernel(const float* __restrict__ x, size_t per_sample, int C, int c0,
                                                           int nch, float* __restrict__ out) {
  __shared__ float smin[256], smax[256];
  const int b = blockIdx.x / nch, j = blockIdx.x % nch;
  const float* p = x + (size_t)b * per_sample * C + c0 + j;
  float lo = INFINITY, hi = -INFINITY;
  for (size_t i = threadIdx.x; i < per_sample; i += 256) {
    const float v = p[i * C];
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + o]);
      smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = smin[0]; out[blockIdx.x * 2 + 1] = smax[0]; }
}

}  // namespace ics

// ==========================================================================================
// C ABI
// ==========================================================================================
using namespace ics;

struct ics_net {
  Net n;
};

extern "C" {

const char* ics_last_error(void) { return g_err.c_str(); }
const char* ics_version(void) { return "icsg3d_amd 0.6 (gfx950, fp32 MFMA implicit-GEMM + Winograd)"; }
long long ics_kernel_launches(void) { return g_kernel_launches.load(std::memory_order_relaxed); }

int ics_device_count(int* count) {
  ICS_HIP(hipGetDeviceCount(count));
  return 0;
}
int ics_set_device(int device) {
  ICS_HIP(hipSetDevice(device));
  return 0;
}
int ics_device_info(char* name, int* cus, size_t* hbm) {
  int dev = 0;
  ICS_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  ICS_HIP(hipGetDeviceProperties(&p, dev));
  if (name) { std::strncpy(name, p.name, 255); name[255] = 0; }
  if (cus) *cus = p.multiProcessorCount;
  if (hbm) *hbm = p.totalGlobalMem;
  return 0;
}

// A C caller that never goes through icsg3d_amd/_lib.py gets the same default (see there): eight hardware queues, unless the
// environment already says otherwise.  Runs when the library is loaded; the runtime reads it when HIP initialises.
__attribute__((constructor)) static void ics_runtime_env_defaults() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

static int net_common_init(Net& n) {
  n.flags = conv_flags_from_env();
  ICS_HIP(hipGetDevice(&n.device));
  ICS_HIP(hipStreamCreateWithFlags(&n.st, hipStreamNonBlocking));
  ICS_HIP(hipStreamCreateWithFlags(&n.st2, hipStreamNonBlocking));
  ICS_HIP(hipEventCreateWithFlags(&n.ev_fork, hipEventDisableTiming));
  ICS_HIP(hipEventCreateWithFlags(&n.ev_join, hipEventDisableTiming));
  n.side_on = false;   // enabled per engine kind in *_build
  return 0;
}

int ics_unet_create(const ics_unet_config* cfg, ics_net** out) {
  ICS_CHECK(cfg && out, "null argument");
  ICS_CHECK(cfg->d >= 8 && (cfg->d & (cfg->d - 1)) == 0, "d must be a power of two >= 8");
  ICS_CHECK(cfg->in_channels >= 1 && cfg->max_batch >= 1 && cfg->num_classes >= 2 && cfg->num_classes <= 127,
            "bad U-Net config");
  auto h = std::make_unique<ics_net>();
  ICS_TRY(net_common_init(h->n));
  ICS_TRY(unet_build(h->n, *cfg));
  ICS_HIP(hipStreamSynchronize(h->n.st));
  *out = h.release();
  return 0;
}

int ics_vae_create(const ics_vae_config* cfg, ics_net* pm, ics_net** out) {
  ICS_CHECK(cfg && out, "null argument");
  ICS_CHECK(pm == nullptr || pm->n.kind == 0, "perceptual model must be a U-Net engine");
  auto h = std::make_unique<ics_net>();
  ICS_TRY(net_common_init(h->n));
  ICS_TRY(vae_build(h->n, *cfg, pm ? &pm->n : nullptr));
  ICS_HIP(hipStreamSynchronize(h->n.st));
  *out = h.release();
  return 0;
}

int ics_net_destroy(ics_net* net) {
  if (!net) return 0;
  (void)hipStreamSynchronize(net->n.st);
  delete net;            // (with ICSG3D_DEBUG_CANARY=1 the destructor reports overwritten guard bytes on stderr)
  return 0;
}
int ics_net_sync(ics_net* net) {
  ICS_CHECK(net, "null handle");
  ICS_HIP(hipStreamSynchronize(net->n.st));
  net->n.prof.resolve();
  return 0;
}

int ics_net_num_tensors(ics_net* net, int* count) {
  ICS_CHECK(net && count, "null argument");
  *count = (int)net->n.tensors.size();
  return 0;
}
int ics_net_tensor_info(ics_net* net, int index, const char** name, int* ndim, int64_t dims[5], int* trainable) {
  ICS_CHECK(net && index >= 0 && index < (int)net->n.tensors.size(), "tensor index out of range");
  const Tensor& t = net->n.tensors[index];
  if (name) *name = t.name.c_str();
  if (ndim) *ndim = (int)t.dims.size();
  if (dims) for (size_t i = 0; i < t.dims.size() && i < 5; ++i) dims[i] = t.dims[i];
  if (trainable) *trainable = t.trainable ? 1 : 0;
  return 0;
}
static int find_tensor(ics_net* net, const char* name, Tensor** t) {
  ICS_CHECK(net && name, "null argument");
  auto it = net->n.tindex.find(name);
  ICS_CHECK(it != net->n.tindex.end(), std::string("unknown tensor: ") + name);
  *t = &net->n.tensors[it->second];
  return 0;
}
int ics_net_set_tensor(ics_net* net, const char* name, const float* host, size_t count) {
  Tensor* t;
  ICS_TRY(find_tensor(net, name, &t));
  ICS_CHECK(count == t->count, std::string("size mismatch for tensor ") + name);
  ICS_HIP(hipMemcpyAsync(t->ptr, host, count * sizeof(float), hipMemcpyHostToDevice, net->n.st));
  ICS_HIP(hipStreamSynchronize(net->n.st));
  net->n.packed_valid = false;
  return 0;
}
int ics_net_get_tensor(ics_net* net, const char* name, float* host, size_t count) {
  Tensor* t;
  ICS_TRY(find_tensor(net, name, &t));
  ICS_CHECK(count == t->count, std::string("size mismatch for tensor ") + name);
  ICS_HIP(hipMemcpyAsync(host, t->ptr, count * sizeof(float), hipMemcpyDeviceToHost, net->n.st));
  ICS_HIP(hipStreamSynchronize(net->n.st));
  return 0;
}
int ics_net_get_grad(ics_net* net, const char* name, float* host, size_t count) {
  Tensor* t;
  ICS_TRY(find_tensor(net, name, &t));
  ICS_CHECK(t->trainable, "tensor is not trainable");
  ICS_CHECK(count == t->count, std::string("size mismatch for tensor ") + name);
  ICS_HIP(hipMemcpyAsync(host, net->n.G + t->off, count * sizeof(float), hipMemcpyDeviceToHost, net->n.st));
  ICS_HIP(hipStreamSynchronize(net->n.st));
  return 0;
}
int ics_net_get_activation(ics_net* net, const char* layer, float* host, size_t count) {
  ICS_CHECK(net && layer && host, "null argument");
  Net& n = net->n;
  // "name" = the stored pre-BatchNorm activation s; parity / diagnosis aids: "name:dy" (gradient w.r.t. s of the last backward
  // pass), "name:dA" (gradient w.r.t. the layer's input, [rows][padded Cin]), "name:pooled" (the max-pooled BatchNorm output)
  std::string lname = layer, what;
  const size_t colon = lname.find(':');
  if (colon != std::string::npos) { what = lname.substr(colon + 1); lname = lname.substr(0, colon); }
  for (auto& Lp : n.layers) {
    if (Lp->name != lname) continue;
    const float* src = Lp->s;
    size_t cnt = n.rows(*Lp, n.last_batch) * Lp->Cout;
    if (what == "dy") src = Lp->dy;
    else if (what == "dA") { src = Lp->dA; cnt = n.rows(*Lp, n.last_batch) * Lp->CinG; }
    else if (what == "pooled") { src = Lp->pooled; cnt = n.rows(*Lp, n.last_batch) / 8 * Lp->Cout; }
    else ICS_CHECK(what.empty(), std::string("unknown activation kind: ") + what);
    ICS_CHECK(src != nullptr, std::string("layer keeps no such buffer: ") + layer);
    ICS_CHECK(cnt == count, std::string("size mismatch for activation ") + layer);
    ICS_HIP(hipMemcpyAsync(host, src, cnt * sizeof(float), hipMemcpyDeviceToHost, n.st));
    ICS_HIP(hipStreamSynchronize(n.st));
    return 0;
  }
  set_error(std::string("unknown layer: ") + layer);
  return -1;
}
// diagnosis aid: with ICSG3D_DEBUG_CANARY=1 at creation, the number of buffers whose guard bytes were overwritten; the first
// few are described in ics_last_error() (allocation index, payload bytes, first dirty offset)
int ics_net_check_canaries(ics_net* net, int* dirty) {
  ICS_CHECK(net && dirty, "null argument");
  Net& n = net->n;
  ICS_HIP(hipStreamSynchronize(n.st));
  std::string msg;
  *dirty = n.canaries_dirty(&msg);
  if (*dirty) set_error(msg);
  return 0;
}

int ics_net_get_bn_affine(ics_net* net, const char* layer, float* scale, float* shift, size_t count) {
  ICS_CHECK(net && layer && scale && shift, "null argument");
  Net& n = net->n;
  for (auto& Lp : n.layers) {
    if (Lp->name != layer) continue;
    ICS_CHECK(Lp->has_bn, "layer has no BatchNorm");
    ICS_CHECK(count == (size_t)Lp->Cout, std::string("size mismatch for layer ") + layer);
    ICS_HIP(hipMemcpyAsync(scale, Lp->scale, count * sizeof(float), hipMemcpyDeviceToHost, n.st));
    ICS_HIP(hipMemcpyAsync(shift, Lp->shift, count * sizeof(float), hipMemcpyDeviceToHost, n.st));
    ICS_HIP(hipStreamSynchronize(n.st));
    return 0;
  }
  set_error(std::string("unknown layer: ") + layer);
  return -1;
}
int ics_net_set_lr(ics_net* net, float lr) {
  ICS_CHECK(net, "null handle");
  net->n.lr = lr;
  return 0;
}
int ics_net_reset_optimizer(ics_net* net) {
  ICS_CHECK(net, "null handle");
  Net& n = net->n;
  n.adam_t = 0;
  ICS_HIP(hipMemsetAsync(n.Mo, 0, n.nparams * sizeof(float), n.st));
  ICS_HIP(hipMemsetAsync(n.Vo, 0, n.nparams * sizeof(float), n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  return 0;
}

int ics_net_get_optimizer_state(ics_net* net, float* m, float* v, size_t count, int* step) {
  ICS_CHECK(net, "null handle");
  Net& n = net->n;
  ICS_CHECK(count == n.nparams, "optimizer state size mismatch");
  if (m) ICS_HIP(hipMemcpyAsync(m, n.Mo, count * sizeof(float), hipMemcpyDeviceToHost, n.st));
  if (v) ICS_HIP(hipMemcpyAsync(v, n.Vo, count * sizeof(float), hipMemcpyDeviceToHost, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  if (step) *step = n.adam_t;
  return 0;
}
int ics_net_set_optimizer_state(ics_net* net, const float* m, const float* v, size_t count, int step) {
  ICS_CHECK(net && m && v, "null argument");
  Net& n = net->n;
  ICS_CHECK(count == n.nparams && step >= 0, "optimizer state size mismatch");
  ICS_HIP(hipMemcpyAsync(n.Mo, m, count * sizeof(float), hipMemcpyHostToDevice, n.st));
  ICS_HIP(hipMemcpyAsync(n.Vo, v, count * sizeof(float), hipMemcpyHostToDevice, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  n.adam_t = step;
  return 0;
}
int ics_net_num_params(ics_net* net, size_t* count) {
  ICS_CHECK(net && count, "null argument");
  *count = net->n.nparams;
  return 0;
}

int ics_net_profile_enable(ics_net* net, int on) {
  ICS_CHECK(net, "null handle");
  ICS_HIP(hipStreamSynchronize(net->n.st));
  net->n.prof.reset();
  net->n.prof.on = on != 0;
  return 0;
}
int ics_net_profile_filter(ics_net* net, const char* prefix) {
  ICS_CHECK(net, "null handle");
  net->n.prof.filter = prefix ? prefix : "";
  return 0;
}
int ics_net_profile_count(ics_net* net, int* rows) {
  ICS_CHECK(net && rows, "null argument");
  ICS_HIP(hipStreamSynchronize(net->n.st));
  net->n.prof.resolve();
  *rows = (int)net->n.prof.rows.size();
  return 0;
}
int ics_net_profile_row(ics_net* net, int row, const char** label, int64_t* launches, double* ms, double* flop,
                        double* bytes) {
  ICS_CHECK(net && row >= 0 && row < (int)net->n.prof.rows.size(), "profile row out of range");
  const auto& r = net->n.prof.rows[row];
  if (label) *label = r.label.c_str();
  if (launches) *launches = r.launches;
  if (ms) *ms = r.ms;
  if (flop) *flop = r.flop;
  if (bytes) *bytes = r.bytes;
  return 0;
}

// ---------------------------------------------------------------- U-Net entry points
static int require_kind(ics_net* net, int kind) {
  ICS_CHECK(net, "null handle");
  ICS_CHECK(net->n.kind == kind, kind == 0 ? "handle is not a U-Net engine" : "handle is not a VAE engine");
  ICS_HIP(hipSetDevice(net->n.device));
  return 0;
}

int ics_unet_predict(ics_net* net, const float* x, int batch, float* soft, float* sig) {
  ICS_TRY(require_kind(net, 0));
  Net& n = net->n;
  ICS_TRY(unet_upload(n, x, nullptr, batch));
  ICS_TRY(unet_forward_trunk(n, batch, false, false, false, n.x_in));
  ICS_TRY(unet_head_loss(n, batch, 0, 0));
  const size_t M = n.rows(*n.head, batch);
  const int ld = n.ncls + 1;
  float* stage = n.head->dy;   // [M][ncls+1] scratch: soft packed first, sig after it
  if (soft) {
    const size_t cnt = M * (size_t)n.ncls;
    ICS_LAUNCH(gather_cols_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, n.st, n.head->s, ld, 0,
                       n.ncls, M, stage);
    ICS_HIP(hipGetLastError());
    ICS_HIP(hipMemcpyAsync(soft, stage, cnt * sizeof(float), hipMemcpyDeviceToHost, n.st));
  }
  if (sig) {
    ICS_LAUNCH(gather_cols_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, n.st, n.head->s, ld, n.ncls,
                       1, M, stage + M * (size_t)n.ncls);
    ICS_HIP(hipGetLastError());
    ICS_HIP(hipMemcpyAsync(sig, stage + M * (size_t)n.ncls, M * sizeof(float), hipMemcpyDeviceToHost, n.st));
  }
  ICS_HIP(hipStreamSynchronize(n.st));
  n.prof.resolve();
  return 0;
}

int ics_unet_predict_labels(ics_net* net, const float* x, int batch, float thresh, uint8_t* species, uint8_t* mask) {
  ICS_TRY(require_kind(net, 0));
  Net& n = net->n;
  ICS_TRY(unet_upload(n, x, nullptr, batch));
  ICS_TRY(unet_forward_trunk(n, batch, false, false, false, n.x_in));
  const size_t M = n.rows(*n.head, batch);
  unsigned char* d_species = reinterpret_cast<unsigned char*>(n.head->dy);   // [M][ncls+1] float scratch
  unsigned char* d_mask = d_species + M;
  ICS_TRY(unet_head_labels(n, batch, thresh, d_species, d_mask, n.st));
  if (species) ICS_HIP(hipMemcpyAsync(species, d_species, M, hipMemcpyDeviceToHost, n.st));
  if (mask) ICS_HIP(hipMemcpyAsync(mask, d_mask, M, hipMemcpyDeviceToHost, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  return 0;
}

// Benchmark path of model.predict (BASELINE configs[0]): the forward pass on the batch ics_unet_upload_batch left in HBM,
// enqueued without a host round trip; the outputs stay on the device (probabilities [M][ncls+1] in the head's buffer, or
// -- labels_only -- the uint8 argmax / threshold volumes of generate.py:221-225 in its scratch).
int ics_unet_predict_resident(ics_net* net, int labels_only, float thresh) {
  ICS_TRY(require_kind(net, 0));
  Net& n = net->n;
  ICS_CHECK(n.resident_batch > 0, "no resident batch: call ics_unet_upload_batch first");
  const int batch = n.resident_batch;
  ICS_TRY(unet_forward_trunk(n, batch, false, false, false, n.x_in));
  if (labels_only) {
    const size_t M = n.rows(*n.head, batch);
    unsigned char* d_species = reinterpret_cast<unsigned char*>(n.head->dy);
    return unet_head_labels(n, batch, thresh, d_species, d_species + M, n.st);
  }
  return unet_head_loss(n, batch, 0, 0);
}

// Two engines of one process on ONE stream (joint U-Net + DFC-VAE training on one GPU): `net` gives up its own stream and
// enqueues on `other`'s from now on, so the two steps alternate in program order -- no events, no host waits.  (Measured
// alternatives, U-Net step 27.9 + DFC-VAE step 5.4 ms: two free-running streams on their own hardware queues share the chip
// and BOTH slow down, 35.0 ms per pair; two streams chained with hipStreamWaitEvent in both directions, 40.9 ms -- the
// cross-queue waits cost milliseconds per iteration.)  `other` must outlive `net`'s use of it.
int ics_net_share_stream(ics_net* net, ics_net* other) {
  ICS_CHECK(net && other && net != other, "need two different handles");
  Net& a = net->n;
  Net& b = other->n;
  ICS_CHECK(a.device == b.device, "engines on different devices");
  ICS_HIP(hipSetDevice(a.device));
  ICS_HIP(hipStreamSynchronize(a.st));
  ICS_HIP(hipStreamSynchronize(b.st));
  if (!a.st_own) a.st_own = a.st;       // kept for the destructor
  a.st = b.st;
  return 0;
}

// Device-clock bracket on the engine's stream (bench.py's gpu_active_s): start records an event, stop records a second
// one, waits for it and returns the milliseconds between the two as the GPU saw them.
int ics_net_timer_start(ics_net* net) {
  ICS_CHECK(net, "null handle");
  Net& n = net->n;
  ICS_HIP(hipSetDevice(n.device));
  if (!n.timer_ev[0]) { ICS_HIP(hipEventCreate(&n.timer_ev[0])); ICS_HIP(hipEventCreate(&n.timer_ev[1])); }
  ICS_HIP(hipEventRecord(n.timer_ev[0], n.st));
  return 0;
}
int ics_net_timer_stop(ics_net* net, double* ms) {
  ICS_CHECK(net && ms, "null argument");
  Net& n = net->n;
  ICS_CHECK(n.timer_ev[0] != nullptr, "ics_net_timer_stop without ics_net_timer_start");
  ICS_HIP(hipEventRecord(n.timer_ev[1], n.st));
  ICS_HIP(hipEventSynchronize(n.timer_ev[1]));
  float f = 0.f;
  ICS_HIP(hipEventElapsedTime(&f, n.timer_ev[0], n.timer_ev[1]));
  *ms = (double)f;
  return 0;
}

// Measurement aid (round 6): the resident train step of `net` eagerly and as a replayed hipGraph.
// The step is captured from the engine's stream (its second stream joins the capture through the fork / join events the
// step records anyway) and replayed `iters` times; both forms are timed with events on the engine's stream.  The replay
// re-runs the CAPTURED launches: Adam's bias-corrected step size is a launch argument computed on the host, so a replayed
// step repeats the captured step's lr_t -- good enough to time, not to train; a product path would keep the step count in
// device memory.  Not available with a communicator (RCCL calls are not captured here) or while profiling.
int ics_net_graph_probe(ics_net* net, int iters, double* eager_ms, double* graph_ms, int* graph_nodes) {
  ICS_CHECK(net && eager_ms && graph_ms && iters >= 1, "bad arguments");
  Net& n = net->n;
  ICS_CHECK(n.resident_batch > 0, "no resident batch: upload one first");
  ICS_CHECK(n.comm == nullptr && !n.prof.on, "graph probe: no communicator, no profiler");
  const int B = n.resident_batch;
  auto step = [&]() { return n.kind == 1 ? vae_step(n, B, true, nullptr) : unet_train_resident(n, B, nullptr); };
  hipEvent_t e0, e1;
  ICS_HIP(hipEventCreate(&e0)); ICS_HIP(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) ICS_TRY(step());
  ICS_HIP(hipStreamSynchronize(n.st));
  ICS_HIP(hipEventRecord(e0, n.st));
  for (int i = 0; i < iters; ++i) ICS_TRY(step());
  ICS_HIP(hipEventRecord(e1, n.st));
  ICS_HIP(hipEventSynchronize(e1));
  float f = 0.f;
  ICS_HIP(hipEventElapsedTime(&f, e0, e1));
  *eager_ms = (double)f / iters;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  ICS_HIP(hipStreamBeginCapture(n.st, hipStreamCaptureModeRelaxed));
  const int rc = step();
  const hipError_t ec = hipStreamEndCapture(n.st, &graph);
  ICS_TRY(rc);
  ICS_CHECK(ec == hipSuccess && graph != nullptr, std::string("hipStreamEndCapture: ") + hipGetErrorString(ec));
  size_t nodes = 0;
  (void)hipGraphGetNodes(graph, nullptr, &nodes);
  if (graph_nodes) *graph_nodes = (int)nodes;
  ICS_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) ICS_HIP(hipGraphLaunch(exec, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  ICS_HIP(hipEventRecord(e0, n.st));
  for (int i = 0; i < iters; ++i) ICS_HIP(hipGraphLaunch(exec, n.st));
  ICS_HIP(hipEventRecord(e1, n.st));
  ICS_HIP(hipEventSynchronize(e1));
  ICS_HIP(hipEventElapsedTime(&f, e0, e1));
  *graph_ms = (double)f / iters;
  (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  n.adam_t += 6 + 2 * iters;      // keep the host's step count in line with the updates the replays applied
  return 0;
}

int ics_unet_upload_batch(ics_net* net, const float* x, const uint8_t* labels, int batch) {
  ICS_TRY(require_kind(net, 0));
  ICS_TRY(unet_upload(net->n, x, labels, batch));
  ICS_HIP(hipStreamSynchronize(net->n.st));
  return 0;
}
int ics_unet_train_step_resident(ics_net* net, float* metrics) {
  ICS_TRY(require_kind(net, 0));
  ICS_CHECK(net->n.resident_batch > 0, "no resident batch: call ics_unet_upload_batch first");
  return unet_train_resident(net->n, net->n.resident_batch, metrics);
}
int ics_unet_train_step(ics_net* net, const float* x, const uint8_t* labels, int batch, float metrics[5]) {
  ICS_TRY(require_kind(net, 0));
  ICS_CHECK(x && labels && metrics, "null argument");
  ICS_TRY(unet_upload(net->n, x, labels, batch));
  return unet_train_resident(net->n, batch, metrics);
}
int ics_unet_test_step(ics_net* net, const float* x, const uint8_t* labels, int batch, float metrics[5]) {
  ICS_TRY(require_kind(net, 0));
  ICS_CHECK(x && labels && metrics, "null argument");
  Net& n = net->n;
  ICS_TRY(unet_upload(n, x, labels, batch));
  ICS_TRY(unet_forward_trunk(n, batch, false, false, false, n.x_in));
  ICS_TRY(unet_head_loss(n, batch, 1, 0));
  ICS_HIP(hipMemcpyAsync(metrics, n.d_metrics, 5 * sizeof(float), hipMemcpyDeviceToHost, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  return 0;
}

int ics_unet_metric_sums(ics_net* net, double sums[7]) {
  ICS_TRY(require_kind(net, 0));
  ICS_CHECK(sums, "null argument");
  Net& n = net->n;
  ICS_HIP(hipMemcpyAsync(sums, n.d_red + 8, 7 * sizeof(double), hipMemcpyDeviceToHost, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  return 0;
}

// ---------------------------------------------------------------- VAE entry points
int ics_vae_encode(ics_net* net, const float* x, const float* cond, const float* eps, int batch, float* z_mean,
                   float* z_log_var, float* z) {
  ICS_TRY(require_kind(net, 1));
  Net& n = net->n;
  ICS_CHECK(x && cond && eps, "null argument");
  ICS_TRY(vae_upload(n, x, cond, eps, batch));
  ICS_TRY(vae_encode_fwd(n, batch, false));
  VaeRefs r = vae_refs(n);
  const size_t lat = n.latent;
  if (z_mean) ICS_HIP(hipMemcpy2DAsync(z_mean, lat * 4, r.zml->s, 2 * lat * 4, lat * 4, batch, hipMemcpyDeviceToHost, n.st));
  if (z_log_var) ICS_HIP(hipMemcpy2DAsync(z_log_var, lat * 4, r.zml->s + lat, 2 * lat * 4, lat * 4, batch, hipMemcpyDeviceToHost, n.st));
  if (z) ICS_HIP(hipMemcpyAsync(z, n.z_buf, batch * lat * 4, hipMemcpyDeviceToHost, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  return 0;
}
int ics_vae_decode(ics_net* net, const float* z, const float* cond, int batch, float* recon) {
  ICS_TRY(require_kind(net, 1));
  Net& n = net->n;
  ICS_CHECK(z && cond && recon, "null argument");
  ICS_CHECK(batch >= 1 && batch <= n.maxB, "batch exceeds max_batch");
  // zc = [z | cond] assembled on the host side of the ABI: two strided uploads
  const size_t lat = n.latent, W = lat + n.ncond;
  ICS_HIP(hipMemcpy2DAsync(n.zc, W * 4, z, lat * 4, lat * 4, batch, hipMemcpyHostToDevice, n.st));
  ICS_HIP(hipMemcpy2DAsync(n.zc + lat, W * 4, cond, n.ncond * 4, n.ncond * 4, batch, hipMemcpyHostToDevice, n.st));
  ICS_TRY(vae_decode_fwd(n, batch, false));
  const size_t cnt = (size_t)batch * n.d * n.d * n.d * n.C;
  ICS_HIP(hipMemcpyAsync(recon, n.recon, cnt * 4, hipMemcpyDeviceToHost, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  return 0;
}
int ics_vae_upload_batch(ics_net* net, const float* x, const float* cond, const float* eps, int batch) {
  ICS_TRY(require_kind(net, 1));
  ICS_TRY(vae_upload(net->n, x, cond, eps, batch));
  ICS_HIP(hipStreamSynchronize(net->n.st));
  return 0;
}
int ics_vae_train_step_resident(ics_net* net, float* metrics) {
  ICS_TRY(require_kind(net, 1));
  ICS_CHECK(net->n.resident_batch > 0, "no resident batch: call ics_vae_upload_batch first");
  return vae_step(net->n, net->n.resident_batch, true, metrics);
}
int ics_vae_train_step(ics_net* net, const float* x, const float* cond, const float* eps, int batch, float metrics[4]) {
  ICS_TRY(require_kind(net, 1));
  ICS_CHECK(x && cond && eps && metrics, "null argument");
  ICS_TRY(vae_upload(net->n, x, cond, eps, batch));
  return vae_step(net->n, batch, true, metrics);
}
int ics_vae_test_step(ics_net* net, const float* x, const float* cond, const float* eps, int batch, float metrics[4]) {
  ICS_TRY(require_kind(net, 1));
  ICS_CHECK(x && cond && eps && metrics, "null argument");
  ICS_TRY(vae_upload(net->n, x, cond, eps, batch));
  return vae_step(net->n, batch, false, metrics);
}

// Fused inference tail of generate.py:204-225 / eval.py:163-175: decoder.predict -> unet.model.predict ->
// argmax / threshold, all on the device.  The reconstruction never leaves HBM: the U-Net reads it where the
// decoder wrote it (driven on the VAE's stream), and what comes back is 2 bytes per voxel plus, on request,
// the density channel (watershed input) and the coordinate channels' min/max (to_lattice_params).
// device part of the tail: decoder -> U-Net -> labels; leaves species / mask (uint8 [M] each) and aux (density [M] |
// minmax [B][3][2]) in the U-Net head's gradient buffer, which no inference path reads
static int decode_to_labels_device(Net& n, Net& u, const float* z, const float* cond, int batch, float thresh,
                                   bool want_density, bool want_minmax, unsigned char** d_species_out,
                                   unsigned char** d_mask_out, float** d_aux_out) {
  ICS_CHECK(z && cond, "null argument");
  ICS_CHECK(u.d == n.d && u.C == n.C && u.device == n.device, "U-Net / VAE engines do not match (grid, channels, device)");
  ICS_CHECK(batch >= 1 && batch <= n.maxB && batch <= u.maxB, "batch exceeds max_batch");
  const size_t lat = n.latent, W = lat + n.ncond;
  ICS_HIP(hipStreamSynchronize(u.st));
  ICS_HIP(hipMemcpy2DAsync(n.zc, W * 4, z, lat * 4, lat * 4, batch, hipMemcpyHostToDevice, n.st));
  ICS_HIP(hipMemcpy2DAsync(n.zc + lat, W * 4, cond, n.ncond * 4, n.ncond * 4, batch, hipMemcpyHostToDevice, n.st));
  ICS_TRY(vae_decode_fwd(n, batch, false));
  hipStream_t saved = u.st;
  u.st = n.st;
  int rc = 0;
  const size_t M = (size_t)batch * n.d * n.d * n.d;
  unsigned char* d_species = reinterpret_cast<unsigned char*>(u.head->dy);   // [M][ncls+1] float scratch
  unsigned char* d_mask = d_species + M;
  float* d_aux = reinterpret_cast<float*>(d_mask + M);                       // density [M] | minmax [B][3][2]
  do {
    if ((rc = unet_forward_trunk(u, batch, false, false, false, n.recon))) break;
    if ((rc = unet_head_labels(u, batch, thresh, d_species, d_mask, n.st))) break;
    if (want_density) {
      ICS_LAUNCH(gather_cols_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, n.st, n.recon, n.C, 0, 1,
                         M, d_aux);
    }
    if (want_minmax && n.C >= 4) {
      ICS_LAUNCH(chan_minmax_kernel, dim3(batch * 3), dim3(256), 0, n.st, n.recon, M / batch, n.C, 1, 3,
                         d_aux + M);
    }
    if (hipGetLastError() != hipSuccess) { set_error("inference tail launch failed"); rc = -1; break; }
  } while (0);
  u.st = saved;
  ICS_TRY(rc);
  *d_species_out = d_species; *d_mask_out = d_mask; *d_aux_out = d_aux;
  return 0;
}

static int tail_copy_out(Net& n, int batch, const unsigned char* d_species, const unsigned char* d_mask,
                         const float* d_aux, uint8_t* species, uint8_t* mask, float* density, float* coord_minmax) {
  const size_t M = (size_t)batch * n.d * n.d * n.d;
  if (species) ICS_HIP(hipMemcpyAsync(species, d_species, M, hipMemcpyDeviceToHost, n.st));
  if (mask) ICS_HIP(hipMemcpyAsync(mask, d_mask, M, hipMemcpyDeviceToHost, n.st));
  if (density) ICS_HIP(hipMemcpyAsync(density, d_aux, M * 4, hipMemcpyDeviceToHost, n.st));
  if (coord_minmax) {
    if (n.C >= 4) ICS_HIP(hipMemcpyAsync(coord_minmax, d_aux + M, (size_t)batch * 6 * 4, hipMemcpyDeviceToHost, n.st));
    else std::memset(coord_minmax, 0, (size_t)batch * 6 * 4);
  }
  return 0;
}

int ics_vae_decode_to_unet_labels(ics_net* vae, ics_net* unet, const float* z, const float* cond, int batch,
                                  float thresh, uint8_t* species, uint8_t* mask, float* density,
                                  float* coord_minmax) {
  ICS_TRY(require_kind(vae, 1));
  ICS_TRY(require_kind(unet, 0));
  Net& n = vae->n;
  unsigned char *d_species, *d_mask;
  float* d_aux;
  ICS_TRY(decode_to_labels_device(n, unet->n, z, cond, batch, thresh, density != nullptr, coord_minmax != nullptr,
                                  &d_species, &d_mask, &d_aux));
  ICS_TRY(tail_copy_out(n, batch, d_species, d_mask, d_aux, species, mask, density, coord_minmax));
  ICS_HIP(hipStreamSynchronize(n.st));
  return 0;
}

// connected components + region statistics of device-resident mask / species volumes; results to the host
static int segment_to_host(Net& n, const unsigned char* d_mask, const unsigned char* d_species, int batch, int d,
                           int min_voxels, int max_atoms, int nbins, int32_t* regions, int32_t* counts,
                           int32_t* atom_stats, int64_t* bounds) {
  ICS_CHECK(counts && atom_stats && max_atoms >= 1, "null argument");
  const size_t M = (size_t)batch * d * d * d;
  const size_t wsb = segment_workspace_bytes(batch, d, max_atoms, nbins);
  unsigned char* ws = nullptr;
  int* d_R = nullptr;
  const size_t wsb16 = (wsb + 15) & ~(size_t)15;
  ICS_HIP(hipMalloc(reinterpret_cast<void**>(&ws), wsb16 + (regions ? M * 4 : 0)));
  int rc = 0;
  do {
    if (regions) d_R = reinterpret_cast<int*>(ws + wsb16);
    int *d_counts = nullptr, *d_stats = nullptr;
    long long* d_bounds = nullptr;
    if ((rc = launch_segment_atoms(n.st, d_mask, d_species, batch, d, min_voxels, max_atoms, nbins, ws, wsb, d_R,
                                   &d_counts, &d_stats, bounds ? &d_bounds : nullptr)))
      break;
    hipError_t e = hipSuccess;
    if (regions) e = hipMemcpyAsync(regions, d_R, M * 4, hipMemcpyDeviceToHost, n.st);
    if (e == hipSuccess) e = hipMemcpyAsync(counts, d_counts, (size_t)batch * 2 * 4, hipMemcpyDeviceToHost, n.st);
    if (e == hipSuccess)
      e = hipMemcpyAsync(atom_stats, d_stats, (size_t)batch * max_atoms * kSegStatInts * 4, hipMemcpyDeviceToHost, n.st);
    if (e == hipSuccess && bounds)
      e = hipMemcpyAsync(bounds, d_bounds, (size_t)batch * max_atoms * kSegBoundInts * 8, hipMemcpyDeviceToHost, n.st);
    if (e == hipSuccess) e = hipStreamSynchronize(n.st);
    if (e != hipSuccess) { set_error(std::string("segmentation copy-out: ") + hipGetErrorString(e)); rc = -1; }
  } while (0);
  (void)hipFree(ws);
  ICS_TRY(rc);
  // counts[2 b + 1] > max_atoms marks sample b as FAILED (its statistics rows are truncated): the caller skips that
  // sample, as the reference does ("Failed", continue: generate.py:228-236) -- one bad sample does not fail the batch
  return 0;
}

int ics_vae_decode_to_unet_atoms(ics_net* vae, ics_net* unet, const float* z, const float* cond, int batch,
                                 float thresh, int min_voxels, int max_atoms, uint8_t* species, uint8_t* mask,
                                 float* density, float* coord_minmax, int32_t* regions, int32_t* counts,
                                 int32_t* atom_stats, int64_t* convexity_bounds) {
  ICS_TRY(require_kind(vae, 1));
  ICS_TRY(require_kind(unet, 0));
  Net& n = vae->n;
  unsigned char *d_species, *d_mask;
  float* d_aux;
  ICS_TRY(decode_to_labels_device(n, unet->n, z, cond, batch, thresh, density != nullptr, coord_minmax != nullptr,
                                  &d_species, &d_mask, &d_aux));
  ICS_TRY(tail_copy_out(n, batch, d_species, d_mask, d_aux, species, mask, density, coord_minmax));
  ICS_TRY(segment_to_host(n, d_mask, d_species, batch, n.d, min_voxels, max_atoms, unet->n.ncls, regions, counts,
                          atom_stats, convexity_bounds));
  return 0;
}

int ics_op_segment_atoms(const uint8_t* mask, const uint8_t* species, int batch, int d, int min_voxels, int max_atoms,
                         int num_species, int32_t* regions, int32_t* counts, int32_t* atom_stats, int64_t* convexity_bounds) {
  ICS_CHECK(mask && species && batch >= 1 && max_atoms >= 1 && min_voxels >= 0, "bad segmentation arguments");
  ICS_CHECK(d >= 16 && d <= 256 && (d & (d - 1)) == 0, "grid must be a power of two in [16, 256]");
  Net n;
  n.arena_next = (size_t)1 << 22;      // a temporary Net: see op_prepare
  ICS_TRY(net_common_init(n));
  const size_t M = (size_t)batch * d * d * d;
  unsigned char *dm, *ds;
  ICS_TRY(n.alloc(&dm, M)); ICS_TRY(n.alloc(&ds, M));
  ICS_HIP(hipMemcpyAsync(dm, mask, M, hipMemcpyHostToDevice, n.st));
  ICS_HIP(hipMemcpyAsync(ds, species, M, hipMemcpyHostToDevice, n.st));
  return segment_to_host(n, dm, ds, batch, d, min_voxels, max_atoms, num_species, regions, counts, atom_stats, convexity_bounds);
}

// The three box-level entry points run tens of times per sample from the host recursion of segment_nuclei
// (icsg3d_amd/watershed.py): one stream per host thread, created on first use and kept (a handle with two streams, two
// events and its own allocations per call cost 10 - 16 ms against kernels of tens of microseconds).
namespace {
struct OpStream { hipStream_t st = nullptr; int device = -1; };
// as segment.hip's scratch: a thread that exits hands its stream to a list of orphans instead of leaking it (or calling into
// a runtime that may be shutting down); the next op_stream() creation and ics_release_caches destroy them
std::mutex g_stream_orphan_mu;
std::vector<hipStream_t> g_stream_orphans;
void reap_stream_orphans() {
  std::vector<hipStream_t> take;
  { std::lock_guard<std::mutex> lk(g_stream_orphan_mu); take.swap(g_stream_orphans); }
  for (hipStream_t s : take) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
}
struct OpStreamTL {
  OpStream o;
  ~OpStreamTL() {
    if (o.st) { std::lock_guard<std::mutex> lk(g_stream_orphan_mu); g_stream_orphans.push_back(o.st); }
  }
};
thread_local OpStreamTL tl_op_stream_holder;
#define tl_op_stream (tl_op_stream_holder.o)
int op_stream(hipStream_t* out) {
  int dev = 0;
  ICS_HIP(hipGetDevice(&dev));
  if (tl_op_stream.st == nullptr || tl_op_stream.device != dev) {
    reap_stream_orphans();
    if (tl_op_stream.st) (void)hipStreamDestroy(tl_op_stream.st);
    tl_op_stream.st = nullptr;
    ICS_HIP(hipStreamCreateWithFlags(&tl_op_stream.st, hipStreamNonBlocking));
    tl_op_stream.device = dev;
  }
  *out = tl_op_stream.st;
  return 0;
}
}  // namespace
int ics_release_caches() {
  if (tl_op_stream.st) { (void)hipStreamSynchronize(tl_op_stream.st); (void)hipStreamDestroy(tl_op_stream.st); }
  tl_op_stream = OpStream{};
  reap_stream_orphans();
  segment_release_scratch();
  return 0;
}
#undef tl_op_stream

int ics_op_label_boxes(const int32_t* vols, const int32_t* dims, int nbox, int connectivity, int max_labels,
                       int32_t* labels, int32_t* nlabels, int32_t* stats) {
  hipStream_t st;
  ICS_TRY(op_stream(&st));
  return segment_label_boxes(st, vols, dims, nbox, connectivity, max_labels, labels, nlabels, stats);
}
int ics_op_region_stats(const int32_t* R, const uint8_t* species, int D, int H, int W, int num_labels, int num_species,
                        int32_t* stats) {
  hipStream_t st;
  ICS_TRY(op_stream(&st));
  return segment_region_stats(st, R, species, D, H, W, num_labels, num_species, stats);
}
int ics_op_component_bounds(const int32_t* labels, const int32_t* dims, int nbox, const int32_t* nlabels, const int32_t* stats,
                            int max_labels, int min_voxels, double hull_threshold, int64_t* bounds) {
  static_assert(sizeof(long long) == sizeof(int64_t), "int64 layout");
  return segment_component_bounds(labels, dims, nbox, nlabels, stats, max_labels, min_voxels, hull_threshold,
                                  reinterpret_cast<long long*>(bounds));
}
int ics_op_watershed_split(const int32_t* boxes, const int32_t* dims, const int32_t* cls, int nbox, int tie, int32_t* wss) {
  // Round 6: the boxes of a level go to host threads (segment_watershed_split_host: the flood is sequential per box and a
  // CPU core walks it ~30x faster than one GPU lane); ICSG3D_WS_DEVICE=1 keeps the kernel (A/B, tests run both).
  const char* env = getenv("ICSG3D_WS_DEVICE");       // read per call: tests flip it inside one process
  const bool on_device = env != nullptr && atoi(env) != 0;
  if (!on_device) return segment_watershed_split_host(boxes, dims, cls, nbox, tie, wss);
  hipStream_t st;
  ICS_TRY(op_stream(&st));
  return segment_watershed_split(st, boxes, dims, cls, nbox, tie, wss);
}

// ---------------------------------------------------------------- data parallel
int ics_comm_unique_id(char uid[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  ICS_CHECK(r == ncclSuccess, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
  std::memcpy(uid, &id, 128);
  return 0;
}
int ics_net_comm_init(ics_net* net, int rank, int nranks, const char uid[128]) {
  ICS_CHECK(net && uid && nranks >= 1 && rank >= 0 && rank < nranks, "bad comm arguments");
  Net& n = net->n;
  ICS_CHECK(n.comm == nullptr, "communicator already initialised");
  ICS_HIP(hipSetDevice(n.device));
  ncclUniqueId id;
  std::memcpy(&id, uid, 128);
  ncclResult_t r = ncclCommInitRank(&n.comm, nranks, id, rank);
  ICS_CHECK(r == ncclSuccess, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  n.rank = rank; n.nranks = nranks;
  ICS_HIP(hipStreamCreateWithFlags(&n.comm_st, hipStreamNonBlocking));
  ICS_HIP(hipEventCreateWithFlags(&n.ev_grad, hipEventDisableTiming));
  ICS_HIP(hipEventCreateWithFlags(&n.ev_comm, hipEventDisableTiming));
  n.overlap = getenv("ICSG3D_DP_NO_OVERLAP") == nullptr;
  // ~4 buckets, last-layer-first; small nets (the VAE: 3.4 MB of gradients) go out as one or two messages
  n.bucket_min = std::max<size_t>(n.nparams / 4, (size_t)1 << 18);
  // SyncBN exchange buffers, wide enough for the perceptual U-Net a VAE engine drives
  int cmax = n.sync_cmax;
  if (n.pm) cmax = std::max(cmax, n.pm->sync_cmax);
  cmax = std::max(cmax, 1);
  ICS_TRY(n.alloc(&n.sync_local, (size_t)3 * cmax));
  ICS_TRY(n.alloc(&n.sync_gathered, (size_t)nranks * 3 * cmax));
  // RCCL serialises the operations of ONE communicator whatever streams they are issued on: with the BatchNorm
  // statistics on the gradient communicator every SyncBN collective of the backward pass waited for the bucket in
  // flight (and the next bucket for it), and the overlap was lost.  A second communicator over the same ranks
  // (ncclCommSplit, color 0, key = rank: no second unique id to hand around) removes the coupling.
  // Not fatal if the split fails: the small collectives then share the gradient communicator as before round 4 --
  // correct, only without the overlap.  The ranks AGREE on the outcome (a min all-reduce of the success flag on the
  // gradient communicator): a rank that fell back alone while the others used comm_bn would hang the first collective.
  // ICSG3D_NO_COMM_SPLIT=1 (set it on every rank) is the escape hatch: one communicator, everything serialised on it.
  // Two communicators in flight on one device: every rank issues the same operations in the same host order on the same
  // two streams; an RCCL kernel that waits for its peers holds a few workgroups, the compute kernels ahead of the other
  // stream's collective never wait on anything, so both collectives become resident on every rank and complete.
  const bool want_split = getenv("ICSG3D_NO_COMM_SPLIT") == nullptr;
  ncclResult_t rs = want_split ? ncclCommSplit(n.comm, 0, rank, &n.comm_bn, nullptr) : ncclInvalidUsage;
  double ok = (rs == ncclSuccess && n.comm_bn != nullptr) ? 1.0 : 0.0;
  const double mine = ok;
  ICS_HIP(hipMemcpyAsync(n.d_red, &ok, sizeof(double), hipMemcpyHostToDevice, n.st));
  ncclResult_t ra = ncclAllReduce(n.d_red, n.d_red, 1, ncclDouble, ncclMin, n.comm, n.st);
  ICS_CHECK(ra == ncclSuccess, std::string("ncclAllReduce(split agreement): ") + ncclGetErrorString(ra));
  ICS_HIP(hipMemcpyAsync(&ok, n.d_red, sizeof(double), hipMemcpyDeviceToHost, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  if (ok < 0.5) {
    if (want_split)
      fprintf(stderr, "icsg3d: ncclCommSplit failed on %s (%s): BatchNorm / metric collectives share the gradient communicator "
                      "on every rank\n", mine < 0.5 ? "this rank" : "another rank", ncclGetErrorString(rs));
    if (n.comm_bn) ncclCommDestroy(n.comm_bn);
    n.comm_bn = nullptr;
  }
  n.bn_sync = BnSync{n.small(), nranks, n.sync_local, n.sync_gathered};
  ICS_HIP(hipStreamSynchronize(n.st));
  return 0;
}
/* Replicas must start identical: parameters, BN moving statistics, Adam moments and step count of `root`
 * overwrite every other rank's (the class API draws its initial weights from an unseeded RNG). */
int ics_net_comm_broadcast_state(ics_net* net, int root) {
  ICS_CHECK(net, "null handle");
  Net& n = net->n;
  if (!n.comm) return 0;
  ICS_CHECK(root >= 0 && root < n.nranks, "bad root rank");
  ICS_HIP(hipSetDevice(n.device));
  float* bufs[3] = {n.P, n.Mo, n.Vo};
  for (float* b : bufs) {
    ncclResult_t r = ncclBroadcast(b, b, n.nparams, ncclFloat, root, n.comm, n.st);
    ICS_CHECK(r == ncclSuccess, std::string("ncclBroadcast: ") + ncclGetErrorString(r));
  }
  if (n.bn_slab_n) {
    ncclResult_t r = ncclBroadcast(n.bn_slab, n.bn_slab, n.bn_slab_n, ncclFloat, root, n.comm, n.st);
    ICS_CHECK(r == ncclSuccess, std::string("ncclBroadcast(bn): ") + ncclGetErrorString(r));
  }
  double t = (double)n.adam_t;
  ICS_HIP(hipMemcpyAsync(n.d_red, &t, sizeof(double), hipMemcpyHostToDevice, n.st));
  ncclResult_t r = ncclBroadcast(n.d_red, n.d_red, 1, ncclDouble, root, n.comm, n.st);
  ICS_CHECK(r == ncclSuccess, std::string("ncclBroadcast(t): ") + ncclGetErrorString(r));
  ICS_HIP(hipMemcpyAsync(&t, n.d_red, sizeof(double), hipMemcpyDeviceToHost, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  n.adam_t = (int)t;
  n.packed_valid = false;
  return 0;
}
int ics_net_set_sync_bn(ics_net* net, int on) {
  ICS_CHECK(net, "null handle");
  net->n.sync_bn = on ? 1 : 0;
  return 0;
}
int ics_net_comm_info(ics_net* net, int* rank, int* nranks, int* buckets_last_step) {
  ICS_CHECK(net, "null handle");
  if (rank) *rank = net->n.rank;
  if (nranks) *nranks = net->n.comm ? net->n.nranks : 0;
  if (buckets_last_step) *buckets_last_step = net->n.buckets_issued;
  return 0;
}
int ics_net_comm_allreduce_max(ics_net* net, double* value) {
  ICS_CHECK(net && value, "null argument");
  Net& n = net->n;
  if (!n.comm) { ICS_HIP(hipStreamSynchronize(n.st)); return 0; }
  ICS_HIP(hipMemcpyAsync(n.d_red, value, sizeof(double), hipMemcpyHostToDevice, n.st));
  ncclResult_t r = ncclAllReduce(n.d_red, n.d_red, 1, ncclDouble, ncclMax, n.comm, n.st);
  ICS_CHECK(r == ncclSuccess, std::string("ncclAllReduce(max): ") + ncclGetErrorString(r));
  ICS_HIP(hipMemcpyAsync(value, n.d_red, sizeof(double), hipMemcpyDeviceToHost, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  return 0;
}

// ---------------------------------------------------------------- single-op entry points
// Short-lived Nets behind the single-op entry points: their buffers are a few KB to MB, so the arena starts at 4 MB (doubling
// as usual) instead of the 128 MB first slab of a long-lived engine handle -- a 128 MB hipMalloc + a synchronising hipFree per
// call was the very overhead the box-level ops got rid of in round 5.
static int op_prepare(Net& n) {
  n.arena_next = (size_t)1 << 22;
  return net_common_init(n);
}

int ics_op_conv3d_forward(const float* x, const float* w, const float* bias, int B, int S, int Cin, int Cout,
                          int taps, int pre_act, float* y) {
  ICS_CHECK(x && w && y && (taps == 27 || taps == 1) && S >= 1 && (S & (S - 1)) == 0, "bad conv arguments");
  Net n;
  ICS_TRY(op_prepare(n));
  const size_t M = (size_t)B * S * S * S;
  const int Kpad = round_up(taps * Cin, 32), Npad = round_up(Cout, 32);
  float *dx, *dw, *db, *dwp, *dyv;
  ICS_TRY(n.alloc(&dx, M * Cin)); ICS_TRY(n.alloc(&dw, (size_t)taps * Cin * Cout)); ICS_TRY(n.alloc(&db, (size_t)Cout));
  ICS_TRY(n.alloc(&dwp, (size_t)Kpad * Npad)); ICS_TRY(n.alloc(&dyv, M * Cout));
  ICS_HIP(hipMemcpyAsync(dx, x, M * Cin * 4, hipMemcpyHostToDevice, n.st));
  ICS_HIP(hipMemcpyAsync(dw, w, (size_t)taps * Cin * Cout * 4, hipMemcpyHostToDevice, n.st));
  if (bias) ICS_HIP(hipMemcpyAsync(db, bias, (size_t)Cout * 4, hipMemcpyHostToDevice, n.st));
  ICS_TRY(launch_pack_fwd(n.st, dw, taps * Cin, Cout, dwp, Kpad, Npad, 0, 0, 1));
  ConvGeom g{B, S, ilog2(S), Cin, Cout, taps, Kpad, Npad, n.flags};
  ConvSrc s = src_plain(dx, Cin);
  if (conv_wino_ok(g, &s, 1)) {          // the path the engine takes for this shape (ICSG3D_NO_WINO: the direct kernels)
    float* ww = nullptr;
    ICS_TRY(n.alloc(&ww, conv_wino_weight_floats(Cin, Cout)));
    const int layout = conv_wino_layout(g);
    ICS_TRY(launch_pack_wino(n.st, dw, Cin, Cout, 0, Cin, 0, ww, layout));
    ICS_TRY(launch_conv_fwd_wino(n.st, g, s, ww, bias ? db : nullptr, dyv, Cout, pre_act, nullptr, nullptr, 0, layout));
  } else if (conv_winog_ok(g, &s, 1)) {  // S = 4: Winograd-domain GEMMs (ICSG3D_NO_WINOG: the direct kernels)
    float *wg = nullptr, *sv = nullptr, *sm = nullptr;
    size_t v, m, z;
    conv_winog_scratch_floats(g, &v, &m, &z);
    ICS_TRY(n.alloc(&wg, conv_winog_weight_floats(Cin, Cout))); ICS_TRY(n.alloc(&sv, v)); ICS_TRY(n.alloc(&sm, m));
    ICS_TRY(launch_pack_wino(n.st, dw, Cin, Cout, 0, Cin, 0, wg, 2));
    ICS_TRY(launch_conv_fwd_winog(n.st, g, s, wg, bias ? db : nullptr, dyv, Cout, pre_act, nullptr, nullptr, sv, sm, nullptr));
  } else {
    ICS_TRY(launch_conv_fwd(n.st, g, &s, 1, dwp, bias ? db : nullptr, dyv, Cout, pre_act, nullptr, nullptr));
  }
  ICS_HIP(hipMemcpyAsync(y, dyv, M * Cout * 4, hipMemcpyDeviceToHost, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  return 0;
}

// The two 1x1x1 heads + losses + metrics on a given trunk output (kernel parity tests against the reference's own formulas,
// tests/golden/loss_golden.npz): the same two launch sequences unet_head_loss chooses between.
int ics_op_unet_head(const float* x, const float* wsoft, const float* bsoft, const float* wsig, const float* bsig,
                     const uint8_t* labels, size_t M, int ncls, float loss_weight, int mode, int fused, float* out,
                     float metrics[5], double sums[7]) {
  const int bce = ((mode >> 2) & 1) << 1;
  mode &= 3;
  ICS_CHECK(x && wsoft && bsoft && wsig && bsig && labels && M >= 1 && ncls >= 2 && ncls <= 128 && mode >= 0 && mode <= 2,
            "bad head arguments");
  Net n;
  ICS_TRY(op_prepare(n));
  const int nz = ncls + 1, Kpad = 128, Npad = round_up(nz, 32);
  float *dx, *dws, *dwg, *db, *dwp, *dz, *dmet, *dcol;
  double *dpart, *dsum;
  unsigned char* dlab;
  ICS_TRY(n.alloc(&dx, M * 128)); ICS_TRY(n.alloc(&dws, (size_t)128 * ncls)); ICS_TRY(n.alloc(&dwg, (size_t)128));
  ICS_TRY(n.alloc(&db, (size_t)nz)); ICS_TRY(n.alloc(&dwp, (size_t)Kpad * Npad)); ICS_TRY(n.alloc(&dz, M * nz));
  ICS_TRY(n.alloc(&dmet, (size_t)8)); ICS_TRY(n.alloc(&dcol, (size_t)2048 * (nz + 4))); ICS_TRY(n.alloc(&dpart, (size_t)2048 * 6));
  ICS_TRY(n.alloc(&dsum, (size_t)8)); ICS_TRY(n.alloc(&dlab, M));
  ICS_HIP(hipMemcpyAsync(dx, x, M * 128 * 4, hipMemcpyHostToDevice, n.st));
  ICS_HIP(hipMemcpyAsync(dws, wsoft, (size_t)128 * ncls * 4, hipMemcpyHostToDevice, n.st));
  ICS_HIP(hipMemcpyAsync(dwg, wsig, (size_t)128 * 4, hipMemcpyHostToDevice, n.st));
  ICS_HIP(hipMemcpyAsync(db, bsoft, (size_t)ncls * 4, hipMemcpyHostToDevice, n.st));
  ICS_HIP(hipMemcpyAsync(db + ncls, bsig, 4, hipMemcpyHostToDevice, n.st));
  ICS_HIP(hipMemcpyAsync(dlab, labels, M, hipMemcpyHostToDevice, n.st));
  const float lw = loss_weight > 0.f ? loss_weight : (float)ncls;       // unet/unet.py:253: the integer 95
  const int hmode = mode == 0 ? 0 : 1, want_grad = mode == 2;
  if (fused) {
    ICS_CHECK(head_fused_ok(ncls, 128, M, ACT_NONE, 0), "this shape has no fused head kernel");
    ICS_TRY(launch_head_fused(n.st, dx, 128, nullptr, nullptr, dws, dwg, db, db + ncls, dz, dlab, M, hmode, want_grad | bce, lw,
                              dpart, 2048, dmet, nullptr, want_grad ? dcol : nullptr, dsum));
  } else {
    ICS_TRY(launch_pack_fwd(n.st, dws, 128, ncls, dwp, Kpad, Npad, 0, 0, 1));
    ICS_TRY(launch_pack_fwd(n.st, dwg, 128, 1, dwp, Kpad, Npad, 0, ncls, 0));
    ConvGeom g{1, 1, 0, 128, nz, 1, Kpad, Npad, n.flags};
    g.B = (int)M;                                                      // 1x1x1: rows are rows
    ConvSrc s = src_plain(dx, 128);
    ICS_TRY(launch_conv_fwd(n.st, g, &s, 1, dwp, db, dz, nz, ACT_NONE, nullptr, nullptr));
    ICS_TRY(launch_head(n.st, dz, nz, ncls, dlab, M, hmode, want_grad | bce, lw, dpart, 2048, dmet, nullptr,
                        want_grad ? dcol : nullptr, dsum));
  }
  if (out) ICS_HIP(hipMemcpyAsync(out, dz, M * nz * 4, hipMemcpyDeviceToHost, n.st));
  if (hmode && metrics) ICS_HIP(hipMemcpyAsync(metrics, dmet, 5 * 4, hipMemcpyDeviceToHost, n.st));
  if (hmode && sums) ICS_HIP(hipMemcpyAsync(sums, dsum, 7 * 8, hipMemcpyDeviceToHost, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  return 0;
}

// micro-benchmark of one conv layer on device-resident data (HIP events): mode 0 fwd (ablate 0/1/2),
// 1 backward-data, 2 backward-weight.  Returns average milliseconds per launch.
int ics_op_conv3d_bench(int B, int S, int Cin, int Cout, int taps, int mode, int ablate, int iters, float* ms_out) {
  ICS_CHECK(ms_out && iters >= 1, "bad arguments");
  Net n;
  ICS_TRY(op_prepare(n));
  const size_t M = (size_t)B * S * S * S;
  const int Kpad = round_up(taps * Cin, 32), Npad = round_up(Cout, 32);
  const int Kpad_b = round_up(taps * Cout, 32), Npad_b = round_up(Cin, 32);
  float *dx, *dw, *dwp, *dwf, *dyv, *dgx, *dgw, *ws;
  ICS_TRY(n.alloc(&dx, M * Cin)); ICS_TRY(n.alloc(&dw, (size_t)taps * Cin * Cout));
  ICS_TRY(n.alloc(&dwp, (size_t)Kpad * Npad)); ICS_TRY(n.alloc(&dwf, (size_t)Kpad_b * Npad_b));
  ICS_TRY(n.alloc(&dyv, M * Cout)); ICS_TRY(n.alloc(&dgx, M * Cin)); ICS_TRY(n.alloc(&dgw, (size_t)taps * Cin * Cout));
  ICS_TRY(fill(n, dx, M * Cin, 0.37f)); ICS_TRY(fill(n, dw, (size_t)taps * Cin * Cout, 0.011f));
  ICS_TRY(fill(n, dyv, M * Cout, 0.23f));
  ICS_TRY(launch_pack_fwd(n.st, dw, taps * Cin, Cout, dwp, Kpad, Npad, 0, 0, 1));
  ICS_TRY(launch_pack_bwd(n.st, dw, taps, Cin, Cout, dwf, Kpad_b, Npad_b, Cout, 0, 1));
  ConvGeom g{B, S, ilog2(S), Cin, Cout, taps, Kpad, Npad, n.flags};
  ConvGeom gb{B, S, ilog2(S), Cout, Cin, taps, Kpad_b, Npad_b, n.flags};
  ConvSrc sx = src_plain(dx, Cin), sd = src_plain(dyv, Cout);
  // the path the engine takes for this shape: Winograd where conv_wino_ok / conv_wino_wgrad_ok accept it
  // ablate >= 8 (Winograd paths only): feature bits of the launch the engine makes -- 8: BatchNorm-affine source,
  // 16: per-block BatchNorm statistics, 32: bias
  const int feat = ablate >= 8 ? ablate : 0;
  if (feat) ablate = 0;
  const bool wf = !ablate && conv_wino_ok(g, &sx, 1), wb = !ablate && conv_wino_ok(gb, &sd, 1);
  const bool ww = !ablate && conv_wino_wgrad_ok(g, &sx, 1);
  float *f_aff = nullptr, *f_stat = nullptr, *f_bias = nullptr;
  if (feat) {
    ICS_TRY(n.alloc(&f_aff, (size_t)2 * std::max(Cin, Cout))); ICS_TRY(fill(n, f_aff, (size_t)2 * std::max(Cin, Cout), 0.9f));
    ICS_TRY(n.alloc(&f_stat, (M / 64 + 1) * 3 * (size_t)Npad)); ICS_TRY(n.alloc(&f_bias, (size_t)Cout));
    if (feat & 8) { sx.scale = f_aff; sx.shift = f_aff + Cin; }
  }
  float *wwf = nullptr, *wwb = nullptr;
  if (wf) { ICS_TRY(n.alloc(&wwf, conv_wino_weight_floats(Cin, Cout))); ICS_TRY(launch_pack_wino(n.st, dw, Cin, Cout, 0, Cin, 0, wwf, conv_wino_layout(g))); }
  if (wb) { ICS_TRY(n.alloc(&wwb, conv_wino_weight_floats(Cin, Cout))); ICS_TRY(launch_pack_wino(n.st, dw, Cin, Cout, 0, Cin, 1, wwb, conv_wino_layout(gb))); }
  const size_t wsn = ww ? conv_wino_wgrad_workspace_floats(g) : conv_wgrad_workspace_floats(g, &sx, 1);
  ICS_TRY(n.alloc(&ws, wsn + 16));
  hipEvent_t e0, e1;
  ICS_HIP(hipEventCreate(&e0)); ICS_HIP(hipEventCreate(&e1));
  for (int it = -2; it < iters; ++it) {
    if (it == 0) ICS_HIP(hipEventRecord(e0, n.st));
    if (mode == 0) {
      if (ablate) ICS_TRY(launch_conv_fwd_ablate(n.st, g, &sx, 1, dwp, dyv, Cout, ablate));
      else if (wf) ICS_TRY(launch_conv_fwd_wino(n.st, g, sx, wwf, (feat & 32) ? f_bias : nullptr, dyv, Cout, ACT_RELU,
                                                (feat & 16) ? f_stat : nullptr, nullptr, 0, conv_wino_layout(g)));
      else ICS_TRY(launch_conv_fwd(n.st, g, &sx, 1, dwp, nullptr, dyv, Cout, ACT_RELU, nullptr, nullptr));
    } else if (mode == 1) {
      if (wb) ICS_TRY(launch_conv_fwd_wino(n.st, gb, sd, wwb, nullptr, dgx, Cin, ACT_NONE, nullptr, nullptr, 0,
                                           conv_wino_layout(gb)));
      else ICS_TRY(launch_conv_fwd(n.st, gb, &sd, 1, dwf, nullptr, dgx, Cin, ACT_NONE, nullptr, nullptr));
    } else if (ablate) {
      ICS_TRY(launch_conv_wgrad_ablate(n.st, g, &sx, dyv, Cout, ws, ablate));
    } else if (ww) {
      ICS_TRY(launch_conv_wgrad_wino(n.st, g, sx, dyv, Cout, dgw, Cout, ws, wsn, 0, 0, 0, 0));
    } else {
      ICS_TRY(launch_conv_wgrad(n.st, g, &sx, 1, dyv, Cout, dgw, Cout, ws, wsn));
    }
  }
  ICS_HIP(hipEventRecord(e1, n.st));
  ICS_HIP(hipStreamSynchronize(n.st));
  float ms = 0.f;
  ICS_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return 0;
}

int ics_op_conv3d_backward(const float* x, const float* w, const float* dy, int B, int S, int Cin, int Cout,
                           int taps, float* dxo, float* dwo) {
  ICS_CHECK(x && w && dy && (taps == 27 || taps == 1) && S >= 1 && (S & (S - 1)) == 0, "bad conv arguments");
  Net n;
  ICS_TRY(op_prepare(n));
  const size_t M = (size_t)B * S * S * S;
  const int Kpad_b = round_up(taps * Cout, 32), Npad_b = round_up(Cin, 32);
  const int Kpad = round_up(taps * Cin, 32), Npad = round_up(Cout, 32);
  float *dx, *dw, *ddy, *dwf, *dgx, *dgw, *ws;
  ICS_TRY(n.alloc(&dx, M * Cin)); ICS_TRY(n.alloc(&dw, (size_t)taps * Cin * Cout)); ICS_TRY(n.alloc(&ddy, M * Cout));
  ICS_TRY(n.alloc(&dwf, (size_t)Kpad_b * Npad_b)); ICS_TRY(n.alloc(&dgx, M * Cin));
  ICS_TRY(n.alloc(&dgw, (size_t)taps * Cin * Cout));
  ICS_HIP(hipMemcpyAsync(dx, x, M * Cin * 4, hipMemcpyHostToDevice, n.st));
  ICS_HIP(hipMemcpyAsync(dw, w, (size_t)taps * Cin * Cout * 4, hipMemcpyHostToDevice, n.st));
  ICS_HIP(hipMemcpyAsync(ddy, dy, M * Cout * 4, hipMemcpyHostToDevice, n.st));
  ConvGeom g{B, S, ilog2(S), Cin, Cout, taps, Kpad, Npad, n.flags};
  ConvSrc s = src_plain(dx, Cin);
  const bool wino_w = conv_wino_wgrad_ok(g, &s, 1);
  const size_t wsn = wino_w ? conv_wino_wgrad_workspace_floats(g) : conv_wgrad_workspace_floats(g, &s, 1);
  ICS_TRY(n.alloc(&ws, wsn + 16));
  if (dwo && conv_winog_wgrad_ok(g, &s, 1)) {
    float *sv = nullptr, *sm = nullptr, *svt = nullptr, *sz = nullptr, *sdu = nullptr, *wg = nullptr, *tmp = nullptr;
    size_t v, m, z;
    conv_winog_scratch_floats(g, &v, &m, &z);
    ICS_TRY(n.alloc(&sv, v)); ICS_TRY(n.alloc(&sm, m)); ICS_TRY(n.alloc(&svt, (size_t)64 * B * 8 * Cin)); ICS_TRY(n.alloc(&sz, z));
    ICS_TRY(n.alloc(&sdu, (size_t)64 * Cin * Cout)); ICS_TRY(n.alloc(&wg, conv_winog_weight_floats(Cin, Cout)));
    ICS_TRY(n.alloc(&tmp, M * Cout));
    // the forward pass (as the engine runs it) leaves the transposed transform of x behind; its output is not needed here
    ICS_TRY(launch_pack_wino(n.st, dw, Cin, Cout, 0, Cin, 0, wg, 2));
    ICS_TRY(launch_conv_fwd_winog(n.st, g, s, wg, nullptr, tmp, Cout, ACT_NONE, nullptr, nullptr, sv, sm, svt));
    ICS_TRY(launch_conv_wgrad_winog(n.st, g, svt, ddy, Cout, dgw, Cout, 0, 0, sz, sdu));
    ICS_HIP(hipMemcpyAsync(dwo, dgw, (size_t)taps * Cin * Cout * 4, hipMemcpyDeviceToHost, n.st));
  } else if (dwo) {
    if (wino_w) {
      ICS_TRY(launch_conv_wgrad_wino(n.st, g, s, ddy, Cout, dgw, Cout, ws, wsn, 0, 0, 0, 0));
    } else
    ICS_TRY(launch_conv_wgrad(n.st, g, &s, 1, ddy, Cout, dgw, Cout, ws, wsn));
    ICS_HIP(hipMemcpyAsync(dwo, dgw, (size_t)taps * Cin * Cout * 4, hipMemcpyDeviceToHost, n.st));
  }
  if (dxo) {
    ICS_TRY(launch_pack_bwd(n.st, dw, taps, Cin, Cout, dwf, Kpad_b, Npad_b, Cout, 0, 1));
    ConvGeom gb{B, S, ilog2(S), Cout, Cin, taps, Kpad_b, Npad_b, n.flags};
    ConvSrc sd = src_plain(ddy, Cout);
    if (conv_wino_ok(gb, &sd, 1)) {
      float* wwb = nullptr;
      ICS_TRY(n.alloc(&wwb, conv_wino_weight_floats(Cin, Cout)));
      const int layout = conv_wino_layout(gb);
      ICS_TRY(launch_pack_wino(n.st, dw, Cin, Cout, 0, Cin, 1, wwb, layout));
      ICS_TRY(launch_conv_fwd_wino(n.st, gb, sd, wwb, nullptr, dgx, Cin, ACT_NONE, nullptr, nullptr, 0, layout));
    } else if (conv_winog_ok(gb, &sd, 1)) {
      float *wgb = nullptr, *sv = nullptr, *sm = nullptr;
      size_t v, m, z;
      conv_winog_scratch_floats(gb, &v, &m, &z);
      ICS_TRY(n.alloc(&wgb, conv_winog_weight_floats(Cin, Cout))); ICS_TRY(n.alloc(&sv, v)); ICS_TRY(n.alloc(&sm, m));
      ICS_TRY(launch_pack_wino(n.st, dw, Cin, Cout, 0, Cin, 1, wgb, 2));
      ICS_TRY(launch_conv_fwd_winog(n.st, gb, sd, wgb, nullptr, dgx, Cin, ACT_NONE, nullptr, nullptr, sv, sm, nullptr));
    } else
    ICS_TRY(launch_conv_fwd(n.st, gb, &sd, 1, dwf, nullptr, dgx, Cin, ACT_NONE, nullptr, nullptr));
    ICS_HIP(hipMemcpyAsync(dxo, dgx, M * Cin * 4, hipMemcpyDeviceToHost, n.st));
  }
  ICS_HIP(hipStreamSynchronize(n.st));
  return 0;
}

}  // extern "C"
