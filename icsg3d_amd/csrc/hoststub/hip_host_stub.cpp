// Host-only stand-in for the HIP runtime and RCCL, used ONLY by `make asan` (CPU AddressSanitizer / UBSan build of
// the engine's host code: tensor tables, workspace planners, split-K / bucket planning, the C ABI).  "Device" memory is
// plain malloc memory, copies are memcpy, kernel launches and collectives are no-ops -- so every host-side offset,
// size and lifetime error in csrc/*.hip shows up under ASan on a CPU.  GPU sanitizers are not available on the pool
// (and numerics are not the point here: kernels do not run).  Never linked into libicsg3d_hip.so.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>

extern "C" {

hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
  std::memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind,
                            hipStream_t) {
  for (size_t r = 0; r < h; ++r) std::memmove((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)std::malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)std::malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
// stream capture / graphs (ics_net_graph_probe): a captured "graph" is an opaque allocation, a replay is a no-op
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = (hipGraph_t)std::malloc(8); return hipSuccess; }
hipError_t hipGraphGetNodes(hipGraph_t, hipGraphNode_t*, size_t* n) { *n = 0; return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) {
  *e = (hipGraphExec_t)std::malloc(8);
  return hipSuccess;
}
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { std::free(e); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { std::free(g); return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  std::memset(p, 0, sizeof(*p));
  std::strcpy(p->name, "host-stub (asan build)");
  p->multiProcessorCount = 256;
  p->totalGlobalMem = (size_t)288 << 30;
  return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "host-stub error"; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipSuccess; }
hipError_t hipExtLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t, hipEvent_t, hipEvent_t, int) { return hipSuccess; }

// clang's host-side launch / registration glue
static dim3 g_grid, g_block;
static size_t g_shmem;
static hipStream_t g_stream;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t s) {
  g_grid = g; g_block = b; g_shmem = sh; g_stream = s;
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* s) {
  *g = g_grid; *b = g_block; *sh = g_shmem; *s = g_stream;
  return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, char*, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void**) {}

// ---- roctx markers
int roctxRangePushA(const char*) { return 0; }
int roctxRangePop(void) { return 0; }

// ---- RCCL: a one-rank world
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { std::memset(id, 7, sizeof(*id)); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t* c, int, ncclUniqueId, int) { *c = (ncclComm_t)std::malloc(8); return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t c) { std::free(c); return ncclSuccess; }
ncclResult_t ncclCommSplit(ncclComm_t, int, int, ncclComm_t* c, ncclConfig_t*) { *c = (ncclComm_t)std::malloc(8); return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t) { return "host-stub nccl error"; }
static size_t nccl_size(ncclDataType_t t) { return (t == ncclDouble || t == ncclInt64 || t == ncclUint64) ? 8 : 4; }
ncclResult_t ncclAllReduce(const void* s, void* r, size_t n, ncclDataType_t t, ncclRedOp_t, ncclComm_t, hipStream_t) {
  if (s != r) std::memmove(r, s, n * nccl_size(t));     // touches the whole range: ASan checks its extent
  else { volatile char c = ((const char*)s)[n * nccl_size(t) - 1]; (void)c; }
  return ncclSuccess;
}
ncclResult_t ncclAllGather(const void* s, void* r, size_t n, ncclDataType_t t, ncclComm_t, hipStream_t) {
  std::memmove(r, s, n * nccl_size(t));
  return ncclSuccess;
}
ncclResult_t ncclBroadcast(const void* s, void* r, size_t n, ncclDataType_t t, int, ncclComm_t, hipStream_t) {
  if (s != r) std::memmove(r, s, n * nccl_size(t));
  else { volatile char c = ((const char*)s)[n * nccl_size(t) - 1]; (void)c; }
  return ncclSuccess;
}

}  // extern "C"
