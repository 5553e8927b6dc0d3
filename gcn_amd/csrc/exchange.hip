// exchange.hip — helpers for the "push" form of the multi-GPU layer exchange (gcn_amd/dist.py, exchange="push"):
// every rank writes its shard of a layer's output straight into its peers' exchange buffers (mapped once through IPC
// handles) with the runtime's copy path — hipMemcpyAsync between devices: the SDMA engines, no compute units — then
// raises one flag per peer and layer behind the data; the consumer waits for its world-1 flags with ONE wave.  What
// this avoids: collective kernels (RCCL's all-gather / send-recv run on tens of workgroups) competing for the CUs with
// the two oversubscribed SpMM main kernels of a layer.  No reference counterpart: the reference is
// single-GPU (flexspmm.cu:507).  UNMEASURED on multi-GPU hardware (no such node was available to this build); exercised
// by a two-rank rehearsal on one GPU (tests/test_bench_gpu.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gcn_spmm.h"

namespace gcn {

// lane i < count (i != skip) polls flags[i] until it equals `value`; every lane gives up after ~timeout_ticks of the
// 100 MHz wall clock and reports which flag it was waiting for: the wave always ends (a peer that died must not hang
// this GPU).  Flags are written by peers' copy engines: system-scope loads that bypass the caches.
__global__ void __launch_bounds__(64)
wait_flags_kernel(const int* __restrict__ flags, int count, int skip, int value, int* __restrict__ status,
                  unsigned long long timeout_ticks) {
  const int i = threadIdx.x;
  if (i >= count || i == skip) return;
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != value) {
    if (wall_clock64() - t0 > timeout_ticks) {
      __hip_atomic_store(status, i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    __builtin_amdgcn_s_sleep(32);
  }
}

}  // namespace gcn

extern "C" {

// The flags live in FINE-GRAINED device memory (coherent at system scope: a peer's copy engine writes them, this GPU's
// wave polls them — ordinary hipMalloc memory may sit stale in this GPU's L2) and travel between processes as raw
// 64-byte IPC handles.
int gcn_exchange_flags_create(int32_t count, int32_t** flags_dev_out, void* ipc_handle_out_64) {
  if (count <= 0 || count > 64 || !flags_dev_out || !ipc_handle_out_64) return GCN_ERR_INVALID_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI hands the IPC handle over as 64 raw bytes");
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, sizeof(int32_t) * 64, hipDeviceMallocFinegrained) != hipSuccess) return GCN_ERR_ALLOC;
  hipIpcMemHandle_t h;
  if (hipMemset(p, 0, sizeof(int32_t) * 64) != hipSuccess || hipIpcGetMemHandle(&h, p) != hipSuccess) { (void)hipFree(p); return GCN_ERR_HIP; }
  __builtin_memcpy(ipc_handle_out_64, &h, 64);
  *flags_dev_out = (int32_t*)p;
  return GCN_OK;
}

int gcn_exchange_flags_open(const void* ipc_handle_64, int32_t** flags_peer_out) {
  if (!ipc_handle_64 || !flags_peer_out) return GCN_ERR_INVALID_ARG;
  hipIpcMemHandle_t h;
  __builtin_memcpy(&h, ipc_handle_64, 64);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return GCN_ERR_HIP;
  *flags_peer_out = (int32_t*)p;
  return GCN_OK;
}

int gcn_exchange_flags_close(int32_t* flags_peer) {
  if (!flags_peer) return GCN_OK;
  return hipIpcCloseMemHandle(flags_peer) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
}

int gcn_exchange_flags_destroy(int32_t* flags_dev) {
  if (!flags_dev) return GCN_OK;
  return hipFree(flags_dev) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
}

int gcn_exchange_push(void* dst_peer, const void* src, size_t bytes, void* stream) {
  if (bytes == 0) return GCN_OK;
  if (!dst_peer || !src) return GCN_ERR_INVALID_ARG;
  return hipMemcpyAsync(dst_peer, src, bytes, hipMemcpyDefault, (hipStream_t)stream) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
}

int gcn_exchange_signal(int32_t* flag_peer, const int32_t* value_dev, void* stream) {
  if (!flag_peer || !value_dev) return GCN_ERR_INVALID_ARG;
  return hipMemcpyAsync(flag_peer, value_dev, sizeof(int32_t), hipMemcpyDefault, (hipStream_t)stream) == hipSuccess ? GCN_OK : GCN_ERR_HIP;
}

int gcn_exchange_wait(const int32_t* flags_dev, int32_t count, int32_t skip, int32_t value, int32_t* status_dev,
                      double timeout_seconds, void* stream) {
  if (!flags_dev || !status_dev || count < 0 || count > 64 || !(timeout_seconds > 0.0)) return GCN_ERR_INVALID_ARG;
  if (count == 0) return GCN_OK;
  const unsigned long long ticks = (unsigned long long)(timeout_seconds * 1.0e8);     // wall_clock64: 100 MHz
  gcn::wait_flags_kernel<<<1, 64, 0, (hipStream_t)stream>>>(flags_dev, count, skip, value, status_dev, ticks);
  return hipGetLastError() == hipSuccess ? GCN_OK : GCN_ERR_HIP;
}

}  // extern "C"
