// plan_policy.h — what the three translation units behind the native SpMM API share (internal):
//   plan_policy.cpp  the dispatch policy: which kernel family, tile, slice count and slice set a k-wide call takes
//   plan_build.cpp   plan construction: chunk table, column slicing, streams, value factors, panels
//   api_spmm.cpp     the C entry points that launch
// The public contract is include/gcn_spmm.h.
#pragma once
#include "plan.h"

namespace gcn {

// thresholds of the policy (measured; the experiments behind them are cited where they are used)
constexpr int kSliceMinK = 33;          // smallest k the sliced copy is used for on the four-per-gather kernel
constexpr int kGroupMinK = 12;          // ... when the group kernels walk it
constexpr int kVallessMinPerCol = 48;   // non-zeros per column from which the scaled copy of B (value-free pass) pays
constexpr double kPanelMfmaDensity = 0.25;   // window density from which a panel becomes a dense MFMA tile

// The slice set ONE call runs on — the plan's own, or the narrow set built for k <= 32 (plan.h: group_alt) — decided once
// per call and handed down as an argument (it used to be per-call state inside the plan: two streams sharing a plan at
// different widths raced on it).
struct SliceSet {
  const GroupStream* g = nullptr;
  int S = 0;
  int alt = -1;                         // width class of group_alt, -1: the plan's own set
  long long table_rows() const { return (long long)S * ((long long)(g ? g->w : 0) + 1); }
};
SliceSet own_slice_set(const gcn_spmm_plan* p);
int alt_class(int k);                   // width class of the narrow slice sets (0: k <= 32), -1: none

// development switches that select an alternate code path a TEST needs (read once per process)
bool group8_enabled();                  // GCN_AMD_GROUP8=0: k <= 32 on the 64-column group pass
bool group_fused_fixup();               // GCN_AMD_GROUP_FUSED_FIXUP=0: cut rows' pieces added by a pass of their own

int group_chunk(long long entries, int cu);
size_t ws_elems(const gcn_spmm_plan* p, int k);
long long group_table_rows(const gcn_spmm_plan* p);

bool sliced_for(const gcn_spmm_plan* p, int k);
bool valless_pays(const gcn_spmm_plan* p, int k, int ldb);
bool value_free_plan(const gcn_spmm_plan* p);
bool group_plan(const gcn_spmm_plan* p);
bool weighted_pass(const gcn_spmm_plan* p, int k, int ldb);
bool group_launch(const gcn_spmm_plan* p, bool valless, bool weighted);
bool odd_width_detour(const gcn_spmm_plan* p, int k);

// Which slice set does a k-wide call run on (k already rounded up to a multiple of 4; *ldb the row stride it would gather
// with, lowered to 48 for the widths the five-engine kernel serves from 192-byte rows: *relay = the call lays that copy
// out itself)?  `build`: build the narrow set at first use (needs the CSR); without it only what exists is chosen.
SliceSet pick_slice_set(gcn_spmm_plan* p, int k, int* ldb, bool* relay, bool build, const int32_t* rowptr, const int32_t* col,
                        const float* val, hipStream_t st);

// plan_build.cpp
int count_empty(const int* rowptr, int m, int* out, hipStream_t st);
void build_sliced_streams(gcn_spmm_plan* p, hipStream_t st);
void drop_streams(gcn_spmm_plan* p);
void maybe_build_alt(gcn_spmm_plan* p, int cls, const int32_t* rowptr, const int32_t* col, const float* val, hipStream_t st);

// grow-only scratch of a plan; plans may be shared between host threads, so growth is serialised
template <class T>
int grow(DevBuf<T>& buf, size_t count) {
  std::lock_guard<std::mutex> lk(g_plan_mu);
  return buf.grow(count) == hipSuccess ? GCN_OK : GCN_ERR_ALLOC;
}

}  // namespace gcn
