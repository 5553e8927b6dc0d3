// plan.h — the SpMM plan object behind gcn_spmm_plan_t and what the C-ABI translation units share.
// Internal; the public contract is include/gcn_spmm.h.
//
// Every device buffer of a plan is a DevBuf: freed by its destructor, reset by assignment of an empty
// value — the optional parts of a plan (column slicing, the value-free streams, LDS panels, value factors)
// are aggregates of DevBufs that are dropped as a whole (`p->slicing = {}`), so there is no hand-kept free
// list to fall out of step with the struct.
#pragma once
#include "../../include/gcn_spmm.h"

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

#include "spmm_kernels.h"

namespace gcn {

template <class T>
class DevBuf {                                      // owning pointer to `count` device elements
 public:
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p_(o.p_), count_(o.count_) { o.p_ = nullptr; o.count_ = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { reset(); p_ = o.p_; count_ = o.count_; o.p_ = nullptr; o.count_ = 0; }
    return *this;
  }
  ~DevBuf() { reset(); }
  void reset() { if (p_) (void)hipFree(p_); p_ = nullptr; count_ = 0; }
  // (re)allocate exactly `count` elements (at least one); the old contents are gone
  hipError_t alloc(size_t count) {
    reset();
    const hipError_t e = hipMalloc((void**)&p_, sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) { p_ = nullptr; return e; }
    count_ = count;
    return hipSuccess;
  }
  // grow-only scratch: keeps the buffer when it is large enough (contents are never preserved)
  hipError_t grow(size_t count) { return (p_ && count <= count_) ? hipSuccess : alloc(count); }
  void adopt(T* q, size_t count) { reset(); p_ = q; count_ = count; }   // take over a hipMalloc'ed pointer
  T* get() const { return p_; }
  operator T*() const { return p_; }
  size_t count() const { return count_; }
 private:
  T* p_ = nullptr;
  size_t count_ = 0;
};

// HIP event pairs around the main kernel of the next launches (gcn_spmm_profile_begin/_end)
class EventPairs {
 public:
  EventPairs() = default;
  EventPairs(const EventPairs&) = delete;
  EventPairs& operator=(const EventPairs&) = delete;
  ~EventPairs() { clear(); }
  void clear() { for (auto& e : ev_) (void)hipEventDestroy(e); ev_.clear(); cap_ = n_ = 0; }
  hipError_t begin(int capacity) {                  // all events or none: a failed create leaves nothing behind
    clear();
    ev_.reserve(2 * (size_t)capacity);
    for (int i = 0; i < 2 * capacity; ++i) {
      hipEvent_t e;
      const hipError_t st = hipEventCreate(&e);
      if (st != hipSuccess) { clear(); return st; }
      ev_.push_back(e);
    }
    cap_ = capacity;
    return hipSuccess;
  }
  bool armed() const { return cap_ > 0 && n_ < cap_; }
  // the next pair (start, stop); call only when armed()
  std::pair<hipEvent_t, hipEvent_t> next() { const int i = n_++; return {ev_[2 * i], ev_[2 * i + 1]}; }
  int capacity() const { return cap_; }
  int recorded() const { return n_; }
  hipEvent_t start(int i) const { return ev_[2 * i]; }
  hipEvent_t stop(int i) const { return ev_[2 * i + 1]; }
 private:
  std::vector<hipEvent_t> ev_;
  int cap_ = 0, n_ = 0;
};

// XCD-aware column slicing (slicing.hip): slice-major copy of the matrix with S*m virtual rows
struct Slicing {
  int S = 0;                                        // 0 = off
  int empty_vrows = -1;                             // empty rows of the virtual CSR (SpmmArgs::empty_rows), -1: not counted
  DevBuf<int> vrowptr;                              // [S*m+1]
  DevBuf<int> vcol;                                 // [nnz]
  DevBuf<float> vval;                               // [nnz]
  DevBuf<int> vchunk_row;                           // [nchunks] rows of the virtual CSR
};

// value-free pass, slices <= 65 535 columns and <= 8 of them: 16-bit column stream of the four-per-gather kernel
struct Col16Stream {
  DevBuf<int> vrowptr16, vchunk_row16;              // [S*m+1], [nchunks16]
  DevBuf<unsigned short> vcol16;                    // [nnz16] offsets inside the slice, 0xFFFF = padding marker
  int nnz16 = 0, nchunks16 = 0, start16[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  bool ready() const { return vcol16 != nullptr; }
};

// slices <= 32 767 columns: 15-bit slice-major stream of the group kernel (spmm_group.hip), value-free or weighted
struct GroupStream {
  DevBuf<unsigned short> stream;                    // [nchunks*T]
  DevBuf<float> vals;                               // [nchunks*T] values in stream order (weighted pass), empty: value-free
  DevBuf<int> chunk_row, vrowptr, chunk_meta;       // [nchunks], [S*m+1], int2 [nchunks]
  DevBuf<int> fix;                                  // int4 [nfix]: rows cut by chunk ends {virtual row, c, c1, 0}
  DevBuf<int> cutptr, cutchunk;                     // the same pieces per OUTPUT row (CutLists): [m+1], [ncut] chunk numbers
  int nchunks = 0, T = 0, w = 0, nfix = 0, ncut = 0;
  bool ready() const { return stream != nullptr; }
};

// rank-1 values: val[r, c] = u_row[r] * u_col[c]
struct Factors {
  DevBuf<float> u_row, u_col_own;                   // [m]; [n] when distinct from u_row
  const float* u_col = nullptr;                     // = u_col_own, or u_row for a square normalised adjacency
  bool ready() const { return u_row != nullptr; }
};

// LDS-staged row panels (spmm_panel.hip): A = A_in + A_out
struct Panels {
  int R = 0;                                        // rows per panel, 0 = off
  double coverage = 0.0;                            // fraction of the non-zeros inside their panel's window
  DevBuf<int> w0;                                   // [ceil(m / R)]: first column of each panel's window
  DevBuf<int> in_rowptr, in_off;                    // staged entries: [m+1], [nnz_in] (LDS byte offsets)
  DevBuf<float> in_val;
  DevBuf<int> out_rowptr, out_col, out_chunk_row;   // the rest: plain CSR + its chunk plan
  DevBuf<float> out_val;
  int out_nnz = 0, out_T = 0, out_nchunks = 0;
  // dense panels on the matrix cores (spmm_panel_dense_mfma_kernel)
  int ndense = 0;
  DevBuf<int> dense_slot, dense_panel;              // [panels]: slot or -1; [ndense]: panel of every slot
  DevBuf<float> adense;                             // [ndense x 128 x 512] in MFMA fragment order
};

}  // namespace gcn

struct gcn_spmm_plan {
  int32_t m = 0, n = 0, nnz = 0, T = 0, nchunks = 0;
  int cu_count = 0, device = 0;
  int empty_rows = -1;                              // rows without an entry (counted at plan creation; -1: unknown): SpmmArgs::empty_rows
  gcn::DevBuf<int> chunk_row;                       // [nchunks]
  gcn::DevBuf<float> ws;                            // partial slab [2*chunks x k], grow-only
  gcn::DevBuf<float> cv;                            // partial outputs [S*m x k], grow-only
  gcn::DevBuf<float> bpad;                          // B re-laid (rows padded to whole lines and/or scaled by u_col), grow-only
  gcn::DevBuf<float> cpad;                          // result with k rounded up to a multiple of 4 (k % 4 != 0), grow-only
  gcn::DevBuf<int> dyn;                             // drop-in flexspmm: {recognised, chunks, cut rows, 0} of the current call (device)
  gcn::EventPairs prof;
  int tile_cols = 0;                                // 0 = auto
  int gather_width = 0;                             // non-zeros per gather instruction of the 64-column kernel: 0 auto, 1, 4
  int blocks_per_cu = 32;                           // grid size: blocks of 4 waves per CU (oversubscribed on purpose)
  gcn::Slicing slicing;
  bool slices_auto = false;                         // the slice count was chosen by auto_slices (enable_slicing(-1))
  gcn::Col16Stream col16;
  gcn::GroupStream group;
  // Narrow widths (value-free plans whose slice count was automatic): the same matrix cut into FEWER, wider slices —
  // class 0: k <= 32, a row of the table is 128 bytes, so half as many slices fill an L2 and the partial rows (whose cost
  // goes with the slice count) halve.  (Class 1, 33..48 on 192-byte rows, was measured and is not built.)  Built at the
  // first call of the class or by gcn_spmm_plan_prepare_width (plan_build.cpp, maybe_build_alt).  Which set a call runs on is
  // decided per call and passed down as an argument (plan_policy.h, SliceSet) — never stored here.
  gcn::GroupStream group_alt[1];                    // (indexed by width class; one class so far)
  int alt_S[1] = {0};
  bool alt_tried[1] = {false};
  gcn::Factors factors;
  gcn::Panels panels;
};

namespace gcn {

extern std::mutex g_plan_mu;                        // serialises workspace growth and the scratch plans

int cu_count_cached();
int auto_chunk_nnz(long long nnz, int cu);
int auto_tile_cols(long long n, int k);
int auto_slices(long long m, long long n, long long nnz, bool value_free = false);
int padded_ldb(long long n, int k);
[[noreturn]] void die(const char* what, hipError_t e);
bool verbose();
// scratch plan of the stateless entry points (oneshot / cuspmm / flexspmm): one per (device, stream), never freed
gcn_spmm_plan* scratch_plan(void* stream);

}  // namespace gcn
