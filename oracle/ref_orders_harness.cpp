// oracle/ref_orders_harness.cpp — TEST INFRASTRUCTURE ONLY.
// A thin extern "C" wrapper (our code) around the reference's INTERNAL ordering
// functions, which renumber.so does not export: order_deg (order_deg.cu:41-56),
// order_rcm (order_rcm.cu:15-33) and complete_gorder (order_gorder.cu:13-31).
// It is compiled together with the reference's sources where they lie
// (oracle/Makefile, target `ref`); nothing of the reference is copied.
#include <vector>
#include "edgelist.cuh"
#include "order_deg.cuh"
#include "order_rcm.cuh"
#include "order_gorder.cuh"

extern "C" {

// which: 0 total, 1 out, 2 in.  out[old] = new.
void ref_order_deg(int* rowptr, int* col, int n, int nnz, int which, int desc, long long* out) {
  Edgelist h(rowptr, col, n, nnz);
  std::vector<ul> r = which == 0 ? order_deg(h, desc != 0)
                    : which == 1 ? order_degOut(h, desc != 0) : order_degIn(h, desc != 0);
  for (int i = 0; i < n; ++i) out[i] = (long long)r[i];
}

void ref_order_rcm(int* rowptr, int* col, int n, int nnz, int directed, long long* out) {
  Edgelist h(rowptr, col, n, nnz);
  std::vector<ul> r = order_rcm(h, directed != 0);
  for (int i = 0; i < n; ++i) out[i] = (long long)r[i];
}

void ref_complete_gorder(int* rowptr, int* col, int n, int nnz, int window, long long* out) {
  Edgelist h(rowptr, col, n, nnz);
  std::vector<unsigned int> r = complete_gorder(h, (ul)window);
  for (int i = 0; i < n; ++i) out[i] = (long long)r[i];
}

}
