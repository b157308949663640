// segments.h -- host-only: the nnz-balanced schedule of the balanced SpMM (spmm.hip spmm_balanced_kernel, ppr.hip ppr_spmm_kernel).
//
// Every scheduled row is cut into segments of <= seg_edges stored entries; a row with s segments occupies an ALIGNED block of p = pow2ceil(s)
// consecutive lane groups of ONE workgroup (waves x 2^gpw_log2 groups), rows too long for that take the whole workgroup with longer
// segments.  Rows are placed in order of non-increasing block size (then non-increasing length), so blocks stay aligned without
// gaps; the tail of the last workgroup is padded with empty descriptors (row -1).  One int4 per lane group:
//     {row, first entry, entry count, log2 p | 0x100 if some row of the workgroup spans more than one wave}
// The kernels trust these descriptors blindly (a wrong one reads or writes out of bounds on the device), so this routine is kept free
// of HIP and is run under ASan / UBSan with random degree sequences by tests/test_native_host.py.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace gss {

struct SegItem {
  int32_t row, first, len;   // output row, first entry (index into the col / val arrays the kernel is given), entry count
};

// -> number of workgroups; segs receives 4 int32 per lane group.  `items`: the rows to schedule, in row order (rows that are not
// listed get no descriptor: nothing computes or writes them)
inline int build_segments_items(const SegItem *items_in, size_t n_items, int waves, int gpw_log2, int seg_edges, std::vector<int32_t> &segs) {
  const int gpw = 1 << gpw_log2;
  const int ngb = waves * gpw;  // groups per workgroup
  int ngb_log2 = 0;
  while ((1 << ngb_log2) < ngb) ++ngb_log2;
  struct RowItem {
    int32_t row, first, len, plog;
  };
  std::vector<RowItem> items;
  items.reserve(n_items);
  int64_t nnz = 0;
  for (size_t i = 0; i < n_items; ++i) {
    const int32_t len = items_in[i].len;
    nnz += len;
    const int sgm = len <= seg_edges ? 1 : (len + seg_edges - 1) / seg_edges;
    int plog = 0;
    while ((1 << plog) < sgm && plog < ngb_log2) ++plog;  // longer rows: whole workgroup, longer segments
    items.push_back({items_in[i].row, items_in[i].first, len, plog});
  }
  std::stable_sort(items.begin(), items.end(), [](const RowItem &x, const RowItem &y) {
    return x.plog != y.plog ? x.plog > y.plog : x.len > y.len;
  });
  segs.clear();
  segs.reserve(((size_t)items.size() + (size_t)(nnz / (seg_edges > 0 ? seg_edges : 1)) + 2 * (size_t)ngb) * 4);
  for (const RowItem &it : items) {  // sizes are non-increasing powers of two: blocks stay aligned
    const int p = 1 << it.plog;
    const int per = (it.len + p - 1) / p;
    for (int k = 0; k < p; ++k) {
      const int b0 = std::min(it.len, k * per), b1 = std::min(it.len, (k + 1) * per);
      segs.push_back(it.row);
      segs.push_back(it.first + b0);
      segs.push_back(b1 - b0);
      segs.push_back(it.plog);
    }
  }
  while ((segs.size() / 4) % (size_t)ngb != 0) {
    segs.push_back(-1);
    segs.push_back(0);
    segs.push_back(0);
    segs.push_back(0);
  }
  const int nblk = (int)(segs.size() / 4 / (size_t)ngb);
  for (int bi = 0; bi < nblk; ++bi) {  // flag workgroups that need the LDS step
    bool multi = false;
    for (int k = 0; k < ngb; ++k) multi |= (1 << segs[((size_t)bi * ngb + k) * 4 + 3]) > gpw;
    if (multi)
      for (int k = 0; k < ngb; ++k) segs[((size_t)bi * ngb + k) * 4 + 3] |= 0x100;
  }
  return nblk;
}

// every row of a CSR
inline int build_segments(const int32_t *rowptr, int32_t n_rows, int waves, int gpw_log2, int seg_edges, std::vector<int32_t> &segs) {
  std::vector<SegItem> items;
  items.reserve((size_t)(n_rows > 0 ? n_rows : 0));
  for (int32_t r = 0; r < n_rows; ++r) items.push_back({r, rowptr[r], rowptr[r + 1] - rowptr[r]});
  return build_segments_items(items.data(), items.size(), waves, gpw_log2, seg_edges, segs);
}

// Giant rows (spmm.hip GiantRows): rows with more than `thr` stored entries are cut into chunks of thr / 4.  Three item lists over the rows of
// one CSR: `shortv` = every other row as it is; `chunks` = one item per chunk (its "row" is the chunk's index, its entries a range of the
// ORIGINAL arrays); `finish` = one item per giant row over a [rows] x [chunks] matrix of ones (first = its first chunk, len = its chunks);
// chunk_row[k] = the row chunk k belongs to.  Returns the number of giant rows.
struct GiantItems {
  std::vector<SegItem> shortv, chunks, finish;
  std::vector<int32_t> chunk_row;
};
inline int32_t giant_items(const int32_t *rowptr, int32_t n_rows, int32_t thr, GiantItems &out) {
  out = GiantItems{};
  if (thr <= 0) return 0;
  const int32_t chunk = thr / 4 > 0 ? thr / 4 : 1;
  int32_t n_giant = 0;
  for (int32_t r = 0; r < n_rows; ++r) {
    const int32_t len = rowptr[r + 1] - rowptr[r];
    if (len <= thr) {
      out.shortv.push_back({r, rowptr[r], len});
      continue;
    }
    ++n_giant;
    const int32_t c0 = (int32_t)out.chunk_row.size();
    for (int32_t b0 = 0; b0 < len; b0 += chunk) {
      out.chunks.push_back({(int32_t)out.chunk_row.size(), rowptr[r] + b0, std::min(chunk, len - b0)});
      out.chunk_row.push_back(r);
    }
    out.finish.push_back({r, c0, (int32_t)out.chunk_row.size() - c0});
  }
  return n_giant;
}

}  // namespace gss
