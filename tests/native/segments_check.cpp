// Host-side check of csrc/segments.h (the schedule the balanced SpMM kernels trust blindly), built with -fsanitize=address,undefined by
// tests/test_native_host.py.  Random degree sequences (power law, empty rows, hubs longer than a whole workgroup's capacity, n = 0 / 1)
// x every lane-group width x several segment lengths; every invariant the kernels rely on is asserted.  TEST INFRASTRUCTURE.
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

#include "segments.h"

#define CHECK(cond, ...)                       \
  do {                                         \
    if (!(cond)) {                             \
      fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);            \
      fprintf(stderr, "\n");                   \
      exit(1);                                 \
    }                                          \
  } while (0)

static long check_one(const std::vector<int32_t> &rowptr, int waves, int gpw_log2, int seg_edges) {
  const int32_t n = (int32_t)rowptr.size() - 1;
  std::vector<int32_t> segs;
  const int nblk = gss::build_segments(rowptr.data(), n, waves, gpw_log2, seg_edges, segs);
  const int gpw = 1 << gpw_log2, ngb = waves * gpw;
  int ngb_log2 = 0;
  while ((1 << ngb_log2) < ngb) ++ngb_log2;
  CHECK(segs.size() % 4 == 0 && (segs.size() / 4) % (size_t)ngb == 0, "descriptor count %zu not a multiple of %d", segs.size() / 4, ngb);
  CHECK((size_t)nblk * ngb == segs.size() / 4, "nblk %d does not match %zu descriptors", nblk, segs.size() / 4);
  std::vector<int> seen((size_t)(n > 0 ? n : 0), 0);
  const size_t nd = segs.size() / 4;
  bool padding = false;
  for (size_t i = 0; i < nd;) {
    const int32_t row = segs[i * 4], plog = segs[i * 4 + 3] & 0xff;
    if (row < 0) {
      padding = true;
      CHECK(segs[i * 4 + 2] == 0, "padding descriptor %zu has %d entries", i, segs[i * 4 + 2]);
      ++i;
      continue;
    }
    CHECK(!padding, "row %d after padding", row);
    CHECK(row < n, "row %d out of range", row);
    CHECK(plog <= ngb_log2, "row %d: block of 2^%d groups exceeds the workgroup", row, plog);
    const size_t p = (size_t)1 << plog;
    CHECK(i % p == 0, "row %d: block of %zu groups starts at unaligned group %zu", row, p, i);
    CHECK((i % (size_t)ngb) + p <= (size_t)ngb, "row %d: block crosses a workgroup boundary", row);
    CHECK(!seen[(size_t)row], "row %d scheduled twice", row);
    seen[(size_t)row] = 1;
    int32_t next = rowptr[(size_t)row];
    for (size_t k = 0; k < p; ++k) {
      CHECK(i + k < nd && segs[(i + k) * 4] == row && (segs[(i + k) * 4 + 3] & 0xff) == plog, "row %d: descriptor %zu of its block differs", row, k);
      const int32_t first = segs[(i + k) * 4 + 1], cnt = segs[(i + k) * 4 + 2];
      CHECK(cnt >= 0 && first == next, "row %d: segment %zu starts at %d, expected %d", row, k, first, next);
      CHECK(cnt <= seg_edges || plog == ngb_log2, "row %d: segment of %d entries (> %d) in a block that is not the whole workgroup", row, cnt, seg_edges);
      next = first + cnt;
    }
    CHECK(next == rowptr[(size_t)row + 1], "row %d: segments end at %d, row ends at %d", row, next, rowptr[(size_t)row + 1]);
    i += p;
  }
  for (int32_t r = 0; r < n; ++r) CHECK(seen[(size_t)r], "row %d never scheduled", r);
  for (int b = 0; b < nblk; ++b) {
    bool multi = false, flagged_all = true, flagged_any = false;
    for (int k = 0; k < ngb; ++k) {
      const int32_t w = segs[((size_t)b * ngb + k) * 4 + 3];
      multi |= (1 << (w & 0xff)) > gpw;
      flagged_all &= (w & 0x100) != 0;
      flagged_any |= (w & 0x100) != 0;
    }
    CHECK(multi ? flagged_all : !flagged_any, "workgroup %d: multi-wave flag inconsistent", b);
  }
  return (long)nd;
}

// the three item lists a CSR with giant rows is scheduled by (segments.h giant_items; spmm.hip GiantRows): together they must cover every
// stored entry exactly once -- a row's entries either in the short schedule or in its chunks --, a chunk never exceeds thr / 4 entries, a
// giant row's chunks are consecutive and in entry order, and each list's schedule is well-formed over its own index space
static long check_giant(const std::vector<int32_t> &rowptr, int32_t thr, int gpw_log2, int seg_edges) {
  const int32_t n = (int32_t)rowptr.size() - 1;
  gss::GiantItems it;
  const int32_t n_giant = gss::giant_items(rowptr.data(), n, thr, it);
  const int64_t nnz = n > 0 ? rowptr[(size_t)n] : 0;
  std::vector<unsigned char> covered((size_t)nnz, 0);
  std::vector<int> row_seen((size_t)(n > 0 ? n : 0), 0);
  for (const gss::SegItem &s : it.shortv) {
    CHECK(s.row >= 0 && s.row < n && s.first == rowptr[(size_t)s.row] && s.len == rowptr[(size_t)s.row + 1] - s.first && s.len <= thr, "short item of row %d", s.row);
    CHECK(!row_seen[(size_t)s.row]++, "row %d listed twice", s.row);
    for (int32_t e = s.first; e < s.first + s.len; ++e) CHECK(!covered[(size_t)e]++, "entry %d covered twice", e);
  }
  CHECK(it.chunk_row.size() == it.chunks.size(), "chunk_row has %zu entries for %zu chunks", it.chunk_row.size(), it.chunks.size());
  const int32_t chunk = thr / 4 > 0 ? thr / 4 : 1;
  for (size_t k = 0; k < it.chunks.size(); ++k) {
    const gss::SegItem &c = it.chunks[k];
    const int32_t r = it.chunk_row[k];
    CHECK(c.row == (int32_t)k && r >= 0 && r < n && c.len >= 1 && c.len <= chunk, "chunk %zu: row %d, %d entries", k, c.row, c.len);
    CHECK(c.first >= rowptr[(size_t)r] && c.first + c.len <= rowptr[(size_t)r + 1], "chunk %zu leaves row %d", k, r);
    for (int32_t e = c.first; e < c.first + c.len; ++e) CHECK(!covered[(size_t)e]++, "entry %d covered twice", e);
  }
  CHECK((int32_t)it.finish.size() == n_giant, "%zu finish items for %d giant rows", it.finish.size(), n_giant);
  int32_t next_chunk = 0;
  for (const gss::SegItem &f : it.finish) {
    CHECK(f.row >= 0 && f.row < n && rowptr[(size_t)f.row + 1] - rowptr[(size_t)f.row] > thr, "finish item of row %d, which is not giant", f.row);
    CHECK(!row_seen[(size_t)f.row]++, "row %d listed twice", f.row);
    CHECK(f.first == next_chunk && f.len >= 5, "row %d: chunks [%d, +%d), expected to start at %d (> thr entries = at least 5 chunks)", f.row, f.first, f.len, next_chunk);
    int32_t e = rowptr[(size_t)f.row];
    for (int32_t k = f.first; k < f.first + f.len; ++k) {
      CHECK(it.chunk_row[(size_t)k] == f.row && it.chunks[(size_t)k].first == e, "row %d: chunk %d out of order", f.row, k);
      e += it.chunks[(size_t)k].len;
    }
    CHECK(e == rowptr[(size_t)f.row + 1], "row %d: chunks end at %d", f.row, e);
    next_chunk += f.len;
  }
  CHECK(next_chunk == (int32_t)it.chunks.size(), "%d chunks listed by giant rows of %zu", next_chunk, it.chunks.size());
  for (int64_t e = 0; e < nnz; ++e) CHECK(covered[(size_t)e] == 1, "entry %lld not covered", (long long)e);
  for (int32_t r = 0; r < n; ++r) CHECK(row_seen[(size_t)r] == 1, "row %d not listed", r);
  // each list as a schedule: descriptors in range of ITS index space, every item's entries covered once
  long nd = 0;
  struct View {
    const std::vector<gss::SegItem> *items;
    int64_t rows, entries;
  } views[3] = {{&it.shortv, n, nnz}, {&it.chunks, (int64_t)it.chunks.size(), nnz}, {&it.finish, n, (int64_t)it.chunks.size()}};
  for (const View &v : views) {
    std::vector<int32_t> segs;
    const int nblk = gss::build_segments_items(v.items->data(), v.items->size(), 16, gpw_log2, seg_edges, segs);
    const int ngb = 16 << gpw_log2;
    CHECK((size_t)nblk * ngb * 4 == segs.size(), "view: %d blocks, %zu ints", nblk, segs.size());
    std::vector<int64_t> got((size_t)v.rows, 0);
    for (size_t i = 0; i < segs.size() / 4; ++i) {
      const int32_t row = segs[i * 4], first = segs[i * 4 + 1], cnt = segs[i * 4 + 2];
      if (row < 0) continue;
      CHECK(row < v.rows && first >= 0 && cnt >= 0 && (int64_t)first + cnt <= v.entries, "view: descriptor %zu: row %d, entries [%d, +%d) of %lld", i, row, first, cnt, (long long)v.entries);
      got[(size_t)row] += cnt;
    }
    for (const gss::SegItem &s : *v.items) CHECK(got[(size_t)s.row] == s.len, "view: row %d scheduled with %lld of %d entries", s.row, (long long)got[(size_t)s.row], s.len);
    nd += (long)(segs.size() / 4);
  }
  return nd;
}

int main() {
  std::mt19937_64 rng(12345);
  long total = 0;
  int cases = 0;
  for (int trial = 0; trial < 60; ++trial) {
    const int n = trial == 0 ? 0 : trial == 1 ? 1 : (int)(rng() % 3000) + 1;
    std::vector<int32_t> rowptr((size_t)n + 1, 0);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (int r = 0; r < n; ++r) {
      double x = u(rng);
      int len = (int)(1.0 / (0.002 + x * x * 4.0));          // heavy tail: up to 500
      if (rng() % 7 == 0) len = 0;                            // empty rows
      if (rng() % 400 == 0) len = 20000 + (int)(rng() % 60000);   // a hub longer than a workgroup of 32-entry segments holds
      rowptr[(size_t)r + 1] = rowptr[(size_t)r] + len;
    }
    for (int gpw_log2 = 0; gpw_log2 <= 4; ++gpw_log2)
      for (int seg_edges : {4, 8, 32, 128, 1024}) {
        total += check_one(rowptr, 16, gpw_log2, seg_edges);
        ++cases;
        if (trial % 6 == 0 && (seg_edges == 32 || seg_edges == 8))
          for (int32_t thr : {64, 1000, 32768}) {
            total += check_giant(rowptr, thr, gpw_log2, seg_edges);
            ++cases;
          }
      }
  }
  printf("segments ok: %d schedules, %ld descriptors\n", cases, total);
  return 0;
}
