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
      }
  }
  printf("segments ok: %d schedules, %ld descriptors\n", cases, total);
  return 0;
}
