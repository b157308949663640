// Host-side sanitizer run of csrc/textio.hip (no device code in that file; GPU ASan is not available on the pool), built by
// tests/test_native_host.py with hipcc --offload-host-only -fsanitize=address,undefined.  Writes a matrix, reads it back, feeds the two
// readers well-formed, ragged, empty, non-finite and random-garbage files, formats random float32 bit patterns against snprintf.
// TEST INFRASTRUCTURE.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <string>
#include <vector>

#include "gssgcn.h"

namespace gss {
thread_local char g_err[512] = "";
}
extern "C" const char *gss_last_error(void) { return gss::g_err; }

#define CHECK(cond, ...)                                      \
  do {                                                        \
    if (!(cond)) {                                            \
      fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__);  \
      fprintf(stderr, __VA_ARGS__);                           \
      fprintf(stderr, " [%s]\n", gss::g_err);                 \
      exit(1);                                                \
    }                                                         \
  } while (0)

static void put(const std::string &path, const std::string &text) {
  FILE *f = fopen(path.c_str(), "wb");
  CHECK(f, "cannot write %s", path.c_str());
  fwrite(text.data(), 1, text.size(), f);
  fclose(f);
}

int main(int argc, char **argv) {
  CHECK(argc == 2, "usage: textio_check <scratch dir>");
  const std::string dir = argv[1];
  std::mt19937_64 rng(7);
  // 1. the formatter against glibc on random bit patterns and the specials
  char a[32], b[64];
  const uint32_t specials[] = {0u, 0x80000000u, 1u, 0x7f7fffffu, 0x00800000u, 0x007fffffu, 0x3f800000u, 0x33800000u};
  for (int i = 0; i < 60000; ++i) {
    uint32_t bits = i < 8 ? specials[i] : (uint32_t)rng();
    if ((bits & 0x7f800000u) == 0x7f800000u) bits &= 0x7f7fffffu;   // finite values only (np.savetxt never sees others here)
    float v;
    memcpy(&v, &bits, 4);
    const int n = gss_format_e18(v, a);
    snprintf(b, sizeof(b), "%.18e", (double)v);
    CHECK(n == (int)strlen(b) && strcmp(a, b) == 0, "format_e18(%08x) = '%s', printf '%s'", bits, a, b);
  }
  // 2. writer -> reader round trip (the reader wants a name column and a header: build such a file from the writer's rows)
  const int64_t n = 3000;
  const int d = 37;
  std::vector<float> x((size_t)n * d);
  std::normal_distribution<float> g(0.f, 1.f);
  for (auto &v : x) v = g(rng);
  const std::string out = dir + "/o.txt";
  for (int threads : {1, 3, 7}) CHECK(gss_write_embs_text(out.c_str(), x.data(), n, d, threads) == 0, "write_embs_text");
  CHECK(gss_write_embs_text((dir + "/nope/o.txt").c_str(), x.data(), n, d, 2) != 0, "write into a missing directory succeeded");
  {
    FILE *f = fopen(out.c_str(), "rb");
    std::string text, line;
    char chunk[65536];
    size_t got;
    while ((got = fread(chunk, 1, sizeof(chunk), f)) > 0) text.append(chunk, got);
    fclose(f);
    std::string embs = std::to_string(n) + " " + std::to_string(d) + "\n";
    size_t pos = 0;
    int64_t r = 0;
    while (pos < text.size()) {
      const size_t e = text.find('\n', pos);
      embs += "node" + std::to_string(r++) + " " + text.substr(pos, e - pos) + "\n";
      pos = e + 1;
    }
    CHECK(r == n, "writer produced %lld rows", (long long)r);
    put(dir + "/in.embs.txt", embs);
  }
  for (int threads : {1, 5}) {
    gss_embs_file *h = nullptr;
    CHECK(gss_embs_open(&h, (dir + "/in.embs.txt").c_str(), threads) == 0, "embs_open");
    CHECK(gss_embs_rows(h) == n && gss_embs_cols(h) == d, "embs shape %lld x %d", (long long)gss_embs_rows(h), gss_embs_cols(h));
    std::vector<double> back((size_t)n * d);
    std::vector<char> names((size_t)gss_embs_names_bytes(h));
    int64_t hn = 0;
    CHECK(gss_embs_copy(h, back.data(), names.data(), (int64_t)names.size(), &hn) == 0 && hn == n, "embs_copy");
    for (size_t i = 0; i < back.size(); ++i) CHECK((float)back[i] == x[i], "round trip differs at %zu", i);
    gss_embs_close(h);
  }
  // 3. malformed inputs: every one must be refused (or read as empty), none may touch memory it does not own
  const char *bad[] = {"", "2 2\n", "2 2\nn0 1 2\nn1 3\n", "2 2\nn0 1 2 3\nn1 4 5 6 7\n", "2 2\nn0 1 x\n", "2 2\nn0 inf 1\n", "2 2\nn0 0x10 1\n",
                       "2 2\nn0\n", "\n\n\n", "2 2\nn0 1e999999 2\n", "2 2\nn0 1.5abc 2\n"};
  for (const char *t : bad) {
    put(dir + "/bad.txt", t);
    gss_embs_file *h = nullptr;
    if (gss_embs_open(&h, (dir + "/bad.txt").c_str(), 3) == 0) {
      CHECK(gss_embs_rows(h) == 0 || strstr(t, "1e999999"), "malformed file '%s' was read as %lld rows", t, (long long)gss_embs_rows(h));
      gss_embs_close(h);
    }
  }
  for (int i = 0; i < 200; ++i) {   // random garbage
    std::string junk((size_t)(rng() % 4000), ' ');
    for (auto &c : junk) c = (char)(rng() % 96 + 9);
    put(dir + "/junk.txt", junk);
    gss_embs_file *h = nullptr;
    if (gss_embs_open(&h, (dir + "/junk.txt").c_str(), 2) == 0) gss_embs_close(h);
    gss_edgelist_file *e = nullptr;
    const char names_[] = "a\nb\nc";
    if (gss_edgelist_open(&e, (dir + "/junk.txt").c_str(), names_, 5, 3, 2) == 0) gss_edgelist_close(e);
  }
  // 4. the edgelist reader
  {
    const char names_[] = "a\nb\nc";
    std::string el = "# comment\n\na b 0.5\nb c\nc a 2e-3\n";
    for (int i = 0; i < 30000; ++i) el += (i % 2 ? "a c " : "b a ") + std::to_string(i * 0.25) + "\n";
    put(dir + "/g.edgelist", el);
    gss_edgelist_file *e = nullptr;
    CHECK(gss_edgelist_open(&e, (dir + "/g.edgelist").c_str(), names_, 5, 3, 4) == 0, "edgelist_open");
    CHECK(gss_edgelist_bad_line(e) == -1 && gss_edgelist_edges(e) == 30003, "edgelist: %lld edges, bad line %lld", (long long)gss_edgelist_edges(e),
          (long long)gss_edgelist_bad_line(e));
    std::vector<int32_t> s((size_t)30003), t((size_t)30003);
    std::vector<double> w((size_t)30003);
    CHECK(gss_edgelist_copy(e, s.data(), t.data(), w.data()) == 0 && s[0] == 0 && t[0] == 1 && w[0] == 0.5 && w[1] == 1.0 && w[2] == 2e-3, "edgelist_copy");
    gss_edgelist_close(e);
    for (const char *line : {"a b inf\n", "a b 0x1p3\n", "a b 1 2\n", "a zzz 1\n", "a\n", "a b nan\n"}) {
      put(dir + "/g2.edgelist", std::string("a b 1\n") + line);
      CHECK(gss_edgelist_open(&e, (dir + "/g2.edgelist").c_str(), names_, 5, 3, 1) == 0, "edgelist_open 2");
      CHECK(gss_edgelist_bad_line(e) == 1, "line '%s' was accepted", line);
      gss_edgelist_close(e);
    }
    CHECK(gss_edgelist_open(&e, (dir + "/g.edgelist").c_str(), names_, 5, 4, 1) != 0, "a short name table was accepted");
  }
  printf("textio ok\n");
  return 0;
}
