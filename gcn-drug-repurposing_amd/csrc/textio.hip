// textio.hip -- host-only: the writer of graph_embs.txt (train.py:193, np.savetxt(path, hidden_emb)) and the reader of the
// '.embs.txt' input (train.py:79-80), both on all host cores.
//
// np.savetxt prints every value of the float32 array with '%.18e' after widening it to double, ' ' between the values of
// a row, '\n' after each row; consumers read the file back with np.loadtxt (predict_drug.py:52).  Python's formatting
// costs ~0.3 us per value on one thread -- 1.2 s at N = 29,960, d = 128, 7 minutes at N = 10M -- more than the training
// run it follows.  Here the conversion is done exactly, for doubles that are widened float32s (24-bit significands):
//     |v| = m 2^e,  m < 2^24.   e < 0:  |v| = m 5^k / 10^k  (k = -e <= 149),   e >= 0:  |v| = m 2^e  (e <= 104)
// so the exact decimal expansion is the integer M = m 5^k (or m 2^e) -- at most 118 digits -- shifted by k places.  M is
// one multiplication of a precomputed base-10^9 power table by m; the leading 19 digits are rounded half-to-even on the
// exact remainder, which is what glibc's printf does.  Rows are formatted by a pool of threads into per-thread buffers and
// written in order.  Checked byte for byte against Python's '%.18e' in tests/test_host.py (random bit patterns, denormals,
// powers of two, ties).
#include <errno.h>
#include <stdlib.h>

#include <algorithm>

#include <exception>
#include <functional>
#include <memory>
#include <string>
#include <system_error>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace gss {
namespace {

constexpr int kLimbs = 14;          // 14 x 9 = 126 decimal digits >= 118
constexpr uint32_t kBase = 1000000000u;

struct PowTable {
  uint32_t p5[150][kLimbs];   // 5^k, little endian base 1e9
  uint32_t p2[105][kLimbs];   // 2^e
  uint8_t n5[150], n2[105];   // limbs in use
  char d2[100][2];            // two decimal digits of 0..99
  PowTable() {
    auto fill = [](uint32_t(*t)[kLimbs], uint8_t *len, int count, uint32_t mul) {
      for (int j = 0; j < kLimbs; ++j) t[0][j] = 0;
      t[0][0] = 1;
      len[0] = 1;
      for (int k = 1; k < count; ++k) {
        uint64_t carry = 0;
        for (int j = 0; j < kLimbs; ++j) {
          const uint64_t v = (uint64_t)t[k - 1][j] * mul + carry;
          t[k][j] = (uint32_t)(v % kBase);
          carry = v / kBase;
        }
        int n = kLimbs;
        while (n > 1 && t[k][n - 1] == 0) --n;
        len[k] = (uint8_t)n;
      }
    };
    fill(p5, n5, 150, 5);
    fill(p2, n2, 105, 2);
    for (int i = 0; i < 100; ++i) {
      d2[i][0] = (char)('0' + i / 10);
      d2[i][1] = (char)('0' + i % 10);
    }
  }
};
const PowTable &tables() {
  static const PowTable t;
  return t;
}

// '%.18e' of (double)f into out (at most 26 bytes incl. sign), returns the length
inline int format_e18(float f, char *out) {
  uint32_t bits;
  memcpy(&bits, &f, 4);
  char *p = out;
  if (bits >> 31) *p++ = '-';
  const uint32_t ex = (bits >> 23) & 0xff, frac = bits & 0x7fffff;
  if (ex == 0xff) {  // Python's '%.18e' prints 'inf' / '-inf' / 'nan' (no sign on a nan)
    if (frac) p = out;
    memcpy(p, frac ? "nan" : "inf", 3);
    return (int)(p - out) + 3;
  }
  if (ex == 0 && frac == 0) {
    memcpy(p, "0.000000000000000000e+00", 24);
    return (int)(p - out) + 24;
  }
  const uint32_t m = ex ? (frac | 0x800000u) : frac;
  const int e = (ex ? (int)ex : 1) - 150;  // |v| = m 2^e
  const PowTable &T = tables();
  const uint32_t *pw = e < 0 ? T.p5[-e] : T.p2[e];
  const int plen = e < 0 ? T.n5[-e] : T.n2[e];
  uint32_t limb[kLimbs + 1];
  uint64_t carry = 0;
  for (int j = 0; j < plen; ++j) {
    const uint64_t v = (uint64_t)pw[j] * m + carry;
    limb[j] = (uint32_t)(v % kBase);
    carry = v / kBase;
  }
  int top = plen - 1;
  if (carry) limb[++top] = (uint32_t)carry;   // m < 2^24 < 10^9: at most one more limb
  // only the leading digits matter: the top limb (>= 1 digit) and the three below it (27 digits: >= 20 digits whenever that
  // many exist), plus whether anything below those is non-zero
  char dig[40];
  int nd = 0;
  {
    uint32_t v = limb[top];
    char tmp[10];
    int len = 0;
    do {
      tmp[len++] = (char)('0' + v % 10);
      v /= 10;
    } while (v);
    for (int i = 0; i < len; ++i) dig[i] = tmp[len - 1 - i];
    nd = len;
  }
  int low = top - 1;
  for (int taken = 0; taken < 3 && low >= 0; ++taken, --low) {
    uint32_t v = limb[low];
    const uint32_t a2 = v % 100; v /= 100;
    const uint32_t b2 = v % 100; v /= 100;
    const uint32_t c2 = v % 100; v /= 100;
    const uint32_t d2 = v % 100; v /= 100;   // v is now the leading digit
    dig[nd] = (char)('0' + v);
    memcpy(dig + nd + 1, T.d2[d2], 2);
    memcpy(dig + nd + 3, T.d2[c2], 2);
    memcpy(dig + nd + 5, T.d2[b2], 2);
    memcpy(dig + nd + 7, T.d2[a2], 2);
    nd += 9;
  }
  bool tail = false;                          // a non-zero digit below the ones extracted
  for (int j = low; j >= 0 && !tail; --j) tail = limb[j] != 0;
  const int total_digits = nd + 9 * (low + 1);
  int dexp = total_digits - 1 - (e < 0 ? -e : 0);  // decimal exponent of the leading digit
  char sig[20];
  for (int i = 0; i < 19; ++i) sig[i] = i < nd ? dig[i] : '0';
  if (nd > 19) {  // round half to even on the exact remainder (nd <= 19 implies nothing below: all digits are in `dig`)
    bool up = false;
    if (dig[19] > '5') {
      up = true;
    } else if (dig[19] == '5') {
      bool rest = tail;
      for (int i = 20; i < nd && !rest; ++i) rest = dig[i] != '0';
      up = rest || ((sig[18] - '0') & 1);
    }
    if (up) {
      int i = 18;
      while (i >= 0 && sig[i] == '9') sig[i--] = '0';
      if (i >= 0) {
        ++sig[i];
      } else {  // 9.99...9 -> 1.00...0 e+1
        sig[0] = '1';
        ++dexp;
      }
    }
  }
  *p++ = sig[0];
  *p++ = '.';
  memcpy(p, sig + 1, 18);
  p += 18;
  *p++ = 'e';
  *p++ = dexp < 0 ? '-' : '+';
  const int ae = dexp < 0 ? -dexp : dexp;
  *p++ = (char)('0' + ae / 10);   // |exponent| <= 45 for float32
  *p++ = (char)('0' + ae % 10);
  return (int)(p - out);
}

}  // namespace
}  // namespace gss

// ---- reader of '.embs.txt' (train.py:79-80; written by multiscale/openne/node2vec.py:40-47) --------------------------------
struct gss_embs_file {
  int64_t n = 0, header_n = -1;
  int32_t d = 0;
  std::vector<double> x;
  std::string names;  // '\n'-joined
};

namespace gss {
namespace {
inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r'; }

// work(0) .. work(count - 1) on up to `count` host threads.  Thread creation can be refused (std::system_error: process / cgroup
// limits): the threads that did start are joined and the remaining items run inline -- unwinding a vector of joinable std::threads
// would call std::terminate before any catch is reached.
template <typename F>
void run_parallel(int count, F &&work) {
  if (count <= 1) {
    if (count == 1) work(0);
    return;
  }
  // an exception of any item -- thrown on a pool thread or by the inline leftovers -- is parked, every started thread is joined, and the
  // first one is rethrown on the caller's thread, where GSS_NOTHROW turns it into an error code (the inline loop used
  // to unwind past joinable threads)
  std::mutex mu;
  std::exception_ptr first;
  auto guarded = [&](int t) {
    try {
      work(t);
    } catch (...) {
      std::lock_guard<std::mutex> lk(mu);
      if (!first) first = std::current_exception();
    }
  };
  std::vector<std::thread> pool;
  pool.reserve((size_t)count);
  int started = 0;
  try {
    for (; started < count; ++started) pool.emplace_back(guarded, started);
  } catch (const std::system_error &) {
    // started threads keep running; the rest is done here
  }
  for (int t = started; t < count; ++t) guarded(t);
  for (auto &th : pool) th.join();
  if (first) std::rethrow_exception(first);
}

struct FileCloser {
  void operator()(FILE *f) const {
    if (f) fclose(f);
  }
};
using FilePtr = std::unique_ptr<FILE, FileCloser>;

// the whole file into `text`; false (and errno) on a read error
inline bool read_all(FILE *f, std::string &text) {
  char chunk[1 << 16];
  size_t got;
  while ((got = fread(chunk, 1, sizeof(chunk), f)) > 0) text.append(chunk, got);
  return ferror(f) == 0;
}

// a number as Python's float() reads the token [p, end): strtod's grammar minus C hex floats, "inf" / "nan" spellings (a weight or
// feature that is not finite is a broken file here) and anything behind the number inside the token
inline bool parse_number(const char *p, const char *end, double *out, const char **stop) {
  char *q = nullptr;
  const double v = strtod(p, &q);
  if (q == p || q > end) return false;
  for (const char *c = p; c < q; ++c) {
    const char ch = *c;
    const bool ok = (ch >= '0' && ch <= '9') || ch == '+' || ch == '-' || ch == '.' || ch == 'e' || ch == 'E';
    if (!ok) return false;   // 'x' (hex), 'i' / 'n' (inf, nan) ...
  }
  *out = v;
  *stop = q;
  return true;
}

// parse lines [l0, l1) (offsets into text, each [begin, end)) into x / names; returns false on a malformed line
bool parse_lines(const char *text, const std::vector<std::pair<size_t, size_t>> &lines, size_t l0, size_t l1, int d, double *x,
                 std::vector<std::pair<size_t, size_t>> &name_span) {
  for (size_t li = l0; li < l1; ++li) {
    const char *p = text + lines[li].first, *end = text + lines[li].second;
    while (p < end && is_space(*p)) ++p;
    const char *nb = p;
    while (p < end && !is_space(*p)) ++p;
    name_span[li] = {(size_t)(nb - text), (size_t)(p - text)};
    double *row = x + li * (size_t)d;
    for (int k = 0; k < d; ++k) {
      while (p < end && is_space(*p)) ++p;
      if (p >= end) return false;
      const char *q = nullptr;    // the line ends in '\n' or the buffer's terminating NUL: strtod stops there
      if (!parse_number(p, end, &row[k], &q)) return false;
      if (q < end && !is_space(*q)) return false;   // '1.5abc'
      p = q;
    }
    while (p < end && is_space(*p)) ++p;
    if (p != end) return false;  // more tokens than the first data line had
  }
  return true;
}
}  // namespace
}  // namespace gss

// ---- reader of a weighted edgelist ('u v w' per line, nx.write_weighted_edgelist, predict_drug.py:224-226) ----------------------
struct gss_edgelist_file {
  std::vector<int32_t> src, dst;
  std::vector<double> w;
  int64_t bad_line = -1;  // 0-based line of the first unknown node name, -1 if none
};

using namespace gss;

extern "C" {

// a C++ exception (out of memory on a huge file) must not cross the C boundary: reported as an error code.  (A refused thread creation
// never gets this far: run_parallel joins what started and finishes inline.)
#define GSS_NOTHROW(call, name)                                                   \
  try {                                                                           \
    return call;                                                                  \
  } catch (const std::exception &ex) {                                            \
    return fail(GSS_EINVAL, name ": %s", ex.what());                              \
  } catch (...) {                                                                 \
    return fail(GSS_EINVAL, name ": unknown C++ exception");                      \
  }

static int edgelist_open_impl(gss_edgelist_file **out, const char *path, const char *names, int64_t names_bytes, int64_t n_names, int32_t threads) {
  GSS_REQUIRE(out && path && (names || n_names == 0) && n_names >= 0, "edgelist_open: bad argument");
  // node name -> row (the .embs.txt order); views into the caller's '\n'-joined name buffer
  struct Key {
    const char *p;
    size_t len;
    bool operator==(const Key &o) const { return len == o.len && memcmp(p, o.p, len) == 0; }
  };
  struct KeyHash {
    size_t operator()(const Key &k) const {
      uint64_t h = 1469598103934665603ull;  // FNV-1a
      for (size_t i = 0; i < k.len; ++i) h = (h ^ (unsigned char)k.p[i]) * 1099511628211ull;
      return (size_t)h;
    }
  };
  std::unordered_map<Key, int32_t, KeyHash> index;
  index.reserve((size_t)n_names * 2);
  {
    const char *p = names, *end = names + names_bytes;
    int32_t row = 0;
    while (p < end && row < n_names) {
      const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
      const char *e = nl ? nl : end;
      index.emplace(Key{p, (size_t)(e - p)}, row++);
      p = e + 1;
    }
    GSS_REQUIRE(row == n_names, "edgelist_open: the name buffer holds %d names, %lld announced", row, (long long)n_names);
  }
  std::string text;
  {
    FilePtr f(fopen(path, "rb"));
    if (!f) return fail(GSS_EINVAL, "edgelist_open: cannot open %s: %s", path, strerror(errno));
    if (!read_all(f.get(), text)) return fail(GSS_EINVAL, "edgelist_open: read error on %s: %s", path, strerror(errno));
  }
  std::vector<std::pair<size_t, size_t>> lines;
  for (size_t pos = 0; pos < text.size();) {
    const char *nl = (const char *)memchr(text.data() + pos, '\n', text.size() - pos);
    const size_t end = nl ? (size_t)(nl - text.data()) : text.size();
    size_t s0 = pos;
    while (s0 < end && is_space(text[s0])) ++s0;
    if (s0 < end && text[s0] != '#') lines.push_back({s0, end});
    pos = end + 1;
  }
  std::unique_ptr<gss_edgelist_file> owner(new gss_edgelist_file());
  gss_edgelist_file *e = owner.get();
  const size_t m = lines.size();
  e->src.resize(m);
  e->dst.resize(m);
  e->w.resize(m);
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  nt = std::max(1, std::min(nt, 64));
  nt = (int)std::min<size_t>((size_t)nt, std::max<size_t>(1, m / 4096));
  std::vector<int64_t> bad((size_t)nt, -1);
  auto work = [&](int t) {
    const size_t l0 = m * (size_t)t / (size_t)nt, l1 = m * (size_t)(t + 1) / (size_t)nt;
    const char *base = text.c_str();
    for (size_t li = l0; li < l1; ++li) {
      const char *p = base + lines[li].first, *end = base + lines[li].second;
      const char *tok[2];
      size_t len[2];
      for (int k = 0; k < 2; ++k) {
        while (p < end && is_space(*p)) ++p;
        tok[k] = p;
        while (p < end && !is_space(*p)) ++p;
        len[k] = (size_t)(p - tok[k]);
      }
      const auto iu = index.find(Key{tok[0], len[0]}), iv = index.find(Key{tok[1], len[1]});
      if (len[1] == 0 || iu == index.end() || iv == index.end()) {
        if (bad[(size_t)t] < 0) bad[(size_t)t] = (int64_t)li;
        continue;
      }
      while (p < end && is_space(*p)) ++p;
      double wt = 1.0;  // an edgelist without weights: unit weights
      if (p < end) {
        // exactly one more token, a finite decimal number: anything else (hex, inf / nan, 'u v w extra') is reported as a bad line,
        // and the caller falls back to the Python parser and ITS error message (embio.read_edgelist)
        const char *q = nullptr;
        bool ok = parse_number(p, end, &wt, &q);
        if (ok) {
          while (q < end && is_space(*q)) ++q;
          ok = q == end;
        }
        if (!ok) {
          if (bad[(size_t)t] < 0) bad[(size_t)t] = (int64_t)li;
          continue;
        }
      }
      e->src[li] = iu->second;
      e->dst[li] = iv->second;
      e->w[li] = wt;
    }
  };
  run_parallel(nt, work);
  for (int64_t b : bad)
    if (b >= 0 && (e->bad_line < 0 || b < e->bad_line)) e->bad_line = b;
  *out = owner.release();
  return GSS_OK;
}
int gss_edgelist_open(gss_edgelist_file **out, const char *path, const char *names, int64_t names_bytes, int64_t n_names, int32_t threads) {
  GSS_NOTHROW(edgelist_open_impl(out, path, names, names_bytes, n_names, threads), "edgelist_open")
}
int64_t gss_edgelist_edges(const gss_edgelist_file *e) { return e ? (int64_t)e->src.size() : 0; }
int64_t gss_edgelist_bad_line(const gss_edgelist_file *e) { return e ? e->bad_line : -1; }
int gss_edgelist_copy(const gss_edgelist_file *e, int32_t *src, int32_t *dst, double *w) {
  GSS_REQUIRE(e && ((src && dst && w) || e->src.empty()), "edgelist_copy: null argument");
  if (!e->src.empty()) {
    memcpy(src, e->src.data(), sizeof(int32_t) * e->src.size());
    memcpy(dst, e->dst.data(), sizeof(int32_t) * e->dst.size());
    memcpy(w, e->w.data(), sizeof(double) * e->w.size());
  }
  return GSS_OK;
}
void gss_edgelist_close(gss_edgelist_file *e) { delete e; }

static int embs_open_impl(gss_embs_file **out, const char *path, int32_t threads) {
  GSS_REQUIRE(out && path, "embs_open: null argument");
  std::string text;
  {
    FilePtr f(fopen(path, "rb"));
    if (!f) return fail(GSS_EINVAL, "embs_open: cannot open %s: %s", path, strerror(errno));
    if (!read_all(f.get(), text)) return fail(GSS_EINVAL, "embs_open: read error on %s: %s", path, strerror(errno));
  }
  // line table (the first line is the '<N> <d>' header of node2vec.py:42; blank lines are skipped like the reference's loadtxt)
  std::vector<std::pair<size_t, size_t>> lines;
  size_t pos = 0;
  bool first = true;
  int64_t header_n = -1;
  while (pos < text.size()) {
    const char *nl = (const char *)memchr(text.data() + pos, '\n', text.size() - pos);
    const size_t end = nl ? (size_t)(nl - text.data()) : text.size();
    if (first) {
      first = false;
      const std::string h = text.substr(pos, end - pos);
      long long a = 0, b = 0;
      char extra;
      if (sscanf(h.c_str(), " %lld %lld %c", &a, &b, &extra) == 2) header_n = a;
    } else {
      size_t s0 = pos;
      while (s0 < end && is_space(text[s0])) ++s0;
      if (s0 < end) lines.push_back({pos, end});
    }
    pos = end + 1;
  }
  std::unique_ptr<gss_embs_file> owner(new gss_embs_file());
  gss_embs_file *e = owner.get();
  e->header_n = header_n;
  e->n = (int64_t)lines.size();
  if (e->n == 0) {
    *out = owner.release();
    return GSS_OK;
  }
  {  // d = tokens of the first data line - 1
    const char *p = text.data() + lines[0].first, *end = text.data() + lines[0].second;
    int tok = 0;
    while (p < end) {
      while (p < end && is_space(*p)) ++p;
      if (p < end) ++tok;
      while (p < end && !is_space(*p)) ++p;
    }
    e->d = tok - 1;
  }
  if (e->d < 1) {
    return fail(GSS_EINVAL, "embs_open: %s: the first data line has no values", path);
  }
  e->x.resize((size_t)e->n * (size_t)e->d);
  std::vector<std::pair<size_t, size_t>> span((size_t)e->n);
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  nt = std::max(1, std::min(nt, 64));
  nt = (int)std::min<int64_t>(nt, std::max<int64_t>(1, e->n / 256));
  std::vector<char> ok((size_t)nt, 1);
  auto work = [&](int t) {
    const size_t l0 = (size_t)e->n * (size_t)t / (size_t)nt, l1 = (size_t)e->n * (size_t)(t + 1) / (size_t)nt;
    ok[(size_t)t] = parse_lines(text.c_str(), lines, l0, l1, e->d, e->x.data(), span) ? 1 : 0;
  };
  run_parallel(nt, work);
  for (char c : ok)
    if (!c) return fail(GSS_EINVAL, "embs_open: %s: a line does not hold a name and %d numbers", path, e->d);
  size_t total = 0;
  for (auto &s : span) total += s.second - s.first + 1;
  e->names.reserve(total);
  for (auto &s : span) {
    e->names.append(text, s.first, s.second - s.first);
    e->names.push_back('\n');
  }
  *out = owner.release();
  return GSS_OK;
}
int gss_embs_open(gss_embs_file **out, const char *path, int32_t threads) { GSS_NOTHROW(embs_open_impl(out, path, threads), "embs_open") }

int64_t gss_embs_rows(const gss_embs_file *e) { return e ? e->n : 0; }
int32_t gss_embs_cols(const gss_embs_file *e) { return e ? e->d : 0; }
int64_t gss_embs_names_bytes(const gss_embs_file *e) { return e ? (int64_t)e->names.size() : 0; }
int gss_embs_copy(const gss_embs_file *e, double *x_out, char *names_out, int64_t names_cap, int64_t *header_n) {
  GSS_REQUIRE(e && (x_out || e->n == 0) && (names_out || e->names.empty()) && names_cap >= (int64_t)e->names.size(), "embs_copy: bad argument");
  if (!e->x.empty()) memcpy(x_out, e->x.data(), sizeof(double) * e->x.size());
  if (!e->names.empty()) memcpy(names_out, e->names.data(), e->names.size());
  if (header_n) *header_n = e->header_n;
  return GSS_OK;
}
void gss_embs_close(gss_embs_file *e) { delete e; }

int gss_format_e18(float value, char *out26) {
  GSS_REQUIRE(out26, "format_e18: null buffer");
  const int n = format_e18(value, out26);
  out26[n] = '\0';
  return n;
}

static int write_embs_text_impl(const char *path, const float *h_emb, int64_t n, int32_t d, int32_t threads) {
  GSS_REQUIRE(path && (h_emb || n == 0) && n >= 0 && d >= 1, "write_embs_text: bad argument");
  FilePtr fp(fopen(path, "wb"));   // closed on every path out of here, an exception included
  if (!fp) return fail(GSS_EINVAL, "write_embs_text: cannot open %s: %s", path, strerror(errno));
  FILE *f = fp.get();
  (void)tables();
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if (nt > 64) nt = 64;
  int64_t rows_per_block = (n + (int64_t)nt * 4 - 1) / ((int64_t)nt * 4);   // ~4 blocks per thread
  rows_per_block = std::max<int64_t>(64, std::min<int64_t>(rows_per_block, 8192));
  const int64_t blocks = (n + rows_per_block - 1) / rows_per_block;
  // waves of nt blocks: format in parallel, write in order (bounded memory: nt x 4096 rows x 26 d bytes)
  std::vector<std::string> buf((size_t)nt);
  int rc = GSS_OK;
  for (int64_t b0 = 0; b0 < blocks && rc == GSS_OK; b0 += nt) {
    const int cnt = (int)std::min<int64_t>(nt, blocks - b0);
    auto work = [&](int t) {
      std::string &s = buf[(size_t)t];
      const int64_t r0 = (b0 + t) * rows_per_block, r1 = std::min(n, r0 + rows_per_block);
      s.resize((size_t)(r1 - r0) * (size_t)d * 26);
      char *p = &s[0];
      for (int64_t r = r0; r < r1; ++r) {
        const float *row = h_emb + (size_t)r * d;
        for (int k = 0; k < d; ++k) {
          p += format_e18(row[k], p);
          *p++ = k + 1 < d ? ' ' : '\n';
        }
      }
      s.resize((size_t)(p - &s[0]));
    };
    run_parallel(cnt, work);
    for (int t = 0; t < cnt && rc == GSS_OK; ++t)
      if (fwrite(buf[(size_t)t].data(), 1, buf[(size_t)t].size(), f) != buf[(size_t)t].size())
        rc = fail(GSS_EINVAL, "write_embs_text: short write to %s: %s", path, strerror(errno));
  }
  if (fclose(fp.release()) != 0 && rc == GSS_OK) rc = fail(GSS_EINVAL, "write_embs_text: close of %s failed: %s", path, strerror(errno));
  return rc;
}
int gss_write_embs_text(const char *path, const float *h_emb, int64_t n, int32_t d, int32_t threads) {
  GSS_NOTHROW(write_embs_text_impl(path, h_emb, n, d, threads), "write_embs_text")
}
}
