#include "common.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cerrno>
#include <cstdlib>
#include <exception>
#include <new>
#include <mutex>

namespace pantas {

MappedFile::~MappedFile() { close(); }

void MappedFile::open(const char* path) {
  close();
  fd = ::open(path, O_RDONLY);
  if (fd < 0) throw Error(errno == ENOENT ? "FileNotFoundError" : "OSError", std::string(path) + ": " + strerror(errno));
  struct stat st;
  if (fstat(fd, &st) != 0) { ::close(fd); fd = -1; throw Error("OSError", std::string(path) + ": fstat failed"); }
  size = size_t(st.st_size);
  if (size == 0) { data = ""; return; }
  void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  if (p == MAP_FAILED) { ::close(fd); fd = -1; size = 0; throw Error("OSError", std::string(path) + ": mmap failed"); }
  madvise(p, size, MADV_SEQUENTIAL);
  data = static_cast<const char*>(p);
}

void MappedFile::close() {
  if (data && size && !borrowed) munmap(const_cast<char*>(data), size);
  borrowed = false;
  if (fd >= 0) ::close(fd);
  data = nullptr;
  size = 0;
  fd = -1;
}

void* big_alloc(size_t bytes) {
  constexpr size_t HUGE = size_t(2) << 20;
  if (bytes < 4 * HUGE) {
    void* p = ::operator new(bytes ? bytes : 1);
    return p;
  }
  const size_t rounded = (bytes + HUGE - 1) / HUGE * HUGE;
  void* p = aligned_alloc(HUGE, rounded);
  if (!p) throw std::bad_alloc();
  (void)madvise(p, rounded, MADV_HUGEPAGE);      // advice only: without transparent huge pages this is an ordinary block
  return p;
}

void big_free(void* p, size_t bytes) {
  constexpr size_t HUGE = size_t(2) << 20;
  if (bytes < 4 * HUGE) ::operator delete(p);
  else free(p);
}

unsigned hw_threads() {
  static unsigned n = [] {
    const char* e = getenv("PANTAS_THREADS");
    if (e && atoi(e) > 0) return unsigned(atoi(e));
    unsigned h = std::thread::hardware_concurrency();
    if (h == 0) h = 1;
    return std::min(h, 64u);
  }();
  return n;
}

void parallel_chunks(size_t n, unsigned chunks, const std::function<void(unsigned, size_t, size_t)>& fn) {
  if (chunks == 0) chunks = 1;
  if (chunks == 1 || n == 0) { fn(0, 0, n); return; }
  std::vector<std::thread> th;
  std::exception_ptr first;
  std::mutex mu;
  for (unsigned c = 0; c < chunks; ++c) {
    size_t b = n * c / chunks, e = n * (c + 1) / chunks;
    th.emplace_back([&, c, b, e] {
      try { fn(c, b, e); } catch (...) { std::lock_guard<std::mutex> g(mu); if (!first) first = std::current_exception(); }
    });
  }
  for (auto& t : th) t.join();
  if (first) std::rethrow_exception(first);
}

std::vector<std::pair<size_t, size_t>> split_at_newlines(const char* data, size_t size, unsigned want) {
  std::vector<std::pair<size_t, size_t>> out;
  if (want == 0) want = 1;
  size_t begin = 0;
  for (unsigned c = 1; c <= want && begin < size; ++c) {
    size_t end = (c == want) ? size : std::max(begin, size * c / want);
    if (end < size) {
      const void* nl = memchr(data + end, '\n', size - end);
      end = nl ? size_t(static_cast<const char*>(nl) - data) + 1 : size;
    }
    if (end > begin) out.emplace_back(begin, end);
    begin = end;
  }
  if (out.empty()) out.emplace_back(0, size);
  return out;
}

void prefault(void* p, size_t bytes) {
  if (bytes < (size_t(64) << 20)) return;
  char* c = static_cast<char*>(p);
  const size_t pages = (bytes + 4095) / 4096;
  parallel_chunks(pages, std::min<unsigned>(hw_threads(), 64), [&](unsigned, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
      volatile char* q = c + std::min(i * 4096, bytes - 1);
      *q = 0;
    }
  });
}

PhaseTimer::PhaseTimer(const char* w) : what(w), t0(now_s()), last(t0), on(getenv("PANTAS_TIMING") != nullptr) {}
void PhaseTimer::lap(const char* phase) {
  if (!on) return;
  double t = now_s();
  fprintf(stderr, "[timing] %s: %-28s %8.3f s (total %8.3f)\n", what, phase, t - last, t - t0);
  last = t;
}

// parallel LSD radix sort of (key, idx) pairs, 16-bit digits
template <class Vec>
static void radix_sort_pairs_impl(Vec& key, Vec& idx, unsigned threads) {
  const size_t n = key.size();
  Vec key2(n), idx2(n);
  const unsigned T = n < (1u << 16) ? 1 : threads;
  uint32_t maxk = 0;
  for (uint32_t k : key) maxk = std::max(maxk, k);
  for (int pass = 0; pass < 2; ++pass) {
    const int shift = pass * 16;
    if (pass == 1 && (maxk >> 16) == 0) break;
    std::vector<std::vector<uint32_t>> hist(T, std::vector<uint32_t>(65536, 0));
    parallel_chunks(n, T, [&](unsigned c, size_t b, size_t e) {
      uint32_t* h = hist[c].data();
      for (size_t i = b; i < e; ++i) ++h[(key[i] >> shift) & 0xFFFFu];
    });
    uint64_t run = 0;
    for (uint32_t d = 0; d < 65536; ++d)
      for (unsigned c = 0; c < T; ++c) { uint32_t x = hist[c][d]; hist[c][d] = uint32_t(run); run += x; }
    parallel_chunks(n, T, [&](unsigned c, size_t b, size_t e) {
      uint32_t* h = hist[c].data();
      for (size_t i = b; i < e; ++i) {
        const uint32_t o = h[(key[i] >> shift) & 0xFFFFu]++;
        key2[o] = key[i];
        idx2[o] = idx[i];
      }
    });
    key.swap(key2);
    idx.swap(idx2);
  }
}


void radix_sort_pairs(RawVector<uint32_t>& key, RawVector<uint32_t>& idx, unsigned threads) { radix_sort_pairs_impl(key, idx, threads); }
void radix_sort_pairs(std::vector<uint32_t>& key, std::vector<uint32_t>& idx, unsigned threads) { radix_sort_pairs_impl(key, idx, threads); }

double now_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace pantas
