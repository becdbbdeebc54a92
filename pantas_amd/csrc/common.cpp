#include "common.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cerrno>
#include <exception>
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

PhaseTimer::PhaseTimer(const char* w) : what(w), t0(now_s()), last(t0), on(getenv("PANTAS_TIMING") != nullptr) {}
void PhaseTimer::lap(const char* phase) {
  if (!on) return;
  double t = now_s();
  fprintf(stderr, "[timing] %s: %-28s %8.3f s (total %8.3f)\n", what, phase, t - last, t - t0);
  last = t;
}

double now_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace pantas
