#include "common.h"
#include <cstring>
#include <cstdio>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cerrno>
#include <cstdlib>
#include <exception>
#include <new>
#include <condition_variable>
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

unsigned writer_threads() {
  static unsigned n = [] {
    const char* e = getenv("PANTAS_WRITER_THREADS");
    if (e && atoi(e) > 0) return unsigned(atoi(e));
    return hw_threads();
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

// ---- OrderedOutput ---------------------------------------------------------------------------------------------------
struct OrderedOutput::Impl {
  int fd;
  int map_fd = -1;                  // descriptor the mapping and the truncations go through (fd itself or a read-write reopening)
  bool own_map_fd = false;
  int mode;                         // 0 stream (every piece written by the thread that made it, in turn), 1 pwrite, 2 mmap, 3 one writer thread
  // mode 3
  struct Slot { const char* p = nullptr; size_t n = 0; bool ready = false, done = false; };
  std::vector<Slot> slots;
  std::mutex mu;
  static constexpr size_t DONE_CVS = 128;      // a maker waits on the variable of its piece's residue: a finished write wakes one thread,
  std::condition_variable cv_ready, cv_done[DONE_CVS];      // not every maker that is waiting
  std::thread writer;
  bool failed = false;
  std::string failure;
  off_t file_pos = 0;               // where the output starts in the file
  std::vector<std::atomic<uint64_t>> end_of;
  char* map = nullptr;              // maps [map_base, map_base + map_len) of the file
  uint64_t map_base = 0, map_len = 0;
  std::mutex grow_mu;
  std::atomic<uint64_t> file_size{0};   // current (sparse) size of the file while mapped
  uint64_t total = 0;
  explicit Impl(size_t n) : end_of(n + 1) {}
};

OrderedOutput::OrderedOutput(int fd, size_t n_pieces, uint64_t expected_bytes, uint64_t bound_bytes) : im(new Impl(n_pieces)) {
  im->fd = fd;
  im->mode = 0;
  for (auto& x : im->end_of) x.store(UINT64_MAX, std::memory_order_relaxed);
  im->end_of[0].store(0, std::memory_order_relaxed);
  if (fd < 0) return;
  if (getenv("PANTAS_WRITER_SEQUENTIAL")) return;
  struct stat st;
  const int fl = fcntl(fd, F_GETFL);
  const char* pw = getenv("PANTAS_OUTPUT_PWRITE");
  const char* mm = getenv("PANTAS_OUTPUT_MMAP");
  const bool positional = ((pw && !strcmp(pw, "1")) || (mm && !strcmp(mm, "1"))) && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && fl != -1 && !(fl & O_APPEND);
  if (!positional) {
    // One thread does every write, in piece order, back to back.  Many threads writing one file queue for the file's lock: on the
    // GPU box's tmpfs (16 cores of CPU quota) 2 - 64 threads pwrite at 3 - 4 GB/s, now and then 9; ONE thread does 9 GB/s every time
    // (profiles/r2_pwrite_probe.txt), and threads that wait on a condition variable instead of a lock or a spin leave the quota to
    // the formatting.  The makers of the pieces block in put() until their piece is written (their buffers are the queue).
    im->mode = 3;
    im->slots.resize(n_pieces);
    Impl* q = im;
    im->writer = std::thread([q] {
      const bool timing = getenv("PANTAS_TIMING") != nullptr;
      double waited = 0, wrote = 0;
      uint64_t bytes = 0;
      for (size_t k = 0; k < q->slots.size(); ++k) {
        const char* p;
        size_t left;
        const double t0 = timing ? now_s() : 0;
        {
          std::unique_lock<std::mutex> lk(q->mu);
          q->cv_ready.wait(lk, [&] { return q->slots[k].ready || q->failed; });
          if (q->failed) break;
          p = q->slots[k].p;
          left = q->slots[k].n;
        }
        const double t1 = timing ? now_s() : 0;
        waited += t1 - t0;
        bytes += left;
        while (left) {
          const ssize_t w = ::write(q->fd, p, left);
          if (w < 0) {
            if (errno == EINTR) continue;
            std::lock_guard<std::mutex> lk(q->mu);
            q->failed = true;
            q->failure = std::string("write failed: ") + strerror(errno);
            break;
          }
          p += w; left -= size_t(w);
        }
        if (timing) wrote += now_s() - t1;
        std::lock_guard<std::mutex> lk(q->mu);
        if (q->failed) break;
        q->total += q->slots[k].n;
        q->slots[k].done = true;
        q->cv_done[k % Impl::DONE_CVS].notify_all();
      }
      if (timing) fprintf(stderr, "[timing] writer thread: %.3f s in write() for %.2f GB (%.2f GB/s), %.3f s waiting for the next piece, %zu pieces\n",
                          wrote, bytes / 1e9, wrote > 0 ? bytes / 1e9 / wrote : 0.0, waited, q->slots.size());
      std::lock_guard<std::mutex> lk(q->mu);
      for (auto& cv : q->cv_done) cv.notify_all();
    });
    return;
  }
  const off_t pos = lseek(fd, 0, SEEK_CUR);
  if (pos < 0) return;
  im->file_pos = pos;
  im->mode = 1;
  // the shared mapping is opt-in: on the GPU box's tmpfs it was 4x SLOWER than pwrite (22.4 s against 5.6 s for 23 GB on
  // 64 threads: the page faults of a shared file mapping contend harder than the writes they were meant to spread)
  const char* e = getenv("PANTAS_OUTPUT_MMAP");
  if (!e || strcmp(e, "1")) return;
  im->map_fd = fd;
  if ((fl & O_ACCMODE) != O_RDWR) {
    // a shared writable mapping needs a descriptor opened for reading too, and the shell's `> file` is write-only:
    // the same file once more through /proc (no /proc, no permission, another file: pwrite then)
    const std::string link = "/proc/self/fd/" + std::to_string(fd);
    const int fd2 = ::open(link.c_str(), O_RDWR);
    if (fd2 < 0) return;
    struct stat st2;
    if (fstat(fd2, &st2) != 0 || st2.st_dev != st.st_dev || st2.st_ino != st.st_ino) { ::close(fd2); return; }
    im->map_fd = fd2;
    im->own_map_fd = true;
  }
  struct statvfs vs;
  if (fstatvfs(fd, &vs) != 0 || uint64_t(vs.f_bavail) * vs.f_frsize < expected_bytes + expected_bytes / 8 + (uint64_t(1) << 30)) return;
  const uint64_t page = 4096;
  im->map_base = uint64_t(pos) / page * page;
  im->map_len = (uint64_t(pos) - im->map_base + bound_bytes + page - 1) / page * page;
  void* m = mmap(nullptr, im->map_len, PROT_READ | PROT_WRITE, MAP_SHARED, im->map_fd, off_t(im->map_base));
  if (m == MAP_FAILED) return;
  im->map = static_cast<char*>(m);
  im->file_size.store(uint64_t(st.st_size));
  im->mode = 2;
}

OrderedOutput::~OrderedOutput() {
  if (im->writer.joinable()) {
    { std::lock_guard<std::mutex> lk(im->mu); im->failed = true; }      // (a maker threw: nobody will bring the missing pieces)
    im->cv_ready.notify_all();
    im->writer.join();
  }
  if (im->map) munmap(im->map, im->map_len);
  if (im->own_map_fd && im->map_fd >= 0) ::close(im->map_fd);
  delete im;
}

const char* OrderedOutput::mode_name() const {
  return im->mode == 3 ? "one writer thread" : im->mode == 2 ? "shared mapping" : im->mode == 1 ? "pwrite" : "write in order";
}

void OrderedOutput::put(size_t piece, const char* data, size_t n) {
  if (im->mode == 3) {
    std::unique_lock<std::mutex> lk(im->mu);
    Impl::Slot& sl = im->slots[piece];
    sl.p = data; sl.n = n; sl.ready = true;
    im->cv_ready.notify_one();
    im->cv_done[piece % Impl::DONE_CVS].wait(lk, [&] { return sl.done || im->failed; });
    if (im->failed) throw Error("OSError", im->failure.empty() ? std::string("output abandoned") : im->failure);
    return;
  }
  uint64_t start;
  for (unsigned spins = 0; (start = im->end_of[piece].load(std::memory_order_acquire)) == UINT64_MAX; ++spins) {
    if (spins < 200) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  auto fail = [&](const char* what) {
    im->end_of[piece + 1].store(start + n, std::memory_order_release);     // nobody may wait for ever
    throw Error("OSError", std::string(what) + ": " + strerror(errno));
  };
  if (im->mode == 0) {
    if (im->fd >= 0) {
      const char* p = data;
      size_t left = n;
      while (left) {
        const ssize_t w = ::write(im->fd, p, left);
        if (w < 0) { if (errno == EINTR) continue; fail("write failed"); }
        p += w; left -= size_t(w);
      }
    }
    im->end_of[piece + 1].store(start + n, std::memory_order_release);
    return;
  }
  im->end_of[piece + 1].store(start + n, std::memory_order_release);
  const uint64_t at = uint64_t(im->file_pos) + start;
  if (im->mode == 2 && at - im->map_base + n <= im->map_len) {
    // the file has to reach behind the piece before its pages may be touched (beyond the end of the file: SIGBUS)
    const uint64_t need = at + n;
    if (im->file_size.load(std::memory_order_acquire) < need) {
      std::lock_guard<std::mutex> lk(im->grow_mu);
      uint64_t cur = im->file_size.load(std::memory_order_relaxed);
      if (cur < need) {
        const uint64_t to = std::min<uint64_t>(im->map_base + im->map_len, std::max<uint64_t>(need + (uint64_t(2) << 30), cur + cur / 2));
        if (ftruncate(im->map_fd, off_t(to)) != 0) throw Error("OSError", std::string("ftruncate failed: ") + strerror(errno));
        im->file_size.store(to, std::memory_order_release);
      }
    }
    memcpy(im->map + (at - im->map_base), data, n);
    return;
  }
  const char* p = data;
  size_t left = n;
  off_t o = off_t(at);
  while (left) {
    const ssize_t w = ::pwrite(im->fd, p, left, o);
    if (w < 0) { if (errno == EINTR) continue; throw Error("OSError", std::string("write failed: ") + strerror(errno)); }
    p += w; left -= size_t(w); o += w;
  }
}

void OrderedOutput::finish() {
  if (im->mode == 3) {
    if (im->writer.joinable()) im->writer.join();
    if (im->failed) throw Error("OSError", im->failure.empty() ? std::string("output abandoned") : im->failure);
    for (const Impl::Slot& sl : im->slots)
      if (!sl.done) throw Error("InternalError", "OrderedOutput::finish before every piece was put");
    return;
  }
  const uint64_t total = im->end_of.back().load(std::memory_order_acquire);
  if (total == UINT64_MAX) throw Error("InternalError", "OrderedOutput::finish before every piece was put");
  if (im->mode == 0) return;
  const off_t end = im->file_pos + off_t(total);
  if (im->mode == 2) {
    munmap(im->map, im->map_len);
    im->map = nullptr;
    if (ftruncate(im->map_fd, std::max<off_t>(end, 0)) != 0) throw Error("OSError", std::string("ftruncate failed: ") + strerror(errno));
  }
  if (lseek(im->fd, end, SEEK_SET) < 0) throw Error("OSError", "lseek failed on the output");
}

double now_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

namespace {
std::mutex g_deferred_mu;
std::vector<std::function<void()>> g_deferred;

size_t mem_available_bytes() {
  FILE* f = fopen("/proc/meminfo", "r");
  if (!f) return SIZE_MAX;
  char line[256];
  size_t kb = SIZE_MAX;
  while (fgets(line, sizeof line, f)) {
    unsigned long long v = 0;
    if (sscanf(line, "MemAvailable: %llu kB", &v) == 1) { kb = size_t(v); break; }
  }
  fclose(f);
  return kb == SIZE_MAX ? SIZE_MAX : kb * 1024;
}
}  // namespace

void defer_free(size_t bytes, std::function<void()> fn) {
  if (getenv("PANTAS_KEEP_TEMPORARIES")) return;            // experiment switch: leak them
  const size_t avail = mem_available_bytes();
  if (getenv("PANTAS_FREE_AT_ONCE") || avail / 4 < bytes) {
    std::thread(std::move(fn)).detach();
    return;
  }
  std::lock_guard<std::mutex> lk(g_deferred_mu);
  g_deferred.push_back(std::move(fn));
}

void release_deferred() {
  std::vector<std::function<void()>> todo;
  {
    std::lock_guard<std::mutex> lk(g_deferred_mu);
    todo.swap(g_deferred);
  }
  if (todo.empty()) return;
  std::thread([todo = std::move(todo)] { for (const auto& fn : todo) fn(); }).detach();
}

bool is_lone_cr_error(const std::exception& e) { return strstr(e.what(), LONE_CR_MARK) != nullptr; }

bool has_lone_cr(const char* p, size_t n) {
  const char* q = p;
  const char* end = p + n;
  while (q < end) {
    q = static_cast<const char*>(memchr(q, '\r', size_t(end - q)));
    if (!q) return false;
    if (q + 1 < end && q[1] != '\n') return true;
    ++q;
  }
  return false;
}

void universal_newlines_copy(const char* p, size_t n, RawVector<char>& out) {
  out.resize(n);
  char* o = out.data();
  parallel_chunks(n, n < (size_t(1) << 22) ? 1u : hw_threads(), [&](unsigned, size_t b, size_t e) {
    memcpy(o + b, p + b, e - b);
    for (size_t i = b; i < e;) {
      const char* q = static_cast<const char*>(memchr(o + i, '\r', e - i));
      if (!q) break;
      const size_t at = size_t(q - o);
      if (at + 1 < n && p[at + 1] != '\n') o[at] = '\n';      // (p, not o: the next byte may belong to another thread's chunk)
      i = at + 1;
    }
  });
}

}  // namespace pantas
