// Probe (GPU box, host side only): how fast can a whole-genome GFA in the page cache / tmpfs be brought under a parser?
//   hostio_probe <file> <threads>
// mmap + first-touch scan, mmap + MADV_POPULATE_READ per thread range, MAP_POPULATE, parallel pread into a huge-page buffer.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class F> static void par(unsigned T, size_t n, F f) {
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; ++t) th.emplace_back([=] { f(t, n / T * t, t + 1 == T ? n : n / T * (t + 1)); });
  for (auto& x : th) x.join();
}
static uint64_t count_nl(const char* p, size_t T_, size_t size, unsigned T) {
  std::vector<uint64_t> c(T, 0);
  par(T, size, [&](unsigned t, size_t b, size_t e) {
    uint64_t n = 0;
    const char* q = p + b;
    const char* end = p + e;
    while (q < end) { const void* x = memchr(q, '\n', size_t(end - q)); if (!x) break; ++n; q = static_cast<const char*>(x) + 1; }
    c[t] = n;
  });
  uint64_t s = 0; for (auto x : c) s += x; (void)T_; return s;
}
int main(int argc, char** argv) {
  const char* path = argv[1];
  const unsigned T = unsigned(atoi(argv[2]));
  int fd = open(path, O_RDONLY);
  struct stat st; fstat(fd, &st);
  const size_t size = size_t(st.st_size);
  printf("file %.2f GB, %u threads\n", size / 1e9, T);
  {
    double t0 = now();
    char* p = static_cast<char*>(mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0));
    uint64_t n = count_nl(p, 0, size, T);
    double t1 = now();
    uint64_t n2 = count_nl(p, 0, size, T);
    double t2 = now();
    munmap(p, size);
    double t3 = now();
    printf("A mmap first-touch scan   %.3f s (%.1f GB/s)   second scan %.3f s (%.1f GB/s)   munmap %.3f s   lines %llu %llu\n", t1 - t0, size / 1e9 / (t1 - t0), t2 - t1,
           size / 1e9 / (t2 - t1), t3 - t2, (unsigned long long)n, (unsigned long long)n2);
  }
  {
    double t0 = now();
    char* p = static_cast<char*>(mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0));
    par(T, size, [&](unsigned, size_t b, size_t e) {
      b &= ~size_t(4095);
      e = (e + 4095) & ~size_t(4095);
      if (e > size) e = (size + 4095) & ~size_t(4095);
      if (madvise(p + b, e - b, MADV_POPULATE_READ) != 0) perror("madvise");
    });
    double t1 = now();
    uint64_t n = count_nl(p, 0, size, T);
    double t2 = now();
    munmap(p, size);
    printf("B MADV_POPULATE_READ par  %.3f s, scan %.3f s (%.1f GB/s)  lines %llu\n", t1 - t0, t2 - t1, size / 1e9 / (t2 - t1), (unsigned long long)n);
  }
  {
    double t0 = now();
    char* p = static_cast<char*>(mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0));
    double t1 = now();
    uint64_t n = count_nl(p, 0, size, T);
    double t2 = now();
    munmap(p, size);
    printf("C MAP_POPULATE            %.3f s, scan %.3f s  lines %llu\n", t1 - t0, t2 - t1, (unsigned long long)n);
  }
  {
    double t0 = now();
    char* buf = static_cast<char*>(mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    madvise(buf, size, MADV_HUGEPAGE);
    par(T, size, [&](unsigned, size_t b, size_t e) {
      while (b < e) { ssize_t r = pread(fd, buf + b, std::min<size_t>(e - b, size_t(16) << 20), off_t(b)); if (r <= 0) break; b += size_t(r); }
    });
    double t1 = now();
    uint64_t n = count_nl(buf, 0, size, T);
    double t2 = now();
    munmap(buf, size);
    double t3 = now();
    printf("D parallel pread -> THP   %.3f s (%.1f GB/s), scan %.3f s (%.1f GB/s), munmap %.3f  lines %llu\n", t1 - t0, size / 1e9 / (t1 - t0), t2 - t1, size / 1e9 / (t2 - t1), t3 - t2,
           (unsigned long long)n);
  }
  {
    // fresh anonymous memory: what sizing gigabytes of std::vector costs
    const size_t bytes = size_t(4) << 30;
    double t0 = now();
    char* a = static_cast<char*>(mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    par(T, bytes, [&](unsigned, size_t b, size_t e) { for (size_t i = b; i < e; i += 4096) a[i] = 1; });
    double t1 = now();
    munmap(a, bytes);
    char* h = static_cast<char*>(mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    madvise(h, bytes, MADV_HUGEPAGE);
    double t2 = now();
    par(T, bytes, [&](unsigned, size_t b, size_t e) { for (size_t i = b; i < e; i += 4096) h[i] = 1; });
    double t3 = now();
    munmap(h, bytes);
    printf("E first touch of 4 GB anonymous: 4K pages %.3f s, THP %.3f s\n", t1 - t0, t3 - t2);
  }
  return 0;
}
