// How fast pieces of 4 MB reach ONE file (what both output writers do), under the box's CPU quota:
//   g++ -O2 -pthread tools/pwrite_probe.cpp -o /tmp/pwrite_probe && /tmp/pwrite_probe /dev/shm/pwrite_probe.bin 16
// Part 1: N threads fill a buffer and pwrite it at its offset.  Part 2: N threads fill buffers, ONE other thread writes them in order
// (the buffers come from other cores: what OrderedOutput's writer thread sees).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void write_all(int fd, const char* p, size_t n, off_t at) {
  size_t off = 0;
  while (off < n) {
    const ssize_t k = at < 0 ? write(fd, p + off, n - off) : pwrite(fd, p + off, n - off, at + off_t(off));
    if (k <= 0) { perror("write"); exit(2); }
    off += size_t(k);
  }
}
int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "/dev/shm/pwrite_probe.bin";
  const size_t gb = argc > 2 ? size_t(atoi(argv[2])) : 16;
  const size_t piece = size_t(4) << 20, n_pieces = (gb << 30) / piece;
  for (int rep = 0; rep < 2; ++rep)
    for (unsigned T : {1u, 2u, 8u, 16u, 64u}) {
      const int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
      if (fd < 0) { perror(path); return 2; }
      std::atomic<size_t> next{0};
      const double t0 = now();
      std::vector<std::thread> th;
      for (unsigned t = 0; t < T; ++t) th.emplace_back([&] {
        std::vector<char> buf(piece, 'x');
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= n_pieces) break;
          memset(buf.data(), int('a' + i % 26), piece);          // stands for the formatting
          write_all(fd, buf.data(), piece, off_t(i * piece));
        }
      });
      for (auto& x : th) x.join();
      const double dt = now() - t0;
      close(fd);
      unlink(path);
      printf("pwrite by its maker   rep %d threads %2u: %6.2f s  %5.2f GB/s\n", rep, T, dt, double(gb) / dt * 1.0737);
      fflush(stdout);
    }
  for (int rep = 0; rep < 2; ++rep)
    for (unsigned T : {1u, 4u, 12u, 64u}) {
      const int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
      if (fd < 0) { perror(path); return 2; }
      std::mutex mu;
      std::condition_variable cv_ready, cv_done;
      std::vector<const char*> ready(n_pieces, nullptr);
      std::vector<char> done(n_pieces, 0);
      std::atomic<size_t> next{0};
      const double t0 = now();
      std::thread writer([&] {
        for (size_t k = 0; k < n_pieces; ++k) {
          const char* p;
          { std::unique_lock<std::mutex> lk(mu); cv_ready.wait(lk, [&] { return ready[k] != nullptr; }); p = ready[k]; }
          write_all(fd, p, piece, -1);
          { std::lock_guard<std::mutex> lk(mu); done[k] = 1; }
          cv_done.notify_all();
        }
      });
      std::vector<std::thread> th;
      for (unsigned t = 0; t < T; ++t) th.emplace_back([&] {
        std::vector<char> buf(piece, 'x');
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= n_pieces) break;
          memset(buf.data(), int('a' + i % 26), piece);
          std::unique_lock<std::mutex> lk(mu);
          ready[i] = buf.data();
          cv_ready.notify_one();
          cv_done.wait(lk, [&] { return done[i] != 0; });
        }
      });
      for (auto& x : th) x.join();
      writer.join();
      const double dt = now() - t0;
      close(fd);
      unlink(path);
      printf("one writer, %2u makers rep %d:             %6.2f s  %5.2f GB/s\n", T, rep, dt, double(gb) / dt * 1.0737);
      fflush(stdout);
    }
  return 0;
}
