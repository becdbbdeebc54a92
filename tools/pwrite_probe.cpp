// How fast N threads pwrite 4 MB pieces into ONE file (what both output writers do), under the box's CPU quota:
//   g++ -O2 -pthread tools/pwrite_probe.cpp -o /tmp/pwrite_probe && /tmp/pwrite_probe /dev/shm/pwrite_probe.bin 16
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "/dev/shm/pwrite_probe.bin";
  const size_t gb = argc > 2 ? size_t(atoi(argv[2])) : 16;
  const size_t piece = size_t(4) << 20, n_pieces = (gb << 30) / piece;
  for (int rep = 0; rep < 3; ++rep)
    for (unsigned T : {1u, 2u, 4u, 8u, 12u, 16u, 32u, 64u}) {
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
          size_t off = 0;
          while (off < piece) {
            const ssize_t k = pwrite(fd, buf.data() + off, piece - off, off_t(i * piece + off));
            if (k <= 0) { perror("pwrite"); exit(2); }
            off += size_t(k);
          }
        }
      });
      for (auto& x : th) x.join();
      const double dt = now() - t0;
      close(fd);
      const double t1 = now();
      unlink(path);
      printf("rep %d threads %2u: %6.2f s  %5.2f GB/s   (unlink %.2f s)\n", rep, T, dt, double(gb) / dt * 1.0737, now() - t1);
      fflush(stdout);
    }
  return 0;
}
