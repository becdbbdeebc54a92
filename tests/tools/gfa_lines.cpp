// Test helper (not product code): prints, in file order, the S lines of a GFA whose segment name is in a given set and the
// L lines whose two ends both are.  tests/test_gpu_scale_parity.py uses it to cut the part of a whole-genome graph that a
// slice of reads touches out of (a) the input GFA -- the graph the Python oracle can hold -- and (b) the engine's
// augmented output, so that the two can be compared byte for byte.
//   gfa_lines <gfa> <names.txt> [threads]      names: one canonical decimal per line
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

static bool is_ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s gfa names.txt [threads]\n", argv[0]); return 2; }
  const unsigned T = argc > 3 ? unsigned(atoi(argv[3])) : std::max(1u, std::thread::hardware_concurrency());
  std::vector<uint64_t> ids;
  {
    FILE* f = fopen(argv[2], "r");
    if (!f) { perror(argv[2]); return 2; }
    char buf[64];
    while (fgets(buf, sizeof buf, f)) if (buf[0] >= '0' && buf[0] <= '9') ids.push_back(strtoull(buf, nullptr, 10));
    fclose(f);
  }
  if (ids.empty()) return 0;
  const uint64_t lo = *std::min_element(ids.begin(), ids.end()), hi = *std::max_element(ids.begin(), ids.end());
  std::vector<uint8_t> in(size_t(hi - lo + 1), 0);
  for (uint64_t x : ids) in[size_t(x - lo)] = 1;
  int fd = open(argv[1], O_RDONLY);
  if (fd < 0) { perror(argv[1]); return 2; }
  struct stat st;
  fstat(fd, &st);
  const size_t size = size_t(st.st_size);
  const char* base = static_cast<const char*>(mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0));
  if (base == MAP_FAILED) { perror("mmap"); return 2; }
  auto member = [&](const char* b, const char* e) {
    if (b >= e || e - b > 18) return false;
    uint64_t v = 0;
    for (const char* p = b; p < e; ++p) { if (*p < '0' || *p > '9') return false; v = v * 10 + uint64_t(*p - '0'); }
    return v >= lo && v <= hi && in[size_t(v - lo)] != 0;
  };
  std::vector<size_t> cut(T + 1, size);
  cut[0] = 0;
  for (unsigned t = 1; t < T; ++t) {
    size_t p = size / T * t;
    const void* nl = p < size ? memchr(base + p, '\n', size - p) : nullptr;
    cut[t] = nl ? size_t(static_cast<const char*>(nl) - base) + 1 : size;
  }
  std::vector<std::string> out(T);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t] {
    const char* p = base + cut[t];
    const char* end = base + cut[t + 1];
    std::string& o = out[t];
    while (p < end) {
      const char* nl = static_cast<const char*>(memchr(p, '\n', size_t(end - p)));
      const char* le = nl ? nl : end;
      if (le > p && (*p == 'S' || *p == 'L')) {
        const char* tok[5][2];
        int n = 0;
        const char* q = p;
        const int want = *p == 'S' ? 2 : 4;
        while (n < want && q < le) {
          while (q < le && is_ws((unsigned char)*q)) ++q;
          if (q >= le) break;
          tok[n][0] = q;
          while (q < le && !is_ws((unsigned char)*q)) ++q;
          tok[n][1] = q;
          ++n;
        }
        bool keep = false;
        if (*p == 'S' && n == 2) keep = member(tok[1][0], tok[1][1]);
        if (*p == 'L' && n == 4) keep = member(tok[1][0], tok[1][1]) && member(tok[3][0], tok[3][1]);
        if (keep) { o.append(p, size_t(le - p)); o.push_back('\n'); }
      }
      p = nl ? nl + 1 : end;
    }
  });
  for (auto& x : th) x.join();
  for (auto& o : out) fwrite(o.data(), 1, o.size(), stdout);
  return 0;
}
