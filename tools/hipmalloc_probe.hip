// What hipMalloc / hipFree cost on this box by size and count (the GFA loader and the first tokenisation allocate tens of GB in
// a few dozen calls; their host-side phase times vary 10x between leases).  hipcc -O2 -o /tmp/hipmalloc_probe tools/hipmalloc_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipSetDevice(0);
  void* w = nullptr; hipMalloc(&w, 1 << 20); hipFree(w);
  const size_t GB = size_t(1) << 30;
  struct Case { int n; size_t bytes; } cases[] = {{1, 64 * GB / 64}, {32, GB / 2}, {4, 4 * GB}, {1, 16 * GB}, {1, 16 * GB}, {64, GB / 4}, {1, 32 * GB}};
  for (const Case& c : cases) {
    std::vector<void*> p(c.n);
    double t0 = now();
    for (int i = 0; i < c.n; ++i) if (hipMalloc(&p[i], c.bytes) != hipSuccess) { printf("malloc failed\n"); return 1; }
    double t1 = now();
    for (int i = 0; i < c.n; ++i) hipMemsetAsync(p[i], 0, 4096, nullptr);
    hipDeviceSynchronize();
    double t2 = now();
    for (int i = 0; i < c.n; ++i) hipFree(p[i]);
    double t3 = now();
    printf("%3d x %6.2f GB: hipMalloc %8.3f ms (%.3f ms/GB), first touch %7.3f ms, hipFree %8.3f ms\n", c.n, c.bytes / double(GB), (t1 - t0) * 1e3,
           (t1 - t0) * 1e3 / (c.n * c.bytes / double(GB)), (t2 - t1) * 1e3, (t3 - t2) * 1e3);
  }
  // the pattern of the GFA loader: tens of GB allocated, WRITTEN, freed -- and then the next allocations
  for (int rep = 0; rep < 3; ++rep) {
    std::vector<void*> p(6);
    const size_t sz[6] = {23 * GB, 4 * GB, 8 * GB, 2 * GB, 2 * GB, 6 * GB};
    double t0 = now();
    for (int i = 0; i < 6; ++i) hipMalloc(&p[i], sz[i]);
    double t1 = now();
    for (int i = 0; i < 6; ++i) hipMemsetAsync(p[i], 1, sz[i], nullptr);
    hipDeviceSynchronize();
    double t2 = now();
    for (int i = 0; i < 6; ++i) hipFree(p[i]);
    double t3 = now();
    void* q[8];
    for (int i = 0; i < 8; ++i) hipMalloc(&q[i], (i + 1) * GB);
    double t4 = now();
    for (int i = 0; i < 8; ++i) hipMemsetAsync(q[i], 2, (i + 1) * GB, nullptr);
    hipDeviceSynchronize();
    double t5 = now();
    for (int i = 0; i < 8; ++i) hipFree(q[i]);
    double t6 = now();
    printf("45 GB: hipMalloc %.1f ms, written in %.1f ms, hipFree %.1f ms; then 36 GB in 8 blocks: hipMalloc %.1f ms, written in %.1f ms, hipFree %.1f ms\n",
           (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3);
  }
  return 0;
}
