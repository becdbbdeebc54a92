// What the first HIP calls of a process cost on the GPU box (the start-up second of both command lines):
//   hipcc --offload-arch=gfx950 -O2 tools/hip_startup_probe.hip -o build/hip_startup_probe && build/hip_startup_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k() {}
int main() {
  double t = now();
  auto lap = [&](const char* w) { const double n = now(); printf("%-40s %.3f s\n", w, n - t); t = n; };
  (void)hipSetDevice(0); (void)hipFree(nullptr); lap("hipSetDevice + hipFree(nullptr)");
  hipStream_t s[4];
  for (int i = 0; i < 4; ++i) { (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); char b[64]; snprintf(b, sizeof b, "hipStreamCreateWithFlags #%d", i + 1); lap(b); }
  hipEvent_t e[6];
  for (auto& x : e) (void)hipEventCreate(&x);
  lap("6 x hipEventCreate");
  void* p[5];
  for (int i = 0; i < 3; ++i) (void)hipMalloc(&p[i], size_t(1) << 30);
  lap("3 x hipMalloc 1 GiB");
  (void)hipMalloc(&p[3], size_t(5) << 30); lap("hipMalloc 5 GiB");
  (void)hipMemsetAsync(p[3], 0, size_t(5) << 30, s[0]); (void)hipStreamSynchronize(s[0]); lap("memset 5 GiB + sync");
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[0]); (void)hipStreamSynchronize(s[0]); lap("first kernel launch + sync");
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[1]); (void)hipStreamSynchronize(s[1]); lap("first launch on stream 2 + sync");
  void* h = nullptr;
  (void)hipHostMalloc(&h, size_t(64) << 20, hipHostMallocDefault); lap("hipHostMalloc 64 MiB");
  (void)hipMemcpyAsync(p[0], h, size_t(64) << 20, hipMemcpyHostToDevice, s[2]); (void)hipStreamSynchronize(s[2]); lap("first H2D copy 64 MiB (stream 3) + sync");
  return 0;
}
