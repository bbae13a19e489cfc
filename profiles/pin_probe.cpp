// What does it cost to get a 250 MB pinned download buffer?  hipHostMalloc against an anonymous mapping on transparent
// huge pages registered with hipHostRegister, one thread and two threads at once (two workers of the command line
// start at the same time).  hipcc -O2 profiles/pin_probe.cpp -o /tmp/pin_probe -lpthread && /tmp/pin_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const size_t N = 250u << 20;
static double t_malloc() { double t0 = now(); void *p = nullptr; if (hipHostMalloc(&p, N, hipHostMallocDefault) != hipSuccess) return -1; double t = now() - t0; memset(p, 1, 4096); hipHostFree(p); return t; }
static double t_register(bool huge, bool touch) {
  double t0 = now();
  void *p = mmap(nullptr, N + (2u << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) return -1;
  uint8_t *q = (uint8_t *)(((uintptr_t)p + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1));
  if (huge) madvise(q, N, MADV_HUGEPAGE);
  if (touch) for (size_t i = 0; i < N; i += 4096) q[i] = 0;
  double t1 = now();
  if (hipHostRegister(q, N, hipHostRegisterDefault) != hipSuccess) return -2;
  double t = now() - t0;
  fprintf(stderr, "    (map%s%s %.1f ms, register %.1f ms)\n", huge ? " + huge" : "", touch ? " + touch" : "", 1e3 * (t1 - t0), 1e3 * (now() - t1));
  hipHostUnregister(q); munmap(p, N + (2u << 20));
  return t;
}
int main() {
  hipSetDevice(0); hipFree(0);
  for (int rep = 0; rep < 2; rep++) {
    printf("hipHostMalloc(250 MB): %.1f ms\n", 1e3 * t_malloc());
    printf("mmap + hipHostRegister: %.1f ms\n", 1e3 * t_register(false, false));
    printf("mmap + touch + hipHostRegister: %.1f ms\n", 1e3 * t_register(false, true));
    printf("mmap + MADV_HUGEPAGE + touch + hipHostRegister: %.1f ms\n", 1e3 * t_register(true, true));
    double a[2]; std::thread t1([&] { a[0] = t_malloc(); }), t2([&] { a[1] = t_malloc(); }); t1.join(); t2.join();
    printf("two hipHostMalloc at once: %.1f / %.1f ms\n", 1e3 * a[0], 1e3 * a[1]);
    std::thread t3([&] { a[0] = t_register(true, true); }), t4([&] { a[1] = t_register(true, true); }); t3.join(); t4.join();
    printf("two huge + register at once: %.1f / %.1f ms\n", 1e3 * a[0], 1e3 * a[1]);
  }
  return 0;
}
