// What do unaligned 16-byte stores cost on gfx950?  k_bam_encode / k_bam_rows write the output stream with 16-byte
// stores at arbitrary byte addresses (a wave's lanes contiguous); this measures the same shape against the aligned one.
//   store: lane g writes 16 bytes at base + off + 16 g          (off = 0, 3, 8)
//   copy : the same with a 16-byte load at src + off2 + 16 g    (aligned / unaligned source)
// Build: hipcc --offload-arch=gfx950 -O3 profiles/calib_store.hip -o profiles/bin/calib_store ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct __attribute__((packed, aligned(1))) W4 { uint32_t a, b, c, d; };

__global__ void k_store(uint8_t *dst, uint64_t n16, uint32_t off) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    W4 v; v.a = (uint32_t)i; v.b = v.a + 1; v.c = v.a + 2; v.d = v.a + 3;
    *(W4 *)(dst + off + 16 * i) = v;
  }
}
__global__ void k_copy(uint8_t *dst, const uint8_t *src, uint64_t n16, uint32_t off_d, uint32_t off_s) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
    *(W4 *)(dst + off_d + 16 * i) = *(const W4 *)(src + off_s + 16 * i);
}
// the old encoder's shape: 8 lanes per 222-byte row, regions of 20 / 50 / 100 / 30 bytes copied one after the other
__global__ void k_rows8(uint8_t *dst, const uint8_t *src, uint64_t n_rows) {
  const int lane = threadIdx.x & 7;
  uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  if (r >= n_rows) return;
  uint8_t *d = dst + r * 222; const uint8_t *s = src + (r / 5) * 222;
  const uint32_t at[5] = {36, 56, 64, 114, 214}, len[4] = {20, 50, 100, 8};
  for (int k = 0; k < 4; k++) {
    const uint32_t a = at[k] + (k == 3 ? 0 : 0), n = len[k];
    for (uint32_t i = 16u * lane; i < n; i += 128u) { uint32_t o = i + 16u <= n ? i : n - 16u; if (n >= 16u) *(W4 *)(d + a + o) = *(const W4 *)(s + a - 4 + o); }
  }
}

int main() {
  const uint64_t BYTES = 8ull << 30;
  uint8_t *d, *s;
  CK(hipMalloc(&d, BYTES + 4096)); CK(hipMalloc(&s, BYTES + 4096));
  CK(hipMemset(d, 0, BYTES + 4096)); CK(hipMemset(s, 1, BYTES + 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint64_t n16 = BYTES / 16;
  const uint32_t offs[4] = {0, 3, 8, 16};
  for (int rep = 0; rep < 2; rep++) {
    for (int k = 0; k < 4; k++) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_store, dim3(256 * 32), dim3(256), 0, 0, d, n16, offs[k]);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("store  off %2u: %7.2f ms  %7.1f GB/s\n", offs[k], ms, BYTES / ms / 1e6);
    }
    for (int k = 0; k < 4; k++) for (int j = 0; j < 2; j++) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_copy, dim3(256 * 32), dim3(256), 0, 0, d, s, n16, offs[k], j ? 5u : 0u);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("copy   dst off %2u src off %u: %7.2f ms  %7.1f GB/s (read + write)\n", offs[k], j ? 5u : 0u, ms, 2.0 * BYTES / ms / 1e6);
    }
    {
      const uint64_t n_rows = BYTES / 222;
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_rows8, dim3((unsigned)((n_rows * 8 + 255) / 256)), dim3(256), 0, 0, d, s, n_rows);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("rows8 (178 of 222 bytes per row): %7.2f ms  %7.1f GB/s written\n", ms, n_rows * 178.0 / ms / 1e6);
    }
  }
  return 0;
}
