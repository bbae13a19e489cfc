// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of the projection kernels:
// the guide's "double FETCH_SIZE" rule is measured on coalesced 16 B/lane streams; the count and emit passes issue
// random 32-byte row gathers, 16-byte record gathers and 4-byte key gathers.  Every kernel below reads a known number
// of useful bytes from a 1 GiB table (4x the Infinity Cache) at random, line-aligned offsets, so that
//   factor = useful bytes / (FETCH_SIZE * 1024)
// can be read off per shape.  Build: hipcc --offload-arch=gfx950 -O3 profiles/calib_fetch.hip -o profiles/bin/calib_fetch
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/calib -o run -- profiles/bin/calib_fetch
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ void k_cal_stream16(const uint4 *t, uint64_t n16, uint32_t *sink) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { uint4 v = t[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}
// W bytes per lane at a random W-aligned offset (W = 4, 16, 32, 64)
template <int W>
__global__ void k_cal_gather(const uint8_t *t, uint64_t n_slots, uint64_t n_lanes, uint32_t *sink) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_lanes) return;
  const uint8_t *p = t + (mix(i * 0x9E3779B97F4A7C15ull + W) % n_slots) * W;
  uint32_t acc = 0;
  if (W == 4) acc = *(const uint32_t *)p;
  else for (int k = 0; k < W / 16; k++) { uint4 v = ((const uint4 *)p)[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}
// the emit pass's shape: eight 4-byte keys per lane, each from its own random 64-byte line
__global__ void k_cal_gather4x8(const uint8_t *t, uint64_t n_slots, uint64_t n_lanes, uint32_t *sink) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_lanes) return;
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) acc ^= *(const uint32_t *)(t + (mix(i * 8 + k) % n_slots) * 4);
  if (acc == 0x12345678u) *sink = acc;
}

int main() {
  const uint64_t BYTES = 1ull << 30;
  uint8_t *t; uint32_t *sink;
  CK(hipMalloc(&t, BYTES)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(t, 1, BYTES));
  const uint64_t lanes = 1ull << 24;   // 16.8 M gathers per kernel
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_cal_stream16, dim3(4096), dim3(256), 0, 0, (const uint4 *)t, BYTES / 16, sink);
    hipLaunchKernelGGL((k_cal_gather<4>), dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, t, BYTES / 4, lanes, sink);
    hipLaunchKernelGGL((k_cal_gather<16>), dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, t, BYTES / 16, lanes, sink);
    hipLaunchKernelGGL((k_cal_gather<32>), dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, t, BYTES / 32, lanes, sink);
    hipLaunchKernelGGL((k_cal_gather<64>), dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, t, BYTES / 64, lanes, sink);
    hipLaunchKernelGGL(k_cal_gather4x8, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, t, BYTES / 4, lanes, sink);
  }
  CK(hipDeviceSynchronize());
  printf("useful bytes per launch: stream16 %llu, gather4 %llu, gather16 %llu, gather32 %llu, gather64 %llu, gather4x8 %llu\n",
         (unsigned long long)BYTES, (unsigned long long)(lanes * 4), (unsigned long long)(lanes * 16), (unsigned long long)(lanes * 32),
         (unsigned long long)(lanes * 64), (unsigned long long)(lanes * 32));
  return 0;
}
