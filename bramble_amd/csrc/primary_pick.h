// Tie-break of the primary alignment (src/core.cpp:214-218,298-299):
//   seed = std::hash<std::string>{}(read_name);  mt19937_64 gen(seed);
//   uniform_int_distribution<uint32_t>(0, n_tied - 1)(gen)
// restated for host and device from libstdc++ (GCC 11; the result depends on the
// standard library the reference is built with -- this is the x86-64 Linux one):
//   * _Hash_bytes (libstdc++ hash_bytes.cc, 64-bit Murmur-style, seed 0xc70f6907)
//   * mersenne_twister_engine<uint64_t, 64, 312, 156, 31, ...>::seed / operator()
//   * uniform_int_distribution::_S_nd (Lemire's nearly divisionless, 128-bit product)
// Only the first few generator outputs are ever needed, so the 312-word state is
// never materialised: output k depends on x[k], x[k+1], x[k+156] of the seeded
// sequence (valid for k < 156).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BR_HD __host__ __device__ __forceinline__
#else
#define BR_HD inline
#endif

namespace br {

BR_HD uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }

BR_HD uint64_t hash_bytes(const uint8_t *buf, uint64_t len) {
  const uint64_t mul = (0xc6a4a793ull << 32) + 0x5bd1e995ull;
  const uint64_t seed = 0xc70f6907ull;
  const uint64_t len_aligned = len & ~(uint64_t)0x7;
  uint64_t hash = seed ^ (len * mul);
  for (uint64_t p = 0; p < len_aligned; p += 8) {
    uint64_t w = 0;
    for (int k = 7; k >= 0; k--) w = (w << 8) | buf[p + k];  // unaligned little-endian load
    uint64_t data = shift_mix(w * mul) * mul;
    hash ^= data;
    hash *= mul;
  }
  if ((len & 0x7) != 0) {
    uint64_t data = 0;
    for (int k = (int)(len & 0x7) - 1; k >= 0; k--) data = (data << 8) + buf[len_aligned + k];  // load_bytes
    hash ^= data;
    hash *= mul;
  }
  hash = shift_mix(hash) * mul;
  hash = shift_mix(hash);
  return hash;
}

struct Mt64Lazy {
  uint64_t xk, xk1, xhi;  // x[k], x[k+1], x[k+156] of the seeded sequence
  uint32_t k;
  BR_HD static uint64_t next_seed(uint64_t prev, uint32_t i) { return 6364136223846793005ull * (prev ^ (prev >> 62)) + i; }
  BR_HD void seed(uint64_t s) {
    k = 0; xk = s; xk1 = next_seed(s, 1);
    uint64_t x = xk1;
    for (uint32_t i = 2; i <= 156; i++) x = next_seed(x, i);
    xhi = x;
  }
  BR_HD uint64_t operator()() {
    uint64_t y = (xk & 0xFFFFFFFF80000000ull) | (xk1 & 0x7FFFFFFFull);
    uint64_t z = xhi ^ (y >> 1) ^ ((y & 1) ? 0xB5026F5AA96619E9ull : 0ull);
    z ^= (z >> 29) & 0x5555555555555555ull;
    z ^= (z << 17) & 0x71D67FFFEDA60000ull;
    z ^= (z << 37) & 0xFFF7EEE000000000ull;
    z ^= (z >> 43);
    // advance to output k + 1
    xk = xk1; xk1 = next_seed(xk1, k + 2); xhi = next_seed(xhi, k + 157); k++;
    return z;
  }
};

BR_HD void mul64wide(uint64_t a, uint64_t b, uint64_t &hi, uint64_t &lo) {
  uint64_t a0 = a & 0xffffffffull, a1 = a >> 32, b0 = b & 0xffffffffull, b1 = b >> 32;
  uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
  uint64_t mid = (p00 >> 32) + (p01 & 0xffffffffull) + (p10 & 0xffffffffull);
  lo = (p00 & 0xffffffffull) | (mid << 32);
  hi = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
}

// get_rand(n_tied, std::hash<std::string>(name))
BR_HD uint32_t primary_pick(const uint8_t *name, uint64_t len, uint32_t n_tied) {
  Mt64Lazy g; g.seed(hash_bytes(name, len));
  uint64_t range = n_tied;  // __uerange
  uint64_t hi, lo;
  mul64wide(g(), range, hi, lo);
  if (lo < range) {
    uint64_t threshold = (0 - range) % range;
    while (lo < threshold) mul64wide(g(), range, hi, lo);
  }
  return (uint32_t)hi;
}

}  // namespace br
