// Check of k_ppm_gather's quotient (rt_ppm.inc: the division's instruction sequence with the denominator's part hoisted) against the plain
// division, bit for bit, over random operands of the ranges the gather sees and far beyond:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// -o check_quot tools/exp/check_quot.hip && ./check_quot
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ void k(unsigned long long *bad, unsigned long long *n, int mode)
{
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long b = 0, c = 0;
  for (int it = 0; it < 256; it++) {
    const uint64_t a = mix(id * 977 + it * 2 + 1 + (uint64_t)mode * 0x1234567ull), e = mix(a);
    double d, x;
    if (mode == 0) {            // the gather's ranges: denom = pi R^2 n / 3000 with R in [1e-3, 10], n in [1e3, 1e9]; colours in [0, 1e6]
      const double R = 1e-3 + (double)(a >> 11) * (1.0 / 9007199254740992.0) * 10.0, nn = 1e3 + (double)(e >> 11) * (1.0 / 9007199254740992.0) * 1e9;
      d = (3.1415926535 * R * R * nn) / 3000; x = (double)(mix(e) >> 11) * (1.0 / 9007199254740992.0) * 1e6;
    } else {                    // any mantissa, exponents within +-300
      d = __longlong_as_double((long long)((a & 0x000FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 300 + (a >> 52) % 600) << 52)));
      x = __longlong_as_double((long long)((e & 0x000FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 300 + (e >> 52) % 600) << 52)));
    }
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const bool fast = d > 1e-100 && d < 1e100 && (x == 0.0 || (x > 1e-100 && x < 1e100));
    double q;
    if (fast) { const double q0 = x * r; q = fma(fma(-d, q0, x), r, q0); } else q = x / d;
    const double ref = x / d;
    if (__double_as_longlong(q) != __double_as_longlong(ref)) b++;
    c++;
  }
  atomicAdd(bad, b); atomicAdd(n, c);
}
int main()
{
  unsigned long long *d; hipMalloc(&d, 16);
  for (int mode = 0; mode < 2; mode++) {
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, d, d + 1, mode);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("mode %d: %llu of %llu quotients differ from x / d\n", mode, h[0], h[1]);
    if (h[0]) return 1;
  }
  return 0;
}
