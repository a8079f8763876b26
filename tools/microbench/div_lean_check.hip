// Brute-force check on gfx950: div_lean(a, b) (device_core.hpp: the IEEE division's own arithmetic without its scaling and special-case instructions)
// against `a / b`, bit for bit, over the range its callers guarantee.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I raymond_amd/csrc -o /tmp/div_lean_check tools/microbench/div_lean_check.hip && /tmp/div_lean_check [log2 samples per mode]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "device_core.hpp"
__device__ inline uint64_t splitmix(uint64_t &s) {
	uint64_t z = (s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
__device__ inline double mkd(uint64_t mant, int e, bool neg) { return __builtin_bit_cast(double, ((uint64_t)neg << 63) | ((uint64_t)(1023 + e) << 52) | (mant & 0xFFFFFFFFFFFFFull)); }
__global__ void check(uint64_t seed, int per_thread, int mode, unsigned long long *bad, double *example) {
	uint64_t s = seed + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x632BE59BD9B4E019ull;
	unsigned long long nb = 0;
	for (int i = 0; i < per_thread; i++) {
		uint64_t ma = splitmix(s), mb = splitmix(s);
		const uint64_t x = splitmix(s);
		int ea = (int)(x % 601) - 300, eb = (int)((x >> 20) % 601) - 300; // |exponent difference| <= 600
		if (mode == 1) mb = 0xFFFFFFFFFFFFFull - (mb & 0xFF);                 // divisors next to all-ones
		if (mode == 2) mb = mb & 0xFF;                                         // divisors next to a power of two
		if (mode == 3) ma = 0xFFFFFFFFFFFFFull - (ma & 0xFF);                 // numerators next to all-ones
		if (mode == 4) { ea = (int)(x % 21) - 10, eb = (int)((x >> 20) % 21) - 20; } // the axis rule's shape: numerators of order 1, divisors in (1e-6, 1]
		if (mode == 5) { ea = (int)(x % 54) - 53, eb = (int)((x >> 20) % 54) - 53; ma &= ~0ull << (ea + 53 < 52 ? 52 - (ea + 53) : 0), mb &= ~0ull << (eb + 53 < 52 ? 52 - (eb + 53) : 0); } // r2 / (1 - r2): multiples of 2^-53 below 1
		if (mode == 6) { ea = 0, ma = 0; eb = (int)((x >> 20) % 2); }      // -1 / (sign + n.z): a = -1, b in [1, 2] (and the power of two itself)
		double a = mkd(ma, ea, x >> 63), b = mkd(mb, eb, (x >> 62) & 1);
		if (mode == 7) a = 0.0, b = __builtin_fabs(b);                         // +0 over a positive divisor (r2 = 0 in r2 / (1 - r2)): +0 either way
		if (mode == 8) a = (x & 1) ? 0.0 : -0.0;                               // zeros of either sign over divisors of either sign: NOT admitted (the quotient's sign of zero is lost) — reported, not counted
		const double fast = rmd::div_lean(a, b), ref = a / b;
		if (__builtin_bit_cast(uint64_t, fast) != __builtin_bit_cast(uint64_t, ref)) { nb++; example[0] = a, example[1] = b; }
	}
	if (nb) atomicAdd(bad, nb);
}
int main(int argc, char **argv) {
	const int lg = argc > 1 ? atoi(argv[1]) : 34;
	unsigned long long *d, h;
	double *ex, hex[2];
	hipMalloc(&d, 16), hipMalloc(&ex, 16);
	const int blocks = 256 * 16, threads = 256, per_thread = (int)((1ull << lg) / ((uint64_t)blocks * threads));
	unsigned long long total = 0, total_bad = 0;
	for (int mode = 0; mode < 9; mode++) {
		hipMemset(d, 0, 16), hipMemset(ex, 0, 16);
		check<<<blocks, threads>>>(0x5151ull + mode, per_thread, mode, d, ex);
		hipDeviceSynchronize();
		hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost), hipMemcpy(hex, ex, 16, hipMemcpyDeviceToHost);
		printf("mode %d: %llu quotients, mismatches %llu (example %a / %a)\n", mode, (unsigned long long)blocks * threads * per_thread, h, hex[0], hex[1]);
		fflush(stdout);
		if (mode != 8) total += (unsigned long long)blocks * threads * per_thread, total_bad += h;
	}
	printf("total over the admitted range (modes 0 - 7): %llu quotients, %llu mismatches\n", total, total_bad);
	return total_bad != 0;
}
