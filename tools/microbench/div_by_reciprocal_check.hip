// Brute-force check on gfx950: a / b through the correctly rounded reciprocal r = 1.0 / b (Markstein's correction:
// q = a*r; rem = fma(-b, q, a); q' = fma(rem, r, q)) against the IEEE quotient, bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/div_check tools/microbench/div_by_reciprocal_check.hip && /tmp/div_check [log2 samples]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__device__ inline uint64_t splitmix(uint64_t &s) {
	uint64_t z = (s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
__device__ inline double mk(uint64_t mant, int e, bool neg) { return __builtin_bit_cast(double, ((uint64_t)neg << 63) | ((uint64_t)(1023 + e) << 52) | (mant & 0xFFFFFFFFFFFFFull)); }
__global__ void check(uint64_t seed, int per_thread, int mode, unsigned long long *bad, double *example) {
	uint64_t s = seed + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x632BE59BD9B4E019ull;
	unsigned long long nb = 0, nones = 0;
	for (int i = 0; i < per_thread; i++) {
		uint64_t ma = splitmix(s), mb = splitmix(s);
		const uint64_t x = splitmix(s);
		int ea = (int)(x % 401) - 200, eb = (int)((x >> 20) % 401) - 200;
		if (mode == 1) mb = 0xFFFFFFFFFFFFFull - (mb & 0xFF);           // divisors next to all-ones
		if (mode == 2) mb = mb & 0xFF;                                   // divisors next to a power of two
		if (mode == 3) { mb = (uint64_t)(mb % 65536) << 36; eb = 15; }   // small integers (image widths, grid resolutions)
		if (mode == 4) { ma = 0xFFFFFFFFFFFFFull - (ma & 0xFF); }        // numerators next to all-ones
		const double a = mk(ma, ea, x >> 63), b = mk(mb, eb, (x >> 62) & 1);
		const double r = 1.0 / b;
		const double q = a * r;
		const double fast = __builtin_fma(__builtin_fma(-b, q, a), r, q);
		const double ref = a / b;
		const bool ones = (__builtin_bit_cast(uint64_t, b) & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull;
		if (ones) nones++;
		if (__builtin_bit_cast(uint64_t, fast) != __builtin_bit_cast(uint64_t, ref) && !ones) { nb++; example[0] = a, example[1] = b; }
	}
	if (nb) atomicAdd(bad, nb);
	if (nones) atomicAdd(bad + 1, nones);
}
int main(int argc, char **argv) {
	const int lg = argc > 1 ? atoi(argv[1]) : 32;
	unsigned long long *d, h[2];
	double *ex, hex[2];
	hipMalloc(&d, 16), hipMalloc(&ex, 16);
	const int blocks = 256 * 16, threads = 256, per_thread = (int)((1ull << lg) / ((uint64_t)blocks * threads));
	for (int mode = 0; mode < 5; mode++) {
		hipMemset(d, 0, 16), hipMemset(ex, 0, 16);
		check<<<blocks, threads>>>(0x9876ull + mode, per_thread, mode, d, ex);
		hipDeviceSynchronize();
		hipMemcpy(h, d, 16, hipMemcpyDeviceToHost), hipMemcpy(hex, ex, 16, hipMemcpyDeviceToHost);
		printf("mode %d: %llu quotients, mismatches with a divisor that is not all-ones: %llu (example %a / %a); all-ones divisors seen: %llu\n", mode,
		       (unsigned long long)blocks * threads * per_thread, h[0], hex[0], hex[1], h[1]);
		fflush(stdout);
	}
	return 0;
}
