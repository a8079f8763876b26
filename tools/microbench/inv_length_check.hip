// Brute-force check on gfx950: the reciprocal of a correctly rounded square root obtained from the sqrt refinement's own
// half-reciprocal (one Newton step + the division's final correction) against IEEE 1.0 / sqrt(x), bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/inv_check tools/microbench/inv_length_check.hip && /tmp/inv_check [log2 samples]
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
__device__ inline void sqrt_and_inverse(double x, double &root, double &inv) {
	const double y = __builtin_amdgcn_rsq(x);
	double g = x * y, h = y * 0.5;
	const double r = __builtin_fma(-h, g, 0.5);
	g = __builtin_fma(g, r, g), h = __builtin_fma(h, r, h);
	g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
	g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
	root = g;
	const double r0 = h + h;
	const double r1 = __builtin_fma(__builtin_fma(-g, r0, 1.0), r0, r0);
	inv = __builtin_fma(__builtin_fma(-g, r1, 1.0), r1, r1);
	// A root whose significand is all ones is the one divisor Markstein's final correction cannot round: 1/root lies
	// 2^-106 (relative) above a midpoint.  There the quotient is the power of two r1 was rounded to, plus one ulp.
	const uint64_t gb = __builtin_bit_cast(uint64_t, g);
	if ((uint32_t)gb == 0xFFFFFFFFu && ((uint32_t)(gb >> 32) | 0xFFF00000u) == 0xFFFFFFFFu) inv = __builtin_bit_cast(double, __builtin_bit_cast(uint64_t, r1) | 1ull);
}
__global__ void check(uint64_t seed, int per_thread, int mode, unsigned long long *bad_root, unsigned long long *bad_inv, double *example) {
	uint64_t s = seed + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x632BE59BD9B4E019ull;
	unsigned long long br = 0, bi = 0, guarded = 0;
	for (int i = 0; i < per_thread; i++) {
		const uint64_t m = splitmix(s);
		uint64_t bits;
		if (mode == 0) { // random mantissa, exponent in [-700, 700]
			const int e = (int)(splitmix(s) % 1401) - 700;
			bits = ((uint64_t)(1023 + e) << 52) | (m & 0xFFFFFFFFFFFFFull);
		} else if (mode == 1) { // sums of three squares of unit-ish components: the arguments normalize() really sees
			const double a = (double)(m >> 11) * 0x1p-53, b = (double)(splitmix(s) >> 11) * 0x1p-53, c = (double)(splitmix(s) >> 11) * 0x1p-53;
			const double v = (a * a + b * b) + c * c;
			bits = __builtin_bit_cast(uint64_t, v > 0x1p-600 ? v : 1.0);
		} else if (mode == 3) { // squares of random doubles (exactly or nearly representable roots)
			const int e = (int)(splitmix(s) % 601) - 300;
			const double q = __builtin_bit_cast(double, ((uint64_t)(1023 + e) << 52) | (m & 0xFFFFFFFFFFFFFull) & ~((splitmix(s) & 1) ? 0x3FFFFFFull : 0ull));
			bits = __builtin_bit_cast(uint64_t, q * q);
		} else if (mode == 4) { // roots next to all-ones: squares of 2^e * (2 - j * 2^-52), small j, nudged by a few ulps
			const int e = (int)(splitmix(s) % 601) - 300;
			const double q = __builtin_bit_cast(double, ((uint64_t)(1023 + e) << 52) | (0xFFFFFFFFFFFFFull - (m & 7)));
			bits = __builtin_bit_cast(uint64_t, q * q) + (splitmix(s) % 9) - 4;
		} else { // mantissas near all-ones / all-zeros
			const uint64_t k = m & 0xFFFFF;
			const uint64_t mant = (splitmix(s) & 1) ? (0xFFFFFFFFFFFFFull - k) : k;
			const int e = (int)(splitmix(s) % 1401) - 700;
			bits = ((uint64_t)(1023 + e) << 52) | mant;
		}
		const double x = __builtin_bit_cast(double, bits);
		const double root_ref = __builtin_sqrt(x);
		const double inv_ref = 1.0 / root_ref;
		double root, inv;
		sqrt_and_inverse(x, root, inv);
		if ((uint32_t)__builtin_bit_cast(uint64_t, root) == 0xFFFFFFFFu) guarded++; // how many roots had an all-ones low word (the all-ones significands are among them)
		if (__builtin_bit_cast(uint64_t, root) != __builtin_bit_cast(uint64_t, root_ref)) br++;
		if (__builtin_bit_cast(uint64_t, inv) != __builtin_bit_cast(uint64_t, inv_ref)) { bi++; example[0] = x; }
	}
	if (br) atomicAdd(bad_root, br);
	if (bi) atomicAdd(bad_inv, bi);
	if (guarded) atomicAdd(bad_inv + 1, guarded);
}
int main(int argc, char **argv) {
	const int lg = argc > 1 ? atoi(argv[1]) : 30;
	unsigned long long *d, h[3];
	double *ex, hex = 0;
	hipMalloc(&d, 24), hipMalloc(&ex, 8);
	const int blocks = 256 * 16, threads = 256, per_thread = (int)((1ull << lg) / ((uint64_t)blocks * threads));
	for (int mode = 0; mode < 5; mode++) {
		hipMemset(d, 0, 24), hipMemset(ex, 0, 8);
		check<<<blocks, threads>>>(0x1234ull + mode, per_thread, mode, d, d + 1, ex);
		hipDeviceSynchronize();
		hipMemcpy(h, d, 24, hipMemcpyDeviceToHost), hipMemcpy(&hex, ex, 8, hipMemcpyDeviceToHost);
		printf("mode %d: %llu samples, sqrt mismatches %llu, reciprocal mismatches %llu (example x = %a), %llu roots with an all-ones low word\n", mode,
		       (unsigned long long)blocks * threads * per_thread, h[0], h[1], hex, h[2]);
		fflush(stdout);
	}
	return 0;
}
