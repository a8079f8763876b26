// Issue-rate probe for a few VALU opcodes on gfx950: each wave runs N iterations of 8 independent chains of one opcode.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/microbench/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHAINS 8
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int n, uint32_t seed) {
	uint32_t a[CHAINS];
	double d[CHAINS];
	for (int i = 0; i < CHAINS; i++) a[i] = seed + threadIdx.x * 977u + i, d[i] = 1.0 + 1e-9 * a[i];
	for (int it = 0; it < n; it++) {
#pragma unroll
		for (int i = 0; i < CHAINS; i++) {
			if (OP == 0) a[i] = a[i] + 0x9E3779B9u ^ (a[i] >> 3);           // add+xor+shift baseline (3 ops)
			if (OP == 1) a[i] = a[i] * 0xD2511F53u;                           // v_mul_lo_u32
			if (OP == 2) a[i] = __umulhi(a[i], 0xD2511F53u) + 1u;             // v_mul_hi_u32 (+add)
			if (OP == 3) { uint64_t p = (uint64_t)a[i] * 0xD2511F53u; a[i] = (uint32_t)p ^ (uint32_t)(p >> 32); } // mad_u64_u32 (+xor)
			if (OP == 4) d[i] = d[i] * 1.0000001;                             // v_mul_f64
			if (OP == 5) d[i] = __builtin_fma(d[i], 1.0000001, 1e-12);        // v_fma_f64
			if (OP == 6) d[i] = d[i] + 1e-7;                                  // v_add_f64
			if (OP == 7) d[i] = __builtin_amdgcn_rcp(d[i]) + 1.0;             // v_rcp_f64 (+add)
			if (OP == 8) d[i] = __builtin_amdgcn_rsq(d[i]) + 1.0;             // v_rsq_f64 (+add)
			if (OP == 9) d[i] = __builtin_amdgcn_div_fixup(d[i], 1.5, 2.5);   // v_div_fixup_f64
			if (OP == 10) d[i] = __builtin_amdgcn_div_fmas(d[i], 1.0000001, 1e-9, (a[i] & 1u) != 0u); // v_div_fmas_f64 (+ and/cmp)
			if (OP == 11) { bool f; d[i] = __builtin_amdgcn_div_scale(d[i], 3.0, true, &f); } // v_div_scale_f64
			if (OP == 12) d[i] = __builtin_ldexp(d[i], 1) * 0.5;             // v_ldexp_f64 + mul
			if (OP == 13) d[i] = __builtin_rint(d[i] * 1.7);                  // v_rndne_f64 + mul
			if (OP == 14) d[i] = (double)(int)d[i] + 1.5;                     // v_cvt_i32_f64 + v_cvt_f64_i32 + add
			if (OP == 15) d[i] = (d[i] < 2.0) ? d[i] + 1.0 : 1.0;             // v_cmp_lt_f64 + add + 2 cndmask
			if (OP == 16) d[i] = d[i] / 1.0000001;                            // full IEEE division
			if (OP == 17) d[i] = __builtin_sqrt(d[i]) + 1.0;                  // full IEEE sqrt + add
			if (OP == 18) a[i] = __umul24(a[i], 0x51F53u) + 1u; // v_mul_u32_u24 (+add; or one v_mad_u32_u24)
			if (OP == 19) { uint32_t h; asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(h) : "v"(a[i]), "v"(0x51F53u)); a[i] = h + a[i]; } // v_mul_hi_u32_u24 + add
			if (OP == 20) d[i] = (double)(uint32_t)a[i] * 1.0000001, a[i] = (uint32_t)d[i]; // v_cvt_f64_u32 + mul + v_cvt_u32_f64
			if (OP == 21) a[i] = (a[i] ^ 0x9E3779B9u ^ (uint32_t)it) + 1u;     // v_xor3 + add
			if (OP == 22) d[i] = __builtin_fabs(d[i] - 1.5) < 0.25 ? 1.0 : d[i] * 1.0000001; // sub-free: cmp with modifiers + mul + 2 cndmask
			// compares in isolation (round 5): one compare + one v_cndmask_b32 on an integer chain; the f64 operand does not change
			if (OP == 23) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(d[i]), "v"(d[(i + 1) % CHAINS]), "v"(a[(i + 1) % CHAINS]) : "vcc");
			if (OP == 24) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(a[(i + 2) % CHAINS]), "v"(a[(i + 3) % CHAINS]), "v"(a[(i + 1) % CHAINS]) : "vcc");
			if (OP == 25) asm volatile("v_cmp_lt_u64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(d[i]), "v"(d[(i + 1) % CHAINS]), "v"(a[(i + 1) % CHAINS]) : "vcc");
			if (OP == 26) asm volatile("v_cmp_class_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(d[i]), "v"(a[(i + 2) % CHAINS]), "v"(a[(i + 1) % CHAINS]) : "vcc");
			if (OP == 27) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) % CHAINS]) : "vcc"); // the select alone
			if (OP == 28) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) % CHAINS]));
			if (OP == 29) asm volatile("v_cmp_lt_f64 s[20:21], %1, %2\n\tv_cndmask_b32 %0, %0, %3, s[20:21]" : "+v"(a[i]) : "v"(d[i]), "v"(d[(i + 1) % CHAINS]), "v"(a[(i + 1) % CHAINS]) : "s20", "s21");
			if (OP == 30) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cmp_lt_f64 s[20:21], %2, %1\n\ts_and_b64 vcc, vcc, s[20:21]\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(d[i]), "v"(d[(i + 1) % CHAINS]), "v"(a[(i + 1) % CHAINS]) : "vcc", "s20", "s21", "scc");
			if (OP == 31) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(a[(i + 2) % CHAINS]), "v"(a[(i + 3) % CHAINS]), "v"(a[(i + 1) % CHAINS]) : "vcc");
			if (OP == 32) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(d[(i + 1) % CHAINS]));
			if (OP == 33) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc" : "+v"(a[i]), "+v"(a[(i + 4) % CHAINS]) : "v"(a[(i + 1) % CHAINS]), "v"(a[(i + 2) % CHAINS]) : "vcc");
		}
	}
	uint32_t r = 0;
	for (int i = 0; i < CHAINS; i++) r ^= a[i] ^ (uint32_t)(long long)d[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP>
void run(const char *name, int ops_per_step) {
	uint32_t *out;
	const int blocks = 256 * 8, n = 4096;
	hipMalloc(&out, blocks * 256 * 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	k<OP><<<blocks, 256>>>(out, 16, 1);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	k<OP><<<blocks, 256>>>(out, n, 1);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	double wave_instr = (double)blocks * 4 * n * CHAINS; // wave-level chain steps
	// per SIMD: 256 CUs * 4 SIMDs
	double per_simd = wave_instr / (256.0 * 4.0);
	printf("%-28s %8.3f ms  -> %.2f ns per wave-step per SIMD  (%d op(s) per step)\n", name, ms, ms * 1e6 / per_simd, ops_per_step);
	hipFree(out);
}
int main() {
	run<0>("add+xor+shift (3 int ops)", 3);
	run<1>("v_mul_lo_u32", 1);
	run<2>("v_mul_hi_u32 + add", 2);
	run<3>("mad_u64_u32 + xor", 2);
	run<4>("v_mul_f64", 1);
	run<5>("v_fma_f64", 1);
	run<6>("v_add_f64", 1);
	run<7>("v_rcp_f64 + add", 2);
	run<8>("v_rsq_f64 + add", 2);
	run<9>("v_div_fixup_f64", 1);
	run<10>("v_div_fmas_f64 + and + cmp", 3);
	run<11>("v_div_scale_f64", 1);
	run<12>("v_ldexp_f64 + mul", 2);
	run<13>("v_rndne_f64 + mul", 2);
	run<14>("cvt_i32_f64 + cvt_f64_i32 + add", 3);
	run<15>("cmp_lt_f64 + add + 2 cndmask", 4);
	run<16>("IEEE f64 division", 13);
	run<17>("IEEE f64 sqrt + add", 20);
	run<18>("v_mul_u32_u24 + add (or v_mad_u32_u24)", 2);
	run<19>("v_mul_hi_u32_u24 + add", 2);
	run<20>("cvt_f64_u32 + mul_f64 + cvt_u32_f64", 3);
	run<21>("v_xor3 + add", 2);
	run<22>("add + cmp + mul + 2 cndmask", 5);
	run<23>("v_cmp_lt_f64 (vcc) + cndmask", 2);
	run<24>("v_cmp_lt_u32 (vcc) + cndmask", 2);
	run<25>("v_cmp_lt_u64 (vcc) + cndmask", 2);
	run<26>("v_cmp_class_f64 + cndmask", 2);
	run<27>("v_cndmask_b32", 1);
	run<28>("v_max_f64", 1);
	run<29>("v_cmp_lt_f64 (sgpr) + cndmask", 2);
	run<30>("2 v_cmp_lt_f64 + s_and + cndmask", 4);
	run<31>("v_cmp_lt_f32 (vcc) + cndmask", 2);
	run<32>("v_mov_b64", 1);
	run<33>("2 v_cndmask_b32", 2);
	return 0;
}
