// Semantics check of global_load_lds_dwordx4 on gfx950 (the record staging of grid_walk.hpp relies on it): lane l's 16 bytes land at
// the wave-uniform LDS base + 16 * l, per-lane global addresses are honoured, lanes outside exec write nothing, completion = vmcnt; the instruction offset
// applies to the global AND the LDS address (form 1).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_dma_test tools/microbench/lds_dma_test.hip && /tmp/lds_dma_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GLOBAL __attribute__((address_space(1)))
#define LDS __attribute__((address_space(3)))
typedef uint4 Stage[5][64];
__global__ void k(const unsigned char *g, const unsigned *rec, uint4 *out, unsigned stage_offset, int use_inst_offset) {
	extern __shared__ __align__(16) unsigned char smem[];
	Stage *buf = reinterpret_cast<Stage *>(smem + stage_offset); // one stage per wave, at a chosen distance into the 160 KB
	const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	for (int p = 0; p < 5; p++) buf[wave][p][lane] = make_uint4(0xdeadbeefu, 0, 0, 0);
	__syncthreads();
	const unsigned char *src = g + (size_t)rec[threadIdx.x] * 80u;
	if (lane != 7u) { // lane 7 sits out: its slots must keep the fill value
		if (use_inst_offset) { // the form grid_walk.hpp uses: one address register, the piece selected by the instruction offset
			const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)&buf[wave][0][0]);
			asm volatile("s_mov_b32 m0, %[l]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g], off\n\t"
			             "s_add_u32 m0, m0, 0x3f0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g], off offset:16\n\t"
			             "s_add_u32 m0, m0, 0x3f0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g], off offset:32\n\t"
			             "s_add_u32 m0, m0, 0x3f0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g], off offset:48\n\t"
			             "s_add_u32 m0, m0, 0x3f0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[g], off offset:64"
			             :
			             : [l] "s"(lds0), [g] "v"((const GLOBAL unsigned char *)src)
			             : "memory", "scc");
		} else {
			for (int p = 0; p < 5; p++)
				__builtin_amdgcn_global_load_lds((const GLOBAL void *)(src + p * 16), (LDS void *)&buf[wave][p][0], 16, 0, 0);
		}
	}
	__builtin_amdgcn_s_waitcnt(0);
	for (int p = 0; p < 5; p++) out[(threadIdx.x) * 5 + p] = buf[wave][p][lane];
}
int main() {
	const int n_rec = 1000, threads = 128;
	std::vector<unsigned> data(n_rec * 20), rec(threads);
	for (size_t i = 0; i < data.size(); i++) data[i] = (unsigned)i * 2654435761u;
	for (int i = 0; i < threads; i++) rec[i] = (unsigned)((i * 37 + 11) % n_rec);
	unsigned char *d_g; unsigned *d_rec; uint4 *d_out;
	hipMalloc((void **)&d_g, data.size() * 4), hipMalloc((void **)&d_rec, threads * 4), hipMalloc((void **)&d_out, threads * 80);
	hipMemcpy(d_g, data.data(), data.size() * 4, hipMemcpyHostToDevice), hipMemcpy(d_rec, rec.data(), threads * 4, hipMemcpyHostToDevice);
	hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	int failed = 0;
	for (int form = 0; form < 2; form++)
	for (unsigned off : {0u, 60u * 1024u, 64u * 1024u, 100u * 1024u, 140u * 1024u}) {
		hipMemset(d_out, 0, threads * 80);
		hipLaunchKernelGGL(k, dim3(1), dim3(threads), off + 2 * sizeof(Stage), 0, d_g, d_rec, d_out, off, form);
		std::vector<unsigned> out(threads * 20);
		hipMemcpy(out.data(), d_out, threads * 80, hipMemcpyDeviceToHost);
		int bad = 0;
		for (int t = 0; t < threads; t++)
			for (int j = 0; j < 20; j++) {
				unsigned want = (t & 63) == 7 ? (j % 4 == 0 ? 0xdeadbeefu : 0u) : data[rec[t] * 20 + j];
				if (out[t * 20 + j] != want && bad++ < 4) std::printf("offset %u thread %d word %d: got %08x want %08x\n", off, t, j, out[t * 20 + j], want);
			}
		std::printf(bad ? "form %d stage at %u: FAILED, %d words differ\n" : "form %d stage at %u: ok (%d)\n", form, off, bad);
		failed += bad != 0;
	}
	return failed != 0;
}
