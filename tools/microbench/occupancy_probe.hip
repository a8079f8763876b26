// How many workgroups of a given shape does a CU of this device hold at once?  Asks the runtime (hipOccupancyMaxActiveBlocksPerMultiprocessor) and MEASURES it:
// a kernel whose workgroups record, by an atomic counter per CU (the hardware's CU id), how many of them were resident together.
// hipcc --offload-arch=gfx950 -O2 tools/microbench/occupancy_probe.hip -o tools/microbench/bin/occupancy_probe && tools/microbench/bin/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int VGPRS>
__global__ __launch_bounds__(1024) void resident_kernel(unsigned *now, unsigned *peak, unsigned long long spin, double *sink) {
	extern __shared__ unsigned char lds[];
	// a few live registers (VGPRS / 2 doubles)
	double acc[VGPRS / 2];
#pragma unroll
	for (int i = 0; i < VGPRS / 2; i++) acc[i] = (double)(threadIdx.x + i);
	unsigned cu;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cu));
	const unsigned key = (cu >> 8) & 0xFFu; // HW_ID bits 15:8 = cu id (4) | sh id (1) | se id (3): one value per CU of an XCD
	unsigned xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	const unsigned slot = ((xcc & 0xFu) << 11) | (key & 0x7FFu);
	if (threadIdx.x == 0) {
		const unsigned n = atomicAdd(&now[slot], 1u) + 1u;
		atomicMax(&peak[slot], n);
	}
	lds[threadIdx.x] = (unsigned char)threadIdx.x;
	__syncthreads();
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < spin) {
#pragma unroll
		for (int i = 0; i < VGPRS / 2; i++) acc[i] = acc[i] * 1.0000001 + (double)lds[(threadIdx.x + i) & 1023];
	}
	double s = 0;
	for (int i = 0; i < VGPRS / 2; i++) s += acc[i];
	if (s == 12345.678) sink[0] = s;
	__syncthreads();
	if (threadIdx.x == 0) atomicSub(&now[slot], 1u);
}

template <int VGPRS>
void probe(int threads, size_t lds_bytes) {
	hipFuncSetAttribute(reinterpret_cast<const void *>(&resident_kernel<VGPRS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	int blocks = -1;
	hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, resident_kernel<VGPRS>, threads, lds_bytes);
	unsigned *now, *peak;
	double *sink;
	hipMalloc(&now, 65536 * 4), hipMalloc(&peak, 65536 * 4), hipMalloc(&sink, 8);
	hipMemset(now, 0, 65536 * 4), hipMemset(peak, 0, 65536 * 4);
	hipFuncAttributes fa;
	hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&resident_kernel<VGPRS>));
	hipLaunchKernelGGL(resident_kernel<VGPRS>, dim3(256 * 8), dim3(threads), lds_bytes, 0, now, peak, 20000000ull /* 0.2 s at 100 MHz */, sink);
	hipError_t e = hipDeviceSynchronize();
	std::vector<unsigned> h(65536);
	hipMemcpy(h.data(), peak, 65536 * 4, hipMemcpyDeviceToHost);
	unsigned mx = 0, used = 0;
	unsigned long long sum = 0;
	for (unsigned v : h)
		if (v) mx = v > mx ? v : mx, used++, sum += v;
	std::printf("VGPRs asked %3d (compiled %3d) threads %4d LDS %6zu B: runtime says %d workgroups per CU; measured peak per CU slot %u (slots seen %u, mean peak %.2f) %s\n", VGPRS,
	            fa.numRegs, threads, lds_bytes, blocks, mx, used, used ? (double)sum / used : 0.0, e == hipSuccess ? "" : hipGetErrorString(e));
	hipFree(now), hipFree(peak), hipFree(sink);
}

int main() {
	probe<16>(1024, 158 * 1024);
	probe<16>(640, 75 * 1024);
	probe<16>(640, 79 * 1024);
	probe<16>(640, 64 * 1024);
	probe<16>(640, 32 * 1024);
	probe<16>(512, 75 * 1024);
	probe<16>(320, 37 * 1024);
	probe<16>(256, 30 * 1024);
	return 0;
}
