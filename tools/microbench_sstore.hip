// Do scalar stores work on gfx950?  build: hipcc --offload-arch=gfx950 -O2 tools/microbench_sstore.hip -o tools/microbench_sstore
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned long long* out) {
	unsigned long long b = __ballot((threadIdx.x & 3) == 1) + blockIdx.x;
	unsigned long long* p = out + blockIdx.x;
	asm volatile("s_store_dwordx2 %0, %1, 0x0\n\ts_dcache_wb" ::"s"(b), "s"(p) : "memory");
}
int main() {
	unsigned long long* d;
	hipMalloc(&d, 64 * 8);
	hipMemset(d, 0, 64 * 8);
	hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d);
	hipError_t e = hipDeviceSynchronize();
	printf("sync: %s\n", hipGetErrorString(e));
	unsigned long long h[64];
	hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	int ok = 0;
	for (int i = 0; i < 64; i++) ok += h[i] == 0x2222222222222222ull + i;
	printf("scalar store: %d/64 correct (first %llx)\n", ok, h[0]);
	return 0;
}
