// tools/microbench_clock.hip -- what does s_memtime count, and how fast does a lightly loaded MI355X run?  One wave per CU (or per SIMD)
// runs a dependent v_add_f32 chain for a fixed number of instructions; wall time from HIP events, ticks from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R16(x) x x x x x x x x x x x x x x x x
__global__ __launch_bounds__(64) void k_chain(unsigned long long* out, float seed, int iters) {
	float a = seed, c = 0.25f;
	unsigned long long t0, t1;
	asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
	for (int k = 0; k < iters; k++) { R16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));) }
	asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
	if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
	if (seed == 12345.0f) out[1] = (unsigned long long)a;
}
int main() {
	unsigned long long* d; hipMalloc(&d, 8 * 4096);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int grid : { 256, 1024, 4096 }) for (int rep = 0; rep < 3; rep++) {
		const int iters = 200000; // 3.2 M dependent instructions
		hipEventRecord(a); hipLaunchKernelGGL(k_chain, dim3(grid), dim3(64), 0, 0, d, 1.5f, iters); hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
		printf("grid %4d: %.3f ms, %llu ticks -> s_memtime %.1f MHz; %.2f ticks, %.2f ns per dependent v_add_f32\n", grid, ms, h, (double)h / ms / 1e3, (double)h / (16.0 * iters), ms * 1e6 / (16.0 * iters));
	}
	return 0;
}
