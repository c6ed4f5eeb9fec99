// tools/microbench_anyorder.hip -- does hipExtAnyOrderLaunch let the next kernel of a stream begin while the previous one drains?
// (hip_ext.h says the flag is "not supported on AMD GFX9xx boards" for the module launch; this measures what gfx950 does.)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_anyorder.hip -o /tmp/mb_anyorder && /tmp/mb_anyorder
// Test 1: two launches of a kernel whose workgroup 0 runs 2 ms and all others 10 us.  In order: 4 ms; overlapped: ~2 ms.
// Test 2: the front end's shape -- 6,144 one-wave workgroups of ~100 us +- 20 % on 3,072 slots (LDS-capped), 20 launches back to back.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ __launch_bounds__(64) void spin(unsigned long long* sink, long long long_ticks, long long short_ticks, int ragged) {
	__shared__ char cap[13000]; // 12 workgroups per CU
	cap[threadIdx.x] = 0;
	long long ticks = short_ticks;
	if (ragged) ticks = short_ticks + (long long)((blockIdx.x * 2654435761u >> 16) % 40) * short_ticks / 100; // +0 .. 39 %
	else if (blockIdx.x == 0) ticks = long_ticks;
	const long long t0 = wall_clock64(); // 100 MHz
	while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
	if (threadIdx.x == 0 && cap[1] == 77) sink[0] = t0;
}

static float run(int mode, int launches, int grid, long long lt, long long st, int ragged, unsigned long long* d) {
	hipStream_t s; hipStreamCreate(&s);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	float best = 1e30f;
	for (int rep = 0; rep < 3; rep++) {
		hipEventRecord(a, s);
		for (int i = 0; i < launches; i++) {
			if (mode == 0) hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, s, d, lt, st, ragged);
			else hipExtLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, s, nullptr, nullptr, mode == 2 ? hipExtAnyOrderLaunch : 0, d, lt, st, ragged);
		}
		hipEventRecord(b, s);
		hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		if (ms < best) best = ms;
	}
	hipStreamDestroy(s);
	return best;
}

int main() {
	unsigned long long* d; hipMalloc(&d, 64);
	const char* names[3] = { "hipLaunchKernelGGL", "hipExtLaunchKernelGGL flags=0", "hipExtLaunchKernelGGL any-order" };
	printf("test 1: 2 launches, workgroup 0 runs 2 ms, 3071 others 10 us\n");
	for (int m = 0; m < 3; m++) printf("  %-34s %.3f ms\n", names[m], run(m, 2, 3072, 200000, 1000, 0, d));
	printf("test 2: 20 launches of 6144 workgroups (100 us +0..39 %%) on 3072 slots\n");
	for (int m = 0; m < 3; m++) { const float t = run(m, 20, 6144, 0, 10000, 1, d); printf("  %-34s %.3f ms  (%.4f per launch)\n", names[m], t, t / 20); }
	printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
	return 0;
}
