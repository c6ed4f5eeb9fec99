// Issue cost of v_lshrrev_b64 against 32-bit VALU ops on gfx950 (one wave per SIMD and four waves per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench_shift tools/microbench_shift.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(unsigned long long* out, int iters, unsigned long long seed) {
	unsigned long long a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = seed * 5, d = seed * 7;
	unsigned sh = threadIdx.x & 31;
	long long t0 = clock64();
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int u = 0; u < 16; u++) {
			if (MODE == 0) { // four independent 64-bit shifts
				asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a) : "v"(sh));
				asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(b) : "v"(sh));
				asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(c) : "v"(sh));
				asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(d) : "v"(sh));
			} else if (MODE == 1) { // four independent 32-bit shifts
				unsigned &x = *(unsigned*)&a, &y = *(unsigned*)&b, &z = *(unsigned*)&c, &w = *(unsigned*)&d;
				asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x) : "v"(sh));
				asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(y) : "v"(sh));
				asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(z) : "v"(sh));
				asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(w) : "v"(sh));
			} else if (MODE == 2) { // packed fp32 multiply
				asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(a));
				asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(b));
				asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(c));
				asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(d));
			} else { // v_bfe_u32
				unsigned &x = *(unsigned*)&a, &y = *(unsigned*)&b, &z = *(unsigned*)&c, &w = *(unsigned*)&d;
				asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(x) : "v"(sh));
				asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(y) : "v"(sh));
				asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(z) : "v"(sh));
				asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(w) : "v"(sh));
			}
		}
	}
	long long t1 = clock64();
	if (threadIdx.x == 0) out[blockIdx.x] = (unsigned long long)(t1 - t0);
	if (a + b + c + d == 12345) out[0] = 1;
}
template <int MODE> void run(const char* name, int wg) {
	unsigned long long* d; hipMalloc(&d, 8 * 4096);
	const int iters = 2000;
	hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(wg), 0, 0, d, iters, 99ull); hipDeviceSynchronize();
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(wg), 0, 0, d, iters, 99ull); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	// one workgroup per CU: wg/64 waves per CU = wg/256 per SIMD
	const double instr_per_wave = (double)iters * 64;
	printf("%-16s wg %4d: %.3f ms -> %.2f ns per wave-instruction per SIMD slot (%.1f cycles at 2.4 GHz, %d wave(s)/SIMD)\n", name, wg, ms,
	       ms * 1e6 / (instr_per_wave * (wg >= 256 ? wg / 256 : 1)), ms * 1e6 / (instr_per_wave * (wg >= 256 ? wg / 256 : 1)) * 2.4, wg >= 256 ? wg / 256 : 1);
	hipFree(d);
}
int main() {
	for (int wg : { 256, 1024 }) { run<0>("v_lshrrev_b64", wg); run<1>("v_lshrrev_b32", wg); run<2>("v_pk_mul_f32", wg); run<3>("v_bfe_u32", wg); }
	return 0;
}
