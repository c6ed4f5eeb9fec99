// tools/microbench_rot.hip -- how fast can ONE wave run a strictly sequential complex-product recurrence
// on gfx950?  (The CGF derotation, DSP/DSP.cpp:460-463, is 24,576 dependent steps per channel per block.)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench_rot.hip -o tools/microbench_rot
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define DPP_SHR1(oldv, srcv) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(oldv), __float_as_int(srcv), 0x138, 0xF, 0xF, false))

// V0: merged register (recurrence in lane 0 + history in lanes 1..63), packed math, 5 ops/step
__global__ __launch_bounds__(64) void v0(float2* out, float2 stp, int nchunk) {
	v2f cur = { 1.0f, 0.0f };
	const v2f st = { stp.x, stp.y }, sw = { -stp.y, stp.x };
	for (int c = 0; c < nchunk; c++) {
#pragma unroll
		for (int k = 0; k < 64; k++) {
			v2f nw = cur.xx * st + cur.yy * sw;
			nw.x = DPP_SHR1(nw.x, cur.x);
			nw.y = DPP_SHR1(nw.y, cur.y);
			cur = nw;
		}
		out[(size_t)blockIdx.x * 64 + threadIdx.x] = make_float2(cur.x, cur.y);
	}
}
// V1: recurrence register kept apart from the history register (DPP off the dependency chain)
__global__ __launch_bounds__(64) void v1(float2* out, float2 stp, int nchunk) {
	v2f rot = { 1.0f, 0.0f };
	const v2f st = { stp.x, stp.y }, sw = { -stp.y, stp.x };
	float hx = 0, hy = 0;
	for (int c = 0; c < nchunk; c++) {
#pragma unroll
		for (int k = 0; k < 64; k++) {
			rot = rot.xx * st + rot.yy * sw;
			hx = DPP_SHR1(rot.x, hx);
			hy = DPP_SHR1(rot.y, hy);
		}
		out[(size_t)blockIdx.x * 64 + threadIdx.x] = make_float2(hx, hy);
	}
}
// V2: scalar (unpacked) math, history apart
__global__ __launch_bounds__(64) void v2(float2* out, float2 stp, int nchunk) {
	float rx = 1.0f, ry = 0.0f, hx = 0, hy = 0;
	for (int c = 0; c < nchunk; c++) {
#pragma unroll
		for (int k = 0; k < 64; k++) {
			float a = rx * stp.x, b = ry * stp.y, cc = rx * stp.y, d = ry * stp.x;
			rx = a - b; ry = cc + d;
			hx = DPP_SHR1(rx, hx);
			hy = DPP_SHR1(ry, hy);
		}
		out[(size_t)blockIdx.x * 64 + threadIdx.x] = make_float2(hx, hy);
	}
}
// V3: floor, packed recurrence only (no history)
__global__ __launch_bounds__(64) void v3(float2* out, float2 stp, int nchunk) {
	v2f rot = { 1.0f, 0.0f };
	const v2f st = { stp.x, stp.y }, sw = { -stp.y, stp.x };
	for (int c = 0; c < nchunk; c++) {
#pragma unroll
		for (int k = 0; k < 64; k++) rot = rot.xx * st + rot.yy * sw;
		out[(size_t)blockIdx.x * 64 + threadIdx.x] = make_float2(rot.x, rot.y);
	}
}
// V4: floor, scalar recurrence only
__global__ __launch_bounds__(64) void v4(float2* out, float2 stp, int nchunk) {
	float rx = 1.0f, ry = 0.0f;
	for (int c = 0; c < nchunk; c++) {
#pragma unroll
		for (int k = 0; k < 64; k++) {
			float a = rx * stp.x, b = ry * stp.y, cc = rx * stp.y, d = ry * stp.x;
			rx = a - b; ry = cc + d;
		}
		out[(size_t)blockIdx.x * 64 + threadIdx.x] = make_float2(rx, ry);
	}
}
// V5: history latched into LDS by lane 0 only (exec-masked ds_write), packed math
__global__ __launch_bounds__(64) void v5(float2* out, float2 stp, int nchunk) {
	__shared__ float2 ring[64];
	v2f rot = { 1.0f, 0.0f };
	const v2f st = { stp.x, stp.y }, sw = { -stp.y, stp.x };
	for (int c = 0; c < nchunk; c++) {
#pragma unroll
		for (int k = 0; k < 64; k++) {
			rot = rot.xx * st + rot.yy * sw;
			if (threadIdx.x == 0) ring[k] = make_float2(rot.x, rot.y);
		}
		__syncthreads();
		out[(size_t)blockIdx.x * 64 + threadIdx.x] = ring[threadIdx.x];
		__syncthreads();
	}
}
// V6: two independent chains interleaved in one wave (does ILP hide the dependent latency?)
__global__ __launch_bounds__(64) void v6(float2* out, float2 stp, int nchunk) {
	v2f r0 = { 1.0f, 0.0f }, r1 = { 0.0f, 1.0f };
	const v2f st = { stp.x, stp.y }, sw = { -stp.y, stp.x };
	for (int c = 0; c < nchunk; c++) {
#pragma unroll
		for (int k = 0; k < 64; k++) {
			r0 = r0.xx * st + r0.yy * sw;
			r1 = r1.xx * st + r1.yy * sw;
		}
		out[(size_t)blockIdx.x * 64 + threadIdx.x] = make_float2(r0.x + r1.x, r0.y + r1.y);
	}
}

template <typename F>
static void run(const char* name, F kern, int waves, int nchunk, float2* d) {
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	float2 stp = make_float2(0.99992470f, 0.01227154f);
	hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, d, stp, 8);
	hipDeviceSynchronize();
	hipEventRecord(a);
	hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, d, stp, nchunk);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	printf("%-28s waves=%5d steps=%7d  %8.3f ms  %7.2f ns/step\n", name, waves, nchunk * 64, ms, ms * 1e6 / (nchunk * 64.0));
}

int main() {
	float2* d;
	hipMalloc(&d, 8192 * 64 * sizeof(float2));
	const int nchunk = 24576 / 64;
	for (int waves : { 512, 1024, 2048, 4096 }) {
		run("v0 merged pk+dpp", v0, waves, nchunk, d);
		run("v1 pk, dpp off chain", v1, waves, nchunk, d);
		run("v2 scalar, dpp off chain", v2, waves, nchunk, d);
		run("v3 pk only (floor)", v3, waves, nchunk, d);
		run("v4 scalar only (floor)", v4, waves, nchunk, d);
		run("v5 pk + lds latch lane0", v5, waves, nchunk, d);
		run("v6 two chains per wave", v6, waves, nchunk, d);
	}
	return 0;
}
