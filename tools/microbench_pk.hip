// tools/microbench_pk.hip -- packed (v_pk_mul_f32 / v_pk_add_f32) against scalar f32 VALU on gfx950, as a dependent recurrence
// (the CGF phasor: rot *= step, one wave per SIMD) and as a throughput stream (many waves).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench_pk.hip -o /tmp/mb_pk && /tmp/mb_pk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float smul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sadd(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float ssub(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__global__ __launch_bounds__(64) void rec_packed(float2* io, int steps) {
	float2 r0 = io[blockIdx.x * 64 + threadIdx.x];
	v2f cur = { r0.x, r0.y };
	const v2f st = { 0.99999f, 0.0031f }, st_sw = { -0.0031f, 0.99999f };
#pragma unroll 8
	for (int k = 0; k < steps; k++) cur = cur.xx * st + cur.yy * st_sw;
	io[blockIdx.x * 64 + threadIdx.x] = make_float2(cur.x, cur.y);
}
__global__ __launch_bounds__(64) void rec_scalar(float2* io, int steps) {
	float2 r0 = io[blockIdx.x * 64 + threadIdx.x];
	float x = r0.x, y = r0.y;
	const float sx = 0.99999f, sy = 0.0031f, nsy = -0.0031f;
#pragma unroll 8
	for (int k = 0; k < steps; k++) {
		const float a = smul(x, sx), b = smul(y, nsy), c = smul(x, sy), d = smul(y, sx);
		x = sadd(a, b); y = sadd(c, d);
	}
	io[blockIdx.x * 64 + threadIdx.x] = make_float2(x, y);
}
// two lanes per chain: each lane one component, the other one's through DPP (quad_perm [1,0,3,2])
__global__ __launch_bounds__(64) void rec_dpp(float* io, int steps) {
	float v = io[blockIdx.x * 64 + threadIdx.x];
	const bool im = threadIdx.x & 1;
	const float s_own = 0.99999f, s_oth = im ? 0.0031f : -0.0031f;
#pragma unroll 8
	for (int k = 0; k < steps; k++) {
		const float o = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
		v = sadd(smul(v, s_own), smul(o, s_oth));
	}
	io[blockIdx.x * 64 + threadIdx.x] = v;
}
// throughput: 8 independent accumulators per lane
__global__ __launch_bounds__(256) void thr_packed(float2* io, int steps) {
	v2f a[8];
	for (int i = 0; i < 8; i++) { float2 t = io[(blockIdx.x * 256 + threadIdx.x) * 8 + i]; a[i] = v2f{ t.x, t.y }; }
	const v2f m = { 0.999f, 1.001f }, c = { 0.001f, -0.001f };
	for (int k = 0; k < steps; k++)
#pragma unroll
		for (int i = 0; i < 8; i++) a[i] = a[i] * m + c;
	for (int i = 0; i < 8; i++) io[(blockIdx.x * 256 + threadIdx.x) * 8 + i] = make_float2(a[i].x, a[i].y);
}
__global__ __launch_bounds__(256) void thr_scalar(float2* io, int steps) {
	float ax[8], ay[8];
	for (int i = 0; i < 8; i++) { float2 t = io[(blockIdx.x * 256 + threadIdx.x) * 8 + i]; ax[i] = t.x; ay[i] = t.y; }
	for (int k = 0; k < steps; k++)
#pragma unroll
		for (int i = 0; i < 8; i++) { ax[i] = sadd(smul(ax[i], 0.999f), 0.001f); ay[i] = sadd(smul(ay[i], 1.001f), -0.001f); }
	for (int i = 0; i < 8; i++) io[(blockIdx.x * 256 + threadIdx.x) * 8 + i] = make_float2(ax[i], ay[i]);
}

template <class F> static float timeit(F f) {
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	f(); hipDeviceSynchronize();
	hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
	float2* d; hipMalloc(&d, 256 << 20); hipMemset(d, 0, 256 << 20);
	const int steps = 1 << 20;
	float t;
	t = timeit([&] { hipLaunchKernelGGL(rec_packed, dim3(8), dim3(64), 0, 0, d, steps); });
	printf("recurrence, packed (3 v_pk per step), 8 waves:  %.3f ms  %.1f ns/step\n", t, t * 1e6 / steps);
	t = timeit([&] { hipLaunchKernelGGL(rec_scalar, dim3(8), dim3(64), 0, 0, d, steps); });
	printf("recurrence, scalar (4 mul + 2 add), 8 waves:    %.3f ms  %.1f ns/step\n", t, t * 1e6 / steps);
	t = timeit([&] { hipLaunchKernelGGL(rec_dpp, dim3(16), dim3(64), 0, 0, (float*)d, steps); });
	printf("recurrence, 2 lanes per chain (dpp), 16 waves:  %.3f ms  %.1f ns/step\n", t, t * 1e6 / steps);
	const int ts = 1 << 12, blocks = 256 * 8 * 4;
	t = timeit([&] { hipLaunchKernelGGL(thr_packed, dim3(blocks), dim3(256), 0, 0, d, ts); });
	printf("throughput, packed: %.3f ms  %.2f Tflop/s (mul+add)\n", t, (double)blocks * 256 * 8 * 2 * 2 * ts / (t * 1e-3) / 1e12);
	t = timeit([&] { hipLaunchKernelGGL(thr_scalar, dim3(blocks), dim3(256), 0, 0, d, ts); });
	printf("throughput, scalar: %.3f ms  %.2f Tflop/s (mul+add)\n", t, (double)blocks * 256 * 8 * 2 * 2 * ts / (t * 1e-3) / 1e12);
	return 0;
}
