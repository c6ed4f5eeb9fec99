// tools/microbench_stream.hip -- read-bandwidth ceilings on MI355X for the access patterns of the front end.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_stream.hip -o tools/microbench_stream
#include <hip/hip_runtime.h>
#include <cstdio>

// (a) classic grid-stride streaming read, 16 B per lane
__global__ __launch_bounds__(256) void gridstride(const float4* __restrict__ in, float* out, size_t n4) {
	float acc = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
		float4 v = in[i];
		acc += v.x + v.y + v.z + v.w;
	}
	if (acc == 12345.678f) out[0] = acc;
}
// (b) span walk: each workgroup streams a contiguous span tile by tile (NV float4 per thread per tile), prefetch 1 tile
template <int NT, int NV, int LDSB>
__global__ __launch_bounds__(NT) void spanwalk(const float4* __restrict__ in, float* out, int tiles_per_span, size_t span_stride4) {
	extern __shared__ float4 lds[];
	const float4* src = in + (size_t)blockIdx.x * span_stride4;
	float4 pre[NV];
#pragma unroll
	for (int e = 0; e < NV; e++) pre[e] = src[e * NT + threadIdx.x];
	float acc = 0;
	for (int t = 0; t < tiles_per_span; t++) {
		float4 cur[NV];
#pragma unroll
		for (int e = 0; e < NV; e++) cur[e] = pre[e];
		const int tn = t + 1 < tiles_per_span ? t + 1 : t;
#pragma unroll
		for (int e = 0; e < NV; e++) pre[e] = src[(size_t)tn * NT * NV + e * NT + threadIdx.x];
#pragma unroll
		for (int e = 0; e < NV; e++) acc += cur[e].x + cur[e].y + cur[e].z + cur[e].w;
		if (LDSB && threadIdx.x == 0 && acc == 1.2345f) lds[0] = cur[0];
	}
	if (acc == 12345.678f) out[0] = acc;
}

// (c) the front end's own pattern: one-wave workgroups, an 8 KB tile per step fetched straight into LDS (global_load_lds, 16 B per
// lane), read back from LDS, and WR 16-byte stores per lane and tile (the 48 kHz output is 1/16 of the input: WR = 1 every second tile)
template <int WR>
__global__ __launch_bounds__(64) void ldsdma_walk(const float4* __restrict__ in, float4* __restrict__ wout, float* out, int tiles_per_span, size_t span_stride4) {
	__shared__ float4 tile[512];
	const float4* src = in + (size_t)blockIdx.x * span_stride4;
	float4* dst = wout + (size_t)blockIdx.x * (span_stride4 / (WR == 5 ? 8 : 16));
	const int lane = threadIdx.x;
	float acc = 0;
	for (int e = 0; e < 8; e++)
		__builtin_amdgcn_global_load_lds((const void*)(src + e * 64 + lane), (__attribute__((address_space(3))) void*)(tile + e * 64), 16, 0, 2);
	for (int t = 0; t < tiles_per_span; t++) {
		__builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0)
		float4 cur[8];
		for (int e = 0; e < 8; e++) cur[e] = tile[e * 64 + lane];
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
		const int tn = t + 1 < tiles_per_span ? t + 1 : t;
		for (int e = 0; e < 8; e++)
			__builtin_amdgcn_global_load_lds((const void*)(src + (size_t)tn * 512 + e * 64 + lane), (__attribute__((address_space(3))) void*)(tile + e * 64), 16, 0, 2);
		for (int e = 0; e < 8; e++) acc += cur[e].x + cur[e].y + cur[e].z + cur[e].w;
		const float4 v = make_float4(acc, cur[1].x, cur[2].y, cur[3].z);
		if (WR == 1 && (t & 1)) dst[(size_t)(t >> 1) * 64 + lane] = v;                              // 1 KB per wave every second tile
		if (WR == 2 && (t & 1)) { typedef float v4f __attribute__((ext_vector_type(4))); v4f nv = { v.x, v.y, v.z, v.w }; __builtin_nontemporal_store(nv, reinterpret_cast<v4f*>(dst + (size_t)(t >> 1) * 64 + lane)); } // the same, non-temporal
		if (WR == 3 && (t & 7) == 7) {                                                              // 4 KB per wave every eighth tile
			for (int e = 0; e < 4; e++) dst[(size_t)(t >> 3) * 256 + e * 64 + lane] = v;
		}
		if (WR == 5) dst[(size_t)t * 64 + lane] = v;                                                // 1 KB per wave every tile: 1/8 written
		if (WR == 4 && (t & 1)) { // 16 B per lane, but lanes 1 KB apart (64 B per line touched: partial lines)
			dst[(size_t)lane * 64 + (t >> 1)] = v;
		}
	}
	if (acc == 12345.678f) out[0] = acc;
}

// (d) writes alone: every wave stores NT x 1 KB, contiguous per workgroup
__global__ __launch_bounds__(64) void write_only(float4* __restrict__ w, int per_wg) {
	float4* dst = w + (size_t)blockIdx.x * per_wg * 64;
	const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
	for (int t = 0; t < per_wg; t++) dst[(size_t)t * 64 + threadIdx.x] = v;
}

template <typename F>
static void timeit(const char* name, F launch, double bytes) {
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	launch();
	hipDeviceSynchronize();
	hipEventRecord(a);
	for (int i = 0; i < 5; i++) launch();
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	ms /= 5;
	printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
}

int main() {
	const size_t bytes = (size_t)256 * 786432 * 8; // 1.61 GB, the bench block
	float4* d; float* o;
	hipMalloc(&d, bytes + (1 << 20)); hipMalloc(&o, 64);
	hipMemset(d, 1, bytes);
	const size_t n4 = bytes / 16;
	for (int g : { 2048, 8192, 32768 })
		timeit(("gridstride grid=" + std::to_string(g)).c_str(), [&] { hipLaunchKernelGGL(gridstride, dim3(g), dim3(256), 0, 0, d, o, n4); }, (double)bytes);
	// K1-like: 1024 workgroups of 256 threads, 32 KB tiles, 70 KB LDS (2 per CU)
	{
		auto k = spanwalk<256, 8, 1>;
		hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 70208);
		timeit("spanwalk 1024 WG x256, 32KB tile, 70KB LDS", [&] { hipLaunchKernelGGL(k, dim3(1024), dim3(256), 70208, 0, d, o, 48, (size_t)48 * 2048); }, (double)bytes);
		timeit("spanwalk 2048 WG x256, 32KB tile, 70KB LDS", [&] { hipLaunchKernelGGL(k, dim3(2048), dim3(256), 70208, 0, d, o, 24, (size_t)24 * 2048); }, (double)bytes);
	}
	{
		auto k = spanwalk<256, 8, 0>;
		timeit("spanwalk 1024 WG x256, 32KB tile, no LDS", [&] { hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, o, 48, (size_t)48 * 2048); }, (double)bytes);
		timeit("spanwalk 4096 WG x256, 32KB tile, no LDS", [&] { hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, o, 12, (size_t)12 * 2048); }, (double)bytes);
	}
	{
		auto k = spanwalk<64, 8, 1>;
		hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 18432);
		timeit("spanwalk 8192 WG x64, 8KB tile, 18KB LDS", [&] { hipLaunchKernelGGL(k, dim3(8192), dim3(64), 18432, 0, d, o, 24, (size_t)24 * 512); }, (double)bytes);
		auto k2 = spanwalk<64, 8, 0>;
		timeit("spanwalk 8192 WG x64, 8KB tile, no LDS", [&] { hipLaunchKernelGGL(k2, dim3(8192), dim3(64), 0, 0, d, o, 24, (size_t)24 * 512); }, (double)bytes);
		timeit("spanwalk 32768 WG x64, 8KB tile, no LDS", [&] { hipLaunchKernelGGL(k2, dim3(32768), dim3(64), 0, 0, d, o, 6, (size_t)6 * 512); }, (double)bytes);
	}
	{
		float4* w; hipMalloc(&w, bytes / 16 + (1 << 20));
		timeit("LDS-DMA walk 8192 WG x64, 8KB tile, read only", [&] { hipLaunchKernelGGL(ldsdma_walk<0>, dim3(8192), dim3(64), 0, 0, d, w, o, 24, (size_t)24 * 512); }, (double)bytes);
		timeit("LDS-DMA walk 8192 WG x64, 8KB tile, + 1/16 written", [&] { hipLaunchKernelGGL(ldsdma_walk<1>, dim3(8192), dim3(64), 0, 0, d, w, o, 24, (size_t)24 * 512); }, (double)bytes * (1.0 + 1.0 / 16));
		timeit("LDS-DMA walk 6144 WG x64 (32 tiles), + 1/16 written", [&] { hipLaunchKernelGGL(ldsdma_walk<1>, dim3(6144), dim3(64), 0, 0, d, w, o, 32, (size_t)32 * 512); }, (double)bytes * (1.0 + 1.0 / 16));
		{
			float4* w2; hipMalloc(&w2, bytes / 8 + (1 << 20));
			timeit("  ... 1/8 written (1 KB per wave every tile)", [&] { hipLaunchKernelGGL(ldsdma_walk<5>, dim3(8192), dim3(64), 0, 0, d, w2, o, 24, (size_t)24 * 512); }, (double)bytes * (1.0 + 1.0 / 8));
			timeit("writes alone, 0.2 GB (8192 WG x 24 KB)", [&] { hipLaunchKernelGGL(write_only, dim3(8192), dim3(64), 0, 0, w2, 24); }, (double)8192 * 24 * 1024);
			timeit("writes alone, 1.61 GB (8192 WG x 192 KB)", [&] { hipLaunchKernelGGL(write_only, dim3(8192), dim3(64), 0, 0, d, 192); }, (double)8192 * 192 * 1024);
		}
		timeit("  ... non-temporal stores", [&] { hipLaunchKernelGGL(ldsdma_walk<2>, dim3(8192), dim3(64), 0, 0, d, w, o, 24, (size_t)24 * 512); }, (double)bytes * (1.0 + 1.0 / 16));
		timeit("  ... 4 KB per wave every eighth tile", [&] { hipLaunchKernelGGL(ldsdma_walk<3>, dim3(8192), dim3(64), 0, 0, d, w, o, 24, (size_t)24 * 512); }, (double)bytes * (1.0 + 1.0 / 16));
		timeit("  ... 16 B per lane, lanes 1 KB apart", [&] { hipLaunchKernelGGL(ldsdma_walk<4>, dim3(8192), dim3(64), 0, 0, d, w, o, 24, (size_t)24 * 512); }, (double)bytes * (1.0 + 1.0 / 16));
	}
	return 0;
}
