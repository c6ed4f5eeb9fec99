// tools/microbench_lonewave.hip -- what a wave that is ALONE on its SIMD pays per instruction when the instructions DEPEND on each
// other (kv2_engine's loops are one long dependent chain): cycles per instruction for chains of several instruction kinds, and for the
// hand-overs between the vector and the scalar unit.  One wave per CU, s_memtime around 64 x 16 instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_lonewave.hip -o /tmp/mb_lw && /tmp/mb_lw
#include <hip/hip_runtime.h>
#include <cstdio>
#define R16(x) x x x x x x x x x x x x x x x x
#define BENCH(name, init, body) \
	__global__ __launch_bounds__(64) void name(unsigned long long* out, float seed) { \
		float a = seed, b = seed * 0.5f, c = 0.25f; int i = (int)seed, j = 3; unsigned long long m = 0; (void)a; (void)b; (void)c; (void)i; (void)j; (void)m; \
		init; \
		unsigned long long t0, t1; \
		asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)); \
		for (int k = 0; k < 64; k++) { R16(body) } \
		asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)); \
		if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0; \
		if (seed == 12345.0f) out[1] = (unsigned long long)(a + b + c) + i + j + m; \
	}
BENCH(k_dep_add, , asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));)
BENCH(k_dep_mul_add, , asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));)
BENCH(k_ind2_add, , asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(b) : "v"(c));)
BENCH(k_dep_xor, , asm volatile("v_xor_b32 %0, %0, %1" : "+v"(i) : "v"(j));)
BENCH(k_dep_floor, , asm volatile("v_floor_f32 %0, %0" : "+v"(a));)
BENCH(k_dep_bfe, , asm volatile("v_bfe_i32 %0, %0, 1, 3" : "+v"(i));)
BENCH(k_dep_bitop3, , asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x36" : "+v"(i) : "v"(j));)
BENCH(k_dep_fma, , asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c));)
BENCH(k_dep_cmp_cnd, , asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(c) : "vcc");)
BENCH(k_dep_cmp_sgpr_cnd, , asm volatile("v_cmp_gt_f32_e64 %1, %0, %2\n v_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(a), "=s"(m) : "v"(c));)
BENCH(k_dep_readfirstlane, , asm volatile("v_readfirstlane_b32 %1, %0\n v_add_u32 %0, %1, %0" : "+v"(i), "=s"(j));)
BENCH(k_dep_salu, , asm volatile("s_add_u32 %0, %0, 1" : "+s"(j) : : "scc");)
BENCH(k_dep_cmp_branch, , asm volatile("v_cmp_gt_f32 vcc, %0, %1\n s_cbranch_vccz 1f\n v_add_f32 %0, %0, %1\n1:" : "+v"(a) : "v"(c) : "vcc");)
typedef float v2f_t __attribute__((ext_vector_type(2)));
BENCH(k_dep_pk, v2f_t p = seed;, asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p));)
BENCH(k_lds_rt, __shared__ int sm[64]; sm[threadIdx.x] = 0;, asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(i));)
BENCH(k_mad24, , asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i) : "v"(j));)
BENCH(k_mul_lo, , asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i) : "v"(j));)
BENCH(k_lshr64, unsigned long long w = (unsigned long long)seed;, asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(w));)
BENCH(k_cvt, , asm volatile("v_cvt_u32_f32 %0, %1\n v_cvt_f32_u32 %1, %0" : "+v"(i), "+v"(a));)
BENCH(k_ffbl, , asm volatile("v_ffbl_b32 %0, %0" : "+v"(i));)
typedef void (*kern_t)(unsigned long long*, float);
int main() {
	unsigned long long* d; hipMalloc(&d, 8 * 512);
	struct { const char* n; kern_t k; int per; } e[] = {
		{ "dependent v_add_f32", k_dep_add, 1 }, { "dependent v_mul_f32 -> v_add_f32", k_dep_mul_add, 2 }, { "two independent v_add_f32 chains", k_ind2_add, 2 },
		{ "dependent v_xor_b32", k_dep_xor, 1 }, { "dependent v_floor_f32", k_dep_floor, 1 }, { "dependent v_bfe_i32", k_dep_bfe, 1 }, { "dependent v_bitop3_b32", k_dep_bitop3, 1 },
		{ "dependent v_fma_f32", k_dep_fma, 1 }, { "v_cmp (vcc) -> v_cndmask, dependent", k_dep_cmp_cnd, 2 }, { "v_cmp (sgpr pair) -> v_cndmask, dependent", k_dep_cmp_sgpr_cnd, 2 },
		{ "v_readfirstlane -> v_add (sgpr operand)", k_dep_readfirstlane, 2 }, { "dependent s_add_u32", k_dep_salu, 1 }, { "v_cmp -> s_cbranch_vccz (not taken) -> v_add", k_dep_cmp_branch, 3 },
		{ "dependent v_pk_add_f32", k_dep_pk, 1 }, { "ds_read_b32 round trip", k_lds_rt, 1 }, { "dependent v_mad_u32_u24", k_mad24, 1 }, { "dependent v_mul_lo_u32", k_mul_lo, 1 },
		{ "dependent v_lshrrev_b64", k_lshr64, 1 }, { "v_cvt_u32_f32 -> v_cvt_f32_u32", k_cvt, 2 }, { "dependent v_ffbl_b32", k_ffbl, 1 } };
	printf("a wave alone on its SIMD (256 one-wave workgroups), cycles (s_memtime) per group of instructions, 1024 groups\n");
	for (auto& x : e) {
		hipLaunchKernelGGL(x.k, dim3(256), dim3(64), 0, 0, d, 1.5f); hipDeviceSynchronize();
		hipLaunchKernelGGL(x.k, dim3(256), dim3(64), 0, 0, d, 1.5f); hipDeviceSynchronize();
		unsigned long long h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
		unsigned long long mn = ~0ull; for (auto v : h) if (v < mn) mn = v;
		printf("%-48s %8.2f per group of %d  (%.2f per instruction)\n", x.n, (double)mn / 1024.0, x.per, (double)mn / 1024.0 / x.per);
	}
	return 0;
}
