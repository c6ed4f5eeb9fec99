// ais-catcher_amd/csrc/kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the AIS GMSK demodulation chain.
//
// Written for 64-wide wavefronts, 160 KiB LDS per CU and HBM3E streaming; compiled with
// -ffp-contract=off: every float operation below is an individually rounded IEEE binary32 op in
// exactly the association order of the reference, so results are
// bit-identical to the reference CPU chain built with strict FP flags.
// File:line citations are relative to the reference's Source/ directory.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace aisk {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// std::complex<float> product as the strict-FP reference evaluates it: (ac-bd, ad+bc), 4 mul + 2 add
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
	return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// glibc 2.35 hypotf == (float)sqrt((double)x*x + (double)y*y)  (sysdeps/ieee754/flt-32/e_hypotf.c);
// std::abs(std::complex<float>) in the reference resolves to it (DSP/DSP.cpp:315,434,439,449,465).
__device__ __forceinline__ float hypot_ref(float x, float y) {
	double dx = (double)x, dy = (double)y;
	return (float)__dsqrt_rn(dx * dx + dy * dy);
}

// Decimating CIC5 on a register chunk (DSP/DSP.cpp:85-117):
// v[0 .. 2*NOUT+3] = x[2*j0-5 .. 2*j0+2*NOUT-2]; out[q] = s4[2*(j0+q)] * 2^-5 with
// s_k[n] = s_{k-1}[n] + s_{k-1}[n-1].  In-place Pascal triangle; pairing identical to the reference.
template <int NOUT>
__device__ __forceinline__ void cic5_dec_chunk(float2 (&v)[2 * NOUT + 4], float2 (&out)[NOUT]) {
#pragma unroll
	for (int lvl = 0; lvl < 4; lvl++) {
#pragma unroll
		for (int i = 0; i < 2 * NOUT + 3 - lvl; i++) v[i] = cadd(v[i + 1], v[i]);
	}
#pragma unroll
	for (int q = 0; q < NOUT; q++) {
		float2 s = cadd(v[2 * q + 1], v[2 * q]);
		out[q] = make_float2(s.x * 0.03125f, s.y * 0.03125f);
	}
}

// ------------------------------------------------------------------------------------------
// K1: front end.  RAW -> [CU8 convert] -> K x Downsample2CIC5 -> FDC -> Rotate -> 2 x (DS2 + FilterCIC5)
//
// One workgroup (256 threads) streams a span of consecutive tiles of one receiver.  A tile is 256
// samples at 96 kHz (= 256 << K input samples); every stage keeps the few samples of history it
// needs in LDS between tiles, so nothing is recomputed inside a span.  A span starts with one
// warm-up tile whose outputs are discarded: every stage is feed-forward with a dependency cone of
// < 348 input samples, so after one tile all carried histories are exact.
// Global loads are fully coalesced float4 (16 B / lane) and are issued one tile ahead into registers
// (async-stage split), so HBM latency overlaps the LDS ladder of the current tile.
//
// LDS chunk layouts: a thread owns C consecutive samples of a stage input; rows of C samples are
// padded by one 16-byte slot so that the per-thread ds_read_b128 of a 16-lane group covers all 64
// banks (row stride 144/80/48 B -> conflict free).
// ------------------------------------------------------------------------------------------
constexpr int K1_THREADS = 256;
constexpr int R0 = 18, R1 = 10, R2 = 6; // padded row lengths (float2) for chunk sizes 16, 8, 4

struct __align__(16) K1Smem {
	float2 x0[256 * R0]; // stage-1 input (body). x2 aliases the front of it once stage 1 is done.
	float2 x1[256 * R1]; // stage-2 input
	float2 x3[8 + 512];  // stage-4 input, 8 leading history samples
	float2 x4[8 + 256];  // 96 kHz (FDC input)
	float2 x5[2][8 + 256]; // rotated up/down (DS2_a/b input)
	float2 x6[2][8 + 128]; // FilterCIC5 input
	float2 h0[8], h1[8], h2[8]; // last 8 samples of x0 / x1 / x2 of the previous tile
};

template <int K, bool CU8>
__global__ __launch_bounds__(K1_THREADS) void k1_frontend(K1Params p) {
	static_assert(K >= 1 && K <= 4, "ladder depth");
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	K1Smem& sm = *reinterpret_cast<K1Smem*>(smem_raw);
	float2* x2 = sm.x0; // alias: x0 is dead after stage 1 (its tail lives in h0)

	const int t = threadIdx.x;
	const int rx = blockIdx.y;
	const int span = blockIdx.x;
	constexpr int TILE_IN = 256 << K; // input samples per tile

	// zero all carried histories (true zero state at stream start; warm-up overwrites them otherwise)
	if (t < 8) {
		sm.h0[t] = sm.h1[t] = sm.h2[t] = make_float2(0.f, 0.f);
		sm.x3[t] = sm.x4[t] = make_float2(0.f, 0.f);
		sm.x5[0][t] = sm.x5[1][t] = sm.x6[0][t] = sm.x6[1][t] = make_float2(0.f, 0.f);
	}

	const int tile_first = span * p.tiles_per_span - 1; // warm-up tile
	int tile_last = tile_first + p.tiles_per_span;        // inclusive
	if (tile_last >= p.tiles_per_block) tile_last = p.tiles_per_block - 1;

	// ---- register prefetch of one tile: NV float4 (CF32) or uint4 (CU8) per thread, coalesced
	constexpr int NV = CU8 ? ((TILE_IN * 2) / (256 * 16) > 0 ? (TILE_IN * 2) / (256 * 16) : 1) : (TILE_IN * 8) / (256 * 16);
	uint4 pre[NV];
	auto prefetch = [&](int tile) {
		// tile -1 lives in the history buffer (last TILE_IN samples of the previous block)
		const unsigned char* base;
		if (tile < 0) base = (const unsigned char*)p.hist + (size_t)rx * TILE_IN * (CU8 ? 2 : 8);
		else base = (const unsigned char*)p.in + ((size_t)rx * p.in_stride + (size_t)tile * TILE_IN) * (CU8 ? 2 : 8);
		const uint4* src = (const uint4*)base;
#pragma unroll
		for (int e = 0; e < NV; e++) {
			if (CU8 && (TILE_IN * 2) < 256 * 16) { // tiny CU8 tiles (K small): not every thread loads
				pre[e] = (t * 16 < TILE_IN * 2) ? src[t] : make_uint4(0, 0, 0, 0);
			} else pre[e] = src[e * 256 + t];
		}
	};
	// first-stage buffer the tile enters at (K = 4: x0, 3: x1, 2: x2, 1: x3)
	auto store_sample_pair = [&](int s, float4 v) { // s even sample index inside the tile
		if (K == 4) *reinterpret_cast<float4*>(&sm.x0[(s >> 4) * R0 + (s & 15)]) = v;
		else if (K == 3) *reinterpret_cast<float4*>(&sm.x1[(s >> 3) * R1 + (s & 7)]) = v;
		else if (K == 2) *reinterpret_cast<float4*>(&x2[(s >> 2) * R2 + (s & 3)]) = v;
		else *reinterpret_cast<float4*>(&sm.x3[8 + s]) = v;
	};
	auto stage_in = [&]() {
		if (!CU8) {
#pragma unroll
			for (int e = 0; e < NV; e++) {
				int s = (e * 256 + t) * 2;
				store_sample_pair(s, *reinterpret_cast<float4*>(&pre[e]));
			}
		} else {
#pragma unroll
			for (int e = 0; e < NV; e++) {
				int s = (e * 256 + t) * 8; // 16 bytes = 8 CU8 samples
				if ((TILE_IN * 2) < 256 * 16 && s >= TILE_IN) continue;
				const unsigned w[4] = { pre[e].x, pre[e].y, pre[e].z, pre[e].w };
#pragma unroll
				for (int q = 0; q < 4; q++) { // Utilities/Convert.cpp:255-264: ((int)u - 128) / 128.0f (exact)
					float4 v;
					v.x = (float)((int)(w[q] & 255u) - 128) * 0.0078125f;
					v.y = (float)((int)((w[q] >> 8) & 255u) - 128) * 0.0078125f;
					v.z = (float)((int)((w[q] >> 16) & 255u) - 128) * 0.0078125f;
					v.w = (float)((int)(w[q] >> 24) - 128) * 0.0078125f;
					store_sample_pair(s + 2 * q, v);
				}
			}
		}
	};

	prefetch(tile_first);

	for (int tile = tile_first; tile <= tile_last; tile++) {
		__syncthreads(); // previous tile fully consumed (x0/x2 alias, x6 reads, tail copies)
		stage_in();
		__syncthreads();
		if (tile < tile_last) prefetch(tile + 1); // in flight during the whole ladder

		// ---- stage 1: x0 (4096) -> x1 (2048); thread owns x0[16t..16t+15], needs x0[16t-5..16t+14]
		if (K >= 4) {
			float2 v[20];
			const float4* own = reinterpret_cast<const float4*>(&sm.x0[t * R0]);
			const float4* halo = (t == 0) ? reinterpret_cast<const float4*>(&sm.h0[2])
			                              : reinterpret_cast<const float4*>(&sm.x0[(t - 1) * R0 + 10]);
			float4 hv[3], ov[8];
#pragma unroll
			for (int e = 0; e < 3; e++) hv[e] = halo[e];
#pragma unroll
			for (int e = 0; e < 8; e++) ov[e] = own[e];
			// v[i] = x0[16t-5+i]
			v[0] = make_float2(hv[0].z, hv[0].w);
			v[1] = make_float2(hv[1].x, hv[1].y); v[2] = make_float2(hv[1].z, hv[1].w);
			v[3] = make_float2(hv[2].x, hv[2].y); v[4] = make_float2(hv[2].z, hv[2].w);
#pragma unroll
			for (int e = 0; e < 7; e++) { v[5 + 2 * e] = make_float2(ov[e].x, ov[e].y); v[6 + 2 * e] = make_float2(ov[e].z, ov[e].w); }
			v[19] = make_float2(ov[7].x, ov[7].y);
			float2 o[8];
			cic5_dec_chunk<8>(v, o);
			float4* dst = reinterpret_cast<float4*>(&sm.x1[t * R1]);
#pragma unroll
			for (int e = 0; e < 4; e++) dst[e] = make_float4(o[2 * e].x, o[2 * e].y, o[2 * e + 1].x, o[2 * e + 1].y);
			__syncthreads();
			if (t == 255) { // tail of x0 for the next tile (x0 body is dead from here on)
#pragma unroll
				for (int e = 0; e < 4; e++) reinterpret_cast<float4*>(sm.h0)[e] = ov[4 + e];
			}
		}
		// ---- stage 2: x1 (2048) -> x2 (1024); thread owns x1[8t..8t+7], needs x1[8t-5..8t+6]
		if (K >= 3) {
			float2 v[12];
			const float4* own = reinterpret_cast<const float4*>(&sm.x1[t * R1]);
			const float4* halo = (t == 0) ? reinterpret_cast<const float4*>(&sm.h1[2])
			                              : reinterpret_cast<const float4*>(&sm.x1[(t - 1) * R1 + 2]);
			float4 hv[3], ov[4];
#pragma unroll
			for (int e = 0; e < 3; e++) hv[e] = halo[e];
#pragma unroll
			for (int e = 0; e < 4; e++) ov[e] = own[e];
			v[0] = make_float2(hv[0].z, hv[0].w);
			v[1] = make_float2(hv[1].x, hv[1].y); v[2] = make_float2(hv[1].z, hv[1].w);
			v[3] = make_float2(hv[2].x, hv[2].y); v[4] = make_float2(hv[2].z, hv[2].w);
#pragma unroll
			for (int e = 0; e < 3; e++) { v[5 + 2 * e] = make_float2(ov[e].x, ov[e].y); v[6 + 2 * e] = make_float2(ov[e].z, ov[e].w); }
			v[11] = make_float2(ov[3].x, ov[3].y);
			float2 o[4];
			cic5_dec_chunk<4>(v, o);
			float4* dst = reinterpret_cast<float4*>(&x2[t * R2]);
			dst[0] = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
			dst[1] = make_float4(o[2].x, o[2].y, o[3].x, o[3].y);
			__syncthreads();
			if (t == 255) {
#pragma unroll
				for (int e = 0; e < 4; e++) reinterpret_cast<float4*>(sm.h1)[e] = ov[e];
			}
		}
		// ---- stage 3: x2 (1024) -> x3 (512); thread owns x2[4t..4t+3], needs x2[4t-5..4t+2]
		if (K >= 2) {
			float2 v[8];
			// x2[4t-6..4t-5], x2[4t-4..4t-1], own x2[4t..4t+3]
			const float4* a = (t >= 2) ? reinterpret_cast<const float4*>(&x2[(t - 2) * R2 + 2])
			                           : reinterpret_cast<const float4*>(&sm.h2[2 + 4 * t]);
			const float4* b = (t >= 1) ? reinterpret_cast<const float4*>(&x2[(t - 1) * R2])
			                           : reinterpret_cast<const float4*>(&sm.h2[4]);
			const float4* own = reinterpret_cast<const float4*>(&x2[t * R2]);
			float4 av = a[0], b0 = b[0], b1 = b[1], o0 = own[0], o1 = own[1];
			v[0] = make_float2(av.z, av.w);
			v[1] = make_float2(b0.x, b0.y); v[2] = make_float2(b0.z, b0.w);
			v[3] = make_float2(b1.x, b1.y); v[4] = make_float2(b1.z, b1.w);
			v[5] = make_float2(o0.x, o0.y); v[6] = make_float2(o0.z, o0.w);
			v[7] = make_float2(o1.x, o1.y);
			float2 o[2];
			cic5_dec_chunk<2>(v, o);
			*reinterpret_cast<float4*>(&sm.x3[8 + 2 * t]) = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
			__syncthreads();
			if (t == 255) { // x2[1016..1023]
				reinterpret_cast<float4*>(sm.h2)[0] = b0; reinterpret_cast<float4*>(sm.h2)[1] = b1;
				reinterpret_cast<float4*>(sm.h2)[2] = o0; reinterpret_cast<float4*>(sm.h2)[3] = o1;
			}
		}
		// ---- stage 4: x3 (512) -> x4 (256 @ 96 kHz); thread t: x3[2t-5..2t]
		{
			float2 v[6];
			const float4* src = reinterpret_cast<const float4*>(&sm.x3[8 + 2 * t - 6]);
			float4 c0 = src[0], c1 = src[1], c2 = src[2], c3 = src[3];
			v[0] = make_float2(c0.z, c0.w);
			v[1] = make_float2(c1.x, c1.y); v[2] = make_float2(c1.z, c1.w);
			v[3] = make_float2(c2.x, c2.y); v[4] = make_float2(c2.z, c2.w);
			v[5] = make_float2(c3.x, c3.y);
			float2 o[1];
			cic5_dec_chunk<1>(v, o);
			sm.x4[8 + t] = o[0];
		}
		__syncthreads();
		// ---- FDC (DSP.cpp:283-293) + Rotate (DSP.cpp:296-316) at 96 kHz
		{
			float2 xm2 = sm.x4[8 + t - 2], xm1 = sm.x4[8 + t - 1], x = sm.x4[8 + t];
			float2 y = x;
			if (p.has_fdc) {
				// alpha * (h1 + x) + h2 * beta, evaluated componentwise: add, mul, mul, add
				float2 s = cadd(xm2, x);
				y = make_float2(p.alpha * s.x + xm1.x * p.beta, p.alpha * s.y + xm1.y * p.beta);
			}
			float2 rot = p.rot[(size_t)(tile + 1) * 256 + t]; // table has 256 leading entries (previous block's tail)
			float RR = y.x * rot.x, II = y.y * rot.y, RI = y.x * rot.y, IR = y.y * rot.x;
			sm.x5[0][8 + t] = make_float2(RR - II, IR + RI); // up   -> channel A
			sm.x5[1][8 + t] = make_float2(RR + II, IR - RI); // down -> channel B
		}
		__syncthreads();
		// ---- DS2_a / DS2_b (96k -> 48k), waves 0-1: channel A, waves 2-3: channel B
		{
			const int ch = t >> 7, j = t & 127;
			float2 v[6];
			const float4* src = reinterpret_cast<const float4*>(&sm.x5[ch][8 + 2 * j - 6]);
			float4 c0 = src[0], c1 = src[1], c2 = src[2], c3 = src[3];
			v[0] = make_float2(c0.z, c0.w);
			v[1] = make_float2(c1.x, c1.y); v[2] = make_float2(c1.z, c1.w);
			v[3] = make_float2(c2.x, c2.y); v[4] = make_float2(c2.z, c2.w);
			v[5] = make_float2(c3.x, c3.y);
			float2 o[1];
			cic5_dec_chunk<1>(v, o);
			sm.x6[ch][8 + j] = o[0];
		}
		__syncthreads();
		// ---- FilterCIC5 (DSP.cpp:132-157): same binomial filter, no decimation -> 48 kHz output
		{
			const int ch = t >> 7, j = t & 127;
			float2 v[6];
#pragma unroll
			for (int e = 0; e < 6; e++) v[e] = sm.x6[ch][8 + j - 5 + e];
#pragma unroll
			for (int lvl = 0; lvl < 5; lvl++) {
#pragma unroll
				for (int i = 0; i < 5 - lvl; i++) v[i] = cadd(v[i + 1], v[i]);
			}
			if (tile >= 0 && tile > tile_first) {
				float2* dst = p.c48 + ((size_t)rx * 2 + ch) * p.c48_stride + (size_t)tile * 128 + j;
				*dst = make_float2(v[0].x * 0.03125f, v[0].y * 0.03125f);
			}
		}
		__syncthreads();
		// ---- carry the tails of the small in-buffer-history stages to their leading slots
		if (t < 8) sm.x3[t] = sm.x3[512 + t];
		else if (t < 16) sm.x4[t - 8] = sm.x4[256 + t - 8];
		else if (t < 24) sm.x5[0][t - 16] = sm.x5[0][256 + t - 16];
		else if (t < 32) sm.x5[1][t - 24] = sm.x5[1][256 + t - 24];
		else if (t < 40) sm.x6[0][t - 32] = sm.x6[0][128 + t - 32];
		else if (t < 48) sm.x6[1][t - 40] = sm.x6[1][128 + t - 40];
	}
}

// K1b: keep the last tile of the block as history for the next block's warm-up tile
__global__ void k1_tail(const unsigned char* in, long long in_stride_bytes, long long block_bytes,
                        unsigned char* hist, int tail_bytes) {
	const int rx = blockIdx.y;
	const uint4* src = (const uint4*)(in + (size_t)rx * in_stride_bytes + block_bytes - tail_bytes);
	uint4* dst = (uint4*)(hist + (size_t)rx * tail_bytes);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tail_bytes / 16; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------
// K2a: SquareFreqOffsetCorrection analysis (DSP/DSP.cpp:417-456 + 475-489, DSP/FFT.h:94-129).
// One wave per 512-sample window: x^2 scattered bit-reversed into LDS, radix-2 DIT with the
// reference's float twiddle table (every butterfly is the same three complex ops as the
// reference's; butterflies of one stage are independent, so lane-parallel evaluation is exact),
// |X| via the glibc-equivalent hypot, the *sequential* float prefix sum (one lane, 511 adds: a
// parallel scan would round differently), then the two first-maximum searches as wave reductions.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_argmax_first(float& v, int& i) {
	// larger value wins; ties -> lower index (the reference scans upward with a strict '>')
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		float ov = __shfl_xor(v, off);
		int oi = __shfl_xor(i, off);
		if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
	}
}

__global__ __launch_bounds__(64) void k2_cgf_analyse(K2Params p) {
	__shared__ __attribute__((aligned(16))) float2 X[512];
	__shared__ __attribute__((aligned(16))) float mag[520]; // mag[q] = |X[(q + 256) % 512]|, q in [0, 512]
	__shared__ __attribute__((aligned(16))) float cs[512];

	const int lane = threadIdx.x;
	const int w = blockIdx.x, chan = blockIdx.y; // chan = rx * 2 + ch
	const float2* x = p.c48 + (size_t)chan * p.c48_stride + (size_t)w * 512;

#pragma unroll
	for (int q = 0; q < 8; q++) {
		int n = lane + 64 * q;
		float2 v = x[n];
		// data[i] * data[i]: (a*a - b*b, a*b + b*a)
		X[__brev((unsigned)n) >> 23] = make_float2(v.x * v.x - v.y * v.y, v.x * v.y + v.y * v.x);
	}
	__syncthreads();
#pragma unroll 1
	for (int s = 0; s < 9; s++) {
		const int m2 = 1 << s;
#pragma unroll
		for (int q = 0; q < 4; q++) {
			int b = lane + 64 * q;
			int j = b & (m2 - 1);
			int i0 = ((b >> s) << (s + 1)) + j, i1 = i0 + m2;
			float2 o = p.omega[j << (8 - s)];
			float2 a = X[i0], c = X[i1];
			float2 tt = cmul(o, c);
			X[i1] = csub(a, tt);
			X[i0] = cadd(a, tt);
		}
		__syncthreads();
	}
#pragma unroll
	for (int q = 0; q < 8; q++) {
		int k = lane + 64 * q;
		float2 v = X[k];
		float m = hypot_ref(v.x, v.y);
		mag[(k + 256) & 511] = m;
		if (k == 256) mag[512] = m; // wrap slot: shifted index 512 == 0
	}
	__syncthreads();

	int wi = 0;
	if (p.wide) {
		if (lane == 0) { // cumsum[0] = 0; cumsum[i] = cumsum[i-1] + mag[i]  (DSP.cpp:431-436)
			float acc = 0.0f;
			cs[0] = 0.0f;
			for (int i = 1; i < 512; i += 1) {
				acc = acc + mag[i];
				cs[i] = acc;
			}
		}
		__syncthreads();
		// M = 133, ofs = 15, delta = 102: v = cs[i+M] - cs[i] + 0.6f * (mag[i+ofs] + mag[i+ofs+delta]), i < 379
		float best = -1.0f;
		int bi = 0;
#pragma unroll
		for (int q = 0; q < 6; q++) {
			int i = lane + 64 * q;
			if (i < 512 - 133) {
				float v = cs[i + 133] - cs[i] + 0.6f * (mag[i + 15] + mag[i + 117]);
				if (v > best) { best = v; bi = i; }
			}
		}
		wave_argmax_first(best, bi);
		wi = bi + 66 - 256; // wi + M/2 - N/2
	}
	// i in [wi+187, wi+223): h = mag[i] + mag[i+102] (shifted indices, wrap at 512); first strict max > 0
	float h = 0.0f;
	int hi = 0x7fffffff;
	if (lane < 36) {
		int i = wi + 187 + lane;
		float v = mag[(i + 512) & 511] + mag[(i + 102 + 512) & 511];
		if (v > 0.0f) { h = v; hi = i; }
	}
	wave_argmax_first(h, hi);
	if (lane == 0) {
		// fz = N/2 - (i + delta/2) = 205 - i (integer valued); default -1
		int fz = (h > 0.0f) ? (205 - hi) : -1;
		p.fz[(size_t)chan * p.n_windows + w] = fz;
		p.ppm[(size_t)chan * p.n_windows + w] = p.ppm_table[fz + 205];
	}
}

// ------------------------------------------------------------------------------------------
// K2b: CGF derotation (DSP/DSP.cpp:457-466).  rot *= rot_step; output[i] *= rot per sample, rot
// renormalised per window and carried across windows -> a strictly sequential float recurrence per
// (receiver, channel).  One wave per chain: all 64 lanes run the recurrence redundantly (no
// divergence, no LDS), a DPP wave shift hands the phasor of step k to lane 63-k, which applies it to
// its own sample, so global traffic stays fully coalesced (8 B / lane).
// ------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64) void k2_cgf_derotate(K2Params p) {
	const int lane = threadIdx.x;
	const int chan = blockIdx.x;
	const float2* x = p.c48 + (size_t)chan * p.c48_stride;
	float2* y = p.cgf + (size_t)chan * p.cgf_stride; // row has CGF_HIST leading history samples
	const int L = p.n_windows * 512;

	// carry the tail of the previous block (FIR-17 history + partial ScatterPLL group) to the front
	if (lane < CGF_HIST) {
		float2 tv = y[L + lane];
		y[lane] = tv;
	}
	const float2 r0 = p.rot_state[chan];
	// `cur` is at the same time the recurrence variable (lane 0) and a 64-deep history (lane l holds the
	// phasor of l steps ago): each step computes rot*rot_step in every lane -- only lane 0's result is
	// meaningful -- and a DPP wave_shr:1 move then refills lanes 1..63 from the previous register while
	// lane 0 (no source lane) keeps the new phasor.  5 VALU ops per sample, no LDS, no select.
	// The product is 3 packed ops: P = (rx*sx, rx*sy), Q = (ry*-sy, ry*sx), rot' = P + Q; x*-y == -(x*y)
	// exactly, so this is the reference's (ac - bd, ad + bc) bit for bit (DSP/DSP.cpp:460-463).
	v2f cur = { r0.x, r0.y };
	const int rl = 63 - lane; // after 64 steps lane l holds the phasor of step 63-l of the chunk
	float2 xv = x[rl];        // sample of the first 64-chunk, prefetched
	for (int w = 0; w < p.n_windows; w++) {
		const int fz = p.fz[(size_t)chan * p.n_windows + w];
		const float2 stp = p.step_table[fz + 205];
		const v2f st = { stp.x, stp.y }, st_sw = { -stp.y, stp.x };
#pragma unroll 1
		for (int c = 0; c < 8; c++) {
			const int base = w * 512 + c * 64;
			float2 xn = xv;
			if (base + 64 < L) xn = x[base + 64 + rl]; // next chunk in flight during the 64 serial steps
#pragma unroll
			for (int k = 0; k < 64; k++) {
				const v2f P = cur.xx * st;
				const v2f Q = cur.yy * st_sw;
				v2f nw = P + Q;
				nw.x = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(nw.x), __float_as_int(cur.x), 0x138, 0xF, 0xF, false));
				nw.y = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(nw.y), __float_as_int(cur.y), 0x138, 0xF, 0xF, false));
				cur = nw;
			}
			y[CGF_HIST + base + rl] = cmul(xv, make_float2(cur.x, cur.y)); // output[i] *= rot
			xv = xn;
		}
		// rot /= std::abs(rot) once per window (DSP.cpp:465); only lane 0 carries the chain
		const float a = hypot_ref(cur.x, cur.y);
		const float nx = __fdiv_rn(cur.x, a), ny = __fdiv_rn(cur.y, a);
		cur.x = lane == 0 ? nx : cur.x;
		cur.y = lane == 0 ? ny : cur.y;
	}
	if (lane == 0) p.rot_state[chan] = make_float2(cur.x, cur.y);
}

// ------------------------------------------------------------------------------------------
// K3: FilterComplex(Filters::Coherent) + ScatterPLL (DSP/DSP.cpp:215-246, DSP/DSP.h:95-117).
// One thread per complete 5-sample group: five 17-tap dot products accumulated left-to-right
// from 0 (DSP.h:224-230), the group level ((((0+n0)+n1)+n2)+n3)+n4)/5, and the de-interleave
// into five per-phase sample streams for the PhaseSearchEMA chains.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k3_fir_scatter(K3Params p) {
	const int chan = blockIdx.y;
	const int g = blockIdx.x * blockDim.x + threadIdx.x; // group index inside the block
	if (g >= p.n_groups) return;
	// stream index of the group's first sample relative to the block start (may be -4..0 for g = 0)
	const long long n_rel = (p.first_group + g) * 5 - p.first_sample48;
	const float2* x = p.cgf + (size_t)chan * p.cgf_stride + CGF_HIST + n_rel - 16;
	float2 win[21];
#pragma unroll
	for (int i = 0; i < 21; i++) win[i] = x[i];
	float level = 0.0f;
#pragma unroll
	for (int j = 0; j < 5; j++) {
		float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
		for (int i = 0; i < 17; i++) {
			float tp = p.taps[i];
			acc = make_float2(acc.x + tp * win[j + i].x, acc.y + tp * win[j + i].y);
		}
		level = level + (acc.x * acc.x + acc.y * acc.y); // std::norm
		p.sym[((size_t)chan * 5 + j) * p.sym_stride + g] = acc;
		if (p.fir_tap) p.fir_tap[(size_t)chan * p.fir_tap_stride + (n_rel + j + 4)] = acc;
	}
	p.lvl[(size_t)chan * p.sym_stride + g] = __fdiv_rn(level, 5.0f);
}

// ------------------------------------------------------------------------------------------
// K4: PhaseSearchEMA (DSP/Demod.cpp:39-101).  One 16-lane row per (receiver, channel, phase)
// chain, one lane per phase hypothesis: lane k keeps ma[k] and bits[k]; the +-1 neighbourhood
// argmax becomes two per-lane predicates gathered with wave ballots, and every lane of the row
// tracks max_idx redundantly from the ballot words (pure integer recurrence).
// ------------------------------------------------------------------------------------------
__constant__ float2 c_ps_phase[8] = { // DSP/Demod.h:29-31
	{ 9.9518472640441780e-01f, 9.8017143048367339e-02f }, { 9.5694033335306883e-01f, 2.9028468509743588e-01f },
	{ 8.8192125790916542e-01f, 4.7139674887287397e-01f }, { 7.7301044123076901e-01f, 6.3439329894649099e-01f },
	{ 6.3439326515712957e-01f, 7.7301046896098113e-01f }, { 4.7139671032286945e-01f, 8.8192127851457169e-01f },
	{ 2.9028464326824349e-01f, 9.5694034604181499e-01f }, { 9.8017099547459546e-02f, 9.9518473068888236e-01f } };

// neighbour exchange inside a 16-lane row: DPP row rotate (a VALU modifier, no LDS round trip).
// The rotate direction is probed once per wave (mode 0: row_ror:1 delivers lane k-1; mode 1: lane k+1;
// mode 2: unexpected -> fall back to ds_bpermute), so correctness never rests on the ISA manual's wording.
__device__ __forceinline__ float dpp_ror1(float v) {
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_ror15(float v) {
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x12F, 0xF, 0xF, false));
}

struct PsLane { // per-lane PhaseSearchEMA state: hypothesis k of one chain
	float ma;
	unsigned bits;
};

// One symbol for all 16 hypotheses of a row (DSP/Demod.cpp:39-101).  Returns the emitted bit (0/1).
template <int MODE>
__device__ __forceinline__ unsigned ps_step(float2 v, int rot, float pc, float psn, PsLane& s, int& idx, int k, int rowbase) {
	const float w = 0.85f;
	const float w1 = 1 - w; // (1 - weight) evaluated in float (Demod.cpp:71)
	// multiply by (1j)^rot via swaps/negations (Demod.cpp:44-61); branch-free because rot differs per row
	const bool sw = (rot & 1) != 0;
	float re = sw ? v.y : v.x, im = sw ? v.x : v.y;
	re = (rot == 1 || rot == 2) ? -re : re;
	im = (rot >= 2) ? -im : im;
	const float a = re * pc, b = im * psn;
	const float tt = a + b;
	s.bits = ((s.bits << 1) | (tt > 0 ? 1u : 0u)) & 0xFFu; // uint8_t shift register
	s.ma = w * s.ma + w1 * fabsf(tt);
	float left, right;
	if (MODE == 2) {
		left = __shfl(s.ma, (k + 15) & 15, 16);
		right = __shfl(s.ma, (k + 1) & 15, 16);
	} else {
		const float r1 = dpp_ror1(s.ma), r15 = dpp_ror15(s.ma);
		left = MODE == 0 ? r1 : r15;
		right = MODE == 0 ? r15 : r1;
	}
	const bool p0 = s.ma > left;         // centre beats idx-1
	const float bestc = p0 ? s.ma : left;
	const bool p1 = right > bestc;       // idx+1 beats the better of the two
	// nDelay = 3 (Model.cpp:560-561): bit(nDelay) XOR bit(nDelay + 1) of the winning hypothesis
	const bool xb = (((s.bits >> 4) ^ (s.bits >> 3)) & 1u) != 0;
	const unsigned long long B0 = __ballot(p0), B1 = __ballot(p1), BX = __ballot(xb);
	const unsigned m0 = (unsigned)(B0 >> rowbase), m1 = (unsigned)(B1 >> rowbase), mx = (unsigned)(BX >> rowbase);
	const int q0 = (int)((m0 >> idx) & 1u), q1 = (int)((m1 >> idx) & 1u);
	idx = (idx + (q1 ? 1 : q0 - 1)) & 15; // prev-1, prev, prev+1 with first-maximum preference
	return (mx >> idx) & 1u;
}

constexpr int PS_BATCH = 16; // symbols whose samples are fetched together (multiple of 4: rot phase is preserved)

template <int MODE>
__device__ __forceinline__ void ps_chain(const float2* __restrict__ x, uint32_t* __restrict__ out, int n, bool writer, float pc,
                                         float psn, PsLane& s, int& idx, int& rot, int k, int rowbase) {
	const int nb = n - (n % PS_BATCH);
	uint32_t word = 0;
	float2 cur[PS_BATCH];
	if (nb > 0) {
#pragma unroll
		for (int e = 0; e < PS_BATCH; e++) cur[e] = x[e];
	}
#pragma unroll 1
	for (int g0 = 0; g0 < nb; g0 += PS_BATCH) {
		float2 nxt[PS_BATCH];
		const bool more = g0 + PS_BATCH < nb;
		if (more) { // next batch in flight while this one is processed
#pragma unroll
			for (int e = 0; e < PS_BATCH; e++) nxt[e] = x[g0 + PS_BATCH + e];
		}
		uint32_t part = 0;
#pragma unroll
		for (int e = 0; e < PS_BATCH; e++) part |= ps_step<MODE>(cur[e], (rot + e) & 3, pc, psn, s, idx, k, rowbase) << e;
		word |= part << (g0 & 31);
		if (((g0 + PS_BATCH) & 31) == 0) {
			if (writer) out[g0 >> 5] = word;
			word = 0;
		}
		if (more) {
#pragma unroll
			for (int e = 0; e < PS_BATCH; e++) cur[e] = nxt[e];
		}
	}
	for (int g = nb; g < n; g++) {
		word |= ps_step<MODE>(x[g], rot, pc, psn, s, idx, k, rowbase) << (g & 31);
		rot = (rot + 1) & 3;
		if ((g & 31) == 31) {
			if (writer) out[g >> 5] = word;
			word = 0;
		}
	}
	if ((n & 31) != 0 && writer) out[n >> 5] = word;
}

__global__ __launch_bounds__(64) void k4_phase_search(K4Params p) {
	const int lane = threadIdx.x;
	const int k = lane & 15, row = lane >> 4;
	const int chain = blockIdx.x * 4 + row; // (rx*2 + ch) * 5 + j
	const bool live = chain < p.n_chains;
	const int cidx = live ? chain : p.n_chains - 1;
	const int rowbase = row * 16;

	const int src = __builtin_amdgcn_update_dpp(0, k, 0x121, 0xF, 0xF, false);
	const bool all_left = __all(src == ((k + 15) & 15)), all_right = __all(src == ((k + 1) & 15));

	const int jj = k < 8 ? k : 15 - k;
	const float pc = c_ps_phase[jj].x;
	const float psn = k < 8 ? c_ps_phase[jj].y : -c_ps_phase[jj].y; // a - b == a + (im * -s) exactly

	EmaState* st = p.state + cidx;
	PsLane s;
	s.ma = st->ma[k];
	s.bits = st->bits[k];
	int idx = st->max_idx, rot = st->rot;

	const float2* x = p.sym + (size_t)cidx * p.sym_stride;
	uint32_t* out = p.bits + (size_t)cidx * p.bits_stride;
	const bool writer = live && k == 0;
	if (all_left) ps_chain<0>(x, out, p.n_groups, writer, pc, psn, s, idx, rot, k, rowbase);
	else if (all_right) ps_chain<1>(x, out, p.n_groups, writer, pc, psn, s, idx, rot, k, rowbase);
	else ps_chain<2>(x, out, p.n_groups, writer, pc, psn, s, idx, rot, k, rowbase);
	if (live) {
		st->ma[k] = s.ma;
		st->bits[k] = s.bits;
		if (k == 0) { st->max_idx = idx; st->rot = rot; }
	}
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
template <int K, bool CU8>
static hipError_t launch_k1_t(const K1Params& p, int spans, int n_rx, hipStream_t s) {
	static bool attr_set = false;
	if (!attr_set) {
		hipError_t e = hipFuncSetAttribute((const void*)k1_frontend<K, CU8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(K1Smem));
		if (e != hipSuccess) return e;
		attr_set = true;
	}
	hipLaunchKernelGGL((k1_frontend<K, CU8>), dim3(spans, n_rx), dim3(K1_THREADS), sizeof(K1Smem), s, p);
	return hipGetLastError();
}

hipError_t launch_k1(const K1Params& p, int K, bool cu8, int spans, int n_rx, hipStream_t s) {
	switch (K * 2 + (cu8 ? 1 : 0)) {
	case 8: return launch_k1_t<4, false>(p, spans, n_rx, s);
	case 9: return launch_k1_t<4, true>(p, spans, n_rx, s);
	case 6: return launch_k1_t<3, false>(p, spans, n_rx, s);
	case 7: return launch_k1_t<3, true>(p, spans, n_rx, s);
	case 4: return launch_k1_t<2, false>(p, spans, n_rx, s);
	case 5: return launch_k1_t<2, true>(p, spans, n_rx, s);
	case 2: return launch_k1_t<1, false>(p, spans, n_rx, s);
	case 3: return launch_k1_t<1, true>(p, spans, n_rx, s);
	}
	return hipErrorInvalidValue;
}

hipError_t launch_k1_tail(const void* in, long long in_stride_bytes, long long block_bytes, void* hist, int tail_bytes,
                          int n_rx, hipStream_t s) {
	int blocks = (tail_bytes / 16 + 255) / 256;
	if (blocks > 8) blocks = 8;
	hipLaunchKernelGGL(k1_tail, dim3(blocks, n_rx), dim3(256), 0, s, (const unsigned char*)in, in_stride_bytes, block_bytes,
	                   (unsigned char*)hist, tail_bytes);
	return hipGetLastError();
}

hipError_t launch_k2(const K2Params& p, int n_chan, hipStream_t s) {
	hipLaunchKernelGGL(k2_cgf_analyse, dim3(p.n_windows, n_chan), dim3(64), 0, s, p);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(k2_cgf_derotate, dim3(n_chan), dim3(64), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k3(const K3Params& p, int n_chan, hipStream_t s) {
	if (p.n_groups <= 0) return hipSuccess;
	hipLaunchKernelGGL(k3_fir_scatter, dim3((p.n_groups + 255) / 256, n_chan), dim3(256), 0, s, p);
	return hipGetLastError();
}

hipError_t launch_k4(const K4Params& p, hipStream_t s) {
	if (p.n_groups <= 0) return hipSuccess;
	hipLaunchKernelGGL(k4_phase_search, dim3((p.n_chains + 3) / 4), dim3(64), 0, s, p);
	return hipGetLastError();
}

} // namespace aisk
