// ais-catcher_amd/csrc/aisgpu.cpp -- host side of libaisgpu.so: context, tables, buffers, launch order.
//
// Everything data-independent that the reference computes with libm or as a sequential float
// recurrence is produced HERE on the host, with the host's own libm, exactly as the reference
// would on the same machine: FFT twiddles (DSP/FFT.h:83), the Rotate phasor
// sequence incl. its once-per-Receive renormalisation (DSP/DSP.cpp:309,315), and the finite set
// of CGF rot_step phasors (DSP/DSP.cpp:457-458).  The device never calls sin/cos.
// Compiled with -ffp-contract=off (host and device).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/aisgpu.h"
#include "kernels.h"

using namespace aisk;

namespace {

const float PI_F = 3.14159265358979323846f; // Library/Common.h:318: PI is a float constant
const float TAPS_COHERENT[17] = { // DSP/Filters.h:35-41
	2.06995719e-06f, 3.18610148e-05f, 3.40605309e-04f, 2.52892989e-03f, 1.30411453e-02f, 4.67076746e-02f,
	1.16186141e-01f, 2.00730781e-01f, 2.40861391e-01f, 2.00730781e-01f, 1.16186141e-01f, 4.67076746e-02f,
	1.30411453e-02f, 2.52892989e-03f, 3.40605309e-04f, 3.18610148e-05f, 2.06995719e-06f };

struct EvPair { hipEvent_t a, b; };
constexpr int NBUF = 3; // ring depth of the buffers that cross from the front-end stream to the others

} // namespace

struct aisgpu {
	aisgpu_cfg cfg;
	int K = 0;            // CIC5 stages in front of the 96 kHz point
	int tile96 = 64;      // 96 kHz samples per front-end tile
	int depth = 1;        // tiles prefetched ahead by the front end
	int k1_threads = 64;  // front-end workgroup size (64: one autonomous wave per workgroup)
	int tile_in = 0;      // input samples per front-end tile (tile96 << K)
	int in_bytes = 0;     // bytes per input sample
	int n96 = 0, L = 0, W = 0; // per block: 96 kHz samples, 48 kHz samples per channel, CGF windows
	int Gcap = 0, words = 0;   // group capacity per block, bit words per chain
	int n_chan = 0, n_chains = 0;
	int tiles_per_block = 0, tiles_per_span = 0, spans = 0;
	float alpha = 0, beta = 1; int has_fdc = 0;

	// Three streams software-pipeline consecutive blocks:
	//   s0 (stream): Rotate table upload -> K1 -> K1-tail -> K2a   (bandwidth-bound front end)
	//   s3: K2b                                         (sequential CGF phasor recurrence, 8 waves, latency bound)
	//   s1 (= s2): K2c -> K3 -> K4 (+ D2H of the outputs) (apply phasors, FIR/ScatterPLL, PhaseSearchEMA)
	// so block b+1's front end overlaps block b's phasor recurrence and back end.  Buffers that cross a
	// stream boundary are double buffered by block parity.
	hipStream_t stream = nullptr, s1 = nullptr, s2 = nullptr, s3 = nullptr; // s3: the CGF phasor recurrence alone
	hipEvent_t ev_phasor[NBUF] = {}; // s3: phasor(b) done -> s1 may apply it
	bool serial = false;
	hipEvent_t ev_front[NBUF] = {}; // s0: K2a(b) done           -> s3 may start K2b(b)
	hipEvent_t ev_c48free[NBUF] = {}; // s1: K2c(b) done (c48/fz/rotT[q] consumed) -> s0 may run K1(b+NBUF)
	hipEvent_t ev_mid[2] = { nullptr, nullptr };   // s1: K3(b) done            -> s2 may start K4(b)
	hipEvent_t ev_ema[2] = { nullptr, nullptr };   // s2: K4(b) done (sym/lvl[p] consumed) -> s1 may run K3(b+2)
	// device buffers
	void* d_in = nullptr; void* d_hist = nullptr;
	float2* d_rot[2] = { nullptr, nullptr };
	float2 *d_c48[NBUF] = {}, *d_sym[2] = { nullptr, nullptr };
	float2 *d_rotT[NBUF] = {};
	float2 *d_cgf = nullptr, *d_omega = nullptr, *d_step = nullptr, *d_rotstate = nullptr, *d_firtap = nullptr;
	float *d_ppmtab = nullptr, *d_ppm[NBUF] = {}, *d_lvl[2] = { nullptr, nullptr };
	int* d_fz[NBUF] = {};
	uint32_t* d_bits[2] = { nullptr, nullptr };
	EmaState* d_ema[2] = { nullptr, nullptr }; // state before / after the current block (swapped per block)
	uint32_t* d_pswords = nullptr; float *d_psma0 = nullptr, *d_psma1 = nullptr; unsigned* d_psfin = nullptr; int* d_psflag = nullptr;
	int ps_chunks = 1, ps_warm = 256; bool ps_parallel = true;
	// host (pinned)
	void* h_in = nullptr;
	float2* h_rot[2] = { nullptr, nullptr };
	hipEvent_t rot_ev[2] = { nullptr, nullptr };
	uint32_t* h_bits = nullptr; float* h_lvl = nullptr; float* h_ppm = nullptr;
	// stream state
	long long block_idx = 0;     // blocks run so far
	long long n48 = 0;           // 48 kHz samples consumed before the current block
	float2 rot = { 1.0f, 0.0f }; // Rotate::rot carried across blocks
	float2 mult = { 1.0f, 0.0f };
	std::vector<float2> rot_tail; // last ROT_HIST phasors of the previous block
	const void* cur_in = nullptr; long long cur_in_stride = 0;
	bool submitted = false, have_out = false;
	// last block's output geometry
	int out_groups = 0; long long out_first_group = 0, out_first48 = 0;
	// timing
	bool timing = false;
	std::vector<EvPair> ev_busy, ev_free;
	double k1_ms = 0; int k1_launches = 0;
	std::string err;
};

namespace {

int fail(aisgpu_t* h, int code, const char* what, hipError_t e) {
	if (h) {
		char b[256];
		snprintf(b, sizeof b, "%s: %s", what, hipGetErrorString(e));
		h->err = b;
	}
	return code;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(h, AISGPU_ERR_HIP, #call, e_); } while (0)

template <typename T>
hipError_t dalloc(T** p, size_t n) {
	hipError_t e = hipMalloc((void**)p, n * sizeof(T));
	if (e == hipSuccess) e = hipMemset(*p, 0, n * sizeof(T));
	return e;
}

// Rotate phasor table of one Receive() call (DSP/DSP.cpp:296-316): entry i multiplies sample i,
// then rot *= mult; after the call rot /= |rot|.  tab = [ROT_HIST previous tail][n96 new].
void gen_rot_table(aisgpu_t* h, float2* tab) {
	for (int i = 0; i < ROT_HIST; i++) tab[i] = h->rot_tail[i];
	float2 r = h->rot;
	const float2 m = h->mult;
	float2* t = tab + ROT_HIST;
	for (int i = 0; i < h->n96; i++) {
		t[i] = r;
		float re = r.x * m.x - r.y * m.y;
		float im = r.x * m.y + r.y * m.x;
		r.x = re; r.y = im;
	}
	float a = hypotf(r.x, r.y);
	r.x /= a; r.y /= a;
	h->rot = r;
	for (int i = 0; i < ROT_HIST; i++) h->rot_tail[i] = t[h->n96 - ROT_HIST + i];
}

void drain_events(aisgpu_t* h) {
	for (auto& p : h->ev_busy) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { h->k1_ms += ms; h->k1_launches++; }
		h->ev_free.push_back(p);
	}
	h->ev_busy.clear();
}

} // namespace

extern "C" {

const char* aisgpu_strerror(int code) {
	switch (code) {
	case AISGPU_OK: return "ok";
	case AISGPU_ERR_ARG: return "invalid argument or unsupported configuration";
	case AISGPU_ERR_NODEV: return "no usable HIP device";
	case AISGPU_ERR_HIP: return "HIP runtime error";
	case AISGPU_ERR_STATE: return "call sequence error";
	case AISGPU_ERR_OVERFLOW: return "output buffer too small";
	}
	return "unknown error";
}
const char* aisgpu_last_error(aisgpu_t* h) { return h ? h->err.c_str() : ""; }

int aisgpu_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

void aisgpu_default_cfg(aisgpu_cfg* c) {
	memset(c, 0, sizeof *c);
	c->sample_rate = 1536000;
	c->n_receivers = 1;
	c->block_len = 786432; // the reference file reader's CF32 block (Device/FileRAW.h:43)
	c->model = AISGPU_MODEL_DEFAULT;
	c->input_format = AISGPU_FMT_CF32;
	c->afc_wide = 1;
	c->droop = 1;
}

int aisgpu_create(const aisgpu_cfg* cfg, aisgpu_t** out) {
	if (!cfg || !out) return AISGPU_ERR_ARG;
	*out = nullptr;
	int K = -1;
	float alpha = 0;
	// pure 2^k ladders of ModelFrontend::buildModel (DSP/Model.cpp:157-338) that fit the LDS tile
	switch (cfg->sample_rate) {
	case 192000: K = 1; alpha = -0.8f; break;
	case 384000: K = 2; alpha = -1.1f; break;
	case 768000: K = 3; alpha = -1.2f; break;
	case 1536000: K = 4; alpha = -1.2f; break;
	default: return AISGPU_ERR_ARG;
	}
	const int dec = 2 << K; // input samples per 48 kHz sample
	if (cfg->model != AISGPU_MODEL_DEFAULT) return AISGPU_ERR_ARG;
	if (cfg->input_format != AISGPU_FMT_CU8 && cfg->input_format != AISGPU_FMT_CF32) return AISGPU_ERR_ARG;
	if (cfg->n_receivers < 1 || cfg->n_receivers > 65535) return AISGPU_ERR_ARG;
	if (cfg->block_len < 512 * dec || cfg->block_len % (512 * dec) != 0) return AISGPU_ERR_ARG;
	if (aisgpu_device_count() <= cfg->device_id || cfg->device_id < 0) return AISGPU_ERR_NODEV;

	aisgpu_t* h = new (std::nothrow) aisgpu();
	if (!h) return AISGPU_ERR_ARG;
	h->cfg = *cfg;
	h->K = K;
	// front-end geometry (tuning knobs; the defaults are the measured best): workgroup size, 96 kHz samples per
	// tile, prefetch depth.  Valid combinations: 256 threads x {256,128}; 64 threads x {64,32}.
	h->k1_threads = 64; h->tile96 = 64; h->depth = 1; // measured best: autonomous waves (profiles/r01_k1_geometry_sweep.txt)
	if (const char* e = getenv("AISGPU_K1")) { // "threads,tile96,depth"
		int a = 0, b = 0, d = 0;
		if (sscanf(e, "%d,%d,%d", &a, &b, &d) == 3) {
			const bool ok = (a == 256 && (b == 256 || b == 128) && (d == 1 || d == 2 || (b == 128 && d == 3)) && !(b == 128 && d == 1)) ||
			                (a == 64 && (b == 64 || b == 32) && (d == 1 || d == 2));
			if (ok) { h->k1_threads = a; h->tile96 = b; h->depth = d; }
		}
	}
	h->tile_in = h->tile96 << K;
	h->in_bytes = cfg->input_format == AISGPU_FMT_CU8 ? 2 : 8;
	h->n96 = cfg->block_len >> K;
	h->L = h->n96 / 2;
	h->W = h->L / 512;
	h->Gcap = ((h->L + 4) / 5 + 1 + 31) / 32 * 32;
	h->words = h->Gcap / 32;
	h->n_chan = cfg->n_receivers * 2;
	h->n_chains = h->n_chan * 5;
	h->tiles_per_block = cfg->block_len / h->tile_in;
	h->has_fdc = cfg->droop ? 1 : 0;
	h->alpha = alpha;
	h->beta = 1 - 2 * alpha; // DSP/DSP.h:296, evaluated in float
	// span length: enough workgroups to fill 256 CUs twice over, at most 1/8 warm-up overhead
	int tps = cfg->tiles_per_span;
	if (tps <= 0) {
		tps = h->tiles_per_block;
		const long long want = h->k1_threads == 64 ? 8192 : (h->tile96 >= 256 ? 1024 : 2048); // workgroups: a few per CU per residency slot
		while (tps > 8 && (long long)cfg->n_receivers * ((h->tiles_per_block + tps - 1) / tps) < want) tps = (tps + 1) / 2;
	}
	if (tps > h->tiles_per_block) tps = h->tiles_per_block;
	h->tiles_per_span = tps;
	h->spans = (h->tiles_per_block + tps - 1) / tps;
	*out = h; // from here on the caller destroys it on failure

	HIPCHK(hipSetDevice(cfg->device_id));
	HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
	if ((cfg->flags & AISGPU_FLAG_SERIAL) || getenv("AISGPU_SERIAL")) { // profiling aid: no cross-block overlap, every kernel runs alone
		h->s1 = h->s2 = h->s3 = h->stream;
		h->serial = true;
	} else {
		// HIP maps streams onto a small number of hardware queues (4 by default, one is the application's
		// null stream): two of our streams sharing a queue would serialise.  So: three streams.
		HIPCHK(hipStreamCreateWithFlags(&h->s1, hipStreamNonBlocking));
		HIPCHK(hipStreamCreateWithFlags(&h->s3, hipStreamNonBlocking));
		h->s2 = h->s1; // apply + FIR + PhaseSearchEMA of a block run back to back on one stream
	}
	for (int i = 0; i < NBUF; i++) {
		HIPCHK(hipEventCreateWithFlags(&h->ev_front[i], hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&h->ev_phasor[i], hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&h->ev_c48free[i], hipEventDisableTiming));
	}
	for (int i = 0; i < 2; i++) {
		HIPCHK(hipEventCreateWithFlags(&h->ev_mid[i], hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&h->ev_ema[i], hipEventDisableTiming));
	}

	// ---- constant tables (host libm, like the reference on this machine)
	{
		float angle = (float)((double)PI_F * 25000.0 / 48000.0); // Model.cpp:31
		h->mult = make_float2(cosf(angle), sinf(angle));         // std::polar(1.0f, angle), DSP.h:311
		h->rot_tail.assign(ROT_HIST, make_float2(1.0f, 0.0f));
		std::vector<float2> omega(512), step(FZ_COUNT);
		std::vector<float> ppm(FZ_COUNT);
		for (int s = 0; s < 512; s++) { // FFT.h:83
			float th = ((float)(-2.0 * (double)PI_F) * (float)s) / (float)512;
			omega[s] = make_float2(cosf(th), sinf(th));
		}
		for (int i = 0; i < FZ_COUNT; i++) { // DSP.cpp:453-458,466
			float fz = (float)(FZ_MIN + i);
			float f = fz / 2.0f / 512;
			float ang = (float)(f * 2 * PI_F);
			step[i] = make_float2(cosf(ang), sinf(ang));
			ppm[i] = f * 48000.0f / 162.0f;
		}
		HIPCHK(dalloc(&h->d_omega, 512));
		HIPCHK(dalloc(&h->d_step, FZ_COUNT));
		HIPCHK(dalloc(&h->d_ppmtab, FZ_COUNT));
		HIPCHK(hipMemcpy(h->d_omega, omega.data(), 512 * sizeof(float2), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(h->d_step, step.data(), FZ_COUNT * sizeof(float2), hipMemcpyHostToDevice));
		HIPCHK(hipMemcpy(h->d_ppmtab, ppm.data(), FZ_COUNT * sizeof(float), hipMemcpyHostToDevice));
	}
	const size_t R = cfg->n_receivers, C = h->n_chan;
	HIPCHK(dalloc((unsigned char**)&h->d_hist, R * h->tile_in * h->in_bytes));
	// zero signal before the stream starts: CU8 zero is the byte 128 (Utilities/Convert.cpp:255-264)
	if (cfg->input_format == AISGPU_FMT_CU8) HIPCHK(hipMemset(h->d_hist, 0x80, R * h->tile_in * h->in_bytes));
	for (int i = 0; i < 2; i++) {
		HIPCHK(dalloc(&h->d_rot[i], (size_t)ROT_HIST + h->n96));
		HIPCHK(hipHostMalloc((void**)&h->h_rot[i], ((size_t)ROT_HIST + h->n96) * sizeof(float2), hipHostMallocDefault));
		HIPCHK(hipEventCreateWithFlags(&h->rot_ev[i], hipEventDisableTiming));
	}
	for (int i = 0; i < NBUF; i++) {
		HIPCHK(dalloc(&h->d_c48[i], C * h->L));
		HIPCHK(dalloc(&h->d_fz[i], C * h->W));
		HIPCHK(dalloc(&h->d_ppm[i], C * h->W));
		HIPCHK(dalloc(&h->d_rotT[i], (size_t)h->L * ((C + 63) / 64 * 64)));
	}
	for (int i = 0; i < 2; i++) {
		HIPCHK(dalloc(&h->d_sym[i], C * 5 * h->Gcap));
		HIPCHK(dalloc(&h->d_lvl[i], C * h->Gcap));
		HIPCHK(dalloc(&h->d_bits[i], C * 5 * h->words));
	}
	HIPCHK(dalloc(&h->d_cgf, C * (CGF_HIST + h->L)));
	HIPCHK(dalloc(&h->d_rotstate, C));
	{
		std::vector<float2> ones(C, make_float2(1.0f, 0.0f)); // SquareFreqOffsetCorrection::rot = 1.0f (DSP.h:379)
		HIPCHK(hipMemcpy(h->d_rotstate, ones.data(), C * sizeof(float2), hipMemcpyHostToDevice));
	}
	HIPCHK(dalloc(&h->d_ema[0], C * 5));
	HIPCHK(dalloc(&h->d_ema[1], C * 5));
	h->ps_chunks = (h->Gcap + PS_CHUNK - 1) / PS_CHUNK;
	if (const char* e = getenv("AISGPU_PS_WARM")) { int v = atoi(e); if (v >= 1 && v <= PS_CHUNK) h->ps_warm = (v + 15) / 16 * 16; } // test hook: small values force the exact fallback
	if (getenv("AISGPU_PS_SEQUENTIAL")) h->ps_parallel = false;
	HIPCHK(dalloc(&h->d_pswords, C * 5 * h->ps_chunks * (PS_CHUNK / 32) * 16));
	HIPCHK(dalloc(&h->d_psma0, C * 5 * h->ps_chunks * 16));
	HIPCHK(dalloc(&h->d_psma1, C * 5 * h->ps_chunks * 16));
	HIPCHK(dalloc(&h->d_psfin, C * 5 * h->ps_chunks * 16));
	HIPCHK(dalloc(&h->d_psflag, 4));
	if (cfg->flags & AISGPU_FLAG_TAPS) HIPCHK(dalloc(&h->d_firtap, C * (8 + h->L)));
	HIPCHK(hipHostMalloc((void**)&h->h_bits, C * 5 * h->words * sizeof(uint32_t), hipHostMallocDefault));
	HIPCHK(hipHostMalloc((void**)&h->h_lvl, C * h->Gcap * sizeof(float), hipHostMallocDefault));
	HIPCHK(hipHostMalloc((void**)&h->h_ppm, C * h->W * sizeof(float), hipHostMallocDefault));
	HIPCHK(hipDeviceSynchronize());
	return AISGPU_OK;
}

void aisgpu_destroy(aisgpu_t* h) {
	if (!h) return;
	if (h->stream) hipStreamSynchronize(h->stream);
	if (h->s1) hipStreamSynchronize(h->s1);
	if (h->s2) hipStreamSynchronize(h->s2);
	if (h->s3) hipStreamSynchronize(h->s3);
	drain_events(h);
	for (int i = 0; i < NBUF; i++) {
		if (h->ev_front[i]) hipEventDestroy(h->ev_front[i]);
		if (h->ev_phasor[i]) hipEventDestroy(h->ev_phasor[i]);
		if (h->ev_c48free[i]) hipEventDestroy(h->ev_c48free[i]);
		hipFree(h->d_rotT[i]); hipFree(h->d_c48[i]); hipFree(h->d_fz[i]); hipFree(h->d_ppm[i]);
	}
	for (int i = 0; i < 2; i++) {
		if (h->ev_mid[i]) hipEventDestroy(h->ev_mid[i]);
		if (h->ev_ema[i]) hipEventDestroy(h->ev_ema[i]);
		hipFree(h->d_sym[i]); hipFree(h->d_lvl[i]); hipFree(h->d_bits[i]);
	}
	for (auto& p : h->ev_free) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
	hipFree(h->d_in); hipFree(h->d_hist);
	for (int i = 0; i < 2; i++) { hipFree(h->d_rot[i]); if (h->h_rot[i]) hipHostFree(h->h_rot[i]); if (h->rot_ev[i]) hipEventDestroy(h->rot_ev[i]); }
	hipFree(h->d_cgf); hipFree(h->d_omega); hipFree(h->d_step);
	hipFree(h->d_rotstate); hipFree(h->d_firtap); hipFree(h->d_ppmtab); hipFree(h->d_ema[0]); hipFree(h->d_ema[1]);
	hipFree(h->d_pswords); hipFree(h->d_psma0); hipFree(h->d_psma1); hipFree(h->d_psfin); hipFree(h->d_psflag);
	if (h->h_in) hipHostFree(h->h_in);
	if (h->h_bits) hipHostFree(h->h_bits);
	if (h->h_lvl) hipHostFree(h->h_lvl);
	if (h->h_ppm) hipHostFree(h->h_ppm);
	if (h->stream) hipStreamDestroy(h->stream);
	if (h->s1 && !h->serial) hipStreamDestroy(h->s1);
	if (h->s3 && !h->serial) hipStreamDestroy(h->s3);
	delete h;
}

int aisgpu_submit(aisgpu_t* h, int rx, const void* iq, int n_iq) {
	if (!h || !iq || rx < 0 || rx >= h->cfg.n_receivers || n_iq != h->cfg.block_len) return AISGPU_ERR_ARG;
	const size_t row = (size_t)h->cfg.block_len * h->in_bytes;
	if (!h->d_in) {
		HIPCHK(hipMalloc(&h->d_in, row * h->cfg.n_receivers));
		HIPCHK(hipHostMalloc(&h->h_in, row * h->cfg.n_receivers, hipHostMallocDefault));
	}
	// the caller's buffer is only borrowed for this call (Device/FileRAW.cpp:131-136): copy to pinned staging
	// (the previous block's H2D copies are complete: run() orders them before its kernels on the same stream,
	//  and a new block is only staged after the previous run() was enqueued; we wait for it here)
	if (!h->submitted && h->block_idx > 0) HIPCHK(hipStreamSynchronize(h->stream));
	memcpy((char*)h->h_in + row * rx, iq, row);
	HIPCHK(hipMemcpyAsync((char*)h->d_in + row * rx, (char*)h->h_in + row * rx, row, hipMemcpyHostToDevice, h->stream));
	h->cur_in = h->d_in;
	h->cur_in_stride = h->cfg.block_len;
	h->submitted = true;
	return AISGPU_OK;
}

int aisgpu_submit_device(aisgpu_t* h, const void* iq_dev, long long rx_stride_samples) {
	if (!h || !iq_dev || rx_stride_samples < h->cfg.block_len) return AISGPU_ERR_ARG;
	if (((uintptr_t)iq_dev & 15) || ((rx_stride_samples * h->in_bytes) & 15)) return AISGPU_ERR_ARG; // 16-byte vector loads
	h->cur_in = iq_dev;
	h->cur_in_stride = rx_stride_samples;
	h->submitted = true;
	return AISGPU_OK;
}

int aisgpu_run(aisgpu_t* h) {
	if (!h) return AISGPU_ERR_ARG;
	if (!h->submitted) return AISGPU_ERR_STATE;
	HIPCHK(hipSetDevice(h->cfg.device_id));
	const int pb = (int)(h->block_idx & 1);
	const int q = (int)(h->block_idx % NBUF); // ring slot of c48 / fz / ppm / rotT
	// the pinned phasor buffer `pb` was last used two blocks ago; wait until that upload has been consumed
	// (only blocks when the host runs more than one block ahead of the device)
	if (h->block_idx >= 2) HIPCHK(hipEventSynchronize(h->rot_ev[pb]));
	gen_rot_table(h, h->h_rot[pb]);
	HIPCHK(hipMemcpyAsync(h->d_rot[pb], h->h_rot[pb], ((size_t)ROT_HIST + h->n96) * sizeof(float2), hipMemcpyHostToDevice, h->stream));
	HIPCHK(hipEventRecord(h->rot_ev[pb], h->stream));

	// c48/fz/ppm[q] were last read by K2b/K2c of block b-NBUF
	HIPCHK(hipStreamWaitEvent(h->stream, h->ev_c48free[q], 0));
	K1Params k1;
	k1.in = h->cur_in; k1.in_stride = h->cur_in_stride; k1.hist = h->d_hist; k1.rot = h->d_rot[pb];
	k1.c48 = h->d_c48[q]; k1.c48_stride = h->L;
	k1.tiles_per_block = h->tiles_per_block; k1.tiles_per_span = h->tiles_per_span;
	k1.alpha = h->alpha; k1.beta = h->beta; k1.has_fdc = h->has_fdc;
	EvPair ev{};
	if (h->timing) {
		if (h->ev_free.empty()) { HIPCHK(hipEventCreate(&ev.a)); HIPCHK(hipEventCreate(&ev.b)); }
		else { ev = h->ev_free.back(); h->ev_free.pop_back(); }
		HIPCHK(hipEventRecord(ev.a, h->stream));
	}
	HIPCHK(launch_k1(k1, h->K, h->cfg.input_format == AISGPU_FMT_CU8, h->tile96, h->depth, h->k1_threads, h->spans, h->cfg.n_receivers, h->stream));
	if (h->timing) { HIPCHK(hipEventRecord(ev.b, h->stream)); h->ev_busy.push_back(ev); }
	HIPCHK(launch_k1_tail(h->cur_in, h->cur_in_stride * h->in_bytes, (long long)h->cfg.block_len * h->in_bytes, h->d_hist,
	                      h->tile_in * h->in_bytes, h->cfg.n_receivers, h->stream));

	K2Params k2;
	k2.c48 = h->d_c48[q]; k2.c48_stride = h->L; k2.cgf = h->d_cgf; k2.cgf_stride = CGF_HIST + h->L;
	k2.omega = h->d_omega; k2.step_table = h->d_step; k2.ppm_table = h->d_ppmtab; k2.fz = h->d_fz[q]; k2.ppm = h->d_ppm[q];
	k2.rot_state = h->d_rotstate; k2.n_windows = h->W; k2.wide = h->cfg.afc_wide ? 1 : 0;
	k2.rotT = h->d_rotT[q]; k2.rotT_stride = (h->n_chan + 63) / 64 * 64; k2.n_chan = h->n_chan;
	HIPCHK(launch_k2a(k2, h->n_chan, h->stream));
	HIPCHK(hipEventRecord(h->ev_front[q], h->stream));

	// ---- s3: sequential CGF phasor recurrence (needs fz of this block; rotT[q] was last read by apply(b-NBUF))
	HIPCHK(hipStreamWaitEvent(h->s3, h->ev_front[q], 0));
	HIPCHK(hipStreamWaitEvent(h->s3, h->ev_c48free[q], 0));
	HIPCHK(launch_k2b(k2, h->n_chan, h->s3));
	HIPCHK(hipEventRecord(h->ev_phasor[q], h->s3));
	// ---- s1: apply the phasors, then FIR-17 + ScatterPLL
	HIPCHK(hipStreamWaitEvent(h->s1, h->ev_phasor[q], 0));
	HIPCHK(launch_k2c(k2, h->n_chan, h->s1));
	HIPCHK(hipEventRecord(h->ev_c48free[q], h->s1));
	// ScatterPLL groups completed inside this block (DSP/DSP.h:95-117): group g completes with sample 5g+4
	const long long g0 = h->n48 / 5, g1 = (h->n48 + h->L) / 5;
	K3Params k3;
	k3.cgf = h->d_cgf; k3.cgf_stride = CGF_HIST + h->L; k3.sym = h->d_sym[pb]; k3.sym_stride = h->Gcap; k3.lvl = h->d_lvl[pb];
	k3.fir_tap = h->d_firtap; k3.fir_tap_stride = 8 + h->L;
	memcpy(k3.taps, TAPS_COHERENT, sizeof k3.taps);
	k3.first_group = g0; k3.first_sample48 = h->n48; k3.n_groups = (int)(g1 - g0);
	HIPCHK(hipStreamWaitEvent(h->s1, h->ev_ema[pb], 0)); // sym/lvl[pb] were last read by K4 of block b-2
	HIPCHK(launch_k3(k3, h->n_chan, h->s1));
	HIPCHK(hipEventRecord(h->ev_mid[pb], h->s1));

	// ---- s2: sequential PhaseSearchEMA chains
	K4Params k4;
	k4.sym = h->d_sym[pb]; k4.sym_stride = h->Gcap; k4.bits = h->d_bits[pb]; k4.bits_stride = h->words;
	k4.state_in = h->d_ema[pb]; k4.state_out = h->d_ema[pb ^ 1];
	k4.words = h->d_pswords; k4.ma_start = h->d_psma0; k4.ma_fin = h->d_psma1; k4.fin = h->d_psfin; k4.flag = h->d_psflag;
	k4.n_chains = h->n_chains; k4.n_groups = (int)(g1 - g0);
	k4.n_chunks = (k4.n_groups + PS_CHUNK - 1) / PS_CHUNK; k4.warm = h->ps_warm;
	HIPCHK(hipStreamWaitEvent(h->s2, h->ev_mid[pb], 0));
	if (h->ps_parallel && k4.n_chunks > 1) HIPCHK(launch_k4(k4, h->s2));
	else HIPCHK(launch_k4_sequential(k4, h->s2));
	HIPCHK(hipEventRecord(h->ev_ema[pb], h->s2));

	h->out_groups = (int)(g1 - g0); h->out_first_group = g0; h->out_first48 = h->n48;
	h->n48 += h->L;
	h->block_idx++;
	h->submitted = false;
	h->have_out = false;
	return AISGPU_OK;
}

static int sync_all(aisgpu_t* h) {
	HIPCHK(hipStreamSynchronize(h->stream));
	HIPCHK(hipStreamSynchronize(h->s1));
	HIPCHK(hipStreamSynchronize(h->s2));
	HIPCHK(hipStreamSynchronize(h->s3));
	drain_events(h);
	return AISGPU_OK;
}

int aisgpu_sync(aisgpu_t* h) {
	if (!h) return AISGPU_ERR_ARG;
	return sync_all(h);
}

int aisgpu_sync_outputs(aisgpu_t* h) {
	if (!h) return AISGPU_ERR_ARG;
	if (h->block_idx == 0) return AISGPU_ERR_STATE;
	const size_t C = h->n_chan;
	const int pb = (int)((h->block_idx - 1) & 1); // buffers of the last block run
	const int q = (int)((h->block_idx - 1) % NBUF);
	// s2 is ordered after K4 of that block, which is ordered after everything that produced lvl/ppm
	HIPCHK(hipMemcpyAsync(h->h_bits, h->d_bits[pb], C * 5 * h->words * sizeof(uint32_t), hipMemcpyDeviceToHost, h->s2));
	HIPCHK(hipMemcpyAsync(h->h_lvl, h->d_lvl[pb], C * h->Gcap * sizeof(float), hipMemcpyDeviceToHost, h->s2));
	HIPCHK(hipMemcpyAsync(h->h_ppm, h->d_ppm[q], C * h->W * sizeof(float), hipMemcpyDeviceToHost, h->s2));
	int rc = sync_all(h);
	if (rc != AISGPU_OK) return rc;
	h->have_out = true;
	return AISGPU_OK;
}

int aisgpu_fetch(aisgpu_t* h, int rx, int ch, aisgpu_out* o) {
	if (!h || !o || rx < 0 || rx >= h->cfg.n_receivers || ch < 0 || ch > 1) return AISGPU_ERR_ARG;
	if (!h->have_out) return AISGPU_ERR_STATE;
	const size_t chan = (size_t)rx * 2 + ch;
	o->n_groups = h->out_groups;
	o->first_group = h->out_first_group;
	for (int j = 0; j < 5; j++) o->bits[j] = h->h_bits + (chan * 5 + j) * h->words;
	o->lvl = h->h_lvl + chan * h->Gcap;
	o->n_windows = h->W;
	o->ppm = h->h_ppm + chan * h->W;
	o->group_window = nullptr;
	o->first_sample48 = h->out_first48;
	return AISGPU_OK;
}

long long aisgpu_tap(aisgpu_t* h, int which, int rx, float* dst, long long cap) {
	if (!h || which < 0 || which > 5 || rx < 0 || rx >= h->cfg.n_receivers) return -AISGPU_ERR_ARG;
	if (!(h->cfg.flags & AISGPU_FLAG_TAPS)) return -AISGPU_ERR_STATE;
	const size_t chan = (size_t)rx * 2 + (which & 1);
	const float2* src;
	long long n = h->L;
	if (h->block_idx == 0) return -AISGPU_ERR_STATE;
	const int q = (int)((h->block_idx - 1) % NBUF);
	if (which < 2) src = h->d_c48[q] + chan * h->L;
	else if (which < 4) src = h->d_cgf + chan * (CGF_HIST + h->L) + CGF_HIST;
	else {
		// FIR outputs exist for every sample that belongs to a group completed in this block:
		// block-relative indices [-carry, 5*n_groups - carry)
		const long long carry = h->out_first48 - h->out_first_group * 5;
		src = h->d_firtap + chan * (8 + h->L) + 4 - carry;
		n = 5LL * h->out_groups;
	}
	if (sync_all(h) != AISGPU_OK) return -AISGPU_ERR_HIP;
	if (dst) {
		long long c = n < cap ? n : cap;
		if (hipMemcpy(dst, src, (size_t)c * sizeof(float2), hipMemcpyDeviceToHost) != hipSuccess) return -AISGPU_ERR_HIP;
	}
	return n;
}

void* aisgpu_stream(aisgpu_t* h) { return h ? (void*)h->stream : nullptr; }

void aisgpu_timing(aisgpu_t* h, int enable) {
	if (!h) return;
	sync_all(h);
	h->timing = enable != 0;
	h->k1_ms = 0;
	h->k1_launches = 0;
}

float aisgpu_frontend_ms(aisgpu_t* h, int* launches) {
	if (!h) return 0;
	sync_all(h);
	if (launches) *launches = h->k1_launches;
	return h->k1_launches ? (float)(h->k1_ms / h->k1_launches) : 0.0f;
}

} // extern "C"
