// ais-catcher_amd/csrc/aisgpu.cpp -- host side of libaisgpu.so: context, tables, buffers, launch order.
//
// Everything data-independent that the reference computes with libm or as a sequential float
// recurrence is produced HERE on the host, with the host's own libm, exactly as the reference
// would on the same machine: FFT twiddles (DSP/FFT.h:83), the Rotate phasor
// sequence incl. its once-per-Receive renormalisation (DSP/DSP.cpp:309,315), the finite set
// of CGF rot_step phasors (DSP/DSP.cpp:457-458) and the fractional resampler's (index, alpha)
// sequence (DSP/DSP.cpp:192-212).  The device never calls sin/cos.
// Compiled with -ffp-contract=off (host and device).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <algorithm>
#include <cctype>
#include <map>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/aisgpu.h"
#include "kernels.h"

using namespace aisk;

namespace {

const float PI_F = 3.14159265358979323846f; // Library/Common.h:318: PI is a float constant
const float TAPS_RECEIVER[37] = { // DSP/Filters.h:24-33
	0.00119025f, -0.00148464f, -0.00282428f, -0.00200561f, -0.00068852f, 0.00343044f, 0.00902093f, 0.01367867f,
	0.01147965f, 0.0027259f, -0.01766614f, -0.04244429f, -0.0577468f, -0.05245161f, -0.01072754f, 0.0732564f,
	0.17643278f, 0.25582214f, 0.28200453f, 0.25582214f, 0.17643278f, 0.0732564f, -0.01072754f, -0.05245161f,
	-0.0577468f, -0.04244429f, -0.01766614f, 0.0027259f, 0.01147965f, 0.01367867f, 0.00902093f, 0.00343044f,
	-0.00068852f, -0.00200561f, -0.00282428f, -0.00148464f, 0.00119025f };
const float TAPS_COHERENT[17] = { // DSP/Filters.h:35-41
	2.06995719e-06f, 3.18610148e-05f, 3.40605309e-04f, 2.52892989e-03f, 1.30411453e-02f, 4.67076746e-02f,
	1.16186141e-01f, 2.00730781e-01f, 2.40861391e-01f, 2.00730781e-01f, 1.16186141e-01f, 4.67076746e-02f,
	1.30411453e-02f, 2.52892989e-03f, 3.40605309e-04f, 3.18610148e-05f, 2.06995719e-06f };

const float TAPS_BH_28_3[26] = { // DSP/Filters.h:45-53 (Filters::BlackmanHarris_28_3)
	6.32542387e-05f, -2.90015252e-04f, -1.54206250e-03f, -1.64972455e-03f, 3.12793899e-03f, 1.09494413e-02f, 9.04975801e-03f,
	-1.43685846e-02f, -4.45615933e-02f, -3.44883647e-02f, 5.53474269e-02f, 2.01827915e-01f, 3.16534610e-01f, 3.16534610e-01f,
	2.01827915e-01f, 5.53474269e-02f, -3.44883647e-02f, -4.45615933e-02f, -1.43685846e-02f, 9.04975801e-03f, 1.09494413e-02f,
	3.12793899e-03f, -1.64972455e-03f, -1.54206250e-03f, -2.90015252e-04f, 6.32542387e-05f };

// Test hooks (aisgpu_set_option, include/aisgpu.h): process-wide key/value pairs read by aisgpu_create().  The release library
// reads NO environment variable; a -DAISGPU_EXPERIMENTS build also looks for AISGPU_<KEY> there.
std::mutex g_opt_mtx;
std::map<std::string, std::string> g_opt;
std::string opt_str(const char* key) {
	{
		std::lock_guard<std::mutex> l(g_opt_mtx);
		auto it = g_opt.find(key);
		if (it != g_opt.end()) return it->second;
	}
#ifdef AISGPU_EXPERIMENTS
	std::string env = "AISGPU_";
	for (const char* c = key; *c; c++) env += (char)toupper((unsigned char)*c);
	if (const char* e = getenv(env.c_str())) return e;
#endif
	return std::string();
}
int opt_int(const char* key, int dflt) {
	const std::string v = opt_str(key);
	return v.empty() ? dflt : atoi(v.c_str());
}

struct EvPair { hipEvent_t a, b; };
struct TraceRec { const char* name; long long block; hipEvent_t a, b; }; // AISGPU_TRACE=1: kernel timeline from HIP events
#ifndef AISGPU_NBUF
#define AISGPU_NBUF 4
#endif
#ifndef KP_PASS_MAX
#define KP_PASS_MAX 5
#endif
#ifndef K_DIRECT_MAX
#define K_DIRECT_MAX 6
#endif
constexpr int XR = 6;  // resampled ladders: ring of pre-decimated input blocks (see d_xpre)
constexpr int NBUF = AISGPU_NBUF;    // ring depth of the buffers that cross from the front-end stream to the others (4 against 3: -1.5 % per step, profiles/r03_expA.txt)
constexpr int MAXSUB = 4;  // downstream blocks ("flushes" of the resampler) that one input block can complete

// How the samples get from the input rate to the two 48 kHz channels (ModelFrontend::buildModel, Model.cpp:129-346)
enum Mode {
	MODE_DIRECT,    // rate == 96k * 2^k, k <= 6: one fused front-end kernel
	MODE_PRE,       // rate == 96k * 2^k, k = 7 (12288 kSPS): three CIC5 stages in a pre-decimation pass, then the fused kernel with four
	MODE_RESAMPLE,  // rate between two buckets: (k-2) CIC5 stages, Upsample to the bucket, DS2_2, DS2_1, ...
	MODE_96K,       // rate == 96k: no decimation in front of Rotate at all (Model.cpp:332-334)
	MODE_DSK,       // rate == 288k * 2^k: k CIC5 stages (or a plain conversion), DownsampleKFilter (/3), Rotate, ...
};

struct SubOut { int pb, lv, q, groups; long long first_group, first48; };

// Rotate phasor tables of the direct / pre-decimated ladders come from a worker thread that runs ahead of the caller: the table
// of a block is a 49,152-step DEPENDENT float recurrence (0.15-0.2 ms on a host core), data independent, so it has no business
// on the thread that enqueues the kernels.  Ring of NR pinned tables; table k lives in slot k % NR until its upload has been
// consumed (the slot's event).  The caller takes the tables strictly in order.
struct RotWorker {
	static constexpr int NR = 4;
	std::thread th; std::mutex m; std::condition_variable cv;
	float2* tab[NR] = {}; float2* tab_dev[NR] = {}; hipEvent_t ev[NR] = {};
	long long produced = 0; // tables 0 .. produced-1 are complete
	long long launched = 0; // tables 0 .. launched-1 have been handed to the device (their slot's event is recorded)
	bool stop = false, started = false;
};

// The same for the resampled ladders (round 4): the tables of a flush -- (input index, alpha) per Upsample output and the Rotate
// phasors -- are 100,000 dependent float steps per input block, data independent: a worker thread builds them ahead, the caller
// only launches the kernels that copy them.  Flush k lives in slot k % USR; run r (input block r) completes nflush[r % NRUN] flushes.
struct UsWorker {
	static constexpr int NRUN = 8;
	std::thread th; std::mutex m; std::condition_variable cv;
	int nflush[NRUN] = {};
	long long produced_runs = 0, taken_runs = 0;    // runs whose tables are complete / whose copies the caller has enqueued
	long long produced_flush = 0, copied_flush = 0; // flushes built / handed to the device (their slot's copy event is recorded)
	bool stop = false, started = false;
};

} // namespace

struct aisgpu {
	aisgpu_cfg cfg;
	Mode mode = MODE_DIRECT;
	int K = 0;            // CIC5 stages executed by the fused front-end kernel (MODE_DIRECT / MODE_PRE)
	int ma_m = 0;         // > 0: `-go MA on`, input samples per 96 kHz sample (the flow of MODE_96K behind launch_ma_rows)
	int KP = 0;           // CIC5 stages of the pre-decimation pass
	int tile96 = 64;      // output samples per front-end tile
	int in_bytes = 0;     // bytes per input sample
	int kfmt = 0;         // kernel numbering of the input format
	int n_pre = 0;        // samples per receiver per input block after the pre-decimation pass
	int n96 = 0, L = 0, W = 0; // per downstream block: 96 kHz samples, 48 kHz samples per channel, CGF windows
	int Gcap = 0, words = 0;   // group capacity per block, bit words per chain
	long long c48s = 0;        // row stride of the 48 kHz arrays: L + 32 so that the rows of consecutive chains start in different
	                           // HBM channels (the fused FIR kernel walks 64 rows in lock step, one per lane)
	int n_chan = 0, n_chains = 0;
	float alpha = 0, beta = 1; int has_fdc = 0;
	float us_increment = 1.0f;

	// Streams software-pipeline consecutive blocks:
	//   s0 (stream): table uploads -> [pre-decimation] -> K1 -> tails -> K2a   (bandwidth-bound front end)
	//   s3: K2b                                         (sequential CGF phasor recurrence, 8 waves, latency bound)
	//   s1 (= s2): K2c -> K3 -> K4 (+ D2H of the outputs) (apply phasors, FIR/ScatterPLL, PhaseSearchEMA)
	// Buffers that cross a stream boundary are ring buffered by downstream-block index.
	hipStream_t stream = nullptr, s1 = nullptr, s2 = nullptr, s3 = nullptr, s4 = nullptr, s5 = nullptr;
	// ds: the stream of everything BEHIND the 48 kHz front-end output on the resampled ladders (FFT, searches, apply, FIR, FM branch).
	// Normally the front stream itself; on a resampled ladder with a pre-decimation pass (6 MSPS: BASELINE configs[2]) it is s4, so
	// that the HBM-bound passes over the NEXT input block (pre-decimation, resampler front end) overlap the dozen small
	// latency-bound kernels behind this one.
	hipStream_t ds = nullptr; hipEvent_t ev_pre[NBUF] = {}; // ev_pre[q]: the resampler front end of downstream block q is done (front stream -> ds)
	hipEvent_t ev_phasor[NBUF] = {};  // s3: phasor(f) done -> s1 may apply it
	hipEvent_t ev_search[NBUF] = {};  // s4: fz(f) known -> s3 may run the phasor recurrence
	hipEvent_t k1_done[NBUF] = {};    // the event bound to the front-end launch of block f (hipExtLaunchKernelGGL), or nullptr: ev_search is recorded behind it
	bool serial = false;
	hipEvent_t ev_front[NBUF] = {};   // s0: K2a(f) done -> s3 may start K2b(f)
	hipEvent_t ev_c48free[NBUF] = {}; // s1: K2c(f) done (c48/fz/rotT[q] consumed) -> s0 may run the front end of f+NBUF
	hipEvent_t ev_ema[4] = {};        // block f (slot f & 3) completely done: bits[f & 1] and lvl[f & 3] written AND consumed by the frame decoder
	hipEvent_t ev_sym[2] = {};        // PhaseSearch has read sym[p] of block f
	hipEvent_t ev_k3[2] = {};         // front stream: sym/lvl[p] of block f written -> s2 may run K4(f)
	hipEvent_t ev_k4[2] = {};         // s2: bits[p] of block f written -> s5 may run the frame decoder
	// device buffers
	void* d_in[2] = {}; void* d_hist[2] = {}; void* d_hist2[2] = {}; // input tails, double buffered (read by span 0, written for the next block)
	// staging of host blocks (aisgpu_submit), double buffered by input block: block f+1 is copied in (pinned buffer, then H2D on a
	// copy stream of its own) while block f is still being computed
	hipStream_t sc = nullptr; hipEvent_t ev_h2d[2] = {}, ev_in_free[2] = {}; bool in_used[2] = {}; std::mutex submit_mtx; bool staged = false;
	float2* d_xpre[XR] = {};           // pre-decimated stream: [R][xh + n_pre], ping-pong by input block (MODE_PRE uses [0] only); resampled ladders: ring of
	                                  // XR [R][n_pre]: a flush may reach a whole input block back (no history copy: K1uParams::xprev / xprev2), and the
	                                  // pass over block f+1 must not wait for the resampler kernels of block f
	bool x_direct = false; float2* d_xhist[2] = {}; // CF32 input of the ladders without a pass at the input rate read in place: the kept tails [R][xh]
	float2* d_xmid = nullptr;         // [R][block_len >> KPa]: between the two passes of a pre-decimation of more than four stages
	bool mode_x = false;              // channel mode X: single-channel front end K1x (npost stages down to 48 kHz); receivers packed (chain = receiver), no channel B
	std::vector<float> h_silent, h_silent_ppm; // what aisgpu_fetch hands out for "channel B" in that mode
	int npost = 2;                    // CIC5 stages behind the resampler (K1u): 2, 1 (192k bucket), 0 (96 kSPS input: no resampler either)
	bool us_dsk = false;              // Upsample in front of DownsampleKFilter (rates below a decimate-by-3 bucket): resampler flow, K1k front end
	int KPa = 0;                      // != 0: the pre-decimation runs as KPa stages, then four (rates above 6144k that are resampled: 8 / 10 MSPS)
	float2* d_rot[4] = {}; // Rotate phasor tables: ring of 4 on the main path (block f & 3, staged two blocks ahead), [f & 1] on the others
	static const int USR = 8; // resampler tables: ring of eight flushes (generated and copied one input block ahead)
	int* d_usidx[USR] = {}; float* d_usalpha[USR] = {}; float2* d_usrot[USR] = {}; float2* h_usrot[USR] = {};
	float2 *us_dev_idx[USR] = {}, *us_dev_alpha[USR] = {}, *us_dev_rot[USR] = {}; bool us_by_kernel = true; // device views of the pinned table buffers
	float2 *d_c48[NBUF] = {}, *d_sym[2] = {};
	float2 *d_rotT[NBUF] = {};
	float2 *d_cgf = nullptr, *d_omega = nullptr, *d_step = nullptr, *d_rotstate = nullptr, *d_firtap = nullptr;
	float *d_ppmtab = nullptr, *d_ppm[NBUF] = {}, *d_lvl[4] = {}; // lvl: ring of 4 (block f & 3): it lives until the block's (deferred) walk and decoder are done
	int* d_fz[NBUF] = {};
	int phasor_simds = 1024;  // SIMDs the phasor recurrence's stream may use (the reserved CUs'): picks the form of the kernel
	float* d_magT[NBUF] = {}; // shifted FFT magnitudes (written by the FFT on the front stream, read by the searches on s3)
	bool fft_in_k1 = false;   // the spectral analysis rides at the end of the front-end waves (k1_fft_tail): fz / ppm come from K1
	bool front_fft = false;   // ... of this block's front-end waves of k1x_wave / k1k_wave (set per block by aisgpu_run, read by enqueue_downstream_fused)
	bool front_fft_us = false; // ... the same on a resampled decimate-by-3 ladder: the caller has recorded k1_done[q] on the stream those waves ran on
	uint32_t* d_bits[4] = {}; // ring of 4 (block f & 3), like lvl: the frame decoder of block f-2 may still be reading while PhaseSearch of block f writes
	bool challenger = false;
	bool v2 = false; float2* h_c48 = nullptr; // ModelEngineV2: front end only, the 48 kHz channels go to the host (MAXSUB slots)
	V2ChanState* d_v2st = nullptr; float2* d_slotcs = nullptr; int* d_v2locked = nullptr; // AISGPU_FLAG_GPU_DECODE with ModelEngineV2: the engine itself on the device (kv2_engine)
	int v2_roles = 1; // kv2_engine: trackers and FM decoder on two waves (test hook "v2_roles" = 0: the one-wave form of round 5)
	bool v2_assist = true; float2* d_v2hist = nullptr; float *d_v2f = nullptr, *d_v2prom = nullptr, *d_v2en = nullptr, *h_v2f = nullptr, *h_v2prom = nullptr, *h_v2en = nullptr; // decoder-independent part of V2::Engine on the device
	// the engine on the device (round 6): on a stream of its own, beside the next block's front end and assist kernels -- what it reads of
	// the assist kernels' outputs exists twice (by block parity): look-back, estimates, energies, the previous block's last discriminator signs
	float2* d_v2hist2 = nullptr; float *d_v2f2 = nullptr, *d_v2prom2 = nullptr, *d_v2en2 = nullptr; uint32_t* d_v2fmtail[2] = { nullptr, nullptr };
	hipEvent_t ev_v2assist = nullptr, ev_v2engine[2] = { nullptr, nullptr }, ev_v2front = nullptr, ev_v2fm = nullptr; int v2_par = 0; hipStream_t v2_stream = nullptr;
	bool k46 = false; // default path: derotation + FIR + PhaseSearch in one kernel (k46_window_search; test hook "k46" = 0: k6_window_fir + k4_phase_chunks)
	bool base = false; float2* d_fmprev[2] = {}; // ModelBase: FM receiver on the 48 kHz channels, no coherent chain
	float* d_fm = nullptr; float* d_fmhist[2] = {}; uint32_t* d_fmbits[2] = {}; uint32_t* h_fmbits = nullptr; // ModelChallenger FM branch
	float* d_fmfir = nullptr; // [n_chan][L] Filter(Receiver) output of the last downstream block (AISGPU_FLAG_TAPS)
	EmaState* d_ema[2] = {}; // state before / after the current downstream block (swapped per block)
	uint32_t* d_pswords = nullptr; float *d_psma0 = nullptr, *d_psma1 = nullptr; unsigned* d_psfin = nullptr; int* d_psflag = nullptr;
	int ps_warm = 256; bool ps_parallel = true;
	struct { bool valid = false; int pb = 0, lv = 0, n_groups = 0; long long g0 = 0; unsigned block = 0, sub = 0; } dpend; // frame decoders not yet enqueued (dec_defer)
	bool dec_defer = false; // the frame decoders of block f are enqueued behind the derotation / FIR kernel of block f+1 (they share its stream)
	// host (pinned)
	void* h_in[2] = {};
	float2* h_rot[4] = {};
	int* h_usidx[USR] = {}; float* h_usalpha[USR] = {};
	// resampled ladders: tables of flush b (slot b & 3) copied (us_copy_ev, table stream) / consumed by its resampler front end (us_used_ev);
	// pre-decimated input block g (slot g % XR) written (ev_xin, front stream) / read for the last time by the flushes of run g (ev_xread)
	hipEvent_t us_copy_ev[USR] = {}, us_used_ev[USR] = {}, ev_xin[XR] = {}, ev_xread[XR] = {}; bool us_slot_used[USR] = {};
	UsWorker uw; long long run_flush = 0, run_idx = 0; int next_nflush = 0; // (run_flush: flushes whose resampler front end has been launched)
	hipEvent_t rot_ev[4] = {}; bool rot_ev_used[4] = {}; long long rot_next = 0; // first block whose table has not been staged yet
	RotWorker rw; int rot_slot[4] = {};
	float2* h_rot_dev[4] = {}; bool rot_by_kernel = true; // device view of the pinned table buffers
	uint32_t* h_bits = nullptr; float* h_lvl = nullptr; float* h_ppm = nullptr; // MAXSUB slots each
	// stream state
	long long in_blocks = 0;     // input blocks run so far
	int rot_period = 0;          // Rotate renormalisation period in 96 kHz samples (0: once per block)
	long long block_idx = 0;     // downstream blocks run so far
	long long n48 = 0;           // 48 kHz samples consumed before the current downstream block
	float2 rot = { 1.0f, 0.0f }; // Rotate::rot carried across Receive() calls
	float2 mult = { 1.0f, 0.0f };
	std::vector<float2> rot_tail; // last ROT_HIST phasors of the previous downstream block
	// resampler replay (DSP.cpp:192-212): alpha carried, outputs waiting for a full flush
	float us_alpha = 0.0f;
	long long us_in = 0;          // inputs consumed so far (pre-decimated samples)
	std::vector<int> us_pend_idx; std::vector<float> us_pend_alpha; int us_pend_n = 0; // (the table thread's)
	std::vector<long long> us_tail_idx; std::vector<float> us_tail_alpha; // last US_HIST entries of the previous flush
	const void* cur_in = nullptr; long long cur_in_stride = 0;
	bool submitted = false, have_out = false;
	SubOut sub[MAXSUB]; int n_sub = 0;
	// Resamplers that complete more than two downstream blocks per input block (channel mode X below 24 kSPS: up to four) would
	// overwrite the two-deep device output rings before aisgpu_sync_outputs() copies them: there every downstream block's outputs
	// are copied to its host slot as soon as they exist (two sets of host slots, by input block)
	bool eager_out = false; int out_set = 0, oset = 0;
	SubOut osub[MAXSUB]; int n_osub = 0; // the downstream blocks whose outputs the last aisgpu_sync_outputs() brought to the host (what fetch serves)
	struct { bool valid = false; int q = 0, pb = 0, lv = 0; long long g0 = 0, g1 = 0, first48 = 0; unsigned block = 0, sub = 0; } pend; // deferred second half
	bool defer = true;
	// device frame decoder (AISGPU_FLAG_GPU_DECODE)
	int dec_kind = 0; uint32_t* d_fmrows[2] = {}; float* d_last_lvl[2] = {}; int fmrow_words = 0; // device decoders of ModelStandard (1) / ModelChallenger (2) / ModelBase (3)
	bool gpu_decode = false; DecState* d_dec = nullptr; uint32_t* d_frames = nullptr; unsigned* d_frame_count = nullptr;
	bool k7_alt = false; // test hook (AISGPU_K7=alt): the two decoder implementations take turns, block by block, on the same DecState
	long long k7e_pass = 0; // passes of the event-driven decoder kernels so far (parity: which overflow flag a pass uses)
	bool k7_event = true; uint32_t *d_k7ev = nullptr, *d_k7cnt = nullptr; uint16_t* d_k7open = nullptr; K7Slot* d_k7slot = nullptr; int* d_k7ovf = nullptr;
	uint32_t* h_frames = nullptr; unsigned frames_seen = 0; int max_frames = 0; std::vector<aisgpu_frame> frames;
	// ModelBase's sampler + decoder loop, chunk-parallel (k7b_*).  Two sets of the speculative pass's scratch, by block parity: the pass
	// of block f+1 (s4) runs beside the boundary tasks of block f (s1); ev_spec[p]: the pass of the block with parity p is done.
	K7bParams k7b[2] = {}; bool base_chunked = false; hipEvent_t ev_spec[2] = {};
	bool ps_box = false; PsBoxState* d_box[2] = {}; // Demod::PhaseSearch (boxcar) instead of PhaseSearchEMA
	bool trace = false; std::vector<TraceRec> trace_recs; hipEvent_t trace_origin = nullptr;
	// fused derotation + FIR path (no phasor / derotated-sample arrays in HBM); off when taps or the FM branch need them
	int k1u_spw = 0; // test hook "k1u_spw": forces the resampler front end's span walk (launch_k1u); channel mode X at 96 kSPS: 2 = k1x_single_channel, 4 / 8 = k1x_wave's span length
	// Round 6: the tail of a resampled ladder with two stages behind Upsample (every bucket from 384k up: 6 MSPS, 2.4 MSPS, 10 MSPS ...)
	// as one-wave workgroups of the front-end kernel itself -- k1_dpp<2, 5, false>: Upsample outputs computed where the other formats
	// convert, DS2_2 / DS2_1 in registers, and with the fused back end the spectral analysis at the end of its waves -- instead of
	// k1u_resample_frontend (320-thread workgroups, six barriers per 128 outputs) + k2_fft_search_win.  Test hook "us_k1" = 0: the old pair.
	bool us_k1 = false, us_fft_in_k1 = false; int us_tiles_per_block = 0, us_tiles_per_span = 0;
	bool us_on_ds = false; // resampled ladders: the resampler front end on the downstream stream, the second half of a flush one flush late (fixed by the mode at create, not an option)
	bool fm_on_s1 = false, fm_ev_used = false; hipEvent_t ev_fm = nullptr; // where the device decoders' regrouping of ModelChallenger's FM bits (k7_pack) runs: in front of PhaseSearch on s1 on the resampled ladders, else behind K6 on s4 (fixed by the mode; ev_fm only exists with challenger + gpu_decode + a resampled ladder)
	bool fused = false; // derotation + FIR + ScatterPLL as one kernel behind the checkpointed phasor recurrence (the default)
	struct { bool valid = false; int q = 0, pb = 0, lv = 0, n_groups = 0, n_rel0 = 0, S = 0; long long g0 = 0; unsigned block = 0, sub = 0; } fpend;
	float2 *d_ck[NBUF] = {}, *d_dfhist[2] = {};
	// per-kernel geometry
	int tile_in = 0, tiles_per_block = 0, tiles_per_span = 0, spans = 0;         // fused front end (its own input)
	int ptile_in = 0, ptiles_per_block = 0, ptiles_per_span = 0, pspans = 0;     // pre-decimation pass
	int xh = 0;                                                                  // history samples in front of d_xpre rows
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

// HIP's current device is a per-thread setting and the entry points are called from the receivers' own threads
// (GpuBatch::submitAndWait): every entry point that touches HIP selects the context's device and restores the caller's.
struct DevGuard {
	int prev = -1;
	explicit DevGuard(const aisgpu_t* h) {
		if (hipGetDevice(&prev) != hipSuccess) prev = -1;
		if (prev != h->cfg.device_id) (void)hipSetDevice(h->cfg.device_id); else prev = -1;
	}
	~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Cross-stream dependency.  A wait on an event that has already completed is elided on the host (hipEventQuery): a barrier
// packet costs the command processor microseconds even when its signal is long satisfied, and the front stream had two of
// them between consecutive front-end launches.
hipError_t wait_event(aisgpu_t* h, hipStream_t s, hipEvent_t ev);
#define WAITEV(stream_, ev_) HIPCHK(wait_event(h, (stream_), (ev_)))

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
	const int period = h->rot_period > 0 ? h->rot_period : h->n96; // samples per Rotate::Receive call
	// (a countdown, not (i + 1) % period: the division was two thirds of this loop, and the loop half of the resampled ladders'
	// table thread -- 0.19 of its 0.40 ms per input block, which at 6 MSPS was the step)
	for (int i = 0, left = period; i < h->n96; i++) {
		t[i] = r;
		float re = r.x * m.x - r.y * m.y;
		float im = r.x * m.y + r.y * m.x;
		r.x = re; r.y = im;
		if (--left == 0) { // rot /= std::abs(rot) at the end of every call (DSP.cpp:315)
			float a = hypotf(r.x, r.y);
			r.x /= a; r.y /= a;
			left = period;
		}
	}
	h->rot = r;
	for (int i = 0; i < ROT_HIST; i++) h->rot_tail[i] = t[h->n96 - ROT_HIST + i];
}

void rot_worker_main(aisgpu_t* h) {
	RotWorker& w = h->rw;
	(void)hipSetDevice(h->cfg.device_id);
	for (long long k = 0;; k++) {
		const int slot = (int)(k % RotWorker::NR);
		{
			std::unique_lock<std::mutex> l(w.m);
			w.cv.wait(l, [&] { return w.stop || k < RotWorker::NR || w.launched > k - RotWorker::NR; });
			if (w.stop) return;
		}
		if (k >= RotWorker::NR) (void)hipEventSynchronize(w.ev[slot]); // the upload of table k - NR has been consumed
		gen_rot_table(h, w.tab[slot]); // (h->rot / rot_tail belong to this thread from now on)
		{ std::lock_guard<std::mutex> l(w.m); w.produced = k + 1; }
		w.cv.notify_all();
	}
}

// next table (strictly in order) -> d_rot[b] on stream st
int stage_rot_from_worker(aisgpu_t* h, int b, hipStream_t st) {
	RotWorker& w = h->rw;
	if (!w.started) {
		for (int i = 0; i < RotWorker::NR; i++) {
			HIPCHK(hipHostMalloc((void**)&w.tab[i], ((size_t)ROT_HIST + h->n96) * sizeof(float2), hipHostMallocDefault));
			HIPCHK(hipEventCreateWithFlags(&w.ev[i], hipEventDisableTiming));
			if (hipHostGetDevicePointer((void**)&w.tab_dev[i], w.tab[i], 0) != hipSuccess) h->rot_by_kernel = false;
		}
		w.started = true;
		w.th = std::thread(rot_worker_main, h);
	}
	const long long k = w.launched; // only this thread writes it
	{
		std::unique_lock<std::mutex> l(w.m);
		w.cv.wait(l, [&] { return w.produced > k; });
	}
	const int slot = (int)(k % RotWorker::NR);
	if (h->rot_by_kernel) HIPCHK(launch_copy_rows(w.tab_dev[slot], 0, h->d_rot[b], 0, ROT_HIST + h->n96, 1, st));
	else HIPCHK(hipMemcpyAsync(h->d_rot[b], w.tab[slot], ((size_t)ROT_HIST + h->n96) * sizeof(float2), hipMemcpyHostToDevice, st));
	HIPCHK(hipEventRecord(w.ev[slot], st));
	h->rot_slot[b] = slot;
	{ std::lock_guard<std::mutex> l(w.m); w.launched = k + 1; }
	w.cv.notify_all();
	return AISGPU_OK;
}

void rot_worker_stop(aisgpu_t* h) {
	RotWorker& w = h->rw;
	if (!w.started) return;
	{ std::lock_guard<std::mutex> l(w.m); w.stop = true; }
	w.cv.notify_all();
	if (w.th.joinable()) w.th.join();
	for (int i = 0; i < RotWorker::NR; i++) { if (w.tab[i]) hipHostFree(w.tab[i]); if (w.ev[i]) hipEventDestroy(w.ev[i]); }
	w.started = false;
}

hipError_t wait_event(aisgpu_t* h, hipStream_t s, hipEvent_t ev) {
	if (hipEventQuery(ev) == hipSuccess) return hipSuccess;
	(void)hipGetLastError(); // (hipErrorNotReady is not an error)
	return hipStreamWaitEvent(s, ev, 0);
}

void drain_events(aisgpu_t* h) {
	for (auto& p : h->ev_busy) {
		float ms = 0;
		if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { h->k1_ms += ms; h->k1_launches++; }
		h->ev_free.push_back(p);
	}
	h->ev_busy.clear();
}

// AISGPU_TRACE=1: bracket a launch with events on its stream; dumped (relative to the first record) by trace_dump()
struct TraceScope {
	aisgpu_t* h; hipStream_t s; TraceRec r{};
	TraceScope(aisgpu_t* h_, const char* name, hipStream_t s_) : h(h_), s(s_) {
		if (!h->trace) return;
		if (!h->trace_origin) { hipEventCreate(&h->trace_origin); hipEventRecord(h->trace_origin, s); }
		r.name = name; r.block = h->block_idx;
		hipEventCreate(&r.a); hipEventCreate(&r.b);
		hipEventRecord(r.a, s);
	}
	~TraceScope() {
		if (!h->trace) return;
		hipEventRecord(r.b, s);
		h->trace_recs.push_back(r);
	}
};
void trace_dump(aisgpu_t* h) {
	if (!h->trace || h->trace_recs.empty()) return;
	for (auto& r : h->trace_recs) {
		float a = 0, b = 0;
		hipEventElapsedTime(&a, h->trace_origin, r.a);
		hipEventElapsedTime(&b, h->trace_origin, r.b);
		fprintf(stderr, "TRACE %-8s block %3lld  %10.1f -> %10.1f  (%7.1f us)\n", r.name, r.block, a * 1e3, b * 1e3, (b - a) * 1e3);
		hipEventDestroy(r.a); hipEventDestroy(r.b);
	}
	h->trace_recs.clear();
}

int span_tiles(int tiles_per_block, int n_rx, int requested) {
	int tps = requested;
	if (tps <= 0) {
		tps = tiles_per_block;
		const long long want = 8192; // workgroups: a few per CU per residency slot
		while (tps > 8 && (long long)n_rx * ((tiles_per_block + tps - 1) / tps) < want) tps = (tps + 1) / 2;
	}
	if (tps > tiles_per_block) tps = tiles_per_block;
	return tps;
}

int enqueue_downstream_fused(aisgpu_t* h, int q, int pb);
int enqueue_fused_back(aisgpu_t* h);

// what follows PhaseSearch of a block: the optional device frame decoder, and the event that frees sym/lvl/bits[pb]
int enqueue_decode(aisgpu_t* h, int pb, int lv, long long g0, int n_groups, unsigned block, unsigned sub, hipStream_t s);
// eager_out: this downstream block's outputs to their host slot, on the stream that has just produced the last of them
int copy_out_eager(aisgpu_t* h, int pb, int lv, unsigned block, unsigned sub, hipStream_t s) {
	if (!h->eager_out || sub >= (unsigned)MAXSUB) return AISGPU_OK;
	const size_t C = h->n_chan, slot = (size_t)h->out_set * MAXSUB + sub;
	const int q = (int)(block % NBUF);
	HIPCHK(hipMemcpyAsync(h->h_bits + slot * C * 5 * h->words, h->d_bits[lv], C * 5 * h->words * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
	HIPCHK(hipMemcpyAsync(h->h_lvl + slot * C * h->Gcap, h->d_lvl[lv], C * h->Gcap * sizeof(float), hipMemcpyDeviceToHost, s));
	HIPCHK(hipMemcpyAsync(h->h_ppm + slot * C * h->W, h->d_ppm[q], C * h->W * sizeof(float), hipMemcpyDeviceToHost, s));
	return AISGPU_OK;
}

// the frame decoders of the block whose PhaseSearch has been enqueued (dec_defer: one block later, see aisgpu_create)
int flush_decode(aisgpu_t* h) {
	if (!h->dpend.valid) return AISGPU_OK;
	h->dpend.valid = false;
	const int pb = h->dpend.pb, lv = h->dpend.lv;
	WAITEV(h->s5, h->ev_k4[pb]);
	int rc = enqueue_decode(h, pb, lv, h->dpend.g0, h->dpend.n_groups, h->dpend.block, h->dpend.sub, h->s5);
	if (rc) return rc;
	rc = copy_out_eager(h, pb, lv, h->dpend.block, h->dpend.sub, h->s5);
	if (rc) return rc;
	HIPCHK(hipEventRecord(h->ev_ema[lv], h->s5));
	return AISGPU_OK;
}

int finish_k4(aisgpu_t* h, int pb, int lv, long long g0, int n_groups, unsigned block, unsigned sub, hipStream_t s) {
	if (h->gpu_decode && h->dec_defer) {
		{ int rc = flush_decode(h); if (rc) return rc; }
		HIPCHK(hipEventRecord(h->ev_k4[pb], s));
		h->dpend.valid = true; h->dpend.pb = pb; h->dpend.lv = lv; h->dpend.g0 = g0; h->dpend.n_groups = n_groups; h->dpend.block = block; h->dpend.sub = sub;
		if (h->serial) return flush_decode(h);
		return AISGPU_OK;
	}
	if (h->gpu_decode) { // the frame decoder is a long latency-bound kernel of a few waves: own stream, so that the next
		// block's PhaseSearchEMA does not queue behind it; sym/lvl/bits[pb] are free again only when IT is done
		HIPCHK(hipEventRecord(h->ev_k4[pb], s));
		WAITEV(h->s5, h->ev_k4[pb]);
		int rc = enqueue_decode(h, pb, lv, g0, n_groups, block, sub, h->s5);
		if (rc) return rc;
		rc = copy_out_eager(h, pb, lv, block, sub, h->s5);
		if (rc) return rc;
		HIPCHK(hipEventRecord(h->ev_ema[lv], h->s5));
	} else {
		int rc = copy_out_eager(h, pb, lv, block, sub, s);
		if (rc) return rc;
		HIPCHK(hipEventRecord(h->ev_ema[lv], s));
	}
	return AISGPU_OK;
}

// PhaseSearchEMA / PhaseSearch of one downstream block (sym/lvl parity pb) on stream s
int enqueue_k4(aisgpu_t* h, int pb, int lv, long long g0, int n_groups, unsigned block, unsigned sub, hipStream_t s) {
	// bits[lv] was last read by the frame decoder / the copies of block f-4: long done, and ordered here.  (A ring of two made
	// PhaseSearch(f) wait for the frame decoders of block f-2, which start behind PhaseSearch(f-2): a loop of two steps that
	// had to hold a PhaseSearch and a decoder pass one after the other -- 0.62 ms per step with the decoders on the device.)
	WAITEV(s, h->ev_ema[lv]);
	K4Params k4;
	k4.sym = h->d_sym[pb]; k4.sym_stride = h->Gcap; k4.bits = h->d_bits[lv]; k4.bits_stride = h->words;
	k4.state_in = h->d_ema[pb]; k4.state_out = h->d_ema[pb ^ 1];
	k4.words = h->d_pswords; k4.ma_start = h->d_psma0; k4.ma_fin = h->d_psma1; k4.fin = h->d_psfin; k4.fb_count = h->d_psflag + 2;
	k4.n_chains = h->n_chains; k4.n_groups = n_groups;
	k4.n_chunks = (k4.n_groups + PS_CHUNK - 1) / PS_CHUNK; k4.warm = h->ps_warm;
	k4.box_in = h->d_box[pb]; k4.box_out = h->d_box[pb ^ 1]; k4.first_group = g0;
	if (h->ps_box) { k4.chunked = h->ps_parallel ? 1 : 0; HIPCHK(launch_k4_box(k4, s)); }
	else if (h->ps_parallel && k4.n_chunks > 1) HIPCHK(launch_k4(k4, s));
	else HIPCHK(launch_k4_sequential(k4, s));
	HIPCHK(hipEventRecord(h->ev_sym[pb], s));
	return finish_k4(h, pb, lv, g0, n_groups, block, sub, s);
}

// AIS::Decoder on the device, behind PhaseSearchEMA of the same block (same stream)
K7Params make_k7(aisgpu_t* h, int pb, int lv, long long g0, int n_groups, unsigned block, unsigned sub) {
	K7Params k7;
	k7.bits = h->d_bits[lv]; k7.bits_stride = h->words; k7.lvl = h->d_lvl[lv]; k7.lvl_stride = h->Gcap;
	k7.state = h->d_dec; k7.frames = h->d_frames; k7.frame_count = h->d_frame_count; k7.max_frames = h->max_frames;
	k7.first_group = g0; k7.n_groups = n_groups; k7.n_chan = h->n_chan; k7.block = block; k7.sub = sub;
	k7.kind = h->dec_kind;
	if (h->dec_kind != 0) { // ModelStandard / ModelChallenger / ModelBase wirings
		k7.fm_cur = h->d_fmbits[pb]; k7.fm_prev = h->d_fmbits[pb ^ 1]; k7.fm_stride = h->L / 32;
		k7.fmrows = h->d_fmrows[pb]; k7.fmrows_stride = h->fmrow_words; k7.last_lvl_in = h->d_last_lvl[pb ^ 1]; k7.last_lvl = h->d_last_lvl[pb];
		k7.n_rel0 = (int)(g0 * 5 - (long long)block * h->L); k7.L = h->L; // (every downstream block has L samples: block * L is its first one)
	}
	return k7;
}

// the event-driven decoder kernels, and behind them the sequential kernel of the same wiring as their exact fallback (it runs only
// for a block in which some decoder has more candidates / frame starts than the lists hold: the event-driven kernels then touch
// nothing).  kq: the parameters as the event-driven kernels see them; kseq: as the sequential kernel does (ModelStandard differs).
int launch_decoders_event(aisgpu_t* h, const K7Params& kq, K7Params kseq, hipStream_t s) {
	if (kq.n_groups <= 0) return AISGPU_OK; // (nothing is launched: the pass parity -- which overflow flag a pass uses / clears -- must not advance)
	const int par = (int)(h->k7e_pass++ & 1);
	K7eParams q;
	q.k = kq; q.ev = h->d_k7ev; q.cnt = h->d_k7cnt; q.open_c = h->d_k7open; q.slot = h->d_k7slot;
	q.overflow = h->d_k7ovf + par; q.overflow_clear = h->d_k7ovf + (par ^ 1);
	HIPCHK(launch_k7e_runs(q, s));
	// (The walk -- a few dozen latency-bound waves -- runs 5-8 times slower beside the front end than alone.  On the phasor recurrence's
	// stream, whose CUs the back end does not use, the kernel itself took 0.08 instead of 0.22-0.30 ms, but it then sits behind the
	// next block's recurrence and the decoders' stream waits for it: 0.65 against 0.54 ms per step.  A stream of its own is a fifth
	// one: see aisgpu_create.)
	HIPCHK(launch_k7e_resolve(q, s));
	kseq.cond = q.overflow; kseq.cond_count = h->d_k7ovf + 2;
	if (kseq.kind == 0) HIPCHK(launch_k7(kseq, s)); else HIPCHK(launch_k7_mesh(kseq, s));
	return AISGPU_OK;
}

int enqueue_decode(aisgpu_t* h, int pb, int lv, long long g0, int n_groups, unsigned block, unsigned sub, hipStream_t s) {
	if (!h->gpu_decode) return AISGPU_OK;
	const K7Params k7 = make_k7(h, pb, lv, g0, n_groups, block, sub);
	if (h->dec_kind != 0 && !(h->dec_kind == 2 && h->k7_event && !(h->k7_alt && (block & 1)))) { // (the FM bits were regrouped on the stream that produced them, see launch_k7_pack)
		HIPCHK(launch_k7_mesh(k7, s));
		return AISGPU_OK;
	}
	if (h->k7_event && !(h->k7_alt && (block & 1))) { // event-driven decoders (kernels.h): same DecState between blocks, so the two can even alternate
		return launch_decoders_event(h, k7, k7, s);
	} else if (h->dec_kind == 2) HIPCHK(launch_k7_mesh(k7, s));
	else HIPCHK(launch_k7(k7, s));
	return AISGPU_OK;
}

K2Params make_k2(aisgpu_t* h, int q) {
	K2Params k2;
	k2.c48 = h->d_c48[q]; k2.c48_stride = h->c48s; k2.cgf = h->d_cgf; k2.cgf_stride = CGF_HIST + h->L;
	k2.omega = h->d_omega; k2.step_table = h->d_step; k2.ppm_table = h->d_ppmtab; k2.magT = h->d_magT[q]; k2.fz = h->d_fz[q]; k2.ppm = h->d_ppm[q];
	k2.rot_state = h->d_rotstate; k2.n_windows = h->W; k2.wide = h->cfg.afc_wide ? 1 : 0;
	k2.rotT = h->d_rotT[q]; k2.rotT_stride = (h->n_chan + 63) / 64 * 64; k2.n_chan = h->n_chan;
	return k2;
}

// Second half of a downstream block (deferred, see enqueue_downstream): apply the phasors, FIR-17 + ScatterPLL
// (+ the Challenger FM branch) on the front stream, PhaseSearchEMA on s1.
int enqueue_back(aisgpu_t* h) {
	if (!h->pend.valid) return AISGPU_OK;
	h->pend.valid = false;
	const int q = h->pend.q, pb = h->pend.pb, lv = h->pend.lv;
	const long long g0 = h->pend.g0, g1 = h->pend.g1;
	const K2Params k2 = make_k2(h, q);
	WAITEV(h->ds, h->ev_phasor[q]);
	HIPCHK(launch_k2c(k2, h->n_chan, h->ds));
	HIPCHK(hipEventRecord(h->ev_c48free[q], h->ds));
	K3Params k3;
	k3.cgf = h->d_cgf; k3.cgf_stride = CGF_HIST + h->L; k3.sym = h->d_sym[pb]; k3.sym_stride = h->Gcap; k3.lvl = h->d_lvl[lv];
	k3.fir_tap = h->d_firtap; k3.fir_tap_stride = 8 + h->L;
	memcpy(k3.taps, TAPS_COHERENT, sizeof k3.taps);
	k3.first_group = g0; k3.first_sample48 = h->pend.first48; k3.n_groups = (int)(g1 - g0);
	WAITEV(h->ds, h->ev_sym[pb]); // sym[pb] was last read by PhaseSearch of block f-2,
	WAITEV(h->ds, h->ev_ema[lv]); // lvl[lv] by the frame decoder / the copies of block f-4
	HIPCHK(launch_k3(k3, h->n_chan, h->ds));
	if (h->challenger) { // FM branch on the same derotated samples (Model.cpp:638-639)
		K5Params k5;
		k5.x = h->d_cgf; k5.x_stride = CGF_HIST + h->L; k5.x_off = CGF_HIST; k5.prev_in = nullptr; k5.prev_out = nullptr; k5.fm = h->d_fm; k5.fm_stride = FM_HIST + h->L;
		k5.hist_in = h->d_fmhist[pb]; k5.hist_out = h->d_fmhist[pb ^ 1];
		k5.fmbits = h->d_fmbits[pb]; k5.fmbits_stride = h->L / 32; k5.L = h->L;
		k5.fir_out = h->d_fmfir; k5.fir_stride = h->L;
		memcpy(k5.taps, TAPS_RECEIVER, sizeof k5.taps);
		HIPCHK(launch_k5(k5, h->n_chan, h->ds));
		if (h->gpu_decode) { // device decoders: the FM bits regrouped per decoder, here, where this and the previous block's bits are in order
			WAITEV(h->ds, h->ev_ema[lv ^ 2]); // fmrows[pb] was last read by the decoders of block f-2
			HIPCHK(launch_k7_pack(make_k7(h, pb, lv, g0, (int)(g1 - g0), h->pend.block, h->pend.sub), h->ds));
		}
	}
	HIPCHK(hipEventRecord(h->ev_k3[pb], h->ds));

	// ---- PhaseSearchEMA chains on s1: VALU-bound, overlaps the HBM-bound front end of the next block
	WAITEV(h->s2, h->ev_k3[pb]);
	return enqueue_k4(h, pb, lv, g0, (int)(g1 - g0), h->pend.block, h->pend.sub, h->s2);
}

// Default path: the phasor recurrence keeps checkpoints only, and one fused kernel derotates, filters and scatters.
// Nothing behind the spectral analysis touches the front stream, so nothing needs to be deferred:
//   front stream: front end with the spectral analysis inside its waves (k1_fft_tail); only where a span is not a whole number
//                 of windows the FFT and search kernels follow it here (four streams are the limit)
//   s3: phasor recurrence, own CUs          (latency-bound)
//   s4: derotation + FIR + ScatterPLL       (VALU/latency-bound)
//   s1: PhaseSearchEMA                      (VALU-bound; both overlap the next block's front end)
// second half of the fused path for one block: derotation + FIR + ScatterPLL (s4), then PhaseSearch (s1)
int enqueue_fused_back(aisgpu_t* h) {
	if (!h->fpend.valid) return AISGPU_OK;
	h->fpend.valid = false;
	const int q = h->fpend.q, pb = h->fpend.pb, lv = h->fpend.lv, n_groups = h->fpend.n_groups;
	const long long g0 = h->fpend.g0;
	K6Params k6;
	k6.c48 = h->d_c48[q]; k6.c48_stride = h->c48s; k6.ck = h->d_ck[q]; k6.ck_stride = (h->n_chan + 63) / 64 * 64;
	k6.step_table = h->d_step; k6.fz = h->d_fz[q];
	k6.hist_in = h->d_dfhist[pb ^ 1]; k6.hist_out = h->d_dfhist[pb];
	k6.sym = h->d_sym[pb]; k6.sym_stride = h->Gcap; k6.lvl = h->d_lvl[lv];
	memcpy(k6.taps, TAPS_COHERENT, sizeof k6.taps);
	k6.first_group = g0; k6.n_rel0 = h->fpend.n_rel0; k6.n_groups = n_groups; k6.L = h->L; k6.n_windows = h->W; k6.n_chan = h->n_chan;
	const int n_chunks = (n_groups + PS_CHUNK - 1) / PS_CHUNK;
	if (h->k46 && n_chunks > 1) {
		// Round 5: derotation + FIR + ScatterPLL + PhaseSearch as ONE kernel on the PhaseSearch stream (k46_window_search): the FIR outputs
		// stay in LDS, `sym` is neither written nor read (the exact fallback inside k46_assemble materialises its rows when it has to).
		hipStream_t s = h->s1;
		WAITEV(s, h->ev_phasor[q]);
		WAITEV(s, h->ev_sym[pb]); // (sym[pb]: the fallback's rows; last read by the fallback of block f-2, same stream)
		WAITEV(s, h->ev_ema[lv]); // lvl[lv] / bits[lv]: the frame decoder / the copies of block f-4
		{
			K2Params k2r = make_k2(h, q);
			k2r.ck = h->d_ck[q]; k2r.ck_stride = k6.ck_stride;
			TraceScope t(h, "refine", s);
			HIPCHK(launch_k2b_refine(k2r, h->n_chan, s));
		}
		K46Params kq;
		kq.f = k6;
		K4Params& k4 = kq.s;
		k4.sym = h->d_sym[pb]; k4.sym_stride = h->Gcap; k4.bits = h->d_bits[lv]; k4.bits_stride = h->words;
		k4.state_in = h->d_ema[pb]; k4.state_out = h->d_ema[pb ^ 1];
		k4.words = h->d_pswords; k4.ma_start = h->d_psma0; k4.ma_fin = h->d_psma1; k4.fin = h->d_psfin; k4.fb_count = h->d_psflag + 2;
		k4.n_chains = h->n_chains; k4.n_groups = n_groups; k4.n_chunks = n_chunks; k4.warm = h->ps_warm;
		k4.box_in = nullptr; k4.box_out = nullptr; k4.first_group = g0;
		kq.trips_pad = 0;
		{ TraceScope t(h, "fir+psearch", s); HIPCHK(launch_k46(kq, s)); }
		HIPCHK(hipEventRecord(h->ev_c48free[q], s));
		HIPCHK(hipEventRecord(h->ev_k3[pb], s));
		HIPCHK(hipEventRecord(h->ev_sym[pb], s));
		{ int rc = flush_decode(h); if (rc) return rc; } // (dec_defer) the previous block's frame decoders on their stream
		return finish_k4(h, pb, lv, g0, n_groups, h->fpend.block, h->fpend.sub, s);
	}
	WAITEV(h->s4, h->ev_phasor[q]);
	WAITEV(h->s4, h->ev_sym[pb]); // sym[pb] was last read by PhaseSearch of block f-2,
	WAITEV(h->s4, h->ev_ema[lv]); // lvl[lv] by the frame decoder / the copies of block f-4
	if (h->challenger) { // ModelChallenger: the FM branch inside the derotation / FIR kernel (Demod::FM + Filter(Receiver)): bits out, no derotated samples in HBM
		k6.fmbits = h->d_fmbits[pb]; k6.fmbits_stride = h->L / 32;
		memcpy(k6.fm_taps, TAPS_RECEIVER, sizeof k6.fm_taps);
	}
	// (the previous block's regrouping kernel on s1 reads what this launch overwrites: the FM bits of the block before, its "previous")
	if (h->challenger && h->fm_on_s1 && h->fm_ev_used) WAITEV(h->s4, h->ev_fm);
	{
		K2Params k2r = make_k2(h, q);
		k2r.ck = h->d_ck[q]; k2r.ck_stride = k6.ck_stride;
		TraceScope t(h, "refine", h->s4);
		HIPCHK(launch_k2b_refine(k2r, h->n_chan, h->s4));
	}
	{ TraceScope t(h, "derotfir", h->s4); HIPCHK(launch_k6(k6, h->s4)); }
	HIPCHK(hipEventRecord(h->ev_c48free[q], h->s4));
	// Where the device decoders' regrouping of the FM bits runs: behind the derotation / FIR kernel on s4, or (fm_on_s1) in front of
	// PhaseSearch on s1.  On the resampled ladders s4 also carries the spectral analysis and, with the recurrence on s3 in the middle of
	// its chain, is the stream that sets the step (BASELINE configs[2]); s1 has the time.
	hipStream_t fs = h->fm_on_s1 ? h->s1 : h->s4;
	if (h->challenger && h->fm_on_s1) { HIPCHK(hipEventRecord(h->ev_k3[pb], h->s4)); WAITEV(h->s1, h->ev_k3[pb]); }
	if (h->challenger && h->gpu_decode) { // device decoders: the FM bits regrouped per decoder, where this and the previous block's bits are in order
		WAITEV(fs, h->ev_ema[lv ^ 2]); // fmrows[pb] was last read by the decoders of block f-2
		HIPCHK(launch_k7_pack(make_k7(h, pb, lv, g0, n_groups, h->fpend.block, h->fpend.sub), fs));
		if (h->fm_on_s1) { HIPCHK(hipEventRecord(h->ev_fm, h->s1)); h->fm_ev_used = true; }
	}
	if (!(h->challenger && h->fm_on_s1)) HIPCHK(hipEventRecord(h->ev_k3[pb], h->s4));
	{ int rc = flush_decode(h); if (rc) return rc; } // (dec_defer) the previous block's frame decoders, behind this block's derotation / FIR kernel
	WAITEV(h->s1, h->ev_k3[pb]);
	TraceScope t(h, "psearch", h->s1);
	return enqueue_k4(h, pb, lv, g0, n_groups, h->fpend.block, h->fpend.sub, h->s1);
}

#ifndef FRONT_FFT_IN_WAVES
#define FRONT_FFT_IN_WAVES 1
#endif
int enqueue_downstream_fused(aisgpu_t* h, int q, int pb) {
	const int lv = (int)(h->block_idx & 3);
	K2Params k2 = make_k2(h, q);
	const long long g0 = h->n48 / 5, g1 = (h->n48 + h->L) / 5; // groups completed inside this block (DSP/DSP.h:95-117)
	const int n_groups = (int)(g1 - g0), n_rel0 = (int)(g0 * 5 - h->n48);
	k2.ck = h->d_ck[q]; k2.ck_stride = k2.rotT_stride;
	if (h->us_fft_in_k1 || h->front_fft_us) {
		// (resampled ladder: the caller has recorded k1_done[q] on the stream its front-end waves ran on)
	} else if (h->fft_in_k1 || h->front_fft) {
		// the front-end waves have done the whole analysis (k1_fft_tail; k1x_wave / k1k_wave: wave_fft_tail): fz / ppm of this block are there when K1 is
		if (h->front_fft) h->k1_done[q] = nullptr;
		if (!h->k1_done[q]) { HIPCHK(hipEventRecord(h->ev_search[q], h->stream)); h->k1_done[q] = h->ev_search[q]; }
	} else { // FFT + searches follow the front end on its stream (four busy streams are the limit; on s4 in front of this block's refine +
		// derotation / FIR kernels -- tried when the front ends of these ladders became one-wave workgroups, round 6 -- s4 idles through
		// every recurrence: 288 kSPS 1.60 -> 1.83 ms per step, mode X 2.16 -> 2.46: profiles/r06_expK)
		h->k1_done[q] = nullptr;
		{ TraceScope t(h, "fft+search", h->ds); HIPCHK(launch_k2a_fft_search(k2, h->n_chan, h->ds)); }
		HIPCHK(hipEventRecord(h->ev_search[q], h->ds));
	}
	WAITEV(h->s3, h->k1_done[q] ? h->k1_done[q] : h->ev_search[q]);
	WAITEV(h->s3, h->ev_c48free[q]); // ck[q] was last read by K6 of block f-NBUF
	{ TraceScope t(h, "phasor", h->s3); HIPCHK(launch_k2b_ck(k2, h->n_chan, h->s3, h->phasor_simds)); }
	HIPCHK(hipEventRecord(h->ev_phasor[q], h->s3));

	h->fpend.valid = true; h->fpend.q = q; h->fpend.pb = pb; h->fpend.lv = lv; h->fpend.g0 = g0; h->fpend.n_groups = n_groups;
	h->fpend.n_rel0 = n_rel0; h->fpend.block = (unsigned)h->block_idx; h->fpend.sub = (unsigned)h->n_sub;
	if (h->n_sub < MAXSUB) {
		SubOut& so = h->sub[h->n_sub++];
		so.pb = pb; so.lv = lv; so.q = q; so.groups = n_groups; so.first_group = g0; so.first48 = h->n48;
	}
	h->n48 += h->L;
	h->block_idx++;
	if (h->us_on_ds) return AISGPU_OK; // (the second half follows the NEXT flush's resampler front end, or the caller's sync)
	return enqueue_fused_back(h);
}

// Path with the phasor and derotated-sample arrays materialised (taps, Challenger FM branch): everything behind the
// 48 kHz front-end output of one downstream block, ring slot q, parity pb.
//
// Stream plan.  The HBM-bound kernels (front end, FFT, phasor apply, FIR/ScatterPLL) run one after the other on
// the front stream: run side by side they only take bandwidth from each other.  What overlaps them are the
// kernels that want something else: PhaseSearchEMA (VALU-bound, s1), the spectral searches (latency-bound, s4)
// and the phasor recurrence (latency-bound, s3, on CUs of its own).  The recurrence of block f is hidden behind
// the next block's front end by DEFERRING the second half of block f (apply ... PhaseSearchEMA) until the first
// half of block f+1 has been enqueued -- or until the caller asks for results (sync_all / aisgpu_sync_outputs).
// ModelBase (Model.cpp:419-438): the two 48 kHz channels go straight into Demod::FM -> Filter(Receiver); the sign of
// every filtered sample is all that SimplePLL and the decoder look at (DSP.cpp:30, AIS.h:96)
int enqueue_downstream_base(aisgpu_t* h, int q, int pb) {
	K5Params k5;
	k5.x = h->d_c48[q]; k5.x_stride = h->c48s; k5.x_off = 0; k5.prev_in = h->d_fmprev[pb]; k5.prev_out = h->d_fmprev[pb ^ 1];
	k5.fm = h->d_fm; k5.fm_stride = FM_HIST + h->L;
	k5.hist_in = h->d_fmhist[pb]; k5.hist_out = h->d_fmhist[pb ^ 1];
	k5.fmbits = h->d_fmbits[pb]; k5.fmbits_stride = h->L / 32; k5.L = h->L;
	k5.fir_out = h->d_fmfir; k5.fir_stride = h->L;
	memcpy(k5.taps, TAPS_RECEIVER, sizeof k5.taps);
	if (h->ds != h->stream) { // the FM receiver on a stream of its own: behind this block's front end, beside the next one's
		HIPCHK(hipEventRecord(h->ev_pre[q], h->stream));
		WAITEV(h->ds, h->ev_pre[q]);
	}
	if (h->gpu_decode && h->base_chunked) {
		WAITEV(h->ds, h->ev_k4[pb]);       // fmbits[pb] was last read by the boundary tasks of block f-2 (on s1)
		WAITEV(h->ds, h->ev_spec[pb ^ 1]); // and by the speculative pass of block f-1 (its first chunk's warm-up, on s4)
	}
	HIPCHK(launch_k5(k5, h->n_chan, h->ds));
	HIPCHK(hipEventRecord(h->ev_c48free[q], h->ds));
	if (h->gpu_decode) { // SimplePLL + decoder (ModelBase) / Deinterleave + five decoders (ModelStandard) on the device, behind the filter
		const long long g0 = h->n48 / 5, g1 = (h->n48 + h->L) / 5;
		K7Params k7 = make_k7(h, pb, 0, g0, (int)(g1 - g0), (unsigned)h->block_idx, (unsigned)h->n_sub);
		if (h->dec_kind == 1 && h->k7_event && !h->serial && !(h->k7_alt && (h->block_idx & 1))) {
			// ModelStandard's five decoders per channel are ModelDefault's mesh on other bits: the event-driven kernels take the FM
			// rows as their decision rows (no level: tag.sample_lvl is never set in this engine), on PhaseSearch's otherwise idle
			// stream, next to the next block's front end (the sequential mesh kernel held the front stream for 2.5 ms per step)
			WAITEV(h->ds, h->ev_k4[pb]); // fmrows[pb] was last read by the decoders of block f-2
			HIPCHK(launch_k7_pack(k7, h->ds));
			HIPCHK(hipEventRecord(h->ev_sym[pb], h->ds));
			WAITEV(h->s1, h->ev_sym[pb]);
			K7Params kq = k7;
			kq.bits = h->d_fmrows[pb]; kq.bits_stride = h->fmrow_words; kq.lvl = nullptr; kq.kind = 0;
			{ int rc = launch_decoders_event(h, kq, k7, h->s1); if (rc) return rc; }
			HIPCHK(hipEventRecord(h->ev_k4[pb], h->s1));
		} else if (h->dec_kind == 3 && h->base_chunked && !(h->k7_alt && (h->block_idx & 1))) {
			// ModelBase: the chunk-parallel sampler + decoder kernels on PhaseSearch's otherwise idle stream, next to the next block's
			// front end (k7_base alone held the front stream for 5 ms per step of 256 receivers)
			// The speculative pass reads nothing but the FM rows: on s4 (idle in this engine), beside the tasks of the block before.
			// Its scratch set was last read by the tasks of block f-2: this block's FM receiver has waited for them (above).
			K7bParams& kb = h->k7b[pb];
			kb.k = k7;
			// (with the FM receiver on s4 too -- measured in round 4, not kept -- the pass queues behind it there; on s1, in front of its block's tasks, it
			// costs 256 distinct receivers 0.15 ms per step: measured, not kept)
			HIPCHK(hipEventRecord(h->ev_sym[pb], h->ds));
			WAITEV(h->s4, h->ev_sym[pb]);
			HIPCHK(launch_k7b_spec(kb, h->s4));
			HIPCHK(hipEventRecord(h->ev_spec[pb], h->s4));
			WAITEV(h->s1, h->ev_spec[pb]);
			WAITEV(h->s1, h->ev_sym[pb]); // (and behind whatever ran on the FM receiver's stream before: k7_base of an alternating test run)
			HIPCHK(launch_k7b_finish(kb, h->s1));
			HIPCHK(hipEventRecord(h->ev_k4[pb], h->s1));
		} else {
			if ((h->dec_kind == 1 || h->dec_kind == 3) && h->k7_alt) WAITEV(h->ds, h->ev_k4[pb ^ 1]); // (test hook: the previous block's decoders ran on s1)
			if (h->dec_kind == 1) HIPCHK(launch_k7_pack(k7, h->ds));
			HIPCHK(launch_k7_mesh(k7, h->ds));
		}
	}
	HIPCHK(hipEventRecord(h->ev_k3[pb], h->ds));
	WAITEV(h->s2, h->ev_k3[pb]); // aisgpu_sync_outputs copies on s2
	if (h->n_sub < MAXSUB) {
		SubOut& so = h->sub[h->n_sub++];
		so.pb = pb; so.lv = 0; so.q = q; so.groups = 0; so.first_group = h->n48 / 5; so.first48 = h->n48;
	}
	h->n48 += h->L;
	h->block_idx++;
	return AISGPU_OK;
}

// ModelEngineV2 (Model.cpp:440-463): nothing behind the front end runs here; the block's two 48 kHz channels travel to the host
#ifndef V2_FM_BESIDE
#define V2_FM_BESIDE 1
#endif
#ifndef V2_ENGINE_OWN_STREAM
#define V2_ENGINE_OWN_STREAM 0
#endif
int enqueue_downstream_v2(aisgpu_t* h, int q, int pb) {
	if (h->n_sub < MAXSUB) {
		const size_t C = h->n_chan;
		const size_t s_ = (size_t)h->out_set * MAXSUB + h->n_sub; // host slot: two sets, by input block
		const bool on_device = h->gpu_decode; // the whole engine on the device: frames out, the channels never cross PCIe
		if (!on_device)
			HIPCHK(hipMemcpy2DAsync(h->h_c48 + s_ * C * h->L, (size_t)h->L * sizeof(float2), h->d_c48[q], (size_t)h->c48s * sizeof(float2),
			                        (size_t)h->L * sizeof(float2), C, hipMemcpyDeviceToHost, h->ds));
		if (h->v2_assist) { // FreqOffset::Estimate of every offset-0 / offset-256 window, midWins' energies, the FM branch up to its sign
			KV2Params k{};
			k.c48 = h->d_c48[q]; k.c48_stride = h->c48s; k.hist = h->d_v2hist; k.hist_out = h->d_v2hist; k.fmtail_out = nullptr; k.omega = h->d_omega;
			k.est_f = h->d_v2f; k.est_prom = h->d_v2prom; k.energy = h->d_v2en;
			k.disc = h->d_fm; k.disc_full = (h->cfg.flags & AISGPU_FLAG_TAPS) != 0; k.fmprev = h->d_fmprev[0]; k.fmbits = h->d_fmbits[pb]; k.fmbits_stride = h->L / 32;
			k.fir_out = h->d_fmfir; k.fir_stride = h->L;
			memcpy(k.taps, TAPS_RECEIVER, sizeof k.taps);
			k.n_windows = h->W; k.L = h->L; k.n_chan = h->n_chan;
			if (on_device) {
				const int par = h->v2_par; // this block reads pair member `par`; its tail goes to the other one
				h->v2_par ^= 1;
				k.hist = par ? h->d_v2hist2 : h->d_v2hist; k.hist_out = par ? h->d_v2hist : h->d_v2hist2;
				k.est_f = par ? h->d_v2f2 : h->d_v2f; k.est_prom = par ? h->d_v2prom2 : h->d_v2prom; k.energy = par ? h->d_v2en2 : h->d_v2en;
				k.fmtail_out = h->d_v2fmtail[par ^ 1];
				KV2EParams e{};
				e.k = k; e.fm_prev = h->d_v2fmtail[par]; e.st = h->d_v2st; e.dec = h->d_dec; e.slot_cs = h->d_slotcs;
				e.w_train = 0.75f; e.w_track = 0.86f; // PhaseTracker's defaults (V2Engine.h:70-71)
				e.frames = h->d_frames; e.frame_count = h->d_frame_count; e.max_frames = h->max_frames;
				e.block = (unsigned)h->block_idx; e.sub = (unsigned)h->n_sub; e.locked_estimates = h->d_v2locked; e.roles = h->v2_roles;
				memcpy(e.taps17, TAPS_COHERENT, sizeof e.taps17);
				// ds: assist kernels of this block; engine stream: the engine behind them; ds again: the carry, which overwrites what the
				// engine of the PREVIOUS block read (the other pair member, fmbits[pb ^ 1] is next) -- so it waits for that engine, not this one
				if (h->v2_stream != h->ds) WAITEV(h->ds, h->ev_v2engine[par ^ 1]); // (only this block's FRONT END ran beside the previous block's engine)
				bool carried = false;
				if (V2_FM_BESIDE && !h->serial && h->s4 != h->ds) {
					// the FM branch (discriminator + 37-tap filter in one kernel, the half-block energies as its first workgroups) beside the
					// estimates: two kernels that do not fill the chip.  The carry follows the FM branch on s4 (round 6, late): everything it
					// reads is there by then, what it writes -- the OTHER members of the look-back pairs, the FM branch's own carries -- is next
					// read by the assist kernels and the engine of block f+1, and the engine of block f-1, the last reader of those members, was
					// through before this block's front end began (the step is one chain).  So the 5 us kernel and its two launch gaps leave the
					// chain; the next front end waits for it through its table's event, recorded on s4 behind it (aisgpu_run).
					HIPCHK(hipEventRecord(h->ev_v2front, h->ds)); // (the front end of this block is through)
					WAITEV(h->s4, h->ev_v2front);
					HIPCHK(launch_kv2_assist(k, h->s4, 2 | 4));
					HIPCHK(hipEventRecord(h->ev_v2fm, h->s4));
					if (h->v2_stream == h->ds) {
						HIPCHK(launch_kv2_carry(k, h->s4));
						carried = true;
					}
					HIPCHK(launch_kv2_assist(k, h->ds, 1));
					WAITEV(h->ds, h->ev_v2fm);
				} else HIPCHK(launch_kv2_assist(k, h->ds, 7));
				HIPCHK(hipEventRecord(h->ev_v2assist, h->ds));
				WAITEV(h->v2_stream, h->ev_v2assist);
				HIPCHK(launch_kv2_engine(e, h->v2_stream));
				HIPCHK(hipEventRecord(h->ev_v2engine[par], h->v2_stream));
				if (!carried) { // (carried: the next front end waits for its table's event, recorded on s4 behind the carry -- aisgpu_run)
					// ds: the carry, which overwrites what the engine of the PREVIOUS block read (the other pair member, fmbits[pb ^ 1] is next)
					// -- so it waits for that engine, not this one
					WAITEV(h->ds, h->ev_v2engine[par ^ 1]);
					HIPCHK(launch_kv2_carry(k, h->ds));
				}
				HIPCHK(hipEventRecord(h->ev_c48free[q], h->v2_stream)); // (the engine is the block's last reader of its 48 kHz channels)
			} else {
				HIPCHK(launch_kv2(k, h->ds));
				HIPCHK(hipMemcpyAsync(h->h_v2f + s_ * C * 2 * h->W, h->d_v2f, C * 2 * h->W * sizeof(float), hipMemcpyDeviceToHost, h->ds));
				HIPCHK(hipMemcpyAsync(h->h_v2prom + s_ * C * 2 * h->W, h->d_v2prom, C * 2 * h->W * sizeof(float), hipMemcpyDeviceToHost, h->ds));
				HIPCHK(hipMemcpyAsync(h->h_v2en + s_ * C * (h->W + 1), h->d_v2en, C * (h->W + 1) * sizeof(float), hipMemcpyDeviceToHost, h->ds));
				HIPCHK(hipMemcpyAsync(h->h_fmbits + s_ * C * (h->L / 32), h->d_fmbits[pb], C * (h->L / 32) * sizeof(uint32_t), hipMemcpyDeviceToHost, h->ds));
			}
		}
		SubOut& so = h->sub[h->n_sub++];
		so.pb = pb; so.lv = 0; so.q = q; so.groups = 0; so.first_group = h->n48 / 5; so.first48 = h->n48;
	}
	if (!(h->gpu_decode && h->v2_assist)) HIPCHK(hipEventRecord(h->ev_c48free[q], h->ds));
	h->n48 += h->L;
	h->block_idx++;
	return AISGPU_OK;
}

int enqueue_downstream(aisgpu_t* h, int q, int pb) {
	if (h->v2) return enqueue_downstream_v2(h, q, pb);
	if (h->base) return enqueue_downstream_base(h, q, pb);
	if (h->fused) return enqueue_downstream_fused(h, q, pb);
	const K2Params k2 = make_k2(h, q);
	HIPCHK(launch_k2a_fft(k2, h->n_chan, h->ds));
	HIPCHK(hipEventRecord(h->ev_front[q], h->ds));
	// ---- s4: the sequential spectral searches, then on s3 the sequential CGF phasor recurrence (needs fz of this
	// block; rotT[q] was last read by apply(f-NBUF))
	WAITEV(h->s4, h->ev_front[q]);
	HIPCHK(launch_k2a_search(k2, h->n_chan, h->s4));
	HIPCHK(hipEventRecord(h->ev_search[q], h->s4));
	{ int rc = flush_decode(h); if (rc) return rc; } // (dec_defer) the frame decoders of the block before the previous one, behind this block's searches
	WAITEV(h->s3, h->ev_search[q]);
	WAITEV(h->s3, h->ev_c48free[q]);
	HIPCHK(launch_k2b(k2, h->n_chan, h->s3));
	HIPCHK(hipEventRecord(h->ev_phasor[q], h->s3));

	int rc = enqueue_back(h); // the previous block's second half
	if (rc) return rc;
	// ScatterPLL groups completed inside this block (DSP/DSP.h:95-117): group g completes with sample 5g+4
	const long long g0 = h->n48 / 5, g1 = (h->n48 + h->L) / 5;
	h->pend.valid = true; h->pend.q = q; h->pend.pb = pb; h->pend.lv = (int)(h->block_idx & 3); h->pend.g0 = g0; h->pend.g1 = g1; h->pend.first48 = h->n48;
	h->pend.block = (unsigned)h->block_idx; h->pend.sub = (unsigned)h->n_sub;
	if (h->n_sub < MAXSUB) {
		SubOut& s = h->sub[h->n_sub++];
		s.pb = pb; s.lv = (int)(h->block_idx & 3); s.q = q; s.groups = (int)(g1 - g0); s.first_group = g0; s.first48 = h->n48;
	}
	h->n48 += h->L;
	h->block_idx++;
	if (h->serial || !h->defer) return enqueue_back(h);
	return AISGPU_OK;
}

// the frames the device decoders completed since the previous call, in the reference's emission order
int gather_frames(aisgpu_t* h) {
	unsigned total = 0;
	HIPCHK(hipMemcpy(&total, h->d_frame_count, sizeof total, hipMemcpyDeviceToHost));
#ifdef AISGPU_EXPERIMENTS
	if (h->k7_event && opt_int("k7e_stats", 0)) { // experiment aid: events / runs per decoder in the last block
		std::vector<uint32_t> cnt((size_t)h->n_chan * 5);
		HIPCHK(hipMemcpy(cnt.data(), h->d_k7cnt, cnt.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
		unsigned long long se = 0, sr = 0; unsigned me = 0, mr = 0;
		for (uint32_t v : cnt) { se += v & 0xFFFFu; sr += v >> 16; me = std::max(me, v & 0xFFFFu); mr = std::max(mr, v >> 16); }
		fprintf(stderr, "K7E_STATS decoders %zu: events avg %.1f max %u, runs avg %.1f max %u\n", cnt.size(), (double)se / cnt.size(), me, (double)sr / cnt.size(), mr);
	}
	if (h->base_chunked && opt_int("k7b_stats", 0)) { // experiment aid: the boundary tasks of the last block
		const K7bParams& b = h->k7b[(h->block_idx + 1) & 1]; // (the last block's parity)
		std::vector<int> m((size_t)b.n_chunks * b.n_chan_pad);
		HIPCHK(hipMemcpy(m.data(), b.task_merge, m.size() * sizeof(int), hipMemcpyDeviceToHost));
		std::vector<int> fb(b.n_chan_pad);
		HIPCHK(hipMemcpy(fb.data(), b.fallback, fb.size() * sizeof(int), hipMemcpyDeviceToHost));
		long long n = 0, sum = 0; int mx = 0, hist[8] = {0};
		for (int c = 0; c < b.n_chunks; c++)
			for (int ch = 0; ch < h->n_chan; ch++) {
				const int v = m[(size_t)c * b.n_chan_pad + ch];
				if (v < 0) continue;
				const int len = v - c * K7B_CH;
				n++; sum += len; mx = std::max(mx, len);
				hist[len < 128 ? 0 : len < 256 ? 1 : len < 512 ? 2 : len < 1024 ? 3 : len < 2048 ? 4 : len < 4096 ? 5 : len < 8192 ? 6 : 7]++;
			}
		int nfb = 0; for (int ch = 0; ch < h->n_chan; ch++) nfb += fb[ch] != 0;
		fprintf(stderr, "K7B_STATS boundaries %d: tasks %lld (avg %.0f samples, max %d), <128 %d <256 %d <512 %d <1k %d <2k %d <4k %d <8k %d more %d; fallback channels %d\n",
		        b.n_chunks * h->n_chan, n, n ? (double)sum / n : 0.0, mx, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], nfb);
	}
#endif
	const unsigned fresh = total - h->frames_seen;
	h->frames.clear();
	if (fresh > (unsigned)h->max_frames) { h->err = "frame ring overflow: call aisgpu_sync_outputs() more often"; h->frames_seen = total; return AISGPU_ERR_OVERFLOW; }
	if (fresh == 0) return AISGPU_OK;
	const unsigned first = h->frames_seen % (unsigned)h->max_frames;
	const size_t rec = DEC_FRAME_WORDS * sizeof(uint32_t);
	const unsigned n1 = first + fresh <= (unsigned)h->max_frames ? fresh : (unsigned)h->max_frames - first;
	HIPCHK(hipMemcpy(h->h_frames, h->d_frames + (size_t)first * DEC_FRAME_WORDS, n1 * rec, hipMemcpyDeviceToHost));
	if (n1 < fresh) HIPCHK(hipMemcpy(h->h_frames + (size_t)n1 * DEC_FRAME_WORDS, h->d_frames, (fresh - n1) * rec, hipMemcpyDeviceToHost));
	h->frames_seen = total;
	h->frames.resize(fresh);
	for (unsigned i = 0; i < fresh; i++) {
		const uint32_t* f = h->h_frames + (size_t)i * DEC_FRAME_WORDS;
		aisgpu_frame& o = h->frames[i];
		const unsigned dec = f[0];
		if (h->dec_kind == 2) { // ModelChallenger: dec = channel * 10 + position in the reference's order (FM0..FM3, coherent 0..4, FM4)
			const unsigned ord = dec % 10;
			o.rx = (int)(dec / 20); o.ch = (int)(dec / 10 % 2); o.phase = ord < 4 ? 5 + (int)ord : ord == 9 ? 9 : (int)ord - 4; // 5..9: the FM decoders
		} else if (h->dec_kind == 3) { o.rx = (int)(dec / 2); o.ch = (int)(dec % 2); o.phase = 0; } // ModelBase: one decoder per channel
		else if (h->dec_kind == 4) { o.rx = (int)(dec / 12); o.ch = (int)(dec / 6 % 2); o.phase = (int)(dec % 6); } // ModelEngineV2: 0..4 behind the trackers, 5 the FM decoder
		else if (h->mode_x) { o.rx = (int)(dec / 5); o.ch = 0; o.phase = (int)(dec % 5); } // (receivers packed: chain = receiver)
		else { o.rx = (int)(dec / 10); o.ch = (int)(dec / 5 % 2); o.phase = (int)(dec % 5); }
		o.group = (int)f[1]; o.position = (int)f[2];
		memcpy(&o.level_sum, &f[3], 4);
		o.start_idx = (long long)((unsigned long long)f[4] | (unsigned long long)f[5] << 32);
		o.end_idx = (long long)((unsigned long long)f[6] | (unsigned long long)f[7] << 32);
		o.sub = (int)f[9];
		memset(o.data, 0, sizeof o.data);
		memcpy(o.data, &f[10], DEC_DATA_WORDS * 4);
	}
	std::vector<aisgpu_frame> sorted(fresh);
	std::vector<unsigned> idx(fresh);
	for (unsigned i = 0; i < fresh; i++) idx[i] = i;
	// On the decimate-by-3 ladders Rotate hands over DownsampleKFilter's 8192-sample blocks: the reference alternates between the
	// channels every 4096 samples at 48 kHz (and GpuChain::process replays in that order); elsewhere channel A's whole block
	// comes first.  slice = the 4096-sample piece of its downstream block in which the frame closed.
	const auto slice_of = [&](const aisgpu_frame& x) -> long long {
		if (h->rot_period <= 0 || x.sub < 0 || x.sub >= h->n_osub) return 0;
		const SubOut& so = h->osub[x.sub];
		const long long n_rel = h->dec_kind == 3 ? x.group : 5 * (so.first_group + x.group) + (h->dec_kind == 1 ? x.phase : 4) - so.first48;
		return n_rel / 4096;
	};
	std::sort(idx.begin(), idx.end(), [&](unsigned a, unsigned b) {
		const uint32_t* fa = h->h_frames + (size_t)a * DEC_FRAME_WORDS; const uint32_t* fb = h->h_frames + (size_t)b * DEC_FRAME_WORDS;
		const aisgpu_frame &x = h->frames[a], &y = h->frames[b];
		if (x.rx != y.rx) return x.rx < y.rx;
		if (fa[8] != fb[8]) return (int)(fa[8] - fb[8]) < 0;
		const long long sx = slice_of(x), sy = slice_of(y);
		if (sx != sy) return sx < sy;
		if (x.ch != y.ch) return x.ch < y.ch;
		if (h->dec_kind == 4) { if (x.end_idx != y.end_idx) return x.end_idx < y.end_idx; return x.phase < y.phase; } // (group carries tag.ppm there)
		if (x.group != y.group) return x.group < y.group;
		if (h->dec_kind == 2) return fa[0] % 10 < fb[0] % 10; // the reference's order inside a group
		return x.phase < y.phase;
	});
	for (unsigned i = 0; i < fresh; i++) sorted[i] = h->frames[idx[i]];
	h->frames.swap(sorted);
	return AISGPU_OK;
}

int sync_all(aisgpu_t* h) {
	{ int rc = enqueue_back(h); if (rc) return rc; }
	{ int rc = enqueue_fused_back(h); if (rc) return rc; }
	{ int rc = flush_decode(h); if (rc) return rc; }
	HIPCHK(hipStreamSynchronize(h->stream));
	HIPCHK(hipStreamSynchronize(h->s1));
	HIPCHK(hipStreamSynchronize(h->s3));
	HIPCHK(hipStreamSynchronize(h->s4));
	HIPCHK(hipStreamSynchronize(h->s5));
	drain_events(h);
	trace_dump(h);
	return AISGPU_OK;
}

// Resampled ladders, host part (worker thread): replay Upsample::Receive over one input block's n_pre inputs (DSP.cpp:192-212);
// every time `len` outputs are complete the reference flushes them downstream as one Receive() call -> one downstream block here.
// The tables of such a flush -- (input index, alpha) per output, the Rotate phasors -- depend on the stream position alone.
void us_worker_main(aisgpu_t* h) {
	UsWorker& w = h->uw;
	(void)hipSetDevice(h->cfg.device_id);
	const int len = h->n_pre;
	for (long long run = 0;; run++) {
		{
			std::unique_lock<std::mutex> l(w.m);
			w.cv.wait(l, [&] { return w.stop || run < w.taken_runs + 3; }); // at most three input blocks ahead of the caller
			if (w.stop) return;
		}
		const long long in0 = h->us_in;
		int n_flush = 0;
		// The outputs of the flush being filled: input index relative to THIS run's first input (entries of the previous run: negative)
		// and alpha, in plain arrays -- the loop is the float recurrence alpha += increment (DSP.cpp:192-212) and little else.
		if ((int)h->us_pend_idx.size() != len) { h->us_pend_idx.assign(len, 0); h->us_pend_alpha.assign(len, 0.0f); h->us_pend_n = 0; }
		int* const pi = h->us_pend_idx.data();
		float* const pa = h->us_pend_alpha.data();
		int n = h->us_pend_n;
		for (int e = 0; e < n; e++) pi[e] -= len;
		float alpha = h->us_alpha;
		const float inc = h->us_increment;
		for (int i = 0; i < len; i++) {
			do {
				pi[n] = i; pa[n] = alpha; n++;
				alpha += inc;
				if (n == len) {
					const long long k = w.produced_flush;
					const int slot = (int)(k % aisgpu::USR);
					if (k >= aisgpu::USR) { // the pinned buffers of slot k % USR were last read by the copy kernels of flush k - USR
						{
							std::unique_lock<std::mutex> l(w.m);
							w.cv.wait(l, [&] { return w.stop || w.copied_flush > k - aisgpu::USR; });
							if (w.stop) return;
						}
						(void)hipEventSynchronize(h->us_copy_ev[slot]);
					}
					gen_rot_table(h, h->h_usrot[slot]);
					int* ti = h->h_usidx[slot]; float* ta = h->h_usalpha[slot];
					for (int e = 0; e < US_HIST; e++) { // halo: the tail of the previous flush, re-based to this block
						const long long a = h->us_tail_idx[e];
						ti[e] = a < 0 ? -1 : (int)(a - in0);
						ta[e] = h->us_tail_alpha[e];
					}
					memcpy(ti + US_HIST, pi, (size_t)len * sizeof(int));
					memcpy(ta + US_HIST, pa, (size_t)len * sizeof(float));
					for (int e = 0; e < US_HIST; e++) { h->us_tail_idx[e] = in0 + pi[len - US_HIST + e]; h->us_tail_alpha[e] = pa[len - US_HIST + e]; }
					n = 0;
					{ std::lock_guard<std::mutex> l(w.m); w.produced_flush = k + 1; }
					n_flush++;
				}
			} while (alpha < 1.0f);
			alpha -= 1.0f;
		}
		h->us_alpha = alpha;
		h->us_pend_n = n;
		h->us_in += len;
		{ std::lock_guard<std::mutex> l(w.m); w.nflush[run % UsWorker::NRUN] = n_flush; w.produced_runs = run + 1; }
		w.cv.notify_all();
	}
}

void us_worker_stop(aisgpu_t* h) {
	UsWorker& w = h->uw;
	if (!w.started) return;
	{ std::lock_guard<std::mutex> l(w.m); w.stop = true; }
	w.cv.notify_all();
	if (w.th.joinable()) w.th.join();
	w.started = false;
}

// The caller's part: the copies of run `run`'s tables to the device (table stream cs), one input block ahead of the pass over the
// raw input -- by copy kernels that read the pinned buffers through their device view: hipMemcpyAsync from pinned memory behind
// pending kernels costs the calling thread and the stream 0.2 ms per flush here.  Returns the run's number of flushes.
int stage_resample_run(aisgpu_t* h, long long run, int* n_flush_out) {
	UsWorker& w = h->uw;
	if (!w.started) { w.started = true; w.th = std::thread(us_worker_main, h); }
	hipStream_t cs = h->serial ? h->stream : h->s3;
	const int len = h->n_pre;
	int n_flush = 0;
	{
		std::unique_lock<std::mutex> l(w.m);
		w.cv.wait(l, [&] { return w.produced_runs > run; });
		n_flush = w.nflush[run % UsWorker::NRUN];
	}
	// The table ring (USR slots, staged one run ahead) and the output slots (MAXSUB per input block) are sized for at most MAXSUB
	// flushes per run -- every ratio aisgpu_create() accepts stays below (increment >= 1/4): refuse anything else instead of reusing
	// a slot whose tables are still unread.
	static_assert(aisgpu::USR >= 2 * MAXSUB, "resampler table ring: two runs of MAXSUB flushes each");
	if (n_flush > MAXSUB) { h->err = "resampler: more downstream blocks per input block than the table ring holds"; return AISGPU_ERR_STATE; }
	for (int i = 0; i < n_flush; i++) {
		const long long k = w.copied_flush; // (only this thread writes it)
		const int slot = (int)(k % aisgpu::USR);
		if (h->us_slot_used[slot]) WAITEV(cs, h->us_used_ev[slot]); // the device tables of the slot: read by the front end of flush k - USR
		if (h->us_by_kernel) {
			HIPCHK(launch_copy_rows(h->us_dev_rot[slot], 0, h->d_usrot[slot], 0, ROT_HIST + h->n96, 1, cs));
			HIPCHK(launch_copy_rows(h->us_dev_idx[slot], 0, reinterpret_cast<float2*>(h->d_usidx[slot]), 0, (US_HIST + len) / 2, 1, cs));
			HIPCHK(launch_copy_rows(h->us_dev_alpha[slot], 0, reinterpret_cast<float2*>(h->d_usalpha[slot]), 0, (US_HIST + len) / 2, 1, cs));
		} else {
			HIPCHK(hipMemcpyAsync(h->d_usrot[slot], h->h_usrot[slot], ((size_t)ROT_HIST + h->n96) * sizeof(float2), hipMemcpyHostToDevice, cs));
			HIPCHK(hipMemcpyAsync(h->d_usidx[slot], h->h_usidx[slot], ((size_t)US_HIST + len) * sizeof(int), hipMemcpyHostToDevice, cs));
			HIPCHK(hipMemcpyAsync(h->d_usalpha[slot], h->h_usalpha[slot], ((size_t)US_HIST + len) * sizeof(float), hipMemcpyHostToDevice, cs));
		}
		HIPCHK(hipEventRecord(h->us_copy_ev[slot], cs));
		{ std::lock_guard<std::mutex> l(w.m); w.copied_flush = k + 1; }
		w.cv.notify_all();
	}
	{ std::lock_guard<std::mutex> l(w.m); w.taken_runs = run + 1; }
	w.cv.notify_all();
	*n_flush_out = n_flush;
	return AISGPU_OK;
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

int aisgpu_set_option(const char* key, const char* value) {
	if (!key || !*key) return AISGPU_ERR_ARG;
	static const char* const known[] = { "serial", "ps_warm", "ps_sequential", "k7", "k7b_fcap", "fused", "fft_in_k1", "k46", "k1u_spw", "us_k1", "v2_roles", "trace", "k7e_stats", "k7b_stats" };
	bool ok = false;
	for (const char* k : known) ok = ok || strcmp(k, key) == 0;
	if (!ok) return AISGPU_ERR_ARG;
	std::lock_guard<std::mutex> l(g_opt_mtx);
	if (value && *value) g_opt[key] = value; else g_opt.erase(key);
	return AISGPU_OK;
}

int aisgpu_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

long long aisgpu_selftest(int device_id, int which, const void* in, long long n) {
	if (which != 0 || !in || n <= 0 || n > (1ll << 30)) return -AISGPU_ERR_ARG;
	if (hipSetDevice(device_id) != hipSuccess) return -AISGPU_ERR_NODEV;
	float2* d_in = nullptr;
	unsigned* d_cnt = nullptr;
	unsigned cnt = 0;
	hipError_t e = hipMalloc((void**)&d_in, (size_t)n * sizeof(float2));
	if (e == hipSuccess) e = hipMalloc((void**)&d_cnt, sizeof(unsigned));
	if (e == hipSuccess) e = hipMemcpy(d_in, in, (size_t)n * sizeof(float2), hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemset(d_cnt, 0, sizeof(unsigned));
	if (e == hipSuccess) e = launch_selftest_hypot(d_in, (int)n, d_cnt, 0);
	if (e == hipSuccess) e = hipMemcpy(&cnt, d_cnt, sizeof(unsigned), hipMemcpyDeviceToHost);
	hipFree(d_in); hipFree(d_cnt);
	return e == hipSuccess ? (long long)cnt : -AISGPU_ERR_HIP;
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
	// ---- ladder selection, ModelFrontend::buildModel (DSP/Model.cpp:129-338): smallest bucket >= rate
	static const int buckets[8] = { 96000, 192000, 384000, 768000, 1536000, 3072000, 6144000, 12288000 };
	static const float alphas[8] = { 0.0f, -0.8f, -1.1f, -1.2f, -1.2f, -1.5f, -2.0f, -2.0f };
	// the reference's bucket lists also hold 288k and, with `-go DSK on`, 576k / 1152k / 2304k (Model.cpp:129-130)
	static const int buckets3[4] = { 288000, 576000, 1152000, 2304000 };
	const int n3 = (cfg->flags & AISGPU_FLAG_DSK) ? 4 : 1;
	int k = -1, k3 = -1;
	for (int i = 0; i < 8; i++) if (buckets[i] >= cfg->sample_rate) { k = i; break; }
	for (int i = 0; i < n3; i++) if (buckets3[i] >= cfg->sample_rate && (k < 0 || buckets3[i] < buckets[k])) { k3 = i; break; }
	// channel mode X (Model.cpp:35-107): one channel at 48k / 96k / 192k, or resampled into the next of these; 12k .. 192k
	const bool mode_x = (cfg->flags & AISGPU_FLAG_MODE_X) != 0;
	int kx = -1;
	if (mode_x) {
		static const int bx[3] = { 48000, 96000, 192000 };
		// (12k .. 192k like the reference; below 24k the resampler completes up to four downstream blocks per input block:
		// eager_out)
		if (cfg->sample_rate < 12000 || cfg->sample_rate > 192000) return AISGPU_ERR_ARG; // Model.cpp:37-38
		if (cfg->model != AISGPU_MODEL_DEFAULT || (cfg->flags & (AISGPU_FLAG_FP_DS | AISGPU_FLAG_DSK))) return AISGPU_ERR_ARG;
		for (int i = 0; i < 3; i++) if (bx[i] >= cfg->sample_rate) { kx = i; break; }
		k = kx; k3 = -1;
	} else
	if (cfg->sample_rate < 96000 || (k < 0 && k3 < 0)) return AISGPU_ERR_ARG;
	Mode mode; int K, KP;
	if (mode_x) { // the flows of the 96k input (exact bucket) / of the resampler (in between), with the single-channel front end K1x
		static const int bx[3] = { 48000, 96000, 192000 };
		mode = bx[kx] != cfg->sample_rate ? MODE_RESAMPLE : MODE_96K; K = 0; KP = 0;
	} else
	if (k3 >= 0) { // a decimate-by-3 bucket is the smallest one >= rate
		// below the bucket: convert >> DS2.. >> US >> DSK (Model.cpp:213-219 etc.): the resampler flow with the decimate-by-3 front end
		mode = buckets3[k3] != cfg->sample_rate ? MODE_RESAMPLE : MODE_DSK; K = 0; KP = k3; k = 0;
	} else {
		const bool interpolated = buckets[k] != cfg->sample_rate;
		if (!interpolated) {
			if (k == 0) { mode = MODE_96K; K = 0; KP = 0; }
			else if (k <= K_DIRECT_MAX) { mode = MODE_DIRECT; K = k; KP = 0; } // (3072 / 6144 kSPS: five / six stages in the front-end waves since round 5 -- one pass over the input instead of 1 + 1/2 + 1/2 / 1 + 1/4 + 1/4)
			else { mode = MODE_PRE; K = 4; KP = k - 4; }
		} else {
			// the resampler sits two CIC5 stages in front of 96 kHz (k == 2: on the input itself; k == 1, the 192k bucket: one stage)
			mode = MODE_RESAMPLE; K = 0; KP = k >= 2 ? k - 2 : 0;
		}
	}
	// `-go MA on` (Model.cpp:111-126): the moving-average branch comes before the ladders and before FP_DS / DSK are looked at:
	// convert >> DS_MA >> ROT, i.e. the flow of a 96 kSPS input behind an integrate-and-dump pass, Rotate called once per
	// 8192-sample output block of the downsampler (DSP.cpp:60-82)
	int ma_m = 0;
	if ((cfg->flags & AISGPU_FLAG_MA_DS) != 0) {
		if (mode_x) return AISGPU_ERR_ARG;
		if (cfg->sample_rate < 192000 || cfg->sample_rate > 12288000 || cfg->sample_rate % 96000 != 0) return AISGPU_ERR_ARG;
		ma_m = cfg->sample_rate / 96000;
		if (cfg->block_len % ma_m != 0 || (cfg->block_len / ma_m) % 8192 != 0) return AISGPU_ERR_ARG;
		mode = MODE_96K; K = 0; KP = 0; k = 0; k3 = -1;
	}
	const bool by3 = k3 >= 0 && ma_m == 0;
	if (cfg->model != AISGPU_MODEL_DEFAULT && cfg->model != AISGPU_MODEL_CHALLENGER && cfg->model != AISGPU_MODEL_BASE &&
	    cfg->model != AISGPU_MODEL_STANDARD && cfg->model != AISGPU_MODEL_V2) return AISGPU_ERR_ARG;
	if (cfg->input_format != AISGPU_FMT_CU8 && cfg->input_format != AISGPU_FMT_CF32 && cfg->input_format != AISGPU_FMT_CS8 &&
	    cfg->input_format != AISGPU_FMT_CS16) return AISGPU_ERR_ARG;
	if (cfg->n_receivers < 1 || cfg->n_receivers > 65535) return AISGPU_ERR_ARG;
	// a downstream block must be a whole number of 512-sample CGF windows
	const int dec48 = ma_m ? 2 * ma_m : mode_x ? 1 << kx : by3 ? 6 << KP : 2 << k; // input samples per 48 kHz sample (bucket rate)
	if (cfg->block_len < 512 * dec48 || cfg->block_len % (512 * dec48) != 0) return AISGPU_ERR_ARG;
	// DownsampleKFilter hands its output on in blocks of 8192 samples (DSP.h:193), whatever the input block was: only
	// input blocks that are a whole number of them reproduce the reference's call pattern (its file block does)
	if (by3 && cfg->block_len % ((3 * 8192) << KP) != 0) return AISGPU_ERR_ARG;
	if (aisgpu_device_count() <= cfg->device_id || cfg->device_id < 0) return AISGPU_ERR_NODEV;

	aisgpu_t* h = new (std::nothrow) aisgpu();
	if (!h) return AISGPU_ERR_ARG;
	h->cfg = *cfg;
	h->mode = mode; h->K = K; h->KP = KP;
	h->challenger = cfg->model == AISGPU_MODEL_CHALLENGER;
	h->base = cfg->model == AISGPU_MODEL_BASE || cfg->model == AISGPU_MODEL_STANDARD; // both: FM receiver on the 48 kHz channels
	h->v2 = cfg->model == AISGPU_MODEL_V2;
	h->tile96 = 64; // one autonomous wave per workgroup, register/DPP ladder: 64 samples at the kernel's output rate per tile
	h->in_bytes = cfg->input_format == AISGPU_FMT_CF32 ? 8 : cfg->input_format == AISGPU_FMT_CS16 ? 4 : 2;
	// kernel numbering of the formats: 0 = CF32, 1 = CU8, 2 = CS8, 3 = CS16
	h->kfmt = cfg->input_format == AISGPU_FMT_CF32 ? 0 : cfg->input_format == AISGPU_FMT_CU8 ? 1 : cfg->input_format == AISGPU_FMT_CS8 ? 2 : 3;
	if ((cfg->flags & AISGPU_FLAG_FP_DS) && cfg->sample_rate == 1536000 && !ma_m) { // Model.cpp:224-237: only this ladder has a fixed-point twin,
		if (h->kfmt != 1) { delete h; return AISGPU_ERR_ARG; }              // and only ConvertRAW::outCU8 feeds it
		h->kfmt = 4;
	}
	h->n_pre = ma_m ? cfg->block_len / ma_m : cfg->block_len >> KP;
	h->ma_m = ma_m;
	h->us_dsk = by3 && mode == MODE_RESAMPLE;
	if (by3) h->n96 = h->n_pre / 3;
	else if (mode == MODE_RESAMPLE) { h->npost = k >= 2 ? 2 : 1; h->n96 = h->n_pre >> h->npost; } // one flush of n_pre samples at the bucket rate >> KP (384 kHz, or 192 kHz)
	else if (mode == MODE_96K) { h->npost = 0; h->n96 = h->n_pre; }
	else h->n96 = h->n_pre >> K;
	h->mode_x = mode_x;
	h->eager_out = mode_x && cfg->sample_rate < 24000;
	if (mode_x) { h->npost = kx; h->n96 = 2 * (h->n_pre >> kx); } // (no Rotate, no 96 kHz point: n96 only sizes the unused phasor table; L = n96 / 2)
	h->L = h->n96 / 2;
	h->W = h->L / 512;
	h->Gcap = ((h->L + 4) / 5 + 1 + 31) / 32 * 32;
	h->c48s = (long long)h->L + 32;
	h->words = h->Gcap / 32;
	// Channel mode X (round 6): ONE channel per receiver, so the receivers are PACKED -- chain r is receiver r, and everything behind
	// the 48 kHz channels runs over R chains instead of 2 R of which every second one is silence (half of that path's time in rounds
	// 3-5).  The kernels that take the two channels of "a receiver" together (spectral analysis) see pairs of receivers; an odd batch
	// gets one silent row at the end.
	h->n_chan = mode_x ? (cfg->n_receivers + 1) / 2 * 2 : cfg->n_receivers * 2;
	h->n_chains = h->n_chan * 5;
	h->has_fdc = cfg->droop && !by3 && k > 0 ? 1 : 0; // no droop filter on the decimate-by-3 ladders (Model.cpp:207-219) nor at 96 kSPS
	h->alpha = mode_x ? (kx == 2 ? -1.1f : -0.8f) : alphas[k]; // Model.cpp:64,76
	h->beta = 1 - 2 * h->alpha; // DSP/DSP.h:296, evaluated in float
	h->us_increment = (float)cfg->sample_rate / (float)(mode_x ? 48000 << kx : by3 ? buckets3[k3] : buckets[k]); // DSP/DSP.h:172-176
	h->rot_period = by3 || ma_m ? 8192 : 0; // Rotate is called once per DownsampleKFilter / DownsampleMovingAverage output block
	if (K > 0) {
		h->tile_in = h->tile96 << K;
		if (h->n_pre % h->tile_in) { delete h; return AISGPU_ERR_ARG; }
		h->tiles_per_block = h->n_pre / h->tile_in;
		h->tiles_per_span = span_tiles(h->tiles_per_block, cfg->n_receivers, cfg->tiles_per_span);
		h->spans = (h->tiles_per_block + h->tiles_per_span - 1) / h->tiles_per_span;
	}
	// (only the resampled 12288k bucket -- 8 / 10 MSPS -- has five stages in front of the resampler, Model.cpp:166-172: one pass of five stages since round 5;
	// -DKP_PASS_MAX=4 brings back the two passes of rounds 3-4, one stage + four on an intermediate stream of half the input's size)
	if (KP > KP_PASS_MAX) h->KPa = KP - 4;
	if (KP > 0) {
		h->ptile_in = h->tile96 << (h->KPa ? h->KPa : KP);
		if (cfg->block_len % h->ptile_in) { delete h; return AISGPU_ERR_ARG; }
		h->ptiles_per_block = cfg->block_len / h->ptile_in;
		h->ptiles_per_span = span_tiles(h->ptiles_per_block, cfg->n_receivers, cfg->tiles_per_span);
		// Round 6: where the heuristic lands on 24 tiles (the bench batch: 768 tiles x 256 receivers) 32 are better -- one warm-up tile per
		// 32 instead of per 24, 6,144 waves instead of 8,192: BASELINE configs[2] 0.405-0.436 -> 0.395-0.396 ms per step, 48 and 96 lose
		// 10 % and 30 % (profiles/r06_expD_pass_span_length.txt); the main front end has used 32 since round 1.
		if (cfg->tiles_per_span <= 0 && h->ptiles_per_span >= 17 && h->ptiles_per_span < 32 && h->ptiles_per_block % 32 == 0) h->ptiles_per_span = 32;
		h->pspans = (h->ptiles_per_block + h->ptiles_per_span - 1) / h->ptiles_per_span;
	}
	if (mode == MODE_RESAMPLE) h->xh = 0; // (ring of three input blocks instead of a copied history)
	if (mode == MODE_DSK || mode == MODE_96K) h->xh = DSK_HIST;
	*out = h; // from here on the caller destroys it on failure

	HIPCHK(hipSetDevice(cfg->device_id));
	const std::string k7_opt = opt_str("k7"); // test hook: "seq" = the symbol-by-symbol decoder kernels, "alt" = both implementations take turns
	if ((cfg->flags & AISGPU_FLAG_SERIAL) || opt_int("serial", 0)) { // profiling aid: no cross-block overlap, every kernel runs alone
		HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
		h->s1 = h->s2 = h->s3 = h->s4 = h->s5 = h->stream;
		h->serial = true;
	} else {
		// The chip dispatches from four compute queues at a time, so the pipeline has exactly four busy streams:
		// front end, phasor recurrence, derotation / FIR, PhaseSearch.
		//
		// The CGF phasor recurrence (s3) is a handful of latency-bound waves and the longest dependency chain of the
		// pipeline; sharing a SIMD with throughput kernels more than doubles its run time (every foreign VALU
		// instruction delays its next dependent one).  It therefore gets CUs of its own: CU-mask bits 0..7 are one
		// CU in each of the 8 XCDs (tools/microbench_cumask.hip), 3% of the chip; the back-end streams use the rest.
		// The front stream is a plain stream of the LOWEST queue priority: the workgroup dispatcher then prefers the back end's
		// workgroups whenever a slot frees up (2-3 % per step, profiles/r02_expI.txt .. r02_expK.txt).
		int n_cu = 0;
		HIPCHK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, cfg->device_id));
#ifndef AISGPU_RESERVE_CUS
#define AISGPU_RESERVE_CUS 8
#endif
		const int reserve = AISGPU_RESERVE_CUS;
		// Where the frame decoders run.  A fifth active stream shares a pipe with one of the others and their kernels take turns
		// (0.59 - 0.62 ms per step with ModelDefault's event-driven decoders on a stream of their own, whatever those kernels cost).
		// So the event-driven decoders share the derotation / FIR stream, a block late (dec_defer): 0.52 ms.  Only the sequential
		// decoder kernels of the other engines, which run for a whole step, get a stream of their own.
		const bool evt = k7_opt != "seq" && (h->L + 4) / 5 + 1 <= 8191; // (ModelChallenger's mesh of ten: event-driven for blocks of at most 8191 groups)
		const bool dec_on_fir_stream = (cfg->model != AISGPU_MODEL_STANDARD && cfg->model != AISGPU_MODEL_CHALLENGER && cfg->model != AISGPU_MODEL_BASE) ||
		                               (cfg->model == AISGPU_MODEL_CHALLENGER && evt);
		const bool own_dec_stream = (cfg->flags & AISGPU_FLAG_GPU_DECODE) && !dec_on_fir_stream;
		bool masked = false;
		if (n_cu >= 64) {
			const int words = (n_cu + 31) / 32;
			std::vector<uint32_t> lat(words, 0u), rest(words, 0u);
			for (int i = 0; i < n_cu; i++) (i < reserve ? lat : rest)[i / 32] |= 1u << (i % 32);
			int least = 0, greatest = 0;
			const bool prio = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest; // (the front stream at the lowest queue priority: the dispatcher prefers the back end's workgroups whenever a slot frees up)
			// (a runtime without CU masks falls back to ordinary streams: same results, the recurrence just shares its SIMDs)
						masked = (prio ? hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, least) : hipExtStreamCreateWithCUMask(&h->stream, (uint32_t)words, rest.data())) == hipSuccess;
			masked = masked && hipExtStreamCreateWithCUMask(&h->s1, (uint32_t)words, rest.data()) == hipSuccess;
			masked = masked && hipExtStreamCreateWithCUMask(&h->s3, (uint32_t)words, lat.data()) == hipSuccess;
			masked = masked && hipExtStreamCreateWithCUMask(&h->s4, (uint32_t)words, rest.data()) == hipSuccess;
			if (own_dec_stream) masked = masked && hipExtStreamCreateWithCUMask(&h->s5, (uint32_t)words, rest.data()) == hipSuccess;
		}
		if (!masked) {
			(void)hipGetLastError();
			hipStream_t* all[5] = { &h->stream, &h->s1, &h->s3, &h->s4, &h->s5 };
			for (auto ps : all) { if (*ps) hipStreamDestroy(*ps); *ps = nullptr; }
			HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
			HIPCHK(hipStreamCreateWithFlags(&h->s1, hipStreamNonBlocking));
			HIPCHK(hipStreamCreateWithFlags(&h->s3, hipStreamNonBlocking));
			HIPCHK(hipStreamCreateWithFlags(&h->s4, hipStreamNonBlocking));
			if (own_dec_stream) HIPCHK(hipStreamCreateWithFlags(&h->s5, hipStreamNonBlocking));
		}
		h->phasor_simds = (masked ? reserve : n_cu) * 4;
		if (!h->s5) h->s5 = h->s4;
		h->s2 = h->s1; // apply + FIR + PhaseSearchEMA of a block run back to back on one stream
		h->dec_defer = h->s5 == h->s4 && h->s4 != h->stream;
	}
	// ModelBase / ModelStandard (round 4, late): the FM receiver leaves the front stream as well -- front end 0.34 + FM receiver 0.10 ms
	// one behind the other WERE these engines' step (0.467 ms; with the decoders on the device 0.497), whatever the decoders cost
	h->ds = (!h->serial && ((h->mode == MODE_RESAMPLE && h->KP > 0) || h->base)) ? h->s4 : h->stream;
	for (int i = 0; i < NBUF; i++) HIPCHK(hipEventCreateWithFlags(&h->ev_pre[i], hipEventDisableTiming));
	HIPCHK(hipEventCreateWithFlags(&h->ev_fm, hipEventDisableTiming));
	h->fm_on_s1 = !h->serial && h->mode == MODE_RESAMPLE; // ModelChallenger's FM bits regrouped in front of PhaseSearch on ITS stream where s4 carries the analysis too (resampled ladders)
	for (int i = 0; i < NBUF; i++) {
		HIPCHK(hipEventCreateWithFlags(&h->ev_front[i], hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&h->ev_phasor[i], hipEventDisableTiming));
		HIPCHK(hipEventCreate(&h->ev_search[i])); // (may be bound to a launch as its stop event)
		HIPCHK(hipEventCreateWithFlags(&h->ev_c48free[i], hipEventDisableTiming));
	}
	for (int i = 0; i < 4; i++) HIPCHK(hipEventCreateWithFlags(&h->ev_ema[i], hipEventDisableTiming));
	for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&h->ev_sym[i], hipEventDisableTiming));
	for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&h->ev_k3[i], hipEventDisableTiming));
	for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&h->ev_k4[i], hipEventDisableTiming));
#ifdef AISGPU_EXPERIMENTS
	h->trace = opt_int("trace", 0) != 0; // kernel timeline from HIP events on stderr
#endif

	// ---- constant tables (host libm, like the reference on this machine)
	{
		float angle = (float)((double)PI_F * 25000.0 / 48000.0); // Model.cpp:31
		h->mult = make_float2(cosf(angle), sinf(angle));         // std::polar(1.0f, angle), DSP.h:311
		h->rot_tail.assign(ROT_HIST, make_float2(1.0f, 0.0f));
		h->us_tail_idx.assign(US_HIST, -1); h->us_tail_alpha.assign(US_HIST, 0.0f); // zero signal before the stream starts
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
		if (mode_x) { // "channel B" of aisgpu_fetch: zero decisions / levels, the ppm of a window without a peak (fz = -1, DSP.cpp:449-456, 484)
			h->h_silent.assign((size_t)h->Gcap + 64, 0.0f);
			h->h_silent_ppm.assign((size_t)h->W + 1, ppm[-1 - FZ_MIN]);
		}
	}
	const size_t R = cfg->n_receivers, C = h->n_chan;
	// history of the raw input: the last tile of the previous block (for the first kernel that touches the input)
	{
		const size_t first_tile = KP > 0 ? h->ptile_in : (h->tile_in > 0 ? h->tile_in : 64);
		for (int i = 0; i < 2; i++) {
			HIPCHK(dalloc((unsigned char**)&h->d_hist[i], R * first_tile * h->in_bytes));
			// zero signal before the stream starts: CU8 zero is the byte 128 (Utilities/Convert.cpp:255-264)
			// (the fixed-point ladder starts from h0..h4 = 0, DSP.h:397: bytes of zero)
			if (cfg->input_format == AISGPU_FMT_CU8 && h->kfmt != 4) HIPCHK(hipMemset(h->d_hist[i], 0x80, R * first_tile * h->in_bytes));
		}
	}
	if (KP > 0 || mode == MODE_DSK || mode == MODE_RESAMPLE || mode == MODE_96K) {
		h->x_direct = KP == 0 && (mode == MODE_DSK || mode == MODE_96K) && h->kfmt == 0 && !h->ma_m && h->xh <= h->n_pre;
		if (h->x_direct) for (int i = 0; i < 2; i++) HIPCHK(dalloc(&h->d_xhist[i], R * (size_t)h->xh)); // zero = silence before the stream
		const int nx = h->x_direct ? 0 : mode == MODE_RESAMPLE ? XR : (mode == MODE_DSK || mode == MODE_96K) ? 2 : 1;
		for (int i = 0; i < nx; i++) HIPCHK(dalloc(&h->d_xpre[i], R * ((size_t)h->xh + h->n_pre)));
		if (mode == MODE_PRE) for (int i = 0; i < 2; i++) HIPCHK(dalloc((unsigned char**)&h->d_hist2[i], R * h->tile_in * 8));
		if (h->KPa) { // second pre-decimation pass: four stages on the CF32 stream of the first
			if ((cfg->block_len >> h->KPa) % (h->tile96 << 4)) return AISGPU_ERR_ARG;
			HIPCHK(dalloc(&h->d_xmid, R * (size_t)(cfg->block_len >> h->KPa)));
			for (int i = 0; i < 2; i++) HIPCHK(dalloc((unsigned char**)&h->d_hist2[i], R * (size_t)(h->tile96 << 4) * 8));
		}
	}
	for (int i = 0; i < 4; i++) {
		HIPCHK(dalloc(&h->d_rot[i], (size_t)ROT_HIST + h->n96));
		HIPCHK(hipHostMalloc((void**)&h->h_rot[i], ((size_t)ROT_HIST + h->n96) * sizeof(float2), hipHostMallocDefault));
		HIPCHK(hipEventCreateWithFlags(&h->rot_ev[i], hipEventDisableTiming));
		if (hipHostGetDevicePointer((void**)&h->h_rot_dev[i], h->h_rot[i], 0) != hipSuccess) h->rot_by_kernel = false;
	}
	if (mode == MODE_RESAMPLE) {
		for (int i = 0; i < XR; i++) { HIPCHK(hipEventCreateWithFlags(&h->ev_xin[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&h->ev_xread[i], hipEventDisableTiming)); }
		for (int i = 0; i < aisgpu::USR; i++) {
			HIPCHK(hipEventCreateWithFlags(&h->us_copy_ev[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&h->us_used_ev[i], hipEventDisableTiming));
			HIPCHK(dalloc(&h->d_usidx[i], (size_t)US_HIST + h->n_pre));
			HIPCHK(dalloc(&h->d_usalpha[i], (size_t)US_HIST + h->n_pre));
			HIPCHK(dalloc(&h->d_usrot[i], (size_t)ROT_HIST + h->n96));
			HIPCHK(hipHostMalloc((void**)&h->h_usidx[i], ((size_t)US_HIST + h->n_pre) * sizeof(int), hipHostMallocDefault));
			HIPCHK(hipHostMalloc((void**)&h->h_usalpha[i], ((size_t)US_HIST + h->n_pre) * sizeof(float), hipHostMallocDefault));
			HIPCHK(hipHostMalloc((void**)&h->h_usrot[i], ((size_t)ROT_HIST + h->n96) * sizeof(float2), hipHostMallocDefault));
			if (hipHostGetDevicePointer((void**)&h->us_dev_idx[i], h->h_usidx[i], 0) != hipSuccess || hipHostGetDevicePointer((void**)&h->us_dev_alpha[i], h->h_usalpha[i], 0) != hipSuccess ||
			    hipHostGetDevicePointer((void**)&h->us_dev_rot[i], h->h_usrot[i], 0) != hipSuccess || ((US_HIST + h->n_pre) & 1)) h->us_by_kernel = false;
		}
	}
	h->gpu_decode = (cfg->flags & AISGPU_FLAG_GPU_DECODE) != 0;
	if (h->gpu_decode) {
		// ModelEngineV2 (4): six decoders per channel inside the engine kernel (kv2_engine): tone gate, derotation, trackers, decoders
		h->dec_kind = cfg->model == AISGPU_MODEL_STANDARD ? 1 : cfg->model == AISGPU_MODEL_CHALLENGER ? 2 : cfg->model == AISGPU_MODEL_BASE ? 3 : cfg->model == AISGPU_MODEL_V2 ? 4 : 0;
		if (h->dec_kind == 4) {
			// std::polar in V2::FreqOffset::Derotate is the HOST's sinf / cosf in the reference; the device restates glibc 2.35's FMA variant.
			// Where the host's libm computes anything else (another libm, a CPU without FMA) the engine on the device would silently differ
			// from the reference on this host in the last bit of a phasor: refuse instead (the host-engine path has no such dependency).
			static const bool libm_ok = sincos_restatement_matches_host_libm(); // (3.6 M arguments, once per process)
			if (!libm_ok) { h->err = "AISGPU_FLAG_GPU_DECODE with ModelEngineV2: the host's sinf / cosf differ from the device's restatement (glibc 2.35, FMA variant); run the engine on the host (without the flag)"; return AISGPU_ERR_ARG; }
			std::vector<V2ChanState> init(C);
			memset(init.data(), 0, C * sizeof(V2ChanState));
			for (size_t i = 0; i < C; i++) init[i].rot = make_float2(1.0f, 0.0f); // FreqOffset::rot (V2Engine.h:34)
			HIPCHK(dalloc(&h->d_v2st, C));
			HIPCHK(hipMemcpy(h->d_v2st, init.data(), C * sizeof(V2ChanState), hipMemcpyHostToDevice));
			std::vector<float2> cs(1280); // learnSlotPhase (V2Engine.cpp:331-333): th = (float)m * (2 pi / SLOT), the host's cosf / sinf
			for (int m = 0; m < 1280; m++) { const float th = (float)m * (2.0f * PI_F / 1280); cs[m] = make_float2(cosf(th), sinf(th)); }
			HIPCHK(dalloc(&h->d_slotcs, 1280));
			HIPCHK(hipMemcpy(h->d_slotcs, cs.data(), 1280 * sizeof(float2), hipMemcpyHostToDevice));
			HIPCHK(dalloc(&h->d_v2locked, 1));
		}
		h->max_frames = (int)C * 64; // ring between two aisgpu_sync_outputs(): a slot holds ~2 frames per channel and block
		HIPCHK(dalloc(&h->d_dec, (size_t)C * 10)); // zero = State::TRAINING, lastBit = prev = 0 (Marine/AIS.h:44-56); up to ten decoders per channel
		// chunk-parallel sampler + decoder loop (kernels.h: K7b); "seq" / blocks of unusual length: k7_base alone
		if (cfg->model == AISGPU_MODEL_BASE && k7_opt != "seq" && (h->L + K7B_CH - 1) / K7B_CH <= K7B_MAXC && h->L < 65534 && h->L % 32 == 0 && h->L >= K7B_WARM) {
			for (int i = 0; i < 2; i++) {
				K7bParams& b = h->k7b[i];
				b.n_chunks = (h->L + K7B_CH - 1) / K7B_CH;
				b.n_chan_pad = (int)((C + 63) / 64 * 64);
				{ const int v = opt_int("k7b_fcap", 0); if (v >= 1 && v <= K7B_FCAP) b.fcap = v; } // test hook: small values force the exact fallback (k7_base)
				const size_t slots = (size_t)b.n_chunks * b.n_chan_pad;
				HIPCHK(dalloc(&b.ckpt, slots * (K7B_CH / 32)));
				HIPCHK(dalloc(&b.end, slots));
				HIPCHK(dalloc(&b.frames, slots * (1 + K7B_FCAP * K7B_FREC)));
				HIPCHK(dalloc(&b.fallback, (size_t)b.n_chan_pad + 1)); b.fallback_count = b.fallback + b.n_chan_pad;
				HIPCHK(dalloc(&b.sum_spec, (size_t)b.n_chan_pad * K7B_MAXC));
				if (i == 0) { // (the tasks' own scratch: one block at a time)
					HIPCHK(dalloc(&b.task_end, slots));
					HIPCHK(dalloc(&b.task_frames, slots * (1 + K7B_FCAP * K7B_FREC)));
					HIPCHK(dalloc(&b.task_merge, slots));
					HIPCHK(dalloc(&b.take_spec, slots)); HIPCHK(dalloc(&b.take_task, slots));
					HIPCHK(dalloc(&b.sum_task, (size_t)b.n_chan_pad * K7B_MAXC));
					HIPCHK(dalloc(&b.out_base, (size_t)b.n_chan_pad)); HIPCHK(dalloc(&b.fin_sel, (size_t)b.n_chan_pad));
				} else { b.task_end = h->k7b[0].task_end; b.task_frames = h->k7b[0].task_frames; b.task_merge = h->k7b[0].task_merge; b.take_spec = h->k7b[0].take_spec; b.take_task = h->k7b[0].take_task; b.sum_task = h->k7b[0].sum_task; b.out_base = h->k7b[0].out_base; b.fin_sel = h->k7b[0].fin_sel; }
				HIPCHK(hipEventCreateWithFlags(&h->ev_spec[i], hipEventDisableTiming));
			}
			h->base_chunked = true;
			// (The FM receiver stays on the front stream; on s4 in front of the speculative pass -- ds = s4 -- the step is the same.)
		}
		if (h->dec_kind == 1 || h->dec_kind == 2) {
			h->fmrow_words = h->Gcap / 32;
			for (int i = 0; i < 2; i++) HIPCHK(dalloc(&h->d_fmrows[i], C * 5 * (size_t)h->fmrow_words));
			for (int i = 0; i < 2; i++) HIPCHK(dalloc(&h->d_last_lvl[i], C));
		}
		if (h->dec_kind == 4) h->dec_defer = false; // (ModelEngineV2: the decoders run inside the engine kernel)
		if (h->dec_kind > 2) h->k7_event = false; // the event-driven form: ModelDefault's wiring, ModelStandard's (the same mesh of five on the FM rows), ModelChallenger's mesh of ten
		if (h->dec_kind == 1 || h->dec_kind == 3) h->dec_defer = false; // (ModelStandard / ModelBase: their decoders are enqueued by the FM receiver's own flow)
		// (on the decimate-by-3 ladders Rotate alternates between the channels every 4096 samples, and with it the level the FM
		// decoders of ModelChallenger inherit through the shared TAG: that variant of the mesh kernel does not exist)
		if (h->dec_kind == 2 && (by3 || ma_m)) { h->err = "AISGPU_FLAG_GPU_DECODE with ModelChallenger: not on the decimate-by-3 ladders / behind the moving-average downsampler"; return AISGPU_ERR_ARG; }
		HIPCHK(dalloc(&h->d_frames, (size_t)h->max_frames * DEC_FRAME_WORDS));
		if (!k7_opt.empty()) { h->k7_event = h->dec_kind <= 2 && k7_opt != "seq"; h->k7_alt = k7_opt == "alt"; } // "seq": one lane per decoder, symbol by symbol
		// the event words hold a symbol index in 13 bits: blocks of more than 8191 groups (e.g. the reference's CU8 file block of
		// 3,145,728 samples at 1536 kSPS) go through the sequential decoder kernel
		if ((h->L + 4) / 5 + 1 > 8191) h->k7_event = false;
		if (h->k7_event) {
			const size_t n_dec = (size_t)h->n_chains * (h->dec_kind == 2 ? 2 : 1); // (ModelChallenger: ten decoders per channel)
			HIPCHK(dalloc(&h->d_k7ev, n_dec * K7E_EVCAP));
			HIPCHK(dalloc(&h->d_k7cnt, n_dec));
			HIPCHK(dalloc(&h->d_k7open, n_dec * K7E_OPENCAP));
			HIPCHK(dalloc(&h->d_k7slot, n_dec * K7E_OPENCAP));
			HIPCHK(dalloc(&h->d_k7ovf, 4));
		}
		HIPCHK(dalloc(&h->d_frame_count, 1));
		HIPCHK(hipHostMalloc((void**)&h->h_frames, (size_t)h->max_frames * DEC_FRAME_WORDS * sizeof(uint32_t), hipHostMallocDefault));
	}
	h->ps_box = (cfg->flags & AISGPU_FLAG_PS_BOXCAR) != 0;
	if (h->ps_box) for (int i = 0; i < 2; i++) HIPCHK(dalloc(&h->d_box[i], (size_t)h->n_chains));
	// default: the fused derotation + FIR path (no phasor array in HBM, and no derotated-sample array either -- except for
	// ModelChallenger, whose FM branch demodulates those samples: there the fused kernel stores them on its way); the materialised
	// path serves the taps, which need those arrays, and stays selectable (option "fused" = 0: test hook)
	// (round 4: also ModelChallenger on the resampled ladders -- BASELINE configs[2], 6 MSPS: with the lanes-over-time derotation / FIR
	// kernel 0.558 -> 0.538 ms per step; with round 3's lane-per-chain kernel it had been 0.70 against 0.53)
	h->fused = !(cfg->flags & AISGPU_FLAG_TAPS) && !h->base && !h->v2 && opt_int("fused", 1) != 0;
	// Measured (profiles/r05_expA_k46.txt): bit-exact, 240 MB less traffic per step, and 5 % SLOWER per step -- so it is off unless asked for
	h->k46 = h->fused && !h->challenger && !(cfg->flags & AISGPU_FLAG_PS_BOXCAR) && opt_int("k46", 0) != 0 && opt_int("ps_sequential", 0) == 0;
	h->us_on_ds = !h->serial && h->fused && h->mode == MODE_RESAMPLE && h->ds != h->stream;
	// The spectral analysis rides at the end of the front-end waves (k1_fft_tail) when every span is a whole number of 512-sample
	// windows of the 48 kHz channels (16 tiles each) and whole spans make up the block; the automatic span length is rounded up
	// to such a value, an explicit one (cfg.tiles_per_span) is taken as it is.  Option "fft_in_k1" = 0 (test hook): the FFT / search kernels.
	h->fft_in_k1 = h->fused && (mode == MODE_DIRECT || mode == MODE_PRE) && opt_int("fft_in_k1", 1) != 0;
	if (h->fft_in_k1) {
		int t = h->tiles_per_span;
		if (cfg->tiles_per_span <= 0) {
			// two windows per channel and span where the block allows it: one warm-up tile per 32 instead of per 16 tiles (3 % less
			// front-end work); round 1 measured 16 < 32 < 48, round 2 with the non-temporal input stream 32 < 16 < 48 (r02_expJ.txt)
			t = (h->tiles_per_block % 32 == 0 && h->tiles_per_block >= 64) ? 32 : 16;
			while (t <= h->tiles_per_block && h->tiles_per_block % t) t += 16;
		}
		if (t >= 16 && t <= h->tiles_per_block && t % 16 == 0 && h->tiles_per_block % t == 0) {
			h->tiles_per_span = t;
			h->spans = h->tiles_per_block / t;
		} else h->fft_in_k1 = false;
	}
	h->us_k1 = mode == MODE_RESAMPLE && h->npost == 2 && !h->mode_x && !h->us_dsk && h->n_pre % (16 * 256) == 0 && opt_int("us_k1", 1) != 0;
	if (h->us_k1) { // tiles of 256 Upsample outputs (64 samples at 96 kHz); spans of whole 512-sample windows (16 tiles), two where the flush allows it
		h->us_tiles_per_block = h->n_pre / 256;
		int t = (h->us_tiles_per_block % 32 == 0 && h->us_tiles_per_block >= 64) ? 32 : 16;
		if (cfg->tiles_per_span >= 16 && cfg->tiles_per_span % 16 == 0 && h->us_tiles_per_block % cfg->tiles_per_span == 0) t = cfg->tiles_per_span;
		h->us_tiles_per_span = t;
		h->us_fft_in_k1 = h->fused && opt_int("fft_in_k1", 1) != 0;
	}
	if (h->fused) {
		for (int i = 0; i < NBUF; i++) HIPCHK(dalloc(&h->d_ck[i], (C + 63) / 64 * 64 * (size_t)h->W * CK_SLOTS)); // phasor checkpoints per (window, slot, chain)
		for (int i = 0; i < 2; i++) HIPCHK(dalloc(&h->d_dfhist[i], C * DF_HIST)); // zero = silence before the stream
	}
	// the FFT magnitudes and the per-sample phasors exist only on the materialised path (taps, option fused = 0): 0.6 GB per 256 receivers
	const bool materialised = !h->fused && !h->base && !h->v2;
	for (int i = 0; i < NBUF; i++) {
		if (materialised) HIPCHK(dalloc(&h->d_magT[i], (C * h->W + 63) / 64 * (size_t)(512 * 64)));
		HIPCHK(dalloc(&h->d_c48[i], C * h->c48s + 64)); // + over-read slack of the fused FIR kernel's last segment
		HIPCHK(dalloc(&h->d_fz[i], C * h->W));
		HIPCHK(dalloc(&h->d_ppm[i], C * h->W));
		if (materialised) HIPCHK(dalloc(&h->d_rotT[i], (size_t)h->L * ((C + 63) / 64 * 64)));
	}
	for (int i = 0; i < 4; i++) { HIPCHK(dalloc(&h->d_lvl[i], C * h->Gcap)); HIPCHK(dalloc(&h->d_bits[i], C * 5 * h->words)); }
	for (int i = 0; i < 2; i++) {
		HIPCHK(dalloc(&h->d_sym[i], sym_elems((int)C, h->Gcap))); // SymRow layout (kernels.h): channels padded to 64
		HIPCHK(dalloc(&h->d_ema[i], C * 5));
	}
	if (!h->fused) HIPCHK(dalloc(&h->d_cgf, C * (CGF_HIST + h->L))); // (the materialised path only: the fused back end keeps the derotated samples on chip)
	if (h->base) for (int i = 0; i < 2; i++) HIPCHK(dalloc(&h->d_fmprev[i], C)); // Demod::FM::prev = 0 (Demod.h)
	if (h->challenger || h->base) {
		if (cfg->flags & AISGPU_FLAG_TAPS) HIPCHK(dalloc(&h->d_fm, C * (FM_HIST + h->L))); // (the discriminator output is a tap only)
		for (int i = 0; i < 2; i++) HIPCHK(dalloc(&h->d_fmhist[i], C * FM_HIST)); // zero: DSP::Filter starts on zeros
		for (int i = 0; i < 2; i++) HIPCHK(dalloc(&h->d_fmbits[i], C * (h->L / 32)));
		HIPCHK(hipHostMalloc((void**)&h->h_fmbits, MAXSUB * C * (h->L / 32) * sizeof(uint32_t), hipHostMallocDefault));
		if (cfg->flags & AISGPU_FLAG_TAPS) HIPCHK(dalloc(&h->d_fmfir, C * (size_t)h->L));
	}
	HIPCHK(dalloc(&h->d_rotstate, C));
	{
		std::vector<float2> ones(C, make_float2(1.0f, 0.0f)); // SquareFreqOffsetCorrection::rot = 1.0f (DSP.h:379)
		HIPCHK(hipMemcpy(h->d_rotstate, ones.data(), C * sizeof(float2), hipMemcpyHostToDevice));
	}
	const int ps_chunks = (h->Gcap + PS_CHUNK - 1) / PS_CHUNK;
	{ const int v = opt_int("ps_warm", 0); if (v >= 1 && v <= PS_CHUNK) h->ps_warm = (v + 15) / 16 * 16; } // test hook: small values force the exact fallback
	{ const int v = opt_int("k1u_spw", 0); if (v == 2 || v == 4 || v == 8) h->k1u_spw = v; }
	if (opt_int("ps_sequential", 0)) h->ps_parallel = false; // test hook: the plain sequential row kernel
	const size_t n_ma = C * 5 * ps_chunks * 16;
	HIPCHK(dalloc(&h->d_pswords, C * 5 * ps_chunks * (PS_CHUNK / 32) * 16));
	HIPCHK(dalloc(&h->d_psma0, n_ma));
	HIPCHK(dalloc(&h->d_psma1, n_ma));
	HIPCHK(dalloc(&h->d_psfin, C * 5 * ps_chunks * 16));
	HIPCHK(dalloc(&h->d_psflag, 4));
	if (cfg->flags & AISGPU_FLAG_TAPS) HIPCHK(dalloc(&h->d_firtap, C * (8 + h->L)));
	if (h->v2) {
		// kv2_engine_roles: three waves per channel.  Up to 512 channels every wave has a SIMD of its own and the kernel runs as it compiles
		// (217 registers, two waves per SIMD: 1); bigger batches run the instance compiled for 168 registers (three waves per SIMD, four
		// workgroups of 35 KB of LDS per CU: 2) -- 256 distinct receivers 75.1 / 73.6 GS/s, 512: 77.2 / 80.6, 1,024: 76.5 / 96.2 (round 5's
		// one-wave kernel: 32 / 53 / 90; profiles/r06_expG_v2_engine.txt).  Test hook "v2_roles" = 0 / 1 / 2 forces one of the three.
		h->v2_roles = opt_int("v2_roles", h->n_chan <= 512 ? 1 : 2);
		// With the engine on the device (AISGPU_FLAG_GPU_DECODE) nothing but frames goes to the host: no pinned slots for the channels,
		// the estimates, the energies or the discriminator signs (several GB at 2,048 receivers), and aisgpu_fetch_sub() returns NULL for them.
		const bool v2_host = !h->gpu_decode;
		if (v2_host) HIPCHK(hipHostMalloc((void**)&h->h_c48, 2 * MAXSUB * C * h->L * sizeof(float2), hipHostMallocDefault)); // (two sets of slots: see out_set)
		if (h->v2_assist) {
			HIPCHK(dalloc(&h->d_v2hist, C * V2_HIST));
			HIPCHK(dalloc(&h->d_v2f, C * 2 * h->W)); HIPCHK(dalloc(&h->d_v2prom, C * 2 * h->W)); HIPCHK(dalloc(&h->d_v2en, C * (h->W + 1)));
			if (!v2_host) {
				HIPCHK(dalloc(&h->d_v2hist2, C * V2_HIST));
				HIPCHK(dalloc(&h->d_v2f2, C * 2 * h->W)); HIPCHK(dalloc(&h->d_v2prom2, C * 2 * h->W)); HIPCHK(dalloc(&h->d_v2en2, C * (h->W + 1)));
				for (int i = 0; i < 2; i++) { HIPCHK(dalloc(&h->d_v2fmtail[i], C * 16)); HIPCHK(hipEventCreateWithFlags(&h->ev_v2engine[i], hipEventDisableTiming)); }
				HIPCHK(hipEventCreateWithFlags(&h->ev_v2assist, hipEventDisableTiming));
				HIPCHK(hipEventCreateWithFlags(&h->ev_v2front, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&h->ev_v2fm, hipEventDisableTiming));
				// (measured, round 6: the engine on s1 beside the next block's front end and assist kernels takes 4.1 ms instead of 2.6 -- its
				// workgroups need 50 KB of LDS each and wait for CUs the throughput kernels fill, and every shared SIMD delays its dependent
				// chains -- so the step got slower, 4.1 against 3.1 ms.  With only the next block's FRONT END beside it (V2_ENGINE_OWN_STREAM = 1,
				// the assist kernels held back until the engine is through): 3.9 against 2.58 ms.  The engine stays behind its assist kernels
				// on their stream; the pairs of buffers stay, they cost nothing.)
				h->v2_stream = (V2_ENGINE_OWN_STREAM && !h->serial) ? h->s1 : h->ds;
			}
			if (v2_host) {
				HIPCHK(hipHostMalloc((void**)&h->h_v2f, 2 * MAXSUB * C * 2 * h->W * sizeof(float), hipHostMallocDefault));
				HIPCHK(hipHostMalloc((void**)&h->h_v2prom, 2 * MAXSUB * C * 2 * h->W * sizeof(float), hipHostMallocDefault));
				HIPCHK(hipHostMalloc((void**)&h->h_v2en, 2 * MAXSUB * C * (h->W + 1) * sizeof(float), hipHostMallocDefault));
			}
			HIPCHK(dalloc(&h->d_fm, C * (FM_HIST + h->L)));
			for (int i = 0; i < 2; i++) HIPCHK(dalloc(&h->d_fmbits[i], C * (h->L / 32)));
			if (v2_host) HIPCHK(hipHostMalloc((void**)&h->h_fmbits, 2 * MAXSUB * C * (h->L / 32) * sizeof(uint32_t), hipHostMallocDefault));
			if (cfg->flags & AISGPU_FLAG_TAPS) HIPCHK(dalloc(&h->d_fmfir, C * (size_t)h->L));
			// FMDemod::prev: the engine decodes an all-zero block first (its look-back before the stream, V2Engine.cpp:275-279), which
			// leaves prev = 0 -- not the 1 + 0j of the constructor (V2Engine.h:84) -- in front of the first real sample
			HIPCHK(dalloc(&h->d_fmprev[0], C));
		}
	}
	const size_t osets = (h->eager_out || h->v2) ? 2 : 1; // (two sets of host slots where outputs are copied inside aisgpu_run(): see out_set)
	HIPCHK(hipHostMalloc((void**)&h->h_bits, osets * MAXSUB * C * 5 * h->words * sizeof(uint32_t), hipHostMallocDefault));
	HIPCHK(hipHostMalloc((void**)&h->h_lvl, osets * MAXSUB * C * h->Gcap * sizeof(float), hipHostMallocDefault));
	HIPCHK(hipHostMalloc((void**)&h->h_ppm, osets * MAXSUB * C * h->W * sizeof(float), hipHostMallocDefault));
	HIPCHK(hipDeviceSynchronize());
	return AISGPU_OK;
}

void aisgpu_destroy(aisgpu_t* h) {
#ifdef V2_PROF
	if (h && h->v2) { hipDeviceSynchronize(); aisk::v2_prof_dump(); }
#endif
	if (!h) return;
	DevGuard dg(h);
	rot_worker_stop(h);
	us_worker_stop(h);
	h->pend.valid = false;
	if (h->stream) hipStreamSynchronize(h->stream);
	if (h->s1) hipStreamSynchronize(h->s1);
	if (h->s3) hipStreamSynchronize(h->s3);
	if (h->s4) hipStreamSynchronize(h->s4);
	if (h->s5) hipStreamSynchronize(h->s5);
	drain_events(h);
	for (int i = 0; i < NBUF; i++) {
		if (h->ev_front[i]) hipEventDestroy(h->ev_front[i]);
		if (i == 0 && h->ev_fm) hipEventDestroy(h->ev_fm);
		if (h->ev_phasor[i]) hipEventDestroy(h->ev_phasor[i]);
		if (h->ev_search[i]) hipEventDestroy(h->ev_search[i]);
		if (h->ev_c48free[i]) hipEventDestroy(h->ev_c48free[i]);
		hipFree(h->d_rotT[i]); hipFree(h->d_c48[i]); hipFree(h->d_fz[i]); hipFree(h->d_ppm[i]);
	}
	for (int i = 0; i < 4; i++) { if (h->ev_ema[i]) hipEventDestroy(h->ev_ema[i]); hipFree(h->d_lvl[i]); hipFree(h->d_bits[i]);
		hipFree(h->d_rot[i]); if (h->h_rot[i]) hipHostFree(h->h_rot[i]); if (h->rot_ev[i]) hipEventDestroy(h->rot_ev[i]); }
	for (int i = 0; i < 2; i++) {
		if (h->ev_sym[i]) hipEventDestroy(h->ev_sym[i]);
		if (h->ev_k3[i]) hipEventDestroy(h->ev_k3[i]);
		if (h->ev_k4[i]) hipEventDestroy(h->ev_k4[i]);
		hipFree(h->d_sym[i]); hipFree(h->d_ema[i]);
	}
	for (int i = 0; i < aisgpu::USR; i++) {
		hipFree(h->d_usidx[i]); hipFree(h->d_usalpha[i]); hipFree(h->d_usrot[i]);
		if (h->h_usidx[i]) hipHostFree(h->h_usidx[i]);
		if (h->h_usalpha[i]) hipHostFree(h->h_usalpha[i]);
		if (h->h_usrot[i]) hipHostFree(h->h_usrot[i]);
		if (h->us_copy_ev[i]) hipEventDestroy(h->us_copy_ev[i]);
		if (h->us_used_ev[i]) hipEventDestroy(h->us_used_ev[i]);
	}
	for (int i = 0; i < XR; i++) hipFree(h->d_xpre[i]);
	hipFree(h->d_xhist[0]); hipFree(h->d_xhist[1]);
	for (int i = 0; i < XR; i++) { if (h->ev_xin[i]) hipEventDestroy(h->ev_xin[i]); if (h->ev_xread[i]) hipEventDestroy(h->ev_xread[i]); }
	for (auto& p : h->ev_free) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
	hipFree(h->d_xmid);
	hipFree(h->d_in[0]); hipFree(h->d_in[1]); hipFree(h->d_hist[0]); hipFree(h->d_hist[1]); hipFree(h->d_hist2[0]); hipFree(h->d_hist2[1]);
	hipFree(h->d_fm); hipFree(h->d_fmhist[0]); hipFree(h->d_fmhist[1]); hipFree(h->d_fmfir); hipFree(h->d_fmbits[0]); hipFree(h->d_fmbits[1]);
	if (h->h_fmbits) hipHostFree(h->h_fmbits);
	if (h->h_c48) hipHostFree(h->h_c48);
	hipFree(h->d_v2hist); hipFree(h->d_v2f); hipFree(h->d_v2prom); hipFree(h->d_v2en); hipFree(h->d_v2st); hipFree(h->d_slotcs); hipFree(h->d_v2locked);
	hipFree(h->d_v2hist2); hipFree(h->d_v2f2); hipFree(h->d_v2prom2); hipFree(h->d_v2en2); hipFree(h->d_v2fmtail[0]); hipFree(h->d_v2fmtail[1]);
	if (h->ev_v2assist) hipEventDestroy(h->ev_v2assist);
	if (h->ev_v2front) hipEventDestroy(h->ev_v2front);
	if (h->ev_v2fm) hipEventDestroy(h->ev_v2fm);
	for (int i = 0; i < 2; i++) if (h->ev_v2engine[i]) hipEventDestroy(h->ev_v2engine[i]);
	if (h->h_v2f) hipHostFree(h->h_v2f); if (h->h_v2prom) hipHostFree(h->h_v2prom); if (h->h_v2en) hipHostFree(h->h_v2en);
	for (int i = 0; i < NBUF; i++) { hipFree(h->d_magT[i]); hipFree(h->d_ck[i]); }
	hipFree(h->d_dfhist[0]); hipFree(h->d_dfhist[1]);
	hipFree(h->d_box[0]); hipFree(h->d_box[1]);
	for (int i = 0; i < 2; i++) {
		hipFree(h->k7b[i].ckpt); hipFree(h->k7b[i].end); hipFree(h->k7b[i].frames); hipFree(h->k7b[i].fallback); hipFree(h->k7b[i].sum_spec);
		if (h->ev_spec[i]) hipEventDestroy(h->ev_spec[i]);
	}
	hipFree(h->k7b[0].task_end); hipFree(h->k7b[0].task_frames); hipFree(h->k7b[0].task_merge); hipFree(h->k7b[0].take_spec); hipFree(h->k7b[0].take_task); hipFree(h->k7b[0].sum_task); hipFree(h->k7b[0].out_base); hipFree(h->k7b[0].fin_sel);
	hipFree(h->d_dec); hipFree(h->d_frames); hipFree(h->d_frame_count); hipFree(h->d_fmrows[0]); hipFree(h->d_fmrows[1]); hipFree(h->d_last_lvl[0]); hipFree(h->d_last_lvl[1]);
	hipFree(h->d_k7ev); hipFree(h->d_k7cnt); hipFree(h->d_k7open); hipFree(h->d_k7slot); hipFree(h->d_k7ovf);
	if (h->h_frames) hipHostFree(h->h_frames);
	hipFree(h->d_fmprev[0]); hipFree(h->d_fmprev[1]);
	hipFree(h->d_cgf); hipFree(h->d_omega); hipFree(h->d_step);
	hipFree(h->d_rotstate); hipFree(h->d_firtap); hipFree(h->d_ppmtab);
	hipFree(h->d_pswords); hipFree(h->d_psma0); hipFree(h->d_psma1); hipFree(h->d_psfin); hipFree(h->d_psflag);
	for (int i = 0; i < 2; i++) { if (h->h_in[i]) hipHostFree(h->h_in[i]); if (h->ev_h2d[i]) hipEventDestroy(h->ev_h2d[i]); if (h->ev_in_free[i]) hipEventDestroy(h->ev_in_free[i]); }
	if (h->sc) { hipStreamSynchronize(h->sc); hipStreamDestroy(h->sc); }
	if (h->h_bits) hipHostFree(h->h_bits);
	if (h->h_lvl) hipHostFree(h->h_lvl);
	if (h->h_ppm) hipHostFree(h->h_ppm);
	for (int i = 0; i < NBUF; i++) if (h->ev_pre[i]) hipEventDestroy(h->ev_pre[i]);
	if (h->stream) hipStreamDestroy(h->stream);
	if (h->s1 && !h->serial) hipStreamDestroy(h->s1);
	if (h->s3 && !h->serial) hipStreamDestroy(h->s3);
	if (h->s4 && !h->serial) hipStreamDestroy(h->s4);
	if (h->s5 && !h->serial && h->s5 != h->s4 && h->s5 != h->s1) hipStreamDestroy(h->s5);
	delete h;
}

int aisgpu_submit(aisgpu_t* h, int rx, const void* iq, int n_iq) {
	if (!h || !iq || rx < 0 || rx >= h->cfg.n_receivers || n_iq != h->cfg.block_len) return AISGPU_ERR_ARG;
	DevGuard dg(h);
	const size_t row = (size_t)h->cfg.block_len * h->in_bytes;
	int p;
	{ // bookkeeping under a lock (receiver threads submit concurrently); the copy itself is done outside it
		std::lock_guard<std::mutex> l(h->submit_mtx);
		p = (int)(h->in_blocks & 1);
		if (!h->sc) {
			HIPCHK(hipStreamCreateWithFlags(&h->sc, hipStreamNonBlocking));
			for (int i = 0; i < 2; i++) {
				HIPCHK(hipMalloc(&h->d_in[i], row * h->cfg.n_receivers));
				HIPCHK(hipHostMalloc(&h->h_in[i], row * h->cfg.n_receivers, hipHostMallocDefault));
				HIPCHK(hipEventCreateWithFlags(&h->ev_h2d[i], hipEventDisableTiming));
				HIPCHK(hipEventCreateWithFlags(&h->ev_in_free[i], hipEventDisableTiming));
			}
		}
		if (!h->submitted) { // first row of a new block: this staging pair was last used two blocks ago
			if (h->in_used[p]) HIPCHK(hipEventSynchronize(h->ev_in_free[p]));
			h->cur_in = h->d_in[p];
			h->cur_in_stride = h->cfg.block_len;
			h->submitted = true;
			h->staged = true;
		}
	}
	// the caller's buffer is only borrowed for this call (Device/FileRAW.cpp:131-136): copy to pinned staging, then to the device
	memcpy((char*)h->h_in[p] + row * rx, iq, row);
	HIPCHK(hipMemcpyAsync((char*)h->d_in[p] + row * rx, (char*)h->h_in[p] + row * rx, row, hipMemcpyHostToDevice, h->sc));
	return AISGPU_OK;
}

int aisgpu_submit_device(aisgpu_t* h, const void* iq_dev, long long rx_stride_samples) {
	if (!h || !iq_dev || rx_stride_samples < h->cfg.block_len) return AISGPU_ERR_ARG;
	if (((uintptr_t)iq_dev & 15) || ((rx_stride_samples * h->in_bytes) & 15)) return AISGPU_ERR_ARG; // 16-byte vector loads
	h->cur_in = iq_dev;
	h->cur_in_stride = rx_stride_samples;
	h->submitted = true;
	h->staged = false;
	return AISGPU_OK;
}

#ifndef ROT_COPY_ON_S3_WITH_DECODERS
#define ROT_COPY_ON_S3_WITH_DECODERS 1
#endif
int aisgpu_run(aisgpu_t* h) {
	if (!h) return AISGPU_ERR_ARG;
	if (!h->submitted) return AISGPU_ERR_STATE;
	DevGuard dg(h);
	const bool cu8 = h->kfmt != 0; // an integer format: converted on the fly by the front end
	const int R = h->cfg.n_receivers;
	h->n_sub = 0;
	const int in_p = (int)(h->in_blocks & 1);
	h->out_set = (h->eager_out || h->v2) ? in_p : 0; // (both copy outputs to the host inside aisgpu_run(): a pipelined caller is still reading the previous block's)
	if (h->staged) { // the rows' host -> device copies run on the copy stream: the front stream waits for them
		HIPCHK(hipEventRecord(h->ev_h2d[in_p], h->sc));
		HIPCHK(hipStreamWaitEvent(h->stream, h->ev_h2d[in_p], 0));
	}

	EvPair ev{};
	auto time_begin = [&]() -> int {
		if (!h->timing) return AISGPU_OK;
		if (h->ev_free.empty()) { HIPCHK(hipEventCreate(&ev.a)); HIPCHK(hipEventCreate(&ev.b)); }
		else { ev = h->ev_free.back(); h->ev_free.pop_back(); }
		HIPCHK(hipEventRecord(ev.a, h->stream));
		return AISGPU_OK;
	};
	auto time_end = [&]() -> int {
		if (!h->timing) return AISGPU_OK;
		HIPCHK(hipEventRecord(ev.b, h->stream));
		h->ev_busy.push_back(ev);
		return AISGPU_OK;
	};

	// ---- resampled ladders: the tables of the flushes this input block completes were staged during the previous call; stage the
	// NEXT block's now (they depend on the stream position alone: us_worker_main / stage_resample_run)
	int n_flush = 0;
	if (h->mode == MODE_RESAMPLE) {
		if (h->run_idx == 0) { int rc = stage_resample_run(h, 0, &h->next_nflush); if (rc) return rc; }
		n_flush = h->next_nflush;
		{ int rc = stage_resample_run(h, h->run_idx + 1, &h->next_nflush); if (rc) return rc; }
		h->run_idx++;
		// The pass of run g overwrites ring slot g % XR = the pre-decimated block of run g - XR, whose last reader is the resampler front
		// end of run g - XR + 2 (as xprev2).  The pass nevertheless waits for the flushes of run g - 2 -- more than the ring of six needs,
		// and deliberately: with the minimal wait (run g - 4; measured in round 5, A/B on one box) the passes run up to four blocks
		// ahead of everything behind them, the steady state is the same (0.464-0.466 against 0.463-0.475 ms per step at 6 MSPS: what
		// separates two passes is the resampler front end's 12,288 workgroups taking the CUs first at every boundary, not this wait) and
		// a 20-step region drains a longer backlog (0.490-0.498 against 0.479-0.483 ms).
		WAITEV(h->stream, h->ev_xread[(h->in_blocks + XR - 2) % XR]);
	}

	// ---- pre-decimation pass (MODE_PRE / MODE_RESAMPLE): KP CIC5 stages at the input rate -> d_xpre
	float2* xcur = nullptr;
	long long xstride = 0;
	if (h->KP == 0 && (h->mode == MODE_DSK || h->mode == MODE_RESAMPLE || h->mode == MODE_96K)) { // 288 kHz input: no CIC5 stage in front of DownsampleKFilter, only the
		// conversion; likewise rates resampled into the 384k bucket: Upsample works on the converted input itself (Model.cpp:295-301)
		const bool ring = h->mode == MODE_RESAMPLE;
		const int xb = ring ? (int)(h->in_blocks % XR) : (int)(h->in_blocks & 1);
		xcur = h->d_xpre[xb];
		xstride = (long long)h->xh + h->n_pre;
		if (h->x_direct) {
			// CF32 rows are read where the caller put them (round 4: the converted copy was 0.5 of the 2.2 ms the front stream of the
			// 288 kSPS ladder took per 1.6 GB of input); what the next block needs of this one is its tail: d_xhist, by block parity
		} else {
		if (!ring && h->in_blocks > 0) HIPCHK(launch_copy_rows(h->d_xpre[xb ^ 1] + h->n_pre, xstride, xcur, xstride, h->xh, R, h->stream));
		if (h->ma_m) HIPCHK(launch_ma_rows(h->cur_in, h->cur_in_stride, h->kfmt, h->ma_m, xcur + h->xh, xstride, h->n_pre, R, h->stream));
		else HIPCHK(launch_convert_rows(h->cur_in, h->cur_in_stride, h->kfmt, xcur + h->xh, xstride, h->n_pre, R, h->stream));
		}
	}
	if (h->KP > 0) {
		const bool ring = h->mode == MODE_RESAMPLE, two = h->mode == MODE_DSK;
		const int xb = ring ? (int)(h->in_blocks % XR) : two ? (int)(h->in_blocks & 1) : 0;
		xcur = h->d_xpre[xb];
		xstride = (long long)h->xh + h->n_pre;
		if (two && h->in_blocks > 0) // history = the last xh samples before this block
			HIPCHK(launch_copy_rows(h->d_xpre[xb ^ 1] + h->n_pre, xstride, xcur, xstride, h->xh, R, h->stream));
		K1Params kp{};
		const int hb = (int)(h->in_blocks & 1);
		const int KP1 = h->KPa ? h->KPa : h->KP; // stages of the (first) pass over the raw input
		const long long n_mid = (long long)h->cfg.block_len >> KP1;
		const bool pre_saves = !cu8 && KP1 >= 2; // the LDS-DMA form of the kernel saves the tail itself
		kp.in = h->cur_in; kp.in_stride = h->cur_in_stride; kp.hist = h->d_hist[hb]; kp.hist_out = pre_saves ? h->d_hist[hb ^ 1] : nullptr; kp.rot = nullptr;
		kp.c48 = nullptr; kp.c48_stride = 0;
		kp.tiles_per_block = h->ptiles_per_block; kp.tiles_per_span = h->ptiles_per_span;
		kp.alpha = 0; kp.beta = 1; kp.has_fdc = 0;
		kp.pre_out = h->KPa ? h->d_xmid : xcur + h->xh; kp.pre_stride = h->KPa ? n_mid : xstride;
		int rc = time_begin(); if (rc) return rc;
		HIPCHK(launch_k1(kp, KP1, h->kfmt, h->pspans, R, h->stream));
		rc = time_end(); if (rc) return rc;
		if (h->KPa) { // four more stages on the CF32 stream of the first pass (its own tail tile: d_hist2)
			K1Params kb{};
			const int tile_b = h->tile96 << 4;
			const bool b_saves = true;
			kb.in = h->d_xmid; kb.in_stride = n_mid; kb.hist = h->d_hist2[hb]; kb.hist_out = b_saves ? h->d_hist2[hb ^ 1] : nullptr; kb.rot = nullptr;
			kb.c48 = nullptr; kb.c48_stride = 0;
			kb.tiles_per_block = (int)(n_mid / tile_b);
			kb.tiles_per_span = span_tiles(kb.tiles_per_block, (int)R, h->cfg.tiles_per_span);
			kb.alpha = 0; kb.beta = 1; kb.has_fdc = 0;
			kb.pre_out = xcur + h->xh; kb.pre_stride = xstride;
			HIPCHK(launch_k1(kb, 4, 0, (kb.tiles_per_block + kb.tiles_per_span - 1) / kb.tiles_per_span, R, h->stream));
			if (!b_saves) HIPCHK(launch_k1_tail(h->d_xmid, n_mid * 8, n_mid * 8, h->d_hist2[hb ^ 1], tile_b * 8, R, h->stream));
		}
		if (!pre_saves) HIPCHK(launch_k1_tail(h->cur_in, h->cur_in_stride * h->in_bytes, (long long)h->cfg.block_len * h->in_bytes, h->d_hist[hb ^ 1],
		                                      h->ptile_in * h->in_bytes, R, h->stream));
	}

	if (h->mode == MODE_96K) {
		// ---- 96 kSPS: the converted input goes straight into Rotate; one downstream block per input block
		const int pb = (int)(h->block_idx & 1);
		const int q = (int)(h->block_idx % NBUF);
		if (!h->mode_x) { // (channel mode X has no Rotate: no table to make and to copy in front of every block)
			if (h->block_idx >= 2) HIPCHK(hipEventSynchronize(h->rot_ev[pb]));
			gen_rot_table(h, h->h_rot[pb]);
			HIPCHK(hipMemcpyAsync(h->d_rot[pb], h->h_rot[pb], ((size_t)ROT_HIST + h->n96) * sizeof(float2), hipMemcpyHostToDevice, h->stream));
			HIPCHK(hipEventRecord(h->rot_ev[pb], h->stream));
			h->rot_ev_used[pb] = true;
		}
		WAITEV(h->stream, h->ev_c48free[q]);
		K1uParams ku;
		ku.xin = xcur; ku.xin_stride = xstride; ku.xin_off = h->xh;
		if (h->x_direct) { ku.xin = static_cast<const float2*>(h->cur_in); ku.xin_stride = h->cur_in_stride; ku.xin_off = 0; ku.xhist = h->d_xhist[h->in_blocks & 1]; ku.xhist_len = h->xh; }
		ku.us_idx = nullptr; ku.us_alpha = nullptr; ku.rot = h->d_rot[pb];
		ku.c48 = h->d_c48[q]; ku.c48_stride = h->c48s;
		ku.alpha = h->alpha; ku.beta = h->beta; ku.has_fdc = h->mode_x ? h->has_fdc : 0; ku.L = h->L;
		if (h->mode_x) {
			ku.c48_rows_per_rx = 1; ku.spw_force = h->k1u_spw;
			// the analysis at the end of the front-end waves (k1x_wave): a block of an even number of windows, the fused back end
			h->front_fft = h->fused && k1x_wave_form(ku, h->npost) && h->W % 2 == 0 && FRONT_FFT_IN_WAVES;
			if (h->front_fft) { ku.omega = h->d_omega; ku.ppm_table = h->d_ppmtab; ku.fz = h->d_fz[q]; ku.ppm = h->d_ppm[q]; ku.n_windows = h->W; ku.wide = h->cfg.afc_wide ? 1 : 0; }
			HIPCHK(launch_k1x(ku, h->npost, R, h->stream));
		} // (test hook k1u_spw = 2: the workgroup form of K1x at 96 kSPS, 4 / 8: k1x_wave with spans of that many tiles)
		else {
			ku.spw_force = h->k1u_spw;
			h->front_fft = h->fused && k1u96_wave_form(ku, 0) && FRONT_FFT_IN_WAVES; // the analysis at the end of the front-end waves (k1k_wave<false>)
			if (h->front_fft) { ku.omega = h->d_omega; ku.ppm_table = h->d_ppmtab; ku.fz = h->d_fz[q]; ku.ppm = h->d_ppm[q]; ku.n_windows = h->W; ku.wide = h->cfg.afc_wide ? 1 : 0; }
			HIPCHK(launch_k1u(ku, 0, R, h->stream));
		}
		if (h->x_direct) HIPCHK(launch_copy_rows(ku.xin + h->n_pre - h->xh, ku.xin_stride, h->d_xhist[(h->in_blocks & 1) ^ 1], h->xh, h->xh, R, h->stream));
		int rc = enqueue_downstream(h, q, pb);
		if (rc) return rc;
	} else if (h->mode == MODE_DSK) {
		// ---- decimate-by-3 ladder: one downstream block per input block (block_len is a whole number of the
		// filter's 8192-sample output blocks, so the reference hands everything on within the same call)
		const int pb = (int)(h->block_idx & 1);
		const int q = (int)(h->block_idx % NBUF);
		if (h->block_idx >= 2) HIPCHK(hipEventSynchronize(h->rot_ev[pb]));
		gen_rot_table(h, h->h_rot[pb]);
		HIPCHK(hipMemcpyAsync(h->d_rot[pb], h->h_rot[pb], ((size_t)ROT_HIST + h->n96) * sizeof(float2), hipMemcpyHostToDevice, h->stream));
		HIPCHK(hipEventRecord(h->rot_ev[pb], h->stream));
		h->rot_ev_used[pb] = true;
		WAITEV(h->stream, h->ev_c48free[q]);
		K1kParams kk;
		kk.xin = xcur; kk.xin_stride = xstride; kk.xin_off = h->xh;
		if (h->x_direct) { kk.xin = static_cast<const float2*>(h->cur_in); kk.xin_stride = h->cur_in_stride; kk.xin_off = 0; kk.xhist = h->d_xhist[h->in_blocks & 1]; kk.xhist_len = h->xh; }
		kk.rot = h->d_rot[pb]; kk.c48 = h->d_c48[q]; kk.c48_stride = h->c48s; kk.L = h->L;
		kk.us_idx = nullptr; kk.us_alpha = nullptr;
		memcpy(kk.taps, TAPS_BH_28_3, sizeof kk.taps);
		h->front_fft = h->fused && k1k_wave_form(kk, h->k1u_spw) && FRONT_FFT_IN_WAVES; // the analysis at the end of the front-end waves (k1k_wave)
		if (h->front_fft) { kk.omega = h->d_omega; kk.ppm_table = h->d_ppmtab; kk.fz = h->d_fz[q]; kk.ppm = h->d_ppm[q]; kk.n_windows = h->W; kk.wide = h->cfg.afc_wide ? 1 : 0; }
		HIPCHK(launch_k1k(kk, R, h->stream, h->k1u_spw));
		if (h->x_direct) HIPCHK(launch_copy_rows(kk.xin + h->n_pre - h->xh, kk.xin_stride, h->d_xhist[(h->in_blocks & 1) ^ 1], h->xh, h->xh, R, h->stream));
		int rc = enqueue_downstream(h, q, pb);
		if (rc) return rc;
	} else if (h->mode != MODE_RESAMPLE) {
		// ---- one downstream block per input block
		const int pb = (int)(h->block_idx & 1);
		const int q = (int)(h->block_idx % NBUF); // ring slot of c48 / fz / ppm / rotT
		// the pinned phasor buffer `pb` was last used two blocks ago; wait until that upload has been consumed
		// (only blocks when the host runs more than one block ahead of the device)
		// The Rotate phasor table of this block: staged one block AHEAD on s3 (below), so that neither the 10 us copy nor its launch
		// gap sits between two kernels of the front stream; only the first block (and the single-stream mode) stages it here.
		const auto stage_rot = [&](int b, hipStream_t st) -> int { return stage_rot_from_worker(h, b, st); }; // table generated ahead by the worker thread
		const int rb = (int)(h->block_idx & 3); // ring slot of this block's table
		if (h->rot_next <= h->block_idx) { int rc = stage_rot(rb, h->stream); if (rc) return rc; h->rot_next = h->block_idx + 1; }
		else WAITEV(h->stream, h->rw.ev[h->rot_slot[rb]]);
		// c48/fz/ppm[q] were last read by K2b/K2c of block f-NBUF (ModelEngineV2 on the device: by the engine, on this very stream -- no wait to enqueue)
		if (!(h->v2 && h->gpu_decode && h->v2_assist && h->v2_stream == h->stream)) WAITEV(h->stream, h->ev_c48free[q]);
		K1Params k1{};
		const bool from_pre = h->mode == MODE_PRE;
		k1.in = from_pre ? (const void*)xcur : h->cur_in;
		k1.in_stride = from_pre ? xstride : h->cur_in_stride;
		const int hb = (int)(h->in_blocks & 1);
		const bool saves = (from_pre || !cu8) && h->K >= 2; // the LDS-DMA form of the kernel saves the tail itself
		k1.hist = from_pre ? h->d_hist2[hb] : h->d_hist[hb];
		k1.hist_out = !saves ? nullptr : from_pre ? h->d_hist2[hb ^ 1] : h->d_hist[hb ^ 1];
		k1.rot = h->d_rot[rb];
		k1.c48 = h->d_c48[q]; k1.c48_stride = h->c48s;
		k1.tiles_per_block = h->tiles_per_block; k1.tiles_per_span = h->tiles_per_span;
		k1.alpha = h->alpha; k1.beta = h->beta; k1.has_fdc = h->has_fdc; k1.stream_start = h->in_blocks == 0;
		k1.pre_out = nullptr; k1.pre_stride = 0;
		if (h->fft_in_k1) {
			k1.fft_windows = h->tiles_per_span / 16; k1.n_windows = h->W; k1.wide = h->cfg.afc_wide ? 1 : 0;
			k1.omega = h->d_omega; k1.ppm_table = h->d_ppmtab; k1.fz = h->d_fz[q]; k1.ppm = h->d_ppm[q];
		}
		// the launch's events ride on its dispatch packet where they can: "front end of block f done" (what s3 waits for) and,
		// while the launch is being timed, its two time stamps
		const bool bound = !from_pre && h->fft_in_k1 && h->fused && !h->serial && !h->trace;
		h->k1_done[q] = nullptr;
		if (bound) {
			hipEvent_t e0 = nullptr, e1 = h->ev_search[q];
			if (h->timing) {
				if (h->ev_free.empty()) { HIPCHK(hipEventCreate(&ev.a)); HIPCHK(hipEventCreate(&ev.b)); }
				else { ev = h->ev_free.back(); h->ev_free.pop_back(); }
				e0 = ev.a; e1 = ev.b;
				h->ev_busy.push_back(ev);
			}
			HIPCHK(launch_k1(k1, h->K, h->kfmt, h->spans, R, h->stream, e0, e1));
			h->k1_done[q] = e1;
		} else {
		if (!from_pre) { int rc = time_begin(); if (rc) return rc; }
		{ TraceScope t(h, "front", h->stream); HIPCHK(launch_k1(k1, h->K, from_pre ? 0 : h->kfmt, h->spans, R, h->stream)); }
		if (!from_pre) { int rc = time_end(); if (rc) return rc; }
		}
		if (saves) {}
		else if (from_pre) HIPCHK(launch_k1_tail(xcur, xstride * 8, (long long)h->n_pre * 8, h->d_hist2[hb ^ 1], h->tile_in * 8, R, h->stream));
		else HIPCHK(launch_k1_tail(h->cur_in, h->cur_in_stride * h->in_bytes, (long long)h->cfg.block_len * h->in_bytes, h->d_hist[hb ^ 1],
		                           h->tile_in * h->in_bytes, R, h->stream));
		if (!h->serial && h->fused) {
			// The tables of the next blocks -- TWO blocks ahead, on s4 (refine + derotation / FIR) since round 6.  Rounds 2-5 had the copy on
			// s3 in front of this block's phasor recurrence: with a lead of one the front end of block f+1 had to wait for the recurrence of
			// block f-1 to be through -- a loop front end(f-1) -> recurrence(f-1) -> table(f+1) -> front end(f+1) that held the step at
			// (front end + recurrence + copy) / 2 = 0.47 ms -- hence the lead of two; but s3 is also where a LONE receiver's step is
			// decided: its recurrences run back to back there (0.181 ms each, a chain across blocks that nothing shortens), and the
			// 20-25 us copy kernel between two of them (it reads pinned host memory) was a tenth of BASELINE configs[1]'s 0.219 ms per
			// block: 0.195-0.203 with the copy on s4, whose kernels of block f-1 also run behind that block's recurrence, so the slot
			// d_rot[(f+2) & 3] -- last read by the front end of block f-2 -- is free as before.  256 receivers: unchanged (+-0.3 %).
			// With the frame decoders on the device, one block late on s4 (dec_defer), the copy stays where rounds 2-5 had it: s4 is that
			// plan's longest stream (measured: bench.py --gpu-decode 20 steps, 375 GS/s with the copy on s4, see profiles/r06_expH).
			const bool rot_on_s3 = ROT_COPY_ON_S3_WITH_DECODERS && h->gpu_decode && h->dec_defer;
			while (h->rot_next <= h->block_idx + 2) {
				if (!rot_on_s3 && h->rot_next >= 4) { // (explicitly, for the stream plans in which s4 carries nothing of block f-1: option k46)
					const int qr = (int)((h->rot_next - 4) % NBUF);
					WAITEV(h->s4, h->k1_done[qr] ? h->k1_done[qr] : h->ev_search[qr]);
				}
				int rc = stage_rot((int)(h->rot_next & 3), rot_on_s3 ? h->s3 : h->s4);
				if (rc) return rc;
				h->rot_next++;
			}
		}
		int rc = enqueue_downstream(h, q, pb);
		if (rc) return rc;
		if (!h->serial && V2_FM_BESIDE && h->v2 && h->gpu_decode && h->v2_assist && h->s4 != h->ds && h->v2_stream == h->ds) {
			// ModelEngineV2 with its engine on the device: the step is ONE chain (front end -> assist kernels -> engine), so the 10 us table
			// copy and its launch gap in front of every front end were 0.7 % of it.  The next block's table goes to s4 instead, BEHIND this
			// block's FM branch and carry there: the next front end waits for the table's event anyway, which now also says that the carry
			// is through (it writes the look-back the next block's assist kernels read) -- one wait in front of the front end, not two.
			// Slot (f+1) & 3 was last read by the front end of block f-3.
			while (h->rot_next <= h->block_idx) { // (block_idx counts this block already)
				int rc2 = stage_rot((int)(h->rot_next & 3), h->s4);
				if (rc2) return rc2;
				h->rot_next++;
			}
		}
	} else {
		// ---- resampled ladders, device part: the flushes this block completes (tables: see the top).  The resampler front end stays on
		// the front stream, behind the pass over the raw input: next to the NEXT block's pass (on the downstream stream) it takes
		// 0.27-0.39 ms instead of 0.09 -- both want the CUs' LDS and the same memory system -- and the downstream stream becomes the
		// pipeline's longest (0.60 ms per step); everything behind the 48 kHz channels runs on `ds`.
		HIPCHK(hipEventRecord(h->ev_xin[h->in_blocks % XR], h->stream)); // the pre-decimated block is there
		// Round 4, late: with us_on_ds the resampler front end is the FIRST thing on the downstream stream when the pass over the input
		// ends -- that stream is idle then (the previous block's derotation / FIR kernel is deferred behind it: enqueue_fused_back
		// below), its queue outranks the front stream's, so the resampler takes the CUs at the boundary between two passes instead
		// of getting them one by one from a pass already resident -- and the front stream is the passes back to back.
		hipStream_t st = h->us_on_ds ? h->ds : h->stream;
		const int len = h->n_pre;
		(void)len;
		for (int fi = 0; fi < n_flush; fi++) {
			const int pb = (int)(h->block_idx & 1);
			const int q = (int)(h->block_idx % NBUF);
			const int slot = (int)(h->run_flush++ % aisgpu::USR);
			WAITEV(st, h->ev_xin[h->in_blocks % XR]);
			WAITEV(st, h->us_copy_ev[slot]);
			WAITEV(st, h->ev_c48free[q]);
			if (h->eager_out) WAITEV(st, h->ev_ema[(h->block_idx + 4 - NBUF % 4) & 3]); // ppm[q] of block f-NBUF has been copied out
			if (h->us_dsk) { // US >> DSK >> ROT: the flush is a whole number of the filter's 8192-sample output blocks
				K1kParams kk;
				kk.xin = xcur; kk.xin_stride = xstride; kk.xin_off = h->xh;
				kk.xprev = h->d_xpre[(h->in_blocks + XR - 1) % XR]; kk.xprev2 = h->d_xpre[(h->in_blocks + XR - 2) % XR]; kk.n_in = h->n_pre;
				kk.rot = h->d_usrot[slot]; kk.c48 = h->d_c48[q]; kk.c48_stride = h->c48s; kk.L = h->L;
				kk.us_idx = h->d_usidx[slot]; kk.us_alpha = h->d_usalpha[slot];
				memcpy(kk.taps, TAPS_BH_28_3, sizeof kk.taps);
				// (round 6, last: the one-wave front end with Upsample in its lanes, and the spectral analysis at the end of its waves)
				h->front_fft_us = h->fused && k1k_wave_form(kk, h->k1u_spw) && FRONT_FFT_IN_WAVES;
				if (h->front_fft_us) { kk.omega = h->d_omega; kk.ppm_table = h->d_ppmtab; kk.fz = h->d_fz[q]; kk.ppm = h->d_ppm[q]; kk.n_windows = h->W; kk.wide = h->cfg.afc_wide ? 1 : 0; }
				HIPCHK(launch_k1k(kk, R, st, h->k1u_spw));
				if (h->front_fft_us) { HIPCHK(hipEventRecord(h->ev_search[q], st)); h->k1_done[q] = h->ev_search[q]; } // fz / ppm of the flush are there when these waves are
			} else {
				K1uParams ku;
				ku.xin = xcur; ku.xin_stride = xstride; ku.xin_off = h->xh;
				ku.xprev = h->d_xpre[(h->in_blocks + XR - 1) % XR]; ku.xprev2 = h->d_xpre[(h->in_blocks + XR - 2) % XR]; ku.n_in = h->n_pre;
				ku.us_idx = h->d_usidx[slot]; ku.us_alpha = h->d_usalpha[slot]; ku.rot = h->d_usrot[slot];
				ku.c48 = h->d_c48[q]; ku.c48_stride = h->c48s;
				ku.alpha = h->alpha; ku.beta = h->beta; ku.has_fdc = h->has_fdc; ku.L = h->L;
				if (h->mode_x) { ku.c48_rows_per_rx = 1; HIPCHK(launch_k1x(ku, h->npost, R, st)); }
				else if (h->us_k1) { // the same stages as one-wave workgroups of the front-end kernel (see us_k1)
					K1Params k1{};
					k1.in = nullptr; k1.in_stride = 0; k1.hist = nullptr; k1.hist_out = nullptr;
					k1.us_idx = ku.us_idx; k1.us_alpha = ku.us_alpha;
					k1.xin = ku.xin; k1.xin_stride = ku.xin_stride; k1.xin_off = ku.xin_off; k1.xprev = ku.xprev; k1.xprev2 = ku.xprev2; k1.n_in = ku.n_in;
					k1.rot = ku.rot; k1.c48 = ku.c48; k1.c48_stride = ku.c48_stride;
					k1.tiles_per_block = h->us_tiles_per_block; k1.tiles_per_span = h->us_tiles_per_span;
					k1.alpha = h->alpha; k1.beta = h->beta; k1.has_fdc = h->has_fdc; k1.stream_start = 0;
					k1.pre_out = nullptr; k1.pre_stride = 0;
					if (h->us_fft_in_k1) {
						k1.fft_windows = h->us_tiles_per_span / 16; k1.n_windows = h->W; k1.wide = h->cfg.afc_wide ? 1 : 0;
						k1.omega = h->d_omega; k1.ppm_table = h->d_ppmtab; k1.fz = h->d_fz[q]; k1.ppm = h->d_ppm[q];
					}
					{ TraceScope t(h, "usfront", st); HIPCHK(launch_k1(k1, 2, 5, h->us_tiles_per_block / h->us_tiles_per_span, R, st)); }
					if (h->us_fft_in_k1) { HIPCHK(hipEventRecord(h->ev_search[q], st)); h->k1_done[q] = h->ev_search[q]; } // fz / ppm of the flush are there when these waves are
				}
				else { ku.spw_force = h->k1u_spw; HIPCHK(launch_k1u(ku, h->npost, R, st)); }
			}
			HIPCHK(hipEventRecord(h->us_used_ev[slot], st));
			HIPCHK(hipEventRecord(h->ev_xread[h->in_blocks % XR], st)); // (the run's last flush leaves the event that counts)
			h->us_slot_used[slot] = true;
			if (h->ds != st) { // the 48 kHz channels of this flush exist: everything behind them runs on ds, next to the next input block's pass
				HIPCHK(hipEventRecord(h->ev_pre[q], st));
				WAITEV(h->ds, h->ev_pre[q]);
			}
			if (h->us_on_ds) { int rc = enqueue_fused_back(h); if (rc) return rc; } // the previous flush's second half: behind this flush's resampler on ds
			int rc = enqueue_downstream(h, q, pb); // (advances block_idx)
			if (rc) return rc;
		}
		if (n_flush == 0) HIPCHK(hipEventRecord(h->ev_xread[h->in_blocks % XR], st)); // (behind the earlier runs' flushes on st)
	}
	if (h->staged) { // everything that reads the staged input (front end, tail copies, conversions) is on the front stream, in front of this
		HIPCHK(hipEventRecord(h->ev_in_free[in_p], h->stream));
		h->in_used[in_p] = true;
	}
	{ // (a receiver evicted by its batch may still be inside aisgpu_submit: its bookkeeping reads these under the same lock)
		std::lock_guard<std::mutex> l(h->submit_mtx);
		h->in_blocks++;
		h->submitted = false;
	}
	return AISGPU_OK;
}

int aisgpu_sync(aisgpu_t* h) {
	if (!h) return AISGPU_ERR_ARG;
	DevGuard dg(h);
	return sync_all(h);
}

int aisgpu_sync_outputs(aisgpu_t* h) {
	if (!h) return AISGPU_ERR_ARG;
	if (h->in_blocks == 0) return AISGPU_ERR_STATE;
	DevGuard dg(h);
	const size_t C = h->n_chan;
	{ int rc = enqueue_back(h); if (rc) return rc; }
	{ int rc = enqueue_fused_back(h); if (rc) return rc; }
	{ int rc = flush_decode(h); if (rc) return rc; }
	for (int s = 0; s < h->n_sub; s++) {
		const SubOut& so = h->sub[s];
		// ev_ema[pb]: PhaseSearch (and the frame decoder) of that block are done, wherever their last kernel ran; they are
		// ordered after everything that produced lvl/ppm
		if (!h->base && !h->v2 && !h->eager_out) {
		WAITEV(h->s2, h->ev_ema[so.lv]);
		HIPCHK(hipMemcpyAsync(h->h_bits + (size_t)s * C * 5 * h->words, h->d_bits[so.lv], C * 5 * h->words * sizeof(uint32_t), hipMemcpyDeviceToHost, h->s2));
		HIPCHK(hipMemcpyAsync(h->h_lvl + (size_t)s * C * h->Gcap, h->d_lvl[so.lv], C * h->Gcap * sizeof(float), hipMemcpyDeviceToHost, h->s2));
		HIPCHK(hipMemcpyAsync(h->h_ppm + (size_t)s * C * h->W, h->d_ppm[so.q], C * h->W * sizeof(float), hipMemcpyDeviceToHost, h->s2));
		}
		if (h->challenger || h->base)
			HIPCHK(hipMemcpyAsync(h->h_fmbits + (size_t)s * C * (h->L / 32), h->d_fmbits[so.pb], C * (h->L / 32) * sizeof(uint32_t), hipMemcpyDeviceToHost, h->s2));
	}
	int rc = sync_all(h);
	if (rc != AISGPU_OK) return rc;
	h->n_osub = h->n_sub;
	h->oset = h->out_set;
	for (int i = 0; i < h->n_sub; i++) h->osub[i] = h->sub[i];
	h->have_out = true; // (the decisions are there even if the frame ring below has overflowed)
	if (h->gpu_decode) { rc = gather_frames(h); if (rc != AISGPU_OK) return rc; }
	return AISGPU_OK;
}

int aisgpu_decoder_fallbacks(aisgpu_t* h, long long* count) {
	if (!h || !count) return AISGPU_ERR_ARG;
	DevGuard dg(h);
	int v = 0;
	if (h->d_k7ovf) HIPCHK(hipMemcpy(&v, h->d_k7ovf + 2, sizeof v, hipMemcpyDeviceToHost));
	long long total = v;
	if (h->base_chunked) // ModelBase: channel-blocks whose frame lists overflowed and went through k7_base (both scratch sets)
		for (int i = 0; i < 2; i++) {
			HIPCHK(hipMemcpy(&v, h->k7b[i].fallback_count, sizeof v, hipMemcpyDeviceToHost));
			total += v;
		}
	if (h->d_v2locked) { // ModelEngineV2 on the device: FreqOffset::Estimate() calls at a learned slot phase (windows the assist kernels cannot know)
		HIPCHK(hipMemcpy(&v, h->d_v2locked, sizeof v, hipMemcpyDeviceToHost));
		total += v;
	}
	*count = total;
	return AISGPU_OK;
}

int aisgpu_ps_fallbacks(aisgpu_t* h, long long* count) {
	if (!h || !count) return AISGPU_ERR_ARG;
	DevGuard dg(h);
	int v = 0;
	if (h->d_psflag) HIPCHK(hipMemcpy(&v, h->d_psflag + 2, sizeof v, hipMemcpyDeviceToHost));
	*count = v;
	return AISGPU_OK;
}

int aisgpu_frames(aisgpu_t* h, const aisgpu_frame** frames, int* count) {
	if (!h || !frames || !count) return AISGPU_ERR_ARG;
	if (!h->gpu_decode || !h->have_out) return AISGPU_ERR_STATE;
	*frames = h->frames.data();
	*count = (int)h->frames.size();
	return AISGPU_OK;
}

int aisgpu_out_count(aisgpu_t* h) { return h ? h->n_osub : 0; }

int aisgpu_fetch_sub(aisgpu_t* h, int sub, int rx, int ch, aisgpu_out* o) {
	if (!h || !o || rx < 0 || rx >= h->cfg.n_receivers || ch < 0 || ch > 1) return AISGPU_ERR_ARG;
	if (!h->have_out) return AISGPU_ERR_STATE;
	if (sub < 0 || sub >= h->n_osub) return AISGPU_ERR_ARG;
	const size_t C = h->n_chan;
	const size_t chan = h->mode_x ? (size_t)rx : (size_t)rx * 2 + ch;
	const SubOut& so = h->osub[sub];
	o->n_groups = so.groups;
	o->first_group = so.first_group;
	if (h->mode_x && ch == 1) { // channel mode X has no second channel: what a chain fed with silence puts out (decisions 0, level 0, the ppm of "no peak")
		for (int j = 0; j < 5; j++) o->bits[j] = reinterpret_cast<const uint32_t*>(h->h_silent.data());
		o->lvl = h->h_silent.data();
		o->n_windows = h->W;
		o->ppm = h->h_silent_ppm.data();
		o->group_window = nullptr;
		o->first_sample48 = so.first48;
		o->fm_bits = nullptr; o->c48 = nullptr; o->v2_f = o->v2_prom = o->v2_energy = nullptr;
		return AISGPU_OK;
	}
	const size_t oslot = (size_t)h->oset * MAXSUB + sub; // (eager_out: the set of host slots the synced run wrote)
	for (int j = 0; j < 5; j++) o->bits[j] = h->h_bits + oslot * C * 5 * h->words + (chan * 5 + j) * h->words;
	o->lvl = h->h_lvl + oslot * C * h->Gcap + chan * h->Gcap;
	o->n_windows = h->W;
	o->ppm = h->h_ppm + oslot * C * h->W + chan * h->W;
	o->group_window = nullptr;
	o->first_sample48 = so.first48;
	const size_t vslot = h->v2 ? oslot : (size_t)sub; // (ModelEngineV2's outputs are copied inside aisgpu_run(): two sets of slots)
	// (ModelEngineV2 with the engine on the device: none of these was copied -- or allocated -- on the host side)
	o->fm_bits = ((h->challenger || h->base || (h->v2 && h->v2_assist)) && h->h_fmbits) ? h->h_fmbits + vslot * C * (h->L / 32) + chan * (h->L / 32) : nullptr;
	o->c48 = (h->v2 && h->h_c48) ? (const float*)(h->h_c48 + (vslot * C + chan) * h->L) : nullptr;
	const bool va = h->v2 && h->v2_assist && h->h_v2f;
	o->v2_f = va ? h->h_v2f + (vslot * C + chan) * 2 * h->W : nullptr;
	o->v2_prom = va ? h->h_v2prom + (vslot * C + chan) * 2 * h->W : nullptr;
	o->v2_energy = va ? h->h_v2en + (vslot * C + chan) * (h->W + 1) : nullptr;
	return AISGPU_OK;
}

int aisgpu_fetch(aisgpu_t* h, int rx, int ch, aisgpu_out* o) { return aisgpu_fetch_sub(h, 0, rx, ch, o); }

long long aisgpu_tap(aisgpu_t* h, int which, int rx, float* dst, long long cap) {
	if (!h || which < 0 || which > 9 || rx < 0 || rx >= h->cfg.n_receivers) return -AISGPU_ERR_ARG;
	if (!(h->cfg.flags & AISGPU_FLAG_TAPS)) return -AISGPU_ERR_STATE;
	if (h->block_idx == 0 || h->n_sub == 0) return -AISGPU_ERR_STATE;
	DevGuard dg(h);
	if (h->mode_x && (which & 1)) return -AISGPU_ERR_ARG; // (channel mode X: one channel)
	const size_t chan = h->mode_x ? (size_t)rx : (size_t)rx * 2 + (which & 1);
	const SubOut& so = h->sub[h->n_sub - 1]; // taps show the last downstream block
	if (which >= 6) { // real-valued taps of the FM receivers (ModelChallenger FM branch, ModelBase, ModelStandard)
		if (!h->d_fm || !h->d_fmfir) return -AISGPU_ERR_STATE;
		const float* fsrc = which < 8 ? h->d_fm + chan * (FM_HIST + h->L) + FM_HIST : h->d_fmfir + chan * (size_t)h->L;
		if (sync_all(h) != AISGPU_OK) return -AISGPU_ERR_HIP;
		if (dst) {
			const long long c = h->L < cap ? h->L : cap;
			if (hipMemcpy(dst, fsrc, (size_t)c * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return -AISGPU_ERR_HIP;
		}
		return h->L;
	}
	const float2* src;
	long long n = h->L;
	if (which < 2) src = h->d_c48[so.q] + chan * h->c48s;
	else if (which < 4) src = h->d_cgf + chan * (CGF_HIST + h->L) + CGF_HIST;
	else {
		// FIR outputs exist for every sample that belongs to a group completed in this block:
		// block-relative indices [-carry, 5*n_groups - carry)
		const long long carry = so.first48 - so.first_group * 5;
		src = h->d_firtap + chan * (8 + h->L) + 4 - carry;
		n = 5LL * so.groups;
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
	DevGuard dg(h);
	sync_all(h);
	h->timing = enable != 0;
	h->k1_ms = 0;
	h->k1_launches = 0;
}

float aisgpu_frontend_ms(aisgpu_t* h, int* launches) {
	if (!h) return 0;
	DevGuard dg(h);
	sync_all(h);
	if (launches) *launches = h->k1_launches;
	return h->k1_launches ? (float)(h->k1_ms / h->k1_launches) : 0.0f;
}

} // extern "C"
