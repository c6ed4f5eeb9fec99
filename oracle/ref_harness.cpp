/*
 * oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A thin C-ABI shim around the *unmodified* reference sources, compiled where they
 * lie under /root/reference by oracle/Makefile into oracle/_ref/libaisref_{strict,fast}.so.
 * It instantiates the reference's own AIS::ModelDefault / AIS::ModelChallenger
 * (Source/DSP/Model.cpp:520-577, :601-678) on a stub Device::Device (Device.h:58-77),
 * pushes fixed-size RAW blocks through device.out.Send() exactly like
 * Device/FileRAW.cpp:135 does, and records
 *   - the NMEA sentences per channel (Marine/Message.cpp:569-631),
 *   - float taps after every stage of the hot path (recorder sinks Connect()ed to the
 *     reference's own Connection<> objects; no reference arithmetic is re-implemented here).
 *
 * This TU is compiled with -fno-access-control so the recorders can reach the private
 * block members of ModelFrontend/ModelDefault (Model.h:138-213).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the result.
 */
#include <atomic>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include <sched.h>

#include "Model.h"
#include "Device.h"
#include "FileRAW.h" // the reference's own file reader (Device/FileRAW.cpp: reader thread + FIFO + run thread), for ref_create_file()
#include "Filters.h"
#ifdef HASMI355X
#include "ModelGPU.h" // the reference-side binding of libaisgpu.so (integration/reference/Source/DSP/GPU): engines 12 / 14
#endif

// normally provided by Application/Main.cpp:38-49 (which we do not link)
std::atomic<bool> stop;
std::atomic<bool> stop_process;
void StopRequest() { stop = true; }

namespace {

template <typename T>
struct Rec : public StreamIn<T> {
	std::vector<T> buf;
	std::vector<float> ppm;   // tag.ppm seen at each call
	bool on = false;
	void Receive(const T* d, int len, TAG& tag) {
		if (!on) return;
		buf.insert(buf.end(), d, d + len);
		ppm.push_back(tag.ppm);
	}
};

struct BitRec : public StreamIn<FLOAT32> {
	std::vector<float> bits;
	std::vector<float> lvl;
	std::vector<long long> idx;
	bool on = false, connected = false;
	void Receive(const FLOAT32* d, int len, TAG& tag) {
		if (!on) return;
		for (int i = 0; i < len; i++) {
			bits.push_back(d[i]);
			lvl.push_back(tag.sample_lvl);
			idx.push_back(tag.sample_idx);
		}
	}
};

struct MsgSink : public StreamIn<AIS::Message> {
	std::string text;       // "<nmea>\n" per sentence, in emission order
	std::vector<float> level, ppm;
	int count = 0;
	void Receive(const AIS::Message* m, int len, TAG& tag) {
		for (int i = 0; i < len; i++) {
			for (const auto& s : m[i].sentences()) { text += s; text += '\n'; }
			level.push_back(tag.level);
			ppm.push_back(tag.ppm);
			count++;
		}
	}
};

struct CallCounter : public StreamIn<RAW> { // what the device's calls looked like
	int calls = 0, max_size = 0;
	void Receive(const RAW* r, int len, TAG&) { calls++; if (r->size > max_size) max_size = r->size; }
};

struct Harness {
	CallCounter cc;
	Device::Device stub;
	Device::RAWFile* file = nullptr; // ref_create_file(): the model hangs on the reference's real RAWFile instead of the stub
	Device::Device& dev;
	AIS::ModelDefault* md = nullptr;
	AIS::ModelChallenger* mc = nullptr;
	AIS::ModelBase* mb = nullptr;
	AIS::ModelStandard* ms = nullptr;
	AIS::ModelEngineV2* mv = nullptr;
#ifdef HASMI355X
	AIS::ModelDefaultGPU* mgpu = nullptr; // the GPU engines (refgpu build only)
#else
	AIS::Model* mgpu = nullptr;
#endif
	AIS::Model* model = nullptr;
	TAG tag;
	Format fmt;
	MsgSink sink;
	// taps: 0/1 = 48k front-end output A/B (FCIC5_a/b.out), 2/3 = CGF out, 4/5 = FIR-17 out
	Rec<CFLOAT32> tap[6];
	Rec<FLOAT32> ftap[4]; // 6/7 = Demod::FM output A/B, 8/9 = Filter(Receiver) output A/B (the FM receivers)
	BitRec bits[2][5];   // after PhaseSearchEMA, per channel/phase
	BitRec fmbits[2][5]; // challenger FM branch (input of DEC_af/bf)
	double seconds = 0;
	bool taps_connected = false, ftap_connected = false;

	Harness(Format f, int rate, Device::RAWFile* rf = nullptr) : stub(f, rate, Type::RAWFILE, "stub"), file(rf), dev(rf ? *(Device::Device*)rf : stub), fmt(f) {}
};

} // namespace

extern "C" {

// kind: 0 = ModelStandard, 1 = ModelBase, 2 = ModelDefault, 4 = ModelChallenger, 11 = ModelEngineV2; libaisrefgpu.so only: 12 = ModelDefaultGPU,
// 14 = ModelChallengerGPU, 20 = ModelStandardGPU, 21 = ModelBaseGPU, 31 = ModelEngineV2GPU.  fmt: 0 = CU8, 1 = CF32, 2 = CS8, 3 = CS16.
// flags: bit 0 record float taps, bit 1 `-go DSK on`, bit 2 `-go PS_EMA off`, bit 3 `-go FP_DS on`, bit 4 channel mode X (`-c X`), bit 5 `-go MA on`;
// GPU engines: bit 6 the AIS::Decoder state machines on the device (GpuPool::setGpuDecode), bit 7 pipelined hand-off (GpuPool::setPipelined).
// GPU engines created with the same configuration before their first block share ONE GPU context (GpuPool): feed them from one thread each.
void* ref_create_file(int kind, int sample_rate, int fmt, int flags, const char* filename);
void* ref_create(int kind, int sample_rate, int fmt, int flags) { return ref_create_file(kind, sample_rate, fmt, flags, nullptr); }

// filename != NULL: the model is built on the reference's own Device::RAWFile (what `-r <fmt> <file> -s <rate>` sets up,
// Application/Receiver.cpp + Device/FileRAW.cpp:230-251) instead of the stub device; ref_play_file() then runs the file through it
// with the reference's reader thread, FIFO and run thread -- including the 1-or-2-block hand-offs of FIFO::Front(-1)
// (Library/FIFO.h:99-109).  flags bit 8: `-go AFC_WIDE off`, bit 9: `-go DROOP off`.
void* ref_create_file(int kind, int sample_rate, int fmt, int flags, const char* filename) {
	const int taps = flags & 1;
	try {
		Format f = fmt == 0 ? Format::CU8 : fmt == 2 ? Format::CS8 : fmt == 3 ? Format::CS16 : Format::CF32;
		Device::RAWFile* rf = nullptr;
		if (filename) {
			rf = new Device::RAWFile();
			rf->SetKey(AIS::KEY_SETTING_FILE, filename);
			rf->setFormat(f);
			rf->setSampleRate(sample_rate);
		}
		Harness* h = new Harness(f, sample_rate, rf);
		if (kind == 4) { h->mc = new AIS::ModelChallenger(); h->model = h->mc; }
		else if (kind == 1) { h->mb = new AIS::ModelBase(); h->model = h->mb; }
		else if (kind == 0) { h->ms = new AIS::ModelStandard(); h->model = h->ms; }
		else if (kind == 11) { h->mv = new AIS::ModelEngineV2(); h->model = h->mv; }
#ifdef HASMI355X
		// what Receiver::addModel (Application/Receiver.cpp:155-195) would do for the new engine numbers
		else if (kind == 12 || kind == 14 || kind == 20 || kind == 21 || kind == 31) {
			AIS::GpuPool::instance().setGpuDecode((flags & 64) != 0);
			AIS::GpuPool::instance().setPipelined((flags & 128) != 0);
			h->mgpu = kind == 12 ? new AIS::ModelDefaultGPU() : kind == 14 ? (AIS::ModelDefaultGPU*)new AIS::ModelChallengerGPU()
			        : kind == 20 ? (AIS::ModelDefaultGPU*)new AIS::ModelStandardGPU() : kind == 31 ? (AIS::ModelDefaultGPU*)new AIS::ModelEngineV2GPU()
			        : (AIS::ModelDefaultGPU*)new AIS::ModelBaseGPU();
			h->model = h->mgpu;
		}
#endif
		else { h->md = new AIS::ModelDefault(); h->model = h->md; }
		if (flags & 2) h->model->SetKey(AIS::KEY_SETTING_DSK, "ON");
		if (flags & 4) h->model->SetKey(AIS::KEY_SETTING_PS_EMA, "OFF");
		if (flags & 8) h->model->SetKey(AIS::KEY_SETTING_FP_DS, "ON");
		if (flags & 32) h->model->SetKey(AIS::KEY_SETTING_MA, "ON");
		if (flags & 256) h->model->SetKey(AIS::KEY_SETTING_AFC_WIDE, "OFF");
		if (flags & 512) h->model->SetKey(AIS::KEY_SETTING_DROOP, "OFF");
		if (flags & 16) { h->model->setMode(AIS::Mode::X); h->model->buildModel('X', 'X', sample_rate, false, &h->dev); } // Receiver.cpp:87-98,220
		else h->model->buildModel('A', 'B', sample_rate, false, &h->dev);
		h->model->Output() >> h->sink;
		h->dev.setTag(h->tag);
		if (h->file) h->file->out.Connect(&h->cc);
		if (taps && !h->mgpu) {
			AIS::ModelFrontend* fe = static_cast<AIS::ModelFrontend*>(h->model);
			for (auto& t : h->tap) t.on = true;
			*fe->C_a >> h->tap[0];
			*fe->C_b >> h->tap[1];
			h->taps_connected = true;
			if (h->mb || h->ms || h->mc) { h->ftap_connected = true; for (auto& t : h->ftap) t.on = true; }
			if (h->mb) { h->mb->FM_a.out >> h->ftap[0]; h->mb->FM_b.out >> h->ftap[1]; h->mb->FR_a.out >> h->ftap[2]; h->mb->FR_b.out >> h->ftap[3]; }
			if (h->ms) { h->ms->FM_a.out >> h->ftap[0]; h->ms->FM_b.out >> h->ftap[1]; h->ms->FR_a.out >> h->ftap[2]; h->ms->FR_b.out >> h->ftap[3]; }
			if (h->mc) { h->mc->FM_af.out >> h->ftap[0]; h->mc->FM_bf.out >> h->ftap[1]; h->mc->FR_af.out >> h->ftap[2]; h->mc->FR_bf.out >> h->ftap[3]; }
			if (h->mb) { // sampler output (what the decoder gets) and the filtered discriminator in front of it
				h->bits[0][0].on = h->bits[1][0].on = h->fmbits[0][0].on = h->fmbits[1][0].on = true;
				h->mb->sampler_a.out >> h->bits[0][0]; h->mb->sampler_b.out >> h->bits[1][0];
				h->mb->FR_a.out >> h->fmbits[0][0];    h->mb->FR_b.out >> h->fmbits[1][0];
			} else if (h->mv) { // (the engine's internals are private arrays; the 48 kHz taps and the messages are compared)
			} else if (h->ms) { // what each of the five decoders of a channel gets (Deinterleave outputs)
				for (int j = 0; j < 5; j++) {
					h->fmbits[0][j].on = h->fmbits[1][j].on = true;
					h->ms->S_a.out[j] >> h->fmbits[0][j];
					h->ms->S_b.out[j] >> h->fmbits[1][j];
				}
			} else if (h->md) {
				h->md->CGF_a.out >> h->tap[2]; h->md->CGF_b.out >> h->tap[3];
				h->md->FC_a.out >> h->tap[4];  h->md->FC_b.out >> h->tap[5];
				for (int j = 0; j < 5; j++) {
					h->bits[0][j].on = h->bits[1][j].on = true;
					if (flags & 4) { h->md->CD_a[j].out >> h->bits[0][j]; h->md->CD_b[j].out >> h->bits[1][j]; }
					else { h->md->CD_EMA_a[j].out >> h->bits[0][j]; h->md->CD_EMA_b[j].out >> h->bits[1][j]; }
				}
			} else {
				h->mc->CGF_a.out >> h->tap[2]; h->mc->CGF_b.out >> h->tap[3];
				h->mc->FC_a.out >> h->tap[4];  h->mc->FC_b.out >> h->tap[5];
				for (int j = 0; j < 5; j++) {
					h->bits[0][j].on = h->bits[1][j].on = true;
					h->fmbits[0][j].on = h->fmbits[1][j].on = true;
					h->mc->CD_EMA_a[j].out >> h->bits[0][j];
					h->mc->CD_EMA_b[j].out >> h->bits[1][j];
					h->mc->S_af.out[j] >> h->fmbits[0][j];
					h->mc->S_bf.out[j] >> h->fmbits[1][j];
				}
			}
		}
		for (int c = 0; c < 2; c++) for (int j = 0; j < 5; j++) { h->bits[c][j].connected = h->bits[c][j].on; h->fmbits[c][j].connected = h->fmbits[c][j].on; }
		return h;
	} catch (const std::exception& e) {
		fprintf(stderr, "ref_create: %s\n", e.what());
		return nullptr;
	}
}

// one Receive() call == one device block (Device/FileRAW.cpp:129-136)
int ref_feed(void* hv, const void* data, int nbytes) {
	Harness* h = (Harness*)hv;
	RAW r = { h->fmt, (void*)data, nbytes };
	auto t0 = std::chrono::high_resolution_clock::now();
	try {
		h->dev.out.Send(&r, 1, h->tag);
	} catch (const std::exception& e) { // set-up errors of a model surface at its first block (the reference catches them in CommandLine::run)
		fprintf(stderr, "ref_feed: %s\n", e.what());
		return -1;
	}
	auto t1 = std::chrono::high_resolution_clock::now();
	h->seconds += std::chrono::duration<double>(t1 - t0).count();
	return 0;
}

// Runs the whole file through the model on the reference's own threads (Device/FileRAW.cpp:199-206); returns when the run thread
// has seen the end of the input (RAWFile::isStreaming() turns false, :139-140), -1 after timeout_s, -2 if the chain stopped the process.
// calls / max_blocks (may be NULL): the number of Receive() calls the device made and the largest number of FIFO blocks in one.
int ref_play_file(void* hv, double timeout_s, int* calls, int* max_blocks) {
	Harness* h = (Harness*)hv;
	if (!h->file) return -3;
	CallCounter& cc = h->cc;
	stop = false;
	int rc = 0;
	try {
		h->file->Play();
		const auto t0 = std::chrono::steady_clock::now();
		while (h->file->isStreaming() && !stop) {
			std::this_thread::sleep_for(std::chrono::milliseconds(2));
			if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) { rc = -1; break; }
		}
		if (stop) rc = -2;
		h->file->Stop();
		h->file->Close();
	} catch (const std::exception& e) {
		fprintf(stderr, "ref_play_file: %s\n", e.what());
		rc = -4;
	}
	if (calls) *calls = cc.calls;
	if (max_blocks) *max_blocks = cc.max_size / (24 * 16 * 16384);
	return rc;
}

// bytes per GPU block for the GPU engines built from now on (0: by the device) -- what `-go GPU_BLOCK n` of the patched Receiver sets
void ref_gpu_block_bytes(int n) {
#ifdef HASMI355X
	AIS::GpuPool::instance().setBlockBytes(n);
#else
	(void)n;
#endif
}

// pipelined GPU batches: collect the last block's outputs (what the patched Receiver does when its device has delivered its last block)
void ref_flush(void* hv) {
#ifdef HASMI355X
	Harness* h = (Harness*)hv;
	if (h->mgpu) h->mgpu->Flush(h->tag);
#else
	(void)hv;
#endif
}

double ref_seconds(void* hv) { return ((Harness*)hv)->seconds; }
int ref_msg_count(void* hv) { return ((Harness*)hv)->sink.count; }

// copies the accumulated NMEA text; returns the required size
int ref_nmea(void* hv, char* dst, int cap) {
	Harness* h = (Harness*)hv;
	int n = (int)h->sink.text.size();
	if (dst && cap > 0) {
		int c = n < cap - 1 ? n : cap - 1;
		memcpy(dst, h->sink.text.data(), c);
		dst[c] = 0;
	}
	return n + 1;
}

int ref_msg_meta(void* hv, float* level, float* ppm, int cap) {
	Harness* h = (Harness*)hv;
	int n = (int)h->sink.level.size();
	for (int i = 0; i < n && i < cap; i++) { level[i] = h->sink.level[i]; ppm[i] = h->sink.ppm[i]; }
	return n;
}

// complex taps: returns number of complex samples, copies up to cap
long long ref_tap(void* hv, int which, float* dst, long long cap) {
	Harness* h = (Harness*)hv;
	auto& b = h->tap[which].buf;
	long long n = (long long)b.size();
	if (dst) memcpy(dst, b.data(), sizeof(CFLOAT32) * (size_t)(n < cap ? n : cap));
	return n;
}

// recorders on / off from now on (only those connected at creation, i.e. created with the taps flag)
void ref_set_taps(void* hv, int on) {
	Harness* h = (Harness*)hv;
	if (!h->taps_connected) return;
	for (auto& t : h->tap) t.on = on;
	for (auto& t : h->ftap) t.on = on && h->ftap_connected;
	for (int c = 0; c < 2; c++) for (int j = 0; j < 5; j++) { h->bits[c][j].on = on && h->bits[c][j].connected; h->fmbits[c][j].on = on && h->fmbits[c][j].connected; }
}

// real taps of the FM receivers: 6/7 = Demod::FM output A/B, 8/9 = Filter(Receiver) output A/B
long long ref_tapf(void* hv, int which, float* dst, long long cap) {
	Harness* h = (Harness*)hv;
	if (which < 6 || which > 9) return 0;
	auto& b = h->ftap[which - 6].buf;
	long long n = (long long)b.size();
	if (dst) memcpy(dst, b.data(), sizeof(float) * (size_t)(n < cap ? n : cap));
	return n;
}

// tag.ppm observed at each Send() into tap `which` (one per 512-window for taps 2..5)
long long ref_tap_ppm(void* hv, int which, float* dst, long long cap) {
	Harness* h = (Harness*)hv;
	auto& b = h->tap[which].ppm;
	long long n = (long long)b.size();
	if (dst) memcpy(dst, b.data(), sizeof(float) * (size_t)(n < cap ? n : cap));
	return n;
}

// hard bits of channel ch (0/1), phase j (0..4); fm=1 selects the Challenger FM branch
long long ref_bits(void* hv, int ch, int j, int fm, float* bits, float* lvl, long long* idx, long long cap) {
	Harness* h = (Harness*)hv;
	BitRec& r = fm ? h->fmbits[ch][j] : h->bits[ch][j];
	long long n = (long long)r.bits.size();
	long long c = n < cap ? n : cap;
	if (bits) memcpy(bits, r.bits.data(), sizeof(float) * (size_t)c);
	if (lvl) memcpy(lvl, r.lvl.data(), sizeof(float) * (size_t)c);
	if (idx) memcpy(idx, r.idx.data(), sizeof(long long) * (size_t)c);
	return n;
}

// ---- V2::Engine's decoder-independent stages as stand-alone functions (the structs are public, V2Engine.h:32-101)
// FreqOffset::Estimate of one 512-sample window (complex, interleaved): f and the prominence it leaves behind
void ref_v2_estimate(const float* window, float* f, float* prom) {
	V2::FreqOffset fo;
	*f = fo.Estimate((const CFLOAT32*)window);
	*prom = fo.prominence;
}
// FMDemod + FilterFL37 over n (multiple of 512) samples of a stream that starts behind the engine's all-zero look-back block
void ref_v2_fm(const float* x, int n, float* disc, float* filt) {
	V2::FMDemod fm;
	V2::FilterFL37 fl(Filters::Receiver.data());
	std::vector<CFLOAT32> zero(V2::BLOCK_SIZE, CFLOAT32(0.0f, 0.0f));
	std::vector<float> d(V2::BLOCK_SIZE), o(V2::BLOCK_SIZE);
	fm.Run(zero.data(), d.data());
	fl.Run(d.data(), o.data());
	for (int b = 0; b + V2::BLOCK_SIZE <= n; b += V2::BLOCK_SIZE) {
		fm.Run((const CFLOAT32*)x + b, disc + b);
		fl.Run(disc + b, filt + b);
	}
}

void ref_destroy(void* hv);
// CPU baseline (bench.py): `nthreads` independent receivers, one per thread, each thread pinned to cpus[i] (if
// given), its model built and its input blocks copied INSIDE the thread (first touch on the thread's own NUMA node), all started
// together, each feeding its blocks round robin until the deadline.  counts[i] = blocks thread i completed.  Returns the
// wall-clock seconds from the common start to the last thread's end.
double ref_bench_threads(int kind, int sample_rate, int fmt, const void* blocks, int nblk, int block_bytes, int nthreads, const int* cpus,
                         double seconds, long long* counts) {
	std::atomic<int> ready(0);
	std::atomic<bool> go(false);
	std::vector<std::thread> th;
	std::vector<double> t_end(nthreads, 0.0);
	const auto clk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double t_start = 0;
	for (int i = 0; i < nthreads; i++) {
		counts[i] = 0;
		th.emplace_back([&, i] {
			if (cpus && cpus[i] >= 0) {
				cpu_set_t set; CPU_ZERO(&set); CPU_SET(cpus[i], &set);
				sched_setaffinity(0, sizeof set, &set);
			}
			void* h = ref_create(kind, sample_rate, fmt, 0);
			std::vector<char> local((size_t)nblk * block_bytes);
			memcpy(local.data(), blocks, local.size());
			if (h) ref_feed(h, local.data(), block_bytes); // warm-up: page in, size every block's output vector
			ready++;
			while (!go.load()) std::this_thread::yield();
			const double deadline = t_start + seconds;
			long long k = 0;
			while (h && clk() < deadline) {
				ref_feed(h, local.data() + (size_t)(k % nblk) * block_bytes, block_bytes);
				k++;
			}
			t_end[i] = clk();
			counts[i] = k;
			if (h) ref_destroy(h);
		});
	}
	while (ready.load() < nthreads) std::this_thread::yield();
	t_start = clk();
	go = true;
	for (auto& t : th) t.join();
	double last = t_start;
	for (double e : t_end) if (e > last) last = e;
	return last - t_start;
}

// Message::ID is the process-global multi-sentence sequence counter (Marine/Message.cpp:28-39)
void ref_reset_seq(void) { AIS::Message::ID.store(0); }

void ref_destroy(void* hv) {
	Harness* h = (Harness*)hv;
	delete h->md; delete h->mc; delete h->mb; delete h->ms; delete h->mv; delete h->mgpu;
	Device::RAWFile* rf = h->file;
	delete h;
	delete rf;
}

} // extern "C"
