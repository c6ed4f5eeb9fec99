// ais-catcher_amd/host/aishost_c.cpp -- small C surface over the C++ host classes so the Python tests
// (and any FFI) can drive ModelDefaultGPU exactly like the reference's device thread drives a Model.
#include <cstring>
#include <string>
#include <vector>

#include "gpu_model.h"

using namespace aisamd;

namespace {
struct Sink : public StreamIn<AIS::Message> {
	std::string text;
	std::vector<float> level, ppm;
	int count = 0;
	void Receive(const AIS::Message* m, int len, TAG& tag) override {
		for (int i = 0; i < len; i++) {
			for (const auto& s : m[i].sentences()) { text += s; text += '\n'; }
			level.push_back(tag.level);
			ppm.push_back(tag.ppm);
			count++;
		}
	}
};
struct Model {
	ModelDefaultGPU m;
	Sink sink;
	TAG tag;
	Format fmt = Format::CF32;
	std::string err;
};
} // namespace

extern "C" {

void* aishost_batch_create(const aisgpu_cfg* cfg, char* errbuf, int errcap) {
	try {
		return new GpuBatch(*cfg);
	} catch (const std::exception& e) {
		if (errbuf && errcap > 0) { strncpy(errbuf, e.what(), errcap - 1); errbuf[errcap - 1] = 0; }
		return nullptr;
	}
}
void aishost_batch_destroy(void* b) { delete (GpuBatch*)b; }

// batch == NULL and detached == 0: stand-alone single receiver with its own context
// detached != 0: no GPU at all, only aishost_model_replay() may be used (host-logic tests)
void* aishost_model_create(void* batch, int rx, int sample_rate, int block_len, int input_format, char ch1, char ch2,
                           int detached, int model, char* errbuf, int errcap) {
	Model* m = new Model();
	try {
		m->fmt = input_format == AISGPU_FMT_CU8 ? Format::CU8 : input_format == AISGPU_FMT_CS8 ? Format::CS8 : input_format == AISGPU_FMT_CS16 ? Format::CS16 : Format::CF32;
		m->m.setFormat(m->fmt);
		m->m.setBlockLength(block_len);
		m->m.setChallenger((model & 0xff) == AISGPU_MODEL_CHALLENGER);
		m->m.setBase((model & 0xff) == AISGPU_MODEL_BASE);
		m->m.setStandard((model & 0xff) == AISGPU_MODEL_STANDARD);
		m->m.setEngineV2((model & 0xff) == AISGPU_MODEL_V2);
		m->m.setGpuDecode((model & 0x100) != 0);
		m->m.setModeX((model & 0x400) != 0);      // bit 10: channel mode X
		// (bit 8 of `model`: AISGPU_FLAG_GPU_DECODE for a stand-alone receiver)  FP_DS first: like the reference's SetKey, it clears MA
		m->m.setFixedPoint((model & 0x200) != 0);    // bit 9: AISGPU_FLAG_FP_DS
		m->m.setMovingAverage((model & 0x800) != 0); // bit 11: AISGPU_FLAG_MA_DS (`-go MA on`)
		if (batch) m->m.useBatch((GpuBatch*)batch, rx);
		if (!detached) m->m.buildModel(ch1, ch2, sample_rate, false, nullptr);
		else m->m.wireDecoders(ch1, ch2); // no GPU context
		m->m.Output() >> m->sink;
		return m;
	} catch (const std::exception& e) {
		if (errbuf && errcap > 0) { strncpy(errbuf, e.what(), errcap - 1); errbuf[errcap - 1] = 0; }
		delete m;
		return nullptr;
	}
}
void aishost_model_destroy(void* mv) { delete (Model*)mv; }

// One device block, exactly what the reference's run thread sends (RAW{format,data,size}, Device/FileRAW.cpp:135)
int aishost_model_receive(void* mv, const void* data, int nbytes) {
	Model* m = (Model*)mv;
	RAW r = { m->fmt, (void*)data, nbytes };
	return m->m.Receive(&r, m->tag); // AISGPU_* status of the block
}
// end of this receiver's input: a shared batch stops waiting for it
void aishost_model_leave(void* mv) { ((Model*)mv)->m.Chain().detach(); }
void aishost_batch_set_timeout(void* b, int ms) { ((GpuBatch*)b)->setTimeout(ms); }
void aishost_batch_set_pipelined(void* b, int on) { ((GpuBatch*)b)->setPipelined(on != 0); }
// after the last block of a pipelined batch: collect (and decode) the last block's outputs
void aishost_model_flush(void* mv) { Model* m = (Model*)mv; m->m.Flush(m->tag); }
int aishost_batch_active(void* b) { return ((GpuBatch*)b)->activeReceivers(); }

int aishost_model_replay(void* mv, int ch, long long first_group, long long first_sample48, int n_groups,
                         const uint32_t* const* bits5, const float* lvl, int n_windows, const float* ppm, const uint32_t* fm_bits) {
	Model* m = (Model*)mv;
	aisgpu_out o;
	memset(&o, 0, sizeof o);
	o.n_groups = n_groups; o.first_group = first_group; o.first_sample48 = first_sample48;
	for (int j = 0; j < 5; j++) o.bits[j] = bits5[j];
	o.lvl = lvl; o.n_windows = n_windows; o.ppm = ppm; o.fm_bits = fm_bits;
	m->m.replay(ch, o, m->tag);
	return 0;
}

// ModelEngineV2 host logic without a GPU: one block of a 48 kHz channel (n complex samples), as the device would hand it over
// one frame of the device decoders (aisgpu_frame) to the tail of its decoder: validation, NMEA text.  For callers that hold the
// frames themselves (bench.py's gate of the timed --gpu-decode run, on a detached model); tag.ppm / tag.level are the caller's.
int aishost_model_frame(void* mv, const aisgpu_frame* f) {
	Model* m = (Model*)mv;
	if (!f) return AISGPU_ERR_ARG;
	m->tag.sample_idx = f->end_idx;
	m->m.Chain().emitFrame(*f, m->tag);
	return 0;
}

int aishost_model_feed48(void* mv, int ch, const float* iq, int n) {
	Model* m = (Model*)mv;
	(ch == 0 ? m->m.Chain().outC48a : m->m.Chain().outC48b).Send((const CFLOAT32*)iq, n, m->tag);
	return 0;
}

int aishost_model_msg_count(void* mv) { return ((Model*)mv)->sink.count; }
int aishost_model_nmea(void* mv, char* dst, int cap) {
	Model* m = (Model*)mv;
	int n = (int)m->sink.text.size();
	if (dst && cap > 0) { int c = n < cap - 1 ? n : cap - 1; memcpy(dst, m->sink.text.data(), c); dst[c] = 0; }
	return n + 1;
}
int aishost_model_msg_meta(void* mv, float* level, float* ppm, int cap) {
	Model* m = (Model*)mv;
	int n = (int)m->sink.level.size();
	for (int i = 0; i < n && i < cap; i++) { level[i] = m->sink.level[i]; ppm[i] = m->sink.ppm[i]; }
	return n;
}
void aishost_reset_sequence(void) { AIS::Message::resetSequence(); }

} // extern "C"
