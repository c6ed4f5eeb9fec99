// ais-catcher_amd/host/gpu_model.cpp -- see gpu_model.h
#include "gpu_model.h"
#include <cstring>

#include <algorithm>
#include <chrono>
#include <cstdlib>

namespace aisamd {

void GpuChain::replay(Connection<FLOAT32>* out, const aisgpu_out& o, TAG& tag, int n0, int n1) {
	for (int g = 0; g < o.n_groups; g++) {
		const long long n_last = 5 * (o.first_group + g) + 4; // the sample that completes the group
		if (n_last - o.first_sample48 < n0 || n_last - o.first_sample48 >= n1) continue;
		const int w = (int)((n_last - o.first_sample48) / 512);
		if (o.ppm && w >= 0 && w < o.n_windows) tag.ppm = o.ppm[w];
		if (tag.mode & 1) tag.sample_lvl = o.lvl[g];
		for (int j = 0; j < N_SAMPLES_PER_SYMBOL; j++) {
			tag.sample_idx = 5 * (o.first_group + g) + j;
			const FLOAT32 b = ((o.bits[j][g >> 5] >> (g & 31)) & 1u) ? 1.0f : -1.0f;
			out[j].Send(&b, 1, tag);
		}
	}
}

void GpuChain::replayChallenger(Connection<FLOAT32>* coh, Connection<FLOAT32>* fm, const aisgpu_out& o, TAG& tag, int n0, int n1) {
	const int L = o.n_windows * 512;
	for (int n = n0 > 0 ? n0 : 0; n < L && n < n1; n++) {
		const long long N = o.first_sample48 + n;
		if (o.ppm) tag.ppm = o.ppm[n >> 9]; // set by the CGF before it hands the window to the throttle (DSP.cpp:484)
		if (N % 5 == 4) { // ScatterPLL has its five samples (DSP.h:101-113)
			const long long g = N / 5;
			const int gi = (int)(g - o.first_group);
			if (tag.mode & 1) tag.sample_lvl = o.lvl[gi];
			for (int j = 0; j < N_SAMPLES_PER_SYMBOL; j++) {
				tag.sample_idx = 5 * g + j;
				const FLOAT32 b = ((o.bits[j][gi >> 5] >> (gi & 31)) & 1u) ? 1.0f : -1.0f;
				coh[j].Send(&b, 1, tag);
			}
		}
		tag.sample_idx = N; // Deinterleave<FLOAT32> S_af counts every sample (DSP.h:65-71)
		const FLOAT32 f = ((o.fm_bits[n >> 5] >> (n & 31)) & 1u) ? 1.0f : -1.0f;
		fm[(int)(N % 5)].Send(&f, 1, tag);
	}
}

void GpuChain::replayBase(Connection<FLOAT32>& fm, const aisgpu_out& o, TAG& tag, int n0, int n1) {
	const int L = o.n_windows * 512;
	for (int n = n0 > 0 ? n0 : 0; n < L && n < n1; n++) { // FM, Filter and SimplePLL leave the tag alone (Demod.cpp:27-37, DSP.cpp:249-280, 28-44)
		const FLOAT32 f = ((o.fm_bits[n >> 5] >> (n & 31)) & 1u) ? 1.0f : -1.0f;
		fm.Send(&f, 1, tag);
	}
}

void GpuChain::process(const void* data, int len, TAG& tag) {
	if (!batch) { last_status = AISGPU_ERR_STATE; return; }
	if (failed) return; // (last_status keeps the code of the failure)
	int rc = batch->submitAndWait(rx, data, len);
	last_status = rc;
	if (rc != AISGPU_OK) {
		failed = true;
		if (on_error) on_error(std::string("GpuChain: ") + aisgpu_strerror(rc) + ": " + batch->lastError());
		return;
	}
	// One downstream block per Receive() call of the chain behind the (optional) resampler; within it Rotate hands
	// the whole block to channel A before channel B (reference DSP/DSP.cpp:312-313)
	const int nsub = batch->outCount(); // (a pipelined batch serves the PREVIOUS block's outputs here: none after the first call)
	if (nsub == 0) return;
	if (batch->config().flags & AISGPU_FLAG_GPU_DECODE) { // the device ran the decoders: only completed frames come back
		const auto frames = batch->frames(); // (this generation's list, valid for as long as this reference lives)
		for (size_t i = 0; i < frames->size(); i++) {
			const aisgpu_frame& f = (*frames)[i];
			if (f.rx != rx || f.sub >= nsub) continue;
			aisgpu_out o;
			if ((rc = batch->fetch(f.sub, rx, f.ch, &o)) != AISGPU_OK) { failed = true; last_status = rc; return; }
			if (o.n_groups > 0) { // (the FM receivers -- ModelBase / ModelStandard -- never touch tag.ppm / tag.sample_lvl)
				// what the tag held when the frame closed: a coherent decoder runs when its group completes (sample 5g+4, like replay()),
				// an FM decoder of ModelChallenger with its own sample (end_idx)
				const long long n_last = f.phase >= 5 ? f.end_idx : 5 * (o.first_group + f.group) + 4;
				const int w = (int)((n_last - o.first_sample48) / 512);
				if (o.ppm && w >= 0 && w < o.n_windows) tag.ppm = o.ppm[w];
				if (tag.mode & 1) tag.sample_lvl = o.lvl[f.group];
			}
			tag.sample_idx = f.end_idx;
			if (on_frame) on_frame(f, tag);
		}
		return;
	}
	// On the decimate-by-3 ladders DownsampleKFilter hands its output on in blocks of 8192 samples (DSP/DSP.h:193), so Rotate --
	// and with it the A-then-B order -- works on 4096 samples at 48 kHz at a time, however long the input block was.
	const aisgpu_cfg& cf = batch->config();
	bool by3 = false; // the smallest bucket >= rate is a decimate-by-3 one (Model.cpp:129-145), exact or resampled into
	if (!(cf.flags & AISGPU_FLAG_MODE_X)) {
		static const int b2[8] = { 96000, 192000, 384000, 768000, 1536000, 3072000, 6144000, 12288000 };
		static const int b3[4] = { 288000, 576000, 1152000, 2304000 };
		int best = 0;
		for (int i = 0; i < 8 && !best; i++) if (b2[i] >= cf.sample_rate) best = b2[i];
		const int n3 = (cf.flags & AISGPU_FLAG_DSK) ? 4 : 1;
		for (int i = 0; i < n3; i++) if (b3[i] >= cf.sample_rate && (!best || b3[i] < best)) { by3 = true; break; }
		if (cf.flags & AISGPU_FLAG_MA_DS) by3 = true; // DownsampleMovingAverage hands on blocks of 8192 samples too (DSP.h:128)
	}
	for (int s = 0; s < nsub; s++) {
		aisgpu_out o[2];
		for (int ch = 0; ch < 2; ch++)
			if ((rc = batch->fetch(s, rx, ch, &o[ch])) != AISGPU_OK) { failed = true; last_status = rc; return; }
		const int L = o[0].n_windows * 512, step = by3 ? 4096 : L;
		for (int n0 = 0; n0 < L; n0 += step) {
			for (int ch = 0; ch < 2; ch++) {
				const aisgpu_out& c = o[ch];
				const int n1 = n0 + step < L ? n0 + step : L;
				if (c.c48) {
					if (on_c48) on_c48(ch, c, n0);
					(ch == 0 ? outC48a : outC48b).Send((const CFLOAT32*)c.c48 + n0, n1 - n0, tag);
					if (on_c48_done && n1 == L) on_c48_done(ch, L);
				}
				else if (c.fm_bits && c.n_groups == 0) replayBase(ch == 0 ? outFMa : outFMb, c, tag, n0, n1);
				else if (c.fm_bits) replayChallenger(ch == 0 ? outA : outB, ch == 0 ? outAf : outBf, c, tag, n0, n1);
				else replay(ch == 0 ? outA : outB, c, tag, n0, n1);
			}
		}
	}
}

ModelDefaultGPU::~ModelDefaultGPU() {
	if (own_batch) delete batch;
	else chain.detach(); // a shared batch stops waiting for this receiver
}

void ModelDefaultGPU::buildModel(char CH1, char CH2, int sample_rate, bool /*timerOn*/, void* /*device*/) {
	if (!batch) { // stand-alone receiver: a batch of one
		aisgpu_cfg c;
		aisgpu_default_cfg(&c);
		c.sample_rate = sample_rate;
		c.n_receivers = 1;
		c.block_len = block_len;
		c.input_format = format == Format::CU8 ? AISGPU_FMT_CU8 : format == Format::CS8 ? AISGPU_FMT_CS8 : format == Format::CS16 ? AISGPU_FMT_CS16 : AISGPU_FMT_CF32;
		c.afc_wide = CGF_wide;
		c.droop = droop_compensation;
		c.model = v2 ? AISGPU_MODEL_V2 : standard ? AISGPU_MODEL_STANDARD : base ? AISGPU_MODEL_BASE : challenger ? AISGPU_MODEL_CHALLENGER : AISGPU_MODEL_DEFAULT;
		if (gpu_decode) c.flags |= AISGPU_FLAG_GPU_DECODE;
		if (fixedpointDS) c.flags |= AISGPU_FLAG_FP_DS;
		if (MA_DS) c.flags |= AISGPU_FLAG_MA_DS;
		if (mode_x) c.flags |= AISGPU_FLAG_MODE_X;
		batch = new GpuBatch(c); // throws std::runtime_error on unsupported rate / missing GPU
		own_batch = true;
		chain.attach(batch, 0);
	} else if (batch->config().sample_rate != sample_rate) {
		throw std::runtime_error("ModelDefaultGPU: batch was created for a different sample rate");
	}
	wireDecoders(CH1, CH2);
}

void ModelDefaultGPU::wireDecoders(char CH1, char CH2) {
	fan.o = &output;
	// AISGPU_FLAG_GPU_DECODE: the state machines ran on the device, a completed frame goes to the tail of ITS decoder
	// (phase 0..4: DEC_x[phase] -- DEC_base_x for ModelBase --, 5..9: the FM decoders DEC_xf[phase - 5] of ModelChallenger)
	chain.setFrameHandler([this](const aisgpu_frame& f, TAG& tag) {
		AIS::Decoder* d;
		if (v2) { // the engine ran on the device (kv2_engine): decoder 0..4 behind the trackers, 5 the FM decoder; group = the bits of tag.ppm
			d = &(f.ch == 0 ? V2_a : V2_b).getDecoder(f.phase < V2Engine::N_DECODERS ? f.phase : 0);
			memcpy(&tag.ppm, &f.group, sizeof(float));
		} else if (base) d = f.ch == 0 ? &DEC_base_a : &DEC_base_b;
		else if (f.phase >= 5) d = &(f.ch == 0 ? DEC_af : DEC_bf)[f.phase - 5];
		else d = &(f.ch == 0 ? DEC_a : DEC_b)[f.phase];
		d->emitFrame(f.data, f.position, f.level_sum, f.start_idx, f.end_idx, tag);
	});
	if (base) { // Model.cpp:428-435
		DEC_base_a.setOrigin(CH1, station, own_mmsi);
		DEC_base_b.setOrigin(CH2, station, own_mmsi);
		chain.outFMa >> sampler_a; sampler_a.out >> DEC_base_a;
		chain.outFMb >> sampler_b; sampler_b.out >> DEC_base_b;
		DEC_base_a.out.Connect(&fan);
		DEC_base_b.out.Connect(&fan);
		DEC_base_a.DecoderMessage.Connect(sampler_a);
		DEC_base_b.DecoderMessage.Connect(sampler_b);
		return;
	}
	if (v2) { // Model.cpp:440-463
		V2_a.setOrigin(CH1, station, own_mmsi);
		V2_b.setOrigin(CH2, station, own_mmsi);
		V2_a.setWeights(dd_train, dd_weight);
		V2_b.setWeights(dd_train, dd_weight);
		chain.outC48a >> V2_a;
		chain.outC48b >> V2_b;
		// what the device computed ahead for the engine blocks of this data (estimates, energies, discriminator signs)
		chain.on_c48 = [this](int ch, const aisgpu_out& o, int n0) {
			(ch == 0 ? V2_a : V2_b).setAssist(o.v2_f, o.v2_prom, o.v2_energy, o.v2_f ? o.fm_bits : nullptr, n0 / V2Engine::BLOCK);
		};
		chain.on_c48_done = [this](int ch, int L) { (ch == 0 ? V2_a : V2_b).finishAssist(L); };
		for (int i = 0; i < V2Engine::N_DECODERS; i++) {
			V2_a.getDecoder(i).out.Connect(&fan);
			V2_b.getDecoder(i).out.Connect(&fan);
		}
		return;
	}
	if (standard) { // Model.cpp:484-518
		S_a.setConnections(N_SAMPLES_PER_SYMBOL);
		S_b.setConnections(N_SAMPLES_PER_SYMBOL);
		chain.outFMa >> S_a;
		chain.outFMb >> S_b;
		for (int i = 0; i < N_SAMPLES_PER_SYMBOL; i++) {
			DEC_a[i].setOrigin(CH1, station, own_mmsi);
			DEC_b[i].setOrigin(CH2, station, own_mmsi);
			S_a.out[i] >> DEC_a[i];
			S_b.out[i] >> DEC_b[i];
			DEC_a[i].out.Connect(&fan);
			DEC_b[i].out.Connect(&fan);
			for (int j = 0; j < N_SAMPLES_PER_SYMBOL; j++)
				if (i != j) { DEC_a[i].DecoderMessage.Connect(DEC_a[j]); DEC_b[i].DecoderMessage.Connect(DEC_b[j]); }
		}
		return;
	}
	for (int i = 0; i < N_SAMPLES_PER_SYMBOL; i++) {
		DEC_a[i].setOrigin(CH1, station, own_mmsi);
		DEC_b[i].setOrigin(CH2, station, own_mmsi);
		chain.outA[i] >> DEC_a[i];
		chain.outB[i] >> DEC_b[i];
		DEC_a[i].out.Connect(&fan);
		DEC_b[i].out.Connect(&fan);
		if (challenger) {
			DEC_af[i].setOrigin(CH1, station, own_mmsi);
			DEC_bf[i].setOrigin(CH2, station, own_mmsi);
			chain.outAf[i] >> DEC_af[i];
			chain.outBf[i] >> DEC_bf[i];
			DEC_af[i].out.Connect(&fan);
			DEC_bf[i].out.Connect(&fan);
		}
		for (int j = 0; j < N_SAMPLES_PER_SYMBOL; j++) {
			if (challenger) { // coherent and FM decoders of a channel reset each other too (Model.cpp:658-674)
				DEC_af[i].DecoderMessage.Connect(DEC_a[j]);
				DEC_a[i].DecoderMessage.Connect(DEC_af[j]);
				DEC_bf[i].DecoderMessage.Connect(DEC_b[j]);
				DEC_b[i].DecoderMessage.Connect(DEC_bf[j]);
			}
			if (i != j) { // a decoder that finds a message resets its four siblings (Model.cpp:566-573)
				DEC_a[i].DecoderMessage.Connect(DEC_a[j]);
				DEC_b[i].DecoderMessage.Connect(DEC_b[j]);
				if (challenger) {
					DEC_af[i].DecoderMessage.Connect(DEC_af[j]);
					DEC_bf[i].DecoderMessage.Connect(DEC_bf[j]);
				}
			}
		}
	}
}

int ModelDefaultGPU::Receive(const RAW* raw, TAG& tag) {
	if (raw->format == Format::CU8) chain.Receive((const CU8*)raw->data, raw->size / 2, tag);
	else if (raw->format == Format::CF32) chain.Receive((const CFLOAT32*)raw->data, raw->size / (int)sizeof(CFLOAT32), tag);
	else if (raw->format == Format::CS8) chain.Receive((const CS8*)raw->data, raw->size / 2, tag);
	else if (raw->format == Format::CS16) chain.Receive((const CS16*)raw->data, raw->size / 4, tag);
	else return AISGPU_ERR_ARG;
	return chain.status();
}

} // namespace aisamd
