// ais-catcher_amd/host/gpu_model.h -- the GPU chain behind the reference's block API.
//
//   GpuBatch        owns one aisgpu_t (include/aisgpu.h) for R batched receivers on one GPU; receiver
//                   threads hand in their Receive() blocks, the last one to arrive launches the chain,
//                   every thread then replays ITS receiver's symbol decisions into its decoders.
//   GpuChain        : StreamIn<CFLOAT32>, StreamIn<CU8> -- the drop-in for everything between
//                   Util::ConvertRAW and the AIS::Decoder objects of AIS::ModelDefault
//                   (reference DSP/Model.cpp:27-356 + :520-577).  Exposes Connection<FLOAT32> outA[5],
//                   outB[5], fed in exactly the reference's order: channel A's whole block
//                   first, then channel B; per 5-sample group the phases j = 0..4 with
//                   tag.sample_idx = 5g+j, tag.sample_lvl = level[g], tag.ppm = ppm[window of g].
//   ModelDefaultGPU same contract as AIS::Model (reference DSP/Model.h:76-126): buildModel(CH1, CH2,
//                   sample_rate, timerOn, device) wires GpuChain -> 2 x 5 AIS::Decoder with the Reset mesh
//                   of Model.cpp:566-573 -> Output().
// Errors: set-up problems throw std::runtime_error like the reference's models (Model.cpp:109-110);
// run-time failures inside Receive() are reported through the error callback and stop the chain
// (the reference logs and calls StopRequest(), Device/FileRAW.cpp:111-115).
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/aisgpu.h"
#include "ais_frame.h"
#include "gpu_batch.h"
#include "stream.h"
#include "v2_engine.h"

namespace aisamd {

constexpr int N_SAMPLES_PER_SYMBOL = 5; // reference DSP/Model.h:37

// DSP::SimplePLL of the reference (DSP/DSP.h:35-50, DSP/DSP.cpp:28-57): the symbol sampler of ModelBase.  Sequential, one
// float of state, and switched between a fast and a slow loop by the decoder behind it -- host logic by nature.
// DSP::Deinterleave (DSP/DSP.h:51-74): round-robin over n outputs, one sample per Send, stamps tag.sample_idx
template <typename T>
class Deinterleave : public StreamIn<T> {
	int lastSymbol = 0;
	long long sample_idx = 0;

public:
	std::vector<Connection<T>> out;
	void setConnections(int n) { out.resize(n); }
	void Receive(const T* data, int len, TAG& tag) override {
		for (int i = 0; i < len; i++) {
			tag.sample_idx = sample_idx++;
			out[lastSymbol].Send(&data[i], 1, tag);
			lastSymbol = (lastSymbol + 1) % (int)out.size();
		}
	}
};

class SimplePLL : public SimpleStreamInOut<FLOAT32, FLOAT32>, public SignalIn<DecoderSignals> {
	bool prev = false;
	float PLL = 0.0f;
	bool FastPLL = true;

public:
	void Receive(const FLOAT32* data, int len, TAG& tag) override {
		for (int i = 0; i < len; i++) {
			const bool bit = data[i] > 0;
			if (bit != prev) PLL += (0.5f - PLL) * (FastPLL ? 0.6f : 0.05f);
			PLL += 0.2f;
			if (PLL >= 1.0f) {
				Send(&data[i], 1, tag);
				PLL -= (int)PLL;
			}
			prev = bit;
		}
	}
	void Signal(const DecoderSignals& in) override {
		if (in == DecoderSignals::StartTraining) FastPLL = true;
		else if (in == DecoderSignals::StopTraining) FastPLL = false;
	}
};

class GpuChain : public StreamIn<CFLOAT32>, public StreamIn<CU8>, public StreamIn<CS8>, public StreamIn<CS16> {
	GpuBatch* batch = nullptr;
	int rx = 0;
	std::function<void(const std::string&)> on_error;
	// AISGPU_FLAG_GPU_DECODE: the decoders' state machines ran on the device; a completed frame goes to its decoder's tail
	std::function<void(const aisgpu_frame&, TAG&)> on_frame;
	bool failed = false;
	int last_status = AISGPU_OK;

	void process(const void* data, int len, TAG& tag);

public:
	Connection<FLOAT32> outA[N_SAMPLES_PER_SYMBOL], outB[N_SAMPLES_PER_SYMBOL];
	// ModelChallenger's FM branch (reference S_af / S_bf, DSP/Model.cpp:638-639); unconnected for ModelDefault
	Connection<FLOAT32> outAf[N_SAMPLES_PER_SYMBOL], outBf[N_SAMPLES_PER_SYMBOL];
	// ModelBase: the filtered FM discriminator of each channel (FR_a / FR_b, DSP/Model.cpp:431-432); only its sign is known
	// here, and only its sign is looked at downstream (DSP.cpp:30, Marine/AIS.h:96)
	Connection<FLOAT32> outFMa, outFMb;
	// ModelEngineV2: the 48 kHz channels themselves (C_a / C_b, DSP/Model.cpp:345-346), one Send per block
	Connection<CFLOAT32> outC48a, outC48b;

	void attach(GpuBatch* b, int receiver) { batch = b; rx = receiver; }
	void detach() { if (batch) batch->leave(rx); batch = nullptr; } // end of this receiver's stream: the batch stops waiting for it
	void flush(TAG& tag) { if (batch && batch->isPipelined()) process(nullptr, 0, tag); } // pipelined batch: collect the last block's outputs
	int status() const { return last_status; }                      // AISGPU_* status of the last Receive()
	void setErrorHandler(std::function<void(const std::string&)> f) { on_error = f; }
	void setFrameHandler(std::function<void(const aisgpu_frame&, TAG&)> f) { on_frame = f; }
	void emitFrame(const aisgpu_frame& f, TAG& tag) { if (on_frame) on_frame(f, tag); } // a device frame to the tail of ITS decoder (what process() does for every frame of a block)
	// ModelEngineV2: called before / after the 48 kHz samples of a device block go out on outC48x (device assist of the engines)
	std::function<void(int ch, const aisgpu_out&, int first_sample)> on_c48;
	std::function<void(int ch, int L)> on_c48_done;
	void Receive(const CFLOAT32* data, int len, TAG& tag) override { process(data, len, tag); }
	void Receive(const CU8* data, int len, TAG& tag) override { process(data, len, tag); }
	// the raw integer formats Util::ConvertRAW turns into CFLOAT32 (Utilities/StreamHelpers.cpp:91-106): converted on the device
	void Receive(const CS8* data, int len, TAG& tag) override { process(data, len, tag); }
	void Receive(const CS16* data, int len, TAG& tag) override { process(data, len, tag); }
	// Replay one channel's symbol decisions of a block into the five phase outputs (host logic,
	// also used stand-alone by the CPU tests).
	// (n0, n1: only the part of the block whose 48 kHz samples [n0, n1) complete it -- Rotate's sub-blocks on the decimate-by-3 ladders)
	static void replay(Connection<FLOAT32>* out, const aisgpu_out& o, TAG& tag, int n0 = 0, int n1 = 1 << 30);
	// ModelChallenger: the reference throttles the CGF output one sample at a time into both branches
	// (Deinterleave n=1, Model.cpp:630-639), so per 48 kHz sample N: first the coherent branch (which fires its five
	// decoders when N completes a group), then FM decoder N % 5.
	static void replayChallenger(Connection<FLOAT32>* coh, Connection<FLOAT32>* fm, const aisgpu_out& o, TAG& tag, int n0 = 0, int n1 = 1 << 30);
	// ModelBase: every 48 kHz sample of the block, in order, into the sampler
	static void replayBase(Connection<FLOAT32>& fm, const aisgpu_out& o, TAG& tag, int n0 = 0, int n1 = 1 << 30);
};

class ModelDefaultGPU {
	GpuBatch* batch = nullptr;
	bool own_batch = false;
	GpuChain chain;
	AIS::Decoder DEC_a[N_SAMPLES_PER_SYMBOL], DEC_b[N_SAMPLES_PER_SYMBOL];
	AIS::Decoder DEC_af[N_SAMPLES_PER_SYMBOL], DEC_bf[N_SAMPLES_PER_SYMBOL]; // ModelChallenger only
	bool challenger = false;
	bool gpu_decode = false; // AIS::Decoder state machines on the device (ModelDefault only)
	bool v2 = false; // AIS::ModelEngineV2 wiring: outC48 -> V2Engine (six decoders inside) per channel
	V2Engine V2_a, V2_b;
	float dd_train = 0.75f, dd_weight = 0.86f; // DSP/Model.h:272
	bool standard = false; // AIS::ModelStandard wiring: outFM -> Deinterleave(5) -> DEC_a/b[5] with their Reset mesh
	Deinterleave<FLOAT32> S_a, S_b;
	bool base = false; // AIS::ModelBase wiring: outFM -> SimplePLL -> one decoder per channel, decoder -> sampler feedback
	SimplePLL sampler_a, sampler_b;
	AIS::Decoder DEC_base_a, DEC_base_b;
	StreamOut<AIS::Message> output;
	int own_mmsi = -1, station = 0;
	int block_len = 786432;
	Format format = Format::CF32;
	bool CGF_wide = true, droop_compensation = true;
	bool fixedpointDS = false; // KEY_SETTING_FP_DS (Model.cpp:362-363)
	bool MA_DS = false;        // KEY_SETTING_MA (Model.cpp:376-380): DownsampleMovingAverage instead of the CIC5 ladder
	bool mode_x = false;       // channel mode X (Model.cpp:35-107)

	struct Fan : public StreamIn<AIS::Message> { // PassThrough<Message> of the reference (Model.h:88)
		StreamOut<AIS::Message>* o = nullptr;
		void Receive(const AIS::Message* d, int len, TAG& tag) override { o->Send(d, len, tag); }
	} fan;

public:
	ModelDefaultGPU() {}
	~ModelDefaultGPU();
	// share one GPU context between several receivers (receiver index rx inside the batch)
	void useBatch(GpuBatch* b, int rx) { batch = b; chain.attach(b, rx); }
	void setBlockLength(int n) { block_len = n; }
	void setFormat(Format f) { format = f; }
	void setOwnMMSI(int m) { own_mmsi = m; }
	void setAFCWide(bool b) { CGF_wide = b; }
	void setDroop(bool b) { droop_compensation = b; }
	void setFixedPoint(bool b) { fixedpointDS = b; MA_DS = false; } // KEY_SETTING_FP_DS clears MA_DS whatever its argument (Model.cpp:362-365)
	void setMovingAverage(bool b) { MA_DS = b; }
	void setModeX(bool b) { mode_x = b; }          // AIS::Model::setMode(Mode::X) (DSP/Model.h:104, Receiver.cpp:87-98): one channel, 12k .. 192k
	void setChallenger(bool b) { challenger = b; } // AIS::ModelChallenger wiring (Model.cpp:601-678) instead of ModelDefault
	void setBase(bool b) { base = b; }             // AIS::ModelBase wiring (Model.cpp:419-438)
	void setStandard(bool b) { standard = b; }     // AIS::ModelStandard wiring (Model.cpp:484-518)
	void setEngineV2(bool b) { v2 = b; }           // AIS::ModelEngineV2 wiring (Model.cpp:440-463)
	void setV2Weights(float train, float track) { dd_train = train; dd_weight = track; } // KEY_SETTING_DD_TRAIN / DD_WEIGHT
	void setGpuDecode(bool b) { gpu_decode = b; }  // frames from aisgpu_frames() instead of replaying decisions
	// same signature as AIS::Model::buildModel (the Device* of the reference is only used for wiring there)
	void buildModel(char CH1, char CH2, int sample_rate, bool timerOn, void* device);
	// decoder wiring only, no GPU context (CPU tests of the replay/decoder host logic)
	void wireDecoders(char CH1, char CH2);
	StreamOut<AIS::Message>& Output() { return output; }
	GpuChain& Chain() { return chain; }
	// entry used by the C API: the RAW block as the device thread delivers it (Device/FileRAW.cpp:135)
	// (returns the AISGPU_* status of the block: the reference's Receive() is void and reports through Error()/StopRequest(),
	// Device/FileRAW.cpp:111-115 -- here the caller gets the code, the error handler the text)
	int Receive(const RAW* raw, TAG& tag);
	void Flush(TAG& tag) { chain.flush(tag); } // after the last block of a pipelined batch
	// CPU-only replay entry (host-logic tests): decisions produced elsewhere
	void replay(int ch, const aisgpu_out& o, TAG& tag) {
		if (base || standard) GpuChain::replayBase(ch == 0 ? chain.outFMa : chain.outFMb, o, tag);
		else if (challenger) GpuChain::replayChallenger(ch == 0 ? chain.outA : chain.outB, ch == 0 ? chain.outAf : chain.outBf, o, tag);
		else GpuChain::replay(ch == 0 ? chain.outA : chain.outB, o, tag);
	}
};

// same contract, AIS::ModelChallenger wiring ("-m 4")
class ModelChallengerGPU : public ModelDefaultGPU {
public:
	ModelChallengerGPU() { setChallenger(true); }
};

// AIS::ModelEngineV2 wiring ("-m 11")
class ModelEngineV2GPU : public ModelDefaultGPU {
public:
	ModelEngineV2GPU() { setEngineV2(true); }
};

// AIS::ModelStandard wiring ("-m 0")
class ModelStandardGPU : public ModelDefaultGPU {
public:
	ModelStandardGPU() { setStandard(true); }
};

// AIS::ModelBase wiring ("-m 1")
class ModelBaseGPU : public ModelDefaultGPU {
public:
	ModelBaseGPU() { setBase(true); }
};

} // namespace aisamd
