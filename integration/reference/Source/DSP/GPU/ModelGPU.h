/*
 * Source/DSP/GPU/ModelGPU.h -- the file a maintainer adds to the REFERENCE tree (jvde-github/AIS-catcher v0.70) to run the
 * hot path of AIS::ModelDefault / ModelChallenger / ModelStandard / ModelBase / ModelEngineV2 on an MI355X through the C ABI of libaisgpu.so
 * (include/aisgpu.h).
 *
 * It is written against the reference's own headers -- Stream.h (StreamIn / Connection, Library/Stream.h:36-167), Model.h
 * (AIS::Model, DSP/Model.h:76-126), AIS.h (AIS::Decoder, Marine/AIS.h:38-191), DSP.h (SimplePLL, Deinterleave), Device.h -- and
 * calls nothing but the C ABI and GpuBatch (ais-catcher_amd/host/gpu_batch.{h,cpp}: the thread meeting point, C ABI + standard
 * library only, added to the tree next to this file).  oracle/Makefile (target refgpu) compiles it together with the unmodified
 * reference sources; tests/test_gpu_parity.py checks that the GPU engines print the same NMEA as the reference's CPU engines from
 * the same binary -- one receiver alone, and EIGHT receivers in one process sharing ONE GPU context.
 *
 * Replaced: everything between the device's RAW output and the AIS::Decoder objects (DSP/Model.cpp:27-356, 419-438, 484-518,
 *           520-577, 601-678).
 * Kept:     the reference's Device, TAG, AIS::Decoder (with its Reset mesh), DSP::SimplePLL, DSP::Deinterleave, AIS::Message,
 *           outputs, Setting/SetKey.
 *
 * Batching.  The reference runs one thread per receiver (Device/FileRAW.cpp:205-206), each delivering its own blocks; one GPU wants
 * the blocks of ALL receivers in one launch.  Every GPU model registers with the process-wide GpuPool when it is built
 * (buildModel: all receivers are built before any device starts, Application/Receiver.cpp:155-195); the receivers whose
 * configuration is the same (rate, engine, options) form a group, and the group's context is created by the first block that
 * arrives, for as many receivers as have registered.  From then on the receiver threads meet in GpuBatch::submitAndWait() once per
 * block.  A receiver that stops delivering is evicted after a timeout; a model that is destroyed leaves its group.
 */
#pragma once

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "Model.h"
#include "AIS.h"
#include "DSP.h"
#include "V2Engine.h"
#include "aisgpu.h"
#include "gpu_batch.h"

namespace AIS
{
	// process-wide registry of the GPU contexts: one per group of receivers with the same configuration
	class GpuPool
	{
	public:
		struct Group
		{
			aisgpu_cfg cfg;
			int reserved = 0;                 // receivers registered (buildModel) before the context exists
			int users = 0;                    // chains that still refer to this group
			aisamd::GpuBatch *batch = nullptr; // created by the first block
		};

	private:
		std::mutex mtx;
		std::vector<Group *> groups;
		bool pipelined = false, gpu_decode = false;
		int timeout_ms = 10000, block_bytes = 0;

	public:
		static GpuPool &instance();
		// register a receiver with configuration c (n_receivers / block_len / input_format are filled in when the context is
		// created); returns its group and its row in the group's batch
		Group *reserve(const aisgpu_cfg &c, int &rx);
		// the group's batch; the first caller creates the context for the block length / format of its block
		aisamd::GpuBatch *open(Group *g, int block_len, int input_format);
		void release(Group *g, int rx);
		// process-wide options, set before the models are built: what a `-go GPU_DECODE on` / `-go GPU_PIPELINE on` of the
		// patched Receiver would call.  gpu_decode: the AIS::Decoder state machines run on the device (AISGPU_FLAG_GPU_DECODE),
		// completed frames are finished by GpuEmitFrame(); pipelined: GpuBatch::setPipelined (messages come one block later,
		// GpuChain::Flush() after the last block).
		void setPipelined(bool b) { std::lock_guard<std::mutex> l(mtx); pipelined = b; }
		void setGpuDecode(bool b) { std::lock_guard<std::mutex> l(mtx); gpu_decode = b; }
		void setTimeout(int ms) { std::lock_guard<std::mutex> l(mtx); timeout_ms = ms; }
		bool gpuDecode() { std::lock_guard<std::mutex> l(mtx); return gpu_decode; }
		// bytes per GPU block (`-go GPU_BLOCK n` of the patched Receiver); 0 = by the device (GpuChain::Receive)
		void setBlockBytes(int n) { std::lock_guard<std::mutex> l(mtx); block_bytes = n; }
		int blockBytes() { std::lock_guard<std::mutex> l(mtx); return block_bytes; }
	};

	// The tail of AIS::Decoder::Run for a frame whose bits the GPU decoders collected (Marine/AIS.h:150-163: tag.level, end_idx,
	// processData -> CRC, validate, buildNMEA, Send).  It touches private members of AIS::Decoder: in the reference tree this is
	// the 12-line patch "add `friend void GpuEmitFrame(...)` to class Decoder" (INTEGRATION.md); the test build of this repository
	// compiles this one translation unit with -fno-access-control instead, so that the reference stays unmodified.
	void GpuEmitFrame(Decoder &d, const aisgpu_frame &f, TAG &tag);

	// StreamIn<RAW> in place of Util::ConvertRAW and everything behind it: the device's bytes go to the GPU as they are (CU8, CS8,
	// CS16 and CF32 are converted inside the front-end kernel), cut into blocks of the context's size whatever the size of the
	// device's calls (RAWFile hands over one OR two FIFO blocks per call, Device/FileRAW.cpp:120-136); the symbol decisions come back and are replayed into the
	// decoders in the reference's order (channel A's whole block first, DSP/DSP.cpp:312-313; per group the phases 0..4 with
	// tag.sample_idx / tag.sample_lvl / tag.ppm as ScatterPLL and the CGF set them, DSP/DSP.h:95-117, DSP/DSP.cpp:484).
	class GpuChain : public StreamIn<RAW>
	{
		GpuPool::Group *group = nullptr;
		aisamd::GpuBatch *batch = nullptr;
		int rx = 0;
		bool failed = false;
		// re-blocking (Receive): the context's block in bytes, bytes per IQ sample, the AISGPU_FMT_* of the stream, and the part of
		// the next block that earlier calls left behind
		int block_bytes = 0, sample_bytes = 0, format = 0;
		std::vector<char> carry;

		bool runBlock(const void *iq, TAG &tag);
		void deliver(TAG &tag);
		void replay(Connection<FLOAT32> *out, const aisgpu_out &o, TAG &tag, int n0, int n1);
		void replayChallenger(Connection<FLOAT32> *coh, Connection<FLOAT32> *fm, const aisgpu_out &o, TAG &tag, int n0, int n1);
		void replayFM(Connection<FLOAT32> &fm, const aisgpu_out &o, TAG &tag, int n0, int n1);

	public:
		Connection<FLOAT32> outA[N_SAMPLES_PER_SYMBOL], outB[N_SAMPLES_PER_SYMBOL];	  // what CD_EMA_a/b[i].out carry (Model.cpp:563-564)
		Connection<FLOAT32> outAf[N_SAMPLES_PER_SYMBOL], outBf[N_SAMPLES_PER_SYMBOL]; // ModelChallenger: S_af / S_bf .out[i] (Model.cpp:638-639), sign only
		Connection<FLOAT32> outFMa, outFMb;											  // ModelBase / ModelStandard: FR_a / FR_b .out (Model.cpp:431-432, 495-496), sign only
		Connection<CFLOAT32> outC48a, outC48b;										  // ModelEngineV2 with the engine on the host: FCIC5_a / FCIC5_b .out, the 48 kHz channels (Model.cpp:345-346, 452-453)
		// AISGPU_FLAG_GPU_DECODE: decoder of (channel, phase) -- phase 5..9: ModelChallenger's FM decoders; ModelEngineV2: 0..4 behind the
		// trackers, 5 the FM decoder -- set by the model
		Decoder *dec[2][2 * N_SAMPLES_PER_SYMBOL] = {};
		Type device_type = Type::NONE; // the device's driver (Device::getDriver), set by the model: picks the block size for the file reader

		virtual ~GpuChain();
		void join(const aisgpu_cfg &c) { group = GpuPool::instance().reserve(c, rx); }
		void Receive(const RAW *data, int len, TAG &tag);
		void Flush(TAG &tag); // pipelined batches: collect the last block's outputs when the device has delivered its last block
	};

	// AIS::ModelDefault with the DSP on the GPU ("-m 12").  Same keys as ModelDefault / ModelFrontend (Model.cpp:358-402, 579-594).
	class ModelDefaultGPU : public Model
	{
	protected:
		GpuChain chain;
		AIS::Decoder DEC_a[N_SAMPLES_PER_SYMBOL], DEC_b[N_SAMPLES_PER_SYMBOL];

		bool PS_EMA = true, CGF_wide = true, droop_compensation = true, fixedpointDS = false, allowDSK = false, MA_DS = false;

		void buildFrontend(int sample_rate, bool timerOn, Device::Device *dev, int model);

	public:
		ModelDefaultGPU() { setName("AIS engine v1 base (MI355X)"); }

		void buildModel(char, char, int, bool, Device::Device *);
		Setting &SetKey(AIS::Keys key, const std::string &arg);
		std::string Get();
		void Flush(TAG &tag) { chain.Flush(tag); }
	};

	// AIS::ModelChallenger with the DSP on the GPU ("-m 14"): the 20-decoder mesh of Model.cpp:641-674 on the host
	class ModelChallengerGPU : public ModelDefaultGPU
	{
		AIS::Decoder DEC_af[N_SAMPLES_PER_SYMBOL], DEC_bf[N_SAMPLES_PER_SYMBOL];

	public:
		ModelChallengerGPU() { setName("AIS engine v1 high (MI355X)"); }
		void buildModel(char, char, int, bool, Device::Device *);
	};

	// AIS::ModelEngineV2 (Model.cpp:440-463, "-m 31" here).  Two forms: by default the GPU runs the front end and hands
	// the two 48 kHz channels to the reference's own V2::Engine objects (chain.outC48a >> V2_a, exactly where *C_a >> V2_a sits in the
	// reference); with GpuPool::setGpuDecode(true) the whole engine runs on the device (kv2_engine) and only completed frames come back,
	// each to the tail of ITS decoder of the reference's engine object (V2_x.getDecoder(phase)).  The device engine has PhaseTracker's
	// default weights (V2Engine.h:70-71): other DD_TRAIN / DD_WEIGHT values are refused in that form.
	class ModelEngineV2GPU : public ModelDefaultGPU
	{
		V2::Engine V2_a, V2_b;
		float dd_train = 0.75f, dd_weight = 0.86f; // DSP/Model.h:272

	public:
		ModelEngineV2GPU() { setName("AIS engine v2 base (MI355X)"); }
		void buildModel(char, char, int, bool, Device::Device *);
		Setting &SetKey(AIS::Keys key, const std::string &arg);
		std::string Get();
	};

	// AIS::ModelStandard (Model.cpp:484-518): front end + FM receiver on the GPU, the reference's Deinterleave(5) + five decoders here
	class ModelStandardGPU : public ModelDefaultGPU
	{
		DSP::Deinterleave<FLOAT32> S_a, S_b;

	public:
		ModelStandardGPU() { setName("Standard (non-coherent) (MI355X)"); }
		void buildModel(char, char, int, bool, Device::Device *);
	};

	// AIS::ModelBase (Model.cpp:419-438): front end + FM receiver on the GPU, the reference's SimplePLL + decoder (with its feedback) here
	class ModelBaseGPU : public ModelDefaultGPU
	{
		DSP::SimplePLL sampler_a, sampler_b;

	public:
		ModelBaseGPU() { setName("Base (non-coherent) (MI355X)"); }
		void buildModel(char, char, int, bool, Device::Device *);
	};
}
