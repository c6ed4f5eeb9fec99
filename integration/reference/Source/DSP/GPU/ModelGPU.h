/*
 * Source/DSP/GPU/ModelGPU.h -- the file a maintainer adds to the REFERENCE tree (jvde-github/AIS-catcher v0.70) to run the
 * hot path of AIS::ModelDefault / AIS::ModelChallenger on an MI355X through the C ABI of libaisgpu.so (include/aisgpu.h).
 *
 * It is written against the reference's own headers -- Stream.h (StreamIn / Connection, Library/Stream.h:36-167), Model.h
 * (AIS::Model, DSP/Model.h:76-126), AIS.h (AIS::Decoder, Marine/AIS.h:38-191), Device.h -- and calls nothing but the C ABI.
 * oracle/Makefile (target refgpu) compiles it together with the unmodified reference sources; tests/test_gpu_parity.py checks
 * that engine 12 (this class) prints the same NMEA as engine 2 (AIS::ModelDefault) from the same binary.
 *
 * Replaced: everything between the device's RAW output and the AIS::Decoder objects (DSP/Model.cpp:27-356, 520-577, 601-678).
 * Kept:     the reference's Device, TAG, AIS::Decoder (with its Reset mesh), AIS::Message, outputs, Setting/SetKey.
 */
#pragma once

#include <string>

#include "Model.h"
#include "AIS.h"
#include "aisgpu.h"

namespace AIS
{
	// StreamIn<RAW> in place of Util::ConvertRAW and everything behind it: the device block goes to the GPU as it is (CU8, CS8,
	// CS16 and CF32 are converted inside the front-end kernel); the symbol decisions come back and are replayed into the
	// decoders in the reference's order (channel A's whole block first, DSP/DSP.cpp:312-313; per group the phases 0..4 with
	// tag.sample_idx / tag.sample_lvl / tag.ppm as ScatterPLL and the CGF set them, DSP/DSP.h:95-117, DSP/DSP.cpp:484).
	class GpuChain : public StreamIn<RAW>
	{
		aisgpu_t *ctx = nullptr;
		aisgpu_cfg cfg;
		bool failed = false;

		void open(const RAW *raw);
		void replay(Connection<FLOAT32> *out, const aisgpu_out &o, TAG &tag, int n0, int n1);
		void replayChallenger(Connection<FLOAT32> *coh, Connection<FLOAT32> *fm, const aisgpu_out &o, TAG &tag, int n0, int n1);

	public:
		Connection<FLOAT32> outA[N_SAMPLES_PER_SYMBOL], outB[N_SAMPLES_PER_SYMBOL];	  // what CD_EMA_a/b[i].out carry (Model.cpp:563-564)
		Connection<FLOAT32> outAf[N_SAMPLES_PER_SYMBOL], outBf[N_SAMPLES_PER_SYMBOL]; // ModelChallenger: S_af / S_bf .out[i] (Model.cpp:638-639), sign only

		GpuChain() { aisgpu_default_cfg(&cfg); }
		virtual ~GpuChain() { aisgpu_destroy(ctx); }

		aisgpu_cfg &config() { return cfg; } // filled by the model before the first block; the context is created with the first block's length
		void Receive(const RAW *data, int len, TAG &tag);
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
	};

	// AIS::ModelChallenger with the DSP on the GPU ("-m 14"): the 20-decoder mesh of Model.cpp:641-674 on the host
	class ModelChallengerGPU : public ModelDefaultGPU
	{
		AIS::Decoder DEC_af[N_SAMPLES_PER_SYMBOL], DEC_bf[N_SAMPLES_PER_SYMBOL];

	public:
		ModelChallengerGPU() { setName("AIS engine v1 high (MI355X)"); }
		void buildModel(char, char, int, bool, Device::Device *);
	};
}
