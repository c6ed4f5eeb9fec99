// ais-catcher_amd/host/v2_engine.h -- the host half of AIS::ModelEngineV2 ("-m 11", reference DSP/Model.cpp:440-463).
//
// V2::Engine (reference DSP/Decoder/V2/V2Engine.{h,cpp}) works on the 48 kHz channel in blocks of 512 samples, and what it
// does with a block depends on what its six decoders made of the previous one: whether a frame is in flight (the tone
// gate of the frequency estimate, the training weights of the phase trackers, the fast / slow bit PLL) and where the last
// good frame started (the learned SOTDMA slot phase decides where the next estimate is taken).  That feedback closes
// every 512 samples through the HDLC state machines, so this part of the model is sequential by construction; the device
// runs what is in front of it -- the decimation ladder, the data-parallel 97 % of the arithmetic -- and hands over the
// channel (aisgpu_out.c48).  Everything here is a behavioural mirror: same blocks, same operation order, the host's libm
// for cosf / sinf / atan2f / hypotf like the reference on the same machine.
#pragma once
#include <vector>

#include "ais_frame.h"
#include "stream.h"

namespace aisamd {

class V2Engine : public StreamIn<CFLOAT32> {
public:
	static const int BLOCK = 512;     // samples per processing block at 48 kHz (V2Engine.h:30)
	static const int N_DECODERS = 6;  // five strobe decoders behind the phase trackers + one behind the FM branch
	static const int FM_DEC = 5;

private:
	static const int SLOT = 1280; // one SOTDMA slot at 48 kHz
	static const int PRE = 155;   // start-flag anchor -> burst start

	// ---- frequency offset: FFT of the squared block, energy window, peak pair 9600 Hz apart, sub-bin interpolation
	struct Tone {
		CFLOAT32 rot = CFLOAT32(1.0f, 0.0f);
		float last_f = 0.0f, prominence = 0.0f;
		std::vector<CFLOAT32> omega, work;
		float mag[BLOCK];
		Tone();
		float estimate(const CFLOAT32* window);
		void derotate(float f, const CFLOAT32* src, CFLOAT32* dst, int len);
	} tone;
	// ---- symmetric FIRs with the centre tap in the middle of the block (16 / 36 samples of carry)
	CFLOAT32 carry17[32];
	float carry37[72];
	void fir17(const CFLOAT32* in, CFLOAT32* out);
	void fir37(const float* in, float* out);
	// ---- decision-directed phase tracker per strobe
	struct Tracker {
		unsigned rot = 0;
		CFLOAT32 s = CFLOAT32(0.0f, 0.0f);
		int prev_decision = 0;
		float weight = 0.86f, weight_train = 0.75f;
		int run(CFLOAT32 z, bool training);
	} trk[5];
	CFLOAT32 fm_prev = CFLOAT32(1.0f, 0.0f);
	struct BitClock {
		float phase = 0.0f;
		int last_bit = 0;
		bool run(float sample, bool training);
	} fm_clock;

	AIS::Decoder dec[N_DECODERS];

	CFLOAT32 raw[2 * BLOCK]; // [0, BLOCK): the block that is decoded now, [BLOCK, 2 BLOCK): look-ahead
	CFLOAT32 derot[BLOCK], coh[BLOCK];
	float disc[BLOCK], disc_f[BLOCK];
	CFLOAT32 slot_ema = CFLOAT32(0.0f, 0.0f);
	int slot_phase = 0, di = 0, fill = 0, ppm_split = 0;
	long long sample_idx = 0;
	float ppm = 0.0f, ppm_prev = 0.0f;

	// ---- what the device has already computed for the engine blocks of the data that is being received (aisgpu_out.v2_*):
	// the frequency estimates of the offset-0 / offset-256 windows, midWins' energies and the sign of the filtered discriminator
	const float *as_f = nullptr, *as_prom = nullptr, *as_en = nullptr;
	const uint32_t* as_fm = nullptr;
	int as_i = 0;              // engine block (within the device's block) that the next block() call decodes the look-back of
	uint32_t fm_tail[BLOCK / 32] = {}; // discriminator signs of the previous device block's last 512 samples

	bool laterHalfIsLouder(const CFLOAT32* in) const;
	void correctFrequency(const CFLOAT32* in, CFLOAT32* out, bool busy);
	void learnSlot(const AIS::Decoder& d);
	void resetAll();
	void block(TAG& tag);

public:
	V2Engine();
	void setWeights(float train, float track) {
		for (auto& t : trk) { t.weight_train = train; t.weight = track; }
	}
	void setOrigin(char channel, int station, int own_mmsi) {
		for (auto& d : dec) d.setOrigin(channel, station, own_mmsi);
	}
	AIS::Decoder& getDecoder(int i) { return dec[i]; }
	// Device assist for the NEXT Receive() call: the arrays of aisgpu_out (v2_f, v2_prom, v2_energy, fm_bits) of the device block the
	// data belongs to and the index of the data's first 512-sample engine block inside it.  nullptr switches back to host-only.
	void setAssist(const float* f, const float* prom, const float* energy, const uint32_t* fm_bits, int first_engine_block) {
		as_f = f; as_prom = prom; as_en = energy; as_fm = fm_bits; as_i = first_engine_block;
	}
	// after the last Receive() of a device block of L samples: keep the discriminator signs of its last 512 samples
	void finishAssist(int L) {
		if (as_fm) for (int w = 0; w < BLOCK / 32; w++) fm_tail[w] = as_fm[(L - BLOCK) / 32 + w];
		as_f = as_prom = as_en = nullptr; as_fm = nullptr;
	}
	void Receive(const CFLOAT32* data, int len, TAG& tag) override;
};

} // namespace aisamd
