// ais-catcher_amd/host/ais_frame.h -- HDLC frame decoder + NMEA armouring that consumes the GPU chain's
// hard bits on the host.  Behavioural mirror of the reference's AIS::Decoder (Marine/AIS.h:38-191,
// Marine/AIS.cpp:33-142) and of the parts of AIS::Message it needs (Marine/Message.h:36-41,171-183,
// 264-281; Marine/Message.cpp:398-413,569-686): same states, same acceptance rules, same text.
// In an integrated build the reference's own classes are used instead; this restatement
// exists so the repository is stand-alone and testable.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "stream.h"

namespace AIS {

constexpr int MAX_AIS_LENGTH = 1064;                     // payload bits
constexpr int MAX_AIS_FRAME_LENGTH = MAX_AIS_LENGTH + 16 + 7; // + FCS + closing flag capture
constexpr int MAX_AIS_FRAME_BYTES = (MAX_AIS_FRAME_LENGTH + 7) / 8;

class Message {
	uint8_t data[MAX_AIS_FRAME_BYTES + 4];
	int length = 0;
	char channel = '?';
	int own_mmsi = -1;
	long long start_idx = 0, end_idx = 0;
	std::vector<std::string> NMEA;
	static std::atomic<int> ID; // multi-sentence sequence id, process global like the reference's

public:
	Message() { clear(); }
	void clear() {
		length = 0;
		start_idx = end_idx = 0;
		NMEA.clear();
		std::memset(data, 0, sizeof data);
	}
	void setBit(int i, bool b) {
		if (i < 0 || i >= MAX_AIS_FRAME_LENGTH) return;
		const uint8_t m = (uint8_t)(1u << (i & 7)); // bytes arrive LSB first on air
		if (b) data[i >> 3] |= m; else data[i >> 3] &= (uint8_t)~m;
	}
	bool getBit(int i) const { return i >= 0 && i < MAX_AIS_FRAME_LENGTH && ((data[i >> 3] >> (i & 7)) & 1); }
	unsigned type() const { return data[0] >> 2; }
	unsigned mmsi() const { return ((unsigned)data[1] << 22) | ((unsigned)data[2] << 14) | ((unsigned)data[3] << 6) | (data[4] >> 2); }
	void setLength(int l) { if (l >= 0 && l <= MAX_AIS_LENGTH) length = l; }
	int getLength() const { return length; }
	void setOrigin(char c, int own) { channel = c; own_mmsi = own; }
	char getChannel() const { return channel; }
	void setStartIdx(long long s) { start_idx = s; }
	void setEndIdx(long long e) { end_idx = e; }
	long long getStartIdx() const { return start_idx; }
	long long getEndIdx() const { return end_idx; }
	bool validate() const;
	void buildNMEA();
	const std::vector<std::string>& sentences() const { return NMEA; }
	static void resetSequence() { ID.store(0); }
};

enum class State { TRAINING, STARTFLAG, STOPFLAG, DATAFCS, FOUNDMESSAGE };

class Decoder : public SimpleStreamInOut<FLOAT32, Message>, public SignalIn<DecoderSignals> {
	char channel = '?';
	int own_mmsi = -1;
	State state = State::TRAINING;
	BIT lastBit = 0, prev = 0;
	int position = 0, one_seq_count = 0;
	FLOAT32 level = 0.0f;
	long long start_idx = 0, end_idx = 0;
	Message msg;

	void NextState(State s, int pos);
	bool CRC16(int len) const;
	bool processData(int len, TAG& tag);
	bool cannotBeValid(int len) const;
	bool step(FLOAT32 sample, TAG& tag); // true: this bit completed a frame with a good CRC

public:
	// Per-bit entry (Marine/AIS.h:88-181): the V2 engine calls it directly.  FOUNDMESSAGE on the completing bit, else the resting state.
	State Run(FLOAT32 sample, TAG& tag) { return step(sample, tag) ? State::FOUNDMESSAGE : state; }
	void reset() { NextState(State::TRAINING, 0); }
	State getState() const { return state; }
	long long getStartIdx() const { return start_idx; }
	void setOrigin(char c, int /*station*/, int own) { channel = c; own_mmsi = own; }
	void Receive(const FLOAT32* data, int len, TAG& tag) override {
		for (int i = 0; i < len; i++) step(data[i], tag);
	}
	void Signal(const DecoderSignals& in) override {
		if (in == DecoderSignals::Reset) NextState(State::TRAINING, 0);
	}
	// A frame whose state machine ran on the GPU (aisgpu_frames): the part of Run()/processData() that follows the closing
	// flag -- tag.level, CRC (checked again), length, validate, NMEA, Send (Marine/AIS.h:150-160, Marine/AIS.cpp:66-96)
	bool emitFrame(const uint8_t* bits_as_received, int position, FLOAT32 level_sum, long long start, long long end, TAG& tag);
	SignalHub<DecoderSignals> DecoderMessage;
};

} // namespace AIS
