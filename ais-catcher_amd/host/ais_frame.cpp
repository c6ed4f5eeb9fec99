// ais-catcher_amd/host/ais_frame.cpp -- see ais_frame.h
#include "ais_frame.h"

#include <cmath>

namespace AIS {

std::atomic<int> Message::ID{0};

// minimum payload length per message type (Marine/Message.cpp:398-413)
bool Message::validate() const {
	static const int min_len[28] = { 149, 149, 149, 168, 418, 88, 72, 56, 168, 70, 168, 72, 40, 40,
	                                 88, 92, 80, 168, 312, 70, 271, 145, 154, 160, 72, 60, 96, 168 };
	if (length == 0) return true;
	if (length > MAX_AIS_LENGTH) return false;
	const unsigned t = type();
	return t >= 1 && t <= 28 && length >= min_len[t - 1];
}

// "!AIVDM,<n>,<k>,<seq>,<channel>,<payload<=56>,<fill>*<xor>" (Marine/Message.cpp:569-631)
void Message::buildNMEA() {
	static const char armour[65] = "0123456789:;<=>?@ABCDEFGHIJKLMNOPQRSTUVW`abcdefghijklmnopqrstuvw";
	static const char hex[17] = "0123456789ABCDEF";
	const int letters_total = (length + 5) / 6;
	const int n_sent = letters_total == 0 ? 1 : (letters_total + 55) / 56;
	char seq = 0;
	if (n_sent > 1) {
		int cur = ID.load(std::memory_order_relaxed);
		while (!ID.compare_exchange_weak(cur, (cur + 1) % 10, std::memory_order_relaxed)) {}
		seq = (char)('0' + cur);
	}
	NMEA.clear();
	int done = 0;
	for (int s = 0; s < n_sent; s++) {
		std::string line = (own_mmsi == (int)mmsi()) ? "!AIVDO," : "!AIVDM,";
		line += (char)('0' + n_sent);
		line += ',';
		line += (char)('1' + s);
		line += ',';
		if (seq) line += seq;
		line += ',';
		if (channel != '?') line += channel;
		line += ',';
		const int take = letters_total - done < 56 ? letters_total - done : 56;
		for (int k = 0; k < take; k++) {
			const int start = (done + k) * 6, end = start + 6;
			char c = 0;
			if (end <= MAX_AIS_LENGTH) {
				const unsigned w = ((unsigned)data[start >> 3] << 8) | data[(start >> 3) + 1];
				int v = (w >> (10 - (start & 7))) & 0x3F;
				if (end > length) v &= 0x3F << (end - length); // bits beyond the payload read as zero
				c = armour[v];
			}
			line += c;
		}
		done += take;
		line += ',';
		line += (char)('0' + (s == n_sent - 1 ? letters_total * 6 - length : 0));
		unsigned x = 0;
		for (size_t k = 1; k < line.size(); k++) x ^= (unsigned char)line[k];
		line += '*';
		line += hex[(x >> 4) & 15];
		line += hex[x & 15];
		NMEA.push_back(line);
	}
}

void Decoder::NextState(State s, int pos) { // Marine/AIS.cpp:33-53
	state = s;
	position = pos;
	one_seq_count = 0;
	if (s == State::TRAINING) DecoderMessage.Send(DecoderSignals::StartTraining);
	else if (s == State::STARTFLAG) DecoderMessage.Send(DecoderSignals::StopTraining);
	else if (s == State::FOUNDMESSAGE) DecoderMessage.Send(DecoderSignals::Reset);
}

bool Decoder::CRC16(int len) const { // CRC-16/X.25 residue check over the received order (AIS.cpp:55-64)
	uint16_t crc = 0xFFFF;
	for (int i = 0; i < len; i++) crc = (((uint16_t)msg.getBit(i) ^ crc) & 1) ? (uint16_t)((crc >> 1) ^ 0x8408) : (uint16_t)(crc >> 1);
	return crc == (uint16_t)~0x0F47;
}

bool Decoder::processData(int len, TAG& tag) { // AIS.cpp:66-96
	if (len < 16 || !CRC16(len)) return false;
	if ((tag.mode & 1) && tag.level != 0.0) tag.level = 10.0f * log10(tag.level);
	msg.setOrigin(channel, own_mmsi);
	msg.setLength(len - 16);
	msg.setStartIdx(start_idx);
	msg.setEndIdx(end_idx);
	if (msg.validate()) {
		msg.buildNMEA();
		Send(&msg, 1, tag);
	}
	return true;
}

bool Decoder::emitFrame(const uint8_t* bits, int pos, FLOAT32 level_sum, long long start, long long end, TAG& tag) {
	if (pos < 7 || pos > MAX_AIS_FRAME_LENGTH) return false;
	msg.clear();
	for (int i = 0; i < pos; i++) msg.setBit(i, (bits[i >> 3] >> (i & 7)) & 1);
	if (tag.mode & 1) tag.level = level_sum / pos;
	start_idx = start;
	end_idx = end;
	return processData(pos - 7, tag);
}

bool Decoder::cannotBeValid(int len) const { // early-abort heuristics (AIS.cpp:111-142)
	const int END = 24;
	if (len < 6 + END) return false;
	const int t = (int)msg.type();
	switch (len) {
	case 6 + END: return t > 28 || t == 0;
	case 8 + 30 + END: return msg.mmsi() > 999999999;
	case 72 + END: return t == 10;
	case 144 + END: return t == 16;
	case 160 + END: return t == 15 || t == 20 || t == 23;
	case 168 + END: return t == 1 || t == 2 || t == 3 || t == 4 || t == 7 || t == 9 || t == 11 || t == 18 || t == 22 || t == 24 || t == 25 || t == 27 || t == 28;
	case 312 + END: return t == 19;
	case 361 + END: return t == 21;
	case 424 + END: return t == 5;
	}
	return false;
}

bool Decoder::step(FLOAT32 sample, TAG& tag) { // Marine/AIS.h:91-181
	bool found = false;
	const BIT d = sample > 0;
	const BIT Bit = !(d ^ prev); // NRZI
	prev = d;
	switch (state) {
	case State::TRAINING:
		if (Bit != lastBit) position++;
		else if (position > 4) {
			start_idx = tag.sample_idx;
			NextState(State::STARTFLAG, Bit ? 3 : 1);
		} else NextState(State::TRAINING, 0);
		break;
	case State::STARTFLAG:
		if (position == 7) {
			if (Bit == 0) {
				NextState(State::DATAFCS, 0);
				level = 0.0f;
				msg.clear();
			} else NextState(State::TRAINING, 0);
		} else if (Bit == 1) position++;
		else NextState(State::TRAINING, 0);
		break;
	case State::DATAFCS:
		msg.setBit(position++, Bit);
		if (tag.mode & 1) level += tag.sample_lvl;
		if (Bit == 1) {
			if (one_seq_count == 5) { // six ones: closing flag (or abort)
				if (tag.mode & 1) tag.level = level / position;
				end_idx = tag.sample_idx;
				found = processData(position - 7, tag);
				if (found) NextState(State::FOUNDMESSAGE, 0);
				NextState(State::TRAINING, 0);
			} else one_seq_count++;
		} else {
			if (one_seq_count == 5) position--; // bit de-stuffing
			one_seq_count = 0;
		}
		if (position == MAX_AIS_FRAME_LENGTH || cannotBeValid(position)) NextState(State::TRAINING, 0);
		break;
	default: break;
	}
	lastBit = Bit;
	return found;
}

} // namespace AIS
