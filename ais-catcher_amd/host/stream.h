// ais-catcher_amd/host/stream.h -- the reference's block API, restated for the host side of the GPU chain.
//
// Same names, argument meaning and call conventions as the reference's Library/Stream.h:36-167 and
// Library/Common.h:77-87,240-295, so that GpuChain / ModelDefaultGPU read like reference blocks and can
// be moved into the reference tree by replacing this header with `#include "Stream.h"`.
// Only the members the hot path touches are kept.
#pragma once
#include <complex>
#include <cstdint>
#include <string>
#include <vector>

typedef float FLOAT32;
typedef std::complex<FLOAT32> CFLOAT32;
typedef std::complex<uint8_t> CU8;
typedef std::complex<int8_t> CS8;    // Library/Common.h
typedef std::complex<int16_t> CS16;
typedef char BIT;

enum class Format { CU8, CF32, CS8, CS16, UNKNOWN };

// side-band struct that travels with every Receive() (Library/Common.h:240-288)
struct TAG {
	unsigned mode = 3;       // bit 0: signal level, bit 1: timestamp
	float sample_lvl = 0;    // written by the chain per 5-sample group (DSP/DSP.h:105-106)
	float level = 0;         // written by the decoder per message (Marine/AIS.h:147-156)
	float ppm = 0;           // written by the chain per CGF window (DSP/DSP.cpp:484)
	long long sample_idx = 0;
};

struct RAW {
	Format format;
	void* data;
	int size; // bytes
};

template <typename T>
class StreamIn {
public:
	virtual ~StreamIn() {}
	// `data` is borrowed for the duration of the call only; len counts T, not bytes
	virtual void Receive(const T* data, int len, TAG& tag) {}
};

template <typename S>
class Connection {
	std::vector<StreamIn<S>*> connections;

public:
	void Send(const S* data, int len, TAG& tag) {
		for (auto c : connections) c->Receive(data, len, tag);
	}
	void Connect(StreamIn<S>* s) { connections.push_back(s); }
	bool isConnected() const { return !connections.empty(); }
};

template <typename S>
class StreamOut {
public:
	Connection<S> out;
	void Send(const S* data, int len, TAG& tag) { out.Send(data, len, tag); }
};

template <typename T, typename S>
class SimpleStreamInOut : public StreamOut<S>, public StreamIn<T> {};

template <typename S>
inline StreamIn<S>& operator>>(Connection<S>& a, StreamIn<S>& b) {
	a.Connect(&b);
	return b;
}
template <typename S>
inline StreamIn<S>& operator>>(StreamOut<S>& a, StreamIn<S>& b) {
	a.out.Connect(&b);
	return b;
}

// decoder-to-decoder signalling (Library/Signals.h:23-51)
enum class DecoderSignals { StopTraining, StartTraining, Reset };

template <typename T>
class SignalIn {
public:
	virtual ~SignalIn() {}
	virtual void Signal(const T&) {}
};

template <typename T>
class SignalHub {
	std::vector<SignalIn<T>*> destinations;

public:
	void Send(const T& m) {
		for (auto d : destinations) d->Signal(m);
	}
	void Connect(SignalIn<T>& s) { destinations.push_back(&s); }
};
