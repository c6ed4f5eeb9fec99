"""ais-catcher_amd: MI355X-native AIS GMSK demodulation chain (the ModelDefault hot path)."""
