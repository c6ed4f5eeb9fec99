"""CPU: the synthetic GMSK generator builds frames the (oracle) decoder accepts."""
import numpy as np

import checkers
from ais_catcher_amd import synth


def test_crc_and_stuffing():
    bits = synth.frame_bits(synth.PAYLOADS[0])
    body = bits[8 + 24 + 8:-16]
    run = 0
    for b in body:  # never six ones inside the stuffed body
        run = run + 1 if b else 0
        assert run < 6
    tx = synth.dearmour(synth.PAYLOADS[0]).reshape(-1, 8)[:, ::-1].reshape(-1)
    assert synth.crc16_x25(tx) == synth.crc16_x25(list(tx))


def test_oracle_decodes_every_scheduled_burst():
    x, sched = synth.receiver_stream(786432 * 3, receiver_id=21, return_schedule=True)
    o = checkers.Oracle()
    o.feed_blocks(x, 786432)
    assert sorted(o.nmea()) == sorted(synth.expected_nmea(sched)) and len(sched) > 10


def test_cu8_quantisation_is_reference_convention():
    x = np.array([0 + 0j, 0.5 - 0.25j, 2 + -2j], np.complex64)
    assert synth.to_cu8(x).tolist() == [128, 128, 192, 96, 255, 0]
