"""N4: the host-side tail -- HDLC deframer and pdu_to_nmea -- of
libaisx.so against the oracle's restatement and against first principles.
CPU only."""
import numpy as np

import ais_amd
import oracle_py as orc
import synth


def _frame_bits(payload_bits):
    return synth.FLAG + synth.bit_stuff(list(payload_bits) + synth.crc16_hdlc(payload_bits)) + synth.FLAG


def test_pdu_to_nmea_first_principles_and_oracle():
    rng = np.random.default_rng(1)
    for length in (1, 11, 21, 42, 43, 60, 64):
        pdu = bytes(rng.integers(0, 256, length, dtype=np.uint8))
        s = ais_amd.pdu_to_nmea("B").msg_to_sentence(pdu)
        assert s == orc.pdu_to_nmea("B", pdu)
        # independent re-derivation (lib/pdu_to_nmea_impl.cc:63-124), including its padding
        # quirk: the last sextet is shifted left npad times AFTER its bits were placed
        # MSB-aligned, in a uint8_t, and the armouring compares as (signed) char
        bits = np.unpackbits(np.frombuffer(pdu, np.uint8))
        npad = (6 - bits.size % 6) % 6
        six = np.concatenate([bits, np.zeros(npad, np.uint8)]).reshape(-1, 6)
        vals = (six * (1 << np.arange(5, -1, -1))).sum(axis=1).astype(np.int64)
        if npad:
            vals[-1] = (int(vals[-1]) << npad) & 0xFF
        chars = []
        for v in vals:
            c = int(v) - 256 if v >= 128 else int(v)
            if c > 39:
                c += 8
            c += 48
            chars.append(chr(c & 0xFF))
        ascii_ = "".join(chars)
        frags = [ascii_[k:k + 56] for k in range(0, len(ascii_), 56)]
        want = []
        for i, f in enumerate(frags):
            body = "AIVDM,%d,%d,,B,%s,%d" % (len(frags), i + 1, f, npad)
            cs = 0
            for ch in body:
                cs ^= ord(ch)
            want.append("!%s*%02X" % (body, cs))
        assert s == "\n".join(want)


def test_hdlc_deframer_matches_oracle_and_recovers_payloads():
    rng = np.random.default_rng(2)
    bits, sent = [], []
    for k in range(40):
        bits += rng.integers(0, 2, int(rng.integers(0, 200))).tolist()
        nbytes = int(rng.choice([5, 11, 21, 40, 64, 70]))
        payload = rng.integers(0, 2, nbytes * 8).tolist()
        fb = _frame_bits(payload)
        if k % 7 == 3:
            fb[40] ^= 1  # corrupt: CRC must reject
        else:
            sent.append((nbytes, bytes(np.packbits(np.array(payload, np.uint8).reshape(-1, 8)[:, ::-1], axis=1).ravel())))
        bits += fb
    bits = np.array(bits, np.uint8)
    got = []
    d = ais_amd.hdlc_deframer_bp(11, 64)
    o = orc.Hdlc(11, 64)
    want = []
    for k in range(0, bits.size, 777):  # state carries across calls
        got += d.work(bits[k:k + 777])
        want += o.work(bits[k:k + 777])
    assert got == want
    # a frame passes when payload + 2 CRC bytes is within [length_min, length_max]
    for n, pl in sent:
        if 11 <= n + 2 <= 64:
            assert pl in got
        else:
            assert pl not in got
    assert len(got) >= 10


def test_end_to_end_decode_on_the_oracle_chain_bits():
    lv = [1 if b else -1 for b in synth.sync_bits("P")]
    tmpl = synth.gmsk_waveform(np.array(lv, float), 4)[: len(lv) * 4].astype(np.complex64)
    x, infos = synth.make_channel(77, 65536, "P", 4, amp=0.3)
    bits, _, _ = orc.Demod(4, tmpl, stages=3).step(x)
    pdus = ais_amd.hdlc_deframer_bp(11, 64).work(bits)
    assert pdus == orc.Hdlc(11, 64).work(bits)
    sent = set()
    for inf in infos:
        pb = np.array(inf["payload"], np.uint8).reshape(-1, 8)
        sent.add(bytes((pb * (1 << np.arange(8))).sum(axis=1).astype(np.uint8)))
    assert len(pdus) >= len(infos) // 2 and all(p in sent for p in pdus)
    nm = ais_amd.pdu_to_nmea("A")
    for p in pdus:
        s = nm.msg_to_sentence(p)
        assert s.startswith("!AIVDM,1,1,,A,") and s == orc.pdu_to_nmea("A", p)


def test_hdlc_rejects_frames_without_room_for_the_fcs_and_nmea_padding_quirk():
    import pytest

    # a frame is payload + two FCS octets: length_min < 2 would check a CRC over a negative length
    with pytest.raises(ValueError):
        ais_amd.hdlc_deframer_bp(1, 64)
    with pytest.raises(ValueError):
        ais_amd.hdlc_deframer_bp(20, 10)
    # lib/pdu_to_nmea_impl.cc:70-78: with 4 fill bits (len % 3 == 1) the last group loses its two
    # data bits to the second shift and always armours to '0'; with 2 fill bits a group >= 0x80
    # skips the +8 step (signed char compare)
    for last in (0x00, 0x01, 0x02, 0x03, 0xFF):
        s = ais_amd.pdu_to_nmea("A").msg_to_sentence(bytes([0x12, 0x34, 0x56, last]))
        body = s.split(",")[5]
        assert len(body) == 6 and body[-1] == "0" and s.split(",")[6].startswith("4*")
    s = ais_amd.pdu_to_nmea("A").msg_to_sentence(bytes([0xFF, 0xFF]))
    assert s.split(",")[5] == "ww" + chr((0xF0 - 256 + 48) & 0xFF) and s.split(",")[6].startswith("2*")


def test_framing_helpers_do_not_import_torch():
    """The host-only helpers (HDLC deframer, NMEA) load libaisx.so without importing torch on their account."""
    import os
    import subprocess
    import sys

    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gr-ais_amd")
    code = ("import sys; sys.path.insert(0, %r); import ais_amd; d = ais_amd.hdlc_deframer_bp(11, 64); "
            "s = ais_amd.pdu_to_nmea('A').msg_to_sentence(b'abc'); "
            "assert 'torch' not in sys.modules, 'torch imported'; print('ok', s)") % pkg
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-1500:]
